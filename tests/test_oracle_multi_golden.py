"""CPU: oracle/reflect_np.py:oe_multiple_reflect (OE.multiple_reflect, oes/reflect.py:165-264,
with the isMulti bracketing of oes/base.py:1279-1289 and find_dz(derivOrder=1), :842-845)
against the goldens the imported reference produced (oracle/gen_fixtures_multi.py)."""
import numpy as np
import pytest

from oracle import fixture_io, reflect_np as rn

CASES = ['g2_multi_cylinder', 'g2_multi_toroid', 'g2_multi_edges', 'g2_multi_flat',
         'g2_multi_capillary']
EXTRA = ('theta', 'elevationD', 'elevationX', 'elevationY', 'elevationZ', 's', 'phi', 'r')


def check(mine, g, prefix):
    for f in mine.fields():
        m, r = getattr(mine, f), g[prefix + f]
        if f == 'state':
            assert np.array_equal(m, r), (prefix, f)
        else:
            assert np.abs(m - r).max() <= 1e-13 * max(np.abs(r).max(), 1e-300), (prefix, f)
    assert np.array_equal(mine.nRefl, g[prefix + 'nRefl'])
    for f in EXTRA:
        assert hasattr(mine, f) == (prefix + f in g.files), (prefix, f)
        if hasattr(mine, f):
            assert np.allclose(getattr(mine, f), g[prefix + f], rtol=1e-13, atol=1e-15), f


@pytest.mark.parametrize('name', CASES)
def test_multiple_reflect_matches_reference(name):
    p, beam, g = fixture_io.load_case(name)
    info = []
    gb, lbN = rn.oe_multiple_reflect(p, beam, int(g['maxReflections']),
                                     bool(g['needElevationMap']), info=info)
    assert len(lbN.x) == int(g['bounces']) * len(beam.x)
    check(gb, g, 'gb_')
    check(lbN, g, 'lbN_')
    # the root-finding method and iteration count of every find_intersection call
    brent, numit = [], []
    for k, one in enumerate(info):
        if k:
            brent.append(one['tangency']['brent'])
            numit.append(one['tangency']['numit'])
        brent.append(one['brent'])
        numit.append(one['numit'])
    assert brent == [bool(b) for b in g['brent']]
    assert numit == g['numit'].tolist()


def test_no_ray_enters():
    p, beam, _ = fixture_io.load_case('g2_multi_flat')
    beam.state[:] = -1
    gb, lbN = rn.oe_multiple_reflect(p, beam)
    assert lbN is gb and np.array_equal(gb.x, beam.x) and not hasattr(gb, 'nRefl')
