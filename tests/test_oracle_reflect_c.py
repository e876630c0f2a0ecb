"""CPU: the C/OpenMP restatement of OE.reflect (oracle/reflect_c.c, bench.py's all-cores
baseline) against the numpy restatement that is pinned to the reference's golden vectors
(oracle/reflect_np.py) -- on the golden input of G2 and on cfg2's synthetic rays."""
import numpy as np
import pytest

from oracle import fixture_io, reflect_c as rc, reflect_np as rn


def _compare(mine, ref):
    assert np.array_equal(mine.state, ref.state)
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E'):
        r = getattr(ref, f)
        assert np.abs(getattr(mine, f) - r).max() <= 1e-12 * max(np.abs(r).max(), 1e-300), f
    scale = max(np.abs(ref.Jss).max(), np.abs(ref.Jpp).max(), 1e-300)
    for f in ('Jss', 'Jpp', 'Jsp'):
        assert np.abs(getattr(mine, f) - getattr(ref, f)).max() <= 1e-9 * scale, f


def _cfg2_params():
    tb = fixture_io.tables()
    from oracle import materials_np as mn
    p, q, pitch = 20000., 10000., 4e-3
    return dict(
        center=[0., p, 0.], azimuth_sc=(0., 1.), pitch=pitch, roll=0., yaw=0.,
        positionRoll=0., rotationSequence='RzRyRx', dx=0, shape='rect', overEdge='yMax',
        lostNum=-1, surfPhysX=[-10., 10.], surfPhysY=[-300., 300.], surfOptX=None,
        surfOptY=None,
        surface=dict(kind='toroid', R=2*p*q/(p+q)/np.sin(pitch), r=2*p*q/(p+q)*np.sin(pitch)),
        material=mn.make_material([mn.load_element(tb, 'Pt')], kind='mirror', rho=21.45))


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    b = rn.Beam(n)
    b.x, b.z, b.y = rng.normal(0, 0.1, n), rng.normal(0, 0.1, n), np.zeros(n)
    b.a, b.c = rng.normal(0, 2e-4, n), rng.normal(0, 2e-5, n)
    b.b = np.sqrt(1 - b.a**2 - b.c**2)
    b.E = rng.uniform(8990., 9010., n)
    b.state = np.ones(n, dtype=np.int32)
    b.Jss, b.Jpp = rng.uniform(0.5, 1., n), rng.uniform(0., 0.5, n)
    b.Jsp = 0.2 * (rng.normal(size=n) + 1j * rng.normal(size=n))
    return b


def test_c_restatement_equals_numpy_oracle_on_cfg2_rays():
    params = _cfg2_params()
    beam = _rays(60000, 42)
    beam.state[::97] = -3                 # rays that do not enter
    beam.state[5::211] = 2
    gb, lb = rc.oe_reflect(params, beam)
    ogb, olb = rn.oe_reflect(params, beam.copy())
    _compare(gb, ogb)
    _compare(lb, olb)
    assert np.abs(lb.theta - olb.theta).max() < 1e-14
    hit = (olb.state == 1).mean()
    assert 0.9 < hit < 0.99 and (olb.state == 3).any() and (olb.state == -1).any()


def test_c_restatement_on_the_golden_input():
    """G2's toroid + Pt case: the reference's own output (golden file), not only the
    numpy restatement of it. The golden beam carries amplitudes; the C code restates
    the intensity path, so Es / Ep are left out of the comparison."""
    params, beam, g = fixture_io.load_case('g2_toroid_pt')
    gb, lb = rc.oe_reflect(params, beam)
    for mine, tag in ((gb, 'gb_'), (lb, 'lb_')):
        assert np.array_equal(mine.state, g[tag + 'state'])
        for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path'):
            r = g[tag + f]
            assert np.abs(getattr(mine, f) - r).max() <= 1e-12 * np.abs(r).max(), (tag, f)
        scale = max(np.abs(g[tag + 'Jss']).max(), np.abs(g[tag + 'Jpp']).max())
        for f in ('Jss', 'Jpp', 'Jsp'):
            assert np.abs(getattr(mine, f) - g[tag + f]).max() <= 1e-9 * scale, (tag, f)
    assert np.abs(lb.theta - g['lb_theta']).max() < 1e-14


def test_a_batch_that_needs_brent_is_refused():
    params = _cfg2_params()
    beam = _rays(2000, 3)
    beam.c = np.random.default_rng(3).normal(0, 3e-2, 2000)
    beam.b = np.sqrt(1 - beam.a**2 - beam.c**2)
    info = {}
    rn.oe_reflect(params, beam.copy(), info=info)
    if info.get('brent'):
        with pytest.raises(NotImplementedError):
            rc.oe_reflect(params, beam)
