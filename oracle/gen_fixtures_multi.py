"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g2_multi_*.npz by RUNNING THE REFERENCE
(imported from /root/reference, build container only): OE.multiple_reflect
(oes/reflect.py:165-264) with its isMulti bracketing (oes/base.py:1279-1289) and the
derivOrder = 1 form of find_dz (base.py:842-845). The cases are in tests/multi_cases.py. While
generating, oracle/reflect_np.py:oe_multiple_reflect is asserted against the reference's beams:
every field of gb and of lbN (all footprints), nRefl, theta, the elevation fields, and the
brackets the reference hands to each of its find_intersection calls (tangency and hit).

  g2_multi_cylinder    the reference's example geometry (Cylinder.py:96), Au, 2048 rays
  g2_multi_toroid      toroid at 3 mrad, needElevationMap=True, 2048 rays
  g2_multi_edges       optical limits, roll / yaw, dead and 'out' incoming rays, rays that miss
                       or leave over the end, maxReflections = 3 (the loop is cut short)
  g2_multi_flat        flat mirror: one bounce, then nothing
  g2_multi_capillary   EllipsoidCapillaryMirror (parametric, closed surface of revolution) lit
                       off its focus: up to four bounces down the bore, lb.s / phi / r, the
                       elevation map through param_to_xyz

Run:  python -m oracle.gen_fixtures_multi
"""
import os
import sys

import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1
from . import reflect_np as rn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                'tests'))

EXTRA = ('nRefl', 'theta', 'elevationD', 'elevationX', 'elevationY', 'elevationZ', 's', 'phi',
         'r')


def multi_dict(prefix, beam):
    d = g1.beam_dict(prefix, beam)
    for name in EXTRA:
        if hasattr(beam, name):
            d[prefix + name] = np.array(getattr(beam, name))
    return d


def assert_multi(tag, mine, ref):
    g1.assert_beams(tag, mine, ref)
    for name in EXTRA:
        assert hasattr(mine, name) == hasattr(ref, name), (tag, name)
        if not hasattr(ref, name):
            continue
        m, r = getattr(mine, name), getattr(ref, name)
        if name == 'nRefl':
            assert np.array_equal(m, r), (tag, name)
        else:
            assert np.allclose(m, r, rtol=1e-13, atol=1e-15), (tag, name, np.abs(m - r).max())


def run_multi(tag, oe, params, beam, maxReflections=1000, needElevationMap=False, **extra):
    """The reference's multiple_reflect on *beam*, the oracle's on the same rays, compared;
    -> the golden file. The brackets of every find_intersection call are spied on."""
    calls = []
    orig = oe.find_intersection

    def find_spy(local_f, t1, t2, *a, **k):
        calls.append((np.array(t1), np.array(t2), k.get('derivOrder', a[7] if len(a) > 7 else 0)))
        return orig(local_f, t1, t2, *a, **k)
    oe.find_intersection = find_spy
    gb, lbN = oe.multiple_reflect(beam, maxReflections=maxReflections,
                                  needElevationMap=needElevationMap)
    oe.find_intersection = orig
    info = []
    mgb, mlbN = rn.oe_multiple_reflect(params, g1.to_oracle_beam(beam), maxReflections,
                                       needElevationMap, info=info)
    assert_multi(tag + ':gb', mgb, gb)
    assert_multi(tag + ':lbN', mlbN, lbN)
    # the brackets: bounce 0 makes one call, every later bounce two (tangency, then hit)
    k = 0
    brent, numit = [], []
    for bounce, one in enumerate(info):
        if bounce > 0:
            t1, t2, order = calls[k]
            assert order == 1 and np.all(t1 == 0), (tag, bounce)
            k += 1
            brent.append(one['tangency']['brent'])
            numit.append(one['tangency']['numit'])
        t1, t2, order = calls[k]
        k += 1
        assert order == 0
        good = one['tMin'] != 0 if bounce == 0 else None
        if good is not None and good.sum() == len(t1):
            assert np.array_equal(one['tMin'][good], t1), (tag, bounce)
        brent.append(one['brent'])
        numit.append(one['numit'])
    assert k == len(calls), (tag, k, len(calls))
    n = len(beam.x)
    nb = len(lbN.x) // n
    st, cnt = np.unique(gb.state, return_counts=True)
    print(tag, 'bounces', nb, 'nRefl', np.bincount(gb.nRefl).tolist(), 'gb states',
          dict(zip(st.tolist(), cnt.tolist())), 'brent', brent, 'numit', numit)
    out = {}
    out.update(g1.beam_dict('in_', beam))
    out.update(multi_dict('gb_', gb))
    out.update(multi_dict('lbN_', lbN))
    out.update(g1.flat_params(params))
    out.update(maxReflections=np.array(maxReflections),
               needElevationMap=np.array(int(needElevationMap)), bounces=np.array(nb),
               brent=np.array(brent, dtype=int), numit=np.array(numit))
    out.update(extra)
    g1.save(tag, **out)


def main(only=None):
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    import multi_cases as case
    from .fixture_io import tables as load_tables
    raycing._VERBOSITY_ = 0
    tb = load_tables()

    # the reference's example: whispering-gallery bounces along a cylinder
    bl = raycing.BeamLine(height=0)
    au = rm.Material('Au', rho=19.3, kind='mirror')
    cyl = case.cylinder_subclass(roe)(bl, 'Cylinder', material=au, **case.CYL)
    beam = case.point_source_rays(rs, 2048, 71)
    par = g1.oe_params(cyl, dict(kind='user', z=case.numpy_cyl_z, n=case.numpy_cyl_n))
    par['material'] = g1.material_dict(tb, au)
    run_multi('g2_multi_cylinder', cyl, par, beam, maxReflections=100,
              surface_parameters=np.array([case.CYL_RM]), mat_rho=np.array(19.3))

    # toroid, with the elevation map
    bl = raycing.BeamLine(height=0)
    tor = roe.ToroidMirror(bl, 'toroid', material=au, **case.TOROID)
    beam = case.point_source_rays(rs, 2048, 73)
    par = g1.oe_params(tor, dict(kind='toroid', R=case.TOROID['R'], r=case.TOROID['r']))
    par['material'] = g1.material_dict(tb, au)
    run_multi('g2_multi_toroid', tor, par, beam, maxReflections=100, needElevationMap=True,
              surf_R=np.array(case.TOROID['R']), surf_r=np.array(case.TOROID['r']),
              mat_rho=np.array(19.3))

    # edges: optical limits, dead rays, misses, the loop cut at three bounces
    bl = raycing.BeamLine(azimuth=0.3, height=0)
    pt = rm.Material('Pt', rho=21.45, kind='mirror')
    dummy = roe.OE(bl, 'first')                                    # lostNum = -2 - 1
    kw = case.edges_on(bl)
    edge = roe.ToroidMirror(bl, 'edges', material=pt, **kw)
    assert dummy is not None
    beam = case.edge_rays(rs, 2048, 79)
    # the element sits on a beamline with an azimuth: bring the rays into its frame
    raycing.virgin_local_to_global(bl, beam, None)
    par = g1.oe_params(edge, dict(kind='toroid', R=kw['R'], r=kw['r']))
    par['material'] = g1.material_dict(tb, pt)
    run_multi('g2_multi_edges', edge, par, beam, maxReflections=3, needElevationMap=True,
              surf_R=np.array(kw['R']), surf_r=np.array(kw['r']), mat_rho=np.array(21.45))

    # flat: one bounce and out
    bl = raycing.BeamLine(height=0)
    flat = roe.OE(bl, 'flat', material=au, **case.FLAT)
    beam = case.point_source_rays(rs, 1024, 83, dzprime=4e-5, amplitudes=False)
    par = g1.oe_params(flat, dict(kind='flat'))
    par['material'] = g1.material_dict(tb, au)
    run_multi('g2_multi_flat', flat, par, beam, maxReflections=10, mat_rho=np.array(19.3))

    # a capillary: the parametric branches (s, phi, r), ray . normal in them
    bl = raycing.BeamLine()
    cap = roe.EllipsoidCapillaryMirror(bl, 'cap', material=au, **case.CAPILLARY)
    beam = case.point_source_rays(rs, 1024, 89, dxprime=3e-3, dzprime=3e-3, E=9000.)
    par = g1.oe_params(cap, dict(kind='ellipse_capillary', ellipseA=cap.ellipseA,
                                 ellipseB=cap.ellipseB, ctd=cap.ctd))
    par['material'] = g1.material_dict(tb, au)
    run_multi('g2_multi_capillary', cap, par, beam, maxReflections=20, needElevationMap=True,
              mat_rho=np.array(19.3), cap_ctd=np.array(cap.ctd))


if __name__ == '__main__':
    main()
