"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g2_polygon.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only): a flat Pt mirror whose
optical surface is a POLYGON (``shape`` = list of (x, y) vertices, oes/base.py:1156-1160,
matplotlib's Path.contains_points decides), plus the reference's ``rays_good`` on a list of
hand-made points that sit on vertices, on edges and on the horizontals through vertices.
While generating, the numpy restatement (oracle/reflect_np.py) is asserted against the
reference on the same inputs.

Run:  python -m oracle.gen_fixtures_polygon
"""
import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1

# a non-convex, integer-cornered outline in the mirror's (x, y) plane [mm]
POLYGON = [(-6., -120.), (6., -120.), (8., 0.), (3., 40.), (6., 140.), (-2., 100.),
           (-8., 140.), (-5., 20.)]


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    tables = dict(np.load(g1.os.path.join(g1.OUT, 'g6_element_tables.npz')))
    bl = raycing.BeamLine(azimuth=0.1)
    mPt = rm.Material('Pt', rho=21.45, kind='mirror')
    ce = [np.sin(0.1)*12000., np.cos(0.1)*12000., 0.]
    oe = roe.OE(bl, 'poly', center=ce, pitch=5e-3, material=mPt, limPhysX=[-8, 8],
                limPhysY=[-120, 140], shape=list(POLYGON))
    n = 4096
    beam = g1.make_rays(rs, n, 77, sx=4.0, sz=0.35, sa=1e-4, sc=3e-5,
                        E=(7000., 11000.), amplitudes=False, pol='mixed')
    xx, yy = beam.x.copy(), beam.y.copy()
    aa, bb = beam.a.copy(), beam.b.copy()
    beam.x[:], beam.y[:] = raycing.rotate_z(xx, yy, bl.cosAzimuth, -bl.sinAzimuth)
    beam.a[:], beam.b[:] = raycing.rotate_z(aa, bb, bl.cosAzimuth, -bl.sinAzimuth)
    beam.state[5] = 2
    beam.state[7] = -1
    par = g1.oe_params(oe, dict(kind='flat'))
    par['shape'] = [list(v) for v in POLYGON]
    par['material'] = g1.material_dict(tables, mPt)
    # rays_good of the reference on hand-made points
    v = np.array(POLYGON)
    mid = (v + np.roll(v, -1, axis=0)) / 2
    rng = np.random.default_rng(5)
    px = np.concatenate([v[:, 0], mid[:, 0], rng.uniform(-9, 9, 40), np.repeat(v[:, 0], 3),
                         rng.integers(-9, 10, 300).astype(float)])
    py = np.concatenate([v[:, 1], mid[:, 1], np.tile(v[:, 1], 5),
                         np.tile(np.array([-121., 10., 150.]), len(v)),
                         rng.integers(-13, 15, 300) * 10.])
    st = oe.rays_good(px, py, None)
    from . import reflect_np as rn
    assert np.array_equal(st, rn.rays_good(par, px, py))
    orig_flat = g1.flat_params

    def flat_without_shape(p):
        q = dict(p)
        q['shape'] = 'polygon'
        return orig_flat(q)
    g1.flat_params = flat_without_shape
    try:
        g1.run_reflect('g2_polygon', rs, oe, par, beam, mat_rho=np.array(21.45),
                       polygon=v, pip_x=px, pip_y=py, pip_state=st)
    finally:
        g1.flat_params = orig_flat


if __name__ == '__main__':
    main()
