"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g2_support_*.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only): mirrors on their mechanical
supports (oes/__init__.py:212-587, stages.py) --

  g2_support_vcm     VCM with two coating stripes; the second one is moved into the beam
                     (select_surface: x shift by the stage, its own limits and material)
  g2_support_vfm     VFM: sagittal cylinder levelled off beyond the optical x limits
  g2_support_dualvfm DualVFM with its second cylinder selected

After construction the jacks / stages are moved and get_orientation() sets the angles (stored
with the case, so that the test also checks the product's stage arithmetic). While generating,
oracle/reflect_np.py ('vfm', 'dualvfm' surfaces) is asserted against the reference.

Run:  python -m oracle.gen_fixtures_supports
"""
import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1

SUPPORT = dict(jack1=[-50., 24700., 0.], jack2=[60., 25000., 0.], jack3=[-40., 25300., 0.],
               tx1=[0., -300.], tx2=[0., 300.])
STRIPES = {'Si': (('Si',), (1,), 2.33), 'Rh': (('Rh',), (1,), 12.41), 'Pt': (('Pt',), (1,), 21.45)}
CASES = {
    'vcm': ('VCM', dict(surface=('Si', 'Rh'), limPhysX=((-15., 3.), (-3., 15.)),
                        limPhysY=(-600., 600.), limOptX=((-13., 5.), (-5., 13.)),
                        limOptY=((-550., -550.), (550., 550.)), R=5e6, pitch=2.5e-3),
            ('Si', 'Rh'), 'Rh'),
    'vfm': ('VFM', dict(surface=None, limPhysX=(-20., 20.), limPhysY=(-500., 500.),
                        limOptX=(-3., 3.), limOptY=(-480., 480.), R=6e6, r=35., pitch=2.5e-3),
            ('Pt',), None),
    'dualvfm': ('DualVFM', dict(surface=('Rh', 'Pt'), limPhysX=((2., -40.), (40., -2.)),
                                limPhysY=(-500., 500.), limOptX=((10., -35.), (35., -12.)),
                                limOptY=((-480., -480.), (480., 480.)), R=5.5e6,
                                pitch=2.5e-3),
                ('Rh', 'Pt'), 'Pt'),
}
MOVES = dict(jack1=0.05, jack3=-0.04, tx1=0.3)      # added to the jack heights / stage x


def materials(M, names):
    return tuple(M.Material(STRIPES[n][0], quantities=STRIPES[n][1], rho=STRIPES[n][2],
                            kind='mirror')
                 for n in names)


def build(R, O, M, case):
    cls, kw, stripes, select = CASES[case]
    bl = R.BeamLine(azimuth=0.02)
    support = {k: list(v) for k, v in SUPPORT.items()}
    oe = getattr(O, cls)(bl, case, [np.sin(0.02)*25000., np.cos(0.02)*25000., 0.],
                         material=materials(M, stripes), **kw, **support)
    if select is not None:
        oe.select_surface(select)
    if hasattr(oe, 'hCylinder'):        # lift the bottom of the chosen groove into the beam
        oe.center[2] += oe.hCylinder
        oe.set_jacks()
    oe.jack1[2] += MOVES['jack1']
    oe.jack3[2] += MOVES['jack3']
    oe.tx1[0] += MOVES['tx1']
    oe.get_orientation()
    return bl, oe


def surface_of(case, oe):
    if case == 'vcm':
        return dict(kind='bentflat', R=oe.R, y0=oe.limPhysY[0])
    if case == 'vfm':
        return dict(kind='vfm', r=oe.r, R=oe.R, y0=oe.limPhysY[0], limOptX=list(oe.limOptX))
    return dict(kind='dualvfm', R=oe.R, y0=oe.limPhysY[0],
                **{k: getattr(oe, k) for k in ('r1', 'r2', 'xCylinder1', 'xCylinder2',
                                               'hCylinder1', 'hCylinder2')})


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    from .fixture_io import tables as load_tables
    tables = load_tables()
    for seed, case in enumerate(CASES):
        bl, oe = build(raycing, roe, rm, case)
        beam = g1.make_rays(rs, 1024, 170 + seed, sx=4., sz=0.3, sa=5e-5, sc=1e-5,
                            E=(8000., 12000.), amplitudes=True, pol='mixed')
        # the source sits on the (rotated) beamline axis
        beam.x[:], beam.y[:] = (np.cos(0.02)*beam.x + np.sin(0.02)*beam.y,
                                -np.sin(0.02)*beam.x + np.cos(0.02)*beam.y)
        beam.a[:], beam.b[:] = (np.cos(0.02)*beam.a + np.sin(0.02)*beam.b,
                                -np.sin(0.02)*beam.a + np.cos(0.02)*beam.b)
        beam.state[1] = 2
        par = g1.oe_params(oe, surface_of(case, oe))
        stripe = CASES[case][2][oe.curSurface]
        par['material'] = g1.material_dict(tables, materials(rm, (stripe,))[0])
        g1.run_reflect('g2_support_' + case, rs, oe, par, beam,
                       orientation=np.array([oe.pitch, oe.roll, oe.yaw, oe.dx, oe.center[2]]),
                       stripe=np.array(stripe))


if __name__ == '__main__':
    main()
