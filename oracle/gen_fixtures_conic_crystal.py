"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g3_conic_crystal_*.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only): a Bragg crystal on a surface
OUTSIDE the flat / bent-crystal families -- the reference's _reflect_local is generic in
(surface, material), oes/reflect.py:551-1139 --

  g3_conic_crystal_parabola   ParabolicalMirrorParam (focusing paraboloid, parametric root
                              solve) with Si(111) at its Bragg angle for 9 keV
  g3_conic_crystal_vfm        VFM (sagittal cylinder + meridional parabola) with Si(111)

While generating, oracle/reflect_np.py is asserted against the reference.

Run:  python -m oracle.gen_fixtures_conic_crystal
"""
import numpy as np

from . import _refenv
from . import reflect_np as rn
from .gen_fixtures_p1 import oe_params, make_rays, run_reflect, crystal_dict
from .fixture_io import tables as load_tables


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    raycing._VERBOSITY_ = 0
    tables = load_tables()
    si = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
    thB = float(np.ravel(si.get_Bragg_angle(9000.) - si.get_dtheta(9000.))[0])
    n = 2048
    # ---- paraboloid -------------------------------------------------------------------
    bl = raycing.BeamLine()
    m = roe.ParabolicalMirrorParam(bl, 'analyser', center=[0, 30000., 0], material=si,
                                   pitch=thB, p=None, q=8000., limPhysX=(-1.5, 1.5),
                                   limPhysY=(-20., 20.), alarmLevel=None)
    beam = make_rays(rs, n, 81, sx=0.05, sz=0.05, sa=1.5e-5, sc=1.5e-5, E=(8998., 9002.),
                     amplitudes=True, pol='mixed')
    beam.x[0] = 5.
    beam.state[1] = 3
    beam.state[2] = -2
    keys = ('cosGamma', 'sinGamma', 'y0', 'z0', 'parabParam')
    surf = dict(kind='parabola_param', isCylindrical=bool(m.isCylindrical),
                isClosed=bool(m.isClosed))
    for k in keys:
        surf[k] = float(getattr(m, k))
    mine = rn.make_parabola_param(None, 8000., abs(np.arcsin(np.sin(thB))), False)
    for k in keys:
        assert abs(mine[k] - surf[k]) <= 1e-15 * max(1., abs(surf[k])), k
    par = oe_params(m, surf)
    par['material'] = crystal_dict(tables, si)
    extra = {'surf_' + k: np.array(surf[k]) for k in keys}
    run_reflect('g3_conic_crystal_parabola', rs, m, par, beam, bragg=np.array(thB),
                surf_q=np.array(8000.), **extra)
    # ---- VFM ------------------------------------------------------------------------------
    bl = raycing.BeamLine()
    support = dict(jack1=[-50., 24700., 0.], jack2=[60., 25000., 0.], jack3=[-40., 25300., 0.],
                   tx1=[0., -300.], tx2=[0., 300.])
    v = roe.VFM(bl, 'vfm', [0., 25000., 0.], material=(si,), surface=None,
                limPhysX=(-20., 20.), limPhysY=(-60., 60.), limOptX=(-3., 3.),
                limOptY=(-50., 50.), R=6e6, r=35., pitch=thB, **support)
    beam = make_rays(rs, n, 82, sx=1.5, sz=0.3, sa=5e-5, sc=1e-5, E=(8998., 9002.),
                     amplitudes=True, pol='mixed')
    beam.state[1] = 2
    par = oe_params(v, dict(kind='vfm', r=v.r, R=v.R, y0=v.limPhysY[0],
                            limOptX=list(v.limOptX)))
    par['material'] = crystal_dict(tables, si)
    run_reflect('g3_conic_crystal_vfm', rs, v, par, beam, bragg=np.array(thB),
                surf_rR=np.array([v.r, v.R]))


if __name__ == '__main__':
    main()
