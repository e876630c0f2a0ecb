"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g3_bent_*.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only): bent crystal analysers
(oes/bragg.py:104-343) with Si(111) on the Rowland circle,

  g3_bent_johann_cyl        JohannCylinder, circular cross section
  g3_bent_johann_parab_asym JohannCylinder, parabolic, asymmetric cut alpha = 3 deg
  g3_bent_johansson_cyl     JohanssonCylinder (ground: planes of radius 2 Rm), alpha = -2 deg
  g3_bent_johann_tor        JohannToroid, Rs = Rm sin^2(theta_B)
  g3_bent_johann_tor_asym   JohannToroid with alpha = 4 deg
  g3_bent_johansson_tor     JohanssonToroid, alpha = 2 deg (the reference turns the tilted
                            plane normal sagittally twice, bragg.py:290-293)
  g3_bent_general_tor       GeneralBraggToroid, RmBragg = 2 Rm, RsBragg = 1.5 Rs

While generating, oracle/reflect_np.py ('bent_cylinder' / 'bent_toroid' surfaces) is
asserted against the reference.

Run:  python -m oracle.gen_fixtures_bent
"""
import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1
from .fixture_io import tables as load_tables

E0, RM = 9000., 1000.

CASES = (
    ('g3_bent_johann_cyl', 'JohannCylinder', dict(), 'johann'),
    ('g3_bent_johann_parab_asym', 'JohannCylinder',
     dict(crossSection='parabolic', alpha=np.radians(3.)), 'johann'),
    ('g3_bent_johansson_cyl', 'JohanssonCylinder', dict(alpha=np.radians(-2.)), 'johansson'),
    ('g3_bent_johann_tor', 'JohannToroid', dict(Rs='sagittal'), 'johann'),
    ('g3_bent_johann_tor_asym', 'JohannToroid', dict(Rs='sagittal', alpha=np.radians(4.)),
     'johann'),
    ('g3_bent_johansson_tor', 'JohanssonToroid', dict(Rs='sagittal', alpha=np.radians(2.)),
     'johansson'),
    ('g3_bent_general_tor', 'GeneralBraggToroid',
     dict(Rs='sagittal', RmBragg=2*RM, RsBragg='1.5 sagittal'), 'general'),
)


def surface_of(cls_name, kw, planes, thB):
    rs_ = RM * np.sin(thB)**2
    surf = dict(kind='bent_toroid' if 'Toroid' in cls_name else 'bent_cylinder', Rm=RM,
                planes=planes, alpha=kw.get('alpha'),
                crossSection=kw.get('crossSection', 'circular'))
    if 'Toroid' in cls_name:
        surf['Rs'] = rs_
        surf['RmBragg'] = kw.get('RmBragg', RM)
        surf['RsBragg'] = 1.5 * rs_ if 'RsBragg' in kw else rs_
    return surf


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    tables = load_tables()
    n = 1024
    for seed, (tag, cls_name, kw, planes) in enumerate(CASES):
        bl = raycing.BeamLine()
        si = rm.CrystalSi(hkl=(1, 1, 1))
        thB = float(si.get_Bragg_angle(E0))
        surf = surface_of(cls_name, kw, planes, thB)
        args = {k: v for k, v in kw.items() if k not in ('Rs', 'RsBragg')}
        if 'Rs' in kw:
            args['Rs'] = surf['Rs']
        if 'RsBragg' in kw:
            args['RsBragg'] = surf['RsBragg']
        alpha = kw.get('alpha') or 0.
        p = RM * np.sin(thB + alpha)          # source on the Rowland circle
        oe = getattr(roe, cls_name)(bl, 'an', center=[0, p, 0], pitch=thB + alpha, Rm=RM,
                                    material=si, limPhysX=[-12, 12], limPhysY=[-35, 35],
                                    **args)
        beam = g1.make_rays(rs, n, 120 + seed, sx=0.02, sz=0.02, sa=1.5e-2, sc=1.5e-2,
                            E=(E0 - 2., E0 + 2.), amplitudes=True, pol='mixed')
        beam.state[1] = 2
        beam.state[2] = -4
        par = g1.oe_params(oe, surf)
        par['material'] = g1.crystal_dict(tables, si)
        extra = {'surf_' + k: np.array(v if v is not None else 0.) for k, v in surf.items()
                 if k not in ('kind', 'planes', 'crossSection')}
        g1.run_reflect(tag, rs, oe, par, beam, surf_class=np.array(cls_name),
                       surf_crossSection=np.array(surf['crossSection']),
                       cr_d=np.array(si.d), cr_chiToF=np.array(si.chiToF),
                       cr_V=np.array(si.V), **extra)


if __name__ == '__main__':
    main()
