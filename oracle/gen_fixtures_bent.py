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


# bent crystals in Laue geometry (oes/laue.py:26-227, 455-507): Si(111) 'Laue reflected', 0.1 mm
LAUE_CASES = (
    ('g3_bent_laue_cyl', 'BentLaueCylinder', dict(R=3000.), 'laue'),
    ('g3_bent_laue_cyl_circ_asym', 'BentLaueCylinder',
     dict(R=2500., crossSection='circular', alpha=np.radians(6.)), 'laue'),
    ('g3_bent_laue_ground', 'GroundBentLaueCylinder',
     dict(R=2000., crossSection='circular', alpha=np.radians(-4.)), 'laue_ground'),
    ('g3_bent_laue_sphere', 'BentLaueSphere', dict(R=4000., crossSection='circular'), 'laue'),
    ('g3_bent_laue_paraboloid', 'BentLaueSphere', dict(R=4000.), 'laue'),
    ('g3_bent_laue_2d', 'BentLaue2D', dict(Rm=3000., Rs=-9000., alpha=np.radians(5.)), 'laue'),
)


# diced elements (oes/bragg.py:8-101, 345-375)
DICED_CASES = (
    ('g3_diced_flat', 'DicedOE', dict(dxFacet=2.1, dyFacet=1.4, dxGap=0.3, dyGap=0.2,
                                      alpha=np.radians(1.5)), 'johann'),
    ('g3_diced_johann_tor', 'DicedJohannToroid', dict(Rs='sagittal', dxFacet=1.8, dyFacet=2.5,
                                                      dxGap=0.2, dyGap=0.25), 'johann'),
    ('g3_diced_johansson_tor', 'DicedJohanssonToroid',
     dict(Rs='sagittal', alpha=np.radians(2.), dxFacet=3., dyFacet=4., dxGap=0.1, dyGap=0.3),
     'johansson'),
)


def diced_surface_of(cls_name, kw, planes, thB):
    surf = dict(kind='diced', base='flat' if cls_name == 'DicedOE' else 'toroid', planes=planes,
                alpha=kw.get('alpha'), Rm=RM, Rs=RM * np.sin(thB)**2, crossSection='circular',
                xStep=kw['dxFacet'] + kw['dxGap'], yStep=kw['dyFacet'] + kw['dyGap'],
                dxFacet=kw['dxFacet'], dyFacet=kw['dyFacet'])
    surf['RmBragg'], surf['RsBragg'] = surf['Rm'], surf['Rs']
    return surf


def laue_surface_of(cls_name, kw, planes):
    if cls_name == 'BentLaue2D':
        return dict(kind='laue_2d', Rm=kw['Rm'], Rs=kw['Rs'], alpha=kw.get('alpha'),
                    planes=planes, crossSection='parabolic')
    return dict(kind='laue_sphere' if 'Sphere' in cls_name else 'bent_cylinder', Rm=kw['R'],
                planes=planes, alpha=kw.get('alpha'),
                crossSection=kw.get('crossSection', 'parabolic'))


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
    for seed, (tag, cls_name, kw, planes) in enumerate(DICED_CASES):
        bl = raycing.BeamLine()
        si = rm.CrystalSi(hkl=(1, 1, 1))
        thB = float(si.get_Bragg_angle(E0))
        surf = diced_surface_of(cls_name, kw, planes, thB)
        args = {k: v for k, v in kw.items() if k != 'Rs'}
        flat = cls_name == 'DicedOE'
        if not flat:
            args.update(Rs=surf['Rs'], Rm=RM)
        alpha = kw.get('alpha') or 0.
        p = RM * np.sin(thB + alpha)
        oe = getattr(roe, cls_name)(bl, 'dc', center=[0, p, 0], pitch=thB + alpha, material=si,
                                    limPhysX=[-12, 12], limPhysY=[-35, 35], **args)
        spread = 1e-5 if flat else 1.5e-2
        size = 2.5 if flat else 0.02
        beam = g1.make_rays(rs, n, 210 + seed, sx=size, sz=size, sa=spread, sc=spread,
                            E=(E0 - 2., E0 + 2.), amplitudes=True, pol='mixed')
        beam.state[1] = 2
        par = g1.oe_params(oe, surf)
        par['material'] = g1.crystal_dict(tables, si)
        extra = {'surf_' + k: np.array(v if v is not None else 0.) for k, v in surf.items()
                 if k not in ('kind', 'planes', 'crossSection', 'base')}
        g1.run_reflect(tag, rs, oe, par, beam, surf_class=np.array(cls_name),
                       cr_d=np.array(si.d), cr_chiToF=np.array(si.chiToF),
                       cr_V=np.array(si.V), **extra)
    for seed, (tag, cls_name, kw, planes) in enumerate(LAUE_CASES):
        bl = raycing.BeamLine()
        si = rm.CrystalSi(hkl=(1, 1, 1), geom='Laue reflected', t=0.1)
        alpha = kw.get('alpha')
        thB = si.get_Bragg_angle(E0) - si.get_dtheta(E0, alpha)
        thB = float(thB[0] if np.ndim(thB) else thB)
        oe = getattr(roe, cls_name)(bl, 'bl', center=[0, 10000., 0],
                                    pitch=thB + (alpha if alpha else 0) + np.pi/2, material=si,
                                    limPhysX=[-5, 5], limPhysY=[-1.5, 1.8], **kw)
        beam = g1.make_rays(rs, n, 190 + seed, sx=0.6, sz=0.5, sa=3e-5, sc=3e-5,
                            E=(E0 - 1., E0 + 1.), amplitudes=True, pol='mixed')
        beam.state[2] = 2
        beam.state[3] = -3
        surf = laue_surface_of(cls_name, kw, planes)
        par = g1.oe_params(oe, surf)
        par['material'] = g1.crystal_dict(tables, si)
        g1.run_reflect(tag, rs, oe, par, beam, surf_class=np.array(cls_name),
                       surf_crossSection=np.array(surf['crossSection']),
                       surf_Rm=np.array(surf['Rm']), surf_Rs=np.array(surf.get('Rs', 0.)),
                       surf_alpha=np.array(alpha if alpha else 0.),
                       cr_d=np.array(si.d), cr_chiToF=np.array(si.chiToF), cr_V=np.array(si.V),
                       cr_t=np.array(0.1))


if __name__ == '__main__':
    main()
