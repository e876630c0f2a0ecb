"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g2_user_surface.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only): an OE subclass that defines
its surface the reference's usual way, by overriding local_z / local_n with numpy code
(tests/user_surface_case.py), reflecting 4096 rays with a Pt coating: the bulk of the cfg2
generator plus rays that miss, fall off the edges and graze the surface. While generating,
oracle/reflect_np.py (surface kind 'user' = the same two callables) is asserted against the
reference's beams.

  g3_user_crystal  the same figured surface with Si(111) at its Bragg angle (9 keV)
  g2_user_multilayer / g2_user_coated  the figured surface under a W/Si multilayer at its
                   Bragg angle and under a Rh coating (layered flavour of the unit)
  g2_user_grating  a plane grating whose groove vector is a function of (x, y) given by the
                   subclass's local_g (a fan of lines with a quadratic density law), order -1

Run:  python -m oracle.gen_fixtures_user_surface
"""
import os
import sys

import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                'tests'))


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    import user_surface_case as case
    from .fixture_io import tables as load_tables
    bl = raycing.BeamLine()
    pt = rm.Material('Pt', rho=21.45, kind='mirror')
    oe = case.subclass(roe)(bl, 'figured', center=[0, case.P, 0], pitch=case.PITCH,
                            material=pt, **case.LIMITS)
    beam = g1.make_rays(rs, 4096, 61, amplitudes=True, pol='mixed')
    # edge rays: wide in x (off the sides), steep (over the ends / missing), one grazing
    beam.x[:64] = np.linspace(-14., 14., 64)
    beam.c[64:128] = np.linspace(-3e-5, 3e-5, 64)
    beam.z[128:160] = np.linspace(-1.5, 1.5, 32)
    beam.a[:], beam.c[:] = beam.a, beam.c
    beam.b[:] = np.sqrt(1 - beam.a**2 - beam.c**2)
    beam.state[200] = 2
    beam.state[201] = -3
    par = g1.oe_params(oe, dict(kind='user', z=case.numpy_local_z, n=case.numpy_local_n))
    par['material'] = g1.material_dict(load_tables(), pt)
    g1.run_reflect('g2_user_surface', rs, oe, par, beam,
                   surface_parameters=np.array([case.RS, case.RM, case.K3, case.KT]),
                   mat_rho=np.array(21.45))
    # a grating whose groove vector the subclass defines (order -1, Au, 280 eV)
    bl = raycing.BeamLine()
    au = rm.Material('Au', rho=19.32, kind='grating')
    gr = case.grating_subclass(roe)(bl, 'fan', center=[0, 2000., 0.], pitch=np.radians(2.2),
                                    material=au, order=-1, alarmLevel=None, **case.G_LIMITS)
    beam = g1.make_rays(rs, 2048, 63, sx=1.0, sz=0.9, sa=3e-5, sc=2e-5, E=(270., 290.),
                        amplitudes=True, pol='mixed')
    beam.state[3] = 3
    beam.state[4] = -4
    par = g1.oe_params(gr, dict(kind='flat'))
    par['material'] = g1.material_dict(load_tables(), au)
    par['local_g'] = case.numpy_local_g
    par['order'] = -1
    g1.run_reflect('g2_user_grating', rs, gr, par, beam,
                   groove_parameters=np.array([case.G_RHO0, case.G_B1, case.G_B2, case.G_BX]))
    # the figured surface as a Si(111) crystal at the Bragg angle for 9 keV
    raycing._VERBOSITY_ = 0
    bl = raycing.BeamLine()
    si = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
    thB = float(np.ravel(si.get_Bragg_angle(9000.) - si.get_dtheta(9000.))[0])
    xt = case.crystal_element(roe, bl, si, thB)
    beam = g1.make_rays(rs, 2048, 67, sx=0.3, sz=0.05, sa=2e-5, sc=8e-6, E=(8999., 9001.),
                        amplitudes=True, pol='mixed')
    beam.x[0] = 9.
    beam.state[1] = 3
    beam.state[2] = -2
    par = g1.oe_params(xt, dict(kind='user', z=case.numpy_local_z, n=case.numpy_local_n))
    par['material'] = g1.crystal_dict(load_tables(), si)
    g1.run_reflect('g3_user_crystal', rs, xt, par, beam, bragg=np.array(thB),
                   surface_parameters=np.array([case.RS, case.RM, case.K3, case.KT]))
    # the figured surface under layered materials: a periodic W/Si multilayer at its
    # refraction-corrected Bragg angle for 9 keV (deflects like a crystal of its period) and a
    # Rh coating on Si at 4 mrad (a mirror); stacks of oracle/gen_fixtures_multilayer.py
    from . import gen_fixtures_multilayer as gm
    tables = gm.all_tables()
    for tag, stack, energy in (('g2_user_multilayer', 'wsi', 9000.),
                               ('g2_user_coated', 'rh_coated', None)):
        bl = raycing.BeamLine()
        ml = gm.ref_stack(rm, stack)
        pitch = case.PITCH if energy is None else \
            float(ml.get_Bragg_angle(energy) - ml.get_dtheta(energy))
        oe = case.subclass(roe)(bl, 'figured', center=[0, case.P, 0], pitch=pitch, material=ml,
                                **case.LIMITS)
        beam = g1.make_rays(rs, 2048, 71, sx=0.3, sz=0.1, sa=5e-5, sc=1.5e-4,
                            E=(8950., 9050.), amplitudes=True, pol='mixed')
        beam.state[3] = 2
        beam.state[4] = -3
        beam.x[5] = 14.
        par = g1.oe_params(oe, dict(kind='user', z=case.numpy_local_z, n=case.numpy_local_n))
        par['material'] = gm.oracle_stack(tables, stack)
        g1.run_reflect(tag, rs, oe, par, beam, stack=np.array(stack), pitch=np.array(pitch),
                       surface_parameters=np.array([case.RS, case.RM, case.K3, case.KT]))


if __name__ == '__main__':
    main()
