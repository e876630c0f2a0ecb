"""TEST INFRASTRUCTURE ONLY — regenerates by RUNNING THE REFERENCE (imported
from /root/reference, build container only) the golden vectors of the surface
kinds the SoftiMAX wave benchmark needs (SURVEY 8f row N4,
tests/speed/3_Softi_CXIw2D_speed.py:176-246):

  g2_blazed_au.npz      BlazedGrating (gratings.py:316-535) in ray mode: the ad
                        hoc first-facet intersection, facet normals, Au mirror
                        amplitudes, positionRoll = pi as mounted in the PGM
  g2_ellipse_cyl.npz    EllipticalMirrorParam(isCylindrical=True) (the KB pair
                        M4/M5): parametric bracketed root solve in (s, phi, r)
  g2_ellipse_full.npz   same, ellipsoid of revolution (isCylindrical=False)
  g2_ellipse_cyl_nis.npz  reflect(noIntersectionSearch=True) on points that
                        already lie on the surface (what follows a diffract)
  g2_parabola_q.npz / g2_parabola_p_cyl.npz / g2_hyperbola.npz
                        ParabolicalMirrorParam (focusing paraboloid; collimating
                        parabolic cylinder) and HyperbolicMirrorParam
                        (parametric.py:252-716): the other two conics of the
                        same parametric family
  g2_grating_vls.npz    plane grating with the grating EQUATION (material
                        kind='grating', reflect.py:840-861, 451-469): VLS line
                        density polynomial along y, order -1, as a PGM grating
  g2_grating_const.npz  a subclass with a constant local_g (the SoftiMAX
                        example's `Grating`), order +1, lines along x
  g2_grating_orders.npz a sequence of orders (1, -1, 2, 0): one per hit ray from
                        numpy's global generator (reflect.py:455-458), seeded

While generating, oracle/reflect_np.py is asserted against the reference.

Run:  python -m oracle.gen_fixtures_softi
"""
import numpy as np

from . import _refenv
from . import reflect_np as rn
from .gen_fixtures_p1 import (oe_params, make_rays, run_reflect, material_dict,
                              to_oracle_beam, assert_beams, beam_dict,
                              flat_params, save)
from .fixture_io import tables as load_tables

SURF_KEYS_BLAZED = ('blaze', 'antiblaze', 'rho0')
SURF_KEYS_ELL = ('p', 'q', 'cosGamma', 'sinGamma', 'y0', 'z0', 'ellipseA',
                 'ellipseB')


def ellipse_surface(oe):
    d = dict(kind='ellipse_param', isCylindrical=bool(oe.isCylindrical),
             isClosed=bool(oe.isClosed))
    for k in SURF_KEYS_ELL:
        d[k] = float(getattr(oe, k))
    return d


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    raycing._VERBOSITY_ = 0
    tables = load_tables()
    mAu = rm.Material('Au', rho=19.32, kind='mirror')
    n = 2048

    # ---------------- blazed grating as in the SoftiMAX PGM ---------------
    bl = raycing.BeamLine()
    blaze, rho = np.radians(0.6), 300.
    beta = -np.radians(86.5)
    pg = roe.BlazedGrating(
        bl, 'pg', center=[0, 2000., 20.], pitch=-(beta + np.pi/2),
        positionRoll=np.pi, material=mAu, blaze=blaze, rho=rho,
        limPhysX=(-2, 2), limPhysY=(-40, 40), alarmLevel=None)
    # a fan that arrives from below (the grating looks down) at ~alpha
    beam = make_rays(rs, n, 61, sx=1.0, sz=0.9, sa=3e-5, sc=2e-5,
                     E=(275., 285.), amplitudes=True, pol='mixed')
    inc = np.radians(2.0)                 # elevation of the incoming fan
    beam.y[:] = 2000. - 1500.*np.cos(inc)
    beam.z[:] += 20. - 1500.*np.sin(inc)
    cc = beam.c + np.sin(inc)
    beam.c[:] = cc
    beam.b[:] = np.sqrt(1 - beam.a**2 - beam.c**2)
    beam.state[5] = 2
    beam.state[6] = -2
    surf = rn.make_blazed(blaze, rho)
    for k in ('sinBlaze', 'cosBlaze', 'tanBlaze', 'sinAntiblaze', 'cosAntiblaze',
              'tanAntiblaze', 'rho_1'):
        assert surf[k] == getattr(pg, k), k
    par = oe_params(pg, surf)
    par['material'] = material_dict(tables, mAu)
    run_reflect('g2_blazed_au', rs, pg, par, beam, mat_rho=np.array(19.32),
                surf_blaze=np.array(blaze), surf_antiblaze=np.array(pg.antiblaze),
                surf_rho=np.array(rho))

    # ---------------- grating equation (ray mode) -------------------------
    mAuG = rm.Material('Au', rho=19.32, kind='grating')
    for tag, kw, order in (
            ('g2_grating_vls',
             dict(gratingDensity=['y', 300., 1., 2.4e-4, -3.1e-8]), -1),
            ('g2_grating_const', dict(), 1),
            ('g2_grating_orders', dict(gratingDensity=['y', 300., 1., 2.4e-4]),
             (1, -1, 2, 0)),
            ('g2_grating_efficiency', dict(gratingDensity=['y', 300., 1., 2.4e-4]),
             (1, -1, 2, 0))):
        bl = raycing.BeamLine()
        if kw:
            cls = roe.OE
        else:
            class XGrating(roe.OE):
                def local_g(self, x, y, rho=120.):
                    return rho, 0, 0          # grooves along y: sagittal mount
            cls = XGrating
        efficiency = [[1, 0.31], [-1, 0.12], [2, 0.045]] if tag.endswith('efficiency') \
            else None               # order 0 is not listed: those rays get no intensity
        mG = rm.Material('Au', rho=19.32, kind='grating', efficiency=efficiency) \
            if efficiency else mAuG
        gr = cls(bl, 'gr', center=[0, 2000., 0.], pitch=np.radians(2.2),
                 material=mG, order=order, limPhysX=(-3, 3), limPhysY=(-45, 45),
                 alarmLevel=None, **kw)
        several = isinstance(order, tuple)      # one order per hit ray, drawn at random
        beam = make_rays(rs, n, ((68 if tag.endswith('efficiency') else 66) if several
                                 else 64) if kw else 65, sx=1.0, sz=0.9, sa=3e-5,
                         sc=2e-5, E=(270., 290.), amplitudes=True, pol='mixed')
        beam.state[3] = 3
        beam.state[4] = -4
        par = oe_params(gr, dict(kind='flat'))
        par['material'] = material_dict(tables, mG)
        if efficiency:
            par['material']['efficiency'] = efficiency
        par['order'] = order
        if kw:
            par['gratingDensity'] = kw['gratingDensity']
        else:
            par['gVector'] = (120., 0, 0)
        extra = dict(mat_rho=np.array(19.32), order=np.array(order))
        if several:
            extra['np_seed'] = 20260928 + int(bool(efficiency))
        if efficiency:
            extra['efficiency'] = np.array(efficiency, dtype=float)
        if kw:
            extra['gd_axis'] = np.array(kw['gratingDensity'][0])
            extra['gd_coeffs'] = np.array(kw['gratingDensity'][1:], dtype=float)
        else:
            extra['g_vector'] = np.array([120., 0., 0.])
        run_reflect(tag, rs, gr, par, beam, **extra)

    # ---------------- elliptical mirrors (parametric) ---------------------
    pitch = np.radians(1)
    for tag, cyl, kw in (('g2_ellipse_cyl', True, dict(positionRoll=np.pi/2)),
                         ('g2_ellipse_full', False, dict())):
        bl = raycing.BeamLine()
        m = roe.EllipticalMirrorParam(
            bl, 'm4', center=[0, 43000., 0], material=mAu, pitch=pitch,
            isCylindrical=cyl, p=43000., q=5000., limPhysX=(-0.5, 0.5),
            limPhysY=(-70., 70.), alarmLevel=None, **kw)
        beam = make_rays(rs, n, 62 if cyl else 63, sx=0.02, sz=0.02, sa=1.2e-5,
                         sc=1.2e-5, E=(275., 285.), amplitudes=True, pol='mixed')
        beam.x[0] = 3.        # misses
        beam.state[1] = 2
        beam.state[2] = -1
        surf = ellipse_surface(m)
        absPitch = abs(np.arcsin(np.sin(pitch)))
        mine = rn.make_ellipse_param(43000., 5000., absPitch, cyl)
        for k in SURF_KEYS_ELL:
            assert abs(mine[k] - surf[k]) <= 1e-15 * max(1., abs(surf[k])), k
        par = oe_params(m, surf)
        par['material'] = material_dict(tables, mAu)
        extra = {'surf_' + k: np.array(surf[k]) for k in SURF_KEYS_ELL}
        extra['surf_isCylindrical'] = np.array(float(cyl))
        run_reflect(tag, rs, m, par, beam, mat_rho=np.array(19.32), **extra)

        if cyl:
            # noIntersectionSearch: start from the local beam of the ray pass
            # (points on the surface), expressed in the global frame again
            gb, lb = m.reflect(beam)
            onsurf = rs.Beam(copyFrom=lb)
            good = lb.state == 1
            onsurf.filter_by_index(good)
            m.local_to_global(onsurf)
            # local_to_global rotated the amplitudes as well; fine: it is just
            # another input beam lying on the surface
            onsurf.a[:], onsurf.b[:], onsurf.c[:] = 0., 1., 0.
            inb = rs.Beam(copyFrom=onsurf)
            gb2, lb2 = m.reflect(rs.Beam(copyFrom=inb), noIntersectionSearch=True)
            mg, ml = rn.oe_reflect(par, to_oracle_beam(inb),
                                   noIntersectionSearch=True)
            assert_beams(tag + '_nis:gb', mg, gb2)
            assert_beams(tag + '_nis:lb', ml, lb2)
            out = {}
            out.update(beam_dict('in_', inb))
            out.update(beam_dict('gb_', gb2))
            out.update(beam_dict('lb_', lb2))
            out.update(flat_params(par))
            out.update(extra)
            out['mat_rho'] = np.array(19.32)
            save(tag + '_nis', **out)
            st, cnt = np.unique(lb2.state, return_counts=True)
            print(tag + '_nis', 'states', dict(zip(st.tolist(), cnt.tolist())))


def conics(raycing, rs, roe, rm, tables):
    mAu = rm.Material('Au', rho=19.32, kind='mirror')
    n = 2048
    pitch = np.radians(1.2)
    cases = (
        ('g2_parabola_q', roe.ParabolicalMirrorParam,
         dict(p=None, q=8000.), 71, rn.make_parabola_param),
        ('g2_parabola_p_cyl', roe.ParabolicalMirrorParam,
         dict(p=30000., isCylindrical=True, positionRoll=np.pi/2), 72,
         rn.make_parabola_param),
        ('g2_hyperbola', roe.HyperbolicMirrorParam, dict(p=30000., q=6000.), 73,
         rn.make_hyperbola_param))
    for tag, cls, kw, seed, maker in cases:
        bl = raycing.BeamLine()
        m = cls(bl, 'm', center=[0, 30000., 0], material=mAu, pitch=pitch,
                limPhysX=(-1.5, 1.5), limPhysY=(-90., 90.), alarmLevel=None, **kw)
        beam = make_rays(rs, n, seed, sx=0.05, sz=0.05, sa=1.5e-5, sc=1.5e-5,
                         E=(7000., 9000.), amplitudes=True, pol='mixed')
        beam.x[0] = 5.
        beam.state[1] = 3
        beam.state[2] = -2
        keys = ('cosGamma', 'sinGamma', 'y0', 'z0') + (
            ('parabParam',) if 'parabola' in tag else ('hyperbolaA', 'hyperbolaB'))
        surf = dict(kind='parabola_param' if 'parabola' in tag else 'hyperbola_param',
                    isCylindrical=bool(m.isCylindrical), isClosed=bool(m.isClosed))
        for k in keys:
            surf[k] = float(getattr(m, k))
        mine = maker(kw.get('p'), kw.get('q'), abs(np.arcsin(np.sin(pitch))),
                     bool(m.isCylindrical))
        for k in keys:
            assert abs(mine[k] - surf[k]) <= 1e-15 * max(1., abs(surf[k])), (tag, k)
        par = oe_params(m, surf)
        par['material'] = material_dict(tables, mAu)
        extra = {'surf_' + k: np.array(surf[k]) for k in keys}
        extra['surf_isCylindrical'] = np.array(float(m.isCylindrical))
        extra['surf_p'] = np.array(np.nan if kw.get('p') is None else kw['p'])
        extra['surf_q'] = np.array(np.nan if kw.get('q') is None else kw['q'])
        run_reflect(tag, rs, m, par, beam, mat_rho=np.array(19.32), **extra)


def main_conics():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    raycing._VERBOSITY_ = 0
    conics(raycing, rs, roe, rm, load_tables())


if __name__ == '__main__':
    import sys
    if 'conics' in sys.argv:
        main_conics()
        sys.exit(0)
    main()
