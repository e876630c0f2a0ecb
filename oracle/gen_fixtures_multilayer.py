"""TEST INFRASTRUCTURE ONLY -- regenerates the multilayer goldens by RUNNING THE REFERENCE
(imported from /root/reference, build container only; materials/multilayer.py):

  g6_layer_tables.npz          f0 / f1 / f2 tables of W, Mo, B, C (Si, Rh are in g6)
  g5_multilayer_amplitudes.npz Multilayer / GradedMultilayer / Coated .get_amplitude on
                               random (E, angle) points: periodic W/Si, depth-graded Mo/Si
                               with interdiffusion, W/B4C in transmission through a finite
                               substrate, a vacuum-spaced stack (no bottom layer), Rh-coated
                               Si with surface and substrate roughness
  g2_multilayer_flat.npz       OE + periodic W/Si at the 9 keV Bragg angle (the beam's
                               divergence scans the rocking curve)
  g2_ellipse_multilayer.npz    EllipticalMirrorParam + depth-graded Mo/Si
  g2_multilayer_tran.npz       OE + W/B4C, geom 'transmitted': rays go straight on
  g2_coated_toroid.npz         ToroidMirror + Coated(Rh on Si)

While generating, the numpy restatement (oracle/materials_np.py: multilayer_amplitude,
oracle/reflect_np.py) is asserted against the reference.

Run:  python -m oracle.gen_fixtures_multilayer
"""
import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1
from . import materials_np as mn
from . import reflect_np as rn

# formula -> (elements, quantities, density)
COMPOUNDS = {'W': (('W',), (1,), 19.3), 'Si': (('Si',), (1,), 2.33),
             'Mo': (('Mo',), (1,), 10.22), 'B4C': (('B', 'C'), (4, 1), 2.52),
             'Rh': (('Rh',), (1,), 12.41), 'C': (('C',), (1,), 2.2)}

# name -> Multilayer keyword arguments, materials by formula
STACKS = {
    'wsi': dict(tLayer='W', tThickness=12., bLayer='Si', bThickness=18., nPairs=40,
                substrate='Si'),
    'mosi_graded': dict(tLayer='Mo', tThickness=28., bLayer='Si', bThickness=41., nPairs=30,
                        substrate='Si', tThicknessLow=20., bThicknessLow=30., power=2.,
                        idThickness=3.),
    'wb4c_tran': dict(tLayer='W', tThickness=10., bLayer='B4C', bThickness=15., nPairs=25,
                      substrate='Si', idThickness=2., substThickness=2e4,
                      geom='transmitted'),
    'w_vacuum': dict(tLayer='W', tThickness=15., bLayer=None, bThickness=20., nPairs=8,
                     substrate='Si', power=1.5, tThicknessLow=11.),
    'c_on_si': dict(tLayer=None, tThickness=0., bLayer='C', bThickness=400., nPairs=1,
                    substrate='Si', substRoughness=4.),
    'rh_coated': dict(coating='Rh', cThickness=300., substrate='Si', surfaceRoughness=3.,
                      substRoughness=5.),
}


def ref_material(rm, formula):
    if formula is None:
        return None
    els, q, rho = COMPOUNDS[formula]
    return rm.Material(els, quantities=q, rho=rho)


def oracle_material(tables, formula):
    if formula is None:
        return None
    els, q, rho = COMPOUNDS[formula]
    return mn.make_material([mn.load_element(tables, e) for e in els], list(q), 'mirror', rho)


def ref_stack(rm, name):
    kw = dict(STACKS[name])
    for key in ('tLayer', 'bLayer', 'substrate', 'coating'):
        if key in kw:
            kw[key] = ref_material(rm, kw[key])
    return (rm.Coated if 'coating' in kw else rm.Multilayer)(**kw)


def oracle_stack(tables, name):
    kw = dict(STACKS[name])
    for key in ('tLayer', 'bLayer', 'substrate', 'coating'):
        if key in kw:
            kw[key] = oracle_material(tables, kw[key])
    if 'coating' in kw:
        return mn.make_coated(kw['coating'], kw['cThickness'], kw['substrate'],
                              kw['surfaceRoughness'], kw['substRoughness'])
    return mn.make_multilayer(**kw)


def all_tables():
    from . import fixture_io
    tb = dict(fixture_io.tables())
    tb.update(np.load(g1.os.path.join(g1.OUT, 'g6_layer_tables.npz')))
    return tb


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    extra = {}
    for name in ('W', 'Mo', 'B', 'C'):
        e = rm.Element(name, table='Chantler total')
        for key, val in (('Z', e.Z), ('mass', e.mass), ('f0', e.f0coeffs), ('E', e.E),
                         ('f1', e.f1), ('f2', e.f2)):
            extra['%s_%s' % (name, key)] = np.array(val, dtype=float)
    g1.save('g6_layer_tables', **extra)
    tables = all_tables()

    # ---- amplitudes on random (E, grazing angle) points -------------------------------
    rng = np.random.default_rng(2024)
    npts = 1500
    out = {}
    for name in STACKS:
        ml = ref_stack(rm, name)
        if name in ('c_on_si', 'rh_coated'):
            E = rng.uniform(3000., 25000., npts)
            theta = rng.uniform(0.5e-3, 12e-3, npts)
        else:
            E = rng.uniform(7000., 16000., npts)
            thB = np.arcsin(np.clip(12398.42 / (2 * ml.d * E), 0, 1))
            theta = thB * rng.uniform(0.3, 1.6, npts)     # through and around the peak
        bdn = -np.sin(theta)
        ref = ml.get_amplitude(E.copy(), bdn.copy())
        mine = mn.multilayer_amplitude(oracle_stack(tables, name), E, bdn)
        for r, m in zip(ref, mine):
            assert np.abs(m - r).max() <= 1e-13 * max(np.abs(r).max(), 1e-300), name
        peak = np.abs(ref[0]).max()
        print(name, 'max |s| %.4f  max |p| %.4f' % (peak, np.abs(ref[1]).max()))
        assert peak > (0.5 if name in ('wsi', 'mosi_graded', 'c_on_si', 'rh_coated')
                       else 0.05), name
        out.update({name + '_E': E, name + '_bdn': bdn, name + '_s': ref[0],
                    name + '_p': ref[1]})
    g1.save('g5_multilayer_amplitudes', **out)

    # ---- reflect passes ----------------------------------------------------------------
    n = 2048

    def run(tag, oe, surface, stack, beam, **more):
        par = g1.oe_params(oe, surface)
        par['material'] = oracle_stack(tables, stack)
        g1.run_reflect(tag, rs, oe, par, beam, stack=np.array(stack), **more)

    # flat element, periodic W/Si at its refraction-corrected Bragg angle
    bl = raycing.BeamLine()
    ml = ref_stack(rm, 'wsi')
    thB = float(ml.get_Bragg_angle(9000.) - ml.get_dtheta(9000.))
    oe = roe.OE(bl, 'ml', center=[0, 1000., 0], pitch=thB, material=ml,
                limPhysX=[-5, 5], limPhysY=[-60, 60])
    beam = g1.make_rays(rs, n, 91, sx=0.5, sz=0.2, sa=1e-4, sc=4e-4, E=(8950., 9050.),
                        amplitudes=True, pol='mixed')
    beam.state[3] = 2
    beam.state[4] = -3
    beam.x[5] = 9.
    run('g2_multilayer_flat', oe, dict(kind='flat'), 'wsi', beam)

    # elliptical mirror (parametric surface), depth-graded Mo/Si with interdiffusion
    bl = raycing.BeamLine()
    ml = ref_stack(rm, 'mosi_graded')
    p_, q_ = 20000., 1500.
    thB = float(ml.get_Bragg_angle(10000.) - ml.get_dtheta(10000.))
    em = roe.EllipticalMirrorParam(bl, 'em', center=[0, p_, 0], pitch=thB, p=p_, q=q_,
                                   material=ml, limPhysX=[-3, 3], limPhysY=[-60, 60],
                                   isCylindrical=True)
    beam = g1.make_rays(rs, n, 92, sx=0.3, sz=0.1, sa=5e-5, sc=1e-5, E=(9500., 10500.),
                        amplitudes=True, pol='mixed')
    beam.state[2] = 3
    from .gen_fixtures_softi import ellipse_surface, SURF_KEYS_ELL
    surf = ellipse_surface(em)
    more = {'surf_' + k: np.array(surf[k]) for k in SURF_KEYS_ELL}
    run('g2_ellipse_multilayer', em, surf, 'mosi_graded', beam,
        surf_isCylindrical=np.array(1.), **more)

    # transmission through W/B4C on a thin Si membrane: directions unchanged
    bl = raycing.BeamLine()
    ml = ref_stack(rm, 'wb4c_tran')
    thB = float(ml.get_Bragg_angle(8000.))
    oe = roe.OE(bl, 'mlt', center=[0, 1000., 0], pitch=thB, material=ml,
                limPhysX=[-5, 5], limPhysY=[-60, 60])
    beam = g1.make_rays(rs, n, 93, sx=0.5, sz=0.2, sa=1e-4, sc=6e-4, E=(7900., 8100.),
                        amplitudes=True, pol='mixed')
    run('g2_multilayer_tran', oe, dict(kind='flat'), 'wb4c_tran', beam)

    # toroid with a rhodium coating on silicon
    bl = raycing.BeamLine()
    ml = ref_stack(rm, 'rh_coated')
    p_, q_, pitch = 20000., 10000., 3e-3
    tm = roe.ToroidMirror(bl, 'tm', center=[0, p_, 0], pitch=pitch, R=(p_, q_), r=(p_, q_),
                          material=ml, limPhysX=[-10, 10], limPhysY=[-300, 300])
    beam = g1.make_rays(rs, n, 94, amplitudes=True, pol='mixed', E=(5000., 24000.))
    beam.z[0] = 2.5
    beam.c[0] = 0.
    beam.state[1] = 2
    run('g2_coated_toroid', tm, dict(kind='toroid', R=tm.R, r=tm.r), 'rh_coated', beam,
        surf_R=np.array(tm.R), surf_r=np.array(tm.r))


if __name__ == '__main__':
    main()
