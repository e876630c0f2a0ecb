"""CPU: the numpy restatement of the ray-surface hot path (oracle/reflect_np.py,
oracle/materials_np.py) against golden vectors produced by the imported
reference (oracle/gen_fixtures_p1.py)."""
import os

import numpy as np
import pytest

from oracle import fixture_io, materials_np as mn, reflect_np as rn

BEAM_TOL = 1e-13


def check_beam(mine, g, prefix):
    for f in mine.fields():
        if prefix + f not in g.files:
            continue
        m = getattr(mine, f)
        r = g[prefix + f]
        if f == 'state':
            assert np.array_equal(m, r), f
        else:
            scale = max(np.abs(r).max(), 1e-300)
            assert np.abs(m - r).max() <= BEAM_TOL * scale, (prefix, f)


@pytest.mark.parametrize('name', ['g2_toroid_pt', 'g2_flat_general',
                                  'g2_toroid_brent', 'g2_bentflat_rh',
                                  'g2_blazed_au', 'g2_ellipse_cyl',
                                  'g2_ellipse_full', 'g2_grating_vls',
                                  'g2_grating_const', 'g2_parabola_q',
                                  'g2_parabola_p_cyl', 'g2_hyperbola', 'g2_polygon',
                                  'g2_cone_rh', 'g2_capillary_parab', 'g2_capillary_ellipse',
                                  'g2_capillary_hyperbola', 'g3_laue_plate', 'g3_laue_plate_asym',
                                  'g3_laue_plate_transmitted', 'g2_multilayer_flat',
                                  'g2_ellipse_multilayer', 'g2_multilayer_tran',
                                  'g2_coated_toroid', 'g3_bent_johann_cyl',
                                  'g3_bent_johann_parab_asym', 'g3_bent_johansson_cyl',
                                  'g3_bent_johann_tor', 'g3_bent_johann_tor_asym',
                                  'g3_bent_johansson_tor', 'g3_bent_general_tor',
                                  'g3_bent_laue_cyl', 'g3_bent_laue_cyl_circ_asym',
                                  'g3_bent_laue_ground', 'g3_bent_laue_sphere',
                                  'g3_bent_laue_paraboloid', 'g3_bent_laue_2d', 'g3_diced_flat',
                                  'g3_diced_johann_tor', 'g3_diced_johansson_tor',
                                  'g3_cell_quartz_flat', 'g3_cell_graphite_johann',
                                  'g2_support_vcm', 'g2_support_vfm', 'g2_support_dualvfm'])
def test_oe_reflect_matches_reference(name):
    p, beam, g = fixture_io.load_case(name)
    info = {}
    gb, lb = rn.oe_reflect(p, beam, info=info)
    check_beam(gb, g, 'gb_')
    check_beam(lb, g, 'lb_')
    assert np.allclose(lb.theta, g['lb_theta'], rtol=0, atol=1e-15)
    assert bool(g['brent']) == info['brent']
    assert int(g['numit']) == info['numit']
    assert int(g['axis']) == info['axis']
    good = g['in_state'] > 0
    assert np.array_equal(info['tMin'][good], g['tMin'][good])
    assert np.array_equal(info['tMax'][good], g['tMax'][good])


@pytest.mark.parametrize('name', ['g2_lens_crl3', 'g2_lens_cyl2', 'g2_lens_single'])
def test_lens_stacks_match_reference(name):
    """Refractive lenses / CRL stacks (oes/refractive.py:237-663): paraboloid and
    parabolic-cylinder faces with their flat rim, multiple_refract's walk."""
    p, beam, g = fixture_io.load_case(name)
    gb, lo1, lo2 = rn.lens_multiple_refract(p, beam)
    check_beam(gb, g, 'gb_')
    check_beam(lo1, g, 'lo1_')
    check_beam(lo2, g, 'lo2_')


@pytest.mark.parametrize('name', ['g2_fzp_first', 'g2_fzp_orders', 'g2_gfzp_normal',
                                  'g2_gfzp_grazing'])
def test_zone_plate_matches_reference(name):
    """NormalFZP in ray mode (gratings.py:10-137): opaque zones absorb, the others
    deflect by the local zone density; one order or a seeded draw per ray."""
    p, beam, g = fixture_io.load_case(name)
    if 'np_seed' in g.files:
        np.random.seed(int(g['np_seed']))
    gb, lb = rn.oe_reflect(p, beam)
    check_beam(gb, g, 'gb_')
    check_beam(lb, g, 'lb_')
    assert int(g['axis']) == (1 if name == 'g2_gfzp_grazing' else 2)
    if 'gfzp' in p:
        assert p['gfzp']['minHalfLambda'] is not None
    if 'np_seed' in g.files:
        assert np.array_equal(lb.order, g['lb_order'])


@pytest.mark.parametrize('name', ['g2_grating_orders', 'g2_grating_efficiency',
                                  'g2_grating_efffile'])
def test_random_diffraction_orders_follow_the_references_draw(name):
    """order=(1, -1, 2, 0): one order per hit ray from numpy's global generator
    (reflect.py:455-458); with the reference's seed the oracle draws the same. With a
    table of efficiencies per order (material.py:391-413) the amplitudes are its square
    roots, zero for an order the table does not list; with an efficiency FILE the table is
    a function of energy (np.interp, :403-410)."""
    p, beam, g = fixture_io.load_case(name)
    np.random.seed(int(g['np_seed']))
    gb, lb = rn.oe_reflect(p, beam)
    check_beam(gb, g, 'gb_')
    check_beam(lb, g, 'lb_')
    assert np.array_equal(lb.order, g['lb_order'])
    hit = g['lb_state'] == 1
    assert set(np.unique(lb.order[hit])) == {-1., 0., 1., 2.} and not lb.order[~hit].any()


def test_polygon_outline_states_are_the_references():
    """The oracle's point-in-polygon is matplotlib's: the reference's rays_good on
    vertices, edge points and points level with vertices (golden file)."""
    p, _, g = fixture_io.load_case('g2_polygon')
    assert np.array_equal(rn.rays_good(p, g['pip_x'], g['pip_y']), g['pip_state'])


def test_parametric_mirror_without_intersection_search():
    """reflect(noIntersectionSearch=True) on an elliptical (parametric) mirror:
    the points are converted to (s, phi, r) and back (reflect.py:679-682,
    1066-1071)."""
    p, beam, g = fixture_io.load_case('g2_ellipse_cyl_nis')
    gb, lb = rn.oe_reflect(p, beam, noIntersectionSearch=True)
    check_beam(gb, g, 'gb_')
    check_beam(lb, g, 'lb_')
    assert (lb.state == 1).all()


@pytest.mark.parametrize('name', ['g3_dcm_si111', 'g3_dcm_si111_asym', 'g3_dcm_sagittal'])
def test_dcm_double_reflect_matches_reference(name):
    p, beam, g = fixture_io.load_case(name)
    gb2, lo1, lo2 = rn.dcm_double_reflect(p, beam)
    check_beam(gb2, g, 'gb_')
    check_beam(lo1, g, 'lo1_')
    check_beam(lo2, g, 'lo2_')


def test_plate_double_refract_matches_reference():
    p, beam, g = fixture_io.load_case('g2_plate_be')
    gb2, lo1, lo2 = rn.dcm_double_reflect(p, beam, fromVacuum1=True,
                                          fromVacuum2=False, is_plate=True)
    check_beam(gb2, g, 'gb_')
    check_beam(lo1, g, 'lo1_')
    check_beam(lo2, g, 'lo2_')
    T = (gb2.Jss + gb2.Jpp)[gb2.state == 1]
    assert 0.5 < T.mean() < 1.0 and T.max() <= 1.0      # an absorbing Be window


def test_edge_rays_are_present_in_toroid_fixture():
    """The fixture must exercise lost / over / out / untouched rays."""
    _, _, g = fixture_io.load_case('g2_toroid_pt')
    st = set(np.unique(g['lb_state']).tolist())
    assert {1, 3, -1, 0} <= st
    assert g['gb_state'][7] == -1 and g['gb_x'][7] == g['in_x'][7]


def test_material_grid(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g5_material_grid.npz'))
    tb = fixture_io.tables()
    el = lambda s: mn.load_element(tb, s)  # noqa: E731
    mats = dict(
        Pt=mn.make_material([el('Pt')], None, 'mirror', 21.45),
        Rh=mn.make_material([el('Rh')], None, 'mirror', 12.41),
        Si=mn.make_material([el('Si')], None, 'mirror', 2.33),
        SiO2=mn.make_material([el('Si'), el('O')], [1, 2], 'mirror', 2.2),
        PtThin=mn.make_material([el('Pt')], None, 'thin mirror', 21.45, 30e-6),
        SiPlate=mn.make_material([el('Si')], None, 'plate', 2.33))
    for name, m in mats.items():
        assert np.allclose(mn.refractive_index(m, g['Egrid']), g[name + '_n'],
                           rtol=1e-15, atol=0)
        for d, fv in (('in', True), ('out', False)):
            key = '%s_%s' % (name, d)
            if key + '_rs' not in g.files:
                continue
            res = mn.material_amplitude(m, g['E'].copy(), g[key + '_bdn'].copy(), fv)
            for i, lab in enumerate(('rs', 'rp', 'mu', 'nk')):
                assert np.allclose(res[i], g[key + '_' + lab], rtol=1e-14, atol=0)


def test_rocking_curves(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g3_rocking_curves.npz'))
    si = mn.load_element(fixture_io.tables(), 'Si')
    keys = sorted(k[:-3] for k in g.files if k.endswith('_in'))
    assert len(keys) == 2 * 3 * (1 + 4 * 2)
    for key in keys:
        hkl = tuple(int(c) for c in key[2:5])
        geom = 'Bragg' if 'Bragg' in key else 'Laue'
        geom += ' transmitted' if 'transmitted' in key else ' reflected'
        d, V, chiToF, t = g[key + '_par']
        cr = mn.make_crystal(si, hkl, d, 'diamond', geom,
                             None if np.isnan(t) else t, 1., V)
        E, g0, gh, hns = g[key + '_in']
        S, P = mn.crystal_amplitude(cr, E.copy(), g0.copy(), gh.copy(), hns.copy())
        for mine, ref in ((S, g[key + '_S']), (P, g[key + '_P'])):
            fin = np.isfinite(ref)
            assert np.array_equal(fin, np.isfinite(mine))
            assert np.abs(mine[fin] - ref[fin]).max() <= 1e-12 * np.abs(ref[fin]).max()


def test_thick_bragg_peak_reflectivity_is_physical(golden_dir):
    """Loose known-answer pin (the reference's XOP curves pin |R|^2 only at
    percent level, tests/raycing/test_materials.py:239-349): Si(111) at 9 keV,
    symmetric thick Bragg: peak |R_s|^2 ~ 0.95, Darwin width ~ 25-35 urad."""
    g = np.load(os.path.join(golden_dir, 'g3_rocking_curves.npz'))
    S = g['Si111_Braggreflected_thick_+0_S']
    E, g0, gh, hns = g['Si111_Braggreflected_thick_+0_in']
    R = np.abs(S)**2
    assert 0.9 < R.max() < 1.0
    theta = -np.arcsin(hns)
    width = np.ptp(theta[R > 0.5 * R.max()])
    assert 20e-6 < width < 40e-6


def test_multilayer_amplitudes_match_reference():
    """Multilayer / Coated .get_amplitude (materials/multilayer.py:257-566): Parratt's
    recursion on periodic, depth-graded, transmitted, vacuum-spaced and single-coating
    stacks."""
    from oracle import gen_fixtures_multilayer as gm
    from oracle import materials_np as mn
    g = np.load(os.path.join(fixture_io.GOLDEN, 'g5_multilayer_amplitudes.npz'))
    tb = gm.all_tables()
    for name in gm.STACKS:
        s, p = mn.multilayer_amplitude(gm.oracle_stack(tb, name), g[name + '_E'],
                                       g[name + '_bdn'])
        for mine, ref in ((s, g[name + '_s']), (p, g[name + '_p'])):
            assert np.abs(mine - ref).max() <= 1e-13 * np.abs(ref).max(), name


def _cell_curve_keys(g):
    for key in sorted(k[:-3] for k in g.files if k.endswith('_in')):
        name, geom, thick, alpha = key.rsplit('_', 3)
        geom = geom.replace('Bragg', 'Bragg ').replace('Laue', 'Laue ')
        yield key, name, geom, None if thick == 'thick' else float(thick[:-2]) * 1e-3


def test_cell_crystal_rocking_curves(golden_dir):
    """CrystalFromCell (crystals_basic.py:157-440): structure factor summed over the atoms
    of a hexagonal two-element cell, partial occupancies, Debye-Waller factor."""
    from oracle import gen_fixtures_cell as gc
    g = np.load(os.path.join(golden_dir, 'g3_cell_rocking_curves.npz'))
    tb = gc.all_tables()
    count = 0
    for key, name, geom, t in _cell_curve_keys(g):
        cr = gc.oracle_cell(tb, name, geom=geom, t=t)
        E, g0, gh, hns = g[key + '_in']
        S, P = mn.crystal_amplitude(cr, E.copy(), g0.copy(), gh.copy(), hns.copy())
        for mine, ref in ((S, g[key + '_S']), (P, g[key + '_P'])):
            assert np.abs(mine - ref).max() <= 1e-12 * np.abs(ref).max(), key
        count += 1
    assert count == 3 * 5 * 2


def test_fixed_refractive_index(golden_dir):
    """Material(refractiveIndex = a number) (material.py:240-262, 364-373): Fresnel
    amplitudes at visible-light energies, and a glass plate traversed by rays."""
    from oracle.gen_fixtures_index import INDEX, oracle_material
    g = np.load(os.path.join(golden_dir, 'g5_fixed_index.npz'))
    for name in INDEX:
        for kind, t in (('mirror', None), ('thin mirror', 2e-4), ('plate', None)):
            for fv in ((True, False) if kind == 'plate' else (True,)):
                key = '%s_%s_%d' % (name, kind.replace(' ', ''), fv)
                res = mn.material_amplitude(oracle_material(name, kind, t), g['E'].copy(),
                                            g[key + '_bdn'].copy(), fv)
                for i, lab in enumerate(('rs', 'rp', 'mu', 'nk')):
                    assert np.allclose(res[i], g[key + '_' + lab], rtol=1e-13, atol=0), key
    p, beam, g = fixture_io.load_case('g2_plate_glass')
    gb2, lo1, lo2 = rn.dcm_double_reflect(p, beam, fromVacuum1=True, fromVacuum2=False,
                                          is_plate=True)
    check_beam(gb2, g, 'gb_')
    check_beam(lo1, g, 'lo1_')
    check_beam(lo2, g, 'lo2_')


def test_energy_outside_the_efficiency_file_is_an_error():
    """material.py:399-407: the reference raises rather than extrapolate."""
    p, beam, g = fixture_io.load_case('g2_grating_efffile')
    beam.E[100] = 400.
    np.random.seed(int(g['np_seed']))
    with pytest.raises(ValueError, match='out of the efficiency table range'):
        rn.oe_reflect(p, beam)
