"""TEST INFRASTRUCTURE ONLY — turns the committed golden files
(tests/golden/g2_*.npz, g3_dcm_*.npz) back into oracle parameter dictionaries and
Beam records. The configurations mirror oracle/gen_fixtures_p1.py."""
import os

import numpy as np

from . import materials_np as mn
from . import reflect_np as rn

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                      'tests', 'golden')


def tables():
    return np.load(os.path.join(GOLDEN, 'g6_element_tables.npz'))


def _unflatten(g):
    p = {}
    for key in g.files:
        if not key.startswith('oe_'):
            continue
        v = g[key]
        name = key[3:]
        if v.dtype.kind in 'US':
            p[name] = str(v)
        elif v.ndim == 0:
            p[name] = None if np.isnan(v) else float(v)
        else:
            p[name] = [float(t) for t in v]
    p['azimuth_sc'] = tuple(p['azimuth_sc'])
    p['lostNum'] = int(p['lostNum'])
    if 'invertNormal' in p:
        p['invertNormal'] = int(p['invertNormal'])
    return p


def load_case(name):
    """-> (params, beam_in, golden npz)."""
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    tb = tables()
    p = _unflatten(g)
    if name == 'g2_polygon':
        p['surface'] = dict(kind='flat')
        p['shape'] = [list(v) for v in g['polygon']]
        p['material'] = mn.make_material([mn.load_element(tb, 'Pt')], None,
                                         'mirror', float(g['mat_rho']))
    elif name == 'g2_toroid_pt':
        p['surface'] = dict(kind='toroid', R=float(g['surf_R']), r=float(g['surf_r']))
        p['material'] = mn.make_material([mn.load_element(tb, 'Pt')], None,
                                         'mirror', float(g['mat_rho']))
    elif name == 'g2_flat_general':
        p['surface'] = dict(kind='flat')
        p['material'] = mn.make_material([mn.load_element(tb, 'Rh')], None,
                                         'thin mirror', float(g['mat_rho']),
                                         float(g['mat_t']))
    elif name == 'g2_toroid_brent':
        p['surface'] = dict(kind='toroid', R=float(g['surf_R']), r=float(g['surf_r']))
        p['material'] = None
    elif name == 'g2_bentflat_rh':
        p['surface'] = dict(kind='bentflat', R=float(g['surf_R']), y0=p['surfPhysY'][0])
        p['material'] = mn.make_material([mn.load_element(tb, 'Rh')], None,
                                         'mirror', float(g['mat_rho']))
    elif name.startswith('g2_grating'):
        p['surface'] = dict(kind='flat')
        p['material'] = mn.make_material([mn.load_element(tb, 'Au')], None,
                                         'grating', float(g['mat_rho']))
        p['order'] = int(g['order']) if g['order'].ndim == 0 else \
            tuple(int(o) for o in g['order'])
        if 'efficiency' in g.files:
            p['material']['efficiency'] = [[int(o), float(v)] for o, v in g['efficiency']]
        if 'eff_E' in g.files:            # efficiencyFile: [order, row of the table]
            p['material']['efficiency_table'] = (g['eff_E'], g['eff_I'])
        if 'gd_axis' in g.files:
            p['gratingDensity'] = [str(g['gd_axis'])] + \
                [float(v) for v in g['gd_coeffs']]
        else:
            p['gVector'] = tuple(float(v) for v in g['g_vector'])
    elif name == 'g2_blazed_au':
        p['surface'] = rn.make_blazed(float(g['surf_blaze']), float(g['surf_rho']),
                                      float(g['surf_antiblaze']))
        p['material'] = mn.make_material([mn.load_element(tb, 'Au')], None,
                                         'mirror', float(g['mat_rho']))
    elif name.startswith('g2_parabola') or name == 'g2_hyperbola':
        keys = ('cosGamma', 'sinGamma', 'y0', 'z0') + (
            ('parabParam',) if 'parabola' in name else ('hyperbolaA', 'hyperbolaB'))
        p['surface'] = dict(
            kind='parabola_param' if 'parabola' in name else 'hyperbola_param',
            isClosed=False, isCylindrical=bool(float(g['surf_isCylindrical'])),
            **{k: float(g['surf_' + k]) for k in keys})
        p['material'] = mn.make_material([mn.load_element(tb, 'Au')], None,
                                         'mirror', float(g['mat_rho']))
    elif name in ('g2_multilayer_flat', 'g2_multilayer_tran', 'g2_coated_toroid'):
        from . import gen_fixtures_multilayer as gm
        p['surface'] = dict(kind='toroid', R=float(g['surf_R']), r=float(g['surf_r'])) \
            if 'surf_R' in g.files else dict(kind='flat')
        p['material'] = gm.oracle_stack(gm.all_tables(), str(g['stack']))
    elif name.startswith('g2_ellipse'):
        p['surface'] = dict(
            kind='ellipse_param', isClosed=False,
            isCylindrical=bool(float(g['surf_isCylindrical'])),
            **{k: float(g['surf_' + k]) for k in
               ('p', 'q', 'cosGamma', 'sinGamma', 'y0', 'z0', 'ellipseA',
                'ellipseB')})
        if 'stack' in g.files:
            from . import gen_fixtures_multilayer as gm
            p['material'] = gm.oracle_stack(gm.all_tables(), str(g['stack']))
        else:
            p['material'] = mn.make_material([mn.load_element(tb, 'Au')], None,
                                             'mirror', float(g['mat_rho']))
    elif name == 'g2_plate_be':
        p['surface'] = dict(kind='flat')
        p['surface2'] = dict(kind='flat')
        be = dict(name='Be', Z=int(g['Be_Z']), mass=float(g['Be_mass']),
                  f0coeffs=np.array(g['Be_f0']), E=np.array(g['Be_E']),
                  f1=np.array(g['Be_f1']), f2=np.array(g['Be_f2']))
        p['material'] = mn.make_material([be], None, 'plate', float(g['mat_rho']))
        p['material2'] = p['material']
    elif name == 'g2_plate_glass':
        from . import gen_fixtures_index as gi
        p['surface'] = dict(kind='flat')
        p['surface2'] = dict(kind='flat')
        p['material'] = p['material2'] = gi.oracle_material('glass', 'plate')
    elif name.startswith('g2_gfzp'):
        from .gen_fixtures_fzp import GENERAL
        from .consts import CH
        kw = GENERAL[name]
        p['surface'] = dict(kind='flat')
        p['material'] = dict(kind='FZP')
        p['order'] = 1
        p['gfzp'] = dict(f1=kw['f1'], f2=kw['f2'], lambdaE=CH / kw['E'] * 1e-7, N=kw['N'],
                         phaseShift=float(g['gfzp_phaseShift']), vorticity=0,
                         grazingAngle=kw['pitch'], minHalfLambda=None)
    elif name.startswith('g2_fzp'):
        p['surface'] = dict(kind='flat')
        p['material'] = dict(kind='FZP')
        p['order'] = int(g['order']) if g['order'].ndim == 0 else \
            tuple(int(o) for o in g['order'])
        p['fzp'] = dict(zones=np.arange(len(g['fzp_rn'])), rn=np.array(g['fzp_rn']),
                        black=bool(g['fzp_black']))
    elif name.startswith('g2_lens'):
        zmax = None if np.isnan(g['lens_zmax']) else float(g['lens_zmax'])
        kind = str(g['lens_class'])
        p['surface'] = p['surface2'] = dict(kind='paraboloid', focus=float(g['lens_focus']),
                                            zmax=zmax, cylinder='Cylinder' in kind)
        be = dict(name='Be', Z=int(g['Be_Z']), mass=float(g['Be_mass']),
                  f0coeffs=np.array(g['Be_f0']), E=np.array(g['Be_E']),
                  f1=np.array(g['Be_f1']), f2=np.array(g['Be_f2']))
        p['material'] = p['material2'] = mn.make_material([be], None, 'lens',
                                                          float(g['mat_rho']))
        p.update(nCRL=int(g['lens_nCRL']), zmax=zmax, t=float(g['lens_t']),
                 double_sided=kind.startswith('Double'))
    elif name.startswith('g2_capillary'):
        shape = name.split('_')[-1]
        if shape == 'parab':
            q, r0 = float(g['cap_q']), float(g['cap_r0'])
            focus = -0.5*(q-(q**2+r0**2)**0.5)
            p['surface'] = dict(kind='parab_capillary', s0=focus + q, focus=focus)
        else:
            A, B = float(g['cap_%sA' % shape]), float(g['cap_%sB' % shape])
            half = 0.5*np.abs(p['surfPhysY'][-1]-p['surfPhysY'][0])
            wd = float(g['cap_workingDistance'])
            ctd = (A**2 - B**2)**0.5 - wd - half if shape == 'ellipse' else \
                (A**2 + B**2)**0.5 + wd + half
            p['surface'] = {'kind': shape + '_capillary', shape + 'A': A, shape + 'B': B,
                            'ctd': ctd}
        p['material'] = mn.make_material([mn.load_element(tb, 'Au')], None,
                                         'mirror', float(g['mat_rho']))
    elif name.startswith('g2_multi_'):        # OE.multiple_reflect (gen_fixtures_multi.py)
        import sys
        sys.path.insert(0, os.path.dirname(GOLDEN))
        import multi_cases as case
        if name == 'g2_multi_cylinder':
            p['surface'] = dict(kind='user', z=case.numpy_cyl_z, n=case.numpy_cyl_n)
        elif name == 'g2_multi_flat':
            p['surface'] = dict(kind='flat')
        elif name == 'g2_multi_capillary':
            p['surface'] = dict(kind='ellipse_capillary', ellipseA=case.CAPILLARY['ellipseA'],
                                ellipseB=case.CAPILLARY['ellipseB'], ctd=float(g['cap_ctd']))
        else:
            p['surface'] = dict(kind='toroid', R=float(g['surf_R']), r=float(g['surf_r']))
        p['material'] = mn.make_material(
            [mn.load_element(tb, 'Pt' if name == 'g2_multi_edges' else 'Au')], None, 'mirror',
            float(g['mat_rho']))
    elif name == 'g2_cone_rh':
        p['surface'] = rn.make_cone(float(g['surf_L0']), float(g['surf_theta']))
        p['material'] = mn.make_material([mn.load_element(tb, 'Rh')], None,
                                         'mirror', float(g['mat_rho']))
    elif name.startswith('g3_diced_'):
        cls = str(g['surf_class'])
        alpha = float(g['surf_alpha'])
        surf = dict(kind='diced', base='flat' if cls == 'DicedOE' else 'toroid',
                    planes='johansson' if 'Johansson' in cls else 'johann',
                    alpha=alpha if alpha else None, crossSection='circular')
        for key in ('Rm', 'Rs', 'RmBragg', 'RsBragg', 'xStep', 'yStep', 'dxFacet', 'dyFacet'):
            surf[key] = float(g['surf_' + key])
        p['surface'] = surf
        si = mn.load_element(tb, 'Si')
        p['material'] = mn.make_crystal(si, (1, 1, 1), float(g['cr_d']), 'diamond',
                                        'Bragg reflected', None, 1., float(g['cr_V']))
    elif name.startswith('g3_bent_laue'):
        cls = str(g['surf_class'])
        alpha = float(g['surf_alpha'])
        p['surface'] = dict(kind='laue_2d' if cls == 'BentLaue2D' else
                            'laue_sphere' if 'Sphere' in cls else 'bent_cylinder',
                            Rm=float(g['surf_Rm']), Rs=float(g['surf_Rs']),
                            alpha=alpha if alpha else None,
                            planes='laue_ground' if 'Ground' in cls else 'laue',
                            crossSection=str(g['surf_crossSection']))
        si = mn.load_element(tb, 'Si')
        p['material'] = mn.make_crystal(si, (1, 1, 1), float(g['cr_d']), 'diamond',
                                        'Laue reflected', float(g['cr_t']), 1., float(g['cr_V']))
    elif name.startswith('g3_bent_'):
        cls = str(g['surf_class'])
        planes = 'general' if 'General' in cls else 'johansson' if 'Johansson' in cls \
            else 'johann'
        alpha = float(g['surf_alpha'])
        surf = dict(kind='bent_toroid' if 'Toroid' in cls else 'bent_cylinder',
                    Rm=float(g['surf_Rm']), planes=planes, alpha=alpha if alpha else None,
                    crossSection=str(g['surf_crossSection']))
        for key in ('Rs', 'RmBragg', 'RsBragg'):
            if 'surf_' + key in g.files:
                surf[key] = float(g['surf_' + key])
        p['surface'] = surf
        si = mn.load_element(tb, 'Si')
        p['material'] = mn.make_crystal(si, (1, 1, 1), float(g['cr_d']), 'diamond',
                                        'Bragg reflected', None, 1., float(g['cr_V']))
        assert p['material']['chiToF'] == float(g['cr_chiToF'])
    elif name.startswith('g3_cell_'):
        from . import gen_fixtures_cell as gc
        if 'surf_Rm' in g.files:
            p['surface'] = dict(kind='bent_cylinder', Rm=float(g['surf_Rm']), planes='johann',
                                alpha=None, crossSection='circular')
        else:
            p['surface'] = dict(kind='flat', alpha=float(g['alpha']))
        p['material'] = gc.oracle_cell(gc.all_tables(), str(g['cell']))
    elif name.startswith('g2_support_'):
        from . import gen_fixtures_supports as gs
        case = name[len('g2_support_'):]
        cls, kw, stripes, select = gs.CASES[case]
        y0 = kw['limPhysY'][0]
        if case == 'vcm':
            p['surface'] = dict(kind='bentflat', R=kw['R'], y0=y0)
        elif case == 'vfm':
            p['surface'] = dict(kind='vfm', r=kw['r'], R=kw['R'], y0=y0,
                                limOptX=list(kw['limOptX']))
        else:
            p['surface'] = dict(kind='dualvfm', R=kw['R'], y0=y0, r1=70.0, xCylinder1=23.5,
                                hCylinder1=3.7035, r2=35.98, xCylinder2=-25.0,
                                hCylinder2=6.9504)
        els, q, rho = gs.STRIPES[str(g['stripe'])]
        p['material'] = mn.make_material([mn.load_element(tb, e) for e in els], list(q),
                                         'mirror', rho)
    elif name.startswith('g3_laue_plate'):
        alpha = float(g['alpha'])
        p['surface'] = dict(kind='flat', laue=True, alpha=alpha if alpha else None)
        si = mn.load_element(tb, 'Si')
        p['material'] = mn.make_crystal(si, (1, 1, 1), float(g['cr_d']), 'diamond',
                                        str(g['cr_geom']), float(g['cr_t']), 1.,
                                        float(g['cr_V']))
        assert p['material']['chiToF'] == float(g['cr_chiToF'])
    elif name.startswith('g3_dcm'):
        alpha = float(g['alpha'])
        p['surface'] = dict(kind='flat', alpha=alpha)
        p['surface2'] = dict(kind='sagittal', Rs=float(g['Rs'])) if 'Rs' in g.files else \
            dict(kind='flat', alpha=alpha, flip_n_y=True)
        si = mn.load_element(tb, 'Si')
        for key in ('material', 'material2'):
            p[key] = mn.make_crystal(si, (1, 1, 1), float(g['cr_d']), 'diamond',
                                     'Bragg reflected', None, 1., float(g['cr_V']))
            assert p[key]['chiToF'] == float(g['cr_chiToF'])
    else:
        raise KeyError(name)
    return p, rn.Beam.from_dict(g, 'in_'), g
