"""Test helper: builds xrt_amd (product) optical elements and beams for the
golden P1 configurations (mirrors oracle/gen_fixtures_p1.py and
oracle/fixture_io.py, which build the reference / oracle versions)."""
import os

import numpy as np

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.materials as rm
import xrt_amd.backends.raycing.oes as roe
import xrt_amd.backends.raycing.sources as rs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def product_beam(src, prefix='in_'):
    """Beam from a golden npz (or any mapping of arrays)."""
    n = len(src[prefix + 'x'])
    b = rs.Beam(nrays=n, withAmplitudes=(prefix + 'Es') in src)
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp',
              'state'):
        setattr(b, f, np.array(src[prefix + f]))
    if (prefix + 'Es') in src:
        b.Es = np.array(src[prefix + 'Es'])
        b.Ep = np.array(src[prefix + 'Ep'])
    return b


def beam_from_oracle(ob):
    b = rs.Beam(nrays=len(ob.x), withAmplitudes=hasattr(ob, 'Es'))
    for f in ob.fields():
        setattr(b, f, np.array(getattr(ob, f)))
    return b


def _pad_ordinal(bl, lostNum):
    """The element's ordinal decides the 'lost' state value."""
    class _Dummy(object):
        pass
    while len(bl.oes) < -int(lostNum) - 1:
        bl.oes.append(_Dummy())


def _az(g):
    s, c = [float(v) for v in g['oe_azimuth_sc']]
    return float(np.arctan2(s, c))


def _opt(g, key):
    v = g[key]
    return None if v.ndim == 0 else [float(t) for t in v]


def product_stack(name):
    """The xrt_amd Multilayer / Coated of the stack *name* of
    oracle/gen_fixtures_multilayer.py."""
    from oracle.gen_fixtures_multilayer import COMPOUNDS, STACKS
    kw = dict(STACKS[name])
    for key in ('tLayer', 'bLayer', 'substrate', 'coating'):
        if kw.get(key) is not None:
            els, q, rho = COMPOUNDS[kw[key]]
            kw[key] = rm.Material(els, quantities=q, rho=rho)
    return (rm.Coated if 'coating' in kw else rm.Multilayer)(**kw)


def product_cell(name, **over):
    """The xrt_amd CrystalFromCell of the cell *name* of oracle/gen_fixtures_cell.py."""
    from oracle.gen_fixtures_cell import CELLS
    kw = dict(CELLS[name])
    kw.update(over)
    return rm.CrystalFromCell(name, **kw)


def product_oe(name, g):
    """-> the xrt_amd optical element for golden case *name*."""
    bl = raycing.BeamLine(azimuth=_az(g))
    # use the exact sin/cos the fixture was generated with
    bl.sinAzimuth, bl.cosAzimuth = [float(v) for v in g['oe_azimuth_sc']]
    _pad_ordinal(bl, g['oe_lostNum'])
    common = dict(
        center=[float(v) for v in g['oe_center']], pitch=float(g['oe_pitch']),
        roll=float(g['oe_roll']), yaw=float(g['oe_yaw']),
        positionRoll=float(g['oe_positionRoll']),
        rotationSequence=str(g['oe_rotationSequence']),
        extraPitch=float(g['oe_extraPitch']), extraRoll=float(g['oe_extraRoll']),
        extraYaw=float(g['oe_extraYaw']),
        extraRotationSequence=str(g['oe_extraRotationSequence']),
        limPhysX=_opt(g, 'oe_surfPhysX'), limPhysY=_opt(g, 'oe_surfPhysY'),
        limOptX=_opt(g, 'oe_surfOptX'), limOptY=_opt(g, 'oe_surfOptY'),
        shape=str(g['oe_shape']), overEdge=str(g['oe_overEdge']))
    if name == 'g2_polygon':
        common['shape'] = [tuple(v) for v in g['polygon']]
        m = rm.Material('Pt', rho=float(g['mat_rho']), kind='mirror')
        oe = roe.OE(bl, 'poly', material=m, **common)
    elif name == 'g2_toroid_pt':
        m = rm.Material('Pt', rho=float(g['mat_rho']), kind='mirror')
        oe = roe.ToroidMirror(bl, 'tm', R=float(g['surf_R']), r=float(g['surf_r']),
                              material=m, **common)
    elif name == 'g2_flat_general':
        m = rm.Material('Rh', rho=float(g['mat_rho']), kind='thin mirror',
                        t=float(g['mat_t']))
        oe = roe.OE(bl, 'fm', material=m, **common)
    elif name == 'g2_toroid_brent':
        oe = roe.ToroidMirror(bl, 'tm2', R=float(g['surf_R']), r=float(g['surf_r']),
                              material=None, **common)
    elif name == 'g2_bentflat_rh':
        m = rm.Material('Rh', rho=float(g['mat_rho']), kind='mirror')
        oe = roe.BentFlatMirror(bl, 'vcm', R=float(g['surf_R']), material=m, **common)
    elif name.startswith('g2_grating'):
        eff = [[int(o), float(v)] for o, v in g['efficiency']] if 'efficiency' in g.files \
            else None
        if 'eff_E' in g.files:            # the columns of the data file next to the golden
            eff = [[int(o), int(c)] for o, c in g['efficiency']]
            m = rm.Material('Au', rho=float(g['mat_rho']), kind='grating', efficiency=eff,
                            efficiencyFile=os.path.join(GOLDEN, name + '.txt'))
        else:
            m = rm.Material('Au', rho=float(g['mat_rho']), kind='grating', efficiency=eff)
        if 'gd_axis' in g.files:
            oe = roe.OE(bl, 'gr', material=m,
                        order=int(g['order']) if g['order'].ndim == 0 else
                        [int(o) for o in g['order']],
                        gratingDensity=[str(g['gd_axis'])] +
                        [float(v) for v in g['gd_coeffs']], **common)
        else:
            gv = [float(v) for v in g['g_vector']]

            class ConstGrating(roe.OE):
                def local_g(self, x, y, rho=None):
                    return gv[0], gv[1], gv[2]
            oe = ConstGrating(bl, 'gr', material=m, order=int(g['order']), **common)
    elif name == 'g2_blazed_au':
        m = rm.Material('Au', rho=float(g['mat_rho']), kind='mirror')
        oe = roe.BlazedGrating(bl, 'pg', material=m, blaze=float(g['surf_blaze']),
                               antiblaze=float(g['surf_antiblaze']),
                               rho=float(g['surf_rho']), **common)
    elif name.startswith('g2_parabola') or name == 'g2_hyperbola':
        m = rm.Material('Au', rho=float(g['mat_rho']), kind='mirror')
        pq = {k: (None if np.isnan(float(g['surf_' + k])) else float(g['surf_' + k]))
              for k in ('p', 'q')}
        cls = roe.ParabolicalMirrorParam if 'parabola' in name else \
            roe.HyperbolicMirrorParam
        oe = cls(bl, 'm', material=m, isCylindrical=bool(float(g['surf_isCylindrical'])),
                 **pq, **common)
        keys = ('cosGamma', 'sinGamma', 'y0', 'z0') + (
            ('parabParam',) if 'parabola' in name else ('hyperbolaA', 'hyperbolaB'))
        for k in keys:
            assert abs(getattr(oe, k) - float(g['surf_' + k])) <= \
                1e-15 * max(1., abs(float(g['surf_' + k]))), k
    elif name in ('g2_multilayer_flat', 'g2_multilayer_tran', 'g2_coated_toroid'):
        m = product_stack(str(g['stack']))
        if 'surf_R' in g.files:
            oe = roe.ToroidMirror(bl, 'tm', R=float(g['surf_R']), r=float(g['surf_r']),
                                  material=m, **common)
        else:
            oe = roe.OE(bl, 'ml', material=m, **common)
    elif name.startswith('g2_ellipse'):
        m = product_stack(str(g['stack'])) if 'stack' in g.files else \
            rm.Material('Au', rho=float(g['mat_rho']), kind='mirror')
        oe = roe.EllipticalMirrorParam(
            bl, 'm4', material=m, p=float(g['surf_p']), q=float(g['surf_q']),
            isCylindrical=bool(float(g['surf_isCylindrical'])), **common)
        for k in ('cosGamma', 'sinGamma', 'y0', 'z0', 'ellipseA', 'ellipseB'):
            assert abs(getattr(oe, k) - float(g['surf_' + k])) <= \
                1e-15 * max(1., abs(float(g['surf_' + k]))), k
    elif name == 'g2_plate_be':
        m = rm.Material('Be', rho=float(g['mat_rho']), kind='plate')
        oe = roe.Plate(bl, 'win', material=m, t=float(g['plate_t']), **common)
    elif name == 'g2_plate_glass':
        from oracle.gen_fixtures_index import INDEX
        m = rm.Material(kind='plate', refractiveIndex=INDEX['glass'])
        oe = roe.Plate(bl, 'win', material=m, t=float(g['plate_t']), **common)
    elif name.startswith('g2_gfzp'):
        from oracle.gen_fixtures_fzp import GENERAL
        kw = {k: v for k, v in GENERAL[name].items()
              if k in ('f1', 'f2', 'E', 'N', 'phaseShift')}
        oe = roe.GeneralFZPin0YZ(bl, 'gfzp', material=rm.Material('Au', rho=19.3, kind='FZP'),
                                 **kw, **common)
        assert oe.phaseShift == float(g['gfzp_phaseShift'])
    elif name.startswith('g2_fzp'):
        m = rm.Material('Au', rho=19.3, kind='FZP')
        for key in ('limPhysX', 'limPhysY'):        # the zone plate sets its own outline
            common.pop(key)
        kw = dict(thinnestZone=float(g['fzp_thinnestZone'])) if 'fzp_thinnestZone' in g.files \
            else dict(N=int(g['fzp_N']))
        oe = roe.NormalFZP(
            bl, 'fzp', material=m, f=float(g['fzp_f']), E=float(g['fzp_E']),
            isCentralZoneBlack=bool(g['fzp_black']),
            order=int(g['order']) if g['order'].ndim == 0 else [int(o) for o in g['order']],
            **kw, **common)
        assert np.array_equal(oe.rn, g['fzp_rn'])
    elif name.startswith('g2_lens'):
        m = rm.Material('Be', rho=float(g['mat_rho']), kind='lens')
        zmax = None if np.isnan(g['lens_zmax']) else float(g['lens_zmax'])
        oe = getattr(roe, str(g['lens_class']))(
            bl, 'crl', material=m, t=float(g['lens_t']), focus=float(g['lens_focus']),
            zmax=zmax, nCRL=int(g['lens_nCRL']), **common)
    elif name.startswith('g2_capillary'):
        m = rm.Material('Au', rho=float(g['mat_rho']), kind='mirror')
        cls = {'parab': roe.ParaboloidCapillaryMirror, 'ellipse': roe.EllipsoidCapillaryMirror,
               'hyperbola': roe.HyperboloidCapillaryMirror}[name.split('_')[-1]]
        kw = {k[4:]: float(g[k]) for k in g.files if k.startswith('cap_')}
        oe = cls(bl, 'cap', material=m, **kw, **common)
    elif name == 'g2_cone_rh':
        m = rm.Material('Rh', rho=float(g['mat_rho']), kind='mirror')
        oe = roe.ConicalMirror(bl, 'cone', L0=float(g['surf_L0']),
                               theta=float(g['surf_theta']), material=m, **common)
    elif name.startswith('g3_diced_'):
        si = rm.CrystalSi(hkl=(1, 1, 1))
        cls = str(g['surf_class'])
        alpha = float(g['surf_alpha'])
        dx, dy = float(g['surf_dxFacet']), float(g['surf_dyFacet'])
        kw = dict(dxFacet=dx, dyFacet=dy, dxGap=float(g['surf_xStep']) - dx,
                  dyGap=float(g['surf_yStep']) - dy)
        if cls != 'DicedOE':
            kw.update(Rm=float(g['surf_Rm']), Rs=float(g['surf_Rs']))
        oe = getattr(roe, cls)(bl, 'dc', material=si, alpha=alpha if alpha else None, **kw,
                               **common)
        assert oe.xStep == float(g['surf_xStep']) and oe.yStep == float(g['surf_yStep'])
    elif name.startswith('g3_bent_laue'):
        si = rm.CrystalSi(hkl=(1, 1, 1), geom='Laue reflected', t=float(g['cr_t']))
        assert si.d == float(g['cr_d']) and si.chiToF == float(g['cr_chiToF'])
        alpha = float(g['surf_alpha'])
        if str(g['surf_class']) == 'BentLaue2D':
            oe = roe.BentLaue2D(bl, 'bl', material=si, Rm=float(g['surf_Rm']),
                                Rs=float(g['surf_Rs']), alpha=alpha if alpha else None, **common)
        else:
            oe = getattr(roe, str(g['surf_class']))(
                bl, 'bl', material=si, R=float(g['surf_Rm']), alpha=alpha if alpha else None,
                crossSection=str(g['surf_crossSection']), **common)
    elif name.startswith('g3_bent_'):
        si = rm.CrystalSi(hkl=(1, 1, 1))
        assert si.d == float(g['cr_d']) and si.chiToF == float(g['cr_chiToF'])
        cls = str(g['surf_class'])
        kw = dict(Rm=float(g['surf_Rm']))
        if 'Cylinder' in cls:
            kw['crossSection'] = str(g['surf_crossSection'])
        else:
            kw['Rs'] = float(g['surf_Rs'])
        if 'General' in cls:
            kw.update(RmBragg=float(g['surf_RmBragg']), RsBragg=float(g['surf_RsBragg']))
        alpha = float(g['surf_alpha'])
        oe = getattr(roe, cls)(bl, 'an', material=si, alpha=alpha if alpha else None, **kw,
                               **common)
    elif name.startswith('g3_cell_'):
        m = product_cell(str(g['cell']))
        if 'surf_Rm' in g.files:
            oe = roe.JohannCylinder(bl, 'gr', Rm=float(g['surf_Rm']), material=m, **common)
        else:
            oe = roe.OE(bl, 'qz', material=m, alpha=float(g['alpha']), **common)
    elif name.startswith('g3_laue_plate'):
        alpha = float(g['alpha'])
        si = rm.CrystalSi(hkl=(1, 1, 1), geom=str(g['cr_geom']), t=float(g['cr_t']))
        assert si.d == float(g['cr_d']) and si.chiToF == float(g['cr_chiToF'])
        oe = roe.LauePlate(bl, 'lp', material=si, alpha=alpha if alpha else None, **common)
    elif name.startswith('g3_dcm'):
        alpha = float(g['alpha'])
        si1 = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
        si2 = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
        assert si1.d == float(g['cr_d']) and si1.chiToF == float(g['cr_chiToF'])
        bent = dict(Rs=float(g['Rs'])) if 'Rs' in g.files else {}
        oe = (roe.DCMwithSagittalFocusing if bent else roe.DCM)(
            bl, 'dcm', bragg=float(g['oe_bragg']), material=si1, material2=si2, **bent,
            cryst1roll=float(g['oe_cryst1roll']), cryst2roll=float(g['oe_cryst2roll']),
            cryst2pitch=float(g['oe_cryst2pitch']),
            cryst2finePitch=float(g['oe_cryst2finePitch']),
            cryst2perpTransl=float(g['oe_cryst2perpTransl']),
            cryst2longTransl=float(g['oe_cryst2longTransl']),
            limPhysX2=_opt(g, 'oe_surfPhysX2'), limPhysY2=_opt(g, 'oe_surfPhysY2'),
            limOptX2=_opt(g, 'oe_surfOptX2'), limOptY2=_opt(g, 'oe_surfOptY2'),
            alpha=alpha if alpha else None, **common)
    else:
        raise KeyError(name)
    assert oe.lostNum == int(g['oe_lostNum'])
    return oe


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


from xrt_amd.workloads import cfg2_toroid, cfg3_dcm, synthetic_rays  # noqa: E402,F401
