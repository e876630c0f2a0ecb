"""TEST INFRASTRUCTURE ONLY — regenerates the P1 golden vectors by RUNNING THE
REFERENCE (imported from /root/reference, build container only):

  g6_element_tables.npz   E/f1/f2 ('Chantler total'), f0 coefficients, mass of
                          Si, Pt, Rh, Au, O (data the GPU box needs)
  g5_material_grid.npz    Material.get_amplitude / get_refractive_index grids
  g3_rocking_curves.npz   Crystal.get_amplitude on theta grids, Si111/Si333 x
                          asymmetry x thickness x Bragg/Laue x refl/transm
  g2_*.npz                OE.reflect: toroid+Pt (cfg2 geometry, with edge rays),
                          flat mirror with azimuth/roll/yaw/limOpt, Brent case
  g3_dcm_*.npz            DCM.double_reflect Si(111) (cfg3 geometry) and an
                          asymmetric-cut variant
  g1_source_screen.npz    GeometricSource.shine -> Screen.expose

While generating, the numpy restatement (oracle/reflect_np.py,
oracle/materials_np.py) is asserted against the reference on the same inputs.

Run:  python -m oracle.gen_fixtures_p1
"""
import os

import numpy as np

from . import _refenv
from . import materials_np as mn
from . import reflect_np as rn
from . import elements_np as en

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')
BEAM_FIELDS = ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp',
               'state')


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def beam_dict(prefix, beam):
    d = {prefix + f: np.array(getattr(beam, f)) for f in BEAM_FIELDS}
    if hasattr(beam, 'Es'):
        d[prefix + 'Es'] = np.array(beam.Es)
        d[prefix + 'Ep'] = np.array(beam.Ep)
    if hasattr(beam, 'theta'):
        d[prefix + 'theta'] = np.array(beam.theta)
    return d


def ref_beam(rs, n, with_amplitudes):
    return rs.Beam(nrays=n, withAmplitudes=with_amplitudes)


def to_oracle_beam(b):
    o = rn.Beam(len(b.x), with_amplitudes=hasattr(b, 'Es'))
    for f in o.fields():
        setattr(o, f, np.array(getattr(b, f)))
    return o


def assert_beams(tag, mine, ref, fields=None):
    for f in (fields or mine.fields()):
        m = getattr(mine, f)
        r = getattr(ref, f)
        if f == 'state':
            assert np.array_equal(m, r), (tag, f, (m != r).sum())
        else:
            scale = max(np.abs(r).max(), 1e-300)
            err = np.abs(m - r).max() / scale
            assert err <= 1e-13, (tag, f, err)


# --------------------------------------------------------------------------
def oe_params(oe, surface):
    """Reference OE -> the oracle's parameter dictionary."""
    p = dict(
        center=[float(c) for c in oe.center],
        azimuth_sc=(oe.bl.sinAzimuth, oe.bl.cosAzimuth),
        pitch=oe.pitch, roll=oe.roll, yaw=oe.yaw, positionRoll=oe.positionRoll,
        rotationSequence=oe.rotationSequence, extraPitch=oe.extraPitch,
        extraRoll=oe.extraRoll, extraYaw=oe.extraYaw,
        extraRotationSequence=oe.extraRotationSequence, dx=oe.dx,
        shape=oe.shape, overEdge=oe.overEdge, lostNum=oe.lostNum,
        surfPhysX=list(oe.surfPhysX), surfPhysY=list(oe.surfPhysY),
        surfOptX=None if oe.surfOptX is None else list(oe.surfOptX),
        surfOptY=None if oe.surfOptY is None else list(oe.surfOptY),
        surface=surface)
    if hasattr(oe, 'invertNormal'):          # e.g. HyperbolicMirrorParam
        p['invertNormal'] = int(oe.invertNormal)
    if hasattr(oe, 'cryst2pitch'):
        p.update(
            bragg=oe.bragg, cryst1roll=oe.cryst1roll, cryst2roll=oe.cryst2roll,
            cryst2pitch=oe.cryst2pitch, cryst2finePitch=oe.cryst2finePitch,
            cryst2perpTransl=oe.cryst2perpTransl,
            cryst2longTransl=oe.cryst2longTransl,
            surfPhysX2=list(oe.surfPhysX2), surfPhysY2=list(oe.surfPhysY2),
            surfOptX2=None if oe.surfOptX2 is None else list(oe.surfOptX2),
            surfOptY2=None if oe.surfOptY2 is None else list(oe.surfOptY2))
    return p


def flat_params(p):
    """Flatten the parameter dict into npz-storable scalars/arrays."""
    out = {}
    for k, v in p.items():
        if k in ('surface', 'surface2', 'material', 'material2', 'gratingDensity',
                 'gVector', 'order', 'fzp', 'gfzp', 'local_g'):
            continue
        if v is None:
            out['oe_' + k] = np.array(np.nan)
        elif isinstance(v, str):
            out['oe_' + k] = np.array(v)
        else:
            out['oe_' + k] = np.array(v, dtype=float)
    return out


def make_rays(rs, n, seed, sx=0.1, sz=0.1, sa=2e-4, sc=2e-5, E=(8990., 9010.),
              amplitudes=False, pol=None):
    rng = np.random.default_rng(seed)
    b = ref_beam(rs, n, amplitudes)
    b.x[:] = rng.normal(0, sx, n)
    b.z[:] = rng.normal(0, sz, n)
    b.y[:] = 0.
    b.a[:] = rng.normal(0, sa, n)
    b.c[:] = rng.normal(0, sc, n)
    b.b[:] = np.sqrt(1 - b.a**2 - b.c**2)
    b.E[:] = rng.uniform(E[0], E[1], n)
    b.state[:] = 1
    if pol == 'mixed':
        ang = rng.uniform(0, np.pi, n)
        ph = rng.uniform(-np.pi, np.pi, n)
        es = np.cos(ang)
        ep = np.sin(ang) * np.exp(1j*ph)
        b.Jss[:] = es*es
        b.Jpp[:] = (ep*np.conj(ep)).real
        b.Jsp[:] = es*np.conj(ep)
        if amplitudes:
            b.Es[:] = es
            b.Ep[:] = ep
    else:
        b.Jss[:] = 1.
        b.Jpp[:] = 0.
        b.Jsp[:] = 0.
        if amplitudes:
            b.Es[:] = 1.
            b.Ep[:] = 0.
    return b


def run_reflect(tag, rs, oe, params, beam, brent_expected=None, **extra):
    import xrt.backends.raycing as raycing
    spy = {}
    orig = oe.find_intersection

    def find_spy(local_f, t1, t2, *a, **k):
        spy['t1'] = np.array(t1)
        spy['t2'] = np.array(t2)
        return orig(local_f, t1, t2, *a, **k)
    oe.find_intersection = find_spy
    verb = raycing._VERBOSITY_
    seed = extra.pop('np_seed', None)
    if seed is not None:            # elements that draw from numpy's global generator
        np.random.seed(seed)
    gb, lb = oe.reflect(beam)
    raycing._VERBOSITY_ = verb
    oe.find_intersection = orig
    info = {}
    if seed is not None:
        np.random.seed(seed)
        extra['np_seed'] = np.array(seed)
        extra['lb_order'] = np.array(lb.order)
    mgb, mlb = rn.oe_reflect(params, to_oracle_beam(beam), info=info)
    if seed is not None:
        assert np.array_equal(mlb.order, lb.order)
    assert_beams(tag + ':gb', mgb, gb)
    assert_beams(tag + ':lb', mlb, lb)
    assert np.allclose(mlb.theta, lb.theta, rtol=0, atol=1e-15)
    good = beam.state > 0
    assert np.array_equal(info['tMin'][good], spy['t1'])
    assert np.array_equal(info['tMax0'][good], spy['t2'])
    if brent_expected is not None:
        assert info['brent'] == brent_expected, info['brent']
    st, cnt = np.unique(lb.state, return_counts=True)
    print(tag, 'states', dict(zip(st.tolist(), cnt.tolist())), 'brent',
          info['brent'], 'numit', info['numit'], 'axis', info['axis'])
    out = {}
    out.update(beam_dict('in_', beam))
    out.update(beam_dict('gb_', gb))
    out.update(beam_dict('lb_', lb))
    out.update(flat_params(params))
    out.update(tMin=info['tMin'], tMax0=info['tMax0'], tMax=info['tMax'],
               brent=np.array(info['brent']), numit=np.array(info['numit']),
               axis=np.array(info['axis']))
    out.update(extra)
    save(tag, **out)


def material_dict(tables, m):
    """Reference Material -> oracle material dict (same tables)."""
    elems = [mn.load_element(tables, e.name) for e in m.elements]
    return mn.make_material(elems, list(m.quantities), m.kind, m.rho, m.t)


def crystal_dict(tables, c):
    elem = mn.load_element(tables, c.elements[0].name)
    structure = 'diamond' if any('Diamond' in k.__name__
                                 for k in type(c).__mro__) else 'fcc'
    cr = mn.make_crystal(elem, c.hkl, c.d, structure, c.geom, c.t, c.factDW, c.V)
    assert cr['chiToF'] == c.chiToF
    return cr


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    import xrt.backends.raycing.screens as rsc
    from xrt.backends.raycing.physconsts import CH, CHBAR
    from .consts import CH as myCH, CHBAR as myCHBAR
    assert CH == myCH and CHBAR == myCHBAR
    os.makedirs(OUT, exist_ok=True)

    # ---------------- G6: element tables ---------------------------------
    tables = {}
    for name in ('Si', 'Pt', 'Rh', 'Au', 'O'):
        e = rm.Element(name, table='Chantler total')
        tables[name + '_Z'] = np.array(e.Z)
        tables[name + '_mass'] = np.array(e.mass)
        tables[name + '_f0'] = np.array(e.f0coeffs, dtype=float)
        tables[name + '_E'] = np.array(e.E, dtype=float)
        tables[name + '_f1'] = np.array(e.f1, dtype=float)
        tables[name + '_f2'] = np.array(e.f2, dtype=float)
    save('g6_element_tables', **tables)

    # ---------------- G5: amorphous material amplitudes -------------------
    g5 = {}
    Egrid = np.linspace(2000., 30000., 57)
    thgrid = np.linspace(1e-3, 10e-3, 19)
    EE, TH = [v.ravel() for v in np.meshgrid(Egrid, thgrid)]
    g5['E'] = EE
    g5['theta'] = TH
    bdn = -np.sin(TH)          # beamInDotNormal for grazing angle theta
    mats = dict(
        Pt=rm.Material('Pt', rho=21.45, kind='mirror'),
        Rh=rm.Material('Rh', rho=12.41, kind='mirror'),
        Si=rm.Material('Si', rho=2.33, kind='mirror'),
        SiO2=rm.Material(('Si', 'O'), quantities=(1, 2), rho=2.2, kind='mirror'),
        PtThin=rm.Material('Pt', rho=21.45, kind='thin mirror', t=30e-6),
        SiPlate=rm.Material('Si', rho=2.33, kind='plate'))
    for name, m in mats.items():
        md = material_dict(tables, m)
        for fromVacuum in ((True, False) if m.kind == 'plate' else (True,)):
            arg = bdn if fromVacuum else -np.cos(TH)   # steep inside a plate
            if m.kind == 'plate':
                arg = -np.cos(TH * 50) if fromVacuum else np.cos(TH * 50)
            ref = m.get_amplitude(EE.copy(), arg.copy(), fromVacuum)
            mine = mn.material_amplitude(md, EE.copy(), arg.copy(), fromVacuum)
            key = '%s_%s' % (name, 'in' if fromVacuum else 'out')
            for i, lab in enumerate(('rs', 'rp', 'mu', 'nk')):
                assert np.allclose(mine[i], ref[i], rtol=1e-14, atol=0), (key, lab)
                g5[key + '_' + lab] = np.array(ref[i])
            g5[key + '_bdn'] = arg
        nref = m.get_refractive_index(Egrid)
        assert np.allclose(mn.refractive_index(md, Egrid), nref, rtol=1e-15)
        g5[name + '_n'] = nref
    g5['Egrid'] = Egrid
    save('g5_material_grid', **g5)

    # ---------------- G3a: rocking curves ---------------------------------
    g3 = {}
    E0 = 9000.
    npts = 400
    for hkl in ((1, 1, 1), (3, 3, 3)):
        for geom in ('Bragg reflected', 'Bragg transmitted', 'Laue reflected',
                     'Laue transmitted'):
            for tmm in (None, 0.1, 0.007):
                if tmm is None and geom != 'Bragg reflected':
                    continue
                for alphaDeg in (-5., 0., 5.):
                    c = rm.CrystalSi(hkl=hkl, geom=geom, t=tmm)
                    cr = crystal_dict(tables, c)
                    alpha = np.radians(alphaDeg)
                    thetaB = c.get_Bragg_angle(E0)
                    dth = np.linspace(-60, 60, npts) * 4.848e-6 * \
                        (1 if hkl == (1, 1, 1) else 0.2)
                    theta = thetaB + dth
                    E = np.ones(npts) * E0
                    if geom.startswith('Bragg'):
                        g0 = -np.sin(theta + alpha)
                        gh = np.sin(theta - alpha)
                    else:
                        g0 = -np.cos(theta + alpha)
                        gh = -np.cos(theta - alpha)
                    hns = -np.sin(theta)
                    ref = c.get_amplitude(E.copy(), g0.copy(), gh.copy(), hns.copy())
                    mine = mn.crystal_amplitude(cr, E.copy(), g0.copy(), gh.copy(),
                                                hns.copy())
                    key = 'Si%d%d%d_%s_%s_%+d' % (
                        hkl + (geom.replace(' ', ''),
                               'thick' if tmm is None else '%gum' % (tmm*1e3),
                               int(alphaDeg)))
                    for i, lab in enumerate(('S', 'P')):
                        fin = np.isfinite(ref[i])
                        assert np.array_equal(fin, np.isfinite(mine[i])), key
                        sc = np.abs(ref[i][fin]).max()
                        assert np.abs(mine[i][fin] - ref[i][fin]).max() <= 1e-12*sc, key
                        g3[key + '_' + lab] = np.array(ref[i])
                    g3[key + '_in'] = np.array([E, g0, gh, hns])
                    g3[key + '_par'] = np.array(
                        [c.d, c.V, c.chiToF, np.nan if tmm is None else tmm])
    save('g3_rocking_curves', **g3)

    # ---------------- G2a: toroid + Pt (cfg2 geometry) --------------------
    p, q, pitch = 20000., 10000., 4e-3
    bl = raycing.BeamLine()
    mPt = rm.Material('Pt', rho=21.45, kind='mirror')
    tm = roe.ToroidMirror(
        bl, 'tm', center=[0, p, 0], pitch=pitch, R=(p, q), r=(p, q),
        material=mPt, limPhysX=[-10, 10], limPhysY=[-300, 300])
    n = 2048
    beam = make_rays(rs, n, 42, amplitudes=True, pol='mixed')
    # hand-placed edge rays: miss / over / lost / on limPhys
    beam.z[0] = 2.5                   # passes over the far end -> "over"
    beam.c[0] = 0.
    beam.z[1] = -3.0                  # below the mirror at entrance -> lost
    beam.x[2] = 10.0                  # exactly on limPhysX
    beam.a[2] = 0.
    beam.x[3] = np.nextafter(10.0, 11)    # one ulp outside
    beam.a[3] = 0.
    beam.x[4] = 25.                   # far outside in x
    beam.state[5] = 2                 # "out" rays still enter (state > 0)
    beam.state[6] = 3                 # "over" from a previous element too
    beam.state[7] = -1                # already lost: untouched
    beam.state[8] = 0
    beam.c[9] = -1.2e-3               # steeper: hits near the upstream edge
    beam.c[10] = 3.99e-3              # almost parallel to the surface (grazing)
    beam.b[9:11] = np.sqrt(1 - beam.a[9:11]**2 - beam.c[9:11]**2)
    par = oe_params(tm, dict(kind='toroid', R=tm.R, r=tm.r))
    par['material'] = material_dict(tables, mPt)
    run_reflect('g2_toroid_pt', rs, tm, par, beam, surf_R=np.array(tm.R),
                surf_r=np.array(tm.r), mat_rho=np.array(21.45))

    # ---------------- G2b: flat mirror, general orientation ---------------
    bl = raycing.BeamLine(azimuth=0.3)
    mRh = rm.Material('Rh', rho=12.41, kind='thin mirror', t=40e-6)
    ce = [np.sin(0.3)*15000., np.cos(0.3)*15000., 0.]
    fm = roe.OE(bl, 'fm', center=ce, pitch=3e-3, roll=2e-3, yaw=-1e-3,
                positionRoll=np.pi/2, material=mRh, limPhysX=[-8, 8],
                limPhysY=[-200, 150], limOptX=[-5, 5], limOptY=[-150, 100],
                overEdge='xMin yMax')
    beam = make_rays(rs, n, 43, sx=0.5, sz=3.0, sa=1.5e-4, sc=4e-5,
                     E=(6000., 12000.), amplitudes=False, pol='mixed')
    # positionRoll = pi/2 deflects horizontally: rotate the fan into the
    # beamline azimuth so that it travels along the local y of the mirror
    xx, yy = beam.x.copy(), beam.y.copy()
    aa, bb = beam.a.copy(), beam.b.copy()
    beam.x[:], beam.y[:] = raycing.rotate_z(xx, yy, bl.cosAzimuth, -bl.sinAzimuth)
    beam.a[:], beam.b[:] = raycing.rotate_z(aa, bb, bl.cosAzimuth, -bl.sinAzimuth)
    par = oe_params(fm, dict(kind='flat'))
    par['material'] = material_dict(tables, mRh)
    run_reflect('g2_flat_general', rs, fm, par, beam, mat_rho=np.array(12.41),
                mat_t=np.array(40e-6))

    # ---------------- G2c: a geometry that selects Brent ------------------
    bl = raycing.BeamLine()
    tm2 = roe.ToroidMirror(
        bl, 'tm2', center=[0, 2000., 0], pitch=0.2, R=3000., r=40.,
        material=None, limPhysX=[-30, 30], limPhysY=[-5, 400])
    beam = make_rays(rs, 2048, 44, sx=2.0, sz=0.5, sa=2e-3, sc=1e-4,
                     E=(8990., 9010.))
    par = oe_params(tm2, dict(kind='toroid', R=tm2.R, r=tm2.r))
    run_reflect('g2_toroid_brent', rs, tm2, par, beam, brent_expected=True,
                surf_R=np.array(tm2.R), surf_r=np.array(tm2.r))

    # ---------------- G3b: DCM Si(111) (cfg3 geometry) --------------------
    for tag, alphaDeg in (('g3_dcm_si111', 0.), ('g3_dcm_si111_asym', 3.),
                          ('g3_dcm_sagittal', 0.)):
        sagittal = tag.endswith('sagittal')     # 2nd crystal bent to Rs (:639-664)
        bl = raycing.BeamLine()
        si1 = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
        si2 = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
        E0 = 9000.
        thB = si1.get_Bragg_angle(E0) - si1.get_dtheta(E0, np.radians(alphaDeg))
        kw = dict(alpha=np.radians(alphaDeg)) if alphaDeg else {}
        if sagittal:
            kw['Rs'] = 2 * 20000. * 10000. / 30000. * np.sin(float(thB))   # 20 m : 10 m
        dcm = (roe.DCMwithSagittalFocusing if sagittal else roe.DCM)(
            bl, 'dcm', center=[0, 20000., 0], bragg=thB,
            pitch=np.radians(alphaDeg), material=si1,
            material2=si2, cryst2perpTransl=10., limPhysX=[-10, 10],
            limPhysY=[-50, 50], limPhysX2=[-10, 10], limPhysY2=[-50, 150], **kw)
        beam = make_rays(rs, n, 73 if sagittal else 45, sx=2.0 if sagittal else 0.1,
                         sa=1e-4, sc=2e-5, E=(8995., 9005.),
                         amplitudes=(alphaDeg == 0), pol='mixed')
        beam.x[0] = 30.                # misses crystal 1
        beam.z[1] = 8.                 # far above: lost at crystal 1 physical edge
        beam.state[2] = -3
        gb2, lo1, lo2 = dcm.double_reflect(beam)
        par = oe_params(dcm, dict(kind='flat', alpha=dcm.alpha))
        par['surface2'] = dict(kind='sagittal', Rs=dcm.Rs) if sagittal else \
            dict(kind='flat', alpha=dcm.alpha, flip_n_y=True)
        par['material'] = crystal_dict(tables, si1)
        par['material2'] = crystal_dict(tables, si2)
        info = {}
        m2, m1l, m2l = rn.dcm_double_reflect(par, to_oracle_beam(beam), info=info)
        assert_beams(tag + ':gb2', m2, gb2)
        assert_beams(tag + ':lo1', m1l, lo1)
        assert_beams(tag + ':lo2', m2l, lo2)
        st, cnt = np.unique(gb2.state, return_counts=True)
        print(tag, 'states', dict(zip(st.tolist(), cnt.tolist())),
              'mean J', (gb2.Jss + gb2.Jpp)[gb2.state == 1].mean())
        out = {}
        out.update(beam_dict('in_', beam))
        out.update(beam_dict('gb_', gb2))
        out.update(beam_dict('lo1_', lo1))
        out.update(beam_dict('lo2_', lo2))
        out.update(flat_params(par))
        out.update(alpha=np.array(dcm.alpha if dcm.alpha else 0.),
                   cr_d=np.array(si1.d), cr_V=np.array(si1.V),
                   cr_chiToF=np.array(si1.chiToF))
        if sagittal:
            out['Rs'] = np.array(dcm.Rs)
        save(tag, **out)

    # ---------------- G2e: BentFlatMirror (VCM) + Rh ----------------------
    bl = raycing.BeamLine()
    mRhM = rm.Material('Rh', rho=12.41, kind='mirror')
    vcm = roe.BentFlatMirror(
        bl, 'vcm', center=[0, 15000., 0], pitch=2.5e-3, R=(15000., 1e9),
        material=mRhM, limPhysX=[-15, 15], limPhysY=[-400, 400])
    beam = make_rays(rs, n, 48, sx=0.5, sz=0.2, sa=3e-4, sc=2e-5,
                     E=(6000., 14000.), amplitudes=True, pol='mixed')
    par = oe_params(vcm, dict(kind='bentflat', R=vcm.R, y0=vcm.limPhysY[0]))
    par['material'] = material_dict(tables, mRhM)
    run_reflect('g2_bentflat_rh', rs, vcm, par, beam, surf_R=np.array(vcm.R),
                mat_rho=np.array(12.41))

    # ---------------- G2o: ConicalMirror (oes/__init__.py:589-636) ---------
    bl = raycing.BeamLine()
    mRhC = rm.Material('Rh', rho=12.41, kind='mirror')
    cone = roe.ConicalMirror(bl, 'cone', center=[0, 12000., 0], pitch=4e-3, L0=900.,
                             theta=3e-3, material=mRhC, limPhysX=[-1.5, 1.5],
                             limPhysY=[-250, 250])
    beam = make_rays(rs, n, 72, sx=0.4, sz=0.3, sa=5e-5, sc=2e-5,
                     E=(7000., 12000.), amplitudes=True, pol='mixed')
    beam.state[1] = 2
    beam.state[2] = -2
    surfc = rn.make_cone(cone.L0, cone.theta)
    for k in ('tt', 't2t', 'redfocus'):
        assert surfc[k] == getattr(cone, k), k
    par = oe_params(cone, surfc)
    par['material'] = material_dict(tables, mRhC)
    run_reflect('g2_cone_rh', rs, cone, par, beam, surf_L0=np.array(cone.L0),
                surf_theta=np.array(cone.theta), mat_rho=np.array(12.41))

    # ---------------- G2p: capillaries (oes/parametric.py:717-988) ---------
    mAuC = rm.Material('Au', rho=19.3, kind='mirror')
    for tag, cls, kw, spread in (
            ('g2_capillary_parab', roe.ParaboloidCapillaryMirror, dict(q=500., r0=2.5), 2.0),
            ('g2_capillary_ellipse', roe.EllipsoidCapillaryMirror,
             dict(ellipseA=1000., ellipseB=3., workingDistance=100.), 2.0),
            ('g2_capillary_hyperbola', roe.HyperboloidCapillaryMirror,
             dict(hyperbolaA=1000., hyperbolaB=3., workingDistance=100.), 1.0)):
        bl = raycing.BeamLine()
        cap = cls(bl, 'cap', center=[0, 1000., 0], material=mAuC, limPhysY=[-50, 50], **kw)
        beam = make_rays(rs, n, 74, sx=spread, sz=spread, sa=2e-5, sc=2e-5,
                         E=(8999., 9001.), amplitudes=True, pol='mixed')
        beam.state[1] = 2
        beam.state[2] = -2
        if 'parab' in tag:
            surf = dict(kind='parab_capillary', s0=cap.s0, focus=cap.focus)
        elif 'ellipse' in tag:
            surf = dict(kind='ellipse_capillary', ellipseA=cap.ellipseA,
                        ellipseB=cap.ellipseB, ctd=cap.ctd)
        else:
            surf = dict(kind='hyperbola_capillary', hyperbolaA=cap.hyperbolaA,
                        hyperbolaB=cap.hyperbolaB, ctd=cap.ctd)
        par = oe_params(cap, surf)
        par['material'] = material_dict(tables, mAuC)
        run_reflect(tag, rs, cap, par, beam, mat_rho=np.array(19.3),
                    **{'cap_' + k: np.array(float(v)) for k, v in kw.items()})

    # ---------------- G3c: LauePlate (oes/laue.py:11-23) -------------------
    for tag, alpha, geom in (('g3_laue_plate', None, 'Laue reflected'),
                             ('g3_laue_plate_asym', np.radians(5.), 'Laue reflected'),
                             ('g3_laue_plate_transmitted', np.radians(-3.),
                              'Laue transmitted')):
        bl = raycing.BeamLine()
        siL = rm.CrystalSi(hkl=(1, 1, 1), geom=geom, t=0.1)
        thL = siL.get_Bragg_angle(9000.) - siL.get_dtheta(9000., alpha)
        lp = roe.LauePlate(bl, 'lp', center=[0, 10000., 0],
                           pitch=float(thL[0] if np.ndim(thL) else thL) +
                           (alpha if alpha else 0) + np.pi/2, material=siL, alpha=alpha,
                           limPhysX=[-5, 5], limPhysY=[-1.2, 1.5])
        beam = make_rays(rs, n, 71, sx=0.5, sz=0.5, sa=2e-5, sc=2e-5,
                         E=(8999., 9001.), amplitudes=True, pol='mixed')
        beam.state[2] = 2
        beam.state[3] = -3
        par = oe_params(lp, dict(kind='flat', laue=True, alpha=alpha))
        par['material'] = crystal_dict(tables, siL)
        run_reflect(tag, rs, lp, par, beam, alpha=np.array(alpha if alpha else 0.),
                    cr_d=np.array(siL.d), cr_t=np.array(0.1), cr_geom=np.array(geom),
                    cr_chiToF=np.array(siL.chiToF), cr_V=np.array(siL.V))

    # ---------------- G2d: Plate.double_refract (Be window) ---------------
    bl = raycing.BeamLine()
    mBe = rm.Material('Be', rho=1.848, kind='plate')
    plate = roe.Plate(bl, 'win', center=[0, 5000., 0], pitch=np.pi/2 - 0.05,
                      material=mBe, t=0.5, limPhysX=[-3, 4], limPhysY=[-2, 2])
    beam = make_rays(rs, n, 46, sx=1.0, sz=0.6, sa=3e-4, sc=3e-4,
                     E=(5000., 15000.), amplitudes=True, pol='mixed')
    gbp, lp1, lp2 = plate.double_refract(beam)
    par = oe_params(plate, dict(kind='flat'))
    par['surface2'] = dict(kind='flat')
    tbBe = dict(tables)
    eBe = rm.Element('Be', table='Chantler total')
    for key, val in (('Z', eBe.Z), ('mass', eBe.mass), ('f0', eBe.f0coeffs),
                     ('E', eBe.E), ('f1', eBe.f1), ('f2', eBe.f2)):
        tbBe['Be_' + key] = np.array(val, dtype=float)
    par['material'] = material_dict(tbBe, mBe)
    par['material2'] = par['material']
    m2, m1l, m2l = rn.dcm_double_reflect(par, to_oracle_beam(beam),
                                         fromVacuum1=True, fromVacuum2=False,
                                         is_plate=True)
    assert_beams('g2_plate:gb2', m2, gbp)
    assert_beams('g2_plate:lo1', m1l, lp1)
    assert_beams('g2_plate:lo2', m2l, lp2)
    st, cnt = np.unique(gbp.state, return_counts=True)
    print('g2_plate_be states', dict(zip(st.tolist(), cnt.tolist())), 'mean T',
          (gbp.Jss + gbp.Jpp)[gbp.state == 1].mean())
    out = {}
    out.update(beam_dict('in_', beam))
    out.update(beam_dict('gb_', gbp))
    out.update(beam_dict('lo1_', lp1))
    out.update(beam_dict('lo2_', lp2))
    out.update(flat_params(par))
    out.update(plate_t=np.array(0.5), mat_rho=np.array(1.848))
    for key in ('Z', 'mass', 'f0', 'E', 'f1', 'f2'):
        out['Be_' + key] = tbBe['Be_' + key]
    save('g2_plate_be', **out)

    # ---------------- G1: GeometricSource -> Screen -----------------------
    np.random.seed(0)
    bl = raycing.BeamLine(azimuth=0.05)
    src = rs.GeometricSource(
        bl, 'src', nrays=2048, dx=0.32, dz=0.018, dxprime=1e-3, dzprime=1e-4,
        distE='lines', energies=(9000.,), polarization='h')
    scr = rsc.Screen(bl, 'scr', center=[np.sin(0.05)*10000., np.cos(0.05)*10000., 0])
    b0 = src.shine()
    b0.state[5] = -2
    b0.state[6] = 3
    lo = scr.expose(b0)
    out = {}
    out.update(beam_dict('in_', b0))
    out.update(beam_dict('lo_', lo))
    out.update(scr_center=np.array(scr.center, dtype=float),
               scr_x=np.array(scr.x, dtype=float), scr_y=np.array(scr.y, dtype=float),
               scr_z=np.array(scr.z, dtype=float), azimuth=np.array(0.05),
               scr_lostNum=np.array(scr.lostNum))
    save('g1_source_screen', **out)
    mine = en.screen_expose(to_oracle_beam(b0), (scr.x, scr.y, scr.z), scr.center,
                            scr.lostNum)
    assert_beams('g1:screen', mine, lo)

    # ---------------- G1b: MeshSource / NESWSource -------------------------
    kwm = dict(center=(1., 2., 3.), minxprime=-2e-4, maxxprime=3e-4, minzprime=-1e-4,
               maxzprime=1.5e-4, nx=7, nz=5, distE='flat', energies=(8000., 9000.),
               polarization='+45', totalFlux=1e12)
    outm = {}
    for cls in ('MeshSource', 'NESWSource'):
        np.random.seed(4)
        bm = getattr(rs, cls)(raycing.BeamLine(azimuth=0.03), name='m', **kwm).shine()
        outm.update(beam_dict(cls + '_', bm))
        outm[cls + '_sourceWeight'] = np.array(getattr(bm, 'sourceWeight', np.nan))
    save('g1_mesh_sources', **outm)

    # ---------------- G7: RectangularAperture.propagate -------------------
    import xrt.backends.raycing.apertures as ra
    bl = raycing.BeamLine(azimuth=-0.02)
    slit = ra.RectangularAperture(
        bl, 'slit', center=[np.sin(-0.02)*8000., np.cos(-0.02)*8000., 0.2],
        kind=('left', 'right', 'bottom', 'top'), opening=[-0.8, 1.1, -0.3, 0.25])
    dummy = ra.RectangularAperture(bl, 'pad', [0, 1, 0], ('left',), [0])  # ordinal 2
    assert slit.lostNum == -1001
    beam = make_rays(rs, n, 47, sx=0.4, sz=0.15, sa=1e-4, sc=3e-5,
                     E=(7000., 9000.), amplitudes=True, pol='mixed')
    xx, yy, aa, bb = beam.x.copy(), beam.y.copy(), beam.a.copy(), beam.b.copy()
    beam.x[:], beam.y[:] = raycing.rotate_z(xx, yy, bl.cosAzimuth, -bl.sinAzimuth)
    beam.a[:], beam.b[:] = raycing.rotate_z(aa, bb, bl.cosAzimuth, -bl.sinAzimuth)
    beam.state[3] = 2
    beam.state[4] = 3
    beam.state[5] = -7
    beam.state[6] = 0
    b_in = rs.Beam(copyFrom=beam)
    glo, lo = slit.propagate(beam, needNewGlobal=True)
    ob = to_oracle_beam(b_in)
    mglo, mlo = en.aperture_propagate(
        ob, slit.xyz, slit.center, dict(slit.blades), slit.lostNum,
        (bl.sinAzimuth, bl.cosAzimuth), needNewGlobal=True)
    assert_beams('g7:lo', mlo, lo)
    assert_beams('g7:glo', mglo, glo)
    assert np.array_equal(ob.state, beam.state)
    st, cnt = np.unique(lo.state, return_counts=True)
    print('g7_aperture states', dict(zip(st.tolist(), cnt.tolist())))
    out = {}
    out.update(beam_dict('in_', b_in))
    out.update(beam_dict('lo_', lo))
    out.update(beam_dict('glo_', glo))
    out.update(in_state_after=np.array(beam.state), azimuth=np.array(-0.02),
               center=np.array(slit.center, dtype=float),
               opening=np.array([-0.8, 1.1, -0.3, 0.25]),
               lostNum=np.array(slit.lostNum))
    save('g7_aperture', **out)

    # ---------------- G7b: beam stops and round apertures ------------------
    out = {}
    for tag, cls, kw in (
            ('rect_stop', ra.RectangularBeamStop,
             dict(kind=('left', 'right', 'bottom', 'top'), opening=[-0.3, 0.4, -0.1, 0.12])),
            ('round', ra.RoundAperture, dict(r=0.45)),
            ('round_stop', ra.RoundBeamStop, dict(r=0.2)),
            ('double', ra.DoubleSlit,
             dict(kind=('left', 'right', 'bottom', 'top'), opening=[-0.5, 0.6, -0.25, 0.3],
                  shadeFraction=0.4)),
            ('polygon', ra.PolygonalAperture,
             dict(vertices=[(-0.5, -0.2), (0.1, -0.3), (0.6, 0.0), (0.2, 0.3), (-0.3, 0.15)])),
            ('polygon_stop', ra.PolygonalBeamStop,
             dict(vertices=[(-0.2, -0.1), (0.2, -0.1), (0.0, 0.2)]))):
        bl = raycing.BeamLine(azimuth=-0.02)
        ap = cls(bl, tag, center=[np.sin(-0.02)*8000., np.cos(-0.02)*8000., 0.05], **kw)
        beam = rs.Beam(copyFrom=b_in)
        glo, lo = ap.propagate(beam, needNewGlobal=True)
        ob = to_oracle_beam(b_in)
        mglo, mlo = en.aperture_propagate(
            ob, ap.xyz, ap.center,
            dict(getattr(ap, 'blades', {})) if tag in ('rect_stop', 'double') else {},
            ap.lostNum, (bl.sinAzimuth, bl.cosAzimuth), isBeamStop=ap.isBeamStop,
            needNewGlobal=True, radius=kw.get('r'), shadeFraction=kw.get('shadeFraction'),
            vertices=kw.get('vertices'))
        assert_beams('g7b:lo', mlo, lo)
        assert_beams('g7b:glo', mglo, glo)
        assert np.array_equal(ob.state, beam.state)
        st, cnt = np.unique(lo.state, return_counts=True)
        print('g7b', tag, 'states', dict(zip(st.tolist(), cnt.tolist())))
        # (the full local / global records of a slit are in g7_aperture: here what the
        # stop shapes decide, plus the complete global beam where it is special)
        full = beam_dict(tag + '_lo_', lo)
        out.update({k: v for k, v in full.items()
                    if k.rsplit('_', 1)[1] in ('state', 'x', 'y', 'z', 'path', 'Es')})
        if tag in ('round', 'double'):
            out.update(beam_dict(tag + '_glo_', glo))
        out[tag + '_in_state_after'] = np.array(beam.state)
        out[tag + '_center'] = np.array(ap.center, dtype=float)
        out[tag + '_lostNum'] = np.array(ap.lostNum)
    out.update(beam_dict('in_', b_in))
    out.update(azimuth=np.array(-0.02), rect_stop_opening=np.array([-0.3, 0.4, -0.1, 0.12]),
               round_r=np.array(0.45), round_stop_r=np.array(0.2),
               double_opening=np.array([-0.5, 0.6, -0.25, 0.3]), double_shade=np.array(0.4),
               polygon_vertices=np.array([(-0.5, -0.2), (0.1, -0.3), (0.6, 0.0), (0.2, 0.3),
                                          (-0.3, 0.15)]),
               polygon_stop_vertices=np.array([(-0.2, -0.1), (0.2, -0.1), (0.0, 0.2)]))
    save('g7_stops_round', **out)


if __name__ == '__main__':
    main()
