"""CPU: host-side logic of xrt_amd that needs no GPU — rotation bookkeeping, the
parameter block of a reflect pass, wave meshes, hull area, Beam container,
plot descriptors."""
import os

import numpy as np
import pytest

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.apertures as ra
import xrt_amd.backends.raycing.materials as rm
import xrt_amd.backends.raycing.oes as roe
import xrt_amd.backends.raycing.screens as rsc
import xrt_amd.backends.raycing.sources as rs
import xrt_amd.backends.raycing.waves as rw
from oracle import reflect_np as rn
from xrt_amd import _structs, plotter as xrtp


def test_rotation_steps_match_oracle_for_all_sequences():
    rng = np.random.default_rng(0)
    axis_name = {0: 'x', 1: 'y', 2: 'z'}
    for seq in ('RzRyRx', 'RxRyRz', 'RyRzRx', '-RzRyRx', '-RxRzRy'):
        for _ in range(20):
            ang = rng.normal(0, 0.5, 3) * (rng.random(3) > 0.3)     # some exact zeros
            mine = raycing.rotation_steps(seq, *ang)
            ref = rn.rotation_steps(seq, *ang)
            assert [(axis_name[a], c, s) for a, c, s in mine] == \
                [(a, float(c), float(s)) for a, c, s in ref]


def _apply(rot, v):
    x, y, z = v
    for i in range(rot.n):
        c, s, ax = rot.cosa[i], rot.sina[i], rot.axis[i]
        if ax == 2:
            x, y = c*x - s*y, s*x + c*y
        elif ax == 1:
            x, z = c*x + s*z, -s*x + c*z
        else:
            y, z = c*y - s*z, s*y + c*z
    return np.array([x, y, z])


def test_pass_block_of_a_general_element():
    bl = raycing.BeamLine(azimuth=0.3)
    oe = roe.OE(bl, 'm', center=[1, 2, 3], pitch=3e-3, roll=2e-3, yaw=-1e-3,
                positionRoll=np.pi/2, extraPitch=1e-4, extraYaw=2e-4,
                limPhysX=[-8, 8], limPhysY=[-200, 150], limOptX=[-5, 5],
                overEdge='xMin yMax', shape='rect')
    p = oe._make_pass(oe.pitch, oe.roll + oe.positionRoll, oe.yaw, oe.dx)
    assert p.to_local.n == 5 and p.to_virgin.n == 5     # 3 main + 2 extra, zero roll skipped
    v = np.array([0.3, -1.2, 0.7])
    back = _apply(p.to_virgin, _apply(p.to_local, v))
    assert np.abs(back - v).max() < 1e-15               # to_virgin undoes to_local
    assert (p.sin_az, p.cos_az) == (bl.sinAzimuth, bl.cosAzimuth)
    assert p.over_mask == _structs.OVER_XMIN | _structs.OVER_YMAX
    assert p.has_opt_x == 1 and p.has_opt_y == 0 and p.lost_num == -1
    assert list(p.n_const) == [0, 0, 1, 0, 0, 1] and p.asymmetric == 0
    assert p.invert_normal == 1 and p.good_mode == 0 and p.out_to_global == 1
    # second crystal of a DCM: roll -pi first, flipped signs, asymmetric normals
    si = rm.CrystalSi(hkl=(1, 1, 1))
    dcm = roe.DCM(bl, 'dcm', bragg=0.22, material=si, material2=si,
                  cryst2perpTransl=10., alpha=0.05)
    p2 = dcm._make_pass(-dcm.pitch - dcm.bragg, dcm.roll, -dcm.yaw, -dcm.dx,
                        dcm.cryst2longTransl, -dcm.cryst2perpTransl, is2ndXtal=True,
                        in_is_global=False, good_mode=1)
    assert p2.to_local.axis[0] == 1 and p2.to_local.cosa[0] == np.cos(-np.pi)
    assert p2.shift[2] == -10. and p2.asymmetric == 1 and p2.in_is_global == 0
    n1, n2 = dcm.local_n1(0., 0.), dcm.local_n2(0., 0.)
    assert n2[1] == -n1[1] and list(p2.n_const) == [float(t) for t in n2]
    assert dcm.lostNum == -2


def test_toroid_radii_from_coddington_and_reciprocals():
    bl = raycing.BeamLine()
    tm = roe.ToroidMirror(bl, 'tm', pitch=4e-3, R=(20000., 10000.), r=(20000., 10000.))
    assert tm.R == 2*20000.*10000./30000./np.sin(4e-3)
    assert tm.r == 2*20000.*10000./30000.*np.sin(4e-3)
    p = tm._make_pass(tm.pitch, 0., 0.)
    assert p.surf_kind == _structs.SURF_TOROID and p.surf_p[4] == 1.
    assert p.surf_p[2] == 1.0 / tm.R and p.surf_p[3] == 1.0 / tm.r
    flat = roe.ToroidMirror(bl, 'flat', R=None, r=None)    # R = r = 1e100: no reciprocals
    assert flat._make_pass(0., 0., 0.).surf_p[4] == 0.


@pytest.mark.parametrize('name', ['g4_slit_2000x32', 'g4_slit_4000x48'])
def test_screen_prepare_wave_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    bl = raycing.BeamLine()
    slit = ra.RectangularAperture(bl, 'slit', [float(v) for v in g['slit_center']],
                                  ('left', 'right', 'bottom', 'top'),
                                  [-0.1, 0.1, -0.1, 0.1])
    scr = rsc.Screen(bl, 'scr', [float(v) for v in g['screen_center']])
    w = scr.prepare_wave(slit, g['xmesh'], g['zmesh'])
    assert np.array_equal(w.xDiffr, g['px']) and np.array_equal(w.yDiffr, g['py'])
    assert np.array_equal(w.zDiffr, g['pz']) and w.dS == float(g['w_dS'])
    assert w.fromOE is slit and w.toOE is scr and not w.EsAcc.any()


def test_convex_hull_area_against_scipy():
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(3)
    for n in (3, 10, 1000):
        pts = rng.normal(size=(n, 2)) * [2.0, 300.0]
        assert abs(rw.convex_hull_area(pts[:, 0], pts[:, 1]) -
                   ConvexHull(pts).volume) <= 1e-10 * ConvexHull(pts).volume
    with pytest.raises(ValueError):
        rw.convex_hull_area(np.zeros(5), np.zeros(5))


def test_beam_container():
    b = rs.Beam(nrays=5, withAmplitudes=True)
    assert len(b) == 5 and b.b.tolist() == [1.] * 5 and b.Jss.sum() == 5
    assert b.state.dtype == np.int32 and b.Jsp.dtype == np.complex128
    b.x[:] = np.arange(5.)
    b.state[:] = [1, 2, 3, -1, 0]
    c = rs.Beam(copyFrom=b)
    c.x[0] = 99.
    assert b.x[0] == 0. and c.x[1] == 1. and hasattr(c, 'Es')
    c.filter_by_index(c.state > 0)
    assert len(c) == 3 and c.state.tolist() == [1, 2, 3] and len(c.Es) == 3
    assert not hasattr(rs.Beam(nrays=2), 'Es')
    with pytest.raises(AttributeError):
        rs.Beam(nrays=2).nonexistent
    assert rs.Beam(nrays=3, forceState=1).state.tolist() == [1, 1, 1]


def test_plot_descriptors():
    p = xrtp.XYCPlot('beam', (1, 3, -1), xrtp.XYCAxis("x'", u'µrad', bins=16),
                     xrtp.XYCAxis('energy', 'keV'), fluxKind='power')
    assert p.ray_flag_mask == 1 | 4 | 8 and p.flux_kind_code == 5
    assert p.xaxis.field() == 'xprime' and p.xaxis.factor == 1e6
    assert p.yaxis.field() == 'E' and p.yaxis.factor == 1e-3
    assert p.total2D.shape == (128, 16)
    with pytest.raises(NotImplementedError):
        xrtp.XYCPlot('b', fluxKind='EsPCA')


def test_grating_block_of_the_pass():
    """gratingDensity / constant local_g -> xrt_hip_pass grating fields."""
    bl = raycing.BeamLine()
    m = rm.Material('Au', rho=19.3, kind='grating')
    g = roe.OE(bl, 'g', material=m, gratingDensity=['y', 300., 1., 2e-4], order=-1)
    p = g._make_pass(g.pitch, g.roll, g.yaw)
    assert (p.grating, p.grating_axis, p.grating_order, p.g_ncoef) == (1, 1, -1, 2)
    assert (p.g_rho0, p.g_coef[0], p.g_coef[1]) == (300., 1., 2e-4)
    gx, gy, gz = g.local_g(np.array([0., 1.]), np.array([0., 10.]))
    assert np.array_equal(gy, 300. * (1. + 2 * 2e-4 * np.array([0., 10.])))

    class C(roe.OE):
        def local_g(self, x, y, rho=None):
            return 0, -250., 0
    c = C(bl, 'c', material=m)
    p = c._make_pass(0., 0., 0.)
    assert (p.grating, p.grating_axis, p.grating_order) == (1, -1, 1)
    assert list(p.g_const) == [0., -250., 0.]
    mirror = roe.OE(bl, 'm', material=rm.Material('Au', rho=19.3, kind='mirror'))
    assert mirror._make_pass(0., 0., 0.).grating == 0


def test_out_of_scope_requests_fail_loudly():
    bl = raycing.BeamLine()
    several = roe.OE(bl, 'g', gratingDensity=['y', 300., 1.], order=(1, 2))   # sequences are in
    assert several.order == [1, 2]
    with pytest.raises(NotImplementedError):
        roe.OE(bl, 'p', isParametric=True)
    with pytest.raises(NotImplementedError):
        roe.BlazedGrating(bl, 'b', blaze=0.01, rho=300., gratingDensity=['y', 300., 1.])
    with pytest.raises(NotImplementedError):
        roe.EllipticalMirrorParam(bl, 'e', p=1000., q=100., f1=[0, 0, 0])
    poly = roe.OE(bl, 'poly', shape=[(0, 0), (1, 0), (0, 1)])      # polygons are in
    assert poly.shape == [(0, 0), (1, 0), (0, 1)]
    class Bump(roe.OE):                       # a user-defined surface: refused, not flattened
        def local_z(self, x, y):
            return 1e-3 * np.exp(-x**2 - y**2)
    with pytest.raises(NotImplementedError):
        Bump(bl, 'bump')._make_pass(0., 0., 0.)
    assert roe.LauePlate(bl, 'lp', alpha=0.1)._make_pass(0., 0., 0.).asymmetric == 1
    with pytest.raises(ValueError):
        roe.OE(bl, 'odd', shape=3.5)
    with pytest.raises(ValueError):
        rm.Element('Si', table='Henke')


def test_beam_utilities(tmp_path):
    a = rs.Beam(nrays=4, withAmplitudes=True)
    b = rs.Beam(nrays=3, withAmplitudes=True)
    a.x[:] = np.arange(4.)
    b.x[:] = 10 + np.arange(3.)
    a.state[:] = [1, 2, 1, -1]
    b.state[:] = 1
    a.E[:] = [100., 200., 300., 400.]
    b.Es[:] = 1j
    a.sourceWeight, b.sourceWeight = 2., 3.
    a.concatenate(b)
    assert len(a) == 7 and a.x.tolist() == [0, 1, 2, 3, 10, 11, 12]
    assert a.Es[4] == 1j and a.sourceWeight.tolist() == [2.] * 4 + [3.] * 3
    good = rs.Beam(copyFrom=a)
    good.filter_good()
    assert len(good) == 5 and (good.state == 1).all()
    c = rs.Beam(copyFrom=a)
    c.Jss[:] = 0.25
    c.absorb_intensity(a)
    assert np.allclose(c.Jss, 0.75) and c.displayAsAbsorbedPower
    a.project_energy_to_band(1000., 2000.)
    assert a.E.min() == 1000. and a.E.max() == 2000.
    d = rs.Beam(copyFrom=a)
    d.x[:] = -1.
    a.replace_by_index(np.array([0, 6]), d)
    assert a.x.tolist() == [-1, 1, 2, 3, 10, 11, -1]
    w = rs.Beam(copyFrom=a)
    w.Es[:] = 2.
    w.Ep[:] = 0.
    a.Es[:] = 1.
    a.Ep[:] = 1j
    a.add_wave(w, sign=-1)
    assert np.allclose(a.Jss, 1.) and np.allclose(a.Jsp, (-1.) * np.conj(1j))
    path = str(tmp_path / 'beam')
    a.export_beam(path)
    back = np.load(path + '.npy', allow_pickle=True).item()
    assert np.array_equal(back['x'], a.x) and 'Es' in back


def test_multilayer_thickness_profiles_and_angles():
    """Multilayer host side (materials/multilayer.py:167-240): depth grading by the power
    law through the two end thicknesses, re-laid when a parameter changes; Bragg angle of
    the period; Coated = one period without a top layer, kind 'mirror'."""
    import xrt_amd.backends.raycing.materials as rm
    from oracle import materials_np as mn
    si, w = rm.Material('Si', rho=2.33), rm.Material('W', rho=19.3)
    ml = rm.GradedMultilayer(w, 28., si, 41., 30, si, tThicknessLow=20., bThicknessLow=30.,
                             power=2.)
    ref = mn.make_multilayer(tThickness=28., bThickness=41., nPairs=30, tThicknessLow=20.,
                             bThicknessLow=30., power=2.)
    assert np.array_equal(ml.dti, ref['dti']) and np.array_equal(ml.dbi, ref['dbi'])
    assert ml.dti[0] == 28. and abs(ml.dti[-1] - 20.) < 1e-12 and ml.d == 69.
    ml.nPairs = 12
    assert len(ml.dti) == len(ml.dbi) == 12 and abs(ml.dbi[-1] - 30.) < 1e-12
    ml.tThicknessLow = 0.
    assert np.array_equal(ml.dti, np.full(12, 28.))
    assert ml.get_t_thickness(None, None, 3) == 28.
    E = np.array([8000., 12000.])
    assert np.allclose(np.sin(ml.get_Bragg_angle(E)), 12398.419297617678 / (2 * 69. * E))
    assert ml.get_sin_Bragg_angle(10.) == 1 - 1e-16
    c = rm.Coated(coating=w, cThickness=250., substrate=si, surfaceRoughness=3.,
                  substRoughness=4.)
    assert c.kind == 'mirror' and c.nPairs == 1 and c.tLayer is None and c.coating is w
    assert c.cThickness == 250. and c.dbi[0] == 250. and c.dti[0] == 0.
    c.cThickness = 300.
    assert c.dbi[0] == 300. and c.surfaceRoughness == 3.


def test_z_actuator_aperture_set_and_collimated_mesh():
    """SetOfRectangularAperturesOnZActuator.select_aperture (apertures.py:600-665) and
    CollimatedMeshSource.shine (sources/geoms.py:1111-1245) -- host arithmetic, values
    worked out by hand; run against the reference itself in
    tests/test_dropin_with_reference.py."""
    import xrt_amd.backends.raycing as raycing
    import xrt_amd.backends.raycing.apertures as ra
    import xrt_amd.backends.raycing.sources as rs
    bl = raycing.BeamLine(height=1400.)
    s = ra.SetOfRectangularAperturesOnZActuator(
        bl, 'set', [0, 1000., 1400.], ['big', 'small', 'top-edge'], [5., -4., 12.],
        [2., 0.5], [1., 0.2])
    s.select_aperture('small', 1399.5)
    assert s.blades == {'left': -0.25, 'right': 0.25, 'bottom': -0.6, 'top': -0.4}
    assert s.zActuator == 1400. + 1399.5 + 4. and s.curAperture == 1
    s.select_aperture('top-edge', 1400.)
    assert s.blades == {'bottom': 12. - 1400.} and s.zActuator == 1400.
    src = rs.CollimatedMeshSource(raycing.BeamLine(), 'c', dx=2., dz=1., nx=3, nz=2,
                                  withCentralRay=False)
    b = src.shine()
    assert src.nrays == 6
    assert np.array_equal(b.x, [-1., 0., 1., -1., 0., 1.])
    assert np.array_equal(b.z, [0.5, 0.5, 0.5, -0.5, -0.5, -0.5])
    assert not b.a.any() and not b.c.any() and np.array_equal(b.b, np.ones(6))


def test_predefined_material_catalogues():
    """materials.elemental / .compounds / .crystals (reference materials/elemental.py,
    compounds.py, crystals.py): classes built from the data extract, importable as
    submodules, taking their base class's keyword arguments."""
    import xrt_amd.backends.raycing.materials as rm
    import xrt_amd.backends.raycing.materials.elemental as xel
    import xrt_amd.backends.raycing.materials.compounds as xco
    import xrt_amd.backends.raycing.materials.crystals as xcr
    assert (len(xel.__all__), len(xco.__all__), len(xcr.__all__)) == (92, 76, 36)
    assert rm.elemental is xel and rm.crystals.Ge is xcr.Ge
    gold = xel.Au(kind='mirror')
    assert gold.rho == 19.32 and gold.kind == 'mirror' and gold.elements[0].Z == 79
    water = xco.Water()
    assert [e.name for e in water.elements] == ['H', 'O'] and water.quantities == [2., 1.]
    ge = xcr.Ge(hkl=(2, 2, 0))
    assert isinstance(ge, rm.CrystalDiamond) and abs(ge.d - ge.a / 8**0.5) < 1e-15
    quartz = xcr.AlphaQuartz(hkl=(1, 0, 2))
    assert isinstance(quartz, rm.CrystalFromCell) and len(quartz.atoms) == 9
    assert abs(quartz.d - 2.2811) < 1e-4 and abs(quartz.rho - 2.649) < 1e-3
    for name in xcr.__all__:
        getattr(xcr, name)()


def test_beam_files_round_trip_and_beam_from_file(tmp_path):
    """Beam.export_beam -> Beam(copyFrom=file) in the three formats, and the BeamFromFile
    source (sources/geoms.py:1247-1300, beams.py:122-149)."""
    import xrt_amd.backends.raycing as raycing
    import xrt_amd.backends.raycing.sources as rs
    np.random.seed(3)
    bl = raycing.BeamLine()
    src = rs.GeometricSource(bl, 'g', nrays=500, distE='flat', energies=(8000., 9000.))
    beam = src.shine(withAmplitudes=True)
    beam.state[7] = -3
    for fmt, ext in (('npy', 'npy'), ('mat', 'mat'), ('pickle', 'pickle')):
        path = str(tmp_path / ('b.' + ext))
        beam.export_beam(path, fmt)
        back = rs.Beam(copyFrom=path)
        for f in ('x', 'y', 'z', 'a', 'b', 'c', 'E', 'Jss', 'Jpp', 'Jsp', 'Es', 'Ep', 'state',
                  'path'):
            assert np.array_equal(getattr(back, f), getattr(beam, f)), (fmt, f)
        assert back.state.dtype == np.int32
    again = rs.BeamFromFile(raycing.BeamLine(), 'file', fileName=str(tmp_path / 'b.npy'))
    assert again.nrays == 500 and np.array_equal(again.shine().E, beam.E)
    assert rs.BeamFromFile(None, 'empty').nrays == raycing.nrays
