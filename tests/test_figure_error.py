"""OE(figureError=...) (reference: xrt/backends/raycing/figure_error.py; hooks on the ray path
oes/base.py:826-830 and oes/reflect.py:767-775).

CPU: the product's map generators and spline equal the reference's (the goldens hold the
spline scipy made inside the reference: knots, coefficients, the map itself); the derivative
coefficients handed to the kernels reproduce scipy's own partial derivatives; records and
refusals. GPU: the four goldens of oracle/gen_fixtures_figure.py -- the reference traced the
same elements -- states bit for bit, geometry 1e-12, amplitudes 1e-9; 1e6 rays against the
oracle with the golden's spline; hit points on the distorted surface."""
import os

import numpy as np
import pytest

import figure_cases as fc
from oracle import reflect_np as rn
from oracle.adapters import oracle_params, to_oracle_beam

GEOM = ('x', 'y', 'z', 'a', 'b', 'c', 'path')


def load(name):
    return np.load(os.path.join(fc.GOLDEN, name + '.npz'))


# ------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize('name', fc.NAMES)
def test_maps_and_splines_equal_the_references(name):
    g = load(name)
    fe = fc.figure_error(name)
    ty, tx, c = fe.local_z_spline.tck
    assert np.array_equal(ty, g['fe_ty']) and np.array_equal(tx, g['fe_tx'])
    # (the random map passes through two FFTs: pocketfft is deterministic, the same bits)
    assert np.array_equal(c, g['fe_c'])
    assert np.array_equal(fe.z2d, g['fe_z2d'])
    assert [fe.xShift, fe.yShift] == g['fe_shift'].tolist()


@pytest.mark.parametrize('name', fc.NAMES)
def test_derivative_coefficients_reproduce_scipys_partial_derivatives(name):
    """spline_arrays() forms the coefficients of d/dy and d/dx as FITPACK's parder does; a
    plain de Boor evaluation with them equals RectBivariateSpline.ev(dx=.., dy=..)."""
    from scipy.interpolate import BSpline
    fe = fc.figure_error(name)
    k, ty, tx, c, cy, cx = fe.spline_arrays()
    rng = np.random.default_rng(3)
    y = rng.uniform(ty[0] - 2., ty[-1] + 2., 400)       # also outside the map: clamped
    x = rng.uniform(tx[0] - 1., tx[-1] + 1., 400)
    yc, xc = np.clip(y, ty[0], ty[-1]), np.clip(x, tx[0], tx[-1])

    def tensor(tu, ku, tv, kv, coef, u, v):
        bu = BSpline.design_matrix(u, tu, ku, extrapolate=False).toarray()
        bv = BSpline.design_matrix(v, tv, kv, extrapolate=False).toarray()
        return np.einsum('pi,ij,pj->p', bu, coef, bv)
    spl = fe.local_z_spline
    for got, want in ((tensor(ty, k, tx, k, c, yc, xc), spl.ev(y, x)),
                      (tensor(ty[1:-1], k - 1, tx, k, cy, yc, xc), spl.ev(y, x, dx=1, dy=0)),
                      (tensor(ty, k, tx[1:-1], k - 1, cx, yc, xc), spl.ev(y, x, dx=0, dy=1))):
        assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1e-300)


def test_host_methods_and_rebuilds():
    fe = fc.figure_error('g2_figure_flat')
    g = load('g2_figure_flat')
    oe = fc.element('g2_figure_flat', g, fe)
    x, y = np.linspace(-7, 7, 50), np.linspace(-140, 140, 50)
    assert np.array_equal(oe.local_z_distorted(x, y), fe.local_z_distorted(x, y))
    d_pitch, d_roll = oe.local_n_distorted(x, y)
    assert d_pitch.shape == x.shape and np.abs(d_pitch).max() < 1e-5 and np.abs(d_roll).max() < 1e-4
    assert 3. < fe.get_rms() < 12.
    before = fe.local_z_spline.tck[2].copy()
    fe.amplitude = 12.                       # a parameter change rebuilds the spline
    assert not np.array_equal(before, fe.local_z_spline.tck[2])
    plain = fc.roe.OE(None, 'p')
    assert plain.local_z_distorted(x, y) is None and plain.local_n_distorted(x, y) is None


def test_imported_map_errors(tmp_path):
    with pytest.raises(ValueError, match='does not exist'):
        fc.rfe.FigureErrorImported(fileName=str(tmp_path / 'nothing.txt'))
    bad = tmp_path / 'two_columns.txt'
    np.savetxt(bad, np.zeros((5, 2)))
    with pytest.raises(ValueError, match='Invalid'):
        fc.rfe.FigureErrorImported(fileName=str(bad))
    empty = fc.rfe.FigureErrorImported()           # no file: a flat 5 x 5 map
    assert np.all(empty.local_z_distorted(np.zeros(3), np.zeros(3)) == 0.)


def test_the_c_abi_refuses_what_the_kernels_do_not_hold():
    import ctypes
    from xrt_amd import _lib, _structs
    lib = _lib.load(build_if_missing=False)
    lib.xrt_hip_last_error.restype = ctypes.c_char_p

    def call(kind=0, mat=_structs.MAT_MIRROR, **fields):
        p, m = _structs.Pass(), _structs.Material()
        p.surf_kind, p.invert_normal, m.kind = kind, 1, mat
        p.fe_c = p.fe_cx = p.fe_cy = p.fe_tx = p.fe_ty = 8
        p.fe_k, p.fe_ntx, p.fe_nty = 3, 12, 12
        for key, value in fields.items():
            setattr(p, key, value)
        rc = lib.xrt_hip_reflect_pass_f64_dev(ctypes.byref(p), ctypes.byref(m), None, None, None,
                                              None, None, None, ctypes.c_size_t(0), None, None,
                                              None)
        return rc, lib.xrt_hip_last_error()
    rc, why = call(fe_k=4)
    assert rc != 0 and b'degree 1..3' in why
    rc, why = call(fe_cx=0)
    assert rc != 0 and b'derivative' in why
    rc, why = call(kind=_structs.SURF_ELLIPSE_PARAM)
    assert rc != 0 and b'parametric or user-defined' in why
    rc, why = call(mat=_structs.MAT_MULTILAYER)
    assert rc != 0 and b'layered material' in why
    rc, why = call()                     # accepted: the call fails later, on the empty beams
    assert rc != 0 and b'figure' not in why


def test_linspace_knots_and_paired_rows():
    """What the kernels compute instead of loading: the knots of a generated map are a linspace's
    (bit for bit, or the flag stays down -- a map from a file with uneven steps); coefficients
    travel as pairs of rows."""
    fe = fc.figure_error('g2_figure_flat')
    k, ty, tx, c, cy, cx = fe.spline_arrays()
    for t, nodes in ((ty, fe.y1d), (tx, fe.x1d)):
        lo, step, hi = fe.linspace_of(t, k)
        assert (lo, hi) == (nodes[0], nodes[-1]) and step == (hi - lo) / (len(nodes) - 1)
        i = np.arange(4, len(t) - 4)
        assert np.array_equal((i - 2) * step + lo, t[4:-4])
    uneven = ty.copy()
    uneven[10] += 1e-9
    assert fe.linspace_of(uneven, k) is None and fe.linspace_of(ty, 2) is None
    # (the map file's 1-mm and 0.5-mm steps ARE such grids; one moved node is not)
    imported = fc.figure_error('g2_figure_imported')
    assert imported.linspace_of(imported.spline_arrays()[1], 3) == (-80., 1., 80.)
    p = fe.paired_rows(c)
    assert p.shape == c.shape + (2,) and np.array_equal(p[..., 0], c)
    assert np.array_equal(p[:-1, :, 1], c[1:]) and not p[-1, :, 1].any()


def test_pass_record_carries_the_spline():
    g = load('g2_figure_flat')
    oe = fc.element('g2_figure_flat', g)
    import torch
    if not torch.cuda.is_available():
        # (the record needs HBM: without a GPU only the refusal of foreign objects is checked)
        class Foreign(object):
            def local_z_distorted(self, x, y):
                return x * 0.
        oe.figureError = Foreign()
        from xrt_amd import _structs
        with pytest.raises(NotImplementedError, match='figure_error'):
            oe._figure_params(_structs.Pass())
        return
    rec = oe.figureError.device_record(torch.device('cuda', 0))
    assert rec['k'] == 3 and rec['nty'] == len(g['fe_ty']) and rec['ntx'] == len(g['fe_tx'])
    assert all(grid is not None for grid in rec['grid'])


# ------------------------------------------------------------------------------ GPU
def _close(got, want, tol, what):
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
    assert err <= tol, (what, err)


def _golden_beam(g):
    import xrt_amd.backends.raycing.sources as rs
    beam = rs.Beam(nrays=len(g['in_x']), withAmplitudes=True)
    for f in GEOM + ('E', 'Jss', 'Jpp', 'Jsp', 'state', 'Es', 'Ep'):
        setattr(beam, f, g['in_' + f])
    return beam


@pytest.mark.gpu
@pytest.mark.parametrize('name', fc.NAMES)
def test_elements_with_a_figure_error_match_the_reference(name):
    g = load(name)
    oe = fc.element(name, g)
    info = {}
    gb, lb = oe.reflect(_golden_beam(g), _info=info)
    for tag, out in (('gb', gb), ('lb', lb)):
        assert np.array_equal(out.state, g[tag + '_state']), tag
        for f in GEOM:
            _close(getattr(out, f), g['%s_%s' % (tag, f)], 1e-12, (tag, f))
        for f in ('Jss', 'Jpp', 'Jsp', 'Es', 'Ep'):
            _close(getattr(out, f), g['%s_%s' % (tag, f)], 1e-9, (tag, f))
    _close(lb.theta, g['lb_theta'], 1e-12, 'theta')
    assert info['axis'] == int(g['axis']) and bool(info['brent']) == bool(g['brent'])
    # the map matters: the same element without it sends the rays elsewhere
    plain = fc.element(name, g)
    plain.figureError = None
    gb0, lb0 = plain.reflect(_golden_beam(g))
    hit = g['lb_state'] == 1
    assert np.abs(gb0.c - g['gb_c'])[hit].max() > 1e-8
    assert np.abs(gb.c - g['gb_c'])[hit].max() < 1e-14


@pytest.mark.gpu
def test_1e6_rays_on_a_rough_toroid_match_the_oracle():
    """The optimistic single pass at size: every hit point lies on the DISTORTED surface, the
    outgoing directions are those of the turned normals; a 50k-ray subset against the oracle
    (its figure functions = scipy on the golden's spline)."""
    from xrt_amd import workloads
    g = load('g2_figure_toroid')
    oe = fc.element('g2_figure_toroid', g)
    n = 1_000_000
    beam = workloads.synthetic_rays(n, 17)
    t = {}
    gb, lb = oe.reflect(beam, _timing=t)
    assert not t['exact_sequence']
    good = lb.state == 1
    assert good.mean() > 0.95
    x, y, z = lb.x[good], lb.y[good], lb.z[good]
    surf = oe.local_z(x, y) + oe.local_z_distorted(x, y)
    assert np.abs(z - surf).max() < 2e-12
    assert np.abs(z - oe.local_z(x, y)).max() > 1e-6          # (nm-scale: 1e-6 mm)
    idx = np.sort(np.random.default_rng(1).choice(n, 50_000, replace=False))
    idx[0] = 0
    sub = rn.Beam(len(idx))
    for f in sub.fields():
        setattr(sub, f, beam.peek(f)[idx].copy())
    par = oracle_params(oe)
    par['surface'].update(fc.oracle_hooks(g))
    ogb, olb = rn.oe_reflect(par, sub)
    assert np.array_equal(lb.state[idx], olb.state)
    for f in GEOM:
        r = getattr(ogb, f)
        assert np.abs(getattr(gb, f)[idx] - r).max() <= 1e-12 * np.abs(r).max(), f
    scale = (ogb.Jss + ogb.Jpp).max()
    assert np.abs(gb.Jss[idx] - ogb.Jss).max() <= 1e-10 * scale


@pytest.mark.gpu
def test_map_on_an_uneven_grid_takes_the_loaded_knots(tmp_path):
    """A measured map whose nodes are NOT equally spaced (np.unique of the file's columns, as the
    reference builds its grid): the knots are no linspace's, the kernels load them (fe_interval /
    fe_basis, FITPACK's sequence as it stands) -- 20000 rays on a bent mirror against the oracle
    with scipy evaluating the same spline; and the same map on the even grid next to it takes
    the computed knots and differs."""
    import xrt_amd.backends.raycing as raycing
    import xrt_amd.backends.raycing.materials as rm
    from xrt_amd import workloads
    rng = np.random.default_rng(8)
    x = np.sort(np.concatenate([[-10., 10.], rng.uniform(-10, 10, 38)]))
    y = np.sort(np.concatenate([[-80., 80.], rng.uniform(-80, 80, 118)]))
    X, Y = np.meshgrid(x, y, indexing='ij')
    Z = 5. * np.cos(2 * np.pi * Y / 47.) * (1 + 0.1 * X) + 2. * np.sin(X)          # [nm]
    path = tmp_path / 'uneven.txt'
    np.savetxt(path, np.column_stack([X.ravel(), Y.ravel(), Z.ravel()]), fmt='%.17g')
    fe = fc.rfe.FigureErrorImported(fileName=str(path))
    k, ty, tx, *_ = fe.spline_arrays()
    assert fe.linspace_of(ty, k) is None and fe.linspace_of(tx, k) is None
    pt = rm.Material('Pt', rho=21.45, kind='mirror')
    bm = fc.roe.BentFlatMirror(raycing.BeamLine(), 'bent', center=[0, 18000., 0], pitch=3.5e-3,
                               material=pt, R=5e6, limPhysX=[-10, 10], limPhysY=[-80, 80],
                               figureError=fe)
    beam = workloads.synthetic_rays(20000, 9)
    beam.x[:] = beam.x * 20.                    # over the whole width, some rays past the edges
    gb, lb = bm.reflect(beam)
    par = oracle_params(bm)
    par['surface'] = dict(par['surface'], figure_z=fe.local_z_distorted,
                          figure_n=fe.local_n_distorted)
    ogb, olb = rn.oe_reflect(par, to_oracle_beam(beam))
    assert np.array_equal(lb.state, olb.state) and (olb.state == 1).mean() > 0.5
    assert (olb.state != 1).sum() > 100
    for f in GEOM:
        r = getattr(ogb, f)
        assert np.abs(getattr(gb, f) - r).max() <= 1e-12 * max(np.abs(r).max(), 1.), f
    good = olb.state == 1
    assert np.abs(lb.z[good] - bm.local_z(lb.x[good], lb.y[good])
                  - fe.local_z_distorted(lb.x[good], lb.y[good])).max() < 2e-12


@pytest.mark.gpu
@pytest.mark.parametrize('order', [2])
def test_spline_orders_below_three(order):
    """figure_error.py keeps the spline order settable (splineOrder, default 3): a quadratic
    spline goes through the same kernels (degree at run time, knots loaded). (Order 1 has no
    slopes in scipy -- pardeu refuses the derivative of a linear spline -- so the reference
    cannot trace it either.)"""
    import xrt_amd.backends.raycing as raycing
    import xrt_amd.backends.raycing.materials as rm
    from xrt_amd import workloads
    fe = fc.rfe.Waviness(amplitude=8., xWaveLength=6., yWaveLength=40., limPhysX=[-8, 8],
                         limPhysY=[-150, 150], gridStep=1.)
    fe.splineOrder = order
    assert fe.local_z_spline.degrees == (order, order)
    oe = fc.roe.OE(raycing.BeamLine(), 'flat', center=[0, 15000., 0], pitch=5e-3,
                   material=rm.Material('Pt', rho=21.45, kind='mirror'), limPhysX=[-8, 8],
                   limPhysY=[-150, 150], figureError=fe)
    beam = workloads.synthetic_rays(8000, 4)
    beam.x[:] = beam.x * 15.
    gb, lb = oe.reflect(beam)
    par = oracle_params(oe)
    par['surface'] = dict(par['surface'], figure_z=fe.local_z_distorted,
                          figure_n=fe.local_n_distorted)
    ogb, olb = rn.oe_reflect(par, to_oracle_beam(beam))
    assert np.array_equal(lb.state, olb.state) and (olb.state == 1).mean() > 0.5
    for f in GEOM:
        r = getattr(ogb, f)
        assert np.abs(getattr(gb, f) - r).max() <= 1e-12 * max(np.abs(r).max(), 1.), f


@pytest.mark.gpu
def test_figure_error_on_a_dcm_and_refusals():
    """Both crystals of a DCM take the map (two passes, exact sequence each); parametric
    surfaces refuse it in Python."""
    import xrt_amd.backends.raycing as raycing
    import xrt_amd.backends.raycing.materials as rm
    from xrt_amd import workloads
    fe = fc.rfe.Waviness(amplitude=5., xWaveLength=6., yWaveLength=15., limPhysX=[-10, 10],
                         limPhysY=[-30, 30], gridStep=0.5)
    dcm = workloads.cfg3_dcm()
    beam = workloads.synthetic_rays(20000, 5)
    ref = dcm.double_reflect(beam)
    dcm.figureError = fe
    out = dcm.double_reflect(beam)
    assert np.array_equal(out[0].state, ref[0].state)
    good = ref[0].state == 1
    dev = np.abs(out[0].c - ref[0].c)[good]
    assert 1e-9 < dev.max() < 1e-4
    # ... and against the oracle with the map on both crystals (find_dz adds it whatever the
    # surface function is, base.py:826-830; the normal of either crystal is turned)
    hooks = dict(figure_z=fe.local_z_distorted, figure_n=fe.local_n_distorted)
    par = oracle_params(dcm)
    par['surface'] = dict(par['surface'], **hooks)
    par['surface2'] = dict(par['surface2'], **hooks)
    o2, o1l, o2l = rn.dcm_double_reflect(par, to_oracle_beam(beam))
    for got, want in ((out[0], o2), (out[1], o1l), (out[2], o2l)):
        assert np.array_equal(got.state, want.state)
        for f in GEOM:
            r = getattr(want, f)
            # (the local z of a flat crystal IS the map, nanometres: held to the solver's zEps)
            assert np.abs(getattr(got, f) - r).max() <= 1e-12 * max(np.abs(r).max(), 1.), f
        scale = (want.Jss + want.Jpp).max()
        assert np.abs(got.Jss - want.Jss).max() <= 1e-10 * scale
    em = fc.roe.EllipticalMirrorParam(raycing.BeamLine(), 'e', center=[0, 10000., 0], pitch=4e-3,
                                      p=10000., q=1000., material=rm.Material('Pt', rho=21.45),
                                      figureError=fe)
    with pytest.raises(NotImplementedError, match='parametric'):
        em.reflect(workloads.synthetic_rays(1000, 5))


def test_imported_map_on_top_of_a_base_map(tmp_path):
    """FigureErrorImported(fileName=..., baseFE=...): the base map is added (it was
    dropped without a word). The measured heights plus the base map's on the same grid; with
    the reference at hand (build container), its map bit for bit."""
    from oracle import _refenv
    x = np.linspace(-10, 10, 41)
    y = np.linspace(-80, 80, 161)
    X, Y = np.meshgrid(x, y, indexing='ij')
    Z = 5. * np.cos(2 * np.pi * Y / 47.) * (1 + 0.1 * X)
    path = tmp_path / 'map.txt'
    np.savetxt(path, np.column_stack([X.ravel(), Y.ravel(), Z.ravel()]), fmt='%.17g')

    def maps(module):
        bump = module.GaussianBump(bumpHeight=30., sigmaX=3., sigmaY=20., limPhysX=[-10, 10],
                                   limPhysY=[-80, 80], gridStep=0.5)
        both = module.FigureErrorImported(fileName=str(path), baseFE=bump)
        alone = module.FigureErrorImported(fileName=str(path))
        return bump, both, alone
    bump, both, alone = maps(fc.rfe)
    assert both.baseFE is bump
    under = bump.local_z_distorted(both.x2d, both.y2d) * 1e6
    assert np.abs(both.z2d - alone.z2d).max() > 29. and \
        np.allclose(both.z2d - alone.z2d, under, rtol=0, atol=1e-9)
    if _refenv.available():
        _refenv.activate()
        import xrt.backends.raycing.figure_error as ref
        assert np.array_equal(maps(ref)[1].z2d, both.z2d)
