"""GPU parity tests of the Fresnel-Kirchhoff kernel, through the C ABI.

Bar (BASELINE.md / SURVEY 0.5): field amplitudes within 1e-5 relative to the
array maximum (norm-wise). Because r and k*r are computed with numpy's exact
operation order the observed error is ~1e-12; the tests assert 1e-9 so that an
accidental FMA contraction on the phase path (which costs ~1e-5) is caught.
"""
import os

import numpy as np
import pytest
import torch

from oracle import kirchhoff_np as kn
from oracle.consts import CHBAR

pytestmark = pytest.mark.gpu
TOL = 1e-9
CASES = ['g4_slit_2000x32', 'g4_slit_4000x48', 'g4_toroid_3000x24']


def dev(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device='cuda')


def run_hip(px, py, pz, sx, sy, sz, n, nl, E, Es, Ep, **kw):
    from xrt_amd import hipcalls
    k = E / CHBAR * 1e7
    n = [np.broadcast_to(np.asarray(c, dtype=float), sx.shape) for c in n]
    out = hipcalls.kirchhoff(
        dev(px), dev(py), dev(pz), dev(sx), dev(sy), dev(sz), dev(n[0]),
        dev(n[1]), dev(n[2]), dev(nl), dev(k), dev(Es, torch.complex128),
        dev(Ep, torch.complex128), **kw)
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in out[:5]]


def assert_close(mine, ref, tol=TOL):
    for m, r in zip(mine, ref):
        scale = max(np.abs(r).max(), 1e-300)
        err = np.abs(m - r).max() / scale
        assert err <= tol, err


def golden_inputs(g):
    good = g['s_state'] == 1
    n = [g['n'][i][good] for i in range(3)]
    return (g['px'], g['py'], g['pz'], g['s_x'][good], g['s_y'][good],
            g['s_z'][good], n, g['nl'][good], g['s_E'][good], g['s_Es'][good],
            g['s_Ep'][good])


# ---- building blocks -------------------------------------------------------
def test_sqrt_is_correctly_rounded_and_rinv_accurate():
    from xrt_amd import hipcalls
    rng = np.random.default_rng(1)
    x = np.concatenate([
        rng.uniform(1e7, 2e8, 2_000_000),          # r^2 of the Kirchhoff geometry
        10.0 ** rng.uniform(-200, 200, 1_000_000),
        np.nextafter(np.arange(1, 200001, dtype=float)**2, 0),   # just below squares
        np.arange(1, 200001, dtype=float)**2])
    r, ri = hipcalls.debug_sqrt(dev(x))
    r = r.cpu().numpy()
    ri = ri.cpu().numpy()
    assert np.array_equal(r, np.sqrt(x))           # IEEE: numpy sqrt is correctly rounded
    assert np.abs(ri * np.sqrt(x) - 1).max() < 1e-14   # one Goldschmidt step on v_rsq_f64


def test_seeded_sqrt_is_correctly_rounded_at_the_seed_error_limit():
    """The fast-geometry loop seeds the root with 1/|dy| instead of v_rsq_f64; it is
    used while the seed is within 2^-27 of 1/sqrt(x) (kirchhoff_fast). Checked here at
    twice that error, both signs, on r^2 of the Kirchhoff geometry and on hard cases."""
    from xrt_amd import hipcalls
    rng = np.random.default_rng(21)
    sq = np.arange(3000, 203000, dtype=float)**2
    x = np.concatenate([rng.uniform(1e7, 2e8, 4_000_000), rng.uniform(1., 4., 2_000_000),
                        np.nextafter(sq, 0), sq, np.nextafter(sq, np.inf)])
    for err in (2.**-26, -2.**-26, 2.**-30, 0.):
        seed = (1. + err * rng.uniform(0.5, 1., x.size)) / np.sqrt(x)
        r, h = hipcalls.debug_sqrt_seeded(dev(x), dev(seed))
        assert np.array_equal(r.cpu().numpy(), np.sqrt(x)), err
        assert np.abs(2. * h.cpu().numpy() * np.sqrt(x) - 1).max() < 1e-14


@pytest.mark.parametrize('table', [0, 1, 2])
def test_sincos_of_large_phases(table):
    """The sincos forms of the Kirchhoff kernel: the general one (|phi| < 2^50) and the
    LDS-table ones it takes when the whole launch has |k r| < 2^42 -- 2048 entries, or
    4096 (cos of the remainder to second order: 1.5e-14) with four points per lane."""
    from xrt_amd import hipcalls
    rng = np.random.default_rng(2)
    top = {0: 1e14, 1: 4e12, 2: 2e12}[table]      # 2^50, 2^42, 2^41 rad
    phi = np.concatenate([rng.uniform(0, 1e12, 2_000_000),
                          rng.uniform(-top, top, 500_000),
                          rng.uniform(-1e6, 1e6, 500_000),
                          rng.uniform(-10, 10, 500_000),
                          np.arange(-4096, 4097) * (np.pi / 1024),   # table nodes
                          (np.arange(-4096, 4097) + 0.5) * (np.pi / 1024),
                          np.arange(-8192, 8193) * (np.pi / 2048),    # nodes of the 4096 table
                          (np.arange(-8192, 8193) + 0.5) * (np.pi / 2048),
                          np.arange(-64, 65) * (np.pi / 4)])
    s, c = hipcalls.debug_sincos(dev(phi), table=table)
    s = s.cpu().numpy()
    c = c.cpu().numpy()
    # glibc sin/cos are < 1 ulp with exact argument reduction
    # (4096 entries, second-order cos: (pi/4096)^4/24 = 1.4e-14 at mid-step, up to 1.8 x
    # that at the largest phases, where the low word of N/2pi shifts the remainder by 0.08)
    tol = {0: 5e-16, 1: 6e-16, 2: 2.7e-14}[table]
    assert np.abs(s - np.sin(phi)).max() < tol
    assert np.abs(c - np.cos(phi)).max() < tol


# ---- golden vectors from the reference --------------------------------------
@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('ppt', [1, 2, 4])
def test_matches_reference_golden(golden_dir, name, ppt):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    mine = run_hip(*golden_inputs(g), ppt=ppt)
    assert_close(mine, g['raw'])


def test_opencl_convention(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g4_slit_2000x32.npz'))
    mine = run_hip(*golden_inputs(g), convention=1)
    assert_close(mine, kn.to_cl_convention(*g['raw']))


# ---- seeded random inputs vs the oracle, ragged shapes and splits ----------
def random_case(npix, ns, seed, ep_zero=False, plane=False, axis_y=False, one_k=False,
                mesh=0, dist=10000.):
    """plane: receiving points on y = const; axis_y: every normal (0, ny, 0);
    one_k: one photon energy; mesh: row length of an x-fastest receiving mesh."""
    rng = np.random.default_rng(seed)
    if mesh:
        rows = -(-npix // mesh)
        gx, gz = np.meshgrid(np.sort(rng.uniform(-0.5, 0.5, mesh)),
                             np.sort(rng.uniform(-0.5, 0.5, rows)))
        px, pz = gx.ravel()[:npix].copy(), gz.ravel()[:npix].copy()
    else:
        px = rng.uniform(-0.5, 0.5, npix)
        pz = rng.uniform(-0.5, 0.5, npix)
    py = np.full(npix, dist) if plane else dist + rng.uniform(-1, 1, npix)
    sx = rng.uniform(-0.1, 0.1, ns)
    sz = rng.uniform(-0.1, 0.1, ns)
    sy = rng.uniform(-0.01, 0.01, ns)
    nrm = rng.normal(size=(3, ns)) * 0.01 + np.array([[0.], [1.], [0.]])
    nrm /= np.sqrt((nrm**2).sum(axis=0))
    if axis_y:
        nrm[0] = nrm[2] = 0.
    nl = rng.uniform(0.9, 1.0, ns)
    E = np.full(ns, 7900.) if one_k else rng.uniform(7899.5, 7900.5, ns)
    Es = rng.normal(size=ns) + 1j * rng.normal(size=ns)
    Ep = np.zeros(ns, dtype=complex) if ep_zero else \
        0.3 * (rng.normal(size=ns) + 1j * rng.normal(size=ns))
    return px, py, pz, sx, sy, sz, list(nrm), nl, E, Es, Ep


# ---- every loop of kirchhoff_stream against the oracle ----------------------
# name -> (random_case options, run options, variant that must have run)
NO_FAST, NO_SHARE = 0x100, 0x200
LOOPS = {
    'gen_s_y': (dict(ep_zero=True, axis_y=True), 0),
    'gen_s_n': (dict(ep_zero=True), 0),
    'gen_sp_y': (dict(axis_y=True), 0),
    'gen_sp_n': (dict(), 0),
    'gen_s_notab': (dict(ep_zero=True, dist=3e5), 0),
    'gen_sp_notab': (dict(dist=3e5), 0),
    'fast_s': (dict(ep_zero=True, axis_y=True, plane=True), 0),
    'fast_s_unik': (dict(ep_zero=True, axis_y=True, plane=True, one_k=True), 0),
    'fast_sp': (dict(axis_y=True, plane=True), 0),
    'fast_s_share': (dict(ep_zero=True, axis_y=True, plane=True, mesh=96), 0),
    'fast_s_share_unik': (dict(ep_zero=True, axis_y=True, plane=True, one_k=True,
                               mesh=96), 0),
    'fast_sp_share': (dict(axis_y=True, plane=True, mesh=96), 0),
    'fast_s_notab': (dict(ep_zero=True, axis_y=True, plane=True, dist=3e5), 0),
    'fast_s_notab_unik': (dict(ep_zero=True, axis_y=True, plane=True, one_k=True,
                               dist=3e5), 0),
    'fast_sp_notab': (dict(axis_y=True, plane=True, dist=3e5), 0),
    # the same planar inputs kept off the specialised loops
    'gen_s_y/no_fast': (dict(ep_zero=True, axis_y=True, plane=True, one_k=True,
                             mesh=96), NO_FAST),
    'gen_sp_y/no_fast': (dict(axis_y=True, plane=True, mesh=96), NO_FAST),
    'fast_s_unik/no_share': (dict(ep_zero=True, axis_y=True, plane=True, one_k=True,
                                  mesh=96), NO_SHARE),
}


@pytest.mark.parametrize('ppt', [1, 2, 4])
@pytest.mark.parametrize('loop', sorted(LOOPS))
def test_every_loop_variant_matches_oracle(loop, ppt):
    from xrt_amd import hipcalls
    opts, knobs = LOOPS[loop]
    want = loop.split('/')[0]
    if ppt == 1 and 'share' in want:
        want = want.replace('_share', '')        # one point per lane has nothing to share
    case = random_case(96 * 37 + 5, 700, seed=31 + ppt, **opts)
    ref = kn.kirchhoff_conv(*case)
    mine = run_hip(*case, ppt=ppt | knobs, nsplit=8)
    rep = hipcalls.kirchhoff_report()
    assert rep['variants'] == {want}, rep
    assert_close(mine, ref)


# ---- the opt-in relaxed loops (include/xrt_hip.h XRT_HIP_KIRCHHOFF_RELAXED) --
# What the relaxed loops can and cannot keep: at these distances k r ~ 4e11 rad, one ulp of
# which is 6e-5 rad. numpy rounds r and then k r; ANY other rounding of either -- the contracted
# d.d, the root one correction short, the unrounded product -- moves a pair's phase by up to
# that much (as numpy's own roundings do against the true product). On sums of samples with
# RANDOM amplitudes (these cases) the result then differs by the same 1e-5..1e-4 norm-wise, on
# the smooth fields of a beamline by 1e-6..1e-7 (bench.py reports both). The exact loops stay at
# 1e-12 only because they are numpy's operations bit for bit. So relaxed is NOT inside the 1e-5
# parity bar for incoherent sums at hard-X-ray distances: it is an opt-in for throughput.
RELAXED = 0x400



def relaxed_tol(case):
    """Three ulps of the largest phase k r of the case, norm-wise (exact mode: 1e-9)."""
    px, py, pz, sx, sy, sz, n, nl, E = case[:9]
    reach = np.abs(np.asarray(py)).max() + 1.
    return 3 * 2.**-52 * (np.max(E) / CHBAR * 1e7) * reach


def _normwise(mine, ref):
    return max(np.abs(m - r).max() / max(np.abs(r).max(), 1e-300) for m, r in zip(mine, ref))


@pytest.mark.parametrize('ppt', [1, 2, 4])
@pytest.mark.parametrize('loop,opts', [('gen_s_n', dict(ep_zero=True)), ('gen_sp_n', dict()),
                                       ('gen_sp_n', dict(dist=40000., one_k=True))])
def test_relaxed_loops_against_the_oracle(loop, opts, ppt, record_property):
    """General normals take the relaxed loop when asked; what they lose is REPORTED (pytest
    -rP / junit property) and bounded (see RELAXED_TOL above); the exact call on the same
    inputs stays at 1e-9 and runs the exact loop."""
    from xrt_amd import hipcalls
    case = random_case(96 * 37 + 5, 3000, seed=77 + ppt, **opts)
    ref = kn.kirchhoff_conv(*case)
    mine = run_hip(*case, ppt=ppt | RELAXED, nsplit=8)
    assert hipcalls.kirchhoff_report()['variants'] == {loop + '_relaxed'}
    err = _normwise(mine, ref)
    record_property('relaxed_normwise_error', err)
    print('relaxed %s ppt %d: norm-wise error %.2e' % (loop, ppt, err))
    assert err <= relaxed_tol(case), (err, relaxed_tol(case))
    exact = run_hip(*case, ppt=ppt, nsplit=8)
    assert hipcalls.kirchhoff_report()['variants'] == {loop}
    assert_close(exact, ref)
    # through the keyword of the Python call
    again = run_hip(*case, ppt=ppt, nsplit=8, relaxed=True)
    assert all(np.array_equal(a, b) for a, b in zip(again, mine))


def test_relaxed_leaves_the_other_loops_alone():
    """Normals along y (apertures, screens, sources: cfg4's geometry) have no relaxed form:
    the flag changes nothing, bit for bit -- neither on the fast loops nor with NO_FAST."""
    from xrt_amd import hipcalls
    for opts, knobs, want in ((dict(ep_zero=True, axis_y=True, plane=True, one_k=True, mesh=96),
                               0, 'fast_s_share_unik'),
                              (dict(axis_y=True), 0, 'gen_sp_y'),
                              (dict(dist=3e5), 0, 'gen_sp_notab')):
        case = random_case(96 * 21, 500, seed=5, **opts)
        a = run_hip(*case, ppt=2 | knobs, nsplit=4)
        b = run_hip(*case, ppt=2 | knobs | RELAXED, nsplit=4)
        assert hipcalls.kirchhoff_report()['variants'] == {want}
        assert all(np.array_equal(u, v) for u, v in zip(a, b))


def test_relaxed_on_the_reference_golden(golden_dir, record_property):
    """G4 toroid -> screen (samples on a mirror: general normals) in relaxed mode against the
    reference's own integrals."""
    g = np.load(os.path.join(golden_dir, 'g4_toroid_3000x24.npz'))
    mine = run_hip(*golden_inputs(g), relaxed=True)
    err = _normwise(mine, g['raw'])
    record_property('relaxed_normwise_error_g4_toroid', err)
    print('relaxed, g4_toroid_3000x24: norm-wise error %.2e' % err)
    assert err <= relaxed_tol(golden_inputs(g))


def test_planar_but_wide_angle_keeps_the_general_loop():
    """receiving plane 2 mm from the samples: 1/|dy| is no seed for the root"""
    from xrt_amd import hipcalls
    case = random_case(3000, 500, seed=41, ep_zero=True, axis_y=True, plane=True,
                       mesh=100, dist=2.)
    mine = run_hip(*case, ppt=2)
    assert hipcalls.kirchhoff_report()['variants'] == {'gen_s_y'}
    assert_close(mine, kn.kirchhoff_conv(*case))


def test_mesh_with_broken_rows_falls_back_per_wave():
    """px repeats with period 96 except in one row: the waves holding that row take the
    unshared loop, the result is the same"""
    from xrt_amd import hipcalls
    case = list(random_case(96 * 40, 600, seed=42, ep_zero=True, axis_y=True, plane=True,
                            one_k=True, mesh=96))
    case[0] = case[0].copy()
    case[0][96 * 7 + 13] += 1e-3
    mine = run_hip(*case, ppt=2)
    assert hipcalls.kirchhoff_report()['variants'] == {'fast_s_share_unik', 'fast_s_unik'}
    assert_close(mine, kn.kirchhoff_conv(*case))


@pytest.mark.parametrize('npix,ns,nsplit,ppt', [
    (1, 1, 0, 1), (1, 500, 0, 1), (63, 65, 0, 2), (257, 1000, 1, 1),
    (513, 777, 8, 2), (1000, 129, 0, 1), (300, 2048, 16, 1), (2049, 100, 3, 2)])
def test_ragged_shapes_match_oracle(npix, ns, nsplit, ppt):
    case = random_case(npix, ns, seed=npix * 7919 + ns)
    ref = kn.kirchhoff_conv(*case)
    mine = run_hip(*case, nsplit=nsplit, ppt=ppt)
    assert_close(mine, ref)


def test_empty_inputs():
    case = random_case(16, 0, seed=3)
    mine = run_hip(*case)
    for m in mine:
        assert m.shape == (16,) and not m.any()
    case = random_case(0, 16, seed=4)
    mine = run_hip(*case)
    assert all(m.shape == (0,) for m in mine)


def test_constant_scalar_normal_like_aperture():
    case = list(random_case(200, 300, seed=5))
    case[6] = [0, 1, 0]                      # waves.py:687-689: n = [0, 1, 0]
    ref = kn.kirchhoff_conv(*case)
    mine = run_hip(*case)
    assert_close(mine, ref)


def test_split_results_are_deterministic():
    case = random_case(512, 4096, seed=6)
    a = run_hip(*case, nsplit=8)
    b = run_hip(*case, nsplit=8)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


# ---- host entry point with the reference's OpenCL marshalling --------------
def _marshal_like_reference(case):
    """Exactly what _diffraction_integral_CL builds (waves.py:860-890)."""
    px, py, pz, sx, sy, sz, n, nl, E, Es, Ep = case
    ns = len(sx)
    n = [np.broadcast_to(np.asarray(c, dtype=float), sx.shape) for c in n]
    k = E / CHBAR * 1e7
    scalarArgs = [np.int32(ns)]
    slicedRO = [np.float64(px), np.float64(py), np.float64(pz)]
    nonSlicedRO = [np.float64(nl), np.complex128(Es), np.complex128(Ep),
                   np.float64(k),
                   np.array([sx, sy, sz, 0 * sz], order='F', dtype=np.float64),
                   np.array([n[0], n[1], n[2], 0 * n[2]], order='F',
                            dtype=np.float64)]
    slicedRW = [np.zeros(len(px), dtype=np.complex128) for _ in range(5)]
    return scalarArgs, slicedRO, nonSlicedRO, slicedRW


@pytest.mark.parametrize('devices', [[0], [0, 0, 0]])
def test_run_parallel_dropin(devices):
    from xrt_amd.backends.raycing.myhip import XRT_HIP
    case = random_case(1001, 900, seed=8)
    ref = kn.kirchhoff_conv(*case)
    cl = XRT_HIP(devices=devices)                      # pixel range split over "devices"
    assert cl.lastTargetOpenCL is not None
    sa, sro, nsro, srw = _marshal_like_reference(case)
    res = cl.run_parallel('integrate_kirchhoff', sa, sro, nsro, srw, None, len(case[0]))
    assert len(res) == 5 and all(r is w for r, w in zip(res, srw))   # in place AND returned
    assert_close(res, kn.to_cl_convention(*ref))
    cl2 = XRT_HIP(convention='numpy', devices=devices)
    sa, sro, nsro, srw = _marshal_like_reference(case)
    res = cl2.run_parallel('integrate_kirchhoff', sa, sro, nsro, srw, None, len(case[0]))
    assert_close(res, ref)
    with pytest.raises(NotImplementedError):
        cl.run_parallel('undulator_nf_byparts', [], [], [], [], None, 1)   # no caller in the reference


# ---- BASELINE-size properties (the oracle cannot run these sizes) -----------
def test_full_size_additivity_and_linearity():
    """cfg4 size (1e6 samples x 512x512): the integral is linear in the sample
    set, K(S1 u S2) = K(S1) + K(S2), and in the field, K(2 Es) = 2 K(Es)."""
    npix, ns = 512 * 512, 1_000_000
    case = random_case(npix, ns, seed=9)
    full = run_hip(*case)
    half = ns // 2
    cut = lambda c, sl: [c[0], c[1], c[2]] + [  # noqa: E731
        a[sl] if not isinstance(a, list) else [b[sl] for b in a] for a in c[3:]]
    a = run_hip(*cut(case, slice(0, half)))
    b = run_hip(*cut(case, slice(half, ns)))
    for f, x, y in zip(full, a, b):
        scale = np.abs(f).max()
        assert np.abs(f - (x + y)).max() <= 1e-11 * scale
    case2 = list(case)
    case2[9] = 2 * case[9]
    case2[10] = 2 * case[10]
    dbl = run_hip(*case2)
    for f, d in zip(full, dbl):
        assert np.abs(d - 2 * f).max() <= 1e-13 * np.abs(f).max()
    # spot-check 64 pixels of the full-size result against the oracle
    idx = np.random.default_rng(10).choice(npix, 64, replace=False)
    sub = list(case)
    sub[0], sub[1], sub[2] = case[0][idx], case[1][idx], case[2][idx]
    ref = kn.kirchhoff_conv(*sub)
    for f, r in zip(full, ref):
        assert np.abs(f[idx] - r).max() <= TOL * np.abs(f).max()


def _on_device(h):
    n = [np.broadcast_to(np.asarray(c, dtype=float), h['sx'].shape) for c in h['n']]
    return [dev(h[f]) for f in ('px', 'py', 'pz', 'sx', 'sy', 'sz')] + \
        [dev(c) for c in n] + [dev(h['nl']), dev(h['k']),
                               dev(h['Es'], torch.complex128), dev(h['Ep'], torch.complex128)]


def _oracle_on_pixels(h, idx):
    return kn.kirchhoff_conv(h['px'][idx], h['py'][idx], h['pz'][idx], h['sx'], h['sy'],
                             h['sz'], h['n'], h['nl'], h['E'], h['Es'], h['Ep'])


@pytest.mark.parametrize('ppt', [0, 1, 2, 4])
def test_cfg4_workload_itself_against_the_oracle(ppt):
    """The very inputs bench.py times (workloads.kirchhoff_case(4): 1e6 samples,
    512 x 512 mesh, Ep = 0, one energy, planar): 64 receiving points of the full-size
    launch against the oracle, and the loop the launch took."""
    from xrt_amd import hipcalls, workloads
    h = workloads.kirchhoff_case(4)
    out = hipcalls.kirchhoff(*_on_device(h), ppt=ppt)
    torch.cuda.synchronize()
    rep = hipcalls.kirchhoff_report()
    assert rep['row'] == 512
    assert rep['variants'] == ({'fast_s_unik'} if ppt == 1 else {'fast_s_share_unik'}), rep
    idx = np.random.default_rng(4).choice(h['px'].size, 64, replace=False)
    ref = _oracle_on_pixels(h, idx)
    for o, r in zip(out, ref):
        o = o.cpu().numpy()
        assert np.abs(o[idx] - r).max() <= TOL * max(np.abs(o).max(), 1e-300)


@pytest.mark.timeout(900)
def test_cfg5_on_one_gpu():
    """cfg5 (4e6 samples x 2048 x 2048) on one GPU: additive over two halves of the
    sample set, 64 receiving points against the oracle."""
    from xrt_amd import hipcalls, workloads
    h = workloads.kirchhoff_case(5)
    d = _on_device(h)
    full = [o.clone() for o in hipcalls.kirchhoff(*d)]
    assert hipcalls.kirchhoff_report()['variants'] == {'fast_s_share_unik'}
    ns = h['ns']
    parts = []
    for sl in (slice(0, ns // 2), slice(ns // 2, ns)):
        part = hipcalls.kirchhoff(*(d[:3] + [a[sl].contiguous() for a in d[3:]]))
        parts.append([o.clone() for o in part])
    for f, a, b in zip(full, *parts):
        scale = float(f.abs().max())
        assert float((f - (a + b)).abs().max()) <= 1e-11 * max(scale, 1e-300)
    idx = np.random.default_rng(5).choice(h['px'].size, 64, replace=False)
    ref = _oracle_on_pixels(h, idx)
    for f, r in zip(full, ref):
        f = f.cpu().numpy()
        assert np.abs(f[idx] - r).max() <= TOL * max(np.abs(f).max(), 1e-300)


@pytest.mark.parametrize('scale,ppt', [(1., 1), (30., 2), (2000., 1), (6., 4), (3., 4)])
def test_phase_magnitudes_across_the_table_bound(scale, ppt):
    """|k r| < 2^42 goes through the LDS-table sincos, larger phases (hard X-rays over
    long distances) through the general one; the switch is per wave from
    max k (|p|_1 + max |s|_1). scale 1: 4e11 rad (table); 30: 1.2e13 (general);
    2000: 8e14, where one ulp of k r is already 0.1 rad - the reference rounds k r
    the same way, so the results still agree."""
    px, py, pz, sx, sy, sz, nrm, nl, E, Es, Ep = random_case(300, 400, seed=11)
    case = (px, py * scale, pz, sx, sy, sz, nrm, nl, E, Es, Ep)
    ref = kn.kirchhoff_conv(*case)
    mine = run_hip(*case, ppt=ppt)
    assert_close(mine, ref, tol=1e-9)


def test_mixed_waves_on_both_sides_of_the_table_bound():
    """pixels near and far in one launch: some waves take the table, others not"""
    px, py, pz, sx, sy, sz, nrm, nl, E, Es, Ep = random_case(1024, 300, seed=12)
    py = py.copy()
    py[512:] *= 40.
    case = (px, py, pz, sx, sy, sz, nrm, nl, E, Es, Ep)
    assert_close(run_hip(*case), kn.kirchhoff_conv(*case), tol=1e-9)
