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


@pytest.mark.parametrize('table', [False, True])
def test_sincos_of_large_phases(table):
    """Both sincos forms of the Kirchhoff kernel: the general one (|phi| < 2^50) and
    the LDS-table one it takes when the whole launch has |k r| < 2^42."""
    from xrt_amd import hipcalls
    rng = np.random.default_rng(2)
    top = 4e12 if table else 1e14
    phi = np.concatenate([rng.uniform(0, 1e12, 2_000_000),
                          rng.uniform(-top, top, 500_000),
                          rng.uniform(-1e6, 1e6, 500_000),
                          rng.uniform(-10, 10, 500_000),
                          np.arange(-4096, 4097) * (np.pi / 1024),   # table nodes
                          (np.arange(-4096, 4097) + 0.5) * (np.pi / 1024),
                          np.arange(-64, 65) * (np.pi / 4)])
    s, c = hipcalls.debug_sincos(dev(phi), table=table)
    s = s.cpu().numpy()
    c = c.cpu().numpy()
    # glibc sin/cos are < 1 ulp with exact argument reduction
    tol = 6e-16 if table else 5e-16
    assert np.abs(s - np.sin(phi)).max() < tol
    assert np.abs(c - np.cos(phi)).max() < tol


# ---- golden vectors from the reference --------------------------------------
@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('ppt', [1, 2])
def test_matches_reference_golden(golden_dir, name, ppt):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    mine = run_hip(*golden_inputs(g), ppt=ppt)
    assert_close(mine, g['raw'])


def test_opencl_convention(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g4_slit_2000x32.npz'))
    mine = run_hip(*golden_inputs(g), convention=1)
    assert_close(mine, kn.to_cl_convention(*g['raw']))


# ---- seeded random inputs vs the oracle, ragged shapes and splits ----------
def random_case(npix, ns, seed, ep_zero=False):
    rng = np.random.default_rng(seed)
    px = rng.uniform(-0.5, 0.5, npix)
    pz = rng.uniform(-0.5, 0.5, npix)
    py = 10000. + rng.uniform(-1, 1, npix)
    sx = rng.uniform(-0.1, 0.1, ns)
    sz = rng.uniform(-0.1, 0.1, ns)
    sy = rng.uniform(-0.01, 0.01, ns)
    nrm = rng.normal(size=(3, ns)) * 0.01 + np.array([[0.], [1.], [0.]])
    nrm /= np.sqrt((nrm**2).sum(axis=0))
    nl = rng.uniform(0.9, 1.0, ns)
    E = rng.uniform(7899.5, 7900.5, ns)
    Es = rng.normal(size=ns) + 1j * rng.normal(size=ns)
    Ep = np.zeros(ns, dtype=complex) if ep_zero else \
        0.3 * (rng.normal(size=ns) + 1j * rng.normal(size=ns))
    return px, py, pz, sx, sy, sz, list(nrm), nl, E, Es, Ep


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
        cl.run_parallel('get_trajectory', [], [], [], [], None, 1)


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


@pytest.mark.parametrize('scale,ppt', [(1., 1), (30., 2), (2000., 1)])
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
