"""GPU: the HIP undulator field sums (xrt_amd/csrc/undulator.hip, C ABI
xrt_hip_undulator_f64[_dev]) against the golden vectors of the reference's
numpy path (tests/golden/g9_undulator_*.npz) and the numpy oracle.

Tolerance: the sums are fp64 with the reference's operation order; what
differs is sin/cos (~2e-16 absolute here, <= 1 ulp in numpy) entering a sum of
O(50..640) terms -> norm-wise 1e-12 (north_star asks 1e-5 for field
amplitudes)."""
import os

import numpy as np
import pytest
import torch

from oracle import undulator_np as un

pytestmark = pytest.mark.gpu
CASES = ['far_planar', 'far_helical', 'taper', 'nf']
TOL = 1e-12


def load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, 'g9_undulator_%s.npz' % tag))


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def run_dev(g, sl=slice(None)):
    from xrt_amd import hipcalls
    tabs = [dev(g[k]) for k in ('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph')]
    rays = [dev(g[k][sl]) for k in ('gamma', 'wu', 'w', 'ww1', 'ddphi', 'ddpsi')]
    tv = float(g['taperVal'])
    Is, Ip = hipcalls.undulator(
        int(g['mode']), float(g['Kx']), float(g['Ky']), tabs, *rays,
        nper=int(g['Np']), alpha_s=0. if np.isnan(tv) else tv / un.E2WC,
        r0z=float(g['r0z']))
    torch.cuda.synchronize()
    return Is.cpu().numpy(), Ip.cpu().numpy()


@pytest.mark.parametrize('tag', CASES)
def test_device_sums_match_reference_golden(golden_dir, tag):
    g = load(golden_dir, tag)
    Is, Ip = run_dev(g)
    assert rel(Is, g['Is']) < TOL, rel(Is, g['Is'])
    assert rel(Ip, g['Ip']) < TOL, rel(Ip, g['Ip'])
    # per-ray worst case, relative to the largest amplitude in the batch
    scale = max(np.abs(g['Is']).max(), np.abs(g['Ip']).max())
    assert np.abs(Is - g['Is']).max() < 1e-11 * scale
    assert np.abs(Ip - g['Ip']).max() < 1e-11 * scale


@pytest.mark.parametrize('tag', CASES)
def test_run_parallel_dropin_matches_device_path(golden_dir, tag):
    """The XRT_CL-shaped entry (host arrays, reference marshalling of
    synchr.py:2132-2160) gives bit-identical numbers to the device path and
    fills the caller's arrays in place."""
    from xrt_amd.backends.raycing.myhip import XRT_HIP
    g = load(golden_dir, tag)
    mode = int(g['mode'])
    name = ('undulator', 'undulator_taper', 'undulator_nf')[mode]
    first = (0., float(g['taperVal']), float(g['r0z']))[mode]
    scalar = [np.float64(first), np.float64(g['Kx']), np.float64(g['Ky']),
              np.int32(len(g['tg']))]
    if mode:
        scalar.append(np.int32(g['Np']))
    n = len(g['w'])
    rw = [np.zeros(n, dtype=np.complex128), np.zeros(n, dtype=np.complex128)]
    hip = XRT_HIP()
    out = hip.run_parallel(
        name, scalar, [g[k] for k in ('gamma', 'wu', 'w', 'ww1', 'ddphi', 'ddpsi')],
        [g[k] for k in ('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph')], rw,
        dimension=n)
    assert out[0] is rw[0] and out[1] is rw[1]
    Is, Ip = run_dev(g)
    assert np.array_equal(rw[0], Is) and np.array_equal(rw[1], Ip)
    assert hip.lastKernelMs > 0


@pytest.mark.parametrize('tag', CASES)
def test_rays_are_independent_of_the_batch(golden_dir, tag):
    g = load(golden_dir, tag)
    Is, Ip = run_dev(g)
    Is2, Ip2 = run_dev(g, slice(100, 357))
    assert np.array_equal(Is[100:357], Is2) and np.array_equal(Ip[100:357], Ip2)


def test_scaled_intensity_map_against_reference(golden_dir):
    """(I, Es, Ep) of _build_I_map_conv from the device sums."""
    g = load(golden_dir, 'far_helical')
    Is, Ip = run_dev(g)
    bw = 0.001
    a2f = un.FINE_STR * bw * float(g['eI']) / un.SIE0
    ds = float(g['dstep'])
    I = a2f * g['ab']**2 * 0.25 * ds**2 * (np.abs(Is)**2 + np.abs(Ip)**2)
    Es = np.sqrt(a2f) * g['ab'] * Is * 0.5 * ds
    assert rel(I, g['I']) < TOL and rel(Es, g['Es']) < TOL


def test_large_batch_against_oracle_sample_and_symmetry():
    """1e6 rays (the size of one reference shine() batch): a random sample is
    checked against the numpy oracle; mirror symmetry psi -> -psi of a planar
    undulator (Is even, Ip odd) holds for every ray."""
    rng = np.random.RandomState(11)
    n = 1_000_000
    Kx, Ky, Np, L0, gamma0 = 0., 1.3, 40, 30., 5870.85
    tab = un.node_tables(24, 2, 0.)
    w = rng.uniform(900., 1100., n)
    th = rng.uniform(-4e-5, 4e-5, n)
    ps = rng.uniform(-4e-5, 4e-5, n)
    th[n // 2:], ps[n // 2:], w[n // 2:] = th[:n // 2], -ps[:n // 2], w[:n // 2]
    gamma, wu, ww1, ab = un.prefactors(Kx, Ky, Np, L0, gamma0, w, th, ps, True)
    g = dict(tab, gamma=gamma, wu=wu, w=w, ww1=ww1, ddphi=th, ddpsi=ps, mode=0, Kx=Kx,
             Ky=Ky, Np=Np, taperVal=np.nan, r0z=0.)
    Is, Ip = run_dev(g)
    idx = rng.choice(n, 2000, replace=False)
    rIs, rIp = un.sp_sum(0, Kx, Ky, Np, tab, ww1[idx], w[idx], wu[idx], gamma[idx],
                         th[idx], ps[idx])
    assert rel(Is[idx], rIs) < TOL and rel(Ip[idx], rIp) < TOL
    h = n // 2
    scale = np.abs(Is).max()
    assert np.abs(Is[:h] - Is[h:]).max() < 1e-9 * scale
    assert np.abs(Ip[:h] + Ip[h:]).max() < 1e-9 * scale


def test_empty_inputs():
    from xrt_amd import hipcalls
    e = torch.empty(0, dtype=torch.float64, device='cuda')
    tab = un.node_tables(8, 2, 0.)
    tabs = [dev(tab[k]) for k in ('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph')]
    Is, Ip = hipcalls.undulator(0, 0., 1., tabs, e, e, e, e, e, e)
    assert Is.numel() == 0 and Ip.numel() == 0
    one = torch.ones(5, dtype=torch.float64, device='cuda')
    Is, Ip = hipcalls.undulator(0, 0., 1., [e] * 6, one * 5000, one, one, one, one * 0,
                                one * 0)
    torch.cuda.synchronize()
    assert torch.all(Is == 0) and torch.all(Ip == 0)


def test_bad_arguments_fail_loudly():
    from xrt_amd import hipcalls, _lib
    tab = un.node_tables(8, 2, 0.)
    tabs = [dev(tab[k]) for k in ('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph')]
    one = torch.ones(5, dtype=torch.float64, device='cuda')
    with pytest.raises(_lib.XrtHipError):
        hipcalls.undulator(7, 0., 1., tabs, one, one, one, one, one, one)
    with pytest.raises(_lib.XrtHipError):
        hipcalls.undulator(1, 0., 1., tabs, one, one, one, one, one, one, nper=0)


# ---- tabulated magnetic field (SourceFromField) -------------------------------
CUSTOM = ['far', 'filament', 'nf']
CTABS = ('tg', 'ag', 'Bx', 'By', 'Bz', 'betax', 'betay', 'trajx', 'trajy', 'trajz')


def run_custom(g, sl=slice(None)):
    from xrt_amd import hipcalls
    R0 = float(g['R0'])
    Is, Ip = hipcalls.custom_field(
        [dev(g[k]) for k in CTABS], dev(g['emcg'][sl]), dev(g['gamma'][sl]),
        dev(g['w'][sl]), dev(g['ddphi'][sl]), dev(g['ddpsi'][sl]), float(g['betam']),
        filament=bool(g['filament']), R0=None if np.isnan(R0) else R0)
    torch.cuda.synchronize()
    return Is.cpu().numpy(), Ip.cpu().numpy()


@pytest.mark.parametrize('tag', CUSTOM)
def test_custom_field_sums_match_reference_golden(golden_dir, tag):
    """800 nodes along a 10-period device; the phase wc*(tg - n.traj) is a
    difference of ~1e3 mm terms times 1e7/mm: operation order reproduced, sin/cos
    differ at 1e-16 -> norm-wise 1e-10 asserted (north_star 1e-5)."""
    g = np.load(os.path.join(golden_dir, 'g12_custom_field_%s.npz' % tag))
    Is, Ip = run_custom(g)
    assert rel(Is, g['Is']) < 1e-10, rel(Is, g['Is'])
    assert rel(Ip, g['Ip']) < 1e-10, rel(Ip, g['Ip'])
    Is2, Ip2 = run_custom(g, slice(200, 333))
    assert np.array_equal(Is[200:333], Is2) and np.array_equal(Ip[200:333], Ip2)


def test_custom_field_dropin_marshalling(golden_dir):
    """XRT_HIP.run_parallel('custom_field', ...) with the argument layout of
    SourceFromField._build_I_map_custom_field_CL gives the device-path numbers
    (emcg rebuilt from gamma like the numpy path)."""
    from xrt_amd.backends.raycing.myhip import XRT_HIP
    g = np.load(os.path.join(golden_dir, 'g12_custom_field_far.npz'))
    n = len(g['w'])
    rw = [np.zeros(n, dtype=np.complex128), np.zeros(n, dtype=np.complex128)]
    out = XRT_HIP().run_parallel(
        'custom_field', [np.int32(len(g['tg'])), np.float64(g['betam']), None],
        [g['gamma'], g['w'], g['ddphi'], g['ddpsi']], [g[k] for k in CTABS], rw, None, n)
    assert out[0] is rw[0]
    assert rel(rw[0], g['Is']) < 1e-10 and rel(rw[1], g['Ip']) < 1e-10


def test_custom_field_empty_and_bad_arguments():
    from xrt_amd import hipcalls, _lib
    e = torch.empty(0, dtype=torch.float64, device='cuda')
    one = torch.ones(4, dtype=torch.float64, device='cuda')
    tabs = [one.clone() for _ in CTABS]
    Is, Ip = hipcalls.custom_field(tabs, e, e, e, e, e, 1.0)
    assert Is.numel() == 0
    Is, Ip = hipcalls.custom_field([e] * 10, one, one * 5000, one, one * 0, one * 0, 1.0)
    torch.cuda.synchronize()
    assert torch.all(Is == 0) and torch.all(Ip == 0)
    with pytest.raises(_lib.XrtHipError):
        hipcalls.custom_field(tabs, one, one, one, one, one, 1.0, R0=-5.)


def test_node_records_are_packed_once_per_table_set(golden_dir):
    """The workspace keeps the packed node records between calls with the same table tensors
    (xrt_hip_undulator.workspace_packed); a table changed in place, other tensors, another K
    or a custom-field call in between make the next call pack again."""
    import torch
    from xrt_amd import hipcalls
    g = np.load(os.path.join(golden_dir, 'g9_undulator_far_planar.npz'))
    dev = torch.device('cuda', torch.cuda.current_device())
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
    tabs = [up(g[k]) for k in ('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph')]
    rays = [up(g[k]) for k in ('gamma', 'wu', 'w', 'ww1', 'ddphi', 'ddpsi')]
    Kx, Ky = float(g['Kx']), float(g['Ky'])

    def run(tables, ky=Ky):
        Is, Ip = hipcalls.undulator(0, Kx, ky, tables, *rays)
        return Is.cpu().numpy(), Ip.cpu().numpy()

    def packed():
        ws = hipcalls.workspace(dev, 256, 'undulator')
        return getattr(ws, '_xrt_packed', None)
    first = run(tabs)
    key = packed()
    again = run(tabs)                                   # served from the packed records
    assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1])
    assert packed()[1] == key[1]
    # a table modified in place: version counter moves, the records are made again
    tabs[1].mul_(2.)
    doubled = run(tabs)
    assert np.allclose(doubled[0], 2. * first[0], rtol=1e-13, atol=0)
    tabs[1].mul_(0.5)
    assert np.array_equal(run(tabs)[0], first[0])
    # other tensor objects with other contents
    other = [t.clone() for t in tabs]
    other[1] *= 3.
    assert np.allclose(run(other)[0], 3. * first[0], rtol=1e-13, atol=0)
    assert np.array_equal(run(tabs)[0], first[0])
    # another deflection parameter
    assert not np.array_equal(run(tabs, ky=1.1 * Ky)[0], first[0])
    assert np.array_equal(run(tabs)[0], first[0])
