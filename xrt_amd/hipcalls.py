"""Typed Python wrappers over the device-pointer entry points of libxrt_hip.so.

torch is used only as the owner of device memory and streams: tensors are
passed to the C ABI as raw pointers (``data_ptr()``) together with torch's
current HIP stream.
"""
import ctypes
import threading

import torch

from . import _lib

_tls = threading.local()


def raw_stream(index=None):
    """The HIP stream the calling thread launches on (torch's current stream of the device) as a
    number: ``torch.cuda.current_stream().cuda_stream`` without the Stream object it makes
    (3 us a time, five times per element chain)."""
    if index is None:
        index = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(index)


def stream_ptr(index=None):
    return ctypes.c_void_p(raw_stream(index))


_stream_ptr = stream_ptr


def _f64(t, n=None, name='array'):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError('%s must be a CUDA/HIP tensor' % name)
    if t.dtype != torch.float64:
        raise TypeError('%s must be float64, got %s' % (name, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)
    if n is not None and t.numel() != n:
        raise ValueError('%s has %d elements, expected %d' % (name, t.numel(), n))
    return ctypes.c_void_p(t.data_ptr())


def _c128(t, n, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError('%s must be a CUDA/HIP tensor' % name)
    if t.dtype != torch.complex128:
        raise TypeError('%s must be complex128, got %s' % (name, t.dtype))
    if not t.is_contiguous() or t.numel() != n:
        raise ValueError('%s must be contiguous with %d elements' % (name, n))
    return ctypes.c_void_p(t.data_ptr())


def workspace(device, nbytes, tag='default'):
    """Grow-only scratch buffer owned by torch's allocator, one per (thread, device,
    HIP stream, tag): kernels of one stream reuse it in stream order; two Python
    threads, or two streams of one thread, never share one."""
    cache = _tls.__dict__.setdefault('workspaces', {})
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, raw_stream(index), tag)
    ws = cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        cache[key] = ws
    return ws


def packed_workspace(device, nbytes, tables, scalars):
    """The undulator workspace and whether it still holds the node records of *tables* (the
    same tensor objects, unmodified since: identity and version counters) with *scalars*."""
    import weakref
    ws = workspace(device, nbytes, 'undulator')
    was = getattr(ws, '_xrt_packed', None)
    versions = tuple(t._version for t in tables)
    same = was is not None and was[1] == versions and was[2] == scalars and \
        len(was[0]) == len(tables) and all(r() is t for r, t in zip(was[0], tables))
    # the record of what the workspace holds is set by mark_packed() AFTER the call that packs
    # succeeded: a call that failed before its pack kernel must not leave the claim behind
    ws._xrt_packed = None
    ws._xrt_packing = (tuple(weakref.ref(t) for t in tables), versions, scalars)
    return ws, same


def mark_packed(ws):
    """The C call that packed (or reused) the node records of *ws* returned success."""
    ws._xrt_packed = ws._xrt_packing


def kirchhoff_plan(npix, ns, nsplit=0, ppt=0):
    lib = _lib.load()
    wsb = ctypes.c_size_t(0)
    ns_out = ctypes.c_int(0)
    ppt_out = ctypes.c_int(0)
    _lib.check(lib.xrt_hip_kirchhoff_plan(
        npix, ns, nsplit, ppt, ctypes.byref(wsb), ctypes.byref(ns_out),
        ctypes.byref(ppt_out)), 'xrt_hip_kirchhoff_plan')
    return wsb.value, ns_out.value, ppt_out.value


KIRCHHOFF_NO_FAST = 0x100     # XRT_HIP_KIRCHHOFF_NO_FAST
KIRCHHOFF_NO_SHARE = 0x200    # XRT_HIP_KIRCHHOFF_NO_SHARE
KIRCHHOFF_RELAXED = 0x400     # XRT_HIP_KIRCHHOFF_RELAXED
# loop variants of csrc/kirchhoff.hip (KV_*), as reported by kirchhoff_report()
KIRCHHOFF_VARIANTS = (
    'gen_s_y', 'gen_s_n', 'gen_sp_y', 'gen_sp_n', 'gen_s_notab', 'gen_sp_notab',
    'fast_s', 'fast_s_unik', 'fast_sp', 'fast_s_share', 'fast_s_share_unik',
    'fast_sp_share', 'fast_s_notab', 'fast_s_notab_unik', 'fast_sp_notab',
    'gen_s_n_relaxed', 'gen_sp_n_relaxed')


def kirchhoff_report(device=None):
    """What the last kirchhoff() call on this device found and ran: dict with the
    classification flags, the set of loop-variant names, the mesh row length."""
    lib = _lib.load()
    dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
    # the record lives at the head of the workspace the call used: none yet on this thread /
    # stream -> nothing to report (a fresh buffer would hold garbage)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (index, raw_stream(index), 'kirchhoff')
    ws = _tls.__dict__.get('workspaces', {}).get(key)
    if ws is None:
        return None
    f, v, row = ctypes.c_uint(0), ctypes.c_uint(0), ctypes.c_int64(0)
    with torch.cuda.device(dev):
        _lib.check(lib.xrt_hip_kirchhoff_report(
            ctypes.c_void_p(ws.data_ptr()), _stream_ptr(), ctypes.byref(f),
            ctypes.byref(v), ctypes.byref(row)), 'xrt_hip_kirchhoff_report')
    names = {n for i, n in enumerate(KIRCHHOFF_VARIANTS) if (v.value >> i) & 1}
    return dict(flags=f.value, variants=names, row=row.value)


def kirchhoff(px, py, pz, sx, sy, sz, nx, ny, nz, nl, k, Es, Ep, convention=0,
              nsplit=0, ppt=0, out=None, timing=False, relaxed=False):
    """Fresnel-Kirchhoff integral on device-resident arrays.

    Returns (S, P, A, B, C) complex128 tensors [npix] = the (Es, Ep, aE, bE, cE)
    of xrt's _diffraction_integral_conv (convention 0) or of its OpenCL kernel
    (convention 1); with ``timing=True`` also the main kernel's milliseconds
    (the call then synchronises). *relaxed* (opt-in, XRT_HIP_KIRCHHOFF_RELAXED): samples with
    general normals take the loop with 5 of 60 issue slots less, whose sums agree with numpy's
    to ~1e-8 norm-wise instead of ~1e-12."""
    lib = _lib.load()
    if relaxed:
        ppt = int(ppt) | KIRCHHOFF_RELAXED
    npix = px.numel()
    ns = sx.numel()
    dev = px.device
    if out is None:
        out = tuple(torch.empty(npix, dtype=torch.complex128, device=dev)
                    for _ in range(5))
    wsb, _, _ = kirchhoff_plan(npix, ns, nsplit, ppt)
    ws = workspace(dev, wsb, 'kirchhoff')
    ms = ctypes.c_float(0.)
    with torch.cuda.device(dev):
        rc = lib.xrt_hip_kirchhoff_f64_dev(
            npix, _f64(px, npix, 'px'), _f64(py, npix, 'py'), _f64(pz, npix, 'pz'),
            ns, _f64(sx, ns, 'sx'), _f64(sy, ns, 'sy'), _f64(sz, ns, 'sz'),
            _f64(nx, ns, 'nx'), _f64(ny, ns, 'ny'), _f64(nz, ns, 'nz'),
            _f64(nl, ns, 'nl'), _f64(k, ns, 'k'), _c128(Es, ns, 'Es'),
            _c128(Ep, ns, 'Ep'), int(convention),
            *[_c128(o, npix, 'out') for o in out],
            ctypes.c_void_p(ws.data_ptr()), ws.numel(), int(nsplit), int(ppt),
            _stream_ptr(), ctypes.byref(ms) if timing else None)
    _lib.check(rc, 'xrt_hip_kirchhoff_f64_dev')
    if timing:
        return out + (ms.value,)
    return out


def debug_sqrt(x):
    lib = _lib.load()
    r = torch.empty_like(x)
    ri = torch.empty_like(x)
    _lib.check(lib.xrt_hip_debug_sqrt_f64_dev(
        x.numel(), _f64(x), _f64(r), _f64(ri), _stream_ptr()), 'debug_sqrt')
    return r, ri


def debug_sqrt_seeded(x, seed):
    lib = _lib.load()
    r = torch.empty_like(x)
    h = torch.empty_like(x)
    _lib.check(lib.xrt_hip_debug_sqrt_seeded_f64_dev(
        x.numel(), _f64(x), _f64(seed, x.numel()), _f64(r), _f64(h), _stream_ptr()),
        'debug_sqrt_seeded')
    return r, h


def debug_sincos(phi, table=False):
    lib = _lib.load()
    s = torch.empty_like(phi)
    c = torch.empty_like(phi)
    fn = {0: lib.xrt_hip_debug_sincos_f64_dev, 1: lib.xrt_hip_debug_sincos_tab_f64_dev,
          2: lib.xrt_hip_debug_sincos_tab4k_f64_dev}[int(table)]
    _lib.check(fn(phi.numel(), _f64(phi), _f64(s), _f64(c), _stream_ptr()), 'debug_sincos')
    return s, c


def debug_divconst(a, b):
    lib = _lib.load()
    q = torch.empty_like(a)
    _lib.check(lib.xrt_hip_debug_divconst_f64_dev(
        a.numel(), _f64(a), float(b), _f64(q), _stream_ptr()), 'debug_divconst')
    return q


UND_FAR, UND_TAPER, UND_NF = 0, 1, 2


def undulator(mode, Kx, Ky, tables, gamma, wu, w, ww1, ddphi, ddpsi, nper=1,
              alpha_s=0., r0z=0., Is=None, Ip=None, timing=False):
    """Undulator field sums on device tensors (xrt_hip_undulator_f64_dev).

    tables = (tg, ag, sintg, costg, sintgph, costgph) float64 CUDA tensors of
    equal length; the six ray arrays float64 CUDA tensors of equal length.
    Returns (Is, Ip) complex128 tensors (and the kernel ms when timing)."""
    from ._structs import Undulator
    lib = _lib.load()
    n = gamma.numel()
    jend = tables[0].numel()
    dev = gamma.device
    if Is is None:
        Is = torch.empty(n, dtype=torch.complex128, device=dev)
    if Ip is None:
        Ip = torch.empty(n, dtype=torch.complex128, device=dev)
    u = Undulator()
    u.mode, u.nper = int(mode), int(nper)
    u.Kx, u.Ky = float(Kx), float(Ky)
    u.alpha_s, u.r0z = float(alpha_s), float(r0z)
    u.jend = jend
    for name, t in zip(('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph'),
                       tables):
        setattr(u, name, _f64(t, jend, name).value)
    wsb = lib.xrt_hip_undulator_workspace_bytes(jend)
    ws, packed = packed_workspace(dev, wsb, tables, (u.Kx, u.Ky, jend))
    u.workspace_packed = 1 if packed else 0
    ms = ctypes.c_float(0.)
    with torch.cuda.device(dev):
        rc = lib.xrt_hip_undulator_f64_dev(
            ctypes.byref(u), n, _f64(gamma, n, 'gamma'), _f64(wu, n, 'wu'),
            _f64(w, n, 'w'), _f64(ww1, n, 'ww1'), _f64(ddphi, n, 'ddphi'),
            _f64(ddpsi, n, 'ddpsi'), _c128(Is, n, 'Is'), _c128(Ip, n, 'Ip'),
            ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr(),
            ctypes.byref(ms) if timing else None)
    _lib.check(rc, 'xrt_hip_undulator_f64_dev')
    mark_packed(ws)
    return (Is, Ip, ms.value) if timing else (Is, Ip)


def _undulator_struct(mode, Kx, Ky, tables, nper, alpha_s, r0z):
    from ._structs import Undulator
    u = Undulator()
    jend = tables[0].numel()
    u.mode, u.nper = int(mode), int(nper)
    u.Kx, u.Ky = float(Kx), float(Ky)
    u.alpha_s, u.r0z = float(alpha_s), float(r0z)
    u.jend = jend
    for name, t in zip(('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph'),
                       tables):
        setattr(u, name, _f64(t, jend, name).value)
    return u


def undulator_imap(mode, Kx, Ky, tables, w, theta, psi, L0, Np, gamma0, eI, dstep,
                   dist_bw, gamma=None, harmonic=None, alpha_s=0., r0z=0.):
    """Whole ``Undulator.build_I_map`` in one launch
    (xrt_hip_undulator_imap_f64_dev): returns device tensors (I, Es, Ep)."""
    from ._structs import UndulatorMap
    lib = _lib.load()
    n = w.numel()
    dev = w.device
    u = _undulator_struct(mode, Kx, Ky, tables, int(Np) if mode else 1, alpha_s, r0z)
    m = UndulatorMap()
    m.L0, m.Np, m.gamma0, m.eI, m.dstep = (float(L0), float(Np), float(gamma0),
                                            float(eI), float(dstep))
    m.has_harmonic = 0 if harmonic is None else 1
    m.harmonic = 0. if harmonic is None else float(harmonic)
    m.dist_bw = 1 if dist_bw else 0
    I = torch.empty(n, dtype=torch.float64, device=dev)
    Es = torch.empty(n, dtype=torch.complex128, device=dev)
    Ep = torch.empty(n, dtype=torch.complex128, device=dev)
    wsb = lib.xrt_hip_undulator_workspace_bytes(u.jend)
    ws, packed = packed_workspace(dev, wsb, tables, (u.Kx, u.Ky, int(u.jend)))
    u.workspace_packed = 1 if packed else 0
    with torch.cuda.device(dev):
        rc = lib.xrt_hip_undulator_imap_f64_dev(
            ctypes.byref(u), ctypes.byref(m), n, _f64(w, n, 'w'),
            _f64(theta, n, 'theta'), _f64(psi, n, 'psi'),
            None if gamma is None else _f64(gamma, n, 'gamma'), _f64(I, n, 'I'),
            _c128(Es, n, 'Es'), _c128(Ep, n, 'Ep'),
            ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr())
    _lib.check(rc, 'xrt_hip_undulator_imap_f64_dev')
    mark_packed(ws)
    return I, Es, Ep


CUSTOM_TABLES = ('tg', 'ag', 'Bx', 'By', 'Bz', 'betax', 'betay', 'trajx', 'trajy',
                 'trajz')


def trajectory(wt, Bx, By, Bz, gamma=None, emcg=1.):
    """Electron trajectory through a field tabulated on the half-step grid of *wt*
    (xrt_hip_trajectory_f64_dev) -> (betax, betay, trajx, trajy, trajz, betam) device
    tensors; *gamma* given = a filament beam's electron (then *emcg* is its
    e/(m c gamma) factor)."""
    lib = _lib.load()
    n = wt.numel()
    dev = wt.device
    outs = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(5)]
    betam = torch.empty(1, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.xrt_hip_trajectory_f64_dev(
            0 if gamma is None else 1, n, _f64(wt, n, 'wt'),
            *[_f64(t, 2 * n - 1, name) for t, name in ((Bx, 'Bx'), (By, 'By'), (Bz, 'Bz'))],
            0. if gamma is None else float(gamma), float(emcg),
            *[ctypes.c_void_p(t.data_ptr()) for t in outs],
            ctypes.c_void_p(betam.data_ptr()), _stream_ptr())
    _lib.check(rc, 'xrt_hip_trajectory_f64_dev')
    return tuple(outs) + (betam,)


def bend_imap(E, theta, psi, gamma0, B, eI, poles=1., K=0., wiggler=False, per_bandwidth=True,
              gamma=None):
    """Flux and amplitudes of a bending magnet / wiggler per ray
    (xrt_hip_bend_imap_f64_dev) -> (I, Es, Ep) device tensors."""
    from ._structs import Bend
    lib = _lib.load()
    n = E.numel()
    dev = E.device
    m = Bend(float(gamma0), float(B), float(K), float(poles), float(eI),
             1 if wiggler else 0, 1 if per_bandwidth else 0)
    I = torch.empty(n, dtype=torch.float64, device=dev)
    Es = torch.empty(n, dtype=torch.complex128, device=dev)
    Ep = torch.empty(n, dtype=torch.complex128, device=dev)
    with torch.cuda.device(dev):
        rc = lib.xrt_hip_bend_imap_f64_dev(
            ctypes.byref(m), n, _f64(E, n, 'E'), _f64(theta, n, 'theta'), _f64(psi, n, 'psi'),
            None if gamma is None else _f64(gamma, n, 'gamma'),
            ctypes.c_void_p(I.data_ptr()), _c128(Es, n, 'Es'), _c128(Ep, n, 'Ep'),
            _stream_ptr())
    _lib.check(rc, 'xrt_hip_bend_imap_f64_dev')
    return I, Es, Ep


def debug_bessel_k(x):
    lib = _lib.load()
    n = x.numel()
    k13, k23 = torch.empty_like(x), torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.xrt_hip_debug_bessel_k_f64_dev(
            n, _f64(x, n, 'x'), ctypes.c_void_p(k13.data_ptr()),
            ctypes.c_void_p(k23.data_ptr()), _stream_ptr()), 'xrt_hip_debug_bessel_k_f64_dev')
    return k13, k23


def custom_field(tables, emcg, gamma, w, ddphi, ddpsi, betam, filament=False, R0=None,
                 wc=0., timing=False, carrier_form=0):
    """Field sums of a tabulated-field source on device tensors
    (xrt_hip_custom_field_f64_dev). tables: dict or sequence of the ten node
    tables in CUSTOM_TABLES order. Returns (Is, Ip[, kernel ms])."""
    from ._structs import CustomField
    lib = _lib.load()
    if isinstance(tables, dict):
        tables = [tables[k] for k in CUSTOM_TABLES]
    n = w.numel()
    jend = tables[0].numel()
    dev = w.device
    f = CustomField()
    f.filament = 1 if filament else 0
    f.near_field = 0 if R0 is None else 1
    f.betam = float(betam)
    f.R0 = 0. if R0 is None else float(R0)
    f.wc = float(wc)
    f.carrier_form = int(carrier_form)
    f.jend = jend
    for name, t in zip(CUSTOM_TABLES, tables):
        setattr(f, name, _f64(t, jend, name).value)
    Is = torch.empty(n, dtype=torch.complex128, device=dev)
    Ip = torch.empty(n, dtype=torch.complex128, device=dev)
    wsb = lib.xrt_hip_undulator_workspace_bytes(jend)
    ws = workspace(dev, wsb, 'undulator')
    ws._xrt_packed = None          # (its records replace the undulator's)
    ms = ctypes.c_float(0.)
    with torch.cuda.device(dev):
        rc = lib.xrt_hip_custom_field_f64_dev(
            ctypes.byref(f), n, _f64(emcg, n, 'emcg'), _f64(gamma, n, 'gamma'),
            _f64(w, n, 'w'), _f64(ddphi, n, 'ddphi'), _f64(ddpsi, n, 'ddpsi'),
            _c128(Is, n, 'Is'), _c128(Ip, n, 'Ip'), ctypes.c_void_p(ws.data_ptr()),
            ws.numel(), _stream_ptr(), ctypes.byref(ms) if timing else None)
    _lib.check(rc, 'xrt_hip_custom_field_f64_dev')
    return (Is, Ip, ms.value) if timing else (Is, Ip)
