"""Wave propagation — host-side mirror of xrt/backends/raycing/waves.py:505-896.

``diffract(oeLocal, wave)`` keeps xrt's semantics: the O(Ns*Np) Fresnel-Kirchhoff
integral runs in the HIP kernel (csrc/kirchhoff.hip, through
``hipcalls.kirchhoff``); the O(Ns+Np) bookkeeping around it (sample selection,
footprint area, normalisation, phase strip, frame changes) is host glue, like in
the reference. Sign convention: the numpy path of the reference
(``_diffraction_integral_conv``, +i k/4pi).
"""
import ctypes
import threading
import time

import numpy as np
import torch

from ... import hipcalls as _hipcalls

from .. import raycing
from ... import _lib, _structs, hipcalls
from . import sources as rs
from .physconsts import CH, CHBAR

_DEBUG = 0
# Milliseconds of the last Kirchhoff kernel -- only measured when `timeKernels` is set: the
# measurement waits for the stream (bench.py sets it around the calls it times).
lastKernelMs = None
timeKernels = False
# GPUs a diffract() call spreads its receiving points over, as the reference spreads them
# over its OpenCL devices (myopencl.py:455-533): a list of ordinals, 'all', or None = what
# the call's targetOpenCL / the XRT_HIP_DEVICES environment variable say (default: the
# current device only).
devices = None
# 'exact' (default): the integrals are the doubles of the reference's numpy kernel
# (_diffraction_integral_conv, waves.py:834-851) to ~1e-12. 'relaxed' (opt-in): transfers from
# samples with general normals (mirror -> mirror) run a loop with 8 % fewer instructions whose
# sums agree to ~1e-8 norm-wise (hipcalls.kirchhoff(relaxed=True), include/xrt_hip.h).
precision = 'exact'


_ACCUMULATORS = ('EsAcc', 'EpAcc', 'aEacc', 'bEacc', 'cEacc')


def _into_frame_of(element, xglo, yglo, zglo):
    """Global points -> the local frame of *element*: shift to its centre, undo
    the beamline azimuth and, for an optical element, its own rotations
    (waves.py:529-560)."""
    if isinstance(xglo, torch.Tensor):      # points on the GPU: the same arithmetic there
        px = xglo - float(element.center[0])
        py = yglo - float(element.center[1])
        pz = zglo - float(element.center[2])
    else:
        px = np.array(xglo, dtype=float) - element.center[0]
        py = np.array(yglo, dtype=float) - element.center[1]
        pz = np.array(zglo, dtype=float) - element.center[2]
    bl = element.bl
    px, py = raycing.rotate_z(px, py, bl.cosAzimuth, bl.sinAzimuth)
    if hasattr(element, 'rotationSequence'):
        if hasattr(element, 'local_n2') and hasattr(element, 'cryst2pitch'):
            raise NotImplementedError('wave propagation from a DCM/plate 2nd '
                                      'surface')
        raycing.rotate_xyz(px, py, pz, rotationSequence=element.rotationSequence,
                           pitch=-element.pitch,
                           roll=-(element.roll + element.positionRoll),
                           yaw=-element.yaw)
        if element.extraPitch or element.extraRoll or element.extraYaw:
            raycing.rotate_xyz(px, py, pz,
                               rotationSequence=element.extraRotationSequence,
                               pitch=-element.extraPitch, roll=-element.extraRoll,
                               yaw=-element.extraYaw)
    return px, py, pz


def prepare_wave(fromOE, wave, xglo, yglo, zglo):
    """Binds the receiving samples *wave* (global positions *xglo, yglo, zglo*)
    to the diffracting element *fromOE*: their coordinates in its local frame
    (``xDiffr`` ...), unit directions from its centre, and empty field
    accumulators (reference: waves.py:505-584)."""
    if isinstance(xglo, torch.Tensor):
        # a wave that lives on the GPU (OE.prepare_wave): fields and accumulators are made
        # there, the frame change runs there
        dev = xglo.device
        nsamples = xglo.numel()
        zeros_c = lambda: torch.zeros(nsamples, dtype=torch.complex128, device=dev)  # noqa: E731
        zeros_f = lambda: torch.zeros(nsamples, dtype=torch.float64, device=dev)     # noqa: E731
        wave.Es, wave.Ep, wave.Jsp = zeros_c(), zeros_c(), zeros_c()
        for name in _ACCUMULATORS:
            setattr(wave, name, zeros_c())
        wave.Jss, wave.Jpp, wave.path = zeros_f(), zeros_f(), zeros_f()
        px, py, pz = _into_frame_of(fromOE, xglo, yglo, zglo)
        dist = torch.sqrt(px * px + py * py + pz * pz)
        wave.xDiffr, wave.yDiffr, wave.zDiffr, wave.rDiffr = px, py, pz, dist
        wave.a, wave.b, wave.c = px / dist, py / dist, pz / dist
    else:
        nsamples = len(wave.x)
        if hasattr(wave, 'Es'):
            wave.Es[:] = 0
            wave.Ep[:] = 0
        else:
            wave.Es = np.zeros(nsamples, dtype=complex)
            wave.Ep = np.zeros(nsamples, dtype=complex)
        for name in _ACCUMULATORS:
            setattr(wave, name, np.zeros(nsamples, dtype=complex))
        for component in (wave.Jss, wave.Jpp, wave.Jsp):
            component[:] = 0
        px, py, pz = _into_frame_of(fromOE, xglo, yglo, zglo)
        dist = (px**2 + py**2 + pz**2)**0.5
        wave.xDiffr, wave.yDiffr, wave.zDiffr, wave.rDiffr = px, py, pz, dist
        wave.a[:] = px / dist
        wave.b[:] = py / dist
        wave.c[:] = pz / dist
        wave.path[:] = 0.
    wave.fromOE = fromOE
    wave.beamReflRays = np.int64(0)       # running sums over repeated diffract()
    wave.beamReflSumJ = 0.
    wave.beamReflSumJnl = 0.
    wave.diffract_repeats = np.int64(0)
    return wave


def receiving_wave(element, prevOE, local, glob, dS, area, parent):
    """Wave samples at the points *local* (x, y, z in the frame of *element*, which
    owns them) = *glob* in the global frame, bound to the diffracting *prevOE*."""
    if isinstance(local[0], torch.Tensor):          # points that live on the GPU: so does the wave
        wave = rs.Beam.on_device(local[0].numel(), local[0].device, withAmplitudes=True, state=1)
        wave.x, wave.y, wave.z = local
    else:
        wave = rs.Beam(nrays=len(local[0]), forceState=1, withAmplitudes=True)
        for name, values in zip('xyz', local):
            getattr(wave, name)[:] = values
    wave.dS, wave.area, wave.toOE, wave.parentId = dS, area, element, parent
    return prepare_wave(prevOE, wave, *glob)


def qualify_sampling(wave, E, goodlen):
    """Effective Fresnel number and samples per Fresnel zone
    (waves.py:587-603)."""
    a = wave.xDiffr / wave.rDiffr
    c = wave.zDiffr / wave.rDiffr
    NAx = (a.max() - a.min()) * 0.5
    NAz = (c.max() - c.min()) * 0.5
    invLambda = E / CH * 1e7
    fn = (NAx**2 + NAz**2) * wave.rDiffr.mean() * invLambda
    samplesPerZone = abs(goodlen / fn)
    return fn, samplesPerZone


def _outside_polygon(px, py, corners):
    """Mask of the points NOT strictly inside the convex polygon *corners* (counter-
    clockwise): only those can be vertices of the hull."""
    inside = np.ones(len(px), dtype=bool)
    for (x1, y1), (x2, y2) in zip(corners, corners[1:] + corners[:1]):
        inside &= (x2-x1)*(py-y1) - (y2-y1)*(px-x1) > 0
    return ~inside


def _extremes(px, py, keys):
    """The points that are extreme along the given directions, in the order of the
    directions, consecutive duplicates dropped."""
    ext = []
    for key in keys:
        i = int(np.argmax(key))
        if not ext or (px[i], py[i]) != ext[-1]:
            ext.append((px[i], py[i]))
    if len(ext) > 1 and ext[0] == ext[-1]:
        ext.pop()
    return ext


def convex_hull_area(px, py):
    """Area of the convex hull of 2-D points (Andrew's monotone chain; the reference
    takes it from scipy.spatial.ConvexHull, waves.py:661-668). Points that cannot be hull
    vertices are filtered out first -- a box inside a subsample's octagon, then the
    Akl-Toussaint octagon of the 8 axis / diagonal extremes, then the polygon of the extremes
    in 64 directions --, so that the Python loop of the chain only sees a handful of
    points."""
    px = np.asarray(px, dtype=float)
    py = np.asarray(py, dtype=float)
    if len(px) > 4096:
        # round 0: an axis-parallel box inside the octagon of a 1/16 subsample's extremes --
        # four comparisons per point throw out the bulk before any polygon test
        qx, qy = px[::16], py[::16]
        ext = _extremes(qx, qy, (qx, qx + qy, qy, qy - qx, -qx, -qx - qy, -qy, qx - qy))
        if len(ext) >= 3:
            cx, cy = np.mean([e[0] for e in ext]), np.mean([e[1] for e in ext])
            wx = max(abs(e[0] - cx) for e in ext)
            wy = max(abs(e[1] - cy) for e in ext)
            t = 1.
            for _ in range(12):      # shrink until the four corners lie inside the octagon
                bx = np.array([cx - t*wx, cx + t*wx, cx + t*wx, cx - t*wx])
                by = np.array([cy - t*wy, cy - t*wy, cy + t*wy, cy + t*wy])
                if not _outside_polygon(bx, by, ext).any():
                    break
                t *= 0.8
            else:
                t = 0.
            if t > 0.:
                keep = (np.abs(px - cx) >= t*wx) | (np.abs(py - cy) >= t*wy)
                px, py = px[keep], py[keep]
    if len(px) > 64:
        ext = _extremes(px, py, (px, px + py, py, py - px, -px, -px - py, -py, px - py))
        if len(ext) >= 3:
            keep = _outside_polygon(px, py, ext)
            px, py = px[keep], py[keep]
    if len(px) > 256:
        ang = np.arange(64) * (2 * np.pi / 64)
        ext = _extremes(px, py, [px * c + py * s_ for c, s_ in zip(np.cos(ang), np.sin(ang))])
        if len(ext) >= 3:
            keep = _outside_polygon(px, py, ext)
            px, py = px[keep], py[keep]
    pts = np.unique(np.column_stack((px, py)), axis=0)
    if len(pts) < 3:
        raise ValueError('cannot normalize this way!')

    def half(points):
        hull = []
        for p in points:
            while len(hull) >= 2:
                o, a = hull[-2], hull[-1]
                if (a[0]-o[0])*(p[1]-o[1]) - (a[1]-o[1])*(p[0]-o[0]) <= 0:
                    hull.pop()
                else:
                    break
            hull.append((p[0], p[1]))
        return hull
    lower = half(pts)
    upper = half(pts[::-1])
    outer = np.array(lower[:-1] + upper[:-1])
    x1, y1 = outer[:, 0], outer[:, 1]
    x2, y2 = np.roll(x1, -1), np.roll(y1, -1)
    return 0.5 * abs(np.sum(x1*y2 - x2*y1))


_pinned = threading.local()


def _later(tensor):
    """Starts a copy of a small device tensor to the host that does not hold the host up;
    -> a function that returns it as a numpy array (waiting for the copy if it has to)."""
    n = tensor.numel()
    pool = _pinned.__dict__.setdefault('buffers', {})
    buf = pool.get(tensor.dtype)
    if buf is None or buf.numel() < n:
        buf = pool[tensor.dtype] = torch.empty(max(n, 4096), dtype=tensor.dtype, pin_memory=True)
    view = buf[:n].view(tensor.shape)
    view.copy_(tensor, non_blocking=True)
    done = torch.cuda.Event()
    done.record()

    def fetch():
        done.synchronize()
        return view.numpy().copy()
    return fetch


def convex_hull_area_on_device(px, py, deferred=False):
    """The same area for points that live on the GPU: the polygon of the extremes in 64
    directions is found there and everything strictly inside it dropped there; the few points
    that are left (every hull vertex is among them) go through the host's monotone chain.
    *deferred*: -> a function that does the host part when it is called; the copy of the
    candidates is under way meanwhile (waves.diffract calls it after it has launched the
    integral, so that the GPU is busy while the host walks the chain)."""
    keep = _hull_candidates(px, py) if px.numel() >= 4096 else None
    both = torch.stack((px, py)) if keep is None else torch.stack((px[keep], py[keep]))
    if deferred:
        fetch = _later(both)

        def area():
            q = fetch()
            return convex_hull_area(q[0], q[1])
        return area
    q = both.cpu().numpy()
    return convex_hull_area(q[0], q[1])


def _hull_candidates(px, py):
    """Mask of the points that are not strictly inside the polygon of the extremes in 64
    directions (None if that polygon is degenerate)."""
    ang = torch.arange(64, dtype=torch.float64, device=px.device) * (2 * np.pi / 64)
    proj = torch.cos(ang)[:, None] * px[None, :] + torch.sin(ang)[:, None] * py[None, :]
    pick = proj.argmax(dim=1)
    ext = torch.stack((px[pick], py[pick]), dim=1).cpu().numpy()
    corners = []
    for q in map(tuple, ext):           # consecutive duplicates dropped, like _extremes
        if not corners or q != corners[-1]:
            corners.append(q)
    if len(corners) > 1 and corners[0] == corners[-1]:
        corners.pop()
    if len(corners) < 3:
        return None
    c = torch.tensor(corners, dtype=torch.float64, device=px.device)
    x1, y1 = c[:, 0], c[:, 1]
    x2, y2 = torch.roll(x1, -1), torch.roll(y1, -1)
    cross = (x2 - x1)[None, :] * (py[:, None] - y1[None, :]) - \
        (y2 - y1)[None, :] * (px[:, None] - x1[None, :])
    return ~(cross > 0).all(dim=1)


def _kirchhoff_on_gpu(points, samples, targetOpenCL='auto'):
    """The five integrals (Es, Ep, aE, bE, cE) of the reference's numpy kernel
    (waves.py:834-851) for receiving *points* (3 device tensors) and *samples* (the ten
    device tensors of ``_sample_arrays``), by the HIP kernel -> 5 device tensors. On
    several GPUs if asked (``devices``): pixel tiles, see multigpu.kirchhoff_devices."""
    global lastKernelMs
    from ... import multigpu
    if precision not in ('exact', 'relaxed'):
        raise ValueError("waves.precision is 'exact' or 'relaxed'")
    relaxed = precision == 'relaxed'
    devs = multigpu.parse_devices(devices if devices is not None else targetOpenCL,
                                  torch.cuda.device_count())
    if devs is not None and len(devs) > 1:
        lastKernelMs = None
        return multigpu.kirchhoff_devices(points, samples, devs, convention=0,
                                          relaxed=relaxed)
    if devs is not None and devs[0] != points[0].device.index:
        raise ValueError('the wave lives on cuda:%d, not on cuda:%d'
                         % (points[0].device.index, devs[0]))
    out = hipcalls.kirchhoff(*points, *samples, convention=0, timing=timeKernels,
                             relaxed=relaxed)
    lastKernelMs = out[5] if timeKernels else None
    return out[:5]


def _illuminated_area(oe, field):
    """Footprint of the lit samples on the diffracting element, for the flux
    normalisation: given by whoever made *field*, else the convex hull in the
    element's surface coordinates (waves.py:642-670). -> a function that returns the area:
    the device part of the hull (which points can be vertices at all) is queued now, the
    host part runs when the function is called."""
    area = getattr(field, 'area', None)
    if area is not None and area > 0:
        return lambda: area
    if hasattr(oe, 'rotationSequence'):
        second = 'y'                         # an optical element: (x, y)
    elif hasattr(oe, 'propagate') or hasattr(oe, 'prepare_wave') or \
            hasattr(oe, 'shine'):
        second = 'z'                         # aperture / screen / source: (x, z)
    else:
        raise ValueError('Unknown diffracting element!')
    fraction = getattr(field, 'areaFraction', None)
    if all(name in field._d for name in ('state', 'x', second)):
        lit = field._d['state'] == 1
        hull = convex_hull_area_on_device(field._d['x'][lit], field._d[second][lit],
                                          deferred=True)
    else:
        lit = field.peek('state') == 1
        px, py = field.peek('x')[lit], field.peek(second)[lit]
        hull = lambda: convex_hull_area(px, py)     # noqa: E731
    return hull if fraction is None else (lambda: hull() * fraction)


def _element_pass(element, lib_pass=None):
    """(pass record, is_oe) that tells the kernels about *element*'s surface and frame."""
    if hasattr(element, 'rotationSequence'):
        if hasattr(element, 'cryst2pitch'):
            raise NotImplementedError('wave propagation from / onto a DCM or plate')
        return element._make_pass(*element._own_angles()[:4]), 1
    p = _structs.Pass()
    p.invert_normal = 1
    bl = getattr(element, 'bl', None)
    p.sin_az, p.cos_az = (0., 1.) if bl is None else (bl.sinAzimuth, bl.cosAzimuth)
    return p, 0


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _sample_arrays(oe, field, dev):
    """Inputs of the integral from the samples on the diffracting element (device):
    -> (ten tensors sx, sy, sz, nx, ny, nz, nl, k, Es, Ep; flux, |flux . nl|, lit count)."""
    n = field.nrays
    f64 = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(8)]
    c128 = [torch.empty(n, dtype=torch.complex128, device=dev) for _ in range(2)]
    p, is_oe = _element_pass(oe)
    ws = hipcalls.workspace(dev, 8192, 'diffract')
    sums = (ctypes.c_double * 3)()
    _lib.check(_lib.load().xrt_hip_diffract_pre_f64_dev(
        ctypes.byref(p), is_oe, ctypes.byref(field.to_struct(dev)), *[_ptr(t) for t in f64],
        *[_ptr(t) for t in c128], _ptr(ws), ws.numel(),
        _hipcalls.stream_ptr(), sums),
        'xrt_hip_diffract_pre_f64_dev')
    return f64 + c128, sums[0], abs(sums[1]), int(sums[2])


_WAVE_FIELDS = ('a', 'b', 'c', 'E', 'Jss', 'Jpp', 'Jsp', 'Es', 'Ep')


def _as_global_beam(oe, wave, dev):
    """The field on the receiving points as a beam in the global frame (waves.py:756-770):
    a copy of *wave* positioned at the points (known in the frame of *oe*), taken out of
    that frame on the GPU."""
    glo = rs.Beam.empty_like_on_device(wave, dev)
    for name, source in (('x', 'xDiffr'), ('y', 'yDiffr'), ('z', 'zDiffr'), ('path', 'path'),
                         ('state', 'state')) + tuple((f, f) for f in _WAVE_FIELDS):
        glo._d[name].copy_(wave.dev(source, dev))
    rs.inherit_scalars(glo, wave)
    glo.parentId = oe.uuid
    stream = _hipcalls.stream_ptr()
    if hasattr(oe, 'rotationSequence'):
        oe.local_to_global(glo)
    elif hasattr(oe, 'local_to_global'):
        # a screen moves bare points, an aperture whole rays (positions and directions)
        frame = _structs.Screen()
        for k in range(3):
            frame.center[k], frame.ex[k], frame.ey[k], frame.ez[k] = (
                float(oe.center[k]), float(oe.x[k]), float(oe.y[k]), float(oe.z[k]))
        _lib.check(_lib.load().xrt_hip_basis_to_global_f64_dev(
            ctypes.byref(frame), ctypes.byref(glo.to_struct(dev)),
            0 if hasattr(oe, 'expose') else 1, stream), 'xrt_hip_basis_to_global_f64_dev')
    return glo


def _forget_host(beam, names):
    for name in names:
        beam._h.pop(name, None)


def diffract(oeLocal, wave, targetOpenCL=raycing.targetOpenCL,
             precisionOpenCL=raycing.precisionOpenCL):
    """Field diffracted from the samples *oeLocal* on ``wave.fromOE`` onto the
    points of *wave*, by the Fresnel-Kirchhoff integral on the GPU. *wave* is
    updated (it accumulates over repeated calls); returns the same field as a
    beam in the global frame. Interface of the reference's ``waves.diffract``
    (waves.py:606-831); every array stays in HBM between the steps."""
    _lib.require_gpu()
    lib = _lib.load()
    dev = torch.device('cuda', torch.cuda.current_device())
    stream = _hipcalls.stream_ptr()
    oe = wave.fromOE
    t0 = time.time()
    samples, flux, flux_nl, nlit = _sample_arrays(oe, oeLocal, dev)
    if nlit < 1e2:
        print("Not enough good rays at {0}: {1} of {2}".format(
            oe.name, nlit, oeLocal.nrays))
        return rs.Beam(nlit)
    if wave.nrays == 0 or 'xDiffr' not in wave.array_fields():
        print("No wave samples on {0}".format(oe.name))
        return rs.Beam(nlit)
    footprint = _illuminated_area(oe, oeLocal)     # (started; finished behind the launch below)
    wave.diffract_repeats += 1
    wave.beamReflRays += nlit
    wave.beamReflSumJ += flux
    wave.beamReflSumJnl += flux_nl
    points = [wave.dev(name, dev) for name in ('xDiffr', 'yDiffr', 'zDiffr')]
    # everything that has to be uploaded goes up BEFORE the integral is launched: a copy from
    # the host queued behind it would hold the host until the kernel is through, and with it
    # the next element's prepare_wave, which could run meanwhile. Accumulators that
    # prepare_wave has just zeroed are made on the device.
    for name in _ACCUMULATORS:
        if name not in wave._d and name in wave._h and not np.any(wave._h[name]):
            wave._h.pop(name)
            wave._d[name] = torch.zeros(wave.nrays, dtype=torch.complex128, device=dev)
    acc = [wave.dev(name, dev) for name in _ACCUMULATORS]
    wave_rec = wave.to_struct(dev)
    energy = oeLocal.dev('E', dev)
    fresh = _kirchhoff_on_gpu(points, samples, targetOpenCL)
    oeLocal.area = footprint()       # the host's share of the hull, while the integral runs
    # Monte-Carlo weight of the integral: receiving cell x illuminated area x incoming flux
    # over (samples x obliquity-weighted flux x repeats), waves.py:735-749
    denom = wave.beamReflRays * wave.beamReflSumJnl * wave.diffract_repeats
    scale = wave.dS * oeLocal.area * wave.beamReflSumJ / denom if denom > 0 else 0
    pointers = ctypes.c_void_p * 5
    _lib.check(lib.xrt_hip_wave_fields_f64_dev(
        wave.nrays, pointers(*[t.data_ptr() for t in fresh]),
        pointers(*[t.data_ptr() for t in acc]), _ptr(energy), float(scale),
        1 if hasattr(oe, 'rotationSequence') else 0, ctypes.byref(wave_rec),
        stream), 'xrt_hip_wave_fields_f64_dev')
    _forget_host(wave, _WAVE_FIELDS + _ACCUMULATORS)
    if hasattr(oeLocal, 'accepted'):         # source bookkeeping for absolute flux
        wave.accepted = oeLocal.accepted
        wave.acceptedE = oeLocal.acceptedE
        wave.seeded = oeLocal.seeded
        wave.seededI = oeLocal.seededI * wave.nrays / oeLocal.nrays
    glo = _as_global_beam(oe, wave, dev)
    if hasattr(wave, 'toOE'):
        p, is_oe = _element_pass(wave.toOE)
        if is_oe:        # the polarisation frame turns by roll + positionRoll
            turn = wave.toOE.roll + wave.toOE.positionRoll
            p.cos_roll, p.sin_roll = float(np.cos(turn)), float(np.sin(turn))
        _lib.check(lib.xrt_hip_wave_receive_f64_dev(
            ctypes.byref(p), is_oe, ctypes.byref(wave.to_struct(dev)),
            ctypes.byref(glo.to_struct(dev)), stream), 'xrt_hip_wave_receive_f64_dev')
        _forget_host(wave, _WAVE_FIELDS)
    if _DEBUG > 10:
        print("diffract on {0} completed in {1:.4f} s".format(
            oe.name, time.time()-t0))
    glo.createdByDiffract = True
    return glo
