"""Pixel-tile sharding of the Kirchhoff integral over the ranks of one node
(one process per GPU, torch.distributed; backend 'nccl' = RCCL over xGMI on the
GPU box, 'gloo' in the CPU tests).

This is what the reference does across OpenCL devices inside one process
(XRT_CL.run_parallel_max splits the pixel range evenly and replicates the
samples, myopencl.py:455-533): output pixels are independent, so each rank
integrates its own contiguous tile over ALL samples and the only exchange is the
assembly of the five result arrays. No reduction, hence no change of summation
order with the number of GPUs.
"""
import os

import torch


def tile_range(n, rank, world):
    """[p0, p1) of rank's contiguous tile; same formula as the in-process
    multi-device split of xrt_hip_kirchhoff_f64."""
    return n * rank // world, n * (rank + 1) // world


def all_gather_tiles(local, n_total, dist, rank, world, width=1):
    """Assembles the full array from the ranks' tiles. *local* holds this rank's
    tile (``width`` scalars per item); uneven tiles are padded to the largest one
    for the collective and trimmed afterwards. Complex tiles travel as their
    interleaved (re, im) doubles (RCCL has no complex dtypes)."""
    if dist is None or world == 1:
        return local
    if local.is_complex():
        real = torch.view_as_real(local.contiguous()).reshape(-1)
        out = all_gather_tiles(real, n_total, dist, rank, world, width=2 * width)
        return torch.view_as_complex(out.reshape(-1, 2))
    sizes = [(p1 - p0) * width
             for p0, p1 in (tile_range(n_total, r, world) for r in range(world))]
    if local.numel() != sizes[rank]:
        raise ValueError('rank %d holds %d values, its tile has %d'
                         % (rank, local.numel(), sizes[rank]))
    maxn = max(sizes)
    if all(sz == maxn for sz in sizes):
        out = torch.empty(n_total * width, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    pad = torch.zeros(maxn, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:sz] for b, sz in zip(bufs, sizes)])


def all_gather_packed(tiles, n_total, dist, rank, world):
    """The full arrays of several equally tiled complex (or real) arrays in ONE collective:
    the rank's tiles are packed into one [len(tiles), largest tile] buffer of doubles,
    gathered with a single ``all_gather_into_tensor`` and cut apart again -- one launch and
    one ring pass over xGMI instead of one per array (the five results of a Kirchhoff call:
    5 x 16 B per receiving point). -> list of full tensors, bit-identical to gathering
    them one by one."""
    if dist is None or world == 1:
        return list(tiles)
    cplx = [t.is_complex() for t in tiles]
    real = [torch.view_as_real(t.contiguous()).reshape(-1) if c else t.contiguous().reshape(-1)
            for t, c in zip(tiles, cplx)]
    width = [2 if c else 1 for c in cplx]
    edges = [tile_range(n_total, r, world) for r in range(world)]
    mine = edges[rank][1] - edges[rank][0]
    for t, w in zip(real, width):
        if t.numel() != mine * w:
            raise ValueError('rank %d holds %d values, its tile has %d' % (rank, t.numel(), mine * w))
        if t.dtype != real[0].dtype:
            raise ValueError('arrays of one packed gather share their real dtype')
    most = max(p1 - p0 for p0, p1 in edges)
    offs = [0]
    for w in width:
        offs.append(offs[-1] + most * w)
    pack = torch.zeros(offs[-1], dtype=real[0].dtype, device=real[0].device)
    for t, o in zip(real, offs):
        pack[o:o + t.numel()] = t
    if pack.is_cuda and dist.get_backend() != 'nccl':
        # (a CPU backend under device tensors -- rehearsals only: through the host)
        host = torch.empty(world * offs[-1], dtype=pack.dtype)
        dist.all_gather_into_tensor(host, pack.cpu())
        full = host.to(pack.device)
    else:
        full = torch.empty(world * offs[-1], dtype=pack.dtype, device=pack.device)
        dist.all_gather_into_tensor(full, pack)
    full = full.reshape(world, offs[-1])
    out = []
    for k, (w, c) in enumerate(zip(width, cplx)):
        parts = [full[r, offs[k]:offs[k] + (p1 - p0) * w] for r, (p0, p1) in enumerate(edges)]
        whole = torch.cat(parts)
        out.append(torch.view_as_complex(whole.reshape(-1, 2)) if c else whole)
    return out


def kirchhoff_tiled(px, py, pz, samples, dist, rank, world, convention=0):
    """px, py, pz: FULL receiving-point arrays (device tensors, replicated);
    samples: dict of device tensors (sx, sy, sz, nx, ny, nz, nl, k, Es, Ep).
    Returns the five full result arrays on every rank."""
    from . import hipcalls
    n = px.numel()
    p0, p1 = tile_range(n, rank, world)
    s = samples
    out = hipcalls.kirchhoff(
        px[p0:p1].contiguous(), py[p0:p1].contiguous(), pz[p0:p1].contiguous(),
        s['sx'], s['sy'], s['sz'], s['nx'], s['ny'], s['nz'], s['nl'], s['k'],
        s['Es'], s['Ep'], convention=convention)
    return tuple(all_gather_packed(out, n, dist, rank, world))


# ---------------------------------------------------------------------------
# The same split INSIDE one process -- what the reference does on every diffract() call
# (myopencl.py:455-533: the pixel range cut into one slice per OpenCL device, the samples
# handed to every device, blocking copy-back): one HIP stream per device, the samples
# copied device to device over xGMI, every tile integrated with the plan (sample splits,
# points per lane) of the WHOLE launch so that a point's sum does not depend on how many
# devices shared the work, the tiles copied into the result arrays on the first device.
# ---------------------------------------------------------------------------
def parse_devices(spec, visible):
    """Device ordinals from targetOpenCL-like input: None / 'auto' -> the XRT_HIP_DEVICES
    environment variable ('all', or '0,1,2'), else the current device only (None);
    'all' / 'GPU' (any case) -> every visible device; 'CPU' -> the current device (there is
    no CPU backend to fall to); an int or a LIST of ints -> those GPUs; the reference's
    ``(platform, device)`` TUPLE and lists of such tuples (myopencl.py:187-231) -> their
    device fields -- there is one platform here."""
    if spec is None or spec == 'auto':
        env = os.environ.get('XRT_HIP_DEVICES', '').strip()
        if not env:
            return None
        spec = env if env.lower() in ('all', 'gpu') else \
            [int(t) for t in env.replace(';', ',').split(',') if t.strip()]
    if isinstance(spec, str):
        word = spec.strip().lower()
        if word in ('all', 'gpu'):
            devs = list(range(visible))
        elif word in ('cpu', 'auto', ''):
            return None
        else:
            try:
                devs = [int(t) for t in word.replace(';', ',').split(',') if t.strip()]
            except ValueError:
                raise ValueError('unknown device specification %r' % (spec,))
    elif isinstance(spec, int):
        devs = [spec]
    else:
        was_tuple = isinstance(spec, tuple)
        spec = list(spec)
        if spec and all(isinstance(e, (tuple, list)) for e in spec):
            devs = [int(e[-1]) for e in spec]          # [(platform, device), ...]
        elif was_tuple and len(spec) == 2:
            devs = [int(spec[1])]                      # the reference's (platform, device)
        else:
            devs = [int(d) for d in spec]              # a LIST of GPU ordinals
    for d in devs:
        if not 0 <= d < visible:
            raise ValueError('GPU ordinal %d out of range (%d visible)' % (d, visible))
    return devs or None


_side_streams = {}


def _side_stream(device_index, slot):
    key = (device_index, slot)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device_index)
    return st


def kirchhoff_devices(points, samples, devices, convention=0, relaxed=False):
    """The five integrals for *points* (3 tensors) from *samples* (10 tensors), all on the
    current device, computed on *devices* (ordinals, repeats allowed) -> 5 tensors on the
    current device, ordered after its current stream.

    Streams: tile r runs on a stream of its own on its device AND has a stream of its own on
    the home device. A device-to-device copy is issued on the source device's current stream
    and makes the destination device's current stream wait for it, so while tile r is set up
    both current streams are the tile's own: no copy of one tile ever parks the home device's
    main stream or another tile's stream behind a kernel. All uploads are queued first, then
    all kernels, then all copies back; the caller's stream waits for the tiles only at the end."""
    from . import hipcalls
    home = points[0].device
    n = points[0].numel()
    ns = samples[0].numel()
    world = len(devices)
    _, nsplit, ppt = hipcalls.kirchhoff_plan(n, ns)          # the plan of the whole launch
    out = tuple(torch.empty(n, dtype=torch.complex128, device=home) for _ in range(5))
    home_index = home.index if home.index is not None else torch.cuda.current_device()
    main = torch.cuda.current_stream(home)
    ready = torch.cuda.Event()
    ready.record(main)
    work = []
    for r, d in enumerate(devices):
        p0, p1 = tile_range(n, r, world)
        if p1 > p0:
            work.append(dict(r=r, d=d, p0=p0, p1=p1, here=torch.device('cuda', d),
                             st=_side_stream(d, ('tile', r)),
                             hst=_side_stream(home_index, ('home', r))))

    def on(w):
        """both devices' current streams = the tile's own"""
        import contextlib
        stack = contextlib.ExitStack()
        stack.enter_context(torch.cuda.stream(w['hst']))
        stack.enter_context(torch.cuda.device(w['d']))
        stack.enter_context(torch.cuda.stream(w['st']))
        return stack
    for w in work:                       # 1. uploads
        with on(w):
            w['hst'].wait_event(ready)
            w['st'].wait_event(ready)
            w['smp'] = [t if t.device == w['here'] else t.to(w['here'], non_blocking=True)
                        for t in samples]
            w['pts'] = [t[w['p0']:w['p1']].to(w['here'], non_blocking=True).contiguous()
                        for t in points]
    for w in work:                       # 2. kernels
        with on(w):
            w['tile'] = hipcalls.kirchhoff(*w['pts'], *w['smp'], convention=convention,
                                           nsplit=nsplit, ppt=ppt, relaxed=relaxed)
    for w in work:                       # 3. copies back into the arrays on the home device
        with on(w):
            for o, t in zip(out, w['tile']):
                o[w['p0']:w['p1']].copy_(t, non_blocking=True)
            w['done'] = [torch.cuda.Event(), torch.cuda.Event()]
            w['done'][0].record(w['st'])
            w['done'][1].record(w['hst'])
            # (tensors made on the side streams are used by them only; the caching allocator
            # must not hand their memory to another stream before the work is done)
            for t in w['smp'] + w['pts'] + list(w['tile']):
                if t.device == w['here']:
                    t.record_stream(w['st'])
    for w in work:
        for ev in w['done']:
            main.wait_event(ev)
    return out
