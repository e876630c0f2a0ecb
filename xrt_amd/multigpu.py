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
    return tuple(all_gather_tiles(o, n, dist, rank, world) for o in out)


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
    'all' / 'GPU' -> every visible device; an int or a sequence of ints -> those."""
    if spec is None or spec == 'auto':
        env = os.environ.get('XRT_HIP_DEVICES', '').strip()
        if not env:
            return None
        spec = env if env in ('all', 'GPU') else [int(t) for t in env.replace(';', ',').split(',') if t]
    if spec in ('all', 'GPU'):
        devs = list(range(visible))
    elif isinstance(spec, int):
        devs = [spec]
    else:
        devs = [int(d) for d in spec]
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


def kirchhoff_devices(points, samples, devices, convention=0):
    """The five integrals for *points* (3 tensors) from *samples* (10 tensors), all on the
    current device, computed on *devices* (ordinals, repeats allowed) -> 5 tensors on the
    current device, ordered after its current stream."""
    from . import hipcalls
    home = points[0].device
    n = points[0].numel()
    ns = samples[0].numel()
    world = len(devices)
    _, nsplit, ppt = hipcalls.kirchhoff_plan(n, ns)          # the plan of the whole launch
    out = tuple(torch.empty(n, dtype=torch.complex128, device=home) for _ in range(5))
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(home))
    done = []
    for r, d in enumerate(devices):
        p0, p1 = tile_range(n, r, world)
        if p1 <= p0:
            continue
        st = _side_stream(d, r)
        with torch.cuda.device(d), torch.cuda.stream(st):
            st.wait_event(ready)
            here = torch.device('cuda', d)
            smp = [t if t.device == here else t.to(here, non_blocking=True) for t in samples]
            pts = [t[p0:p1].to(here, non_blocking=True).contiguous() for t in points]
            tile = hipcalls.kirchhoff(*pts, *smp, convention=convention, nsplit=nsplit,
                                      ppt=ppt)
            for o, t in zip(out, tile):
                o[p0:p1].copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
            done.append(ev)
    for ev in done:
        torch.cuda.current_stream(home).wait_event(ev)
    return out
