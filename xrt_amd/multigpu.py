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
