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


def all_gather_tiles(local, n_total, dist, rank, world):
    """Assembles the full array from the ranks' tiles (uneven tiles are padded to
    the largest one for the collective and trimmed afterwards)."""
    if dist is None or world == 1:
        return local
    sizes = [tile_range(n_total, r, world) for r in range(world)]
    maxn = max(p1 - p0 for p0, p1 in sizes)
    if all(p1 - p0 == maxn for p0, p1 in sizes):
        out = torch.empty(n_total, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    pad = torch.zeros(maxn, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:p1 - p0] for b, (p0, p1) in zip(bufs, sizes)])


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
