"""CPU, world_size 2, gloo: the pixel-tile sharding + gather used for the
multi-GPU Kirchhoff path (xrt_amd/multigpu.py) reassembles exactly the
single-rank result. The per-tile integral is computed by the numpy oracle here
(no GPU in this environment); on the GPU box the same functions wrap the HIP
kernel (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import kirchhoff_np as kn
from xrt_amd import multigpu, workloads


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, npix_side, ns, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=60))
    try:
        h = workloads.kirchhoff_custom(ns, npix_side, seed=11)
        n = h['px'].size
        p0, p1 = multigpu.tile_range(n, rank, world)
        tile = kn.kirchhoff_conv(h['px'][p0:p1], h['py'][p0:p1], h['pz'][p0:p1],
                                 h['sx'], h['sy'], h['sz'], h['n'], h['nl'],
                                 h['E'], h['Es'], h['Ep'])
        full = [multigpu.all_gather_tiles(torch.from_numpy(t), n, dist, rank, world)
                for t in tile]
        # the five arrays in ONE collective (what bench.py and kirchhoff_tiled use)
        packed = multigpu.all_gather_packed([torch.from_numpy(t) for t in tile], n, dist,
                                            rank, world)
        assert all(torch.equal(a, b) for a, b in zip(full, packed))
        # a real-valued array rides along with complex ones of the same tiling
        mixed = multigpu.all_gather_packed(
            [torch.from_numpy(tile[0]), torch.from_numpy(h['px'][p0:p1].copy())], n, dist,
            rank, world)
        assert torch.equal(mixed[1], torch.from_numpy(h['px'])) and torch.equal(mixed[0], full[0])
        ones = torch.ones(1, dtype=torch.float64)
        dist.all_reduce(ones)
        assert int(ones.item()) == world == dist.get_world_size()
        if rank == 0:
            q.put([f.numpy() for f in packed])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize('world,npix_side', [
    (2, 8), (2, 7),      # 64 pixels (even) / 49 (uneven tiles)
    (4, 7), (8, 7),      # 49 pixels over 4 / 8 ranks: tiles of 12-13 / 6-7 points
    (8, 2)])             # 4 pixels over 8 ranks: half of the ranks hold an EMPTY tile
def test_pixel_tiling_over_ranks_matches_single_rank(world, npix_side):
    ns = 300 if world == 2 else 60
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, npix_side, ns, q))
             for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = q.get(timeout=120)     # raises queue.Empty if a rank died
        for p in procs:
            p.join(60)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    finally:
        for p in procs:          # never leave a rank blocked in a collective
            if p.is_alive():
                p.terminate()
                p.join(5)
    h = workloads.kirchhoff_custom(ns, npix_side, seed=11)
    ref = kn.kirchhoff_conv(h['px'], h['py'], h['pz'], h['sx'], h['sy'], h['sz'],
                            h['n'], h['nl'], h['E'], h['Es'], h['Ep'])
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)          # tiles are independent: bit-identical


def test_tile_ranges_cover_everything():
    for n in (0, 1, 7, 64, 262144, 4194304):
        for world in (1, 2, 3, 4, 8):
            edges = [multigpu.tile_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))


@pytest.mark.timeout(300)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` as the driver types it (no torch.distributed
    environment): the script re-launches itself under torch.distributed.run with one
    rank per GPU and rank 0 prints ONE JSON line. --dry-run stops after the rendezvous
    (gloo), so this runs without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2',
                        '--steps', '2', '--warmup', '1', '--dry-run'],
                       capture_output=True, text=True, timeout=280, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    # ONE line on stdout, nothing else (gloo's connection notes and the like go to stderr)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), r.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['warmup'] == 1


@pytest.mark.timeout(600)
@pytest.mark.parametrize('gpus', [1, 8])
def test_bench_rank_choreography_without_gpus(gpus):
    """`bench.py --gpus N --dry-ranks`: everything bench.py does AROUND its kernels on a
    multi-GPU node -- self-launch, rendezvous, uneven pixel tiles, barriers, max over ranks, the
    packed all_gather (every rank checks every tile where it belongs), who-was-there, cfg5 at 8
    ranks -- with stand-in kernels on gloo, and the line the driver keeps: both halves of
    BASELINE.json's metric inside `roofline`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '2'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(gpus),
                        '--steps', '3', '--warmup', '1', '--dry-ranks'],
                       capture_output=True, text=True, timeout=580, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), r.stdout
    line = json.loads(lines[0])
    assert line['dry_ranks'] and line['n_gpus'] == gpus and line['steps'] == 3
    assert line['metric'].startswith('ray-surface intersections/sec/GPU; Kirchhoff')
    # the driver's record keeps the SCALAR members of `roofline` (nested objects are dropped,
    # VERDICT r5 weak #2) and the last 2 KB of stdout: flat keys, and a short summary at the end
    roof = line['roofline']
    assert all(not isinstance(v, (dict, list)) for v in roof.values()), roof
    assert roof['kirchhoff_cfg4_rccl_ranks'] == gpus and roof['kirchhoff_cfg4_n_gpus'] == gpus
    assert roof['kirchhoff_cfg4_pairs_per_s'] > 0 and roof['kirchhoff_cfg4_frac'] > 0
    assert roof['kirchhoff_cfg4_kernel_ms'] > 0 and roof['kirchhoff_cfg4_ms_per_step'] > 0
    assert line['kirchhoff']['kernel_ms_by_rank'] == [1.0 + r for r in range(gpus)]
    assert ('kirchhoff_cfg5_pairs_per_s' in roof) == (gpus == 8)
    assert roof['frac'] > 0 and line['value'] > 0
    assert list(line)[-1] == 'summary' and lines[0].rstrip().endswith('}}')
    tail = lines[0][-2000:]
    short = json.loads(tail[tail.index('"summary": ') + len('"summary": '):-1])
    assert short == line['summary'] and len(json.dumps(short)) <= 1500
    assert short['kirchhoff_cfg4_pairs_per_s'] == pytest.approx(
        roof['kirchhoff_cfg4_pairs_per_s'], rel=1e-4)
    assert short['intersections_per_s'] == pytest.approx(line['value'], rel=1e-4)
    assert ('kirchhoff_cfg5_pairs_per_s' in short) == (gpus == 8)


def test_device_specifications_of_the_reference():
    """targetOpenCL values that are legal in the reference (myopencl.py:187-231) select
    sensible GPUs instead of failing: 'CPU' / 'auto' -> the current device,
    'ALL' / 'GPU' in any case -> all, a (platform, device) tuple -> that device, a list of
    such tuples -> those devices; a LIST of ints stays a list of ordinals."""
    pd = multigpu.parse_devices
    old = os.environ.pop('XRT_HIP_DEVICES', None)
    try:
        assert pd('CPU', 8) is None and pd('cpu', 8) is None
        assert pd('ALL', 3) == [0, 1, 2] and pd('gpu', 2) == [0, 1]
        assert pd((0, 1), 8) == [1]
        assert pd([(0, 0), (0, 1)], 8) == [0, 1] and pd(((0, 2),), 8) == [2]
        assert pd([0, 1], 8) == [0, 1] and pd('0,3', 8) == [0, 3]
        with pytest.raises(ValueError):
            pd('fpga', 8)
        with pytest.raises(ValueError):
            pd((0, 9), 8)
    finally:
        if old is not None:
            os.environ['XRT_HIP_DEVICES'] = old


def test_in_process_device_list_and_tiles():
    """The in-process split of waves.diffract (multigpu.kirchhoff_devices): which devices a
    targetOpenCL-like spec / the XRT_HIP_DEVICES variable select, and that the tiles of any
    device count cover every receiving point exactly once, in order."""
    import os
    from xrt_amd import multigpu
    old = os.environ.pop('XRT_HIP_DEVICES', None)
    try:
        assert multigpu.parse_devices(None, 8) is None
        assert multigpu.parse_devices('auto', 8) is None
        assert multigpu.parse_devices('all', 8) == list(range(8))
        assert multigpu.parse_devices('GPU', 2) == [0, 1]
        assert multigpu.parse_devices(3, 8) == [3]
        assert multigpu.parse_devices([0, 0, 1], 2) == [0, 0, 1]
        os.environ['XRT_HIP_DEVICES'] = '0, 2,5'
        assert multigpu.parse_devices('auto', 8) == [0, 2, 5]
        os.environ['XRT_HIP_DEVICES'] = 'all'
        assert multigpu.parse_devices(None, 4) == [0, 1, 2, 3]
        try:
            multigpu.parse_devices([4], 4)
            raise AssertionError('ordinal 4 of 4 accepted')
        except ValueError:
            pass
    finally:
        os.environ.pop('XRT_HIP_DEVICES', None)
        if old is not None:
            os.environ['XRT_HIP_DEVICES'] = old
    for n in (0, 1, 7, 262144, 4194304, 1000003):
        for world in (1, 2, 3, 4, 8):
            edges = [multigpu.tile_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            assert all(p1 >= p0 for p0, p1 in edges)
