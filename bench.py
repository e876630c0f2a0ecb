#!/usr/bin/env python
"""Benchmark of the accelerated hot path (BASELINE.json metric):

  primary    ray-surface intersections/s   cfg2: 1e7 rays -> ToroidMirror(Pt),
                                           one OE.reflect per step (P1)
  kirchhoff  sample*pixel pairs/s          cfg4: 1e6 samples -> 512x512 screen
                                           (cfg5 4e6 x 2048^2 on 8 GPUs / request)

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One process per GPU: started without a torch.distributed environment and with
--gpus N > 1, the script launches itself under torch.distributed.run (N ranks on
127.0.0.1) and relays rank 0's line -- the reference also splits the devices inside
one call (myopencl.py:455-533). P1 does not shard (geometric tracing stays single-GPU,
SURVEY 8e): N ranks run N independent replicas (weak scaling). The Kirchhoff
integral shards over output-pixel tiles (strong scaling) with one RCCL
all_gather of the five complex result arrays.

Rank 0 prints ONE JSON line. ``value`` is whole-job throughput with all inputs
resident in HBM. ``roofline`` times the dominant kernel with HIP events on the
launch stream; ``cpu_baseline`` times the numpy oracle (test infrastructure) on
a bounded sample of the same workload on the host.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md chip table (spec)
FP64_PEAK = 78.6e12        # flop/s vector = matrix fp64 (256 CU x 4 SIMD x 16 lanes
#                            x 2 flop x 2.4 GHz)
BYTES_PER_INTERSECTION = 308   # SURVEY 8d / BASELINE.md: 100 B in + 2 x 100 B out + 8 B theta
FLOP_PER_PAIR = 57             # SURVEY 8d / BASELINE.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--rays', type=float, default=1e7)
    ap.add_argument('--kirchhoff-config', type=int, default=0,
                    help='4 or 5; 0 = 4, plus 5 when --gpus 8')
    ap.add_argument('--kirchhoff-steps', type=int, default=0)
    ap.add_argument('--skip-kirchhoff', action='store_true')
    ap.add_argument('--skip-undulator', action='store_true')
    ap.add_argument('--skip-softimax', action='store_true')
    ap.add_argument('--skip-balder', action='store_true')
    ap.add_argument('--skip-e2e', action='store_true')
    ap.add_argument('--skip-cpu-baseline', action='store_true')
    ap.add_argument('--with-softi-shapes', action='store_true',
                    help='also time the two Kirchhoff shapes of the reference\'s '
                         'published speed test (2e5 x 2e5 and 2e5 x 64^2)')
    ap.add_argument('--with-dcm', action='store_true',
                    help='(default on one GPU) also time cfg3 (DCM Si111, 2 intersections '
                         'per ray)')
    ap.add_argument('--skip-dcm', action='store_true')
    ap.add_argument('--dry-ranks', action='store_true',
                    help='no GPU work: the whole rank choreography of --gpus N (tiles, barriers, '
                         'max over ranks, the packed gather, who-was-there, the line with every '
                         'key the driver keeps) on gloo with stand-in kernels -- runs in the CPU '
                         'suite at N = 8')
    ap.add_argument('--dry-run', action='store_true',
                    help='no GPU work: the ranks only rendezvous (gloo) and rank 0 prints '
                         'the line skeleton -- checks the launch path on a CPU box')
    return ap.parse_args()


def self_launch(args):
    """--gpus N without a torch.distributed environment: become the launcher."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
        'HSA_ENABLE_IPC_MODE_LEGACY', '0'), OMP_NUM_THREADS=os.environ.get(
        'OMP_NUM_THREADS', '8'))
    raise SystemExit(subprocess.call(cmd, env=env))


def dry_run(args, line_out=sys.stdout):
    import datetime
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=120))
        t = torch.ones(1, dtype=torch.float64)
        dist.all_reduce(t)                       # every rank is there
        assert int(t.item()) == world
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        line_out.write(json.dumps(dict(dry_run=True, n_gpus=world, steps=args.steps,
                                       warmup=args.warmup)) + '\n')
        line_out.flush()


DRY_RANKS = False     # --dry-ranks: stand-in kernels, CPU tensors, gloo


def setup_dist(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    global CPU_GROUP
    if DRY_RANKS:
        dist = None
        if world > 1:
            import datetime
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
            dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=180))
            CPU_GROUP = dist.new_group(backend='gloo')
        return world, rank, local, dist
    # (rehearsal on a box with fewer GPUs than ranks: XRT_BENCH_SHARE_GPU=1 maps the ranks onto
    # the visible GPUs in turn and XRT_BENCH_BACKEND=gloo carries the collectives -- RCCL refuses
    # two ranks on one GPU; the driver's runs use neither)
    share = os.environ.get('XRT_BENCH_SHARE_GPU', '') == '1'
    backend = os.environ.get('XRT_BENCH_BACKEND', 'nccl')
    if share:
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
            dist.init_process_group(backend)
        # a CPU-side group for waits during which the GPUs must stay free (the in-process
        # multi-GPU leg: rank 0 drives every GPU while the others wait; an RCCL barrier would
        # park a spinning kernel on each of them)
        os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')    # one node: the loopback will do
        try:
            CPU_GROUP = dist.new_group(backend='gloo')
        except Exception:  # noqa: BLE001
            CPU_GROUP = None
    return world, rank, local, dist


CPU_GROUP = None


def barrier(dist):
    if dist is not None:
        dist.barrier()
    if not DRY_RANKS:
        torch.cuda.synchronize()


def max_over_ranks(dist, seconds):
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64,
                     device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ----------------------------------------------------------------------------
# P1
# ----------------------------------------------------------------------------
def bench_reflect_nolocal(nrays, steps=20):
    """BELOW the 308-B contract, as an extension: the same cfg2 pass with
    ``needLocal=False`` -- the reference's own switch (oes/reflect.py:104-108) for scripts that
    never look at the footprint: no local beam, no theta, 200 B per ray."""
    from xrt_amd import workloads as pc
    oe = pc.cfg2_toroid()
    beam = pc.synthetic_rays(nrays, 42)
    for f in beam.array_fields():
        beam.dev(f)
    out = None
    for _ in range(5):
        out = oe.reflect(beam, needLocal=False, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = oe.reflect(beam, needLocal=False, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n_enter = int((beam.peek('state') > 0).sum())
    return dict(metric='ray-surface intersections/s, OE.reflect(needLocal=False)',
                value=n_enter / dt, ms_per_step=dt * 1e3, bytes_per_intersection=200,
                roofline=dict(bound='hbm', kernel='reflect_fused without the local beam (whole '
                                                  'pass, host clock)',
                              achieved=200. * n_enter / dt / 1e9, peak=HBM_PEAK / 1e9,
                              unit='GB/s', frac=200. * n_enter / dt / HBM_PEAK,
                              traffic=load_traffic('reflect_fused_nolocal', 200. * nrays)
                              if nrays == 10_000_000 else None, traffic_source=TRAFFIC_SOURCE),
                note='an extension beside the primary metric, which keeps the 308-B contract of '
                     'OE.reflect with both beams')


def bench_multiple_reflect(nrays, reps=3, cpu=True):
    """OE.multiple_reflect (round 5): the toroid of tests/multi_cases.py (3 mrad, 190 mm, a point
    source 1 m upstream -- the geometry of the reference's 10_MultipleReflect example with a
    toroid), all bounces of one call; every bounce is a pass over ALL rays (the footprint
    beam lbN holds them all)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import multi_cases as case
    import xrt_amd.backends.raycing as raycing
    import xrt_amd.backends.raycing.materials as rm
    import xrt_amd.backends.raycing.oes as roe
    import xrt_amd.backends.raycing.sources as rs
    bl = raycing.BeamLine(height=0)
    oe = roe.ToroidMirror(bl, 'toroid', material=rm.Material('Au', rho=19.3, kind='mirror'),
                          **case.TOROID)
    beam = case.point_source_rays(rs, nrays, 5)
    beam.to_struct(torch.device('cuda', torch.cuda.current_device()))
    # (the batch statistics -- _info -- come from the exact phases only: one untimed call for the
    # counts, the timed ones without, as a script calls it: full bounces in their optimistic form)
    info = []
    oe.multiple_reflect(beam, maxReflections=100, _info=info)
    times = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gb, lbN = oe.multiple_reflect(beam, maxReflections=100)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = min(times[1:])
    bounces = lbN.nrays // nrays
    entered = sum(one['n_enter'] for one in info)
    per_ray = 112 + (32 if beam.has_amplitudes() else 0)       # x..Jsp, state, nRefl, theta
    moved = bounces * nrays * 2 * per_ray
    res = dict(metric='ray-surface intersections/s, OE.multiple_reflect', value=entered / dt,
               unit='intersections/s', rays=nrays, bounces=bounces, intersections=entered,
               ms_per_call=dt * 1e3, ms_per_bounce=dt * 1e3 / bounces,
               hbm_fraction=moved / dt / HBM_PEAK,
               note='a full bounce in its optimistic form (reflect_multi_opt: tangency search, hit '
                    'search, reflection and stores per ray without batch statistics, verified per '
                    'ray; reflect_multi redoes a contradicted bounce exactly), a bounce that few '
                    'rays still enter over an index of them (three launches); instruction-bound '
                    'by the reference\'s own bracket-keeping secant (20-50 iterations per ray), '
                    'not by HBM')
    if cpu:
        from oracle import fixture_io, reflect_np as rn
        p, _, _ = fixture_io.load_case('g2_multi_toroid')
        m = 200000
        ob = rn.Beam(m, with_amplitudes=True)
        src = case.point_source_rays(rs, m, 5)
        for f in ob.fields():
            setattr(ob, f, np.array(getattr(src, f)))
        t0 = time.perf_counter()
        cnt = []
        rn.oe_multiple_reflect(p, ob, 100, False, info=cnt)
        cdt = time.perf_counter() - t0
        centered = sum(int((one['tMin'] != 0).sum()) if k == 0 else 0 for k, one in enumerate(cnt))
        res['cpu_baseline'] = dict(
            value=entered / nrays * m / cdt, unit='intersections/s', cores=1, kind='port',
            sample='%d rays through oracle/reflect_np.py:oe_multiple_reflect (numpy, 1 thread), '
                   '%.1f s' % (m, cdt))
        del centered
    return res


def bench_reflect_figure(nrays, steps=10):
    """The cfg2 toroid under a figure error (OE(figureError=RandomRoughness): a 512 x 128 height
    map as a bicubic spline, evaluated per ray in every step of the intersection search and
    twice more for the normal: oes/base.py:826-830, reflect.py:767-775) -- the Figured kernels
    against the lean pass of the primary metric."""
    from xrt_amd import workloads as pc
    from xrt_amd.backends.raycing import figure_error as rfe
    oe = pc.cfg2_toroid()
    oe.figureError = rfe.RandomRoughness(rms=3., corrLength=4., seed=11, limPhysX=[-10, 10],
                                         limPhysY=[-300, 300], gridStep=2.)
    beam = pc.synthetic_rays(nrays, 42)
    for f in beam.array_fields():
        beam.dev(f)
    out = None
    for _ in range(3):
        out = oe.reflect(beam, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = oe.reflect(beam, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n_enter = int((beam.peek('state') > 0).sum())
    good = out[1].dev('state') == 1
    z = out[1].dev('z')[good][:100000].cpu().numpy()
    x, y = out[1].dev('x')[good][:100000].cpu().numpy(), out[1].dev('y')[good][:100000].cpu().numpy()
    return dict(metric='ray-surface intersections/s, toroid with a figure-error map',
                value=n_enter / dt, ms_per_step=dt * 1e3, map='RandomRoughness 3 nm rms, 512 x 128 '
                'nodes, bicubic spline',
                hit_points_off_the_distorted_surface_mm=float(np.abs(
                    z - oe.local_z(x, y) - oe.local_z_distorted(x, y)).max()),
                note='~11 spline evaluations per ray (16 coefficients each) inside the root '
                     'search: bound by their dependent loads and fp64 arithmetic, not by HBM')


def dry_reflect(args, world, rank, dist):
    """--dry-ranks: the timed region of bench_reflect with a sleep in place of the pass."""
    n = int(args.rays)
    barrier(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3 * (1 + rank % 2))
    barrier(dist)
    dt = max_over_ranks(dist, time.perf_counter() - t0)
    k = 1e-3
    return dict(value=world * n * args.steps / dt, ms_per_step=dt / args.steps * 1e3, rays=n,
                n_enter=n, surfaces=1, good_fraction=1., kernel_ms=k * 1e3, pass_ms=k * 1e3,
                roofline=dict(bound='hbm', kernel='reflect_fused (stand-in)',
                              achieved=BYTES_PER_INTERSECTION * n / k / 1e9, peak=HBM_PEAK / 1e9,
                              unit='GB/s', frac=BYTES_PER_INTERSECTION * n / k / HBM_PEAK,
                              traffic=None, traffic_source=None))


def bench_reflect(args, world, rank, dist, dcm=False):
    if DRY_RANKS:
        return dry_reflect(args, world, rank, dist)
    # The primary metric is the FULL pass: every step writes the local and the global beam
    # (308 B per intersection), an immediate launch per call -- the beams-on-demand route of
    # round 5 (oes.fuseConsumers) is switched off for these legs.
    from xrt_amd.backends.raycing import oes as roe
    fuse, roe.fuseConsumers = roe.fuseConsumers, False
    try:
        return _bench_reflect(args, world, rank, dist, dcm)
    finally:
        roe.fuseConsumers = fuse


def _bench_reflect(args, world, rank, dist, dcm=False):
    from xrt_amd import workloads as pc
    n = int(args.rays)
    seed = (43 if dcm else 42) + 1000 * rank          # replicas: own rays per rank
    if dcm:
        oe = pc.cfg3_dcm()
        beam = pc.synthetic_rays(n, seed, sa=1e-4, E=(8995., 9005.))
        op = oe.double_reflect
        surfaces = 2
    else:
        oe = pc.cfg2_toroid()
        beam = pc.synthetic_rays(n, seed)
        op = oe.reflect
        surfaces = 1
    for f in beam.array_fields():                      # inputs resident in HBM
        beam.dev(f)
    torch.cuda.synchronize()
    out = None
    kw = {'out': None}
    # untimed spin-up (not a step): a sub-millisecond step measured right after an idle
    # GPU sees its clocks and the allocator still ramping (5 steps after 2 warm-up steps
    # measured 10 % slower than 20 after 3)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.25:
        out = op(beam, **kw)
        kw['out'] = out
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = op(beam, **kw)
        kw['out'] = out              # steady state: outputs overwritten in place
    # HIP events around every pass of the timed region and around its dominant kernel,
    # recorded on the launch stream by the library without a host sync
    # (xrt_hip_reflect_time_next_pass) and read after the region
    from xrt_amd import _lib
    lib = _lib.load()
    # (every 4th step carries events: four event records per pass are not free on the
    # stream -- up to ~40 us of gaps per pass were seen -- and a sample is all the roofline
    # figure needs)
    events = {}
    if not dcm:
        for k in range(0, args.steps, 4):
            quad = [ctypes.c_void_p() for _ in range(4)]
            for e in quad:
                _lib.check(lib.xrt_hip_event_create(ctypes.byref(e)), 'event_create')
            events[k] = quad
    barrier(dist)
    t0 = time.perf_counter()
    for k in range(args.steps):
        if k in events:
            lib.xrt_hip_reflect_time_next_pass(*events[k])
        out = op(beam, **kw)
    barrier(dist)
    dt = max_over_ranks(dist, time.perf_counter() - t0)
    n_enter = int((beam.peek('state') > 0).sum())
    value = world * n_enter * surfaces * args.steps / dt
    kms, pms = [], []
    for quad in events.values():
        ms = ctypes.c_float(0.)
        _lib.check(lib.xrt_hip_event_elapsed_ms(quad[0], quad[1], ctypes.byref(ms)), 'elapsed')
        pms.append(ms.value)
        _lib.check(lib.xrt_hip_event_elapsed_ms(quad[2], quad[3], ctypes.byref(ms)), 'elapsed')
        kms.append(ms.value)
        for e in quad:
            lib.xrt_hip_event_destroy(e)
    st = out[0].peek('state')
    res = dict(value=value, ms_per_step=dt / args.steps * 1e3, rays=n,
               n_enter=n_enter, surfaces=surfaces,
               good_fraction=float((st == 1).mean()))
    if kms:
        k = float(np.mean(kms)) * 1e-3
        res['kernel_ms'] = k * 1e3
        res['pass_ms'] = float(np.mean(pms))
        res['roofline'] = dict(
            bound='hbm', kernel='reflect_fused',
            achieved=BYTES_PER_INTERSECTION * n_enter / k / 1e9,
            peak=HBM_PEAK / 1e9, unit='GB/s',
            frac=BYTES_PER_INTERSECTION * n_enter / k / HBM_PEAK,
            traffic=load_traffic('reflect_fused', BYTES_PER_INTERSECTION * n)
            if n == 10_000_000 else None,
            traffic_source=TRAFFIC_SOURCE)
    return res


def host_cpu():
    """CPU model string and core count of the box (SURVEY 8d asks for both)."""
    model = None
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.lower().startswith('model name'):
                    model = ln.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(cpu_model=model, nproc=os.cpu_count())


def cpu_baseline_reflect(nrays=10_000_000, numpy_rays=10_000_000):
    """The same cfg2 workload on the host: the numpy oracle on one core (all *nrays* rays,
    BASELINE.md section 3; ~16 s) and its C/OpenMP restatement on all cores."""
    from xrt_amd import workloads as pc
    from oracle.adapters import oracle_params, to_oracle_beam
    from oracle import reflect_np as rn
    oe = pc.cfg2_toroid()
    beam = to_oracle_beam(pc.synthetic_rays(nrays, 42))
    params = oracle_params(oe)
    m = min(nrays, numpy_rays)
    part = rn.Beam(m)
    for f in part.fields():
        setattr(part, f, getattr(beam, f)[:m].copy())
    t0 = time.perf_counter()
    rn.oe_reflect(params, part)
    dt = time.perf_counter() - t0
    res = dict(value=m / dt, unit='intersections/s', cores=1, kind='port',
               sampled=m < nrays,
               sample=('a SAMPLE of the workload: the first %d of its %d rays' % (m, nrays)
                       if m < nrays else 'the whole workload: %d rays' % nrays) +
               ' through oracle/reflect_np.py (numpy, 1 thread), %.1f s' % dt)
    res.update(host_cpu())
    # all host cores: the C/OpenMP restatement of the same pass (oracle/reflect_c.c,
    # validated against reflect_np and the reference's golden G2 by
    # tests/test_oracle_reflect_c.py), same 1e7 rays
    try:
        from oracle import reflect_c as rc
        rc.oe_reflect(params, to_oracle_beam(pc.synthetic_rays(20000, 1)))    # warm up
        t0 = time.perf_counter()
        rc.oe_reflect(params, beam)
        dtc = time.perf_counter() - t0
        res['all_cores'] = dict(
            value=nrays / dtc, unit='intersections/s', cores=rc.max_threads(), kind='port',
            sample='%d rays of cfg2 through oracle/reflect_c.c (gcc -O2 -fopenmp, %d '
                   'threads), %.2f s' % (nrays, rc.max_threads(), dtc))
    except Exception as e:          # a baseline must never take the bench line down
        res['all_cores'] = dict(error=repr(e))
    return res


# ----------------------------------------------------------------------------
# P2
# ----------------------------------------------------------------------------
def multi_gpu_self_check(world, rank, dist):
    """N > 1 only, before anything is timed: a small Kirchhoff integral (2e4 samples x 64 x 64
    points) computed THREE ways -- every rank its pixel tile + the packed all_gather over RCCL
    (what the timed steps do), rank 0 alone on its own GPU, and rank 0 driving two DISTINCT GPUs
    in one process (multigpu.kirchhoff_devices on devices [0, 1]: peer copies between real
    devices, the parametrisations of tests/test_gpu_kirchhoff.py and test_gpu_diffract.py that
    skip on a one-GPU box) -- must agree to 1e-12 norm-wise. The outcome goes into the line
    (`roofline.multi_gpu_self_check` 1 / 0 and a message): a failure is reported, it does not
    take the measured numbers down with it (the split mirrored is
    xrt/backends/raycing/myopencl.py:455-533)."""
    from xrt_amd import hipcalls, multigpu, workloads
    out = dict(ok=False, ranks=world)
    ok = False
    dev = torch.device('cuda', torch.cuda.current_device())
    alone = smp = h = up = None
    # (every rank reaches every collective whatever happens on another one: the rank-local parts
    # have their own try blocks, the all_gather and the all_reduce are outside of them)
    try:
        h = workloads.kirchhoff_custom(20000, 64)
        up = lambda a, dt=np.float64: torch.from_numpy(  # noqa: E731
            np.ascontiguousarray(a, dtype=dt)).to(dev)
        ns, npix = h['ns'], h['px'].size
        smp = [up(h['sx']), up(h['sy']), up(h['sz']), up(np.zeros(ns)), up(np.ones(ns)),
               up(np.zeros(ns)), up(h['nl']), up(h['k']), up(h['Es'], np.complex128),
               up(h['Ep'], np.complex128)]
        p0, p1 = multigpu.tile_range(npix, rank, world)
        tile = hipcalls.kirchhoff(up(h['px'][p0:p1]), up(h['py'][p0:p1]), up(h['pz'][p0:p1]),
                                  *smp)[:5]
    except Exception as e:  # noqa: BLE001  (reported, never fatal: see the docstring)
        out['error'] = repr(e)[:300]
        tile = None
    try:
        if tile is None:      # (this rank has nothing: zeros of the tile's shape keep the gather whole)
            npix = 64 * 64
            p0, p1 = multigpu.tile_range(npix, rank, world)
            tile = [torch.zeros(p1 - p0, dtype=torch.complex128, device=dev) for _ in range(5)]
        full = multigpu.all_gather_packed(tile, npix, dist, rank, world)
        if 'error' not in out:
            alone = hipcalls.kirchhoff(up(h['px']), up(h['py']), up(h['pz']), *smp)[:5]
            worst = 0.
            for a, b in zip(full, alone):
                worst = max(worst, float((a - b).abs().max() / b.abs().max().clamp_min(1e-300)))
            out['gather_vs_one_gpu'] = worst
            ok = worst <= 1e-12
    except Exception as e:  # noqa: BLE001
        out['error'] = repr(e)[:300]
    if alone is not None and rank == 0 and torch.cuda.device_count() >= 2 and \
            not os.environ.get('XRT_BENCH_SHARE_GPU'):
        try:
            both = multigpu.kirchhoff_devices((up(h['px']), up(h['py']), up(h['pz'])), smp, [0, 1])
            for d in (0, 1):
                torch.cuda.synchronize(d)
            w2 = 0.
            for a, b in zip(both, alone):
                w2 = max(w2, float((a.to(dev) - b).abs().max() / b.abs().max().clamp_min(1e-300)))
            out['two_devices_in_one_process_vs_one_gpu'] = w2
            ok = ok and w2 <= 1e-12
        except Exception as e:  # noqa: BLE001
            out['two_devices_error'] = repr(e)[:300]
            ok = False
    try:
        flag = torch.tensor([1. if ok else 0.], dtype=torch.float64,
                            device=dev if dist.get_backend() == 'nccl' else torch.device('cpu'))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        out['ok'] = bool(flag.item() > 0.5)
    except Exception as e:  # noqa: BLE001
        out.setdefault('error', repr(e)[:300])
    return out


def kirchhoff_inputs(cfg, device):
    """SURVEY 8d cfg4 / cfg5 (xrt_amd.workloads.kirchhoff_case), samples
    uploaded to HBM."""
    from xrt_amd import workloads
    h = workloads.kirchhoff_case(cfg)
    ns = h['ns']
    up = lambda a, dt=np.float64: torch.from_numpy(  # noqa: E731
        np.ascontiguousarray(a, dtype=dt)).to(device)
    samples = dict(sx=up(h['sx']), sy=up(h['sy']), sz=up(h['sz']),
                   nx=up(np.zeros(ns)), ny=up(np.ones(ns)), nz=up(np.zeros(ns)),
                   nl=up(h['nl']), k=up(h['k']), Es=up(h['Es'], np.complex128),
                   Ep=up(h['Ep'], np.complex128))
    return samples, (h['px'], h['py'], h['pz']), h, ns, h['side']


def bench_kirchhoff(cfg, steps, warmup, world, rank, dist):
    from xrt_amd import multigpu
    if DRY_RANKS:
        # stand-in: a small mesh whose tiles are uneven, the "kernel" writes the rank's number
        dev = torch.device('cpu')
        ns, side = {4: (1000, 35), 5: (4000, 51)}[cfg]
        npix = side * side
        host, s = None, None
        p0, p1 = multigpu.tile_range(npix, rank, world)
        out = tuple(torch.empty(p1 - p0, dtype=torch.complex128) for _ in range(5))

        def step(timing=False):
            for j, t in enumerate(out):
                t[:] = complex(rank, j)
            time.sleep(1e-3)
            full = out
            if dist is not None:
                full = multigpu.all_gather_packed(out, npix, dist, rank, world)
                for r in range(world):          # every rank sees every tile where it belongs
                    q0, q1 = multigpu.tile_range(npix, r, world)
                    assert all(bool((full[j][q0:q1] == complex(r, j)).all()) for j in range(5))
            return list(full) + [1.0 + rank]
    else:
        from xrt_amd import hipcalls
        dev = torch.device('cuda', torch.cuda.current_device())
        s, (px, py, pz), host, ns, side = kirchhoff_inputs(cfg, dev)
        npix = px.size
        p0, p1 = multigpu.tile_range(npix, rank, world)             # pixel tile
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a[p0:p1])).to(dev)  # noqa: E731
        tx, ty, tz = up(px), up(py), up(pz)
        out = tuple(torch.empty(p1 - p0, dtype=torch.complex128, device=dev)
                    for _ in range(5))

        def step(timing=False):
            r = hipcalls.kirchhoff(tx, ty, tz, s['sx'], s['sy'], s['sz'], s['nx'],
                                   s['ny'], s['nz'], s['nl'], s['k'], s['Es'], s['Ep'],
                                   convention=0, out=out, timing=timing)
            if dist is not None:        # assemble the full field on every rank: ONE RCCL
                multigpu.all_gather_packed(out, npix, dist, rank, world)   # all_gather of [5, tile]
            return r
    for _ in range(warmup):
        step()
    barrier(dist)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier(dist)
    dt = max_over_ranks(dist, time.perf_counter() - t0)
    pairs = float(ns) * float(npix)
    kms = [step(timing=True)[5] for _ in range(2)]
    k = float(np.mean(kms)) * 1e-3
    my_pairs = float(ns) * float(p1 - p0)
    # who was there: the size of the RCCL group as the collective itself sees it, and every
    # rank's kernel time (the slowest tile sets the step)
    rccl_ranks, kernel_ms_by_rank = 1, [k * 1e3]
    if dist is not None:
        cdev = dev if dist.get_backend() == 'nccl' else torch.device('cpu')
        ones = torch.ones(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        assert rccl_ranks == dist.get_world_size() == world
        mine = torch.zeros(world, dtype=torch.float64, device=cdev)
        mine[rank] = k * 1e3
        dist.all_reduce(mine)
        kernel_ms_by_rank = [float(v) for v in mine.tolist()]
    # the same launch through ONE process that tiles the receiving points over the GPUs itself
    # (what waves.diffract does with devices = [...], multigpu.kirchhoff_devices): rank 0
    # drives all of them while the other ranks wait
    in_process = None
    if world > 1 and CPU_GROUP is not None and not DRY_RANKS:
        barrier(dist)
        if rank == 0:
            try:        # (a figure beside the main one: it must never take the line down)
                fx, fy, fz = (torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                              for a in (px, py, pz))
                smp = [s[f] for f in ('sx', 'sy', 'sz', 'nx', 'ny', 'nz', 'nl', 'k', 'Es', 'Ep')]
                devs = [d % torch.cuda.device_count() for d in range(world)]

                def sync_all():
                    for d in devs:
                        torch.cuda.synchronize(d)
                multigpu.kirchhoff_devices((fx, fy, fz), smp, devs)
                sync_all()
                t0 = time.perf_counter()
                for _ in range(steps):
                    multigpu.kirchhoff_devices((fx, fy, fz), smp, devs)
                sync_all()
                dti = time.perf_counter() - t0
                in_process = dict(value=pairs * steps / dti, unit='pairs/s',
                                  ms_per_step=dti / steps * 1e3, devices=devs,
                                  note='one process, one stream per device, samples copied '
                                       'device to device, tiles copied into the arrays on '
                                       'device 0')
            except Exception as e:  # noqa: BLE001
                in_process = dict(error=repr(e))
        dist.barrier(group=CPU_GROUP)       # (the other ranks wait here on the CPU)
        barrier(dist)
    res = dict(
        metric='Kirchhoff sample*pixel pairs/s', value=pairs * steps / dt,
        unit='pairs/s', n_gpus=world, steps=steps, warmup=warmup,
        ms_per_step=dt / steps * 1e3, scaling='strong', dtype='f64',
        config=dict(workload='cfg%d: %d samples -> %dx%d screen, fp64, pixel-tiled'
                             % (cfg, ns, side, side), samples=ns, pixels=npix,
                    parallelism='pixel tiles x%d + all_gather' % world),
        kernel_ms=k * 1e3, kernel_ms_by_rank=kernel_ms_by_rank, rccl_ranks=rccl_ranks,
        gather='one all_gather_into_tensor of the packed [5, tile] complex results per step '
               '(%.1f MB per rank)' % (5 * 16 * (p1 - p0) / 1e6) if world > 1 else None,
        in_process=in_process,
        roofline=dict(bound='valu_fp64', kernel='kirchhoff_stream',
                      note='fp64 VALU kernel (no MFMA: profiles/r01_mfma_f64_probe.txt); '
                           'MI355X vector fp64 peak = matrix fp64 peak = 78.6 TFLOP/s '
                           'at 2.4 GHz; 57 flop per pair (sqrt, div, sin, cos counted '
                           'as 1)',
                      achieved=FLOP_PER_PAIR * my_pairs / k / 1e12,
                      peak=FP64_PEAK / 1e12, unit='TFLOP/s',
                      frac=FLOP_PER_PAIR * my_pairs / k / FP64_PEAK,
                      traffic=load_traffic('kirchhoff_stream')
                      if cfg == 4 and world == 1 else None,
                      traffic_source=TRAFFIC_SOURCE,
                      traffic_note='the packed sample records re-streamed through the '
                                   'scalar cache by every block; unique bytes are ~0.15 GB '
                                   '(compute-bound kernel, HBM at <1 % of peak)'))
    return res, host


def bench_kirchhoff_general(reps=3):
    """The loop every mirror -> mirror wave transfer runs (both polarisations, per-sample
    normals, receiving points off a plane: kirchhoff_stream's GEN_SP_N variant) on
    workloads.kirchhoff_general: 2e5 samples x 2e5 points, the shape of the seven large
    integrals of the reference's SoftiMAX speed test."""
    from xrt_amd import hipcalls, workloads
    dev = torch.device('cuda', torch.cuda.current_device())
    h = workloads.kirchhoff_general()
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    args = [up(h[f]) for f in ('px', 'py', 'pz', 'sx', 'sy', 'sz', 'nx', 'ny', 'nz', 'nl', 'k',
                               'Es', 'Ep')]
    hipcalls.kirchhoff(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        hipcalls.kirchhoff(*args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kms = float(np.mean([hipcalls.kirchhoff(*args, timing=True)[5] for _ in range(2)]))
    pairs = float(h['ns']) * float(h['npix'])
    variants = sorted(hipcalls.kirchhoff_report()['variants'])
    # the opt-in relaxed loop on the same inputs: its time, and what it loses against the exact
    # sums of the same launch (norm-wise)
    exact = [o.clone() for o in hipcalls.kirchhoff(*args)[:5]]
    rel = hipcalls.kirchhoff(*args, relaxed=True, timing=True)
    rms = float(np.mean([hipcalls.kirchhoff(*args, relaxed=True, timing=True)[5]
                         for _ in range(2)]))
    rvariants = sorted(hipcalls.kirchhoff_report()['variants'])
    loss = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(rel[:5], exact))
    relaxed = dict(kernel_ms=rms, frac=FLOP_PER_PAIR * pairs / (rms * 1e-3) / FP64_PEAK,
                   loop_variants=rvariants, normwise_difference_from_exact=loss,
                   note='opt-in (waves.precision = "relaxed", XRT_HIP_KIRCHHOFF_RELAXED): d.d '
                        'contracted, the root without its last correction, k folded into the '
                        'phase reduction; 57 flop per pair counted as for the exact loop')
    return dict(
        relaxed=relaxed,
        metric='Kirchhoff sample*pixel pairs/s, general loop', value=pairs / dt, unit='pairs/s',
        samples=h['ns'], pixels=h['npix'], ms_per_call=dt * 1e3, kernel_ms=kms,
        loop_variants=variants,
        roofline=dict(
            bound='valu_fp64', kernel='kirchhoff_stream (general geometry, Es and Ep, '
                                      'per-sample normals)',
            achieved=FLOP_PER_PAIR * pairs / (kms * 1e-3) / 1e12, peak=FP64_PEAK / 1e12,
            unit='TFLOP/s', frac=FLOP_PER_PAIR * pairs / (kms * 1e-3) / FP64_PEAK, traffic=None,
            note='57 flop per pair; this loop issues 57 VALU instructions per pair, one of them '
                 'the quarter-rate v_rsq_f64 (60 issue slots): 57 / (2 x 60) = 0.475 of peak is '
                 'what the formulation allows at full issue (DESIGN 5.1)'))


def bench_hist(nrays):
    """N2: the histograms of one XYCPlot (2-D flux + RGB planes, three 1-D histograms with
    flux / R / G / B weights, ray counters) from a device-resident beam of *nrays* rays --
    the step after the hot path in every run_ray_tracing iteration. Algorithmic bytes: 44 B
    per ray (x, y, colour datum, state, Jss, Jpp), read once."""
    from xrt_amd import workloads, plotter as xrtp, runner
    oe = workloads.cfg2_toroid()
    beam = workloads.synthetic_rays(nrays, 42)
    for f in beam.array_fields():
        beam.dev(f)
    gb, lb = oe.reflect(beam)
    res = dict(metric='plot histograms of one XYCPlot, rays/s', rays=nrays, unit='rays/s',
               dtype='f64')
    for bins in (128, 256):
        plot = xrtp.XYCPlot('b', (1,), xrtp.XYCAxis('x', 'mm', bins=bins),
                            xrtp.XYCAxis('y', 'mm', bins=bins),
                            caxis=xrtp.XYCAxis('energy', 'eV', bins=bins))
        runner.accumulate_plot(plot, {'b': lb})          # sets the limits
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            runner.accumulate_plot(plot, {'b': lb})
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        total = plot.total2D.sum()                       # brings the accumulators home
        t2 = time.perf_counter()
        ms = (t1 - t0) / reps * 1e3
        key = 'bins%d' % bins
        res[key] = dict(ms_per_plot=ms, value=nrays / ms * 1e3, read_back_ms=(t2 - t1) * 1e3,
                        flux_in_range=float(total),
                        note='accumulate_plot per iteration (the plot\'s accumulators stay on the '
                             'device: no copy, no sync per iteration); read_back_ms = once, when '
                             'the plot is read')
        if bins == 256:
            res['value'] = nrays / ms * 1e3
            res['ms_per_plot'] = ms
            res['roofline'] = dict(
                bound='hbm', kernel='plot_hist_rays + plot_hist_tiles + plot_hist_reduce '
                                    '(256 x 256 bins + 3 x 1-D)',
                achieved=44. * nrays / (ms * 1e-3) / 1e9, peak=HBM_PEAK / 1e9, unit='GB/s',
                frac=44. * nrays / (ms * 1e-3) / HBM_PEAK,
                traffic=load_hist_traffic(bins) if nrays == 10_000_000 else None,
                traffic_source='profiles/hist_traffic.json: PMC passes of tools/pmc_hist.sh, '
                               'committed -- static, not collected in this run',
                note='44 B per ray algorithmic (x, y, colour datum, state, Jss, Jpp, read once); '
                     'the four fp64 planes of the plot are 2 MB, a CU has 160 KB of LDS, and '
                     'global fp64 atomics run at 2.4e10 /s on this chip (tools/probes/'
                     'probe_atomics.hip = 1.7 ms for the 4e7 updates): the rays are sorted by '
                     'tile of 64 x 64 bins on the way (20 B per ray written and read again) and '
                     'accumulated in LDS, so the traffic is 84 B per ray + the per-CU plane '
                     'copies; the blocks of the tile pass are shared out by the tiles\' ray '
                     'counts and table walks (round 4)')
    return res


def bench_e2e(nrays, repeats=20):
    """One whole ``run_ray_tracing`` job, nothing resident beforehand: every iteration makes its
    rays (GeometricSource on the device: csrc/source.hip), reflects them on the cfg2 toroid,
    exposes a screen at the focus and adds the screen beam to one 256 x 256 XYCPlot (2-D flux +
    RGB, three 1-D histograms); the reference's loop is xrt/runner.py:513-719. Reported: ms per
    iteration by the host clock over *repeats* iterations, the GPU time of the same work
    (HIP events around the four steps of separately instrumented iterations) and their ratio
    = the fraction of the wall time the GPU is busy; beside it the same job with the host
    source (numpy in the reference's RNG order, rays uploaded)."""
    from xrt_amd import workloads, runner
    from xrt_amd.backends.raycing import run as rr
    bl, run_process, make_plot = workloads.e2e_beamline(nrays)
    rr.run_process = run_process
    runner.run_ray_tracing([make_plot()], repeats=3, beamLine=bl)           # warm up
    torch.cuda.synchronize()
    # three blocks of *repeats* iterations, as the Balder leg does: the first follows seconds of
    # host-side preparation on an idle GPU; the steady state is the figure, all three are reported
    walls = []
    for _ in range(3):
        plot = make_plot()
        t0 = time.perf_counter()
        runner.run_ray_tracing([plot], repeats=repeats, beamLine=bl)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) / repeats)
    wall = min(walls)
    t1 = time.perf_counter()
    flux = float(plot.total2D.sum())
    read_back = time.perf_counter() - t1
    # the same job with the plot as launches of its own (round 5: the screen alone rides the pass,
    # the 100-B image is written and read back) ...
    os.environ['XRT_PLOT_TAIL_OFF'] = '1'
    try:
        runner.run_ray_tracing([make_plot()], repeats=3, beamLine=bl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run_ray_tracing([make_plot()], repeats=repeats, beamLine=bl)
        torch.cuda.synchronize()
        wall_own = (time.perf_counter() - t0) / repeats
    finally:
        del os.environ['XRT_PLOT_TAIL_OFF']
    # ... and without any plot (the pass with the screen in its tail alone): what the plot adds
    from xrt_amd.backends.raycing import sources as _rs

    def no_plot():
        beams = run_process(bl)
        beams['focus'].nrays                  # (the first look at the image launches the pass)
        _rs.flush_pending()
    for _ in range(3):
        no_plot()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(repeats):
        no_plot()
    torch.cuda.synchronize()
    wall_pass = (time.perf_counter() - t0) / repeats
    # the same job with every call an immediate launch and every beam written, looked at or not
    # (what rounds 1-4 did)
    from xrt_amd.backends.raycing import oes as roe
    roe.fuseConsumers = False
    try:
        runner.run_ray_tracing([make_plot()], repeats=3, beamLine=bl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run_ray_tracing([make_plot()], repeats=repeats, beamLine=bl)
        torch.cuda.synchronize()
        wall_all = (time.perf_counter() - t0) / repeats
    finally:
        roe.fuseConsumers = True
    # GPU time per step of an iteration: events on the launch stream
    steps = ('source', 'reflect', 'screen', 'histograms')
    dev_ms = dict.fromkeys(steps, 0.)
    probe = make_plot()
    n_probe = 5
    for _ in range(n_probe):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        src = bl.source.shine()
        ev[1].record()
        gb, lb = bl.mirror.reflect(src)
        ev[2].record()
        img = bl.screen.expose(gb)
        ev[3].record()
        runner.accumulate_plot(probe, {'focus': img})
        ev[4].record()
        torch.cuda.synchronize()
        for k, name in enumerate(steps):
            dev_ms[name] += ev[k].elapsed_time(ev[k + 1]) / n_probe
    gpu_ms = sum(dev_ms.values())
    res = dict(
        metric='run_ray_tracing end to end: source -> toroid mirror -> screen -> XYCPlot, '
               'rays/s', rays=nrays, repeats=repeats, ms_per_iteration=wall * 1e3,
        ms_per_iteration_by_block=[w * 1e3 for w in walls],
        ms_per_iteration_first_block=walls[0] * 1e3,
        ms_per_iteration_note='best of three blocks of %d iterations (all in '
                              'ms_per_iteration_by_block; the first is what rounds 1-5 reported)' % repeats,
        value=nrays / wall, unit='rays/s', dtype='f64', source='device (Philox4x32-10)',
        gpu_ms_per_iteration=gpu_ms, gpu_ms_by_step=dev_ms, gpu_busy=gpu_ms / (wall * 1e3),
        gpu_busy_note='GPU time of one iteration (HIP events around its four steps, %d '
                      'instrumented iterations with a sync each) / host-clock time per iteration '
                      'of the free-running loop; ~1 = the loop is GPU-bound (the instrumented '
                      'iterations carry a few us of event gaps: the ratio can exceed 1)' % n_probe,
        read_back_ms_once=read_back * 1e3, flux_in_plot=flux,
        ms_per_iteration_every_beam_written=wall_all * 1e3,
        ms_per_iteration_plot_as_own_launches=wall_own * 1e3,
        ms_per_iteration_without_plot=wall_pass * 1e3,
        plot_adds_ms=(wall - wall_pass) * 1e3,
        bytes_per_ray=dict(plot_records_written=20.5, plot_records_read=20.5,
                           round5_image_and_histograms=184, as_separate_passes=652),
        fused='GeometricSource.shine, OE.reflect, Screen.expose and the XYCPlot of the image are '
              'ONE pass (reflect_fused_gen_scr_plot, round 6): the rays are made in the '
              'registers of the mirror kernel, the screen\'s image stays in registers, the tail '
              'forms weight, hue and bins and every wave writes its 64 rays sorted by tile of '
              'the 2-D histogram as 20-B records; plot_tail_tiles and plot_hist_reduce add them '
              'up. Neither the source beam, the mirror\'s beams nor the image are written (each '
              'is made on demand, by the same kernels, the first time somebody looks at it, '
              'and written at once from then on -- tests/test_gpu_fusion.py). '
              'ms_per_iteration_plot_as_own_launches = round 5\'s route (the image written, '
              'plot_hist_rays / tiles / reduce); ms_per_iteration_without_plot = the pass with '
              'the screen alone; ms_per_iteration_every_beam_written = four immediate launches '
              'that write everything (oes.fuseConsumers = False). The "histograms" step of '
              'gpu_ms_by_step is the pass AND the plot now (accumulate_plot launches both)',
        roofline=dict(bound='hbm', kernel='the pass and the plot of one iteration',
                      achieved=41. * nrays / wall / 1e9, peak=HBM_PEAK / 1e9, unit='GB/s',
                      frac=41. * nrays / wall / HBM_PEAK,
                      traffic=load_plot_tail_traffic() if nrays == 10_000_000 else None,
                      traffic_source='profiles/plot_tail_traffic.json (tools/pmc_plot_tail.sh)',
                      note='41 B per ray algorithmic as built (20.5 written by the pass, 20.5 '
                           'read by plot_tail_tiles; round 5: 184) -- the pass is bound by its '
                           'arithmetic (Philox, Box-Muller, the root search, the bins), not by '
                           'HBM.'))
    # beams of the size most xrt scripts trace (1e5 rays per iteration): the host, not the GPU,
    # bounds the eager loop; run_ray_tracing(graph=True) replays one HIP graph per iteration
    small = {}
    for n_small in () if os.environ.get('XRT_E2E_NO_SMALL') else (100_000, 1_000_000):
        bls, run_s, make_s = workloads.e2e_beamline(n_small)
        rr.run_process = run_s
        row = {}
        for mode in ('eager', 'graph'):
            # (long enough for the recording -- ~6 ms -- and the 44 iterations of the contest of
            # the graph route to be what they are in a real run: a small part)
            reps = 1000 if n_small <= 100_000 else 500
            runner.run_ray_tracing([make_s()], repeats=3, beamLine=bls, graph=mode == 'graph')
            torch.cuda.synchronize()
            ps = make_s()
            t0 = time.perf_counter()
            runner.run_ray_tracing([ps], repeats=reps, beamLine=bls, graph=mode == 'graph')
            torch.cuda.synchronize()
            # (the graph run spends its first two iterations eagerly and records the third)
            row[mode + '_ms_per_iteration'] = (time.perf_counter() - t0) / reps * 1e3
            row[mode + '_flux'] = float(ps.total2D.sum())
            if mode == 'graph':     # what the graph route measured on this box and chose
                row['graph_choice'] = getattr(ps, 'graphChoice', None)
        row['speedup'] = row['eager_ms_per_iteration'] / row['graph_ms_per_iteration']
        small['%d_rays' % n_small] = row
    if small:
        small['note'] = ('the same job at 1e5 and 1e6 rays per iteration, 1000 / 500 iterations: eager '
                         'loop (Python + ctypes per element) against run_ray_tracing(graph=True) '
                         '(one HIP graph launch per iteration, xrt_amd/graphs.py); the graph time '
                         'includes its two eager iterations and the recording')
        res['small_beams'] = small
    rr.run_process = run_process
    if os.environ.get('XRT_E2E_NO_HOST'):
        return res
    # the same job with the host source
    blh, run_h, make_h = workloads.e2e_beamline(nrays, rng='host')
    rr.run_process = run_h
    np.random.seed(0)
    hp = make_h()
    runner.run_ray_tracing([hp], repeats=1, beamLine=blh)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.run_ray_tracing([hp], repeats=2, beamLine=blh)
    torch.cuda.synchronize()
    host_wall = (time.perf_counter() - t0) / 2
    res['host_source'] = dict(ms_per_iteration=host_wall * 1e3, value=nrays / host_wall,
                              note='GeometricSource(rng=\'host\'): numpy sampling on one core in '
                                   'the reference\'s order + upload of the beam, then the same '
                                   'three GPU steps')
    res['speedup_vs_host_source'] = host_wall / wall
    return res


def bench_softi_shapes():
    """The reference's only published P2 numbers are whole-script times of
    tests/speed/3_Softi_CXIw2D_speed.py: 7 diffract calls of <= 2e5 x 2e5 pairs
    and 3 of <= 2e5 x 4096 (derived A100 rate
    ~1.6e10 pairs/s incl. everything else the script does). This times our
    kernel on synthetic data of those two shapes."""
    from xrt_amd import hipcalls, workloads
    dev = torch.device('cuda', torch.cuda.current_device())
    up = lambda a, dt=np.float64: torch.from_numpy(  # noqa: E731
        np.ascontiguousarray(a, dtype=dt)).to(dev)
    res = {}
    for tag, ns, side in (('2e5x2e5', 200_000, 447), ('2e5x64x64', 200_000, 64)):
        h = workloads.kirchhoff_custom(ns, side)
        args = [up(h['px']), up(h['py']), up(h['pz']), up(h['sx']), up(h['sy']),
                up(h['sz']), up(np.zeros(ns)), up(np.ones(ns)), up(np.zeros(ns)),
                up(h['nl']), up(h['k']), up(h['Es'], np.complex128),
                up(h['Ep'], np.complex128)]
        hipcalls.kirchhoff(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            hipcalls.kirchhoff(*args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        pairs = float(ns) * h['px'].size
        res[tag] = dict(samples=ns, pixels=int(h['px'].size), ms=dt * 1e3,
                        pairs_per_s=pairs / dt)
    total = 7 * res['2e5x2e5']['ms'] + 3 * res['2e5x64x64']['ms']
    res['ten_calls_ms'] = total
    res['note'] = ('7 x (2e5 x 2e5) + 3 x (2e5 x 4096) Kirchhoff calls, kernel + '
                   'pack + finalize, inputs resident; the reference publishes '
                   '17.5 s (1 x A100) for the whole script that contains them')
    return res


def bench_undulator(with_cpu=True):
    """N3: the fused far-field map (Undulator.build_I_map) on 2^20 rays x 48
    nodes of a planar undulator, inputs resident; CPU leg = the numpy oracle of
    the reference's _sp_sum on a bounded sample."""
    from xrt_amd import hipcalls
    dev = torch.device('cuda', torch.cuda.current_device())
    rng = np.random.RandomState(5)
    n, Kx, Ky, Np, L0, gamma0 = 1 << 20, 0., 0.52, 108, 18.5, 5870.853297866972
    quadm, gi = 24, 2
    dstep = 2 * np.pi / gi
    from xrt_amd.backends.raycing.undulator import clenshaw_curtis
    xk, wk = clenshaw_curtis(quadm)
    dI = np.arange(-np.pi + 0.5 * dstep, np.pi, dstep)
    tg = (dI[:, None] + 0.5 * dstep * xk).ravel()
    ag = (dI[:, None] * 0 + wk).ravel()
    tabs_h = (tg, ag, np.sin(tg), np.cos(tg), np.sin(tg), np.cos(tg))
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    tabs = [up(t) for t in tabs_h]
    w = rng.uniform(3900., 4250., n)
    th = rng.uniform(-3e-5, 3e-5, n)
    ps = rng.uniform(-3e-5, 3e-5, n)
    dw, dth, dps = up(w), up(th), up(ps)
    call = lambda: hipcalls.undulator_imap(  # noqa: E731
        0, Kx, Ky, tabs, dw, dth, dps, L0, Np, gamma0, 0.5, dstep, True)
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nodes = len(tg)
    res = dict(metric='undulator field map, ray-nodes/s (far field, fused '
                      'build_I_map)', rays=n, nodes=nodes, ms=ms,
               value=n * nodes / ms * 1e3, unit='ray-nodes/s', dtype='f64')
    # Static count from the gfx950 ISA of und_imap<0>'s node loop: 86 flop
    # (37 mul + 23 add + 13 fma as the reference writes them) in 50 VALU instructions for a
    # planar undulator (Kx = 0: every term with Kx in it is an exact zero added to something
    # and is left out, same bits; 60 with both fields; 81 in round 2), one of them v_rcp_f64,
    # a quarter-rate instruction: 53 issue slots. The sum of krel keeps the reference's
    # roundings, the products behind it are fused.
    flop, slots = 86, 53
    tf = flop * n * nodes / ms * 1e3 / 1e12
    res['roofline'] = dict(
        bound='valu_fp64', kernel='und_imap', achieved=tf, peak=78.6, unit='TFLOP/s',
        frac=tf / 78.6, traffic=None,
        note='%d flop / %d VALU issue slots per ray-node; issue-slot use = %.2f of the '
             '3.93e13 lane-slots/s of 256 CUs x 4 SIMDs at 2.4 GHz (fp64-dense code runs at '
             '2.0-2.15 GHz: tools/probes/probe_fp64_rates.hip); 64 B of HBM traffic per ray' % (flop, slots, slots * n * nodes / ms * 1e3 / 3.93e13))
    if with_cpu:
        from oracle import undulator_np as un
        m = 100_000
        tab = dict(zip(('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph'), tabs_h),
                   dstep=dstep)
        t0 = time.perf_counter()
        un.intensity_map(0, Kx, Ky, Np, L0, gamma0, 0.5, True, tab, w[:m], th[:m],
                         ps[:m])
        dt = time.perf_counter() - t0
        res['cpu_baseline'] = dict(
            value=m * nodes / dt, unit='ray-nodes/s', cores=1, kind='port', sampled=True,
            sample='a SAMPLE of the workload: %d of its %d rays x %d nodes through '
                   'oracle/undulator_np.py (numpy restatement of the reference\'s _sp_sum), '
                   '%.1f s' % (m, n, nodes, dt))
        res['cpu_baseline'].update(host_cpu())
    return res


def bench_balder(nrays, runs=5, both=True):
    """A whole ray-tracing beamline, element after element on device-resident beams: the
    reference's example Balder (examples/withRaycing/02_Balder_BL) from the front-end mask to
    the sample -- diamond filter (two surfaces), bent collimating mirror, DCM Si(111) (two
    crystals), toroidal focusing mirror = six ray-surface intersections per ray, four
    apertures, two screens. The rays are a synthetic pencil that fills the front-end mask
    (the wiggler's sampling is host-side numpy, as in the reference, and not part of this
    number). Parity of this chain with the reference: tests/test_gpu_balder.py (golden
    G17)."""
    import numpy as np
    import torch
    from xrt_amd import workloads
    import xrt_amd.backends.raycing.sources as rs
    rng = np.random.default_rng(17)
    beam = rs.Beam(nrays=nrays)
    beam.x, beam.z = rng.normal(0, 0.05, nrays), rng.normal(0, 0.01, nrays)
    beam.y = np.zeros(nrays)
    a, c = rng.uniform(-1.9e-4, 1.9e-4, nrays), rng.uniform(-4.5e-5, 4.5e-5, nrays)
    beam.a, beam.c, beam.b = a, c, np.sqrt(1 - a**2 - c**2)
    beam.E = rng.uniform(8999., 9001., nrays)
    beam.state = np.ones(nrays, dtype=np.int32)
    beam.Jss, beam.Jpp, beam.Jsp = np.ones(nrays), np.zeros(nrays), np.zeros(nrays, complex)
    for f in beam.array_fields():
        beam.dev(f)
    optics = workloads.balder_optics()
    # three blocks of *runs* passes, the best block counts (one block of the round-6 profile run
    # held an 85-ms stall of the profiler's: all blocks are reported)
    blocks, reserved = [], [torch.cuda.memory_reserved() / 1e9]
    for _ in range(3):
        fresh = [rs.Beam(copyFrom=beam) for _ in range(runs + 1)]    # the chain marks its input
        image = workloads.balder_trace(optics, fresh[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(runs):
            image = workloads.balder_trace(optics, fresh[k + 1])
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / runs)
        reserved.append(torch.cuda.memory_reserved() / 1e9)
    sec = min(blocks)
    arrived = float((image.state_count(1) if hasattr(image, 'state_count')
                     else (image.state == 1).sum()) / nrays)
    # how many launches a pass of the chain is: the chain recorded into a HIP graph, its nodes
    # counted (hipGraphGetNodes)
    launches = None
    try:
        from xrt_amd import graphs
        spare = rs.Beam(copyFrom=beam)
        workloads.balder_trace(optics, rs.Beam(copyFrom=beam))
        torch.cuda.synchronize()

        def one_pass():
            img = workloads.balder_trace(optics, spare)
            rs.flush_pending()
            return img
        rec = graphs.IterationGraph(one_pass, keep_graph=True)
        launches = rec.kernel_nodes()
        rec.close()
        del rec, spare
    except Exception as e:      # noqa: BLE001  (the count is a report, not the measurement)
        launches = 'not counted: %s' % e
    # ... and with every beam of every element written, looked at or not (rounds 1-4)
    from xrt_amd.backends.raycing import oes as roe
    roe.fuseConsumers = not both
    try:
        fresh = [rs.Beam(copyFrom=beam) for _ in range(runs + 1)]
        workloads.balder_trace(optics, fresh[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(runs):
            workloads.balder_trace(optics, fresh[k + 1])
        torch.cuda.synchronize()
        sec_all = (time.perf_counter() - t0) / runs
    finally:
        roe.fuseConsumers = True
    del fresh
    return {'metric': 'Balder example beamline, mask -> sample, seconds per pass of the beam',
            'rays': nrays, 'seconds': sec, 'seconds_every_beam_written': sec_all,
            'seconds_by_block': blocks, 'reserved_gb_before_and_after_each_block': reserved,
            'launches_per_pass': launches,
            'in_one_pass': 'both faces of the filter are one kernel (reflect_fused_plate2), both '
                           'crystals are one kernel, the two slits behind the focusing mirror and '
                           'the sample screen ride in the tail of its pass (round 6)',
            'on_demand': 'the chain hands the global beam from element to element and shows the '
                         'two screens: the local beams of the mirrors and of the filter (308 -> '
                         '200 B per ray and surface), of the two crystals (416 -> 200 B per '
                         'ray) and the beams in the frames of the four '
                         'apertures (200 -> 52 B) are not written; each is made, by the same '
                         'kernels on the same input, the first time somebody looks at it',
            'rays_per_s': nrays / sec,
            'intersections_per_s': 6 * nrays / sec, 'surfaces': 6, 'apertures': 4,
            'screens': 2, 'fraction_at_sample': arrived,
            'note': 'device-resident beams through 12 elements; host glue of every element '
                    'call is inside the time'}


def bench_softimax(runs=3):
    """The reference's published wave benchmark, whole script body
    (tests/speed/3_Softi_CXIw2D_speed.py, BASELINE.md table: 17.5 s on 1 x A100,
    11.5 s on 2 x A100): undulator field on the front-end slit, ten Kirchhoff
    integrals (7 of 2e5 x 2e5, 3 of 2e5 x 4096) interleaved with wave sampling
    (prepare_wave) and reflect on toroid / plane / blazed grating / elliptical
    mirrors, nrays = 2e5, one repeat. Synthetic = the script's own beamline."""
    import types
    import xrt_amd.backends.raycing as raycing
    import xrt_amd.backends.raycing.sources as rs
    import xrt_amd.backends.raycing.apertures as ra
    import xrt_amd.backends.raycing.oes as roe
    import xrt_amd.backends.raycing.materials as rm
    import xrt_amd.backends.raycing.screens as rsc
    import xrt_amd.backends.raycing.waves as rw
    from xrt_amd.workloads import SoftiMAX
    mods = types.SimpleNamespace(raycing=raycing, rs=rs, ra=ra, roe=roe, rm=rm,
                                 rsc=rsc, rw=rw)
    np.random.seed(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scene = SoftiMAX(mods, nrays=200000)
    out = scene.run()                       # the first run, setup included: timed as it is
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    times, kernel = [], []
    for _ in range(runs):
        kms = [0.]
        orig = rw._kirchhoff_on_gpu

        def spy(*a, **k):
            r = orig(*a, **k)
            kms[0] += rw.lastKernelMs
            return r
        # the kernel seconds come from a run of their own (timing a kernel waits for the
        # stream); the run that is timed as a whole does not measure them
        rw._kirchhoff_on_gpu = spy
        rw.timeKernels = True
        try:
            scene.run()
        finally:
            rw._kirchhoff_on_gpu = orig
            rw.timeKernels = False
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = scene.run()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t1)
        kernel.append(kms[0] * 1e-3)
    best = min(times)
    pairs = 7 * 2e5 * 2e5 + 3 * 2e5 * 4096
    focus = out['beamFSMExp01']
    # the same script with waves.precision = 'relaxed' (opt-in): seconds, and the flux in the
    # focus against the exact run's
    rw.precision = 'relaxed'
    try:
        rtimes = []
        for _ in range(2):
            np.random.seed(1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rout = scene.run()
            torch.cuda.synchronize()
            rtimes.append(time.perf_counter() - t1)
        rfocus = rout['beamFSMExp01']
        rflux = float((rfocus.Jss + rfocus.Jpp).sum())
    finally:
        rw.precision = 'exact'
    np.random.seed(1)
    xfocus = scene.run()['beamFSMExp01']
    xflux = float((xfocus.Jss + xfocus.Jpp).sum())
    relaxed = dict(seconds=min(rtimes), focus_flux=rflux, focus_flux_exact_same_seed=xflux,
                   relative_flux_difference=abs(rflux - xflux) / abs(xflux))
    return dict(
        relaxed=relaxed,
        metric='SoftiMAX wave chain (reference speed test 3_Softi_CXIw2D), seconds '
               'per run', seconds=best, seconds_first_run_incl_setup=first,
        kirchhoff_kernel_seconds=min(kernel),
        pairs=pairs, higher_is_better=False, nrays=200000,
        reference_published_seconds={'1xA100': 17.5, '2xA100': 11.5, '1xP100': 53.0},
        speedup_vs_published_1xA100=17.5 / first,
        speedup_note='published whole-script seconds / seconds_first_run_incl_setup (the '
                     'comparable figure: one cold run of the script body); steady state '
                     '(`seconds`) is %.0f x' % (17.5 / best),
        focus_flux=float((focus.Jss + focus.Jpp).sum()),
        note='host glue (numpy sampling / frame changes in the reference\'s RNG '
             'order) is inside the time; parity of this chain vs the reference: '
             'tests/test_gpu_softimax.py (golden G11)')


def cpu_baseline_kirchhoff(host, npix=256):
    from oracle import kirchhoff_np as kn
    idx = np.linspace(0, host['px'].size - 1, npix).astype(int)
    t0 = time.perf_counter()
    kn.kirchhoff_conv(host['px'][idx], host['py'][idx], host['pz'][idx],
                      host['sx'], host['sy'], host['sz'], [0, 1, 0], host['nl'],
                      host['E'], host['Es'], host['Ep'])
    dt = time.perf_counter() - t0
    ns = host['sx'].size
    res = dict(value=npix * ns / dt, unit='pairs/s', cores=1, kind='port', sampled=True,
               sample='a SAMPLE of the workload: %d of its pixels x all %d samples of cfg4 '
                      'through oracle/kirchhoff_np.py (numpy, 1 thread), %.1f s; the '
                      'integral is linear in pixels' % (npix, ns, dt))
    res.update(host_cpu())
    try:        # all host cores: the C/OpenMP restatement
        from oracle import kirchhoff_c as kc
        from oracle.consts import CHBAR
        threads = kc.max_threads()
        npc = int(min(host['px'].size, max(256, 32 * threads)))
        idx = np.linspace(0, host['px'].size - 1, npc).astype(int)
        k = host['E'] / CHBAR * 1e7
        args = (host['px'][idx], host['py'][idx], host['pz'][idx], host['sx'],
                host['sy'], host['sz'], [0., 1., 0.], host['nl'], k, host['Es'],
                host['Ep'])
        kc.kirchhoff(*[a[:64] if i < 3 else a for i, a in enumerate(args)])  # warm up
        t0 = time.perf_counter()
        kc.kirchhoff(*args)
        dtc = time.perf_counter() - t0
        res['all_cores'] = dict(
            value=npc * ns / dtc, unit='pairs/s', cores=threads, kind='port',
            sample='%d pixels x %d samples through oracle/kirchhoff_c.c '
                   '(gcc -O2 -fopenmp, %d threads), %.1f s' % (npc, ns, threads, dtc))
    except Exception as e:  # noqa: BLE001  (no compiler on the box: keep the numpy figure)
        res['all_cores'] = dict(error=str(e))
    return res


TRAFFIC_SOURCE = ('profiles/hbm_traffic.json: PMC passes of tools/refresh_profiles.sh on the '
                  'same command, committed -- static, not collected in this run')


def load_hist_traffic(bins):
    """HBM bytes per plot of the histogram kernels from the committed PMC summary, or None."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'hist_traffic.json')) as f:
            return json.load(f).get('bins%d' % bins, {}).get('hbm_bytes_per_plot')
    except Exception:  # noqa: BLE001
        return None


def load_plot_tail_traffic():
    """HBM bytes of one e2e iteration with the plot in the tail of the pass (PMC summary)."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'plot_tail_traffic.json')) as f:
            return json.load(f)['tail']['hbm_bytes_per_iteration']
    except Exception:  # noqa: BLE001
        return None


def load_traffic(kernel, algorithmic=None):
    """HBM bytes per launch from the committed PMC summary (profiles/), or None. *algorithmic*:
    the bytes the timed launch moves by its contract -- the counter figure is only reported for
    a launch of the SAME shape (a committed figure once came from the 200-B form of the pass
    while the timed one was the 308-B form): one that lies below the
    algorithmic bytes or far above them belongs to another launch and is left out."""
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    try:
        with open(path) as f:
            got = json.load(f).get(kernel, {}).get('hbm_bytes_per_launch')
    except Exception:  # noqa: BLE001
        return None
    if got is not None and algorithmic is not None and \
            not 0.97 * algorithmic <= got <= 1.5 * algorithmic:
        return None
    return got


def claim_stdout():
    """The contract is ONE JSON line on stdout. Libraries write there too (gloo announces its
    connections on fd 1 when the CPU-side group is made): everything but the line goes to
    stderr from here on, the line itself to the original stdout (returned as a file)."""
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    return line_out


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)
    line_out = claim_stdout()
    if args.dry_run:
        return dry_run(args, line_out)
    global DRY_RANKS
    DRY_RANKS = bool(args.dry_ranks)
    if DRY_RANKS:       # the choreography and the line only: every single-GPU leg is skipped
        args.skip_undulator = args.skip_softimax = args.skip_balder = args.skip_e2e = True
        args.skip_cpu_baseline = args.skip_dcm = True
    else:
        from xrt_amd import _lib
        _lib.require_gpu()                    # no CPU fallback: fail loudly
    world, rank, local, dist = setup_dist(args)
    self_check = multi_gpu_self_check(world, rank, dist) \
        if world > 1 and dist is not None and not DRY_RANKS else None
    main_res = bench_reflect(args, world, rank, dist)
    line = dict(
        # BASELINE.json's metric, verbatim; `value` is its first quantity, the second one
        # is the "kirchhoff" object of the same line
        metric='ray-surface intersections/sec/GPU; Kirchhoff sample\u00b7pixel '
               'pairs/sec at 1/2/4/8 GPUs',
        metric_note='value = ray-surface intersections/s over all GPUs (cfg2); the Kirchhoff '
                    'pairs/s of the same run are in "kirchhoff"',
        value=main_res['value'], unit='intersections/s', n_gpus=world,
        steps=args.steps, warmup=args.warmup,
        ms_per_step=main_res['ms_per_step'], higher_is_better=True,
        scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
        config=dict(
            workload='cfg2: %d rays/GPU -> ToroidMirror(Pt, p=20 m, q=10 m, '
                     '4 mrad), one OE.reflect per step' % int(args.rays),
            rays_per_gpu=int(args.rays), entering=main_res['n_enter'],
            good_fraction=main_res['good_fraction'],
            parallelism='replicas x%d (P1 does not shard, SURVEY 8e)' % world),
        per_gpu=main_res['value'] / world, pass_ms=main_res.get('pass_ms'),
        kernel_ms=main_res.get('kernel_ms'), roofline=main_res.get('roofline'))
    if args.with_dcm or (world == 1 and not args.skip_dcm):
        d = bench_reflect(args, world, rank, dist, dcm=True)
        dcm_bytes = 416 * d['n_enter']       # 100 B in, 3 x 100 B + 2 x 8 B out
        dcm_s = d['ms_per_step'] * 1e-3
        line['dcm'] = dict(metric='ray-surface intersections/s (cfg3 DCM Si111)',
                           value=d['value'], ms_per_step=d['ms_per_step'],
                           good_fraction=d['good_fraction'],
                           roofline=dict(
                               bound='hbm', kernel='reflect_fused_dcm + its two small '
                                                   'launches (whole double_reflect pass, '
                                                   'host clock over the steps)',
                               achieved=dcm_bytes / dcm_s / 1e9, peak=HBM_PEAK / 1e9,
                               unit='GB/s', frac=dcm_bytes / dcm_s / HBM_PEAK,
                               note='208 B per intersection (416 B per ray)',
                               traffic=load_traffic('reflect_fused_dcm', 416. * args.rays)
                               if int(args.rays) == 10_000_000 else None,
                               traffic_source=TRAFFIC_SOURCE))
    host = None
    if not args.skip_kirchhoff:
        cfgs = [args.kirchhoff_config] if args.kirchhoff_config else \
            ([4, 5] if world == 8 else [4])
        for cfg in cfgs:
            ks = args.kirchhoff_steps or (max(1, min(args.steps, 3)) if cfg == 4 else 1)
            kw = 1
            kres, h = bench_kirchhoff(cfg, ks, kw, world, rank, dist)
            if cfg == 4 or host is None:
                host = h
            line['kirchhoff' if 'kirchhoff' not in line else 'kirchhoff_cfg%d' % cfg] = kres
    if world == 1 and not args.skip_kirchhoff and not DRY_RANKS:
        line['kirchhoff_general'] = bench_kirchhoff_general()
    if world == 1 and not args.skip_undulator:
        line['undulator'] = bench_undulator(not args.skip_cpu_baseline)
    if world == 1 and not args.skip_softimax:
        line['softimax'] = bench_softimax()
    if world == 1 and not args.skip_balder:
        line['balder'] = bench_balder(int(args.rays))
        line['hist'] = bench_hist(int(args.rays))
    if world == 1 and not args.skip_dcm:
        line['reflect_nolocal'] = bench_reflect_nolocal(int(args.rays))
    if world == 1 and not args.skip_dcm:
        line['reflect_figure'] = bench_reflect_figure(int(args.rays))
    if world == 1 and not args.skip_e2e:
        line['e2e'] = bench_e2e(int(args.rays))
    if world == 1 and not args.skip_dcm:
        line['multiple_reflect'] = bench_multiple_reflect(int(args.rays),
                                                          cpu=not args.skip_cpu_baseline)
    if args.with_softi_shapes and world == 1:
        line['softi_shapes'] = bench_softi_shapes()
    if world == 1 and rank == 0 and not args.skip_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline_reflect()
        line['cpu_baseline']['host_cpus'] = os.cpu_count()
        if host is not None and 'kirchhoff' in line:
            line['kirchhoff']['cpu_baseline'] = cpu_baseline_kirchhoff(host)
    if DRY_RANKS:
        line['dry_ranks'] = True
    if self_check is not None:
        line['multi_gpu_self_check'] = self_check
        line['roofline'] = line.get('roofline') or {}
        line['roofline']['multi_gpu_self_check'] = 1 if self_check.get('ok') else 0
    compact_for_the_record(line, world)          # (writes `summary`, the LAST key of the line)
    if rank == 0:
        line_out.write(json.dumps(line) + '\n')
        line_out.flush()
    if dist is not None:
        dist.destroy_process_group()


def _dig(d, *path):
    for key in path:
        if not isinstance(d, dict) or key not in d:
            return None
        d = d[key]
    return d


def compact_for_the_record(line, world):
    """The driver's record of a run keeps the SCALAR members of `roofline`, `config` and
    `cpu_baseline`, the names of every other key, and the last 2 KB of stdout (VERDICT r5 weak
    #2: nested objects are dropped). So the second half of BASELINE.json's metric -- Kirchhoff
    pairs/s -- and one number per other leg are repeated as FLAT keys of `roofline` /
    `cpu_baseline`, and once more in a short `summary` object that is the LAST key of the line.
    Checks who was there before anything is printed."""
    roof = line.setdefault('roofline', {}) or {}
    line['roofline'] = roof
    base_flat = line.get('cpu_baseline') if isinstance(line.get('cpu_baseline'), dict) else None
    summary = dict(intersections_per_s=line.get('value'), ms_per_step=line.get('ms_per_step'),
                   reflect_frac=roof.get('frac'), reflect_kernel_ms=line.get('kernel_ms'),
                   reflect_traffic_over_algorithmic=(
                       roof['traffic'] / (BYTES_PER_INTERSECTION * line['config']['entering'])
                       if roof.get('traffic') and _dig(line, 'config', 'entering') else None),
                   n_gpus=world)
    for key in ('kirchhoff', 'kirchhoff_cfg5'):
        k = line.get(key)
        if not k:
            continue
        # a multi-rank figure counts only if every rank took part and timed its own tile
        assert k['rccl_ranks'] == world, (key, k['rccl_ranks'], world)
        assert len(k['kernel_ms_by_rank']) == world and min(k['kernel_ms_by_rank']) > 0., \
            (key, k['kernel_ms_by_rank'])
        name = 'kirchhoff_cfg%d' % (5 if 'cfg5' in k['config']['workload'] else 4)
        flat = dict(pairs_per_s=k['value'], ms_per_step=k['ms_per_step'],
                    frac=k['roofline']['frac'], kernel_ms=k['kernel_ms'],
                    rccl_ranks=k['rccl_ranks'], n_gpus=k['n_gpus'])
        for kk, v in flat.items():
            roof['%s_%s' % (name, kk)] = v
            summary['%s_%s' % (name, kk)] = v
        roof[name + '_scaling'] = k['scaling']
        roof[name + '_peak_tflops'] = k['roofline']['peak']
        if _dig(k, 'in_process', 'value') is not None:
            roof[name + '_in_process_pairs_per_s'] = k['in_process']['value']
        base = k.get('cpu_baseline')
        if base and base_flat is not None:
            base_flat[name + '_numpy_pairs_per_s'] = base.get('value')
            base_flat[name + '_numpy_cores'] = base.get('cores')
            base_flat[name + '_numpy_sample'] = base.get('sample')
            summary[name + '_numpy_pairs_per_s'] = base.get('value')
            if 'all_cores' in base:
                base_flat[name + '_openmp_pairs_per_s'] = base['all_cores'].get('value')
                base_flat[name + '_openmp_cores'] = base['all_cores'].get('cores')
                summary[name + '_openmp_pairs_per_s'] = base['all_cores'].get('value')
    if base_flat is not None:
        omp = _dig(base_flat, 'all_cores', 'value')
        if omp is not None:
            base_flat['openmp_intersections_per_s'] = omp
            base_flat['openmp_cores'] = _dig(base_flat, 'all_cores', 'cores')
        summary['numpy_intersections_per_s'] = base_flat.get('value')
        summary['openmp_intersections_per_s'] = omp
    flat_legs = (
        ('dcm_frac', ('dcm', 'roofline', 'frac')),
        ('dcm_ms_per_step', ('dcm', 'ms_per_step')),
        ('dcm_intersections_per_s', ('dcm', 'value')),
        ('dcm_traffic', ('dcm', 'roofline', 'traffic')),
        ('kirchhoff_general_frac', ('kirchhoff_general', 'roofline', 'frac')),
        ('kirchhoff_general_kernel_ms', ('kirchhoff_general', 'kernel_ms')),
        ('kirchhoff_general_relaxed_frac', ('kirchhoff_general', 'relaxed', 'frac')),
        ('und_imap_frac', ('undulator', 'roofline', 'frac')),
        ('und_imap_ms', ('undulator', 'ms')),
        ('hist_frac', ('hist', 'roofline', 'frac')),
        ('hist_ms_per_plot', ('hist', 'ms_per_plot')),
        ('hist_traffic', ('hist', 'roofline', 'traffic')),
        ('nolocal_frac', ('reflect_nolocal', 'roofline', 'frac')),
        ('nolocal_ms_per_step', ('reflect_nolocal', 'ms_per_step')),
        ('nolocal_traffic', ('reflect_nolocal', 'roofline', 'traffic')),
        ('softimax_seconds', ('softimax', 'seconds')),
        ('softimax_first_run_seconds', ('softimax', 'seconds_first_run_incl_setup')),
        ('balder_ms', ('balder', 'seconds')),
        ('balder_every_beam_written_ms', ('balder', 'seconds_every_beam_written')),
        ('balder_launches', ('balder', 'launches_per_pass')),
        ('e2e_ms_per_iteration', ('e2e', 'ms_per_iteration')),
        ('e2e_first_block_ms', ('e2e', 'ms_per_iteration_first_block')),
        ('e2e_plot_adds_ms', ('e2e', 'plot_adds_ms')),
        ('e2e_plot_as_own_launches_ms', ('e2e', 'ms_per_iteration_plot_as_own_launches')),
        ('e2e_traffic', ('e2e', 'roofline', 'traffic')),
        ('e2e_every_beam_written_ms', ('e2e', 'ms_per_iteration_every_beam_written')),
        ('e2e_1e5_eager_ms', ('e2e', 'small_beams', '100000_rays', 'eager_ms_per_iteration')),
        ('e2e_1e5_graph_ms', ('e2e', 'small_beams', '100000_rays', 'graph_ms_per_iteration')),
        ('e2e_1e5_speedup', ('e2e', 'small_beams', '100000_rays', 'speedup')),
        ('e2e_1e5_replay_ms', ('e2e', 'small_beams', '100000_rays', 'graph_choice', 'replay_ms')),
        ('e2e_1e6_eager_ms', ('e2e', 'small_beams', '1000000_rays', 'eager_ms_per_iteration')),
        ('e2e_1e6_graph_ms', ('e2e', 'small_beams', '1000000_rays', 'graph_ms_per_iteration')),
        ('e2e_1e6_speedup', ('e2e', 'small_beams', '1000000_rays', 'speedup')),
        ('multiple_reflect_ms_per_bounce', ('multiple_reflect', 'ms_per_bounce')),
        ('multiple_reflect_intersections_per_s', ('multiple_reflect', 'value')),
        ('multiple_reflect_bounces', ('multiple_reflect', 'bounces')),
        ('multiple_reflect_numpy_intersections_per_s',
         ('multiple_reflect', 'cpu_baseline', 'value')),
    )
    in_summary = ('dcm_frac', 'dcm_ms_per_step', 'kirchhoff_general_frac',
                  'kirchhoff_general_relaxed_frac', 'und_imap_frac', 'hist_frac', 'hist_ms_per_plot',
                  'hist_traffic', 'nolocal_ms_per_step', 'softimax_seconds', 'balder_ms',
                  'e2e_ms_per_iteration', 'e2e_first_block_ms', 'e2e_plot_adds_ms', 'e2e_plot_as_own_launches_ms',
                  'e2e_traffic', 'e2e_every_beam_written_ms', 'e2e_1e5_eager_ms',
                  'e2e_1e5_graph_ms', 'e2e_1e5_replay_ms', 'e2e_1e5_speedup', 'e2e_1e6_speedup',
                  'multiple_reflect_ms_per_bounce', 'multiple_reflect_intersections_per_s')
    for name, path in flat_legs:
        v = _dig(line, *path)
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            roof[name] = v
            if name in in_summary:
                summary[name] = v
    # (balder 'seconds' is per pass of the chain: report it in ms under its flat name)
    for name in ('balder_ms', 'balder_every_beam_written_ms'):
        if name in roof:
            roof[name] = roof[name] * 1e3
            if name in summary:
                summary[name] = roof[name]
    # the short form at the very end of the line: the driver's tail of stdout holds it whole
    text = json.dumps({k: (float('%.5g' % v) if isinstance(v, float) else v)
                       for k, v in summary.items() if v is not None})
    while len(text) > 1500:                   # (never: ~45 numbers; drop from the back if so)
        summary.popitem()
        text = json.dumps({k: (float('%.5g' % v) if isinstance(v, float) else v)
                           for k, v in summary.items() if v is not None})
    line.pop('summary', None)
    line['summary'] = json.loads(text)


if __name__ == '__main__':
    main()
