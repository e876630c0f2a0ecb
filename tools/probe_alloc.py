"""Device memory held by the e2e loops of bench.py (plot in the tail of the pass / plot as launches of its
own / no plot), and their ms per iteration: torch allocator statistics after each loop. Round 6: the records of
deferred passes refer to their beams weakly -- 4.5 GB reserved and no device allocation after the first loop, where
reference cycles had kept 15-20 GB of dropped beams alive between runs of Python's cycle collector (33 GB reserved).
    python tools/probe_alloc.py"""
import os, sys, time, torch
sys.path.insert(0, '.')
from xrt_amd import workloads, runner
from xrt_amd.backends.raycing import run as rr, sources as rs
bl, run_process, make_plot = workloads.e2e_beamline(10_000_000)
rr.run_process = run_process
def stats(tag):
    s = torch.cuda.memory_stats()
    print(tag, 'device_alloc', s.get('num_device_alloc'), 'device_free', s.get('num_device_free'), 'retries', s.get('num_alloc_retries'),
          'reserved GB %.2f' % (torch.cuda.memory_reserved() / 1e9), 'allocated GB %.2f' % (torch.cuda.memory_allocated() / 1e9), flush=True)
def timed(tag, fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); print('%-28s %.3f ms' % (tag, (time.perf_counter() - t0) / reps * 1e3)); stats(tag)
plot = make_plot()
timed('tail', lambda: runner.run_ray_tracing([plot], repeats=1, beamLine=bl))
os.environ['XRT_PLOT_TAIL_OFF'] = '1'
timed('plot as own launches', lambda: runner.run_ray_tracing([plot], repeats=1, beamLine=bl))
del os.environ['XRT_PLOT_TAIL_OFF']
def no_plot():
    beams = run_process(bl); beams['focus'].nrays; rs.flush_pending()
timed('no plot', no_plot)
timed('tail again', lambda: runner.run_ray_tracing([plot], repeats=1, beamLine=bl))
os.environ['XRT_PLOT_TAIL_OFF'] = '1'
timed('own launches again', lambda: runner.run_ray_tracing([plot], repeats=1, beamLine=bl))
