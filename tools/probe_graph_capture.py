"""Feasibility: one run_ray_tracing iteration (source -> mirror -> screen -> plot) captured in a
HIP graph through torch.cuda.graph and replayed: does the capture take the ctypes launches, the
histogram's stream-ordered scratch, and what does a replay cost against the eager iteration?
    PYTHONPATH=. python tools/probe_graph_capture.py"""
import sys
import time

import torch

from xrt_amd import workloads, runner


def clock(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for n in (2000, 100000, 1000000):
    bl, run_process, make_plot = workloads.e2e_beamline(n)
    plot = make_plot()

    def iteration():
        beams = run_process(bl)
        runner.accumulate_plot(plot, beams)
        return beams

    for _ in range(3):
        iteration()
    eager = clock(iteration, 100)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):       # warm-up on the capture stream (workspaces per stream)
        iteration()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=side):
            kept = iteration()
    except Exception as e:      # noqa: BLE001
        print('n = %d: capture failed: %r' % (n, e))
        sys.exit(1)
    before = float(plot.device_accumulator(torch.device('cuda', 0)).sum())
    g.replay()
    torch.cuda.synchronize()
    after = float(plot.device_accumulator(torch.device('cuda', 0)).sum())
    replay = clock(g.replay, 100)
    print('n = %8d rays: eager %7.1f us per iteration, graph replay %7.1f us; '
          'accumulator grew by %.6g on the first replay' % (n, eager, replay, after - before))
