"""run_ray_tracing of the e2e scene (device source -> cfg2 toroid -> screen -> 256^2 XYCPlot) at
several beam sizes: ms per iteration, eager and graph=True, with the options named in the
environment (XRT_HIP_NO_FUSE=1: no screen in the tail of the pass; XRT_HIP_HIST_NO_SMALL=1: the
three-kernel histogram route also for small beams).
    python tools/probe_e2e_sizes.py [iterations [rays ...]]"""
import sys
import time

import torch

from xrt_amd import runner, workloads
from xrt_amd.backends.raycing import run as rr

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sizes = [int(float(a)) for a in sys.argv[2:]] or [2000, 100000, 1000000]
for n in sizes:
    bl, run_process, make_plot = workloads.e2e_beamline(n)
    rr.run_process = run_process
    row = []
    for graph in (False, True):
        runner.run_ray_tracing([make_plot()], repeats=3, beamLine=bl, graph=graph)
        torch.cuda.synchronize()
        plot = make_plot()
        t0 = time.perf_counter()
        runner.run_ray_tracing([plot], repeats=reps, beamLine=bl, graph=graph)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / reps * 1e3)
        choice = getattr(plot, 'graphChoice', None)
    print('%8d rays: eager %.4f ms  graph %.4f ms per iteration  %s' % (n, row[0], row[1], choice))
