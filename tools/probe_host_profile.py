"""cProfile of the host side of one eager e2e iteration (device source -> mirror -> screen -> plot)
on a tiny beam: where the ~0.2 ms per iteration of Python + ctypes go.
    PYTHONPATH=. python tools/probe_host_profile.py [iterations]"""
import cProfile
import pstats
import sys
import time

import torch

from xrt_amd import runner, workloads
from xrt_amd.backends.raycing import sources as rs

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
bl, run_process, make_plot = workloads.e2e_beamline(2000)
plot = make_plot()


def iteration():
    beams = run_process(bl)
    runner.accumulate_plot(plot, beams)
    rs.flush_pending()


for _ in range(20):
    iteration()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    iteration()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('%.1f us per iteration on the host' % ((t1 - t0) / reps * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    iteration()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumulative').print_stats(22)
