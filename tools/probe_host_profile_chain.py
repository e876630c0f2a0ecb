"""cProfile of the host side of the Balder chain (12 elements) on a tiny beam: what an eager
iteration of a real beamline costs in Python + ctypes.
    PYTHONPATH=. python tools/probe_host_profile_chain.py [passes]"""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

from xrt_amd import workloads
import xrt_amd.backends.raycing.sources as rs

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = 2000
rng = np.random.default_rng(17)
beam = rs.Beam(nrays=n)
beam.x, beam.z = rng.normal(0, 0.05, n), rng.normal(0, 0.01, n)
a, c = rng.uniform(-1.9e-4, 1.9e-4, n), rng.uniform(-4.5e-5, 4.5e-5, n)
beam.a, beam.c, beam.b = a, c, np.sqrt(1 - a**2 - c**2)
beam.E = rng.uniform(8999., 9001., n)
for f in beam.array_fields():
    beam.dev(f)
optics = workloads.balder_optics()
fresh = [rs.Beam(copyFrom=beam) for _ in range(2 * reps + 20)]
for b in fresh:
    for f in b.array_fields():
        b.dev(f)
k = 0
for _ in range(20):
    workloads.balder_trace(optics, fresh[k]); k += 1
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    workloads.balder_trace(optics, fresh[k]); k += 1
t1 = time.perf_counter()
torch.cuda.synchronize()
print('%.1f us per pass of the chain on the host' % ((t1 - t0) / reps * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    workloads.balder_trace(optics, fresh[k]); k += 1
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(30)
st.sort_stats('cumulative').print_stats(30)
