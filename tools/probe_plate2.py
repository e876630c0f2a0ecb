"""Both faces of the Balder chain's diamond filter in one pass (reflect_fused_plate2), 1e7 rays."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from xrt_amd import workloads
import xrt_amd.backends.raycing.sources as rs

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
b = workloads.balder_optics()
rng = np.random.default_rng(17)
beam = rs.Beam(nrays=n)
beam.x, beam.z = rng.normal(0, 0.05, n), rng.normal(0, 0.01, n)
beam.y = np.zeros(n)
a, c = rng.uniform(-1.9e-4, 1.9e-4, n), rng.uniform(-4.5e-5, 4.5e-5, n)
beam.a, beam.c, beam.b = a, c, np.sqrt(1 - a**2 - c**2)
beam.E = rng.uniform(8999., 9001., n)
beam.state = np.ones(n, dtype=np.int32)
beam.Jss, beam.Jpp, beam.Jsp = np.ones(n), np.zeros(n), np.zeros(n, complex)
b.mask.propagate(beam)
# (Plate.double_refract hands its beams out before it launches -- oes._DeferredDouble: the launch
# is made when the next element would take the global beam; here, by flushing what is pending)
for _ in range(3):
    g = b.filter1.double_refract(beam)[0]
    rs.flush_pending()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g = b.filter1.double_refract(beam)[0]
    rs.flush_pending()
torch.cuda.synchronize()
print('[%s] double_refract %.3f ms' % (os.environ.get('XRT_HIP_LIBRARY', ''),
                                       (time.perf_counter() - t0) * 50))
