"""Where does a wave of the lean fused kernel spend its life?  Runs cfg2 on a library built with
-DXRT_PROBE_TIMING (tools/build_variant.sh timing -DXRT_PROBE_TIMING) and prints the mean
shader-clock ticks per wave and section:
    XRT_HIP_LIBRARY=xrt_amd/ab/libxrt_timing.so python tools/probe_sections.py [rays]"""
import ctypes
import os
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from xrt_amd import _lib, workloads  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
oe = workloads.cfg2_toroid()
beam = workloads.synthetic_rays(n, 42)
for f in beam.array_fields():
    beam.dev(f)
lib = ctypes.CDLL(os.environ['XRT_HIP_LIBRARY'])
waves = (n + 63) // 64
buf = np.zeros((waves, 8), dtype=np.uint64)
for _ in range(3):
    oe.reflect(beam)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True)
t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10):
    oe.reflect(beam)
t1.record()
torch.cuda.synchronize()
assert lib.xrt_probe_ticks(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(waves)) == 0
v = buf[:, :5].astype(np.float64)
names = ['entry -> record loaded + local frame', 'root solve (+ report)', 'finish (normal, amplitudes, J)',
         'stores issued', 'stores acknowledged']
print('waves %d, pass %.3f ms' % (waves, t0.elapsed_time(t1) / 10))
tot = v.sum(axis=1).mean()
for k, nm in enumerate(names):
    print('%-40s mean %8.0f  median %8.0f ticks  %5.1f %%' % (nm, v[:, k].mean(), np.median(v[:, k]), 100 * v[:, k].mean() / tot))
start = buf[:, 7].astype(np.float64)
span = (start + v.sum(axis=1)).max() - start.min()
print('%-40s %8.0f ticks; launch span %.0f ticks' % ('wave lifetime', tot, span))
