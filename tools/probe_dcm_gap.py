"""Where do the ~50 us between the kernels' sum and a DCM step go? Host time of every call and
the step time with 1e7 rays: PYTHONPATH=. python tools/probe_dcm_gap.py"""
import time
import numpy as np
import torch
from xrt_amd import workloads

n = 10_000_000
dcm = workloads.cfg3_dcm()
b3 = workloads.synthetic_rays(n, 43, sa=1e-4, E=(8995., 9005.))
for f in b3.array_fields():
    b3.dev(f)
for _ in range(5):
    out = dcm.double_reflect(b3)
torch.cuda.synchronize()
for label, keep in (('outputs dropped every step (bench)', False), ('outputs kept alive', True)):
    held = []
    host = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        t = time.perf_counter()
        out = dcm.double_reflect(b3)
        host.append(time.perf_counter() - t)
        if keep and len(held) < 6:
            held.append(out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print('%-38s step %.1f us; host per call median %.0f us, max %.0f us' % (
        label, dt * 1e6, np.median(host) * 1e6, max(host) * 1e6))
tm = {}
dcm.double_reflect(b3, _timing=tm)
print('events: pass %.1f us, fused kernel %.1f us' % (tm['pass_ms'] * 1e3, tm['kernel_ms'] * 1e3))
