"""Host time of one OE.reflect / DCM.double_reflect call (launch only, no sync):
PYTHONPATH=. python tools/probe_call_overhead.py"""
import time
import torch
from xrt_amd import workloads

n = 10_000_000
oe = workloads.cfg2_toroid()
beam = workloads.synthetic_rays(n, 42)
for f in beam.array_fields():
    beam.dev(f)
out = oe.reflect(beam)
out = oe.reflect(beam, out=out)
torch.cuda.synchronize()
for label, fn in (('reflect(out=)', lambda: oe.reflect(beam, out=out)),
                  ('reflect()', lambda: oe.reflect(beam))):
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    ts.sort()
    print('%-14s host %.0f us median, %.0f us min' % (label, ts[15] * 1e6, ts[0] * 1e6))
dcm = workloads.cfg3_dcm()
b3 = workloads.synthetic_rays(n, 43, sa=1e-4, E=(8995., 9005.))
dcm.double_reflect(b3)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    dcm.double_reflect(b3)
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
ts.sort()
print('%-14s host %.0f us median, %.0f us min' % ('double_reflect', ts[10] * 1e6, ts[0] * 1e6))
