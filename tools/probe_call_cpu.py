"""CPU time of one OE.reflect / DCM.double_reflect call (enqueue only) on the GPU box:
PYTHONPATH=. python tools/probe_call_cpu.py"""
import time
import numpy as np
import torch
from xrt_amd import workloads as pc

for name, oe, beam, op in (
        ('reflect', pc.cfg2_toroid(), pc.synthetic_rays(10_000_000, 42), 'reflect'),
        ('dcm', pc.cfg3_dcm(), pc.synthetic_rays(10_000_000, 43, sa=1e-4, E=(8995., 9005.)),
         'double_reflect')):
    for f in beam.array_fields():
        beam.dev(f)
    fn = getattr(oe, op)
    kw = {}
    out = fn(beam)
    if op == 'reflect':
        kw['out'] = out
    torch.cuda.synchronize()
    ts = []
    for k in range(40):
        if k % 8 == 0:
            torch.cuda.synchronize()       # empty queue: pure enqueue cost for the next calls
        t0 = time.perf_counter()
        out = fn(beam, **kw)
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    ts = np.array(ts) * 1e6
    print(name, 'cpu us per call: median %.0f min %.0f max %.0f' % (np.median(ts), ts.min(), ts.max()))
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        out = fn(beam, **kw)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
