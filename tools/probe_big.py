"""One reflect pass on 1e8 rays (10 GB per beam) checked on slices against the oracle:
PYTHONPATH=. python tools/probe_big.py [nrays]"""
import sys
import time
import numpy as np
import torch
from xrt_amd import workloads
import xrt_amd.backends.raycing.sources as rs
from oracle import reflect_np as rn
from oracle.adapters import oracle_params, to_oracle_beam

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
t0 = time.time()
beam = workloads.synthetic_rays(n, 7)
print('generated %.0f s' % (time.time() - t0), flush=True)
oe = workloads.cfg2_toroid()
for f in beam.array_fields():
    beam.dev(f)
torch.cuda.synchronize()
t = {}
gb, lb = oe.reflect(beam, _timing=t)
torch.cuda.synchronize()
print('pass %.2f ms, kernel %.2f ms, exact sequence %s -> %.3e rays/s' % (
    t['pass_ms'], t['kernel_ms'], t['exact_sequence'], n / t['pass_ms'] * 1e3), flush=True)
m = 20000
for lo in (0, n // 2, n - m):
    sub = rs.Beam(nrays=m)
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp', 'state'):
        getattr(sub, f)[:] = beam.peek(f)[lo:lo + m]
    if lo:   # the batch decisions hinge on ray 0: keep it in front of the slice
        for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp', 'state'):
            getattr(sub, f)[0] = beam.peek(f)[0]
    ogb, olb = rn.oe_reflect(oracle_params(oe), to_oracle_beam(sub))
    s = slice(lo + (1 if lo else 0), lo + m)
    o = slice(1 if lo else 0, m)
    assert np.array_equal(lb.peek('state')[s], olb.state[o])
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path'):
        d = np.abs(gb.peek(f)[s] - getattr(ogb, f)[o]).max()
        assert d <= 1e-12 * max(np.abs(getattr(ogb, f)[o]).max(), 1e-300), (f, d)
    print('slice at %d ok' % lo, flush=True)
