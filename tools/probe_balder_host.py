"""Host time of every element call of the Balder chain (launches are asynchronous: a call
that takes long on the host either does real host work or waits for the GPU)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from xrt_amd import workloads                      # noqa: E402
import xrt_amd.backends.raycing.sources as rs      # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(17)
beam = rs.Beam(nrays=n)
beam.x, beam.z = rng.normal(0, 0.05, n), rng.normal(0, 0.01, n)
a, c = rng.uniform(-1.9e-4, 1.9e-4, n), rng.uniform(-4.5e-5, 4.5e-5, n)
beam.a, beam.c, beam.b = a, c, np.sqrt(1 - a**2 - c**2)
beam.E = rng.uniform(8999., 9001., n)
beam.state = np.ones(n, dtype=np.int32)
beam.Jss, beam.Jpp, beam.Jsp = np.ones(n), np.zeros(n), np.zeros(n, complex)
for f in beam.array_fields():
    beam.dev(f)
b = workloads.balder_optics()
for rep in range(3):
    src = rs.Beam(copyFrom=beam)
    torch.cuda.synchronize()
    marks = []
    t00 = time.perf_counter()

    def lap(name, t0):
        marks.append((name, (time.perf_counter() - t0) * 1e3))
    t = time.perf_counter(); b.fsm0.expose(src); lap('fsm0.expose', t)
    t = time.perf_counter(); b.mask.propagate(src); lap('mask.propagate', t)
    t = time.perf_counter(); f1 = b.filter1.double_refract(src)[0]; lap('filter.double_refract', t)
    t = time.perf_counter(); v = b.vcm.reflect(f1)[0]; lap('vcm.reflect', t)
    t = time.perf_counter(); d = b.dcm.double_reflect(v)[0]; lap('dcm.double_reflect', t)
    t = time.perf_counter(); b.slitDCM.propagate(d); lap('slitDCM.propagate', t)
    t = time.perf_counter(); m = b.vfm.reflect(d)[0]; lap('vfm.reflect', t)
    t = time.perf_counter(); b.slitVFM.propagate(m); lap('slitVFM.propagate', t)
    t = time.perf_counter(); b.slitEH.propagate(m); lap('slitEH.propagate', t)
    t = time.perf_counter(); img = b.sample.expose(m); lap('sample.expose', t)
    host = (time.perf_counter() - t00) * 1e3
    torch.cuda.synchronize()
    total = (time.perf_counter() - t00) * 1e3
    if rep == 2:
        for name, ms in marks:
            print('%-24s %7.3f ms host' % (name, ms))
        print('host %.3f ms, with the GPU drained %.3f ms' % (host, total))

# which element calls copy arrays? (count torch-level copies per call)
import collections
counts = collections.Counter()
current = ['?']
for meth in ('clone', 'copy_', 'to', 'cpu', 'contiguous', '__getitem__', '__setitem__'):
    orig = getattr(torch.Tensor, meth)

    def wrap(self, *a, _o=orig, _m=meth, **k):
        if self.is_cuda and self.numel() > 1000000:
            counts[(current[0], _m)] += 1
        return _o(self, *a, **k)
    setattr(torch.Tensor, meth, wrap)
src = rs.Beam(copyFrom=beam)
steps = (('fsm0.expose', lambda s: b.fsm0.expose(s)), ('mask', lambda s: b.mask.propagate(s)))
current[0] = 'fsm0.expose'; b.fsm0.expose(src)
current[0] = 'mask'; b.mask.propagate(src)
current[0] = 'filter'; f1 = b.filter1.double_refract(src)[0]
current[0] = 'vcm'; v = b.vcm.reflect(f1)[0]
current[0] = 'dcm'; d = b.dcm.double_reflect(v)[0]
current[0] = 'slitDCM'; b.slitDCM.propagate(d)
current[0] = 'vfm'; m = b.vfm.reflect(d)[0]
current[0] = 'slitEH'; b.slitEH.propagate(m)
current[0] = 'sample'; b.sample.expose(m)
torch.cuda.synchronize()
for key, val in sorted(counts.items()):
    print(key, val)
