"""Time of OE.multiple_reflect on the GPU: the toroid case of tests/multi_cases.py with n rays
(default 1e6), wall time of the whole call (all bounces, host loop included), repeated.
    python tools/probe_multi.py [n] [repeats]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import multi_cases as case                                   # noqa: E402
import xrt_amd.backends.raycing as raycing                   # noqa: E402
import xrt_amd.backends.raycing.materials as rm              # noqa: E402
import xrt_amd.backends.raycing.oes as roe                   # noqa: E402
import xrt_amd.backends.raycing.sources as rs                # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
bl = raycing.BeamLine(height=0)
au = rm.Material('Au', rho=19.3, kind='mirror')
oe = roe.ToroidMirror(bl, 'toroid', material=au, **case.TOROID)
beam = case.point_source_rays(rs, n, 5)
beam.to_struct(torch.device('cuda', 0))
for elev in (False, True):
    # (the batch statistics -- _info -- are collected by the exact phases only: one call for them,
    # the timed calls without, as a script calls it)
    info = []
    oe.multiple_reflect(beam, maxReflections=100, needElevationMap=elev, _info=info)
    times = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gb, lbN = oe.multiple_reflect(beam, maxReflections=100, needElevationMap=elev)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    nb = lbN.nrays // n
    entered = sum(one['n_enter'] for one in info)
    best = min(times)
    if not elev:
        print('live lanes per bounce (rays that enter / all rays): ' +
              ' '.join('%.3f' % (one['n_enter'] / n) for one in info) +
              '; hit: ' + ' '.join('%.3f' % (one['hit'] / n) for one in info))
    print('n %d elevation %s: %d bounces, %d ray-bounces, best %.3f ms (median %.3f) = %.3g '
          'ray-surface intersections / s; %.3f ms per bounce'
          % (n, elev, nb, entered, best * 1e3, np.median(times) * 1e3, entered / best,
             best * 1e3 / nb))
