"""The plot in the tail of the pass against the separate histogram launches, same box: the e2e
scene of bench.py (device source -> toroid -> screen -> 256 x 256 XYCPlot) at n rays with
  focused: the bench's plot limits (+-1 mm around a ~50 um focus: all rays in a few tiles)
  wide:    limits of +-3 sigma of the image (the rays spread over all 16 tiles)
each with oes.fuseConsumers on / off; ms per iteration of run_ray_tracing.
    python tools/probe_plot_tail.py [n] [repeats]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xrt_amd import plotter as xrtp, runner, workloads                  # noqa: E402
from xrt_amd.backends.raycing import oes as roe, run as rr              # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bl, run_process, make_plot = workloads.e2e_beamline(n)
rr.run_process = run_process
roe.fuseConsumers = False
img = run_process(bl)['focus']
good = img.dev('state') == 1
sx = float(img.dev('x')[good].std())
sz = float(img.dev('z')[good].std())
del img, good
roe.fuseConsumers = True


def wide():
    return xrtp.XYCPlot('focus', (1,),
                        xrtp.XYCAxis('x', 'mm', bins=256, limits=[-3 * sx, 3 * sx]),
                        xrtp.XYCAxis('z', 'mm', bins=256, limits=[-3 * sz, 3 * sz]),
                        caxis=xrtp.XYCAxis('energy', 'eV', bins=256, limits=[8990, 9010]))


print('image sigma x %.4g mm, z %.4g mm; %d rays, %d iterations' % (sx, sz, n, reps))
for name, make in (('focused', make_plot), ('wide', wide)):
    row = {}
    for fuse in (True, False, True, False):
        roe.fuseConsumers = fuse
        runner.run_ray_tracing([make()], repeats=3, beamLine=bl)
        torch.cuda.synchronize()
        plot = make()
        t0 = time.perf_counter()
        runner.run_ray_tracing([plot], repeats=reps, beamLine=bl)
        torch.cuda.synchronize()
        row.setdefault(fuse, []).append((time.perf_counter() - t0) / reps * 1e3)
        flux = float(plot.total2D.sum())
    roe.fuseConsumers = True
    print('%-8s plot in the tail of the pass %.3f / %.3f ms per iteration, separate launches '
          '%.3f / %.3f ms  (flux in plot %.6g)'
          % (name, row[True][0], row[True][1], row[False][0], row[False][1], flux))
