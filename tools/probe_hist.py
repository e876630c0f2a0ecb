"""Time of the per-plot histogram reduce (N2) on 1e7 rays, by bin count:
PYTHONPATH=. python tools/probe_hist.py"""
import time
import torch
from xrt_amd import workloads, plotter as xrtp, runner

n = 10_000_000
oe = workloads.cfg2_toroid()
beam = workloads.synthetic_rays(n, 42)
for f in beam.array_fields():
    beam.dev(f)
gb, lb = oe.reflect(beam)
import sys
for bins in ([int(v) for v in sys.argv[1:]] or [64, 128, 256, 512]):
    for name, b, xa, ya in (('footprint', lb, 'x', 'y'), ('global xz', gb, 'x', 'z')):
        plot = xrtp.XYCPlot('b', (1,), xrtp.XYCAxis(xa, 'mm', bins=bins), xrtp.XYCAxis(ya, 'mm', bins=bins),
                            caxis=xrtp.XYCAxis('energy', 'eV', bins=bins))
        runner.accumulate_plot(plot, {'b': b})          # sets the limits
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            runner.accumulate_plot(plot, {'b': b})
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nsel = plot.nRaysSelected                       # brings the accumulator home
        t2 = time.perf_counter()
        print('%3d bins %-10s %.3f ms per accumulate_plot (device accumulators), %.2f ms to read '
              'the plot back, selected %d' % (bins, name, (t1 - t0) / 10 * 1e3, (t2 - t1) * 1e3, nsel))
