"""The plot of a beam whose colour datum is the same for every ray (a monochromatic source: all
rays in ONE bin of the energy histogram) against one with energies spread over the axis:
PYTHONPATH=. python tools/probe_hist_mono.py"""
import time
import torch
from xrt_amd import workloads, plotter as xrtp, runner

n = 10_000_000
oe = workloads.cfg2_toroid()
beam = workloads.synthetic_rays(n, 42)
for f in beam.array_fields():
    beam.dev(f)
gb, lb = oe.reflect(beam)
for label, fill in (('energies spread over the colour axis', None), ('one energy', 9000.),
                    ('two energies', (8995., 9005.))):
    if fill is not None:
        E = lb.dev('E')
        if isinstance(fill, tuple):
            E[::2] = fill[0]
            E[1::2] = fill[1]
        else:
            E.fill_(fill)
    for bins in (128, 256):
        plot = xrtp.XYCPlot('b', (1,), xrtp.XYCAxis('x', 'mm', bins=bins), xrtp.XYCAxis('y', 'mm', bins=bins),
                            caxis=xrtp.XYCAxis('energy', 'eV', bins=bins, limits=[8990, 9010]))
        runner.accumulate_plot(plot, {'b': lb})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            runner.accumulate_plot(plot, {'b': lb})
        torch.cuda.synchronize()
        print('%-40s %3d bins  %.3f ms per plot' % (label, bins, (time.perf_counter() - t0) / 10 * 1e3))
