"""Pass time of OE(figureError=...) against the same element without a map, 1e7 rays:
PYTHONPATH=.:tests python tools/probe_figure.py"""
import time
import torch
import figure_cases as fc
import numpy as np
from xrt_amd import workloads

n = 10_000_000
beam = workloads.synthetic_rays(n, 42)
for f in beam.array_fields():
    beam.dev(f)
g = np.load(fc.GOLDEN + '/g2_figure_toroid.npz')
cases = [('toroid, lean kernel (no map)', workloads.cfg2_toroid())]
for label, step in (('toroid + roughness map 512 x 128', 2.), ('toroid + roughness map 2048 x 128', 0.5),
                    ('toroid + roughness map 4096 x 256', 0.07)):
    fe = fc.rfe.RandomRoughness(rms=3., corrLength=4., seed=11, limPhysX=[-10, 10],
                                limPhysY=[-300, 300], gridStep=step)
    print(label, 'spline', len(fe.local_z_spline.tck[0]), 'x', len(fe.local_z_spline.tck[1]), 'knots')
    cases.append((label, fc.element('g2_figure_toroid', g, fe)))
# the same rays pulled towards the axis: every ray lands on the mirror (no lane of a wave at the
# ends of the knot sequence: the computed-knot path serves whole waves)
narrow = workloads.synthetic_rays(n, 42)
for f in ('x', 'a'):
    narrow.dev(f).mul_(0.2)
for f in narrow.array_fields():
    narrow.dev(f)
runs = [(name, oe, beam) for name, oe in cases] + [(cases[1][0] + ', underfilled', cases[1][1], narrow)]
for name, oe, beam in runs:
    out = None
    for _ in range(5):
        out = oe.reflect(beam, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = oe.reflect(beam, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    tm = {}
    oe.reflect(beam, out=out, _timing=tm)
    print('%-36s %.3f ms per pass (kernel %.3f ms); good %.4f; optimistic pass %s' % (
        name, ms, tm['kernel_ms'], float((out[0].peek('state') == 1).mean()),
        not tm['exact_sequence']))
