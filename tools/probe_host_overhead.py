"""Host time per element call (tiny beams: the GPU work is negligible, what is left is Python +
ctypes + launch): PYTHONPATH=. python tools/probe_host_overhead.py"""
import time
import torch
from xrt_amd import workloads, plotter as xrtp, runner
import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.screens as rsc

n = 2000
beam = workloads.synthetic_rays(n, 1)
for f in beam.array_fields():
    beam.dev(f)
b3 = workloads.synthetic_rays(n, 2, sa=1e-4, E=(8995., 9005.))
for f in b3.array_fields():
    b3.dev(f)
oe, dcm = workloads.cfg2_toroid(), workloads.cfg3_dcm()
bl, run_process, make_plot = workloads.e2e_beamline(n)
scr = rsc.Screen(raycing.BeamLine(), 'scr', [0, 30000., 0])
plot = make_plot()


def clock(what, fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-44s %7.1f us per call on the host (+ %.1f us to drain)' % (
        what, (t1 - t0) / reps * 1e6, (t2 - t1) * 1e6))


out = [None]


def reuse():
    out[0] = oe.reflect(beam, out=out[0])


clock('OE.reflect, new outputs', lambda: oe.reflect(beam))
clock('OE.reflect(out=...)', reuse)
clock('OE.reflect(needLocal=False)', lambda: oe.reflect(beam, needLocal=False))
clock('DCM.double_reflect', lambda: dcm.double_reflect(b3))
clock('Screen.expose', lambda: scr.expose(beam))
clock('GeometricSource(rng=device).shine', lambda: bl.source.shine())
img = scr.expose(beam)
clock('accumulate_plot', lambda: runner.accumulate_plot(plot, {'focus': img}))
clock('run_process (source, mirror, screen)', lambda: run_process(bl))
import cProfile
import pstats
for what, fn in (('OE.reflect', lambda: oe.reflect(beam)),
                 ('DCM.double_reflect', lambda: dcm.double_reflect(b3))):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        fn()
    pr.disable()
    print('==== cProfile of 200 x %s' % what)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
    pstats.Stats(pr).sort_stats('tottime').print_stats(20)
