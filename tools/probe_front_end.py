"""A monitor and the mask behind it on a RESIDENT beam (Balder's FSM0 and FEFixedMask on the
source beam): one pass over the rays (screen_expose_mark_kernel) against the two launches.
    python tools/probe_front_end.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from xrt_amd import workloads
import xrt_amd.backends.raycing.oes as roe
import xrt_amd.backends.raycing.sources as rs

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
b = workloads.balder_optics()
beam = workloads.synthetic_rays(n, 17, sa=1e-4, sc=3e-5, E=(8999., 9001.))
for f in beam.array_fields():
    beam.dev(f)
reps = 10
fresh = [rs.Beam(copyFrom=beam) for _ in range(2 * (reps + 1))]      # (the mask marks its input)
# (two launches: the image looked at before the mask comes -- the screen's own kernel, then the
# mask's states-only one, what the chain did before)
for name, look in (('one launch', False), ('two launches', True)):
    rays = fresh.pop()
    img = b.fsm0.expose(rays)
    b.mask.propagate(rays)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        rays = fresh.pop()
        img = b.fsm0.expose(rays)
        if look:
            img.nrays
        b.mask.propagate(rays)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('front-end screen + mask, %-13s %.3f ms   stopped by the mask %.4f' % (
        name, dt * 1e3, float((rays.state == b.mask.lostNum).sum()) / n))
