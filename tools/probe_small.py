"""Small kernels on 1e7 rays (run under rocprofv3 --kernel-trace): aperture, screen,
stand-alone amplitudes.  PYTHONPATH=. python tools/probe_small.py"""
import numpy as np
import torch
from xrt_amd import workloads
import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.apertures as ra
import xrt_amd.backends.raycing.screens as rsc
import xrt_amd.backends.raycing.materials as rm

n = 10_000_000
beam = workloads.synthetic_rays(n, 42)
for f in beam.array_fields():
    beam.dev(f)
bl = raycing.BeamLine()
slit = ra.RectangularAperture(bl, 's', [0, 10000., 0], ('left', 'right', 'bottom', 'top'),
                              [-0.2, 0.2, -0.2, 0.2])
scr = rsc.Screen(bl, 'scr', [0, 30000., 0])
for _ in range(3):
    slit.propagate(beam)
    scr.expose(beam)
pt = rm.Material('Pt', rho=21.45)
E = torch.as_tensor(np.random.default_rng(0).uniform(8990, 9010, n), device='cuda')
bdn = torch.full((n,), -4e-3, dtype=torch.float64, device='cuda')
for _ in range(3):
    pt.get_amplitude(E, bdn)
si = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
th = si.get_Bragg_angle(9000.)
g0 = torch.full((n,), -float(np.sin(th)), dtype=torch.float64, device='cuda')
for _ in range(3):
    si.get_amplitude(E, g0, -g0, g0)
torch.cuda.synchronize()
print('done')
