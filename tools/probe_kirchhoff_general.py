"""Kirchhoff kernel time on the SoftiMAX shape (2e5 samples x 2e5 points, Ep != 0, general
normals, points off a plane) over ppt: PYTHONPATH=. python tools/probe_kirchhoff_general.py"""
import numpy as np
import torch
from xrt_amd import hipcalls

ns = npix = 200_000
rng = np.random.default_rng(3)
dev = lambda a: torch.as_tensor(a, device='cuda')
sx = dev(rng.uniform(-5, 5, ns)); sy = dev(rng.uniform(-100, 100, ns)); sz = dev(rng.uniform(-0.1, 0.1, ns))
nrm = rng.normal(size=(3, ns)) * 0.01 + np.array([[0.], [0.], [1.]])
nrm /= np.sqrt((nrm**2).sum(0))
nx, ny, nz = (dev(c.copy()) for c in nrm)
k = dev(np.full(ns, 280. / 1973.2697177417986 * 1e7))
nl = dev(rng.uniform(0.01, 0.02, ns))
Es = dev(rng.normal(size=ns) + 1j * rng.normal(size=ns)); Ep = dev(rng.normal(size=ns) + 1j * rng.normal(size=ns))
px = dev(rng.uniform(-5, 5, npix)); py = dev(2000. + rng.uniform(-100, 100, npix)); pz = dev(20. + rng.uniform(-1, 1, npix))
for ppt in (1, 2, 4):
    for nsplit in (0, 64, 256):
        best = 1e9
        for _ in range(3):
            out = hipcalls.kirchhoff(px, py, pz, sx, sy, sz, nx, ny, nz, nl, k, Es, Ep,
                                     nsplit=nsplit, ppt=ppt, timing=True)
            best = min(best, out[-1])
        print('ppt %d nsplit %3d  %.1f ms  %.3e pairs/s  %s' % (
            ppt, nsplit, best, ns * npix / best * 1e3, sorted(hipcalls.kirchhoff_report()['variants'])))
