"""Kirchhoff kernel time over (ppt, nsplit) on the cfg4 shape: PYTHONPATH=. python tools/probe_kirchhoff.py"""
import sys
import numpy as np
import torch
from xrt_amd import hipcalls

ns, side = 1_000_000, 512
rng = np.random.default_rng(7)
dev = lambda a: torch.as_tensor(a, device='cuda')
sx = dev(rng.uniform(-0.1, 0.1, ns)); sy = dev(np.zeros(ns)); sz = dev(rng.uniform(-0.1, 0.1, ns))
k = dev(np.full(ns, 7900 / 1973.269804593025 * 1e1 * 1e0 * 1e3))
nl = dev(np.ones(ns)); nx = dev(np.zeros(ns)); ny = dev(np.ones(ns)); nz = dev(np.zeros(ns))
Es = dev(rng.normal(size=ns) + 1j * rng.normal(size=ns)); Ep = dev(np.zeros(ns, complex))
g = np.linspace(-0.5, 0.5, side)
X, Z = np.meshgrid(g, g)
px = dev(X.ravel().copy()); pz = dev(Z.ravel().copy()); py = dev(np.full(side * side, 1e4))
cases = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else \
    [(a, b) for a in (1, 2, 4, 2 | 0x200, 2 | 0x100) for b in (0, 8, 32, 64)]
for ppt, nsplit in cases:
    if True:
        best = 1e9
        for _ in range(3):
            out = hipcalls.kirchhoff(px, py, pz, sx, sy, sz, nx, ny, nz, nl, k, Es, Ep,
                                     nsplit=nsplit, ppt=ppt, timing=True)
            best = min(best, out[-1])
        print('ppt %#x nsplit %2d  %.1f ms  %.3e pairs/s' % (ppt, nsplit, best, ns * side * side / best * 1e3))
