"""Quick on-GPU timing sweep of the Kirchhoff kernel (development aid)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from xrt_amd import hipcalls  # noqa: E402


def main():
    npix = int(float(sys.argv[1])) if len(sys.argv) > 1 else 512 * 512
    ns = int(float(sys.argv[2])) if len(sys.argv) > 2 else 200_000
    g = torch.Generator(device='cuda').manual_seed(7)
    r = lambda n, lo, hi: (torch.rand(n, generator=g, device='cuda', dtype=torch.float64) * (hi - lo) + lo)  # noqa: E731
    px, pz = r(npix, -.5, .5), r(npix, -.5, .5)
    py = torch.full((npix,), 10000., device='cuda', dtype=torch.float64)
    sx, sz, sy = r(ns, -.1, .1), r(ns, -.1, .1), torch.zeros(ns, device='cuda', dtype=torch.float64)
    nx = torch.zeros_like(sx); ny = torch.ones_like(sx); nz = torch.zeros_like(sx)
    nl = r(ns, .99, 1.)
    k = torch.full((ns,), 7900. / 1973.2697177417986 * 1e7, device='cuda', dtype=torch.float64)
    Es = torch.complex(r(ns, -1, 1), r(ns, -1, 1))
    Ep = torch.complex(r(ns, -1, 1), r(ns, -1, 1))
    for ppt in (1, 2):
        for nsplit in (0, 1, 2, 4, 8, 16):
            best = 1e30
            for it in range(3):
                *_, ms = hipcalls.kirchhoff(px, py, pz, sx, sy, sz, nx, ny, nz, nl, k, Es, Ep,
                                            nsplit=nsplit, ppt=ppt, timing=True)
                best = min(best, ms)
            pairs = npix * ns / (best * 1e-3)
            print('ppt=%d nsplit=%2d  %8.2f ms  %.3e pairs/s  %.1f%% of 78.6 TF (57 flop/pair)'
                  % (ppt, nsplit, best, pairs, pairs * 57 / 78.6e12 * 100), flush=True)


if __name__ == '__main__':
    main()
