"""Quick on-GPU timing sweep of the Kirchhoff kernel (development aid)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from xrt_amd import hipcalls, workloads  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    h = workloads.kirchhoff_case(cfg) if cfg in (4, 5) else workloads.kirchhoff_custom(200000, 512)
    dev = torch.device('cuda', 0)
    up = lambda a, dt=np.float64: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)  # noqa: E731
    ns = h['ns']
    args = [up(h['px']), up(h['py']), up(h['pz']), up(h['sx']), up(h['sy']), up(h['sz']),
            up(np.zeros(ns)), up(np.ones(ns)), up(np.zeros(ns)), up(h['nl']), up(h['k']),
            up(h['Es'], np.complex128), up(h['Ep'], np.complex128)]
    npix = h['px'].size
    for ppt in (1, 2):
        for nsplit in (1, 2, 4, 8, 16, 32):
            best = 1e30
            for it in range(2):
                *_, ms = hipcalls.kirchhoff(*args, nsplit=nsplit, ppt=ppt, timing=True)
                best = min(best, ms)
            pairs = npix * ns / (best * 1e-3)
            print('ppt=%d nsplit=%2d  %8.2f ms  %.3e pairs/s  %.1f%% of 78.6 TF (57 flop/pair)'
                  % (ppt, nsplit, best, pairs, pairs * 57 / 78.6e12 * 100), flush=True)


if __name__ == '__main__':
    main()
