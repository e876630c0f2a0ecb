"""GPU probe: accuracy against the golden vectors and throughput of the
undulator field sums. Run on the GPU box: python tools/probe_undulator.py"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xrt_amd import hipcalls  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                    'tests', 'golden')
E2WC = 5067.7309392068091


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


for tag in ('far_planar', 'far_helical', 'taper', 'nf'):
    g = np.load(os.path.join(GOLD, 'g9_undulator_%s.npz' % tag))
    tabs = [dev(g[k]) for k in ('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph')]
    tv = float(g['taperVal'])
    kw = dict(nper=int(g['Np']), alpha_s=0. if np.isnan(tv) else tv / E2WC,
              r0z=float(g['r0z']))
    rays = [dev(g[k]) for k in ('gamma', 'wu', 'w', 'ww1', 'ddphi', 'ddpsi')]
    Is, Ip = hipcalls.undulator(int(g['mode']), float(g['Kx']), float(g['Ky']), tabs,
                                *rays, **kw)
    Is, Ip = Is.cpu().numpy(), Ip.cpu().numpy()
    err = [np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in
           ((Is, g['Is']), (Ip, g['Ip']))]
    # throughput: tile the rays to 2^20 (far) / 2^17 (multi-period)
    n = 1 << (20 if int(g['mode']) == 0 else 17)
    big = [t.repeat((n + t.numel() - 1) // t.numel())[:n].contiguous() for t in rays]
    best = 1e9
    for _ in range(5):
        _, _, ms = hipcalls.undulator(int(g['mode']), float(g['Kx']), float(g['Ky']),
                                      tabs, *big, timing=True, **kw)
        best = min(best, ms)
    nodes = len(g['tg']) * (int(g['Np']) if int(g['mode']) else 1)
    print('%-12s rel err Is %.2e Ip %.2e | %d rays x %d nodes: %.3f ms -> %.3e '
          'ray-nodes/s' % (tag, err[0], err[1], n, nodes, best, n * nodes / best * 1e3))
