"""und_imap time against the number of rays: fixed cost vs per-ray cost of the persistent launch.
PYTHONPATH=. python tools/probe_und_scaling.py"""
import numpy as np
import torch
from xrt_amd import hipcalls
from xrt_amd.backends.raycing.undulator import clenshaw_curtis

dev = torch.device('cuda', 0)
rng = np.random.RandomState(5)
Kx, Ky, Np, L0, gamma0 = 0., 0.52, 108, 18.5, 5870.853297866972
for gi in (2, 8):
    dstep = 2 * np.pi / gi
    xk, wk = clenshaw_curtis(24)
    dI = np.arange(-np.pi + 0.5 * dstep, np.pi, dstep)
    tg = (dI[:, None] + 0.5 * dstep * xk).ravel()
    ag = (dI[:, None] * 0 + wk).ravel()
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    tabs = [up(t) for t in (tg, ag, np.sin(tg), np.cos(tg), np.sin(tg), np.cos(tg))]
    for lg in (16, 18, 20, 22, 24):
        n = 1 << lg
        dw, dth, dps = (up(rng.uniform(3900., 4250., n)), up(rng.uniform(-3e-5, 3e-5, n)),
                        up(rng.uniform(-3e-5, 3e-5, n)))
        call = lambda: hipcalls.undulator_imap(0, Kx, Ky, tabs, dw, dth, dps, L0, Np, gamma0, 0.5,  # noqa: E731
                                               dstep, True)
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nodes = len(tg)
        tf = 86. * n * nodes / ms * 1e3 / 1e12
        print('nodes %3d rays 2^%d: %8.4f ms  %.3f of 78.6 TF  (%.2f ns per ray-node x 1e3)' % (
            nodes, lg, ms, tf / 78.6, ms * 1e6 / (n * nodes) * 1e3))
