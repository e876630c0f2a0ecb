import time, numpy as np, torch
from xrt_amd import plotter as xrtp, runner
import xrt_amd.backends.raycing.sources as rs
n = 10_000_000
b = rs.Beam(nrays=n)
rng = np.random.RandomState(0)
b.x = rng.normal(0, 1., n); b.z = rng.normal(0, 1., n); b.E = rng.normal(9000., 1., n)
b.state = np.ones(n, dtype=np.int32); b.Jss = np.ones(n); b.Jpp = np.zeros(n)
for f in b.array_fields():
    b.dev(f)
for lim in (6., 3., 1.):
    for bins in (128, 256):
        plot = xrtp.XYCPlot('b', (1,), xrtp.XYCAxis('x', 'mm', bins=bins, limits=[-lim, lim]),
                            xrtp.XYCAxis('z', 'mm', bins=bins, limits=[-lim, lim]),
                            caxis=xrtp.XYCAxis('energy', 'eV', bins=bins, limits=[8994., 9006.]))
        runner.accumulate_plot(plot, {'b': b}); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            runner.accumulate_plot(plot, {'b': b})
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        ref = np.histogram2d(np.array(b.z), np.array(b.x), bins=[bins, bins], range=[[-lim, lim], [-lim, lim]])[0] * 11
        err = np.abs(plot.total2D - ref).max() / ref.max()
        print('gaussian beam, limits +-%g sigma, %d bins: %.3f ms per accumulate_plot, err %.1e' % (lim, bins, ms, err))
