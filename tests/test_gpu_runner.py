"""GPU: the run_ray_tracing plug-in surface with on-device histogramming against
np.histogram2d on the host (numpy IS the reference's histogram code,
xrt/multipro.py:165-166)."""
import numpy as np
import pytest

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.run as rr
import xrt_amd.backends.raycing.screens as rsc
import xrt_amd.backends.raycing.sources as rs
from xrt_amd import plotter as xrtp, runner as xrtr, workloads

pytestmark = pytest.mark.gpu


def build():
    bl = raycing.BeamLine()
    bl.src = rs.GeometricSource(bl, 'src', nrays=40000, dx=0.1, dz=0.1, dxprime=2e-4,
                                dzprime=2e-5, distE='flat', energies=(8990., 9010.))
    bl.m1 = workloads.cfg2_toroid(bl)
    bl.scr = rsc.Screen(bl, 'scr', [0, 30000., 10000. * np.tan(8e-3)])
    return bl


def test_run_ray_tracing_accumulates_histograms():
    bl = build()
    kept = []

    def run_process(beamLine):
        b0 = beamLine.src.shine()
        gb, lb = beamLine.m1.reflect(b0)
        img = beamLine.scr.expose(gb)
        kept.append((lb, img))
        return {'beamSource': b0, 'beamM1local': lb, 'beamScreen': img}
    rr.run_process = run_process
    np.random.seed(11)
    plots = [
        xrtp.XYCPlot('beamM1local', (1,), xrtp.XYCAxis('x', 'mm', limits=[-1, 1], bins=64),
                     xrtp.XYCAxis('y', 'mm', limits=[-300, 300], bins=48)),
        xrtp.XYCPlot('beamScreen', (1, 3), xrtp.XYCAxis('x', u'µm', limits=[-400, 400],
                                                        bins=100),
                     xrtp.XYCAxis("z'", u'µrad', bins=32), fluxKind='s'),
    ]
    xrtr.run_ray_tracing(plots, repeats=3, beamLine=bl)
    assert len(kept) == 3 and plots[0].iteration == 3
    # plot 0 against numpy on the host copies
    ref = np.zeros((48, 64))
    nsel = 0
    inten = 0.
    for lb, img in kept:
        sel = lb.state == 1
        w = (lb.Jss + lb.Jpp)[sel]
        h, _, _ = np.histogram2d(lb.y[sel], lb.x[sel], bins=[48, 64],
                                 range=[[-300, 300], [-1, 1]], weights=w)
        ref += h
        nsel += sel.sum()
        inten += w.sum()
    assert plots[0].nRaysSelected == nsel and plots[0].nRaysAll == 3 * 40000
    assert abs(plots[0].intensity - inten) <= 1e-10 * inten
    assert np.abs(plots[0].total2D - ref).max() <= 1e-10 * ref.max()
    assert np.allclose(plots[0].total1D_x, ref.sum(axis=0), rtol=1e-9, atol=1e-9)
    # plot 1: unit factors, derived axis z' = c/b, auto limits, two ray flags, Jss
    ylim = plots[1].yaxis.limits
    ref = np.zeros((32, 100))
    for lb, img in kept:
        sel = (img.state == 1) | (img.state == 3)
        h, _, _ = np.histogram2d((img.c / img.b)[sel] * 1e6, img.x[sel] * 1e3,
                                 bins=[32, 100], range=[ylim, [-400, 400]],
                                 weights=img.Jss[sel])
        ref += h
    assert np.abs(plots[1].total2D - ref).max() <= 1e-10 * ref.max()
    assert plots[1].nRaysGood + plots[1].nRaysOver + plots[1].nRaysOut + \
        plots[1].nRaysDead == plots[1].nRaysAll


def test_rays_on_bin_edges_follow_numpy():
    """Edge semantics of np.histogram2d: right-open bins, last bin closed."""
    import ctypes
    import torch
    from xrt_amd import _lib
    n = 11
    b = rs.Beam(nrays=n)
    b.x = np.linspace(-1, 1, n)            # exactly on the edges of 10 bins
    b.z = np.array([-1., 1., 0., 0.2, -0.2, 0.6, 1.0000001, -1.0000001, 0.999999, 0., 0.4])
    b.state = np.ones(n, dtype=np.int32)
    plot = xrtp.XYCPlot('b', (1,), xrtp.XYCAxis('x', 'mm', limits=[-1, 1], bins=10),
                        xrtp.XYCAxis('z', 'mm', limits=[-1, 1], bins=5))
    xrtr.accumulate_plot(plot, {'b': b})
    ref, _, _ = np.histogram2d(b.z, b.x, bins=[5, 10], range=[[-1, 1], [-1, 1]],
                               weights=b.Jss + b.Jpp)
    assert np.array_equal(plot.total2D, ref)
    assert plot.intensityInRange == ref.sum() and plot.intensity == n


@pytest.mark.parametrize('bx,by,bc', [(40, 36, 50), (64, 64, 64), (128, 128, 128),
                                      (256, 256, 256), (100, 90, 2000), (300, 200, 64),
                                      (700, 600, 32)])
def test_colour_and_1d_histograms_match_numpy_and_matplotlib(bx, by, bc):
    """The complete per-plot reduce of multipro.py:316-361: hue from the colour
    axis (energy), hsv_to_rgb with value = flux, three 1-D histograms with
    flux / R / G / B weights, 2-D intensity and RGB histograms. Bin counts: all four
    2-D planes in the LDS of one block (up to 64 x 64); rays sorted by tile and accumulated
    tile by tile (4, 16, 14 tiles); 1-D cells beyond the LDS budget and more tiles than the
    sort takes (both through the global-atomic form)."""
    import matplotlib.colors as mc
    bl = build()
    kept = []

    def run_process(beamLine):
        b0 = beamLine.src.shine()
        gb, lb = beamLine.m1.reflect(b0)
        kept.append(lb)
        return {'beamM1local': lb}
    rr.run_process = run_process
    np.random.seed(12)
    plot = xrtp.XYCPlot(
        'beamM1local', (1, 3), xrtp.XYCAxis('x', 'mm', limits=[-0.8, 0.8], bins=bx),
        xrtp.XYCAxis('y', 'mm', limits=[-250, 250], bins=by),
        caxis=xrtp.XYCAxis('energy', 'eV', limits=[8992., 9008.], bins=bc),
        fluxKind='total')
    xrtr.run_ray_tracing([plot], repeats=2, beamLine=bl)
    ref2 = np.zeros((by, bx))
    ref2rgb = np.zeros((by, bx, 3))
    r1 = {k: np.zeros((n, 4)) for k, n in (('x', bx), ('y', by), ('c', bc))}
    for lb in kept:
        sel = (lb.state == 1) | (lb.state == 3)
        x, y, c = lb.x[sel], lb.y[sel], lb.E[sel]
        flux = (lb.Jss + lb.Jpp)[sel]
        c01 = ((c - 8992.) * plot.colorFactor / (9008. - 8992.)).reshape(-1, 1)
        c01[c01 < 0] = 0.
        c01[c01 > 1] = 1.
        hsv = np.dstack((c01, np.ones_like(c01) * plot.colorSaturation,
                         flux.reshape(-1, 1)))
        rgb = mc.hsv_to_rgb(hsv).reshape(-1, 3)
        ref2 += np.histogram2d(y, x, bins=[by, bx], range=[[-250, 250], [-0.8, 0.8]],
                               weights=flux)[0]
        for k in range(3):
            ref2rgb[:, :, k] += np.histogram2d(
                y, x, bins=[by, bx], range=[[-250, 250], [-0.8, 0.8]],
                weights=rgb[:, k])[0]
        for key, v, n, lim in (('x', x, bx, (-0.8, 0.8)), ('y', y, by, (-250, 250)),
                               ('c', c, bc, (8992., 9008.))):
            r1[key][:, 0] += np.histogram(v, bins=n, range=lim, weights=flux)[0]
            for k in range(3):
                r1[key][:, 1 + k] += np.histogram(v, bins=n, range=lim,
                                                  weights=rgb[:, k])[0]
    tol = 1e-10
    assert np.abs(plot.total2D - ref2).max() <= tol * ref2.max()
    assert np.abs(plot.total2D_RGB - ref2rgb).max() <= tol * ref2rgb.max()
    for key, axis in (('x', plot.xaxis), ('y', plot.yaxis), ('c', plot.caxis)):
        assert np.abs(axis.total1D4 - r1[key]).max() <= tol * r1[key].max(), key
    # the 1-D histograms see rays that fall outside the other axis' range
    assert plot.total1D_x.sum() > plot.total2D.sum() * (1 + 1e-6)
    assert np.array_equal(plot.total1D_c, plot.caxis.total1D4[:, 0])


def test_histograms_of_two_million_rays_match_numpy():
    """The bench's histogram workload at a size numpy still handles: 2e6 rays off the cfg2
    mirror (many chunks per block, ragged last chunk), 256 x 256 bins, accumulated twice on the
    device before the plot is read."""
    n = 2_000_003
    oe = workloads.cfg2_toroid()
    beam = workloads.synthetic_rays(n, 7)
    gb, lb = oe.reflect(beam)
    plot = xrtp.XYCPlot('b', (1,), xrtp.XYCAxis('x', 'mm', bins=256),
                        xrtp.XYCAxis('y', 'mm', bins=256),
                        caxis=xrtp.XYCAxis('energy', 'eV', bins=256))
    xrtr.accumulate_plot(plot, {'b': lb})
    xrtr.accumulate_plot(plot, {'b': lb})
    sel = np.array(lb.state) == 1
    x, y, e = np.array(lb.x)[sel], np.array(lb.y)[sel], np.array(lb.E)[sel]
    flux = (np.array(lb.Jss) + np.array(lb.Jpp))[sel]
    xl, yl, cl = plot.xaxis.limits, plot.yaxis.limits, plot.caxis.limits
    ref = 2 * np.histogram2d(y, x, bins=[256, 256], range=[yl, xl], weights=flux)[0]
    assert np.abs(plot.total2D - ref).max() <= 1e-10 * ref.max()
    for axis, v, lim in ((plot.xaxis, x, xl), (plot.yaxis, y, yl), (plot.caxis, e, cl)):
        r1 = 2 * np.histogram(v, bins=256, range=lim, weights=flux)[0]
        assert np.abs(axis.total1D4[:, 0] - r1).max() <= 1e-10 * r1.max()
    assert plot.nRaysSelected == 2 * int(sel.sum()) and plot.nRaysAll == 2 * n
    assert abs(plot.intensity - 2 * flux.sum()) <= 1e-10 * flux.sum()
    # a second read changes nothing; a zoomed plot drops the rays outside it from the 2-D
    # planes only
    assert np.array_equal(plot.total2D, plot.total2D)
    zoom = xrtp.XYCPlot('b', (1,), xrtp.XYCAxis('x', 'mm', bins=256,
                                                limits=[0.5 * xl[0], 0.5 * xl[1]]),
                        xrtp.XYCAxis('y', 'mm', bins=256, limits=[0.25 * yl[0], 0.25 * yl[1]]),
                        caxis=xrtp.XYCAxis('energy', 'eV', bins=256, limits=cl))
    xrtr.accumulate_plot(zoom, {'b': lb})
    ref = np.histogram2d(y, x, bins=[256, 256], range=[zoom.yaxis.limits, zoom.xaxis.limits],
                         weights=flux)[0]
    assert np.abs(zoom.total2D - ref).max() <= 1e-10 * ref.max()
    r1 = np.histogram(x, bins=256, range=zoom.xaxis.limits, weights=flux)[0]
    assert np.abs(zoom.total1D_x - r1).max() <= 1e-10 * r1.max()
    assert zoom.total1D_x.sum() > zoom.total2D.sum() * 1.01
    assert abs(zoom.intensityInRange - ref.sum()) <= 1e-10 * ref.sum()


@pytest.mark.parametrize('lines', [(9000.,), (8995., 9005.), (8991., 8996., 9000.5, 9009.)])
def test_colour_histogram_of_a_few_energies(lines):
    """A monochromatic beam (every ray in ONE bin of the energy axis) and beams of two and four
    lines: the wave adds the weights of lanes that share a bin across its lanes before one of
    them updates the cell (hist.hip: colour_line_update); sums against numpy / matplotlib."""
    import matplotlib.colors as mc
    n = 300_001
    oe = workloads.cfg2_toroid()
    beam = workloads.synthetic_rays(n, 11)
    rng = np.random.default_rng(2)
    beam.E[:] = np.asarray(lines)[rng.integers(0, len(lines), n)]
    gb, lb = oe.reflect(beam)
    plot = xrtp.XYCPlot('b', (1,), xrtp.XYCAxis('x', 'mm', bins=256),
                        xrtp.XYCAxis('y', 'mm', bins=256),
                        caxis=xrtp.XYCAxis('energy', 'eV', bins=128, limits=[8990., 9010.]))
    xrtr.accumulate_plot(plot, {'b': lb})
    sel = np.array(lb.state) == 1
    e = np.array(lb.E)[sel]
    flux = (np.array(lb.Jss) + np.array(lb.Jpp))[sel]
    c01 = np.clip((e - 8990.) * plot.colorFactor / 20., 0., 1.).reshape(-1, 1)
    hsv = np.dstack((c01, np.ones_like(c01) * plot.colorSaturation, flux.reshape(-1, 1)))
    rgb = mc.hsv_to_rgb(hsv).reshape(-1, 3)
    ref = np.zeros((128, 4))
    ref[:, 0] = np.histogram(e, bins=128, range=(8990., 9010.), weights=flux)[0]
    for k in range(3):
        ref[:, 1 + k] = np.histogram(e, bins=128, range=(8990., 9010.), weights=rgb[:, k])[0]
    assert (ref[:, 0] > 0).sum() == len(lines)
    assert np.abs(plot.caxis.total1D4 - ref).max() <= 1e-10 * ref.max()
    xl, yl = plot.xaxis.limits, plot.yaxis.limits
    r2 = np.histogram2d(np.array(lb.y)[sel], np.array(lb.x)[sel], bins=[256, 256],
                        range=[yl, xl], weights=flux)[0]
    assert np.abs(plot.total2D - r2).max() <= 1e-10 * r2.max()


@pytest.mark.parametrize('n', [1_000_003, 100_003])
@pytest.mark.parametrize('centre', [(0., 0.), (0.31, -0.52), (0.999, 0.999)])
def test_histograms_of_a_focused_beam(centre, n):
    """All rays in a spot of a few bins -- what a screen at a focus shows: the spot sits on the
    corner shared by four tiles of the 256 x 256 plot, inside one tile, and in the last bins of
    the plot (rays beyond the limits). The tile kernel shares its blocks out by the ray counts
    of the tiles (one tile may hold everything); empty tiles get none. Also at 1e5 rays
    (a hundred chunks: most blocks of the tile kernel have one chunk or none)."""
    rng = np.random.default_rng(5)
    beam = rs.Beam(nrays=n)
    beam.x = centre[0] + rng.normal(0, 0.01, n)
    beam.z = centre[1] + rng.normal(0, 0.002, n)
    beam.E = rng.uniform(8990., 9010., n)
    beam.Jss, beam.Jpp = rng.uniform(0.5, 1., n), rng.uniform(0., 0.2, n)
    beam.state = np.where(rng.uniform(size=n) < 0.97, 1, 2).astype(np.int32)
    plot = xrtp.XYCPlot('b', (1,), xrtp.XYCAxis('x', 'mm', bins=256, limits=[-1, 1]),
                        xrtp.XYCAxis('z', 'mm', bins=256, limits=[-1, 1]),
                        caxis=xrtp.XYCAxis('energy', 'eV', bins=256, limits=[8990, 9010]))
    hx, hz, he = beam.x.copy(), beam.z.copy(), beam.E.copy()
    flux, st = beam.Jss + beam.Jpp, beam.state.copy()
    xrtr.accumulate_plot(plot, {'b': beam})
    xrtr.accumulate_plot(plot, {'b': beam})
    sel = st == 1
    ref = 2 * np.histogram2d(hz[sel], hx[sel], bins=[256, 256], range=[[-1, 1], [-1, 1]],
                             weights=flux[sel])[0]
    assert ref.max() > 0 and np.abs(plot.total2D - ref).max() <= 1e-10 * ref.max()
    for axis, v in ((plot.xaxis, hx), (plot.yaxis, hz), (plot.caxis, he)):
        r1 = 2 * np.histogram(v[sel], bins=256, range=axis.limits, weights=flux[sel])[0]
        assert np.abs(axis.total1D4[:, 0] - r1).max() <= 1e-10 * r1.max()
    # the colour planes add up to what the flux plane holds: R + G + B of hsv(h, s, v) is
    # v (3 - s) - s v (f or 1 - f), bounded by v (3 - 2 s) and v (3 - s)
    rgb = plot.total2D_RGB.sum(axis=2)
    s_ = plot.colorSaturation
    assert (rgb <= ref * (3 - s_) * (1 + 1e-9) + 1e-9).all() and \
        (rgb >= ref * (3 - 2 * s_) * (1 - 1e-9) - 1e-9).all()
    assert plot.nRaysSelected == 2 * int(sel.sum())


@pytest.mark.parametrize('n', [300_001, 90_001])
@pytest.mark.parametrize('bx,by,with_counters', [(50, 40, True), (140, 141, False),
                                                 (300, 280, True), (1500, 1200, True)])
def test_plain_2d_histogram_entry_point(bx, by, with_counters, n):
    """xrt_hip_hist2d_f64_dev (one flux plane, no colour axis; what a plot without caxis needs):
    the plane in the LDS of one block, sorted by tile (one plane per tile), and beyond the
    tiles the sort takes; it ADDS into what it is given."""
    import ctypes
    import torch
    from xrt_amd import _lib
    oe = workloads.cfg2_toroid()
    gb, lb = oe.reflect(workloads.synthetic_rays(n, 3))
    dev = torch.device('cuda', torch.cuda.current_device())
    x, y = lb.dev('x', dev), lb.dev('y', dev)
    hx, hy = np.array(lb.x), np.array(lb.y)
    st = np.array(lb.state)
    sel = (st == 1) | (st == 3)
    xl = [float(hx[sel].min()), float(hx[sel].max())]
    yl = [0.6 * float(hy[sel].min()), 0.6 * float(hy[sel].max())]
    hist = torch.full((by, bx), 1.0, dtype=torch.float64, device=dev)
    counters = torch.zeros(8, dtype=torch.float64, device=dev)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    _lib.check(_lib.load().xrt_hip_hist2d_f64_dev(
        ctypes.byref(lb.to_struct(dev)), ptr(x), ptr(y), 1., 1., 1 | 4, 1, 2.5, bx, xl[0], xl[1],
        by, yl[0], yl[1], ptr(hist), ptr(counters) if with_counters else None,
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'xrt_hip_hist2d_f64_dev')
    w = 2.5 * np.array(lb.Jss)[sel]
    ref = np.histogram2d(hy[sel], hx[sel], bins=[by, bx], range=[yl, xl], weights=w)[0]
    got = hist.cpu().numpy() - 1.0
    assert np.abs(got - ref).max() <= 1e-10 * ref.max()
    if with_counters:
        c = counters.cpu().numpy()
        assert c[0] == sel.sum() and abs(c[1] - w.sum()) <= 1e-10 * w.sum()
        assert abs(c[2] - ref.sum()) <= 1e-10 * ref.sum()
        assert c[3] == (st > 0).sum() and c[4] == (st == 1).sum() and c[7] == (st < 0).sum()


@pytest.mark.parametrize('n', [1, 63, 777, 1025, 5000])
def test_small_beams_through_the_tile_sort(n):
    """Fewer rays than one chunk / a ragged second chunk, 200 x 200 bins (sorted by tile), one
    bin on the colour axis."""
    oe = workloads.cfg2_toroid()
    gb, lb = oe.reflect(workloads.synthetic_rays(n, 11))
    plot = xrtp.XYCPlot('b', (1, 2, 3), xrtp.XYCAxis('x', 'mm', bins=200, limits=[-5., 5.]),
                        xrtp.XYCAxis('y', 'mm', bins=200, limits=[-400., 400.]),
                        caxis=xrtp.XYCAxis('energy', 'eV', bins=1, limits=[8000., 10000.]))
    xrtr.accumulate_plot(plot, {'b': lb})
    st = np.array(lb.state)
    sel = (st >= 1) & (st <= 3)
    flux = (np.array(lb.Jss) + np.array(lb.Jpp))[sel]
    ref = np.histogram2d(np.array(lb.y)[sel], np.array(lb.x)[sel], bins=[200, 200],
                         range=[[-400., 400.], [-5., 5.]], weights=flux)[0]
    assert np.abs(plot.total2D - ref).max() <= 1e-10 * max(ref.max(), 1e-300)
    assert plot.nRaysSelected == int(sel.sum()) and plot.nRaysAll == n
    assert abs(plot.caxis.total1D4[0, 0] - flux.sum()) <= 1e-10 * max(flux.sum(), 1e-300)


def test_plot_histograms_of_an_empty_selection():
    """No ray matches the ray flag: all histograms stay zero, counters count."""
    bl = build()

    def run_process(beamLine):
        b0 = beamLine.src.shine()
        b0.state[:] = -1
        return {'beam': b0}
    rr.run_process = run_process
    np.random.seed(2)
    plot = xrtp.XYCPlot('beam', (1,), xrtp.XYCAxis('x', 'mm', limits=[-1, 1], bins=8),
                        xrtp.XYCAxis('z', 'mm', limits=[-1, 1], bins=8),
                        caxis=xrtp.XYCAxis('energy', 'eV', limits=[8990., 9010.], bins=8))
    xrtr.run_ray_tracing([plot], repeats=1, beamLine=bl)
    assert plot.nRaysSelected == 0 and plot.nRaysDead == 40000
    assert not plot.total2D.any() and not plot.total2D_RGB.any()
    assert not plot.xaxis.total1D4.any() and not plot.caxis.total1D4.any()


def test_two_plots_of_one_beam_with_and_without_beam_state():
    """A plot whose states come from ANOTHER beam (beamState) must not leave that
    state array behind in the beam's cached struct: the next plot of the same beam,
    and any later GPU op on it, read the beam's own states again."""
    bl = build()
    kept = []

    def run_process(beamLine):
        b0 = beamLine.src.shine()
        gb, lb = beamLine.m1.reflect(b0)
        img = beamLine.scr.expose(gb)
        lb.state[::7] = 3                 # make the two beams' selections differ
        kept.append((lb, img))
        return {'beamSource': b0, 'beamM1local': lb, 'beamScreen': img}
    rr.run_process = run_process
    np.random.seed(12)
    ax = lambda: (xrtp.XYCAxis('x', 'mm', limits=[-2, 2], bins=40),   # noqa: E731
                  xrtp.XYCAxis('z', 'mm', limits=[-2, 2], bins=40))
    plots = [xrtp.XYCPlot('beamScreen', (1,), *ax(), beamState='beamM1local'),
             xrtp.XYCPlot('beamScreen', (1,), *ax()),
             xrtp.XYCPlot('beamScreen', (1,), *ax(), beamState='beamM1local')]
    xrtr.run_ray_tracing(plots, repeats=2, beamLine=bl)
    ref_own = np.zeros((40, 40))
    ref_m1 = np.zeros((40, 40))
    for lb, img in kept:
        for ref, st in ((ref_own, img.state), (ref_m1, lb.state)):
            sel = st == 1
            h, _, _ = np.histogram2d(img.z[sel], img.x[sel], bins=[40, 40],
                                     range=[[-2, 2], [-2, 2]],
                                     weights=(img.Jss + img.Jpp)[sel])
            ref += h
    assert (ref_own != ref_m1).any()
    assert np.abs(plots[1].total2D - ref_own).max() <= 1e-10 * ref_own.max()
    assert np.abs(plots[0].total2D - ref_m1).max() <= 1e-10 * ref_m1.max()
    assert np.abs(plots[2].total2D - ref_m1).max() <= 1e-10 * ref_m1.max()


def test_two_threads_on_their_own_streams_equal_the_serial_run():
    """Thread safety of the host layer (error slot, armed events and the scratch-buffer
    cache are per thread / per stream): two Python threads trace different beams through
    their own elements at the same time, each on its own HIP stream; every array equals
    what the same work gives serially. (np.random is process-global, so the source
    beams are drawn before the threads start.)"""
    import threading

    import torch

    def source_beam(seed):
        np.random.seed(seed)
        src = rs.GeometricSource(raycing.BeamLine(), 'src', nrays=300000, dx=0.1, dz=0.1,
                                 dxprime=2e-4, dzprime=2e-5, distE='flat',
                                 energies=(8990., 9010.))
        return src.shine()

    def trace(b0, out, key, stream):
        bl = raycing.BeamLine()
        m1 = workloads.cfg2_toroid(bl)
        scr = rsc.Screen(bl, 'scr', [0, 30000., 10000. * np.tan(8e-3)])
        with torch.cuda.stream(stream):
            b = rs.Beam(copyFrom=b0)
            for _ in range(8):
                gb, lb = m1.reflect(b)
                img = scr.expose(gb)
            stream.synchronize()
            out[key] = {(n, f): np.array(getattr(bm, f)) for n, bm in (('lb', lb), ('img', img))
                        for f in ('x', 'z', 'state', 'Jss', 'Jsp')}

    beams = {seed: source_beam(seed) for seed in (3, 4)}
    serial, threaded = {}, {}
    for seed, b0 in beams.items():
        trace(b0, serial, seed, torch.cuda.current_stream())
    ts = [threading.Thread(target=trace, args=(b0, threaded, seed, torch.cuda.Stream()))
          for seed, b0 in beams.items()]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert set(threaded) == {3, 4}
    for seed in (3, 4):
        for key, a in serial[seed].items():
            assert np.array_equal(a, threaded[seed][key], equal_nan=True), (seed, key)
    assert not np.array_equal(serial[3][('lb', 'x')], serial[4][('lb', 'x')])


def test_run_ray_tracing_with_worker_threads_sums_the_same_histograms():
    """threads=3: three run_process calls in flight per step (own streams), the plots'
    copies summed -- with beams that do not depend on the draw order (the source beam is
    made once) the result equals the single-threaded run's."""
    bl = build()
    np.random.seed(5)
    fixed = bl.src.shine()
    calls = []

    def run_process(beamLine):
        b0 = rs.Beam(copyFrom=fixed)
        gb, lb = beamLine.m1.reflect(b0)
        img = beamLine.scr.expose(gb)
        calls.append(1)
        return {'beamM1local': lb, 'beamScreen': img}
    rr.run_process = run_process

    def plots():
        return [xrtp.XYCPlot('beamM1local', (1,), xrtp.XYCAxis('x', 'mm', bins=64),
                             xrtp.XYCAxis('y', 'mm', limits=[-300, 300], bins=48)),
                xrtp.XYCPlot('beamScreen', (1, 3), xrtp.XYCAxis('x', u'µm', limits=[-400, 400],
                                                                bins=100),
                             xrtp.XYCAxis("z'", u'µrad', bins=32), fluxKind='s')]
    one = xrtr.run_ray_tracing(plots(), repeats=7, beamLine=bl)
    assert len(calls) == 7
    many = xrtr.run_ray_tracing(plots(), repeats=7, beamLine=bl, threads=3)
    assert len(calls) == 14
    for a, b in zip(one, many):
        assert a.iteration == b.iteration == 7 and a.nRaysAll == b.nRaysAll
        assert a.xaxis.limits == b.xaxis.limits
        assert np.abs(a.total2D - b.total2D).max() <= 1e-12 * a.total2D.max()
        assert np.abs(a.total2D_RGB - b.total2D_RGB).max() <= 1e-12 * a.total2D_RGB.max()
        assert abs(a.intensity - b.intensity) <= 1e-12 * a.intensity
        assert a.nRaysGood == b.nRaysGood and a.nRaysSelected == b.nRaysSelected
        assert np.allclose(a.total1D_x, b.total1D_x, rtol=1e-12, atol=0)
