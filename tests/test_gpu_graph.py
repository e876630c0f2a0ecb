"""GPU: run_ray_tracing(graph=True) -- the iterations after the first two are replays of ONE HIP
graph that holds the ray generator, the element passes and the histograms (xrt_amd/graphs.py).
The result must be the eager run's: the same rays in the same order of calls (the generator's
call number comes from a device cell the graph increments), hence the same histograms up to the
order of the fp64 accumulation."""
import numpy as np
import pytest

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.apertures as ra
import xrt_amd.backends.raycing.run as rr
import xrt_amd.backends.raycing.screens as rsc
import xrt_amd.backends.raycing.sources as rs
from xrt_amd import graphs, plotter as xrtp, runner as xrtr, workloads

pytestmark = pytest.mark.gpu


def beamline(n=30000, rng='device', seed=11):
    bl = raycing.BeamLine()
    bl.src = rs.GeometricSource(bl, 'src', nrays=n, dx=0.1, dz=0.1, dxprime=1e-4, dzprime=2e-5,
                                distE='flat', energies=(8999., 9001.), polarization='h',
                                rng=rng, seed=seed)
    bl.slit = ra.RectangularAperture(bl, 'slit', [0, 15000., 0], ('left', 'right', 'bottom',
                                                                  'top'), [-2, 2, -0.5, 0.5])
    bl.dcm = workloads.cfg3_dcm(bl)          # two branches from one source beam: the cfg3 DCM
    bl.m1 = workloads.cfg2_toroid(bl)        # and the cfg2 mirror with a screen at its focus
    bl.scr = rsc.Screen(bl, 'scr', [0, 20000. + 10000. * np.cos(8e-3), 10000. * np.sin(8e-3)])
    calls = []

    def run_process(beamLine):
        b0 = beamLine.src.shine()
        beamLine.slit.propagate(b0)
        g1, l1, l2 = beamLine.dcm.double_reflect(b0)
        g2, lm = beamLine.m1.reflect(b0)
        img = beamLine.scr.expose(g2)
        calls.append(1)
        return {'dcm2': l2, 'mirror': lm, 'screen': img}
    return bl, run_process, calls


def plots():
    return [xrtp.XYCPlot('mirror', (1,), xrtp.XYCAxis('x', 'mm', limits=[-3, 3], bins=64),
                         xrtp.XYCAxis('y', 'mm', limits=[-300, 300], bins=48),
                         caxis=xrtp.XYCAxis('energy', 'eV', limits=[8999, 9001], bins=32)),
            xrtp.XYCPlot('screen', (1,), xrtp.XYCAxis('x', 'mm', bins=256),    # automatic limits
                         xrtp.XYCAxis('z', 'mm', bins=256),
                         caxis=xrtp.XYCAxis('energy', 'eV', limits=[8999, 9001], bins=32)),
            xrtp.XYCPlot('dcm2', (1, 3), xrtp.XYCAxis('x', 'mm', limits=[-3, 3], bins=32),
                         xrtp.XYCAxis('y', 'mm', limits=[-50, 150], bins=32), fluxKind='s')]


def same_plots(one, two):
    for a, b in zip(one, two):
        assert a.iteration == b.iteration and a.nRaysAll == b.nRaysAll
        assert a.xaxis.limits == b.xaxis.limits and a.yaxis.limits == b.yaxis.limits
        assert a.total2D.max() > 0
        assert np.abs(a.total2D - b.total2D).max() <= 1e-12 * a.total2D.max()
        assert np.abs(a.total2D_RGB - b.total2D_RGB).max() <= 1e-12 * a.total2D_RGB.max()
        assert abs(a.intensity - b.intensity) <= 1e-12 * a.intensity
        assert a.nRaysGood == b.nRaysGood and a.nRaysSelected == b.nRaysSelected
        assert np.allclose(a.total1D_x, b.total1D_x, rtol=1e-12, atol=0)


@pytest.mark.parametrize('repeats, python_calls', [(9, 2), (40, 2), (150, 24)])
def test_replayed_iterations_equal_the_eager_run(repeats, python_calls):
    bl1, run1, calls1 = beamline()
    rr.run_process = run1
    eager = xrtr.run_ray_tracing(plots(), repeats=repeats, beamLine=bl1)
    assert len(calls1) == repeats
    bl2, run2, calls2 = beamline()
    rr.run_process = run2
    replayed = xrtr.run_ray_tracing(plots(), repeats=repeats, beamLine=bl2, graph=True)
    # the first iteration fixes the automatic limits, the second call of run_process is the
    # recording; the replays run no Python of the beamline
    # (a long run also times twenty eager iterations against twenty replays, alternately, after
    # two warm-up iterations of each, and keeps the graph only if it wins by more than 3 %)
    assert len(calls2) == python_calls or \
        (repeats >= 120 and not replayed[0].graphChoice['replaying'])
    if repeats >= 120:
        assert replayed[0].graphChoice['replay_ms'] > 0 and replayed[0].graphChoice['eager_ms'] > 0
    same_plots(eager, replayed)
    assert bl1.src._calls == bl2.src._calls == repeats
    # ... and the generator goes on where the replays left it
    a, b = bl1.src.shine(), bl2.src.shine()
    for f in ('x', 'z', 'a', 'c', 'E'):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_replay_with_figure_errors_on_the_elements():
    """OE(figureError=...) on the mirror and on the DCM (its crystals take the exact sequence of the
    Figured kernels): the map's spline is in HBM before the recording starts, the replays equal
    the eager run -- and differ from the run without maps."""
    from xrt_amd.backends.raycing import figure_error as rfe

    def with_maps():
        bl, run, calls = beamline()
        bl.m1.figureError = rfe.RandomRoughness(rms=4., corrLength=3., seed=5, limPhysX=[-10, 10],
                                                limPhysY=[-300, 300], gridStep=2.)
        bl.dcm.figureError = rfe.Waviness(amplitude=6., xWaveLength=5., yWaveLength=12.,
                                          limPhysX=[-10, 10], limPhysY=[-40, 160], gridStep=0.5)
        return bl, run, calls
    bl1, run1, _ = with_maps()
    rr.run_process = run1
    eager = xrtr.run_ray_tracing(plots(), repeats=7, beamLine=bl1)
    bl2, run2, calls2 = with_maps()
    rr.run_process = run2
    replayed = xrtr.run_ray_tracing(plots(), repeats=7, beamLine=bl2, graph=True)
    assert len(calls2) == 2
    same_plots(eager, replayed)
    bl3, run3, _ = beamline()
    rr.run_process = run3
    plain = xrtr.run_ray_tracing(plots(), repeats=7, beamLine=bl3)
    screen_maps, screen_plain = eager[1].total2D, plain[1].total2D
    assert eager[1].xaxis.limits != plain[1].xaxis.limits or \
        np.abs(screen_maps - screen_plain).max() > 1e-3 * screen_plain.max()


def test_a_second_run_records_again_and_continues_the_sequence():
    bl1, run1, _ = beamline(n=5000)
    bl2, run2, _ = beamline(n=5000)
    for graph, bl, run in ((False, bl1, run1), (True, bl2, run2)):
        rr.run_process = run
        first = xrtr.run_ray_tracing(plots(), repeats=4, beamLine=bl, graph=graph)
        second = xrtr.run_ray_tracing(plots(), repeats=5, beamLine=bl, graph=graph)
        if graph:
            same_plots(kept[0], first)
            same_plots(kept[1], second)
        kept = (first, second)
    assert bl1.src._calls == bl2.src._calls == 9


def test_host_randomness_is_refused_while_recording():
    bl, run, _ = beamline(n=2000, rng='host')
    rr.run_process = run
    with pytest.raises(graphs.CaptureError):
        xrtr.run_ray_tracing(plots(), repeats=5, beamLine=bl, graph=True)
    # nothing is left in recording state: the same beamline runs eagerly afterwards
    out = xrtr.run_ray_tracing(plots(), repeats=2, beamLine=bl)
    assert out[0].iteration == 2


def test_short_runs_never_record():
    bl, run, calls = beamline(n=2000)
    rr.run_process = run
    out = xrtr.run_ray_tracing(plots(), repeats=2, beamLine=bl, graph=True)
    assert len(calls) == 2 and out[0].iteration == 2


def test_two_shines_of_one_source_in_an_iteration():
    """Every recorded shine() increments the device cell: replay r draws calls 2r and 2r + 1."""
    bl1, _, _ = beamline(n=4000)
    bl2, _, _ = beamline(n=4000)

    def run_process(beamLine):
        first, second = beamLine.src.shine(), beamLine.src.shine()
        return {'mirror': beamLine.m1.reflect(first)[1], 'screen': beamLine.scr.expose(second),
                'dcm2': beamLine.dcm.double_reflect(second)[2]}
    rr.run_process = run_process
    eager = xrtr.run_ray_tracing(plots(), repeats=6, beamLine=bl1)
    replayed = xrtr.run_ray_tracing(plots(), repeats=6, beamLine=bl2, graph=True)
    same_plots(eager, replayed)
    assert bl1.src._calls == bl2.src._calls == 12
