"""``run_ray_tracing`` — the plug-in surface of xrt (xrt/runner.py:513-719,
xrt/multipro.py:235-373) for the accelerated backend: calls the user's
``raycing.run.run_process(beamLine)`` *repeats* times and accumulates each
plot's histogram and ray counters. The histogram reduce runs on the GPU on the
resident beams (csrc/hist.hip), so only bins cross PCIe.

*threads* (or *processes*: one process can drive every GPU, so both mean the same here)
= that many ``run_process`` calls in flight, as the reference starts that many workers per
step (xrt/runner.py:248-330): each worker is a Python thread with its own HIP stream, on
the visible GPUs in turn, filling its own copy of the plots; the copies are summed in the
order of the iterations. While any plot limit is still automatic the first iteration runs
alone, as in the reference (its ``uniqueFirstRun``). numpy's global generator is shared
by the workers (as it is by the reference's threads): reproducible runs use 1 thread."""
import ctypes
import os as _os
import threading
import time

import numpy as np
import torch

from . import hipcalls as _hipcalls

from . import _lib, _structs, graphs, hipcalls
from .backends.raycing import run as rr


def _axis_tensor(beam, field, dev):
    if field == 'xprime':
        return beam.dev('a', dev) / beam.dev('b', dev)
    if field == 'zprime':
        return beam.dev('c', dev) / beam.dev('b', dev)
    return beam.dev(field, dev)


def _plot_record(plot, srcw):
    P = _structs.Plot()
    P.x_factor, P.y_factor, P.c_factor = (float(plot.xaxis.factor),
                                          float(plot.yaxis.factor), float(plot.caxis.factor))
    P.source_weight = float(srcw)
    for lim, axis in ((P.x_lim, plot.xaxis), (P.y_lim, plot.yaxis), (P.c_lim, plot.caxis)):
        lim[0], lim[1] = float(axis.limits[0]), float(axis.limits[1])
    P.color_factor = float(plot.colorFactor)
    P.color_saturation = float(plot.colorSaturation)
    P.bins_x, P.bins_y, P.bins_c = plot.xaxis.bins, plot.yaxis.bins, plot.caxis.bins
    P.ray_flags, P.flux_kind = plot.ray_flag_mask, plot.flux_kind_code
    return P


def _join_the_pass(plot, beam, dev, lib):
    """*beam* is the image of a screen whose pass has not been launched (Screen.expose of the
    pending global beam of OE.reflect, sources.LazyBeam) and this plot is the first to look at
    it: the plot rides in the tail of that pass (oes._DeferredReflect.plot_on) -- the rays'
    weights, hues and bins are formed from the registers that hold the image, which itself is
    not written unless somebody has asked the screen for it before. -> True if it did. Fixed
    limits only (automatic ones are read back from the first beam: the usual route)."""
    from .backends.raycing import oes as roe
    op = beam.__dict__.get('_op')
    if not roe.fuseConsumers or getattr(op, 'image', None) is not beam or op.state != 'pending' \
            or plot.beamState is not None or _os.environ.get('XRT_PLOT_TAIL_OFF', '') == '1':
        return False            # (XRT_PLOT_TAIL_OFF=1: the plot's own launches, for A/B runs)
    axes = (plot.xaxis, plot.yaxis, plot.caxis)
    if any(a.limits is None for a in axes) or \
            any(a.field() not in _structs.PLOT_FIELDS for a in axes):
        return False
    srcw = op.n * beam.sourceWeight if 'sourceWeight' in beam.__dict__ else 1.
    tail = _structs.PlotTail()
    tail.plot = _plot_record(plot, srcw)
    tail.x_field, tail.y_field, tail.c_field = (_structs.PLOT_FIELDS[a.field()] for a in axes)
    need = ctypes.c_size_t(0)
    _lib.check(lib.xrt_hip_plot_tail_workspace_bytes(op.n, ctypes.byref(tail), ctypes.byref(need)),
               'xrt_hip_plot_tail_workspace_bytes')
    if need.value == 0:
        return False
    flat = plot.device_accumulator(dev)
    hist, hist_rgb, hx, hy, hc, counters = torch.split(flat, plot.part_sizes())
    tail.hist2d, tail.hist2d_rgb, tail.hist_x, tail.hist_y = (
        t.data_ptr() for t in (hist, hist_rgb, hx, hy))
    tail.hist_c = hc.data_ptr() if plot.ePos else None
    tail.counters = counters.data_ptr()
    ws = hipcalls.workspace(dev, need.value, 'plot_tail')
    tail.workspace, tail.workspace_bytes = ws.data_ptr(), ws.numel()
    if not op.plot_on(tail):
        return False
    nrays = op.n

    def count():
        plot.nRaysAll += nrays
        plot.iteration += 1
    graphs.per_iteration(count)
    return True


def accumulate_plot(plot, beams, sole=True):
    """One iteration of get_output + do_hist2d for *plot* on the device. *sole*: no other plot
    of this iteration shows the same beam (then the plot may ride in the tail of the pass that
    makes the beam, _join_the_pass)."""
    _lib.require_gpu()
    lib = _lib.load()
    dev = torch.device('cuda', torch.cuda.current_device())
    beam = beams[plot.beam]
    from .backends.raycing import sources as _rs
    if sole and type(beam) is _rs.LazyBeam and _join_the_pass(plot, beam, dev, lib):
        return
    x = _axis_tensor(beam, plot.xaxis.field(), dev).contiguous()
    y = _axis_tensor(beam, plot.yaxis.field(), dev).contiguous()
    for axis, t in ((plot.xaxis, x), (plot.yaxis, y)):
        if axis.limits is None:      # auto limits from the first batch
            graphs.refuse('automatic plot limits (read back from the first beam)')
            sel = beam.dev('state', dev) == 1
            v = t[sel] * axis.factor if bool(sel.any()) else t * axis.factor
            lo, hi = float(v.min()), float(v.max())
            if hi <= lo:
                lo, hi = lo - 0.5, hi + 0.5
            axis.limits = [lo, hi]
    cax = plot.caxis
    cdat = _axis_tensor(beam, cax.field(), dev).contiguous()
    if cax.limits is None:
        graphs.refuse('automatic plot limits (read back from the first beam)')
        sel = beam.dev('state', dev) == 1
        v = cdat[sel] * cax.factor if bool(sel.any()) else cdat * cax.factor
        lo, hi = float(v.min()), float(v.max())
        if hi <= lo:
            lo, hi = lo - 0.5, hi + 0.5
        cax.limits = [lo, hi]
    # all results of the call go into ONE device buffer, the plot's own accumulator: no memset,
    # no copy back, no sync per iteration (the plot brings it home when it is read)
    flat = plot.device_accumulator(dev)
    hist, hist_rgb, hx, hy, hc, counters = torch.split(flat, plot.part_sizes())
    srcw = beam.nrays * beam.sourceWeight if hasattr(beam, 'sourceWeight') else 1.
    state_beam = beam if plot.beamState is None else beams[plot.beamState]
    s = beam.to_struct(dev)
    if state_beam is not beam:
        # the struct is the beam's cached one: the other beam's state goes into a
        # private copy of it, alive for this call only (the state tensor itself is the
        # other beam's; the launch is ordered on this stream)
        cached, s = s, _structs.Beam()
        ctypes.memmove(ctypes.byref(s), ctypes.byref(cached), ctypes.sizeof(s))
        s.state = state_beam.dev('state', dev).data_ptr()
    P = _plot_record(plot, srcw)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    # scratch from torch's allocator, one per (thread, stream): nothing is allocated per call
    # (and a call recorded into a HIP graph keeps pointing at memory the graph owns)
    need = ctypes.c_size_t(0)
    _lib.check(lib.xrt_hip_plot_hist_workspace_bytes(beam.nrays, ctypes.byref(P), 1, 1,
                                                     ctypes.byref(need)),
               'xrt_hip_plot_hist_workspace_bytes')
    ws = hipcalls.workspace(dev, need.value, 'hist')
    _lib.check(lib.xrt_hip_plot_hist_ws_f64_dev(
        ctypes.byref(s), ptr(x), ptr(y), ptr(cdat), ctypes.byref(P), ptr(hist),
        ptr(hist_rgb), ptr(hx), ptr(hy), ptr(hc) if plot.ePos else None, ptr(counters),
        ptr(ws), ws.numel(), _hipcalls.stream_ptr()),
        'xrt_hip_plot_hist_ws_f64_dev')
    nrays = beam.nrays

    def count():
        plot.nRaysAll += nrays
        plot.iteration += 1
    graphs.per_iteration(count)


def _parallel_iterations(plots, beamLine, count):
    """*count* iterations at once: one thread each, own stream, GPUs in turn."""
    ndev = torch.cuda.device_count()
    results, errors = [None] * count, []

    def work(k):
        try:
            with torch.cuda.device(k % ndev):
                stream = torch.cuda.Stream()
                with torch.cuda.stream(stream):
                    mine = [p.spawn() for p in plots]
                    beams = rr.run_process(beamLine)
                    shown = [p.beam for p in mine]
                    for plot in mine:
                        accumulate_plot(plot, beams, sole=shown.count(plot.beam) == 1)
                    stream.synchronize()
            results[k] = mine
        except BaseException as e:   # noqa: BLE001  (re-raised in the caller's thread)
            errors.append(e)
    pool = [threading.Thread(target=work, args=(k,)) for k in range(count)]
    for t in pool:
        t.start()
    for t in pool:
        t.join()
    if errors:
        raise errors[0]
    for mine in results:
        for plot, part in zip(plots, mine):
            plot.absorb(part)


def run_ray_tracing(plots=[], repeats=1, updateEvery=1, pickleEvery=None,
                    energyRange=None, backend='raycing', beamLine=None, threads=1,
                    processes=1, generator=None, generatorArgs=[],
                    generatorKWargs='auto', globalNorm=0, afterScript=None,
                    afterScriptArgs=[], afterScriptKWargs={}, graph=False):
    """Runs ``raycing.run.run_process(beamLine)`` *repeats* times per generator
    step and sums the histograms of *plots* (rays are never accumulated, only
    their histograms — runner.py:520-526).

    *graph* (not in the reference): after the iterations that fix the automatic plot limits
    and one more eager one, an iteration -- ``run_process`` and the histograms of all plots --
    is recorded into a HIP graph and the remaining ones are replays of it: one launch per
    iteration instead of ~200 us of Python per element chain, which is what bounds beams of
    up to ~1e6 rays (xrt_amd/graphs.py). Needs a ``run_process`` whose host side does the
    same thing every time: device ray generator (``GeometricSource(rng='device')``), no
    host random numbers, no host access to the rays; anything else raises
    ``graphs.CaptureError`` while recording. One worker."""
    if backend != 'raycing':
        raise NotImplementedError("only the 'raycing' backend is accelerated")
    if not isinstance(plots, (list, tuple)):
        plots = [plots]

    workers = max(int(threads), int(processes), 1)
    graph_choice = {}            # (what the graph route measured and chose; plots[0].graphChoice)

    def iteration():
        beams = rr.run_process(beamLine)
        shown = [p.beam for p in plots]
        for plot in plots:
            accumulate_plot(plot, beams, sole=shown.count(plot.beam) == 1)
        # (an element pass nobody has looked at yet is launched now: an iteration leaves
        # nothing behind -- also what a recorded iteration has to contain)
        from .backends.raycing import sources as rs
        rs.flush_pending()
        return beams

    def one_scan():
        left = int(repeats)
        recorded, ran_eagerly, refused = None, False, False
        while left > 0:
            if graph and workers == 1 and not any(
                    a.limits is None for p in plots for a in (p.xaxis, p.yaxis, p.caxis)):
                # one iteration has run eagerly (workspaces, tables, compiled units, plot limits:
                # what only a first call does); the next one is recorded.
                # (Several iterations per graph were tried: 0.144 ms per 1e5-ray iteration
                # against 0.135-0.15 with one -- the time of a replay is the gaps between its
                # dependent nodes, not the launch of the graph.)
                if ran_eagerly and recorded is None and left > 1 and not refused:
                    recorded = graphs.IterationGraph(iteration)
                    if left >= 120:
                        # Which is faster HERE: a replay or the eager loop? A stream takes
                        # kernels back to back while the host runs ahead; the nodes of a graph
                        # wait for one another through barrier packets (~8 us each). Beams of
                        # ~1e6 rays and more are GPU-bound and lose by replaying. Both ways
                        # warm (the eager one has been since before the recording; the replay's
                        # first launches upload the graph), then two rounds of ten iterations
                        # each way (a stream sync per block only: at 1e6 rays a sync every
                        # five iterations made the eager loop look 15 % slower than it runs
                        # freely), ALTERNATELY -- a cold eager loop measured after a warm
                        # replay made the graph look better than it is --
                        # and the graph stays only if it wins by more than 3 %.
                        for run in (recorded.replay, recorded.replay, iteration, iteration):
                            run()
                        clock = [0., 0.]
                        for _ in range(2):
                            for which, run in enumerate((recorded.replay, iteration)):
                                torch.cuda.current_stream().synchronize()
                                t0 = time.perf_counter()
                                for _ in range(10):
                                    run()
                                torch.cuda.current_stream().synchronize()
                                clock[which] += time.perf_counter() - t0
                        left -= 44
                        graph_choice['replay_ms'] = clock[0] / 20 * 1e3
                        graph_choice['eager_ms'] = clock[1] / 20 * 1e3
                        if clock[0] > 0.97 * clock[1]:
                            recorded.close()
                            recorded, refused = None, True
                        graph_choice['replaying'] = not refused
                        continue
                if recorded is not None:
                    recorded.replay()
                    left -= 1
                    continue
            auto = any(a.limits is None for p in plots for a in (p.xaxis, p.yaxis, p.caxis))
            batch = 1 if (auto or workers == 1) else min(workers, left)
            if batch == 1:
                iteration()
                ran_eagerly = True
            else:
                _parallel_iterations(plots, beamLine, batch)
            left -= batch
        if recorded is not None:
            torch.cuda.current_stream().synchronize()    # (its replays have run: it can go)
            recorded.close()

    if generator is None:
        one_scan()
    else:
        if generatorKWargs == 'auto':
            kw = dict(plots=plots, beamLine=beamLine) if not generatorArgs else {}
        else:
            kw = generatorKWargs
        for _ in generator(*generatorArgs, **kw):
            one_scan()
            for plot in plots:      # a new scan point starts from empty bins
                plot.lastTotal2D = plot.total2D.copy()
                plot.reset_bins2D()
    if afterScript:
        afterScript(*afterScriptArgs, **afterScriptKWargs)
    if graph and plots:
        plots[0].graphChoice = graph_choice
    return plots
