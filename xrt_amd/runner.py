"""``run_ray_tracing`` — the plug-in surface of xrt (xrt/runner.py:513-719,
xrt/multipro.py:235-373) for the accelerated backend: calls the user's
``raycing.run.run_process(beamLine)`` *repeats* times and accumulates each
plot's histogram and ray counters. The histogram reduce runs on the GPU on the
resident beams (csrc/hist.hip), so only bins cross PCIe.

Threads/processes are not used: one process drives one GPU (the reference warns
that OpenCL and processes>1 cannot be combined, runner.py:560-564)."""
import ctypes

import numpy as np
import torch

from . import _lib
from .backends.raycing import run as rr


def _axis_tensor(beam, field, dev):
    if field == 'xprime':
        return beam.dev('a', dev) / beam.dev('b', dev)
    if field == 'zprime':
        return beam.dev('c', dev) / beam.dev('b', dev)
    return beam.dev(field, dev)


def accumulate_plot(plot, beams):
    """One iteration of get_output + do_hist2d for *plot* on the device."""
    _lib.require_gpu()
    lib = _lib.load()
    dev = torch.device('cuda', torch.cuda.current_device())
    beam = beams[plot.beam]
    x = _axis_tensor(beam, plot.xaxis.field(), dev).contiguous()
    y = _axis_tensor(beam, plot.yaxis.field(), dev).contiguous()
    for axis, t in ((plot.xaxis, x), (plot.yaxis, y)):
        if axis.limits is None:      # auto limits from the first batch
            sel = beam.dev('state', dev) == 1
            v = t[sel] * axis.factor if bool(sel.any()) else t * axis.factor
            lo, hi = float(v.min()), float(v.max())
            if hi <= lo:
                lo, hi = lo - 0.5, hi + 0.5
            axis.limits = [lo, hi]
    hist = torch.zeros((plot.yaxis.bins, plot.xaxis.bins), dtype=torch.float64,
                       device=dev)
    counters = torch.zeros(8, dtype=torch.float64, device=dev)
    srcw = beam.nrays * beam.sourceWeight if hasattr(beam, 'sourceWeight') else 1.
    state_beam = beam if plot.beamState is None else beams[plot.beamState]
    s = beam.to_struct(dev)
    if state_beam is not beam:
        keep = state_beam.dev('state', dev)
        s.state = keep.data_ptr()
        s._keep.append(keep)
    _lib.check(lib.xrt_hip_hist2d_f64_dev(
        ctypes.byref(s), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()),
        float(plot.xaxis.factor), float(plot.yaxis.factor), plot.ray_flag_mask,
        plot.flux_kind_code, float(srcw), plot.xaxis.bins,
        float(plot.xaxis.limits[0]), float(plot.xaxis.limits[1]), plot.yaxis.bins,
        float(plot.yaxis.limits[0]), float(plot.yaxis.limits[1]),
        ctypes.c_void_p(hist.data_ptr()), ctypes.c_void_p(counters.data_ptr()),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
        'xrt_hip_hist2d_f64_dev')
    c = counters.cpu().numpy()
    plot.total2D += hist.cpu().numpy()
    plot.nRaysAll += beam.nrays
    plot.nRaysSelected += int(c[0])
    plot.intensity += float(c[1])
    plot.intensityInRange += float(c[2])
    plot.nRaysAlive += int(c[3])
    plot.nRaysGood += int(c[4])
    plot.nRaysOut += int(c[5])
    plot.nRaysOver += int(c[6])
    plot.nRaysDead += int(c[7])
    plot.iteration += 1


def run_ray_tracing(plots=[], repeats=1, updateEvery=1, pickleEvery=None,
                    energyRange=None, backend='raycing', beamLine=None, threads=1,
                    processes=1, generator=None, generatorArgs=[],
                    generatorKWargs='auto', globalNorm=0, afterScript=None,
                    afterScriptArgs=[], afterScriptKWargs={}):
    """Runs ``raycing.run.run_process(beamLine)`` *repeats* times per generator
    step and sums the histograms of *plots* (rays are never accumulated, only
    their histograms — runner.py:520-526)."""
    if backend != 'raycing':
        raise NotImplementedError("only the 'raycing' backend is accelerated")
    if not isinstance(plots, (list, tuple)):
        plots = [plots]

    def one_scan():
        for _ in range(int(repeats)):
            beams = rr.run_process(beamLine)
            for plot in plots:
                accumulate_plot(plot, beams)

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
    return plots
