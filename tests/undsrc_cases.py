"""Helper: rebuild the G10 undulator-source cases with xrt_amd's classes."""
import json
import os

import numpy as np


def load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, 'g10_undsrc_%s.npz' % tag))


def build(g):
    import xrt_amd.backends.raycing as raycing
    import xrt_amd.backends.raycing.sources as rs
    import xrt_amd.backends.raycing.apertures as ra
    kw = json.loads(str(g['ctor']))
    for k in ('taper', 'center'):
        if k in kw:
            kw[k] = tuple(kw[k])
    bl = raycing.BeamLine()
    src = rs.Undulator(bl, 'und', **kw)
    wave = None
    dist, size, ns = g['wave_geom']
    if ns > 0:
        slit = ra.RectangularAperture(
            bl, 'slit', [0, dist, 0], ('left', 'right', 'bottom', 'top'),
            [-size/2, size/2, -size/2, size/2])
        wave = slit.prepare_wave(src, int(ns))
    return src, wave, json.loads(str(g['shine']))
