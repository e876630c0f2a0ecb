"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g13_*.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only).

G13 (the source class around the custom-field kernel):
  g13_trajectory_plain / _filament   SourceFromField._build_trajectory_conv
        (synchr.py:1049-1147) on a tabulated 10-period field with tapered ends: the
        grids, the field on the half-step grid, the Runge-Kutta tables ON THE GRID (taken
        from the reference's own calls of its spline function) and the splined tables on
        the integration nodes.
  g13_sff_rays / g13_sff_filament    SourceFromField.shine() with a seeded numpy
        generator: the rays the reference returns.

While generating, oracle/undulator_np.py:trajectory is asserted against the reference.

Run:  python -m oracle.gen_fixtures_field_source
"""
import os
import time

import numpy as np

from . import _refenv
from . import undulator_np as un

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')

L0, NP = 30., 10
SOURCE = dict(nrays=400, eE=3.0, eI=0.5, eEspread=0, eEpsilonX=0.263, eEpsilonZ=0.008,
              betaX=9., betaZ=2., eMin=1500, eMax=1700, xPrimeMax=0.1, zPrimeMax=0.1,
              distE='BW', gNodes=40, gIntervals=20)


def tabulated_field():
    z = np.linspace(-L0*NP/2-40, L0*NP/2+40, 2000)
    env = 0.5*(np.tanh((z + L0*NP/2)/8.) - np.tanh((z - L0*NP/2)/8.))
    return np.vstack((z, 0.08*np.cos(2*np.pi*z/L0)*env,
                      0.6*np.sin(2*np.pi*z/L0)*env, 0*z)).T


def beam_arrays(b):
    out = {}
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'E', 'path', 'Jss', 'Jpp', 'Jsp', 'Es', 'Ep',
              'state'):
        out['beam_' + f] = np.array(getattr(b, f))
    for f in ('accepted', 'acceptedE', 'seeded', 'seededI', 'sourceWeight'):
        if hasattr(b, f):
            out['beam_' + f] = np.array(getattr(b, f))
    return out


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.sources.synchr as synchr
    raycing._VERBOSITY_ = 0
    field = tabulated_field()
    for tag, kw in (('plain', {}), ('filament', dict(filamentBeam=True))):
        bl = raycing.BeamLine()
        s = rs.SourceFromField(bl, 'sff', targetOpenCL=None, customField=field,
                               **dict(SOURCE, **kw))
        # ---- the trajectory, with the raw tables caught at the spline calls ----------
        np.random.seed(7)
        s.reset()
        if not hasattr(s, 'tg'):
            s._build_integration_grid()
        Bx, By, Bz = s._magnetic_field()
        raw = []
        orig = synchr.interp1d

        def spy(x, y, **kwargs):
            raw.append(np.array(y))
            return orig(x, y, **kwargs)
        synchr.interp1d = spy
        t0 = time.perf_counter()
        try:
            res = s._build_trajectory_conv(Bx, By, Bz)
        finally:
            synchr.interp1d = orig
        seconds = time.perf_counter() - t0
        gamma = float(s.gamma) if s.filamentBeam else None
        mine = un.trajectory(s.wtGrid, Bx, By, Bz, gamma)
        for a, b in zip((mine[0], mine[1], mine[3], mine[4], mine[5]), raw):
            assert np.array_equal(a, b)
        assert mine[2] == float(res[2][0])
        print('trajectory', tag, len(s.wtGrid), 'grid points, reference loop %.2f s' % seconds,
              'betam', res[2][0])
        np.savez_compressed(
            os.path.join(OUT, 'g13_trajectory_%s.npz' % tag), wtGrid=s.wtGrid, Bx=Bx, By=By,
            Bz=Bz, gamma=np.float64(s.gamma), filament=np.int32(bool(s.filamentBeam)),
            betax=raw[0], betay=raw[1], trajx=raw[2], trajy=raw[3], trajz=raw[4],
            betam=np.float64(res[2][0]), tg=s.tg, betax_tg=res[0], betay_tg=res[1],
            trajx_tg=res[3], trajy_tg=res[4], trajz_tg=res[5],
            reference_seconds=np.float64(seconds))
        # ---- the rays ------------------------------------------------------------------
        bl = raycing.BeamLine()
        s = rs.SourceFromField(bl, 'sff', targetOpenCL=None, customField=field,
                               **dict(SOURCE, **kw))
        np.random.seed(11)
        t0 = time.perf_counter()
        beam = s.shine()
        seconds = time.perf_counter() - t0
        print('shine', tag, len(beam.x), 'rays of', beam.seeded, 'seeded, %.1f s' % seconds,
              'Imax', s.Imax)
        np.savez_compressed(
            os.path.join(OUT, 'g13_sff_%s.npz' % ('rays' if tag == 'plain' else tag)),
            field=field, filament=np.int32(bool(s.filamentBeam)), Imax=np.float64(s.Imax),
            seed=np.int64(11), reference_seconds=np.float64(seconds), **beam_arrays(beam))


def node_search():
    """g13_nodes: the automatic search for the number of nodes (gNodes=None) -- the number
    found, and the probe value |field| dstep / 2 of one ray at the edge of the ranges (ten
    rays or fewer go through the reference's vectorised form _sp -- which, without a filament
    beam, only runs for ONE ray: its per-ray factor does not broadcast against the nodes)."""
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    raycing._VERBOSITY_ = 0
    field = tabulated_field()
    out = {}
    for tag, kw in (('plain', {}), ('filament', dict(filamentBeam=True))):
        cfg = dict(SOURCE, gp=1e-6, gIntervals=6, **kw)
        cfg.pop('gNodes')
        s = rs.SourceFromField(raycing.BeamLine(), name='sff', targetOpenCL=None,
                               customField=field, **cfg)
        np.random.seed(5)
        t0 = time.perf_counter()
        s.reset()
        seconds = time.perf_counter() - t0
        s.convergenceSearchFlag = True
        probe = s.build_I_map(s.E_max * np.ones(1), s.Theta_max * np.ones(1),
                              s.Psi_max * np.ones(1))
        s.convergenceSearchFlag = False
        print('node search', tag, 'quadm', s.quadm, '%.1f s' % seconds, 'probe', probe)
        out.update({tag + '_quadm': np.int64(s.quadm), tag + '_probe': np.array(probe),
                    tag + '_seconds': np.float64(seconds)})
    np.savez_compressed(os.path.join(OUT, 'g13_nodes.npz'), field=field, **out)


if __name__ == '__main__':
    if 'nodes' in os.sys.argv[1:]:
        node_search()
        raise SystemExit
    main()
