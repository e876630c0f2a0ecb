"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g14_bend_sources.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only).

G14 (the remaining synchrotron sources): BendingMagnet / Wiggler
(sources/synchr.py:69-610): the intensity map ``build_I_map`` on random (E, theta, psi) and
the rays of a seeded ``shine()`` for: a magnet by field, a magnet by radius with a filament
beam, a magnet with energy spread and uniform ray density, a wiggler, a wiggler with
energy spread, a wiggler with a filament beam.

Run:  python -m oracle.gen_fixtures_bend_source
"""
import os

import numpy as np

from . import _refenv

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')
COMMON = dict(nrays=1500, eE=3.0, eI=0.5, eEpsilonX=0.263, eEpsilonZ=0.008, betaX=9.,
              betaZ=2., eMin=5000, eMax=15000, xPrimeMax=1.5, zPrimeMax=0.3, distE='BW')
CASES = (
    ('bm_field', 'BendingMagnet', dict(B0=1.7)),
    ('bm_filament', 'BendingMagnet', dict(rho=5.9, filamentBeam=True)),
    ('bm_uniform', 'BendingMagnet', dict(B0=1.7, eEspread=1e-3, uniformRayDensity=True,
                                         distE='eV')),
    ('wiggler', 'Wiggler', dict(K=12., period=80., n=10, pitch=1e-4, yaw=-2e-4)),
    ('wiggler_spread', 'Wiggler', dict(K=12., period=80., n=10, eEspread=1e-3,
                                       xPrimeMax=3.)),
    ('wiggler_filament', 'Wiggler', dict(K=12., period=80., n=10, filamentBeam=True)),
)
SEED = 17


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    raycing._VERBOSITY_ = 0
    out = {}
    for tag, cls, kw in CASES:
        args = dict(COMMON, **kw)
        s = getattr(rs, cls)(raycing.BeamLine(azimuth=0.02), name=tag, center=(1., 2., 3.),
                             **args)
        # ---- the map ------------------------------------------------------------------
        s.reset()
        rng = np.random.RandomState(4)
        n = 1500
        E = rng.uniform(s.E_min, s.E_max, n)
        th = rng.uniform(s.Theta_min, s.Theta_max, n)
        ps = rng.uniform(s.Psi_min, s.Psi_max, n)
        ps[:3] = 0.
        th[:2] = 0.
        np.random.seed(SEED + 1)
        I, Es, Ep = s.build_I_map(E, th, ps)
        out.update({tag + '_map_' + k: v for k, v in dict(E=E, theta=th, psi=ps, I=I, Es=Es,
                                                          Ep=Ep).items()})
        # ---- the rays -----------------------------------------------------------------
        s = getattr(rs, cls)(raycing.BeamLine(azimuth=0.02), name=tag, center=(1., 2., 3.),
                             **args)
        np.random.seed(SEED)
        b = s.shine()
        for f in ('x', 'y', 'z', 'a', 'b', 'c', 'E', 'Jss', 'Jpp', 'Jsp', 'state'):
            out['%s_b_%s' % (tag, f)] = np.array(getattr(b, f))
        for f in ('Es', 'Ep'):
            out['%s_b_%s' % (tag, f)] = np.array(getattr(b, f))
        for f in ('accepted', 'acceptedE', 'seeded', 'seededI', 'sourceWeight'):
            if hasattr(b, f):
                out['%s_b_%s' % (tag, f)] = np.array(getattr(b, f))
        out[tag + '_Imax'] = np.array(s.Imax)
        out[tag + '_limits'] = np.array([s.Theta_min, s.Theta_max, s.Psi_min, s.Psi_max])
        print(tag, len(b.x), 'rays of', getattr(b, 'seeded', None), 'Imax %.4e' % s.Imax,
              'B %.4f T, rho %.4f m' % (s.B, s.ro))
    np.savez_compressed(os.path.join(OUT, 'g14_bend_sources.npz'), seed=np.int64(SEED), **out)


if __name__ == '__main__':
    main()
