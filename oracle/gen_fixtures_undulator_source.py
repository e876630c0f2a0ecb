"""TEST INFRASTRUCTURE ONLY — regenerates tests/golden/g10_undsrc_*.npz by
RUNNING THE REFERENCE (imported from /root/reference, build container only).

G10: outputs of the reference's `Undulator.shine`
(sources/sybase.py:1470-1810 with the numpy field integral, targetOpenCL=None)
for seeded numpy RNG:
  rays_planar    ray mode (rejection sampling), planar, explicit gNodes
  rays_helical   ray mode, Kx = Ky, energy spread, AUTOMATIC node convergence
  rays_taper     ray mode, tapered gap
  wave_filament  the configuration-4 source: filamentBeam + uniformRayDensity,
                 shine(fixedEnergy, wave=slit wave), automatic convergence
  wave_emittance wave mode with a finite-emittance electron beam
  wave_nf        wave mode, near field (R0)
Stored: constructor arguments (JSON), seed, the grid the reference converged
to, limits, the returned beam, and for wave modes the filled wave.

Run:  python -m oracle.gen_fixtures_undulator_source
"""
import json
import os
import numpy as np
from . import _refenv

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')
FIELDS = ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp', 'Es',
          'Ep', 'state')

RING = dict(eE=3.0, eI=0.5, eEpsilonX=0.263, eEpsilonZ=0.008, betaX=9., betaZ=2.)
CASES = {
    'rays_planar': dict(
        kw=dict(RING, nrays=4000, period=18.5, n=108, K=0.52, eMin=4000,
                eMax=4100, xPrimeMax=0.03, zPrimeMax=0.03, distE='BW', gNodes=24),
        seed=21, shine=dict()),
    'rays_helical': dict(
        kw=dict(RING, nrays=3000, period=40., n=40, Kx=0.8, Ky=0.8, phaseDeg=90,
                eEspread=8e-4, eMin=700, eMax=760, xPrimeMax=0.05,
                zPrimeMax=0.05, distE='eV'),
        seed=22, shine=dict(withAmplitudes=False)),
    'rays_taper': dict(
        kw=dict(RING, nrays=1500, period=18.5, n=20, K=1.1, taper=(0.4, 10.),
                eMin=2600, eMax=3100, xPrimeMax=0.03, zPrimeMax=0.03,
                distE='BW', gNodes=16, pitch=1e-5, yaw=-2e-5,
                center=(1., 2., 3.)),
        seed=23, shine=dict()),
    'wave_filament': dict(
        kw=dict(nrays=1500, period=29., n=172, eE=6.08, eI=0.1, eEpsilonX=0.,
                eEpsilonZ=0., betaX=1.2, betaZ=3.95, filamentBeam=True,
                uniformRayDensity=True, xPrimeMax=(0.2/44000.)*2e3,
                zPrimeMax=(0.2/44000.)*2e3, targetE=[7900., 3], eMin=7899.5,
                eMax=7900.5),
        seed=24, shine=dict(fixedEnergy=7900.), wave=(44000., 0.2, 1500)),
    'wave_emittance': dict(
        kw=dict(RING, nrays=1200, period=18.5, n=108, K=0.52, eMin=4060,
                eMax=4070, xPrimeMax=0.02, zPrimeMax=0.02, distE='BW',
                gNodes=20),
        seed=25, shine=dict(), wave=(30000., 0.3, 1200)),
    'wave_nf': dict(
        kw=dict(RING, nrays=800, period=18.5, n=20, K=1.1, R0=25000.,
                eMin=2790, eMax=2800, xPrimeMax=0.02, zPrimeMax=0.02,
                distE='BW', gNodes=16, filamentBeam=True,
                uniformRayDensity=True),
        seed=26, shine=dict(fixedEnergy=2795.), wave=(25000., 0.3, 800)),
}


def build(raycing, rs, ra, spec, extra=None):
    bl = raycing.BeamLine()
    kw = dict(spec['kw'])
    kw.update(extra or {})
    src = rs.Undulator(bl, 'und', **kw)
    wave = None
    if 'wave' in spec:
        dist, size, ns = spec['wave']
        slit = ra.RectangularAperture(
            bl, 'slit', [0, dist, 0], ('left', 'right', 'bottom', 'top'),
            [-size/2, size/2, -size/2, size/2])
        wave = slit.prepare_wave(src, ns)
    return src, wave


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.apertures as ra
    raycing._VERBOSITY_ = 0
    for tag, spec in CASES.items():
        np.random.seed(spec['seed'])
        src, wave = build(raycing, rs, ra, spec, dict(targetOpenCL=None))
        kwargs = dict(spec['shine'])
        if wave is not None:
            kwargs['wave'] = wave
        beam = src.shine(**kwargs)
        out = dict(ctor=json.dumps(spec['kw']), seed=np.int64(spec['seed']),
                   shine=json.dumps(spec['shine']),
                   wave_geom=np.array(spec.get('wave', (0, 0, 0)), dtype=float),
                   quadm=np.int64(src.quadm), gIntervals=np.int64(src.gIntervals),
                   limits=np.array([src.E_min, src.E_max, src.Theta_min,
                                    src.Theta_max, src.Psi_min, src.Psi_max]),
                   Kxy=np.array([src.Kx, src.Ky]), E1=np.float64(src.E1),
                   Imax=np.float64(src.Imax), xzE=np.float64(src.xzE),
                   dxdz=np.array([src.dx, src.dz, src.dxprime, src.dzprime]),
                   tg=src.tg, ag=src.ag)
        for k in ('accepted', 'acceptedE', 'seeded', 'seededI', 'sourceWeight'):
            out['b_' + k] = np.float64(getattr(beam, k))
        for f in FIELDS:
            if hasattr(beam, f):
                out['b_' + f] = np.array(getattr(beam, f))
        if wave is not None:
            for f in FIELDS:
                out['w_' + f] = np.array(getattr(wave, f))
            out['w_rDiffr'] = np.array(wave.rDiffr)
            out['w_xDiffr'] = np.array(wave.xDiffr)
            out['w_zDiffr'] = np.array(wave.zDiffr)
        path = os.path.join(OUT, 'g10_undsrc_%s.npz' % tag)
        np.savez_compressed(path, **out)
        print(tag, 'nodes', src.quadm, 'x', src.gIntervals, 'rays',
              len(beam.x), 'Imax %.4g' % src.Imax,
              os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
