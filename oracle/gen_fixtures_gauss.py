"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g16_gaussian_beams.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only): GaussianBeam,
LaguerreGaussianBeam and HermiteGaussianBeam .shine(wave=...) (sources/geoms.py:538-850) on
the meshes of screens at the waist and 5 m / 40 m downstream, as in the reference's
tests/raycing/laguerre_hermite_gaussian_beam.py: plain, astigmatic (two waists), vortex
(l, p) = (1, 1) and (-2, 0), TEM (2, 1) with an astigmatic waist, a tilted source with a
total flux.

Run:  python -m oracle.gen_fixtures_gauss
"""
import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1

CASES = (
    ('plain', 'GaussianBeam', dict(w0=15e-3, energies=(9000.,)), 5000.),
    ('waist', 'GaussianBeam', dict(w0=15e-3, energies=(9000.,)), 0.),
    ('astig', 'GaussianBeam', dict(w0=(15e-3, 8e-3), energies=(8000.,), polarization='v'),
     40000.),
    ('lg11', 'LaguerreGaussianBeam', dict(w0=15e-3, vortex=(1, 1), energies=(9000.,)), 5000.),
    ('lg20', 'LaguerreGaussianBeam', dict(w0=10e-3, vortex=(-2, 0), energies=(7000.,),
                                          polarization='r'), 40000.),
    ('hg21', 'HermiteGaussianBeam', dict(w0=(15e-3, 12e-3), TEM=(2, 1), energies=(9000.,)),
     5000.),
    ('tilted', 'GaussianBeam', dict(w0=20e-3, energies=(9000., 9010.), distE='flat',
                                    pitch=1e-5, yaw=-2e-5, totalFlux=1e12,
                                    center=(0.1, 3., -0.2)), 20000.),
)


def mesh(dist, w0, E):
    w0 = np.max(w0)
    yR = E / 1973.2697177417986 * 1e7 / 2 * w0**2
    half = 3 * w0 * (1 + (dist/yR)**2)**0.5
    return np.linspace(-half, half, 24), np.linspace(-half*0.8, half*0.8, 20)


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.screens as rsc
    out = {}
    for seed, (tag, cls, kw, dist) in enumerate(CASES):
        bl = raycing.BeamLine(azimuth=0.01)
        src = getattr(rs, cls)(bl, tag, **kw)
        scr = rsc.Screen(bl, 'fsm', [np.sin(0.01)*dist, np.cos(0.01)*dist, 0])
        x, z = mesh(dist, kw['w0'], kw['energies'][0])
        wave = scr.prepare_wave(src, x, z)
        np.random.seed(200 + seed)
        bo = src.shine(wave=wave)
        flux = (wave.Jss + wave.Jpp).sum()
        print(tag, 'flux on the mesh', flux, 'max |Es|', np.abs(wave.Es).max())
        out.update(g1.beam_dict(tag + '_bo_', bo))
        out.update({tag + '_x': x, tag + '_z': z, tag + '_dist': np.array(dist),
                    tag + '_seed': np.array(200 + seed),
                    tag + '_wave_abc': np.array([wave.a, wave.b, wave.c]),
                    tag + '_wave_Es': np.array(wave.Es), tag + '_wave_Ep': np.array(wave.Ep),
                    tag + '_wave_J': np.array([wave.Jss, wave.Jpp])})
        if hasattr(wave, 'sourceWeight'):
            out[tag + '_sourceWeight'] = np.array(wave.sourceWeight)
    out['azimuth'] = np.array(0.01)
    g1.save('g16_gaussian_beams', **out)


if __name__ == '__main__':
    main()
