"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g15_meshes.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only).

G15: the mesh functions of the sources (sources/sybase.py:676-932, synchr.py:1710-1785,
581-609): intensities_on_mesh (Stokes and vortex kinds, with harmonics, energy spread and
emittance), multi_electron_stack (seeded), tuning_curves and power_vs_K of an undulator;
intensities_on_mesh and power_vs_K of a wiggler; intensities_on_mesh of a bending magnet.

Run:  python -m oracle.gen_fixtures_meshes
"""
import os

import numpy as np

from . import _refenv

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')
UND = dict(nrays=1000, eE=3.0, eI=0.5, eEspread=8e-4, eEpsilonX=0.263, eEpsilonZ=0.008,
           betaX=9., betaZ=2., period=18.5, n=108, K=0.52, eMin=3900, eMax=4250,
           xPrimeMax=0.06, zPrimeMax=0.06, distE='BW', gNodes=24, gIntervals=2,
           xPrimeMaxAutoReduce=False, zPrimeMaxAutoReduce=False)
RING = dict(nrays=1000, eE=3.0, eI=0.5, eEpsilonX=0.263, eEpsilonZ=0.008, betaX=9.,
            betaZ=2., eMin=5000, eMax=15000, xPrimeMax=1.5, zPrimeMax=0.3, distE='eV')


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    raycing._VERBOSITY_ = 0
    out = {}
    E = np.linspace(3950., 4200., 9)
    th = np.linspace(-4e-5, 4e-5, 17)
    ps = np.linspace(-3e-5, 3e-5, 13)
    u = rs.Undulator(raycing.BeamLine(), name='u', targetOpenCL=None, **UND)
    for kind in ('Stokes', 'vortex'):
        for tag, h in (('', None), ('_h', [1, 3])):
            res = u.intensities_on_mesh(E, th, ps, h, eSpreadNSamples=8, resultKind=kind)
            for k, a in enumerate(res):
                out['und_%s%s_%d' % (kind, tag, k)] = np.array(a)
    np.random.seed(21)
    Es, Ep = u.multi_electron_stack(E, th, ps, [1, 3])
    out.update(und_stack_Es=Es, und_stack_Ep=Ep)
    auto = u.intensities_on_mesh()           # default meshes from eN, nx, nz
    out['und_auto_s0'] = np.array(auto[0][::6, ::5, ::5])    # a thinned copy + the shape
    out['und_auto_shape'] = np.array(auto[0].shape)
    Ks = [0.4, 0.6]
    tE, tF = u.tuning_curves(np.linspace(3000., 5000., 5), th, ps, [1], Ks)
    out.update(und_tune_E=tE, und_tune_F=tF, Ks=np.array(Ks))
    u0 = rs.Undulator(raycing.BeamLine(), name='u', targetOpenCL=None,
                      **dict(UND, eEspread=0))
    out['und_power'] = u0.power_vs_K(np.linspace(3000., 5000., 6), th, ps, [1, 3], Ks)
    thr, psr = np.linspace(-1e-3, 1e-3, 11), np.linspace(-2e-4, 2e-4, 9)
    Er = np.linspace(6000., 14000., 5)
    w = rs.Wiggler(raycing.BeamLine(), name='w', K=12., period=80., n=10, **RING)
    for k, a in enumerate(w.intensities_on_mesh(Er, thr, psr)):
        out['wig_Stokes_%d' % k] = np.array(a)
    out['wig_power'] = w.power_vs_K(Er, thr, psr, [8., 12.])
    b = rs.BendingMagnet(raycing.BeamLine(), name='b', B0=1.7, **RING)
    for k, a in enumerate(b.intensities_on_mesh(Er, thr, psr)):
        out['bm_Stokes_%d' % k] = np.array(a)
    out.update(E=E, theta=th, psi=ps, Er=Er, thetar=thr, psir=psr)
    np.savez_compressed(os.path.join(OUT, 'g15_meshes.npz'), **out)
    print({k: v.shape for k, v in out.items() if k.endswith('_0') or 'tune' in k
           or 'power' in k or 'auto' in k})


if __name__ == '__main__':
    main()
