"""GPU: mesh functions of the sources (intensities_on_mesh, multi_electron_stack,
tuning_curves, power_vs_K) against the reference's (golden G15,
oracle/gen_fixtures_meshes.py)."""
import os

import numpy as np
import pytest

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.sources as rs

pytestmark = pytest.mark.gpu

UND = dict(nrays=1000, eE=3.0, eI=0.5, eEspread=8e-4, eEpsilonX=0.263, eEpsilonZ=0.008,
           betaX=9., betaZ=2., period=18.5, n=108, K=0.52, eMin=3900, eMax=4250,
           xPrimeMax=0.06, zPrimeMax=0.06, distE='BW', gNodes=24, gIntervals=2,
           xPrimeMaxAutoReduce=False, zPrimeMaxAutoReduce=False)
RING = dict(nrays=1000, eE=3.0, eI=0.5, eEpsilonX=0.263, eEpsilonZ=0.008, betaX=9.,
            betaZ=2., eMin=5000, eMax=15000, xPrimeMax=1.5, zPrimeMax=0.3, distE='eV')


def close(mine, ref, tol=1e-9):
    ref = np.asarray(ref)
    scale = np.abs(ref).max()
    assert np.shape(mine) == ref.shape
    assert np.abs(np.asarray(mine) - ref).max() <= tol * scale, \
        np.abs(np.asarray(mine) - ref).max() / scale


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'g15_meshes.npz'))


@pytest.mark.parametrize('kind', ['Stokes', 'vortex'])
@pytest.mark.parametrize('with_harmonics', [False, True])
def test_undulator_intensities_on_mesh(g, kind, with_harmonics):
    u = rs.Undulator(raycing.BeamLine(), name='u', **UND)
    res = u.intensities_on_mesh(g['E'], g['theta'], g['psi'], [1, 3] if with_harmonics else None,
                                eSpreadNSamples=8, resultKind=kind)
    tag = '_h' if with_harmonics else ''
    assert len(res) == (4 if kind == 'Stokes' else 6)
    for k, a in enumerate(res):
        ref = g['und_%s%s_%d' % (kind, tag, k)]
        if kind == 'Stokes' and k > 0:
            # normalised Stokes parameters: compared where there is light
            lit = g['und_%s%s_0' % (kind, tag)] > 1e-6 * g['und_%s%s_0' % (kind, tag)].max()
            assert np.abs(a - ref)[lit].max() < 1e-7
        elif kind == 'vortex' and k in (2, 3):
            # orbital angular momentum density Re(E* i (dE/dtheta psi - dE/dpsi theta)): for
            # this planar device the phase fronts are flat and the quantity is ~1e-8 of
            # its natural scale |E|^2 psi / dtheta -- rounding noise of the field in the
            # reference as well; compared on that scale
            natural = g['und_vortex%s_%d' % (tag, k - 2)].max() * \
                np.abs(g['psi']).max() / (g['theta'][1] - g['theta'][0])
            assert np.abs(ref).max() < 1e-6 * natural
            assert np.abs(a - ref).max() < 1e-7 * natural
        else:
            close(a, ref)
    if with_harmonics and kind == 'Stokes':
        assert res[0][..., 0].max() > 0 and res[0].shape[-1] == 2


def test_undulator_default_meshes_stack_and_curves(g):
    u = rs.Undulator(raycing.BeamLine(), name='u', **UND)
    auto = u.intensities_on_mesh()[0]
    assert list(auto.shape) == list(g['und_auto_shape'])
    close(auto[::6, ::5, ::5], g['und_auto_s0'])
    np.random.seed(21)
    Es, Ep = u.multi_electron_stack(g['E'], g['theta'], g['psi'], [1, 3])
    close(Es, g['und_stack_Es'])
    close(Ep, g['und_stack_Ep'])
    tE, tF = u.tuning_curves(np.linspace(3000., 5000., 5), g['theta'], g['psi'], [1], list(g['Ks']))
    assert np.array_equal(tE, g['und_tune_E'])
    close(tF, g['und_tune_F'])
    assert u.Ky == 0.52
    u0 = rs.Undulator(raycing.BeamLine(), name='u', **dict(UND, eEspread=0))
    close(u0.power_vs_K(np.linspace(3000., 5000., 6), g['theta'], g['psi'], [1, 3], list(g['Ks'])),
          g['und_power'])


def test_wiggler_and_magnet_meshes(g):
    w = rs.Wiggler(raycing.BeamLine(), name='w', K=12., period=80., n=10, **RING)
    for k, a in enumerate(w.intensities_on_mesh(g['Er'], g['thetar'], g['psir'])):
        close(a, g['wig_Stokes_%d' % k], 1e-9 if k == 0 else 1e-7)
    close(w.power_vs_K(g['Er'], g['thetar'], g['psir'], [8., 12.]), g['wig_power'])
    assert w.K == 12.
    b = rs.BendingMagnet(raycing.BeamLine(), name='b', B0=1.7, **RING)
    for k, a in enumerate(b.intensities_on_mesh(g['Er'], g['thetar'], g['psir'])):
        close(a, g['bm_Stokes_%d' % k], 1e-9 if k == 0 else 1e-7)
