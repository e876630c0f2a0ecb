"""GPU: bending-magnet and wiggler sources (reference sources/synchr.py:69-610) against the
reference's own map and seeded shine() (golden G14, oracle/gen_fixtures_bend_source.py), and
the device Bessel functions against scipy."""
import os

import numpy as np
import pytest
import torch
from scipy import special

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.sources as rs
from xrt_amd import hipcalls

pytestmark = pytest.mark.gpu

COMMON = dict(nrays=1500, eE=3.0, eI=0.5, eEpsilonX=0.263, eEpsilonZ=0.008, betaX=9.,
              betaZ=2., eMin=5000, eMax=15000, xPrimeMax=1.5, zPrimeMax=0.3, distE='BW')
CASES = {
    'bm_field': ('BendingMagnet', dict(B0=1.7)),
    'bm_filament': ('BendingMagnet', dict(rho=5.9, filamentBeam=True)),
    'bm_uniform': ('BendingMagnet', dict(B0=1.7, eEspread=1e-3, uniformRayDensity=True,
                                         distE='eV')),
    'wiggler': ('Wiggler', dict(K=12., period=80., n=10, pitch=1e-4, yaw=-2e-4)),
    'wiggler_spread': ('Wiggler', dict(K=12., period=80., n=10, eEspread=1e-3,
                                       xPrimeMax=3.)),
    'wiggler_filament': ('Wiggler', dict(K=12., period=80., n=10, filamentBeam=True)),
}


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def make(tag):
    cls, kw = CASES[tag]
    return getattr(rs, cls)(raycing.BeamLine(azimuth=0.02), name=tag, center=(1., 2., 3.),
                            **dict(COMMON, **kw))


def test_bessel_k_of_order_one_and_two_thirds():
    x = np.concatenate([np.logspace(-12, np.log10(2.), 4000, endpoint=False),
                        np.linspace(2., 40., 4000), np.logspace(np.log10(40.), np.log10(700.), 500),
                        [2. - 1e-15, 2., 1e-300, 704.9]])
    k13, k23 = hipcalls.debug_bessel_k(torch.from_numpy(x).cuda())
    for mine, order in ((k13, 1./3.), (k23, 2./3.)):
        ref = special.kv(order, x)
        ok = np.isfinite(ref) & (ref > 1e-300)
        err = np.abs(mine.cpu().numpy()[ok] - ref[ok]) / ref[ok]
        worst = np.argmax(err)
        print('K_%.3f: max relative error %.1e at x = %.3g' % (order, err.max(), x[ok][worst]))
        assert err.max() < 2e-13      # scipy (AMOS) and this agree to a few 1e-14
    edge = torch.tensor([0., 800., float('inf')], dtype=torch.float64).cuda()
    k13, k23 = (t.cpu().numpy() for t in hipcalls.debug_bessel_k(edge))
    assert np.isinf(k13[0]) and k13[1] == 0. and k23[2] == 0.


@pytest.mark.parametrize('tag', list(CASES))
def test_map_matches_reference(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, 'g14_bend_sources.npz'))
    src = make(tag)
    src.reset()
    assert np.array_equal([src.Theta_min, src.Theta_max, src.Psi_min, src.Psi_max],
                          g[tag + '_limits'])
    np.random.seed(int(g['seed']) + 1)
    I, Es, Ep = src.build_I_map(g[tag + '_map_E'], g[tag + '_map_theta'], g[tag + '_map_psi'])
    assert rel(I, g[tag + '_map_I']) < 1e-12
    assert rel(Es, g[tag + '_map_Es']) < 1e-12 and rel(Ep, g[tag + '_map_Ep']) < 1e-12
    assert np.abs(Es.real).max() == 0. and np.abs(Ep.imag).max() == 0.


@pytest.mark.parametrize('tag', list(CASES))
def test_shine_returns_the_references_rays(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, 'g14_bend_sources.npz'))
    src = make(tag)
    np.random.seed(int(g['seed']))
    beam = src.shine()
    key = tag + '_b_'
    assert len(beam.x) == len(g[key + 'x'])
    assert np.array_equal(beam.E, g[key + 'E'])                  # the same rays accepted
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'state'):
        assert np.array_equal(getattr(beam, f), g[key + f]), f
    for f in ('Jss', 'Jpp'):
        assert rel(getattr(beam, f), g[key + f]) < 1e-11, f
    assert np.abs(beam.Jsp - g[key + 'Jsp']).max() < 1e-11 * max(np.abs(g[key + 'Jss']).max(), 1.)
    assert abs(src.Imax - float(g[tag + '_Imax'])) <= 1e-12 * float(g[tag + '_Imax'])
    if len(g[key + 'Es']) == len(beam.x) and (key + 'seeded') in g.files and \
            int(g[key + 'seeded']) <= int(np.int64(COMMON['nrays'] * 1.2)):
        for f in ('Es', 'Ep'):          # one batch: the reference's amplitudes are complete
            assert rel(getattr(beam, f), g[key + f]) < 1e-11, f
    for k in ('accepted', 'acceptedE', 'seeded', 'seededI', 'sourceWeight'):
        if (key + k) in g.files:
            ref = float(g[key + k])
            assert abs(getattr(beam, k) - ref) <= 1e-11 * abs(ref), k
        else:
            assert not hasattr(beam, k), k
