"""GPU: the source defined by a tabulated magnetic field (reference SourceFromField,
sources/synchr.py:612-1347): the trajectory kernel against the reference's Runge-Kutta
tables, the class against the reference's seeded shine() (golden G13, made by running the
reference: oracle/gen_fixtures_field_source.py)."""
import os
import time

import numpy as np
import pytest
import torch

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.sources as rs
from xrt_amd import hipcalls

pytestmark = pytest.mark.gpu

SOURCE = dict(nrays=400, eE=3.0, eI=0.5, eEspread=0, eEpsilonX=0.263, eEpsilonZ=0.008,
              betaX=9., betaZ=2., eMin=1500, eMax=1700, xPrimeMax=0.1, zPrimeMax=0.1,
              distE='BW', gNodes=40, gIntervals=20)


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize('tag', ['plain', 'filament'])
def test_trajectory_kernel_matches_reference_tables(golden_dir, tag):
    """Runge-Kutta in the reference's operation order: the tables on the grid agree to
    the last bits (cubes come from a product here and from pow() in numpy)."""
    g = np.load(os.path.join(golden_dir, 'g13_trajectory_%s.npz' % tag))
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    gamma = float(g['gamma'])
    kw = dict(gamma=gamma, emcg=1.602176565e-19 / 9.109383701528e-31 / 2.99792458e10 / 10.
              / gamma) if int(g['filament']) else {}
    out = hipcalls.trajectory(up(g['wtGrid']), up(g['Bx']), up(g['By']), up(g['Bz']), **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = hipcalls.trajectory(up(g['wtGrid']), up(g['Bx']), up(g['By']), up(g['Bz']), **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    exact = 0
    for t, name in zip(out[:5], ('betax', 'betay', 'trajx', 'trajy', 'trajz')):
        mine = t.cpu().numpy()
        scale = np.abs(g[name]).max()
        assert np.abs(mine - g[name]).max() <= 1e-13 * scale, name
        exact += int(np.array_equal(mine, g[name]))
    assert abs(float(out[5][0]) - float(g['betam'])) <= 1e-15 * abs(float(g['betam']))
    print('%s: %d grid points in %.2f ms (the reference loop: %.0f ms); %d of 5 tables '
          'bit-identical' % (tag, len(g['wtGrid']), ms, 1e3 * float(g['reference_seconds']),
                             exact))


def make_source(g, **kw):
    return rs.SourceFromField(raycing.BeamLine(), 'sff', customField=np.array(g['field']),
                              **dict(SOURCE, **kw))


@pytest.mark.parametrize('tag', ['plain', 'filament'])
def test_trajectory_on_the_integration_nodes(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, 'g13_trajectory_%s.npz' % tag))
    rays = np.load(os.path.join(golden_dir, 'g13_sff_%s.npz' % (
        'rays' if tag == 'plain' else tag)))
    src = make_source(rays, filamentBeam=(tag == 'filament'))
    src._reset_limits()
    src._build_integration_grid()
    assert np.array_equal(src.tg, g['tg'])
    Bx, By, Bz = src._magnetic_field()
    assert np.array_equal(src.wtGrid, g['wtGrid']) and np.array_equal(By, g['By'])
    betax, betay, betazav, trajx, trajy, trajz = src.build_trajectory(Bx, By, Bz)
    for mine, name in ((betax, 'betax_tg'), (betay, 'betay_tg'), (trajx, 'trajx_tg'),
                       (trajy, 'trajy_tg'), (trajz, 'trajz_tg')):
        assert np.abs(mine - g[name]).max() <= 1e-12 * np.abs(g[name]).max(), name
    assert abs(betazav[-1] - float(g['betam'])) <= 1e-15 * abs(float(g['betam']))


@pytest.mark.parametrize('tag', ['rays', 'filament'])
def test_shine_returns_the_references_rays(golden_dir, tag):
    """Same numpy seed -> the same accepted rays (energies, positions, directions from
    the host generator: bit-identical), polarisation to 1e-9 norm-wise."""
    g = np.load(os.path.join(golden_dir, 'g13_sff_%s.npz' % tag))
    src = make_source(g, filamentBeam=(tag == 'filament'))
    np.random.seed(int(g['seed']))
    t0 = time.perf_counter()
    beam = src.shine()
    seconds = time.perf_counter() - t0
    assert len(beam.x) == len(g['beam_x']) and beam.seeded == int(g['beam_seeded'])
    assert abs(src.Imax - float(g['Imax'])) <= 1e-10 * float(g['Imax'])
    assert np.array_equal(beam.E, g['beam_E'])
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'state'):
        assert np.array_equal(getattr(beam, f), g['beam_' + f]), f
    for f in ('Jss', 'Jpp', 'Jsp', 'Es', 'Ep'):
        assert rel(getattr(beam, f), g['beam_' + f]) < 1e-9, f
    for k in ('accepted', 'acceptedE', 'seededI', 'sourceWeight'):
        ref = float(g['beam_' + k])
        assert abs(getattr(beam, k) - ref) <= 1e-10 * abs(ref), k
    print('shine(%s): %.2f s (the reference: %.1f s)' % (tag, seconds,
                                                         float(g['reference_seconds'])))


def test_requests_outside_the_mirrored_part_fail_loudly(golden_dir):
    with pytest.raises(NotImplementedError):
        rs.SourceFromField(raycing.BeamLine(), 'a', **SOURCE)


@pytest.mark.parametrize('tag', ['plain', 'filament'])
def test_automatic_number_of_nodes(golden_dir, tag):
    """gNodes=None: doubling + bisection until the probe ray's field is stable to *gp*; the
    probe calls (ten rays or fewer) use the carrier of the reference's vectorised form."""
    g = np.load(os.path.join(golden_dir, 'g13_nodes.npz'))
    cfg = dict(SOURCE, gp=1e-6, gIntervals=6, filamentBeam=(tag == 'filament'))
    cfg.pop('gNodes')
    src = rs.SourceFromField(raycing.BeamLine(), 'sff', customField=np.array(g['field']), **cfg)
    np.random.seed(5)
    t0 = time.perf_counter()
    src.reset()
    seconds = time.perf_counter() - t0
    assert src.quadm == int(g[tag + '_quadm'])
    src.convergenceSearchFlag = True
    probe = src.build_I_map(src.E_max * np.ones(1), src.Theta_max * np.ones(1),
                            src.Psi_max * np.ones(1))
    src.convergenceSearchFlag = False
    assert abs(probe[0] - g[tag + '_probe'][0]) <= 1e-10 * g[tag + '_probe'][0]
    print('node search (%s): %d nodes in %.2f s (the reference: %.1f s)' % (
        tag, src.quadm, seconds, float(g[tag + '_seconds'])))


@pytest.mark.parametrize('tag', ['plain', 'filament'])
def test_get_trajectory_through_the_dropin_object(golden_dir, tag):
    """run_parallel('get_trajectory' | 'get_trajectory_filament', ...) with the arguments
    SourceFromField._build_trajectory_CL passes (synchr.py:1011-1035)."""
    from xrt_amd.backends.raycing.myhip import XRT_HIP
    g = np.load(os.path.join(golden_dir, 'g13_trajectory_%s.npz' % tag))
    n = len(g['wtGrid'])
    scalars = [np.int32(n)] + ([np.float64(g['gamma'])] if tag == 'filament' else [])
    rw = [np.zeros(n) for _ in range(6)]
    name = 'get_trajectory' + ('_filament' if tag == 'filament' else '')
    betax, betay, betazav, trajx, trajy, trajz = XRT_HIP().run_parallel(
        name, scalars, None, [g['wtGrid'], g['Bx'], g['By'], g['Bz']], None, rw, 1)
    assert np.array_equal(betax, g['betax']) and np.array_equal(trajz, g['trajz'])
    assert betazav[-1] == float(g['betam']) and np.array_equal(rw[3], g['trajx'])


def test_shine_onto_a_wave_takes_the_host_map(golden_dir):
    """a wave from prepare_wave holds its points on the GPU; ``shine(wave=...)`` of a
    source WITHOUT a device intensity map (SourceFromField inherits Undulator.shine) must take
    the host path instead of the undulator's device path -- it raised NotImplementedError. The
    same wave with its points pulled to the host first gives the same field."""
    import xrt_amd.backends.raycing.apertures as ra
    g = np.load(os.path.join(golden_dir, 'g13_sff_rays.npz'))
    src = make_source(g)
    assert not src._map_on_device()
    slit = ra.RectangularAperture(src.bl, 'slit', [0, 20000., 0], ('left', 'right', 'bottom', 'top'),
                                  [-1., 1., -1., 1.])
    results = []
    for pull in (False, True):
        np.random.seed(4)
        wave = slit.prepare_wave(src, 500)
        assert 'xDiffr' in wave._d
        if pull:
            for f in ('xDiffr', 'yDiffr', 'zDiffr'):
                getattr(wave, f)                       # host copies become the masters
        np.random.seed(5)
        out = src.shine(wave=wave)
        results.append((np.array(out.Es), np.array(out.Ep), np.array(out.E)))
    for a, b in zip(*results):
        assert np.array_equal(a, b)
    assert np.abs(results[0][0]).max() > 0
