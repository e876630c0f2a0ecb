"""GPU: xrt_amd's Undulator.shine (host sampling in the reference's RNG order +
ONE HIP launch for the field integral) against the reference's shine with its
numpy integral (golden G10, oracle/gen_fixtures_undulator_source.py).

Same seed -> same accepted rays: energies, positions and directions must be
bit-identical (they come from the host RNG; the rejection test compares the
device intensity, equal to the reference's to ~1e-15, with a random number).
Coherency matrix and amplitudes: norm-wise 1e-10 (north_star: 1e-5)."""
import numpy as np
import pytest

from undsrc_cases import load, build

pytestmark = pytest.mark.gpu
RAY_CASES = ['rays_planar', 'rays_helical', 'rays_taper']
WAVE_CASES = ['wave_filament', 'wave_emittance', 'wave_nf']
TOL = 1e-10


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / nb if nb > 0 else np.linalg.norm(a)


def shine(golden_dir, tag):
    g = load(golden_dir, tag)
    np.random.seed(int(g['seed']))
    src, wave, kw = build(g)
    if wave is not None:
        kw['wave'] = wave
    beam = src.shine(**kw)
    return g, src, wave, beam


@pytest.mark.parametrize('tag', RAY_CASES + WAVE_CASES)
def test_converged_grid_and_flux_bookkeeping(golden_dir, tag):
    g, src, wave, beam = shine(golden_dir, tag)
    assert (src.quadm, src.gIntervals) == (int(g['quadm']), int(g['gIntervals']))
    assert abs(src.Imax - float(g['Imax'])) <= 1e-12 * float(g['Imax'])
    assert src.xzE == float(g['xzE'])
    assert beam.seeded == int(g['b_seeded'])
    for k in ('accepted', 'acceptedE', 'seededI', 'sourceWeight'):
        ref = float(g['b_' + k])
        assert abs(getattr(beam, k) - ref) <= 1e-11 * abs(ref), k


@pytest.mark.parametrize('tag', RAY_CASES)
def test_ray_mode_matches_reference(golden_dir, tag):
    g, src, wave, beam = shine(golden_dir, tag)
    assert len(beam.x) == len(g['b_x'])
    assert np.array_equal(beam.state, g['b_state'])
    assert np.array_equal(beam.E, g['b_E'])          # same rays accepted
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path'):
        assert np.array_equal(getattr(beam, f), g['b_' + f]), f
    for f in ('Jss', 'Jpp', 'Jsp'):
        assert rel(getattr(beam, f), g['b_' + f]) < TOL, f
    assert np.allclose(beam.Jss + beam.Jpp, 1., rtol=0, atol=1e-12)
    if 'b_Es' in g.files and len(g['b_Es']) == len(beam.x):
        for f in ('Es', 'Ep'):
            assert rel(getattr(beam, f), g['b_' + f]) < TOL, f


@pytest.mark.parametrize('tag', WAVE_CASES)
def test_wave_mode_matches_reference(golden_dir, tag):
    g, src, wave, beam = shine(golden_dir, tag)
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E'):
        assert np.array_equal(getattr(beam, f), g['b_' + f]), 'beam ' + f
        assert np.array_equal(getattr(wave, f), g['w_' + f]), 'wave ' + f
    assert np.array_equal(wave.rDiffr, g['w_rDiffr'])
    assert np.array_equal(wave.state, g['w_state'])
    for f in ('Jss', 'Jpp', 'Jsp', 'Es', 'Ep'):
        assert rel(getattr(wave, f), g['w_' + f]) < TOL, 'wave ' + f
        assert rel(getattr(beam, f), g['b_' + f]) < TOL, 'beam ' + f


def test_device_map_agrees_with_host_map(golden_dir):
    """build_I_map (host arrays) and build_I_map_device (device tensors in and
    out, no host round trip) are the same launch."""
    import torch
    g = load(golden_dir, 'rays_planar')
    src, _, _ = build(g)
    src.reset()
    rng = np.random.RandomState(1)
    w = rng.uniform(src.E_min, src.E_max, 5000)
    th = rng.uniform(src.Theta_min, src.Theta_max, 5000)
    ps = rng.uniform(src.Psi_min, src.Psi_max, 5000)
    I, Es, Ep = src.build_I_map(w, th, ps)
    dI, dEs, dEp = src.build_I_map_device(torch.from_numpy(w).cuda(),
                                          torch.from_numpy(th).cuda(),
                                          torch.from_numpy(ps).cuda())
    assert dI.is_cuda and np.array_equal(dI.cpu().numpy(), I)
    assert np.array_equal(dEs.cpu().numpy(), Es)
    assert np.array_equal(dEp.cpu().numpy(), Ep)
    # harmonic window (synchr.py:2094-2098): only rays around harmonic 1 survive
    w = rng.uniform(1000., 9000., 5000)
    I, Es, Ep = src.build_I_map(w, th, ps)
    I1, Es1, _ = src.build_I_map(w, th, ps, harmonic=1)
    keep = Es1 != 0
    assert 0 < keep.sum() < len(w)
    assert np.array_equal(Es1[keep], Es[keep])
    assert np.all(I1[~keep] == 0)


@pytest.mark.parametrize('tag,ns,npx,seed', [('slit_2000x32', 2000, 32, 7),
                                             ('slit_4000x48', 4000, 48, 8)])
def test_configuration4_chain_undulator_slit_screen(golden_dir, tag, ns, npx, seed):
    """Configuration 4 end to end on this package's classes with the script of
    oracle/gen_fixtures_p2.py: Undulator.shine(wave=slit) [N3 kernel] ->
    waves.diffract onto a screen [P2 kernel], against what the reference
    produced for the same seed (golden G4: slit field and screen field)."""
    import os
    import xrt_amd.backends.raycing as raycing
    import xrt_amd.backends.raycing.sources as rs
    import xrt_amd.backends.raycing.screens as rsc
    import xrt_amd.backends.raycing.apertures as ra
    import xrt_amd.backends.raycing.waves as rw
    g = np.load(os.path.join(golden_dir, 'g4_%s.npz' % tag))
    R0, E0, slitD = 44000., 7900., 0.2
    np.random.seed(seed)
    bl = raycing.BeamLine()
    src = rs.Undulator(
        bl, nrays=ns, period=29., n=172, eE=6.08, eI=0.1, eEpsilonX=0.,
        eEpsilonZ=0., betaX=1.2, betaZ=3.95, filamentBeam=True,
        uniformRayDensity=True, xPrimeMax=(slitD/R0)*2e3,
        zPrimeMax=(slitD/R0)*2e3, targetE=[E0, 3], eMin=E0-0.5, eMax=E0+0.5)
    slit = ra.RectangularAperture(
        bl, 'slit', [0, R0, 0], ('left', 'right', 'bottom', 'top'),
        [-slitD/2, slitD/2, -slitD/2, slitD/2])
    scr = rsc.Screen(bl, 'scr', [0, R0 + 10000., 0])
    xm = np.linspace(-0.5, 0.5, npx)
    wscr = scr.prepare_wave(slit, xm, xm)
    wslit = slit.prepare_wave(src, ns)
    src.shine(fixedEnergy=E0, wave=wslit)
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'E'):
        assert np.array_equal(getattr(wslit, f), g['s_' + f]), 'slit ' + f
    for f in ('Jss', 'Jpp', 'Es', 'Ep'):
        assert rel(getattr(wslit, f), g['s_' + f]) < TOL, 'slit ' + f
    rw.diffract(wslit, wscr)
    for f in ('Jss', 'Jpp', 'Es', 'Ep'):
        assert rel(getattr(wscr, f), g['w_' + f]) < 1e-9, 'screen ' + f
    for f in ('a', 'b', 'c'):
        assert np.abs(getattr(wscr, f) - g['w_' + f]).max() < 1e-9, 'screen ' + f


def test_worker_threads_share_one_undulator(golden_dir):
    """What run_ray_tracing(threads=N) does to a source: N threads on their own streams call
    build_I_map of ONE Undulator at the same time, the first of them on a cold table cache
    (filled under a lock, keyed by device). Every thread gets the serial map, bit for bit, and
    the cache ends with one entry for this device."""
    import threading

    import torch
    g = load(golden_dir, 'rays_planar')
    src, _, _ = build(g)
    src.reset()
    rng = np.random.RandomState(2)
    args = [(rng.uniform(src.E_min, src.E_max, 20000),
             rng.uniform(src.Theta_min, src.Theta_max, 20000),
             rng.uniform(src.Psi_min, src.Psi_max, 20000)) for _ in range(4)]
    serial = [src.build_I_map(*a) for a in args]
    src.reset()                                   # cold cache again
    src._tables = {}
    got = [None] * len(args)
    errors = []

    def work(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(3):
                    got[i] = src.build_I_map(*args[i])
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(args))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    for s, t in zip(serial, got):
        for a, b in zip(s, t):
            assert np.array_equal(a, b)
    assert list(src._tables) == [str(torch.device('cuda', torch.cuda.current_device()))]
