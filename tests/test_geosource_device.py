"""SURVEY 8 row a2 on the device: ``GeometricSource(rng='device')`` (csrc/source.hip) against its
CPU restatement oracle/geosource_np.py (bit-exact integers and uniform laws, a few ulp on the
transcendental ones) and, as SURVEY 8c prescribes for a2, against the distribution moments of
the host path that reproduces the reference's rays (tests/test_source_screen.py, golden G1).
CPU part: the Philox4x32-10 known-answer vectors, the oracle's own moments, and the
host-side record the kernel is launched with."""
import numpy as np
import pytest

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.sources as rs
from oracle import geosource_np as og

GEOM = ('x', 'y', 'z', 'a', 'b', 'c')


def spec_of(src, call=0, withAmplitudes=False, toGlobal=True):
    """The oracle's Spec of a product source object."""
    bl = src.bl
    return og.Spec(
        src.nrays, seed=int(src.seed), call=call,
        distx=src.distx, dx=src.dx, disty=src.disty, dy=src.dy, distz=src.distz, dz=src.dz,
        distxprime=src.distxprime, dxprime=src.dxprime, distzprime=src.distzprime,
        dzprime=src.dzprime, distE=src.distE, energies=src.energies,
        energyWeights=src.energyWeights, polarization=rs._polarization_state(src.polarization),
        filamentBeam=src.filamentBeam, uniformRayDensity=src.uniformRayDensity,
        withAmplitudes=withAmplitudes,
        steps=raycing.rotation_steps(pitch=src.pitch, roll=src.roll, yaw=src.yaw),
        azimuth=(bl.cosAzimuth, bl.sinAzimuth) if toGlobal else None,
        center=src.center if toGlobal else None)


CASES = {
    'default': dict(),
    'flat_rotated': dict(distx='flat', dx=2., distz='flat', dz=(-0.1, 0.3), disty='flat', dy=5.,
                         distxprime='flat', dxprime=2e-3, distzprime='normal', dzprime=1e-4,
                         distE='flat', energies=(8000., 8100.), polarization='v',
                         pitch=0.01, roll=-0.2, yaw=0.03, center=(1., 2000., -3.)),
    'annulus': dict(distx='annulus', dx=(0.5, 1.5), distz='annulus', dz=(0.3, 2.5),
                    distxprime='annulus', dxprime=(1e-4, 2e-4), distzprime='annulus',
                    dzprime=1e-4, distE='normal', energies=(9000., 2.), polarization='r'),
    'ring_line': dict(distx='annulus', dx=(1.5, 1.5), distz='annulus', dz=0.1,
                      distE='lines', energies=(8000., 9000., 10000.),
                      energyWeights=(0.2, 0.5, 0.3), polarization='+45'),
    'uniform_density': dict(uniformRayDensity=True, dx=0.3, dz=(0.02, 0.05), disty='normal',
                            dy=1.5, dxprime=1e-3, dzprime=(1e-4, 3e-4), polarization=None,
                            distE='lines', energies=(7000., 0., 7100.)),
    'slopes': dict(distxprime='flat', dxprime=3., distzprime='normal', dzprime=0.2,
                   polarization=(0.7, 0.3, 0.1, -0.2), filamentBeam=True, distE='flat',
                   energies=(5000., 5100.)),
    'no_energy_law': dict(distE=None, distx=None, distz='flat', dz=-1., polarization='30'),
}


def make(case, n, azimuth=0.3, seed=11, **kw):
    bl = raycing.BeamLine(azimuth=azimuth)
    args = dict(CASES[case])
    args.update(kw)
    return rs.GeometricSource(bl, 'src', nrays=n, rng='device', seed=seed, **args)


# --------------------------------------------------------------------------- CPU
def test_philox_known_answers():
    """Random123 v1.14 kat_vectors, philox4x32-10."""
    kat = (((0, 0, 0, 0), (0, 0), '6627e8d5 e169c58d bc57ac4c 9b00dbd8'),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), '408f276d 41c83b0e a20bc7c6 6d5451fd'),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            'd16cfe09 94fdcceb 5001e420 24126ea1'))
    for ctr, key, want in kat:
        got = og.philox4x32_10(*[np.array([v], dtype=np.uint64) for v in ctr], *key)
        assert ' '.join('%08x' % int(v[0]) for v in got) == want


def _moments_ok(v, mean, sigma, n, what):
    assert abs(v.mean() - mean) < 5 * sigma / np.sqrt(n), what
    assert abs(v.std() - sigma) < 5 * sigma / np.sqrt(2 * n), what


def test_oracle_moments_and_independence():
    n = 400_000
    src = make('default', n)
    o = og.shine(spec_of(src, toGlobal=False))
    for f, sigma in (('x', 0.32), ('z', 0.018), ('a', 1e-3), ('c', 1e-4)):
        _moments_ok(o[f], 0., sigma, n, f)
    for p, q in (('x', 'z'), ('a', 'c'), ('x', 'a'), ('z', 'c')):
        assert abs(np.corrcoef(o[p], o[q])[0, 1]) < 5 / np.sqrt(n), (p, q)
    assert np.array_equal(o['b'], (1 - (o['a']**2 + o['c']**2))**0.5)
    assert (o['E'] == 9000.).all() and (o['state'] == 1).all() and not o['y'].any()
    # two calls and two seeds are different streams; the same call is the same rays
    again = og.shine(spec_of(src, toGlobal=False))
    other = og.shine(spec_of(src, call=1, toGlobal=False))
    assert np.array_equal(again['x'], o['x']) and not np.array_equal(other['x'], o['x'])
    assert abs(np.corrcoef(other['x'], o['x'])[0, 1]) < 5 / np.sqrt(n)


def test_oracle_against_the_host_path_that_is_the_reference():
    """Same source through rng='host' (bit-identical to the reference, golden G1) and through
    the device laws: the moments of every coordinate agree within the sampling error."""
    n = 200_000
    for case in ('default', 'flat_rotated', 'annulus', 'uniform_density'):
        src = make(case, n)
        o = og.shine(spec_of(src, withAmplitudes=True))
        np.random.seed(3)
        src.rng = 'host'
        h = src.shine(withAmplitudes=True)
        for f in GEOM + ('E', 'Jss', 'Jpp'):
            a, b = o[f], getattr(h, f)
            spread = max(a.std(), b.std())
            assert abs(a.mean() - b.mean()) <= 7 * spread / np.sqrt(n) + 1e-15 * abs(b.mean()), \
                (case, f)
            assert abs(a.std() - b.std()) <= 7 * spread / np.sqrt(n) * max(
                1., np.abs(a - a.mean()).max() / max(spread, 1e-300) / 3), (case, f)
        assert np.allclose(o['a']**2 + o['b']**2 + o['c']**2, 1, atol=1e-15)
        if case != 'uniform_density':
            assert np.array_equal(o['Jsp'], h.Jsp) and np.array_equal(o['Es'], h.Es)


def test_device_record_of_a_source():
    from xrt_amd import _structs
    src = make('flat_rotated', 10)
    g, reach2 = src.device_spec()
    assert list(g.law) == [_structs.LAW_FLAT] * 4 + [_structs.LAW_NORMAL]
    assert (g.p0[1], g.p1[1], g.p0[2], g.p1[2]) == (-1., 1., -0.1, 0.3)
    assert g.e_law == 2 and (g.e_p0, g.e_p1) == (8000., 8100.)
    assert g.rot.n == 3 and list(g.rot.axis)[:3] == [2, 1, 0] and g.to_global == 1
    assert (g.Jss, g.Jpp) == (rs._polarization_state('v')[0], 1.) and reach2 < 1
    g, reach2 = make('slopes', 10).device_spec()
    assert reach2 > 1 and g.filament == 1
    g, _ = make('ring_line', 10).device_spec()
    assert g.annulus_xz == 1 and g.annulus_ac == 0 and g.n_lines == 3
    assert np.allclose(list(g.e_cdf)[:3], (0.2, 0.7, 1.0)) and g.e_cdf[2] == 1.
    with pytest.raises(ValueError):
        make('default', 10, distE='lines', energies=tuple(range(1, 40))).device_spec()
    assert make('default', 10, seed=None).device_spec()[0].seed >= 0
    with pytest.raises(ValueError):
        rs.GeometricSource(rng='gpu')


# --------------------------------------------------------------------------- GPU
def _close(got, want, ulps, what):
    # norm-wise: a Gaussian deviate radius * cos(2 pi u) near a zero of the cosine is tiny, and
    # numpy rounds the angle 2 pi u before taking the cosine (an absolute error of ~1e-16 of
    # the radius), the kernel reduces u exactly -- what matters is the error against sigma
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
    assert err <= ulps * 2.3e-16, (what, err)


@pytest.mark.gpu
@pytest.mark.parametrize('case', sorted(CASES))
def test_device_rays_are_the_oracles(case):
    n = 50_021
    amp = case in ('annulus', 'uniform_density', 'ring_line')
    src = make(case, n)
    for call in range(2):
        want = og.shine(spec_of(src, call=call, withAmplitudes=amp))
        b = src.shine(withAmplitudes=amp)
        assert b.nrays == n and not b._h, 'the beam is born on the device'
        assert np.array_equal(b.peek('state'), want['state'])
        assert b.has_amplitudes() == ('Es' in want)
        for f in sorted(set(want) - {'state'}):
            got = b.peek(f)
            if case in ('flat_rotated', 'slopes', 'no_energy_law') and f in ('E', 'Jss', 'Jpp',
                                                                             'Jsp', 'path'):
                assert np.array_equal(got, want[f]), (case, f)     # no transcendental on the way
            # numpy's and the GPU library's log / sin / cos / exp agree to an ulp or two; a
            # rotation or a cancellation (b = sqrt(1 - ac)) carries that on
            _close(got, want[f], 64 if f in GEOM else 8, (case, f))
    assert b.parentId == src.uuid


@pytest.mark.gpu
def test_device_uniform_laws_are_bit_exact():
    """Flat laws use the same two IEEE operations on the same 53-bit uniforms."""
    src = make('flat_rotated', 100_003, azimuth=0., pitch=0, roll=0, yaw=0, center=(0, 0, 0),
               distzprime='flat', dzprime=1e-3)
    want = og.shine(spec_of(src))
    b = src.shine()
    for f in GEOM + ('E',):
        assert np.array_equal(b.peek(f), want[f]), f


@pytest.mark.gpu
def test_device_moments_on_1e7_rays():
    """mean, sigma and correlation of x, z, a, c within 5 sigma / sqrt(N);
    b^2 = 1 - a^2 - c^2 as the reference forms it."""
    import torch
    n = 10_000_000
    src = make('default', n, azimuth=0., seed=2024)
    b = src.shine(toGlobal=False)
    d = {f: b.dev(f) for f in GEOM}
    for f, sigma in (('x', 0.32), ('z', 0.018), ('a', 1e-3), ('c', 1e-4)):
        assert abs(float(d[f].mean())) < 5 * sigma / np.sqrt(n), f
        assert abs(float(d[f].std()) - sigma) < 5 * sigma / np.sqrt(2 * n), f
        # fourth moment of a Gaussian: kurtosis 3 +- 5 sqrt(96 / n)
        k = float(((d[f] / sigma)**4).mean())
        assert abs(k - 3.) < 5 * np.sqrt(96. / n), (f, k)
    for p, q in (('x', 'z'), ('a', 'c'), ('x', 'a'), ('z', 'c'), ('x', 'c'), ('z', 'a')):
        r = float(torch.corrcoef(torch.stack((d[p], d[q])))[0, 1])
        assert abs(r) < 5 / np.sqrt(n), (p, q, r)
    assert torch.equal(d['b'], torch.sqrt(1 - (d['a'] * d['a'] + d['c'] * d['c'])))
    assert not bool(d['y'].any()) and bool((b.dev('state') == 1).all())
    # successive calls: fresh, uncorrelated rays
    b2 = src.shine(toGlobal=False)
    r = float(torch.corrcoef(torch.stack((d['x'], b2.dev('x'))))[0, 1])
    assert abs(r) < 5 / np.sqrt(n)


@pytest.mark.gpu
@pytest.mark.parametrize('pol', ['h', 'v', '+45', '-45', 'r', 'l', None, 'un', '30', '0.5rad',
                                 17.5, (0.6, 0.4, 0.2, -0.1)])
def test_device_polarisation_equals_the_host_paths(pol):
    n = 4096
    bl = raycing.BeamLine()
    dev = rs.GeometricSource(bl, 'd', nrays=n, polarization=pol, rng='device', seed=1)
    host = rs.GeometricSource(bl, 'h', nrays=n, polarization=pol)
    np.random.seed(0)
    bd, bh = dev.shine(withAmplitudes=True), host.shine(withAmplitudes=True)
    for f in ('Jss', 'Jpp', 'Jsp', 'Es'):
        assert np.array_equal(bd.peek(f), getattr(bh, f)), (pol, f)
    if pol in (None, 'un'):         # random |Ep| up to 1/sqrt(2): same law, other numbers
        ep = bd.peek('Ep')
        assert not ep.imag.any() and 0 <= ep.real.min() and ep.real.max() < 2**-0.5
        assert abs(ep.real.mean() - 2**-1.5) < 5 * 2**-0.5 / np.sqrt(12 * n)
    else:
        assert np.array_equal(bd.peek('Ep'), bh.Ep)


@pytest.mark.gpu
def test_device_source_flux_accu_beam_and_chain():
    """totalFlux normalisation, accuBeam energies, and the beam feeds the next element as it
    is (resident): source -> toroid mirror, states and footprint against the same rays pulled
    to the host and traced from there."""
    from xrt_amd import workloads
    src = make('default', 20_000, azimuth=0., totalFlux=1e13, distE='flat',
               energies=(8990., 9010.), dx=0.1, dz=0.1, dxprime=2e-4, dzprime=2e-5)
    b = src.shine()
    assert b.sourceWeight == 1e13 / 20_000 and b.seeded == 20_000
    u = make('uniform_density', 20_000, totalFlux=1e13).shine()
    assert np.isclose(u.sourceWeight * (u.peek('Jss') + u.peek('Jpp')).sum(), 1e13)
    again = src.shine(accuBeam=b)
    assert np.array_equal(again.peek('E'), b.peek('E'))
    assert not np.array_equal(again.peek('x'), b.peek('x'))
    oe = workloads.cfg2_toroid()
    gb, lb = oe.reflect(b)
    twin = rs.Beam(copyFrom=b)
    for f in twin.array_fields():
        getattr(twin, f)                  # host copies become the masters
    gb2, lb2 = oe.reflect(twin)
    assert np.array_equal(gb.peek('state'), gb2.peek('state'))
    assert np.array_equal(lb.peek('x'), lb2.peek('x')) and (gb.peek('state') == 1).mean() > 0.9


@pytest.mark.gpu
def test_concurrent_shines_take_different_substreams():
    """run_ray_tracing(threads=N) calls shine() from N threads: every call takes its own call
    number (sub-stream of the generator), none is handed out twice."""
    import threading
    import torch
    src = make('default', 10_000, azimuth=0.)
    got, calls = [], 24

    def work():
        with torch.cuda.stream(torch.cuda.Stream()):
            for _ in range(calls // 4):
                b = src.shine()
                torch.cuda.current_stream().synchronize()
                got.append(b.peek('x')[:8].copy())
    pool = [threading.Thread(target=work) for _ in range(4)]
    for t in pool:
        t.start()
    for t in pool:
        t.join()
    assert len(got) == calls and src._calls == calls
    assert len({g.tobytes() for g in got}) == calls
    first = [og.shine(spec_of(src, call=k))['x'][0] for k in range(calls)]
    assert sorted(round(float(g[0]), 9) for g in got) == sorted(round(float(v), 9) for v in first)
