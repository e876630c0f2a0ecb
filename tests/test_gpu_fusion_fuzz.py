"""GPU: randomised scripts over the Balder chain (reference examples/withRaycing/02_Balder_BL:
front-end screen and mask, both faces of the filter, collimating mirror, DCM, focusing mirror,
slits and screens between them) -- with the consumers fused into their producers
(sources.LazyBeam: passes wait for the apertures, screens and elements that take their beams,
beams nobody has asked for are not written, elements remember what was looked at) every beam,
looked at in ANY order and over several iterations of the same script, has the bits of the
run in which every call is an immediate launch (oes.fuseConsumers = False). The fixed-order
tests are tests/test_gpu_fusion.py; this one draws which slits and screens are there and who
looks at what first."""
import numpy as np
import pytest

import xrt_amd.backends.raycing.apertures as ra
import xrt_amd.backends.raycing.oes as roe
import xrt_amd.backends.raycing.screens as rsc
import xrt_amd.backends.raycing.sources as rs
from xrt_amd import workloads

pytestmark = pytest.mark.gpu

FIELDS = ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp', 'state')
WANTS = ('_local_beam_wanted', '_global_beam_wanted', '_local_beams_wanted', '_image_wanted',
         '_beam_wanted')


def pencil(n, seed, amplitudes=False):
    """The synthetic pencil of bench.py's Balder leg, some rays dead or astray on arrival."""
    rng = np.random.default_rng(seed)
    beam = rs.Beam(nrays=n, withAmplitudes=amplitudes)
    beam.x, beam.z = rng.normal(0, 0.05, n), rng.normal(0, 0.01, n)
    beam.y = np.zeros(n)
    a, c = rng.uniform(-1.9e-4, 1.9e-4, n), rng.uniform(-4.5e-5, 4.5e-5, n)
    beam.a, beam.c, beam.b = a, c, np.sqrt(1 - a**2 - c**2)
    beam.E = rng.uniform(8999., 9001., n)
    beam.state = np.ones(n, dtype=np.int32)
    beam.state[::97] = -3
    beam.state[5::101] = 2
    beam.a[::53] *= 40.
    beam.b = np.sqrt(1 - beam.a**2 - beam.c**2)
    beam.Jss, beam.Jpp, beam.Jsp = np.ones(n), np.zeros(n), np.zeros(n, complex)
    if amplitudes:
        beam.Es = np.exp(1j * rng.uniform(0, 6.28, n))
        beam.Ep = 0.3 * np.exp(1j * rng.uniform(0, 6.28, n))
    return beam


def optics():
    b = workloads.balder_optics()
    # more consumers than the example has, so that every producer can get a tail:
    # half-open slits (x is not touched by the vertical deflections) and screens
    b.slitF = ra.RectangularAperture(b.bl, 'afterFilter', (0, 24000, 0), blades={'left': -0.8})
    b.fsmF = rsc.Screen(b.bl, 'FSM-F', (0, 24500, 0))
    b.slitVCM = ra.RectangularAperture(b.bl, 'afterVCM', (0, 26000, 0), blades={'right': 1.2})
    b.slitVCM2 = ra.RectangularAperture(b.bl, 'afterVCM2', (0, 26100, 0), blades={'left': -1.5})
    b.fsmVCM = rsc.Screen(b.bl, 'FSM-VCM', (0, 26300, 0))
    b.fsmDCM = rsc.Screen(b.bl, 'FSM-DCM', (0, 29400, 0))
    # a slit that a script closes on a beam AFTER the next element has taken it (marks in place)
    b.slitLate = ra.RectangularAperture(b.bl, 'late', (0, 26200, 0), blades={'right': 0.6})
    return b


def script(b, beam, has, look=lambda out: None):
    """One pass of the chain with the consumers *has* names -> {name: beam}, every beam kept;
    *look(out)* is called between the steps (a script that prints or plots as it goes)."""
    out = {}
    if 'fsm0' in has:
        out['fsm0'] = b.fsm0.expose(beam)
    if 'mask' in has:
        out['mask'] = b.mask.propagate(beam)
    cur = beam
    if 'filter' in has:
        cur, out['filter.l1'], out['filter.l2'] = b.filter1.double_refract(cur)
        out['filter.g'] = cur
        if 'slitF' in has:
            out['slitF'] = b.slitF.propagate(cur)
        if 'fsmF' in has:
            out['fsmF'] = b.fsmF.expose(cur)
        look(out)
    cur, out['vcm.l'] = b.vcm.reflect(cur)
    out['vcm.g'] = cur
    for name in ('slitVCM', 'slitVCM2'):
        if name in has:
            out[name] = getattr(b, name).propagate(cur)
    if 'fsmVCM' in has:
        out['fsmVCM'] = b.fsmVCM.expose(cur)
    look(out)
    if 'dcm' in has:
        cur, out['dcm.l1'], out['dcm.l2'] = b.dcm.double_reflect(cur)
        out['dcm.g'] = cur
        if 'slitLate' in has:            # (the DCM has seen the beam as it was)
            out['slitLate'] = b.slitLate.propagate(out['vcm.g'])
        if 'slitDCM' in has:
            out['slitDCM'] = b.slitDCM.propagate(cur)
        if 'fsmDCM' in has:
            out['fsmDCM'] = b.fsmDCM.expose(cur)
    look(out)
    cur, out['vfm.l'] = b.vfm.reflect(cur)
    out['vfm.g'] = cur
    for name in ('slitVFM', 'slitEH'):
        if name in has:
            out[name] = getattr(b, name).propagate(cur)
    if 'sample' in has:
        out['sample'] = b.sample.expose(cur)
    return out


def fields(beam):
    return FIELDS + (('Es', 'Ep') if beam.has_amplitudes() else ())


def snapshot(beams):
    return {k: {f: np.array(v.peek(f)) for f in fields(v)} for k, v in beams.items()}


def equal(beam, ref, what):
    assert len(fields(beam)) == len(ref), what
    for f in fields(beam):
        u = beam.peek(f)
        assert np.array_equal(u, ref[f], equal_nan=True), (what, f)


OPTIONAL = ('fsm0', 'mask', 'filter', 'slitF', 'fsmF', 'slitVCM', 'slitVCM2', 'fsmVCM',
            'slitLate', 'slitDCM', 'fsmDCM', 'slitVFM', 'slitEH', 'sample')


@pytest.mark.parametrize('seed', range(96))
def test_random_scripts_over_the_balder_chain(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([3000, 40000, 70001]))
    beam = pencil(n, seed, amplitudes=bool(seed % 3 == 2))
    b = optics()
    has = set(k for k in OPTIONAL if rng.random() < 0.65) | {'dcm'}   # (the VFM sits behind it)
    old = roe.fuseConsumers
    roe.fuseConsumers = False
    try:
        # (a slit marks the beam it is given IN PLACE: a look on the way sees the states as they
        # are at that point of the script, not as they are at its end)
        on_the_way = []
        ref = snapshot(script(b, rs.Beam(copyFrom=beam), has,
                              lambda out: on_the_way.append(snapshot(out))))
    finally:
        roe.fuseConsumers = old
    assert (ref['vfm.g']['state'] == 1).sum() > 0.05 * n, sorted(has)
    for o in vars(b).values():
        for key in WANTS:
            getattr(o, '__dict__', {}).pop(key, None)
    for iteration in range(4):
        seen = []

        def look(out, iteration=iteration, seen=seen):
            # now and then the script looks at a beam it has got so far before it goes on
            seen.append(None)
            if out and rng.random() < (0.0, 0.3, 0.15, 0.3)[iteration]:
                k = list(out)[int(rng.integers(len(out)))]
                seen[-1] = k
                equal(out[k], on_the_way[len(seen) - 1][k],
                      (seed, iteration, k, sorted(has), 'on the way', seen))
        got = script(b, rs.Beam(copyFrom=beam), has, look)
        assert sorted(got) == sorted(ref)
        names = list(got)
        rng.shuffle(names)
        # the first iterations look at little (what a beamline script does), the last at all
        share = (0.15, 0.4, 0.0, 1.0)[iteration]
        looked = [k for k in names if rng.random() < share]
        for k in looked:
            equal(got[k], ref[k], (seed, iteration, k, sorted(has), looked))
        if rng.random() < 0.5:
            rs.flush_pending()
        for k in names:                 # what is looked at after the flush, or after the others
            if k not in looked and rng.random() < share:
                equal(got[k], ref[k], (seed, iteration, k, sorted(has), 'late'))
        del got


# ---- whole run_ray_tracing jobs: which plots, of which beams, ride or not -----------------------
def _job(n, has, seed):
    """Device source -> cfg2 toroid -> [slits] -> screen at the focus -> [a slit behind it];
    -> (beamLine, run_process)."""
    import xrt_amd.backends.raycing as raycing
    bl = raycing.BeamLine()
    bl.source = rs.GeometricSource(bl, 'source', nrays=n, dx=0.1, dz=0.1, dxprime=2e-4,
                                   dzprime=2e-5, distE='flat', energies=(8990., 9010.),
                                   polarization='h', rng='device', seed=seed)
    bl.mirror = workloads.cfg2_toroid(bl)
    way = np.array([0., np.cos(8e-3), np.sin(8e-3)])
    at = lambda d: list(np.array([0., 20000., 0.]) + d * way)      # noqa: E731
    bl.slitA = ra.RectangularAperture(bl, 'A', at(3000.), blades={'left': -1.0})
    bl.slitB = ra.RectangularAperture(bl, 'B', at(6000.), blades={'right': 0.8, 'top': 0.05})
    bl.slitC = ra.RectangularAperture(bl, 'C', at(9000.), blades={'left': -0.1})
    bl.screen = rsc.Screen(bl, 'focus', center=at(10000.))

    def run_process(beamLine):
        out = {}
        src = out['source'] = beamLine.source.shine()
        gb, lb = beamLine.mirror.reflect(src)
        out['mirrorGlobal'], out['mirrorLocal'] = gb, lb
        if 'slitA' in has:
            out['slitA'] = beamLine.slitA.propagate(gb)
        if 'slitB' in has:
            out['slitB'] = beamLine.slitB.propagate(gb)
        out['focus'] = beamLine.screen.expose(gb)
        if 'slitC' in has:
            out['slitC'] = beamLine.slitC.propagate(gb)
        return out
    return bl, run_process


LIMITS = {          # beam -> axis label -> (unit, limits)
    'focus': {'x': ('mm', [-0.4, 0.4]), 'z': ('mm', [-0.05, 0.05]), "x'": ('mrad', [-0.6, 0.6]),
              "z'": ('mrad', [-0.1, 0.1]), 'path': ('mm', [29999., 30001.])},
    'mirrorLocal': {'x': ('mm', [-12., 12.]), 'y': ('mm', [-350., 350.]),
                    'z': ('mm', [-0.01, 0.06])},
    'mirrorGlobal': {'x': ('mm', [-12., 12.]), 'z': ('mm', [-1.5, 1.5]),
                     "z'": ('mrad', [7.8, 8.2])},
    'source': {'x': ('mm', [-0.4, 0.4]), 'z': ('mm', [-0.4, 0.4]), "x'": ('mrad', [-0.8, 0.8])},
    'slitA': {'x': ('mm', [-12., 12.]), 'z': ('mm', [-1., 1.])},
}


def _draw_plots(rng, has):
    from xrt_amd import plotter as xrtp
    beams = [k for k in LIMITS if k != 'slitA' or 'slitA' in has]
    spec = []
    for _ in range(int(rng.integers(1, 4))):
        beam = beams[int(rng.integers(len(beams)))] if rng.random() < 0.5 else 'focus'
        labels = list(LIMITS[beam])
        rng.shuffle(labels)
        spec.append((beam, labels[0], labels[1], int(rng.choice([64, 128, 256])),
                     [(1,), (1, 2), (1, 2, 3, -1)][int(rng.integers(3))],
                     int(rng.choice([32, 128]))))

    def make():
        plots = []
        for beam, lx, ly, bins, flag, cbins in spec:
            (ux, limx), (uy, limy) = LIMITS[beam][lx], LIMITS[beam][ly]
            plots.append(xrtp.XYCPlot(
                beam, flag, xaxis=xrtp.XYCAxis(lx, ux, limits=list(limx), bins=bins),
                yaxis=xrtp.XYCAxis(ly, uy, limits=list(limy), bins=bins),
                caxis=xrtp.XYCAxis('energy', 'eV', limits=[8989., 9011.], bins=cbins)))
        return plots
    return spec, make


def _plots_agree(p, q, what):
    for name in ('total2D', 'total2D_RGB'):
        a, b = getattr(p, name), getattr(q, name)
        assert np.array_equal(a != 0, b != 0), (what, name)
        assert np.abs(a - b).max() <= 1e-12 * max(np.abs(b).max(), 1e-300), (what, name)
    for axis in ('xaxis', 'yaxis', 'caxis'):
        a, b = getattr(p, axis).total1D4, getattr(q, axis).total1D4
        assert np.abs(a - b).max() <= 1e-12 * max(np.abs(b).max(), 1e-300), (what, axis)
    for name in ('nRaysAll', 'nRaysSelected', 'nRaysAlive', 'nRaysGood', 'nRaysOut', 'nRaysOver',
                 'nRaysDead', 'iteration'):
        assert getattr(p, name) == getattr(q, name), (what, name)
    for name in ('intensity', 'intensityInRange'):
        assert abs(getattr(p, name) - getattr(q, name)) <= 1e-12 * abs(getattr(q, name)) + 1e-300, \
            (what, name)


@pytest.mark.parametrize('seed', range(100))
def test_random_run_ray_tracing_jobs(seed):
    """run_ray_tracing (reference xrt/runner.py:513-719) over drawn jobs: one to three plots of
    drawn beams and axes, slits before and behind the screen, eager loop or HIP-graph replays --
    the plots of the run with everything fused (a plot may ride in the tail of the pass, the
    source in its head) equal those of the run of immediate launches."""
    from xrt_amd import runner as xrtr
    import xrt_amd.backends.raycing.run as rr
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([2000, 50000, 120001]))
    has = set(k for k in ('slitA', 'slitB', 'slitC') if rng.random() < 0.5)
    repeats = int(rng.integers(2, 7))
    graph = bool(rng.random() < 0.4)
    bl, run_process = _job(n, has, seed)
    spec, make = _draw_plots(rng, has)
    what = (seed, n, sorted(has), repeats, graph, spec)
    keep = rr.run_process
    rr.run_process = run_process
    old = roe.fuseConsumers
    try:
        roe.fuseConsumers = False
        bl.source._calls = 0
        ref = xrtr.run_ray_tracing(make(), repeats=repeats, beamLine=bl)
        for q, (beam, lx, ly, _, _, _) in zip(ref, spec):
            assert q.nRaysAll == n * repeats and q.iteration == repeats, what
            if {lx, ly} <= {'x', 'y', 'z'}:       # (the limits hold the beam: something to compare)
                assert q.total2D.sum() > 0 and (q.total2D != 0).sum() > 20, (what, beam)
        roe.fuseConsumers = True
        for again in range(2):        # (the second run: elements and screens remember)
            bl.source._calls = 0
            got = xrtr.run_ray_tracing(make(), repeats=repeats, beamLine=bl, graph=graph)
            for k, (p, q) in enumerate(zip(got, ref)):
                _plots_agree(p, q, what + (again, k))
    finally:
        roe.fuseConsumers = old
        rr.run_process = keep


# ---- drawn elements of every kernel family with drawn consumers behind them ---------------------
def _consumers(oe, g0, rng):
    """Up to two apertures and perhaps a screen along the beam that leaves *oe* (*g0*: its
    global beam from the immediate launch), sized by the beam as a wide-open slit sees it."""
    ok = g0.state == 1
    pos = np.array([g0.x[ok].mean(), g0.y[ok].mean(), g0.z[ok].mean()])
    way = np.array([g0.a[ok].mean(), g0.b[ok].mean(), g0.c[ok].mean()])
    way /= np.sqrt((way**2).sum())
    bl = oe.bl
    aps = []
    for k in range(int(rng.integers(0, 3))):
        at = list(pos + 150. * (k + 1) * way)
        seen = ra.RectangularAperture(bl, 'open', at, ('left',), [-1e9]).propagate(
            rs.Beam(copyFrom=g0))
        live = seen.state == 1
        x, z = seen.x[live], seen.z[live]
        what = str(rng.choice(['rect', 'round', 'stop']))
        if what == 'rect':
            aps.append(ra.RectangularAperture(bl, 'slit%d' % k, at, ('left', 'top'),
                                              [float(np.quantile(x, 0.25)),
                                               float(np.quantile(z, 0.8))]))
        elif what == 'round':
            aps.append(ra.RoundAperture(bl, 'pipe%d' % k, at,
                                        r=float(np.median(np.hypot(x, z)))))
        else:
            aps.append(ra.RectangularBeamStop(
                bl, 'stop%d' % k, at, ('left', 'right', 'bottom', 'top'),
                [float(np.quantile(x, 0.4)), float(np.quantile(x, 0.6)),
                 float(np.quantile(z, 0.3)), float(np.quantile(z, 0.7))]))
    scr = rsc.Screen(bl, 'after', list(pos + 700. * way)) if rng.random() < 0.7 else None
    return aps, scr


def _element_script(oe, aps, scr, beam, need_local):
    out = {}
    g, l = oe.reflect(rs.Beam(copyFrom=beam), needLocal=need_local)
    out['g'] = g
    if need_local:
        out['l'] = l
    for k, a in enumerate(aps):
        out['ap%d' % k] = a.propagate(g)
    if scr is not None:
        out['img'] = scr.expose(g)
    return out


@pytest.mark.parametrize('seed', range(5))
@pytest.mark.parametrize('kind', ['flat', 'toroid', 'bentflat', 'ellipse', 'ellipse_cyl',
                                  'parabola', 'hyperbola', 'blazed', 'grating'])
def test_random_elements_with_random_consumers(kind, seed):
    """OE.reflect of drawn elements (every surface family of tests/test_gpu_fuzz.py: lean,
    generic, parametric, gratings; coatings, thin mirrors, no material) followed by drawn slits,
    pipes, beam stops and a screen -- whether the kernel carries them in its tail or they take
    their own launches inside the call, every beam has the bits of the immediate launches, over
    three iterations in which the elements remember what was looked at."""
    from test_gpu_fuzz import KINDS, aimed_beam, make_element
    rng = np.random.default_rng(77000 + 100 * KINDS.index(kind) + seed)
    oe, pitch = make_element(kind, rng)
    beam = aimed_beam(oe, pitch, rng, n=int(rng.choice([400, 6000])))
    need_local = bool(rng.random() < 0.75)
    old = roe.fuseConsumers
    roe.fuseConsumers = False
    try:
        try:
            g0 = oe.reflect(rs.Beam(copyFrom=beam))[0]
        except Exception as e:                 # (a blazed grating: "above both facets")
            pytest.skip(str(e)[:80])
        if (g0.state == 1).sum() < 100:
            pytest.skip('the fan mostly misses')
        aps, scr = _consumers(oe, g0, rng)
        ref = snapshot(_element_script(oe, aps, scr, beam, need_local))
    finally:
        roe.fuseConsumers = old
    what = (kind, seed, need_local, [type(a).__name__ for a in aps], scr is not None)
    for iteration in range(3):
        got = _element_script(oe, aps, scr, beam, need_local)
        names = list(got)
        rng.shuffle(names)
        share = (0.3, 0.0, 1.0)[iteration]
        for k in names:
            if rng.random() < share:
                equal(got[k], ref[k], what + (iteration, k))
        if rng.random() < 0.5:
            rs.flush_pending()
        if iteration == 2:
            for k in names:
                equal(got[k], ref[k], what + (iteration, k, 'again'))
        del got


# ---- drawn monochromators and single crystals with drawn consumers ------------------------------
def _crystal_fan(bl, rng, E0, n):
    """A fan converging on [0, 20000, 0] (turned by the beamline's azimuth), some rays dead or
    over the edge on arrival."""
    import xrt_amd.backends.raycing as raycing
    beam = rs.Beam(nrays=n, withAmplitudes=bool(rng.random() < 0.5))
    beam.x[:] = rng.normal(0, 2.5, n)
    beam.z[:] = rng.normal(0, 0.6, n)
    beam.a[:] = rng.normal(0, 1e-4, n)
    beam.c[:] = rng.normal(0, 2e-5, n)
    beam.b[:] = np.sqrt(1 - beam.a**2 - beam.c**2)
    beam.y[:] = 9900.
    beam.z[:] += -beam.c / beam.b * 100.
    if bl.azimuth:
        for u, v in (('x', 'y'), ('a', 'b')):
            p, q = getattr(beam, u).copy(), getattr(beam, v).copy()
            pu, qv = raycing.rotate_z(p, q, bl.cosAzimuth, -bl.sinAzimuth)
            getattr(beam, u)[:] = pu
            getattr(beam, v)[:] = qv
    beam.E[:] = rng.uniform(E0 - 3., E0 + 3., n)
    ang = rng.uniform(0, np.pi, n)
    es, ep = np.cos(ang), np.sin(ang) * np.exp(1j * rng.uniform(-np.pi, np.pi, n))
    beam.Jss[:], beam.Jpp[:], beam.Jsp[:] = es * es, (ep * np.conj(ep)).real, es * np.conj(ep)
    if hasattr(beam, 'Es'):
        beam.Es[:], beam.Ep[:] = es, ep
    st = np.ones(n, dtype=np.int32)
    st[rng.random(n) < 0.03] = 2
    st[rng.random(n) < 0.02] = -1
    beam.state[:] = st
    return beam


def _random_crystal_optic(rng, pair):
    import xrt_amd.backends.raycing as raycing
    import xrt_amd.backends.raycing.materials as rm
    bl = raycing.BeamLine(azimuth=float(rng.choice([0., 0.15])))
    hkl = [(1, 1, 1), (3, 1, 1), (3, 3, 3)][int(rng.integers(0, 3))]
    E0 = float(rng.uniform(9500. if hkl == (3, 3, 3) else 7000., 16000.))
    thin = dict(t=float(rng.uniform(0.02, 0.2))) if (not pair and rng.random() < 0.4) else {}
    si1 = rm.CrystalSi(hkl=hkl, tK=297.15, **thin)
    alpha = float(rng.choice([0., 0., np.radians(2.), np.radians(-3.)]))
    thB = float(np.ravel(si1.get_Bragg_angle(E0))[0])
    thB -= float(np.ravel(si1.get_dtheta(E0, alpha) if alpha else si1.get_dtheta(E0))[0]) \
        if (alpha or not pair) else 0.
    center = [20000. * bl.sinAzimuth, 20000. * bl.cosAzimuth, 0.]
    if pair:
        perp = float(rng.uniform(5., 25.))
        optic = roe.DCM(
            bl, 'dcm', center=center, bragg=thB, pitch=alpha, material=si1,
            material2=rm.CrystalSi(hkl=hkl, tK=297.15), alpha=alpha if alpha else None,
            cryst2perpTransl=perp,
            cryst2longTransl=float(perp / np.tan(thB) * rng.uniform(0.8, 1.2)),
            cryst2finePitch=float(rng.normal(0, 2e-6)), limPhysX=[-10, 10], limPhysY=[-40, 40],
            limPhysX2=[-10, 10], limPhysY2=[-80, 80])
    else:
        optic = roe.OE(bl, 'xtal', center=center, pitch=thB + alpha, material=si1,
                       alpha=alpha if alpha else None, limPhysX=[-10, 10], limPhysY=[-40, 40])
    return optic, _crystal_fan(bl, rng, E0, int(rng.choice([500, 4000])))


def _crystal_script(optic, pair, aps, scr, beam):
    out = {}
    made = optic.double_reflect(rs.Beam(copyFrom=beam)) if pair else \
        optic.reflect(rs.Beam(copyFrom=beam))
    g = out['g'] = made[0]
    for k, lo in enumerate(made[1:]):
        out['l%d' % k] = lo
    for k, a in enumerate(aps):
        out['ap%d' % k] = a.propagate(g)
    if scr is not None:
        out['img'] = scr.expose(g)
    return out


@pytest.mark.parametrize('seed', range(16))
@pytest.mark.parametrize('pair', [True, False])
def test_random_crystals_with_random_consumers(pair, seed):
    """DCM.double_reflect (reference oes/dcm.py:248-354) and OE.reflect of a single flat Bragg
    crystal, thick or thin, with drawn reflections, energies and asymmetric cuts, followed by
    drawn slits, pipes, beam stops and a screen: in the tail of the crystal kernels or as their
    own launches, every beam has the bits of the immediate launches over three iterations."""
    rng = np.random.default_rng(91000 + 1000 * int(pair) + seed)
    optic, beam = _random_crystal_optic(rng, pair)
    old = roe.fuseConsumers
    roe.fuseConsumers = False
    try:
        made = optic.double_reflect(rs.Beam(copyFrom=beam)) if pair else \
            optic.reflect(rs.Beam(copyFrom=beam))
        g0 = made[0]
        if (g0.state == 1).sum() < 100:
            pytest.skip('the fan mostly misses the rocking curve')
        aps, scr = _consumers(optic, g0, rng)
        ref = snapshot(_crystal_script(optic, pair, aps, scr, beam))
    finally:
        roe.fuseConsumers = old
    what = (pair, seed, [type(a).__name__ for a in aps], scr is not None)
    for iteration in range(3):
        got = _crystal_script(optic, pair, aps, scr, beam)
        names = list(got)
        rng.shuffle(names)
        share = (0.3, 0.0, 1.0)[iteration]
        for k in names:
            if rng.random() < share:
                equal(got[k], ref[k], what + (iteration, k))
        if rng.random() < 0.5:
            rs.flush_pending()
        if iteration == 2:
            for k in names:
                equal(got[k], ref[k], what + (iteration, k, 'again'))
        del got
