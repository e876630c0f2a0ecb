"""GPU: a consumer fused into its producer (N1 of SURVEY 8f). OE.reflect hands
out its beams before anything is launched (sources.LazyBeam); if the script's first use of the
global beam is Screen.expose, the image is made in the tail of the SAME pass
(xrt_hip_reflect_screen_f64_dev, reflect_fused_scr) and the global beam is written only if
somebody asks for it afterwards. Everything must be bit-identical to the two separate launches
(reference: oes/reflect.py:18-163 followed by screens.py:226-302), whatever the order in which
the script touches the beams."""
import os

import numpy as np
import pytest
import torch

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.apertures as ra
import xrt_amd.backends.raycing.materials as rm
import xrt_amd.backends.raycing.oes as roe
import xrt_amd.backends.raycing.run as rr
import xrt_amd.backends.raycing.screens as rsc
import xrt_amd.backends.raycing.sources as rs
from xrt_amd import plotter as xrtp, runner as xrtr, workloads

pytestmark = pytest.mark.gpu

FIELDS = ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp', 'state')


def same(a, b, what, extra=()):
    names = FIELDS + (('Es', 'Ep') if a.has_amplitudes() else ()) + tuple(extra)
    for f in names:
        u, v = a.peek(f), b.peek(f)
        assert np.array_equal(u, v, equal_nan=True), (what, f, np.abs(u - v).max())


def scene(n=200000, amplitudes=True, bad=True):
    bl = raycing.BeamLine(azimuth=0.02)
    oe = workloads.cfg2_toroid(bl)
    scr = rsc.Screen(bl, 'focus', center=[0, 20000. + 10000. * np.cos(8e-3),
                                          10000. * np.sin(8e-3)])
    beam = workloads.synthetic_rays(n, 7, amplitudes=amplitudes)
    if bad:
        beam.state[::97] = -3          # dead on arrival
        beam.state[5::101] = 2
        beam.x[::53] *= 300.           # off the mirror
        beam.c[3::211] = -beam.c[3::211] - 1e-2   # away from it
        beam.b[:] = np.sqrt(1 - beam.a**2 - beam.c**2)
    return bl, oe, scr, beam


def eager(oe, scr, beam, **kw):
    old = roe.fuseConsumers
    roe.fuseConsumers = False
    try:
        gb, lb = oe.reflect(beam)
        img = scr.expose(gb, **kw)
    finally:
        roe.fuseConsumers = old
    return gb, lb, img


@pytest.mark.parametrize('amplitudes', [False, True])
def test_screen_in_the_tail_of_the_pass_is_the_two_launches(amplitudes):
    bl, oe, scr, beam = scene(amplitudes=amplitudes)
    gb0, lb0, img0 = eager(oe, scr, beam)
    gb, lb = oe.reflect(beam)
    assert type(gb) is rs.LazyBeam and gb.__dict__['_op'].state == 'pending'
    img = scr.expose(gb)
    op = gb.__dict__['_op']
    # (round 6: the image is handed out before the launch as well -- a plot may still join)
    assert type(img) is rs.LazyBeam and op.state == 'pending' and not img.__dict__['_filled']
    assert img.nrays == beam.nrays              # looked at: the pass with the screen in its tail
    assert op.state == 'imaged' and not gb.__dict__['_filled']        # the lean kernel took it
    same(img, img0, 'image')
    same(lb, lb0, 'local', extra=('theta',))
    assert img.parentId == img0.parentId and lb.parentId == lb0.parentId
    # ... and the global beam, asked for afterwards, is the eager one
    same(gb, gb0, 'global')
    assert op.state == 'done'
    # a second screen on the now existing beam: the plain launch
    same(scr.expose(gb, onlyPositivePath=True), eager(oe, scr, beam, onlyPositivePath=True)[2],
         'second image')


def test_any_other_first_use_launches_the_plain_pass():
    bl, oe, scr, beam = scene(n=50000)
    gb0, lb0, img0 = eager(oe, scr, beam)
    gb, lb = oe.reflect(beam)
    assert lb.nrays == beam.nrays                 # the local beam is looked at first
    assert gb.__dict__['_op'].state == 'done' and gb.__dict__['_filled']
    same(lb, lb0, 'local', extra=('theta',))
    same(scr.expose(gb), img0, 'image')
    same(gb, gb0, 'global')
    # attributes of an xrt script on a pending beam
    gb, lb = oe.reflect(beam)
    assert np.array_equal(gb.state, gb0.state) and hasattr(lb, 'theta')
    # a copy
    gb, lb = oe.reflect(beam)
    same(rs.Beam(copyFrom=gb), gb0, 'copy')


def test_a_contradicted_pass_is_redone_and_imaged_from_the_real_beam():
    """Rays for which the batch statistics ask for Brent's method (golden g2_toroid_brent): the
    optimistic pass with the screen in its tail is contradicted, the exact sequence writes the
    global beam and the image is made from it."""
    import p1_cases
    g = np.load(os.path.join(p1_cases.GOLDEN, 'g2_toroid_brent.npz'))
    oe = p1_cases.product_oe('g2_toroid_brent', g)
    beam = p1_cases.product_beam(g)
    scr = rsc.Screen(oe.bl, 'after', center=[0, float(g['oe_center'][1]) + 3000., 10.])
    info = {}
    oe.reflect(beam, _info=info)
    assert info['brent']
    gb0, lb0, img0 = eager(oe, scr, beam)
    gb, lb = oe.reflect(beam)
    img = scr.expose(gb)
    same(img, img0, 'image')
    same(lb, lb0, 'local', extra=('theta',))
    same(gb, gb0, 'global')


def test_elements_whose_kernels_do_not_carry_a_screen():
    """A bent Bragg crystal (surface family 2: neither a lean kernel nor the flat crystals'): the
    same call, the screen's own launch inside it."""
    bl = raycing.BeamLine()
    si = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
    thB = float(np.ravel(si.get_Bragg_angle(9000.) - si.get_dtheta(9000.))[0])
    xt = roe.JohannCylinder(bl, 'xtal', center=[0, 20000., 0], pitch=thB, material=si,
                            Rm=1e9, limPhysX=[-10, 10], limPhysY=[-50, 50])
    scr = rsc.Screen(bl, 'after', center=[0, 21000., 1000. * np.tan(2 * thB)])
    beam = workloads.synthetic_rays(60000, 3, sa=1e-4, E=(8995., 9005.), amplitudes=True)
    gb0, lb0, img0 = eager(xt, scr, beam)
    gb, lb = xt.reflect(beam)
    img = scr.expose(gb)
    assert img.nrays == beam.nrays
    assert gb.__dict__['_filled'] and not lb.__dict__['_filled']
    same(img, img0, 'image')
    same(gb, gb0, 'global')
    same(lb, lb0, 'local', extra=('theta',))
    # a hemispheric screen reads the stored beam
    hs = rsc.HemisphericScreen(bl, 'sphere', center=[0, 20000., 0], R=500.)
    m = workloads.cfg2_toroid(bl)
    beam = workloads.synthetic_rays(20000, 4)
    g1, _ = m.reflect(beam)
    i1 = hs.expose(g1)
    roe.fuseConsumers = False
    try:
        g2, _ = m.reflect(beam)
        i2 = hs.expose(g2)
    finally:
        roe.fuseConsumers = True
    same(i1, i2, 'sphere', extra=('theta', 'phi'))


def test_changes_in_place_come_after_the_pending_pass():
    """aperture.propagate writes beam.state in place: a pass still waiting that reads the beam is
    launched first, as in program order."""
    bl, oe, scr, beam = scene(n=40000, bad=False)
    slit = ra.RectangularAperture(bl, 'slit', [0, 15000., 0], ('left', 'right'), [-0.02, 0.02])
    roe.fuseConsumers = False
    try:
        b0 = rs.Beam(copyFrom=beam)
        gb0, lb0 = oe.reflect(b0)
        slit.propagate(b0)
    finally:
        roe.fuseConsumers = True
    b1 = rs.Beam(copyFrom=beam)
    gb, lb = oe.reflect(b1)
    slit.propagate(b1)                                   # (would kill most rays)
    same(lb, lb0, 'local', extra=('theta',))
    same(gb, gb0, 'global')
    assert np.array_equal(b1.state, b0.state) and (b1.state < 0).sum() > 1000
    # the input edited on the host afterwards does not reach the pass either
    b2 = rs.Beam(copyFrom=beam)
    gb, lb = oe.reflect(b2)
    b2.x[:] = 1e6
    same(lb, lb0, 'local', extra=('theta',))


def test_run_ray_tracing_with_and_without_fusion():
    def run(fuse, graph):
        roe.fuseConsumers = fuse
        try:
            bl, run_process, make_plot = workloads.e2e_beamline(100000, seed=5)
            rr.run_process = run_process
            plots = [make_plot(),
                     xrtp.XYCPlot('mirrorLocal', (1,), xrtp.XYCAxis('x', 'mm', limits=[-3, 3]),
                                  xrtp.XYCAxis('y', 'mm', limits=[-300, 300]))]
            xrtr.run_ray_tracing(plots, repeats=6, beamLine=bl, graph=graph)
            torch.cuda.synchronize()
            return plots
        finally:
            roe.fuseConsumers = True
    ref = run(False, False)
    for graph in (False, True):
        got = run(True, graph)
        for a, b in zip(got, ref):
            assert a.nRaysAll == b.nRaysAll and a.total2D.max() > 0
            assert np.abs(a.total2D - b.total2D).max() <= 1e-12 * b.total2D.max()
            assert abs(a.intensity - b.intensity) <= 1e-12 * b.intensity


def test_c_abi_keeps_the_global_beam_on_request():
    """xrt_hip_reflect_screen_f64_dev(keep_virgin=1): all three beams from one call."""
    import ctypes
    from xrt_amd import _lib, hipcalls
    bl, oe, scr, beam = scene(n=30000)
    gb0, lb0, img0 = eager(oe, scr, beam)
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    p = oe._make_pass(oe.pitch, oe.roll + oe.positionRoll, oe.yaw, oe.dx)
    ms = oe._material_struct(oe.material, True, dev, beam)
    lb, gb, img = (rs.Beam.empty_like_on_device(beam, dev) for _ in range(3))
    theta = torch.empty(beam.nrays, dtype=torch.float64, device=dev)
    ws = hipcalls.workspace(dev, lib.xrt_hip_reflect_workspace_bytes(beam.nrays), 'reflect')
    fused = ctypes.c_int(-1)
    s_in = beam.to_struct(dev)
    _lib.check(lib.xrt_hip_reflect_screen_f64_dev(
        ctypes.byref(p), ctypes.byref(ms), ctypes.byref(s_in), ctypes.byref(s_in),
        ctypes.byref(lb.to_struct(dev)), ctypes.byref(gb.to_struct(dev)),
        ctypes.c_void_p(theta.data_ptr()), ctypes.byref(scr._record(False)),
        ctypes.byref(img.to_struct(dev)), 1, ctypes.c_void_p(ws.data_ptr()), ws.numel(),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(fused), None),
        'xrt_hip_reflect_screen_f64_dev')
    assert fused.value == 1
    same(img, img0, 'image')
    same(gb, gb0, 'global')
    same(lb, lb0, 'local')


# ---- the source in the head of the pass ------------------------------------------------------
def source_scene(n=150000, wide=False, amplitudes=False):
    bl = raycing.BeamLine()
    kw = dict(dxprime=0.9, distxprime='flat') if wide else dict(dxprime=2e-4)
    bl.source = rs.GeometricSource(bl, 'source', nrays=n, dx=0.1, dz=0.1, dzprime=2e-5,
                                   distE='flat', energies=(8990., 9010.), polarization='h',
                                   rng='device', seed=17, **kw)
    bl.mirror = workloads.cfg2_toroid(bl)
    bl.screen = rsc.Screen(bl, 'focus', center=[0, 20000. + 10000. * np.cos(8e-3),
                                                10000. * np.sin(8e-3)])
    return bl, amplitudes


def run_chain(bl, amplitudes, fuse):
    old = roe.fuseConsumers
    roe.fuseConsumers = fuse
    try:
        bl.source._calls = 0
        src = bl.source.shine(withAmplitudes=amplitudes)
        gb, lb = bl.mirror.reflect(src)
        img = bl.screen.expose(gb)
        img.nrays                   # (the first look at the image launches the pass)
    finally:
        roe.fuseConsumers = old
    return src, gb, lb, img


@pytest.mark.parametrize('amplitudes', [False, True])
def test_source_mirror_screen_in_one_pass(amplitudes):
    bl, amp = source_scene(amplitudes=amplitudes)
    src0, gb0, lb0, img0 = run_chain(bl, amp, False)
    src, gb, lb, img = run_chain(bl, amp, True)
    assert type(src) is rs.LazyBeam and src.__dict__['_op'].state == 'inflight'
    assert not src.__dict__['_filled'] and gb.__dict__['_op'].state == 'imaged'
    same(img, img0, 'image')
    same(lb, lb0, 'local', extra=('theta',))
    assert img.parentId == img0.parentId
    # the beams nobody had asked for, afterwards: the same rays again
    same(src, src0, 'source')
    same(gb, gb0, 'global')
    assert src.parentId == src0.parentId == bl.source.uuid


def test_source_looked_at_first_and_redo_with_a_source():
    bl, amp = source_scene(n=40000)
    src0, gb0, lb0, img0 = run_chain(bl, amp, False)
    # the script looks at the source before anything else: the generator's own launch
    bl.source._calls = 0
    src = bl.source.shine()
    assert src.nrays == 40000 and src.__dict__['_op'].state == 'done'
    gb, lb = bl.mirror.reflect(src)
    same(bl.screen.expose(gb), img0, 'image')
    same(src, src0, 'source')
    # rays all over the place: the optimistic pass is contradicted (a ray's largest direction
    # cosine is not y), the beam is generated after all and the exact sequence does the pass
    bl, amp = source_scene(n=40000, wide=True)
    src0, gb0, lb0, img0 = run_chain(bl, amp, False)
    src, gb, lb, img = run_chain(bl, amp, True)
    same(img, img0, 'wide image')
    same(lb, lb0, 'wide local', extra=('theta',))
    same(gb, gb0, 'wide global')
    same(src, src0, 'wide source')


def test_an_aperture_on_the_pending_source_beam():
    bl, amp = source_scene(n=30000)
    slit = ra.RectangularAperture(bl, 'slit', [0, 15000., 0], ('left', 'right'), [-0.5, 0.5])

    def chain(fuse):
        roe.fuseConsumers = fuse
        try:
            bl.source._calls = 0
            src = bl.source.shine()
            gb, lb = bl.mirror.reflect(src)      # before the slit: sees every ray
            slit.propagate(src)
            img = bl.screen.expose(gb)
            return src, lb, img
        finally:
            roe.fuseConsumers = True
    s0, l0, i0 = chain(False)
    s1, l1, i1 = chain(True)
    same(i1, i0, 'image')
    same(l1, l0, 'local', extra=('theta',))
    same(s1, s0, 'source after the slit')
    assert (s1.state < 0).sum() > 100


def test_elements_and_sources_remember_what_was_wanted():
    """A global beam (a source's beam) that was left out and then asked for after all costs a
    second launch ONCE: from then on the element's fused pass writes it (the source launches its
    generator at once)."""
    bl, amp = source_scene(n=20000)
    src0, gb0, lb0, img0 = run_chain(bl, amp, False)
    src, gb, lb, img = run_chain(bl, amp, True)
    assert gb.__dict__['_op'].state == 'imaged' and src.__dict__['_op'].state == 'inflight'
    same(gb, gb0, 'global')                  # asked for: the element remembers
    same(src, src0, 'source')                # ... and so does the source
    src, gb, lb, img = run_chain(bl, amp, True)
    assert type(src) is not rs.LazyBeam      # its own launch, at once
    assert gb.__dict__['_filled']            # written by the pass
    assert not lb.__dict__['_filled'] and not bl.mirror.__dict__.get('_local_beams_wanted')
    same(gb, gb0, 'global, kept')
    same(img, img0, 'image')
    same(lb, lb0, 'local', extra=('theta',))     # asked for: remembered as well
    assert bl.mirror.__dict__['_local_beams_wanted']
    src, gb, lb, img = run_chain(bl, amp, True)
    assert gb.__dict__['_op'].state == 'done' and lb.__dict__['_filled']
    same(lb, lb0, 'local, kept', extra=('theta',))


def _apertures(bl, gb=None):
    """Five kinds of aperture half way to the screen, each cutting a good part of the beam."""
    at = [0, 20000. + 5000. * np.cos(8e-3), 5000. * np.sin(8e-3)]
    sx = sz = 1.
    if gb is not None:      # sized by the beam as a wide-open slit sees it
        roe.fuseConsumers = False
        try:
            seen = ra.RectangularAperture(bl, 'open', at, ('left',), [-1e9]).propagate(
                rs.Beam(copyFrom=gb))
        finally:
            roe.fuseConsumers = True
        ok = seen.state == 1
        sx, sz = np.abs(seen.x[ok]).mean() * 2, np.abs(seen.z[ok]).mean() * 2
    return [ra.RectangularAperture(bl, 'slit', at, ('left', 'right', 'top'),
                                   [-0.6 * sx, 0.4 * sx, 0.3 * sz]),
            ra.RectangularBeamStop(bl, 'stop', at, ('left', 'right', 'bottom', 'top'),
                                   [-0.3 * sx, 0.3 * sx, -0.4 * sz, 0.4 * sz]),
            ra.RoundAperture(bl, 'pipe', at, r=0.5 * min(sx, sz)),
            ra.DoubleSlit(bl, 'two', at, ('bottom', 'top'), [-0.5 * sz, 0.5 * sz],
                          shadeFraction=0.3),
            ra.PolygonalAperture(bl, 'tri', at, opening=[(-0.5 * sx, -0.4 * sz),
                                                         (0.6 * sx, -0.3 * sz), (0., 0.5 * sz)])]


@pytest.mark.parametrize('amplitudes', [False, True])
def test_aperture_local_beam_made_on_demand(amplitudes):
    """propagate() marks the stopped rays at once (states only) and makes the beam in the
    aperture's frame when it is first looked at: the bits of the one full launch, also after
    the incoming beam went through another aperture in between."""
    bl, oe, scr, beam = scene(n=60000, amplitudes=amplitudes)
    gb0 = eager(oe, scr, beam)[0]
    kinds = _apertures(bl, gb0)
    for ap, other in zip(kinds, kinds[1:] + kinds[:1]):
        roe.fuseConsumers = False
        try:
            b0 = rs.Beam(copyFrom=gb0)
            l0 = ap.propagate(b0)
            after_first = b0.state.copy()
            other.propagate(b0)
        finally:
            roe.fuseConsumers = True
        b1 = rs.Beam(copyFrom=gb0)
        l1 = ap.propagate(b1)
        assert type(l1) is rs.LazyBeam and not l1.__dict__['_filled']
        assert np.array_equal(b1.state, after_first), ap.name
        assert 1000 < (after_first != gb0.state).sum() < 0.97 * len(b1), ap.name
        l2 = other.propagate(b1)                 # changes b1.state again, in place
        assert not l1.__dict__['_filled']
        assert np.array_equal(b1.state, b0.state), ap.name
        same(l1, l0, 'local of ' + ap.name)
        assert l1.__dict__['_filled'] and not l2.__dict__['_filled']
        # with the new global beam: the full launch at once, as before
        b2 = rs.Beam(copyFrom=gb0)
        g2, l3 = ap.propagate(b2, needNewGlobal=True)
        assert type(l3) is rs.Beam
        same(l3, l0, 'local with global of ' + ap.name)


def test_aperture_local_beam_never_looked_at_costs_nothing_more():
    import gc
    bl, oe, scr, beam = scene(n=20000, bad=False)
    gb = eager(oe, scr, beam)[0]
    slit = _apertures(bl, gb)[0]
    local = slit.propagate(gb)
    op = local.__dict__['_op']
    assert op in rs._PENDING.optional()
    rs.flush_pending()                            # (the end of an iteration)
    assert not local.__dict__['_filled']
    # a change of the incoming arrays in place comes after the reader
    l0 = None
    roe.fuseConsumers = False
    try:
        g0 = rs.Beam(copyFrom=eager(oe, scr, beam)[0])
        l0 = slit.propagate(g0)
    finally:
        roe.fuseConsumers = True
    rs.flush_pending(gb)
    assert local.__dict__['_filled'] and op not in rs._PENDING.optional()
    same(local, l0, 'local before the change')
    # dropped unseen: gone with its beam
    import weakref
    local = slit.propagate(gb)
    gone = weakref.ref(local.__dict__['_op'])
    assert gone() in rs._PENDING.optional()
    del local, op
    gc.collect()
    assert gone() is None and None not in rs._PENDING.optional()


def test_c_abi_aperture_states_only():
    import ctypes
    from xrt_amd import _lib
    bl, oe, scr, beam = scene(n=30000)
    gb = eager(oe, scr, beam)[0]
    slit = _apertures(bl, gb)[0]
    roe.fuseConsumers = False
    try:
        b0 = rs.Beam(copyFrom=gb)
        slit.propagate(b0)
    finally:
        roe.fuseConsumers = True
    dev = torch.device('cuda', 0)
    b1 = rs.Beam(copyFrom=gb)
    glo = rs.Beam.empty_like_on_device(b1, dev)
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rec = slit._record()
    rc = lib.xrt_hip_aperture_propagate_f64_dev(ctypes.byref(rec), ctypes.byref(b1.to_struct(dev)),
                                                None, ctypes.byref(glo.to_struct(dev)), stream)
    assert rc != 0 and b'out_global' in lib.xrt_hip_last_error()
    _lib.check(lib.xrt_hip_aperture_propagate_f64_dev(
        ctypes.byref(rec), ctypes.byref(b1.to_struct(dev)), None, None, stream), 'states only')
    b1._h.pop('state', None)
    assert np.array_equal(b1.state, b0.state)


def test_local_beam_made_on_demand():
    """The global beam goes on to the next element and nobody looks at the local one: the pass
    leaves it out (200 instead of 308 B per ray). Looked at later it is made by the pass run
    again -- on the input as it was, whatever an aperture did to its states since -- and the
    element writes it at once from then on."""
    bl, oe, scr, beam = scene(n=50000)
    gb0, lb0, _ = eager(oe, scr, beam)
    slit = ra.RectangularAperture(bl, 'half', [0., 19000., 0.], ('left',), [0.])   # (upstream)
    b1 = rs.Beam(copyFrom=beam)
    gb, lb = oe.reflect(b1)
    again = oe.reflect(gb)                           # the next element takes the global beam
    op = lb.__dict__['_op']
    assert op.state == 'global' and gb.__dict__['_filled'] and not lb.__dict__['_filled']
    assert op in rs._PENDING.optional()
    same(gb, gb0, 'global')
    before = b1.state.copy()
    slit.propagate(b1)                               # the input's states change in place
    assert (b1.state != before).sum() > 100 and not lb.__dict__['_filled']
    rs.flush_pending()                               # (the end of an iteration)
    assert not lb.__dict__['_filled'] and not oe.__dict__.get('_local_beams_wanted')
    same(lb, lb0, 'local', extra=('theta',))
    assert op.state == 'done' and oe.__dict__['_local_beams_wanted']
    assert op not in rs._PENDING.optional()
    gb, lb = oe.reflect(rs.Beam(copyFrom=beam))
    again = oe.reflect(gb)
    assert lb.__dict__['_filled'] and lb.__dict__['_op'].state == 'done'
    same(lb, lb0, 'local, at once', extra=('theta',))
    del again
    # the local beam looked at FIRST: both beams by the one plain pass
    del oe.__dict__['_local_beams_wanted']
    gb, lb = oe.reflect(rs.Beam(copyFrom=beam))
    same(lb, lb0, 'local first', extra=('theta',))
    assert gb.__dict__['_filled'] and not oe.__dict__.get('_local_beams_wanted')


def test_local_beam_on_demand_after_a_contradicted_pass():
    import p1_cases
    g = np.load(os.path.join(p1_cases.GOLDEN, 'g2_toroid_brent.npz'))
    oe = p1_cases.product_oe('g2_toroid_brent', g)
    beam = p1_cases.product_beam(g)
    scr = rsc.Screen(oe.bl, 'after', center=[0, float(g['oe_center'][1]) + 3000., 10.])
    gb0, lb0, img0 = eager(oe, scr, beam)
    gb, lb = oe.reflect(beam)
    gb.to_struct(torch.device('cuda', 0))
    assert lb.__dict__['_op'].state == 'global'
    same(gb, gb0, 'global')
    same(lb, lb0, 'local', extra=('theta',))
    del oe.__dict__['_local_beams_wanted']
    gb, lb = oe.reflect(beam)                        # ... and with the screen in the tail
    img = scr.expose(gb)
    assert not lb.__dict__['_filled']
    same(img, img0, 'image')
    same(lb, lb0, 'local', extra=('theta',))
    same(gb, gb0, 'global')


@pytest.mark.parametrize('amplitudes', [False, True])
def test_plate_local_beams_made_on_demand(amplitudes):
    bl = raycing.BeamLine(azimuth=-0.1)
    mat = rm.Material(('Si', 'O'), quantities=(1, 2), rho=2.2, kind='plate')
    plate = roe.Plate(bl, 'w', center=[20000. * bl.sinAzimuth, 20000. * bl.cosAzimuth, 0.],
                      pitch=1.1, material=mat, t=0.2, wedgeAngle=2e-3, limPhysX=[-4., 5.],
                      limPhysY=[-4., 4.])
    rng = np.random.default_rng(5)
    n = 30000
    beam = rs.Beam(nrays=n, withAmplitudes=amplitudes)
    beam.x[:], beam.z[:] = rng.normal(0, 1.5, n), rng.normal(0, 1.5, n)
    beam.a[:], beam.c[:] = rng.normal(0, 1e-4, n), rng.normal(0, 1e-4, n)
    beam.b[:] = np.sqrt(1 - beam.a**2 - beam.c**2)
    for u, v in (('x', 'y'), ('a', 'b')):
        pu, qv = raycing.rotate_z(getattr(beam, u).copy(), getattr(beam, v).copy(),
                                  bl.cosAzimuth, -bl.sinAzimuth)
        getattr(beam, u)[:], getattr(beam, v)[:] = pu, qv
    beam.E[:] = rng.uniform(7000., 12000., n)
    beam.state[::41] = -2
    roe.fuseConsumers = False
    try:
        g0, a0, b0 = plate.double_refract(rs.Beam(copyFrom=beam))
    finally:
        roe.fuseConsumers = True
    assert type(a0) is rs.Beam
    g1, a1, b1 = plate.double_refract(rs.Beam(copyFrom=beam))
    assert type(a1) is rs.LazyBeam and not a1.__dict__['_filled']
    same(g1, g0, 'global')
    assert not b1.__dict__['_filled']
    same(b1, b0, 'second local', extra=('theta',))
    assert a1.__dict__['_filled'] and plate.__dict__['_local_beams_wanted']
    same(a1, a0, 'first local', extra=('theta',))
    g2, a2, b2 = plate.double_refract(rs.Beam(copyFrom=beam))
    assert type(a2) is rs.Beam
    same(a2, a0, 'first local, at once', extra=('theta',))
    same(g2, g0, 'global, at once')


@pytest.mark.parametrize('amplitudes', [False, True])
def test_dcm_local_beams_made_on_demand(amplitudes):
    """The fused pass over both crystals with the global beam alone (xrt_hip_double_reflect_f64_dev
    with out_local1 = out_local2 = NULL: 200 instead of 416 B per ray), then the two local beams
    by the pass run again; a single crystal likewise."""
    bl = raycing.BeamLine()
    dcm = workloads.cfg3_dcm(bl)
    beam = workloads.synthetic_rays(80000, 11, sa=1e-4, E=(8995., 9005.), amplitudes=amplitudes)
    beam.state[::37] = -2
    beam.x[::29] *= 400.
    roe.fuseConsumers = False
    try:
        g0, a0, b0 = dcm.double_reflect(rs.Beam(copyFrom=beam))
    finally:
        roe.fuseConsumers = True
    assert (g0.state == 1).sum() > 1000 and (g0.state != 1).sum() > 1000
    b1 = rs.Beam(copyFrom=beam)
    g1, a1, b1l = dcm.double_reflect(b1)
    assert type(a1) is rs.LazyBeam and not a1.__dict__['_filled']
    same(g1, g0, 'global')
    b1.state[:] = 3                                  # (the input's states afterwards)
    same(a1, a0, 'first crystal', extra=('theta',))
    assert b1l.__dict__['_filled'] and dcm.__dict__['_local_beams_wanted']
    same(b1l, b0, 'second crystal', extra=('theta',))
    g2, a2, b2 = dcm.double_reflect(rs.Beam(copyFrom=beam))
    assert type(a2) is rs.Beam
    same(g2, g0, 'global, at once')
    same(b2, b0, 'second crystal, at once', extra=('theta',))
    del dcm.__dict__['_local_beams_wanted']
    os.environ['XRT_HIP_DCM_TWO_PASSES'] = '1'       # ... and as two single-crystal passes
    try:
        g3, a3, b3 = dcm.double_reflect(rs.Beam(copyFrom=beam))
        assert type(a3) is rs.LazyBeam
        same(g3, g0, 'global, two passes')
        same(a3, a0, 'first crystal, two passes', extra=('theta',))
        same(b3, b0, 'second crystal, two passes', extra=('theta',))
    finally:
        del os.environ['XRT_HIP_DCM_TWO_PASSES']


def test_beams_of_a_recorded_iteration_looked_at_after_its_replays():
    """Inside a HIP graph the source is in the head of the pass as well (the graph moves the call
    cell as its last node); the beams that were left out, asked for after the replays, are the
    ones of the last replay."""
    from xrt_amd import graphs

    def iteration(bl, amp):
        src = bl.source.shine(withAmplitudes=amp)
        gb, lb = bl.mirror.reflect(src)
        img = bl.screen.expose(gb)
        rs.flush_pending()
        return src, gb, lb, img
    bl0, amp = source_scene(n=20000)
    roe.fuseConsumers = False
    try:
        for _ in range(4):
            src0, gb0, lb0, img0 = iteration(bl0, amp)
    finally:
        roe.fuseConsumers = True
    bl, amp = source_scene(n=20000)
    iteration(bl, amp)                           # (eagerly once: cells and workspaces exist)
    rec = graphs.IterationGraph(lambda: iteration(bl, amp))
    src, gb, lb, img = rec.result
    assert src.__dict__['_op'].state == 'inflight' and not lb.__dict__['_filled']
    for _ in range(3):
        rec.replay()
    torch.cuda.synchronize()
    assert bl.source._calls == bl0.source._calls == 4
    same(img, img0, 'image of the last replay')
    same(lb, lb0, 'local beam, made afterwards', extra=('theta',))
    same(src, src0, 'source beam, made afterwards')
    same(gb, gb0, 'global beam, made afterwards')
    rec.close()
    nxt0, nxt = bl0.source.shine(), bl.source.shine()       # ... and the sequence goes on
    same(nxt, nxt0, 'next shine')


def test_an_aperture_on_the_source_beam_after_the_fused_pass():
    """The pass made the source's rays in its registers; afterwards the script stops part of the
    source beam with a slit (states change in place) and only then looks at the mirror's local
    beam: it is the one of the rays as they were."""
    bl, amp = source_scene(n=30000)
    slit = ra.RectangularAperture(bl, 'slit', [0, 15000., 0], ('left', 'right'), [-0.5, 0.5])
    roe.fuseConsumers = False
    try:
        bl.source._calls = 0
        s0 = bl.source.shine()
        g0, l0 = bl.mirror.reflect(s0)
        i0 = bl.screen.expose(g0)
        slit.propagate(s0)
    finally:
        roe.fuseConsumers = True
    bl.source._calls = 0
    s1 = bl.source.shine()
    g1, l1 = bl.mirror.reflect(s1)
    i1 = bl.screen.expose(g1)
    assert i1.nrays == 30000 and s1.__dict__['_op'].state == 'inflight'
    slit.propagate(s1)
    assert (s1.state < 0).sum() > 100 and not l1.__dict__['_filled']
    same(i1, i0, 'image')
    same(s1, s0, 'source after the slit')
    same(l1, l0, 'local', extra=('theta',))
    same(g1, g0, 'global')


def test_c_abi_rules_for_a_missing_local_beam():
    """out_local NULL through the C ABI itself: a crystal pass leaves its local beam out (round
    5) and gives the global beam of the full pass; a layered mirror refuses; the fused DCM takes
    both local beams or neither."""
    import ctypes
    from xrt_amd import _lib, hipcalls
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    stream = hipcalls.stream_ptr()
    bl = raycing.BeamLine()
    si = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
    thB = float(np.ravel(si.get_Bragg_angle(9000.) - si.get_dtheta(9000.))[0])
    xt = roe.OE(bl, 'xtal', center=[0, 20000., 0], pitch=thB, material=si, limPhysX=[-10, 10],
                limPhysY=[-50, 50])
    beam = workloads.synthetic_rays(30000, 3, sa=1e-4, E=(8995., 9005.), amplitudes=True)
    roe.fuseConsumers = False
    try:
        gb0, lb0 = xt.reflect(rs.Beam(copyFrom=beam))
    finally:
        roe.fuseConsumers = True
    assert (gb0.state == 1).sum() > 1000

    def run(oe, material):
        p = oe._make_pass(oe.pitch + getattr(oe, 'bragg', 0), oe.roll + oe.positionRoll, oe.yaw,
                          oe.dx)
        ms = oe._material_struct(material, True, dev, beam)
        out = rs.Beam.empty_like_on_device(beam, dev)
        ws = hipcalls.workspace(dev, lib.xrt_hip_reflect_workspace_bytes(beam.nrays), 'reflect')
        s_in = beam.to_struct(dev)
        rc = lib.xrt_hip_reflect_pass_f64_dev(
            ctypes.byref(p), ctypes.byref(ms), ctypes.byref(s_in), ctypes.byref(s_in), None,
            ctypes.byref(out.to_struct(dev)), None, ctypes.c_void_p(ws.data_ptr()), ws.numel(),
            stream, None, None)
        return rc, out
    rc, gb = run(xt, si)
    assert rc == 0, lib.xrt_hip_last_error()
    same(gb, gb0, 'crystal, global beam alone')
    coated = roe.OE(bl, 'coated', center=[0, 20000., 0], pitch=4e-3, limPhysX=[-10, 10],
                    limPhysY=[-300, 300],
                    material=rm.Coated(coating=rm.Material('Rh', rho=12.41), cThickness=300.,
                                       substrate=rm.Material('Si', rho=2.33),
                                       surfaceRoughness=3.))
    rc, _ = run(coated, coated.material)
    assert rc != 0 and b'layered' in lib.xrt_hip_last_error()
    # the fused DCM: one local beam without the other is refused
    dcm = workloads.cfg3_dcm(bl)
    first, second = dcm._own_angles(False), dcm._own_angles(True)
    p1 = dcm._make_pass(*first[:4], out_to_global=False)
    p2 = dcm._make_pass(*second, is2ndXtal=True, in_is_global=False, good_mode=1,
                        out_to_global=True, zero_local_not_entering=True)
    m1, m2 = (dcm._material_struct(m, True, dev) for m in (dcm.material, dcm.material2))
    lo1, gb2 = (rs.Beam.empty_like_on_device(beam, dev) for _ in range(2))
    theta = torch.empty(beam.nrays, dtype=torch.float64, device=dev)
    ws = hipcalls.workspace(dev, lib.xrt_hip_reflect_workspace_bytes(beam.nrays), 'reflect')
    rc = lib.xrt_hip_double_reflect_f64_dev(
        ctypes.byref(p1), ctypes.byref(m1), ctypes.byref(p2), ctypes.byref(m2),
        ctypes.byref(beam.to_struct(dev)), ctypes.byref(lo1.to_struct(dev)), None,
        ctypes.byref(gb2.to_struct(dev)), ctypes.c_void_p(theta.data_ptr()), None,
        ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream, None)
    assert rc != 0 and b'both local beams or neither' in lib.xrt_hip_last_error()


# ---- ADVICE r5 -----------------------------------------------------------------------------------
def test_screen_on_lazy_beams_that_are_not_an_elements_global_beam():
    """Screen.expose of a pending beam that is NOT the global beam of a deferred OE.reflect: a
    device source's beam (source -> screen is the most common script pattern), an aperture's
    local beam made on demand, a DCM's local beams on demand. Each is simply made and imaged."""
    bl, _ = source_scene(n=30000)
    roe.fuseConsumers = False
    try:
        bl.source._calls = 0
        i0 = bl.screen.expose(bl.source.shine())
    finally:
        roe.fuseConsumers = True
    bl.source._calls = 0
    src = bl.source.shine()
    assert type(src) is rs.LazyBeam and not src.__dict__['_filled']
    same(bl.screen.expose(src), i0, 'source -> screen')
    # an aperture's local beam
    slit = ra.RectangularAperture(bl, 'slit', [0, 15000., 0], ('left', 'right'), [-0.5, 0.5])
    roe.fuseConsumers = False
    try:
        bl.source._calls = 0
        l0 = slit.propagate(bl.source.shine())
        j0 = bl.screen.expose(l0)
    finally:
        roe.fuseConsumers = True
    bl.source._calls = 0
    l1 = slit.propagate(bl.source.shine())
    assert type(l1) is rs.LazyBeam and not l1.__dict__['_filled']
    same(bl.screen.expose(l1), j0, 'aperture local -> screen')
    # a DCM's local beams
    dcm = workloads.cfg3_dcm(raycing.BeamLine())
    beam = workloads.synthetic_rays(20000, 11, sa=1e-4, E=(8995., 9005.))
    roe.fuseConsumers = False
    try:
        g0, a0, b0 = dcm.double_reflect(rs.Beam(copyFrom=beam))
        k0 = bl.screen.expose(b0)
    finally:
        roe.fuseConsumers = True
    g1, a1, b1 = dcm.double_reflect(rs.Beam(copyFrom=beam))
    assert type(b1) is rs.LazyBeam and not b1.__dict__['_filled']
    same(bl.screen.expose(b1), k0, 'DCM local -> screen')


def test_out_beams_reused_in_place_wait_for_their_readers():
    """reflect -> propagate (lazy local beam of the slit, reads gb's arrays) -> reflect(out=the
    previous pair) overwrites those arrays: the slit's local beam is made first, from the data of
    ITS iteration. Same for DCM.double_reflect(out=...)."""
    bl, oe, scr, beam = scene(n=50000)
    slit = ra.RectangularAperture(bl, 'slit', [0, 25000., 0], ('left', 'right'), [-0.2, 0.2])
    other = workloads.synthetic_rays(50000, 8, amplitudes=True)
    roe.fuseConsumers = False
    try:
        gb0, lb0 = oe.reflect(beam)
        l0 = slit.propagate(gb0)
    finally:
        roe.fuseConsumers = True
    gb, lb = oe.reflect(beam)
    gb.dev('x')                                      # (filled LazyBeams)
    lb.dev('x')
    loc = slit.propagate(gb)
    assert type(loc) is rs.LazyBeam and not loc.__dict__['_filled']
    gb2, lb2 = oe.reflect(other, out=(gb, lb))
    gb2.dev('x')                                     # the launch that overwrites gb's arrays
    same(loc, l0, 'slit local after out= reuse')
    # the DCM
    dcm = workloads.cfg3_dcm(raycing.BeamLine())
    b3 = workloads.synthetic_rays(30000, 11, sa=1e-4, E=(8995., 9005.))
    b4 = workloads.synthetic_rays(30000, 12, sa=1e-4, E=(8995., 9005.))
    dslit = ra.RectangularAperture(dcm.bl, 'dslit', [0, 25000., 20.], ('left', 'right'), [-0.05, 0.05])
    roe.fuseConsumers = False
    try:
        trio = dcm.double_reflect(b3)
        m0 = dslit.propagate(rs.Beam(copyFrom=trio[0]))
    finally:
        roe.fuseConsumers = True
    dcm.__dict__['_local_beams_wanted'] = True       # (out= needs the three real beams)
    trio = dcm.double_reflect(b3)
    m1 = dslit.propagate(trio[0])
    assert type(m1) is rs.LazyBeam and not m1.__dict__['_filled']
    dcm.double_reflect(b4, out=trio)
    same(m1, m0, 'slit local after DCM out= reuse')


def test_multiple_reflect_refuses_no_bounce_at_all():
    bl, oe, scr, beam = scene(n=1000)
    with pytest.raises(ValueError):
        oe.multiple_reflect(beam, maxReflections=0)


# ---- the plot in the tail of the pass (round 6) -----------------------------------------------------
def _plot(bins=256, rayFlag=(1,), **kw):
    """The XYCPlot of workloads.e2e_beamline with fixed limits around the focus."""
    return xrtp.XYCPlot('focus', rayFlag,
                        xaxis=xrtp.XYCAxis('x', 'mm', limits=[-0.4, 0.4], bins=bins),
                        yaxis=xrtp.XYCAxis('z', 'mm', limits=[-0.05, 0.05], bins=bins),
                        caxis=xrtp.XYCAxis('energy', 'eV', limits=[8989., 9011.], bins=128), **kw)


def _same_plot(p, q, what):
    """Bins and counts identical; sums to rounding (another order of addition)."""
    for name in ('total2D', 'total2D_RGB'):
        a, b = getattr(p, name), getattr(q, name)
        assert (a != 0).sum() > 50, (what, name)
        assert np.array_equal(a != 0, b != 0), (what, name)
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (what, name, np.abs(a - b).max())
    for axis in ('xaxis', 'yaxis', 'caxis'):
        a, b = getattr(p, axis).total1D4, getattr(q, axis).total1D4
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (what, axis)
    for name in ('nRaysAll', 'nRaysSelected', 'nRaysAlive', 'nRaysGood', 'nRaysOut', 'nRaysOver',
                 'nRaysDead'):
        assert getattr(p, name) == getattr(q, name), (what, name)
    for name in ('intensity', 'intensityInRange'):
        assert abs(getattr(p, name) - getattr(q, name)) <= 1e-12 * abs(getattr(q, name)), (what, name)


def _plot_chain(bl, amp, fuse, plot, look=(), source=True, beam=None):
    """source -> mirror -> screen -> plot as run_ray_tracing's iteration does it; *look*: beams
    the script looks at AFTER the plot."""
    old = roe.fuseConsumers
    roe.fuseConsumers = fuse
    try:
        if source:
            bl.source._calls = 0
            src = bl.source.shine(withAmplitudes=amp)
        else:
            src = beam
        gb, lb = bl.mirror.reflect(src)
        img = bl.screen.expose(gb)
        xrtr.accumulate_plot(plot, {'focus': img})
        rs.flush_pending()
        beams = dict(src=src, gb=gb, lb=lb, img=img)
        for name in look:
            beams[name].nrays
    finally:
        roe.fuseConsumers = old
    return beams


@pytest.mark.parametrize('amplitudes', [False, True])
@pytest.mark.parametrize('bins', [64, 256])
def test_the_plot_rides_in_the_tail_of_the_pass(amplitudes, bins):
    """accumulate_plot of the image of a pending pass: weight, hue and bins of every ray come from
    the registers of the ray kernel (reflect_fused_gen_scr_plot / reflect_fused_scr_plot), the
    image is not written; the plot equals the one the separate launches fill, and every beam
    that was left out is made on demand, bit-identical."""
    bl, amp = source_scene(n=300000, amplitudes=amplitudes)
    p0, p1 = _plot(bins), _plot(bins)
    b0 = _plot_chain(bl, amp, False, p0)
    b1 = _plot_chain(bl, amp, True, p1)
    op = b1['gb'].__dict__['_op']
    assert op.state == 'imaged' and not b1['img'].__dict__['_filled']
    assert not b1['gb'].__dict__['_filled'] and b1['src'].__dict__['_op'].state == 'inflight'
    _same_plot(p1, p0, 'source in the head')
    assert p1.nRaysAll == 300000 and p1.iteration == 1
    # the image nobody had asked for: the same rays, and the screen remembers
    same(b1['img'], b0['img'], 'image on demand')
    assert bl.screen.__dict__['_image_wanted']
    same(b1['lb'], b0['lb'], 'local on demand', extra=('theta',))
    same(b1['gb'], b0['gb'], 'global on demand')
    same(b1['src'], b0['src'], 'source on demand')
    # ... from now on the pass writes the image as well (and the other beams that were wanted)
    p2 = _plot(bins)
    b2 = _plot_chain(bl, amp, True, p2)
    assert b2['img'].__dict__['_filled'] and b2['gb'].__dict__['_filled']
    _same_plot(p2, p0, 'image kept')
    same(b2['img'], b0['img'], 'image written by the pass')
    # a resident beam instead of the device source (reflect_fused_scr_plot)
    bl, amp = source_scene(n=300000, amplitudes=amplitudes)
    rays = workloads.synthetic_rays(250000, 5, amplitudes=amplitudes)
    rays.state[::97] = -3
    rays.state[5::101] = 2
    rays.x[::53] *= 300.
    q0, q1 = _plot(bins, rayFlag=(1, 2, 3, -1)), _plot(bins, rayFlag=(1, 2, 3, -1))
    c0 = _plot_chain(bl, amp, False, q0, source=False, beam=rays)
    c1 = _plot_chain(bl, amp, True, q1, source=False, beam=rays)
    assert c1['gb'].__dict__['_op'].state == 'imaged' and not c1['img'].__dict__['_filled']
    _same_plot(q1, q0, 'resident beam')
    same(c1['img'], c0['img'], 'image on demand, resident beam')


def test_plots_that_do_not_ride():
    """Automatic limits, a second plot of the same beam, another beam's states, an axis the tail
    does not know: the usual route, the same numbers."""
    bl, amp = source_scene(n=100000)
    p0 = _plot()
    _plot_chain(bl, amp, False, p0)
    auto = xrtp.XYCPlot('focus', (1,), xaxis=xrtp.XYCAxis('x', 'mm', bins=64),
                        yaxis=xrtp.XYCAxis('z', 'mm', bins=64),
                        caxis=xrtp.XYCAxis('energy', 'eV', bins=32))
    b = _plot_chain(bl, amp, True, auto)
    assert b['img'].__dict__['_filled'] and auto.xaxis.limits is not None
    # two plots of one beam: neither rides (the image is read twice anyway)
    pa, pb = _plot(), _plot(128)
    roe.fuseConsumers = True
    bl.source._calls = 0
    gb, lb = bl.mirror.reflect(bl.source.shine())
    img = bl.screen.expose(gb)
    for plot in (pa, pb):
        xrtr.accumulate_plot(plot, {'focus': img}, sole=False)
    assert img.__dict__['_filled']
    _same_plot(pa, p0, 'one of two plots')


def test_a_contradicted_pass_with_a_plot_in_its_tail():
    """Rays all over the place: the optimistic pass is contradicted (a ray's largest direction
    cosine is not y), the redo makes the plot's records from the real image (reflect_redo_scr);
    likewise for a resident beam whose batch statistics ask for Brent's method."""
    def plot():
        return xrtp.XYCPlot('focus', (1, 2, 3, -1),
                            xaxis=xrtp.XYCAxis('x', 'mm', limits=[-3000., 3000.], bins=96),
                            yaxis=xrtp.XYCAxis('z', 'mm', limits=[-200., 200.], bins=160),
                            caxis=xrtp.XYCAxis('energy', 'eV', limits=[8989., 9011.], bins=64))
    bl, amp = source_scene(n=40000, wide=True)
    p0, p1 = plot(), plot()
    b0 = _plot_chain(bl, amp, False, p0)
    b1 = _plot_chain(bl, amp, True, p1)
    assert b1['gb'].__dict__['_op'].state == 'imaged' and not b1['img'].__dict__['_filled']
    _same_plot(p1, p0, 'wide source, redone')
    same(b1['img'], b0['img'], 'image on demand after the redo')
    same(b1['gb'], b0['gb'], 'global')
    same(b1['src'], b0['src'], 'source')
    # a resident beam, Brent's method
    rays = workloads.synthetic_rays(60000, 9)
    rays.c[::7] = 0.3                       # steep rays: dz at the far bracket end is large
    rays.b[:] = np.sqrt(1 - rays.a**2 - rays.c**2)
    bl, amp = source_scene(n=1000)
    info = {}
    bl.mirror.reflect(rays, _info=info)
    q0, q1 = plot(), plot()
    c0 = _plot_chain(bl, amp, False, q0, source=False, beam=rays)
    c1 = _plot_chain(bl, amp, True, q1, source=False, beam=rays)
    assert c1['gb'].__dict__['_op'].state == 'imaged' and not c1['img'].__dict__['_filled']
    for name in ('total2D', 'total2D_RGB'):
        a, b = getattr(q1, name), getattr(q0, name)
        assert np.array_equal(a != 0, b != 0) and np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
    assert q1.nRaysGood == q0.nRaysGood and q1.nRaysDead == q0.nRaysDead
    same(c1['img'], c0['img'], 'image on demand, resident beam')


def test_the_fused_chain_against_the_oracle_directly():
    """the fused kernels were pinned only through the immediate launches. Here the
    whole chain at 1e6 rays -- device source -> toroid mirror -> screen -> plot as ONE pass --
    against the oracle: geosource_np -> reflect_np.oe_reflect -> elements_np.screen_expose ->
    numpy histograms (reference: geoms.py:420-535, oes/reflect.py:18-163, screens.py:226-302,
    multipro.py:316-361). States bit for bit, geometry 1e-12, the plot 1e-12."""
    from oracle import elements_np as en, geosource_np as og, reflect_np as rn
    from oracle.adapters import oracle_params
    n = 1_000_000
    bl = raycing.BeamLine()
    bl.source = rs.GeometricSource(bl, 'source', nrays=n, dx=0.1, dz=0.1, dxprime=2e-4,
                                   dzprime=2e-5, distE='flat', energies=(8990., 9010.),
                                   polarization='h', rng='device', seed=23)
    bl.mirror = workloads.cfg2_toroid(bl)
    bl.screen = rsc.Screen(bl, 'focus', center=[0, 20000. + 10000. * np.cos(8e-3),
                                                10000. * np.sin(8e-3)])
    plot = _plot(256, rayFlag=(1, 2))
    b = _plot_chain(bl, False, True, plot)
    assert b['gb'].__dict__['_op'].state == 'imaged' and not b['img'].__dict__['_filled']
    # the oracle's chain
    born = og.shine(og.Spec(n, seed=23, call=0, dx=0.1, dz=0.1, dxprime=2e-4, dzprime=2e-5,
                            distE='flat', energies=(8990., 9010.), azimuth=(1., 0.),
                            center=(0, 0, 0)))
    ob = rn.Beam(n, with_amplitudes=False)
    for f in ob.fields():
        if f in born:
            setattr(ob, f, np.array(born[f]))
    # (the normal laws of the device generator agree with the oracle's libm to 1e-14: the rays
    # the GPU made are the input of the oracle's pass, their uniform laws checked bit for bit)
    src = b['src']
    assert np.array_equal(src.peek('E'), born['E'])
    for f in ('x', 'z', 'a', 'c'):
        assert np.abs(src.peek(f) - born[f]).max() <= 1e-14 * np.abs(born[f]).max(), f
    for f in ob.fields():
        setattr(ob, f, np.array(src.peek(f)))
    ogb, olb = rn.oe_reflect(oracle_params(bl.mirror), ob)
    oimg = en.screen_expose(ogb, (bl.screen.x, bl.screen.y, bl.screen.z), bl.screen.center,
                            bl.screen.lostNum)
    img = b['img']
    assert np.array_equal(img.state, oimg.state)
    for f in ('x', 'z', 'path'):
        r = getattr(oimg, f)
        assert np.abs(getattr(img, f) - r).max() <= 1e-12 * np.abs(r).max(), f
    sel = (oimg.state == 1) | (oimg.state == 2)
    w = (oimg.Jss + oimg.Jpp)[sel]
    h2, _, _ = np.histogram2d(oimg.z[sel], oimg.x[sel], bins=[256, 256],
                              range=[[-0.05, 0.05], [-0.4, 0.4]], weights=w)
    assert h2.sum() > 0.2 * w.sum()
    assert np.abs(plot.total2D - h2).max() <= 1e-12 * h2.max()
    hx, _ = np.histogram(oimg.x[sel], bins=256, range=(-0.4, 0.4), weights=w)
    hc, _ = np.histogram(oimg.E[sel], bins=128, range=(8989., 9011.), weights=w)
    assert np.abs(plot.total1D_x - hx).max() <= 1e-12 * hx.max()
    assert np.abs(plot.total1D_c - hc).max() <= 1e-12 * hc.max()
    assert plot.nRaysGood == int((oimg.state == 1).sum())
    assert plot.nRaysOut == int((oimg.state == 2).sum())
    assert plot.nRaysDead == int((oimg.state < 0).sum())


# ---- apertures in the tail of the pass -------------------------------------------------------
def _eager_chain(oe, aps, scr, beam):
    """element -> apertures -> screen, every step its own launch."""
    old = roe.fuseConsumers
    roe.fuseConsumers = False
    try:
        gb, lb = oe.reflect(beam)
        locs = [a.propagate(gb) for a in aps]
        img = scr.expose(gb) if scr is not None else None
    finally:
        roe.fuseConsumers = old
    return gb, lb, locs, img


def _forget(*objs):
    for o in objs:
        for key in ('_local_beam_wanted', '_global_beam_wanted', '_local_beams_wanted',
                    '_image_wanted'):
            o.__dict__.pop(key, None)


@pytest.mark.parametrize('amplitudes', [False, True])
@pytest.mark.parametrize('pair', [(0, 1), (2, 3), (3,)])
def test_apertures_ride_in_the_tail_of_the_pass(amplitudes, pair):
    """aperture.propagate(gb) while the mirror's pass is pending (reference apertures.py:334-413
    after oes/reflect.py): the marks are made on the ray in registers, between the element and
    the screen; image, both beams of the element and -- on demand -- the apertures' local beams
    have the bits of the separate launches."""
    bl, oe, scr, beam = scene(n=80000, amplitudes=amplitudes)
    kinds = _apertures(bl, eager(oe, scr, beam)[0])
    aps = [kinds[k] for k in pair]
    for k, a in enumerate(aps):          # (every aperture its own number, as in a beamline)
        a.lostNum = -(40 + k)
    gb0, lb0, locs0, img0 = _eager_chain(oe, aps, scr, beam)
    assert 1000 < (gb0.state == aps[-1].lostNum).sum()
    _forget(oe, scr, *aps)
    gb, lb = oe.reflect(beam)
    op = gb.__dict__['_op']
    locs = [a.propagate(gb) for a in aps]
    assert op.state == 'pending' and len(op.apertures) == len(aps)
    assert all(type(b) is rs.LazyBeam and not b.__dict__['_filled'] for b in locs)
    img = scr.expose(gb)
    same(img, img0, 'image')
    assert op.state == 'imaged' and not gb.__dict__['_filled']
    same(lb, lb0, 'local', extra=('theta',))
    same(gb, gb0, 'global')              # (made again, the marks in it)
    for a, b, b0 in zip(aps, locs, locs0):
        same(b, b0, 'beam at ' + a.name)
        assert a.__dict__['_local_beam_wanted']
    assert op.state == 'done' and op.beam is None
    # an aperture whose beam has been looked at takes its own launch from then on
    gb, lb = oe.reflect(beam)
    again = aps[0].propagate(gb)
    assert type(again.__dict__['_op']) is ra._DeferredLocal and gb.__dict__['_filled']
    same(again, locs0[0], 'beam at the aperture, its own launch')


def test_apertures_in_the_tail_without_a_screen():
    """mirror -> slit -> next element: the marked global beam is all the pass writes."""
    bl, oe, scr, beam = scene(n=50000)
    kinds = _apertures(bl, eager(oe, scr, beam)[0])
    aps = kinds[:2]
    aps[1].lostNum = -41
    gb0, lb0, locs0, _ = _eager_chain(oe, aps, None, beam)
    _forget(oe, scr, *aps)
    gb, lb = oe.reflect(beam)
    op = gb.__dict__['_op']
    locs = [a.propagate(gb) for a in aps]
    del locs                              # (nobody looks at them: a beamline's slits)
    assert op.state == 'pending'
    assert np.array_equal(gb.state, gb0.state)
    assert op.state == 'global' and not lb.__dict__['_filled']
    same(gb, gb0, 'global')
    same(lb, lb0, 'local', extra=('theta',))
    assert op.state == 'done'
    # the local beam first: still one launch with the marks
    gb, lb = oe.reflect(beam)
    op = gb.__dict__['_op']
    aps[0].propagate(gb), aps[1].propagate(gb)
    same(lb, lb0, 'local first', extra=('theta',))
    assert op.state == 'done'
    same(gb, gb0, 'global then')
    # a third aperture, a polygon and an aperture after the screen take their own launches
    gb, lb = oe.reflect(beam)
    op = gb.__dict__['_op']
    kinds[0].propagate(gb), kinds[1].propagate(gb)
    assert op.state == 'pending'
    kinds[2].propagate(gb)
    assert op.state != 'pending' and len(op.apertures) == 2
    gb, lb = oe.reflect(beam)
    kinds[4].propagate(gb)
    assert gb.__dict__['_op'].state != 'pending'
    gb, lb = oe.reflect(beam)
    img = scr.expose(gb)
    kinds[0].propagate(gb)
    assert not gb.__dict__['_op'].apertures
    g1, _, _, i1 = _eager_chain(oe, [], scr, beam)
    same(img, i1, 'the image from before the aperture')


def test_apertures_in_the_tail_of_a_contradicted_pass_and_of_other_kernels():
    import p1_cases
    g = np.load(os.path.join(p1_cases.GOLDEN, 'g2_toroid_brent.npz'))
    oe = p1_cases.product_oe('g2_toroid_brent', g)
    beam = p1_cases.product_beam(g)
    y0 = float(g['oe_center'][1])
    scr = rsc.Screen(oe.bl, 'after', center=[0, y0 + 3000., 10.])
    gb0 = eager(oe, scr, beam)[0]
    ok = gb0.state == 1
    mid = np.median(gb0.x[ok] + gb0.a[ok] / gb0.b[ok] * (y0 + 1000. - gb0.y[ok]))
    slit = ra.RectangularAperture(oe.bl, 'slit', [0, y0 + 1000., 0], ('left',), [mid])
    gb0, lb0, locs0, img0 = _eager_chain(oe, [slit], scr, beam)
    assert 10 < (gb0.state == slit.lostNum).sum() < ok.sum()
    _forget(oe, scr, slit)
    gb, lb = oe.reflect(beam)
    loc = slit.propagate(gb)
    assert gb.__dict__['_op'].state == 'pending'
    img = scr.expose(gb)
    same(img, img0, 'image after the redo')
    same(gb, gb0, 'global after the redo')
    same(loc, locs0[0], 'beam at the slit')
    same(lb, lb0, 'local', extra=('theta',))
    # a Bragg crystal (no lean kernel): the same call, the apertures' own launches inside it
    bl = raycing.BeamLine()
    si = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
    thB = float(np.ravel(si.get_Bragg_angle(9000.) - si.get_dtheta(9000.))[0])
    xt = roe.OE(bl, 'xtal', center=[0, 20000., 0], pitch=thB, material=si, limPhysX=[-10, 10],
                limPhysY=[-50, 50])
    scr = rsc.Screen(bl, 'after', center=[0, 21000., 1000. * np.tan(2 * thB)])
    pipe = ra.RoundAperture(bl, 'pipe', [0, 20500., 500. * np.tan(2 * thB)], r=2.)
    beam = workloads.synthetic_rays(60000, 3, sa=1e-4, E=(8995., 9005.), amplitudes=True)
    gb0, lb0, locs0, img0 = _eager_chain(xt, [pipe], scr, beam)
    assert 1000 < (gb0.state == pipe.lostNum).sum() < 58000
    _forget(xt, scr, pipe)
    gb, lb = xt.reflect(beam)
    loc = pipe.propagate(gb)
    assert gb.__dict__['_op'].state == 'pending'
    img = scr.expose(gb)
    same(img, img0, 'image')
    same(gb, gb0, 'global')
    same(loc, locs0[0], 'beam in the pipe')
    same(lb, lb0, 'local', extra=('theta',))


@pytest.mark.parametrize('amplitudes', [False, True])
def test_source_mirror_slit_screen_in_one_pass(amplitudes):
    bl, amplitudes = source_scene(amplitudes=amplitudes)
    at = [0, 20000. + 5000. * np.cos(8e-3), 5000. * np.sin(8e-3)]
    slit = ra.RectangularAperture(bl, 'slit', at, ('left', 'top'), [-0.2, 0.05])

    def chain(fuse):
        old = roe.fuseConsumers
        roe.fuseConsumers = fuse
        try:
            bl.source._calls = 0
            src = bl.source.shine(withAmplitudes=amplitudes)
            gb, lb = bl.mirror.reflect(src)
            loc = slit.propagate(gb)
            img = bl.screen.expose(gb)
            img.nrays
        finally:
            roe.fuseConsumers = old
        return src, gb, lb, loc, img
    s0, g0, l0, a0, i0 = chain(False)
    assert 1000 < (g0.state == slit.lostNum).sum() < 0.9 * g0.nrays
    _forget(bl.mirror, bl.screen, slit, bl.source)
    s1, g1, l1, a1, i1 = chain(True)
    assert not g1.__dict__['_filled'] and not s1.__dict__['_filled']
    same(i1, i0, 'image')
    same(g1, g0, 'global')
    same(a1, a0, 'beam at the slit')
    same(l1, l0, 'local', extra=('theta',))
    same(s1, s0, 'source')


def test_c_abi_tail_record():
    """xrt_hip_reflect_tail_f64_dev directly: what rode where (*fused*), and what it refuses."""
    import ctypes
    from xrt_amd import _lib, _structs, hipcalls
    bl, oe, scr, beam = scene(n=30000)
    kinds = _apertures(bl, eager(oe, scr, beam)[0])
    aps = kinds[:2]
    aps[1].lostNum = -41
    gb0, lb0, _, img0 = _eager_chain(oe, aps, scr, beam)
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    p = oe._make_pass(oe.pitch, oe.roll + oe.positionRoll, oe.yaw, oe.dx)
    ms = oe._material_struct(oe.material, True, dev, beam)
    theta = torch.empty(beam.nrays, dtype=torch.float64, device=dev)
    ws = hipcalls.workspace(dev, lib.xrt_hip_reflect_workspace_bytes(beam.nrays), 'reflect')
    s_in = beam.to_struct(dev)
    srec = scr._record(False)

    def call(n_ap, screen, keep, polygon=False):
        lb, gb, img = (rs.Beam.empty_like_on_device(beam, dev) for _ in range(3))
        tail = _structs.Tail()
        tail.n_apertures, tail.keep_screen = n_ap, 1
        for k, a in enumerate(aps[:min(n_ap, 2)]):
            tail.aperture[k] = a._record()
        if polygon:
            tail.aperture[0] = kinds[4]._record()
        if screen:
            tail.screen, tail.out_screen = ctypes.addressof(srec), \
                ctypes.addressof(img.to_struct(dev))
        fused = ctypes.c_int(-1)
        rc = lib.xrt_hip_reflect_tail_f64_dev(
            None, ctypes.byref(p), ctypes.byref(ms), ctypes.byref(s_in), ctypes.byref(s_in),
            ctypes.byref(lb.to_struct(dev)), ctypes.byref(gb.to_struct(dev)),
            ctypes.c_void_p(theta.data_ptr()), ctypes.byref(tail), keep,
            ctypes.c_void_p(ws.data_ptr()), ws.numel(),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(fused))
        return rc, fused.value, lb, gb, img
    rc, fused, lb, gb, img = call(2, True, 1)
    assert rc == 0 and fused == 9
    same(img, img0, 'image')
    same(gb, gb0, 'global')
    same(lb, lb0, 'local')
    rc, fused, lb, gb, img = call(2, False, 0)      # (no screen: the global beam is kept anyway)
    assert rc == 0 and fused == 8
    same(gb, gb0, 'global, no screen')
    rc, fused, lb, gb, img = call(0, True, 1)
    assert rc == 0 and fused == 1
    assert call(3, True, 1)[0] != 0 and b'0 to 2' in lib.xrt_hip_last_error()
    assert call(1, True, 1, polygon=True)[0] != 0 and b'polygon' in lib.xrt_hip_last_error()


# ---- both faces of a plate in one pass -------------------------------------------------------
@pytest.mark.parametrize('amplitudes', [False, True])
@pytest.mark.parametrize('pitch', [1.1, np.pi / 2])
def test_both_faces_of_a_plate_in_one_pass(amplitudes, pitch, monkeypatch):
    """Plate.double_refract (refractive.py:171-235) as ONE kernel with the beam inside the
    plate in registers (reflect_fused_plate2) against its two passes: the same bits in all
    three beams, with rays dead on arrival, rays that miss the plate and rays lost in it; the
    same when the exact sequence is forced."""
    bl = raycing.BeamLine(azimuth=-0.1)
    mat = rm.Material(('Si', 'O'), quantities=(1, 2), rho=2.2, kind='plate')
    plate = roe.Plate(bl, 'w', center=[20000. * bl.sinAzimuth, 20000. * bl.cosAzimuth, 0.],
                      pitch=pitch, material=mat, t=0.2, wedgeAngle=2e-3 if pitch < 1.5 else 0.,
                      limPhysX=[-4., 5.], limPhysY=[-4., 4.])
    rng = np.random.default_rng(5)
    n = 100000
    beam = rs.Beam(nrays=n, withAmplitudes=amplitudes)
    beam.x[:], beam.z[:] = rng.normal(0, 2.5, n), rng.normal(0, 2.5, n)
    beam.a[:], beam.c[:] = rng.normal(0, 1e-4, n), rng.normal(0, 1e-4, n)
    beam.b[:] = np.sqrt(1 - beam.a**2 - beam.c**2)
    for u, v in (('x', 'y'), ('a', 'b')):
        pu, qv = raycing.rotate_z(getattr(beam, u).copy(), getattr(beam, v).copy(),
                                  bl.cosAzimuth, -bl.sinAzimuth)
        getattr(beam, u)[:], getattr(beam, v)[:] = pu, qv
    beam.E[:] = rng.uniform(7000., 12000., n)
    beam.state[::41] = -2
    beam.state[7::43] = 2
    if amplitudes:
        ang = rng.uniform(0, np.pi, n)
        beam.Es[:], beam.Ep[:] = np.cos(ang), np.sin(ang) * np.exp(1j * rng.uniform(-3, 3, n))
        beam.Jss[:], beam.Jpp[:] = np.abs(beam.Es)**2, np.abs(beam.Ep)**2
        beam.Jsp[:] = beam.Es * np.conj(beam.Ep)
    roe.fuseConsumers = False
    try:
        monkeypatch.setenv('XRT_HIP_DCM_TWO_PASSES', '1')
        g0, a0, b0 = plate.double_refract(rs.Beam(copyFrom=beam))
        monkeypatch.delenv('XRT_HIP_DCM_TWO_PASSES')
        # (the exit face asks for Brent's method: the first pass of a new plate learns that
        # from being redone, the element remembers -- xrt_hip_pass.method_hint)
        first, timing = {}, {}
        g1, a1, b1 = plate.double_reflect(rs.Beam(copyFrom=beam), fromVacuum1=True,
                                          fromVacuum2=False, _timing=first)
        same(g1, g0, 'global, first call')
        g1, a1, b1 = plate.double_reflect(rs.Beam(copyFrom=beam), fromVacuum1=True,
                                          fromVacuum2=False, _timing=timing)
        assert not timing['exact_sequence']
        monkeypatch.setenv('XRT_HIP_REFLECT_EXACT', '1')
        g2, a2, b2 = plate.double_refract(rs.Beam(copyFrom=beam))
        monkeypatch.delenv('XRT_HIP_REFLECT_EXACT')
    finally:
        roe.fuseConsumers = True
    alive = int((g0.state == 1).sum())
    assert 500 < alive < n - 500, alive
    for tag, (g, a, b) in (('one pass', (g1, a1, b1)), ('exact', (g2, a2, b2))):
        same(g, g0, 'global, ' + tag)
        same(a, a0, 'front face, ' + tag, extra=('theta',))
        same(b, b0, 'back face, ' + tag, extra=('theta',))
    # without the local beams (the Balder chain's filter): the global beam alone
    g3, a3, b3 = plate.double_refract(rs.Beam(copyFrom=beam))
    assert type(a3) is rs.LazyBeam and not a3.__dict__['_filled']
    same(g3, g0, 'global alone')
    same(b3, b0, 'back face on demand', extra=('theta',))
    same(a3, a0, 'front face on demand', extra=('theta',))


# ---- apertures and a screen in the tail of the DCM -------------------------------------------
def _dcm_scene(n, amplitudes, odd_ray=False):
    bl = raycing.BeamLine()
    dcm = workloads.cfg3_dcm(bl)
    beam = workloads.synthetic_rays(n, 11, sa=1e-4, E=(8995., 9005.), amplitudes=amplitudes)
    beam.state[::37] = -2
    beam.x[::29] *= 400.
    if odd_ray:          # its largest direction cosine is not the head ray's: the pass is redone
        beam.a[778] = 0.9
        beam.b[778] = np.sqrt(1 - beam.a[778]**2 - beam.c[778]**2)
    roe.fuseConsumers = False
    try:
        g0 = dcm.double_reflect(rs.Beam(copyFrom=beam))[0]
    finally:
        roe.fuseConsumers = True
    ok = g0.state == 1
    pos = np.array([g0.x[ok].mean(), g0.y[ok].mean(), g0.z[ok].mean()])
    way = np.array([g0.a[ok].mean(), g0.b[ok].mean(), g0.c[ok].mean()])
    slit = ra.RectangularAperture(bl, 'slit', list(pos + 1000. * way), ('left', 'top'), [0., 0.05])
    scr = rsc.Screen(bl, 'after', center=list(pos + 2000. * way))
    return dcm, beam, slit, scr


def _eager_dcm_chain(dcm, aps, scr, beam):
    old = roe.fuseConsumers
    roe.fuseConsumers = False
    try:
        g, a, b = dcm.double_reflect(rs.Beam(copyFrom=beam))
        locs = [s.propagate(g) for s in aps]
        img = scr.expose(g) if scr is not None else None
    finally:
        roe.fuseConsumers = old
    return g, a, b, locs, img


@pytest.mark.parametrize('amplitudes', [False, True])
@pytest.mark.parametrize('what', ['screen', 'slit + screen', 'slit'])
def test_apertures_and_a_screen_in_the_tail_of_the_dcm(amplitudes, what):
    """DCM.double_reflect is handed out before it is launched too (oes._DeferredDouble): a slit and
    a screen that take its global beam ride in the tail of the pair's kernel
    (reflect_fused_dcm_scr; reference: dcm.py:248-354 -> apertures.py:334-413 ->
    screens.py:226-302). Every beam has the bits of the separate launches."""
    dcm, beam, slit, scr = _dcm_scene(80000, amplitudes)
    aps = [slit] if 'slit' in what else []
    scr = scr if 'screen' in what else None
    g0, a0, b0, locs0, img0 = _eager_dcm_chain(dcm, aps, scr, beam)
    if aps:
        assert 1000 < (g0.state == slit.lostNum).sum() < 70000
    _forget(dcm, slit, *([scr] if scr else []))
    g, a, b = dcm.double_reflect(rs.Beam(copyFrom=beam))
    op = g.__dict__['_op']
    assert type(op) is roe._DeferredDouble and op.state == 'pending'
    locs = [s.propagate(g) for s in aps]
    assert op.state == 'pending' and len(op.apertures) == len(aps)
    if scr is not None:
        img = scr.expose(g)
        assert op.state == 'pending'
        same(img, img0, 'image')
        assert op.state == 'imaged' and not g.__dict__['_filled']
    same(g, g0, 'global')
    same(b, b0, 'second crystal', extra=('theta',))
    same(a, a0, 'first crystal', extra=('theta',))
    for s, l, l0 in zip(aps, locs, locs0):
        same(l, l0, 'beam at the slit')
    assert op.state == 'done' and op.beam is None


def test_a_contradicted_or_forced_dcm_pass_with_its_tail(monkeypatch):
    dcm, beam, slit, scr = _dcm_scene(60000, True, odd_ray=True)
    g0, a0, b0, locs0, img0 = _eager_dcm_chain(dcm, [slit], scr, beam)
    t = {}
    dcm.double_reflect(rs.Beam(copyFrom=beam), _timing=t)
    assert t['exact_sequence']
    for forced in (False, True):
        _forget(dcm, slit, scr)
        if forced:
            monkeypatch.setenv('XRT_HIP_REFLECT_EXACT', '1')
        g, a, b = dcm.double_reflect(rs.Beam(copyFrom=beam))
        loc = slit.propagate(g)
        img = scr.expose(g)
        same(img, img0, 'image %d' % forced)
        same(g, g0, 'global %d' % forced)
        same(loc, locs0[0], 'beam at the slit %d' % forced)
        same(a, a0, 'first crystal %d' % forced, extra=('theta',))
        same(b, b0, 'second crystal %d' % forced, extra=('theta',))


def test_c_abi_double_reflect_tail():
    import ctypes
    from xrt_amd import _lib, _structs, hipcalls
    dcm, beam, slit, scr = _dcm_scene(30000, False)
    g0, a0, b0, locs0, img0 = _eager_dcm_chain(dcm, [slit], scr, beam)
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    first, second = dcm._own_angles(False), dcm._own_angles(True)
    p1 = dcm._make_pass(*first[:4], fromVacuum=True, out_to_global=False)
    p2 = dcm._make_pass(*second, fromVacuum=True, is2ndXtal=True, in_is_global=False, good_mode=1,
                        out_to_global=True, zero_local_not_entering=True, force_lost_out=False)
    m1 = dcm._material_struct(dcm.material, True, dev)
    m2 = dcm._material_struct(dcm.material2, True, dev)
    ws = hipcalls.workspace(dev, lib.xrt_hip_reflect_workspace_bytes(beam.nrays), 'reflect')
    s_in = beam.to_struct(dev)
    srec, arec = scr._record(False), slit._record()
    gb, img = (rs.Beam.empty_like_on_device(beam, dev) for _ in range(2))
    tail = _structs.Tail()
    tail.n_apertures, tail.keep_screen = 1, 1
    tail.aperture[0] = arec
    tail.screen, tail.out_screen = ctypes.addressof(srec), ctypes.addressof(img.to_struct(dev))
    fused = ctypes.c_int(-1)
    args = (ctypes.byref(p1), ctypes.byref(m1), ctypes.byref(p2), ctypes.byref(m2),
            ctypes.byref(s_in), None, None, ctypes.byref(gb.to_struct(dev)), None, None)
    rest = (1, ctypes.c_void_p(ws.data_ptr()), ws.numel(),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(fused))
    _lib.check(lib.xrt_hip_double_reflect_tail_f64_dev(*args, ctypes.byref(tail), *rest), 'tail')
    assert fused.value == 9
    same(img, img0, 'image')
    same(gb, g0, 'global')
    tail.plot = 1
    assert lib.xrt_hip_double_reflect_tail_f64_dev(*args, ctypes.byref(tail), *rest) != 0
    assert b'plot' in lib.xrt_hip_last_error()


# ---- apertures and a screen in the tail of a SINGLE flat Bragg crystal -------------------------
def _xtal_scene(n, amplitudes, thin=False, alpha=None, odd_ray=False):
    bl = raycing.BeamLine()
    si = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15, **({'t': 0.05} if thin else {}))
    thB = float(np.ravel(si.get_Bragg_angle(9000.) - si.get_dtheta(9000.))[0])
    xt = roe.OE(bl, 'xtal', center=[0, 20000., 0], pitch=thB, material=si, alpha=alpha,
                limPhysX=[-10, 10], limPhysY=[-50, 50])
    beam = workloads.synthetic_rays(n, 3, sa=1e-4, E=(8995., 9005.), amplitudes=amplitudes)
    beam.state[::41] = -2
    beam.x[::31] *= 300.
    if odd_ray:          # its largest direction cosine is not the head ray's: the pass is redone
        beam.a[778] = 0.9
        beam.b[778] = np.sqrt(1 - beam.a[778]**2 - beam.c[778]**2)
    scr = rsc.Screen(bl, 'after', center=[0, 21000., 1000. * np.tan(2 * thB)])
    pipe = ra.RoundAperture(bl, 'pipe', [0, 20500., 500. * np.tan(2 * thB)], r=2.)
    return xt, beam, pipe, scr


@pytest.mark.parametrize('amplitudes', [False, True])
@pytest.mark.parametrize('thin', [False, True])
@pytest.mark.parametrize('what', ['screen', 'pipe + screen', 'pipe'])
def test_a_single_flat_crystal_carries_its_tail(amplitudes, thin, what):
    """OE.reflect of a flat Bragg crystal (thick: ThickXtal kernels, thin: the general flat-crystal
    ones) with the aperture and the screen that take its global beam in the tail of its pass
    (reflect_fused_xtal_scr; reference oes/reflect.py -> apertures.py:334-413 ->
    screens.py:226-302): every beam has the bits of the separate launches, and the global beam is
    not written when only the screen reads it."""
    xt, beam, pipe, scr = _xtal_scene(60000, amplitudes, thin=thin)
    aps = [pipe] if 'pipe' in what else []
    one = scr if 'screen' in what else None
    gb0, lb0, locs0, img0 = _eager_chain(xt, aps, one, beam)
    assert ((gb0.state == 1) | (gb0.state == pipe.lostNum)).sum() > 20000
    if aps:
        assert 1000 < (gb0.state == pipe.lostNum).sum() < 58000
    _forget(xt, scr, pipe)
    gb, lb = xt.reflect(rs.Beam(copyFrom=beam))
    op = gb.__dict__['_op']
    locs = [a.propagate(gb) for a in aps]
    assert op.state == 'pending' and len(op.apertures) == len(aps)
    if one is not None:
        img = one.expose(gb)
        assert op.state == 'pending'
        same(img, img0, 'image')
        assert op.state == 'imaged' and not gb.__dict__['_filled']     # (the kernel carried it)
    same(gb, gb0, 'global')
    same(lb, lb0, 'local', extra=('theta',))
    for l, l0 in zip(locs, locs0):
        same(l, l0, 'beam in the pipe')


def test_a_contradicted_mixed_or_forced_crystal_pass_with_its_tail(monkeypatch):
    """The redo behind a crystal pass with a tail (reflect_redo_scr): an assumption contradicted by
    one ray, a batch with both signs of beamInDotNormal (asymmetric cut hit from both sides of the
    cut angle: reflect.py:573-574), and the forced exact sequence -- marks and image are made from
    the real global beam; the bits of the separate launches."""
    for kind in ('odd ray', 'mixed', 'forced'):
        if kind == 'mixed':
            rng = np.random.default_rng(77)
            bl = raycing.BeamLine()
            si = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
            xt = roe.OE(bl, 'xtal', center=[0., 10000., 0.], pitch=np.radians(3.5), material=si,
                        alpha=np.radians(3.), limPhysX=[-20, 20], limPhysY=[-300, 300])
            n = 20000
            beam = rs.Beam(nrays=n, withAmplitudes=True)
            beam.x[:] = rng.normal(0, 1., n)
            beam.z[:] = rng.normal(0, 0.5, n)
            beam.a[:] = rng.normal(0, 1e-4, n)
            beam.c[:] = rng.uniform(-np.radians(2.5), np.radians(2.0), n)
            beam.b[:] = np.sqrt(1 - beam.a**2 - beam.c**2)
            beam.y[:] = 9900.
            beam.z[:] += -beam.c / beam.b * 100.
            beam.E[:] = rng.uniform(8990., 9010., n)
            beam.Jss[:], beam.Jpp[:], beam.Jsp[:] = 0.5, 0.5, 0.
            beam.Es[:], beam.Ep[:] = np.sqrt(0.5), np.sqrt(0.5)
            beam.state[:] = 1
            scr = rsc.Screen(bl, 'after', center=[0, 10400., 50.])
            pipe = ra.RectangularAperture(bl, 'slit', [0, 10200., 25.], ('bottom', 'top'),
                                          [-10., 20.])
            info = {}
            xt.reflect(rs.Beam(copyFrom=beam), _info=info)
            assert info['mixed_sign']
        else:
            xt, beam, pipe, scr = _xtal_scene(60000, True, odd_ray=kind == 'odd ray')
            if kind == 'odd ray':
                t = {}
                xt.reflect(rs.Beam(copyFrom=beam), _timing=t)
                assert t['exact_sequence']
            else:
                monkeypatch.setenv('XRT_HIP_REFLECT_EXACT', '1')
        gb0, lb0, locs0, img0 = _eager_chain(xt, [pipe], scr, beam)
        _forget(xt, scr, pipe)
        gb, lb = xt.reflect(rs.Beam(copyFrom=beam))
        loc = pipe.propagate(gb)
        img = scr.expose(gb)
        same(img, img0, 'image, ' + kind)
        same(gb, gb0, 'global, ' + kind)
        same(loc, locs0[0], 'beam at the aperture, ' + kind)
        same(lb, lb0, 'local, ' + kind, extra=('theta',))


# ---- a screen and the mask behind it on a RESIDENT beam: one pass over the rays -----------------
def _front_end(n, amplitudes):
    """A source's beam, a monitor 15 m downstream and four kinds of mask behind it."""
    bl = raycing.BeamLine()
    beam = workloads.synthetic_rays(n, 23, amplitudes=amplitudes)
    beam.state[::97] = -3
    beam.state[5::101] = 2
    beam.b[:] = np.sqrt(1 - beam.a**2 - beam.c**2)
    beam.b[3::211] *= -1.                      # (flies away: arrives with a negative path only)
    fsm = rsc.Screen(bl, 'FSM0', [0, 15000., 0])
    at = [0, 15750., 0]
    masks = [ra.RectangularAperture(bl, 'mask', at, ('left', 'right', 'bottom', 'top'),
                                    [-2., 2.5, -0.2, 0.25]),
             ra.RectangularBeamStop(bl, 'stop', at, ('left', 'right', 'bottom', 'top'),
                                    [-1., 1., -0.2, 0.2]),
             ra.RoundAperture(bl, 'pipe', at, r=1.5),
             ra.DoubleSlit(bl, 'two', at, ('bottom', 'top'), [-0.4, 0.4], shadeFraction=0.3),
             ra.PolygonalAperture(bl, 'tri', at, opening=[(-2., -0.3), (2.5, -0.2), (0., 0.4)])]
    return bl, beam, fsm, masks


def _two_launches(fsm, mask, beam, **kw):
    old = roe.fuseConsumers
    roe.fuseConsumers = False
    try:
        b = rs.Beam(copyFrom=beam)
        img = fsm.expose(b, **kw)
        loc = mask.propagate(b)
    finally:
        roe.fuseConsumers = old
    return b, img, loc


@pytest.mark.parametrize('amplitudes', [False, True])
@pytest.mark.parametrize('kind', [0, 1, 2, 3])
def test_a_screen_and_the_mask_behind_it_are_one_pass(amplitudes, kind):
    """Screen.expose(beam) then aperture.propagate(beam) of a resident beam (reference
    screens.py:226-302, apertures.py:334-413: the front-end monitor and mask of every beamline)
    = one launch (xrt_hip_screen_expose_mark_f64_dev): the image from the states as they were,
    the marks in the beam, the beam in the mask's frame on demand -- the bits of the two calls."""
    bl, beam, fsm, masks = _front_end(70001, amplitudes)
    mask = masks[kind]
    b0, img0, loc0 = _two_launches(fsm, mask, beam, onlyPositivePath=bool(kind % 2))
    assert 3000 < (b0.state == mask.lostNum).sum() < 60000
    assert ((img0.state == fsm.lostNum).sum() > 100) == bool(kind % 2)
    b = rs.Beam(copyFrom=beam)
    img = fsm.expose(b, onlyPositivePath=bool(kind % 2))
    shot = img.__dict__['_op']
    assert type(shot) is rsc._DeferredExpose and shot.state == 'pending'
    loc = mask.propagate(b)
    assert shot.state == 'done' and img.__dict__['_filled'] and shot.was is None
    assert not loc.__dict__['_filled'] and loc.__dict__['_op'].shot is None
    assert np.array_equal(b.state, b0.state)
    same(img, img0, 'image')
    same(loc, loc0, 'beam at the mask')
    same(b, b0, 'the beam itself')


def test_a_screen_on_a_resident_beam_in_any_other_order():
    bl, beam, fsm, masks = _front_end(40000, False)
    mask = masks[0]
    b0, img0, loc0 = _two_launches(fsm, mask, beam)
    # the image looked at first: the screen's own launch, then the mask's
    b = rs.Beam(copyFrom=beam)
    img = fsm.expose(b)
    same(img, img0, 'image first')
    loc = mask.propagate(b)
    same(loc, loc0, 'mask after'), same(b, b0, 'beam')
    # nobody keeps the image: written all the same (with the mask's marks in the same launch)
    b = rs.Beam(copyFrom=beam)
    fsm.expose(b)
    assert len([op for op in rs._PENDING if type(op) is rsc._DeferredExpose]) == 1
    mask.propagate(b)
    assert not [op for op in rs._PENDING if type(op) is rsc._DeferredExpose]
    assert np.array_equal(b.state, b0.state)
    # an outline of vertices, two screens, a beam changed on the host in between: own launches
    for case in ('polygon', 'two screens', 'host change', 'element'):
        b = rs.Beam(copyFrom=beam)
        img = fsm.expose(b)
        shot = img.__dict__['_op']
        if case == 'polygon':
            m = masks[4]
        else:
            m = mask
        ref_b, ref_img, ref_loc = _two_launches(fsm, m, beam)
        if case == 'two screens':
            img2 = rsc.Screen(bl, 'FSM1', [0, 15500., 0]).expose(b)
        if case == 'host change':
            b.state[::7] = -5          # (after the screen saw the beam)
            ref_b = rs.Beam(copyFrom=beam)
            roe.fuseConsumers = False
            try:
                ref_img = fsm.expose(ref_b)
                ref_b.state[::7] = -5
                ref_loc = m.propagate(ref_b)
            finally:
                roe.fuseConsumers = True
        if case == 'element':        # every element's call launches what waits
            bl2, oe, scr, _ = scene(n=10)
            oe.reflect(workloads.synthetic_rays(1000, 1))[0].nrays
            assert shot.state == 'done'
        loc = m.propagate(b)
        assert shot.state == 'done'
        if case == 'two screens':
            assert img2.__dict__['_op'].state == 'done'
        same(img, ref_img, case + ': image')
        same(loc, ref_loc, case + ': beam at the mask')
        assert np.array_equal(b.state, ref_b.state), case


def test_c_abi_screen_and_mask_refusals():
    import ctypes
    from xrt_amd import _lib
    bl, beam, fsm, masks = _front_end(5000, False)
    dev = torch.device('cuda', 0)
    b = rs.Beam(copyFrom=beam)
    out = rs.Beam.empty_like_on_device(b, dev)
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = lambda s, a: (ctypes.byref(s), ctypes.byref(a), ctypes.byref(b.to_struct(dev)),  # noqa: E731
                         ctypes.byref(out.to_struct(dev)), stream)
    rec = fsm._record(False)
    rec.radius = 100.
    assert lib.xrt_hip_screen_expose_mark_f64_dev(*args(rec, masks[0]._record())) != 0
    assert b'hemispheric' in lib.xrt_hip_last_error()
    assert lib.xrt_hip_screen_expose_mark_f64_dev(*args(fsm._record(False), masks[4]._record())) != 0
    assert b'vertices' in lib.xrt_hip_last_error()
    _lib.check(lib.xrt_hip_screen_expose_mark_f64_dev(*args(fsm._record(False), masks[2]._record())),
               'screen + mask')
    b._h.pop('state', None)
    b0, img0, _ = _two_launches(fsm, masks[2], beam)
    assert np.array_equal(b.state, b0.state)
    same(out, img0, 'image through the C ABI')


@pytest.mark.parametrize('n', [0, 1, 63, 257, 20011])
@pytest.mark.parametrize('kind', [0, 2, 3])
def test_a_screen_and_the_mask_behind_it_against_the_oracle(n, kind):
    """The one-launch form directly against the numpy restatement of screens.py:226-302 and
    apertures.py:334-413 (oracle/elements_np.py, pinned by goldens G1 / G7), also for empty and
    ragged beams: states bit for bit, geometry 1e-13, amplitudes 1e-12."""
    from oracle import elements_np as en
    from oracle.adapters import to_oracle_beam
    bl, beam, fsm, masks = _front_end(n, True)
    mask = masks[kind]
    ob = to_oracle_beam(beam)
    oimg = en.screen_expose(ob, (fsm.x, fsm.y, fsm.z), fsm.center, fsm.lostNum)
    blades = dict(zip(mask.kind, mask.opening)) if kind != 2 else {}
    oloc = en.aperture_propagate(ob, (mask.x, mask.y, mask.z), mask.center, blades, mask.lostNum,
                                 radius=mask.r if kind == 2 else None,
                                 shadeFraction=0.3 if kind == 3 else None)
    b = rs.Beam(copyFrom=beam)
    img = fsm.expose(b)
    loc = mask.propagate(b)
    assert img.__dict__['_op'].state == 'done' or n == 0
    assert np.array_equal(b.state, ob.state) and len(b.state) == n
    for got, ref, what in ((img, oimg, 'image'), (loc, oloc, 'beam at the mask')):
        assert np.array_equal(got.state, ref.state), what
        for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp'):
            r = getattr(ref, f)
            assert np.abs(getattr(got, f) - r).max(initial=0.) <= \
                1e-13 * max(np.abs(r).max(initial=0.), 1e-300), (what, f)
        for f in ('Es', 'Ep'):
            r = getattr(ref, f)
            assert np.abs(getattr(got, f) - r).max(initial=0.) <= 1e-12, (what, f)
