"""CPU, build container only (skipped where /root/reference is absent): the
XRT_HIP drop-in is installed as ``waves.waveCL`` of the REAL reference and the
reference's own ``diffract`` is run through it. The C call is replaced by the
numpy oracle here (no GPU in this environment), so the test pins everything on
the Python side of the boundary: the marshalling contract of
_diffraction_integral_CL (waves.py:854-896), in-place + returned outputs, and
that the OpenCL sign convention gives the same intensities / directions as the
reference's numpy path."""
import numpy as np
import pytest

from oracle import _refenv, kirchhoff_np as kn
from oracle.consts import CHBAR

pytestmark = pytest.mark.skipif(not _refenv.available(),
                                reason='reference tree not present')


def _make_fake(convention):
    from xrt_amd.backends.raycing.myhip import XRT_HIP

    class OracleBackedHIP(XRT_HIP):
        def set_cl(self, targetOpenCL='auto', precisionOpenCL='float64'):
            self.device_ids = [0]
            self.lastTargetOpenCL = targetOpenCL
            self.lastPrecisionOpenCL = precisionOpenCL

        def _call_lib(self, npix, px, py, pz, ns, nl, Es, Ep, k, pos, nrm, conv,
                      outs):
            assert pos.flags.f_contiguous and pos.shape == (4, ns)
            E = k * CHBAR / 1e7
            raw = kn.kirchhoff_conv(px, py, pz, pos[0], pos[1], pos[2],
                                    [nrm[0], nrm[1], nrm[2]], nl, E, Es, Ep)
            if conv == 1:
                raw = kn.to_cl_convention(*raw)
            for o, r in zip(outs, raw):
                o[:] = r
    return OracleBackedHIP(convention=convention)


def _scene(rw, seed):
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.screens as rsc
    import xrt.backends.raycing.apertures as ra
    np.random.seed(seed)
    bl = raycing.BeamLine()
    src = rs.GeometricSource(bl, 'src', nrays=500, dx=0.01, dz=0.01, dxprime=1e-6,
                             dzprime=1e-6, distE='lines', energies=(7900.,),
                             polarization='h')
    slit = ra.RectangularAperture(bl, 'slit', [0, 44000., 0],
                                  ('left', 'right', 'bottom', 'top'),
                                  [-0.1, 0.1, -0.1, 0.1])
    scr = rsc.Screen(bl, 'scr', [0, 54000., 0])
    mesh = np.linspace(-0.5, 0.5, 12)
    wscr = scr.prepare_wave(slit, mesh, mesh)
    wslit = slit.prepare_wave(src, 500)
    # a deterministic field on the slit (what source.shine(wave=...) would fill)
    k = 7900. / CHBAR * 1e7
    rho2 = wslit.x**2 + wslit.z**2
    wslit.Es[:] = np.exp(1j * k * rho2 / (2 * 44000.))
    wslit.Ep[:] = 0.3 * wslit.Es
    wslit.Jss[:] = np.abs(wslit.Es)**2
    wslit.Jpp[:] = np.abs(wslit.Ep)**2
    wslit.E[:] = 7900.
    wslit.a[:] = wslit.x / 44000.
    wslit.c[:] = wslit.z / 44000.
    wslit.b[:] = np.sqrt(1 - wslit.a**2 - wslit.c**2)
    return wslit, wscr


@pytest.mark.parametrize('convention', ['opencl', 'numpy'])
def test_reference_diffract_runs_through_the_dropin(convention):
    _refenv.activate()
    import xrt.backends.raycing.waves as rw
    saved = rw.waveCL
    try:
        rw.waveCL = None
        wslit, wscr = _scene(rw, 3)
        rw.diffract(wslit, wscr)                     # the reference's numpy path
        ref = {f: np.array(getattr(wscr, f)) for f in
               ('Es', 'Ep', 'Jss', 'Jpp', 'Jsp', 'a', 'b', 'c')}
        rw.waveCL = _make_fake(convention)
        wslit, wscr = _scene(rw, 3)
        rw.diffract(wslit, wscr)                     # the reference + XRT_HIP
        for f in ('Jss', 'Jpp', 'Jsp', 'a', 'b', 'c'):
            r = ref[f]
            assert np.abs(getattr(wscr, f) - r).max() <= 1e-10 * np.abs(r).max(), f
        sign = -1. if convention == 'opencl' else 1.
        for f in ('Es', 'Ep'):
            r = ref[f]
            assert np.abs(getattr(wscr, f) - sign * r).max() <= 1e-10 * np.abs(r).max(), f
    finally:
        rw.waveCL = saved


def _make_fake_undulator_backend():
    from xrt_amd.backends.raycing.myhip import XRT_HIP
    from oracle import undulator_np as un

    class OracleBackedHIP(XRT_HIP):
        calls = 0

        def set_cl(self, targetOpenCL='auto', precisionOpenCL='float64'):
            self.device_ids = [0]
            self.lastTargetOpenCL = targetOpenCL
            self.lastPrecisionOpenCL = precisionOpenCL

        def _call_lib_undulator(self, u, n, rays, outs):
            import ctypes
            type(self).calls += 1

            def tab(p):
                return np.ctypeslib.as_array(
                    ctypes.cast(p, ctypes.POINTER(ctypes.c_double)), (u.jend,))
            t = dict(tg=tab(u.tg), ag=tab(u.ag), sintg=tab(u.sintg), costg=tab(u.costg),
                     sintgph=tab(u.sintgph), costgph=tab(u.costgph))
            gamma, wu, w, ww1, th, ps = rays
            taper = u.alpha_s * un.E2WC if u.mode == 1 else None
            Is, Ip = un.sp_sum(u.mode, u.Kx, u.Ky, u.nper, t, ww1, w, wu, gamma, th, ps,
                               taper, u.r0z)
            outs[0][:] = Is
            outs[1][:] = Ip
    return OracleBackedHIP()


@pytest.mark.parametrize('kw', [
    dict(n=30, K=0.9),
    dict(n=12, K=1.1, taper=(0.4, 10.)),
    dict(n=12, K=1.1, R0=25000.),
], ids=['far', 'taper', 'nf'])
def test_reference_undulator_runs_through_the_dropin(kw):
    """The reference's Undulator.build_I_map takes its OpenCL branch
    (_build_I_map_CL, synchr.py:2110-2176) with XRT_HIP attached, and the result
    equals its numpy branch: pins the run_parallel marshalling of the three
    undulator kernels (scalar order, alphaS = taper/E2WC, nper)."""
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    raycing._VERBOSITY_ = 0
    bl = raycing.BeamLine()
    u = rs.Undulator(bl, 'u', nrays=500, eE=3.0, eI=0.5, eEspread=0, eEpsilonX=0.263,
                     eEpsilonZ=0.008, betaX=9., betaZ=2., period=18.5, eMin=2500,
                     eMax=3200, xPrimeMax=0.03, zPrimeMax=0.03, targetOpenCL=None,
                     distE='BW', gNodes=12, gIntervals=2, **kw)
    if u.needReset:
        u.reset()
    rng = np.random.RandomState(5)
    w = rng.uniform(2500, 3200, 64)
    th = rng.uniform(-3e-5, 3e-5, 64)
    ps = rng.uniform(-3e-5, 3e-5, 64)
    ref = u.build_I_map(w, th, ps)                    # numpy branch (cl_ctx is None)
    fake = _make_fake_undulator_backend()
    fake.attach_to_source(u)
    got = u.build_I_map(w, th, ps)                    # OpenCL branch -> XRT_HIP
    assert type(fake).calls == 1
    for a, b in zip(got, ref):
        assert np.linalg.norm(a - b) <= 1e-12 * np.linalg.norm(b)


def test_reference_source_from_field_runs_through_the_dropin():
    """SourceFromField.build_I_map takes its OpenCL branch
    (_build_I_map_custom_field_CL, synchr.py:1157-1272) with XRT_HIP attached:
    pins the 'custom_field' marshalling (scalar order, 10 node tables, emcg
    rebuilt from gamma)."""
    _refenv.activate()
    import ctypes
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    from xrt_amd.backends.raycing.myhip import XRT_HIP
    from oracle import undulator_np as un
    raycing._VERBOSITY_ = 0

    class OracleBackedHIP(XRT_HIP):
        calls = 0
        trajectories = 0

        def set_cl(self, targetOpenCL='auto', precisionOpenCL='float64'):
            self.device_ids = [0]
            self.lastTargetOpenCL = targetOpenCL
            self.lastPrecisionOpenCL = precisionOpenCL

        def _trajectory(self, kernelName, scalarArgs, nonSlicedRO, nonSlicedRW):
            # 'get_trajectory' as _build_trajectory_CL marshals it (synchr.py:1011-1035)
            type(self).trajectories += 1
            assert kernelName == 'get_trajectory' and len(nonSlicedRW) == 6
            grid, Bx, By, Bz = (np.array(a) for a in nonSlicedRO)
            assert int(scalarArgs[0]) == len(grid) and len(Bx) == 2 * len(grid) - 1
            bx, by, bm, tx, ty, tz = un.trajectory(grid, Bx, By, Bz)
            return bx, by, np.full(len(grid), bm), tx, ty, tz

        def _call_lib_custom_field(self, f, n, rays, outs):
            type(self).calls += 1

            def tab(p):
                return np.ctypeslib.as_array(
                    ctypes.cast(p, ctypes.POINTER(ctypes.c_double)), (f.jend,))
            t = {k: tab(getattr(f, k)) for k in ('tg', 'ag', 'Bx', 'By', 'Bz', 'betax',
                                                 'betay', 'trajx', 'trajy', 'trajz')}
            emcg, gamma, w, th, ps = rays
            assert not f.filament and f.wc == 0.
            Is, Ip = un.custom_sp_sum(False, t, emcg, w, gamma, th, ps, f.betam,
                                      f.R0 if f.near_field else None)
            outs[0][:] = Is
            outs[1][:] = Ip

    L0, Np = 30., 6
    z = np.linspace(-L0*Np/2-40, L0*Np/2+40, 1500)
    env = 0.5*(np.tanh((z + L0*Np/2)/8.) - np.tanh((z - L0*Np/2)/8.))
    field = np.vstack((z, 0.6*np.sin(2*np.pi*z/L0)*env)).T
    bl = raycing.BeamLine()
    s = rs.SourceFromField(
        bl, 'sff', nrays=500, eE=3.0, eI=0.5, eEspread=0, eEpsilonX=0.263,
        eEpsilonZ=0.008, betaX=9., betaZ=2., eMin=1500, eMax=1700, xPrimeMax=0.1,
        zPrimeMax=0.1, targetOpenCL=None, distE='BW', customField=field, gNodes=20,
        gIntervals=12, R0=15000.)
    if s.needReset:
        s.reset()
    rng = np.random.RandomState(5)
    w = rng.uniform(1500, 1700, 64)
    th = rng.uniform(-1e-4, 1e-4, 64)
    ps = rng.uniform(-1e-4, 1e-4, 64)
    ref = s.build_I_map(w, th, ps)
    fake = OracleBackedHIP()
    fake.attach_to_source(s)
    got = s.build_I_map(w, th, ps)
    assert type(fake).calls == 1 and type(fake).trajectories == 1
    for a, b in zip(got, ref):
        assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(b)


def test_host_only_classes_agree_with_the_reference():
    """Classes that are pure host arithmetic, run side by side with the reference's:
    SetOfRectangularAperturesOnZActuator (blades, actuator position and limits after
    select_aperture), CollimatedMeshSource (the rays, bit for bit)."""
    _refenv.activate()
    import xrt.backends.raycing as rr
    import xrt.backends.raycing.apertures as rar
    import xrt.backends.raycing.sources as rsr
    import xrt_amd.backends.raycing as mr
    import xrt_amd.backends.raycing.apertures as mar
    import xrt_amd.backends.raycing.sources as msr
    args = dict(apertures=['big', 'small', 'top-edge'], centerZs=[5., -4., 12.],
                dXs=[2., 0.5], dZs=[1., 0.2])
    for name, target in (('big', 1401.), ('small', 1399.5), ('top-edge', 1400.)):
        seen = []
        for R, A in ((rr, rar), (mr, mar)):
            bl = R.BeamLine(height=1400.)
            s = A.SetOfRectangularAperturesOnZActuator(bl, 'set', [0, 1000., 1400.], **args)
            s.select_aperture(name, target)
            seen.append((dict(s.blades), s.zActuator, s.zlims, s.limOptX, s.limOptY,
                         s.curAperture, s.lostNum))
        assert seen[0] == seen[1], name
    kw = dict(center=(1, 2, 3), dx=2., dz=1., nx=5, nz=4, totalFlux=1e10, polarization='v')
    b0 = rsr.CollimatedMeshSource(rr.BeamLine(azimuth=0.1), 'c', **kw).shine()
    b1 = msr.CollimatedMeshSource(mr.BeamLine(azimuth=0.1), 'c', **kw).shine()
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'E', 'Jss', 'Jpp', 'Jsp', 'state'):
        assert np.array_equal(getattr(b0, f), getattr(b1, f)), f
    assert b0.sourceWeight == b1.sourceWeight and len(b1.x) == 21


def test_predefined_materials_agree_with_the_reference():
    """Every class of the three catalogues against the reference's class of the same name:
    formula, density, molar mass; d spacing, cell volume and chi/F factor of the crystals at
    the default and at another reflection."""
    _refenv.activate()
    import xrt.backends.raycing.materials.elemental as rel
    import xrt.backends.raycing.materials.compounds as rco
    import xrt.backends.raycing.materials.crystals as rcr
    import xrt_amd.backends.raycing.materials.elemental as xel
    import xrt_amd.backends.raycing.materials.compounds as xco
    import xrt_amd.backends.raycing.materials.crystals as xcr
    for mine, ref in ((xel, rel), (xco, rco)):
        assert set(mine.__all__) == set(ref.__all__)
        for n in ref.__all__:
            a, b = getattr(mine, n)(), getattr(ref, n)()
            assert a.rho == b.rho and a.mass == b.mass, n
            assert [e.name for e in a.elements] == [e.name for e in b.elements], n
    assert set(xcr.__all__) == set(rcr.__all__)
    for n in rcr.__all__:
        for kw in ({}, dict(hkl=(2, 2, 0))):
            a, b = getattr(xcr, n)(**kw), getattr(rcr, n)(**kw)
            assert a.d == b.d and a.V == b.V and a.chiToF == b.chiToF, n
            assert abs(a.rho - b.rho) <= 1e-14 * abs(b.rho), n
