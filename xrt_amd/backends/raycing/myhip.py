"""XRT_HIP — the drop-in for xrt's ``XRT_CL`` accelerator object
(xrt/backends/raycing/myopencl.py:86-97, 414-583) on MI355X.

``xrt.backends.raycing.waves`` only needs an object with ``run_parallel``,
``cl_precisionF``, ``cl_precisionC``, ``set_cl`` and a non-None
``lastTargetOpenCL`` (waves.py:493-498, 698-705, 855-894); assign an instance
to ``waves.waveCL`` and ``diffract`` runs its Kirchhoff integral on the GPU:

    import xrt.backends.raycing.waves as rw
    from xrt_amd.backends.raycing.myhip import XRT_HIP
    rw.waveCL = XRT_HIP()

Like the OpenCL kernel it replaces (cl/diffract.cl:143-148) the returned
integrals use the -i/4pi sign and the (1+i) direction factor by default
(``convention='opencl'``); ``convention='numpy'`` returns what
``_diffraction_integral_conv`` returns. Both give identical post-`diffract`
intensities and directions.
"""
import ctypes

import numpy as np

from ... import _lib


class XRT_HIP(object):
    kernels = ('integrate_kirchhoff', 'undulator', 'undulator_taper',
               'undulator_nf', 'custom_field', 'custom_field_filament',
               'get_trajectory', 'get_trajectory_filament')

    def __init__(self, filename=None, targetOpenCL='auto',
                 precisionOpenCL='float64', convention='opencl', devices=None):
        self.cl_filename = filename
        self.cl_precisionF = np.float64
        self.cl_precisionC = np.complex128
        self.cl_is_blocking = True
        self.convention = convention
        self.lastTargetOpenCL = None
        self.lastPrecisionOpenCL = None
        self.lastKernelMs = None
        self.devices = devices
        self.set_cl(targetOpenCL, precisionOpenCL)

    def set_cl(self, targetOpenCL='auto', precisionOpenCL='float64'):
        """targetOpenCL: 'auto' / 'GPU' / 'all' = every visible GPU; an int or a
        sequence of ints = those device ordinals. fp32 is not offered: the
        Kirchhoff phase k*r ~ 4e11 rad needs fp64."""
        if precisionOpenCL not in ('auto', 'float64', np.float64):
            raise ValueError('XRT_HIP computes in float64 only')
        n = _lib.require_gpu()
        if self.devices is not None:
            devs = list(self.devices)
        elif isinstance(targetOpenCL, (int, np.integer)):
            devs = [int(targetOpenCL)]
        elif isinstance(targetOpenCL, (list, tuple)) and len(targetOpenCL) and \
                all(isinstance(d, (int, np.integer)) for d in targetOpenCL):
            devs = [int(d) for d in targetOpenCL]
        elif targetOpenCL is None:
            raise ValueError('targetOpenCL=None disables the accelerator; '
                             'do not install XRT_HIP then')
        else:
            devs = list(range(n))
        for d in devs:
            if not 0 <= d < n:
                raise ValueError('GPU ordinal %d out of range (%d visible)' % (d, n))
        self.device_ids = devs
        self.lastTargetOpenCL = targetOpenCL
        self.lastPrecisionOpenCL = precisionOpenCL

    def run_parallel(self, kernelName='', scalarArgs=None, slicedROArgs=None,
                     nonSlicedROArgs=None, slicedRWArgs=None,
                     nonSlicedRWArgs=None, dimension=0, complexity=0,
                     signal=None):
        if kernelName in ('undulator', 'undulator_taper', 'undulator_nf'):
            return self._undulator(kernelName, scalarArgs, slicedROArgs,
                                   nonSlicedROArgs, slicedRWArgs,
                                   int(dimension))
        if kernelName in ('custom_field', 'custom_field_filament'):
            return self._custom_field(kernelName, scalarArgs, slicedROArgs,
                                      nonSlicedROArgs, slicedRWArgs, int(dimension))
        if kernelName in ('get_trajectory', 'get_trajectory_filament'):
            return self._trajectory(kernelName, scalarArgs, nonSlicedROArgs, nonSlicedRWArgs)
        if kernelName != 'integrate_kirchhoff':
            raise NotImplementedError(
                "XRT_HIP implements %s, not %r" % (self.kernels, kernelName))
        return self._integrate_kirchhoff(scalarArgs, slicedROArgs,
                                         nonSlicedROArgs, slicedRWArgs,
                                         int(dimension))

    def _integrate_kirchhoff(self, scalarArgs, slicedRO, nonSlicedRO, slicedRW,
                             dimension):
        ns = int(scalarArgs[0])

        def f64(a, n, name):
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.size != n:
                raise ValueError('%s: %d elements, expected %d' % (name, a.size, n))
            return a

        px, py, pz = (f64(a, dimension, 'mesh') for a in slicedRO)
        nl = f64(nonSlicedRO[0], ns, 'nl')
        Es = np.ascontiguousarray(nonSlicedRO[1], dtype=np.complex128)
        Ep = np.ascontiguousarray(nonSlicedRO[2], dtype=np.complex128)
        k = f64(nonSlicedRO[3], ns, 'k')
        # (4, ns) order='F'  ==  ns x [x, y, z, 0] in memory (waves.py:872-879)
        pos = np.asfortranarray(nonSlicedRO[4], dtype=np.float64)
        nrm = np.asfortranarray(nonSlicedRO[5], dtype=np.float64)
        if pos.shape != (4, ns) or nrm.shape != (4, ns):
            raise ValueError('coordinate / normal arrays must be (4, %d)' % ns)
        outs = []
        for a in slicedRW:
            if not (isinstance(a, np.ndarray) and a.dtype == np.complex128 and
                    a.flags.c_contiguous and a.size == dimension):
                raise ValueError('RW arrays must be contiguous complex128[%d]'
                                 % dimension)
            outs.append(a)
        self._call_lib(dimension, px, py, pz, ns, nl, Es, Ep, k, pos, nrm,
                       0 if self.convention == 'numpy' else 1, outs)
        return tuple(outs)

    # E2WC of sources/synchr.py (module constant): _taperVal / E2WC = alphaS
    E2WC = 5067.7309392068091

    def attach_to_source(self, source):
        """Makes a reference ``Undulator`` built with ``targetOpenCL=None`` use
        this object for its field sums: sets what ``IntegratedSource._set_cl``
        (sources/sybase.py:1087-1104) would have set for an XRT_CL."""
        source.ucl = self
        source.cl_precisionF = self.cl_precisionF
        source.cl_precisionC = self.cl_precisionC
        source.cl_ctx = self            # only tested against None
        source.cl_is_blocking = True
        return source

    def _trajectory(self, kernelName, scalarArgs, nonSlicedRO, nonSlicedRW):
        """Argument order of SourceFromField._build_trajectory_CL (synchr.py:1011-1035):
        scalars [jend(, gamma)], RO [wtGrid, Bx, By, Bz], RW [betax, betay, betazav,
        trajx, trajy, trajz] on the grid; the caller reads betazav[-1]."""
        import torch
        from ... import hipcalls
        filament = kernelName.endswith('filament')
        jend = int(scalarArgs[0])
        grid, Bx, By, Bz = (np.ascontiguousarray(a, dtype=np.float64) for a in nonSlicedRO)
        if grid.size != jend or any(b.size != 2 * jend - 1 for b in (Bx, By, Bz)):
            raise ValueError('%s: grid of %d points needs the field on %d points'
                             % (kernelName, jend, 2 * jend - 1))
        dev = torch.device('cuda', self.device_ids[0])
        up = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
        if filament:
            gamma = float(scalarArgs[1])
            res = hipcalls.trajectory(up(grid), up(Bx), up(By), up(Bz), gamma=gamma,
                                      emcg=self.EMC_NUM / gamma)
        else:
            res = hipcalls.trajectory(up(grid), up(Bx), up(By), up(Bz))
        betax, betay, trajx, trajy, trajz = (t.cpu().numpy() for t in res[:5])
        mean = np.full(jend, float(res[5][0]))
        outs = (betax, betay, mean, trajx, trajy, trajz)
        for target, value in zip(nonSlicedRW or (), outs):
            if isinstance(target, np.ndarray) and target.shape == value.shape:
                target[...] = value
        return outs

    def _undulator(self, kernelName, scalarArgs, slicedRO, nonSlicedRO, slicedRW,
                   dimension):
        """Argument order of Undulator._build_I_map_CL (synchr.py:2132-2160)."""
        from ..._structs import Undulator
        mode = {'undulator': 0, 'undulator_taper': 1, 'undulator_nf': 2}[kernelName]
        if len(scalarArgs) != (4 if mode == 0 else 5):
            raise ValueError('%s takes %d scalar arguments'
                             % (kernelName, 4 if mode == 0 else 5))
        if len(slicedRO) != 6 or len(nonSlicedRO) != 6 or len(slicedRW) != 2:
            raise ValueError('%s: 6 sliced RO, 6 non-sliced RO and 2 RW arrays '
                             'expected' % kernelName)
        jend = int(scalarArgs[3])
        rays = [np.ascontiguousarray(a, dtype=np.float64) for a in slicedRO]
        tabs = [np.ascontiguousarray(a, dtype=np.float64) for a in nonSlicedRO]
        for a in rays:
            if a.size != dimension:
                raise ValueError('ray arrays must have %d elements' % dimension)
        for a in tabs:
            if a.size != jend:
                raise ValueError('node tables must have jend=%d elements' % jend)
        for a in slicedRW:
            if not (isinstance(a, np.ndarray) and a.dtype == np.complex128 and
                    a.flags.c_contiguous and a.size == dimension):
                raise ValueError('RW arrays must be contiguous complex128[%d]'
                                 % dimension)
        u = Undulator()
        u.mode = mode
        u.nper = int(scalarArgs[4]) if mode else 1
        u.Kx, u.Ky = float(scalarArgs[1]), float(scalarArgs[2])
        u.alpha_s = float(scalarArgs[0]) / self.E2WC if mode == 1 else 0.
        u.r0z = float(scalarArgs[0]) if mode == 2 else 0.
        u.jend = jend
        for name, t in zip(('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph'),
                           tabs):
            setattr(u, name, t.ctypes.data)
        self._call_lib_undulator(u, dimension, rays, slicedRW)
        return tuple(slicedRW)

    # SIE0 / SIM0 / C / 10 of synchr.py:1309 (emcg = that / gamma)
    EMC_NUM = 1.602176565e-19 / 9.109383701528e-31 / 2.99792458e10 / 10.

    def _custom_field(self, kernelName, scalarArgs, slicedRO, nonSlicedRO, slicedRW,
                      dimension):
        """Argument order of SourceFromField._build_I_map_custom_field_CL
        (synchr.py:1196-1256)."""
        from ..._structs import CustomField
        fil = kernelName.endswith('filament')
        if len(nonSlicedRO) != 10 or len(slicedRW) != 2:
            raise ValueError('%s: 10 node tables and 2 RW arrays expected' % kernelName)
        jend = int(scalarArgs[0])
        tabs = [np.ascontiguousarray(a, dtype=np.float64) for a in nonSlicedRO]
        for a in tabs:
            if a.size != jend:
                raise ValueError('node tables must have jend=%d elements' % jend)
        f = CustomField()
        f.filament = 1 if fil else 0
        f.jend = jend
        ones = np.ones(dimension)
        if fil:
            # scalars: jend, emcg0, 1/gamma0^2, R0, wc (= w0 E2WC / betam)
            theta, psi = (np.ascontiguousarray(a, dtype=np.float64) for a in slicedRO)
            emcg = float(scalarArgs[1]) * ones
            gamma = ones / np.sqrt(float(scalarArgs[2]))
            R0 = float(scalarArgs[3])
            f.wc = float(scalarArgs[4])
            f.betam = 1.
            w = ones            # unused: the carrier is given
        else:
            # scalars: jend, betam, R0
            gamma, w, theta, psi = (np.ascontiguousarray(a, dtype=np.float64)
                                    for a in slicedRO)
            emcg = self.EMC_NUM / gamma
            f.betam = float(scalarArgs[1])
            R0 = scalarArgs[2]
            R0 = 0. if R0 is None else float(R0)
            f.wc = 0.
        f.near_field = 1 if R0 > 0 else 0
        f.R0 = R0
        for name, t in zip(('tg', 'ag', 'Bx', 'By', 'Bz', 'betax', 'betay', 'trajx',
                            'trajy', 'trajz'), tabs):
            setattr(f, name, t.ctypes.data)
        for a in slicedRW:
            if not (isinstance(a, np.ndarray) and a.dtype == np.complex128 and
                    a.flags.c_contiguous and a.size == dimension):
                raise ValueError('RW arrays must be contiguous complex128[%d]'
                                 % dimension)
        self._call_lib_custom_field(f, dimension, [emcg, gamma, w, theta, psi],
                                    slicedRW)
        return tuple(slicedRW)

    def _call_lib_custom_field(self, f, n, rays, outs):
        lib = _lib.load()
        ms = ctypes.c_float(0.)
        ptr = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
        rays = [np.ascontiguousarray(a, dtype=np.float64) for a in rays]
        rc = lib.xrt_hip_custom_field_f64(
            self.device_ids[0], ctypes.byref(f), n, *[ptr(a) for a in rays],
            ptr(outs[0]), ptr(outs[1]), ctypes.byref(ms))
        _lib.check(rc, 'xrt_hip_custom_field_f64')
        self.lastKernelMs = ms.value

    def _call_lib_undulator(self, u, n, rays, outs):
        lib = _lib.load()
        ms = ctypes.c_float(0.)
        ptr = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
        rc = lib.xrt_hip_undulator_f64(
            self.device_ids[0], ctypes.byref(u), n, *[ptr(a) for a in rays],
            ptr(outs[0]), ptr(outs[1]), ctypes.byref(ms))
        _lib.check(rc, 'xrt_hip_undulator_f64')
        self.lastKernelMs = ms.value

    def _call_lib(self, npix, px, py, pz, ns, nl, Es, Ep, k, pos, nrm, convention,
                  outs):
        """The one place that crosses the C ABI (xrt_hip_kirchhoff_f64)."""
        lib = _lib.load()
        devs = (ctypes.c_int * len(self.device_ids))(*self.device_ids)
        ms = ctypes.c_float(0.)
        ptr = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
        rc = lib.xrt_hip_kirchhoff_f64(
            len(self.device_ids), devs, npix, ptr(px), ptr(py), ptr(pz), ns,
            ptr(nl), ptr(Es), ptr(Ep), ptr(k), ptr(pos), ptr(nrm), convention,
            *[ptr(a) for a in outs], ctypes.byref(ms))
        _lib.check(rc, 'xrt_hip_kirchhoff_f64')
        self.lastKernelMs = ms.value
