"""Optical elements of the accelerated path, with the call signatures and return values
of ``xrt.backends.raycing.oes`` (OE + reflect: oes/base.py, oes/reflect.py; ToroidMirror,
BentFlatMirror: oes/__init__.py; conic mirrors: oes/parametric.py; BlazedGrating:
oes/gratings.py; DCM: oes/dcm.py; Plate: oes/refractive.py).

An element here is a DESCRIPTION: where it stands, how it is turned, which surface and
material it has. Everything per ray happens on the GPU:

* ``reflect`` / ``double_reflect`` turn the description into the parameter record of one
  pass (``xrt_hip_pass``, include/xrt_hip.h) and launch the kernels of csrc/reflect.hip
  on device-resident beams;
* the surface functions ``local_z``, ``local_n``, ``local_r``, ``xyz_to_param``,
  ``param_to_xyz`` are evaluated by ``xrt_hip_surface_eval_f64_dev`` -- the code the ray
  kernels use, not a second implementation;
* ``local_to_global`` is ``xrt_hip_local_to_global_f64_dev``.

What lies outside the accelerated subset (figure error, user-defined parametric
surfaces, mosaic / bent crystals, polygon shapes, multi-stripe coatings) raises
NotImplementedError: there is no CPU fallback.
"""
import ctypes
import os

import numpy as np
import torch

from .. import raycing
from ... import _lib, _structs, graphs, hipcalls
from . import sources as rs
from . import stages as rst
from .physconsts import CH

_WIDE = raycing.maxHalfSizeOfOE
_OVER_EDGE = (('xmin', _structs.OVER_XMIN), ('xmax', _structs.OVER_XMAX),
              ('ymin', _structs.OVER_YMIN), ('ymax', _structs.OVER_YMAX))
_SURF_Z, _SURF_N, _SURF_R, _SURF_TO_PARAM, _SURF_FROM_PARAM, _SURF_STATE = range(6)
_BEAM_FIELDS = ('x', 'y', 'z', 'a', 'b', 'c', 'Jss', 'Jpp', 'Jsp', 'Es', 'Ep')


def _device():
    return torch.device('cuda', torch.cuda.current_device())


def _stream():
    return hipcalls.stream_ptr()


def _fill_rotation(record, steps):
    if len(steps) > _structs.MAX_ROT:
        raise ValueError('too many rotation steps')
    record.n = len(steps)
    for k, (axis, cosine, sine) in enumerate(steps):
        record.axis[k], record.cosa[k], record.sina[k] = axis, cosine, sine


def _radius(value, coddington):
    """A curvature radius as the mirrors take it: a number, 0 / None for flat (1e100),
    or (p, q[, pitch]) for the Coddington radius that images p onto q."""
    if isinstance(value, (list, tuple)):
        return coddington(*value)
    if value is None or value == 0:
        return 1e100
    return value


class OE(object):
    """Flat optical element (mirror, crystal, grating by the grating equation, one
    surface of a plate); base of the curved ones."""

    # constructor arguments stored under their own names
    _PLAIN = ('center', 'pitch', 'roll', 'yaw', 'rotationSequence', 'positionRoll',
              'extraPitch', 'extraRoll', 'extraYaw', 'extraRotationSequence',
              'alarmLevel', 'shape', 'surface', 'material', 'alpha', 'limOptX',
              'limOptY', 'limPhysX', 'limPhysY', 'gratingDensity')

    def __init__(self, bl=None, name='', center=[0, 0, 0], pitch=0, roll=0, yaw=0,
                 positionRoll=0, rotationSequence='RzRyRx',
                 extraPitch=0, extraRoll=0, extraYaw=0, extraRotationSequence='RzRyRx',
                 alarmLevel=None, surface=None, material=None, figureError=None,
                 alpha=None, limPhysX=[-_WIDE, _WIDE], limOptX=None,
                 limPhysY=[-_WIDE, _WIDE], limOptY=None, isParametric=False,
                 shape='rect', gratingDensity=None, order=None, **kwargs):
        given = dict(locals())
        if isParametric:
            raise NotImplementedError('user-defined parametric OEs are outside the '
                                      'accelerated path')
        if not isinstance(shape, str) and not raycing.is_sequence(shape):
            raise ValueError('Unknown shape of OE {0}!'.format(name))
        raycing.enrol(self, bl, 'oes', 0, name, type(self).__name__, kwargs.get('uuid'))
        self.isParametric = False
        for key in self._PLAIN:
            setattr(self, key, given[key])
        self.overEdge = kwargs.get('overEdge', 'yMax')
        # a height map added to the surface (figure_error.py; oes/base.py:160, 310): evaluated
        # by the ray kernels, see _figure_params
        self.figureError = figureError
        # one diffraction order, or a sequence to draw from per hit ray
        if order is None:
            self.order = 1
        elif raycing.is_sequence(order):
            self.order = [int(o) for o in order]
        else:
            self.order = int(order)
        self.curSurface, self.dx, self.footprint = 0, 0, []

    # ---- asymmetric cut: the angle between the Bragg planes and the surface -------
    @property
    def alpha(self):
        return self._alpha

    @alpha.setter
    def alpha(self, angle):
        self._alpha = angle
        if angle is not None:
            self.cosalpha, self.sinalpha = float(np.cos(angle)), float(np.sin(angle))

    # ---- the surface ---------------------------------------------------------------
    def _flat_normals(self, second=False):
        """[n_H, n_surface] of a flat element: the Bragg planes are tilted about x by the
        asymmetry angle, the other way round on the second crystal of a DCM."""
        tilt = self.sinalpha if self.alpha else 0.
        nH = [0., -tilt if second else tilt, self.cosalpha if self.alpha else 1.]
        return nH + [0., 0., 1.]

    def _has_source_surface(self):
        from ... import usersurf
        return usersurf.snippets_of(self) is not None

    def local_z(self, x, y):
        if self._has_source_surface():       # hip_local_z, evaluated by its own kernel code
            return self._eval_surface(_SURF_Z, x, y)[0]
        return np.zeros_like(y)

    def local_n(self, x, y):
        if self._has_source_surface():
            return list(self._eval_surface(_SURF_N, x, y)[3:])
        both = self._flat_normals()
        return both if self.alpha else both[:3]

    def local_n2(self, x, y):
        return self.local_n(x, y)

    def local_z_distorted(self, x, y):
        """The height the figure error adds at (x, y) [mm], or None (oes/base.py:681-686)."""
        fe = self.figureError
        if fe is not None and hasattr(fe, 'local_z_distorted'):
            return fe.local_z_distorted(x, y)

    def local_n_distorted(self, x, y):
        """[d_pitch, d_roll] by which the figure error turns the normal at (x, y), or None
        (oes/base.py:744-772)."""
        fe = self.figureError
        if fe is not None and hasattr(fe, 'local_n_distorted'):
            return fe.local_n_distorted(x, y)

    def _figure_params(self, p):
        """OE(figureError=...): the map's spline in HBM -> xrt_hip_pass.fe_* (the kernels add
        its height inside find_dz and turn the normal at the hit point)."""
        fe = getattr(self, 'figureError', None)
        if fe is None:
            return
        if not hasattr(fe, 'device_record'):
            raise NotImplementedError(
                'figureError must be one of xrt_amd.backends.raycing.figure_error (a height map '
                'as a spline: the kernels evaluate it); %r is not' % type(fe).__name__)
        if self.isParametric or self._has_source_surface():
            raise NotImplementedError('figure error on a parametric or user-defined surface')
        rec = fe.device_record(_device())
        p.fe_k, p.fe_nty, p.fe_ntx = rec['k'], rec['nty'], rec['ntx']
        p.fe_ty, p.fe_tx, p.fe_c = rec['ty'], rec['tx'], rec['c']
        p.fe_cy, p.fe_cx = rec['cy'], rec['cx']
        p.fe_shift[0], p.fe_shift[1] = rec['shift']
        for axis, grid in enumerate(rec['grid']):       # (y, x): knots the kernels compute
            p.fe_grid[axis] = 0 if grid is None else 1
            if grid is not None:
                p.fe_lo[axis], p.fe_step[axis], p.fe_hi[axis] = grid
                for j in range(3):
                    p.fe_inv[axis][j] = 1. / ((j + 1) * grid[1])
        p._keep_fe = rec['_keep']

    def _surface_params(self, p, second=False):
        # a subclass that brings its own numpy local_z / local_n (the usual way to define a
        # surface in an xrt script) cannot be evaluated by the kernels: say so instead of
        # tracing a flat surface
        from ... import usersurf
        if usersurf.snippets_of(self) is not None:
            # the class brings its surface as HIP source (the reference: cl_local_z /
            # cl_local_n, oes/base.py:69-90): the ray kernels compiled around it
            p.surf_kind = _structs.SURF_USER
            for k, value in enumerate(usersurf.parameters_of(self)):
                p.surf_p[k] = value
            # (Multilayer / Coated: the unit's flavour that holds Parratt's recursion)
            from . import materials as rmat
            material = getattr(self, 'material2', None) if second else None
            material = material if material is not None else getattr(self, 'material', None)
            if raycing.is_sequence(material):
                material = material[self.curSurface]
            p.user_unit = usersurf.unit_for(self, layered=isinstance(material, rmat.Multilayer))
            p.asymmetric = 0
            for k, value in enumerate((0., 0., 1., 0., 0., 1.)):
                p.n_const[k] = value
            return
        for name in ('local_z', 'local_n'):
            if getattr(type(self), name) is not getattr(OE, name) and \
                    type(self)._surface_params is OE._surface_params:
                raise NotImplementedError(
                    '%s.%s is user-defined in Python: give the class its surface as source '
                    '(hip_local_z / hip_local_n / hip_plist, see xrt_amd/usersurf.py) to run it '
                    'on the GPU' % (type(self).__name__, name))
        p.surf_kind = _structs.SURF_FLAT
        p.asymmetric = 1 if self.alpha else 0
        for k, value in enumerate(self._flat_normals(second and hasattr(self, 'cryst2pitch'))):
            p.n_const[k] = float(value)

    def _curved(self, p, kind, values):
        """Common tail of the curved surfaces' records: kind, parameters, no Bragg planes."""
        p.surf_kind = kind
        for k, value in enumerate(values):
            p.surf_p[k] = float(value)
        p.asymmetric = 0
        for k, value in enumerate((0., 0., 1., 0., 0., 1.)):
            p.n_const[k] = value

    def _eval_surface(self, what, u, v, w=None):
        """The surface function *what* of this element at host points, evaluated on the
        GPU by the ray kernels' own code -> list of output arrays shaped like *u*."""
        _lib.require_gpu()
        shape = np.shape(u) if np.ndim(u) else np.shape(v)
        pts = [np.ascontiguousarray(np.broadcast_to(np.asarray(c, dtype=float), shape)).ravel()
               for c in ((u, v) if w is None else (u, v, w))]
        count = pts[0].size
        dev = _device()
        nout = (1, 6, 1, 3, 3, 1)[what]
        ins = [torch.from_numpy(c.copy()).to(dev) for c in pts]
        out = torch.empty(nout * max(count, 1), dtype=torch.float64, device=dev)
        if what == _SURF_STATE:
            p = self._make_pass(*self._own_angles()[:4])
        else:
            p = _structs.Pass()
            p.invert_normal = 1
            self._surface_params(p)
        _lib.check(_lib.load().xrt_hip_surface_eval_f64_dev(
            ctypes.byref(p), what, count, *[ctypes.c_void_p(t.data_ptr()) for t in ins],
            *([None] if w is None else []), ctypes.c_void_p(out.data_ptr()), _stream()),
            'xrt_hip_surface_eval_f64_dev')
        res = out.cpu().numpy()[:nout * count].reshape(nout, count)
        return [r.reshape(shape) for r in res]

    def _eval_surface_dev(self, what, u, v):
        """The same for points that are on the GPU already (1-D float64 tensors) -> a
        [nout, n] tensor there."""
        count = u.numel()
        nout = (1, 6, 1, 3, 3, 1)[what]
        out = torch.empty((nout, max(count, 1)), dtype=torch.float64, device=u.device)
        if what == _SURF_STATE:
            p = self._make_pass(*self._own_angles()[:4])
        else:
            p = _structs.Pass()
            p.invert_normal = 1
            self._surface_params(p)
        _lib.check(_lib.load().xrt_hip_surface_eval_f64_dev(
            ctypes.byref(p), what, count, ctypes.c_void_p(u.data_ptr()),
            ctypes.c_void_p(v.data_ptr()), None, ctypes.c_void_p(out.data_ptr()), _stream()),
            'xrt_hip_surface_eval_f64_dev')
        return out[:, :count]

    def rays_good(self, x, y, z=None, is2ndXtal=False):
        """State of a ray that hits the surface at local (x, y): 1 good, 2 out, 3 over,
        ``lostNum`` absorbed (reference oes/base.py:1094-1163) -- evaluated on the GPU."""
        if is2ndXtal:
            raise NotImplementedError('rays_good of a second crystal')
        return self._eval_surface(_SURF_STATE, x, y)[0].astype(np.int32)

    def _surface_height(self, x, y):
        """z of the surface above (x, y) -- on a parametric surface the point of the
        surface that has these x and y."""
        if type(self).local_z is OE.local_z and not self.isParametric and \
                not self._has_source_surface():
            return self.local_z(x, y)
        return self._eval_surface(_SURF_Z, x, y)[0]

    # ---- gratings that deflect by the grating equation -------------------------------
    def local_g(self, x, y, rho=-100.):
        """Reciprocal groove vector [1/mm] at (x, y). With *gratingDensity* = ['x'|'y',
        rho0, p0, p1, ...] the line density along that axis is
        rho0 (p0 + 2 p1 w + 3 p2 w^2 + ...); without it (0, rho, 0). A subclass may
        override this with a CONSTANT vector (it is evaluated once, on the host)."""
        spec = self.gratingDensity
        if spec is None:
            return 0, rho, 0
        along_x = spec[0] == 'x'
        w = x if along_x else y
        density = spec[1] * sum((k + 1) * coef * w**k for k, coef in enumerate(spec[2:]))
        nothing = np.zeros_like(density)
        return (density, nothing, nothing) if along_x else (nothing, density, nothing)

    def _is_grating(self):
        material = self.material
        if raycing.is_sequence(material):
            material = material[0] if len(material) == 1 else None
        kind = getattr(material, 'kind', None)
        return kind in ('grating', 'FZP') or (kind == 'auto' and self.gratingDensity is not None)

    def _grating_params(self, p, second=False):
        p.grating = 0
        if second or not self._is_grating():
            return
        if self.isParametric:
            raise NotImplementedError('grating equation on a parametric surface')
        several = raycing.is_sequence(self.order)
        p.grating, p.grating_order = 1, int(self.order[0] if several else self.order)
        material = self.material[0] if raycing.is_sequence(self.material) else self.material
        pairs = getattr(material, 'efficiency', None)
        if pairs is not None:
            if len(pairs) > 8:
                raise NotImplementedError('more than 8 efficiency entries')
            p.eff_n = len(pairs)
            if getattr(material, 'efficiencyFile', None) is not None:
                # the second number is a column of the file: a row of the table in HBM,
                # interpolated at each ray's energy by the kernel
                tab_E, tab_I = material.efficiency_on_device(_device())
                p.eff_tab_n = int(tab_E.numel())
                p.eff_tab_E, p.eff_tab_I = tab_E.data_ptr(), tab_I.data_ptr()
                p._keep_eff = (tab_E, tab_I)
                p._eff_range = (float(material.efficiency_E[0]), float(material.efficiency_E[-1]))
                for k, (order, column) in enumerate(pairs):
                    p.eff_order[k], p.eff_amp[k] = int(order), 0.
            else:
                for k, (order, value) in enumerate(pairs):
                    p.eff_order[k], p.eff_amp[k] = int(order), float(np.float64(value)**0.5)
        if hasattr(self, '_zones_between_passes'):
            p.grating = 0                 # general zone plate: its first pass is geometry only
            return
        if hasattr(self, 'rn'):           # zone plate: the zone radii instead of a groove vector
            held = self.__dict__.setdefault('_zone_table', {})     # in HBM, per device
            cached = held.get(str(_device()))
            if cached is None or cached[0] is not self.rn:
                table = np.ascontiguousarray(self.rn, dtype=np.float64)
                cached = held[str(_device())] = (
                    self.rn, torch.from_numpy(table.copy()).to(_device()))
            p.grating, p.grating_axis = 2, -1
            p.zone_n, p.zone_black = len(self.rn) - 1, int(bool(self.isCentralZoneBlack))
            p.zone_r = cached[1].data_ptr()
            return
        from ... import usersurf
        if usersurf.snippets_of(self) is not None and usersurf.groove_snippet_of(self):
            p.grating_axis = 2            # the groove function compiled into the class's unit
            return
        spec = self.gratingDensity
        if spec is not None and type(self).local_g is OE.local_g:
            coefs = [float(c) for c in spec[2:]]
            if len(coefs) > 8 or spec[0] not in ('x', 'y'):
                raise NotImplementedError('gratingDensity %r' % (spec,))
            p.grating_axis, p.g_rho0, p.g_ncoef = 'xy'.index(spec[0]), float(spec[1]), len(coefs)
            for k, c in enumerate(coefs):
                p.g_coef[k] = c
            return
        # a subclass's own local_g: accepted if it is the same vector everywhere
        probe_x, probe_y = np.array([-0.7, 0., 0.3, 1.1]), np.array([0.9, 0., -1.3, 0.2])
        vector = [np.broadcast_to(np.asarray(c, dtype=float), probe_x.shape)
                  for c in self.local_g(probe_x, probe_y)]
        if any(np.ptp(c) != 0 for c in vector):
            raise NotImplementedError('a user-defined position-dependent local_g: express '
                                      'it as gratingDensity')
        p.grating_axis = -1
        for k in range(3):
            p.g_const[k] = float(vector[k][0])

    # ---- Coddington radii that image a source at p onto q ----------------------------
    def get_Rmer_from_Coddington(self, p, q, pitch=None):
        grazing = abs(self.pitch if pitch is None else pitch)
        return 2 * p * q / (p+q) / np.sin(grazing)

    def get_rsag_from_Coddington(self, p, q, pitch=None):
        grazing = abs(self.pitch if pitch is None else pitch)
        return 2 * p * q / (p+q) * np.sin(grazing)

    # ---- the parameter record of one pass --------------------------------------------
    def _turns(self, pitch, roll, yaw, second):
        """The two rotation lists of a pass: world -> true local frame and back. On the
        second crystal of a DCM the frame is first rolled by pi and the extra angles of
        pitch and yaw change sign."""
        flip = -1. if second else 1.
        extra = self.extraPitch or self.extraRoll or self.extraYaw
        forth = raycing.rotation_steps(roll=-np.pi) if second else []
        forth += raycing.rotation_steps(self.rotationSequence, -pitch, -roll, -yaw)
        back = []
        if extra:
            forth += raycing.rotation_steps(self.extraRotationSequence, -flip*self.extraPitch,
                                            -self.extraRoll, -flip*self.extraYaw)
            back += raycing.rotation_steps('-' + self.extraRotationSequence,
                                           flip*self.extraPitch, self.extraRoll,
                                           flip*self.extraYaw)
        back += raycing.rotation_steps('-' + self.rotationSequence, pitch, roll, yaw)
        if second:
            back += raycing.rotation_steps(roll=np.pi)
        return forth, back

    def _make_pass(self, pitch, roll, yaw, dx=0, dy=0, dz=0, fromVacuum=True,
                   is2ndXtal=False, noIntersectionSearch=False, in_is_global=True,
                   good_mode=0, out_to_global=True, only_state1_out=False,
                   zero_local_not_entering=False, force_lost_out=False):
        p = _structs.Pass()
        p.good_mode, p.in_is_global = good_mode, int(bool(in_is_global))
        for k in range(3):
            p.center[k] = float(self.center[k])
            p.shift[k] = float((dx, dy, dz)[k] or 0.)
        p.sin_az, p.cos_az = (0., 1.) if self.bl is None else \
            (self.bl.sinAzimuth, self.bl.cosAzimuth)
        forth, back = self._turns(pitch, roll, yaw, is2ndXtal)
        _fill_rotation(p.to_local, forth)
        _fill_rotation(p.to_virgin, back)
        # what the batch statistics of this surface chose last time (secant / Brent): the
        # optimistic pass assumes it again (one int in HBM per surface, see xrt_hip.h)
        if torch.cuda.is_available():       # (the record itself can be made without a GPU)
            hints = self.__dict__.setdefault('_method_hints', {})
            key = (str(_device()), bool(is2ndXtal), bool(fromVacuum))
            if key not in hints:
                hints[key] = torch.zeros(1, dtype=torch.int32, device=_device())
            p.method_hint = hints[key].data_ptr()
        p.invert_normal = int(getattr(self, 'invertNormal', 1 if fromVacuum else -1))
        p.no_intersection_search = int(bool(noIntersectionSearch))
        self._surface_params(p, is2ndXtal)
        # limits of this surface (the second crystal has its own)
        suffix = '2' if is2ndXtal else ''
        for axis, lim in (('x', 'limPhysX'), ('y', 'limPhysY')):
            lo, hi = self._limits_of(lim + suffix)
            target = getattr(p, 'phys_' + axis)
            target[0], target[1] = float(lo), float(hi)
        for axis, lim in (('x', 'limOptX'), ('y', 'limOptY')):
            optical = self._limits_of(lim + suffix)
            setattr(p, 'has_opt_' + axis, 0 if optical is None else 1)
            if optical is not None:
                target = getattr(p, 'opt_' + axis)
                target[0], target[1] = float(optical[0]), float(optical[1])
        shapes = {'re': _structs.SHAPE_RECT, 'ro': _structs.SHAPE_ROUND}
        if not isinstance(self.shape, str):
            # outline of the optical surface as (x, y) vertices: kept in HBM for the kernels
            outline = np.ascontiguousarray(self.shape, dtype=np.float64).reshape(-1, 2)
            held = self.__dict__.setdefault('_outline', {})        # per device
            cached = held.get(str(_device()))
            if cached is None or not np.array_equal(cached[0], outline):
                cached = held[str(_device())] = (
                    outline.copy(), torch.from_numpy(outline.copy()).to(_device()))
            p.shape, p.poly_n, p.poly_xy = _structs.SHAPE_POLYGON, len(outline), \
                cached[1].data_ptr()
        elif self.shape[:2] in shapes:
            p.shape = shapes[self.shape[:2]]
        else:
            raise NotImplementedError('shape %r' % (self.shape,))
        edges = str(getattr(self, 'overEdge', 'yMax')).lower()
        p.over_mask = sum(bit for word, bit in _OVER_EDGE if word in edges)
        p.lost_num = int(self.lostNum)
        p.roll, p.cos_roll, p.sin_roll = float(roll), float(np.cos(roll)), float(np.sin(roll))
        p.out_to_global = int(bool(out_to_global))
        p.only_state1_out = int(bool(only_state1_out))
        p.zero_local_not_entering = int(bool(zero_local_not_entering))
        p.force_lost_out = int(bool(force_lost_out))
        self._grating_params(p, is2ndXtal)
        self._figure_params(p)
        return p

    def _limits_of(self, name):
        """(lo, hi) of the limits *name* for the present stripe: elements with several
        surfaces (``surface`` = their names) give each limit as a sequence per stripe
        (reference oes/base.py:1050-1092)."""
        value = getattr(self, name, None)
        if value is not None and raycing.is_sequence(value[0]):
            return value[0][self.curSurface], value[1][self.curSurface]
        return value

    def get_surface_limits(self):
        for kind in ('Phys', 'Opt'):
            for axis in 'XY':
                setattr(self, 'surf%s%s' % (kind, axis), self._limits_of('lim%s%s' % (kind, axis)))

    def _material_struct(self, material, fromVacuum, device, beam=None):
        if raycing.is_sequence(material):     # coating stripes: the one in the beam
            material = material[self.curSurface]
        if material is not None:
            if beam is not None and isinstance(getattr(material, 'refractiveIndex', None), list):
                # a tabulated index is evaluated at the energies of THIS beam
                return material.to_struct(fromVacuum, device, E=beam.dev('E', device))
            return material.to_struct(fromVacuum, device)
        s = _structs.Material()
        s.kind, s.from_vacuum, s._keep = _structs.MAT_NONE, int(bool(fromVacuum)), []
        return s

    def _adopt(self, beams, parent):
        for b in beams:
            rs.inherit_scalars(b, parent)
            b.parentId = self.uuid

    @staticmethod
    def _check_efficiency_range(p, beam, made, dev):
        """The reference refuses energies outside the efficiency table (material.py:399-407)
        among the rays that HIT the grating (its `good`: state 1 in the beam the pass *made*);
        so does this, right after the launch (one reduction on the device and one number read
        back). Not while a HIP graph is recorded or replayed: the eager iteration before the
        recording checks, the replays interpolate with the table's end values."""
        E = beam.dev('E', dev)
        Emin, Emax = p._eff_range
        if made is None:        # (multiple_reflect: before its loop, the rays that enter it)
            hit = beam.dev('state', dev) > 0
        else:
            hit = made.dev('state', dev) == 1
        bad = hit & ((E < Emin) | (E > Emax))
        if bool(bad.any()):
            raise ValueError(
                'E={0} is out of the efficiency table range [{1}, {2}]!!! Use another '
                'table.'.format(E[bad].cpu().numpy(), Emin, Emax))

    def _run_pass(self, p, material, fromVacuum, beam_in, restore, want_info=False,
                  timing=False, out=None, local=True):
        """-> (lb, vlb) device-resident beams (+ info dict). *out*: an (lb, vlb)
        pair from an earlier call on a beam of the same size to be overwritten
        instead of allocating new arrays. *local* False: no local beam is made (lb is
        returned as None; mirrors, plates and gratings only)."""
        _lib.require_gpu()
        lib = _lib.load()
        dev = _device()
        if out is not None:              # (beams about to be overwritten in place)
            for b in out:
                # (also a LazyBeam that has its arrays: snapshots of them may wait to be read,
                # reads() looks at what is there without filling anything)
                if b is not None and (type(b) is not rs.LazyBeam or b.__dict__['_filled']):
                    rs.flush_pending(b)
        ms = self._material_struct(material, fromVacuum, dev, beam_in)
        s_in = beam_in.to_struct(dev)
        s_re = s_in if restore is beam_in else restore.to_struct(dev)
        n = beam_in.nrays
        usable = lambda b: b is not None and b.nrays == n and not b._h_dirty() and \
            b.has_amplitudes() == beam_in.has_amplitudes()      # noqa: E731
        if not local:
            lb, theta = None, None
            vb = out[1] if out is not None and usable(out[1]) else \
                rs.Beam.empty_like_on_device(beam_in, dev)
        elif out is not None and all(usable(b) for b in out) and 'theta' in out[0]._d:
            lb, vb = out
            theta = lb._d['theta']
        else:
            lb = rs.Beam.empty_like_on_device(beam_in, dev)
            vb = rs.Beam.empty_like_on_device(beam_in, dev)
            theta = torch.empty(n, dtype=torch.float64, device=dev)   # fully written
        ws = hipcalls.workspace(dev, lib.xrt_hip_reflect_workspace_bytes(n), 'reflect')
        info = (ctypes.c_double * 16)() if want_info else None
        ms_out = (ctypes.c_float * 3)() if timing else None
        _lib.check(lib.xrt_hip_reflect_pass_f64_dev(
            ctypes.byref(p), ctypes.byref(ms), ctypes.byref(s_in), ctypes.byref(s_re),
            ctypes.byref(lb.to_struct(dev)) if local else None, ctypes.byref(vb.to_struct(dev)),
            ctypes.c_void_p(theta.data_ptr()) if local else None,
            ctypes.c_void_p(ws.data_ptr()), ws.numel(),
            _stream(), info, ms_out), 'xrt_hip_reflect_pass_f64_dev')
        if local and lb._d.get('theta') is not theta:
            lb._h.pop('theta', None)
            lb._d['theta'] = theta
        if p.eff_tab_n > 0 and graphs.capturing() is None:
            # (recorded into a HIP graph: checked by the eager iteration before it)
            self._check_efficiency_range(p, beam_in, lb if local else vb, dev)
        self._adopt((lb, vb) if local else (vb,), beam_in)
        report = None
        if want_info:
            v = list(info)
            report = dict(axis=int(v[0]), positive=bool(v[1]), brent=bool(v[2]),
                          tMinGlobal=v[3], tMaxGlobal=v[4], maxdz1=v[5], maxdz2=v[6],
                          n_enter=int(v[7]), mixed_sign=bool(v[10] and v[11]))
        if timing:
            report = report or {}
            report.update(pass_ms=ms_out[0], kernel_ms=ms_out[1],
                          exact_sequence=bool(ms_out[2]))
        return lb, vb, report

    # ---- frames --------------------------------------------------------------------------
    def _own_angles(self, second=False):
        """(pitch, roll, yaw, dx, dy, dz) of this surface in the element's virgin frame."""
        if not hasattr(self, 'cryst2pitch'):
            return self.pitch, self.roll + self.positionRoll, self.yaw, 0, 0, 0
        if second:
            return (-self.pitch - self.bragg + self.cryst2pitch + self.cryst2finePitch,
                    self.roll + self.cryst2roll + self.positionRoll, -self.yaw,
                    -self.dx, self.cryst2longTransl, -self.cryst2perpTransl)
        return (self.pitch + self.bragg, self.roll + self.positionRoll + self.cryst1roll,
                self.yaw, self.dx, 0, 0)

    def local_to_global(self, lb, returnBeam=False, is2ndXtal=False, **kwargs):
        """*lb* from the true local frame of this surface into the global frame, in place
        (with *returnBeam* the last step, virgin local -> global, goes into a copy that is
        returned): positions, directions, and the coherency matrix / amplitudes turned
        about the ray like the reference does (oes/base.py:1165-1229). On the GPU."""
        _lib.require_gpu()
        pitch, roll, yaw, dx, dy, dz = self._own_angles(is2ndXtal)
        p = self._make_pass(pitch, roll, yaw, dx, dy, dz, is2ndXtal=is2ndXtal,
                            out_to_global=not returnBeam)
        # the polarisation frame turns by roll + positionRoll, whatever the crystal's own roll
        turn = self.roll + self.positionRoll
        p.roll, p.cos_roll, p.sin_roll = float(turn), float(np.cos(turn)), float(np.sin(turn))
        dev = _device()
        lb.to_struct(dev)
        rs.flush_pending(lb)             # (lb changes in place)
        _lib.check(_lib.load().xrt_hip_local_to_global_f64_dev(
            ctypes.byref(p), ctypes.byref(lb.to_struct(dev)), _stream()),
            'xrt_hip_local_to_global_f64_dev')
        for name in _BEAM_FIELDS:          # the kernel rewrote these arrays in HBM
            lb._h.pop(name, None)
        if returnBeam:
            out = rs.Beam(copyFrom=lb)
            raycing.virgin_local_to_global(self.bl, out, self.center)
            return out

    def _to_global_points(self, x, y, z, second=False):
        """Surface points (true local frame) -> global coordinates."""
        pts = rs.Beam(nrays=len(x), forceState=1, withAmplitudes=True)
        pts.x[:], pts.y[:], pts.z[:] = x, y, z
        self.local_to_global(pts, is2ndXtal=second)
        return pts

    # ---- wave propagation through the element ------------------------------------------------
    def _wave_samples(self, nrays, shape, area):
        """Where the wave is sampled on the surface -> (x, y, area): *nrays* random points
        (one (n, 2) draw from numpy's global generator: column 0 -> x or r^2, column 1 -> y
        or the azimuth), an (nx, ny) mesh over the physical limits, or the mesh of two
        given coordinate arrays."""
        (x0, x1), (y0, y1) = self.limPhysX, self.limPhysY
        rect = (x1 - x0) * (y1 - y0)
        round_ = shape.startswith('ro')
        if not round_ and not shape.startswith('re'):
            raise ValueError('unknown shape!')
        if isinstance(nrays, (int, float)):
            draw = np.random.rand(int(nrays), 2)
            if round_:
                radius = (x1 - x0) / 2
                rho, phi = draw[:, 0]**0.5 * radius, draw[:, 1] * 2*np.pi
                return rho * np.cos(phi), rho * np.sin(phi), \
                    np.pi * radius**2 if area == 'auto' else area
            return draw[:, 0] * (x1 - x0) + x0, draw[:, 1] * (y1 - y0) + y0, \
                rect if area == 'auto' else area
        if not isinstance(nrays, (list, tuple)):
            raise ValueError('wrong type of `nrays`!')
        if round_:
            raise ValueError('must be rectangular')
        if all(isinstance(v, (int, float)) for v in nrays[:2]):
            axes = np.linspace(x0, x1, nrays[0]), np.linspace(y0, y1, nrays[1])
        elif all(isinstance(v, np.ndarray) for v in nrays[:2]):
            axes = nrays[:2]
        else:
            raise ValueError('wrong type of `nrays`!')
        gx, gy = np.meshgrid(*axes)
        return gx.ravel(), gy.ravel(), rect if area == 'auto' else area

    def _anchor(self):
        """The point a wave is aimed from: the middle of the surface, in global
        coordinates."""
        mx = (self.limPhysX[1] + self.limPhysX[0])*0.5
        my = (self.limPhysY[1] + self.limPhysY[0])*0.5
        mz = self._surface_height(np.atleast_1d(float(mx)), np.atleast_1d(float(my)))
        mid = self._to_global_points(np.atleast_1d(mx), np.atleast_1d(my), mz)
        return [mid.x[0], mid.y[0], mid.z[0]]

    def prepare_wave(self, prevOE, nrays, shape='auto', area='auto', rw=None):
        """Wave samples on this surface that will receive the field diffracted by
        *prevOE* (reference oes/reflect.py:266-403). The samples become rays aimed from
        the middle of *prevOE* at the sample points and are sent through ``reflect``:
        what comes back in state 1 or 2 is the local wave, its area the sampled area
        times the surviving fraction."""
        if rw is None:
            from . import waves as rw
        x, y, area = self._wave_samples(nrays, self.shape if shape == 'auto' else shape, area)
        # The sample points are drawn on the host (numpy's generator, in the reference's
        # order); everything after that stays on the GPU: surface height, frame changes,
        # aiming, the ray pass, the selection of the survivors, their coordinates in
        # prevOE's frame -- the same IEEE operations in the same order as the reference's
        # numpy expressions (products and sums as separate elementwise operations).
        dev = _device()
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
        xd, yd = up(x), up(y)
        if type(self).local_z is OE.local_z and not self.isParametric:
            zd = up(self.local_z(x, y))
        else:
            zd = self._eval_surface_dev(_SURF_Z, xd, yd)[0].contiguous()
        aimed = rs.Beam.on_device(len(x), dev, withAmplitudes=True, state=1)
        aimed.x, aimed.y, aimed.z = xd, yd, zd
        self.local_to_global(aimed)
        aimed.parentId = prevOE.uuid
        origin = prevOE._anchor() if hasattr(prevOE, 'rotationSequence') else prevOE.center
        # rays from `origin` towards the sample points
        da = aimed.dev('x', dev) - float(origin[0])
        db = aimed.dev('y', dev) - float(origin[1])
        dc = aimed.dev('z', dev) - float(origin[2])
        length = torch.sqrt(da * da + db * db + dc * dc)
        aimed.a, aimed.b, aimed.c = da / length, db / length, dc / length
        aimed.x = torch.full_like(da, float(origin[0]))
        aimed.y = torch.full_like(da, float(origin[1]))
        aimed.z = torch.full_like(da, float(origin[2]))
        # projection of the area on the line of sight from prevOE: the local normal at the
        # origin of the surface against the direction to that origin
        pole = rs.Beam(nrays=1)
        pole.b[:], pole.c[:] = 0., 1.
        self.local_to_global(pole)
        sight = [getattr(pole, position) - o for position, o in zip('xyz', origin)]
        reach = (sight[0]**2 + sight[1]**2 + sight[2]**2)**0.5
        tilt = abs(float(((sight[0]*pole.a[0] + sight[1]*pole.b[0] + sight[2]*pole.c[0])
                          / reach)[0]))
        waveGlobal, waveLocal = self.reflect(aimed)        # HIP kernels
        state = waveLocal.dev('state', dev)
        alive = torch.nonzero((state == 1) | (state == 2)).squeeze(1)
        waveGlobal.filter_by_index(alive)
        waveLocal.filter_by_index(alive)
        area *= int(alive.numel()) / float(len(x))         # (the one number that crosses back)
        waveLocal.area, waveLocal.areaNormal = area, area * tilt
        waveLocal.dS = area / float(len(x))
        waveLocal.toOE, waveLocal.parentId = self, self.uuid
        rw.prepare_wave(prevOE, waveLocal, waveGlobal.dev('x', dev), waveGlobal.dev('y', dev),
                        waveGlobal.dev('z', dev))
        return waveLocal

    def propagate_wave(self, wave=None, beam=None, nrays='auto'):
        """Brings the field *wave* (the local beam on the previous element, or the beam of a
        source) onto this surface as a wave and reflects it: -> (beamGlobal, beamLocal) usable
        for further ray or wave propagation (oes/reflect.py:405-449). From an optical element
        or aperture: prepare_wave -> diffract -> reflect(noIntersectionSearch) -- the sequence
        the reference's wave examples spell out. From a source: prepare_wave, the source's
        ``shine(wave=...)`` onto those samples, reflect with the intersection search. The
        reference's own method additionally passes *wave* through the auto-alignment hooks,
        which transform it in place (reflect.py:434-438); that side effect is not reproduced
        (golden cases G8 are generated with explicit positions)."""
        from . import waves as rw
        prevOE = self.bl.oesDict[wave.parentId][0]
        size = len(wave.x) if nrays == 'auto' else int(nrays)
        receiving = self.prepare_wave(prevOE, size, rw=rw)
        if hasattr(prevOE, 'shine'):
            arriving = prevOE.shine(wave=receiving)
            glo, loc = self.reflect(arriving)
        else:
            arriving = rw.diffract(wave, receiving)
            glo, loc = self.reflect(arriving, noIntersectionSearch=True)
        loc.parentId = self.uuid
        return glo, loc

    def reflect(self, beam=None, needLocal=True, noIntersectionSearch=False,
                returnLocalAbsorbed=None, _info=None, out=None, _timing=None):
        """-> (beamGlobal, beamLocal). *needLocal* False: no local beam is made and the
        global beam is returned in its place, as in the reference (oes/reflect.py:104-108:
        ``lb = gb``) -- the pass then writes 200 B per ray instead of 308 (mirrors, plates,
        single-order gratings; crystals and zone plates keep theirs). *out* (extension): the
        pair returned by an earlier call, to be overwritten in place (no new HBM
        allocations). *_info* (dict) receives the batch statistics (this takes the exact
        kernel sequence); *_timing* (dict) the pass / kernel milliseconds and whether the
        exact sequence had to run."""
        from . import materials as _rm
        stripes = self.material if raycing.is_sequence(self.material) else (self.material,)
        crystal_like = any(isinstance(m, (_rm.Crystal, _rm.Multilayer)) for m in stripes)
        local = bool(needLocal) or crystal_like or \
            getattr(self, '_zones_between_passes', None) is not None or \
            raycing.is_sequence(getattr(self, 'order', None))
        p = self._make_pass(
            self.pitch + getattr(self, 'bragg', 0), self.roll + self.positionRoll,
            self.yaw, self.dx, noIntersectionSearch=noIntersectionSearch,
            only_state1_out=hasattr(beam, 'createdByDiffract'))
        # (at most one chain -- a source and the pass that takes its beam -- ever waits)
        rs.flush_pending(keep=beam.__dict__.get('_op') if type(beam) is rs.LazyBeam else None)
        if fuseConsumers and local and out is None and _info is None and _timing is None and \
                getattr(self, '_zones_between_passes', None) is None and \
                not (p.grating and raycing.is_sequence(self.order)) and not p.eff_tab_n:
            # not launched yet: Screen.expose of the global beam may still join the pass
            return _DeferredReflect(self, p, beam, out).hand_out()
        lb, gb, report = self._run_pass(
            p, self.material, True, beam, beam, want_info=_info is not None,
            timing=_timing is not None, out=None if out is None else (out[1], out[0]),
            local=local)
        if not local:
            lb = gb
        between = getattr(self, '_zones_between_passes', None)
        if between is not None:       # zones that depend on the whole batch (general FZP)
            lb, gb, report, held = between(p, beam, lb, gb, _info, _timing)
        if p.grating and raycing.is_sequence(self.order):
            lb, gb, report = self._with_ray_orders(p, beam, lb, gb, _info, _timing)
        clock = ('pass_ms', 'kernel_ms', 'exact_sequence')
        if _info is not None:
            _info.update({k: v for k, v in report.items() if k not in clock})
        if _timing is not None:
            _timing.update({k: report[k] for k in clock})
        return gb, lb


# OE.reflect -> Screen.expose as one pass (see sources.LazyBeam). XRT_HIP_NO_FUSE=1 in the
# environment, or oes.fuseConsumers = False, keeps every call an immediate launch.
fuseConsumers = os.environ.get('XRT_HIP_NO_FUSE', '') != '1'


def _as_it_is(beam, own_states=False, sharer=None):
    """-> (a beam of the same device arrays as *beam* has at this moment, their ids). Later
    assignments to *beam* replace its arrays and do not reach the copy; writes INTO the arrays
    come after their readers (sources.flush_pending). *own_states*: with a copy of the states,
    which apertures change in place at any time."""
    snap = rs.Beam.__new__(rs.Beam)
    object.__setattr__(snap, '_h', {})
    object.__setattr__(snap, '_d', dict(beam._d))
    rs.inherit_scalars(snap, beam)
    object.__setattr__(snap, 'parentId', getattr(beam, 'parentId', None))
    if 'createdByDiffract' in beam.__dict__:
        snap.createdByDiffract = beam.createdByDiffract
    held = set(id(t) for t in snap._d.values())
    if own_states and sharer is not None:
        sharer._share_states(snap)        # (its own copy when an aperture is about to write them)
    elif own_states:
        snap._d['state'] = snap._d['state'].clone()
    return snap, held


def _locals_on_demand(oe, *materials):
    """True if the element's passes may leave their local beams out (written -- by the same
    pass run again -- when somebody first looks at one, after which the element writes them
    at once): not the layered kernels, not an element that was asked before."""
    from . import materials as _rm
    if not fuseConsumers or oe.__dict__.get('_local_beams_wanted'):
        return False
    for material in materials:
        stripes = material if raycing.is_sequence(material) else (material,)
        if any(isinstance(m, _rm.Multilayer) for m in stripes):
            return False
    return True


class _LocalsOnDemand(rs.SharesStates, rs.FillsBeams):
    """The local beams of an element whose pass has written the global beam only (308 -> 200 B
    per ray and surface): *run(beam)* -> the real local beams, called with the input as it was
    (its own copy of the states) the first time one of them is looked at. The element then
    remembers (``_local_beams_wanted``) and writes them in its pass from the next call on."""
    optional = True

    def __init__(self, oe, beam, count, run):
        self.oe, self.run = oe, run
        beam.to_struct(_device())                    # everything up in HBM now
        self.was, self.tensors = _as_it_is(beam, own_states=True, sharer=self)
        self.count = count
        oe._adopt([self._make(k) for k in range(count)], beam)
        self.state = 'pending'
        rs._PENDING.add(self)

    def reads(self, beam):
        d = beam.__dict__.get('_real_d', beam.__dict__.get('_d')) or {}
        return any(id(t) in self.tensors for t in d.values())

    def materialize(self, which=None):
        rs._PENDING.discard(self)
        if self.state != 'done':
            self.oe.__dict__['_local_beams_wanted'] = True
            # (the state changes when the launch has returned: a launch that raises is raised
            # again by the next look at the beam instead of leaving it empty)
            made = self.run(self.was)
            self.state = 'done'
            for k, real in enumerate(made):
                rs.adopt_into(self._beam(k), real)
            self.was = self.run = None
            self.tensors = ()


class _DeferredReflect(rs.SharesStates, rs.FillsBeams):
    """OE.reflect not launched yet. States: pending -> done (plain pass: both beams), or
    pending -> global (the next element took the global beam: the pass without its local beam,
    200 instead of 308 B per ray) -> done (the local beam, by the pass run again, if somebody
    looks at it), or pending -> imaged (the pass with a screen in its tail: the image, and the
    local beam if the element knows it is looked at; the global beam was not written) -> done
    (the beam asked for, on demand). From its first launch on the record holds its own copy
    of the input's states and lives as long as its beams do (``optional``)."""

    def __init__(self, oe, p, beam, out=None):
        dev = _device()
        self.src_op = None
        made_by = beam.__dict__.get('_op') if type(beam) is rs.LazyBeam else None
        if isinstance(made_by, rs._DeferredShine) and made_by.state == 'pending':
            # the beam of a device source that has not run either: kept as it is (its rays
            # can be made inside this element's pass, image_on)
            self.src_op, snap = made_by, beam
            self.tensors = set()
        else:
            beam.to_struct(dev)                      # everything up in HBM now
            snap, self.tensors = _as_it_is(beam)     # the input as it is at this moment
        self.oe, self.p, self.beam, self.out = oe, p, snap, out
        # the material (the stripe in the beam) as the element has it NOW: the pass record was
        # made at this moment too, a scan loop may set another before anybody looks at a beam
        material = oe.material
        if raycing.is_sequence(material):
            material = material[oe.curSurface]
        self.material = material
        self.n = self.src_op.n if self.src_op is not None else snap.nrays
        self.screen = self.screen_rec = None
        self.apertures = []           # [(aperture, its record)]: marks_later
        self.state = 'pending'
        oe._adopt((self._make('gb'), self._make('lb')), beam)
        rs._PENDING.add(self)

    # (weak: sources.FillsBeams)
    gb = property(lambda self: self._beam('gb'))
    lb = property(lambda self: self._beam('lb'))
    image = property(lambda self: self._beam('image'))

    def reads(self, beam):
        if beam is self.beam:
            return True
        d = beam.__dict__.get('_real_d', beam.__dict__.get('_d')) or {}
        return any(id(t) in self.tensors for t in d.values())

    def _waits_with_its_own_states(self):
        """After a launch that left a beam out: what is needed to make it later."""
        held = self.beam
        if type(held) is not rs.LazyBeam or held.__dict__['_filled']:
            self.beam, self.tensors = _as_it_is(held, own_states=True, sharer=self)
        # (else: the rays of a source that were made in this pass's registers -- its record
        # makes them again)
        self.optional = True
        rs._PENDING.add(self)

    def takes_aperture(self, aperture):
        """Can ``aperture.propagate(gb)`` wait for the pass? Up to two apertures without an
        outline of vertices right behind the element, before any screen has looked at the beam
        (a screen sees the states as they are when it is exposed)."""
        return self.state == 'pending' and self.screen_rec is None and self.out is None and \
            len(self.apertures) < 2 and bool(self.p.out_to_global) and \
            not hasattr(aperture, 'vertices') and \
            not aperture.__dict__.get('_local_beam_wanted')

    def marks_later(self, aperture):
        """RectangularAperture.propagate of the global beam while the pass is still pending
        (reference apertures.py:334-413 right after oes/reflect.py): the aperture's marks are
        made in the tail of the pass, on the ray in registers (xrt_hip_reflect_tail_f64_dev) --
        a slit between two mirrors costs no launch and no traffic of its own. -> the beam in the
        aperture's frame, made when somebody first looks at it (the pass again + the aperture's
        own kernel; the aperture remembers and takes its own launch from then on)."""
        k = len(self.apertures)
        self.apertures.append((aperture, aperture._record()))
        gb = self.gb
        gb.__dict__.setdefault('_stopped_by', set()).add(aperture.lostNum)
        local = self._make('ap%d' % k)
        rs.inherit_scalars(local, gb)
        return self.hand_out()

    def _aperture_locals_filled(self):
        return all(rs.filled(self._beam('ap%d' % k)) for k in range(len(self.apertures)))

    def _mark(self, gb):
        """The apertures' marks in a global beam that a repeated pass has made."""
        for _, rec in self.apertures:
            _lib.check(_lib.load().xrt_hip_aperture_propagate_f64_dev(
                ctypes.byref(rec), ctypes.byref(gb.to_struct(_device())), None, None, _stream()),
                'xrt_hip_aperture_propagate_f64_dev')
        gb._h.pop('state', None)

    def _aperture_local(self, k):
        """The beam in the frame of aperture *k* after all: the pass again (the global beam with
        the states as the element leaves them), the marks of the apertures before this one,
        then the aperture's full kernel."""
        if self.state == 'pending':
            self.materialize('gb')
        target = self._beam('ap%d' % k)
        if rs.filled(target):
            return
        aperture, rec = self.apertures[k]
        aperture.__dict__['_local_beam_wanted'] = True
        rays = self.beam
        if type(rays) is rs.LazyBeam:      # (a source's, not made when the pass ran)
            rays = self.src_op.rays_again()
        _, gb, _ = self.oe._run_pass(self.p, self.material, True, rays, rays, local=False)
        dev = _device()
        lib = _lib.load()
        for _, before in self.apertures[:k]:
            _lib.check(lib.xrt_hip_aperture_propagate_f64_dev(
                ctypes.byref(before), ctypes.byref(gb.to_struct(dev)), None, None, _stream()),
                'xrt_hip_aperture_propagate_f64_dev')
        local = rs.Beam.empty_like_on_device(gb, dev)
        _lib.check(lib.xrt_hip_aperture_propagate_f64_dev(
            ctypes.byref(rec), ctypes.byref(gb.to_struct(dev)),
            ctypes.byref(local.to_struct(dev)), None, _stream()),
            'xrt_hip_aperture_propagate_f64_dev')
        rs.adopt_into(target, local)
        if rs.filled(self.lb) and rs.filled(self.gb) and rs.filled(self.image) and \
                self._aperture_locals_filled():
            rs._PENDING.discard(self)
            self.state = 'done'
            self.beam, self.tensors = None, ()

    def expose_later(self, screen, rec):
        """Screen.expose of the global beam while the pass is still pending: the image is handed
        out before anything is launched as well, so that a plot of it may still join the pass
        (plot_on: run_ray_tracing's accumulate_plot). Whoever looks at a beam first launches."""
        self.screen, self.screen_rec = screen, rec
        rs.inherit_scalars(self._make('image'), self.beam)
        return self.hand_out()

    def materialize(self, which=None):
        oe = self.oe
        filled = rs.filled
        if which is not None and which.startswith('ap'):
            return self._aperture_local(int(which[2:]))
        if self.state == 'pending' and (self.screen_rec is not None or self.apertures):
            # the pass with the apertures and the screen in its tail; then the beam asked for,
            # if it was left out
            self._launch_with_screen(local=True if which == 'lb' else None)
            if which in (None, 'image') or filled(self.gb if which == 'gb' else self.lb):
                return
        if self.state == 'pending':
            rs._PENDING.discard(self)
            # (the state changes when the launch has returned: one that raises is raised again
            # by the next look at a beam instead of leaving the beams empty)
            if which != 'lb' and self.out is None and _locals_on_demand(oe, self.material):
                _, gb, _ = oe._run_pass(self.p, self.material, True, self.beam, self.beam,
                                        local=False)
                self.state = 'global'
                rs.adopt_into(self.gb, gb)
                self._waits_with_its_own_states()
                return
            lb, gb, _ = oe._run_pass(self.p, self.material, True, self.beam, self.beam,
                                     out=None if self.out is None else (self.out[1], self.out[0]))
            self.state = 'done'
            rs.adopt_into(self.lb, lb)
            rs.adopt_into(self.gb, gb)
        elif self.state in ('global', 'imaged') and which == 'image':
            # the image of a pass that fed a plot and nothing else (plot_on): the pass with the
            # screen again -- and the screen remembers: next time the image is written as well
            if not filled(self.image):
                self.screen.__dict__['_image_wanted'] = True
                rays = self.beam
                if type(rays) is rs.LazyBeam:      # (a source's, not made when the pass ran)
                    rays = self.src_op.rays_again()
                _, _, image, _ = oe._run_pass_screen(self.p, self.material, rays, self.screen_rec,
                                                     keep_global=False, local=False)
                rs.adopt_into(self.image, image)
            if filled(self.lb) and filled(self.gb) and self._aperture_locals_filled():
                rs._PENDING.discard(self)
                self.state = 'done'
        elif self.state in ('global', 'imaged'):
            # somebody wants a beam that was left out after all: the pass again -- and the
            # element remembers: next time its pass writes that beam as well
            want_lb = not filled(self.lb) and which != 'gb'
            want_gb = not filled(self.gb) and which != 'lb'
            if want_lb:
                oe.__dict__['_local_beams_wanted'] = True
            if want_gb:
                oe.__dict__['_global_beam_wanted'] = True
            if want_lb or want_gb:
                rays = self.beam
                if type(rays) is rs.LazyBeam:      # (a source's, not made when the pass ran)
                    rays = self.src_op.rays_again()
                lb, gb, _ = oe._run_pass(self.p, self.material, True, rays, rays, local=want_lb)
                if want_lb:
                    rs.adopt_into(self.lb, lb)
                if want_gb:
                    self._mark(gb)
                    rs.adopt_into(self.gb, gb)
            if filled(self.lb) and filled(self.gb) and filled(self.image) and \
                    self._aperture_locals_filled():
                rs._PENDING.discard(self)
                self.state = 'done'
        if self.state == 'done':
            self.beam, self.tensors = None, ()

    def _launch_with_screen(self, plot=None, local=None):
        """The pass with its tail: the apertures that wait (marks_later), the screen
        (expose_later) and *plot*, a _structs.PlotTail showing the screen's image, behind it
        -> True; False if *plot* cannot ride this pass (nothing has been launched then)."""
        oe = self.oe
        src = self.src_op
        screened = self.screen_rec is not None
        tabulated = isinstance(getattr(self.material, 'refractiveIndex', None), list)
        # (without a screen the marked global beam is what the pass is for)
        keep = bool(oe.__dict__.get('_global_beam_wanted')) or not screened
        keep_image = screened and (plot is None or bool(self.screen.__dict__.get('_image_wanted')))
        if local is None:
            local = not _locals_on_demand(oe, self.material)
        from_source = src is not None and src.state == 'pending' and not tabulated
        made = oe._run_pass_screen(self.p, self.material, None if from_source else self.beam,
                                   self.screen_rec, source=src if from_source else None,
                                   keep_global=keep, local=local, plot=plot,
                                   keep_image=keep_image,
                                   apertures=[rec for _, rec in self.apertures])
        if made is None:
            return False
        rs._PENDING.discard(self)
        lb, gb, image, fused = made
        if local:
            rs.adopt_into(self.lb, lb)
        if fused and not keep:
            self._scratch = gb            # (the redo's scratch: freed with this record)
        else:
            rs.adopt_into(self.gb, gb)
        if keep_image:
            rs.adopt_into(self.image, image)
        if local and (keep_image or not screened) and not (fused and not keep) and \
                self._aperture_locals_filled():
            self.state = 'done'
            self.beam, self.tensors = None, ()
        else:
            self.state = 'imaged' if screened else 'global'
            self._waits_with_its_own_states()
        return True

    def image_on(self, screen, rec):
        """The pass with *screen* in its tail, at once -> the screen's image."""
        image = self.expose_later(screen, rec)
        self._launch_with_screen()
        return image

    def plot_on(self, tail):
        """run_ray_tracing adds the screen's image to a plot and nothing else has looked at
        it: the plot joins the pass (reference: raycing/__init__.py:170-300 + multipro.py:
        316-361 right after screens.py:226-302). The image itself is written only if the
        screen has been asked for it before (``_image_wanted``); -> False if this pass or this
        plot cannot do that (nothing launched: the caller takes the usual route)."""
        if self.state != 'pending' or self.screen_rec is None:
            return False
        return self._launch_with_screen(plot=tail)


class _DeferredDouble(_DeferredReflect):
    """DCM.double_reflect / Plate.double_refract not launched yet (pairs the fused kernels take,
    local beams on demand): the global beam and the beams on the two surfaces are handed out
    first; what the script does next decides the launch, as for OE.reflect -- apertures and a
    flat screen that take the global beam ride in the tail of the pair's kernel
    (xrt_hip_double_reflect_tail_f64_dev; reference: dcm.py:248-354 -> apertures.py:334-413 ->
    screens.py:226-302). States: pending -> global / imaged (the local beams, and with a screen
    the global beam, left out) -> done (on demand, by the pair's pass again)."""

    def __init__(self, oe, p1, p2, from_vacuum, beam):
        dev = _device()
        beam.to_struct(dev)                          # everything up in HBM now
        self.src_op = None
        self.beam, self.tensors = _as_it_is(beam)
        self.oe, self.p1, self.p2, self.from_vacuum = oe, p1, p2, from_vacuum
        self.p, self.out = p2, None                  # (what a screen / an aperture looks at)
        self.materials = (oe.material, oe.material2)
        self.n = self.beam.nrays
        self.screen = self.screen_rec = None
        self.apertures = []
        self.state = 'pending'
        oe._adopt((self._make('gb'), self._make('lo1'), self._make('lo2')), beam)
        rs._PENDING.add(self)

    lo1 = property(lambda self: self._beam('lo1'))
    lo2 = property(lambda self: self._beam('lo2'))

    def plot_on(self, tail):
        return False                 # (a plot of the image: its own launches)

    def _all_filled(self):
        f = rs.filled
        return f(self.gb) and f(self.lo1) and f(self.lo2) and f(self.image) and \
            self._aperture_locals_filled()

    def _pair(self, rays, local, tail=False, keep=True):
        """The pair's pass on *rays* -> (gb, lo1, lo2, image, fused); *tail*: with the apertures
        and the screen that wait."""
        oe = self.oe
        return oe._run_double_tail(
            self.p1, self.p2, self.materials, self.from_vacuum, rays, local,
            self.screen_rec if tail else None,
            [rec for _, rec in self.apertures] if tail else (), keep)

    def _finish(self):
        if self._all_filled():
            rs._PENDING.discard(self)
            self.state = 'done'
            self.beam, self.tensors = None, ()

    def materialize(self, which=None):
        filled = rs.filled
        if which is not None and which.startswith('ap'):
            return self._aperture_local(int(which[2:]))
        oe = self.oe
        if self.state == 'pending':
            rs._PENDING.discard(self)
            screened = self.screen_rec is not None
            keep = bool(oe.__dict__.get('_global_beam_wanted')) or not screened
            local = which in ('lo1', 'lo2')
            gb, lo1, lo2, image, fused = self._pair(self.beam, local, tail=True, keep=keep)
            if local:
                rs.adopt_into(self.lo1, lo1)
                rs.adopt_into(self.lo2, lo2)
            if fused and not keep:
                self._scratch = gb            # (the redo's scratch: freed with this record)
            else:
                rs.adopt_into(self.gb, gb)
            if screened:
                rs.adopt_into(self.image, image)
            if self._all_filled():
                self.state = 'done'
                self.beam, self.tensors = None, ()
                return
            self.state = 'imaged' if screened else 'global'
            self._waits_with_its_own_states()
            if which is None or which == 'image' or \
                    filled({'gb': self.gb, 'lo1': self.lo1, 'lo2': self.lo2}.get(which)):
                return
        if self.state in ('global', 'imaged'):
            # somebody wants a beam that was left out after all: the pair's pass again -- and the
            # element remembers: next time it writes that beam at once
            want_local = which in (None, 'lo1', 'lo2') and not (filled(self.lo1) and filled(self.lo2))
            want_gb = which in (None, 'gb') and not filled(self.gb)
            if which in ('lo1', 'lo2'):
                oe.__dict__['_local_beams_wanted'] = True
            if which == 'gb':
                oe.__dict__['_global_beam_wanted'] = True
            if which is not None and (want_local or want_gb):
                gb, lo1, lo2, _, _ = self._pair(self.beam, want_local)
                if want_local:
                    rs.adopt_into(self.lo1, lo1)
                    rs.adopt_into(self.lo2, lo2)
                if not filled(self.gb):
                    self._mark(gb)
                    rs.adopt_into(self.gb, gb)
            self._finish()

    def _aperture_local(self, k):
        if self.state == 'pending':
            self.materialize('gb')
        target = self._beam('ap%d' % k)
        if rs.filled(target):
            return
        aperture, rec = self.apertures[k]
        aperture.__dict__['_local_beam_wanted'] = True
        gb = self._pair(self.beam, False)[0]
        dev = _device()
        lib = _lib.load()
        for _, before in self.apertures[:k]:
            _lib.check(lib.xrt_hip_aperture_propagate_f64_dev(
                ctypes.byref(before), ctypes.byref(gb.to_struct(dev)), None, None, _stream()),
                'xrt_hip_aperture_propagate_f64_dev')
        local = rs.Beam.empty_like_on_device(gb, dev)
        _lib.check(lib.xrt_hip_aperture_propagate_f64_dev(
            ctypes.byref(rec), ctypes.byref(gb.to_struct(dev)),
            ctypes.byref(local.to_struct(dev)), None, _stream()),
            'xrt_hip_aperture_propagate_f64_dev')
        rs.adopt_into(target, local)
        self._finish()


def _scratch_beam(role, n, dev, amplitudes):
    """A beam-sized scratch that a fused pass writes only if it has to be redone, kept from call
    to call per (thread, stream) for the beams of the size the host bounds (three allocations
    and a record per beam = 13 us of a 0.2-ms iteration); ``taken`` when the call turned it
    into a real beam after all."""
    if n > _SCRATCH_RAYS:
        return rs.Beam.empty_on_device(n, dev, amplitudes)
    held = hipcalls._tls.__dict__.setdefault('scratch_beams', {})
    key = (role, dev.index, hipcalls.raw_stream(dev.index), int(n), bool(amplitudes))
    beam = held.get(key)
    if beam is None:
        beam = held[key] = rs.Beam.empty_on_device(n, dev, amplitudes)
    return beam


def _scratch_taken(role, beam):
    held = hipcalls._tls.__dict__.get('scratch_beams', {})
    for key in [k for k, b in held.items() if b is beam]:
        del held[key]


_scratch_beam.taken = _scratch_taken
_SCRATCH_RAYS = 2_000_000


def _run_pass_screen(self, p, material, beam_in, screen_record, source=None, keep_global=False,
                     local=True, plot=None, keep_image=True, apertures=()):
    """OE.reflect + Screen.expose in one C call (xrt_hip_reflect_screen_f64_dev) ->
    (lb, gb, image, fused): fused = the lean kernel carried the screen and gb holds nothing;
    *local* False: no local beam either (lb None).
    *source* (a pending sources._DeferredShine, with beam_in None): the rays are made by the
    source's record inside the same call (xrt_hip_shine_reflect_screen_f64_dev).
    *plot* (a _structs.PlotTail): the plot of the image behind the screen
    (xrt_hip_reflect_screen_plot_f64_dev); the image itself only with *keep_image*. -> None,
    and nothing is launched, if this pass cannot carry the plot.
    *apertures* (up to two _structs.Aperture records): their marks between the element and the
    screen (xrt_hip_reflect_tail_f64_dev); *screen_record* may then be None (no image)."""
    _lib.require_gpu()
    lib = _lib.load()
    dev = _device()
    if source is not None:
        n, amp = source.n, source.amplitudes
        parent = source.beam
        ms = self._material_struct(material, True, dev, None)
        scratch = _scratch_beam('source', n, dev, amp)      # (written only for a redo)
    else:
        n, amp, parent = beam_in.nrays, beam_in.has_amplitudes(), beam_in
        ms = self._material_struct(material, True, dev, beam_in)
        s_in = beam_in.to_struct(dev)
    if plot is not None and not lib.xrt_hip_reflect_screen_plot_fusable(
            ctypes.byref(p), ctypes.byref(ms), ctypes.byref(screen_record), ctypes.byref(plot), n):
        return None
    # (the global beam nobody keeps: the redo's scratch)
    gb = rs.Beam.empty_on_device(n, dev, amp) if keep_global else _scratch_beam('global', n, dev, amp)
    keep_image = keep_image and screen_record is not None
    image = rs.Beam.empty_on_device(n, dev, amp) if keep_image else None
    image_ref = ctypes.byref(image.to_struct(dev)) if keep_image else None
    lb = rs.Beam.empty_on_device(n, dev, amp) if local else None
    theta = torch.empty(n, dtype=torch.float64, device=dev) if local else None
    lb_ref = ctypes.byref(lb.to_struct(dev)) if local else None
    theta_ref = ctypes.c_void_p(theta.data_ptr()) if local else None
    ws = hipcalls.workspace(dev, lib.xrt_hip_reflect_workspace_bytes(n), 'reflect')
    fused = ctypes.c_int(0)
    if apertures:
        tail = _structs.Tail()
        tail.n_apertures, tail.keep_screen = len(apertures), int(keep_image)
        for k, rec in enumerate(apertures):
            tail.aperture[k] = rec
        if screen_record is not None:
            tail.screen = ctypes.addressof(screen_record)
        if keep_image:
            tail.out_screen = ctypes.addressof(image.to_struct(dev))
        if plot is not None:
            tail.plot = ctypes.addressof(plot)
        rays = scratch.to_struct(dev) if source is not None else s_in
        _lib.check(lib.xrt_hip_reflect_tail_f64_dev(
            ctypes.byref(source.g) if source is not None else None, ctypes.byref(p),
            ctypes.byref(ms), ctypes.byref(rays), ctypes.byref(rays), lb_ref,
            ctypes.byref(gb.to_struct(dev)), theta_ref, ctypes.byref(tail), int(keep_global),
            ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream(), ctypes.byref(fused)),
            'xrt_hip_reflect_tail_f64_dev')
    elif source is not None and plot is not None:
        _lib.check(lib.xrt_hip_shine_reflect_screen_plot_f64_dev(
            ctypes.byref(source.g), ctypes.byref(p), ctypes.byref(ms),
            ctypes.byref(scratch.to_struct(dev)), lb_ref,
            ctypes.byref(gb.to_struct(dev)), theta_ref,
            ctypes.byref(screen_record), image_ref, int(keep_global), int(keep_image),
            ctypes.byref(plot), ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream(),
            ctypes.byref(fused)), 'xrt_hip_shine_reflect_screen_plot_f64_dev')
    elif plot is not None:
        _lib.check(lib.xrt_hip_reflect_screen_plot_f64_dev(
            ctypes.byref(p), ctypes.byref(ms), ctypes.byref(s_in), ctypes.byref(s_in),
            lb_ref, ctypes.byref(gb.to_struct(dev)), theta_ref, ctypes.byref(screen_record),
            image_ref, int(keep_global), int(keep_image), ctypes.byref(plot),
            ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream(), ctypes.byref(fused)),
            'xrt_hip_reflect_screen_plot_f64_dev')
    elif source is not None:
        _lib.check(lib.xrt_hip_shine_reflect_screen_f64_dev(
            ctypes.byref(source.g), ctypes.byref(p), ctypes.byref(ms),
            ctypes.byref(scratch.to_struct(dev)), lb_ref,
            ctypes.byref(gb.to_struct(dev)), theta_ref,
            ctypes.byref(screen_record), ctypes.byref(image.to_struct(dev)), int(keep_global),
            ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream(), ctypes.byref(fused)),
            'xrt_hip_shine_reflect_screen_f64_dev')
    else:
        _lib.check(lib.xrt_hip_reflect_screen_f64_dev(
            ctypes.byref(p), ctypes.byref(ms), ctypes.byref(s_in), ctypes.byref(s_in),
            lb_ref, ctypes.byref(gb.to_struct(dev)),
            theta_ref, ctypes.byref(screen_record),
            ctypes.byref(image.to_struct(dev)), int(keep_global), ctypes.c_void_p(ws.data_ptr()),
            ws.numel(), _stream(), ctypes.byref(fused), None), 'xrt_hip_reflect_screen_f64_dev')
    if source is not None:
        if fused.value & 2:
            rs._PENDING.discard(source)
            source.state = 'inflight'       # (still makes its beam if somebody asks for it)
        else:
            _scratch_beam.taken('source', scratch)
            source.adopt(scratch)           # the generator's own launch has filled it
    if local:
        lb._d['theta'] = theta
    if not (fused.value & 1):
        _scratch_beam.taken('global', gb)      # (it holds the element's global beam after all)
    self._adopt((lb, gb) if local else (gb,), parent)
    if image is not None:
        rs.inherit_scalars(image, parent)
    return lb, gb, image, bool(fused.value & 1)


OE._run_pass_screen = _run_pass_screen


def _ray_orders(self, p, beam, lb, gb, _info, _timing):
    """A sequence of diffraction orders: every ray that hits (state 1 in the local beam —
    which order it takes does not move its hit point) gets one, drawn with numpy's global
    generator the way the reference draws it (oes/reflect.py:455-458, one randint call of
    the size of the hit set, in ray order), then the pass is repeated with the draw as a
    per-ray array. The local beam carries it as *order* like the reference's."""
    graphs.refuse('diffraction orders drawn per ray with numpy\'s generator')
    dev = _device()
    hit = lb.dev('state', dev) == 1
    count = int(hit.sum())
    choice = np.asarray(self.order)[np.random.randint(len(self.order), size=count)]
    per_ray = torch.zeros(beam.nrays, dtype=torch.int32, device=dev)
    per_ray[hit] = torch.as_tensor(choice.astype(np.int32), device=dev)
    p.order_ray = per_ray.data_ptr()
    try:
        lb, gb, report = self._run_pass(
            p, self.material, True, beam, beam, want_info=_info is not None,
            timing=_timing is not None, out=(lb, gb))
        torch.cuda.current_stream().synchronize()     # per_ray is released below
    finally:
        p.order_ray = None
    lb.order = per_ray.to(torch.float64)
    return lb, gb, report


OE._with_ray_orders = _ray_orders


class _Footprints(object):
    """The arrays of lbN -- every ray after every bounce of ``multiple_reflect``, bounce after
    bounce (reference: ``lbN.concatenate(lb)``, oes/reflect.py:231-235) -- in HBM: one tensor
    per field with room for *cap* bounces, doubled when it runs out. A bounce kernel writes
    slice k and reads slice k - 1: the beam never exists anywhere else."""

    def __init__(self, n, dev, amplitudes, elevation, parametric, cap):
        self.n, self.dev, self.cap = int(n), dev, max(int(cap), 1)
        f64 = list(rs._F64) + ['theta']
        if elevation:
            f64 += ['elevationD', 'elevationX', 'elevationY', 'elevationZ']
        if parametric:
            f64 += ['s', 'phi', 'r']
        self.kinds = [(name, torch.float64) for name in f64]
        self.kinds += [('Jsp', torch.complex128)]
        if amplitudes:
            self.kinds += [('Es', torch.complex128), ('Ep', torch.complex128)]
        self.kinds += [('state', torch.int32), ('nRefl', torch.int32)]
        self.t = {name: torch.empty(self.cap * self.n, dtype=dt, device=dev)
                  for name, dt in self.kinds}

    def grow(self, used):
        cap = self.cap * 2
        for name, dt in self.kinds:
            t = torch.empty(cap * self.n, dtype=dt, device=self.dev)
            t[:used * self.n] = self.t[name][:used * self.n]
            self.t[name] = t
        self.cap = cap

    def ptr(self, name, k):
        t = self.t[name]
        return t.data_ptr() + k * self.n * t.element_size()

    def beam_struct(self, k):
        s = _structs.Beam()
        s.n = self.n
        for cname, name in (('x', 'x'), ('y', 'y'), ('z', 'z'), ('a', 'a'), ('b', 'b'),
                            ('c', 'c'), ('path', 'path'), ('E', 'E'), ('Jss', 'Jss'),
                            ('Jpp', 'Jpp'), ('Jsp_ri', 'Jsp'), ('state', 'state')):
            setattr(s, cname, self.ptr(name, k))
        if 'Es' in self.t:
            s.Es_ri, s.Ep_ri = self.ptr('Es', k), self.ptr('Ep', k)
        else:
            s.Es_ri = s.Ep_ri = None
        s._keep = list(self.t.values())
        return s

    def slice(self, name, k):
        return self.t[name][k * self.n:(k + 1) * self.n]

    def beam(self, first, last, skip=()):
        """Beam over the bounces first..last-1 (views of the buffers)."""
        b = rs.Beam.__new__(rs.Beam)
        object.__setattr__(b, '_h', {})
        object.__setattr__(b, '_d', {})
        for name, _ in self.kinds:
            if name not in skip:
                b._d[name] = self.t[name][first * self.n:last * self.n]
        object.__setattr__(b, 'parentId', None)
        return b


def _multiple_reflect(self, beam=None, maxReflections=1000, needElevationMap=False,
                      returnLocalAbsorbed=None, _info=None):
    """-> (beamGlobal, beamLocalN): like :meth:`reflect`, with up to *maxReflections*
    reflections off the same surface (reference oes/reflect.py:165-264). The returned beams
    carry *nRefl*, and with *needElevationMap* *elevationD*, *elevationX/Y/Z*: the greatest
    distance between ray and surface on the way from one impact point to the next, and where.
    beamLocalN holds ALL rays after every bounce (the footprints), bounce after bounce, in the
    element's virgin local frame like the reference's.

    On the GPU one launch per bounce (csrc/reflect_multi_impl.h); the beam stays in HBM, two
    numbers per bounce come back (how many rays are left, how many hit). Mirrors, plates,
    single-order gratings and bare surfaces; flat / toroidal / bent-flat surfaces, parametric
    conics (capillaries), cones, VFMs and user-defined surfaces. *_info* (list) receives one
    dictionary of batch decisions per bounce."""
    from . import materials as _rm
    if int(maxReflections) < 1:
        raise ValueError('maxReflections must be at least 1 (got %r)' % (maxReflections,))
    graphs.refuse('multiple_reflect (the rays decide how many bounces there are)')
    _lib.require_gpu()
    lib = _lib.load()
    dev = _device()
    self.footprint = []
    stripes = self.material if raycing.is_sequence(self.material) else (self.material,)
    if any(isinstance(m, (_rm.Crystal, _rm.Multilayer)) for m in stripes):
        raise NotImplementedError('multiple_reflect with crystals or layered materials')
    if getattr(self, '_zones_between_passes', None) is not None or \
            raycing.is_sequence(getattr(self, 'order', None)):
        raise NotImplementedError('multiple_reflect with zone plates / several orders')
    n = beam.nrays
    if n == 0 or not bool((beam.dev('state', dev) > 0).any()):    # reflect.py:203-205
        gb = rs.Beam(copyFrom=beam)
        self._adopt((gb,), beam)
        return gb, gb
    first = self._make_pass(self.pitch, self.roll + self.positionRoll, self.yaw, self.dx,
                            out_to_global=False)
    later = self._make_pass(self.pitch, self.roll + self.positionRoll, self.yaw, self.dx,
                            in_is_global=False, good_mode=1, out_to_global=False)
    later.is_multi = 1
    first.need_elevation_map = later.need_elevation_map = int(bool(needElevationMap))
    for p in (first, later):
        if p.eff_tab_n > 0:
            self._check_efficiency_range(p, beam, None, dev)
    ms = self._material_struct(self.material, True, dev, beam)
    amplitudes = beam.has_amplitudes()
    fp = _Footprints(n, dev, amplitudes, needElevationMap, bool(self.isParametric),
                     min(int(maxReflections), 8))
    ws = hipcalls.workspace(dev, lib.xrt_hip_bounce_workspace_bytes(n), 'bounce')
    counts = (ctypes.c_int64 * 2)()
    stats = (ctypes.c_double * 16)() if _info is not None else None
    # which method (secant / Brent) the two searches of bounce k took the last time this element
    # was called: the guess the optimistic form of the bounce starts from (xrt_hip_bounce)
    found = (ctypes.c_int32 * 4)()
    methods = self.__dict__.setdefault('_multi_methods', {})
    guess = (0, 0)
    elevation = ('elevationD', 'elevationX', 'elevationY', 'elevationZ')
    k, hits_first, hits_any = 0, False, False
    while k < maxReflections:
        if k == fp.cap:
            fp.grow(k)
        bounce = _structs.Bounce()
        bounce.nrefl_in = None if k == 0 else fp.ptr('nRefl', k - 1)
        bounce.entering_hint = 0 if k == 0 else left     # (rays in state 1 or 2 after bounce k - 1)
        hit_brent, tan_brent = methods.get(k, guess)
        bounce.assume_hit_brent, bounce.assume_tangency_brent = int(hit_brent), int(tan_brent)
        bounce.found_host = ctypes.addressof(found)
        bounce.nrefl_out = fp.ptr('nRefl', k)
        bounce.theta = fp.ptr('theta', k)
        for j, name in enumerate(elevation):
            if needElevationMap:
                bounce.elev_in[j] = None if k == 0 else fp.ptr(name, k - 1)
                bounce.elev_out[j] = fp.ptr(name, k)
        if self.isParametric:
            for j, name in enumerate(('s', 'phi', 'r')):
                bounce.spr_out[j] = fp.ptr(name, k)
        s_in = beam.to_struct(dev) if k == 0 else fp.beam_struct(k - 1)
        s_out = fp.beam_struct(k)
        _lib.check(lib.xrt_hip_reflect_bounce_f64_dev(
            ctypes.byref(first if k == 0 else later), ctypes.byref(ms), ctypes.byref(s_in),
            ctypes.byref(s_out), ctypes.byref(bounce), ctypes.c_void_p(ws.data_ptr()),
            ws.numel(), _stream(), counts, stats), 'xrt_hip_reflect_bounce_f64_dev')
        left, hit = int(counts[0]), int(counts[1])
        guess = methods[k] = (int(found[0]), int(found[1]))   # (the next bounce's guess, too)
        if hit == 0 and k > 0:
            # a bounce in which no ray hits leaves lb.theta as it was (reflect.py:791-796)
            fp.slice('theta', k).copy_(fp.slice('theta', k - 1))
        hits_first = hits_first or (k == 0 and hit > 0)
        hits_any = hits_any or hit > 0
        if _info is not None:
            v = list(stats)
            one = dict(axis=int(v[0]), positive=bool(v[1]), brent=bool(v[2]),
                       n_enter=int(v[3]), tMinGlobal=v[7], tMaxGlobal=v[8], left=left, hit=hit)
            if k > 0:
                one['tangency'] = dict(brent=bool(v[4]), tMinGlobal=v[5], tMaxGlobal=v[6])
            _info.append(one)
        k += 1
        if left == 0:
            break
    # lb.theta exists from the first bounce with a hit on; lbN took (or did not take) it at
    # bounce 0 and concatenate() keeps what both sides have (sources/beams.py:271-272)
    lbN = fp.beam(0, k, skip=() if hits_first else ('theta',))
    last = fp.beam(k - 1, k, skip=() if hits_any else ('theta',))
    gb = rs.Beam.empty_like_on_device(beam, dev)
    _lib.check(lib.xrt_hip_multiple_reflect_out_f64_dev(
        ctypes.byref(first), ctypes.byref(fp.beam_struct(k - 1)),
        ctypes.byref(beam.to_struct(dev)), ctypes.c_void_p(fp.ptr('nRefl', k - 1)),
        ctypes.byref(gb.to_struct(dev)), _stream()), 'xrt_hip_multiple_reflect_out_f64_dev')
    for name in last.array_fields():       # gb is lb: it carries the bounce's other arrays
        if name not in gb._d:
            gb._d[name] = last._d[name].clone()
    self._adopt((gb, lbN), beam)
    return gb, lbN


OE.multiple_reflect = _multiple_reflect


class _Curved(OE):
    """Surfaces whose height and normal the GPU evaluates."""

    def local_z(self, x, y):
        return self._eval_surface(_SURF_Z, x, y)[0]

    def local_n(self, x, y):
        return self._eval_surface(_SURF_N, x, y)[3:]


class ToroidMirror(_Curved):
    """Toroidal mirror, z = y^2/(2R) + r - sqrt(r^2 - x^2): meridional radius *R*,
    sagittal radius *r*, each a number or a (p, q) pair for the Coddington radius
    (reference oes/__init__.py:321-411)."""

    def __init__(self, *args, **kwargs):
        radii = kwargs.pop('R', 5.0e6), kwargs.pop('r', 50.)
        OE.__init__(self, *args, **kwargs)
        self.R, self.r = radii

    R = property(lambda self: self._RVal,
                 lambda self, v: setattr(self, '_RVal',
                                         _radius(v, self.get_Rmer_from_Coddington)))
    r = property(lambda self: self._rVal,
                 lambda self, v: setattr(self, '_rVal',
                                         _radius(v, self.get_rsag_from_Coddington)))

    def _surface_params(self, p, second=False):
        # correctly rounded reciprocals for the constant-divisor division of the
        # kernel; only for ordinary radii (R = 1e100 / inf mean 'flat')
        ok = all(np.isfinite(v) and 1e-100 < abs(v) < 1e100 for v in (self.R, self.r))
        self._curved(p, _structs.SURF_TOROID,
                     (self.R, self.r, 1.0 / float(self.R) if ok else 0.,
                      1.0 / float(self.r) if ok else 0., 1.0 if ok else 0.))


SimpleVFM = ToroidMirror


class BentFlatMirror(_Curved):
    """Meridionally bent parabolic cylinder with fixed ends:
    z = (y^2 - limPhysY[0]^2) / (2R) (oes/__init__.py:240-303)."""

    def __init__(self, *args, **kwargs):
        bend = kwargs.pop('R', 5.0e6)
        OE.__init__(self, *args, **kwargs)
        self.R = bend

    R = property(lambda self: self._RVal,
                 lambda self, v: setattr(self, '_RVal',
                                         _radius(v, self.get_Rmer_from_Coddington)))

    def _surface_params(self, p, second=False):
        ok = bool(np.isfinite(self.R)) and 1e-100 < abs(self.R) < 1e100
        self._curved(p, _structs.SURF_BENTFLAT,
                     (self.R, self.limPhysY[0]**2, 1.0 / float(self.R) if ok else 0., 0.,
                      1.0 if ok else 0.))


SimpleVCM = BentFlatMirror


class ConicalMirror(_Curved):
    """Mirror on the inside of a cone: *L0* = distance from the mirror centre to the
    vertex measured ALONG THE SURFACE, *theta* = half opening angle (axis to surface)
    (reference oes/__init__.py:589-636)."""

    def __init__(self, *args, **kwargs):
        self.L0 = kwargs.pop('L0', 1000.)
        self.theta = kwargs.pop('theta', np.pi/6.)
        OE.__init__(self, *args, **kwargs)

    @property
    def theta(self):
        return self._theta

    @theta.setter
    def theta(self, value):
        self._theta = raycing.auto_units_angle(value)
        self.tt, self.t2t = np.tan(self._theta), np.tan(2*self._theta)
        self.redfocus = np.cos(self._theta)**2 / (1./self.tt - 1./self.t2t)

    def _surface_params(self, p, second=False):
        t2t = self.t2t
        self._curved(p, _structs.SURF_CONE,
                     (self.L0, 0.25*t2t**2, self.redfocus*t2t, -0.5*t2t, np.sign(t2t),
                      self.redfocus, t2t, .5*t2t))


class MirrorOnTripodWithTwoXStages(OE, rst.Tripod, rst.TwoXStages):
    """A mirror on three jacks (*jack1..3*: [x, y, z], global) and two x stages (*tx1*, *tx2*:
    [x, y], local; *dx*: nominal x shift) (reference oes/__init__.py:212-238)."""
    _mirror = OE

    def __init__(self, *args, **kwargs):
        kwargs, jacks = rst.Tripod.pop_kwargs(self, **kwargs)
        kwargs, stages = rst.TwoXStages.pop_kwargs(self, **kwargs)
        self._mirror.__init__(self, *args, **kwargs)
        rst.Tripod.__init__(self, *jacks)
        rst.TwoXStages.__init__(self, *stages)

    def get_orientation(self):
        """x shift, height and the three angles from the stages and the jacks."""
        rst.TwoXStages.get_orientation(self)
        rst.Tripod.get_orientation(self)


class VCM(MirrorOnTripodWithTwoXStages, BentFlatMirror):
    """Vertically collimating (bent flat) mirror on its support (oes/__init__.py:309-317)."""
    _mirror = BentFlatMirror


class VFM(MirrorOnTripodWithTwoXStages, ToroidMirror):
    """Vertically focusing mirror on its support: a sagittal cylinder of radius *r* --
    levelled off beyond the optical x limits --, bent meridionally to *R* with fixed ends
    (oes/__init__.py:417-478)."""
    _mirror = ToroidMirror

    def __init__(self, *args, **kwargs):
        if kwargs.get('limPhysY') is None:
            raise AttributeError('limPhysY must be given')
        MirrorOnTripodWithTwoXStages.__init__(self, *args, **kwargs)

    def _surface_params(self, p, second=False):
        cap = np.inf
        edges = (0., 0.)
        if self.limOptX is not None:
            edges = self.limOptX
            cap = self.r - (self.r**2 - self.limOptX[1]**2)**0.5
        self._curved(p, _structs.SURF_VFM, (self.r, self.r**2, cap, self.limPhysY[0]**2,
                                            self.R, edges[0], edges[1]))


class DualVFM(MirrorOnTripodWithTwoXStages, _Curved):
    """Two focusing cylinders side by side in one substrate: radii *r1*, *r2*, axes at the
    local x *xCylinder1*, *xCylinder2*, sunk by *hCylinder1*, *hCylinder2* below the flat
    top; meridional radius *R* (oes/__init__.py:480-587)."""

    def __init__(self, *args, **kwargs):
        self.R = kwargs.pop('R', 5.0e6)
        self.r1, self.xCylinder1, self.hCylinder1 = (
            kwargs.pop('r1', 70.0), kwargs.pop('xCylinder1', 23.5),
            kwargs.pop('hCylinder1', 3.7035))
        self.r2, self.xCylinder2, self.hCylinder2 = (
            kwargs.pop('r2', 35.98), kwargs.pop('xCylinder2', -25.0),
            kwargs.pop('hCylinder2', 6.9504))
        MirrorOnTripodWithTwoXStages.__init__(self, *args, **kwargs)
        self.hCylinder = 0

    def _surface_params(self, p, second=False):
        y0 = self._limits_of('limPhysY')[0]
        self._curved(p, _structs.SURF_DUALVFM,
                     (self.r1 - self.hCylinder1, self.r1**2, self.xCylinder1,
                      self.r2 - self.hCylinder2, self.r2**2, self.xCylinder2, y0**2, self.R))

    def select_surface(self, surfaceName):
        """Moves the cylinder of that name into the beam."""
        self.curSurface = self.surface.index(surfaceName)
        axis, self.hCylinder, self.r = (
            (self.xCylinder1, self.hCylinder1, self.r1) if self.curSurface == 0 else
            (self.xCylinder2, self.hCylinder2, self.r2))
        self.dx = -axis
        self.get_surface_limits()
        self.set_x_stages()


class _BentBragg(_Curved):
    """Bent crystal analysers (reference oes/bragg.py:104-343): surface kind
    XRT_HIP_SURF_BENT_BRAGG. *_shape*: 0 circular cylinder, 1 parabolic cylinder, 2 toroid;
    *_planes*: 0 the atomic planes follow the surface (Johann), 1 ground to twice the
    radius (Johansson), 2 bent to radii of their own."""
    _shape = _planes = 0

    def local_n(self, x, y):
        both = self._eval_surface(_SURF_N, x, y)
        return both if self._planes or self.alpha else both[3:]

    def _surface_params(self, p, second=False):
        shape = self._shape
        if shape in (0, 3) and self.crossSection.startswith('parab'):
            shape += 1
        tilt = self.alpha if self.alpha else 0.
        Rs = getattr(self, 'Rs', 0.)
        self._curved(p, _structs.SURF_BENT_BRAGG,
                     (shape, self._planes, self.Rm, Rs, np.cos(tilt), np.sin(tilt),
                      1. if self.alpha else 0., getattr(self, 'RmBragg', self.Rm),
                      getattr(self, 'RsBragg', Rs)))
        p.asymmetric = 1 if self._planes or self.alpha else 0


class JohannCylinder(_BentBragg):
    """Cylindrically bent crystal, meridional radius *Rm*; *crossSection* 'circular' or
    'parabolic' (bragg.py:104-176)."""

    def __init__(self, *args, **kwargs):
        self.Rm = kwargs.pop('Rm', 1000.)
        self.crossSection = kwargs.pop('crossSection', 'circular')
        if not self.crossSection.startswith(('circ', 'parab')):
            raise ValueError('unknown crossSection!')
        self.crossSectionInt = 0 if self.crossSection.startswith('circ') else 1
        OE.__init__(self, *args, **kwargs)


class JohanssonCylinder(JohannCylinder):
    """Bent and ground: atomic planes of radius 2 Rm under a surface of radius Rm
    (bragg.py:179-197)."""
    _planes = 1


class BentLaueCylinder(JohannCylinder):
    """Cylindrically bent crystal in Laue geometry (du Mond): the diffracting planes stand
    across the surface, turned by *alpha*; meridional radius *R* (a number or (p, q) for
    the Coddington radius), *crossSection* 'parabolic' (default) or 'circular' (reference
    oes/laue.py:26-227). The volumetric-diffraction and bent-crystal (TT) amplitude models
    of the reference are not on the GPU path (the material says so)."""
    _planes = 3

    def __init__(self, *args, **kwargs):
        bend = kwargs.pop('R', 1.0e4)
        kwargs.setdefault('crossSection', 'parabolic')
        JohannCylinder.__init__(self, *args, **kwargs)
        self.R = bend

    R = property(lambda self: self._RVal,
                 lambda self, v: setattr(self, '_RVal',
                                         np.inf if v in (None, 0) else
                                         _radius(v, self.get_Rmer_from_Coddington)))
    Rm = property(lambda self: self.R, lambda self, v: None)


class GroundBentLaueCylinder(BentLaueCylinder):
    """Bent and ground Laue crystal (laue.py:455-475)."""
    _planes = 4


class BentLaueSphere(BentLaueCylinder):
    """Spherically (or paraboloidally) bent Laue crystal (laue.py:478-507)."""
    _shape = 3


class BentLaue2D(_BentBragg):
    """Doubly bent Laue crystal, z = x^2 / (2 Rs) + y^2 / (2 Rm): *Rm*, *Rs* of either sign
    (concave, convex or saddle), numbers or (p, q) pairs for the Coddington radii (reference
    oes/laue.py:229-452; its volumetric / TT amplitude models are not on the GPU path)."""
    _shape, _planes = 5, 3

    def __init__(self, *args, **kwargs):
        radii = kwargs.pop('Rm', 1.0e4), kwargs.pop('Rs', -5.0e4)
        self.crossSection = 'parabolic'
        OE.__init__(self, *args, **kwargs)
        self.Rm, self.Rs = radii

    Rm = property(lambda self: self._RmVal,
                  lambda self, v: setattr(self, '_RmVal', np.inf if v in (None, 0) else
                                          _radius(v, self.get_Rmer_from_Coddington)))
    Rs = property(lambda self: self._RsVal,
                  lambda self, v: setattr(self, '_RsVal', np.inf if v in (None, 0) else
                                          _radius(v, self.get_rsag_from_Coddington)))


class JohannToroid(_BentBragg):
    """Doubly bent crystal, meridional *Rm* and sagittal *Rs* (= *Rm* if None)
    (bragg.py:200-269)."""
    _shape = 2

    def __init__(self, *args, **kwargs):
        kwargs = self.pop_kwargs(**kwargs)
        OE.__init__(self, *args, **kwargs)

    def pop_kwargs(self, **kwargs):
        self.Rm = kwargs.pop('Rm', 1000.)
        self.Rs = kwargs.pop('Rs', None)
        return kwargs

    Rs = property(lambda self: self.Rm if self._Rs is None else self._Rs,
                  lambda self, v: setattr(self, '_Rs', v))


class JohanssonToroid(JohannToroid):
    """Doubly bent and ground (bragg.py:272-296)."""
    _planes = 1


class GeneralBraggToroid(JohannToroid):
    """Four radii: *Rm*, *Rs* of the surface, *RmBragg*, *RsBragg* of the atomic planes
    (each following its surface radius if None) (bragg.py:299-343)."""
    _planes = 2

    def pop_kwargs(self, **kwargs):
        planes = kwargs.pop('RmBragg', None), kwargs.pop('RsBragg', None)
        kwargs = JohannToroid.pop_kwargs(self, **kwargs)
        self.RmBragg, self.RsBragg = planes
        return kwargs

    RmBragg = property(lambda self: self.Rm if self._RmBragg is None else self._RmBragg,
                       lambda self, v: setattr(self, '_RmBragg', v))
    RsBragg = property(lambda self: self.Rs if self._RsBragg is None else self._RsBragg,
                       lambda self, v: setattr(self, '_RsBragg', v))


class DicedOE(_BentBragg):
    """Flat element cut into facets *dxFacet* x *dyFacet* with gaps *dxGap*, *dyGap* between
    them; the gaps absorb (reference oes/bragg.py:8-101). Base of the diced analysers."""
    _base = 0          # 0 flat, 2 toroid

    def __init__(self, *args, **kwargs):
        self._dice(kwargs)
        OE.__init__(self, *args, **kwargs)

    def _dice(self, kwargs):
        self.dxFacet, self.dyFacet = kwargs.pop('dxFacet', 2.1), kwargs.pop('dyFacet', 1.4)
        self.dxGap, self.dyGap = kwargs.pop('dxGap', 0.05), kwargs.pop('dyGap', 0.05)
        self.xStep, self.yStep = self.dxFacet + self.dxGap, self.dyFacet + self.dyGap

    def local_n(self, x, y):
        both = self._eval_surface(_SURF_N, x, y)
        return both if self._planes or self.alpha else both[3:]

    def _surface_params(self, p, second=False):
        tilt = self.alpha if self.alpha else 0.
        self._curved(p, _structs.SURF_DICED,
                     (self._base, self._planes, getattr(self, 'Rm', 0.), getattr(self, 'Rs', 0.),
                      np.cos(tilt), np.sin(tilt), 1. if self.alpha else 0., self.xStep,
                      self.yStep, self.dxFacet / 2, self.dyFacet / 2))
        p.asymmetric = 1 if self._planes or self.alpha else 0


class DicedJohannToroid(DicedOE, JohannToroid):
    """Flat facets tangent to a Johann toroid at their centres (bragg.py:345-359)."""
    _base = 2

    def __init__(self, *args, **kwargs):
        self._dice(kwargs)
        JohannToroid.__init__(self, *args, **kwargs)


class DicedJohanssonToroid(DicedJohannToroid, JohanssonToroid):
    """Facets ground to the meridional radius, atomic planes of the Johansson toroid
    (bragg.py:362-375)."""


def _tracked(name):
    """An attribute whose assignment re-derives the zone plate (GeneralFZPin0YZ.reset)."""
    def assign(self, value):
        self.__dict__['_' + name] = value
        self.reset()
    return property(lambda self: self.__dict__['_' + name], assign)


class GeneralFZPin0YZ(OE):
    """Zone plate on a flat element at grazing incidence, its zones set by the two foci
    *f1*, *f2* -- points (x, y, z[, -1 for a negative path]) in the LOCAL frame, or a string
    for 'at infinity' -- at the energy *E*: a ray belongs to zone floor((d1 + d2) / (lambda /
    2) - the lowest such value of the first batch - phaseShift [+ vorticity phi / pi]);
    even zones up to *N* transmit, the others absorb; the local groove density comes from
    the extent of the neighbouring zones in the batch (reference oes/gratings.py:140-313).
    The material must be of kind 'FZP'.

    Those quantities are statistics over all rays, so the pass runs twice: hit points and
    outline states first, then zones and groove vectors on the GPU (torch reductions), then
    the pass again with them per ray. A ray that the intersection search loses is not part
    of the statistics here (the reference counts its last trial point)."""

    f1, f2, E, N = _tracked('f1'), _tracked('f2'), _tracked('E'), _tracked('N')
    phaseShift = _tracked('phaseShift')

    def __init__(self, *args, **kwargs):
        # (assignment order and the repeated reset() are the reference's: a phase shift
        # given to the constructor ends up divided by pi three times, one assigned later by
        # pi once)
        self.f1, self.f2, self.E = kwargs.pop('f1'), kwargs.pop('f2'), kwargs.pop('E')
        self.N = kwargs.pop('N', 1000)
        self.phaseShift = kwargs.pop('phaseShift', 0)
        self.vorticity = kwargs.pop('vorticity', 0)
        angle = kwargs.pop('grazingAngle', None)
        OE.__init__(self, *args, **kwargs)
        self.reset()      # (the reference's base constructor resets the element once, through
        # its gratingDensity assignment; with the reset below that makes pi**3 in all)
        self.use_rays_good_gn = True
        self.grazingAngle = raycing.auto_units_angle(self.pitch if angle is None else angle)
        self.reset()

    def assign_auto_material_kind(self, material):
        material.kind = 'FZP'

    def reset(self):
        if '_E' in self.__dict__ and '_phaseShift' in self.__dict__:
            self.lambdaE = CH / self.E * 1e-7
            self.minHalfLambda = None
            self.set_phase_shift(self.phaseShift)

    def set_phase_shift(self, phaseShift):
        self.__dict__['_phaseShift'] = phaseShift
        if phaseShift:
            self.__dict__['_phaseShift'] = phaseShift / np.pi

    def _is_grating(self):
        return True

    def _path_to(self, focus, x, y, z):
        if isinstance(focus, str):
            return y * np.cos(self.grazingAngle)
        d = ((x - focus[0])**2 + (y - focus[1])**2 + (z - focus[2])**2)**0.5
        return d * focus[3] if len(focus) > 3 else d

    def _zones_between_passes(self, p, beam, lb, gb, _info, _timing):
        dev = _device()
        hit = lb.dev('state', dev) == 1
        x, y, z = (lb.dev(f, dev)[hit] for f in 'xyz')
        half = (self._path_to(self.f1, x, y, z) + self._path_to(self.f2, x, y, z)) / \
            (self.lambdaE / 2)
        if self.minHalfLambda is None:
            self.minHalfLambda = float(half.min()) if half.numel() else 0.
        phi = torch.atan2(y * np.sin(self.grazingAngle), x) / np.pi
        half = half - (self.minHalfLambda + self.phaseShift - phi * self.vorticity)
        N = int(self.N)
        zone = torch.floor(half).to(torch.int64)
        open_ = (torch.remainder(zone, 2) == 0) & (zone < N)
        # extent of every zone in the batch (the reference keeps it for the odd ones, which
        # is all the density of an even zone needs)
        inside = (zone >= 0) & (zone < N)
        extent = [torch.zeros(N, dtype=torch.float64, device=dev).scatter_reduce(
            0, zone[inside], c[inside].abs(), 'amax', include_self=True) for c in (x, y)]
        odd = torch.arange(N, device=dev) % 2 == 1
        zo = zone[open_]
        xo, yo = x[open_], y[open_]
        spans = []
        for ext in extent:
            ext = torch.where(odd, ext, torch.zeros_like(ext))
            span = ext[torch.remainder(zo + 1, N)] - ext[torch.remainder(zo - 1, N)]
            spans.append(torch.where(span == 0, torch.full_like(span, 1e20), span))
        r = torch.sqrt(xo * xo + yo * yo)
        density = (xo * xo / spans[0] + yo * yo / spans[1]) / (r * r)
        n = beam.nrays
        state_ray = torch.ones(n, dtype=torch.int32, device=dev)
        gx = torch.zeros(n, dtype=torch.float64, device=dev)
        gy = torch.zeros(n, dtype=torch.float64, device=dev)
        where = torch.nonzero(hit).ravel()
        state_ray[where[~open_]] = int(self.lostNum)
        gx[where[open_]] = -xo * density / r
        gy[where[open_]] = -yo * density / r
        p.grating, p.grating_axis = 2, -1
        p.grating_order = int(self.order[0] if raycing.is_sequence(self.order) else self.order)
        p.state_ray, p.g_ray_x, p.g_ray_y = state_ray.data_ptr(), gx.data_ptr(), gy.data_ptr()
        lb, gb, report = self._run_pass(p, self.material, True, beam, beam,
                                        want_info=_info is not None,
                                        timing=_timing is not None, out=(lb, gb))
        torch.cuda.current_stream().synchronize()
        return lb, gb, report, (state_ray, gx, gy)


class BlazedGrating(_Curved):
    """Saw-tooth grating of constant line density for WAVE propagation: the
    diffraction comes from the surface itself through the Kirchhoff integral,
    the material is a mirror (oes/gratings.py:316-535). Facet geometry, the
    first-facet intersection and the facet normals run in the reflect kernels
    (surface kind XRT_HIP_SURF_BLAZED)."""

    def __init__(self, *args, **kwargs):
        self.blaze = raycing.auto_units_angle(kwargs.pop('blaze'))
        self.antiblaze = raycing.auto_units_angle(kwargs.pop('antiblaze', np.pi*0.4999))
        self.rho0 = kwargs.pop('rho', 1)
        if kwargs.get('gratingDensity') is not None:
            # the reference cannot run this case either: its rho0 setter and reset() call
            # each other without end (gratings.py:378-380, 418-420), so there is nothing
            # to pin a variable-density saw-tooth against
            raise NotImplementedError('variable line density of a blazed grating')
        OE.__init__(self, *args, **kwargs)
        self.gratingDensity = None
        self.reset()

    def reset(self):
        """Groove period and the trigonometry of the two facets."""
        self.rho_1 = 1. / self.rho0
        for facet, angle in (('Blaze', self.blaze), ('Antiblaze', self.antiblaze)):
            for fn in (np.sin, np.cos, np.tan):
                setattr(self, fn.__name__ + facet, fn(angle))

    def get_grating_area_fraction(self):
        """The part of a groove the beam sees at the grating's pitch: length of the lit
        stretch of the blaze facet per period (gratings.py:524-535)."""
        grazing = np.tan(abs(self.pitch))
        y_lit = self.rho_1 * self.tanBlaze / (self.tanBlaze + grazing)
        z_lit = -y_lit * grazing
        return ((self.rho_1-y_lit)**2 + (0-z_lit)**2)**0.5 * self.rho0

    def _surface_params(self, p, second=False):
        self._curved(p, _structs.SURF_BLAZED,
                     (self.rho_1, self.tanBlaze, self.tanAntiblaze, self.sinBlaze,
                      self.cosBlaze, self.sinAntiblaze, self.cosAntiblaze,
                      1 + self.tanAntiblaze/self.tanBlaze,
                      float(self.blaze == np.pi/2), float(self.antiblaze == np.pi/2)))


def _rederive(name):
    """Property whose assignment re-derives the conic (its axis depends on the angles)."""
    def store(self, value):
        setattr(self, '_' + name, value)
        self._reset_pq()
    return property(lambda self: getattr(self, '_' + name), store)


class EllipticalMirrorParam(OE):
    """Elliptical mirror -- ellipsoid of revolution, or elliptical cylinder with
    *isCylindrical* -- between the foci at distances *p* and *q* (the p arm along the
    global y axis), in the parametric coordinates (s, phi, r) of oes/parametric.py:
    s along the major axis, r the distance from it. The root solve runs on
    r - local_r(s, phi) in the reflect kernels (surface kind XRT_HIP_SURF_ELLIPSE_PARAM).
    *f1* / *f2* / *pAxis* are not mirrored."""
    conic = 0
    _NEEDS = ('_p', '_q', '_pitchVal', '_rollVal', '_yawVal', '_positionRollVal',
              'rotationSequence')

    def __init__(self, *args, **kwargs):
        for unsupported in ('f1', 'f2', 'pAxis'):
            if kwargs.pop(unsupported, None) is not None:
                raise NotImplementedError('%s(%s=...)' % (type(self).__name__, unsupported))
        arms = kwargs.pop('p', 1000), kwargs.pop('q', 1000)
        self.isCylindrical = kwargs.pop('isCylindrical', False)
        self.isClosed = kwargs.pop('isClosed', False)
        OE.__init__(self, *args, **kwargs)
        self.isParametric = True
        self._p, self._q = arms
        self._reset_pq()

    p = _rederive('p')
    q = _rederive('q')
    pitch = _rederive('pitchVal')
    roll = _rederive('rollVal')
    yaw = _rederive('yawVal')
    positionRoll = _rederive('positionRollVal')

    def _grazing_angle(self):
        """Angle between the surface at its pole and the global y axis (the incoming arm):
        |asin| of the y component of the pole's normal in the global frame. None while the
        element is still being constructed."""
        if self.bl is None or not all(hasattr(self, v) for v in self._NEEDS):
            return None
        normal = [np.zeros(1), np.zeros(1), np.ones(1)]
        raycing.turn(normal, raycing.rotation_steps(
            '-' + self.rotationSequence, self.pitch, self.roll + self.positionRoll, self.yaw))
        if self.bl.sinAzimuth != 0:
            raycing.turn(normal, [(2, self.bl.cosAzimuth, -self.bl.sinAzimuth)])
        return abs(np.arcsin(normal[1][0]))

    def _place(self, gamma, y0, z0):
        """Axis of the conic in the local frame: tilted by *gamma*, centre at (y0, z0)."""
        self.cosGamma, self.sinGamma, self.y0, self.z0 = np.cos(gamma), np.sin(gamma), y0, z0

    def _reset_pq(self):
        theta = self._grazing_angle()
        if theta is None or not (self.p and self.q):
            return
        half_sum, half_diff = (self.q + self.p)/2., (self.q - self.p)/2.
        self._place(np.arctan2((self.p - self.q) * np.sin(theta),
                               (self.p + self.q) * np.cos(theta)),
                    half_diff * np.cos(theta), half_sum * np.sin(theta))
        self.ellipseA = half_sum
        self.ellipseB = np.sqrt(self.q * self.p) * np.sin(theta)

    def _conic_ab(self):
        return self.ellipseA, self.ellipseB

    def _surface_params(self, p, second=False):
        semi = self._conic_ab()
        self._curved(p, _structs.SURF_ELLIPSE_PARAM,
                     (self.y0, self.z0, self.cosGamma, self.sinGamma, semi[0], semi[1],
                      float(bool(self.isCylindrical)), float(bool(self.isClosed)),
                      float(self.conic)))

    # the parametric surface functions, on the GPU
    def xyz_to_param(self, x, y, z):
        return tuple(self._eval_surface(_SURF_TO_PARAM, x, y, z))

    def param_to_xyz(self, s, phi, r):
        return tuple(self._eval_surface(_SURF_FROM_PARAM, s, phi, r))

    def local_r(self, s, phi):
        return self._eval_surface(_SURF_R, s, phi)[0]

    def local_n(self, s, phi):
        return self._eval_surface(_SURF_N, s, phi)[3:]


EllipticalMirror = EllipticalMirrorParam


class ParabolicalMirrorParam(EllipticalMirrorParam):
    """Paraboloid (or parabolic cylinder): collimates a source at distance *p*
    or focuses a parallel beam at *q* -- exactly one of them is given
    (oes/parametric.py:252-474)."""
    conic = 1

    def __init__(self, *args, **kwargs):
        if kwargs.pop('parabolaAxis', None) is not None:
            raise NotImplementedError('ParabolicalMirrorParam(parabolaAxis=...)')
        kwargs.setdefault('p', 10000)
        kwargs.setdefault('q', None)
        EllipticalMirrorParam.__init__(self, *args, **kwargs)

    def _reset_pq(self):
        theta = self._grazing_angle()
        if theta is None:
            return
        if (self.p is None) == (self.q is None):
            raise ValueError('One and only one of p or q must be None!')
        if self.p is None:      # focusing: the focus lies downstream
            focal, side = self.q, 1.
        else:                   # collimating: the focus is the source
            focal, side = self.p, -1.
        self._place(side * theta, side * focal * np.cos(theta), focal * np.sin(theta))
        self.parabParam = -side * focal * np.sin(theta)**2

    def _conic_ab(self):
        return self.parabParam, 0.


ParabolicMirror = ParabolicalMirrorParam


class HyperbolicMirrorParam(EllipticalMirrorParam):
    """Hyperboloid (or hyperbolic cylinder) between the foci at *p* and *q*; the
    OUTER surface reflects (``invertNormal = -1``), oes/parametric.py:477-716."""
    conic = 2

    def __init__(self, *args, **kwargs):
        EllipticalMirrorParam.__init__(self, *args, **kwargs)
        self.invertNormal = -1

    def _reset_pq(self):
        theta = self._grazing_angle()
        if theta is None or not (self.p and self.q):
            return
        self._place(np.arctan2((self.p + self.q) * np.sin(theta),
                               (self.p - self.q) * np.cos(theta)),
                    -(self.p + self.q)/2. * np.cos(theta), (self.p - self.q)/2. * np.sin(theta))
        self.hyperbolaA = abs(self.p - self.q)/2.
        self.hyperbolaB = np.sqrt(self.p*self.q) * np.sin(theta)

    def _conic_ab(self):
        return self.hyperbolaA, self.hyperbolaB


HyperbolicMirror = HyperbolicMirrorParam


class SurfaceOfRevolution(EllipticalMirrorParam):
    """Closed surfaces about the local y axis -- capillaries -- in cylindrical coordinates:
    s = y along the axis, (phi, r) polar across it (reference oes/parametric.py:717-731).
    The kernels' conic kind with the axis untilted and the full turn open."""
    isClosed, isCylindrical = True, False
    cosGamma, sinGamma, y0, z0 = 1., 0., 0., 0.
    _ctd = 0.

    def __init__(self, *args, **kwargs):
        OE.__init__(self, *args, **kwargs)
        self.isParametric = True

    def _reset_pq(self):         # nothing to re-derive when the element is turned
        pass

    def _surface_params(self, p, second=False):
        semi = self._conic_ab()
        self._curved(p, _structs.SURF_ELLIPSE_PARAM,
                     (0., 0., 1., 0., semi[0], semi[1], 0., 1., float(self.conic),
                      float(self._ctd)))


class ParaboloidCapillaryMirror(SurfaceOfRevolution):
    """Paraboloid of revolution focusing at the distance *q* behind its centre, radius
    *r0* at the centre (yaw = 180 deg collimates) (oes/parametric.py:733-788)."""
    conic = 3

    def __init__(self, *args, **kwargs):
        self.q, self.r0 = kwargs.pop('q', 500.), kwargs.pop('r0', 2.5)
        SurfaceOfRevolution.__init__(self, *args, **kwargs)

    @property
    def focus(self):
        return -0.5*(self.q-(self.q**2+self.r0**2)**0.5)

    @property
    def s0(self):
        return self.focus + self.q

    def _conic_ab(self):
        return self.s0, self.focus


class EllipsoidCapillaryMirror(SurfaceOfRevolution):
    """Ellipsoid of revolution, the inside reflecting: semi-axes *ellipseA*, *ellipseB*
    (not the tube radius), *workingDistance* from the end face to the focus; the centre is
    the middle of the tube, whose length is limPhysY (oes/parametric.py:791-889)."""
    conic = 0

    def __init__(self, *args, **kwargs):
        self.ellipseA = kwargs.pop('ellipseA', 10000)
        self.ellipseB = kwargs.pop('ellipseB', 2.5)
        self.workingDistance = kwargs.pop('workingDistance', 17.)
        SurfaceOfRevolution.__init__(self, *args, **kwargs)

    def _half_length(self):
        ends = self.limPhysY
        return 0.5*np.abs(ends[-1]-ends[0])

    @property
    def ctd(self):
        """Centre of the tube measured from the centre of the ellipse."""
        c = (self.ellipseA**2 - self.ellipseB**2)**0.5
        return c - self.workingDistance - self._half_length()

    _ctd = property(lambda self: self.ctd)

    def _conic_ab(self):
        return self.ellipseA, self.ellipseB


class HyperboloidCapillaryMirror(EllipsoidCapillaryMirror):
    """Hyperboloid of revolution (mirror lens), the OUTSIDE reflecting: *hyperbolaA*,
    *hyperbolaB*, *workingDistance* from the virtual focus to the front face
    (oes/parametric.py:892-988)."""
    conic = 2

    def __init__(self, *args, **kwargs):
        self.hyperbolaA = kwargs.pop('hyperbolaA', 10000)
        self.hyperbolaB = kwargs.pop('hyperbolaB', 2.5)
        self.workingDistance = kwargs.pop('workingDistance', 17.)
        self.invertNormal = -1
        SurfaceOfRevolution.__init__(self, *args, **kwargs)

    @property
    def ctd(self):
        c = (self.hyperbolaA**2 + self.hyperbolaB**2)**0.5
        return c + self.workingDistance + self._half_length()

    def _conic_ab(self):
        return self.hyperbolaA, self.hyperbolaB


class DCM(OE):
    """Double-crystal monochromator with flat crystals (oes/dcm.py): the second
    crystal is described relative to the first -- extra roll / pitch, the translations
    along and across the first crystal's surface, its own limits and material."""
    _SECOND = dict(bragg=0, cryst1roll=0, cryst2roll=0, cryst2pitch=0, cryst2finePitch=0,
                   cryst2perpTransl=0, cryst2longTransl=0, limPhysX2=[-_WIDE, _WIDE],
                   limPhysY2=[-_WIDE, _WIDE], limOptX2=None, limOptY2=None, material2=None)

    def __init__(self, *args, **kwargs):
        for key, default in self._SECOND.items():
            setattr(self, key, kwargs.pop(key, list(default) if isinstance(default, list)
                                          else default))
        offset = kwargs.pop('fixedOffset', None)
        OE.__init__(self, *args, **kwargs)
        if offset not in [0, None]:       # exit beam parallel, this far above the incoming one
            self.cryst2perpTransl = offset/2./np.cos(self.bragg)

    def local_n1(self, x, y):
        return self.local_n(x, y)

    def local_n2(self, x, y):
        both = self._flat_normals(second=True)
        return both if self.alpha else both[:3]

    def double_reflect(self, beam=None, needLocal=True, fromVacuum1=True,
                       fromVacuum2=True, returnLocalAbsorbed=None, _timing=None, out=None):
        """-> (beamGlobal, beamLocal1, beamLocal2), dcm.py:248-354. *out* (extension, as in
        ``OE.reflect``): the triple an earlier call on a beam of the same size returned, to be
        overwritten in place -- a loop over the same beamline then allocates nothing (giving
        three 1-GB beams back to the allocator and taking three new ones every step cost the
        cfg3 loop 20-90 us of idle GPU per pass)."""
        first = self._own_angles(False)
        second = self._own_angles(True)
        p1 = self._make_pass(*first[:4], fromVacuum=fromVacuum1, out_to_global=False)
        p2 = self._make_pass(*second, fromVacuum=fromVacuum2, is2ndXtal=True,
                             in_is_global=False, good_mode=1, out_to_global=True,
                             zero_local_not_entering=True,
                             force_lost_out=hasattr(self, 't'))

        def both(beam, local=True):
            # (XRT_HIP_DCM_TWO_PASSES=1: the two separate passes, for comparison)
            if os.environ.get('XRT_HIP_DCM_TWO_PASSES', '') != '1':
                fused = self._run_double(p1, p2, fromVacuum1, fromVacuum2, beam, _timing, out,
                                         local=local)
                if fused is not None:
                    return fused
            lo1, between, _ = self._run_pass(p1, self.material, fromVacuum1, beam, beam,
                                             local=local)
            lo2, gb2, _ = self._run_pass(p2, self.material2, fromVacuum2, between, beam,
                                         local=local)
            return gb2, lo1, lo2
        if out is None and _timing is None and \
                _locals_on_demand(self, self.material, self.material2):
            if os.environ.get('XRT_HIP_DCM_TWO_PASSES', '') != '1' and \
                    self._pair_is_fusable(p1, p2, fromVacuum1, fromVacuum2):
                # nothing is launched yet: apertures and a screen that take the global beam
                # ride in the tail of the pair's kernel (_DeferredDouble)
                return _DeferredDouble(self, p1, p2, (fromVacuum1, fromVacuum2),
                                       beam).hand_out(always_tuple=True)
            # the global beam now, the beams on the two surfaces when somebody looks at them
            later = _LocalsOnDemand(self, beam, 2, lambda was: both(was)[1:])
            return (both(beam, local=False)[0],) + later.hand_out(always_tuple=True)
        return both(beam)

    def _pair_is_fusable(self, p1, p2, fromVacuum1, fromVacuum2):
        _lib.require_gpu()
        dev = _device()
        m1 = self._material_struct(self.material, fromVacuum1, dev)
        m2 = self._material_struct(self.material2, fromVacuum2, dev)
        return bool(_lib.load().xrt_hip_double_reflect_fusable(
            ctypes.byref(p1), ctypes.byref(m1), ctypes.byref(p2), ctypes.byref(m2)))

    def _run_double_tail(self, p1, p2, materials, from_vacuum, beam, local, screen_rec, apertures,
                         keep_global):
        """The pair's fused pass, optionally with apertures and a screen in its tail
        (xrt_hip_double_reflect_tail_f64_dev) -> (gb2, lo1, lo2, image, fused): fused = the
        kernel carried the screen; with *keep_global* False gb2 then holds nothing."""
        lib = _lib.load()
        dev = _device()
        m1 = self._material_struct(materials[0], from_vacuum[0], dev)
        m2 = self._material_struct(materials[1], from_vacuum[1], dev)
        n, amp = beam.nrays, beam.has_amplitudes()
        screened = screen_rec is not None
        keep_global = keep_global or not screened
        gb2 = rs.Beam.empty_on_device(n, dev, amp) if keep_global else \
            _scratch_beam('global', n, dev, amp)
        image = rs.Beam.empty_on_device(n, dev, amp) if screened else None
        lo1 = lo2 = None
        angles = (None, None)
        if local:
            lo1, lo2 = (rs.Beam.empty_on_device(n, dev, amp) for _ in range(2))
            angles = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(2)]
        ws = hipcalls.workspace(dev, lib.xrt_hip_reflect_workspace_bytes(n), 'reflect')
        common = (ctypes.byref(p1), ctypes.byref(m1), ctypes.byref(p2), ctypes.byref(m2),
                  ctypes.byref(beam.to_struct(dev)),
                  ctypes.byref(lo1.to_struct(dev)) if local else None,
                  ctypes.byref(lo2.to_struct(dev)) if local else None,
                  ctypes.byref(gb2.to_struct(dev)),
                  ctypes.c_void_p(angles[0].data_ptr()) if local else None,
                  ctypes.c_void_p(angles[1].data_ptr()) if local else None)
        fused = ctypes.c_int(0)
        if screened or apertures:
            tail = _structs.Tail()
            tail.n_apertures, tail.keep_screen = len(apertures), 1
            for k, rec in enumerate(apertures):
                tail.aperture[k] = rec
            if screened:
                tail.screen = ctypes.addressof(screen_rec)
                tail.out_screen = ctypes.addressof(image.to_struct(dev))
            _lib.check(lib.xrt_hip_double_reflect_tail_f64_dev(
                *common, ctypes.byref(tail), int(keep_global), ctypes.c_void_p(ws.data_ptr()),
                ws.numel(), _stream(), ctypes.byref(fused)),
                'xrt_hip_double_reflect_tail_f64_dev')
        else:
            _lib.check(lib.xrt_hip_double_reflect_f64_dev(
                *common, ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream(), None),
                'xrt_hip_double_reflect_f64_dev')
        if local:
            lo1._d['theta'], lo2._d['theta'] = angles
        if not (fused.value & 1) and not keep_global:
            _scratch_beam.taken('global', gb2)      # (it holds the global beam after all)
        self._adopt((lo1, lo2, gb2) if local else (gb2,), beam)
        if image is not None:
            rs.inherit_scalars(image, beam)
        return gb2, lo1, lo2, image, bool(fused.value & 1)

    def _run_double(self, p1, p2, fromVacuum1, fromVacuum2, beam, timing=None, out=None,
                    local=True):
        """Both crystals in one pass over the beam (xrt_hip_double_reflect_f64_dev) when
        the pair qualifies (flat Bragg crystals), else None. -> (gb2, lo1, lo2); *local*
        False: the global beam alone (lo1 = lo2 = None), 200 instead of 416 B per ray."""
        _lib.require_gpu()
        lib = _lib.load()
        dev = _device()
        m1 = self._material_struct(self.material, fromVacuum1, dev)
        m2 = self._material_struct(self.material2, fromVacuum2, dev)
        if not lib.xrt_hip_double_reflect_fusable(ctypes.byref(p1), ctypes.byref(m1),
                                                  ctypes.byref(p2), ctypes.byref(m2)):
            return None
        n = beam.nrays
        usable = out is not None and len(out) == 3 and all(
            b is not None and b is not beam and b.nrays == n and not b._h_dirty() and
            b.has_amplitudes() == beam.has_amplitudes() for b in out) and \
            all('theta' in b._d for b in out[1:])
        if not local:
            lo1 = lo2 = None
            gb2 = rs.Beam.empty_like_on_device(beam, dev)
        elif usable:
            gb2, lo1, lo2 = out
            for b in out:               # (overwritten in place: their readers first)
                if type(b) is not rs.LazyBeam or b.__dict__['_filled']:
                    rs.flush_pending(b)
            angles = [lo1._d['theta'], lo2._d['theta']]
        else:
            lo1, lo2, gb2 = (rs.Beam.empty_like_on_device(beam, dev) for _ in range(3))
            angles = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(2)]
        ws = hipcalls.workspace(dev, lib.xrt_hip_reflect_workspace_bytes(n), 'reflect')
        ms = (ctypes.c_float * 3)() if timing is not None else None
        _lib.check(lib.xrt_hip_double_reflect_f64_dev(
            ctypes.byref(p1), ctypes.byref(m1), ctypes.byref(p2), ctypes.byref(m2),
            ctypes.byref(beam.to_struct(dev)),
            ctypes.byref(lo1.to_struct(dev)) if local else None,
            ctypes.byref(lo2.to_struct(dev)) if local else None, ctypes.byref(gb2.to_struct(dev)),
            ctypes.c_void_p(angles[0].data_ptr()) if local else None,
            ctypes.c_void_p(angles[1].data_ptr()) if local else None,
            ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream(), ms),
            'xrt_hip_double_reflect_f64_dev')
        if timing is not None:
            timing.update(pass_ms=ms[0], kernel_ms=ms[1], exact_sequence=bool(ms[2]))
        if local:
            lo1._d['theta'], lo2._d['theta'] = angles
        self._adopt((lo1, lo2, gb2) if local else (gb2,), beam)
        return gb2, lo1, lo2


class DCMOnTripodWithOneXStage(DCM, rst.Tripod, rst.OneXStage):
    """A double-crystal monochromator on three jacks and one x stage
    (oes/__init__.py:669-707)."""

    def __init__(self, *args, **kwargs):
        kwargs, jacks = rst.Tripod.pop_kwargs(self, **kwargs)
        kwargs, stage = rst.OneXStage.pop_kwargs(self, **kwargs)
        DCM.__init__(self, *args, **kwargs)
        rst.Tripod.__init__(self, *jacks)
        rst.OneXStage.__init__(self, *stage)
        stripes = len(self.surface) if self.surface is not None else 0
        for optical in (self.limOptX2, self.limOptY2):
            if optical is None:
                continue
            if not (raycing.is_sequence(optical[0]) and raycing.is_sequence(optical[1])):
                raise ValueError('optical limits of the second crystal: (lows, highs), each a '
                                 'sequence per stripe')
            if not (len(optical[0]) == len(optical[1]) == stripes):
                raise ValueError('one optical limit of the second crystal per stripe, please')
        for edge in (self.limPhysX2[0], self.limPhysX2[1], self.limPhysY2[0],
                     self.limPhysY2[1]):
            if raycing.is_sequence(edge) and len(edge) != stripes:
                raise ValueError('one physical limit of the second crystal per stripe, please')

    def get_orientation(self):
        rst.Tripod.get_orientation(self)


class DCMwithSagittalFocusing(DCM):
    """Double-crystal monochromator whose second crystal is bent sagittally to the radius
    *Rs* (a cylinder along the beam), no miscut (reference oes/__init__.py:639-664)."""

    def __init__(self, *args, **kwargs):
        self.Rs = kwargs.pop('Rs', 1e12)
        DCM.__init__(self, *args, **kwargs)

    def local_z2(self, x, y):
        return self._eval_second(_SURF_Z, x, y)[0]

    def local_n2(self, x, y):
        return self._eval_second(_SURF_N, x, y)[3:]

    def _eval_second(self, what, x, y):
        keep = self._surface_params
        self._surface_params = lambda p, second=False: keep(p, True)
        try:
            return self._eval_surface(what, x, y)
        finally:
            del self._surface_params

    def _surface_params(self, p, second=False):
        if not second:
            return DCM._surface_params(self, p, second)
        if self.alpha:
            raise NotImplementedError('a miscut on the bent crystal')
        self._curved(p, _structs.SURF_SAGITTAL, (self.Rs, self.Rs**2))


class LauePlate(OE):
    """Flat crystal plate in Laue geometry: the diffracting planes stand on the surface
    (turned from its normal by 90 deg + *alpha*); the thickness belongs to the material
    (reference oes/laue.py:11-23)."""

    def _flat_normals(self, second=False):
        planes = [0., self.cosalpha, -self.sinalpha] if self.alpha else [0., 1., 0.]
        return planes + [0., 0., 1.]

    def local_n(self, x, y):
        return self._flat_normals()

    def _surface_params(self, p, second=False):
        OE._surface_params(self, p, second)
        p.asymmetric = 1          # the two normals always differ


class Plate(DCM):
    """A body with two flat surfaces (window, filter): the 2nd 'crystal' of the
    DCM skeleton is the back face at -t (oes/refractive.py:11-235)."""

    def __init__(self, *args, **kwargs):
        thickness = kwargs.pop('t', 0)
        wedge = kwargs.pop('wedgeAngle', 0)
        kwargs.setdefault('overEdge', '')
        DCM.__init__(self, *args, **kwargs)
        self.t, self.wedgeAngle = thickness, wedge
        self.cryst2perpTransl, self.cryst2pitch = -thickness, wedge
        # As constructed, the reference's Plate ends up with the back-face limits
        # EQUAL to the front-face ones: its __init__ mirrors x only `if
        # isinstance(self.limPhysX, (list, tuple))`, but the property returns a
        # Limits array, so the else branch runs (refractive.py:37-46). Reproduced,
        # not "fixed" (golden case g2_plate_be has asymmetric x limits).
        self.limPhysX2, self.limPhysY2 = list(self.limPhysX), list(self.limPhysY)
        self.limOptX2, self.limOptY2 = self.limOptX, self.limOptY
        self.material2 = self.material
        if getattr(self.material, 'kind', None) == 'auto':
            self.material.kind = 'plate'

    def double_refract(self, beam=None, needLocal=True, returnLocalAbsorbed=None):
        """-> (beamGlobal, beamLocal1, beamLocal2), refractive.py:171-235."""
        return self.double_reflect(beam=beam, needLocal=needLocal,
                                   fromVacuum1=True, fromVacuum2=False)


class ParaboloidFlatLens(Plate):
    """Refractive lens, or a stack of *nCRL* of them (compound refractive lens): the
    surface is z = (x^2 + y^2) / (4 focus), cut off at *zmax* (a plate of thickness
    zmax + t with a paraboloid hole). *focus* is the focal length of the PARABOLA, a
    shape parameter; either *focus* or *nCRL* may be a pair (focal distance, E) to be
    derived from the other one and the material (reference oes/refractive.py:237-550).
    As in the reference, the back surface has the shape of the front one in all four
    lens classes (its local_z2 returns local_z); they differ in the lenslet spacing of
    ``multiple_refract`` and in the factor between focus and nCRL."""
    _double_sided = False
    _cylinder = False

    def __init__(self, *args, **kwargs):
        focus, count = kwargs.pop('focus', 1.), kwargs.pop('nCRL', 1)
        self.zmax = kwargs.pop('zmax', None)
        kwargs.setdefault('pitch', np.pi/2)
        Plate.__init__(self, *args, **kwargs)
        if getattr(self.material, 'kind', None) in ('auto', 'plate'):
            self.material.kind = 'lens'
        if raycing.is_sequence(focus) and raycing.is_sequence(count):
            print("'focus' and 'nCRL' cannot be both automatic")
            count = 1
        # the given one first, the derived one after it
        for name, value in sorted((('focus', focus), ('nCRL', count)),
                                  key=lambda item: raycing.is_sequence(item[1])):
            setattr(self, name, value)

    @property
    def nCRL(self):
        return self._nCRL

    @nCRL.setter
    def nCRL(self, value):
        if raycing.is_sequence(value):
            exact = self.get_nCRL(*value)
            self._nCRL = max(int(round(exact)), 1)
        else:
            self._nCRL = max(int(round(value)), 1)

    @property
    def focus(self):
        return self._focus

    @focus.setter
    def focus(self, value):
        self._focus = self.get_focus(*value) if raycing.is_sequence(value) else value

    def _lens_power(self, E):
        """(1 - Re n) of the material times the number of curved faces per lenslet."""
        faces = 1. if self._double_sided else 2.
        index = np.ravel(self.material.get_refractive_index(E))[0]
        return 1. - float(index.real), faces

    def get_nCRL(self, f, E):
        decrement, faces = self._lens_power(E)
        return self.focus / (f*decrement) * faces

    def get_focus(self, f, E):
        decrement, faces = self._lens_power(E)
        return (f*decrement) * self.nCRL / faces

    def _surface_params(self, p, second=False):
        self._curved(p, _structs.SURF_PARABOLOID,
                     (4 * self.focus, 2 * self.focus,
                      0. if self.zmax is None else self.zmax,
                      0. if self.zmax is None else 1., 1. if self._cylinder else 0.))

    def local_z1(self, x, y):
        return self._eval_surface(_SURF_Z, x, y)[0]

    def local_n1(self, x, y):
        return self._eval_surface(_SURF_N, x, y)[3:]

    local_z = local_z2 = local_z1
    local_n = local_n2 = local_n1

    def multiple_refract(self, beam=None, needLocal=True, returnLocalAbsorbed=None):
        """The beam through all *nCRL* lenslets -> (global beam behind the last one, the
        two local beams of the FIRST one). Between lenslets the centre walks by one
        lenslet spacing against the rotated local z (only when *zmax* is given: the
        reference does not move an unbounded paraboloid); the element is back at its own
        centre afterwards and *centerShift* holds the last step."""
        if self.nCRL == 1:
            self.centerShift = np.zeros(3)
            return self.double_refract(beam=beam, needLocal=needLocal)
        depth = 5 if self.zmax is None else self.zmax
        spacing = (2.*depth if self._double_sided else depth) + self.t
        axis = [0, -spacing, 0]
        home = list(self.center)
        self.center = list(home)
        try:
            current, first = beam, None
            for _ in range(self.nCRL):
                gb, lo1, lo2 = self.double_refract(beam=current, needLocal=needLocal)
                if self.zmax is not None:
                    axis = raycing.rotate_point([0, 0, 1], self.rotationSequence, self.pitch,
                                                self.roll + self.positionRoll, self.yaw)
                    for k in range(3):
                        self.center[k] -= spacing * axis[k]
                current = gb
                first = first or (lo1, lo2)
        finally:
            self.center = home
        self.centerShift = spacing * np.array(axis)
        return (gb,) + first


class ParabolicCylinderFlatLens(ParaboloidFlatLens):
    """Lens(es) focusing in one direction: flat along the local x, parabolic along y."""
    _cylinder = True


class DoubleParaboloidLens(ParaboloidFlatLens):
    """Lens(es) with two paraboloid faces."""
    _double_sided = True


class DoubleParabolicCylinderLens(ParabolicCylinderFlatLens):
    """Lens(es) with two parabolic-cylinder faces."""
    _double_sided = True


def _zone_reset(name):
    """A zone-plate parameter: storing it rebuilds the zone table."""
    def store(self, value):
        setattr(self, '_' + name, value)
        self.reset()
    return property(lambda self: getattr(self, '_' + name), store)


class NormalFZP(OE):
    """Circular Fresnel zone plate in the local (x, y) plane, the optical axis along the
    local z (X-Ray Data Booklet 4.4): zones of zero thickness, alternately opaque and
    transparent; the material must be of kind 'FZP'. *f* [mm] is the focal length at the
    energy *E* [eV], *N* the number of zones, given or derived from the width of the
    *thinnestZone* [mm]; *isCentralZoneBlack* False inverts the zones; *order* is one
    diffraction order or a sequence to draw from (reference oes/gratings.py:10-137)."""
    f, E, N, thinnestZone = (_zone_reset(k) for k in ('f', 'E', 'N', 'thinnestZone'))

    def __init__(self, *args, **kwargs):
        self.isCentralZoneBlack = kwargs.pop('isCentralZoneBlack', True)
        given = {k: kwargs.pop(k, default) for k, default in
                 (('f', 50), ('E', 1000), ('N', 1000), ('thinnestZone', None))}
        for key, value in given.items():
            setattr(self, '_' + key, value)
        self.reset()
        kwargs['limPhysX'] = kwargs['limPhysY'] = [-self.rn[-1], self.rn[-1]]
        OE.__init__(self, *args, **kwargs)
        if getattr(self.material, 'kind', None) == 'auto':
            self.material.kind = 'FZP'

    def reset(self):
        """r_n = sqrt(n f lambda + (n lambda / 2)^2), n = 0..N; the outline follows."""
        if not all(hasattr(self, '_' + k) for k in ('f', 'E', 'N', 'thinnestZone')):
            return
        wavelength = CH / self.E * 1e-7
        if self.thinnestZone is not None:
            self._N = wavelength * self.f / 4. / self.thinnestZone**2
        self.zones = np.arange(self.N + 1)
        self.rn = np.sqrt(self.zones*self.f*wavelength + 0.25*(self.zones*wavelength)**2)
        self.limPhysX = self.limPhysY = [-self.rn[-1], self.rn[-1]]

    def rays_good_gn(self, x, y, z=None):
        """-> (state, (gx, gy, gz) of the rays with state 1): rays in opaque zones or
        beyond the last zone are lost (the state comes from the kernels' own function)."""
        state = self.rays_good(x, y)
        hit = state == 1
        xs, ys = np.asarray(x, dtype=float)[hit], np.asarray(y, dtype=float)[hit]
        radius = np.sqrt(xs**2 + ys**2)
        zone = np.searchsorted(self.rn, radius, side='right') - 1
        density = 1. / (self.rn[np.minimum(zone + 1, len(self.rn) - 1)] -
                        np.where(zone >= 1, self.rn[np.maximum(zone - 1, 0)], 0.))
        return state, (-xs / radius * density, -ys / radius * density, np.zeros_like(xs))
