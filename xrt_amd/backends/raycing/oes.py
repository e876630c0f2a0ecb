"""Optical elements on the accelerated path — host-side mirror of
xrt/backends/raycing/oes (OE: oes/base.py:61-330 + oes/reflect.py:18-163,
ToroidMirror: oes/__init__.py:321-411, DCM: oes/dcm.py:12-354).

The objects hold geometry and material; ``reflect`` / ``double_reflect`` build
the parameter block of one ``_reflect_local`` pass (include/xrt_hip.h
``xrt_hip_pass``) and launch the fused HIP kernels on device-resident beams.
Same call signatures and return values (beamGlobal, beamLocal[, beamLocal2])
as the reference. Outside the accelerated subset (figure error, gratings,
parametric surfaces, mosaic/bent crystals, polygon shapes) a
NotImplementedError is raised — there is no CPU fallback.
"""
import ctypes
import os

import numpy as np
import torch

from .. import raycing
from ... import _lib, _structs, hipcalls
from . import sources as rs


def _limits(lim, default_half=raycing.maxHalfSizeOfOE):
    if lim is None:
        return None
    return [float(lim[0]), float(lim[1])]


class OE(object):
    """Generic flat optical element (mirror, crystal, plate surface)."""

    surf_kind = _structs.SURF_FLAT

    def __init__(self, bl=None, name='', center=[0, 0, 0], pitch=0, roll=0, yaw=0,
                 positionRoll=0, rotationSequence='RzRyRx', extraPitch=0,
                 extraRoll=0, extraYaw=0, extraRotationSequence='RzRyRx',
                 alarmLevel=None, surface=None, material=None, figureError=None,
                 alpha=None,
                 limPhysX=[-raycing.maxHalfSizeOfOE, raycing.maxHalfSizeOfOE],
                 limOptX=None,
                 limPhysY=[-raycing.maxHalfSizeOfOE, raycing.maxHalfSizeOfOE],
                 limOptY=None, isParametric=False, shape='rect',
                 gratingDensity=None, order=None, **kwargs):
        if figureError is not None or isParametric:
            raise NotImplementedError('figure error / user-defined parametric OEs '
                                      'are outside the accelerated path')
        if not isinstance(shape, str):
            raise NotImplementedError('polygon-shaped OEs are outside the '
                                      'accelerated path')
        self.bl = bl
        if bl is not None:
            if self not in bl.oes:
                bl.oes.append(self)
                self.ordinalNum = len(bl.oes)
                self.lostNum = -self.ordinalNum
        else:
            self.ordinalNum = 1
            self.lostNum = -1
        self.name = name or '{0}{1}'.format(type(self).__name__, self.ordinalNum)
        self.uuid = kwargs.get('uuid', raycing.new_uuid())
        if bl is not None:
            bl.oesDict[self.uuid] = [self, 1]
        self.center = center
        self.pitch = pitch
        self.roll = roll
        self.yaw = yaw
        self.rotationSequence = rotationSequence
        self.positionRoll = positionRoll
        self.extraPitch = extraPitch
        self.extraRoll = extraRoll
        self.extraYaw = extraYaw
        self.extraRotationSequence = extraRotationSequence
        self.alarmLevel = alarmLevel
        self.isParametric = False
        self.shape = shape
        self.overEdge = kwargs.get('overEdge', 'yMax')
        self.surface = surface
        self.material = material
        self.alpha = alpha
        self.curSurface = 0
        self.dx = 0
        self.limOptX = limOptX
        self.limOptY = limOptY
        self.limPhysX = limPhysX
        self.limPhysY = limPhysY
        if order is not None and not isinstance(order, (int, np.integer)):
            raise NotImplementedError('a sequence of diffraction orders (random '
                                      'order per ray)')
        self.order = 1 if order is None else int(order)      # base.py:507-509
        self.gratingDensity = gratingDensity
        self.footprint = []

    # -- asymmetric cut ----------------------------------------------------
    @property
    def alpha(self):
        return self._alpha

    @alpha.setter
    def alpha(self, alpha):
        self._alpha = alpha
        if alpha is not None:
            self.cosalpha = float(np.cos(alpha))
            self.sinalpha = float(np.sin(alpha))

    # -- surface: flat (oes/base.py:675-742) ----------------------------------
    def local_z(self, x, y):
        return np.zeros_like(y)

    def local_n(self, x, y):
        a, b, c = 0., 0., 1.
        if self.alpha:
            bAlpha, cAlpha = raycing.rotate_x(b, c, self.cosalpha, -self.sinalpha)
            return [a, bAlpha, cAlpha, a, b, c]
        return [a, b, c]

    def _surface_params(self, p, second=False):
        p.surf_kind = _structs.SURF_FLAT
        n = list(self.local_n2(0., 0.) if second else self.local_n(0., 0.))
        if len(n) == 3:
            n = n + n
            p.asymmetric = 0
        else:
            p.asymmetric = 1
        for i in range(6):
            p.n_const[i] = float(n[i])

    def local_n2(self, x, y):
        return self.local_n(x, y)

    # -- gratings by the grating equation (base.py:688-717, reflect.py:840-861) --
    def local_g(self, x, y, rho=-100.):
        """Reciprocal groove vector [1/mm] at (x, y): the *gratingDensity*
        polynomial ['x'|'y', rho0, p0, p1, ...] or (0, rho, 0). Subclasses may
        override it with a CONSTANT vector (evaluated on the host once)."""
        rhoList = self.gratingDensity
        if rhoList is not None:
            coord = x if rhoList[0] == 'x' else y
            poly = 0.
            for ic, coeff in enumerate(rhoList[2:]):
                poly += (ic+1) * coeff * coord**ic
            N = rhoList[1] * poly
            if rhoList[0] == 'x':
                return N, np.zeros_like(N), np.zeros_like(N)
            return np.zeros_like(N), N, np.zeros_like(N)
        return 0, rho, 0

    def _is_grating(self):
        material = self.material
        if raycing.is_sequence(material):
            material = material[0] if len(material) == 1 else None
        if material is None:
            return False
        kind = getattr(material, 'kind', None)
        if kind == 'auto' and self.gratingDensity is not None:   # base.py:1088-1092
            return True
        return kind == 'grating'

    def _grating_params(self, p, second=False):
        p.grating = 0
        if second or not self._is_grating():
            return
        if self.isParametric:
            raise NotImplementedError('grating equation on a parametric surface')
        p.grating = 1
        p.grating_order = int(self.order)
        rhoList = self.gratingDensity
        overridden = type(self).local_g is not OE.local_g
        if rhoList is not None and not overridden:
            coefs = [float(c) for c in rhoList[2:]]
            if len(coefs) > 8 or rhoList[0] not in ('x', 'y'):
                raise NotImplementedError('gratingDensity %r' % (rhoList,))
            p.grating_axis = 0 if rhoList[0] == 'x' else 1
            p.g_rho0 = float(rhoList[1])
            p.g_ncoef = len(coefs)
            for i, c in enumerate(coefs):
                p.g_coef[i] = c
        else:
            xs = np.array([-0.7, 0., 0.3, 1.1])
            ys = np.array([0.9, 0., -1.3, 0.2])
            g = [np.broadcast_to(np.asarray(v, dtype=float), xs.shape)
                 for v in self.local_g(xs, ys)]
            if any(np.ptp(v) != 0 for v in g):
                raise NotImplementedError(
                    'a user-defined position-dependent local_g: express it as '
                    'gratingDensity')
            p.grating_axis = -1
            for i in range(3):
                p.g_const[i] = float(g[i][0])

    def _surface_height(self, x, y):
        """z of the surface above (x, y), also for parametric surfaces
        (reflect.py:336-341)."""
        if self.isParametric:
            s, phi, r = self.xyz_to_param(x, y, 0)
            r = self.local_r(s, phi)
            return self.param_to_xyz(s, phi, r)[2]
        return self.local_z(x, y)

    # -- Coddington radii (oes/base.py:649-673) -------------------------------
    def get_Rmer_from_Coddington(self, p, q, pitch=None):
        if pitch is None:
            pitch = self.pitch
        return 2 * p * q / (p+q) / np.sin(abs(pitch))

    def get_rsag_from_Coddington(self, p, q, pitch=None):
        if pitch is None:
            pitch = self.pitch
        return 2 * p * q / (p+q) * np.sin(abs(pitch))

    # -- the parameter block of one _reflect_local pass ------------------------
    def _limits_for(self, second):
        sfx = '2' if second else ''
        return (getattr(self, 'limPhysX' + sfx), getattr(self, 'limPhysY' + sfx),
                getattr(self, 'limOptX' + sfx, None),
                getattr(self, 'limOptY' + sfx, None))

    def _make_pass(self, pitch, roll, yaw, dx=0, dy=0, dz=0, fromVacuum=True,
                   is2ndXtal=False, noIntersectionSearch=False, in_is_global=True,
                   good_mode=0, out_to_global=True, only_state1_out=False,
                   zero_local_not_entering=False, force_lost_out=False):
        p = _structs.Pass()
        p.good_mode = good_mode
        p.in_is_global = 1 if in_is_global else 0
        for i in range(3):
            p.center[i] = float(self.center[i])
        p.sin_az = self.bl.sinAzimuth if self.bl is not None else 0.
        p.cos_az = self.bl.cosAzimuth if self.bl is not None else 1.
        # rotate the world around the element, reflect.py:617-629 ...
        to_local = []
        extraSign = 1.
        if is2ndXtal:
            to_local += raycing.rotation_steps(roll=-np.pi)
            extraSign = -1.
        to_local += raycing.rotation_steps(self.rotationSequence, pitch=-pitch,
                                           roll=-roll, yaw=-yaw)
        if self.extraPitch or self.extraRoll or self.extraYaw:
            to_local += raycing.rotation_steps(
                self.extraRotationSequence, pitch=-extraSign*self.extraPitch,
                roll=-self.extraRoll, yaw=-extraSign*self.extraYaw)
        # ... and back, reflect.py:1122-1132
        to_virgin = []
        if self.extraPitch or self.extraRoll or self.extraYaw:
            to_virgin += raycing.rotation_steps(
                '-' + self.extraRotationSequence, pitch=extraSign*self.extraPitch,
                roll=self.extraRoll, yaw=extraSign*self.extraYaw)
        to_virgin += raycing.rotation_steps('-' + self.rotationSequence,
                                            pitch=pitch, roll=roll, yaw=yaw)
        if is2ndXtal:
            to_virgin += raycing.rotation_steps(roll=np.pi)
        for rot, steps in ((p.to_local, to_local), (p.to_virgin, to_virgin)):
            if len(steps) > _structs.MAX_ROT:
                raise ValueError('too many rotation steps')
            rot.n = len(steps)
            for i, (ax, c, s) in enumerate(steps):
                rot.axis[i] = ax
                rot.cosa[i] = c
                rot.sina[i] = s
        p.shift[0] = float(dx or 0.)
        p.shift[1] = float(dy or 0.)
        p.shift[2] = float(dz or 0.)
        if hasattr(self, 'invertNormal'):
            p.invert_normal = int(self.invertNormal)
        else:
            p.invert_normal = 1 if fromVacuum else -1
        p.no_intersection_search = 1 if noIntersectionSearch else 0
        self._surface_params(p, is2ndXtal)
        physX, physY, optX, optY = self._limits_for(is2ndXtal)
        if self.shape.startswith('re'):
            p.shape = _structs.SHAPE_RECT
        elif self.shape.startswith('ro'):
            p.shape = _structs.SHAPE_ROUND
        else:
            raise NotImplementedError('shape %r' % (self.shape,))
        p.phys_x[0], p.phys_x[1] = float(physX[0]), float(physX[1])
        p.phys_y[0], p.phys_y[1] = float(physY[0]), float(physY[1])
        p.has_opt_x = 0 if optX is None else 1
        p.has_opt_y = 0 if optY is None else 1
        if optX is not None:
            p.opt_x[0], p.opt_x[1] = float(optX[0]), float(optX[1])
        if optY is not None:
            p.opt_y[0], p.opt_y[1] = float(optY[0]), float(optY[1])
        ovE = str(getattr(self, 'overEdge', 'yMax')).lower()
        mask = 0
        if 'xmin' in ovE:
            mask |= _structs.OVER_XMIN
        if 'xmax' in ovE:
            mask |= _structs.OVER_XMAX
        if 'ymin' in ovE:
            mask |= _structs.OVER_YMIN
        if 'ymax' in ovE:
            mask |= _structs.OVER_YMAX
        p.over_mask = mask
        p.lost_num = int(self.lostNum)
        p.roll = float(roll)
        p.cos_roll = float(np.cos(roll))
        p.sin_roll = float(np.sin(roll))
        p.out_to_global = 1 if out_to_global else 0
        p.only_state1_out = 1 if only_state1_out else 0
        p.zero_local_not_entering = 1 if zero_local_not_entering else 0
        self._grating_params(p, is2ndXtal)
        p.force_lost_out = 1 if force_lost_out else 0
        return p

    @staticmethod
    def _material_struct(material, fromVacuum, device):
        if material is None:
            s = _structs.Material()
            s.kind = _structs.MAT_NONE
            s.from_vacuum = 1 if fromVacuum else 0
            s._keep = []
            return s
        if raycing.is_sequence(material):     # reflect.py:725-728, one stripe only
            if len(material) != 1:
                raise NotImplementedError('multi-stripe material lists')
            material = material[0]
        return material.to_struct(fromVacuum, device)

    def _run_pass(self, p, material, fromVacuum, beam_in, restore, want_info=False,
                  timing=False, out=None):
        """-> (lb, vlb) device-resident beams (+ info dict). *out*: an (lb, vlb)
        pair from an earlier call on a beam of the same size to be overwritten
        instead of allocating new arrays."""
        _lib.require_gpu()
        lib = _lib.load()
        dev = torch.device('cuda', torch.cuda.current_device())
        ms = self._material_struct(material, fromVacuum, dev)
        s_in = beam_in.to_struct(dev)
        s_re = s_in if restore is beam_in else restore.to_struct(dev)
        n = beam_in.nrays
        if out is not None and out[0].nrays == n and out[1].nrays == n and \
                out[0].has_amplitudes() == beam_in.has_amplitudes() and \
                not out[0]._h_dirty() and not out[1]._h_dirty() and 'theta' in out[0]._d:
            lb, vb = out
            theta = lb._d['theta']
        else:
            lb = rs.Beam.empty_like_on_device(beam_in, dev)
            vb = rs.Beam.empty_like_on_device(beam_in, dev)
            theta = torch.empty(n, dtype=torch.float64, device=dev)   # fully written
        s_lb, s_vb = lb.to_struct(dev), vb.to_struct(dev)
        wsb = lib.xrt_hip_reflect_workspace_bytes(n)
        ws = hipcalls.workspace(dev, wsb, 'reflect')
        info = (ctypes.c_double * 16)() if want_info else None
        ms_out = (ctypes.c_float * 3)() if timing else None
        rc = lib.xrt_hip_reflect_pass_f64_dev(
            ctypes.byref(p), ctypes.byref(ms), ctypes.byref(s_in),
            ctypes.byref(s_re), ctypes.byref(s_lb), ctypes.byref(s_vb),
            ctypes.c_void_p(theta.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
            ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
            info, ms_out)
        _lib.check(rc, 'xrt_hip_reflect_pass_f64_dev')
        if 'theta' not in lb._d or lb._d['theta'] is not theta:
            lb._h.pop('theta', None)
            lb._d['theta'] = theta
        for b in (lb, vb):
            for k in rs._SCALAR_ATTRS:
                if k in beam_in.__dict__:
                    object.__setattr__(b, k, beam_in.__dict__[k])
            b.parentId = self.uuid
        res_info = None
        if want_info:
            v = list(info)
            res_info = dict(axis=int(v[0]), positive=bool(v[1]), brent=bool(v[2]),
                            tMinGlobal=v[3], tMaxGlobal=v[4], maxdz1=v[5],
                            maxdz2=v[6], n_enter=int(v[7]),
                            mixed_sign=bool(v[10] and v[11]))
        if timing:
            res_info = res_info or {}
            res_info['pass_ms'] = ms_out[0]
            res_info['kernel_ms'] = ms_out[1]
            res_info['exact_sequence'] = bool(ms_out[2])
        return lb, vb, res_info

    # -- local -> global for a beam on the surface (oes/base.py:1165-1229) ------
    def local_to_global(self, lb, returnBeam=False, is2ndXtal=False, **kwargs):
        """Host-side (wave post-processing glue): *lb* in the true local frame
        -> global frame, coherency matrix and amplitudes included."""
        dx, dy, dz = 0, 0, 0
        extraAnglesSign = 1.
        if hasattr(self, 'cryst2pitch'):
            if is2ndXtal:
                pitch = -self.pitch - self.bragg + self.cryst2pitch +\
                    self.cryst2finePitch
                roll = self.roll + self.cryst2roll + self.positionRoll
                yaw = -self.yaw
                dx = -self.dx
                dy = self.cryst2longTransl
                dz = -self.cryst2perpTransl
                extraAnglesSign = -1.
            else:
                pitch = self.pitch + self.bragg
                roll = self.roll + self.positionRoll + self.cryst1roll
                yaw = self.yaw
                dx = self.dx
        else:
            pitch = self.pitch
            roll = self.roll + self.positionRoll
            yaw = self.yaw
        if dx:
            lb.x += dx
        if dy:
            lb.y += dy
        if dz:
            lb.z += dz
        if self.extraPitch or self.extraRoll or self.extraYaw:
            raycing.rotate_beam(
                lb, rotationSequence='-'+self.extraRotationSequence,
                pitch=extraAnglesSign*self.extraPitch, roll=self.extraRoll,
                yaw=extraAnglesSign*self.extraYaw)
        raycing.rotate_beam(lb, rotationSequence='-'+self.rotationSequence,
                            pitch=pitch, roll=roll, yaw=yaw)
        if hasattr(self, 'cryst2pitch') and is2ndXtal:
            raycing.rotate_beam(lb, roll=np.pi)
        if self.isParametric:                  # base.py:1210-1214
            s, phi, r = self.xyz_to_param(lb.x, lb.y, lb.z)
            oeNormal = list(self.local_n(s, phi))
        else:
            oeNormal = list(self.local_n(lb.x, lb.y))
        roll = self.roll + self.positionRoll +\
            np.arctan2(oeNormal[-3], oeNormal[-1])
        lb.Jss[:], lb.Jpp[:], lb.Jsp[:] =\
            rs.rotate_coherency_matrix(lb, slice(None), roll)
        if hasattr(lb, 'Es'):
            cosY, sinY = np.cos(roll), np.sin(roll)
            lb.Es[:], lb.Ep[:] = raycing.rotate_y(lb.Es, lb.Ep, cosY, sinY)
        if returnBeam:
            retGlo = rs.Beam(copyFrom=lb)
            raycing.virgin_local_to_global(self.bl, retGlo, self.center)
            return retGlo
        raycing.virgin_local_to_global(self.bl, lb, self.center)

    # -- wave propagation through the element (oes/reflect.py:266-449) ---------
    def prepare_wave(self, prevOE, nrays, shape='auto', area='auto', rw=None):
        """Wave samples on this surface that will receive the field diffracted
        by *prevOE*: random over the physical limits (int *nrays*), a uniform mesh
        ((nx, ny)) or given positions ((x array, y array)). Uses the global
        np.random state like the reference."""
        if rw is None:
            from . import waves as rw
        if isinstance(nrays, (int, float)):
            nsamples = int(nrays)
        elif isinstance(nrays, (list, tuple)):
            if isinstance(nrays[0], (int, float)):
                nsamples = nrays[0] * nrays[1]
            elif isinstance(nrays[0], np.ndarray) and isinstance(nrays[1], np.ndarray):
                nsamples = len(nrays[0]) * len(nrays[1])
            else:
                raise ValueError('wrong type of `nrays`!')
        else:
            raise ValueError('wrong type of `nrays`!')
        lb = rs.Beam(nrays=nsamples, forceState=1, withAmplitudes=True)
        lb.parentId = prevOE.uuid
        if shape == 'auto':
            shape = self.shape
        if isinstance(nrays, (int, float)):
            xy = np.random.rand(nsamples, 2)
            if shape.startswith('ro'):
                dR = (self.limPhysX[1] - self.limPhysX[0]) / 2
                r = xy[:, 0]**0.5 * dR
                phi = xy[:, 1] * 2*np.pi
                x = r * np.cos(phi)
                y = r * np.sin(phi)
                if area == 'auto':
                    area = np.pi * dR**2
            elif shape.startswith('re'):
                dX = self.limPhysX[1] - self.limPhysX[0]
                dY = self.limPhysY[1] - self.limPhysY[0]
                x = xy[:, 0] * dX + self.limPhysX[0]
                y = xy[:, 1] * dY + self.limPhysY[0]
                if area == 'auto':
                    area = dX * dY
            else:
                raise ValueError('unknown shape!')
        else:
            if shape.startswith('ro'):
                raise ValueError('must be rectangular')
            if isinstance(nrays[0], (int, float)):
                xx = np.linspace(*self.limPhysX, nrays[0])
                yy = np.linspace(*self.limPhysY, nrays[1])
            else:
                xx, yy = nrays
            X, Y = np.meshgrid(xx, yy)
            x = X.ravel()
            y = Y.ravel()
            if area == 'auto':
                area = (self.limPhysX[1] - self.limPhysX[0]) * \
                    (self.limPhysY[1] - self.limPhysY[0])
        lb.x[:] = x
        lb.y[:] = y
        lb.z[:] = self._surface_height(x, y)
        self.local_to_global(lb)
        if hasattr(prevOE, 'rotationSequence'):   # the previous element is an OE
            cx = (prevOE.limPhysX[1] + prevOE.limPhysX[0])*0.5
            cy = (prevOE.limPhysY[1] + prevOE.limPhysY[0])*0.5
            cz = prevOE._surface_height(np.atleast_1d(float(cx)),
                                        np.atleast_1d(float(cy)))
            lbc = rs.Beam(nrays=1)
            lbc.x[:] = cx
            lbc.y[:] = cy
            lbc.z[:] = cz
            prevOE.local_to_global(lbc)
            prevCenter = [lbc.x[0], lbc.y[0], lbc.z[0]]
        else:
            prevCenter = prevOE.center
        lb.a[:] = lb.x - prevCenter[0]
        lb.b[:] = lb.y - prevCenter[1]
        lb.c[:] = lb.z - prevCenter[2]
        norm = (lb.a**2 + lb.b**2 + lb.c**2)**0.5
        lb.a /= norm
        lb.b /= norm
        lb.c /= norm
        lb.x[:] = prevCenter[0]
        lb.y[:] = prevCenter[1]
        lb.z[:] = prevCenter[2]
        lbn = rs.Beam(nrays=1)
        lbn.b[:] = 0.
        lbn.c[:] = 1.
        self.local_to_global(lbn)
        a = lbn.x - prevCenter[0]
        b = lbn.y - prevCenter[1]
        c = lbn.z - prevCenter[2]
        norm = (a**2 + b**2 + c**2)**0.5
        areaNormalFact = abs(float(((a*lbn.a[0] + b*lbn.b[0] + c*lbn.c[0]) / norm)[0]))
        waveGlobal, waveLocal = self.reflect(lb)        # HIP kernels
        good = (waveLocal.state == 1) | (waveLocal.state == 2)
        waveGlobal.filter_by_index(good)
        waveLocal.filter_by_index(good)
        area *= good.sum() / float(len(good))
        waveLocal.area = area
        waveLocal.areaNormal = area * areaNormalFact
        waveLocal.dS = area / float(len(good))
        waveLocal.toOE = self
        waveLocal.parentId = self.uuid
        rw.prepare_wave(prevOE, waveLocal, waveGlobal.x, waveGlobal.y, waveGlobal.z)
        return waveLocal

    def propagate_wave(self, wave=None, beam=None, nrays='auto'):
        """Kirchhoff-propagates *wave* (the local field on the previous element)
        onto this surface and reflects it: -> (beamGlobal, beamLocal) usable for
        further ray or wave propagation (oes/reflect.py:405-449). This is the
        explicit sequence prepare_wave -> diffract -> reflect(noIntersectionSearch)
        that the reference's wave examples spell out; the reference's own
        propagate_wave additionally transforms *wave* in place while
        auto-aligning (reflect.py:434-438), a side effect that is not reproduced
        (golden case G8 is generated with the explicit sequence)."""
        from . import waves as rw
        waveSize = len(wave.x) if nrays == 'auto' else int(nrays)
        prevOE = self.bl.oesDict[wave.parentId][0]
        if hasattr(prevOE, 'shine'):
            raise NotImplementedError('wave propagation directly from a source')
        waveOnSelf = self.prepare_wave(prevOE, waveSize, rw=rw)
        beamToSelf = rw.diffract(wave, waveOnSelf)
        retGlo, retLoc = self.reflect(beamToSelf, noIntersectionSearch=True)
        retLoc.parentId = self.uuid
        return retGlo, retLoc

    # -- OE.reflect, oes/reflect.py:18-163 ----------------------------------
    def reflect(self, beam=None, needLocal=True, noIntersectionSearch=False,
                returnLocalAbsorbed=None, _info=None, out=None, _timing=None):
        """-> (beamGlobal, beamLocal). *out* (extension): the pair returned by an
        earlier call, to be overwritten in place (no new HBM allocations).
        *_info* (dict) receives the batch statistics (this takes the exact kernel
        sequence); *_timing* (dict) the pass / kernel milliseconds and whether the
        exact sequence had to run."""
        pitch = self.pitch
        if hasattr(self, 'bragg'):
            pitch = pitch + self.bragg
        p = self._make_pass(
            pitch, self.roll + self.positionRoll, self.yaw, self.dx,
            noIntersectionSearch=noIntersectionSearch,
            only_state1_out=hasattr(beam, 'createdByDiffract'))
        lb, gb, info = self._run_pass(
            p, self.material, True, beam, beam, want_info=_info is not None,
            timing=_timing is not None, out=None if out is None else (out[1], out[0]))
        if _info is not None:
            _info.update({k: v for k, v in info.items() if k not in
                          ('pass_ms', 'kernel_ms', 'exact_sequence')})
        if _timing is not None:
            _timing.update({k: info[k] for k in ('pass_ms', 'kernel_ms', 'exact_sequence')})
        return gb, lb


class ToroidMirror(OE):
    """Toroidal mirror: z = y^2/(2R) + r - sqrt(r^2 - x^2)
    (oes/__init__.py:321-411). R and r may be (p, q) tuples for the Coddington
    equations."""

    def __init__(self, *args, **kwargs):
        R = kwargs.pop('R', 5.0e6)
        r = kwargs.pop('r', 50.)
        OE.__init__(self, *args, **kwargs)
        self.R = R
        self.r = r

    @property
    def R(self):
        return self._RVal

    @R.setter
    def R(self, R):
        if isinstance(R, (list, tuple)):
            self._RVal = self.get_Rmer_from_Coddington(*R)
        elif R in [0, None]:
            self._RVal = 1e100
        else:
            self._RVal = R

    @property
    def r(self):
        return self._rVal

    @r.setter
    def r(self, r):
        if isinstance(r, (list, tuple)):
            self._rVal = self.get_rsag_from_Coddington(*r)
        elif r in [0, None]:
            self._rVal = 1e100
        else:
            self._rVal = r

    def local_z(self, x, y):
        rx = 1 - (np.asarray(x)/self.r)**2
        rx[rx < 0] = 0.
        return y**2/2.0/self.R + self.r*(1 - rx**0.5)

    def local_n(self, x, y):
        rx = 1 - (np.asarray(x)/self.r)**2
        with np.errstate(divide='ignore', invalid='ignore'):
            ax = np.where(rx < 0, 0, rx**(-0.5))
        a = -x / self.r * ax
        b = -y / self.R
        c = 1.
        norm = (a**2 + b**2 + 1)**0.5
        return [a/norm, b/norm, c/norm]

    def _surface_params(self, p, second=False):
        p.surf_kind = _structs.SURF_TOROID
        p.surf_p[0] = float(self.R)
        p.surf_p[1] = float(self.r)
        # correctly rounded reciprocals for the constant-divisor division of the
        # kernel; only for ordinary radii (R = 1e100 / inf mean 'flat')
        ok = all(np.isfinite(v) and 1e-100 < abs(v) < 1e100 for v in (self.R, self.r))
        p.surf_p[2] = 1.0 / float(self.R) if ok else 0.
        p.surf_p[3] = 1.0 / float(self.r) if ok else 0.
        p.surf_p[4] = 1.0 if ok else 0.
        p.asymmetric = 0
        for i, v in enumerate((0., 0., 1., 0., 0., 1.)):
            p.n_const[i] = v


SimpleVFM = ToroidMirror


class BlazedGrating(OE):
    """Saw-tooth grating of constant line density for WAVE propagation: the
    diffraction comes from the surface itself through the Kirchhoff integral,
    the material is a mirror (oes/gratings.py:316-535). Facet geometry, the
    first-facet intersection and the facet normals run in the reflect kernels
    (surface kind XRT_HIP_SURF_BLAZED)."""

    def __init__(self, *args, **kwargs):
        self.blaze = raycing.auto_units_angle(kwargs.pop('blaze'))
        self.antiblaze = raycing.auto_units_angle(
            kwargs.pop('antiblaze', np.pi*0.4999))
        self.rho0 = kwargs.pop('rho', 1)
        if kwargs.get('gratingDensity') is not None:
            raise NotImplementedError('variable line density')
        OE.__init__(self, *args, **kwargs)
        self.gratingDensity = None
        self.reset()

    def reset(self):
        self.rho_1 = 1. / self.rho0
        self.sinBlaze, self.cosBlaze, self.tanBlaze = \
            np.sin(self.blaze), np.cos(self.blaze), np.tan(self.blaze)
        self.sinAntiblaze, self.cosAntiblaze, self.tanAntiblaze = \
            np.sin(self.antiblaze), np.cos(self.antiblaze), np.tan(self.antiblaze)

    def local_pre(self, x, y):
        # np.divmod = the same npy_divmod as `//` and `%`, in one pass
        q, yL = np.divmod(y, self.rho_1)
        y0 = q * self.rho_1
        y1 = y0 + self.rho_1
        yC = (y1-y0) / (1 + self.tanAntiblaze/self.tanBlaze)
        return 0, y0, y1, yC, yL

    def local_z(self, x, y):
        y0ind, y0, y1, yC, yL = self.local_pre(x, y)
        return np.where(yL > yC, -(y1-y) * self.tanBlaze, -yL * self.tanAntiblaze)

    def local_n(self, x, y):
        y0ind, y0, y1, yC, yL = self.local_pre(x, y)
        return [np.zeros_like(x),
                np.where(yL > yC, -self.sinBlaze, self.sinAntiblaze),
                np.where(yL > yC, self.cosBlaze, self.cosAntiblaze)]

    def get_grating_area_fraction(self):
        """Illuminated fraction of the groove length (gratings.py:524-535)."""
        tanPitch = np.tan(abs(self.pitch))
        y1 = self.rho_1 * self.tanBlaze / (self.tanBlaze + tanPitch)
        z1 = -y1 * tanPitch
        y2 = self.rho_1
        z2 = 0
        d = ((y2-y1)**2 + (z2-z1)**2)**0.5
        return d * self.rho0

    def _surface_params(self, p, second=False):
        p.surf_kind = _structs.SURF_BLAZED
        vals = (self.rho_1, self.tanBlaze, self.tanAntiblaze, self.sinBlaze,
                self.cosBlaze, self.sinAntiblaze, self.cosAntiblaze,
                1 + self.tanAntiblaze/self.tanBlaze,
                1. if self.blaze == np.pi/2 else 0.,
                1. if self.antiblaze == np.pi/2 else 0.)
        for i, v in enumerate(vals):
            p.surf_p[i] = float(v)
        p.asymmetric = 0
        for i, v in enumerate((0., 0., 1., 0., 0., 1.)):
            p.n_const[i] = v


class EllipticalMirrorParam(OE):
    """Elliptical mirror (ellipsoid of revolution, or elliptical cylinder with
    *isCylindrical*) in the parametric coordinates (s, phi, r) of
    oes/parametric.py:9-249: s along the major axis, r the distance from it.
    The root solve runs on r - local_r(s, phi) in the reflect kernels (surface
    kind XRT_HIP_SURF_ELLIPSE_PARAM). *f1*/*f2*/*pAxis* are not mirrored: give
    *p* and *q* (the p arm along the global y axis)."""

    def __init__(self, *args, **kwargs):
        for k in ('f1', 'f2', 'pAxis'):
            if kwargs.pop(k, None) is not None:
                raise NotImplementedError('EllipticalMirrorParam(%s=...)' % k)
        p = kwargs.pop('p', 1000)
        q = kwargs.pop('q', 1000)
        self.isCylindrical = kwargs.pop('isCylindrical', False)
        self.isClosed = kwargs.pop('isClosed', False)
        OE.__init__(self, *args, **kwargs)
        self.isParametric = True
        self._p, self._q = p, q
        self._reset_pq()

    # every quantity that enters the ellipse is a property so that a later
    # `oe.pitch = ...` re-derives it, like the reference's setters do
    def _get(name):
        return property(lambda self: getattr(self, '_' + name),
                        lambda self, v: (setattr(self, '_' + name, v),
                                         self._reset_pq())[0])
    p = _get('p')
    q = _get('q')
    pitch = _get('pitchVal')
    roll = _get('rollVal')
    yaw = _get('yawVal')
    positionRoll = _get('positionRollVal')
    del _get

    def _reset_pq(self):
        """parametric.py:117-157."""
        need = ('_p', '_q', '_pitchVal', '_rollVal', '_yawVal', '_positionRollVal',
                'rotationSequence')
        if not all(hasattr(self, v) for v in need) or self.bl is None:
            return
        lbn = rs.Beam(nrays=1)
        lbn.a[:], lbn.b[:], lbn.c[:] = 0, 0, 1
        raycing.rotate_beam(lbn, rotationSequence='-'+self.rotationSequence,
                            pitch=self.pitch, roll=self.roll+self.positionRoll,
                            yaw=self.yaw, skip_xyz=True)
        raycing.virgin_local_to_global(self.bl, lbn, self.center, skip_xyz=True)
        normal = lbn.a[0], lbn.b[0], lbn.c[0]
        axis = [0, 1, 0]
        norm = sum([a**2 for a in axis])**0.5
        sintheta = sum([a*n for a, n in zip(axis, normal)]) / norm
        absPitch = abs(np.arcsin(sintheta))
        if self.p and self.q:
            gamma = np.arctan2((self.p - self.q) * np.sin(absPitch),
                               (self.p + self.q) * np.cos(absPitch))
            self.cosGamma = np.cos(gamma)
            self.sinGamma = np.sin(gamma)
            self.y0 = (self.q - self.p)/2. * np.cos(absPitch)
            self.z0 = (self.q + self.p)/2. * np.sin(absPitch)
            self.ellipseA = (self.q + self.p)/2.
            self.ellipseB = np.sqrt(self.q * self.p) * np.sin(absPitch)

    def xyz_to_param(self, x, y, z):
        yNew, zNew = raycing.rotate_x(y - self.y0, z - self.z0, self.cosGamma,
                                      self.sinGamma)
        return yNew, np.arctan2(x, zNew), np.sqrt(x**2 + zNew**2)

    def param_to_xyz(self, s, phi, r):
        x = r * np.sin(phi)
        y = s
        z = r * np.cos(phi)
        yNew, zNew = raycing.rotate_x(y, z, self.cosGamma, -self.sinGamma)
        return x, yNew + self.y0, zNew + self.z0

    def local_r(self, s, phi):
        r = self.ellipseB * np.sqrt(abs(1 - s**2 / self.ellipseA**2))
        if self.isCylindrical:
            r /= abs(np.cos(phi))
        if self.isClosed:
            return r
        return np.where(abs(phi) > np.pi/2, r, np.ones_like(phi)*1e20)

    def local_n(self, s, phi):
        A2s2 = np.array(self.ellipseA**2 - s**2)
        A2s2[A2s2 <= 0] = 1e22
        nr = -self.ellipseB / self.ellipseA * s / np.sqrt(A2s2)
        norm = np.sqrt(nr**2 + 1)
        b = nr / norm
        if self.isCylindrical:
            a = np.zeros_like(phi)
            c = 1. / norm
        else:
            a = -np.sin(phi) / norm
            c = -np.cos(phi) / norm
        bNew, cNew = raycing.rotate_x(b, c, self.cosGamma, -self.sinGamma)
        return [a, bNew, cNew]

    def _surface_params(self, p, second=False):
        p.surf_kind = _structs.SURF_ELLIPSE_PARAM
        vals = (self.y0, self.z0, self.cosGamma, self.sinGamma, self.ellipseA,
                self.ellipseB, 1. if self.isCylindrical else 0.,
                1. if self.isClosed else 0., 0.)
        for i, v in enumerate(vals):
            p.surf_p[i] = float(v)
        p.asymmetric = 0
        for i, v in enumerate((0., 0., 1., 0., 0., 1.)):
            p.n_const[i] = v


EllipticalMirror = EllipticalMirrorParam


class _ConicMirrorParam(EllipticalMirrorParam):
    """Shared parts of the other two conics of revolution: same parametric
    frame (s along the axis, r from it), own generatrix."""
    conic = None

    def _conic_ab(self):
        raise NotImplementedError

    def _abs_pitch(self):
        lbn = rs.Beam(nrays=1)
        lbn.a[:], lbn.b[:], lbn.c[:] = 0, 0, 1
        raycing.rotate_beam(lbn, rotationSequence='-'+self.rotationSequence,
                            pitch=self.pitch, roll=self.roll+self.positionRoll,
                            yaw=self.yaw, skip_xyz=True)
        raycing.virgin_local_to_global(self.bl, lbn, self.center, skip_xyz=True)
        axis = [0, 1, 0]
        norm = sum([a**2 for a in axis])**0.5
        sintheta = sum([a*n for a, n in
                        zip(axis, (lbn.a[0], lbn.b[0], lbn.c[0]))]) / norm
        return abs(np.arcsin(sintheta))

    def _ready(self):
        need = ('_p', '_q', '_pitchVal', '_rollVal', '_yawVal', '_positionRollVal',
                'rotationSequence')
        return all(hasattr(self, v) for v in need) and self.bl is not None

    def _surface_params(self, p, second=False):
        p.surf_kind = _structs.SURF_ELLIPSE_PARAM
        A, B = self._conic_ab()
        vals = (self.y0, self.z0, self.cosGamma, self.sinGamma, A, B,
                1. if self.isCylindrical else 0., 1. if self.isClosed else 0.,
                float(self.conic))
        for i, v in enumerate(vals):
            p.surf_p[i] = float(v)
        p.asymmetric = 0
        for i, v in enumerate((0., 0., 1., 0., 0., 1.)):
            p.n_const[i] = v


class ParabolicalMirrorParam(_ConicMirrorParam):
    """Paraboloid (or parabolic cylinder): collimates a source at distance *p*
    or focuses a parallel beam at *q* — exactly one of them is given
    (oes/parametric.py:252-474)."""
    conic = 1

    def __init__(self, *args, **kwargs):
        if kwargs.pop('parabolaAxis', None) is not None:
            raise NotImplementedError('ParabolicalMirrorParam(parabolaAxis=...)')
        kwargs.setdefault('p', 10000)
        kwargs.setdefault('q', None)
        EllipticalMirrorParam.__init__(self, *args, **kwargs)

    def _reset_pq(self):
        if not self._ready():
            return
        if (self.p is None) == (self.q is None):
            raise ValueError('One and only one of p or q must be None!')
        absPitch = self._abs_pitch()
        if self.p is None:
            self.y0 = self.q * np.cos(absPitch)
            self.z0 = self.q * np.sin(absPitch)
            self.parabParam = -self.q * np.sin(absPitch)**2
            gamma = absPitch
        else:
            self.y0 = -self.p * np.cos(absPitch)
            self.z0 = self.p * np.sin(absPitch)
            self.parabParam = self.p * np.sin(absPitch)**2
            gamma = -absPitch
        self.cosGamma = np.cos(gamma)
        self.sinGamma = np.sin(gamma)

    def _conic_ab(self):
        return self.parabParam, 0.

    def local_r(self, s, phi):
        r2 = self.parabParam*s + self.parabParam**2
        r2[r2 < 0] = 0
        r = 2 * r2**0.5
        if self.isCylindrical:
            r /= abs(np.cos(phi))
        if self.isClosed:
            return r
        return np.where(abs(phi) > np.pi/2, r, np.ones_like(phi)*1e20)

    def local_n(self, s, phi):
        nr = self.parabParam / (self.parabParam*s + self.parabParam**2)**0.5
        return self._normal_from_slope(nr, phi, -1.)

    def _normal_from_slope(self, nr, phi, sign):
        norm = np.sqrt(nr**2 + 1)
        b = nr / norm
        if self.isCylindrical:
            a = np.zeros_like(phi)
            c = 1. / norm
        elif sign < 0:
            a = -np.sin(phi) / norm
            c = -np.cos(phi) / norm
        else:
            a = np.sin(phi) / norm
            c = np.cos(phi) / norm
        bNew, cNew = raycing.rotate_x(b, c, self.cosGamma, -self.sinGamma)
        return [a, bNew, cNew]


ParabolicMirror = ParabolicalMirrorParam


class HyperbolicMirrorParam(_ConicMirrorParam):
    """Hyperboloid (or hyperbolic cylinder) between the foci at *p* and *q*; the
    OUTER surface reflects (``invertNormal = -1``), oes/parametric.py:477-716."""
    conic = 2
    _normal_from_slope = ParabolicalMirrorParam._normal_from_slope

    def __init__(self, *args, **kwargs):
        EllipticalMirrorParam.__init__(self, *args, **kwargs)
        self.invertNormal = -1

    def _reset_pq(self):
        if not self._ready():
            return
        absPitch = self._abs_pitch()
        if self.p and self.q:
            gamma = np.arctan2((self.p + self.q) * np.sin(absPitch),
                               (self.p - self.q) * np.cos(absPitch))
            self.cosGamma = np.cos(gamma)
            self.sinGamma = np.sin(gamma)
            self.y0 = -(self.p + self.q)/2. * np.cos(absPitch)
            self.z0 = (self.p - self.q)/2. * np.sin(absPitch)
            self.hyperbolaA = abs(self.p - self.q)/2.
            self.hyperbolaB = np.sqrt(self.p*self.q) * np.sin(absPitch)

    def _conic_ab(self):
        return self.hyperbolaA, self.hyperbolaB

    def local_r(self, s, phi):
        r = self.hyperbolaB * np.sqrt(abs(s**2/self.hyperbolaA**2 - 1))
        if self.isCylindrical:
            r /= abs(np.cos(phi))
        if self.isClosed:
            return r
        return np.where(abs(phi) < np.pi/2, r, np.ones_like(phi)*1e20)

    def local_n(self, s, phi):
        A2s2 = np.array(s**2 - self.hyperbolaA**2)
        A2s2[A2s2 <= 0] = 1e22
        nr = -self.hyperbolaB / self.hyperbolaA * s / np.sqrt(A2s2)
        return self._normal_from_slope(nr, phi, 1.)


HyperbolicMirror = HyperbolicMirrorParam


class BentFlatMirror(OE):
    """Meridionally bent parabolic cylinder with fixed ends:
    z = (y^2 - limPhysY[0]^2) / (2R) (oes/__init__.py:240-303)."""

    def __init__(self, *args, **kwargs):
        R = kwargs.pop('R', 5.0e6)
        OE.__init__(self, *args, **kwargs)
        self.R = R

    @property
    def R(self):
        return self._RVal

    @R.setter
    def R(self, R):
        if isinstance(R, (list, tuple)):
            self._RVal = self.get_Rmer_from_Coddington(*R)
        elif R is None:
            self._RVal = 1e100
        else:
            self._RVal = R

    def local_z(self, x, y):
        return (y**2 - self.limPhysY[0]**2) / 2.0 / self.R

    def local_n(self, x, y):
        a = 0.
        b = -y / self.R
        c = 1.
        norm = (b**2 + 1)**0.5
        return [a/norm, b/norm, c/norm]

    def _surface_params(self, p, second=False):
        p.surf_kind = _structs.SURF_BENTFLAT
        ok = bool(np.isfinite(self.R)) and 1e-100 < abs(self.R) < 1e100
        p.surf_p[0] = float(self.R)
        p.surf_p[1] = float(self.limPhysY[0]**2)
        p.surf_p[2] = 1.0 / float(self.R) if ok else 0.
        p.surf_p[4] = 1.0 if ok else 0.
        p.asymmetric = 0
        for i, v in enumerate((0., 0., 1., 0., 0., 1.)):
            p.n_const[i] = v


SimpleVCM = BentFlatMirror


class DCM(OE):
    """Double-crystal monochromator with flat crystals (oes/dcm.py)."""

    def __init__(self, *args, **kwargs):
        self.bragg = kwargs.pop('bragg', 0)
        self.cryst1roll = kwargs.pop('cryst1roll', 0)
        self.cryst2roll = kwargs.pop('cryst2roll', 0)
        self.cryst2pitch = kwargs.pop('cryst2pitch', 0)
        self.cryst2finePitch = kwargs.pop('cryst2finePitch', 0)
        self.cryst2perpTransl = kwargs.pop('cryst2perpTransl', 0)
        self.cryst2longTransl = kwargs.pop('cryst2longTransl', 0)
        self.limPhysX2 = kwargs.pop(
            'limPhysX2', [-raycing.maxHalfSizeOfOE, raycing.maxHalfSizeOfOE])
        self.limPhysY2 = kwargs.pop(
            'limPhysY2', [-raycing.maxHalfSizeOfOE, raycing.maxHalfSizeOfOE])
        self.limOptX2 = kwargs.pop('limOptX2', None)
        self.limOptY2 = kwargs.pop('limOptY2', None)
        self.material2 = kwargs.pop('material2', None)
        fixedOffset = kwargs.pop('fixedOffset', None)
        OE.__init__(self, *args, **kwargs)
        if fixedOffset not in [0, None]:
            self.cryst2perpTransl = fixedOffset/2./np.cos(self.bragg)

    def local_n1(self, x, y):
        return self.local_n(x, y)

    def local_n2(self, x, y):
        res = list(self.local_n1(x, y))
        if self.alpha:
            res[1] *= -1
        return res

    def double_reflect(self, beam=None, needLocal=True, fromVacuum1=True,
                       fromVacuum2=True, returnLocalAbsorbed=None, _timing=None):
        """-> (beamGlobal, beamLocal1, beamLocal2), dcm.py:248-354."""
        p1 = self._make_pass(
            self.pitch + self.bragg,
            self.roll + self.positionRoll + self.cryst1roll, self.yaw, self.dx,
            fromVacuum=fromVacuum1, out_to_global=False)
        p2 = self._make_pass(
            -self.pitch - self.bragg + self.cryst2pitch + self.cryst2finePitch,
            self.roll + self.cryst2roll + self.positionRoll, -self.yaw,
            -self.dx, self.cryst2longTransl, -self.cryst2perpTransl,
            fromVacuum=fromVacuum2, is2ndXtal=True, in_is_global=False,
            good_mode=1, out_to_global=True, zero_local_not_entering=True,
            force_lost_out=hasattr(self, 't'))
        # (XRT_HIP_DCM_TWO_PASSES=1: the two separate passes, for comparison)
        if os.environ.get('XRT_HIP_DCM_TWO_PASSES', '') != '1':
            fused = self._run_double(p1, p2, fromVacuum1, fromVacuum2, beam, _timing)
            if fused is not None:
                return fused
        lo1, gb, _ = self._run_pass(p1, self.material, fromVacuum1, beam, beam)
        lo2, gb2, _ = self._run_pass(p2, self.material2, fromVacuum2, gb, beam)
        return gb2, lo1, lo2

    def _run_double(self, p1, p2, fromVacuum1, fromVacuum2, beam, timing=None):
        """Both crystals in one pass over the beam (xrt_hip_double_reflect_f64_dev) when
        the pair qualifies (flat Bragg crystals), else None. -> (gb2, lo1, lo2)"""
        _lib.require_gpu()
        lib = _lib.load()
        dev = torch.device('cuda', torch.cuda.current_device())
        m1 = self._material_struct(self.material, fromVacuum1, dev)
        m2 = self._material_struct(self.material2, fromVacuum2, dev)
        if not lib.xrt_hip_double_reflect_fusable(ctypes.byref(p1), ctypes.byref(m1),
                                                  ctypes.byref(p2), ctypes.byref(m2)):
            return None
        n = beam.nrays
        s_in = beam.to_struct(dev)
        outs = [rs.Beam.empty_like_on_device(beam, dev) for _ in range(3)]
        lo1, lo2, gb2 = outs
        th1 = torch.empty(n, dtype=torch.float64, device=dev)
        th2 = torch.empty(n, dtype=torch.float64, device=dev)
        ws = hipcalls.workspace(dev, lib.xrt_hip_reflect_workspace_bytes(n), 'reflect')
        ms = (ctypes.c_float * 3)() if timing is not None else None
        rc = lib.xrt_hip_double_reflect_f64_dev(
            ctypes.byref(p1), ctypes.byref(m1), ctypes.byref(p2), ctypes.byref(m2),
            ctypes.byref(s_in), ctypes.byref(lo1.to_struct(dev)),
            ctypes.byref(lo2.to_struct(dev)), ctypes.byref(gb2.to_struct(dev)),
            ctypes.c_void_p(th1.data_ptr()), ctypes.c_void_p(th2.data_ptr()),
            ctypes.c_void_p(ws.data_ptr()), ws.numel(),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ms)
        _lib.check(rc, 'xrt_hip_double_reflect_f64_dev')
        if timing is not None:
            timing.update(pass_ms=ms[0], kernel_ms=ms[1], exact_sequence=bool(ms[2]))
        lo1._d['theta'] = th1
        lo2._d['theta'] = th2
        for b in outs:
            for k in rs._SCALAR_ATTRS:
                if k in beam.__dict__:
                    object.__setattr__(b, k, beam.__dict__[k])
            b.parentId = self.uuid
        return gb2, lo1, lo2


class Plate(DCM):
    """A body with two flat surfaces (window, filter): the 2nd 'crystal' of the
    DCM skeleton is the back face at -t (oes/refractive.py:11-235)."""

    def __init__(self, *args, **kwargs):
        t = kwargs.pop('t', 0)
        wedgeAngle = kwargs.pop('wedgeAngle', 0)
        kwargs.setdefault('overEdge', '')
        DCM.__init__(self, *args, **kwargs)
        self.t = t
        self.wedgeAngle = wedgeAngle
        self.cryst2perpTransl = -t
        self.cryst2pitch = wedgeAngle
        # As constructed, the reference's Plate ends up with the back-face limits
        # EQUAL to the front-face ones: its __init__ mirrors x only `if
        # isinstance(self.limPhysX, (list, tuple))`, but the property returns a
        # Limits array, so the else branch runs (refractive.py:37-46). Reproduced,
        # not "fixed" (golden case g2_plate_be has asymmetric x limits).
        self.limPhysX2 = list(self.limPhysX)
        self.limPhysY2 = list(self.limPhysY)
        self.limOptX2 = self.limOptX
        self.limOptY2 = self.limOptY
        self.material2 = self.material
        if self.material is not None and self.material.kind == 'auto':
            self.material.kind = 'plate'

    def double_refract(self, beam=None, needLocal=True, returnLocalAbsorbed=None):
        """-> (beamGlobal, beamLocal1, beamLocal2), refractive.py:171-235."""
        return self.double_reflect(beam=beam, needLocal=needLocal,
                                   fromVacuum1=True, fromVacuum2=False)
