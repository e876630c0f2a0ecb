"""``RectangularAperture`` — host-side mirror of
xrt/backends/raycing/apertures.py:29-499: blade geometry, local frame,
``propagate`` (streaming HIP kernel on device-resident beams) and
``prepare_wave`` for the wave path."""
import ctypes

import numpy as np
import torch

from .. import raycing
from ... import _lib, _structs
from . import sources as rs

_BLADE_ORDER = ('left', 'right', 'bottom', 'top')


class RectangularAperture(object):
    def __init__(self, bl=None, name='', center=[0, 0, 0],
                 kind=('left', 'right', 'bottom', 'top'),
                 opening=(-10, 10, -10, 10), x='auto', z='auto', alarmLevel=None,
                 blades=None, **kwargs):
        self.bl = bl
        if bl is not None:
            if self not in bl.slits:
                bl.slits.append(self)
                self.ordinalNum = len(bl.slits)
                self.lostNum = -self.ordinalNum - 1000     # apertures.py:80
        else:
            self.ordinalNum = 1
            self.lostNum = -1001
        self.name = name or 'Aperture{0}'.format(self.ordinalNum)
        self.uuid = kwargs.get('uuid', raycing.new_uuid())
        if bl is not None:
            bl.oesDict[self.uuid] = [self, 1]
        self.center = center
        self.limOptX = [-raycing.maxHalfSizeOfOE, raycing.maxHalfSizeOfOE]
        self.limOptY = [-raycing.maxHalfSizeOfOE, raycing.maxHalfSizeOfOE]
        if blades is None:
            kinds = [kind] if isinstance(kind, str) else list(kind)
            opens = list(opening) if raycing.is_sequence(opening) else [opening]
            blades = {k: v for k, v in zip(kinds, opens) if v is not None}
        self.blades = {k: blades[k] for k in _BLADE_ORDER if k in blades}
        for akind, d in self.blades.items():
            td = float(d)
            if akind.startswith('l'):
                self.limOptX[0] = td
            elif akind.startswith('r'):
                self.limOptX[1] = td
            elif akind.startswith('b'):
                self.limOptY[0] = td
            elif akind.startswith('t'):
                self.limOptY[1] = td
        self.isBeamStop = False
        self.alarmLevel = alarmLevel
        if isinstance(x, str):
            x = None
        if isinstance(z, str):
            z = None
        self.xyz = raycing.xyz_from_xz(self, x, z)
        self.x, self.y, self.z = self.xyz

    @property
    def kind(self):
        return list(self.blades.keys())

    @property
    def opening(self):
        return list(self.blades.values())

    def local_to_global(self, glo, returnBeam=False, **kwargs):
        """Beam in the aperture's frame -> global frame, in place
        (reference: apertures.py:436-457)."""
        basis = (self.x, self.y, self.z)
        glo.x, glo.y, glo.z = raycing.along_basis(basis, glo.x, glo.y, glo.z,
                                                  self.center)
        glo.a, glo.b, glo.c = raycing.along_basis(basis, glo.a, glo.b, glo.c)

    def propagate(self, beam=None, needNewGlobal=False):
        """Rays stopped by the blades get state ``lostNum`` — in *beam* itself
        too, as in the reference (apertures.py:334-413). Returns the beam in the
        aperture's local frame (and the new global beam if *needNewGlobal*)."""
        _lib.require_gpu()
        lib = _lib.load()
        dev = torch.device('cuda', torch.cuda.current_device())
        a = _structs.Aperture()
        for i in range(3):
            a.center[i] = float(self.center[i])
            a.ex[i] = float(self.x[i])
            a.ey[i] = float(self.y[i])
            a.ez[i] = float(self.z[i])
        a.sin_az = self.bl.sinAzimuth if self.bl is not None else 0.
        a.cos_az = self.bl.cosAzimuth if self.bl is not None else 1.
        mask = 0
        for bit, key in enumerate(_BLADE_ORDER):
            if key in self.blades:
                mask |= 1 << bit
                a.blade[bit] = float(self.blades[key])
        a.blade_mask = mask
        a.is_beam_stop = 1 if self.isBeamStop else 0
        a.lost_num = int(self.lostNum)
        s_in = beam.to_struct(dev)
        lo = rs.Beam.empty_like_on_device(beam, dev)
        s_lo = lo.to_struct(dev)
        glo = s_glo = None
        if needNewGlobal:
            glo = rs.Beam.empty_like_on_device(beam, dev)
            s_glo = glo.to_struct(dev)
        _lib.check(lib.xrt_hip_aperture_propagate_f64_dev(
            ctypes.byref(a), ctypes.byref(s_in), ctypes.byref(s_lo),
            ctypes.byref(s_glo) if s_glo is not None else None,
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
            'xrt_hip_aperture_propagate_f64_dev')
        beam._h.pop('state', None)       # the kernel updated beam.state in HBM
        for b in (lo, glo):
            if b is not None:
                for k in rs._SCALAR_ATTRS:
                    if k in beam.__dict__:
                        object.__setattr__(b, k, beam.__dict__[k])
        if needNewGlobal:
            return glo, lo
        return lo

    def prepare_wave(self, prevOE, nrays, rw=None):
        """*nrays* samples uniformly random over the slit area
        (apertures.py:467-499); uses the global np.random state like xrt."""
        if rw is None:
            from . import waves as rw
        nrays = int(nrays)
        wave = rs.Beam(nrays=nrays, forceState=1, withAmplitudes=True)
        uv = np.random.rand(nrays, 2)             # one (nrays, 2) draw, like xrt
        width = self.limOptX[1] - self.limOptX[0]
        height = self.limOptY[1] - self.limOptY[0]
        wave.x[:] = uv[:, 0] * width + self.limOptX[0]
        wave.z[:] = uv[:, 1] * height + self.limOptY[0]
        wave.area = width * height
        wave.dS = wave.area / nrays
        wave.toOE = self
        wave.parentId = self.uuid
        glo = rs.Beam(copyFrom=wave)
        self.local_to_global(glo)
        rw.prepare_wave(prevOE, wave, glo.x, glo.y, glo.z)
        return wave
