"""``Screen`` — host-side mirror of xrt/backends/raycing/screens.py:21-365.
``expose`` runs the streaming HIP kernel (csrc/screen.hip) on a device-resident
beam; ``prepare_wave`` builds the receiving mesh for ``waves.diffract``."""
import ctypes

import numpy as np
import torch

from .. import raycing
from ... import _lib, _structs
from . import sources as rs


class Screen(object):
    def __init__(self, bl=None, name='', center=[0, 0, 0], x='auto', z='auto',
                 compressX=None, compressZ=None, **kwargs):
        self.bl = bl
        if bl is not None:
            if self not in bl.screens:
                bl.screens.append(self)
                self.ordinalNum = len(bl.screens)
                self.lostNum = -self.ordinalNum - 2000      # screens.py:77
        else:
            self.ordinalNum = 1
            self.lostNum = -2001
        self.name = name or 'Screen{0}'.format(self.ordinalNum)
        self.uuid = kwargs.get('uuid', raycing.new_uuid())
        if bl is not None:
            bl.oesDict[self.uuid] = [self, 1]
        self.center = center
        self.compressX = compressX
        self.compressZ = compressZ
        self.set_orientation(x, z)
        self.footprint = []

    def set_orientation(self, x=None, z=None):
        if isinstance(x, str):
            x = None
        if isinstance(z, str):
            z = None
        self.x, self.y, self.z = raycing.xyz_from_xz(self, x, z)

    def local_to_global(self, x=0, y=0, z=0, **kwargs):
        return tuple(raycing.along_basis((self.x, self.y, self.z), x, y, z,
                                         self.center))

    def expose(self, beam=None, onlyPositivePath=False):
        """*beam* in the global frame -> the image in the screen's local frame
        (screens.py:226-302)."""
        _lib.require_gpu()
        lib = _lib.load()
        dev = torch.device('cuda', torch.cuda.current_device())
        s = _structs.Screen()
        for i in range(3):
            s.center[i] = float(self.center[i])
            s.ex[i] = float(self.x[i])
            s.ey[i] = float(self.y[i])
            s.ez[i] = float(self.z[i])
        s.compress_x = float(self.compressX) if self.compressX else 0.
        s.compress_z = float(self.compressZ) if self.compressZ else 0.
        s.lost_num = int(self.lostNum)
        s.only_positive_path = 1 if onlyPositivePath else 0
        s_in = beam.to_struct(dev)
        blo = rs.Beam.empty_like_on_device(beam, dev)
        s_out = blo.to_struct(dev)
        _lib.check(lib.xrt_hip_screen_expose_f64_dev(
            ctypes.byref(s), ctypes.byref(s_in), ctypes.byref(s_out),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
            'xrt_hip_screen_expose_f64_dev')
        for k in rs._SCALAR_ATTRS:
            if k in beam.__dict__:
                object.__setattr__(blo, k, beam.__dict__[k])
        return blo

    def prepare_wave(self, prevOE, dim1, dim2, dy=0, rw=None, condition=None):
        """Receiving mesh for wave propagation (screens.py:304-365): a
        meshgrid of local x (dim1) and z (dim2)."""
        if rw is None:
            from . import waves as rw
        d1s, d2s = np.meshgrid(dim1, dim2)
        d1s = d1s.flatten()
        d2s = d2s.flatten()
        if hasattr(dim1, '__getitem__') and hasattr(dim2, '__getitem__'):
            try:
                dS = (dim1[1] - dim1[0]) * (dim2[1] - dim2[0])
            except IndexError:
                dS = 1.
        else:
            dS = 1.
        if condition is not None:
            d1s, d2s = condition(d1s, d2s)
        nrays = len(d1s)
        xglo, yglo, zglo = self.local_to_global(x=d1s, z=d2s)
        wave = rs.Beam(nrays=nrays, forceState=1, withAmplitudes=True)
        wave.x[:] = d1s
        wave.y[:] = np.zeros_like(d1s) + dy
        wave.z[:] = d2s
        wave.dS = dS
        wave.toOE = self
        wave.area = (np.ones_like(d1s) * dS).sum()
        wave.parentId = prevOE.uuid
        return rw.prepare_wave(prevOE, wave, xglo, yglo+dy, zglo)
