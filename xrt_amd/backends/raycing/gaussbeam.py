"""Analytical Gaussian modes as wave sources (reference sources/geoms.py:538-850):
``GaussianBeam``, ``LaguerreGaussianBeam`` (vortex beams, indices l, p) and
``HermiteGaussianBeam`` (TEM m, n). They have no rays of their own: ``shine(wave=...)``
writes the mode field on the points of a wave made by ``prepare_wave`` of a screen, slit or
optical element. The field per point is one HIP kernel (``xrt_hip_gaussian_beam_f64_dev``);
energies and polarisation are the host sampling shared with the other sources (numpy RNG in
the reference's order)."""
import ctypes
from math import factorial

import numpy as np
import torch

from ... import hipcalls as _hipcalls

from .. import raycing
from ... import _lib, _structs
from . import sources as rs
from .physconsts import CHBAR


class GaussianBeam(object):
    def __init__(self, bl=None, name='', center=(0, 0, 0), w0=0.1, distE='lines',
                 energies=(rs.defaultEnergy,), energyWeights=None, polarization='horizontal',
                 pitch=0, roll=0, yaw=0, totalFlux=None, **kwargs):
        """*w0*: waist size [mm], one number or (horizontal, vertical); energies,
        polarisation, orientation and *totalFlux* as for ``GeometricSource``."""
        rs._enrol_source(self, bl, name or type(self).__name__, kwargs.get('uuid'))
        self.center, self.distE, self.energies, self.energyWeights = \
            center, distE, energies, energyWeights
        self.polarization, self.totalFlux = polarization, totalFlux
        self.w0 = w0
        self.vortex = self.tem = None
        self.pitch, self.roll, self.yaw = (raycing.auto_units_angle(v)
                                           for v in (pitch, roll, yaw))

    @property
    def w0(self):
        return self._w0

    @w0.setter
    def w0(self, size):
        if raycing.is_sequence(size) and len(size) != 2:
            raise ValueError('w0: a number or a pair (horizontal, vertical)')
        self._w0 = size

    def _waist(self, w0=None):
        if w0 is not None:
            return w0
        return self.w0[0] if raycing.is_sequence(self.w0) else self.w0

    def rayleigh_range(self, E, w0=None):
        """k w0^2 / 2 [mm] at photon energy *E* [eV]."""
        return E / CHBAR * 1e7 / 2 * self._waist(w0)**2

    def w(self, y, E=None, yR=None, w0=None):
        """Beam size at the distance *y* from the waist."""
        w0 = self._waist(w0)
        reduced = y / (self.rayleigh_range(E, w0) if yR is None else yR)
        return w0 * (1 + reduced**2)**0.5

    def _announce_energy(self):
        """The beamline's alignment energy, unless the user fixed one: the first line, or the
        middle of a flat band."""
        if self.bl is None:
            return
        if not isinstance(self.bl.alignE, str):
            self.bl._alignE = float(self.bl.alignE)
            return
        lines = np.atleast_1d(self.energies)
        if len(lines) == 0:
            self.bl._alignE = rs.defaultEnergy
        elif self.distE == 'flat' and len(lines) == 2:
            self.bl._alignE = 0.5 * (lines[0] + (lines[1] or lines[0]))
        else:
            self.bl._alignE = lines[0]

    def _mode(self):
        g = _structs.Gauss()
        pair = raycing.is_sequence(self.w0)
        g.w0x, g.w0z = (float(self.w0[0]), float(self.w0[1])) if pair else (float(self.w0),) * 2
        g.astigmatic = int(pair)
        g.clp = 1.
        if self.vortex is not None:
            g.mode, (g.l, g.p) = 1, [int(v) for v in self.vortex]
            g.clp = (factorial(g.p) * 1. / factorial(abs(g.l) + g.p))**0.5
        elif self.tem is not None:
            g.mode, (g.m, g.n) = 2, [int(v) for v in self.tem]
            g.clp = (2**(g.m + g.n) * factorial(g.m) * factorial(g.n))**(-0.5)
        return g

    def shine(self, toGlobal=True, wave=None, accuBeam=None):
        """The mode field on the points of *wave* -> the wave as a beam (global frame)."""
        self._announce_energy()
        if wave is None or not hasattr(wave, 'rDiffr'):
            raise ValueError("run a `prepare_wave` before shine!")
        count = len(wave.rDiffr)
        if self.distE is not None:
            wave.E[:] = accuBeam.E[:] if accuBeam is not None else rs.make_energy(
                self.distE, self.energies, count, filamentBeam=False,
                energyWeights=self.energyWeights)
        rs.make_polarization(self.polarization, wave, count)

        _lib.require_gpu()
        dev = torch.device('cuda', torch.cuda.current_device())

        def up(values):
            return torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64)).to(dev)
        pts = [up(v) for v in (wave.xDiffr, wave.yDiffr, wave.zDiffr, wave.E)]
        cells = up(wave.dS) if np.ndim(wave.dS) else None
        amp = torch.empty(count, dtype=torch.complex128, device=dev)
        dirs = [torch.empty(count, dtype=torch.float64, device=dev) for _ in range(3)]
        mode = self._mode()
        _lib.check(_lib.load().xrt_hip_gaussian_beam_f64_dev(
            ctypes.byref(mode), count, *[t.data_ptr() for t in pts],
            cells.data_ptr() if cells is not None else None,
            float(wave.dS) if cells is None else 0., amp.data_ptr(),
            *[t.data_ptr() for t in dirs],
            _hipcalls.stream_ptr()),
            'xrt_hip_gaussian_beam_f64_dev')
        amp = amp.cpu().numpy()
        for field, factor in (('Es', amp), ('Ep', amp)) + tuple(
                (j, np.abs(amp)**2) for j in ('Jss', 'Jpp', 'Jsp')):
            setattr(wave, field, getattr(wave, field) * factor)
        if np.isscalar(self.totalFlux) and self.totalFlux > 0:
            total = (wave.Jss + wave.Jpp).sum()
            if total > 0:
                wave.sourceWeight = self.totalFlux / total
                wave.seeded, wave.seededI, wave.accepted, wave.acceptedE = count, 1., 1., 1.
        wave.a[:], wave.b[:], wave.c[:] = (t.cpu().numpy() for t in dirs)

        bo = rs.Beam(copyFrom=wave)
        bo.x[:], bo.y[:], bo.z[:] = wave.xDiffr, wave.yDiffr, wave.zDiffr
        bo.path = (wave.xDiffr**2 + wave.yDiffr**2 + wave.zDiffr**2)**0.5
        if self.pitch or self.roll or self.yaw:
            raycing.rotate_beam(bo, pitch=self.pitch, roll=self.roll, yaw=self.yaw)
        if toGlobal:
            raycing.virgin_local_to_global(self.bl, bo, self.center)
        bo.parentId = self.uuid
        return bo


class LaguerreGaussianBeam(GaussianBeam):
    """*vortex* = (l, p): azimuthal index and radial index p >= 0."""

    def __init__(self, *args, vortex=None, **kwargs):
        super().__init__(*args, **kwargs)
        if raycing.is_sequence(self.w0):
            raise ValueError('w0 must be a value, not a sequence')   # one waist for a vortex
        self.vortex = vortex


class HermiteGaussianBeam(GaussianBeam):
    """*TEM* = (m, n): mode orders along x and z."""

    def __init__(self, *args, TEM=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.tem = TEM
