"""``Beam`` and ``GeometricSource`` — host-side mirror of
xrt/backends/raycing/sources/beams.py:46-186 and sources/geoms.py:194-535.

``Beam`` keeps xrt's attribute API (``beam.x``, ``beam.state``, ``beam.Jsp`` ...
are numpy arrays) but every field can also live in HBM as a torch tensor: the
GPU operators (OE.reflect, DCM.double_reflect, Screen.expose, diffract) read
and write the device copies and beams stay resident between elements. A field
is copied to the host only when user code touches the attribute; since the
returned array may then be modified in place, the device copy is dropped and
re-uploaded on the next GPU operation.
"""
import numpy as np
import torch

from .. import raycing
from ... import _structs
from .physconsts import PI2

defaultEnergy = 9.0e3

_F64 = ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp')
_C128 = ('Jsp', 'Es', 'Ep')
_OPT_F64 = ('theta', 'order')
_SCALAR_ATTRS = ('sourceSIGMAx', 'sourceSIGMAz', 'filamentDX', 'filamentDZ',
                 'filamentDtheta', 'filamentDpsi', 'filamentDgamma', 'accepted',
                 'acceptedE', 'seeded', 'seededI', 'sourceWeight')
_ARRAY_FIELDS = set(_F64) | set(_C128) | set(_OPT_F64) | {'state'}
_TORCH_DTYPE = {np.dtype('float64'): torch.float64,
                np.dtype('complex128'): torch.complex128,
                np.dtype('int32'): torch.int32}


def _np_dtype(name):
    if name == 'state':
        return np.int32
    if name in _C128:
        return np.complex128
    return np.float64


class Beam(object):
    """SoA ray container: x, y, z, a, b, c, path, E, Jss, Jpp (f64), Jsp (c128),
    state (i32), optional Es, Ep (c128) — 100 B/ray, 132 B with amplitudes."""

    def __init__(self, nrays=raycing.nrays, copyFrom=None, forceState=False,
                 withNumberOfReflections=False, withAmplitudes=False,
                 xyzOnly=False, bl=None):
        object.__setattr__(self, '_h', {})
        object.__setattr__(self, '_d', {})
        if copyFrom is not None and hasattr(copyFrom, 'a') and hasattr(copyFrom, 'x'):
            if isinstance(copyFrom, Beam):
                for name in copyFrom.array_fields():
                    if name in copyFrom._d:
                        self._d[name] = copyFrom._d[name].clone()
                    else:
                        self._h[name] = np.copy(copyFrom._h[name])
                for k in _SCALAR_ATTRS:     # listOfAttrs, beams.py:95-105
                    if k in copyFrom.__dict__:
                        object.__setattr__(self, k, copyFrom.__dict__[k])
            else:   # any object with xrt's Beam attributes (e.g. the reference's)
                for name in _ARRAY_FIELDS:
                    if hasattr(copyFrom, name):
                        v = getattr(copyFrom, name)
                        if isinstance(v, np.ndarray):
                            self._h[name] = np.array(v, dtype=_np_dtype(name))
                for k in _SCALAR_ATTRS:
                    if hasattr(copyFrom, k):
                        object.__setattr__(self, k, getattr(copyFrom, k))
        else:
            nrays = int(nrays)
            self._h['x'] = np.zeros(nrays)
            self._h['y'] = np.zeros(nrays)
            self._h['z'] = np.zeros(nrays)
            if not xyzOnly:
                self._h['state'] = np.zeros(nrays, dtype=np.int32)
                self._h['a'] = np.zeros(nrays)
                self._h['b'] = np.ones(nrays)
                self._h['c'] = np.zeros(nrays)
                self._h['path'] = np.zeros(nrays)
                self._h['E'] = np.ones(nrays) * defaultEnergy
                self._h['Jss'] = np.ones(nrays)
                self._h['Jpp'] = np.zeros(nrays)
                self._h['Jsp'] = np.zeros(nrays, dtype=complex)
                if withAmplitudes:
                    self._h['Es'] = np.zeros(nrays, dtype=complex)
                    self._h['Ep'] = np.zeros(nrays, dtype=complex)
        if type(forceState) == int:
            self.state[:] = forceState
        if 'parentId' not in self.__dict__:
            object.__setattr__(self, 'parentId', None)

    # ---- attribute protocol ------------------------------------------------
    def __getattr__(self, name):
        if name in _ARRAY_FIELDS:
            h = object.__getattribute__(self, '_h')
            d = object.__getattribute__(self, '_d')
            if name in d:
                # the host copy becomes the master (it may be edited in place): the
                # device tensor and any struct pointing at it are dropped
                h[name] = d.pop(name).cpu().numpy()
                self.__dict__.pop('_struct', None)
            if name in h:
                return h[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in _ARRAY_FIELDS:
            self.__dict__.pop('_struct', None)
            self._d.pop(name, None)
            if isinstance(value, torch.Tensor):
                if value.is_cuda:
                    self._h.pop(name, None)
                    self._d[name] = value
                    return
                value = value.numpy()
            self._h[name] = np.ascontiguousarray(value, dtype=_np_dtype(name))
        else:
            object.__setattr__(self, name, value)

    def __delattr__(self, name):
        if name in _ARRAY_FIELDS:
            self._h.pop(name, None)
            self._d.pop(name, None)
        else:
            object.__delattr__(self, name)

    def filter_by_index(self, indarr):
        """Keeps the rays selected by *indarr* (sources/beams.py:296-318)."""
        for name in self.array_fields():
            setattr(self, name, getattr(self, name)[indarr])

    def array_fields(self):
        return [n for n in (_F64 + _C128 + _OPT_F64 + ('state',))
                if n in self._h or n in self._d]

    @property
    def nrays(self):
        for store in (self._d, self._h):
            if 'x' in store:
                return int(store['x'].shape[0])
        return 0

    def __len__(self):
        return self.nrays

    def has_amplitudes(self):
        return ('Es' in self._h or 'Es' in self._d)

    # ---- device side ---------------------------------------------------------
    def dev(self, name, device=None):
        """The field as a CUDA/HIP tensor (uploaded on first use)."""
        if name in self._d:
            return self._d[name]
        if name not in self._h:
            raise AttributeError(name)
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        h = np.ascontiguousarray(self._h[name], dtype=_np_dtype(name))
        t = torch.from_numpy(h).to(device)
        self._d[name] = t
        self.__dict__.pop('_struct', None)      # a new tensor: cached pointers are stale
        # the host copy stays valid until a kernel overwrites the tensor; the
        # operators below always write into NEW beams, never into their input
        return t

    def peek(self, name):
        """Host copy of a field WITHOUT invalidating the device copy (read-only
        use)."""
        if name in self._d:
            return self._d[name].cpu().numpy()
        return self._h[name]

    @classmethod
    def empty_like_on_device(cls, other, device):
        """New beam with uninitialised device arrays of other's shape."""
        b = cls.__new__(cls)
        object.__setattr__(b, '_h', {})
        object.__setattr__(b, '_d', {})
        n = other.nrays
        names = list(_F64) + ['Jsp', 'state']
        if other.has_amplitudes():
            names += ['Es', 'Ep']
        for name in names:
            b._d[name] = torch.empty(n, dtype=_TORCH_DTYPE[np.dtype(_np_dtype(name))],
                                     device=device)
        object.__setattr__(b, 'parentId', None)
        return b

    def to_struct(self, device=None):
        """ctypes xrt_hip_beam with device pointers (keeps the tensors alive
        through the returned struct's ``_keep``). Cached until a field changes."""
        cached = self.__dict__.get('_struct')
        if cached is not None and not self._h_dirty():
            return cached
        s = _structs.Beam()
        keep = []
        s.n = self.nrays
        for cname, name in (('x', 'x'), ('y', 'y'), ('z', 'z'), ('a', 'a'),
                            ('b', 'b'), ('c', 'c'), ('path', 'path'), ('E', 'E'),
                            ('Jss', 'Jss'), ('Jpp', 'Jpp'), ('Jsp_ri', 'Jsp'),
                            ('state', 'state')):
            t = self.dev(name, device)
            keep.append(t)
            setattr(s, cname, t.data_ptr())
        if self.has_amplitudes():
            for cname, name in (('Es_ri', 'Es'), ('Ep_ri', 'Ep')):
                t = self.dev(name, device)
                keep.append(t)
                setattr(s, cname, t.data_ptr())
        else:
            s.Es_ri = None
            s.Ep_ri = None
        s._keep = keep
        object.__setattr__(self, '_struct', s)
        return s

    def _h_dirty(self):
        """True if some array field has no device copy (host access drops it)."""
        need = ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp', 'state')
        if any(n not in self._d for n in need):
            return True
        return self.has_amplitudes() and ('Es' not in self._d or 'Ep' not in self._d)


def copy_beam(beamTo, beamFrom, indarr, includeState=False, includeJspEsp=True):
    """Host-side copy_beam (sources/beams.py:409-445), for glue code."""
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E'):
        getattr(beamTo, f)[indarr] = getattr(beamFrom, f)[indarr]
    if includeState:
        beamTo.state[indarr] = beamFrom.state[indarr]
    if includeJspEsp:
        for f in ('Jss', 'Jpp', 'Jsp'):
            getattr(beamTo, f)[indarr] = getattr(beamFrom, f)[indarr]
        if hasattr(beamFrom, 'Es') and hasattr(beamTo, 'Es'):
            beamTo.Es[indarr] = beamFrom.Es[indarr]
            beamTo.Ep[indarr] = beamFrom.Ep[indarr]


def rotate_coherency_matrix(beam, indarr, roll):
    """sources/beams.py:448-479 on host arrays (wave post-processing glue)."""
    c = np.cos(roll)
    s = np.sin(roll)
    c2 = c**2
    s2 = s**2
    cs = c * s
    JssN = beam.Jss[indarr]*c2 + beam.Jpp[indarr]*s2 +\
        2*beam.Jsp[indarr].real*cs
    JppN = beam.Jss[indarr]*s2 + beam.Jpp[indarr]*c2 -\
        2*beam.Jsp[indarr].real*cs
    JspN = (beam.Jpp[indarr]-beam.Jss[indarr])*cs +\
        beam.Jsp[indarr].real*(c2-s2) + beam.Jsp[indarr].imag*1j
    return JssN, JppN, JspN


# ---------------------------------------------------------------------------
# GeometricSource (sources/geoms.py). Sampling is host-side numpy with the
# GLOBAL np.random state and the same call order as the reference, so a script
# that seeds np.random gets the very same rays.
# ---------------------------------------------------------------------------
def make_energy(distE, energies, nrays, filamentBeam=False, energyWeights=None):
    locnrays = 1 if filamentBeam else int(nrays)
    eArr = np.atleast_1d(energies)
    if distE == 'normal':
        eMean = eArr[0]
        eSigma = eArr[1] if len(eArr) == 2 else 0
        E = np.random.normal(
            eMean, 0 if abs(eSigma) > 0.1*abs(eMean) else abs(eSigma), locnrays)
    elif distE == 'flat':
        eMin = eArr[0]
        eMax = eArr[1] or eArr[0] if len(eArr) == 2 else eArr[0]
        E = np.random.uniform(eMin, eMax, locnrays)
    elif distE == 'lines':
        if 0 in eArr:
            eArr = eArr[eArr > 0]
        if energyWeights is not None and len(eArr) == len(np.atleast_1d(energyWeights)):
            E = np.random.choice(eArr, size=locnrays, p=np.atleast_1d(energyWeights))
        else:
            E = np.random.choice(eArr, size=locnrays)
    else:
        raise ValueError('unknown distE')
    return E


def make_polarization(polarization, bo, nrays=raycing.nrays):
    """Coherency matrix (and Es, Ep) of the generated rays, geoms.py:63-179."""
    def _fill_beam(Jss, Jpp, Jsp, Es, Ep):
        bo.Jss.fill(Jss)
        bo.Jpp.fill(Jpp)
        bo.Jsp.fill(Jsp)
        if hasattr(bo, 'Es'):
            bo.Es.fill(Es)
            if isinstance(Ep, str):
                bo.Ep[:] = np.random.uniform(size=int(nrays)) * 2**(-0.5)
            else:
                bo.Ep.fill(Ep)

    def _fill_linear(angle):
        Es = np.cos(angle)
        Ep = np.sin(angle)
        _fill_beam(Es*Es, Ep*Ep, Es*Ep, Es, Ep)

    if polarization is None:
        _fill_beam(0.5, 0.5, 0, 2**(-0.5), 'random phase')
    elif isinstance(polarization, (tuple, list, np.ndarray)):
        if len(polarization) != 4:
            raise ValueError('wrong coherency matrix: must be a 4-sequence!')
        bo.Jss.fill(polarization[0])
        bo.Jpp.fill(polarization[1])
        bo.Jsp.fill(polarization[2] + 1j*polarization[3])
    elif isinstance(polarization, str):
        pol = polarization.lower()
        if pol.startswith('un'):
            _fill_beam(0.5, 0.5, 0, 2**(-0.5), 'random phase')
        elif pol.startswith('r'):
            _fill_beam(0.5, 0.5, 0.5j, 2**(-0.5), -1j * 2**(-0.5))
        elif pol.startswith('l'):
            _fill_beam(0.5, 0.5, -0.5j, 2**(-0.5), 1j * 2**(-0.5))
        elif pol.startswith('h'):
            _fill_linear(0.)
        elif pol.startswith('v'):
            _fill_linear(np.pi / 2.)
        else:
            try:
                if pol.endswith('rad'):
                    angle = float(pol[:-3])
                else:
                    angle = float(pol) * np.pi / 180.
            except ValueError:
                raise ValueError('wrong polarization!')
            _fill_linear(angle)
    else:
        _fill_linear(float(polarization) * np.pi / 180.)


class GeometricSource(object):
    """Rays with origin, divergence and energy sampled from simple laws."""

    def __init__(self, bl=None, name='', center=(0, 0, 0), nrays=raycing.nrays,
                 distx='normal', dx=0.32, disty=None, dy=0, distz='normal',
                 dz=0.018, distxprime='normal', dxprime=1e-3,
                 distzprime='normal', dzprime=1e-4, distE='lines',
                 energies=(defaultEnergy,), energyWeights=None,
                 polarization='horizontal', filamentBeam=False,
                 uniformRayDensity=False, pitch=0, roll=0, yaw=0, **kwargs):
        self.bl = bl
        if bl is not None and self not in bl.sources:
            bl.sources.append(self)
            self.ordinalNum = len(bl.sources)
        self.name = name or 'GeometricSource'
        self.uuid = kwargs.get('uuid', raycing.new_uuid())
        if bl is not None:
            bl.oesDict[self.uuid] = [self, 0]
        self.center = center
        self.nrays = int(nrays)
        self.distx, self.dx = distx, dx
        self.disty, self.dy = disty, dy
        self.distz, self.dz = distz, dz
        self.distxprime, self.dxprime = distxprime, dxprime
        self.distzprime, self.dzprime = distzprime, dzprime
        self.distE = distE
        self.energies = energies
        self.energyWeights = energyWeights
        self.polarization = polarization
        self.filamentBeam = filamentBeam
        self.uniformRayDensity = uniformRayDensity
        self.pitch, self.roll, self.yaw = pitch, roll, yaw

    def _apply_distribution(self, axis, distaxis, daxis, bo=None):
        if distaxis == 'normal':
            if self.uniformRayDensity:
                daxisArr = np.atleast_1d(daxis)
                if len(daxisArr) < 2:
                    sigma = daxisArr[0]
                    cutLim = 5 * abs(sigma)
                else:
                    sigma = daxisArr[0]
                    cutLim = daxisArr[-1]
                axis[:] = np.random.uniform(-cutLim, cutLim, self.nrays)
                amp = np.exp(-axis**2 / sigma**2 / 2) /\
                    PI2**0.5 / sigma * 2 * cutLim
                bo.Jss *= amp
                bo.Jpp *= amp
                bo.Jsp *= amp
                amp = amp**0.5
                bo.Es *= amp
                bo.Ep *= amp
            else:
                sigma = daxis[0] if isinstance(daxis, (list, tuple)) else daxis
                try:
                    axis[:] = np.random.normal(0, sigma, self.nrays)
                except ValueError:
                    axis[:] = np.zeros(self.nrays)
        elif distaxis == 'flat':
            if raycing.is_sequence(daxis):
                aMin, aMax = daxis[0], daxis[1]
            else:
                if daxis <= 0:
                    return
                aMin, aMax = -daxis*0.5, daxis*0.5
            axis[:] = np.random.uniform(aMin, aMax, self.nrays)

    def _set_annulus(self, axis1, axis2, rMin, rMax, phiMin, phiMax):
        if rMax > rMin:
            A = 2. / (rMax**2 - rMin**2)
            r = np.sqrt(2*np.random.uniform(0, 1, self.nrays)/A + rMin**2)
        else:
            r = rMax
        phi = np.random.uniform(phiMin, phiMax, self.nrays)
        axis1[:] = r * np.cos(phi)
        axis2[:] = r * np.sin(phi)

    def _pair(self, bo, n1, n2, dist1, d1, dist2, d2):
        isAnnulus = False
        if (dist1 == 'annulus') or (dist2 == 'annulus'):
            isAnnulus = True
            if raycing.is_sequence(d1):
                rMin, rMax = d1
            else:
                isAnnulus = False
            if raycing.is_sequence(d2):
                phiMin, phiMax = d2
            else:
                phiMin, phiMax = 0, PI2
        if isAnnulus:
            self._set_annulus(getattr(bo, n1), getattr(bo, n2), rMin, rMax,
                              phiMin, phiMax)
        else:
            self._apply_distribution(getattr(bo, n1), dist1, d1, bo)
            self._apply_distribution(getattr(bo, n2), dist2, d2, bo)

    def shine(self, toGlobal=True, withAmplitudes=False, accuBeam=None):
        """The source beam, in the global frame if *toGlobal*
        (geoms.py:420-535)."""
        if self.uniformRayDensity:
            withAmplitudes = True
        bo = Beam(self.nrays, withAmplitudes=withAmplitudes)
        bo.state[:] = 1
        make_polarization(self.polarization, bo, self.nrays)
        self._apply_distribution(bo.y, self.disty, self.dy, bo)
        self._pair(bo, 'x', 'z', self.distx, self.dx, self.distz, self.dz)
        self._pair(bo, 'a', 'c', self.distxprime, self.dxprime, self.distzprime,
                   self.dzprime)
        ac = bo.a**2 + bo.c**2
        if sum(ac > 1) > 0:
            bo.b[:] = (ac + 1)**0.5
            bo.a[:] /= bo.b
            bo.c[:] /= bo.b
            bo.b[:] = 1.0 / bo.b
        else:
            bo.b[:] = (1 - ac)**0.5
        if self.distE is not None:
            if accuBeam is None:
                bo.E[:] = make_energy(self.distE, self.energies, self.nrays,
                                      self.filamentBeam, self.energyWeights)
            else:
                bo.E[:] = accuBeam.E[:]
        if self.pitch or self.roll or self.yaw:
            raycing.rotate_beam(bo, pitch=self.pitch, roll=self.roll,
                                yaw=self.yaw)
        if toGlobal:
            raycing.virgin_local_to_global(self.bl, bo, self.center)
        bo.parentId = self.uuid
        return bo

from .undulator import Undulator  # noqa: E402,F401  (needs Beam from this module)
