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

from ... import hipcalls as _hipcalls

from .. import raycing
from ... import _structs
from .physconsts import PI2

defaultEnergy = 9.0e3

_F64 = ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp')
_C128 = ('Jsp', 'Es', 'Ep')
_OPT_F64 = ('theta', 'phi', 'order', 'xDiffr', 'yDiffr', 'zDiffr', 'rDiffr',
            # OE.multiple_reflect: the points of greatest elevation between two bounces, the
            # impact points in the parametric coordinates (sources/beams.py:80-91)
            'elevationD', 'elevationX', 'elevationY', 'elevationZ', 's', 'r')
_OPT_I32 = ('nRefl',)           # number of reflections (multiple_reflect)
# accumulated Kirchhoff integrals of a receiving wave (waves.diffract)
_OPT_C128 = ('EsAcc', 'EpAcc', 'aEacc', 'bEacc', 'cEacc')
_SCALAR_ATTRS = ('sourceSIGMAx', 'sourceSIGMAz', 'filamentDX', 'filamentDZ',
                 'filamentDtheta', 'filamentDpsi', 'filamentDgamma', 'accepted',
                 'acceptedE', 'seeded', 'seededI', 'sourceWeight')
_ALWAYS = frozenset(_F64 + ('Jsp', 'state'))        # what every beam holds
_ARRAY_FIELDS = set(_F64) | set(_C128) | set(_OPT_F64) | set(_OPT_C128) | set(_OPT_I32) | {'state'}
_TORCH_DTYPE = {np.dtype('float64'): torch.float64,
                np.dtype('complex128'): torch.complex128,
                np.dtype('int32'): torch.int32}


def _np_dtype(name):
    if name == 'state' or name in _OPT_I32:
        return np.int32
    if name in _C128 or name in _OPT_C128:
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
        # (a Beam is recognised by its type: asking it for .a / .x would pull those two arrays
        # of a device-resident beam to the host -- 3 ms per 1e7 rays, and up again later)
        if isinstance(copyFrom, Beam) or (
                copyFrom is not None and not isinstance(copyFrom, str) and
                hasattr(copyFrom, 'a') and hasattr(copyFrom, 'x')):
            if isinstance(copyFrom, Beam):
                for name in copyFrom.array_fields():
                    if name in copyFrom._d:
                        self._d[name] = copyFrom._d[name].clone()
                    else:
                        self._h[name] = np.copy(copyFrom._h[name])
                for k in _SCALAR_ATTRS:     # listOfAttrs, beams.py:95-105
                    if k in copyFrom.__dict__:
                        object.__setattr__(self, k, copyFrom.__dict__[k])
                if '_stopped_by' in copyFrom.__dict__:
                    self.__dict__['_stopped_by'] = set(copyFrom.__dict__['_stopped_by'])
            else:   # any object with xrt's Beam attributes (e.g. the reference's)
                for name in _ARRAY_FIELDS:
                    if hasattr(copyFrom, name):
                        v = getattr(copyFrom, name)
                        if isinstance(v, np.ndarray):
                            self._h[name] = np.array(v, dtype=_np_dtype(name))
                for k in _SCALAR_ATTRS:
                    if hasattr(copyFrom, k):
                        object.__setattr__(self, k, getattr(copyFrom, k))
        elif isinstance(copyFrom, str):
            self._load(copyFrom, bl)
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

    def _load(self, path, bl=None):
        """A beam written by ``export_beam`` (numpy 'npy' dictionary, Matlab 'mat' or
        pickle -- by this class or by the reference's, beams.py:122-149). The elements it
        names (``fromOE``, ``toOE``, ``parentId``) are looked up on *bl*, if given."""
        if path.endswith('mat'):
            import scipy.io
            record = {k: (np.squeeze(v) if isinstance(v, np.ndarray) else v)
                      for k, v in scipy.io.loadmat(path).items() if not k.startswith('__')}
        elif path.endswith('npy'):
            record = np.load(path, allow_pickle=True).item()
        else:
            import pickle
            with open(path, 'rb') as f:
                record = pickle.load(f)
        for key, value in record.items():
            if key in _ARRAY_FIELDS:
                self._h[key] = np.array(value, dtype=_np_dtype(key))
            else:
                if isinstance(value, np.ndarray) and value.size == 1:
                    value = value.item()       # what a 'mat' file makes of scalars and names
                if key in ('fromOE', 'toOE', 'parentId') and bl is not None:
                    try:
                        value = getattr(bl, 'oesDict', {}).get(value, [value])[0]
                    except (KeyError, TypeError):
                        pass                   # not a known element: the name stays as it is
                object.__setattr__(self, key, value)

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
        """Keeps the rays selected by *indarr* (sources/beams.py:296-318). With a mask that
        lives on the GPU the arrays are filtered there and stay there."""
        if isinstance(indarr, torch.Tensor) and indarr.is_cuda:
            # a mask is turned into indices ONCE (one scan, one sync) and every array is
            # gathered with them (one kernel each); index tensors are taken as they are, so
            # that two beams filtered alike share the scan
            index = torch.nonzero(indarr).squeeze(1) if indarr.dtype == torch.bool else indarr
            for name in self.array_fields():
                kept = self.dev(name, indarr.device).index_select(0, index)
                self._h.pop(name, None)
                self._d[name] = kept
            self.__dict__.pop('_struct', None)
            return
        for name in self.array_fields():
            setattr(self, name, getattr(self, name)[indarr])

    def filter_good(self):
        return self.filter_by_index(self.state == 1)

    def _stored(self, name):
        """The array as it is held: device tensor or host array (no transfer)."""
        return self._d[name] if name in self._d else self._h[name]

    def concatenate(self, beam):
        """Appends the rays of *beam* (several sources feeding one beamline,
        sources/beams.py:230-294): every array both beams carry, amplitudes included; two
        different scalar source weights become per-ray weights. Arrays that both beams hold
        on the GPU are joined there."""
        mine, theirs = self.nrays, beam.nrays
        for name in self.array_fields():
            if name not in beam.array_fields():
                if name not in _F64 + _C128 + ('state',):
                    delattr(self, name)
                continue
            a, b = self._stored(name), beam._stored(name)
            if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
                setattr(self, name, torch.cat((a, b)))
            else:
                setattr(self, name, np.concatenate((getattr(self, name), getattr(beam, name))))
        if hasattr(self, 'sourceWeight') and hasattr(beam, 'sourceWeight'):
            w0, w1 = self.sourceWeight, beam.sourceWeight
            if np.ndim(w0) or np.ndim(w1) or w0 != w1:
                self.sourceWeight = np.concatenate((np.broadcast_to(w0, mine),
                                                    np.broadcast_to(w1, theirs)))
        return self

    def replace_by_index(self, indarr, beam):
        """Rays *indarr* take their arrays from the same rays of *beam*."""
        for name in self.array_fields():
            if name in beam.array_fields():
                getattr(self, name)[indarr] = getattr(beam, name)[indarr]
        return self

    def absorb_intensity(self, inBeam, sign=1):
        """The coherency matrix becomes what was lost on the way from *inBeam*."""
        for name in ('Jss', 'Jpp', 'Jsp'):
            setattr(self, name, (inBeam._stored(name) - self._stored(name)) * sign
                    if isinstance(inBeam._stored(name), type(self._stored(name)))
                    else (getattr(inBeam, name) - getattr(self, name)) * sign)
        self.displayAsAbsorbedPower = True

    def add_wave(self, wave, sign=1):
        """Coherent sum with another field on the same points."""
        self.Es = self.Es + sign*wave.Es
        self.Ep = self.Ep + sign*wave.Ep
        self.Jss = (self.Es * self.Es.conjugate()).real
        self.Jpp = (self.Ep * self.Ep.conjugate()).real
        self.Jsp = self.Es * self.Ep.conjugate()

    def project_energy_to_band(self, EnewMin, EnewMax):
        """The energies stretched linearly onto [EnewMin, EnewMax]."""
        lo, hi = np.min(self.E), np.max(self.E)
        if lo < hi:
            self.E = EnewMin + (self.E - lo) / (hi - lo) * (EnewMax - EnewMin)

    def make_uniform_energy_band(self, EnewMin, EnewMax):
        self.E = np.random.uniform(EnewMin, EnewMax, self.nrays)

    def diffract(self, wave):
        from . import waves as rw
        return rw.diffract(self, wave)

    def export_beam(self, fileName, fformat='npy'):
        """The beam's arrays and scalars as one dictionary in a numpy ('npy'), Matlab
        ('mat') or pickle file."""
        record = {name: np.asarray(getattr(self, name)) for name in self.array_fields()}
        record.update({k: v for k, v in self.__dict__.items()
                       if not k.startswith('_') and k not in ('fromOE', 'toOE')})
        for key in ('fromOE', 'toOE'):
            if key in self.__dict__:
                record[key] = getattr(self.__dict__[key], 'name', None)
        kind = str(fformat).lower()
        if kind in ('npy', 'np', 'numpy'):
            np.save(fileName if fileName.endswith('npy') else fileName + '.npy', record)
        elif kind in ('mat', 'matlab'):
            import scipy.io
            scipy.io.savemat(fileName if fileName.endswith('mat') else fileName + '.mat',
                             {k: v for k, v in record.items() if v is not None})
        else:
            import pickle
            with open(fileName if fileName.endswith('pickle') else fileName + '.pickle',
                      'wb') as f:
                pickle.dump(record, f, protocol=2)

    def array_fields(self):
        return [n for n in (_F64 + _C128 + _OPT_F64 + _OPT_C128 + _OPT_I32 + ('state',))
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
        from ... import graphs
        graphs.refuse('upload of the host array %r of a beam' % name)
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
    def on_device(cls, nrays, device, withAmplitudes=True, state=1):
        """What ``Beam(nrays=..., forceState=state, withAmplitudes=...)`` makes (zero positions,
        direction along y, unit Jss, `defaultEnergy`), made on the GPU: nothing to upload."""
        b = cls.__new__(cls)
        object.__setattr__(b, '_h', {})
        object.__setattr__(b, '_d', {})
        f64 = lambda v: torch.full((int(nrays),), float(v), dtype=torch.float64,  # noqa: E731
                                   device=device)
        c128 = lambda: torch.zeros(int(nrays), dtype=torch.complex128, device=device)  # noqa: E731
        for name, value in (('x', 0.), ('y', 0.), ('z', 0.), ('a', 0.), ('b', 1.), ('c', 0.),
                            ('path', 0.), ('E', defaultEnergy), ('Jss', 1.), ('Jpp', 0.)):
            b._d[name] = f64(value)
        b._d['Jsp'] = c128()
        b._d['state'] = torch.full((int(nrays),), int(state), dtype=torch.int32, device=device)
        if withAmplitudes:
            b._d['Es'], b._d['Ep'] = c128(), c128()
        object.__setattr__(b, 'parentId', None)
        return b

    @classmethod
    def empty_like_on_device(cls, other, device):
        """New beam with uninitialised device arrays of other's shape."""
        return cls.empty_on_device(other.nrays, device, other.has_amplitudes())

    @classmethod
    def empty_on_device(cls, nrays, device, withAmplitudes=False):
        """New beam of *nrays* rays with uninitialised device arrays: three allocations (the ten
        f64 arrays are rows of one block, rows 512-B aligned like separate allocations would
        be; the complex ones of another; the states) and the xrt_hip_beam record filled from
        their addresses -- an element call makes two or three of these, and fifteen
        ``torch.empty`` + as many ``data_ptr`` were a quarter of its host time."""
        b = cls.__new__(cls)
        n = int(nrays)
        row = (n + 63) // 64 * 64
        f = torch.empty((len(_F64), row), dtype=torch.float64, device=device)
        c = torch.empty((3 if withAmplitudes else 1, row), dtype=torch.complex128, device=device)
        state = torch.empty(n, dtype=torch.int32, device=device)
        whole = n == row
        d = dict(zip(_F64, (f if whole else f[:, :n]).unbind(0)))
        rows = (c if whole else c[:, :n]).unbind(0)
        d['Jsp'] = rows[0]
        d['state'] = state
        fb, cb, step = f.data_ptr(), c.data_ptr(), row * 8
        if withAmplitudes:
            d['Es'], d['Ep'] = rows[1], rows[2]
        # (positional: n, the ten f64 rows in _F64's order, Jsp, state, Es, Ep -- one call
        # instead of fifteen attribute stores)
        s = _structs.Beam(n, fb, fb + step, fb + 2 * step, fb + 3 * step, fb + 4 * step,
                          fb + 5 * step, fb + 6 * step, fb + 7 * step, fb + 8 * step,
                          fb + 9 * step, cb, state.data_ptr(),
                          cb + 2 * step if withAmplitudes else None,
                          cb + 4 * step if withAmplitudes else None)
        s._keep = (f, c, state)
        vars(b).update(_h={}, _d=d, _struct=s, parentId=None)
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
        held = self._d.keys()
        if not _ALWAYS <= held:
            return True
        return self.has_amplitudes() and not ('Es' in held and 'Ep' in held)


# ---- beams whose pass has not been launched yet -------------------------------------------------
# OE.reflect hands out its two beams before it launches anything: if the first thing the script
# does with the global beam is Screen.expose, the screen's image is made in the tail of the
# SAME pass (csrc/reflect_impl.h: reflect_fused_scr) and the global beam itself is not written
# unless somebody asks for it later. Any other use of either beam -- an attribute, a plot, the
# next element -- launches the plain pass at that moment: same kernels, same bits, a little
# later on the stream.
import threading as _threading
import weakref as _weakref


class _PendingOps(object):
    """The operations not launched yet, per Python thread (every thread of a parallel
    run_ray_tracing traces its own beams on its own stream)."""

    def __init__(self):
        self._tls = _threading.local()

    def _mine(self):
        ops = self._tls.__dict__.get('ops')
        if ops is None:
            ops = self._tls.ops = set()
        return ops

    def _mine_optional(self):
        ops = self._tls.__dict__.get('optional')
        if ops is None:
            ops = self._tls.optional = _weakref.WeakSet()
        return ops

    def add(self, op):
        # an *optional* operation has no effect but the beam it would fill: it lives as long as
        # that beam does and may never run at all
        (self._mine_optional() if getattr(op, 'optional', False) else self._mine()).add(op)

    def discard(self, op):
        self._mine().discard(op)
        self._mine_optional().discard(op)

    def __iter__(self):
        # consumers first: an element's deferred pass makes the rays of a pending device source
        # in its own head (and takes the source's record off this list); the other way round the
        # source would launch its generator for nothing
        ops = list(self._mine())
        return iter([op for op in ops if hasattr(op, 'src_op')] +
                    [op for op in ops if not hasattr(op, 'src_op')])

    def __contains__(self, op):
        return op in self._mine() or op in self._mine_optional()

    def optional(self):
        return list(self._mine_optional())


_PENDING = _PendingOps()

# Records that keep the STATES of a beam as they are at some moment (to make a beam that was left
# out later, from the same input) share the state tensor until somebody is about to change it in
# place -- apertures are the only ones who do -- and take their own copy then (a copy per record
# at once was ten 40-MB copy launches in a pass of the Balder chain, five of them never needed).
_STATE_SHARERS = _weakref.WeakSet()


class SharesStates(object):
    """Mixin of such a record: ``_share_states(snapshot)`` after taking the snapshot."""

    def _share_states(self, snap):
        self._sharing = snap
        _STATE_SHARERS.add(self)

    def own_states_now(self):
        snap = self.__dict__.pop('_sharing', None)
        _STATE_SHARERS.discard(self)
        if snap is not None and 'state' in snap._d:
            snap.state = snap._d['state'].clone()


def before_states_change(beam):
    """Called by whoever is about to write the states of *beam* in place."""
    t = (beam.__dict__.get('_real_d') or beam.__dict__.get('_d') or {}).get('state')
    if t is None:
        return
    for op in list(_STATE_SHARERS):
        snap = op.__dict__.get('_sharing')
        if snap is not None and snap._d.get('state') is t:
            op.own_states_now()
_FILL_LOCK = _threading.RLock()      # a beam looked at from another thread: one launch, complete


def flush_pending(beam=None, keep=None, only_state=False):
    """Launches what is still pending -- all of it (but *keep*), or what reads *beam* (called by
    whoever is about to change a beam's arrays in place; *only_state*: nothing but its states,
    of which the optional operations hold their own copy)."""
    for op in list(_PENDING):
        # (an operation launched earlier in this loop may have taken another one off the list:
        # the pass that makes a pending source's rays itself)
        if op is not keep and op in _PENDING and (beam is None or op.reads(beam)):
            op.materialize()
    if beam is not None and not only_state:
        for op in _PENDING.optional():
            if op is not keep and op in _PENDING and op.reads(beam):
                op.materialize()


class FillsBeams(object):
    """Mixin of a record that will fill LazyBeams: the beams refer to their record, the record
    refers to its beams WEAKLY. As a cycle, a beam the script had dropped kept its record and
    everything that holds -- the input's arrays, a beam-sized scratch, its own gigabyte -- until
    Python's cycle collector came by: tens of GB of garbage between collections, and new device
    allocations (milliseconds each) in loops that had been allocation-free. ``_make(role, ...)``
    creates a beam and keeps it alive until ``hand_out()`` gives it to the caller."""

    def _make(self, role, key=None):
        beam = LazyBeam(self, role)
        self.__dict__.setdefault('_fresh', []).append(beam)
        self.__dict__.setdefault('_weak', {})[role if key is None else key] = _weakref.ref(beam)
        return beam

    def _beam(self, key):
        ref = self.__dict__.get('_weak', {}).get(key)
        return None if ref is None else ref()

    def hand_out(self, always_tuple=False):
        fresh = self.__dict__.pop('_fresh', [])
        return fresh[0] if len(fresh) == 1 and not always_tuple else tuple(fresh)


def filled(beam):
    """A lazy beam has its arrays -- or nobody holds it any more (nothing to fill)."""
    return beam is None or beam.__dict__['_filled']


def adopt_into(beam, real):
    if beam is not None:
        beam._adopt_arrays(real)


class _DeferredShine(FillsBeams):
    """GeometricSource.shine(rng='device') not launched yet: the generator is counter-based, the
    record *g* makes the same rays whenever it runs. States: pending -> done (its own launch), or
    pending -> inflight (an element's pass made the rays in its registers,
    xrt_hip_shine_reflect_screen_f64_dev; the beam itself can still be made on demand) -> done."""

    def __init__(self, source, g, nrays, amplitudes, device, scalars, rec=None):
        self.source, self.g, self.n, self.amplitudes, self.device = \
            source, g, int(nrays), bool(amplitudes), device
        self.rec = rec               # the HIP graph this shine is recorded into, if any
        self.state = 'pending'
        self.scalars = dict(scalars)
        beam = self._make('beam')
        for key, value in scalars.items():
            object.__setattr__(beam, key, value)
        _PENDING.add(self)

    beam = property(lambda self: self._beam('beam'))

    def reads(self, beam):
        return False

    def launch_into(self, bo):
        import ctypes
        from ... import _lib, graphs
        g = self.g
        if self.rec is not None and graphs.capturing() is not self.rec:
            # recorded into a graph and asked for after its replays: the graph's cell has moved
            # on to the next replay, these are the rays of the last one
            g = type(g).from_buffer_copy(g)
            g.call = (g.call - self.rec.pending_calls.get(self.source, 0)) & 0xffffffff
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().xrt_hip_geosource_shine_f64_dev(
                ctypes.byref(g), ctypes.byref(bo.to_struct(self.device)),
                _hipcalls.stream_ptr()),
                'xrt_hip_geosource_shine_f64_dev')

    def materialize(self, which=None):
        _PENDING.discard(self)
        if self.state != 'done':
            if self.state == 'inflight':
                # an element's pass had made these rays in its registers and now somebody wants
                # the beam as well: the source remembers and launches its generator at once
                # from the next shine() on
                self.source.__dict__['_beam_wanted'] = True
            self.state = 'done'
            bo = Beam.empty_on_device(self.n, self.device, self.amplitudes)
            self.launch_into(bo)
            adopt_into(self.beam, bo)

    def rays_again(self):
        """-> the same rays in a NEW beam (for whoever has to run a pass on them again: this
        beam itself stays as it is, asked for or not -- and if it exists, an aperture may have
        changed its states since)."""
        bo = Beam.empty_on_device(self.n, self.device, self.amplitudes)
        self.launch_into(bo)
        for key, value in self.scalars.items():
            object.__setattr__(bo, key, value)
        return bo

    def adopt(self, bo):
        """*bo* holds the rays already (somebody else ran the generator into it)."""
        _PENDING.discard(self)
        if self.state != 'done':
            self.state = 'done'
            adopt_into(self.beam, bo)


class LazyBeam(Beam):
    """A beam that an operation (*op*, with ``materialize(which)``) will fill: its arrays
    come into being when first looked at."""

    def __init__(self, op, role):
        object.__setattr__(self, '_op', op)
        object.__setattr__(self, '_role', role)
        object.__setattr__(self, '_real_h', {})
        object.__setattr__(self, '_real_d', {})
        object.__setattr__(self, '_filled', False)
        object.__setattr__(self, 'parentId', None)

    def _fill(self):
        if not self.__dict__['_filled']:
            with _FILL_LOCK:
                if not self.__dict__['_filled']:
                    self.__dict__['_op'].materialize(self.__dict__['_role'])

    def _adopt_arrays(self, real):
        """Takes over the arrays of the beam the launch made."""
        object.__setattr__(self, '_real_d', real.__dict__['_d'])
        object.__setattr__(self, '_real_h', real.__dict__['_h'])
        if '_struct' in real.__dict__:
            object.__setattr__(self, '_struct', real.__dict__['_struct'])
        object.__setattr__(self, '_filled', True)

    @property
    def _d(self):
        self._fill()
        return self.__dict__['_real_d']

    @property
    def _h(self):
        self._fill()
        return self.__dict__['_real_h']


def inherit_scalars(new, old):
    """Per-beam scalars (source bookkeeping, flags) follow the rays into a new beam."""
    for key in _SCALAR_ATTRS:
        if key in old.__dict__:
            object.__setattr__(new, key, old.__dict__[key])
    if '_stopped_by' in old.__dict__:     # (apertures whose marks the states may carry)
        new.__dict__.setdefault('_stopped_by', set()).update(old.__dict__['_stopped_by'])


def copy_beam(beamTo, beamFrom, indarr, includeState=False, includeJspEsp=True):
    """Rays *indarr* of one host beam into another: geometry, path and energy always,
    the state and the coherency matrix / amplitudes on request."""
    fields = ['x', 'y', 'z', 'a', 'b', 'c', 'path', 'E']
    if includeState:
        fields.append('state')
    if includeJspEsp:
        fields += ['Jss', 'Jpp', 'Jsp']
        if hasattr(beamFrom, 'Es') and hasattr(beamTo, 'Es'):
            fields += ['Es', 'Ep']
    for name in fields:
        getattr(beamTo, name)[indarr] = getattr(beamFrom, name)[indarr]


def rotate_coherency_matrix(beam, indarr, roll):
    """J' = R J R^T of the rays *indarr* for the rotation by *roll* about the ray (host
    arrays): -> (Jss', Jpp', Jsp'); the imaginary part of Jsp is invariant."""
    cr, sr = np.cos(roll), np.sin(roll)
    cc, ss, mixed = cr**2, sr**2, cr * sr
    jss, jpp, jsp = beam.Jss[indarr], beam.Jpp[indarr], beam.Jsp[indarr]
    return (jss*cc + jpp*ss + 2*jsp.real*mixed,
            jss*ss + jpp*cc - 2*jsp.real*mixed,
            (jpp-jss)*mixed + jsp.real*(cc-ss) + jsp.imag*1j)


# ---------------------------------------------------------------------------
# GeometricSource (reference: sources/geoms.py). Sampling is host-side numpy on the
# GLOBAL np.random state, consuming it in the reference's order -- polarisation phase
# (unpolarised beams with amplitudes), y, then x and z, then x' and z', then the energy --
# so a script that seeds np.random gets the very same rays.
# ---------------------------------------------------------------------------
_ROOT_HALF = 2**(-0.5)
_RANDOM_PHASE = 'random phase'
# named polarisation states: Jss, Jpp, Jsp, Es, Ep
_POLARIZATION = {
    'un': (0.5, 0.5, 0, _ROOT_HALF, _RANDOM_PHASE),
    'r': (0.5, 0.5, 0.5j, _ROOT_HALF, -1j * _ROOT_HALF),
    'l': (0.5, 0.5, -0.5j, _ROOT_HALF, 1j * _ROOT_HALF),
}


def _linear_state(angle):
    """Linear polarisation at *angle* [rad] from the horizontal."""
    es, ep = np.cos(angle), np.sin(angle)
    return es*es, ep*ep, es*ep, es, ep


def _polarization_state(spec):
    """-> (Jss, Jpp, Jsp, Es, Ep) of a polarisation given the reference's way: None or
    'unpolarized', 'horizontal', 'vertical', 'right', 'left' (first letters suffice), an
    angle in degrees (number or string; '...rad' for radians), or the four real numbers
    (Jss, Jpp, Re Jsp, Im Jsp), which carry no amplitudes."""
    if spec is None:
        return _POLARIZATION['un']
    if isinstance(spec, (tuple, list, np.ndarray)):
        if len(spec) != 4:
            raise ValueError('wrong coherency matrix: must be a 4-sequence!')
        return spec[0], spec[1], spec[2] + 1j*spec[3], None, None
    if not isinstance(spec, str):
        return _linear_state(float(spec) * np.pi / 180.)
    word = spec.lower()
    for head, state in _POLARIZATION.items():
        if word.startswith(head):
            return state
    if word.startswith('h'):
        return _linear_state(0.)
    if word.startswith('v'):
        return _linear_state(np.pi / 2.)
    try:
        return _linear_state(float(word[:-3]) if word.endswith('rad')
                             else float(word) * np.pi / 180.)
    except ValueError:
        raise ValueError('wrong polarization!')


def make_polarization(polarization, bo, nrays=raycing.nrays):
    """Fills the coherency matrix of *bo* -- and its amplitudes, if it has them; an
    unpolarised beam gets Ep of uniformly random size up to 1/sqrt(2) (one draw)."""
    jss, jpp, jsp, es, ep = _polarization_state(polarization)
    bo.Jss.fill(jss)
    bo.Jpp.fill(jpp)
    bo.Jsp.fill(jsp)
    if es is None or not hasattr(bo, 'Es'):
        return
    bo.Es.fill(es)
    if isinstance(ep, str):
        bo.Ep[:] = np.random.uniform(size=int(nrays)) * _ROOT_HALF
    else:
        bo.Ep.fill(ep)


def make_energy(distE, energies, nrays, filamentBeam=False, energyWeights=None):
    """Photon energies: 'normal' (mean, sigma -- a sigma above a tenth of the mean is
    taken as 0), 'flat' (min, max) or 'lines' (discrete values, optionally weighted); a
    filament beam has one energy."""
    count = 1 if filamentBeam else int(nrays)
    values = np.atleast_1d(energies)
    pair = len(values) == 2
    if distE == 'normal':
        spread = abs(values[1]) if pair else 0
        return np.random.normal(values[0], 0 if spread > 0.1*abs(values[0]) else spread,
                                count)
    if distE == 'flat':
        return np.random.uniform(values[0], (values[1] or values[0]) if pair else values[0],
                                 count)
    if distE == 'lines':
        if 0 in values:
            values = values[values > 0]
        weights = None if energyWeights is None else np.atleast_1d(energyWeights)
        if weights is not None and len(weights) != len(values):
            weights = None
        return np.random.choice(values, size=count, p=weights)
    raise ValueError('unknown distE')


def _enrol_source(source, bl, name, uuid_=None, listed=True):
    """A source on its beamline: numbered in ``bl.sources`` (if *listed*) and findable by
    its uuid."""
    source.bl, source.name = bl, name
    source.uuid = uuid_ or raycing.new_uuid()
    if bl is None:
        return
    if listed and source not in bl.sources:
        bl.sources.append(source)
        source.ordinalNum = len(bl.sources)
    bl.oesDict[source.uuid] = [source, 0]


class GeometricSource(object):
    """Rays with origin, divergence and energy sampled from simple laws."""
    # the sampled ray coordinates in the order the random numbers are drawn: a lone one,
    # then pairs that may form an annulus
    _LONE = ('y',)
    _PAIRS = (('x', 'z'), ('xprime', 'zprime'))
    _FIELD = {'x': 'x', 'y': 'y', 'z': 'z', 'xprime': 'a', 'zprime': 'c'}

    def __init__(self, bl=None, name='', center=(0, 0, 0), nrays=raycing.nrays,
                 distx='normal', dx=0.32, disty=None, dy=0, distz='normal',
                 dz=0.018, distxprime='normal', dxprime=1e-3,
                 distzprime='normal', dzprime=1e-4, distE='lines',
                 energies=(defaultEnergy,), energyWeights=None,
                 polarization='horizontal', filamentBeam=False,
                 uniformRayDensity=False, pitch=0, roll=0, yaw=0, totalFlux=None,
                 rng='host', seed=None, **kwargs):
        """*rng*: 'host' (default) samples with numpy's global generator in the reference's
        order -- a script that seeds ``np.random`` gets the reference's very rays; 'device'
        samples on the GPU (one kernel, counter-based Philox4x32-10 under *seed*, nothing
        crosses PCIe): the same laws, other random numbers. *seed* None = drawn from
        ``np.random`` at the first ``shine`` (a seeded script stays reproducible); every
        ``shine`` call takes the next sub-stream."""
        given = dict(locals())
        _enrol_source(self, bl, name or 'GeometricSource', kwargs.get('uuid'))
        self.nrays = int(nrays)
        if rng not in ('host', 'device'):
            raise ValueError("rng must be 'host' or 'device'")
        self.rng, self.seed, self._calls = rng, seed, 0
        import threading
        self._call_lock = threading.Lock()      # run_ray_tracing(threads=N): one sub-stream per call
        for key in ('center', 'distE', 'energies', 'energyWeights', 'polarization',
                    'filamentBeam', 'uniformRayDensity', 'pitch', 'roll', 'yaw', 'totalFlux'):
            setattr(self, key, given[key])
        for coord in self._FIELD:
            setattr(self, 'dist' + coord, given['dist' + coord])
            setattr(self, 'd' + coord, given['d' + coord])

    def _weigh_down(self, bo, weight):
        """Uniform ray density: the law goes into the intensities instead."""
        for name in ('Jss', 'Jpp', 'Jsp'):
            getattr(bo, name).__imul__(weight)
        for name in ('Es', 'Ep'):
            getattr(bo, name).__imul__(weight**0.5)

    def _apply_distribution(self, axis, distaxis, daxis, bo=None):
        """One coordinate from its law: 'normal' (sigma, or (sigma, cut) for a uniform
        ray density, default cut 5 sigma) or 'flat' (full width, or (min, max))."""
        n = self.nrays
        if distaxis == 'normal' and self.uniformRayDensity:
            widths = np.atleast_1d(daxis)
            sigma = widths[0]
            cut = widths[-1] if len(widths) > 1 else 5 * abs(sigma)
            axis[:] = np.random.uniform(-cut, cut, n)
            self._weigh_down(bo, np.exp(-axis**2 / sigma**2 / 2) / PI2**0.5 / sigma * 2 * cut)
        elif distaxis == 'normal':
            try:
                axis[:] = np.random.normal(
                    0, daxis[0] if isinstance(daxis, (list, tuple)) else daxis, n)
            except ValueError:            # a negative sigma
                axis[:] = 0.
        elif distaxis == 'flat':
            if raycing.is_sequence(daxis):
                axis[:] = np.random.uniform(daxis[0], daxis[1], n)
            elif daxis > 0:
                axis[:] = np.random.uniform(-daxis*0.5, daxis*0.5, n)

    def _set_annulus(self, axis1, axis2, rMin, rMax, phiMin, phiMax):
        """Uniform over the ring between two radii (a circle line if they coincide)."""
        n = self.nrays
        radius = rMax
        if rMax > rMin:
            density = 2. / (rMax**2 - rMin**2)
            radius = np.sqrt(2*np.random.uniform(0, 1, n)/density + rMin**2)
        angle = np.random.uniform(phiMin, phiMax, n)
        axis1[:], axis2[:] = radius * np.cos(angle), radius * np.sin(angle)

    def _sample_pair(self, bo, first, second):
        laws = [getattr(self, 'dist' + c) for c in (first, second)]
        sizes = [getattr(self, 'd' + c) for c in (first, second)]
        fields = [getattr(bo, self._FIELD[c]) for c in (first, second)]
        if 'annulus' in laws and raycing.is_sequence(sizes[0]):
            arc = sizes[1] if raycing.is_sequence(sizes[1]) else (0, PI2)
            self._set_annulus(fields[0], fields[1], sizes[0][0], sizes[0][1], *arc)
        else:
            for field, law, size in zip(fields, laws, sizes):
                self._apply_distribution(field, law, size, bo)

    # ---- rng='device': the same laws as plain numbers for csrc/source.hip ----------------
    def _law(self, coord):
        """(law, p0, p1) of one coordinate (the branches of _apply_distribution)."""
        dist, size = getattr(self, 'dist' + coord), getattr(self, 'd' + coord)
        if dist == 'normal' and self.uniformRayDensity:
            widths = np.atleast_1d(size)
            return (_structs.LAW_NORMAL_UNIFORM, float(widths[0]),
                    float(widths[-1] if len(widths) > 1 else 5 * abs(widths[0])))
        if dist == 'normal':
            sigma = float(size[0] if isinstance(size, (list, tuple)) else size)
            # (numpy refuses a negative sigma and the reference then leaves zeros)
            return (_structs.LAW_NORMAL, sigma, 0.) if sigma >= 0 else (_structs.LAW_NONE, 0., 0.)
        if dist == 'flat':
            if raycing.is_sequence(size):
                return _structs.LAW_FLAT, float(size[0]), float(size[1])
            if size > 0:
                return _structs.LAW_FLAT, -size*0.5, size*0.5
        return _structs.LAW_NONE, 0., 0.

    def _reach(self, law):
        """The largest |value| a coordinate can take under *law*: Box-Muller on 53-bit
        uniforms ends at sqrt(-2 ln 2^-53) = 8.572 sigma."""
        kind, p0, p1 = law
        if kind == _structs.LAW_NORMAL:
            return 8.58 * abs(p0)
        if kind == _structs.LAW_FLAT:
            return max(abs(p0), abs(p1))
        return abs(p1) if kind == _structs.LAW_NORMAL_UNIFORM else 0.

    def device_spec(self, toGlobal=True, call=None):
        """The ``xrt_hip_geosource`` record of this source (all but ``slopes``) for sub-stream
        *call* (default: the next one)."""
        g = _structs.GeoSource()
        with self._call_lock:
            if self.seed is None:
                self.seed = int(np.random.randint(0, 2**62, dtype=np.int64))
            call = self._calls if call is None else call
        g.seed, g.call = int(self.seed) & (2**64 - 1), call & 0xFFFFFFFF
        laws = [self._law(c) for c in ('y', 'x', 'z', 'xprime', 'zprime')]
        for k, (kind, p0, p1) in enumerate(laws):
            g.law[k], g.p0[k], g.p1[k] = kind, p0, p1
        reach = []
        for flag, ring, first, second in (('annulus_xz', g.ann_xz, 'x', 'z'),
                                          ('annulus_ac', g.ann_ac, 'xprime', 'zprime')):
            dists = [getattr(self, 'dist' + c) for c in (first, second)]
            sizes = [getattr(self, 'd' + c) for c in (first, second)]
            if 'annulus' in dists and raycing.is_sequence(sizes[0]):
                arc = sizes[1] if raycing.is_sequence(sizes[1]) else (0, PI2)
                setattr(g, flag, 1)
                ring[0], ring[1], ring[2], ring[3] = (float(sizes[0][0]), float(sizes[0][1]),
                                                      float(arc[0]), float(arc[1]))
                reach.append(max(abs(ring[0]), abs(ring[1]))**2)
            else:
                k = 1 if first == 'x' else 3
                reach.append(self._reach(laws[k])**2 + self._reach(laws[k + 1])**2)
        g.e_p0 = defaultEnergy
        if self.distE is not None:
            values = np.atleast_1d(np.asarray(self.energies, dtype=float))
            pair = len(values) == 2
            g.filament = 1 if self.filamentBeam else 0
            if self.distE == 'normal':
                spread = abs(values[1]) if pair else 0.
                g.e_law, g.e_p0, g.e_p1 = 1, values[0], \
                    0. if spread > 0.1*abs(values[0]) else spread
            elif self.distE == 'flat':
                g.e_law, g.e_p0, g.e_p1 = 2, values[0], \
                    (values[1] or values[0]) if pair else values[0]
            elif self.distE == 'lines':
                if 0 in values:
                    values = values[values > 0]
                weights = None if self.energyWeights is None else \
                    np.atleast_1d(self.energyWeights).astype(float)
                if weights is None or len(weights) != len(values):
                    weights = np.ones(len(values))
                if not 1 <= len(values) <= _structs.MAX_LINES:
                    raise ValueError("rng='device' takes 1..%d energy lines (use rng='host')"
                                     % _structs.MAX_LINES)
                cdf = np.cumsum(weights) / np.sum(weights)
                cdf[-1] = 1.
                g.e_law, g.n_lines = 3, len(values)
                for k in range(len(values)):
                    g.e_lines[k], g.e_cdf[k] = values[k], cdf[k]
            else:
                raise ValueError('unknown distE')
        jss, jpp, jsp, es, ep = _polarization_state(self.polarization)
        g.Jss, g.Jpp = float(jss), float(jpp)
        g.Jsp[0], g.Jsp[1] = complex(jsp).real, complex(jsp).imag
        if es is not None:
            g.Es[0], g.Es[1] = complex(es).real, complex(es).imag
            if isinstance(ep, str):
                g.random_ep = 1
            else:
                g.Ep[0], g.Ep[1] = complex(ep).real, complex(ep).imag
        steps = raycing.rotation_steps(pitch=self.pitch, roll=self.roll, yaw=self.yaw)
        g.rot.n = len(steps)
        for k, (axis, cs, sn) in enumerate(steps):
            g.rot.axis[k], g.rot.cosa[k], g.rot.sina[k] = axis, cs, sn
        g.state = 1
        if toGlobal:
            g.to_global = 1
            g.sin_az, g.cos_az = float(self.bl.sinAzimuth), float(self.bl.cosAzimuth)
            for k in range(3):
                g.center[k] = float(self.center[k])
        return g, reach[1]

    def _replay_cell(self, dev):
        """(device cell, its value as the host knows it) counting the replays of the graphs this
        source was recorded into; made outside any recording."""
        cells = self.__dict__.setdefault('_replay_cells', {})
        held = cells.get(str(dev))
        if held is None:
            from ... import graphs
            if graphs.capturing() is not None:
                raise graphs.CaptureError('the source has not been used eagerly on this device '
                                          'before the recording')
            held = cells[str(dev)] = [torch.zeros(1, dtype=torch.int32, device=dev), 0]
        return held

    def _sync_replay_cell(self):
        """Before a replay: the eager shine() calls since the last one move the device cell."""
        for key, lag in self.__dict__.get('_cell_lag', {}).items():
            held = self.__dict__.get('_replay_cells', {}).get(key)
            if held is not None and lag:
                held[0].add_(lag)
                held[1] += lag
            self._cell_lag[key] = 0

    def _shine_device(self, toGlobal, withAmplitudes, accuBeam):
        import ctypes
        from ... import _lib
        _lib.require_gpu()
        lib = _lib.load()
        dev = torch.device('cuda', torch.cuda.current_device())
        stream = _hipcalls.stream_ptr()
        from ... import graphs
        rec = graphs.capturing()
        with self._call_lock:            # every shine() takes its own sub-stream, also from threads
            if rec is None:
                call = self._calls
                self._calls += 1
                self._replay_cell(dev)   # (exists before any recording)
                # an eager call between the replays of a graph: the graph's cell has to move
                # with the call counter (applied before the next replay, _sync_replay_cell)
                lag = self.__dict__.setdefault('_cell_lag', {})
                lag[str(dev)] = lag.get(str(dev), 0) + 1
            else:
                self.__dict__.setdefault('_cell_lag', {})[str(dev)] = 0   # (absorbed below)
                # recorded into a HIP graph: replay k must draw what the k-th eager call would
                # have -- the kernel adds a device cell to the call number of the record. The
                # graph increments the cell ONCE, as its last node, by the number of shines it
                # holds: within an iteration the cell stands still (the rays of a shine are the
                # same wherever they are made -- its own launch, the head of an element's pass,
                # a redo), and the j-th shine of replay r finds the cell at value + r * shines and
                # draws call _calls + r * shines + j
                cell, value = self._replay_cell(dev)
                j = rec.pending_calls.get(self, 0)
                rec.pending_calls[self] = j + 1
                call = (self._calls - value + j) & 0xffffffff
        g, reach2 = self.device_spec(toGlobal, call)
        if rec is not None:
            g.call_dev = cell.data_ptr()
            if reach2 > 1:
                graphs.refuse('a source whose tangents can leave the unit circle (the whole-batch '
                              'decision on how b is formed is read back)')
        if reach2 > 1:
            # tangents that CAN leave the unit circle: the reference forms b from slopes if
            # any ray of the batch does (geoms.py:497-505) -- ask the generator
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(lib.xrt_hip_geosource_probe_f64_dev(
                ctypes.byref(g), self.nrays, ctypes.c_void_p(flag.data_ptr()), stream),
                'xrt_hip_geosource_probe_f64_dev')
            g.slopes = int(flag.item())
        amplitudes = withAmplitudes or self.uniformRayDensity
        if rec is not None and rec.pending_calls[self] == 1:
            rec.before_end.append(lambda: cell.add_(rec.pending_calls[self]))

            def count():
                with self._call_lock:
                    self._calls += rec.pending_calls[self]
                    self._replay_cells[str(dev)][1] += rec.pending_calls[self]
            rec.after_replay.append(count)
        if reach2 <= 1 and accuBeam is None and not self.uniformRayDensity:
            from . import oes as _oes
            if _oes.fuseConsumers and not self.__dict__.get('_beam_wanted'):
                # not launched yet: an element's pass may make these rays in its own registers
                # (oes._DeferredReflect); anything else that looks at the beam launches the
                # generator -- the record draws the same rays whenever it runs
                flush_pending()
                scalars = dict(parentId=self.uuid)
                if np.isscalar(self.totalFlux) and self.totalFlux > 0:
                    total = self.nrays * (g.Jss + g.Jpp)
                    if total > 0:
                        scalars.update(sourceWeight=self.totalFlux / total, seeded=self.nrays,
                                       seededI=1., accepted=1., acceptedE=1.)
                return _DeferredShine(self, g, self.nrays, amplitudes, dev, scalars, rec).hand_out()
        bo = Beam.empty_on_device(self.nrays, dev, amplitudes)
        _lib.check(lib.xrt_hip_geosource_shine_f64_dev(
            ctypes.byref(g), ctypes.byref(bo.to_struct(dev)), stream),
            'xrt_hip_geosource_shine_f64_dev')
        if np.isscalar(self.totalFlux) and self.totalFlux > 0:      # make_flux_normalization
            if self.uniformRayDensity:
                graphs.refuse('totalFlux of a source with uniformRayDensity (a sum read back)')
            total = self.nrays * (g.Jss + g.Jpp) if not self.uniformRayDensity else \
                float((bo.dev('Jss') + bo.dev('Jpp')).sum())
            if total > 0:
                bo.sourceWeight = self.totalFlux / total
                bo.seeded, bo.seededI, bo.accepted, bo.acceptedE = self.nrays, 1., 1., 1.
        if self.distE is not None and accuBeam is not None:
            bo.E = accuBeam.dev('E', dev).clone()
        bo.parentId = self.uuid
        return bo

    def shine(self, toGlobal=True, withAmplitudes=False, accuBeam=None):
        """One beam of ``nrays`` rays, in the global frame if *toGlobal*."""
        if self.rng == 'device':
            return self._shine_device(toGlobal, withAmplitudes, accuBeam)
        bo = Beam(self.nrays, withAmplitudes=withAmplitudes or self.uniformRayDensity)
        bo.state[:] = 1
        make_polarization(self.polarization, bo, self.nrays)
        for coord in self._LONE:
            self._apply_distribution(getattr(bo, self._FIELD[coord]),
                                     getattr(self, 'dist' + coord),
                                     getattr(self, 'd' + coord), bo)
        for pair in self._PAIRS:
            self._sample_pair(bo, *pair)
        # b from the two tangents: as direction cosines when they leave room for it,
        # else as slopes of a vector of length sqrt(1 + a^2 + c^2)
        transverse = bo.a**2 + bo.c**2
        if (transverse > 1).any():
            length = (transverse + 1)**0.5
            bo.a[:] /= length
            bo.c[:] /= length
            bo.b[:] = 1.0 / length
        else:
            bo.b[:] = (1 - transverse)**0.5
        if np.isscalar(self.totalFlux) and self.totalFlux > 0:      # make_flux_normalization,
            total = (bo.Jss + bo.Jpp).sum()                         # geoms.py:182-190
            if total > 0:
                bo.sourceWeight = self.totalFlux / total
                bo.seeded, bo.seededI, bo.accepted, bo.acceptedE = len(bo.E), 1., 1., 1.
        if self.distE is not None:
            bo.E[:] = accuBeam.E[:] if accuBeam is not None else make_energy(
                self.distE, self.energies, self.nrays, self.filamentBeam,
                self.energyWeights)
        if self.pitch or self.roll or self.yaw:
            raycing.rotate_beam(bo, pitch=self.pitch, roll=self.roll, yaw=self.yaw)
        if toGlobal:
            raycing.virgin_local_to_global(self.bl, bo, self.center)
        bo.parentId = self.uuid
        return bo


class MeshSource(object):
    """Point source sending a regular fan of rays: *nx* by *nz* directions between the given
    angular limits (rows from the top), optionally preceded by the central ray; used to
    find the divergence an element accepts (reference sources/geoms.py:853-1068)."""

    def __init__(self, bl=None, name='', center=(0, 0, 0), minxprime=-1e-4, maxxprime=1e-4,
                 minzprime=-1e-4, maxzprime=1e-4, nx=11, nz=11, distE='lines',
                 energies=(defaultEnergy,), energyWeights=None, polarization='horizontal',
                 withCentralRay=True, autoAppendToBL=False, totalFlux=None, **kwargs):
        given = dict(locals())
        _enrol_source(self, bl, name or 'MeshSource', kwargs.get('uuid'), autoAppendToBL)
        for key in ('center', 'nx', 'nz', 'distE', 'energies', 'energyWeights', 'polarization',
                    'withCentralRay', 'totalFlux'):
            setattr(self, key, given[key])
        for key in ('minxprime', 'maxxprime', 'minzprime', 'maxzprime'):
            setattr(self, key, raycing.auto_units_angle(given[key]))

    nrays = property(lambda self: self.nx * self.nz + int(self.withCentralRay))

    def _fan(self, bo, first, a, c):
        """Tangents (a, c) -> unit directions of the rays from *first* on."""
        bo.a[first:], bo.c[first:] = a, c
        length = (bo.a**2 + 1.0 + bo.c**2)**0.5
        bo.a[:] = bo.a / length
        bo.c[:] = bo.c / length
        bo.b[:] = 1.0 / length

    def shine(self, toGlobal=True):
        bo = Beam(self.nrays)
        bo.state[:] = 1
        self.dxprime = (self.maxxprime-self.minxprime) / (self.nx-1)
        self.dzprime = (self.maxzprime-self.minzprime) / (self.nz-1)
        across, up = np.meshgrid(np.linspace(self.minxprime, self.maxxprime, self.nx),
                                 np.linspace(self.minzprime, self.maxzprime, self.nz))
        self._fan(bo, int(self.withCentralRay), across.flatten(), np.flipud(up).flatten())
        if self.distE is not None:
            bo.E[:] = make_energy(self.distE, self.energies, self.nrays,
                                  energyWeights=self.energyWeights)
        make_polarization(self.polarization, bo, self.nrays)
        if np.isscalar(self.totalFlux) and self.totalFlux > 0:      # absolute flux [ph/s]
            total = (bo.Jss + bo.Jpp).sum()
            if total > 0:
                bo.sourceWeight = self.totalFlux / total
                bo.seeded, bo.seededI, bo.accepted, bo.acceptedE = len(bo.E), 1., 1., 1.
        if toGlobal:
            raycing.virgin_local_to_global(self.bl, bo, self.center)
        bo.parentId = self.uuid
        return bo


class CollimatedMeshSource(MeshSource):
    """Parallel rays from a regular *nx* by *nz* mesh of points over *dx* x *dz* (rows from
    the top), optionally preceded by the central ray (reference sources/geoms.py:1111-1245)."""

    def __init__(self, bl=None, name='', center=(0, 0, 0), dx=1., dz=1., nx=11, nz=11,
                 distE='lines', energies=(defaultEnergy,), energyWeights=None,
                 polarization='horizontal', withCentralRay=True, autoAppendToBL=False,
                 totalFlux=None, **kwargs):
        MeshSource.__init__(self, bl, name or 'CollimatedMeshSource', center, nx=nx, nz=nz,
                            distE=distE, energies=energies, energyWeights=energyWeights,
                            polarization=polarization, withCentralRay=withCentralRay,
                            autoAppendToBL=autoAppendToBL, totalFlux=totalFlux, **kwargs)
        self.dx, self.dz = dx, dz
        for angular in ('minxprime', 'maxxprime', 'minzprime', 'maxzprime'):
            del self.__dict__[angular]

    def shine(self, toGlobal=True):
        bo = Beam(self.nrays)
        bo.state[:] = 1
        across, up = np.meshgrid(np.linspace(-self.dx/2., self.dx/2., self.nx),
                                 np.linspace(-self.dz/2., self.dz/2., self.nz))
        first = int(self.withCentralRay)
        bo.x[first:] = across.flatten()
        bo.z[first:] = np.flipud(up).flatten()
        if self.distE is not None:
            bo.E[:] = make_energy(self.distE, self.energies, self.nrays,
                                  energyWeights=self.energyWeights)
        make_polarization(self.polarization, bo, self.nrays)
        if np.isscalar(self.totalFlux) and self.totalFlux > 0:
            total = (bo.Jss + bo.Jpp).sum()
            if total > 0:
                bo.sourceWeight = self.totalFlux / total
                bo.seeded, bo.seededI, bo.accepted, bo.acceptedE = len(bo.E), 1., 1., 1.
        if toGlobal:
            raycing.virgin_local_to_global(self.bl, bo, self.center)
        bo.parentId = self.uuid
        return bo


class BeamFromFile(object):
    """A source that hands out a beam saved earlier with ``Beam.export_beam`` (reference
    sources/geoms.py:1247-1300): a reproducible stand-in for an expensive source."""

    def __init__(self, bl=None, name='', center=(0, 0, 0), fileName=None, **kwargs):
        _enrol_source(self, bl, name or 'BeamFromFile', kwargs.get('uuid'))
        self.center = center
        self.fileName = fileName

    @property
    def fileName(self):
        return self._fileName

    @fileName.setter
    def fileName(self, path):
        self._fileName = path
        self.fbeam = Beam(copyFrom=path) if path is not None else Beam()

    nrays = property(lambda self: np.int64(np.asarray(self.fbeam.x).size))

    def shine(self):
        return self.fbeam


class NESWSource(MeshSource):
    """The four extreme rays of the fan: up, right, down, left, from 50 um above the
    centre (sources/geoms.py:1071-1108)."""
    nrays = 4

    def shine(self, toGlobal=True):
        bo = Beam(4)
        bo.state[:] = 1
        self._fan(bo, 0, np.array([0, self.maxxprime, 0, self.minxprime]),
                  np.array([self.maxzprime, 0, self.minzprime, 0]))
        bo.z[:] += 0.05
        if toGlobal:
            raycing.virgin_local_to_global(self.bl, bo, self.center)
        return bo


# The source classes that live in modules of their own (they need Beam from this one) are
# re-exported on first use, so that each of those modules can also be imported first.
_LAZY = {'Undulator': 'undulator', 'SourceFromField': 'fieldsource',
         'BendingMagnet': 'bendsource', 'Wiggler': 'bendsource',
         'GaussianBeam': 'gaussbeam', 'LaguerreGaussianBeam': 'gaussbeam',
         'HermiteGaussianBeam': 'gaussbeam'}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        value = getattr(importlib.import_module('.' + _LAZY[name], __package__), name)
        globals()[name] = value
        return value
    raise AttributeError('module %r has no attribute %r' % (__name__, name))


def __dir__():
    return sorted(set(globals()) | set(_LAZY))
