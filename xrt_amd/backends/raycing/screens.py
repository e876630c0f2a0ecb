"""``Screen`` with the interface of the reference's (xrt/backends/raycing/screens.py):
``expose`` is the streaming HIP kernel of csrc/screen.hip on a device-resident beam,
``prepare_wave`` lays the receiving mesh of ``waves.diffract`` on the screen."""
import ctypes

import numpy as np
import torch

from ... import hipcalls as _hipcalls

from .. import raycing
from ... import _lib, _structs
from . import sources as rs


def _pitch_of(axis):
    """Step of a regular axis (its first interval), 1 for anything without one."""
    try:
        return axis[1] - axis[0]
    except (TypeError, IndexError):
        return None


class Screen(object):
    def __init__(self, bl=None, name='', center=[0, 0, 0], x='auto', z='auto',
                 compressX=None, compressZ=None, **kwargs):
        raycing.enrol(self, bl, 'screens', 2000, name, 'Screen', kwargs.get('uuid'))
        self.center, self.footprint = center, []
        self.compressX, self.compressZ = compressX, compressZ
        self.set_orientation(x, z)

    def set_orientation(self, x=None, z=None):
        """Local axes; 'auto' (any string): x horizontal, z vertical."""
        given = [None if isinstance(v, str) else v for v in (x, z)]
        self.x, self.y, self.z = raycing.xyz_from_xz(self, *given)

    def local_to_global(self, x=0, y=0, z=0, **kwargs):
        return tuple(raycing.along_basis((self.x, self.y, self.z), x, y, z,
                                         self.center))

    def _record(self, onlyPositivePath):
        s = _structs.Screen()
        for k in range(3):
            s.center[k], s.ex[k], s.ey[k], s.ez[k] = (
                float(self.center[k]), float(self.x[k]), float(self.y[k]), float(self.z[k]))
        s.compress_x = float(self.compressX or 0.)
        s.compress_z = float(self.compressZ or 0.)
        s.lost_num = int(self.lostNum)
        s.only_positive_path = int(bool(onlyPositivePath))
        return s

    def expose(self, beam=None, onlyPositivePath=False):
        """The image of *beam* (global frame) in the screen's frame: every ray is
        projected on the screen axes and carried along its direction to the plane
        (reference screens.py:226-302)."""
        _lib.require_gpu()
        dev = torch.device('cuda', torch.cuda.current_device())
        rec = self._record(onlyPositivePath)
        op = beam.__dict__.get('_op') if type(beam) is rs.LazyBeam else None
        # (only the global beam of an element's deferred pass: a device source's beam, an
        # aperture's lazy local beam or a DCM's beams on demand are other kinds of record)
        if op is not None and getattr(op, 'gb', None) is beam and op.state == 'pending' and \
                getattr(getattr(op, 'p', None), 'out_to_global', 0):
            # the element's pass has not been launched: the image is made in its tail, the
            # global beam is not written unless somebody else asks for it (sources.LazyBeam).
            # Nor is the pass launched now: the image is handed out first, a plot of it may
            # still join (runner.accumulate_plot -> op.plot_on)
            if op.screen_rec is None and type(self) is Screen:
                return op.expose_later(self, rec)
            op.materialize('gb')
        from . import oes as roe
        if roe.fuseConsumers and type(self) is Screen and type(beam) is rs.Beam:
            # a resident beam (a source's): the image is handed out before the launch, so that
            # aperture.propagate of the same beam right behind this screen -- a front-end
            # monitor and its mask -- can make image and marks in ONE pass over the rays
            # (_DeferredExpose); launched at the latest by the next element's call
            return _DeferredExpose(self, rec, beam, dev).hand_out()
        image = rs.Beam.empty_like_on_device(beam, dev)
        _lib.check(_lib.load().xrt_hip_screen_expose_f64_dev(
            ctypes.byref(rec), ctypes.byref(beam.to_struct(dev)),
            ctypes.byref(image.to_struct(dev)),
            _hipcalls.stream_ptr()),
            'xrt_hip_screen_expose_f64_dev')
        rs.inherit_scalars(image, beam)
        return image

    def prepare_wave(self, prevOE, dim1, dim2, dy=0, rw=None, condition=None):
        """Receiving points for the field diffracted by *prevOE*: the mesh of local
        x values *dim1* times local z values *dim2* (x runs fastest), optionally
        thinned by *condition(x, z)*, in a plane *dy* off the screen's
        (reference screens.py:304-365). The cell size is taken from the first
        intervals of the two axes."""
        if not rw:
            from . import waves as rw
        gx, gz = (g.ravel() for g in np.meshgrid(dim1, dim2))
        steps = _pitch_of(dim1), _pitch_of(dim2)
        cell = 1. if None in steps else steps[0] * steps[1]
        if callable(condition):
            gx, gz = condition(gx, gz)
        area = (np.ones_like(gx) * cell).sum()
        if torch.cuda.is_available():
            # the mesh goes up once; the wave is made and stays on the GPU (the same
            # elementwise operations in the same order as with host arrays)
            dev = torch.device('cuda', torch.cuda.current_device())
            gx, gz = (torch.from_numpy(np.ascontiguousarray(g, dtype=np.float64)).to(dev)
                      for g in (gx, gz))
            gy = torch.zeros_like(gx) + dy
        else:
            gy = np.zeros_like(gx) + dy
        xg, yg, zg = self.local_to_global(x=gx, z=gz)
        return rw.receiving_wave(self, prevOE, (gx, gy, gz), (xg, yg + dy, zg), cell, area,
                                 prevOE.uuid)


class _DeferredExpose(rs.FillsBeams):
    """Screen.expose of a resident beam not launched yet (reference screens.py:226-302). The
    record keeps the beam's arrays as they are now (whoever writes INTO them launches their
    readers first, sources.flush_pending); ``aperture.propagate`` of the same beam finds the
    record and makes the image and its own marks in one launch
    (xrt_hip_screen_expose_mark_f64_dev: the image sees the states from before the marks, as
    in the two calls); otherwise the next flush -- every element's call starts with one --
    launches the screen's own kernel. Not optional: the image is written whether or not the
    script keeps it, as the immediate launch does."""

    def __init__(self, screen, rec, beam, dev):
        self.rec, self.device = rec, dev
        beam.to_struct(dev)                          # everything up in HBM now
        was = rs.Beam.__new__(rs.Beam)
        object.__setattr__(was, '_h', {})
        object.__setattr__(was, '_d', dict(beam._d))
        object.__setattr__(was, 'parentId', None)
        self.was = was
        self.tensors = {id(t) for t in was._d.values()}
        self.state = 'pending'
        rs.inherit_scalars(self._make('shot'), beam)
        rs._PENDING.add(self)

    def reads(self, beam):
        d = beam.__dict__.get('_real_d', beam.__dict__.get('_d')) or {}
        return any(id(t) in self.tensors for t in d.values())

    def _done(self, picture):
        rs._PENDING.discard(self)
        self.state = 'done'
        rs.adopt_into(self._beam('shot'), picture)
        self.was, self.tensors = None, ()

    def materialize(self, which=None):
        if self.state == 'done':
            return
        picture = rs.Beam.empty_like_on_device(self.was, self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().xrt_hip_screen_expose_f64_dev(
                ctypes.byref(self.rec), ctypes.byref(self.was.to_struct(self.device)),
                ctypes.byref(picture.to_struct(self.device)), _hipcalls.stream_ptr()),
                'xrt_hip_screen_expose_f64_dev')
        self._done(picture)       # (a launch that raises is raised again by the next look)

    def with_marks(self, aperture_record, beam):
        """The image and the marks of *aperture_record* in *beam* (whose arrays are the ones
        this record holds) in one launch."""
        picture = rs.Beam.empty_like_on_device(self.was, self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().xrt_hip_screen_expose_mark_f64_dev(
                ctypes.byref(self.rec), ctypes.byref(aperture_record),
                ctypes.byref(beam.to_struct(self.device)),
                ctypes.byref(picture.to_struct(self.device)), _hipcalls.stream_ptr()),
                'xrt_hip_screen_expose_mark_f64_dev')
        self._done(picture)


def pending_expose_of(beam, dev):
    """-> the record of the ONE screen that waits with its launch for *beam* as it is now (the
    very arrays the beam has at this moment, on this device), else None (no such screen, or
    two of them: each then takes its own launch). Other records that read the beam are the
    caller's to flush (sources.flush_pending(beam, keep=...))."""
    found = None
    for op in list(rs._PENDING) + rs._PENDING.optional():
        if not op.reads(beam):
            continue
        if type(op) is _DeferredExpose and found is None and op.state == 'pending' and \
                op.device == dev and \
                op.tensors == {id(t) for t in beam._d.values()}:
            found = op
        elif type(op) is _DeferredExpose:
            return None           # (two screens: each its own launch)
    return found


class HemisphericScreen(Screen):
    """Screen on a sphere of radius *R* about *center* (reference screens.py:422-559): the
    image carries the local position on the sphere and the angles ``theta`` (latitude
    about the polar axis *z*) and ``phi`` (azimuth from *x*), each less its offset. 'auto'
    axes: *x* along the beamline, *z* horizontal across it."""

    def __init__(self, bl=None, name='', center=[0, 0, 0], R=1000., x='auto', z='auto',
                 phiOffset=0, thetaOffset=0, **kwargs):
        Screen.__init__(self, bl, name, center, x, z, **kwargs)
        self.R, self.phiOffset, self.thetaOffset = R, phiOffset, thetaOffset

    def set_orientation(self, x=None, z=None):
        def unit(v):
            v = np.asarray(v, dtype=float)
            return list(v / sum(c**2 for c in v)**0.5)
        s, c = (self.bl.sinAzimuth, self.bl.cosAzimuth) if self.bl is not None else (0., 1.)
        self.x = (s, c, 0.) if x is None or isinstance(x, str) else unit(x)
        self.z = (c, -s, 0.) if z is None or isinstance(z, str) else unit(z)
        if abs(np.dot(self.x, self.z)) > 1e-8:
            print('x and z must be orthogonal, got xz={0:.4e}'.format(np.dot(self.x, self.z)))
        self.y = np.cross(self.z, self.x)

    def local_to_global_sph(self, phi, theta, **kwargs):
        """Angles on the sphere -> (x, y, z local, x, y, z global)."""
        lat, az = theta + self.thetaOffset, phi + self.phiOffset
        local = (np.cos(lat) * np.cos(az) * self.R, np.cos(lat) * np.sin(az) * self.R,
                 np.sin(lat) * self.R)
        return local + tuple(self.local_to_global(*local, **kwargs))

    def expose(self, beam=None, onlyPositivePath=False):
        _lib.require_gpu()
        dev = torch.device('cuda', torch.cuda.current_device())
        image = rs.Beam.empty_like_on_device(beam, dev)
        angles = [torch.empty(len(beam.x), dtype=torch.float64, device=dev) for _ in range(2)]
        rec = self._record(onlyPositivePath)
        rec.radius, rec.theta_offset, rec.phi_offset = \
            float(self.R), float(self.thetaOffset), float(self.phiOffset)
        rec.out_theta, rec.out_phi = angles[0].data_ptr(), angles[1].data_ptr()
        _lib.check(_lib.load().xrt_hip_screen_expose_f64_dev(
            ctypes.byref(rec), ctypes.byref(beam.to_struct(dev)),
            ctypes.byref(image.to_struct(dev)),
            _hipcalls.stream_ptr()),
            'xrt_hip_screen_expose_f64_dev')
        image._d['theta'], image._d['phi'] = angles
        rs.inherit_scalars(image, beam)
        return image

    def expose_global(self, beam=None):
        """The image with its positions back in the global frame."""
        glo = self.expose(beam)
        x, y, z = self.local_to_global_sph(glo.phi, glo.theta)[3:]
        glo.x, glo.y, glo.z = x, y, z
        return glo

    def prepare_wave(self, *args, **kwargs):
        raise NotImplementedError('HemisphericScreen.prepare_wave')
