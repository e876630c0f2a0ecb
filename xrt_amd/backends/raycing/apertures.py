"""Apertures with the interface of the reference's (xrt/backends/raycing/apertures.py):
``RectangularAperture`` (:75-541: four optional blades in the aperture's own frame),
``RoundAperture`` (:668-914) and their beam stops; ``propagate`` is one streaming HIP kernel
over a device-resident beam, ``prepare_wave`` serves the wave path."""
import ctypes

import numpy as np
import torch

from ... import hipcalls as _hipcalls

from .. import raycing
from ... import _lib, _structs
from . import sources as rs

# blade -> (which limit it sets, lower or upper end); also the bit order of the kernel
_BLADES = {'left': ('limOptX', 0), 'right': ('limOptX', 1),
           'bottom': ('limOptY', 0), 'top': ('limOptY', 1)}
_BLADE_ORDER = tuple(_BLADES)


def _stop_of(opening, name, doc):
    """The beam-stop twin of an aperture class: the nominal opening is the solid part."""
    def construct(self, *args, **kwargs):
        opening.__init__(self, *args, **kwargs)
        self.isBeamStop = True
    return type(name, (opening,), {'__init__': construct, '__doc__': doc})


class RectangularAperture(object):
    def __init__(self, bl=None, name='', center=[0, 0, 0],
                 kind=_BLADE_ORDER, opening=(-10, 10, -10, 10),
                 x='auto', z='auto', alarmLevel=None, blades=None, **kwargs):
        raycing.enrol(self, bl, 'slits', 1000, name, 'Aperture', kwargs.get('uuid'))
        self.center = center
        self.alarmLevel = alarmLevel
        self.isBeamStop = False
        if blades is None:        # parallel lists of blade names and positions
            names = [kind] if isinstance(kind, str) else kind
            edges = opening if raycing.is_sequence(opening) else [opening]
            blades = {b: e for b, e in zip(names, edges) if e is not None}
        unknown = set(blades) - set(_BLADES)
        if unknown:
            raise ValueError('unknown blade(s) {0}'.format(sorted(unknown)))
        self.blades = {b: blades[b] for b in _BLADE_ORDER if b in blades}
        RectangularAperture.set_optical_limits(self)    # (a subclass's own comes later)
        axes = [None if isinstance(v, str) else v for v in (x, z)]
        self.xyz = raycing.xyz_from_xz(self, *axes)
        self.x, self.y, self.z = self.xyz

    def set_optical_limits(self):
        """limOptX / limOptY (what prepare_wave samples and what the areas are taken from)
        from the blades (reference apertures.py:107-131)."""
        wide = raycing.maxHalfSizeOfOE
        self.limOptX, self.limOptY = [-wide, wide], [-wide, wide]
        for blade, edge in self.blades.items():
            limit, end = _BLADES[blade]
            getattr(self, limit)[end] = float(edge)

    # kind / opening as in the reference: parallel lists that can be assigned, e.g. in a scan
    @property
    def kind(self):
        return list(self.blades)

    @kind.setter
    def kind(self, names):
        names = [names] if isinstance(names, str) else list(names)
        edges = self.opening
        if len(edges) != len(names):
            raise ValueError('`kind` and `opening` must have equal lengths')
        self._set_blades(names, edges)

    @property
    def opening(self):
        return list(self.blades.values())

    @opening.setter
    def opening(self, edges):
        edges = list(edges) if raycing.is_sequence(edges) else [edges]
        names = self.kind
        if len(edges) != len(names):
            raise ValueError('`kind` and `opening` must have equal lengths')
        self._set_blades(names, edges)

    def _set_blades(self, names, edges):
        unknown = set(names) - set(_BLADES)
        if unknown:
            raise ValueError('unknown blade(s) {0}'.format(sorted(unknown)))
        given = dict(zip(names, edges))
        self.blades = {b: given[b] for b in _BLADE_ORDER if b in given}
        RectangularAperture.set_optical_limits(self)

    def local_to_global(self, glo, returnBeam=False, **kwargs):
        """Positions and directions of a host beam from the aperture's frame to the
        global one, in place (reference apertures.py:436-457)."""
        axes = (self.x, self.y, self.z)
        glo.x, glo.y, glo.z = raycing.along_basis(axes, glo.x, glo.y, glo.z, self.center)
        glo.a, glo.b, glo.c = raycing.along_basis(axes, glo.a, glo.b, glo.c)

    def _record(self):
        a = _structs.Aperture()
        for k in range(3):
            a.center[k], a.ex[k], a.ey[k], a.ez[k] = (
                float(self.center[k]), float(self.x[k]), float(self.y[k]), float(self.z[k]))
        a.sin_az, a.cos_az = (0., 1.) if self.bl is None else \
            (self.bl.sinAzimuth, self.bl.cosAzimuth)
        a.blade_mask = 0
        for bit, blade in enumerate(_BLADE_ORDER):
            if blade in self.blades:
                a.blade_mask |= 1 << bit
                a.blade[bit] = float(self.blades[blade])
        a.is_beam_stop = int(bool(self.isBeamStop))
        a.lost_num = int(self.lostNum)
        if hasattr(self, 'r'):
            a.round, a.radius, a.blade_mask = 1, float(self.r), 0
        if hasattr(self, 'shadeFraction'):      # the band between the two slits
            low = (1 - self.shadeFraction) * 0.5
            bottom, top = self.blades['bottom'], self.blades['top']
            a.has_shade, a.glo_adds_path = 1, 1
            a.shade[0] = bottom + (top - bottom) * low
            a.shade[1] = bottom + (top - bottom) * (low + self.shadeFraction)
        if hasattr(self, 'vertices'):
            outline = np.ascontiguousarray(self.vertices, dtype=np.float64).reshape(-1, 2)
            held = self.__dict__.setdefault('_outline', {})
            dev = torch.device('cuda', torch.cuda.current_device())
            if str(dev) not in held or not np.array_equal(held[str(dev)][0], outline):
                held[str(dev)] = (outline.copy(), torch.from_numpy(outline.copy()).to(dev))
            a.poly_n, a.poly_xz, a.blade_mask = len(outline), held[str(dev)][1].data_ptr(), 0
        return a

    def propagate(self, beam=None, needNewGlobal=False):
        """*beam* (global frame) carried to the aperture plane; rays the blades stop
        get state ``lostNum`` -- also in *beam* itself, as in the reference
        (apertures.py:334-413). Returns the beam in the aperture's frame, preceded by
        the new global beam if *needNewGlobal*."""
        _lib.require_gpu()
        dev = torch.device('cuda', torch.cuda.current_device())
        from . import oes as roe
        op = beam.__dict__.get('_op') if type(beam) is rs.LazyBeam else None
        if isinstance(op, roe._DeferredReflect) and op.gb is beam and roe.fuseConsumers and \
                not needNewGlobal and op.takes_aperture(self):
            # the element's pass has not been launched: the marks are made in its tail
            return op.marks_later(self)
        beam.to_struct(dev)
        shot = None
        if roe.fuseConsumers and not needNewGlobal and type(beam) is rs.Beam and \
                not hasattr(self, 'vertices'):
            # a screen that was exposed to this very beam and has not been launched: its image
            # and this aperture's marks are one pass over the rays (a front-end monitor and the
            # mask behind it; screens._DeferredExpose)
            from . import screens as rsc
            shot = rsc.pending_expose_of(beam, dev)
        rs.flush_pending(beam, keep=shot, only_state=True)   # (beam.state changes in place below)
        rs.before_states_change(beam)
        if not needNewGlobal and roe.fuseConsumers:
            return _DeferredLocal(self, beam, dev, shot).hand_out()
        local = rs.Beam.empty_like_on_device(beam, dev)
        glo = rs.Beam.empty_like_on_device(beam, dev) if needNewGlobal else None
        rec = self._record()
        _lib.check(_lib.load().xrt_hip_aperture_propagate_f64_dev(
            ctypes.byref(rec), ctypes.byref(beam.to_struct(dev)),
            ctypes.byref(local.to_struct(dev)),
            ctypes.byref(glo.to_struct(dev)) if glo is not None else None,
            _hipcalls.stream_ptr()),
            'xrt_hip_aperture_propagate_f64_dev')
        beam._h.pop('state', None)       # the kernel updated beam.state in HBM
        rs.inherit_scalars(local, beam)
        if glo is None:
            return local
        rs.inherit_scalars(glo, beam)
        return glo, local

    def prepare_wave(self, prevOE, nrays, rw=None):
        """*nrays* receiving samples, uniformly random over the open rectangle (one
        (nrays, 2) draw from numpy's global generator: column 0 across, column 1 up --
        the reference's order of consumption, apertures.py:467-499)."""
        if rw is None:
            from . import waves as rw
        count = int(nrays)
        draw = np.random.rand(count, 2)
        spans = [lim[1] - lim[0] for lim in (self.limOptX, self.limOptY)]
        px = draw[:, 0] * spans[0] + self.limOptX[0]
        pz = draw[:, 1] * spans[1] + self.limOptY[0]
        if torch.cuda.is_available():
            # the points go up once; the wave is made and stays on the GPU (the same
            # elementwise operations in the same order as with host arrays)
            dev = torch.device('cuda', torch.cuda.current_device())
            px, pz = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (px, pz))
            py = torch.zeros(count, dtype=torch.float64, device=dev)
        else:
            py = np.zeros(count)
        there = raycing.along_basis((self.x, self.y, self.z), px, py, pz, self.center)
        opened = spans[0] * spans[1]
        return rw.receiving_wave(self, prevOE, (px, py, pz), there, opened / count,
                                 opened, self.uuid)


class _DeferredLocal(rs.SharesStates, rs.FillsBeams):
    """``propagate`` when only the states are certain to be needed: ONE launch reads the
    geometry and marks the stopped rays in the incoming beam (52 B read, <= 4 B written per ray
    -- out_local NULL in xrt_hip_aperture_propagate_f64_dev); the beam in the aperture's frame
    (another ~100 B read and 100 B written per ray) is made by the full kernel, from the very
    same arrays and a copy of the states as they were, when somebody first looks at it --
    the same bits, or no launch at all if nobody does (a slit between two mirrors whose local
    beam no plot shows)."""
    optional = True

    def __init__(self, aperture, beam, dev, shot=None):
        self.aperture, self.device = aperture, dev
        self.shot = shot                 # a screen's pending launch on the same rays
        self.record = aperture._record()
        # (the record points at the polygon's vertices in HBM: they live as long as it does,
        # whatever the script assigns to aperture.vertices in the meantime)
        self._keep = list(aperture.__dict__.get('_outline', {}).values())
        # the rays as they are NOW: the same tensors (whoever writes into them in place flushes
        # the readers first, sources.flush_pending) and the states before this aperture
        was = rs.Beam.__new__(rs.Beam)
        object.__setattr__(was, '_h', {})
        object.__setattr__(was, '_d', dict(beam._d))
        object.__setattr__(was, 'parentId', None)
        # the states the later launch starts from: as THIS launch leaves them (the rays it stops
        # carry its own number: own_marks), shared until another aperture is about to mark the
        # same beam. A polygon also relabels rays that were dead already, and a beam that has
        # been through this aperture before holds its number from then: a copy of the states as
        # they are now, as before round 6.
        seen = beam.__dict__.setdefault('_stopped_by', set())
        self.own_marks = not hasattr(aperture, 'vertices') and aperture.lostNum not in seen
        seen.add(aperture.lostNum)
        if not self.own_marks:
            was._d['state'] = beam._d['state'].clone()
        self.was = was
        self.tensors = {id(t) for t in beam._d.values()}
        self.state = 'pending'
        self._launch(beam, None)
        if self.own_marks:
            self._share_states(was)
        beam._h.pop('state', None)       # the kernel updated beam.state in HBM
        rs.inherit_scalars(self._make('local'), beam)
        rs._PENDING.add(self)

    def _launch(self, beam, local):
        shot, self.shot = self.shot, None
        if shot is not None and local is None:
            return shot.with_marks(self.record, beam)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().xrt_hip_aperture_propagate_f64_dev(
                ctypes.byref(self.record), ctypes.byref(beam.to_struct(self.device)),
                ctypes.byref(local.to_struct(self.device)) if local is not None else None,
                None, _hipcalls.stream_ptr()),
                'xrt_hip_aperture_propagate_f64_dev')

    def reads(self, beam):
        d = beam.__dict__.get('_real_d', beam.__dict__.get('_d')) or {}
        return any(id(t) in self.tensors for t in d.values())

    def materialize(self, which=None):
        rs._PENDING.discard(self)
        if self.state != 'done':
            local = rs.Beam.empty_like_on_device(self.was, self.device)
            self.record.own_marks = int(self.own_marks)
            self._launch(self.was, local)     # (one that raises is raised again by the next look)
            self.state = 'done'
            rs.adopt_into(self._beam('local'), local)
            self.was, self.tensors = None, ()


class SetOfRectangularAperturesOnZActuator(RectangularAperture):
    """Several openings in one plate that a vertical actuator moves into the beam
    (reference apertures.py:555-665): *apertures* = their names, the last of which is the
    plate's edge ('bottom-edge' or 'top-edge'); *centerZs* = heights of their centres (of
    the edge) relative to center[2]; *dXs*, *dZs* = sizes of the openings.
    ``select_aperture(name, targetZ)`` puts that opening at the height *targetZ*."""

    def __init__(self, bl, name, center, apertures, centerZs, dXs, dZs, x='auto', z='auto',
                 alarmLevel=None):
        RectangularAperture.__init__(self, bl, name, center, blades={}, x=x, z=z,
                                     alarmLevel=alarmLevel)
        self.zActuator = self.z0 = center[2]
        self.apertures, self.centerZs, self.dXs, self.dZs = apertures, centerZs, dXs, dZs
        self.surface, self.zlims = apertures, None
        half = [w * 0.5 for w in dXs]
        self.limOptX = [[-h for h in half] + [-500], half + [500]]
        self.limOptY = [0, 0]
        self.shape, self.spotLimits = 'rect', [0, 0, 0, 0]

    def select_aperture(self, apertureName, targetZ):
        which = self.apertures.index(apertureName)
        self.curAperture = which
        level = self.bl.height
        if which < len(self.apertures) - 1:
            hx, hz = self.dXs[which] * 0.5, self.dZs[which] * 0.5
            mid = targetZ - level
            self.blades = {'left': -hx, 'right': hx, 'bottom': mid-hz, 'top': mid+hz}
            self.zActuator = self.z0 + targetZ - self.centerZs[which]
        else:
            edge = self.apertures[-1]
            if edge not in ('top-edge', 'bottom-edge'):
                raise ValueError('not "top-edge" nor "bottom-edge"!')
            self.blades = {'bottom' if edge == 'top-edge' else 'top':
                           self.centerZs[-1] - level}
            self.zActuator = self.z0
        reach = max(self.dZs) * 0.5
        moved = self.zActuator - self.z0
        self.zlims = [min(min(self.centerZs) + moved, targetZ) - level - reach,
                      max(max(self.centerZs) + moved, targetZ) - level + reach]
        self.set_optical_limits()

    def set_optical_limits(self):
        """Outlines of all openings at the present actuator position (for footprints)."""
        shift = -self.bl.height + self.zActuator - self.z0
        pairs = list(zip(self.centerZs, self.dZs))
        self.limOptY[0] = [cz + shift - dz*0.5 for cz, dz in pairs] + [self.centerZs[-1] + shift]
        self.limOptY[1] = [cz + shift + dz*0.5 for cz, dz in pairs] + [200]


RectangularBeamStop = _stop_of(RectangularAperture, "RectangularBeamStop", """The blades enclose the solid part: rays inside are stopped, rays outside pass.""")


class RoundAperture(RectangularAperture):
    """A pipe or a flange: open within the radius *r* around the centre."""

    def __init__(self, bl=None, name='', center=[0, 0, 0], r=1, x='auto', z='auto',
                 alarmLevel=None, **kwargs):
        RectangularAperture.__init__(self, bl, name, center, blades={}, x=x, z=z,
                                     alarmLevel=alarmLevel, **kwargs)
        self.r = r
        self.limOptX, self.limOptY, self.shape = [-r, r], [-r, r], 'round'

    def get_divergence(self, source):
        """Full angle the aperture subtends at *source*."""
        gap = np.subtract(self.center, source.center)
        return self.r * 2 * np.dot(gap, gap) ** -0.5

    def prepare_wave(self, prevOE, nrays, rw=None):
        """*nrays* receiving samples uniform over the disc: one (nrays, 2) draw, column 0
        -> radius (its square root), column 1 -> azimuth (apertures.py:848-874)."""
        if not rw:
            from . import waves as rw
        count = int(nrays)
        draw = np.random.rand(count, 2)
        radius = draw[:, 0]**0.5 * self.r
        angle = draw[:, 1] * 2*np.pi
        px, pz, py = radius * np.cos(angle), radius * np.sin(angle), np.zeros(count)
        there = raycing.along_basis((self.x, self.y, self.z), px, py, pz, self.center)
        disc = np.pi * self.r**2
        return rw.receiving_wave(self, prevOE, (px, py, pz), there, disc / count, disc,
                                 self.uuid)


RoundBeamStop = _stop_of(RoundAperture, "RoundBeamStop", """A disc of radius *r* in the beam: rays inside are stopped.""")


class DoubleSlit(RectangularAperture):
    """Two slits one above the other: the opening between the bottom and the top blade
    with its middle *shadeFraction* (0..1) opaque."""

    def __init__(self, *args, **kwargs):
        self.shadeFraction = kwargs.pop('shadeFraction', 0.5)
        RectangularAperture.__init__(self, *args, **kwargs)
        if not {'bottom', 'top'} <= set(self.blades):
            raise ValueError('a DoubleSlit needs a bottom and a top blade')


DoubleBeamStop = _stop_of(DoubleSlit, "DoubleBeamStop", """The two slit openings are the solid parts.""")


class PolygonalAperture(RectangularAperture):
    """Open inside the polygon of *vertices* [(x0, z0), ...] in the aperture's plane
    (*opening* is an alias); a point on the outline counts as matplotlib's
    ``Path.contains_points`` counts it."""

    def __init__(self, bl=None, name='', center=[0, 0, 0], opening=None, x='auto', z='auto',
                 alarmLevel=None, vertices=((-10, -10), (-10, 10), (10, 10), (10, -10)),
                 **kwargs):
        RectangularAperture.__init__(self, bl, name, center, blades={}, x=x, z=z,
                                     alarmLevel=alarmLevel, **kwargs)
        self.vertices = [tuple(v) for v in (opening if opening is not None else vertices)]
        self.shape = 'polygon'

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name == 'vertices':       # (NaN rows separate the cells of a grid)
            corners = np.array(value, dtype=float)
            self.limOptX = [np.nanmin(corners[:, 0]), np.nanmax(corners[:, 0])]
            self.limOptY = [np.nanmin(corners[:, 1]), np.nanmax(corners[:, 1])]

    def prepare_wave(self, prevOE, nrays, rw=None):
        raise NotImplementedError('wave samples on a polygonal aperture')


PolygonalBeamStop = _stop_of(PolygonalAperture, "PolygonalBeamStop", """The polygon is the solid part.""")


class GridAperture(PolygonalAperture):
    """A regular grid of rectangular openings *dx* x *dz* at the pitches *px*, *pz*:
    (2 nx + 1) x (2 nz + 1) of them about the centre (reference apertures.py:1324-1447).
    One outline: the closed rectangles, separated by NaN rows."""
    _SHAPE = ('dx', 'dz', 'px', 'pz', 'nx', 'nz')

    def __init__(self, bl=None, name='', center=[0, 0, 0], x='auto', z='auto', alarmLevel=None,
                 dx=0.5, dz=0.5, px=1.0, pz=1.0, nx=7, nz=7, **kwargs):
        object.__setattr__(self, '_grid', dict(dx=dx, dz=dz, px=px, pz=pz, nx=int(nx),
                                               nz=int(nz)))
        PolygonalAperture.__init__(self, bl=bl, name=name, center=center, x=x, z=z,
                                   alarmLevel=alarmLevel, vertices=self._cells(), **kwargs)

    def _cells(self):
        q = self._grid
        # one cell, walked from its (+, +) corner and closed, then the separator
        walk_x = np.array([q['dx'], -q['dx'], -q['dx'], q['dx'], q['dx'], np.nan]) * 0.5
        walk_z = np.array([q['dz'], q['dz'], -q['dz'], -q['dz'], q['dz'], np.nan]) * 0.5
        at_x = np.linspace(-1, 1, 2*q['nx'] + 1) * q['px'] * q['nx']
        at_z = np.linspace(-1, 1, 2*q['nz'] + 1) * q['pz'] * q['nz']
        grid_x, grid_z = np.meshgrid(at_x, at_z)
        xs = (grid_x.ravel(order='F') + walk_x[:, np.newaxis]).ravel(order='F')
        zs = (grid_z.ravel(order='F') + walk_z[:, np.newaxis]).ravel(order='F')
        return [tuple(v) for v in np.column_stack((xs, zs))]

    def __getattr__(self, name):
        grid = self.__dict__.get('_grid')
        if grid is not None and name in grid:
            return grid[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in self._SHAPE:
            self._grid[name] = int(value) if name in ('nx', 'nz') else value
            self.vertices = self._cells()
        else:
            PolygonalAperture.__setattr__(self, name, value)

    def get_render_cells(self):
        """(xmin, xmax, zmin, zmax) of every opening."""
        rows = np.array(self.vertices, dtype=float).reshape(-1, 6, 2)[:, :4]
        return [(c[:, 0].min(), c[:, 0].max(), c[:, 1].min(), c[:, 1].max()) for c in rows]


GridBeamStop = _stop_of(GridAperture, "GridBeamStop", """The cells of the grid are the solid parts.""")


class SiemensStar(PolygonalAperture):
    """Star of *nSpokes* open sectors of radius *r* (or semi-axes *rx*, *rz*), turned by
    *phi0*; *vortex* bends the spokes (by *vortex* spoke positions at the rim, drawn in
    *vortexNradial* segments) (reference apertures.py:1462-1528). One closed outline
    through the centre."""

    def __init__(self, bl=None, name='', center=[0, 0, 0], x='auto', z='auto', alarmLevel=None,
                 nSpokes=9, r=1, rx=0, rz=0, phi0=0, vortex=0, vortexNradial=7, **kwargs):
        self.nSpokes, self.phi0 = nSpokes, phi0
        self.rx, self.rz = (r, r) if r else (rx, rz)
        edges = np.linspace(0, 2*np.pi, nSpokes*2, endpoint=False) - np.pi/nSpokes/2 - phi0
        if vortex:
            # every spoke: up its first edge ring by ring, down its second edge
            cols_x, cols_z = [], []
            for ring in reversed(range(vortexNradial)):
                part = (ring + 1.) / vortexNradial
                twist = 2*np.pi * vortex / nSpokes * part
                px = (part * self.rx * np.sin(edges + twist)).reshape((nSpokes, 2))
                pz = (part * self.rz * np.cos(edges + twist)).reshape((nSpokes, 2))
                cols_x = [px[:, 0:1]] + cols_x + [px[:, 1:2]]
                cols_z = [pz[:, 0:1]] + cols_z + [pz[:, 1:2]]
        else:
            cols_x = [(self.rx * np.sin(edges)).reshape((nSpokes, 2))]
            cols_z = [(self.rz * np.cos(edges)).reshape((nSpokes, 2))]
        hub = np.zeros((nSpokes, 1))
        outline = zip(np.hstack(cols_x + [hub]).flatten(), np.hstack(cols_z + [hub]).flatten())
        PolygonalAperture.__init__(self, bl=bl, name=name, center=center, x=x, z=z,
                                   alarmLevel=alarmLevel, vertices=list(outline), **kwargs)

