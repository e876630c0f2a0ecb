"""Host side of the accelerated path, laid out like ``xrt.backends.raycing`` so that
beamline scripts read the same: the module constants (reference
raycing/__init__.py:84-108), plane rotations in the reference's order of operations
(_rotate.py), ``BeamLine`` and the frame changes between the global frame and an
element's frames (beamline.py:230-316, 407-478). Only what the path needs.

Every frame change here is a short list of elementary steps -- (axis, cos, sin) plane
rotations and shifts -- interpreted by ``turn``: the kernels of csrc/reflect.hip take the
same lists (``xrt_hip_rotation``), so host and device walk one description and round
the same way."""
import uuid

import functools as _functools

import numpy as np

# ray states and solver constants
stateGood, stateOut, stateOver = range(1, 4)
zEps, dt, maxIteration = 1e-12, 1e-5, 100
maxHalfSizeOfOE, maxDepthOfOE = 1000., 100.      # bracket sizes [mm] without limits
nrays = 10**5
targetOpenCL = precisionOpenCL = 'auto'          # accepted by the classes, unused
_VERBOSITY_ = 0

_AXIS = {'x': 0, 'y': 1, 'z': 2}
# the two coordinates an axis turns into each other, and the sense of the sine:
# x: (y, z), y: (x, z) with the sine reversed, z: (x, y)
_PLANE = ((1, 2, 1.), (0, 2, -1.), (0, 1, 1.))


def is_sequence(arg):
    return isinstance(arg, (list, tuple, np.ndarray))


def _in_plane(u, v, c, s):
    """(u, v) turned by the angle whose cosine and sine are c, s."""
    return u*c - v*s, u*s + v*c


def rotate_x(y, z, cosangle, sinangle):
    return _in_plane(y, z, cosangle, sinangle)


def rotate_y(x, z, cosangle, sinangle):
    return _in_plane(x, z, cosangle, -sinangle)


def rotate_z(x, y, cosangle, sinangle):
    return _in_plane(x, y, cosangle, sinangle)


def rotation_steps(rotationSequence='RzRyRx', pitch=0, roll=0, yaw=0):
    """[(axis, cos, sin)] in the order the reference's rotate_beam applies them: the
    letters of e.g. 'RzRyRx' left to right, right to left behind a leading '-';
    zero angles drop out; cos/sin of the scalar angle are taken here, once."""
    try:
        return list(_rotation_steps(rotationSequence, pitch, roll, yaw))
    except TypeError:           # (an angle that does not hash: an array of one)
        return list(_rotation_steps.__wrapped__(rotationSequence, pitch, roll, yaw))


@_functools.lru_cache(maxsize=4096)
def _rotation_steps(rotationSequence, pitch, roll, yaw):
    # (a pure function of four numbers that every element call asks for again: numpy's scalar
    # cos / sin are 0.7 us each, twelve of them per pass record)
    angle_of = (pitch, roll, yaw)
    letters = [ch for ch in rotationSequence if ch in _AXIS]
    if rotationSequence.startswith('-'):
        letters.reverse()
    return tuple((_AXIS[ch], float(np.cos(angle_of[_AXIS[ch]])),
                  float(np.sin(angle_of[_AXIS[ch]])))
                 for ch in letters if angle_of[_AXIS[ch]] != 0)


def turn(triple, steps, part=None):
    """Applies the plane rotations *steps* to the three host arrays of *triple*
    (x, y, z or a, b, c) in place, on the rays *part*."""
    part = slice(None) if part is None else part
    for axis, c, s in steps:
        i, j, sense = _PLANE[axis]
        triple[i][part], triple[j][part] = _in_plane(triple[i][part], triple[j][part],
                                                     c, sense*s)
    return triple


def rotate_xyz(x, y, z, indarr=None, rotationSequence='RzRyRx', pitch=0, roll=0,
               yaw=0):
    turn([x, y, z], rotation_steps(rotationSequence, pitch, roll, yaw), indarr)
    return x, y, z


def rotate_point(point, rotationSequence='RzRyRx', pitch=0, roll=0, yaw=0):
    """The 3-sequence *point* through one rotation sequence -> list of three numbers."""
    held = [np.array([float(c)]) for c in point]
    turn(held, rotation_steps(rotationSequence, pitch, roll, yaw))
    return [float(c[0]) for c in held]


def rotate_beam(beam, indarr=None, rotationSequence='RzRyRx', pitch=0, roll=0,
                yaw=0, skip_xyz=False, skip_abc=False, **kw):
    """Positions and directions of a host-resident beam through one rotation sequence."""
    steps = rotation_steps(rotationSequence, pitch, roll, yaw)
    for skipped, names in ((skip_xyz, 'xyz'), (skip_abc, 'abc')):
        if not skipped:
            turn([getattr(beam, n) for n in names], steps, indarr)


def virgin_local_to_global(bl, vlb, center=None, part=None, skip_xyz=False,
                           skip_abc=False, **kw):
    """Out of an element's virgin local frame: back through the beamline azimuth about z,
    then out to *center* (host arrays, in place)."""
    undo_azimuth = [(2, bl.cosAzimuth, -bl.sinAzimuth)] if bl.sinAzimuth != 0 else []
    if not skip_abc:
        turn([vlb.a, vlb.b, vlb.c], undo_azimuth, part)
    if not skip_xyz:
        xyz = turn([vlb.x, vlb.y, vlb.z], undo_azimuth, part)
        if center is not None:
            part = slice(None) if part is None else part
            for arr, c0 in zip(xyz, center):
                arr[part] += c0


def _unit(vec):
    length = sum(c*c for c in vec)**0.5
    return [c/length for c in vec]


def xyz_from_xz(obj, x=None, z=None):
    """The three local axes of a screen or an aperture in global coordinates. Defaults:
    z up, x horizontal across the beamline direction; y completes the right-handed
    triad. *x*, *z* given as 3-sequences are normalised and must be orthogonal."""
    bl = obj.bl
    if is_sequence(x):
        ex = _unit(x)
    elif bl is None:
        ex = 1, 0, 0.
    else:
        ex = bl.cosAzimuth, -bl.sinAzimuth, 0.
    ez = _unit(z) if is_sequence(z) else (0., 0., 1.)
    skew = np.dot(ex, ez)
    if abs(skew) > 1e-8:
        raise ValueError('x and z must be orthogonal, got xz={0:.4e}'.format(skew))
    return [ex, np.cross(ez, ex), ez]


class BeamLine(object):
    """The list of elements along one beam path. Elements register themselves in the
    roster of their kind, which numbers them: a ray lost at element #n of its roster
    carries state -n (OEs), -1000-n (apertures), -2000-n (screens)."""
    rosters = ('sources', 'oes', 'slits', 'screens', 'alarms')

    def __init__(self, azimuth=0., height=0., alignE='auto', name=''):
        for roster in self.rosters:
            setattr(self, roster, [])
        self.oesDict = {}
        self.name, self.height, self.alignE = name, height, alignE
        self.azimuth = azimuth

    def _turn_to(self, angle):
        self._azimuth = angle
        self.cosAzimuth, self.sinAzimuth = float(np.cos(angle)), float(np.sin(angle))

    azimuth = property(lambda self: self._azimuth, _turn_to,
                       doc='angle of the beam path about the global z, from the y axis')


def enrol(element, bl, roster, lost_offset, name, stem, uuid_=None):
    """Registration of *element* on beamline *bl* (None: stand-alone): its number in
    the roster, the state its lost rays get, name and uuid."""
    element.bl = bl
    members = getattr(bl, roster) if bl is not None else None
    if members is None:
        element.ordinalNum = 1
    elif element not in members:
        members.append(element)
        element.ordinalNum = len(members)
    element.lostNum = -element.ordinalNum - lost_offset
    element.name = name or '{0}{1}'.format(stem, element.ordinalNum)
    element.uuid = uuid_ or new_uuid()
    if bl is not None:
        bl.oesDict[element.uuid] = [element, 1]


_ANGLE_UNITS = (('mrad', 1e-3), ('urad', 1e-6), ('nrad', 1e-9), ('rad', 1.),
                ('deg', np.pi / 180.))


def auto_units_angle(angle, defaultFactor=1.):
    """Angle given as a number (times *defaultFactor*) or as a string with a
    unit suffix: '2mrad', '10 urad', '0.5deg' (reference: _flow_utils.py:74-103)."""
    if isinstance(angle, str):
        text = angle.strip()
        if 'auto' in text:
            return angle
        for unit, factor in _ANGLE_UNITS:
            if unit in text:
                value = float(text[:text.index(unit[0])].strip())
                return np.radians(value) if unit == 'deg' else value * factor
        return float(text) * defaultFactor
    plain_number = not (angle is None or isinstance(angle, (list, tuple)))
    return angle * defaultFactor if plain_number else angle


def along_basis(basis, u, v, w, origin=None):
    """u*ex + v*ey + w*ez (+ origin) for the three basis vectors of a screen or
    an aperture, in the reference's summation order (component by component:
    ((o + u ex) + v ey) + w ez)."""
    ex, ey, ez = basis
    out = []
    for k in range(3):
        if origin is None:
            out.append(u*ex[k] + v*ey[k] + w*ez[k])
        else:
            out.append(origin[k] + u*ex[k] + v*ey[k] + w*ez[k])
    return out


def new_uuid():
    return str(uuid.uuid4())
