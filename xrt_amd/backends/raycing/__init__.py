"""Host-side mirror of the part of ``xrt.backends.raycing`` that sits on the
accelerated hot path: module constants (xrt/backends/raycing/__init__.py:84-108),
the rotation primitives (_rotate.py:5-108), ``BeamLine`` and the global<->local
transforms (beamline.py:230-316, 407-478). Only what the path needs."""
import uuid

import numpy as np

# ray states, raycing/__init__.py:84
stateGood, stateOut, stateOver = 1, 2, 3
zEps = 1e-12
maxIteration = 100
dt = 1e-5
maxHalfSizeOfOE = 1000.
maxDepthOfOE = 100.
nrays = 100000
targetOpenCL = 'auto'
precisionOpenCL = 'auto'
_VERBOSITY_ = 0


def is_sequence(arg):
    return isinstance(arg, (list, tuple, np.ndarray))


def rotate_x(y, z, cosangle, sinangle):
    return cosangle*y - sinangle*z, sinangle*y + cosangle*z


def rotate_y(x, z, cosangle, sinangle):
    return cosangle*x + sinangle*z, -sinangle*x + cosangle*z


def rotate_z(x, y, cosangle, sinangle):
    return cosangle*x - sinangle*y, sinangle*x + cosangle*y


_AXIS = {'x': 0, 'y': 1, 'z': 2}


def rotation_steps(rotationSequence='RzRyRx', pitch=0, roll=0, yaw=0):
    """[(axis, cos, sin)] exactly as rotate_beam walks the sequence
    (_rotate.py:30-57): leading '-' reverses, zero angles are skipped and
    cos/sin are taken of the scalar angle on the host."""
    angles = {'z': yaw, 'y': roll, 'x': pitch}
    if rotationSequence[0] == '-':
        seq = rotationSequence[6] + rotationSequence[4] + rotationSequence[2]
    else:
        seq = rotationSequence[1] + rotationSequence[3] + rotationSequence[5]
    steps = []
    for s in seq:
        angle = angles[s]
        if angle != 0:
            steps.append((_AXIS[s], float(np.cos(angle)), float(np.sin(angle))))
    return steps


def rotate_xyz(x, y, z, indarr=None, rotationSequence='RzRyRx', pitch=0, roll=0,
               yaw=0):
    """In-place rotation of three host arrays (_rotate.py:60-82)."""
    if indarr is None:
        indarr = slice(None)
    for ax, cA, sA in rotation_steps(rotationSequence, pitch, roll, yaw):
        if ax == 2:
            x[indarr], y[indarr] = rotate_z(x[indarr], y[indarr], cA, sA)
        elif ax == 1:
            x[indarr], z[indarr] = rotate_y(x[indarr], z[indarr], cA, sA)
        else:
            y[indarr], z[indarr] = rotate_x(y[indarr], z[indarr], cA, sA)
    return x, y, z


def rotate_beam(beam, indarr=None, rotationSequence='RzRyRx', pitch=0, roll=0,
                yaw=0, skip_xyz=False, skip_abc=False, **kw):
    """Host-side rotate_beam for O(N) glue (wave pre/post-processing)."""
    if not skip_xyz:
        rotate_xyz(beam.x, beam.y, beam.z, indarr, rotationSequence, pitch, roll,
                   yaw)
    if not skip_abc:
        rotate_xyz(beam.a, beam.b, beam.c, indarr, rotationSequence, pitch, roll,
                   yaw)


def virgin_local_to_global(bl, vlb, center=None, part=None, skip_xyz=False,
                           skip_abc=False, **kw):
    """beamline.py:267-287 on host arrays."""
    if part is None:
        part = slice(None)
    a0, b0 = bl.sinAzimuth, bl.cosAzimuth
    if a0 != 0:
        if not skip_abc:
            vlb.a[part], vlb.b[part] = rotate_z(vlb.a[part], vlb.b[part], b0, -a0)
        if not skip_xyz:
            vlb.x[part], vlb.y[part] = rotate_z(vlb.x[part], vlb.y[part], b0, -a0)
    if (center is not None) and (not skip_xyz):
        vlb.x[part] += center[0]
        vlb.y[part] += center[1]
        vlb.z[part] += center[2]


def xyz_from_xz(obj, x=None, z=None):
    """Local axes of a screen from optional x and z directions
    (beamline.py:288-316)."""
    bl = obj.bl
    if isinstance(x, (list, tuple, np.ndarray)):
        norm = sum([xc**2 for xc in x])**0.5
        retx = [xc/norm for xc in x]
    else:
        if bl is None:
            retx = 1, 0, 0.
        else:
            retx = bl.cosAzimuth, -bl.sinAzimuth, 0.
    if isinstance(z, (list, tuple, np.ndarray)):
        norm = sum([zc**2 for zc in z])**0.5
        retz = [zc/norm for zc in z]
    else:
        retz = 0., 0., 1.
    xdotz = np.dot(retx, retz)
    if abs(xdotz) > 1e-8:
        raise ValueError('x and z must be orthogonal, got xz={0:.4e}'.format(xdotz))
    rety = np.cross(retz, retx)
    return [retx, rety, retz]


class BeamLine(object):
    """Container of beamline elements (beamline.py:407-478): azimuth, element
    lists that give each element its ordinal (lost rays get state
    -ordinal, oes/base.py:266-267)."""

    def __init__(self, azimuth=0., height=0., alignE='auto', name=''):
        self.azimuth = azimuth
        self.height = height
        self.alignE = alignE
        self.name = name
        self.sources = []
        self.oes = []
        self.slits = []
        self.screens = []
        self.alarms = []
        self.oesDict = {}

    @property
    def azimuth(self):
        return self._azimuth

    @azimuth.setter
    def azimuth(self, value):
        self._azimuth = value
        self.sinAzimuth = float(np.sin(value))
        self.cosAzimuth = float(np.cos(value))


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
    if angle is None or isinstance(angle, (list, tuple)):
        return angle
    return angle * defaultFactor


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
