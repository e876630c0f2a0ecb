"""TEST INFRASTRUCTURE ONLY — numpy restatement of xrt's ray-surface hot path
(the P1 oracle): frame transforms, bracketing, the bracketed secant / Brent
root solve, state classification, direction update, coherency-matrix rotation,
Fresnel and Bragg amplitudes.

Same arithmetic, same operation order as the reference so that hit states come
out bit-identical; the code is organised differently (plain functions over a
small Beam record and parameter dictionaries). Reference anchors, relative to
xrt/backends/raycing/:

* rotate_*/rotate_beam          <- _rotate.py:5-57
* global_to_virgin_local etc.   <- beamline.py:230-287
* bracket                       <- oes/base.py:1231-1295 (_set_t, _bracketing)
* find_dz / find_intersection   <- oes/base.py:801-885
* secant / brent                <- oes/base.py:933-959 / 961-1048
* rays_good                     <- oes/base.py:1094-1163
* flat / toroid surface         <- oes/base.py:675-742, oes/__init__.py:398-411
* reflect_local                 <- oes/reflect.py:551-1139
* oe_reflect                    <- oes/reflect.py:18-163
* dcm_double_reflect            <- oes/dcm.py:248-354
* rotate_coherency_matrix       <- sources/beams.py:448-479
* material / crystal amplitudes <- see oracle/materials_np.py

* blazed grating surface        <- oes/gratings.py:461-522 (local_pre/z/n and the
                                   ad hoc first-facet find_intersection)
* elliptical mirror (parametric)<- oes/parametric.py:117-157, 213-249; the
                                   parametric branches of find_dz (base.py:822-841)
                                   and _reflect_local (reflect.py:676-703, 1066-1071)

Parity pinned: tests/golden/g2_*.npz, g3_*.npz (oracle/gen_fixtures_p1.py,
oracle/gen_fixtures_softi.py).
Supported subset: rectangular / round OEs; flat, toroidal, bent-flat, blazed
(constant line density) and elliptical-parametric surfaces; no figure error, no
grating-equation materials / multilayers / mosaicity (SURVEY 2.1 OOS rows).
"""
import copy

import numpy as np

from . import materials_np as mat
from .consts import CH, CHBAR

zEps = 1e-12            # raycing/__init__.py:86
maxIteration = 100      # :88
dt = 1e-5               # :90 bracket margin [mm]
ds = 0.                 # :91 margin used in multiple reflections [mm]
maxHalfSizeOfOE = 1000.  # :92
maxDepthOfOE = 100.      # :94

F64 = ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp')


class Beam(object):
    """SoA ray record (sources/beams.py:153-182)."""

    def __init__(self, n=0, with_amplitudes=False):
        for f in F64:
            setattr(self, f, np.zeros(n))
        self.b[:] = 1.
        self.Jss[:] = 1.
        self.Jsp = np.zeros(n, dtype=complex)
        self.state = np.zeros(n, dtype=np.int32)
        if with_amplitudes:
            self.Es = np.zeros(n, dtype=complex)
            self.Ep = np.zeros(n, dtype=complex)

    def copy(self):
        return copy.deepcopy(self)

    def concatenate(self, beam):
        """sources/beams.py:230-294: the arrays both beams hold, one after the other."""
        always = list(F64) + ['Jsp', 'state']
        optional = ['nRefl', 'elevationD', 'elevationX', 'elevationY', 'elevationZ', 's',
                    'phi', 'r', 'theta', 'order', 'Es', 'Ep']
        for name in always + [o for o in optional
                              if hasattr(self, o) and hasattr(beam, o)]:
            setattr(self, name, np.concatenate((getattr(self, name), getattr(beam, name))))

    def fields(self):
        names = list(F64) + ['Jsp', 'state']
        if hasattr(self, 'Es'):
            names += ['Es', 'Ep']
        return names

    @staticmethod
    def from_dict(d, prefix=''):
        n = len(d[prefix + 'x'])
        b = Beam(n, with_amplitudes=(prefix + 'Es') in d)
        for f in b.fields():
            setattr(b, f, np.array(d[prefix + f]))
        if (prefix + 'theta') in d:
            b.theta = np.array(d[prefix + 'theta'])
        return b


# --------------------------------------------------------------------------
# plane rotations (_rotate.py:5-20)
# --------------------------------------------------------------------------
def rotate_x(y, z, cosangle, sinangle):
    return cosangle*y - sinangle*z, sinangle*y + cosangle*z


def rotate_y(x, z, cosangle, sinangle):
    return cosangle*x + sinangle*z, -sinangle*x + cosangle*z


def rotate_z(x, y, cosangle, sinangle):
    return cosangle*x - sinangle*y, sinangle*x + cosangle*y


def rotation_steps(rotationSequence, pitch, roll, yaw):
    """The list of (axis, cos, sin) that rotate_beam applies
    (_rotate.py:30-57): axes in the order of the sequence string (reversed for
    a leading '-'), zero angles skipped, cos/sin of the scalar angle."""
    angles = {'z': yaw, 'y': roll, 'x': pitch}
    if rotationSequence[0] == '-':
        seq = rotationSequence[6] + rotationSequence[4] + rotationSequence[2]
    else:
        seq = rotationSequence[1] + rotationSequence[3] + rotationSequence[5]
    steps = []
    for s in seq:
        angle = angles[s]
        if angle != 0:
            steps.append((s, np.cos(angle), np.sin(angle)))
    return steps


def apply_steps(steps, b, idx, xyz=True, abc=True):
    for s, cA, sA in steps:
        if s == 'z':
            if xyz:
                b.x[idx], b.y[idx] = rotate_z(b.x[idx], b.y[idx], cA, sA)
            if abc:
                b.a[idx], b.b[idx] = rotate_z(b.a[idx], b.b[idx], cA, sA)
        elif s == 'y':
            if xyz:
                b.x[idx], b.z[idx] = rotate_y(b.x[idx], b.z[idx], cA, sA)
            if abc:
                b.a[idx], b.c[idx] = rotate_y(b.a[idx], b.c[idx], cA, sA)
        else:
            if xyz:
                b.y[idx], b.z[idx] = rotate_x(b.y[idx], b.z[idx], cA, sA)
            if abc:
                b.b[idx], b.c[idx] = rotate_x(b.b[idx], b.c[idx], cA, sA)


def rotate_beam(b, idx, rotationSequence='RzRyRx', pitch=0, roll=0, yaw=0):
    apply_steps(rotation_steps(rotationSequence, pitch, roll, yaw), b, idx)


# --------------------------------------------------------------------------
# global <-> virgin local (beamline.py:230-287)
# --------------------------------------------------------------------------
def global_to_virgin_local(azimuth_sc, beam, lo, center, part):
    a0, b0 = azimuth_sc           # (sinAzimuth, cosAzimuth)
    lo.x[part] = beam.x[part] - center[0]
    lo.y[part] = beam.y[part] - center[1]
    lo.z[part] = beam.z[part] - center[2]
    if a0 == 0:
        lo.a[part] = beam.a[part]
        lo.b[part] = beam.b[part]
    else:
        lo.x[part], lo.y[part] = rotate_z(lo.x[part], lo.y[part], b0, a0)
        lo.a[part], lo.b[part] = rotate_z(beam.a[part], beam.b[part], b0, a0)
    lo.c[part] = beam.c[part]


def virgin_local_to_global(azimuth_sc, vlb, center, part):
    a0, b0 = azimuth_sc
    if a0 != 0:
        vlb.a[part], vlb.b[part] = rotate_z(vlb.a[part], vlb.b[part], b0, -a0)
        vlb.x[part], vlb.y[part] = rotate_z(vlb.x[part], vlb.y[part], b0, -a0)
    if center is not None:
        vlb.x[part] += center[0]
        vlb.y[part] += center[1]
        vlb.z[part] += center[2]


def copy_beam(to, fr, idx, includeState=False, includeJspEsp=True):
    """sources/beams.py:409-445 (array fields only)."""
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E'):
        getattr(to, f)[idx] = getattr(fr, f)[idx]
    if includeState:
        to.state[idx] = fr.state[idx]
    if includeJspEsp:
        for f in ('Jss', 'Jpp', 'Jsp'):
            getattr(to, f)[idx] = getattr(fr, f)[idx]
        if hasattr(fr, 'Es') and hasattr(to, 'Es'):
            to.Es[idx] = fr.Es[idx]
            to.Ep[idx] = fr.Ep[idx]


def rotate_coherency_matrix(b, idx, roll):
    """sources/beams.py:448-479."""
    c = np.cos(roll)
    s = np.sin(roll)
    c2 = c**2
    s2 = s**2
    cs = c * s
    JssN = b.Jss[idx]*c2 + b.Jpp[idx]*s2 + 2*b.Jsp[idx].real*cs
    JppN = b.Jss[idx]*s2 + b.Jpp[idx]*c2 - 2*b.Jsp[idx].real*cs
    JspN = (b.Jpp[idx]-b.Jss[idx])*cs + b.Jsp[idx].real*(c2-s2) + \
        b.Jsp[idx].imag*1j
    return JssN, JppN, JspN


# --------------------------------------------------------------------------
# surfaces
# --------------------------------------------------------------------------
def local_z(surf, x, y):
    if surf['kind'] == 'user':                    # an OE subclass's own local_z (numpy callables)
        return surf['z'](x, y)
    if surf['kind'] == 'flat':                    # oes/base.py:675-679
        return np.zeros_like(y)
    if surf['kind'] == 'toroid':                  # oes/__init__.py:398-401
        R, r = surf['R'], surf['r']
        rx = 1 - (np.asarray(x)/r)**2
        rx[rx < 0] = 0.
        return y**2/2.0/R + r*(1 - rx**0.5)
    if surf['kind'] == 'bentflat':                # oes/__init__.py:289-293
        return (y**2 - surf['y0']**2) / 2.0 / surf['R']
    if surf['kind'] == 'blazed':                  # oes/gratings.py:475-480
        y0, y1, yC, yL = blazed_pre(surf, y)
        return np.where(yL > yC, -(y1-y) * surf['tanBlaze'],
                        -yL * surf['tanAntiblaze'])
    if surf['kind'] == 'sagittal':                # DCMwithSagittalFocusing.local_z2, :655-656
        return surf['Rs'] - np.sqrt(surf['Rs']**2 - x**2)
    if surf['kind'] == 'vfm':                     # VFM.local_z, oes/__init__.py:458-467
        z = surf['r'] - (surf['r']**2 - x**2)**0.5
        if surf.get('limOptX') is not None:
            zMax = surf['r'] - (surf['r']**2 - surf['limOptX'][1]**2)**0.5
            z[z > zMax] = zMax
        z += (y**2 - surf['y0']**2) / 2.0 / surf['R']
        return z
    if surf['kind'] == 'dualvfm':                 # DualVFM.local_z, oes/__init__.py:532-550
        z = np.zeros_like(x)
        ind = x < 0
        with np.errstate(invalid='ignore'):
            tmp2 = surf['r2']**2 - (x[ind] - surf['xCylinder2'])**2
            z[ind] = surf['r2'] - surf['hCylinder2'] - tmp2**0.5
            tmp1 = surf['r1']**2 - (x[~ind] - surf['xCylinder1'])**2
            z[~ind] = surf['r1'] - surf['hCylinder1'] - tmp1**0.5
        z[np.isnan(z)] = 0.
        z[z > 0] = 0.
        z += (y**2 - surf['y0']**2) / 2.0 / surf['R']
        return z
    if surf['kind'] == 'diced':                   # DicedOE.local_z, bragg.py:52-65
        fx, fy, cz, cn = _diced_facet(surf, x, y)
        dz = fy**2 / 2.0 / surf['Rm'] if surf['planes'] == 'johansson' else np.zeros_like(fx)
        return cz + (dz - cn[-3]*fx - cn[-2]*fy) / cn[-1]
    if surf['kind'] == 'laue_2d':                 # BentLaue2D.local_z, laue.py:364-365
        return 0.5*x**2 / surf['Rs'] + 0.5*y**2 / surf['Rm']
    if surf['kind'] == 'laue_sphere':             # BentLaueSphere.local_z, laue.py:487-491
        if surf['crossSection'].startswith('circ'):
            return surf['Rm'] - np.sqrt(surf['Rm']**2 - x**2 - y**2)
        return (x**2+y**2) / 2.0 / surf['Rm']
    if surf['kind'] == 'bent_cylinder':           # Johann/JohanssonCylinder, bragg.py:138-144
        if surf['crossSection'].startswith('circ'):
            sq = surf['Rm']**2 - y**2
            return surf['Rm'] - np.sqrt(sq)
        return y**2 / 2.0 / surf['Rm']
    if surf['kind'] == 'bent_toroid':             # JohannToroid.local_z, bragg.py:236-241
        Rm, Rs = surf['Rm'], surf['Rs']
        z = Rm - Rs - (Rm**2 - y**2)**0.5
        cosangle, sinangle = (z**2 - x**2)**0.5 / abs(z), -x/abs(z)
        bla, z = rotate_y(0, z, cosangle, sinangle)
        return z + Rs
    if surf['kind'] == 'cone':                    # ConicalMirror, oes/__init__.py:623-627
        t2t, L0, redfocus = surf['t2t'], surf['L0'], surf['redfocus']
        sqroot = np.sqrt(0.25*t2t**2*(y - L0)**2 - redfocus*t2t*x**2)
        z = -0.5*t2t*(y-L0)-np.sign(t2t)*sqroot
        return z
    if surf['kind'] == 'paraboloid':              # oes/refractive.py:394-399, 613-614
        if surf['cylinder']:
            x = 0
        z = (x**2 + y**2) / (4 * surf['focus'])
        if surf['zmax'] is not None:
            z[z > surf['zmax']] = surf['zmax']
        return z
    raise ValueError(surf['kind'])


def blazed_pre(surf, y):
    """Facet bookkeeping of a constant-density blazed grating
    (gratings.py:461-473): groove start/end, apex position, offset in groove."""
    rho_1 = surf['rho_1']
    y0 = (y // rho_1) * rho_1
    y1 = y0 + rho_1
    yL = y % rho_1
    yC = (y1-y0) / (1 + surf['tanAntiblaze']/surf['tanBlaze'])
    return y0, y1, yC, yL


def make_blazed(blaze, rho, antiblaze=np.pi*0.4999):
    """Surface dictionary as BlazedGrating.reset derives it (gratings.py:416-440)."""
    return dict(kind='blazed', blaze=blaze, antiblaze=antiblaze, rho0=rho,
                rho_1=1. / rho, sinBlaze=np.sin(blaze), cosBlaze=np.cos(blaze),
                tanBlaze=np.tan(blaze), sinAntiblaze=np.sin(antiblaze),
                cosAntiblaze=np.cos(antiblaze), tanAntiblaze=np.tan(antiblaze))


def make_cone(L0, theta):
    """ConicalMirror's derived constants (oes/__init__.py:610-616)."""
    tt = np.tan(theta)
    t2t = np.tan(2*theta)
    return dict(kind='cone', L0=L0, theta=theta, tt=tt, t2t=t2t,
                redfocus=np.cos(theta)**2 / (1./tt-1./t2t))


def make_ellipse_param(p, q, abs_pitch, isCylindrical=False, isClosed=False):
    """EllipticalMirrorParam._reset_pq (parametric.py:143-157) for a mirror whose
    p arm lies along the global y axis: *abs_pitch* = |asin(axis . normal)|."""
    gamma = np.arctan2((p - q) * np.sin(abs_pitch), (p + q) * np.cos(abs_pitch))
    return dict(kind='ellipse_param', p=p, q=q, cosGamma=np.cos(gamma),
                sinGamma=np.sin(gamma), y0=(q - p)/2. * np.cos(abs_pitch),
                z0=(q + p)/2. * np.sin(abs_pitch), ellipseA=(q + p)/2.,
                ellipseB=np.sqrt(q * p) * np.sin(abs_pitch),
                isCylindrical=bool(isCylindrical), isClosed=bool(isClosed))


PARAM_KINDS = ('ellipse_param', 'parabola_param', 'hyperbola_param')
# surfaces of revolution about the local y axis (capillaries), parametric.py:717-988
REVOLUTION_KINDS = ('parab_capillary', 'ellipse_capillary', 'hyperbola_capillary')


def make_parabola_param(p, q, abs_pitch, isCylindrical=False, isClosed=False):
    """ParabolicalMirrorParam._reset_pq (parametric.py:411-425): exactly one of
    p (collimating) / q (focusing) is given."""
    if p is None:
        y0, z0 = q * np.cos(abs_pitch), q * np.sin(abs_pitch)
        parabParam = -q * np.sin(abs_pitch)**2
        gamma = abs_pitch
    else:
        y0, z0 = -p * np.cos(abs_pitch), p * np.sin(abs_pitch)
        parabParam = p * np.sin(abs_pitch)**2
        gamma = -abs_pitch
    return dict(kind='parabola_param', cosGamma=np.cos(gamma), sinGamma=np.sin(gamma),
                y0=y0, z0=z0, parabParam=parabParam,
                isCylindrical=bool(isCylindrical), isClosed=bool(isClosed))


def make_hyperbola_param(p, q, abs_pitch, isCylindrical=False, isClosed=False):
    """HyperbolicMirrorParam._reset_pq (parametric.py:611-622)."""
    gamma = np.arctan2((p + q) * np.sin(abs_pitch), (p - q) * np.cos(abs_pitch))
    return dict(kind='hyperbola_param', cosGamma=np.cos(gamma),
                sinGamma=np.sin(gamma), y0=-(p + q)/2. * np.cos(abs_pitch),
                z0=(p - q)/2. * np.sin(abs_pitch), hyperbolaA=abs(p - q)/2.,
                hyperbolaB=np.sqrt(p*q) * np.sin(abs_pitch),
                isCylindrical=bool(isCylindrical), isClosed=bool(isClosed))


def is_param(surf):
    return surf['kind'] in PARAM_KINDS + REVOLUTION_KINDS


def xyz_to_param(surf, x, y, z):                  # parametric.py:213-216
    if surf['kind'] in REVOLUTION_KINDS:          # :726-727
        return y, np.arctan2(x, z), np.sqrt(x**2 + z**2)
    yNew, zNew = rotate_x(y - surf['y0'], z - surf['z0'], surf['cosGamma'],
                          surf['sinGamma'])
    return yNew, np.arctan2(x, zNew), np.sqrt(x**2 + zNew**2)


def param_to_xyz(surf, s, phi, r):                # parametric.py:218-223
    if surf['kind'] in REVOLUTION_KINDS:          # :729-730
        return r * np.sin(phi), s, r * np.cos(phi)
    x = r * np.sin(phi)
    y = s
    z = r * np.cos(phi)
    yNew, zNew = rotate_x(y, z, surf['cosGamma'], -surf['sinGamma'])
    return x, yNew + surf['y0'], zNew + surf['z0']


def local_r(surf, s, phi):        # parametric.py:225-231, 450-458, 690-696
    if surf['kind'] == 'parab_capillary':         # :780-781
        return 2*np.sqrt((surf['s0']-s)*surf['focus'])
    if surf['kind'] == 'ellipse_capillary':       # :877-879
        return surf['ellipseB'] * np.sqrt(abs(1 - (surf['ctd']+s)**2/surf['ellipseA']**2))
    if surf['kind'] == 'hyperbola_capillary':     # :974-977
        ss = surf['ctd'] + s
        return surf['hyperbolaB'] * np.sqrt(abs(ss**2/surf['hyperbolaA']**2 - 1))
    if surf['kind'] == 'parabola_param':
        r2 = surf['parabParam']*s + surf['parabParam']**2
        r2[r2 < 0] = 0
        r = 2 * r2**0.5
    elif surf['kind'] == 'hyperbola_param':
        r = surf['hyperbolaB'] * np.sqrt(abs(s**2/surf['hyperbolaA']**2 - 1))
    else:
        r = surf['ellipseB'] * np.sqrt(abs(1 - s**2 / surf['ellipseA']**2))
    if surf['isCylindrical']:
        r /= abs(np.cos(phi))
    if surf['isClosed']:
        return r
    if surf['kind'] == 'hyperbola_param':           # the branch facing +z
        return np.where(abs(phi) < np.pi/2, r, np.ones_like(phi)*1e20)
    return np.where(abs(phi) > np.pi/2, r, np.ones_like(phi)*1e20)


def _diced_facet(surf, x, y):
    """DicedOE.local_z(skipReturnZ) + facet_center_z / facet_center_n, bragg.py:52-60,
    353-362: facet coordinates, height and normal(s) of the base surface at the centre."""
    cx = (x / surf['xStep']).round() * surf['xStep']
    cy = (y / surf['yStep']).round() * surf['yStep']
    if surf['base'] == 'flat':
        return x - cx, y - cy, np.zeros_like(cy), [0, 0, 1]
    base = dict(surf, kind='bent_toroid')
    return x - cx, y - cy, local_z(base, cx, cy), local_n(base, cx, cy)


def _n_bent_cylinder(surf, x, y, R, alpha):      # JohannCylinder.local_n_cylinder
    a = np.zeros_like(x)
    b = -y / R
    if surf['crossSection'].startswith('circ'):
        c = (R**2 - y**2)**0.5 / R
    else:
        norm = (b**2 + 1)**0.5
        b /= norm
        c = 1. / norm
    if alpha:
        bAlpha, cAlpha = rotate_x(b, c, np.cos(alpha), -np.sin(alpha))
        return [a, bAlpha, cAlpha, a, b, c]
    return [a, b, c]


def _n_bent_toroid(surf, x, y, Rm, Rs, alpha):     # JohannToroid.local_n_toroid
    a = np.zeros_like(x)
    b = -y / Rm
    c = (Rm**2 - y**2)**0.5 / Rm
    if alpha:
        aAlpha = np.zeros_like(x)
        bAlpha, cAlpha = rotate_x(b, c, np.cos(alpha), -np.sin(alpha))
    r = Rs - (Rm - (Rm**2 - y**2)**0.5)
    cosangle, sinangle = (r**2 - x**2)**0.5 / r, -x/r
    a, c = rotate_y(a, c, cosangle, sinangle)
    if alpha:
        aAlpha, cAlpha = rotate_y(aAlpha, cAlpha, cosangle, sinangle)
        return [aAlpha, bAlpha, cAlpha, a, b, c]
    return [a, b, c]


def local_n(surf, x, y):
    """3-list, or 6-list [n_H(3), n_surface(3)] for an asymmetric cut."""
    if surf['kind'] == 'user':
        return list(surf['n'](x, y))
    if surf['kind'] == 'flat' and surf.get('laue'):   # LauePlate.local_n, oes/laue.py:14-20
        a, b, c = 0, 0, 1
        if surf.get('alpha'):
            bB, cB = rotate_x(b, c, -np.sin(surf['alpha']), -np.cos(surf['alpha']))
        else:
            bB, cB = c, -b
        return [a, bB, cB, a, b, c]
    if surf['kind'] == 'flat':                    # oes/base.py:719-742
        a = 0.
        b = 0.
        c = 1.
        alpha = surf.get('alpha')
        if alpha:
            bAlpha, cAlpha = rotate_x(b, c, np.cos(alpha), -np.sin(alpha))
            res = [a, bAlpha, cAlpha, a, b, c]
        else:
            res = [a, b, c]
        if surf.get('flip_n_y') and alpha:        # DCM.local_n2, dcm.py:234-238
            res[1] *= -1
        return res
    if surf['kind'] == 'toroid':                  # oes/__init__.py:403-411
        R, r = surf['R'], surf['r']
        rx = 1 - (np.asarray(x)/r)**2
        with np.errstate(divide='ignore', invalid='ignore'):
            ax = np.where(rx < 0, 0, rx**(-0.5))
        a = -x / r * ax
        b = -y / R
        c = 1.
        norm = (a**2 + b**2 + 1)**0.5
        return [a/norm, b/norm, c/norm]
    if surf['kind'] == 'bentflat':                # oes/__init__.py:295-303
        a = 0.
        b = -y / surf['R']
        c = 1.
        norm = (b**2 + 1)**0.5
        return [a/norm, b/norm, c/norm]
    if surf['kind'] == 'blazed':                  # gratings.py:482-490
        y0, y1, yC, yL = blazed_pre(surf, y)
        return [np.zeros_like(x),
                np.where(yL > yC, -surf['sinBlaze'], surf['sinAntiblaze']),
                np.where(yL > yC, surf['cosBlaze'], surf['cosAntiblaze'])]
    if surf['kind'] == 'vfm':                     # oes/__init__.py:469-477
        a = -x * (surf['r']**2 - x**2)**(-0.5)
        if surf.get('limOptX') is not None:
            a[(x < surf['limOptX'][0]) | (x > surf['limOptX'][1])] = 0.0
        b = -y / surf['R']
        norm = (a**2 + b**2 + 1)**0.5
        return [a/norm, b/norm, 1./norm]
    if surf['kind'] == 'dualvfm':                 # oes/__init__.py:552-571
        a = np.zeros_like(x)
        ind = x < 0
        with np.errstate(invalid='ignore'):
            tmp2 = surf['r2']**2 - (x[ind] - surf['xCylinder2'])**2
            a[ind] = -(x[ind] - surf['xCylinder2']) * tmp2**(-0.5)
            tmp1 = surf['r1']**2 - (x[~ind] - surf['xCylinder1'])**2
            a[~ind] = -(x[~ind] - surf['xCylinder1']) * tmp1**(-0.5)
        z = local_z(surf, x, y)
        a[np.isnan(a)] = 0.
        a[z > 0] = 0.
        b = -y / surf['R']
        norm = (a**2 + b**2 + 1)**0.5
        return [a/norm, b/norm, 1./norm]
    if surf['kind'] == 'diced':                   # DicedOE.local_n, bragg.py:67-90
        fx, fy, cz, cn = _diced_facet(surf, x, y)
        cn = [np.array(v, dtype=float) * np.ones_like(fx) for v in cn]
        if surf['planes'] == 'johansson':         # facet_delta_n, bragg.py:367-372
            b = -fy / surf['Rm']
            norm = (b**2 + 1)**0.5
            cn[-1] = cn[-1] + 1./norm
            cn[-2] = cn[-2] + b/norm
            norm = (cn[-1]**2 + cn[-2]**2 + cn[-3]**2)**0.5
            cn[-1] = cn[-1] / norm
            cn[-2] = cn[-2] / norm
            cn[-3] = cn[-3] / norm
        alpha = surf.get('alpha')
        if alpha:
            bAlpha, cAlpha = rotate_x(cn[1], cn[2], np.cos(alpha), -np.sin(alpha))
            return [cn[0], bAlpha, cAlpha, cn[-3], cn[-2], cn[-1]]
        return cn
    if surf['kind'] == 'laue_2d':                 # BentLaue2D.local_n, laue.py:424-452
        a = -x / surf['Rs']
        b = -y / surf['Rm']
        c = 1.
        norm = np.sqrt(a**2 + b**2 + 1)
        a /= norm
        b /= norm
        c /= norm
        sinpitch = -b
        cospitch = np.sqrt(1 - b**2)
        sinroll = -a
        cosroll = np.sqrt(1 - a**2)
        aB = np.zeros_like(a)
        bB = np.ones_like(a)
        cB = np.zeros_like(a)
        if surf.get('alpha'):
            bB, cB = rotate_x(bB, cB, np.cos(surf['alpha']), -np.sin(surf['alpha']))
        aB, cB = rotate_y(aB, cB, cosroll, -sinroll)
        bB, cB = rotate_x(bB, cB, cospitch, sinpitch)
        normB = (bB**2 + cB**2 + aB**2)**0.5
        return [aB/normB, bB/normB, cB/normB, a/norm, b/norm, c/norm]
    if surf['kind'] == 'laue_sphere':             # laue.py:493-507
        R = surf['Rm']
        if surf['crossSection'].startswith('circ'):
            a = -x * (R**2 - x**2 - y**2)**(-0.5)
            b = -y * (R**2 - x**2 - y**2)**(-0.5)
        else:
            a = -x / R
            b = -y / R
        c = 1.
        norm = (a**2 + b**2 + 1)**0.5
        aB = 0.
        bB = c
        cB = -b
        normB = (b**2 + c**2)**0.5
        return [aB/normB, bB/normB, cB/normB, a/norm, b/norm, c/norm]
    if surf['kind'] == 'bent_cylinder' and surf['planes'] in ('laue', 'laue_ground'):
        alpha = surf.get('alpha')                 # laue.py:153-173, 457-470
        nS = _n_bent_cylinder(surf, x, y, surf['Rm'], None)
        a, b, c = nS
        if surf['planes'] == 'laue_ground':
            b = -y
            c = (surf['Rm']**2 - y**2)**0.5 + surf['Rm']
        if alpha:
            bB, cB = rotate_x(b, c, -np.sin(alpha), -np.cos(alpha))
        else:
            bB, cB = c, -b
        if surf['planes'] == 'laue_ground':
            norm = np.sqrt(bB**2 + cB**2)
            return [a/norm, bB/norm, cB/norm, nS[-3], nS[-2], nS[-1]]
        return [a, bB, cB, nS[0], nS[1], nS[2]]
    if surf['kind'] == 'bent_cylinder':           # bragg.py:146-197
        nSurf = _n_bent_cylinder(surf, x, y, surf['Rm'], surf.get('alpha'))
        if surf['planes'] == 'johann':
            return nSurf
        nSurf = _n_bent_cylinder(surf, x, y, surf['Rm'], None)
        a = np.zeros_like(x)
        b = -y
        c = (surf['Rm']**2 - y**2)**0.5 + surf['Rm']
        if surf.get('alpha'):
            b, c = rotate_x(b, c, np.cos(surf['alpha']), -np.sin(surf['alpha']))
        norm = np.sqrt(b**2 + c**2)
        return [a/norm, b/norm, c/norm, nSurf[-3], nSurf[-2], nSurf[-1]]
    if surf['kind'] == 'bent_toroid':             # bragg.py:243-343
        Rm, Rs = surf['Rm'], surf['Rs']
        if surf['planes'] == 'johann':
            return _n_bent_toroid(surf, x, y, Rm, Rs, surf.get('alpha'))
        nSurf = _n_bent_toroid(surf, x, y, Rm, Rs, None)
        if surf['planes'] == 'general':
            nSurfBr = _n_bent_toroid(surf, x, y, surf['RmBragg'], surf['RsBragg'], None)
            return [nSurfBr[0], nSurfBr[1], nSurfBr[2], nSurf[-3], nSurf[-2], nSurf[-1]]
        a = np.zeros_like(x)
        b = -y
        c = (Rm**2 - y**2)**0.5 + Rm
        norm = np.sqrt(b**2 + c**2)
        b, c = b/norm, c/norm
        alpha = surf.get('alpha')
        if alpha:
            b, c = rotate_x(b, c, np.cos(alpha), -np.sin(alpha))
        r = Rs - (Rm - (Rm**2 - y**2)**0.5)
        cosangle, sinangle = (r**2 - x**2)**0.5 / r, -x/r
        a, c = rotate_y(a, c, cosangle, sinangle)
        if alpha:
            a, c = rotate_y(a, c, cosangle, sinangle)
        return [a, b, c, nSurf[-3], nSurf[-2], nSurf[-1]]
    if surf['kind'] == 'sagittal':                # oes/__init__.py:658-662
        a = -x / surf['Rs']  # -dz/dx
        c = (surf['Rs']**2-x**2)**0.5 / surf['Rs']
        b = np.zeros_like(y)  # -dz/dy
        return [a, b, c]
    if surf['kind'] == 'cone':                    # oes/__init__.py:629-636
        t2t, L0, redfocus = surf['t2t'], surf['L0'], surf['redfocus']
        sqroot = np.sign(t2t)*np.sqrt(0.25*t2t**2*(y - L0)**2 - redfocus*x*x*t2t)
        a = -x*redfocus*t2t/sqroot  # -dz/dx
        b = .5*t2t + 0.25*t2t**2*(y-L0)/sqroot  # -dz/dy
        c = 1.
        norm = (a**2 + b**2 + 1.)**0.5
        return [a/norm, b/norm, c/norm]
    if surf['kind'] == 'paraboloid':              # oes/refractive.py:405-419, 616-617
        if surf['cylinder']:
            x = 0
        a = -x / (2*surf['focus'])  # -dz/dx
        b = -y / (2*surf['focus'])  # -dz/dy
        if surf['zmax'] is not None:
            z = (x**2 + y**2) / (4*surf['focus'])
            if isinstance(a, np.ndarray):
                a[z > surf['zmax']] = 0
            if isinstance(b, np.ndarray):
                b[z > surf['zmax']] = 0
        c = np.ones_like(x)
        norm = (a**2 + b**2 + 1)**0.5
        return [a/norm, b/norm, c/norm]
    if surf['kind'] == 'parab_capillary':         # parametric.py:783-788
        s, phi = x, y
        a = -np.sin(phi)
        b = -np.sqrt(surf['focus']/(surf['s0']-s))
        c = -np.cos(phi)
        norm = np.sqrt(a**2 + b**2 + c**2)
        return [a/norm, b/norm, c/norm]
    if surf['kind'] == 'ellipse_capillary':       # :881-889
        s, phi = x, y
        A2s2 = np.array(surf['ellipseA']**2 - (surf['ctd']+s)**2)
        A2s2[A2s2 <= 0] = 1e22
        nr = -surf['ellipseB'] / surf['ellipseA'] * (surf['ctd']+s) / np.sqrt(A2s2)
        norm = np.sqrt(nr**2 + 1.)
        return [-np.sin(phi) / norm, nr / norm, -np.cos(phi) / norm]
    if surf['kind'] == 'hyperbola_capillary':     # :979-988
        s, phi = x, y
        ss = surf['ctd'] + s
        A2s2 = np.array(ss**2 - surf['hyperbolaA']**2)
        A2s2[A2s2 <= 0] = 1e22
        nr = -surf['hyperbolaB'] / surf['hyperbolaA'] * ss / np.sqrt(A2s2)
        norm = np.sqrt(nr**2 + 1)
        return [np.sin(phi) / norm, nr / norm, np.cos(phi) / norm]
    if surf['kind'] in PARAM_KINDS:   # parametric.py:233-247, 460-472, 698-713
        s, phi = x, y
        sign = -1.
        if surf['kind'] == 'parabola_param':
            nr = surf['parabParam'] / \
                (surf['parabParam']*s + surf['parabParam']**2)**0.5
        elif surf['kind'] == 'hyperbola_param':
            A2s2 = np.array(s**2 - surf['hyperbolaA']**2)
            A2s2[A2s2 <= 0] = 1e22
            nr = -surf['hyperbolaB'] / surf['hyperbolaA'] * s / np.sqrt(A2s2)
            sign = 1.
        else:
            A2s2 = np.array(surf['ellipseA']**2 - s**2)
            A2s2[A2s2 <= 0] = 1e22
            nr = -surf['ellipseB'] / surf['ellipseA'] * s / np.sqrt(A2s2)
        norm = np.sqrt(nr**2 + 1)
        b = nr / norm
        if surf['isCylindrical']:
            a = np.zeros_like(phi)
            c = 1. / norm
        elif sign < 0:
            a = -np.sin(phi) / norm
            c = -np.cos(phi) / norm
        else:
            a = np.sin(phi) / norm
            c = np.cos(phi) / norm
        bNew, cNew = rotate_x(b, c, surf['cosGamma'], -surf['sinGamma'])
        return [a, bNew, cNew]
    raise ValueError(surf['kind'])


# --------------------------------------------------------------------------
# bracketing + root solve
# --------------------------------------------------------------------------
def _set_t(xyz, abc, surfPhys=None, defSize=maxHalfSizeOfOE):
    if surfPhys is None:
        limMin = -defSize
        limMax = defSize
    else:
        limMin = surfPhys[0] if surfPhys[0] > -np.inf else -defSize
        limMax = surfPhys[1] if surfPhys[1] < np.inf else defSize
    if abc[0] > 0:                                # first ray decides, base.py:1239
        tMin = (limMin-xyz)/abc - dt
        tMax = (limMax-xyz)/abc + dt
    else:
        tMin = (limMax-xyz)/abc - dt
        tMax = (limMin-xyz)/abc + dt
    return tMin, tMax


def bracket(oe, x, y, z, a, b, c, is2ndXtal, mainPart, info=None, isMulti=False,
            needElevationMap=False, surf=None, invertNormal=1):
    """-> tMin, tMax, elevation. *isMulti* (a further bounce of multiple_reflect,
    base.py:1279-1289): the search starts where the ray is farthest from the surface it has
    just left -- the root of ray . normal between 0 and tMax (find_intersection with
    derivOrder=1) -- and *elevation* (needElevationMap) = find_dz there."""
    sfx = '2' if is2ndXtal else ''
    surfPhysX = oe['surfPhysX' + sfx]
    surfPhysY = oe['surfPhysY' + sfx]
    try:
        maxa = np.max(abs(a[mainPart]))
        maxb = np.max(abs(b[mainPart]))
        maxc = np.max(abs(c[mainPart]))
    except ValueError:
        maxa, maxb, maxc = 0, 1, 0
    maxMax = max(maxa, maxb, maxc)
    if maxMax == maxa:
        axis = 0
        tMin, tMax = _set_t(x, a, surfPhysX)
    elif maxMax == maxb:
        axis = 1
        tMin, tMax = _set_t(y, b, surfPhysY)
    else:
        axis = 2
        tMin, tMax = _set_t(z, c, defSize=maxDepthOfOE)
    tMin[tMin < -1e6*zEps] = -1e6*zEps            # base.py:1275
    if info is not None:
        info['axis'] = axis
    elevation = None
    if isMulti:                                   # base.py:1279-1289
        tMin[:] = 0
        tMaxTmp = np.copy(tMax)
        tangency = {} if info is not None else None
        tMax = find_intersection(surf, tMin, tMax, x, y, z, a, b, c, invertNormal,
                                 tangency, derivOrder=1)[0]
        if needElevationMap:
            elevation = find_dz(surf, tMax, x, y, z, a, b, c, invertNormal)
        tMin = tMax + ds
        tMax = tMaxTmp
        if info is not None:
            info['tangency'] = tangency
    return tMin, tMax, elevation


def find_dz(surf, t, x0, y0, z0, a, b, c, invertNormal, derivOrder=0):
    x = x0 + a*t
    y = y0 + b*t
    z = z0 + c*t
    if derivOrder:                                # base.py:819-821, 842-845: ray . normal
        if is_param(surf):
            x, y, z = xyz_to_param(surf, x, y, z)
        n = local_n(surf, x, y)
        dz = (a*n[-3] + b*n[-2] + c*n[-1]) * invertNormal
        return dz, x, y, z
    if is_param(surf):                            # base.py:822-841, diffSign = -1
        x, y, z = xyz_to_param(surf, x, y, z)     # s, phi, r
        s = local_r(surf, x, y)
        diffSign = -1
    else:
        s = local_z(surf, x, y)
        diffSign = 1
        if surf.get('figure_z') is not None:      # base.py:826-830: surf += z_distorted
            s = s + surf['figure_z'](x, y)
    ind = np.isnan(s)
    if ind.sum() > 0:
        s[ind] = 0
    dz = (z - s) * diffSign * invertNormal
    return dz, x, y, z


def blazed_find_intersection(surf, x, y, z, a, b, c):
    """First illuminated facet of the saw-tooth, closed form
    (gratings.py:492-522, constant line density)."""
    b_c = b / c
    n = np.floor((y - b_c*z) / surf['rho_1'])
    y0 = surf['rho_1'] * n
    y1 = y0 + surf['rho_1']
    if surf['antiblaze'] == np.pi/2:
        zabl = (y0-y) / b_c + z
    else:
        zabl = -surf['tanAntiblaze'] * (y - b_c*z - y0) /\
            (1 + surf['tanAntiblaze']*b_c)
    if surf['blaze'] == np.pi/2:
        zbl = (y1-y) / b_c + z
    else:
        zbl = surf['tanBlaze'] * (y - b_c*z - y1) / (1 - surf['tanBlaze']*b_c)
    if ((zabl > 0) & (zbl > 0)).any():
        raise ValueError('blazed grating: ray above both facets')
    zabl[zabl > 0] = zbl[zabl > 0] - 1
    zbl[zbl > 0] = zabl[zbl > 0] - 1
    z2 = zbl
    y2 = b_c * (z2 - z) + y
    t2 = (y2 - y) / b
    x2 = x + t2 * a
    return t2, x2, y2, z2


def find_intersection(surf, t1, t2, x, y, z, a, b, c, invertNormal, info=None,
                      derivOrder=0):
    if surf['kind'] == 'blazed':
        if info is not None:
            info.update(brent=False, numit=0, tMinGlobal=np.nan,
                        tMaxGlobal=np.nan)
        return blazed_find_intersection(surf, x, y, z, a, b, c) + (None,)
    dz1, x1, y1, z1 = find_dz(surf, t1, x, y, z, a, b, c, invertNormal, derivOrder)
    dz2, x2, y2, z2 = find_dz(surf, t2, x, y, z, a, b, c, invertNormal, derivOrder)
    tMin = t1.min()
    tMax = t2.max()
    ind1 = dz1 <= 0
    ind2 = dz2 >= 0
    dz2[ind1 | ind2] = 0
    t2[ind1] = t1[ind1]
    x2[ind1] = x1[ind1]
    y2[ind1] = y1[ind1]
    z2[ind1] = z1[ind1]
    ind = ~(ind1 | ind2)
    use_brent = bool(abs(dz2).max() > abs(dz1).max()*20)
    solver = brent if use_brent else secant
    t2, x2, y2, z2, numit = solver(
        surf, t1, t2, x, y, z, a, b, c, invertNormal, dz1, dz2, tMin, tMax,
        x2, y2, z2, ind, derivOrder)
    if info is not None:
        info.update(brent=use_brent, numit=numit, tMinGlobal=tMin,
                    tMaxGlobal=tMax)
    return t2, x2, y2, z2, ind1


def secant(surf, t1, t2, x, y, z, a, b, c, invertNormal, dz1, dz2, tMin, tMax,
           x2, y2, z2, ind, derivOrder=0):
    """base.py:933-959."""
    numit = 2
    while (ind.sum() > 0) and (numit < maxIteration):
        t = t1[ind]
        dz = dz1[ind]
        t1[ind] = t2[ind]
        dz1[ind] = dz2[ind]
        with np.errstate(divide='ignore', invalid='ignore'):
            t2[ind] = t - (t1[ind]-t) * dz / (dz1[ind]-dz)
        where = np.where(ind)[0]
        t2[where[t2[ind] < tMin]] = tMin
        t2[where[t2[ind] > tMax]] = tMax
        dz2[ind], x2[ind], y2[ind], z2[ind] = find_dz(
            surf, t2[ind], x[ind], y[ind], z[ind], a[ind], b[ind], c[ind],
            invertNormal, derivOrder)
        swap = np.sign(dz2[ind]) == np.sign(dz1[ind])
        t1[where[swap]] = t[swap]
        dz1[where[swap]] = dz[swap]
        ind = ind & (abs(dz2) > zEps)
        numit += 1
    return t2, x2, y2, z2, numit


def brent(surf, t1, t2, x, y, z, a, b, c, invertNormal, dz1, dz2, tMin, tMax,
          x2, y2, z2, ind, derivOrder=0):
    """base.py:961-1048."""
    where = np.where(ind)[0]
    swap = abs(dz1[ind]) < abs(dz2[ind])
    if swap.sum() > 0:
        w = where[swap]
        t1[w], t2[w] = t2[w], t1[w]
        dz1[w], dz2[w] = dz2[w], dz1[w]
    t3 = np.copy(t1)
    dz3 = np.copy(dz1)
    t4 = np.zeros_like(t1)
    mflag = np.ones_like(t1, dtype='bool')
    numit = 2
    ind = ind & (abs(dz2) > zEps)
    while (ind.sum() > 0) and (numit < maxIteration):
        xa, xb, xc, xd = t1[ind], t2[ind], t3[ind], t4[ind]
        fa, fb, fc = dz1[ind], dz2[ind], dz3[ind]
        mf = mflag[ind]
        xs = np.empty_like(xa)
        inq = (fa != fc) & (fb != fc)
        with np.errstate(divide='ignore', invalid='ignore'):
            if inq.sum() > 0:
                xai, xbi, xci = xa[inq], xb[inq], xc[inq]
                fai, fbi, fci = fa[inq], fb[inq], fc[inq]
                xs[inq] = \
                    xai * fbi * fci / (fai-fbi) / (fai-fci) + \
                    fai * xbi * fci / (fbi-fai) / (fbi-fci) + \
                    fai * fbi * xci / (fci-fai) / (fci-fbi)
            inx = ~inq
            if inx.sum() > 0:
                xai, xbi = xa[inx], xb[inx]
                fai, fbi = fa[inx], fb[inx]
                xs[inx] = xbi - fbi * (xbi-xai) / (fbi-fai)
        cond1 = ((xs < (3*xa + xb) / 4.) & (xs < xb) |
                 (xs > (3*xa + xb) / 4.) & (xs > xb))
        cond2 = mf & (abs(xs - xb) >= (abs(xb - xc) / 2.))
        cond3 = (~mf) & (abs(xs - xb) >= (abs(xc - xd) / 2.))
        cond4 = mf & (abs(xb - xc) < zEps)
        cond5 = (~mf) & (abs(xc - xd) < zEps)
        conds = cond1 | cond2 | cond3 | cond4 | cond5
        xs[conds] = (xa[conds] + xb[conds]) / 2.
        mf = conds
        fs, x2[ind], y2[ind], z2[ind] = find_dz(
            surf, xs, x[ind], y[ind], z[ind], a[ind], b[ind], c[ind],
            invertNormal, derivOrder)
        xd[:] = xc[:]
        xc[:] = xb[:]
        fc[:] = fb[:]
        fafsNeg = ((fa < 0) & (fs > 0)) | ((fa > 0) & (fs < 0))
        xb[fafsNeg] = xs[fafsNeg]
        fb[fafsNeg] = fs[fafsNeg]
        fafsPos = ~fafsNeg
        xa[fafsPos] = xs[fafsPos]
        fa[fafsPos] = fs[fafsPos]
        swap = abs(fa) < abs(fb)
        xa[swap], xb[swap] = xb[swap], xa[swap]
        fa[swap], fb[swap] = fb[swap], fa[swap]
        t1[ind], t2[ind], t3[ind], t4[ind] = xa, xb, xc, xd
        dz1[ind], dz2[ind], dz3[ind] = fa, fb, fc
        mflag[ind] = mf
        ind = ind & (abs(dz2) > zEps)
        numit += 1
    return t2, x2, y2, z2, numit


def points_in_polygon(vertices, x, y):
    """matplotlib.path.Path(vertices).contains_points(zip(x, y)) with radius 0 -- what
    the reference's polygon-shaped elements call (base.py:1157-1158). matplotlib's
    crossing-number test (src/_path.h, point_in_path_impl): every edge v0 -> v1 of the
    implicitly closed polygon whose ends lie on different sides of the horizontal through
    the point toggles `inside` if
        ((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == (y1 >= ty).
    Pinned against matplotlib itself by the golden file g2_polygon (edge and vertex points
    included)."""
    v = np.asarray(vertices, dtype=float)
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    inside = np.zeros(x.shape, dtype=bool)
    if len(v) < 3:
        return inside
    # non-finite vertices split the outline into sub-polygons (matplotlib's PathNanRemover:
    # the vertex after a gap is a MOVETO), each closed on itself; a point is inside if it
    # is inside any of them (point_in_path_impl: inside_flag |= subpath_flag) -- GridAperture
    finite = np.isfinite(v).all(axis=1)
    k = 0
    while k < len(v):
        if not finite[k]:
            k += 1
            continue
        start = k
        while k < len(v) and finite[k]:
            k += 1
        sub = v[start:k]
        part = np.zeros(x.shape, dtype=bool)
        for j in range(len(sub)):
            x0, y0 = sub[j]
            x1, y1 = sub[(j + 1) % len(sub)]
            above0, above1 = y0 >= y, y1 >= y
            crosses = ((y1 - y) * (x0 - x1) >= (x1 - x) * (y0 - y1)) == above1
            part ^= (above0 != above1) & crosses
        inside |= part
    return inside & np.isfinite(x) & np.isfinite(y)


def rays_good(oe, x, y, is2ndXtal=False):
    """base.py:1094-1163 for shape 'rect' / 'round'."""
    sfx = '2' if is2ndXtal else ''
    surfPhysX, surfPhysY = oe['surfPhysX' + sfx], oe['surfPhysY' + sfx]
    surfOptX, surfOptY = oe.get('surfOptX' + sfx), oe.get('surfOptY' + sfx)
    lostNum = oe['lostNum']
    locState = np.ones(x.size, dtype=np.int32)
    shape = oe.get('shape', 'rect')
    if isinstance(shape, (list, tuple, np.ndarray)):     # base.py:1156-1160
        inside = points_in_polygon(shape, x, y)
        locState[:] = inside
        locState[(locState == 0) & (y < surfPhysY[0])] = lostNum
        locState[locState == 0] = 3
        return locState
    if shape.startswith('re'):
        if surfOptX is not None:
            locState[((surfPhysX[0] <= x) & (x < surfOptX[0])) |
                     ((surfOptX[1] <= x) & (x < surfPhysX[1]))] = 2
        if surfOptY is not None:
            locState[((surfPhysY[0] <= y) & (y < surfOptY[0])) |
                     ((surfOptY[1] <= y) & (y < surfPhysY[1]))] = 2
        ovE = str(oe.get('overEdge', 'yMax')).lower()
        outside = (x < surfPhysX[0]) | (x > surfPhysX[1]) |\
            (y < surfPhysY[0]) | (y > surfPhysY[1])
        over = np.zeros_like(outside)
        if 'xmin' in ovE:
            over |= x < surfPhysX[0]
        if 'xmax' in ovE:
            over |= x > surfPhysX[1]
        if 'ymin' in ovE:
            over |= y < surfPhysY[0]
        if 'ymax' in ovE:
            over |= y > surfPhysY[1]
        locState[outside] = lostNum
        locState[over] = 3
    elif shape.startswith('ro'):
        centerX = (surfPhysX[0]+surfPhysX[1]) * 0.5
        if np.isnan(centerX):
            centerX = 0
        radiusX = (surfPhysX[1]-surfPhysX[0]) * 0.5
        if surfPhysY is not None:
            centerY = (surfPhysY[0]+surfPhysY[1]) * 0.5
            radiusY = (surfPhysY[1]-surfPhysY[0]) * 0.5
        else:
            centerY = 0.
            radiusY = radiusX
        if np.isnan(centerY):
            centerY = 0
        if not np.isinf(radiusX):
            locState[((x-centerX)/radiusX)**2 +
                     ((y-centerY)/radiusY)**2 > 1] = lostNum
    else:
        raise ValueError('unsupported shape')
    return locState


# --------------------------------------------------------------------------
# grating deflection (reflect.py:451-469); crystal-as-grating (:568-612)
# --------------------------------------------------------------------------
def make_fzp(f, E, N, isCentralZoneBlack=True, thinnestZone=None):
    """NormalFZP.reset, gratings.py:97-116 -> the zone table."""
    lambdaE = CH / E * 1e-7
    if thinnestZone is not None:
        N = lambdaE * f / 4. / thinnestZone**2
    zones = np.arange(N+1)
    rn = np.sqrt(zones*f*lambdaE + 0.25*(zones*lambdaE)**2)
    return dict(zones=zones, rn=rn, black=bool(isCentralZoneBlack))


def _table(xp, fp, x):
    """scipy's interp1d(xp, fp, bounds_error=False, fill_value=0) for a float / int
    table: np.interp inside the table, 0 outside (scipy/interpolate/_interpolate.py,
    _call_linear_np + _check_bounds)."""
    x = np.asarray(x)
    y = np.interp(x, xp, fp)
    y[(x < xp[0]) | (x > xp[-1])] = 0
    return y


def fzp_rays_good_gn(oe, x, y):
    """NormalFZP.rays_good_gn, gratings.py:120-137."""
    fzp = oe['fzp']
    rn, zones = fzp['rn'], fzp['zones']
    locState = rays_good(oe, x, y)
    r = np.sqrt(x**2 + y**2)
    i = (_table(rn, zones, r)).astype(int)
    good = ((i % 2 == int(fzp['black'])) & (r < rn[-1]) & (locState == 1))
    locState[~good] = oe['lostNum']
    gz = np.zeros_like(x[good])
    rho = 1./(_table(zones, rn, i[good]+1) - _table(zones, rn, i[good]-1))
    gx = -x[good] / r[good] * rho
    gy = -y[good] / r[good] * rho
    return locState, (gx, gy, gz)


def general_fzp_rays_good_gn(oe, x, y, z):
    """GeneralFZPin0YZ.rays_good_gn, gratings.py:249-313. oe['gfzp'] = dict(f1, f2, lambdaE,
    N, phaseShift (as stored: divided by pi), vorticity, grazingAngle, minHalfLambda)."""
    q = oe['gfzp']
    locState = rays_good(oe, x, y)
    good = locState == 1

    def dist(f):
        if isinstance(f, str):
            return y[good] * np.cos(q['grazingAngle'])
        d = ((x[good]-f[0])**2 + (y[good]-f[1])**2 + (z[good]-f[2])**2)**0.5
        if len(f) > 3:
            d *= f[3]
        return d
    halfLambda = (dist(q['f1'])+dist(q['f2'])) / (q['lambdaE']/2)
    phi = np.arctan2(y[good]*np.sin(q['grazingAngle']), x[good]) / np.pi
    if q.get('minHalfLambda') is None:
        q['minHalfLambda'] = halfLambda.min()
    halfLambda -= q['minHalfLambda'] + q['phaseShift'] - phi*q['vorticity']
    N = q['N']
    zone = np.ones_like(x, dtype=np.int32) * (N+2)
    zone[good] = np.floor(halfLambda).astype(np.int32)
    goodN = (zone % 2 == 0) & (zone < N) & good
    badN = ((zone % 2 == 1) | (zone >= N)) & good
    locState[badN] = oe['lostNum']
    a = np.zeros(N)
    b = np.zeros(N)
    for i in range(1, N+1, 2):
        if (zone == i).sum() == 0:
            continue
        a[i] = max(abs(x[zone == i]))
        b[i] = max(abs(y[zone == i]))
    gz = np.zeros_like(x[goodN])
    r = np.sqrt(x[goodN]**2 + y[goodN]**2)
    diva = a[zone[goodN]+1] - a[zone[goodN]-1]
    diva[diva == 0] = 1e20
    divb = b[zone[goodN]+1] - b[zone[goodN]-1]
    divb[divb == 0] = 1e20
    xy = (x[goodN]**2/diva + y[goodN]**2/divb) / r**2
    gx = -x[goodN] * xy / r
    gy = -y[goodN] * xy / r
    return locState, (gx, gy, gz)


def local_g(oe, x, y):
    """Reciprocal groove vector [1/mm] of OE.local_g (base.py:688-717):
    polynomial line density ['x'|'y', rho0, p0, p1, ...] or a constant vector."""
    if callable(oe.get('local_g')):                 # an OE subclass's own local_g (numpy)
        return oe['local_g'](x, y)
    rhoList = oe.get('gratingDensity')
    if rhoList is not None:
        coord = x if rhoList[0] == 'x' else y
        poly = 0.
        for ic, coeff in enumerate(rhoList[2:]):
            poly += (ic+1) * coeff * coord**ic
        N = rhoList[1] * poly
        if rhoList[0] == 'x':
            return N, np.zeros_like(N), np.zeros_like(N)
        return np.zeros_like(N), N, np.zeros_like(N)
    return oe.get('gVector', (0, -100., 0))


def grating_deflection(a, b, c, E, g, oeNormal, beamInDotNormal, order, sig,
                       drawn=None):
    """reflect.py:451-469. A sequence *order*: one per hit ray from numpy's GLOBAL
    generator, as the reference draws it (:455-456); the draw goes to drawn[0]."""
    beamInDotG = a*g[0] + b*g[1] + c*g[2]
    G2 = g[0]**2 + g[1]**2 + g[2]**2
    locOrder = order if isinstance(order, (int, np.integer)) else \
        np.array(order)[np.random.randint(len(order), size=len(a))]
    if drawn is not None:
        drawn.append(locOrder)
    orderLambda = locOrder * CH / E * 1e-7
    u = beamInDotNormal**2 - 2*beamInDotG*orderLambda - G2*orderLambda**2
    gs = np.sign(beamInDotNormal) if sig is None else sig
    dn = beamInDotNormal + gs*np.sqrt(abs(u))
    a_out = a - oeNormal[-3]*dn + g[0]*orderLambda
    b_out = b - oeNormal[-2]*dn + g[1]*orderLambda
    c_out = c - oeNormal[-1]*dn + g[2]*orderLambda
    norm = (a_out**2 + b_out**2 + c_out**2)**0.5
    return a_out/norm, b_out/norm, c_out/norm


def asymmetric_reflection_grating(matSur, a, b, c, E, oeNormal,
                                  beamInDotSurfaceNormal, beamInDotNormal):
    normalDotSurfNormal = oeNormal[0]*oeNormal[-3] +\
        oeNormal[1]*oeNormal[-2] + oeNormal[2]*oeNormal[-1]
    bdn = beamInDotNormal.sum() / len(beamInDotNormal)
    sgbdn = 1 if bdn < 0 else -1
    wH = 0
    crystd = matSur['d']
    wHd = (1 + wH) / (crystd * 1e-7)
    gNormalCryst = np.asarray((
        (oeNormal[0]-normalDotSurfNormal*oeNormal[-3]) * wHd,
        (oeNormal[1]-normalDotSurfNormal*oeNormal[-2]) * wHd,
        (oeNormal[2]-normalDotSurfNormal*oeNormal[-1]) * wHd),
        order='F') * sgbdn
    sg = 1 if matSur['geom'].startswith('Laue') else -1
    return grating_deflection(a, b, c, E, gNormalCryst, oeNormal,
                              beamInDotSurfaceNormal, 1, sg)


# --------------------------------------------------------------------------
# the per-surface pipeline (reflect.py:551-1139)
# --------------------------------------------------------------------------
def reflect_local(oe, good, lb, vlb, pitch, roll, yaw, dx=None, dy=None,
                  dz=None, surf=None, fromVacuum=True, material=None,
                  is2ndXtal=False, noIntersectionSearch=False, info=None,
                  needElevationMap=False, isMulti=False):
    if surf is None:
        surf = oe['surface']
    rotSeq = oe.get('rotationSequence', 'RzRyRx')
    extra = [oe.get(k, 0) for k in ('extraPitch', 'extraRoll', 'extraYaw')]
    extraSeq = oe.get('extraRotationSequence', 'RzRyRx')
    extraAnglesSign = 1.
    if is2ndXtal:
        rotate_beam(lb, good, roll=-np.pi)
        extraAnglesSign = -1.
    rotate_beam(lb, good, rotSeq, pitch=-pitch, roll=-roll, yaw=-yaw)
    if extra[0] or extra[1] or extra[2]:
        rotate_beam(lb, good, extraSeq, pitch=-extraAnglesSign*extra[0],
                    roll=-extra[1], yaw=-extraAnglesSign*extra[2])
    if dx:
        lb.x[good] -= dx
    if dy:
        lb.y[good] -= dy
    if dz:
        lb.z[good] -= dz

    if 'invertNormal' in oe:
        invertNormal = oe['invertNormal']
    else:
        invertNormal = 1 if fromVacuum else -1

    mainPart = lb.state[good] == 1
    tMin = np.zeros_like(lb.x)
    tMax = np.zeros_like(lb.x)
    tMin[good], tMax[good], elev = bracket(
        oe, lb.x[good], lb.y[good], lb.z[good], lb.a[good], lb.b[good],
        lb.c[good], is2ndXtal, mainPart, info, isMulti, needElevationMap, surf,
        invertNormal)
    if needElevationMap and elev:                 # reflect.py:651-659
        lb.elevationD[good] = elev[0]
        if is_param(surf):
            tX, tY, tZ = param_to_xyz(surf, elev[1], elev[2], elev[3])
        else:
            tX, tY, tZ = elev[1], elev[2], elev[3]
        lb.elevationX[good] = tX
        lb.elevationY[good] = tY
        lb.elevationZ[good] = tZ
    if info is not None:
        info['tMin'] = tMin.copy()
        info['tMax0'] = tMax.copy()

    _lost = None
    if noIntersectionSearch:
        tMax[good] = 0.
        if is_param(surf):                        # reflect.py:679-682
            lb.x[good], lb.y[good], lb.z[good] = xyz_to_param(
                surf, lb.x[good], lb.y[good], lb.z[good])
    else:
        res = find_intersection(
            surf, tMin[good], tMax[good], lb.x[good], lb.y[good], lb.z[good],
            lb.a[good], lb.b[good], lb.c[good], invertNormal, info)
        tMax[good], lb.x[good], lb.y[good], lb.z[good] = res[:4]
        _lost = res[4]

    if is_param(surf):                            # reflect.py:701-704
        tX, tY, _ = param_to_xyz(surf, lb.x[good], lb.y[good], lb.z[good])
    else:
        tX, tY = lb.x[good], lb.y[good]
    gNormal = None
    if 'gfzp' in oe:                              # reflect.py:706-707 (use_rays_good_gn)
        lb.state[good], gNormal = general_fzp_rays_good_gn(oe, tX, tY, lb.z[good])
    elif 'fzp' in oe:                             # reflect.py:706-707
        lb.state[good], gNormal = fzp_rays_good_gn(oe, tX, tY)
    else:
        lb.state[good] = rays_good(oe, tX, tY, is2ndXtal)
        if surf['kind'] == 'diced':               # DicedOE.rays_good, bragg.py:92-101
            fx, fy = _diced_facet(surf, tX, tY)[:2]
            inGaps = (abs(fx) > surf['dxFacet']/2) | (abs(fy) > surf['dyFacet']/2)
            st = lb.state[good]
            st[inGaps] = oe['lostNum']
            lb.state[good] = st
    if _lost is not None:
        lb.state[np.where(good)[0][_lost]] = oe['lostNum']

    goodN = (lb.state == 1)
    goodNsum = goodN.sum()
    if goodNsum > 0:
        lb.path[goodN] += tMax[goodN]
        toWhere = 0
        matSur = material
        kind = None
        if material is not None:
            kind = matSur['kind']
            if kind in ('plate', 'lens'):
                toWhere = 1
            elif kind in ('crystal', 'multilayer'):       # reflect.py:734-741
                if matSur['geom'].endswith('transmitted'):
                    toWhere = 2
            elif kind == 'grating':               # reflect.py:743-744
                toWhere = 3
            elif kind == 'FZP':
                toWhere = 4
            elif kind not in ('mirror', 'thin mirror'):
                raise ValueError('unsupported material kind ' + kind)

        oeNormal = list(local_n(surf, lb.x[goodN], lb.y[goodN]))
        if surf.get('figure_n') is not None:      # reflect.py:767-775: [d_pitch, d_roll]
            d_pitch, d_roll = surf['figure_n'](lb.x[goodN], lb.y[goodN])
            cosX, sinX = np.cos(d_pitch), np.sin(d_pitch)
            oeNormal[-2], oeNormal[-1] = (cosX*oeNormal[-2] - sinX*oeNormal[-1],
                                          sinX*oeNormal[-2] + cosX*oeNormal[-1])
            cosY, sinY = np.cos(d_roll), np.sin(d_roll)
            oeNormal[-3], oeNormal[-1] = (cosY*oeNormal[-3] + sinY*oeNormal[-1],
                                          -sinY*oeNormal[-3] + cosY*oeNormal[-1])
        isAsymmetric = len(oeNormal) == 6
        oeNormal = np.asarray(
            [np.broadcast_to(np.asarray(v, dtype=float), lb.x[goodN].shape)
             for v in oeNormal], order='F')
        beamInDotNormal = lb.a[goodN]*oeNormal[0] +\
            lb.b[goodN]*oeNormal[1] + lb.c[goodN]*oeNormal[2]
        lb.theta = np.zeros_like(lb.x)
        beamInDotNormal[beamInDotNormal < -1] = -1
        beamInDotNormal[beamInDotNormal > 1] = 1
        lb.theta[goodN] = np.arccos(beamInDotNormal) - np.pi/2
        if isAsymmetric:
            beamInDotSurfaceNormal = lb.a[goodN]*oeNormal[-3] +\
                lb.b[goodN]*oeNormal[-2] + lb.c[goodN]*oeNormal[-1]
        else:
            beamInDotSurfaceNormal = beamInDotNormal

        if toWhere == 3:                          # reflect.py:840-861
            if is_param(surf):
                raise ValueError('gratings on parametric surfaces not restated')
            g = local_g(oe, lb.x[goodN], lb.y[goodN])
            drawn = []
            lb.a[goodN], lb.b[goodN], lb.c[goodN] = grating_deflection(
                lb.a[goodN], lb.b[goodN], lb.c[goodN], lb.E[goodN], g, oeNormal,
                beamInDotSurfaceNormal, oe.get('order', 1), -1, drawn)
            lb.order = np.zeros(len(lb.a))               # :457-458
            lb.order[goodN] = drawn[0]
        elif toWhere == 4:                        # zone plate: reflect.py:840, 857-861
            drawn = []
            lb.a[goodN], lb.b[goodN], lb.c[goodN] = grating_deflection(
                lb.a[goodN], lb.b[goodN], lb.c[goodN], lb.E[goodN],
                np.asarray(gNormal, order='F'), oeNormal, beamInDotSurfaceNormal,
                oe.get('order', 1), 1, drawn)
            lb.order = np.zeros(len(lb.a))
            lb.order[goodN] = drawn[0]
        elif toWhere in (0, 2):
            if kind in ('crystal', 'multilayer') and toWhere == 0:   # reflect.py:865-872
                a_out, b_out, c_out = asymmetric_reflection_grating(
                    matSur, lb.a[goodN], lb.b[goodN], lb.c[goodN], lb.E[goodN],
                    oeNormal, beamInDotSurfaceNormal, beamInDotNormal)
            else:
                a_out = lb.a[goodN] - oeNormal[0]*2*beamInDotNormal
                b_out = lb.b[goodN] - oeNormal[1]*2*beamInDotNormal
                c_out = lb.c[goodN] - oeNormal[2]*2*beamInDotNormal
            if toWhere == 0:
                lb.a[goodN] = a_out
                lb.b[goodN] = b_out
                lb.c[goodN] = c_out
        elif toWhere == 1:                        # reflect.py:894-919
            refractive_index = mat.refractive_index(matSur, lb.E[goodN]).real
            if fromVacuum:
                n1overn2 = 1. / refractive_index
            else:
                n1overn2 = refractive_index
            signN = np.sign(-beamInDotNormal)
            n1overn2cosTheta1 = -n1overn2 * beamInDotNormal
            cosTheta2 = signN * \
                np.sqrt(1 - n1overn2**2 + n1overn2cosTheta1**2)
            dn = (n1overn2cosTheta1 - cosTheta2)
            lb.a[goodN] = lb.a[goodN] * n1overn2 + oeNormal[0]*dn
            lb.b[goodN] = lb.b[goodN] * n1overn2 + oeNormal[1]*dn
            lb.c[goodN] = lb.c[goodN] * n1overn2 + oeNormal[2]*dn

        rollAngle = roll + np.arctan2(oeNormal[-3], oeNormal[-1])
        localJ = rotate_coherency_matrix(lb, goodN, -rollAngle)
        if hasattr(lb, 'Es'):
            cosY, sinY = np.cos(rollAngle), np.sin(rollAngle)
            lb.Es[goodN], lb.Ep[goodN] = rotate_y(
                lb.Es[goodN], lb.Ep[goodN], cosY, -sinY)

        if material is not None:
            if kind == 'crystal':
                beamOutDotSurfaceNormal = a_out*oeNormal[-3] + \
                    b_out*oeNormal[-2] + c_out*oeNormal[-1]
                refl = mat.crystal_amplitude(
                    matSur, lb.E[goodN], beamInDotSurfaceNormal,
                    beamOutDotSurfaceNormal, beamInDotNormal)
            elif kind == 'multilayer':                # reflect.py:999-1003
                refl = mat.multilayer_amplitude(
                    matSur, lb.E[goodN], beamInDotSurfaceNormal)
            elif matSur.get('layered'):               # Coated, kind 'mirror': :1031-1032
                refl = mat.multilayer_amplitude(matSur, lb.E[goodN], beamInDotNormal)
            elif kind in ('grating', 'FZP') and matSur.get('efficiency') is not None:
                # Material.get_grating_efficiency, material.py:391-413: constant values, or
                # (efficiencyFile) columns of a table against energy
                resI = np.zeros(goodN.sum())
                order = lb.order[goodN]
                if matSur.get('efficiency_table') is None:
                    for eff in matSur['efficiency']:
                        resI[order == eff[0]] = eff[1]
                else:
                    tabE, tabI = matSur['efficiency_table']
                    E = lb.E[goodN]
                    if np.any(E < tabE[0]) or np.any(E > tabE[-1]):         # :399-407
                        raise ValueError('E={0} is out of the efficiency table range [{1}, {2}]'
                                         .format(E[(E < tabE[0]) | (E > tabE[-1])],
                                                 tabE[0], tabE[-1]))
                    for ieff, eff in enumerate(matSur['efficiency']):
                        resI[order == eff[0]] = np.interp(E[order == eff[0]], tabE, tabI[ieff])
                resA = resI**0.5
                refl = resA, resA, 0
            else:
                refl = mat.material_amplitude(
                    matSur, lb.E[goodN], beamInDotNormal, fromVacuum)
        else:
            refl = 1., 1.
        ras, rap = refl[0], refl[1]
        if isinstance(ras, np.ndarray):
            ras[np.isnan(ras)] = 0.
            rap[np.isnan(rap)] = 0.

        lb.Jss[goodN] = (localJ[0] * ras * np.conjugate(ras)).real
        lb.Jpp[goodN] = (localJ[1] * rap * np.conjugate(rap)).real
        lb.Jsp[goodN] = localJ[2] * ras * np.conjugate(rap)
        if hasattr(lb, 'Es'):
            lb.Es[goodN] *= ras
            lb.Ep[goodN] *= rap

        if (not fromVacuum) and material is not None and \
                kind not in ('crystal', 'multilayer'):
            att = np.exp(-refl[2] * tMax[goodN] * 0.1)
            lb.Jss[goodN] *= att
            lb.Jpp[goodN] *= att
            lb.Jsp[goodN] *= att
            if hasattr(lb, 'Es'):
                mPh = att**0.5 * np.exp(0.1j * refl[3] * tMax[goodN])
                lb.Es[goodN] *= mPh
                lb.Ep[goodN] *= mPh
        else:
            if hasattr(lb, 'Es'):
                mPh = np.exp(1e7j * lb.E[goodN]/CHBAR * tMax[goodN])
                lb.Es[goodN] *= mPh
                lb.Ep[goodN] *= mPh

        vlb.Jss[goodN], vlb.Jpp[goodN], vlb.Jsp[goodN] =\
            rotate_coherency_matrix(lb, goodN, rollAngle)
        if hasattr(lb, 'Es'):
            vlb.Es[goodN], vlb.Ep[goodN] = rotate_y(
                lb.Es[goodN], lb.Ep[goodN], cosY, sinY)

    if is_param(surf):                            # reflect.py:1066-1071
        lb.s = np.copy(lb.x)
        lb.phi = np.copy(lb.y)
        lb.r = np.copy(lb.z)
        lb.x[good], lb.y[good], lb.z[good] = param_to_xyz(
            surf, lb.x[good], lb.y[good], lb.z[good])

    if vlb is not lb:
        copy_beam(vlb, lb, good, includeState=True, includeJspEsp=False)
    if dx:
        vlb.x[good] += dx
    if dy:
        vlb.y[good] += dy
    if dz:
        vlb.z[good] += dz
    if extra[0] or extra[1] or extra[2]:
        rotate_beam(vlb, good, '-' + extraSeq, pitch=extraAnglesSign*extra[0],
                    roll=extra[1], yaw=extraAnglesSign*extra[2])
    rotate_beam(vlb, good, '-' + rotSeq, pitch=pitch, roll=roll, yaw=yaw)
    if is2ndXtal:
        rotate_beam(vlb, good, roll=np.pi)
    if info is not None:
        info['tMax'] = tMax


def oe_reflect(oe, beam, noIntersectionSearch=False, createdByDiffract=False,
               info=None):
    """OE.reflect (reflect.py:18-163) -> (gb, lb)."""
    gb = beam.copy()
    lb = beam.copy()
    good = beam.state > 0
    if good.sum() == 0:
        return gb, lb
    pitch = oe['pitch']
    global_to_virgin_local(oe['azimuth_sc'], beam, lb, oe['center'], good)
    reflect_local(oe, good, lb, gb, pitch, oe['roll'] + oe['positionRoll'],
                  oe['yaw'], oe.get('dx', 0),
                  noIntersectionSearch=noIntersectionSearch,
                  material=oe.get('material'), info=info)
    if createdByDiffract:
        goodAfter = gb.state == 1
    else:
        goodAfter = (gb.state == 1) | (gb.state == 2)
    if goodAfter.sum() > 0:
        virgin_local_to_global(oe['azimuth_sc'], gb, oe['center'], goodAfter)
    notGood = ~goodAfter
    if notGood.sum() > 0:
        copy_beam(gb, beam, notGood)
    return gb, lb


def oe_multiple_reflect(oe, beam, maxReflections=1000, needElevationMap=False,
                        info=None):
    """OE.multiple_reflect (reflect.py:165-264) -> (gb, lbN): up to *maxReflections* bounces
    off the same surface. The beam stays in the element's virgin local frame between the
    bounces (``lb is gb``); lbN holds a copy of ALL rays after every bounce, one after the
    other, with ``nRefl`` (and, on request, the elevation fields of the points between two
    bounces where a ray was farthest from the surface). *info*: a list that receives one
    dictionary of batch statistics per bounce."""
    gb = beam.copy()
    lb = gb
    good = beam.state > 0
    if good.sum() == 0:
        return gb, lb
    global_to_virgin_local(oe['azimuth_sc'], beam, lb, oe['center'], good)
    iRefl = 0
    isMulti = False
    lbN = None
    while iRefl < maxReflections:
        tmpX, tmpY, tmpZ = np.copy(lb.x[good]), np.copy(lb.y[good]), np.copy(lb.z[good])
        if iRefl == 0 and needElevationMap:
            lb.elevationD = -np.ones_like(lb.x)
            lb.elevationX = -np.ones_like(lb.x)*maxHalfSizeOfOE
            lb.elevationY = -np.ones_like(lb.x)*maxHalfSizeOfOE
            lb.elevationZ = -np.ones_like(lb.x)*maxHalfSizeOfOE
        one = {} if info is not None else None
        reflect_local(oe, good, lb, gb, oe['pitch'], oe['roll'] + oe['positionRoll'],
                      oe['yaw'], oe.get('dx', 0), material=oe.get('material'), info=one,
                      needElevationMap=needElevationMap, isMulti=isMulti)
        if info is not None:
            info.append(one)
        if iRefl == 0:
            isMulti = True
            lb.nRefl = np.zeros_like(lb.state)
        ov = lb.state[good] == 3                  # over the edge: back to where it was
        where = np.where(good)[0][ov]
        lb.x[where] = tmpX[ov]
        lb.y[where] = tmpY[ov]
        lb.z[where] = tmpZ[ov]
        good = (lb.state == 1) | (lb.state == 2)
        lb.nRefl[good] += 1
        if iRefl == 0:
            lbN = lb.copy()
        else:
            lbN.concatenate(lb)
        iRefl += 1
        if good.sum() == 0:
            break
    goodAfter = gb.nRefl > 0
    gb.state[goodAfter] = 1
    if goodAfter.sum() > 0:
        virgin_local_to_global(oe['azimuth_sc'], gb, oe['center'], goodAfter)
    notGood = ~goodAfter
    if notGood.sum() > 0:
        copy_beam(gb, beam, notGood)
    return gb, lbN


def dcm_double_reflect(oe, beam, fromVacuum1=True, fromVacuum2=True,
                       is_plate=False, info=None):
    """DCM.double_reflect (dcm.py:248-354) -> (gb2, lo1, lo2)."""
    gb = beam.copy()
    lo1 = beam.copy()
    good1 = beam.state > 0
    if good1.sum() == 0:
        return gb, lo1, lo1
    global_to_virgin_local(oe['azimuth_sc'], beam, lo1, oe['center'], good1)
    i1 = {} if info is not None else None
    reflect_local(
        oe, good1, lo1, gb, oe['pitch'] + oe['bragg'],
        oe['roll'] + oe['positionRoll'] + oe['cryst1roll'], oe['yaw'],
        oe.get('dx', 0), surf=oe['surface'], fromVacuum=fromVacuum1,
        material=oe.get('material'), info=i1)
    goodAfter1 = (gb.state == 1) | (gb.state == 2)
    notGood = ~goodAfter1
    if notGood.sum() > 0:
        copy_beam(gb, beam, notGood)
    gb2 = gb.copy()
    lo2 = gb2.copy()
    good2 = goodAfter1
    if (~good2).sum() > 0:
        lo2.state[~good2] = 0
        lo2.x[~good2] = 0.
        lo2.y[~good2] = 0.
        lo2.z[~good2] = 0.
    if is_plate:
        gb2.state[~good2] = oe['lostNum']
    if good2.sum() == 0:
        return gb2, lo1, lo2
    i2 = {} if info is not None else None
    reflect_local(
        oe, good2, lo2, gb2,
        -oe['pitch'] - oe['bragg'] + oe['cryst2pitch'] + oe['cryst2finePitch'],
        oe['roll'] + oe['cryst2roll'] + oe['positionRoll'], -oe['yaw'],
        -oe.get('dx', 0), oe['cryst2longTransl'], -oe['cryst2perpTransl'],
        surf=oe['surface2'], fromVacuum=fromVacuum2,
        material=oe.get('material2'), is2ndXtal=True, info=i2)
    goodAfter2 = (gb2.state == 1) | (gb2.state == 2)
    virgin_local_to_global(oe['azimuth_sc'], gb2, oe['center'], goodAfter2)
    notGood = ~goodAfter2
    if is_plate:
        gb2.state[notGood] = oe['lostNum']
    if notGood.sum() > 0:
        copy_beam(gb2, beam, notGood)
    if info is not None:
        info['crystal1'] = i1
        info['crystal2'] = i2
    return gb2, lo1, lo2


def rotate_point(point, rotationSequence='RzRyRx', pitch=0, roll=0, yaw=0):
    """_rotate.py:87-108."""
    angles = {'z': yaw, 'y': roll, 'x': pitch}
    rotates = {'z': rotate_z, 'y': rotate_y, 'x': rotate_x}
    ind1 = {'z': 0, 'y': 0, 'x': 1}
    ind2 = {'z': 1, 'y': 2, 'x': 2}
    newp = [coord for coord in point]
    if rotationSequence[0] == '-':
        seq = rotationSequence[6] + rotationSequence[4] + rotationSequence[2]
    else:
        seq = rotationSequence[1] + rotationSequence[3] + rotationSequence[5]
    for s in seq:
        angle, rotate = angles[s], rotates[s]
        if angle != 0:
            cA = np.cos(angle)
            sA = np.sin(angle)
            newp[ind1[s]], newp[ind2[s]] = rotate(
                newp[ind1[s]], newp[ind2[s]], cA, sA)
    return newp


def lens_multiple_refract(oe, beam):
    """ParaboloidFlatLens.multiple_refract (oes/refractive.py:457-519): nCRL
    double_refract passes, the centre of the stack moved by `step` along the rotated
    local z between them; -> (global beam behind the last lenslet, the two local beams
    of the FIRST lenslet). oe: the DCM/Plate dictionary + nCRL, zmax, t, double_sided."""
    if oe['nCRL'] == 1:
        return dcm_double_reflect(oe, beam, True, False, is_plate=True)
    oe = dict(oe)
    oe['center'] = [c for c in oe['center']]
    beamIn = beam
    zmax = 5 if oe['zmax'] is None else oe['zmax']
    step = 2.*zmax + oe['t'] if oe['double_sided'] else zmax + oe['t']
    for ilens in range(oe['nCRL']):
        lglobal, tlocal1, tlocal2 = dcm_double_reflect(oe, beamIn, True, False,
                                                       is_plate=True)
        if oe['zmax'] is not None:
            toward = rotate_point([0, 0, 1], oe['rotationSequence'], oe['pitch'],
                                  oe['roll']+oe['positionRoll'], oe['yaw'])
            oe['center'][0] -= step * toward[0]
            oe['center'][1] -= step * toward[1]
            oe['center'][2] -= step * toward[2]
        beamIn = lglobal
        if ilens == 0:
            llocal1, llocal2 = tlocal1, tlocal2
    return lglobal, llocal1, llocal2
