"""TEST INFRASTRUCTURE ONLY — numpy restatement of the two streaming elements
either side of the ray-surface path:

* screen_expose       <- xrt/backends/raycing/screens.py:226-302 with
                         global_to_virgin_local(xyz basis) beamline.py:253-264
* aperture_propagate  <- xrt/backends/raycing/apertures.py:334-413

Pinned by tests/golden/g1_source_screen.npz and g7_aperture.npz
(oracle/gen_fixtures_p1.py)."""
import numpy as np

from .consts import CHBAR
from .reflect_np import rotate_z


def _to_basis(beam, lo, basis, center, part):
    lx, ly, lz = basis
    lo.x[part] = beam.x[part] - center[0]
    lo.y[part] = beam.y[part] - center[1]
    lo.z[part] = beam.z[part] - center[2]
    xyz = lo.x[part], lo.y[part], lo.z[part]
    lo.x[part], lo.y[part], lo.z[part] = (
        sum(c*b for c, b in zip(lx, xyz)), sum(c*b for c, b in zip(ly, xyz)),
        sum(c*b for c, b in zip(lz, xyz)))
    abc = beam.a[part], beam.b[part], beam.c[part]
    lo.a[part], lo.b[part], lo.c[part] = (
        sum(c*b for c, b in zip(lx, abc)), sum(c*b for c, b in zip(ly, abc)),
        sum(c*b for c, b in zip(lz, abc)))


def screen_expose(beam, basis, center, lostNum, onlyPositivePath=False):
    blo = beam.copy()
    part = np.ones(beam.x.shape, dtype=bool)
    _to_basis(beam, blo, basis, center, part)
    with np.errstate(divide='ignore', invalid='ignore'):
        path = -blo.y / blo.b
    condBad = np.isnan(path) | np.isinf(path)
    if onlyPositivePath:
        condBad = condBad | (path < 0)
    path[condBad] = 0.
    blo.state[condBad] = lostNum
    blo.path += path
    blo.x[:] += blo.a * path
    blo.z[:] += blo.c * path
    blo.y[:] = 0.
    if hasattr(blo, 'Es'):
        propPhase = np.exp(1e7j * (blo.E/CHBAR) * path)
        blo.Es *= propPhase
        blo.Ep *= propPhase
    return blo


def hemispheric_expose(beam, basis, center, R, lostNum, phiOffset=0, thetaOffset=0,
                       onlyPositivePath=False):
    """HemisphericScreen.expose, screens.py:517-559. basis = (x, y, z) axes."""
    blo = beam.copy()
    sqb_2 = (beam.a * (beam.x-center[0]) +
             beam.b * (beam.y-center[1]) +
             beam.c * (beam.z-center[2]))
    sqc = ((beam.x-center[0])**2 +
           (beam.y-center[1])**2 +
           (beam.z-center[2])**2 - R**2)
    with np.errstate(invalid='ignore'):
        path = -sqb_2 + (sqb_2**2 - sqc)**0.5
    condBad = np.isnan(path) | np.isinf(path)
    if onlyPositivePath:
        condBad = condBad | (path < 0)
    path[condBad] = 0.
    blo.state[condBad] = lostNum
    blo.path += path
    rx = beam.x + beam.a*path - center[0]
    ry = beam.y + beam.b*path - center[1]
    rz = beam.z + beam.c*path - center[2]
    ex, ey, ez = basis
    blo.z = rx*ez[0] + ry*ez[1] + rz*ez[2]
    blo.y = rx*ey[0] + ry*ey[1] + rz*ey[2]
    blo.x = rx*ex[0] + ry*ex[1] + rz*ex[2]
    blo.theta = np.arcsin(blo.z / R) - thetaOffset
    blo.phi = np.arctan2(blo.y, blo.x) - phiOffset
    if hasattr(blo, 'Es'):
        propPhase = np.exp(1e7j * (blo.E / CHBAR) * path)
        blo.Es *= propPhase
        blo.Ep *= propPhase
    return blo


def aperture_propagate(beam, basis, center, blades, lostNum, azimuth_sc=(0., 1.),
                       isBeamStop=False, needNewGlobal=False, radius=None,
                       shadeFraction=None, vertices=None):
    """Mutates beam.state like the reference. blades: dict left/right/bottom/top;
    *radius*: a RoundAperture instead (apertures.py:770-846); *shadeFraction*: a DoubleSlit
    (:931-1021); *vertices*: a PolygonalAperture (:1183-1225)."""
    good = beam.state > 0
    lo = beam.copy()
    _to_basis(beam, lo, basis, center, good)
    path = -lo.y[good] / lo.b[good]
    lo.x[good] += lo.a[good] * path
    lo.z[good] += lo.c[good] * path
    lo.path[good] += path
    badIndices = np.zeros(len(beam.x), dtype=bool)
    if radius is not None:
        badIndices[good] = (lo.x[good]**2 + lo.z[good]**2)**0.5 > radius
    for akind, d in blades.items():
        if akind.startswith('l'):
            badIndices[good] = badIndices[good] | (lo.x[good] < d)
        elif akind.startswith('r'):
            badIndices[good] = badIndices[good] | (lo.x[good] > d)
        elif akind.startswith('b'):
            badIndices[good] = badIndices[good] | (lo.z[good] < d)
        elif akind.startswith('t'):
            badIndices[good] = badIndices[good] | (lo.z[good] > d)
    if shadeFraction is not None:
        shadeMin = (1 - shadeFraction) * 0.5
        shadeMax = shadeMin + shadeFraction
        dsb, dst = blades['bottom'], blades['top']
        sb = dsb + (dst - dsb) * shadeMin
        st = dsb + (dst - dsb) * shadeMax
        badIndices[good] = \
            badIndices[good] | ((lo.z[good] > sb) & (lo.z[good] < st))
    if vertices is not None:                      # every ray, entering or not (:1198-1199)
        from .reflect_np import points_in_polygon
        badIndices = np.invert(points_in_polygon(vertices, lo.x, lo.z))
    if isBeamStop:
        badIndices[good] = np.invert(badIndices[good])
    beam.state[badIndices] = lostNum
    lo.state[good] = beam.state[good]
    lo.y[good] = 0.
    if hasattr(lo, 'Es'):
        propPhase = np.exp(1e7j * (lo.E[good]/CHBAR) * path)
        lo.Es[good] *= propPhase
        lo.Ep[good] *= propPhase
    if not needNewGlobal:
        return lo
    glo = lo.copy()
    a0, b0 = azimuth_sc
    if a0 != 0:
        glo.a[good], glo.b[good] = rotate_z(glo.a[good], glo.b[good], b0, -a0)
        glo.x[good], glo.y[good] = rotate_z(glo.x[good], glo.y[good], b0, -a0)
    glo.x[good] += center[0]
    glo.y[good] += center[1]
    glo.z[good] += center[2]
    if shadeFraction is not None:                 # apertures.py:1013
        glo.path[good] += beam.path[good]
    return glo, lo
