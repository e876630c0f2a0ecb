"""Wave propagation — host-side mirror of xrt/backends/raycing/waves.py:505-896.

``diffract(oeLocal, wave)`` keeps xrt's semantics: the O(Ns*Np) Fresnel-Kirchhoff
integral runs in the HIP kernel (csrc/kirchhoff.hip, through
``hipcalls.kirchhoff``); the O(Ns+Np) bookkeeping around it (sample selection,
footprint area, normalisation, phase strip, frame changes) is host glue, like in
the reference. Sign convention: the numpy path of the reference
(``_diffraction_integral_conv``, +i k/4pi).
"""
import time

import numpy as np
import torch

from .. import raycing
from ... import _lib, hipcalls
from . import sources as rs
from .physconsts import CH, CHBAR

_DEBUG = 0
lastKernelMs = None


def prepare_wave(fromOE, wave, xglo, yglo, zglo):
    """Receiving points *xglo, yglo, zglo* expressed in the local frame of the
    diffracting element *fromOE*; zeroed accumulators (waves.py:505-584)."""
    if not hasattr(wave, 'Es'):
        nrays = len(wave.x)
        wave.Es = np.zeros(nrays, dtype=complex)
        wave.Ep = np.zeros(nrays, dtype=complex)
    else:
        wave.Es[:] = 0
        wave.Ep[:] = 0
    wave.EsAcc = np.zeros_like(wave.Es)
    wave.EpAcc = np.zeros_like(wave.Es)
    wave.aEacc = np.zeros_like(wave.Es)
    wave.bEacc = np.zeros_like(wave.Es)
    wave.cEacc = np.zeros_like(wave.Es)
    wave.Jss[:] = 0
    wave.Jpp[:] = 0
    wave.Jsp[:] = 0
    x, y, z = np.array(xglo, dtype=float), np.array(yglo, dtype=float), \
        np.array(zglo, dtype=float)
    x -= fromOE.center[0]
    y -= fromOE.center[1]
    z -= fromOE.center[2]
    a0, b0 = fromOE.bl.sinAzimuth, fromOE.bl.cosAzimuth
    x[:], y[:] = raycing.rotate_z(x, y, b0, a0)
    if hasattr(fromOE, 'rotationSequence'):  # OE
        dt = 0
        extraAnglesSign = 1.
        if hasattr(fromOE, 'local_n2') and hasattr(fromOE, 'cryst2pitch'):
            raise NotImplementedError('wave propagation from a DCM/plate 2nd '
                                      'surface')
        raycing.rotate_xyz(
            x, y, z, rotationSequence=fromOE.rotationSequence,
            pitch=-fromOE.pitch, roll=-(fromOE.roll+fromOE.positionRoll),
            yaw=-fromOE.yaw)
        if fromOE.extraPitch or fromOE.extraRoll or fromOE.extraYaw:
            raycing.rotate_xyz(
                x, y, z, rotationSequence=fromOE.extraRotationSequence,
                pitch=-extraAnglesSign*fromOE.extraPitch,
                roll=-fromOE.extraRoll, yaw=-extraAnglesSign*fromOE.extraYaw)
        if dt:
            z += dt
    wave.xDiffr = x
    wave.yDiffr = y
    wave.zDiffr = z
    wave.rDiffr = (wave.xDiffr**2 + wave.yDiffr**2 + wave.zDiffr**2)**0.5
    wave.a[:] = wave.xDiffr / wave.rDiffr
    wave.b[:] = wave.yDiffr / wave.rDiffr
    wave.c[:] = wave.zDiffr / wave.rDiffr
    wave.path[:] = 0.
    wave.fromOE = fromOE
    wave.beamReflRays = np.int64(0)
    wave.beamReflSumJ = 0.
    wave.beamReflSumJnl = 0.
    wave.diffract_repeats = np.int64(0)
    return wave


def qualify_sampling(wave, E, goodlen):
    """Effective Fresnel number and samples per Fresnel zone
    (waves.py:587-603)."""
    a = wave.xDiffr / wave.rDiffr
    c = wave.zDiffr / wave.rDiffr
    NAx = (a.max() - a.min()) * 0.5
    NAz = (c.max() - c.min()) * 0.5
    invLambda = E / CH * 1e7
    fn = (NAx**2 + NAz**2) * wave.rDiffr.mean() * invLambda
    samplesPerZone = abs(goodlen / fn)
    return fn, samplesPerZone


def convex_hull_area(px, py):
    """Area of the convex hull of 2-D points (Andrew's monotone chain); the
    reference takes it from scipy.spatial.ConvexHull (waves.py:661-668)."""
    px = np.asarray(px, dtype=float)
    py = np.asarray(py, dtype=float)
    if len(px) > 64:
        # Akl-Toussaint pre-filter: points strictly inside the octagon of the
        # extreme points in 8 directions cannot be hull vertices
        ext = []
        for key in (px, px + py, py, py - px, -px, -px - py, -py, px - py):
            i = int(np.argmax(key))
            if not ext or (px[i], py[i]) != ext[-1]:
                ext.append((px[i], py[i]))
        if len(ext) > 1 and ext[0] == ext[-1]:
            ext.pop()
        if len(ext) >= 3:
            inside = np.ones(len(px), dtype=bool)
            for (x1, y1), (x2, y2) in zip(ext, ext[1:] + ext[:1]):
                inside &= (x2-x1)*(py-y1) - (y2-y1)*(px-x1) > 0
            keep = ~inside
            px, py = px[keep], py[keep]
    pts = np.unique(np.column_stack((px, py)), axis=0)
    if len(pts) < 3:
        raise ValueError('cannot normalize this way!')

    def half(points):
        hull = []
        for p in points:
            while len(hull) >= 2:
                o, a = hull[-2], hull[-1]
                if (a[0]-o[0])*(p[1]-o[1]) - (a[1]-o[1])*(p[0]-o[0]) <= 0:
                    hull.pop()
                else:
                    break
            hull.append((p[0], p[1]))
        return hull
    lower = half(pts)
    upper = half(pts[::-1])
    outer = np.array(lower[:-1] + upper[:-1])
    x1, y1 = outer[:, 0], outer[:, 1]
    x2, y2 = np.roll(x1, -1), np.roll(y1, -1)
    return 0.5 * abs(np.sum(x1*y2 - x2*y1))


def _kirchhoff_on_gpu(oeLocal, n, nl, wave, good):
    """(Es, Ep, aE, bE, cE) of waves.py:834-851, computed by the HIP kernel."""
    global lastKernelMs
    _lib.require_gpu()
    dev = torch.device('cuda', torch.cuda.current_device())

    def up(a, dtype=np.float64):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(dev)
    shape = oeLocal.x[good].shape
    n3 = [np.broadcast_to(np.asarray(c, dtype=float), oeLocal.x.shape)[good]
          if np.ndim(c) else np.full(shape, float(c)) for c in n]
    k = oeLocal.E[good] / CHBAR * 1e7                      # waves.py:841
    out = hipcalls.kirchhoff(
        up(wave.xDiffr), up(wave.yDiffr), up(wave.zDiffr),
        up(oeLocal.x[good]), up(oeLocal.y[good]), up(oeLocal.z[good]),
        up(n3[0]), up(n3[1]), up(n3[2]), up(nl[good]), up(k),
        up(oeLocal.Es[good], np.complex128), up(oeLocal.Ep[good], np.complex128),
        convention=0, timing=True)
    lastKernelMs = out[5]
    return [o.cpu().numpy() for o in out[:5]]


def diffract(oeLocal, wave, targetOpenCL=raycing.targetOpenCL,
             precisionOpenCL=raycing.precisionOpenCL):
    """Diffracted field on the points of *wave* from the field *oeLocal* on the
    diffracting surface ``wave.fromOE`` (waves.py:606-831). Returns the global
    beam; *wave* accumulates over repeated calls."""
    oe = wave.fromOE
    t0 = time.time()
    good = oeLocal.state == 1
    goodlen = good.sum()
    if goodlen < 1e2:
        print("Not enough good rays at {0}: {1} of {2}".format(
            oe.name, goodlen, len(oeLocal.x)))
        return rs.Beam(goodlen)
    if len(wave.xDiffr) == 0:
        print("No wave samples on {0}".format(oe.name))
        return rs.Beam(goodlen)

    shouldCalculateArea = False
    if not hasattr(oeLocal, 'area'):
        shouldCalculateArea = True
    elif oeLocal.area is None or (oeLocal.area <= 0):
        shouldCalculateArea = True
    if shouldCalculateArea:
        if hasattr(oe, 'rotationSequence'):
            secondDim = oeLocal.y
        elif hasattr(oe, 'propagate') or hasattr(oe, 'prepare_wave') or \
                hasattr(oe, 'shine'):
            secondDim = oeLocal.z
        else:
            raise ValueError('Unknown diffracting element!')
        oeLocal.area = convex_hull_area(oeLocal.x[good], secondDim[good])
        if hasattr(oeLocal, 'areaFraction'):
            oeLocal.area *= oeLocal.areaFraction

    if hasattr(oe, 'rotationSequence'):  # OE
        local_n = oe.local_n2 if hasattr(oe, 'cryst2pitch') else oe.local_n
        if oe.isParametric:                 # waves.py:680-682
            sp, phi, _ = oe.xyz_to_param(oeLocal.x, oeLocal.y, oeLocal.z)
            n = local_n(sp, phi)
        else:
            n = local_n(oeLocal.x, oeLocal.y)[-3:]
        nl = (oeLocal.a*np.asarray([n[-3]]) + oeLocal.b*np.asarray([n[-2]]) +
              oeLocal.c*np.asarray([n[-1]])).flatten()
    else:
        n = [0, 1, 0]
        nl = oeLocal.a*n[0] + oeLocal.b*n[1] + oeLocal.c*n[2]

    wave.diffract_repeats += 1
    wave.beamReflRays += goodlen
    wave.beamReflSumJ += (oeLocal.Jss[good] + oeLocal.Jpp[good]).sum()
    wave.beamReflSumJnl += abs(((oeLocal.Jss[good] + oeLocal.Jpp[good]) *
                               nl[good]).sum())

    Es, Ep, aE, bE, cE = _kirchhoff_on_gpu(oeLocal, n, nl, wave, good)

    wave.EsAcc += Es
    wave.EpAcc += Ep
    wave.aEacc += aE
    wave.bEacc += bE
    wave.cEacc += cE
    wave.E[:] = oeLocal.E[0]
    wave.Es[:] = wave.EsAcc
    wave.Ep[:] = wave.EpAcc
    wave.Jss[:] = (wave.Es * np.conj(wave.Es)).real
    wave.Jpp[:] = (wave.Ep * np.conj(wave.Ep)).real
    wave.Jsp[:] = wave.Es * np.conj(wave.Ep)

    if hasattr(oe, 'rotationSequence'):  # OE: waves.py:719-722
        toRealComp = wave.cEacc if abs(wave.cEacc[0]) > abs(wave.bEacc[0]) \
            else wave.bEacc
        toReal = np.exp(-1j * np.angle(toRealComp))
    else:
        toReal = np.exp(-1j * np.angle(wave.bEacc))
    wave.a[:] = (wave.aEacc * toReal).real
    wave.b[:] = (wave.bEacc * toReal).real
    wave.c[:] = (wave.cEacc * toReal).real
    norm = (wave.a**2 + wave.b**2 + wave.c**2)**0.5
    norm[norm == 0] = 1.
    wave.a /= norm
    wave.b /= norm
    wave.c /= norm

    norm = wave.dS * oeLocal.area * wave.beamReflSumJ
    de = wave.beamReflRays * wave.beamReflSumJnl * wave.diffract_repeats
    if de > 0:
        norm /= de
    else:
        norm = 0
    wave.Jss *= norm
    wave.Jpp *= norm
    wave.Jsp *= norm
    wave.Es *= norm**0.5
    wave.Ep *= norm**0.5
    if hasattr(oeLocal, 'accepted'):
        wave.accepted = oeLocal.accepted
        wave.acceptedE = oeLocal.acceptedE
        wave.seeded = oeLocal.seeded
        wave.seededI = oeLocal.seededI * len(wave.x) / len(oeLocal.x)

    glo = rs.Beam(copyFrom=wave)
    glo.parentId = oe.uuid
    glo.x[:] = wave.xDiffr
    glo.y[:] = wave.yDiffr
    glo.z[:] = wave.zDiffr
    if hasattr(oe, 'local_to_global'):
        if hasattr(oe, 'expose'):  # a Screen
            glo.x[:], glo.y[:], glo.z[:] = \
                oe.local_to_global(glo.x, glo.y, glo.z)
        else:
            oe.local_to_global(glo)

    if hasattr(wave, 'toOE'):          # waves.py:773-824
        if hasattr(oe, 'rotationSequence'):
            rollAngle = oe.roll + oe.positionRoll
            cosY, sinY = np.cos(rollAngle), np.sin(rollAngle)
            Es[:], Ep[:] = raycing.rotate_y(Es, Ep, cosY, sinY)
        toOE = wave.toOE
        wave.a[:], wave.b[:], wave.c[:] = glo.a, glo.b, glo.c
        wave.Jss[:], wave.Jpp[:], wave.Jsp[:] = glo.Jss, glo.Jpp, glo.Jsp
        wave.Es[:], wave.Ep[:] = glo.Es, glo.Ep
        a0, b0 = toOE.bl.sinAzimuth, toOE.bl.cosAzimuth
        wave.a[:], wave.b[:] = raycing.rotate_z(wave.a, wave.b, b0, a0)
        if hasattr(toOE, 'rotationSequence'):  # the receiver is an OE
            if toOE.isParametric:          # waves.py:791-793
                sp, phi, _ = toOE.xyz_to_param(wave.x, wave.y, wave.z)
                oeNormal = list(toOE.local_n(sp, phi))
            else:
                oeNormal = list(toOE.local_n(wave.x, wave.y))
            rollAngle = toOE.roll + toOE.positionRoll +\
                np.arctan2(oeNormal[-3], oeNormal[-1])
            wave.Jss[:], wave.Jpp[:], wave.Jsp[:] = \
                rs.rotate_coherency_matrix(wave, slice(None), -rollAngle)
            cosY, sinY = np.cos(rollAngle), np.sin(rollAngle)
            wave.Es[:], wave.Ep[:] = raycing.rotate_y(
                wave.Es, wave.Ep, cosY, -sinY)
            raycing.rotate_xyz(
                wave.a, wave.b, wave.c, rotationSequence=toOE.rotationSequence,
                pitch=-toOE.pitch, roll=-toOE.roll-toOE.positionRoll,
                yaw=-toOE.yaw)
            if toOE.extraPitch or toOE.extraRoll or toOE.extraYaw:
                raycing.rotate_xyz(
                    wave.a, wave.b, wave.c,
                    rotationSequence=toOE.extraRotationSequence,
                    pitch=-toOE.extraPitch, roll=-toOE.extraRoll,
                    yaw=-toOE.extraYaw)
            norm = -wave.a*oeNormal[-3] - wave.b*oeNormal[-2] -\
                wave.c*oeNormal[-1]
            norm = np.abs(norm)
            for b in (wave, glo):
                b.Jss *= norm
                b.Jpp *= norm
                b.Jsp *= norm
                b.Es *= norm**0.5
                b.Ep *= norm**0.5
    if _DEBUG > 10:
        print("diffract on {0} completed in {1:.4f} s".format(
            oe.name, time.time()-t0))
    glo.createdByDiffract = True
    return glo
