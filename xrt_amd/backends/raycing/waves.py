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


_ACCUMULATORS = ('EsAcc', 'EpAcc', 'aEacc', 'bEacc', 'cEacc')


def _into_frame_of(element, xglo, yglo, zglo):
    """Global points -> the local frame of *element*: shift to its centre, undo
    the beamline azimuth and, for an optical element, its own rotations
    (waves.py:529-560)."""
    px = np.array(xglo, dtype=float) - element.center[0]
    py = np.array(yglo, dtype=float) - element.center[1]
    pz = np.array(zglo, dtype=float) - element.center[2]
    bl = element.bl
    px[:], py[:] = raycing.rotate_z(px, py, bl.cosAzimuth, bl.sinAzimuth)
    if hasattr(element, 'rotationSequence'):
        if hasattr(element, 'local_n2') and hasattr(element, 'cryst2pitch'):
            raise NotImplementedError('wave propagation from a DCM/plate 2nd '
                                      'surface')
        raycing.rotate_xyz(px, py, pz, rotationSequence=element.rotationSequence,
                           pitch=-element.pitch,
                           roll=-(element.roll + element.positionRoll),
                           yaw=-element.yaw)
        if element.extraPitch or element.extraRoll or element.extraYaw:
            raycing.rotate_xyz(px, py, pz,
                               rotationSequence=element.extraRotationSequence,
                               pitch=-element.extraPitch, roll=-element.extraRoll,
                               yaw=-element.extraYaw)
    return px, py, pz


def prepare_wave(fromOE, wave, xglo, yglo, zglo):
    """Binds the receiving samples *wave* (global positions *xglo, yglo, zglo*)
    to the diffracting element *fromOE*: their coordinates in its local frame
    (``xDiffr`` ...), unit directions from its centre, and empty field
    accumulators (reference: waves.py:505-584)."""
    nsamples = len(wave.x)
    if hasattr(wave, 'Es'):
        wave.Es[:] = 0
        wave.Ep[:] = 0
    else:
        wave.Es = np.zeros(nsamples, dtype=complex)
        wave.Ep = np.zeros(nsamples, dtype=complex)
    for name in _ACCUMULATORS:
        setattr(wave, name, np.zeros(nsamples, dtype=complex))
    for component in (wave.Jss, wave.Jpp, wave.Jsp):
        component[:] = 0
    px, py, pz = _into_frame_of(fromOE, xglo, yglo, zglo)
    dist = (px**2 + py**2 + pz**2)**0.5
    wave.xDiffr, wave.yDiffr, wave.zDiffr, wave.rDiffr = px, py, pz, dist
    wave.a[:] = px / dist
    wave.b[:] = py / dist
    wave.c[:] = pz / dist
    wave.path[:] = 0.
    wave.fromOE = fromOE
    wave.beamReflRays = np.int64(0)       # running sums over repeated diffract()
    wave.beamReflSumJ = 0.
    wave.beamReflSumJnl = 0.
    wave.diffract_repeats = np.int64(0)
    return wave


def receiving_wave(element, prevOE, local, glob, dS, area, parent):
    """Wave samples at the points *local* (x, y, z in the frame of *element*, which
    owns them) = *glob* in the global frame, bound to the diffracting *prevOE*."""
    wave = rs.Beam(nrays=len(local[0]), forceState=1, withAmplitudes=True)
    for name, values in zip('xyz', local):
        getattr(wave, name)[:] = values
    wave.dS, wave.area, wave.toOE, wave.parentId = dS, area, element, parent
    return prepare_wave(prevOE, wave, *glob)


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


def _illuminated_area(oe, field, lit):
    """Footprint of the lit samples on the diffracting element, for the flux
    normalisation: given by whoever made *field*, else the convex hull in the
    element's surface coordinates (waves.py:642-670)."""
    area = getattr(field, 'area', None)
    if area is not None and area > 0:
        return area
    if hasattr(oe, 'rotationSequence'):
        along = field.y                      # an optical element: (x, y)
    elif hasattr(oe, 'propagate') or hasattr(oe, 'prepare_wave') or \
            hasattr(oe, 'shine'):
        along = field.z                      # aperture / screen / source: (x, z)
    else:
        raise ValueError('Unknown diffracting element!')
    area = convex_hull_area(field.x[lit], along[lit])
    if hasattr(field, 'areaFraction'):
        area *= field.areaFraction
    return area


def _surface_normals(oe, field):
    """Normal of the diffracting surface at every sample and the cosine between
    it and the incoming direction (waves.py:674-689)."""
    if not hasattr(oe, 'rotationSequence'):
        normal = [0, 1, 0]
        return normal, field.a*normal[0] + field.b*normal[1] + field.c*normal[2]
    normal_at = oe.local_n2 if hasattr(oe, 'cryst2pitch') else oe.local_n
    if oe.isParametric:
        sp, phi, _ = oe.xyz_to_param(field.x, field.y, field.z)
        normal = normal_at(sp, phi)
    else:
        normal = normal_at(field.x, field.y)[-3:]
    cosine = (field.a*np.asarray([normal[-3]]) + field.b*np.asarray([normal[-2]]) +
              field.c*np.asarray([normal[-1]])).flatten()
    return normal, cosine


def _fields_and_directions(oe, wave, energy):
    """From the accumulated integrals: amplitudes, coherency matrix and the
    propagation direction (the direction integrals share one arbitrary phase,
    removed with the dominant component; waves.py:707-733)."""
    wave.E[:] = energy
    wave.Es[:] = wave.EsAcc
    wave.Ep[:] = wave.EpAcc
    wave.Jss[:] = (wave.Es * np.conj(wave.Es)).real
    wave.Jpp[:] = (wave.Ep * np.conj(wave.Ep)).real
    wave.Jsp[:] = wave.Es * np.conj(wave.Ep)
    carrier = wave.bEacc
    if hasattr(oe, 'rotationSequence') and abs(wave.cEacc[0]) > abs(wave.bEacc[0]):
        carrier = wave.cEacc
    unphase = np.exp(-1j * np.angle(carrier))
    wave.a[:] = (wave.aEacc * unphase).real
    wave.b[:] = (wave.bEacc * unphase).real
    wave.c[:] = (wave.cEacc * unphase).real
    length = (wave.a**2 + wave.b**2 + wave.c**2)**0.5
    length[length == 0] = 1.
    wave.a /= length
    wave.b /= length
    wave.c /= length


def _scale_to_flux(wave, area):
    """Monte-Carlo weight of the integral: receiving cell x illuminated area x
    incoming flux over (samples x obliquity-weighted flux x repeats),
    waves.py:735-749."""
    scale = wave.dS * area * wave.beamReflSumJ
    denom = wave.beamReflRays * wave.beamReflSumJnl * wave.diffract_repeats
    scale = scale / denom if denom > 0 else 0
    wave.Jss *= scale
    wave.Jpp *= scale
    wave.Jsp *= scale
    wave.Es *= scale**0.5
    wave.Ep *= scale**0.5


def _as_global_beam(oe, wave):
    """Copy of *wave* positioned at the receiving points in the global frame
    (waves.py:756-770)."""
    glo = rs.Beam(copyFrom=wave)
    glo.parentId = oe.uuid
    glo.x[:] = wave.xDiffr
    glo.y[:] = wave.yDiffr
    glo.z[:] = wave.zDiffr
    if hasattr(oe, 'local_to_global'):
        if hasattr(oe, 'expose'):            # a screen transforms bare arrays
            glo.x[:], glo.y[:], glo.z[:] = oe.local_to_global(glo.x, glo.y, glo.z)
        else:
            oe.local_to_global(glo)
    return glo


def _into_receiver_frame(wave, glo):
    """The receiving samples live on an element (``wave.toOE``): directions,
    coherency matrix and amplitudes go from the global frame into its local
    s/p frame, and the flux is projected on its surface (waves.py:773-824)."""
    receiver = wave.toOE
    wave.a[:], wave.b[:], wave.c[:] = glo.a, glo.b, glo.c
    wave.Jss[:], wave.Jpp[:], wave.Jsp[:] = glo.Jss, glo.Jpp, glo.Jsp
    wave.Es[:], wave.Ep[:] = glo.Es, glo.Ep
    bl = receiver.bl
    wave.a[:], wave.b[:] = raycing.rotate_z(wave.a, wave.b, bl.cosAzimuth,
                                            bl.sinAzimuth)
    if not hasattr(receiver, 'rotationSequence'):
        return
    if receiver.isParametric:
        sp, phi, _ = receiver.xyz_to_param(wave.x, wave.y, wave.z)
        normal = list(receiver.local_n(sp, phi))
    else:
        normal = list(receiver.local_n(wave.x, wave.y))
    turn = receiver.roll + receiver.positionRoll + np.arctan2(normal[-3], normal[-1])
    wave.Jss[:], wave.Jpp[:], wave.Jsp[:] = \
        rs.rotate_coherency_matrix(wave, slice(None), -turn)
    wave.Es[:], wave.Ep[:] = raycing.rotate_y(wave.Es, wave.Ep, np.cos(turn),
                                              -np.sin(turn))
    raycing.rotate_xyz(wave.a, wave.b, wave.c,
                       rotationSequence=receiver.rotationSequence,
                       pitch=-receiver.pitch,
                       roll=-receiver.roll-receiver.positionRoll, yaw=-receiver.yaw)
    if receiver.extraPitch or receiver.extraRoll or receiver.extraYaw:
        raycing.rotate_xyz(wave.a, wave.b, wave.c,
                           rotationSequence=receiver.extraRotationSequence,
                           pitch=-receiver.extraPitch, roll=-receiver.extraRoll,
                           yaw=-receiver.extraYaw)
    obliquity = np.abs(-wave.a*normal[-3] - wave.b*normal[-2] - wave.c*normal[-1])
    for beam in (wave, glo):
        beam.Jss *= obliquity
        beam.Jpp *= obliquity
        beam.Jsp *= obliquity
        beam.Es *= obliquity**0.5
        beam.Ep *= obliquity**0.5


def diffract(oeLocal, wave, targetOpenCL=raycing.targetOpenCL,
             precisionOpenCL=raycing.precisionOpenCL):
    """Field diffracted from the samples *oeLocal* on ``wave.fromOE`` onto the
    points of *wave*, by the Fresnel-Kirchhoff integral on the GPU. *wave* is
    updated (it accumulates over repeated calls); returns the same field as a
    beam in the global frame. Interface of the reference's ``waves.diffract``
    (waves.py:606-831)."""
    oe = wave.fromOE
    t0 = time.time()
    lit = oeLocal.state == 1
    nlit = lit.sum()
    if nlit < 1e2:
        print("Not enough good rays at {0}: {1} of {2}".format(
            oe.name, nlit, len(oeLocal.x)))
        return rs.Beam(nlit)
    if len(wave.xDiffr) == 0:
        print("No wave samples on {0}".format(oe.name))
        return rs.Beam(nlit)
    oeLocal.area = _illuminated_area(oe, oeLocal, lit)
    normal, cosine = _surface_normals(oe, oeLocal)
    flux = oeLocal.Jss[lit] + oeLocal.Jpp[lit]
    wave.diffract_repeats += 1
    wave.beamReflRays += nlit
    wave.beamReflSumJ += flux.sum()
    wave.beamReflSumJnl += abs((flux * cosine[lit]).sum())
    integrals = _kirchhoff_on_gpu(oeLocal, normal, cosine, wave, lit)
    for name, part in zip(_ACCUMULATORS, integrals):
        getattr(wave, name).__iadd__(part)
    _fields_and_directions(oe, wave, oeLocal.E[0])
    _scale_to_flux(wave, oeLocal.area)
    if hasattr(oeLocal, 'accepted'):         # source bookkeeping for absolute flux
        wave.accepted = oeLocal.accepted
        wave.acceptedE = oeLocal.acceptedE
        wave.seeded = oeLocal.seeded
        wave.seededI = oeLocal.seededI * len(wave.x) / len(oeLocal.x)
    glo = _as_global_beam(oe, wave)
    if hasattr(wave, 'toOE'):
        _into_receiver_frame(wave, glo)
    if _DEBUG > 10:
        print("diffract on {0} completed in {1:.4f} s".format(
            oe.name, time.time()-t0))
    glo.createdByDiffract = True
    return glo
