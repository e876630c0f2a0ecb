"""TEST INFRASTRUCTURE ONLY -- golden vectors by RUNNING THE REFERENCE (imported from
/root/reference, build container only) for the two wave-propagation cases SURVEY 8f row N4
still lacked:

  g8_fzp_wave.npz        a zone plate in WAVE mode (NormalFZP, oes/gratings.py:10-137, used
                         through prepare_wave / diffract / reflect(noIntersectionSearch)):
                         field on a slit --diffract--> samples on the transparent zones of the
                         plate --diffract--> a line of screen points through its first-order
                         focus
  g8_source_mirror.npz   OE.propagate_wave straight from a source (oes/reflect.py:405-449, the
                         'source' branch: prepare_wave on the mirror, Undulator.shine(wave=...)
                         onto those samples, reflect WITH the intersection search)

Stored: seeds, inputs, the waves after every step. Run: python -m oracle.gen_fixtures_n4_waves
"""
import json
import os

import numpy as np

from . import _refenv
from .consts import CHBAR

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')
F = ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp', 'Es', 'Ep', 'state')

FZP = dict(f=50., E=700., N=40)
FZP_Y, SLIT_Y = 1000., 400.
UND = dict(nrays=1500, period=29., n=172, eE=6.08, eI=0.1, eEpsilonX=0., eEpsilonZ=0.,
           betaX=1.2, betaZ=3.95, filamentBeam=True, uniformRayDensity=True,
           xPrimeMax=0.02, zPrimeMax=0.02, targetE=[7900., 3], eMin=7899.5, eMax=7900.5)
MIRROR = dict(center=[0., 25000., 0.], pitch=4e-3, limPhysX=[-0.4, 0.4], limPhysY=[-100., 100.])


def slit_field(w, E0, R0):
    k = E0 / CHBAR * 1e7
    rho2 = w.x**2 + w.z**2
    w.Es[:] = np.exp(1j * k * rho2 / (2 * R0))
    w.Ep[:] = 0.
    w.Jss[:] = np.abs(w.Es)**2
    w.Jpp[:] = 0.
    w.Jsp[:] = 0.
    w.E[:] = E0
    w.a[:] = w.x / R0
    w.c[:] = w.z / R0
    w.b[:] = np.sqrt(1 - w.a**2 - w.c**2)


def fzp_case(raycing, rs, ra, roe, rm, rsc, rw):
    bl = raycing.BeamLine()
    bl.src = rs.GeometricSource(bl, 'src', nrays=10)
    fzp = roe.NormalFZP(bl, 'fzp', center=[0, FZP_Y, 0], pitch=np.pi/2,
                        material=rm.Material('Au', rho=19.3, kind='FZP'), order=1, **FZP)
    half = float(fzp.rn[-1])
    slit = ra.RectangularAperture(bl, 'slit', [0, SLIT_Y, 0], ('left', 'right', 'bottom', 'top'),
                                  [-1.2 * half, 1.2 * half, -1.2 * half, 1.2 * half])
    np.random.seed(41)
    wslit = slit.prepare_wave(bl.src, 1500)
    slit_field(wslit, FZP['E'], SLIT_Y)
    out = {'s_' + f: np.array(getattr(wslit, f)) for f in F}
    out['s_area'] = np.float64(wslit.area)
    np.random.seed(42)
    wz = fzp.prepare_wave(slit, 3000)
    to_fzp = rw.diffract(wslit, wz)
    glo, lo = fzp.reflect(to_fzp, noIntersectionSearch=True)
    lo.parentId = fzp.uuid
    out.update({'z_' + f: np.array(getattr(lo, f)) for f in F})
    out['z_area'] = np.float64(lo.area) if hasattr(lo, 'area') else np.float64(0.)
    # the image of the point source the slit field comes from (the origin), through the plate
    p = FZP_Y
    q = 1. / (1. / FZP['f'] - 1. / p)
    scr = rsc.Screen(bl, 'scr', [0, FZP_Y + q, 0])
    xm = np.linspace(-3e-3, 3e-3, 33)
    zm = np.array([0.])
    wscr = scr.prepare_wave(fzp, xm, zm)
    rw.diffract(lo, wscr)
    out.update({'w_' + f: np.array(getattr(wscr, f)) for f in F})
    out.update(xmesh=xm, zmesh=zm, q=np.float64(q), rn=np.array(fzp.rn),
               fzp=json.dumps(FZP))
    path = os.path.join(OUT, 'g8_fzp_wave.npz')
    np.savez_compressed(path, **out)
    J = wscr.Jss + wscr.Jpp
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB; samples on open zones',
          len(lo.x), 'state 1:', int((lo.state == 1).sum()), '; focus profile peak at',
          xm[np.argmax(J)], 'peak / edge', J.max() / J[0])


def source_case(raycing, rs, roe, rm):
    bl = raycing.BeamLine()
    src = rs.Undulator(bl, 'und', targetOpenCL=None, **UND)
    m1 = roe.ToroidMirror(bl, 'm1', R=1e7, r=60., material=rm.Material('Pt', rho=21.45),
                          **MIRROR)
    trigger = rs.Beam(nrays=8)          # stands for the source's beam: propagate_wave takes
    trigger.parentId = src.uuid         # its size (unless nrays is given) and its parent
    np.random.seed(43)
    glo, lo = m1.propagate_wave(wave=trigger, nrays=1200)
    out = {'m_' + f: np.array(getattr(lo, f)) for f in F}
    out.update({'mg_' + f: np.array(getattr(glo, f)) for f in F})
    out.update(und=json.dumps(UND), mirror=json.dumps(MIRROR),
               limits=np.array([src.E_min, src.E_max, src.Theta_min, src.Theta_max,
                                src.Psi_min, src.Psi_max]),
               quadm=np.int64(src.quadm), gIntervals=np.int64(src.gIntervals))
    path = os.path.join(OUT, 'g8_source_mirror.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB; mirror samples', len(lo.x),
          'state 1:', int((lo.state == 1).sum()), 'max J', float((lo.Jss + lo.Jpp).max()))


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.apertures as ra
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    import xrt.backends.raycing.screens as rsc
    import xrt.backends.raycing.waves as rw
    raycing._VERBOSITY_ = 0
    fzp_case(raycing, rs, ra, roe, rm, rsc, rw)
    source_case(raycing, rs, roe, rm)


if __name__ == '__main__':
    main()
