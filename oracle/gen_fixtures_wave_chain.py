"""TEST INFRASTRUCTURE ONLY — golden vector G8 by RUNNING THE REFERENCE: a
sequential wave-propagation chain

    field on a rectangular slit --diffract--> ToroidMirror(Pt).propagate_wave
    (= prepare_wave: random samples on the mirror; diffract; reflect with
    noIntersectionSearch) --diffract--> 20x20 screen

Stored: the slit field, the np.random seed, the mirror-local wave after
propagate_wave and the final screen wave. Run: python -m oracle.gen_fixtures_wave_chain
"""
import os

import numpy as np

from . import _refenv
from .consts import CHBAR

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')
F = ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp', 'Es', 'Ep',
     'state')


def slit_field(wslit, E0=9000., R0=20000.):
    k = E0 / CHBAR * 1e7
    rho2 = wslit.x**2 + wslit.z**2
    wslit.Es[:] = np.exp(-rho2 / 0.3**2) * np.exp(1j * k * rho2 / (2 * R0))
    wslit.Ep[:] = 0.25j * wslit.Es
    wslit.Jss[:] = np.abs(wslit.Es)**2
    wslit.Jpp[:] = np.abs(wslit.Ep)**2
    wslit.Jsp[:] = wslit.Es * np.conj(wslit.Ep)
    wslit.E[:] = E0
    wslit.a[:] = wslit.x / R0
    wslit.c[:] = wslit.z / R0
    wslit.b[:] = np.sqrt(1 - wslit.a**2 - wslit.c**2)


def build(raycing, ra, roe, rm, rsc, rs):
    bl = raycing.BeamLine()
    bl.src = rs.GeometricSource(bl, 'src', nrays=10)
    bl.slit = ra.RectangularAperture(bl, 'slit', [0, 20000., 0],
                                     ('left', 'right', 'bottom', 'top'),
                                     [-0.4, 0.4, -0.3, 0.3])
    p, q, pitch = 22000., 8000., 4e-3
    bl.m1 = roe.ToroidMirror(bl, 'm1', center=[0, p, 0], pitch=pitch, R=(p, q),
                             r=(p, q), material=rm.Material('Pt', rho=21.45),
                             limPhysX=[-1.5, 1.5], limPhysY=[-120, 120])
    bl.scr = rsc.Screen(bl, 'scr', [0, p + q, q * np.tan(2 * pitch)])
    return bl


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    import xrt.backends.raycing.screens as rsc
    import xrt.backends.raycing.apertures as ra
    import xrt.backends.raycing.waves as rw
    bl = build(raycing, ra, roe, rm, rsc, rs)
    np.random.seed(21)
    wslit = bl.slit.prepare_wave(bl.src, 1500)
    slit_field(wslit)
    out = {'s_' + f: np.array(getattr(wslit, f)) for f in F}
    out['s_area'] = np.float64(wslit.area)
    # The explicit three-call sequence the reference's own wave examples and
    # speed test use (tests/speed/3_Softi_CXIw2D_speed.py:369-407). NOT
    # OE.propagate_wave: that method first passes `wave` through
    # prevOE.local_to_global(wave, returnBeam=True) for auto-alignment
    # (oes/reflect.py:434-438), which transforms `wave` IN PLACE, so the
    # integral that follows sees the samples in the wrong frame.
    np.random.seed(22)
    wm = bl.m1.prepare_wave(bl.slit, 1200)
    beamToM1 = rw.diffract(wslit, wm)
    glo, lo = bl.m1.reflect(beamToM1, noIntersectionSearch=True)
    lo.parentId = bl.m1.uuid
    out.update({'m_' + f: np.array(getattr(lo, f)) for f in F})
    out.update({'mg_' + f: np.array(getattr(glo, f)) for f in F})
    xm = np.linspace(-0.15, 0.15, 20)
    zm = np.linspace(-0.05, 0.05, 20)
    wscr = bl.scr.prepare_wave(bl.m1, xm, zm)
    lo.area = float(lo.area) if hasattr(lo, 'area') else 0.
    rw.diffract(lo, wscr)
    out.update({'w_' + f: np.array(getattr(wscr, f)) for f in F})
    out.update(xmesh=xm, zmesh=zm, m_area=np.float64(lo.area))
    path = os.path.join(OUT, 'g8_wave_chain.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB; mirror samples',
          len(lo.x), 'good', (lo.state == 1).sum(), 'screen max J',
          (wscr.Jss + wscr.Jpp).max())


if __name__ == '__main__':
    main()
