"""TEST INFRASTRUCTURE ONLY — regenerates tests/golden/g4_*.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only).

G4: inputs and outputs of waves.diffract for
  a) undulator -> rectangular slit (2000 samples) -> 32x32 screen,
  b) same, 4000 samples -> 48x48 screen,
  c) toroid mirror (local normals vary per sample) -> 24x24 screen.
Stored: the diffracting-surface beam, the receiving points, the raw integrals
of _diffraction_integral_conv and the post-`diffract` wave/global beam.
While generating, the numpy restatement in oracle/kirchhoff_np.py is checked
against the reference's own function on the same inputs.

Run:  python -m oracle.gen_fixtures_p2
"""
import os
import numpy as np
from . import _refenv
from . import kirchhoff_np as kn

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')
BEAM_FIELDS = ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp',
               'Es', 'Ep', 'state')


def _beam_dict(prefix, beam):
    return {prefix + f: np.array(getattr(beam, f)) for f in BEAM_FIELDS}


def _run_diffract(rw, oeLocal, wave, tag, extra):
    """Run reference diffract while capturing the raw kernel output."""
    captured = {}
    orig = rw._diffraction_integral_conv

    def spy(oeL, n, nl, w, good):
        res = orig(oeL, n, nl, w, good)
        n3 = [np.broadcast_to(np.asarray(c, dtype=float), oeL.x.shape).copy()
              for c in n]
        captured.update(n=np.array(n3), nl=np.array(nl), good=np.array(good),
                        raw=np.array(res))
        # check the restatement on the very same inputs
        mine = kn.kirchhoff_conv(
            w.xDiffr, w.yDiffr, w.zDiffr, oeL.x[good], oeL.y[good],
            oeL.z[good], [c[good] for c in n3], nl[good], oeL.E[good],
            oeL.Es[good], oeL.Ep[good])
        for m, r in zip(mine, res):
            scale = np.abs(r).max()
            assert np.abs(m - r).max() <= 1e-13 * scale, \
                (tag, np.abs(m - r).max() / scale)
        return res

    rw._diffraction_integral_conv = spy
    try:
        inp = _beam_dict('s_', oeLocal)
        glo = rw.diffract(oeLocal, wave)
    finally:
        rw._diffraction_integral_conv = orig
    out = dict(inp)
    out.update(
        s_area=np.float64(oeLocal.area), w_dS=np.float64(wave.dS),
        px=wave.xDiffr, py=wave.yDiffr, pz=wave.zDiffr,
        n=captured['n'], nl=captured['nl'], raw=captured['raw'],
        w_beamReflRays=np.int64(wave.beamReflRays),
        w_beamReflSumJ=np.float64(wave.beamReflSumJ),
        w_beamReflSumJnl=np.float64(wave.beamReflSumJnl),
        w_diffract_repeats=np.int64(wave.diffract_repeats))
    out.update(_beam_dict('w_', wave))
    out.update(_beam_dict('g_', glo))
    out.update(extra)
    path = os.path.join(OUT, 'g4_%s.npz' % tag)
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.screens as rsc
    import xrt.backends.raycing.apertures as ra
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    import xrt.backends.raycing.waves as rw
    os.makedirs(OUT, exist_ok=True)

    R0 = 44000.
    E0 = 7900.
    slitD = 0.2
    for tag, ns, npx, seed in (('slit_2000x32', 2000, 32, 7),
                               ('slit_4000x48', 4000, 48, 8)):
        np.random.seed(seed)
        bl = raycing.BeamLine()
        bl.source = rs.Undulator(
            bl, nrays=ns, period=29., n=172, eE=6.08, eI=0.1, eEpsilonX=0.,
            eEpsilonZ=0., betaX=1.2, betaZ=3.95, filamentBeam=True,
            uniformRayDensity=True, xPrimeMax=(slitD/R0)*2e3,
            zPrimeMax=(slitD/R0)*2e3, targetE=[E0, 3], eMin=E0-0.5,
            eMax=E0+0.5, targetOpenCL=None)
        bl.slit = ra.RectangularAperture(
            bl, 'slit', [0, R0, 0], ('left', 'right', 'bottom', 'top'),
            [-slitD/2, slitD/2, -slitD/2, slitD/2])
        bl.scr = rsc.Screen(bl, 'scr', [0, R0 + 10000., 0])
        xm = np.linspace(-0.5, 0.5, npx)
        zm = np.linspace(-0.5, 0.5, npx)
        wscr = bl.scr.prepare_wave(bl.slit, xm, zm)
        wslit = bl.slit.prepare_wave(bl.source, ns)
        bl.source.shine(fixedEnergy=E0, wave=wslit)
        _run_diffract(rw, wslit, wscr, tag,
                      dict(kind='aperture', xmesh=xm, zmesh=zm,
                           slit_center=np.array([0, R0, 0.]),
                           screen_center=np.array([0, R0 + 10000., 0.])))

    # c) diffraction from a curved mirror: per-sample normals, OE branch of
    # the phase strip (waves.py:719-722) and of local_to_global (waves.py:763-771)
    np.random.seed(9)
    bl = raycing.BeamLine()
    ns = 3000
    src = rs.GeometricSource(
        bl, 'src', nrays=ns, dx=0.05, dz=0.02, dxprime=2e-5, dzprime=5e-6,
        distE='lines', energies=(9000.,), polarization='h')
    pitch = 4e-3
    p, q = 20000., 10000.
    mir = roe.ToroidMirror(
        bl, 'tm', center=[0, p, 0], pitch=pitch, R=(p, q), r=(p, q),
        material=rm.Material('Pt', rho=21.45), limPhysX=[-10, 10],
        limPhysY=[-300, 300])
    scr = rsc.Screen(bl, 'scr', [0, p + q, q*np.tan(2*pitch)])
    beam = src.shine(withAmplitudes=True)
    gb, lb = mir.reflect(beam)
    assert (lb.state == 1).all()
    lb.parentId = mir.uuid
    xm = np.linspace(-0.05, 0.05, 24)
    zm = np.linspace(-0.02, 0.02, 24)
    wscr = scr.prepare_wave(mir, xm, zm)
    _run_diffract(rw, lb, wscr, 'toroid_3000x24',
                  dict(kind='oe', xmesh=xm, zmesh=zm,
                       mirror=np.array([p, q, pitch, mir.R, mir.r]),
                       screen_center=np.array(scr.center, dtype=float)))


if __name__ == '__main__':
    main()
