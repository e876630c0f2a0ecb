"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g1_hemispheric_screen.npz by RUNNING
THE REFERENCE (imported from /root/reference, build container only): HemisphericScreen.expose
(screens.py:422-559) with the default axes on a beamline with an azimuth ('auto'), with given
axes and angular offsets, with onlyPositivePath; rays that start outside the sphere and miss
it (lost), rays inside it, field amplitudes carried along. While generating,
oracle/elements_np.hemispheric_expose is asserted against the reference.

Run:  python -m oracle.gen_fixtures_hemisphere
"""
import numpy as np

from . import _refenv
from . import elements_np as en
from . import gen_fixtures_p1 as g1


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.screens as rsc
    n = 1024
    beam = g1.make_rays(rs, n, 150, sx=2., sz=2., sa=0.3, sc=0.3, E=(8000., 9000.),
                        amplitudes=True, pol='mixed')
    beam.state[3] = -5
    beam.state[4] = 2
    beam.x[5:9] = 300.            # outside the sphere, flying past it
    beam.a[5:9] = 0.
    beam.c[5:9] = 0.
    beam.b[5:9] = 1.
    beam.y[9] = 150.              # outside in front, flying away: negative path only
    wide = beam.a**2 + beam.c**2 > 0.9
    beam.a[wide] *= 0.5
    beam.c[wide] *= 0.5
    beam.b[:] = np.sqrt(1 - beam.a**2 - beam.c**2)
    beam.b[20:30] *= -1           # backwards
    out = g1.beam_dict('in_', beam)
    bl = raycing.BeamLine(azimuth=0.2)
    cases = (
        ('auto', dict(center=[1., 20., -0.5], R=100.), False),
        ('given', dict(center=[0., 10., 0.], R=80., x=(0, 1, 1), z=(1, 0, 0),
                       phiOffset=0.3, thetaOffset=-0.1), False),
        ('positive', dict(center=[0., -120., 0.], R=100.), True),
    )
    for tag, kw, positive in cases:
        scr = rsc.HemisphericScreen(bl, tag, **kw)
        lo = scr.expose(beam, onlyPositivePath=positive)
        mine = en.hemispheric_expose(g1.to_oracle_beam(beam), (scr.x, scr.y, scr.z),
                                     scr.center, scr.R, scr.lostNum, scr.phiOffset,
                                     scr.thetaOffset, positive)
        g1.assert_beams('hemi:' + tag, mine, lo)
        assert np.array_equal(mine.theta, lo.theta, equal_nan=True) and \
            np.array_equal(mine.phi, lo.phi, equal_nan=True)
        st, cnt = np.unique(lo.state, return_counts=True)
        print(tag, dict(zip(st.tolist(), cnt.tolist())))
        out.update(g1.beam_dict(tag + '_', lo))
        out.update({tag + '_theta': lo.theta, tag + '_phi': lo.phi,
                    tag + '_axes': np.array([scr.x, scr.y, scr.z], dtype=float),
                    tag + '_center': np.array(scr.center, dtype=float),
                    tag + '_R': np.array(scr.R), tag + '_lostNum': np.array(scr.lostNum),
                    tag + '_offsets': np.array([scr.phiOffset, scr.thetaOffset])})
        glo = scr.expose_global(beam)
        out[tag + '_global_xyz'] = np.array([glo.x, glo.y, glo.z])
    out['azimuth'] = np.array(0.2)
    g1.save('g1_hemispheric_screen', **out)


if __name__ == '__main__':
    main()
