"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g7_grid_star.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only): GridAperture, GridBeamStop
and SiemensStar (apertures.py:1324-1528) on the input beam of g7_stops_round -- outlines of
many cells separated by NaN rows (matplotlib treats each as its own polygon), straight and
bent (vortex) star spokes. The outlines the reference builds are stored, the product classes
must build the same ones. While generating, oracle/elements_np.aperture_propagate (polygon
branch with sub-polygons) is asserted against the reference.

Run:  python -m oracle.gen_fixtures_grid
"""
import numpy as np

from . import _refenv
from . import elements_np as en
from . import gen_fixtures_p1 as g1
from . import reflect_np as rn

CASES = (
    ('grid', 'GridAperture', dict(dx=0.1, dz=0.08, px=0.25, pz=0.2, nx=2, nz=1)),
    ('grid_stop', 'GridBeamStop', dict(dx=0.2, dz=0.05, px=0.3, pz=0.12, nx=1, nz=2)),
    ('star', 'SiemensStar', dict(nSpokes=7, r=0.5, phi0=0.1)),
    ('star_vortex', 'SiemensStar', dict(nSpokes=5, r=0, rx=0.6, rz=0.4, vortex=0.7,
                                        vortexNradial=5)),
)


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.apertures as ra
    g = np.load(g1.os.path.join(g1.OUT, 'g7_stops_round.npz'))
    ob_in = rn.Beam.from_dict(g, 'in_')
    out = {}
    for tag, cls, kw in CASES:
        bl = raycing.BeamLine(azimuth=float(g['azimuth']))
        ap = getattr(ra, cls)(bl, tag, center=[np.sin(-0.02)*8000., np.cos(-0.02)*8000., 0.05],
                              **kw)
        beam = rs.Beam(nrays=len(ob_in.x), withAmplitudes=True)
        for f in ob_in.fields():
            getattr(beam, f)[:] = getattr(ob_in, f)
        glo, lo = ap.propagate(beam, needNewGlobal=True)
        ob = ob_in.copy()
        mglo, mlo = en.aperture_propagate(
            ob, ap.xyz, ap.center, {}, ap.lostNum, (bl.sinAzimuth, bl.cosAzimuth),
            isBeamStop=ap.isBeamStop, needNewGlobal=True, vertices=ap.vertices)
        g1.assert_beams(tag + ':lo', mlo, lo)
        g1.assert_beams(tag + ':glo', mglo, glo)
        assert np.array_equal(ob.state, beam.state)
        st, cnt = np.unique(lo.state, return_counts=True)
        print(tag, 'states', dict(zip(st.tolist(), cnt.tolist())), len(ap.vertices), 'vertices')
        full = g1.beam_dict(tag + '_lo_', lo)
        out.update({k: v for k, v in full.items()
                    if k.rsplit('_', 1)[1] in ('state', 'x', 'z', 'path')})
        out[tag + '_in_state_after'] = np.array(beam.state)
        out[tag + '_vertices'] = np.array(ap.vertices, dtype=float)
        out[tag + '_center'] = np.array(ap.center, dtype=float)
        out[tag + '_lostNum'] = np.array(ap.lostNum)
    g1.save('g7_grid_star', **out)


if __name__ == '__main__':
    main()
