"""TEST INFRASTRUCTURE ONLY — regenerates tests/golden/g11_softimax_chain.npz by
RUNNING THE REFERENCE (imported from /root/reference, build container only).

G11: the reference's published wave benchmark
(tests/speed/3_Softi_CXIw2D_speed.py: undulator -> FE slit -> M1 -> M2 ->
blazed grating -> M3 -> exit slit -> M4 -> M5 -> 3 focal screens, ten Kirchhoff
integrals) at 1000 samples per wave with the reference's numpy kernels. The
scene is described once, in xrt_amd/workloads.py:SoftiMAX, and instantiated
here on the reference's modules. Stored per stage: positions, state,
intensities and complex amplitudes of the local beam.

The reference at this revision cannot run its own script: ToroidMirror.local_z
(oes/__init__.py:398-401) indexes a numpy scalar when prepare_wave asks for the
height of the previous mirror's centre (reflect.py:356). The generator wraps
that method with np.atleast_1d on the two toroids; nothing else is touched.

Run:  python -m oracle.gen_fixtures_softi_chain
"""
import os
import types

import numpy as np

from . import _refenv

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')
NRAYS = 1000
SEED = 31
GNODES = 32
STAGE_FIELDS = ('x', 'y', 'z', 'a', 'b', 'c', 'state', 'Jss', 'Jpp', 'Es', 'Ep')


def reference_modules():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.apertures as ra
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    import xrt.backends.raycing.screens as rsc
    import xrt.backends.raycing.waves as rw
    raycing._VERBOSITY_ = 0
    return types.SimpleNamespace(raycing=raycing, rs=rs, ra=ra, roe=roe, rm=rm,
                                 rsc=rsc, rw=rw)


def main():
    from xrt_amd.workloads import SoftiMAX
    mods = reference_modules()
    np.random.seed(SEED)
    scene = SoftiMAX(mods, nrays=NRAYS,
                     source_kwargs=dict(targetOpenCL=None, gNodes=GNODES))
    for oe in (scene.bl.m1, scene.bl.m3):
        oe.local_z = (lambda f: (lambda x, y: f(np.atleast_1d(x),
                                                np.atleast_1d(y))))(oe.local_z)
    out = dict(nrays=np.int64(NRAYS), seed=np.int64(SEED), gNodes=np.int64(GNODES),
               screenCenters=np.array(scene.screenCenters),
               pg_areaFraction=np.float64(scene.bl.pg.areaFraction),
               Kxy=np.array([scene.bl.source.Kx, scene.bl.source.Ky]))
    stages = []

    def keep(name, beam):
        stages.append(name)
        for f in STAGE_FIELDS:
            out['%s_%s' % (name, f)] = np.array(getattr(beam, f))
        for k in ('area', 'dS', 'areaNormal'):
            if hasattr(beam, k):
                out['%s_%s' % (name, k)] = np.float64(getattr(beam, k))
        print(name, len(beam.x), 'flux %.6e' % (beam.Jss + beam.Jpp).sum())

    scene.run(keep)
    out['stages'] = np.array(stages)
    path = os.path.join(OUT, 'g11_softimax_chain.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
