"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g17_balder_chain.npz by RUNNING THE
REFERENCE's own example beamline (build container only):
examples/withRaycing/02_Balder_BL/BalderBL.py -- build_beamline(), align_beamline(energy =
9 keV) and run_process() exactly as the example's tracing script calls them (SURVEY 8b names
this run_process as the caller of the ray path): wiggler -> front-end mask -> diamond filter
(Plate.double_refract) -> bent collimating mirror -> DCM Si(111) -> slits -> toroidal
focusing mirror (turned upside down) -> slits -> sample screen.

Stored: the aligned beamline's numbers (they are the example's parameters after its own
alignment arithmetic -- data, not code) and, for the beams of run_process, state / position /
direction / energy / flux per ray. numpy is seeded right before run_process.

Run:  python -m oracle.gen_fixtures_balder
"""
import os
import sys

import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1

NRAYS, SEED = 1500, 1717
BEAMS = ('beamSource', 'beamFSM0', 'beamFilter1global', 'beamFilter1local2A', 'beamVCMglobal',
         'beamVCMlocal', 'beamFSMVCM', 'beamDCMglobal', 'beamDCMlocal1', 'beamDCMlocal2',
         'beamSlitAfterDCMlocal', 'beamVFMglobal', 'beamVFMlocal', 'beamSlitEHLocal',
         'beamFSMSample')
FIELDS = ('state', 'x', 'y', 'z', 'a', 'c', 'E', 'Jss', 'Jpp')


def main():
    _refenv.activate()
    here = os.getcwd()
    example = '/root/reference/examples/withRaycing/02_Balder_BL'
    sys.path.insert(0, example)
    os.chdir(example)
    try:
        import BalderBL
    finally:
        os.chdir(here)
    bl = BalderBL.build_beamline(nrays=NRAYS, eMinRays=8990., eMaxRays=9010.)
    BalderBL.align_beamline(bl, energy=9000.)
    np.random.seed(SEED)
    beams = BalderBL.run_process(bl)
    out = {}
    for name in BEAMS:
        b = beams[name]
        for f in FIELDS:
            out['%s_%s' % (name, f)] = np.array(getattr(b, f))
        st, cnt = np.unique(b.state, return_counts=True)
        print(name, dict(zip(st.tolist(), cnt.tolist())),
              'flux %.4g' % (b.Jss + b.Jpp)[b.state == 1].sum())
    src = bl.sources[0]
    par = dict(
        vcm_pitch=bl.vcm.pitch, vcm_R=bl.vcm.R, dcm_z=bl.dcm.center[2], dcm_bragg=bl.dcm.bragg,
        dcm_perp=bl.dcm.cryst2perpTransl, vfm_pitch=bl.vfm.pitch, vfm_R=bl.vfm.R,
        vfm_z=bl.vfm.center[2], mask=[bl.feFixedMask.blades[k] for k in
                                       ('left', 'right', 'bottom', 'top')],
        slitDCM=[bl.slitAfterDCM.blades[k] for k in ('left', 'right', 'bottom', 'top')],
        slitVFM=[bl.slitAfterVFM.blades[k] for k in ('left', 'right', 'bottom', 'top')],
        slitEH=[bl.slitEH.blades[k] for k in ('left', 'right', 'bottom', 'top')],
        seed=SEED, nrays=NRAYS, src_eMin=src.eMin, src_eMax=src.eMax)
    out.update({'par_' + k: np.array(v, dtype=float) for k, v in par.items()})
    g1.save('g17_balder_chain', **out)


if __name__ == '__main__':
    main()
