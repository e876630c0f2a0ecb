"""Workload to run under rocprofv3 (kernel trace or one --pmc counter at a time):
  * Screen.expose on 1e7 rays   - pure streaming, exactly 100 B read + 100 B
                                  written per ray with the same 8 B/lane SoA
                                  accesses as the reflect kernels: the
                                  CALIBRATION kernel for FETCH_SIZE / WRITE_SIZE
  * OE.reflect cfg2, 1e7 rays   - P1
  * DCM.double_reflect cfg3     - P1, crystal path
  * Kirchhoff cfg4 (1 launch)   - P2
  * GeometricSource(rng='device').shine, 1e7 rays - the ray generator kernel

The cfg2 / cfg3 launches are the ones bench.py times: the FULL pass, both beams written (308 B /
416 B per ray) -- the beams-on-demand route (oes.fuseConsumers) is switched off for them as
bench.py's primary legs do. ``--nolocal`` runs ONLY the passes without their local beams
(200 B per ray: OE.reflect(needLocal=False), DCM.double_reflect with the local beams left out)
after the calibration kernel, for the ``*_nolocal`` entries of profiles/hbm_traffic.json.
"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from xrt_amd import hipcalls, workloads  # noqa: E402
import xrt_amd.backends.raycing as raycing  # noqa: E402
import xrt_amd.backends.raycing.oes as roe  # noqa: E402
import xrt_amd.backends.raycing.screens as rsc  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
reps = 3
oe = workloads.cfg2_toroid()
beam = workloads.synthetic_rays(n, 42)
if '--tight' in sys.argv:      # every ray hits (no divergent lost / over lanes in a wave)
    beam.z = beam.z * 0.2
    beam.c = beam.c * 0.2
for f in beam.array_fields():
    beam.dev(f)
bl0 = raycing.BeamLine()
scr = rsc.Screen(bl0, 'scr', [0, 30000., 0])
for _ in range(reps):
    scr.expose(beam).nrays          # (looked at: the screen's own launch)
if '--nolocal' not in sys.argv:
    # a screen and the mask behind it on a resident beam: one pass over the rays
    # (screen_expose_mark_kernel: 100 B read, 100 B + the marks written per ray)
    import xrt_amd.backends.raycing.apertures as ra  # noqa: E402
    import xrt_amd.backends.raycing.sources as rs  # noqa: E402
    mask = ra.RectangularAperture(bl0, 'mask', [0, 30500., 0], ('left', 'right', 'bottom', 'top'),
                                  [-4., 5., -0.5, 0.6])
    for _ in range(reps):
        rays = rs.Beam(copyFrom=beam)
        scr.expose(rays)
        mask.propagate(rays)
NOLOCAL = '--nolocal' in sys.argv
dcm = workloads.cfg3_dcm()
b3 = workloads.synthetic_rays(n, 43, sa=1e-4, E=(8995., 9005.))
if NOLOCAL:
    # the 200-B forms: what an element does whose global beam goes on to the next element
    for _ in range(reps):
        oe.reflect(beam, needLocal=False)
    for _ in range(reps):
        gb3 = dcm.double_reflect(b3)[0]
        gb3.dev('x')            # the global beam is looked at, the local beams are not
    torch.cuda.synchronize()
    print('done (nolocal)')
    sys.exit(0)
roe.fuseConsumers = False       # the timed shape of bench.py: every beam written, at once
for _ in range(reps):
    oe.reflect(beam)
# the device ray generator (round 4): 100 B written per ray, nothing read
bl_e2e, _, _ = workloads.e2e_beamline(n)
for _ in range(reps):
    bl_e2e.source.shine()
for _ in range(reps):
    dcm.double_reflect(b3)
torch.cuda.synchronize()
roe.fuseConsumers = True
if '--no-kirchhoff' not in sys.argv:
    h = workloads.kirchhoff_case(4)
    dev = torch.device('cuda', 0)
    up = lambda a, dt=np.float64: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)  # noqa: E731
    ns = h['ns']
    hipcalls.kirchhoff(up(h['px']), up(h['py']), up(h['pz']), up(h['sx']), up(h['sy']), up(h['sz']),
                       up(np.zeros(ns)), up(np.ones(ns)), up(np.zeros(ns)), up(h['nl']), up(h['k']),
                       up(h['Es'], np.complex128), up(h['Ep'], np.complex128))
# undulator field map (N3): 2^20 rays x 48 nodes, far field
from xrt_amd.backends.raycing.undulator import clenshaw_curtis  # noqa: E402
rng = np.random.RandomState(5)
nr = 1 << 20
xk, wk = clenshaw_curtis(24)
dstep = np.pi
dI = np.arange(-np.pi + 0.5 * dstep, np.pi, dstep)
tg = (dI[:, None] + 0.5 * dstep * xk).ravel()
ag = (dI[:, None] * 0 + wk).ravel()
dev = torch.device('cuda', 0)
upd = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
tabs = [upd(t) for t in (tg, ag, np.sin(tg), np.cos(tg), np.sin(tg), np.cos(tg))]
w, th, ps = (upd(rng.uniform(3900., 4250., nr)), upd(rng.uniform(-3e-5, 3e-5, nr)),
             upd(rng.uniform(-3e-5, 3e-5, nr)))
for _ in range(reps):
    hipcalls.undulator_imap(0, 0., 0.52, tabs, w, th, ps, 18.5, 108, 5870.853297866972,
                            0.5, dstep, True)
torch.cuda.synchronize()
print('done')
