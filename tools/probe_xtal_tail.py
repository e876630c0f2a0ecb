"""A single flat Si(111) crystal with a round aperture and a screen in the tail of its pass
(reflect_fused_xtal_scr) against the three launches, 1e7 rays (cfg3's beam on one crystal).
    python tools/probe_xtal_tail.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from xrt_amd import workloads
import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.apertures as ra
import xrt_amd.backends.raycing.materials as rm
import xrt_amd.backends.raycing.oes as roe
import xrt_amd.backends.raycing.screens as rsc

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
bl = raycing.BeamLine()
si = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
thB = float(np.ravel(si.get_Bragg_angle(9000.) - si.get_dtheta(9000.))[0])
xt = roe.OE(bl, 'xtal', center=[0, 20000., 0], pitch=thB, material=si, limPhysX=[-10, 10],
            limPhysY=[-50, 50])
scr = rsc.Screen(bl, 'after', center=[0, 21000., 1000. * np.tan(2 * thB)])
pipe = ra.RoundAperture(bl, 'pipe', [0, 20500., 500. * np.tan(2 * thB)], r=2.)
beam = workloads.synthetic_rays(n, 3, sa=1e-4, E=(8995., 9005.))
for f in beam.array_fields():
    beam.dev(f)


def chain(aperture, fuse=True):
    roe.fuseConsumers = fuse
    g = xt.reflect(beam)[0]
    if aperture:
        pipe.propagate(g)
    img = scr.expose(g)
    img.nrays
    roe.fuseConsumers = True
    return g, img


for name, args in (('screen in the tail', (False,)), ('aperture + screen in the tail', (True,)),
                   ('three launches', (True, False))):
    g, img = chain(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g, img = chain(*args)
    torch.cuda.synchronize()
    print('crystal, %-30s %.3f ms   global beam written: %s   arrived %.4f' % (
        name, (time.perf_counter() - t0) * 100, g.__dict__.get('_filled', True),
        float((img.state == 1).sum()) / n))
