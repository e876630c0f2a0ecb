"""Times OE.reflect through multilayer materials on 1e7 rays (cfg2-like beam): the
Parratt recursion inside the fused pass. python tools/probe_multilayer.py [nrays]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import xrt_amd.backends.raycing as raycing            # noqa: E402
import xrt_amd.backends.raycing.materials as rm       # noqa: E402
import xrt_amd.backends.raycing.oes as roe            # noqa: E402
from xrt_amd import workloads                         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
si, w = rm.Material('Si', rho=2.33), rm.Material('W', rho=19.3)
stacks = {
    'W/Si x40 periodic': rm.Multilayer(w, 12., si, 18., 40, si),
    'W/Si x200 periodic': rm.Multilayer(w, 12., si, 18., 200, si),
    'W/Si x40 depth-graded': rm.Multilayer(w, 12., si, 18., 40, si, tThicknessLow=9.,
                                           bThicknessLow=14.),
    'Rh coating': rm.Coated(coating=rm.Material('Rh', rho=12.41), cThickness=300.,
                            substrate=si, surfaceRoughness=3.),
    'Rh bulk mirror': rm.Material('Rh', rho=12.41, kind='mirror'),
}
beam = workloads.synthetic_rays(n, seed=42)
for label, m in stacks.items():
    bl = raycing.BeamLine()
    pitch = 4e-3 if 'Rh' in label else float(m.get_Bragg_angle(9000.))
    oe = roe.OE(bl, 'ml', center=[0, 20000., 0], pitch=pitch, material=m,
                limPhysX=[-10, 10], limPhysY=[-300, 300])
    for _ in range(2):
        gb, lb = oe.reflect(beam)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        gb, lb = oe.reflect(beam)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    good = int((lb.state_dev() == 1).sum()) if hasattr(lb, 'state_dev') else -1
    layers = 2 * getattr(m, 'nPairs', 0)
    print('%-24s %8.3f ms / %d rays  %s' % (
        label, ms, n, '%.1f G ray-layers/s' % (n * layers / ms / 1e6) if layers else ''))
