"""Round-2 elements at ray counts the goldens do not reach (GPU box):
PYTHONPATH=. python tools/probe_elements_scale.py"""
import time

import numpy as np
import torch

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.apertures as ra
import xrt_amd.backends.raycing.materials as rm
import xrt_amd.backends.raycing.oes as roe
from xrt_amd import workloads


def timed(label, fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    beams = out if isinstance(out, tuple) else (out,)
    st = beams[-1].state
    print('%-28s %8.2f ms  %.2e rays/s  good %.3f' % (label, dt * 1e3, n / dt, (st == 1).mean()))


n = 4_000_000
bl = raycing.BeamLine()
be = rm.Material('Be', rho=1.848, kind='lens')
beam = workloads.synthetic_rays(n, 7, sa=1e-5, sc=1e-5)
lens = roe.DoubleParaboloidLens(bl, 'crl', center=[0, 20000., 0], material=be, t=0.05,
                                focus=0.25, zmax=0.4, nCRL=8, limPhysX=[-1, 1], limPhysY=[-1, 1])
timed('CRL x8 (16 surfaces)', lambda: lens.multiple_refract(beam), n)
fzp = roe.NormalFZP(bl, 'fzp', center=[0, 20000., 0], pitch=np.pi/2,
                    material=rm.Material('Au', rho=19.3, kind='FZP'), f=5., E=9000.,
                    N=2000, order=(1, -1, 3))
small = workloads.synthetic_rays(n, 8, sa=1e-6, sc=1e-6)
small.x = small.x * (fzp.rn[-1] / 0.2)
small.z = small.z * (fzp.rn[-1] / 0.2)
timed('zone plate, 3 orders', lambda: fzp.reflect(small), n)
cone = roe.ConicalMirror(bl, 'cone', center=[0, 20000., 0], pitch=4e-3, L0=900., theta=3e-3,
                         material=rm.Material('Rh', rho=12.41, kind='mirror'),
                         limPhysX=[-1.5, 1.5], limPhysY=[-250, 250])
timed('conical mirror', lambda: cone.reflect(beam), n)
pipe = ra.RoundAperture(bl, 'pipe', center=[0, 20000., 0], r=0.3)
timed('round aperture', lambda: pipe.propagate(beam), n)
si = rm.CrystalSi(hkl=(1, 1, 1), geom='Laue reflected', t=0.1)
lp = roe.LauePlate(bl, 'lp', center=[0, 20000., 0],
                   pitch=float(si.get_Bragg_angle(9000.)) + np.pi/2, material=si,
                   limPhysX=[-5, 5], limPhysY=[-5, 5])
timed('Laue plate', lambda: lp.reflect(beam), n)
