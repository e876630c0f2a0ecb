"""Where the host time of one SoftiMAX run goes (cProfile on the GPU box):
PYTHONPATH=. python tools/probe_softimax_host.py"""
import cProfile
import pstats
import time
import types

import numpy as np
import torch

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.sources as rs
import xrt_amd.backends.raycing.apertures as ra
import xrt_amd.backends.raycing.oes as roe
import xrt_amd.backends.raycing.materials as rm
import xrt_amd.backends.raycing.screens as rsc
import xrt_amd.backends.raycing.waves as rw
from xrt_amd.workloads import SoftiMAX

mods = types.SimpleNamespace(raycing=raycing, rs=rs, ra=ra, roe=roe, rm=rm, rsc=rsc, rw=rw)
np.random.seed(1)
scene = SoftiMAX(mods, nrays=200000)
scene.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
scene.run()
torch.cuda.synchronize()
print('one run: %.3f s' % (time.perf_counter() - t0))
pr = cProfile.Profile()
pr.enable()
scene.run()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(14)
st.sort_stats('cumtime').print_stats(45)
