"""Runs the reference's published wave benchmark (SoftiMAX, 2e5 samples per
wave) on this package and prints the time per stage.
    python tools/run_softimax.py [nrays]"""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def product_modules():
    import xrt_amd.backends.raycing as raycing
    import xrt_amd.backends.raycing.sources as rs
    import xrt_amd.backends.raycing.apertures as ra
    import xrt_amd.backends.raycing.oes as roe
    import xrt_amd.backends.raycing.materials as rm
    import xrt_amd.backends.raycing.screens as rsc
    import xrt_amd.backends.raycing.waves as rw
    return types.SimpleNamespace(raycing=raycing, rs=rs, ra=ra, roe=roe, rm=rm,
                                 rsc=rsc, rw=rw)


def main():
    from xrt_amd.workloads import SoftiMAX
    nrays = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    np.random.seed(1)
    t0 = time.perf_counter()
    scene = SoftiMAX(product_modules(), nrays=nrays)
    profile = os.environ.get('SOFTI_PROFILE')
    for rep in range(2):
        last = [time.perf_counter()]
        t1 = last[0]

        def stage(name, beam):
            torch.cuda.synchronize()
            now = time.perf_counter()
            print('  %-14s %8.3f s   flux %.4e' % (
                name, now - last[0], float((beam.Jss + beam.Jpp).sum())), flush=True)
            last[0] = time.perf_counter()
        if profile and rep == 1:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            out = scene.run(stage)
            pr.disable()
            pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
        else:
            out = scene.run(stage)
        torch.cuda.synchronize()
        print('run %d: %.3f s (nrays %d, source grid %d nodes)' % (
            rep, time.perf_counter() - t1, nrays, scene.bl.source.quadm), flush=True)


if __name__ == '__main__':
    main()
