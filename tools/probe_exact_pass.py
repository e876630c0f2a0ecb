"""Per-call time of OE.reflect on the golden cases whose passes run the generic kernels of the
surface families 1 / 2 and of layered materials (2048 rays: the time is launch overhead and
host glue, which is where a private segment of 1 KB per lane shows -- DESIGN 5.2):
    XRT_HIP_LIBRARY=<another build> python tools/probe_exact_pass.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import p1_cases as pc  # noqa: E402

CASES = ('g2_toroid_pt', 'g2_toroid_brent', 'g2_ellipse_cyl', 'g2_ellipse_full', 'g2_parabola_q',
         'g2_blazed_au', 'g2_multilayer_flat', 'g2_coated_toroid', 'g2_plate_be',
         'g2_cone_rh', 'g2_grating_vls')


def main():
    reps = 300
    for name in CASES:
        try:
            g = pc.load(name)
            oe = pc.product_oe(name, g)
        except Exception as e:          # a case this tree does not have
            print('%-22s skipped (%s)' % (name, type(e).__name__))
            continue
        beam = pc.product_beam(g)
        call = (lambda: oe.double_refract(beam)) if name.startswith('g2_plate') else \
            (lambda: oe.reflect(beam))
        np.random.seed(1)
        call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        print('%-22s %8.1f us per call' % (name, (time.perf_counter() - t0) / reps * 1e6))


if __name__ == '__main__':
    main()
