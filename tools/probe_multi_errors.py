"""Per-bounce deviation of OE.multiple_reflect on the GPU from the reference's goldens
(g2_multi_*): max |difference| of every field, bounce by bounce, and on gb.
    python tools/probe_multi_errors.py [case ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_multiple_reflect as t            # noqa: E402
from p1_cases import GOLDEN, product_beam       # noqa: E402

names = sys.argv[1:] or ['g2_multi_cylinder', 'g2_multi_toroid', 'g2_multi_edges',
                         'g2_multi_flat', 'g2_multi_capillary']
os.environ.setdefault('XRT_HIP_USER_CACHE', '/tmp/xrt_units')
for name in names:
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    oe = t.element(name)
    beam = product_beam(g)
    info = []
    gb, lbN = oe.multiple_reflect(beam, maxReflections=int(g['maxReflections']),
                                  needElevationMap=bool(g['needElevationMap']), _info=info)
    n = beam.nrays
    nb = lbN.nrays // n
    print(name, 'bounces', nb, 'golden', int(g['bounces']))
    fields = [f for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'Jss', 'Jsp', 'Es', 'theta',
                          'elevationD', 'elevationY', 's', 'phi', 'r') if 'lbN_' + f in g.files]
    for k in range(min(nb, int(g['bounces']))):
        sl = slice(k * n, (k + 1) * n)
        row = ['%s %.1e' % (f, np.abs(getattr(lbN, f)[sl] - g['lbN_' + f][sl]).max())
               for f in fields]
        same = np.array_equal(lbN.state[sl], g['lbN_state'][sl]) and \
            np.array_equal(lbN.nRefl[sl], g['lbN_nRefl'][sl])
        print('  bounce', k, 'states', 'ok' if same else 'DIFFER', ' '.join(row), info[k])
    print('  gb', ' '.join('%s %.1e' % (f, np.abs(getattr(gb, f) - g['gb_' + f]).max())
                           for f in fields if 'gb_' + f in g.files),
          'states', np.array_equal(gb.state, g['gb_state']))
