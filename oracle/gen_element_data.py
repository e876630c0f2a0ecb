"""Build-container tool: extracts the per-element DATA the backend needs on the
GPU box (tabulated E/f1/f2 'Chantler total', Waasmaier-Kirfel f0 coefficients,
atomic masses, Z = 1..92) through the reference's own Element API and writes
xrt_amd/data/elements.npz. Data only; no reference code is copied.

    python -m oracle.gen_element_data
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _refenv  # noqa: E402


def main():
    _refenv.activate()
    import xrt.backends.raycing.materials as rm
    from xrt.backends.raycing.materials.element import elementsList
    out = {}
    names = []
    for Z in range(1, 93):
        name = elementsList[Z]
        try:
            e = rm.Element(name, table='Chantler total')
        except Exception as ex:  # noqa: BLE001
            print('skip', name, ex)
            continue
        names.append(name)
        out[name + '_Z'] = np.array(e.Z, dtype=np.int32)
        out[name + '_mass'] = np.array(e.mass, dtype=np.float64)
        out[name + '_f0'] = np.array(e.f0coeffs, dtype=np.float64)
        out[name + '_E'] = np.array(e.E, dtype=np.float64)
        out[name + '_f1'] = np.array(e.f1, dtype=np.float64)
        out[name + '_f2'] = np.array(e.f2, dtype=np.float64)
    out['names'] = np.array(names)
    out['table'] = np.array('Chantler total')
    path = os.path.join(ROOT, 'xrt_amd', 'data', 'elements.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB', len(names), 'elements')


if __name__ == '__main__':
    main()
