"""Build-container tool: extracts the DATA of the reference's predefined materials
(materials/elemental.py, compounds.py, crystals.py: chemical formulas, densities, lattice
constants, atoms of the unit cells) by instantiating each class through the reference's API
and writes xrt_amd/data/materials.json. Data only; no reference code is copied.

    python -m oracle.gen_material_data
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _refenv  # noqa: E402


def main():
    _refenv.activate()
    import xrt.backends.raycing.materials.elemental as el
    import xrt.backends.raycing.materials.compounds as co
    import xrt.backends.raycing.materials.crystals as cr
    import xrt.backends.raycing.materials as rm
    out = {'elemental': {}, 'compounds': {}, 'crystals': {}}
    for key, module in (('elemental', el), ('compounds', co)):
        for name in module.__all__:
            m = getattr(module, name)()
            out[key][name] = dict(elements=[e.name for e in m.elements],
                                  quantities=[float(q) for q in m.quantities],
                                  rho=float(m.rho), name=m.name)
    for name in cr.__all__:
        c = getattr(cr, name)()
        if isinstance(c, rm.CrystalFromCell):
            out['crystals'][name] = dict(
                base='cell', name=c.name, a=float(c.a), b=float(c.b), c=float(c.c),
                alpha=float(c.alpha), beta=float(c.beta), gamma=float(c.gamma),
                atoms=[int(e.Z) for e in c.elements],
                atomsXYZ=[[float(v) for v in r] for r in c.atomsXYZ],
                atomsFraction=[float(f) for f in c.atomsFraction])
        else:
            out['crystals'][name] = dict(base='diamond', name=c.name, a=float(c.a),
                                         elements=[e.name for e in c.elements])
    path = os.path.join(ROOT, 'xrt_amd', 'data', 'materials.json')
    with open(path, 'w') as f:
        json.dump(out, f, separators=(',', ':'), sort_keys=True)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB',
          {k: len(v) for k, v in out.items()})


if __name__ == '__main__':
    main()
