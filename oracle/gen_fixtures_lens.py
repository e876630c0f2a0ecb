"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g2_lens_*.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only): refractive lenses
(oes/refractive.py:237-663), single lenslets through ``double_refract`` and stacks
(compound refractive lenses) through ``multiple_refract``:

  g2_lens_crl3.npz     ParaboloidFlatLens, Be, nCRL = 3, zmax given (the stack walks
                       along the rotated local z between the lenslets)
  g2_lens_cyl2.npz     DoubleParabolicCylinderLens, nCRL = 2, slightly off normal
                       incidence and rolled (the walk has all three components)
  g2_lens_single.npz   DoubleParaboloidLens, nCRL = 1 (plain double_refract), rays
                       beyond zmax hit the flat rim

While generating, the numpy restatement (oracle/reflect_np.py: the 'paraboloid'
surface and lens_multiple_refract) is asserted against the reference.

Run:  python -m oracle.gen_fixtures_lens
"""
import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1
from . import reflect_np as rn

CASES = (
    ('g2_lens_crl3', 'ParaboloidFlatLens',
     dict(pitch=np.pi/2, t=0.05, focus=0.25, zmax=0.4, nCRL=3,
          limPhysX=[-1, 1], limPhysY=[-1, 1]), 81, 0.2),
    ('g2_lens_cyl2', 'DoubleParabolicCylinderLens',
     dict(pitch=np.pi/2 - 0.01, roll=0.02, t=0.03, focus=0.4, zmax=0.3, nCRL=2,
          limPhysX=[-0.8, 0.9], limPhysY=[-0.7, 0.7]), 82, 0.25),
    ('g2_lens_single', 'DoubleParaboloidLens',
     dict(pitch=np.pi/2, t=0.1, focus=0.3, zmax=0.2, nCRL=1,
          limPhysX=[-1, 1], limPhysY=[-1, 1]), 83, 0.3),
)


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    tables = dict(np.load(g1.os.path.join(g1.OUT, 'g6_element_tables.npz')))
    eBe = rm.Element('Be', table='Chantler total')
    for key, val in (('Z', eBe.Z), ('mass', eBe.mass), ('f0', eBe.f0coeffs),
                     ('E', eBe.E), ('f1', eBe.f1), ('f2', eBe.f2)):
        tables['Be_' + key] = np.array(val, dtype=float)
    n = 1024
    for tag, cls_name, kw, seed, spread in CASES:
        bl = raycing.BeamLine()
        mBe = rm.Material('Be', rho=1.848, kind='lens')
        lens = getattr(roe, cls_name)(bl, 'crl', center=[0, 1000., 0], material=mBe, **kw)
        beam = g1.make_rays(rs, n, seed, sx=spread, sz=spread, sa=1e-5, sc=1e-5,
                            E=(8990., 9010.), amplitudes=True, pol='mixed')
        beam.state[3] = 3
        beam.state[4] = -4
        beam.x[5] = 7.          # misses the lens
        gb, l1, l2 = lens.multiple_refract(beam)
        surf = dict(kind='paraboloid', focus=lens.focus, zmax=lens.zmax,
                    cylinder='Cylinder' in cls_name)
        par = g1.oe_params(lens, surf)
        par['surface2'] = surf
        par['material'] = g1.material_dict(tables, mBe)
        par['material2'] = par['material']
        par.update(nCRL=lens.nCRL, zmax=lens.zmax, t=lens.t,
                   double_sided=cls_name.startswith('Double'))
        m2, m1l, m2l = rn.lens_multiple_refract(par, g1.to_oracle_beam(beam))
        g1.assert_beams(tag + ':gb', m2, gb)
        g1.assert_beams(tag + ':lo1', m1l, l1)
        g1.assert_beams(tag + ':lo2', m2l, l2)
        st, cnt = np.unique(gb.state, return_counts=True)
        good = gb.state == 1
        print(tag, 'states', dict(zip(st.tolist(), cnt.tolist())), 'mean T',
              (gb.Jss + gb.Jpp)[good].mean(), 'rms x\'', gb.a[good].std())
        out = {}
        for prefix, b in (('in_', beam), ('gb_', gb), ('lo1_', l1), ('lo2_', l2)):
            out.update(g1.beam_dict(prefix, b))
        flat = dict(par)
        for key in ('nCRL', 'zmax', 't', 'double_sided'):
            flat.pop(key)
        out.update(g1.flat_params(flat))
        out.update(lens_class=np.array(cls_name), lens_t=np.array(float(lens.t)),
                   lens_focus=np.array(float(lens.focus)),
                   lens_zmax=np.array(np.nan if lens.zmax is None else float(lens.zmax)),
                   lens_nCRL=np.array(int(lens.nCRL)), mat_rho=np.array(1.848))
        for key in ('Z', 'mass', 'f0', 'E', 'f1', 'f2'):
            out['Be_' + key] = tables['Be_' + key]
        g1.save(tag, **out)


if __name__ == '__main__':
    main()
