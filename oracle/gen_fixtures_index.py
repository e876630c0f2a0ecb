"""TEST INFRASTRUCTURE ONLY -- regenerates the goldens of materials with a user-given
constant refractive index by RUNNING THE REFERENCE (imported from /root/reference, build
container only; materials/material.py:240-262, 364-373) -- visible light, where the tables
of scattering factors do not reach:

  g5_fixed_index.npz   Material(refractiveIndex=n).get_amplitude: mirror, thin mirror, plate
                       from vacuum and from inside, at steep and grazing angles
  g2_plate_glass.npz   Plate.double_refract through 2 mm of glass (n = 1.52 + 1e-7j) at 2 eV

While generating, oracle/materials_np.py / reflect_np.py are asserted against the reference.

Run:  python -m oracle.gen_fixtures_index
"""
import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1
from . import materials_np as mn
from . import reflect_np as rn

INDEX = {'glass': 1.52 + 1e-7j, 'metal': 0.2 + 3.4j, 'real': 1.33}


def oracle_material(name, kind, t=None):
    m = mn.make_material([], None, kind, 0., t)
    m['refractiveIndex'] = complex(INDEX[name])
    return m


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    rng = np.random.default_rng(77)
    npts = 600
    E = rng.uniform(1.5, 3.5, npts)
    ang = rng.uniform(0.02, 1.55, npts)         # angle of incidence from the normal
    out = dict(E=E, bdn=-np.cos(ang))
    for name in INDEX:
        for kind, t in (('mirror', None), ('thin mirror', 2e-4), ('plate', None)):
            for fromVacuum in ((True, False) if kind == 'plate' else (True,)):
                m = rm.Material([], kind=kind, t=t, refractiveIndex=INDEX[name])
                bdn = -np.cos(ang) if fromVacuum else np.cos(ang * 0.4)
                ref = m.get_amplitude(E.copy(), bdn.copy(), fromVacuum)
                mine = mn.material_amplitude(oracle_material(name, kind, t), E.copy(),
                                             bdn.copy(), fromVacuum)
                key = '%s_%s_%d' % (name, kind.replace(' ', ''), fromVacuum)
                for i, lab in enumerate(('rs', 'rp', 'mu', 'nk')):
                    r = np.asarray(ref[i]) * np.ones(npts)
                    mm = np.asarray(mine[i]) * np.ones(npts)
                    assert np.allclose(mm, r, rtol=1e-13, atol=0), (key, lab)
                    out[key + '_' + lab] = r
                out[key + '_bdn'] = bdn
    g1.save('g5_fixed_index', **out)

    bl = raycing.BeamLine()
    glass = rm.Material([], kind='plate', refractiveIndex=INDEX['glass'])
    plate = roe.Plate(bl, 'win', center=[0, 500., 0], pitch=np.pi/2 - 0.3, material=glass,
                      t=2., limPhysX=[-3, 4], limPhysY=[-2, 2])
    beam = g1.make_rays(rs, 1024, 160, sx=0.8, sz=0.5, sa=3e-3, sc=3e-3, E=(1.8, 2.4),
                        amplitudes=True, pol='mixed')
    beam.state[2] = 2
    beam.state[3] = -1
    gb, lo1, lo2 = plate.double_refract(beam)
    par = g1.oe_params(plate, dict(kind='flat'))
    par['surface2'] = dict(kind='flat')
    par['material'] = par['material2'] = oracle_material('glass', 'plate')
    m2, m1l, m2l = rn.dcm_double_reflect(par, g1.to_oracle_beam(beam), fromVacuum1=True,
                                         fromVacuum2=False, is_plate=True)
    g1.assert_beams('glass:gb', m2, gb)
    g1.assert_beams('glass:lo1', m1l, lo1)
    g1.assert_beams('glass:lo2', m2l, lo2)
    st, cnt = np.unique(gb.state, return_counts=True)
    print('g2_plate_glass states', dict(zip(st.tolist(), cnt.tolist())), 'mean T',
          (gb.Jss + gb.Jpp)[gb.state == 1].mean())
    res = {}
    for prefix, b in (('in_', beam), ('gb_', gb), ('lo1_', lo1), ('lo2_', lo2)):
        res.update(g1.beam_dict(prefix, b))
    res.update(g1.flat_params(par))
    res.update(plate_t=np.array(2.))
    g1.save('g2_plate_glass', **res)


if __name__ == '__main__':
    main()
