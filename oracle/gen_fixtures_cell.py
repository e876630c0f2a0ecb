"""TEST INFRASTRUCTURE ONLY -- regenerates the CrystalFromCell goldens by RUNNING THE
REFERENCE (imported from /root/reference, build container only;
materials/crystals_basic.py:157-440):

  g3_cell_rocking_curves.npz  CrystalFromCell.get_amplitude on angle grids: alpha-quartz
                              (1 0 2) -- hexagonal cell, two elements --, graphite (0 0 2),
                              quartz with partial oxygen occupancy; thick Bragg, thin
                              Bragg / Laue, reflected / transmitted, asymmetric cuts
  g3_cell_quartz_flat.npz     OE + quartz (1 0 2) at the 8 keV Bragg angle, asymmetric cut
  g3_cell_graphite_johann.npz JohannCylinder + graphite (0 0 2)

While generating, oracle/materials_np.py ('cell' structure factor) and
oracle/reflect_np.py are asserted against the reference.

Run:  python -m oracle.gen_fixtures_cell
"""
import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1
from . import materials_np as mn

QUARTZ_XYZ = [[0.4697, 0., 0.], [-0.4697, -0.4697, 1./3], [0., 0.4697, 2./3],
              [0.4125, 0.2662, 0.1188], [-0.1463, -0.4125, 0.4521],
              [-0.2662, 0.1463, -0.2145], [0.1463, -0.2662, -0.1188],
              [-0.4125, -0.1463, 0.2145], [0.2662, 0.4125, 0.5479]]
# name -> CrystalFromCell keyword arguments (atoms by symbol)
CELLS = {
    'quartz102': dict(hkl=(1, 0, 2), a=4.91304, c=5.40463, gamma=120,
                      atoms=['Si']*3 + ['O']*6, atomsXYZ=QUARTZ_XYZ),
    'graphite002': dict(hkl=(0, 0, 2), a=2.456, c=6.696, gamma=120, atoms=['C']*4,
                        atomsXYZ=[[0., 0., 0.], [0., 0., 0.5], [1./3, 2./3, 0.],
                                  [2./3, 1./3, 0.5]]),
    'quartz_partial': dict(hkl=(2, 0, 3), a=4.91304, c=5.40463, gamma=120,
                           atoms=['Si']*3 + ['O']*6, atomsXYZ=QUARTZ_XYZ,
                           atomsFraction=[1., 1., 0.9, 0.8, 1., 1., 0.7, 1., 1.],
                           factDW=0.95),
}


def all_tables():
    from . import gen_fixtures_multilayer as gm
    return gm.all_tables()


def oracle_cell(tables, name, **over):
    kw = dict(CELLS[name])
    kw.update(over)
    cache = {}
    elems = [cache.setdefault(a, mn.load_element(tables, a)) for a in kw.pop('atoms')]
    return mn.make_crystal_from_cell(elems, kw.pop('atomsXYZ'), kw.pop('hkl'), **kw)


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    tables = all_tables()

    out = {}
    E0, npts = 8000., 300
    for name in CELLS:
        for geom, tmm in (('Bragg reflected', None), ('Bragg reflected', 0.02),
                          ('Bragg transmitted', 0.02), ('Laue reflected', 0.05),
                          ('Laue transmitted', 0.05)):
            for alphaDeg in (0., 4.):
                c = rm.CrystalFromCell(name, geom=geom, t=tmm, **CELLS[name])
                cr = oracle_cell(tables, name, geom=geom, t=tmm)
                assert cr['d'] == c.d and cr['V'] == c.V and cr['chiToF'] == c.chiToF
                alpha = np.radians(alphaDeg)
                theta = c.get_Bragg_angle(E0) + np.linspace(-80, 80, npts) * 4.848e-6
                E = np.ones(npts) * E0
                if geom.startswith('Bragg'):
                    g0, gh = -np.sin(theta + alpha), np.sin(theta - alpha)
                else:
                    g0, gh = -np.cos(theta + alpha), -np.cos(theta - alpha)
                hns = -np.sin(theta)
                ref = c.get_amplitude(E.copy(), g0.copy(), gh.copy(), hns.copy())
                mine = mn.crystal_amplitude(cr, E.copy(), g0.copy(), gh.copy(), hns.copy())
                key = '%s_%s_%s_%+d' % (name, geom.replace(' ', ''),
                                        'thick' if tmm is None else '%gum' % (tmm*1e3),
                                        int(alphaDeg))
                for i, lab in enumerate(('S', 'P')):
                    sc = np.abs(ref[i]).max()
                    assert np.abs(mine[i] - ref[i]).max() <= 1e-12 * sc, key
                    out[key + '_' + lab] = np.array(ref[i])
                out[key + '_in'] = np.array([E, g0, gh, hns])
                print(key, 'max |S| %.3f' % np.abs(ref[0]).max())
    g1.save('g3_cell_rocking_curves', **out)

    n = 1024
    # flat element, asymmetrically cut quartz
    bl = raycing.BeamLine()
    alpha = np.radians(3.)
    c = rm.CrystalFromCell('quartz102', **CELLS['quartz102'])
    thB = float(c.get_Bragg_angle(E0) - c.get_dtheta(E0, alpha))
    oe = roe.OE(bl, 'qz', center=[0, 2000., 0], pitch=thB + alpha, alpha=alpha, material=c,
                limPhysX=[-8, 8], limPhysY=[-30, 30])
    beam = g1.make_rays(rs, n, 140, sx=0.4, sz=0.4, sa=2e-5, sc=3e-5, E=(E0 - 1., E0 + 1.),
                        amplitudes=True, pol='mixed')
    beam.state[2] = 2
    beam.state[3] = -2
    par = g1.oe_params(oe, dict(kind='flat', alpha=alpha))
    par['material'] = oracle_cell(tables, 'quartz102')
    g1.run_reflect('g3_cell_quartz_flat', rs, oe, par, beam, cell=np.array('quartz102'),
                   alpha=np.array(alpha))

    # graphite on a Johann cylinder
    bl = raycing.BeamLine()
    c = rm.CrystalFromCell('graphite002', **CELLS['graphite002'])
    thB = float(c.get_Bragg_angle(E0))
    Rm = 800.
    oe = roe.JohannCylinder(bl, 'gr', center=[0, Rm*np.sin(thB), 0], pitch=thB, Rm=Rm,
                            material=c, limPhysX=[-10, 10], limPhysY=[-30, 30])
    beam = g1.make_rays(rs, n, 141, sx=0.02, sz=0.02, sa=1.5e-2, sc=1.5e-2,
                        E=(E0 - 5., E0 + 5.), amplitudes=True, pol='mixed')
    surf = dict(kind='bent_cylinder', Rm=Rm, planes='johann', alpha=None,
                crossSection='circular')
    par = g1.oe_params(oe, surf)
    par['material'] = oracle_cell(tables, 'graphite002')
    g1.run_reflect('g3_cell_graphite_johann', rs, oe, par, beam,
                   cell=np.array('graphite002'), surf_Rm=np.array(Rm))


if __name__ == '__main__':
    main()
