"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g5_index_table.npz and
tests/golden/g5_index_table.csv by RUNNING THE REFERENCE (imported from /root/reference, build
container only): Material(refractiveIndex = a table of E, n, k | a file of them)
(materials/material.py:240-262, 284-330, 364-373 -- cubic splines through n + ik, used when the
whole batch of energies lies inside the table, else the element tables):

  table            a synthetic optical-constants table of 40 energies, 8 .. 120 eV (a smooth
                   resonance in n and k: the VUV of a metal)
  amp_*            get_amplitude of a mirror and of a plate (from vacuum) at 500 random
                   (E, angle) inside the table, given as an ARRAY and as the FILE (k on a
                   sparser grid than n: splined onto n's energies first)
  out_*            the same with one energy outside the table: the reference then falls back
                   to the atomic scattering factors of the material's elements for the WHOLE call
  in_/gb_/lb_*     a flat mirror of that material reflecting 1024 rays at 10 degrees

While generating, oracle/materials_np.py ('refractiveIndex' = [energies, spline]) and
reflect_np.py are asserted against the reference.

Run:  python -m oracle.gen_fixtures_index_table
"""
import os

import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1
from . import materials_np as mn

CSV = os.path.join(g1.OUT, 'g5_index_table.csv')


def table():
    E = np.geomspace(8., 120., 40)
    x = (E - 38.) / 9.
    n = 0.92 - 0.25 * x / (1 + x * x) + 0.002 * np.log(E)
    k = 0.05 + 0.55 / (1 + x * x) + 3. / E
    return np.column_stack((E, n, k))


def write_csv(tab, path=CSV):
    """The reference's own text format (material.py:297-318): rows 'E, n' where only n is
    known, 'E, n, k' / 'E, , k' where k is: here k on every second energy only."""
    with open(path, 'w') as f:
        f.write('"Photon energy (eV)", n, k\n')
        for i, (e, n, k) in enumerate(tab):
            if i % 2 == 0 or i == len(tab) - 1:
                f.write('%.17g, %.17g, %.17g\n' % (e, n, k))
            else:
                f.write('%.17g, %.17g\n' % (e, n))


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    from .fixture_io import tables as load_tables
    tab = table()
    write_csv(tab)
    rng = np.random.default_rng(91)
    npts = 500
    E = rng.uniform(9., 110., npts)
    bdn = -np.cos(rng.uniform(0.1, 1.5, npts))
    out = dict(table=tab, E=E, bdn=bdn)
    for form, spec in (('array', tab), ('file', CSV)):
        for kind in ('mirror', 'plate'):
            m = rm.Material('Au', rho=19.32, kind=kind, refractiveIndex=spec)
            assert isinstance(m.refractiveIndex, list)
            ref = m.get_amplitude(E.copy(), bdn.copy(), True)
            om = mn.make_material([mn.load_element(load_tables(), 'Au')], None, kind, 19.32)
            om['refractiveIndex'] = [m.refractiveIndex[0], m.refractiveIndex[1]]
            mine = mn.material_amplitude(om, E.copy(), bdn.copy(), True)
            for i, lab in enumerate(('rs', 'rp', 'mu', 'nk')):
                assert np.allclose(mine[i], ref[i], rtol=1e-13, atol=0), (form, kind, lab)
                out['amp_%s_%s_%s' % (form, kind, lab)] = np.asarray(ref[i])
            out['n_%s' % form] = np.asarray(m.get_refractive_index(E.copy()))
    # one energy outside: atomic factors for the whole call
    Eo = E.copy()
    Eo[7] = 3000.
    m = rm.Material('Au', rho=19.32, kind='mirror', refractiveIndex=tab)
    ref = m.get_amplitude(Eo.copy(), bdn.copy(), True)
    plain = rm.Material('Au', rho=19.32, kind='mirror').get_amplitude(Eo.copy(), bdn.copy(), True)
    for i, lab in enumerate(('rs', 'rp', 'mu', 'nk')):
        assert np.array_equal(np.asarray(ref[i]), np.asarray(plain[i]))
        out['out_' + lab] = np.asarray(ref[i])
    out['E_out'] = Eo
    # a mirror of it in a beamline
    bl = raycing.BeamLine()
    mat = rm.Material('Au', rho=19.32, kind='mirror', refractiveIndex=tab)
    oe = roe.OE(bl, 'vuv', center=[0, 1000., 0], pitch=np.radians(10.), material=mat,
                limPhysX=[-5, 5], limPhysY=[-20, 20])
    beam = g1.make_rays(rs, 1024, 92, sx=0.5, sz=0.5, sa=1e-3, sc=1e-3, E=(20., 100.),
                        amplitudes=True, pol='mixed')
    beam.state[3] = 2
    par = g1.oe_params(oe, dict(kind='flat'))
    om = mn.make_material([mn.load_element(load_tables(), 'Au')], None, 'mirror', 19.32)
    om['refractiveIndex'] = [mat.refractiveIndex[0], mat.refractiveIndex[1]]
    par['material'] = om
    g1.run_reflect('g5_index_table', rs, oe, par, beam, **out)


if __name__ == '__main__':
    main()
