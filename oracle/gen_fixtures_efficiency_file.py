"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g2_grating_efffile.npz and the data
file tests/golden/g2_grating_efffile.txt by RUNNING THE REFERENCE (imported from
/root/reference, build container only): a plane VLS grating whose material is
Material(kind='grating', efficiency=[[order, column], ...], efficiencyFile=...)
(materials/material.py:78-95, 335-346, 391-413): the efficiency of each order against energy
in the columns of a text file, interpolated linearly at every ray's energy (np.interp), the
amplitude its square root; an order the list does not name gets none; an energy outside the
table is an error.

  table            31 energies 255 .. 305 eV; columns 1..3 = three smooth efficiency curves
  in_/gb_/lb_*     2048 rays of 270 .. 290 eV, orders (1, -1, 2, 0) drawn per ray (seeded)

While generating, oracle/reflect_np.py ('efficiency_table') is asserted against the reference.

Run:  python -m oracle.gen_fixtures_efficiency_file
"""
import os

import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1
from .fixture_io import tables as load_tables

TXT = os.path.join(g1.OUT, 'g2_grating_efffile.txt')
PAIRS = [[1, 1], [-1, 3], [2, 2]]       # [order, column of the file]


def table():
    E = np.linspace(255., 305., 31)
    x = (E - 280.) / 25.
    c1 = 0.30 + 0.08 * x - 0.05 * x * x
    c2 = 0.05 + 0.02 * np.sin(3. * x)
    c3 = 0.12 * np.exp(-x * x) + 0.01
    return np.column_stack((E, c1, c2, c3))


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    raycing._VERBOSITY_ = 0
    tab = table()
    np.savetxt(TXT, tab, fmt='%.17g', header='E (eV), efficiency of three orders')
    order = (1, -1, 2, 0)
    density = ['y', 300., 1., 2.4e-4]
    mG = rm.Material('Au', rho=19.32, kind='grating', efficiency=PAIRS, efficiencyFile=TXT)
    assert np.array_equal(mG.efficiency_E, tab[:, 0])
    assert np.array_equal(mG.efficiency_I, tab[:, [1, 3, 2]].T)
    bl = raycing.BeamLine()
    gr = roe.OE(bl, 'gr', center=[0, 2000., 0.], pitch=np.radians(2.2), material=mG,
                order=order, limPhysX=(-3, 3), limPhysY=(-45, 45), alarmLevel=None,
                gratingDensity=density)
    beam = g1.make_rays(rs, 2048, 69, sx=1.0, sz=0.9, sa=3e-5, sc=2e-5, E=(270., 290.),
                        amplitudes=True, pol='mixed')
    beam.E[5] = tab[7, 0]           # exactly on a node of the table
    beam.E[6] = tab[0, 0]           # and on its two ends
    beam.E[7] = tab[-1, 0]
    beam.state[3] = 3
    beam.state[4] = -4
    par = g1.oe_params(gr, dict(kind='flat'))
    par['material'] = g1.material_dict(load_tables(), mG)
    par['material']['efficiency'] = [[o, c] for o, c in PAIRS]
    par['material']['efficiency_table'] = (mG.efficiency_E, mG.efficiency_I)
    par['order'] = order
    par['gratingDensity'] = density
    g1.run_reflect('g2_grating_efffile', rs, gr, par, beam, mat_rho=np.array(19.32),
                   order=np.array(order), np_seed=20260930,
                   efficiency=np.array(PAIRS, dtype=float), eff_E=mG.efficiency_E,
                   eff_I=mG.efficiency_I, gd_axis=np.array(density[0]),
                   gd_coeffs=np.array(density[1:], dtype=float))
    # outside the table: the reference raises
    beam.E[100] = 400.
    try:
        np.random.seed(1)
        gr.reflect(beam)
    except ValueError as e:
        assert 'out of the efficiency table range' in str(e)
    else:
        raise AssertionError('no ValueError outside the table')


if __name__ == '__main__':
    main()
