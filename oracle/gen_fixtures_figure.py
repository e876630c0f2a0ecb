"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g2_figure_*.npz / g3_figure_crystal.npz by
RUNNING THE REFERENCE (imported from /root/reference, build container only): optical elements
with OE(figureError=...) (xrt/backends/raycing/figure_error.py; the hooks on the ray path are
oes/base.py:826-830 -- the map's height inside find_dz -- and oes/reflect.py:767-775 -- the
normal at the hit point turned by [d_pitch, d_roll]).

  g2_figure_toroid    the cfg2 toroid mirror (Pt) under RandomRoughness(rms 3 nm, corrLength
                      4 mm, seed 11): rays that miss, fall off the edges and graze included
  g2_figure_flat      a flat Pt mirror under Waviness on top of a GaussianBump (baseFE): maps add
  g3_figure_crystal   a flat Si(111) crystal at its Bragg angle under Waviness (10 nm, 8 x 20 mm)
  g2_figure_imported  a bent-flat mirror under a FigureErrorImported map read from
                      tests/golden/figure_map_nom.txt (written here: a measured-like profile on a
                      41 x 161 grid, 0.5 mm x 1 mm steps, file columns y x z in m, m, um)

Each golden also holds the map's spline as scipy made it in the reference (knots ty, tx,
coefficients c) so that the tests can (i) pin the product's own map generators and spline
against it and (ii) rebuild the oracle's figure functions without the reference. While
generating, oracle/reflect_np.py with the two figure hooks is asserted against the reference's
beams.

Run:  python -m oracle.gen_fixtures_figure
"""
import os

import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1
from .fixture_io import tables as load_tables

MAP_FILE = os.path.join(g1.OUT, 'figure_map_nom.txt')


def figure_hooks(fe):
    """The oracle's two figure functions from a reference figure-error object."""
    return dict(figure_z=lambda x, y: fe.local_z_distorted(x, y),
                figure_n=lambda x, y: fe.local_n_distorted(x, y))


def spline_extra(fe):
    ty, tx, c = fe.local_z_spline.tck
    return dict(fe_ty=np.array(ty), fe_tx=np.array(tx), fe_c=np.array(c),
                fe_k=np.array(fe.local_z_spline.degrees[0]),
                fe_shift=np.array([fe.xShift, fe.yShift]), fe_z2d=np.array(fe.z2d))


def write_map_file():
    """A map like a slope-measuring instrument would leave: y fastest, columns y [m], x [m],
    height [um]."""
    rng = np.random.default_rng(5)
    x = np.arange(-10., 10.01, 0.5)
    y = np.arange(-80., 80.01, 1.)
    X, Y = np.meshgrid(x, y, indexing='ij')
    Z = 4e-3 * np.cos(2*np.pi*Y/55.) * (1 + 0.2*X/10.) + 1.5e-3 * np.sin(2*np.pi*X/13.) + \
        3e-4 * rng.normal(size=X.shape)
    np.savetxt(MAP_FILE, np.column_stack([Y.ravel()*1e-3, X.ravel()*1e-3, Z.ravel()]),
               fmt='%.9e')


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    import xrt.backends.raycing.figure_error as rfe
    raycing._VERBOSITY_ = 0
    tables = load_tables()
    pt = rm.Material('Pt', rho=21.45, kind='mirror')

    # ---- toroid + random roughness ------------------------------------------------------
    bl = raycing.BeamLine()
    fe = rfe.RandomRoughness(rms=3., corrLength=4., seed=11, limPhysX=[-10, 10],
                             limPhysY=[-300, 300], gridStep=2.)
    p_, q_, pitch = 20000., 10000., 4e-3
    tm = roe.ToroidMirror(bl, 'm1', center=[0, p_, 0], pitch=pitch, material=pt,
                          R=2*p_*q_/((p_+q_)*np.sin(pitch)), r=2*p_*q_*np.sin(pitch)/(p_+q_),
                          limPhysX=[-10, 10], limPhysY=[-300, 300], figureError=fe)
    beam = g1.make_rays(rs, 4096, 101, amplitudes=True, pol='mixed')
    beam.x[:64] = np.linspace(-14., 14., 64)
    beam.c[64:128] = np.linspace(-3e-5, 3e-5, 64)
    beam.z[128:160] = np.linspace(-1.5, 1.5, 32)
    beam.b[:] = np.sqrt(1 - beam.a**2 - beam.c**2)
    beam.state[200] = 2
    beam.state[201] = -3
    par = g1.oe_params(tm, dict(kind='toroid', R=tm.R, r=tm.r, **figure_hooks(fe)))
    par['material'] = g1.material_dict(tables, pt)
    g1.run_reflect('g2_figure_toroid', rs, tm, par, beam, surf_Rr=np.array([tm.R, tm.r]),
                   rough=np.array([3., 4., 11.]), **spline_extra(fe))

    # ---- flat mirror + waviness on a bump -----------------------------------------------
    bl = raycing.BeamLine()
    bump = rfe.GaussianBump(bumpHeight=25., cX=1., cY=-20., sigmaX=3., sigmaY=40.,
                            limPhysX=[-8, 8], limPhysY=[-150, 150], gridStep=1.)
    wav = rfe.Waviness(amplitude=6., xWaveLength=7., yWaveLength=60., baseFE=bump,
                       limPhysX=[-8, 8], limPhysY=[-150, 150], gridStep=1.)
    fm = roe.OE(bl, 'flat', center=[0, 15000., 0], pitch=5e-3, material=pt,
                limPhysX=[-8, 8], limPhysY=[-150, 150], figureError=wav)
    beam = g1.make_rays(rs, 2048, 102, sx=1.5, sz=0.25, sa=1e-4, sc=1e-5,
                        amplitudes=True, pol='mixed')
    beam.state[5] = 3
    beam.x[6] = 9.5
    par = g1.oe_params(fm, dict(kind='flat', **figure_hooks(wav)))
    par['material'] = g1.material_dict(tables, pt)
    g1.run_reflect('g2_figure_flat', rs, fm, par, beam, **spline_extra(wav))

    # ---- flat Bragg crystal + waviness --------------------------------------------------
    bl = raycing.BeamLine()
    si = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
    thB = float(np.ravel(si.get_Bragg_angle(9000.) - si.get_dtheta(9000.))[0])
    wav2 = rfe.Waviness(amplitude=10., xWaveLength=8., yWaveLength=20., limPhysX=[-6, 6],
                        limPhysY=[-30, 30], gridStep=0.25)
    xt = roe.OE(bl, 'xtal', center=[0, 25000., 0], pitch=thB, material=si,
                limPhysX=[-6, 6], limPhysY=[-30, 30], figureError=wav2)
    beam = g1.make_rays(rs, 2048, 103, sx=1., sz=0.4, sa=2e-5, sc=6e-6, E=(8999., 9001.),
                        amplitudes=True, pol='mixed')
    beam.state[1] = 2
    beam.state[2] = -2
    par = g1.oe_params(xt, dict(kind='flat', **figure_hooks(wav2)))
    par['material'] = g1.crystal_dict(tables, si)
    g1.run_reflect('g3_figure_crystal', rs, xt, par, beam, bragg=np.array(thB),
                   **spline_extra(wav2))

    # ---- bent mirror + an imported map --------------------------------------------------
    write_map_file()
    bl = raycing.BeamLine()
    imp = rfe.FigureErrorImported(fileName=MAP_FILE, orientation='YXZ',
                                  columnFactors=[1e3, 1e3, 1e3])
    bm = roe.BentFlatMirror(bl, 'bent', center=[0, 18000., 0], pitch=3.5e-3, material=pt,
                            R=5e6, limPhysX=[-10, 10], limPhysY=[-80, 80], figureError=imp)
    beam = g1.make_rays(rs, 2048, 104, sx=2., sz=0.12, sa=5e-5, sc=8e-6,
                        amplitudes=True, pol='mixed')
    beam.x[3] = -12.
    par = g1.oe_params(bm, dict(kind='bentflat', R=bm.R, y0=bm.limPhysY[0],
                                **figure_hooks(imp)))
    par['material'] = g1.material_dict(tables, pt)
    g1.run_reflect('g2_figure_imported', rs, bm, par, beam, surf_R=np.array(bm.R),
                   **spline_extra(imp))


if __name__ == '__main__':
    main()
