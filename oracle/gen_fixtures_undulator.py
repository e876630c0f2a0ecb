"""TEST INFRASTRUCTURE ONLY — regenerates tests/golden/g9_undulator_*.npz by
RUNNING THE REFERENCE (imported from /root/reference, build container only).

G9: inputs and outputs of the reference's numpy undulator
integrals `Undulator._sp_sum` (synchr.py:1930-2038) and the scaled
`_build_I_map_conv` (synchr.py:2050-2108) for
  far_planar   planar undulator, far field (kernel `undulator`)
  far_helical  Kx = Ky, phase 90 deg
  taper        tapered gap (kernel `undulator_taper`, all Np periods summed)
  nf           near field, R0 = 25 m (kernel `undulator_nf`)
Each file holds the node tables, the per-ray arguments exactly as
`_build_I_map_CL` (synchr.py:2110-2176) would marshal them for
`run_parallel`, the raw sums and the scaled (I, Es, Ep). While generating, the
restatement in oracle/undulator_np.py is checked against the reference's
functions on the very same inputs.

Run:  python -m oracle.gen_fixtures_undulator
"""
import os
import numpy as np
from . import _refenv
from . import undulator_np as un

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')

COMMON = dict(nrays=1000, eE=3.0, eI=0.5, eEspread=0, eEpsilonX=0.263,
              eEpsilonZ=0.008, betaX=9., betaZ=2., period=18.5,
              xPrimeMax=0.03, zPrimeMax=0.03, targetOpenCL=None, distE='BW')

CASES = {
    'far_planar': (dict(n=108, K=0.52, eMin=3900, eMax=4250, gNodes=24,
                        gIntervals=2), 3000),
    'far_helical': (dict(n=60, Kx=0.9, Ky=0.9, phaseDeg=90, eMin=2300,
                         eMax=2700, gNodes=20, gIntervals=2), 3000),
    'taper': (dict(n=20, K=1.1, taper=(0.4, 10.), eMin=2600, eMax=3100,
                   gNodes=16, gIntervals=2), 1500),
    'nf': (dict(n=20, K=1.1, R0=25000., eMin=2600, eMax=3100, gNodes=16,
                gIntervals=2), 1500),
}


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    raycing._VERBOSITY_ = 0
    for tag, (kw, nr) in CASES.items():
        bl = raycing.BeamLine()
        args = dict(COMMON)
        args.update(kw)
        u = rs.Undulator(bl, 'u', **args)
        if u.needReset:
            u.reset()
        rng = np.random.RandomState(
            {'far_planar': 1, 'far_helical': 2, 'taper': 3, 'nf': 4}[tag])
        E = rng.uniform(args['eMin'], args['eMax'], nr)
        th = rng.uniform(-1, 1, nr) * 0.03e-3
        ps = rng.uniform(-1, 1, nr) * 0.03e-3
        # a few exactly on-axis / on-harmonic rays
        th[:3] = 0.
        ps[:2] = 0.
        w = E
        mode = un.MODE_TAPER if u._taperVal is not None else \
            un.MODE_NF if u.R0 is not None else un.MODE_FAR
        tab = dict(tg=u.tg, ag=u.ag, sintg=u.sintg, costg=u.costg,
                   sintgph=u.sintgph, costgph=u.costgph, dstep=u.dstep)
        mytab = un.node_tables(u.quadm, u.gIntervals, u.phase)
        for k in ('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph'):
            assert np.allclose(tab[k], mytab[k], rtol=0, atol=4e-16), (tag, k)
        # raw sums: the reference's own _sp_sum on the pre-factors it computes
        gamma, wu, ww1, ab = un.prefactors(
            u.Kx, u.Ky, u.Np, u.L0, u.gamma, w, th, ps, mode == un.MODE_FAR)
        R0v = None
        r0z = 0.
        if u.R0 is not None:
            R0v = np.array((np.tan(th), np.tan(ps), np.ones_like(ps)))
            r0z = u.R0 * np.pi * 2 / u.L0
            R0v *= r0z
        Is_ref, Ip_ref = u._sp_sum(ww1, w, wu, gamma, th, ps, R0v)
        I_ref, Es_ref, Ep_ref = u._build_I_map_conv(w, th, ps, None)
        Is, Ip = un.sp_sum(mode, u.Kx, u.Ky, u.Np, tab, ww1, w, wu, gamma, th,
                           ps, u._taperVal, r0z)
        I, Es, Ep = un.intensity_map(
            mode, u.Kx, u.Ky, u.Np, u.L0, u.gamma, u.eI, True, tab, w, th, ps,
            u._taperVal, u.R0)
        errs = [_rel(Is, Is_ref), _rel(Ip, Ip_ref) if np.abs(Ip_ref).max() > 0
                else np.abs(Ip).max(), _rel(I, I_ref), _rel(Es, Es_ref)]
        print(tag, 'mode', mode, 'nodes', len(u.tg), 'Np', u.Np,
              'restatement vs reference:', ['%.2e' % e for e in errs])
        assert max(errs) < 1e-12, (tag, errs)
        np.savez_compressed(
            os.path.join(OUT, 'g9_undulator_%s.npz' % tag),
            mode=np.int32(mode), Kx=np.float64(u.Kx), Ky=np.float64(u.Ky),
            Np=np.int32(u.Np), L0=np.float64(u.L0), gamma0=np.float64(u.gamma),
            eI=np.float64(u.eI), phase=np.float64(u.phase),
            quadm=np.int32(u.quadm), gIntervals=np.int32(u.gIntervals),
            taperVal=np.float64(np.nan if u._taperVal is None else u._taperVal),
            R0=np.float64(np.nan if u.R0 is None else u.R0),
            r0z=np.float64(r0z), dstep=np.float64(u.dstep),
            tg=u.tg, ag=u.ag, sintg=u.sintg, costg=u.costg,
            sintgph=u.sintgph, costgph=u.costgph,
            gamma=gamma, wu=wu, w=w, ww1=ww1, ddphi=th, ddpsi=ps, ab=ab,
            Is=Is_ref, Ip=Ip_ref, I=I_ref, Es=Es_ref, Ep=Ep_ref)


if __name__ == '__main__':
    main()
