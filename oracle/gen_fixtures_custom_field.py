"""TEST INFRASTRUCTURE ONLY — regenerates tests/golden/g12_custom_field_*.npz by
RUNNING THE REFERENCE (imported from /root/reference, build container only).

G12 (custom field): inputs and outputs of
SourceFromField._sp_sum (synchr.py:888-973) as called from
_build_I_map_custom_field_conv (:1274-1346) for a tabulated 10-period vertical
field with tapered ends:
  far        electron-energy-normalised tables (non-filament), far field
  filament   filamentBeam=True
  nf         near field, R0 = 20 m
Node tables (field, velocity and trajectory on the integration grid, built by
the reference's own trajectory integration) are stored as the kernel inputs.

Run:  python -m oracle.gen_fixtures_custom_field
"""
import os
import numpy as np
from . import _refenv
from . import undulator_np as un

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    raycing._VERBOSITY_ = 0
    L0, Np = 30., 10
    z = np.linspace(-L0*Np/2-40, L0*Np/2+40, 2000)
    env = 0.5*(np.tanh((z + L0*Np/2)/8.) - np.tanh((z - L0*Np/2)/8.))
    field = np.vstack((z, 0.6*np.sin(2*np.pi*z/L0)*env,
                       0.05*np.cos(2*np.pi*z/L0)*env)).T     # z, B_hor?, B_ver
    field = np.vstack((z, 0.08*np.cos(2*np.pi*z/L0)*env,
                       0.6*np.sin(2*np.pi*z/L0)*env, 0*z)).T
    for tag, kw in (('far', {}), ('filament', dict(filamentBeam=True)),
                    ('nf', dict(R0=20000.))):
        bl = raycing.BeamLine()
        s = rs.SourceFromField(
            bl, 'sff', nrays=500, eE=3.0, eI=0.5, eEspread=0, eEpsilonX=0.263,
            eEpsilonZ=0.008, betaX=9., betaZ=2., eMin=1500, eMax=1700,
            xPrimeMax=0.1, zPrimeMax=0.1, targetOpenCL=None, distE='BW',
            customField=field, gNodes=40, gIntervals=20, **kw)
        np.random.seed(7)
        if s.needReset:
            s.reset()
        rng = np.random.RandomState(3)
        n = 1500
        w = rng.uniform(1500, 1700, n)
        th = rng.uniform(-1e-4, 1e-4, n)
        ps = rng.uniform(-1e-4, 1e-4, n)
        th[:2] = 0.
        ps[:1] = 0.
        cap = {}
        orig = s._sp_sum

        def spy(emcg, w_, gamma, ddphi, ddpsi, Bx, By, Bz, betax, betay, betam,
                trajx, trajy, trajz, R0=None):
            res = orig(emcg, w_, gamma, ddphi, ddpsi, Bx, By, Bz, betax, betay,
                       betam, trajx, trajy, trajz, R0)
            cap.update(emcg=np.array(emcg), gamma=np.array(gamma), Bx=np.array(Bx),
                       By=np.array(By), Bz=np.array(Bz), betax=np.array(betax),
                       betay=np.array(betay), betam=np.float64(betam),
                       trajx=np.array(trajx), trajy=np.array(trajy),
                       trajz=np.array(trajz), Is=np.array(res[0]),
                       Ip=np.array(res[1]))
            return res
        s._sp_sum = spy
        I, Es, Ep = s.build_I_map(w, th, ps)
        tab = dict(tg=s.tg, ag=s.ag, **{k: cap[k] for k in (
            'Bx', 'By', 'Bz', 'betax', 'betay', 'trajx', 'trajy', 'trajz')})
        Is, Ip = un.custom_sp_sum(bool(s.filamentBeam), tab, cap['emcg'], w,
                                  cap['gamma'], th, ps, float(cap['betam']),
                                  s.R0)
        errs = [np.linalg.norm(a - b) / np.linalg.norm(b)
                for a, b in ((Is, cap['Is']), (Ip, cap['Ip']))]
        print(tag, 'nodes', len(s.tg), 'restatement vs reference',
              ['%.1e' % e for e in errs], '|Ip|/|Is| %.2e' % (
                  np.linalg.norm(cap['Ip']) / np.linalg.norm(cap['Is'])))
        assert max(errs) < 1e-12
        np.savez_compressed(
            os.path.join(OUT, 'g12_custom_field_%s.npz' % tag),
            filament=np.int32(bool(s.filamentBeam)),
            R0=np.float64(np.nan if s.R0 is None else s.R0),
            dstep=np.float64(s.dstep), w=w, ddphi=th, ddpsi=ps, I=I, Es=Es, Ep=Ep,
            tg=s.tg, ag=s.ag, **cap)


if __name__ == '__main__':
    main()
