"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/g2_fzp_*.npz by RUNNING THE
REFERENCE (imported from /root/reference, build container only): a circular Fresnel
zone plate (NormalFZP, oes/gratings.py:10-137, material kind 'FZP') in ray mode:

  g2_fzp_first.npz   order +1, N from the thinnest zone, central zone opaque
  g2_fzp_orders.npz  a sequence of orders (1, 0, -1, 3) drawn per transmitted ray from
                     numpy's global generator (seeded), inverted zones

While generating, the numpy restatement (oracle/reflect_np.py: fzp_rays_good_gn and the
toWhere = 4 branch) is asserted against the reference, and so is its zone table.

Run:  python -m oracle.gen_fixtures_fzp
"""
import numpy as np

from . import _refenv
from . import gen_fixtures_p1 as g1
from . import reflect_np as rn

CASES = (
    ('g2_fzp_first', dict(f=5., E=700., thinnestZone=6e-5, order=1), 91, None),
    ('g2_fzp_orders', dict(f=8., E=720., N=300, isCentralZoneBlack=False,
                           order=(1, 0, -1, 3)), 92, 20260929),
)


def main():
    _refenv.activate()
    import xrt.backends.raycing as raycing
    import xrt.backends.raycing.sources as rs
    import xrt.backends.raycing.oes as roe
    import xrt.backends.raycing.materials as rm
    n = 2048
    for tag, kw, seed, np_seed in CASES:
        bl = raycing.BeamLine()
        mat = rm.Material('Au', rho=19.3, kind='FZP')
        fzp = roe.NormalFZP(bl, 'fzp', center=[0, 1000., 0], pitch=np.pi/2,
                            material=mat, **kw)
        half = fzp.rn[-1]
        beam = g1.make_rays(rs, n, seed, sx=half*0.55, sz=half*0.55, sa=2e-6, sc=2e-6,
                            E=(kw['E'] - 1., kw['E'] + 1.), amplitudes=True, pol='mixed')
        beam.state[3] = 3
        beam.state[4] = -4
        beam.x[6], beam.z[6] = fzp.rn[5], 0.         # exactly on a zone boundary
        beam.x[7], beam.z[7] = 0., 0.                # on the axis
        par = g1.oe_params(fzp, dict(kind='flat'))
        par['material'] = dict(kind='FZP')
        par['order'] = kw['order']
        par['fzp'] = rn.make_fzp(kw['f'], kw['E'], kw.get('N', 1000),
                                 kw.get('isCentralZoneBlack', True),
                                 kw.get('thinnestZone'))
        assert np.array_equal(par['fzp']['rn'], fzp.rn)
        extra = dict(fzp_f=np.array(float(kw['f'])), fzp_E=np.array(float(kw['E'])),
                     fzp_N=np.array(float(fzp.N)), fzp_rn=np.array(fzp.rn),
                     fzp_black=np.array(bool(fzp.isCentralZoneBlack)),
                     order=np.array(kw['order']))
        if 'thinnestZone' in kw:
            extra['fzp_thinnestZone'] = np.array(kw['thinnestZone'])
        if np_seed is not None:
            extra['np_seed'] = np_seed
        g1.run_reflect(tag, rs, fzp, par, beam, **extra)
    general_cases(raycing, rs, roe, rm)


# GeneralFZPin0YZ (gratings.py:140-313): zones from two foci; batch statistics inside
GENERAL = {
    'g2_gfzp_normal': dict(center=[0, 10., 0], pitch=np.pi/2, f1='inf', f2=(0, 0, 2.), E=400.,
                           N=100, phaseShift=np.pi),
    'g2_gfzp_grazing': dict(center=[0, 2000., 0], pitch=0.02,
                            f1=(0, -2000.*np.cos(0.02), 2000.*np.sin(0.02)),
                            f2=(0, 900.*np.cos(0.02), 900.*np.sin(0.02)), E=9000., N=200,
                            limPhysX=[-0.5, 0.5], limPhysY=[-12, 12]),
}


def general_cases(raycing, rs, roe, rm):
    for seed, (tag, kw) in enumerate(GENERAL.items()):
        bl = raycing.BeamLine()
        mat = rm.Material('Au', rho=19.3, kind='FZP')
        fzp = roe.GeneralFZPin0YZ(bl, 'gfzp', material=mat, **kw)
        if tag.endswith('normal'):
            beam = g1.make_rays(rs, 2048, 230 + seed, sx=0.02, sz=0.02, sa=1e-6, sc=1e-6,
                                E=(399., 401.), amplitudes=True, pol='mixed')
        else:
            beam = g1.make_rays(rs, 2048, 230 + seed, sx=0.05, sz=0.05, sa=2e-5, sc=2e-5,
                                E=(8999., 9001.), amplitudes=True, pol='mixed')
        beam.state[3] = 3
        beam.state[4] = -4
        par = g1.oe_params(fzp, dict(kind='flat'))
        par['material'] = dict(kind='FZP')
        par['order'] = 1
        par['gfzp'] = dict(f1=kw['f1'], f2=kw['f2'], lambdaE=fzp.lambdaE, N=fzp.N,
                           phaseShift=fzp.phaseShift, vorticity=fzp.vorticity,
                           grazingAngle=fzp.grazingAngle, minHalfLambda=None)
        g1.run_reflect(tag, rs, fzp, par, beam, gfzp_phaseShift=np.array(fzp.phaseShift),
                       order=np.array(1))
        assert par['gfzp']['minHalfLambda'] == fzp.minHalfLambda


if __name__ == '__main__':
    main()
