"""Diagnostic: the drawn multiple_reflect case of tests/test_gpu_multiple_reflect.py
(test_random_toroids_against_the_oracle) for one seed -- which rays differ from the oracle and how.
    PYTHONPATH=.:tests python tools/diag_multi_seed.py SEED [exact]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
import multi_cases as case
import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.materials as rm
import xrt_amd.backends.raycing.oes as roe
import xrt_amd.backends.raycing.sources as rs
from oracle import reflect_np as rn
from oracle.adapters import oracle_params, to_oracle_beam

seed = int(sys.argv[1])
if len(sys.argv) > 2:
    os.environ['XRT_HIP_MULTI_FORM'] = sys.argv[2]
rng = np.random.default_rng(4400 + seed)
bl = raycing.BeamLine(azimuth=float(rng.choice([0., 0.25])), height=0)
for _ in range(int(rng.integers(0, 3))):
    roe.OE(bl, 'before')
mat = [rm.Material('Au', rho=19.3, kind='mirror'), rm.Material('Pt', rho=21.45, kind='mirror'),
       rm.Material('Rh', rho=12.41, kind='mirror')][int(rng.integers(3))]
length = float(rng.uniform(120., 260.))
x, y, z = 0., 1000., -float(rng.uniform(0.03, 0.07))
kw = dict(center=[bl.cosAzimuth * x + bl.sinAzimuth * y, -bl.sinAzimuth * x + bl.cosAzimuth * y, z],
          pitch=float(rng.uniform(2.2e-3, 4e-3)), limPhysX=[-5, 5], limPhysY=[0, length],
          R=float(rng.uniform(3000., 9000.)), r=float(rng.uniform(30., 90.)))
if rng.random() < 0.4:
    kw.update(roll=float(rng.normal(0, 0.01)), yaw=float(rng.normal(0, 5e-4)))
if rng.random() < 0.4:
    kw.update(limOptX=[-3, 3], limOptY=[5., 0.8 * length])
print(kw, bl.azimuth)
oe = roe.ToroidMirror(bl, 'gallery', material=mat, **kw)
n = int(rng.choice([700, 5000]))
src = case.point_source_rays(rs, n, 900 + seed, dxprime=float(rng.uniform(2e-4, 1e-3)),
                             dzprime=float(rng.uniform(5e-6, 3e-5)),
                             E=float(rng.uniform(2000., 12000.)), spread_E=5.,
                             amplitudes=bool(rng.random() < 0.5))
if bl.azimuth:
    for u, v in (('x', 'y'), ('a', 'b')):
        p, q = getattr(src, u).copy(), getattr(src, v).copy()
        pu, qv = raycing.rotate_z(p, q, bl.cosAzimuth, -bl.sinAzimuth)
        getattr(src, u)[:] = pu
        getattr(src, v)[:] = qv
src.state[rng.random(n) < 0.02] = 2
src.state[rng.random(n) < 0.02] = -1
most = int(rng.choice([2, 4, 100]))
elevation = bool(rng.random() < 0.5)
print('n', n, 'most', most, 'elevation', elevation)
ob = to_oracle_beam(src)
oinfo = []
mgb, mlbN = rn.oe_multiple_reflect(oracle_params(oe), ob.copy(), most, elevation, info=oinfo)
info = []
gb, lbN = oe.multiple_reflect(rs.Beam(copyFrom=src), maxReflections=most,
                              needElevationMap=elevation, _info=info)
print('bounces', lbN.nrays // n, len(mlbN.x) // n)
bad = np.nonzero(gb.nRefl != mgb.nRefl)[0]
print('rays whose nRefl differs:', len(bad), bad[:10])
for i in bad[:2]:
    print(i, 'nRefl', gb.nRefl[i], mgb.nRefl[i], 'state', gb.state[i], mgb.state[i],
          'dx', gb.x[i] - mgb.x[i], 'dy', gb.y[i] - mgb.y[i], 'dz', gb.z[i] - mgb.z[i])
    k = lbN.nrays // n
    for b in range(k):
        print('   bounce', b, 'state', lbN.state[b * n + i], mlbN.state[b * n + i],
              'y', lbN.y[b * n + i], mlbN.y[b * n + i], 'x', lbN.x[b * n + i], mlbN.x[b * n + i])
gb2, lbN2 = oe.multiple_reflect(rs.Beam(copyFrom=src), maxReflections=most, needElevationMap=elevation)
print('second call differs from the oracle in', int((gb2.nRefl != mgb.nRefl).sum()), 'rays; from the first in',
      int((gb2.nRefl != gb.nRefl).sum()))
print('oracle:')
for one in oinfo:
    print({k: one[k] for k in one if k in ('brent', 'left', 'entering', 'tangency')})
print('here:')
for one in info:
    print({k: one[k] for k in one if k in ('brent', 'left', 'entering')}, one.get('tangency'))

# Is the bounce the oracle's, given THIS run's footprints of the bounce before? (the oracle's
# single bounce -- reflect_local(isMulti=True) -- started from the footprints b - 1 made here)
p = oracle_params(oe)
k = lbN.nrays // n
for b in range(1, k):
    ob1 = rn.Beam(n, with_amplitudes=hasattr(lbN, 'Es'))
    for f in ob1.fields():
        setattr(ob1, f, np.array(getattr(lbN, f))[(b - 1) * n:b * n])
    ob1.nRefl = np.array(lbN.nRefl)[(b - 1) * n:b * n]
    if elevation:
        for f in ('elevationD', 'elevationX', 'elevationY', 'elevationZ'):
            setattr(ob1, f, np.array(getattr(lbN, f))[(b - 1) * n:b * n])
    good = (ob1.state == 1) | (ob1.state == 2)
    rn.reflect_local(p, good, ob1, ob1, p['pitch'], p['roll'] + p['positionRoll'], p['yaw'],
                     p.get('dx', 0), material=p.get('material'), needElevationMap=elevation,
                     isMulti=True)
    mine = np.array(lbN.state)[b * n:(b + 1) * n]
    diff = np.nonzero(ob1.state != mine)[0]
    print('bounce %d from the footprints of bounce %d made here: the oracle\'s states differ in %d rays %s'
          % (b, b - 1, len(diff), [(int(i), int(ob1.state[i]), int(mine[i])) for i in diff[:4]]))
