"""The OE.multiple_reflect cases, written once for both packages: the reference (imported by
oracle/gen_fixtures_multi.py in the build container, which makes the goldens g2_multi_*) and
xrt_amd (tests/test_multiple_reflect.py, tests/test_gpu_multiple_reflect.py).

  cylinder   the reference's own example (examples/withRaycing/10_MultipleReflect/
             Cylinder.py:23-39, 81-88, 96): a meridional cylinder Rm = 5 m, 190 mm long, at 3 mrad,
             a point source 1 m upstream -- whispering-gallery bounces, 3 to 6 per ray. The
             surface is the user's: numpy methods there, HIP snippets here.
  toroid     ToroidMirror at grazing incidence, 2-7 bounces, with the elevation map
  edges      the toroid with optical limits, a short mirror, dead and 'out' rays in the
             incoming beam, rays that miss, and maxReflections = 3 cutting the loop short
  flat       a flat mirror: the second bounce finds nothing (the loop ends by exhaustion)
  capillary  an ellipsoidal capillary (parametric surface of revolution, closed): rays
             spiralling down the bore
"""
import numpy as np

CYL_RM, CYL_L = 5000., 190.
CYL = dict(center=[0, 1000, -0.05], pitch=3e-3, limPhysX=[-5, 5], limPhysY=[0, CYL_L])


def numpy_cyl_z(x, y, Rm=CYL_RM):
    return Rm - np.sqrt(Rm**2 - y**2)


def numpy_cyl_n(x, y, Rm=CYL_RM):
    a = np.zeros_like(x)
    b = -y * (Rm**2 - y**2)**(-0.5)
    c = 1.
    norm = (b**2 + 1)**0.5
    b /= norm
    c /= norm
    return [a, b, c]


# p = (Rm,). numpy's x**(-0.5) is 1 / sqrt(x) here (npy_pow's fast path for -0.5 does not exist:
# it calls pow(); pow(x, -0.5) is correctly rounded in this range to within an ulp of
# 1 / sqrt(x) -- the states are compared bit for bit, the geometry at 1e-12)
HIP_CYL_Z = 'return p[0] - sqrt(p[0] * p[0] - y * y);'
HIP_CYL_N = '''
  double b = -y * pow(p[0] * p[0] - y * y, -0.5);
  double c = 1.;
  const double norm = sqrt(b * b + 1);
  n[0] = 0.;
  n[1] = b / norm;
  n[2] = c / norm;
'''


def cylinder_subclass(roe):
    class Cylinder(roe.OE):
        hip_local_z, hip_local_n = HIP_CYL_Z, HIP_CYL_N
        hip_plist = (CYL_RM,)

        def local_z(self, x, y):
            return numpy_cyl_z(x, y)

        def local_n(self, x, y):
            return numpy_cyl_n(x, y)
    return Cylinder


TOROID = dict(center=[0, 1000, -0.05], pitch=3e-3, limPhysX=[-5, 5], limPhysY=[0, 190.],
              R=5000., r=50.)
EDGES = dict(center=[0, 1000, -0.04], pitch=2.5e-3, roll=0.02, yaw=1e-3,
             limPhysX=[-3, 3], limPhysY=[-20, 150.], limOptX=[-2, 2], limOptY=[-10, 120],
             R=4000., r=40.)


def edges_on(bl):
    """EDGES for a beamline with an azimuth: the centre 1 m down THAT beamline."""
    kw = dict(EDGES)
    x, y, z = kw['center']
    kw['center'] = [bl.cosAzimuth*x + bl.sinAzimuth*y, -bl.sinAzimuth*x + bl.cosAzimuth*y, z]
    return kw


# an ellipsoidal capillary lit from 300 mm upstream of its middle (far from the ellipsoid's
# focus): the rays spiral down the bore, up to four bounces
CAPILLARY = dict(center=[0, 300., 0], limPhysY=[-200, 200], ellipseA=1000., ellipseB=0.5,
                 workingDistance=100.)
FLAT = dict(center=[0, 1000, 0], pitch=4e-3, limPhysX=[-5, 5], limPhysY=[-100, 100.])


def point_source_rays(rs, n, seed, dxprime=5e-4, dzprime=1e-5, E=2000., amplitudes=True,
                      spread_E=0.):
    """What the example's GeometricSource makes (a point source, normal angular
    distributions), from a seeded generator of its own."""
    rng = np.random.default_rng(seed)
    b = rs.Beam(nrays=n, withAmplitudes=amplitudes)
    b.x[:] = 0.
    b.y[:] = 0.
    b.z[:] = 0.
    b.a[:] = rng.normal(0, dxprime, n)
    b.c[:] = rng.normal(0, dzprime, n)
    b.b[:] = np.sqrt(1 - b.a**2 - b.c**2)
    b.E[:] = E + spread_E * rng.uniform(-1, 1, n)
    b.state[:] = 1
    ang = rng.uniform(0, np.pi, n)
    ph = rng.uniform(-np.pi, np.pi, n)
    es = np.cos(ang)
    ep = np.sin(ang) * np.exp(1j*ph)
    b.Jss[:] = es*es
    b.Jpp[:] = (ep*np.conj(ep)).real
    b.Jsp[:] = es*np.conj(ep)
    if amplitudes:
        b.Es[:] = es
        b.Ep[:] = ep
    return b


def edge_rays(rs, n, seed):
    b = point_source_rays(rs, n, seed, dxprime=1.2e-3, dzprime=4e-5, E=9000., spread_E=10.)
    b.c[:64] = np.linspace(-3e-4, 3e-4, 64)        # under / over the mirror
    b.a[64:128] = np.linspace(-4e-3, 4e-3, 64)     # off the sides
    b.b[:] = np.sqrt(1 - b.a**2 - b.c**2)
    b.state[200:204] = (2, 3, -1, 0)
    b.state[300] = 2
    return b
