"""One user-defined surface, written twice the way a user of either package would: as the numpy
methods of an OE subclass (what the reference runs; oracle/gen_fixtures_user_surface.py makes
golden g2_user_surface with them) and as the HIP source snippets of the same subclass here
(hip_local_z / hip_local_n / hip_plist, xrt_amd/usersurf.py). A paraboloid-like focusing mirror
with a cubic figure term and a twist:
    z = x^2 / (2 rs) + y^2 / (2 Rm) + k3 y^3 + kt x^2 y
Both spell out the same IEEE operations in the same order (products instead of powers, one
division per term), so the ray states agree bit for bit."""
import numpy as np

P, Q, PITCH = 20000., 10000., 4e-3
RM = 2 * P * Q / ((P + Q) * np.sin(PITCH))
RS = 2 * P * Q * np.sin(PITCH) / (P + Q)
K3, KT = 2e-12, 1.5e-9
LIMITS = dict(limPhysX=[-10, 10], limPhysY=[-300, 300])


def numpy_local_z(x, y, rs=RS, rm=RM, k3=K3, kt=KT):
    return x * x / (2 * rs) + y * y / (2 * rm) + k3 * y * y * y + kt * x * x * y


def numpy_local_n(x, y, rs=RS, rm=RM, k3=K3, kt=KT):
    a = -(x / rs + 2 * kt * x * y)
    b = -(y / rm + 3 * k3 * y * y + kt * x * x)
    norm = np.sqrt(a * a + b * b + 1)
    return [a / norm, b / norm, 1 / norm]


# p = (rs, Rm, k3, kt)
HIP_LOCAL_Z = '''
  return x * x / (2 * p[0]) + y * y / (2 * p[1]) + p[2] * y * y * y + p[3] * x * x * y;
'''
HIP_LOCAL_N = '''
  const double a = -(x / p[0] + 2 * p[3] * x * y);
  const double b = -(y / p[1] + 3 * p[2] * y * y + p[3] * x * x);
  const double norm = sqrt(a * a + b * b + 1);
  n[0] = a / norm;
  n[1] = b / norm;
  n[2] = 1 / norm;
'''


def subclass(roe):
    """The OE subclass in the package *roe* (the reference's oes module or xrt_amd's)."""
    class FiguredParaboloid(roe.OE):
        hip_local_z, hip_local_n = HIP_LOCAL_Z, HIP_LOCAL_N
        hip_plist = property(lambda self: (RS, RM, K3, KT))

        def local_z(self, x, y):
            return numpy_local_z(x, y)

        def local_n(self, x, y):
            return numpy_local_n(x, y)
    return FiguredParaboloid


# ---- a user-defined GRATING: a plane whose groove vector is a function of (x, y) -- a fan of
# lines with a quadratic density law, as the reference takes it from a subclass's local_g
# (oes/base.py:688-717; cl_local_g on its OpenCL path). p = (rho0, b1, b2, bx)
G_RHO0, G_B1, G_B2, G_BX = 300., 2.4e-4, -3.1e-8, 1.5e-4
G_LIMITS = dict(limPhysX=(-3, 3), limPhysY=(-45, 45))


def numpy_local_g(x, y, rho0=G_RHO0, b1=G_B1, b2=G_B2, bx=G_BX):
    return rho0 * bx * x, rho0 * (1 + b1 * y + b2 * y * y), x * 0.


HIP_FLAT_Z = 'return 0.;'
HIP_FLAT_N = 'n[0] = 0.; n[1] = 0.; n[2] = 1.;'
HIP_LOCAL_G = '''
  g[0] = p[0] * p[3] * x;
  g[1] = p[0] * (1 + p[1] * y + p[2] * y * y);
  g[2] = x * 0.;
'''


def grating_subclass(roe):
    class FanGrating(roe.OE):
        hip_local_z, hip_local_n, hip_local_g = HIP_FLAT_Z, HIP_FLAT_N, HIP_LOCAL_G
        hip_plist = (G_RHO0, G_B1, G_B2, G_BX)

        def local_g(self, x, y, rho=None):
            return numpy_local_g(x, y)
    return FanGrating


# ---- the same figured surface as a Bragg CRYSTAL (Si 111 at its Bragg angle for 9 keV): the
# atomic planes follow the surface, local_n's one normal serves as both (oes/reflect.py takes
# the last three components of whatever local_n returns)
X_CENTER, X_LIMITS = [0, 30000., 0], dict(limPhysX=[-4, 4], limPhysY=[-40, 40])


def crystal_element(roe, bl, si, pitch):
    return subclass(roe)(bl, 'figured crystal', center=X_CENTER, pitch=pitch, material=si,
                         **X_LIMITS)
