"""Synthetic workloads of the BASELINE configurations: element
definitions and seeded input generators shared by bench.py, the smoke test and
the parity tests. Inputs are generated on the host with numpy's default_rng."""
import numpy as np

from .backends import raycing
from .backends.raycing import materials as rm
from .backends.raycing import oes as roe
from .backends.raycing import sources as rs
from .backends.raycing.physconsts import CHBAR


def cfg2_toroid(bl=None):
    """BASELINE cfg2: toroid mirror + Pt, SURVEY 8d."""
    bl = bl or raycing.BeamLine()
    p, q, pitch = 20000., 10000., 4e-3
    m = rm.Material('Pt', rho=21.45, kind='mirror')
    return roe.ToroidMirror(bl, 'tm', center=[0, p, 0], pitch=pitch, R=(p, q),
                            r=(p, q), material=m, limPhysX=[-10, 10],
                            limPhysY=[-300, 300])


def cfg3_dcm(bl=None):
    """BASELINE cfg3: Si(111) double-crystal monochromator, SURVEY 8d."""
    bl = bl or raycing.BeamLine()
    si1 = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
    si2 = rm.CrystalSi(hkl=(1, 1, 1), tK=297.15)
    thB = float(np.ravel(si1.get_Bragg_angle(9000.) - si1.get_dtheta(9000.))[0])
    return roe.DCM(bl, 'dcm', center=[0, 20000., 0], bragg=thB, material=si1,
                   material2=si2, cryst2perpTransl=10., limPhysX=[-10, 10],
                   limPhysY=[-50, 50], limPhysX2=[-10, 10], limPhysY2=[-50, 150])


def synthetic_rays(n, seed, sa=2e-4, sc=2e-5, E=(8990., 9010.), amplitudes=False):
    """SURVEY 8d cfg2/cfg3 ray generator (numpy default_rng on the host)."""
    rng = np.random.default_rng(seed)
    b = rs.Beam(nrays=n, withAmplitudes=amplitudes)
    b.x = rng.normal(0, 0.1, n)
    b.z = rng.normal(0, 0.1, n)
    b.y = np.zeros(n)
    a = rng.normal(0, sa, n)
    c = rng.normal(0, sc, n)
    b.a = a
    b.c = c
    b.b = np.sqrt(1 - a**2 - c**2)
    b.E = rng.uniform(E[0], E[1], n)
    b.state = np.ones(n, dtype=np.int32)
    b.Jss = np.ones(n)
    b.Jpp = np.zeros(n)
    b.Jsp = np.zeros(n, dtype=complex)
    if amplitudes:
        b.Es = np.ones(n, dtype=complex)
        b.Ep = np.zeros(n, dtype=complex)
    return b


def kirchhoff_case(cfg):
    """cfg4 / cfg5 of SURVEY 8d: Gaussian-spherical field sampled uniformly on a
    0.2 x 0.2 mm slit 44 m from the source point, E = 7900 eV, Ep = 0, receiving
    mesh +-0.5 mm on a screen 10 m downstream, seed 7. Returns host arrays."""
    ns, side = {4: (1_000_000, 512), 5: (4_000_000, 2048)}[cfg]
    return kirchhoff_custom(ns, side)


def kirchhoff_custom(ns, side, seed=7):
    rng = np.random.default_rng(seed)
    sx = rng.uniform(-0.1, 0.1, ns)
    sz = rng.uniform(-0.1, 0.1, ns)
    sy = np.zeros(ns)
    R0 = 44000.
    nl = R0 / np.sqrt(sx**2 + R0**2 + sz**2)   # unit vector from the origin . (0,1,0)
    E = np.full(ns, 7900.)
    k = E / CHBAR * 1e7
    rho2 = sx**2 + sz**2
    Es = np.exp(-rho2 / 0.15**2) * np.exp(1j * k * rho2 / (2 * R0))
    Ep = np.zeros(ns, dtype=complex)
    mesh = np.linspace(-0.5, 0.5, side)
    px, pz = [a.ravel() for a in np.meshgrid(mesh, mesh)]
    py = np.full(px.shape, 10000.)
    return dict(ns=ns, side=side, px=px, py=py, pz=pz, sx=sx, sy=sy, sz=sz,
                n=[0, 1, 0], nl=nl, E=E, k=k, Es=Es, Ep=Ep)
