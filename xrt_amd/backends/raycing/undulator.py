"""``Undulator`` — host-side mirror of the reference's analytic undulator source
(xrt/backends/raycing/sources/synchr.py:1349-2260 on top of
sources/sybase.py:29-560, 933-1810) for the cases its three OpenCL kernels
cover: far field, tapered gap, near field (``R0``).

What runs where: the sampling logic (numpy RNG in the reference's call order,
rejection, electron-beam convolution, coherency matrix) is host code exactly as
in the reference; the field integral ``build_I_map`` — the only expensive
step — is ONE HIP kernel launch (``xrt_hip_undulator_imap_f64_dev``: pre-factors,
node sum, scaling). There is no numpy implementation of the integral here: without
the GPU library ``build_I_map`` raises.

Not mirrored (raise NotImplementedError / absent): custom magnetic field
(``SourceFromField``: fieldsource.py), the Qt flow hooks. The mesh functions
(``intensities_on_mesh``, ``multi_electron_stack``, ``tuning_curves``, ``power_vs_K``) are in
meshes.py.
"""
import threading

import numpy as np
import torch
from scipy import special

from .. import raycing
from ... import hipcalls
from .physconsts import (PI, PI2, C, EV2ERG, CHeVcm, CHBAR, M0, K2B, E2WC, SIE0,
                         SQ2, SQPI)
from .sources import Beam
from .meshes import MeshFunctions

_TABLE_LOCK = threading.Lock()

UND_FAR, UND_TAPER, UND_NF = 0, 1, 2


def clenshaw_curtis(n):
    """n-point Clenshaw–Curtis nodes and weights on [-1, 1] (the reference's
    default rule, sybase.py:1106-1140): w_k = c_k/N · Σ_j g_j cos(2πjk/N),
    g_0 = 1, g_j = −b_j/(4j²−1), evaluated with one length-N FFT."""
    N = n - 1
    x = -np.cos(np.pi * np.arange(n) / N)
    if n == 2:
        return x, np.array([1.0, 1.0])
    half = N // 2
    j = np.arange(1, half + 1)
    g = -2. / (4. * j * j - 1.)
    G = np.zeros(N)
    G[0] = 1.
    if N % 2 == 0:
        g[-1] *= 0.5            # b_{N/2} = 1
        G[j[:-1]] += 0.5 * g[:-1]
        G[N - j[:-1]] += 0.5 * g[:-1]
        G[half] += g[-1]
    else:
        G[j] += 0.5 * g
        G[N - j] += 0.5 * g
    S = np.fft.fft(G).real
    S = np.append(S, S[0])
    c = np.full(n, 2.)
    c[0] = c[-1] = 1.
    return x, c / N * S


class Undulator(MeshFunctions):
    def __init__(self, bl=None, name='GenericSource', center=(0, 0, 0),
                 nrays=raycing.nrays, eE=6.0, eI=0.1, eEspread=0., eSigmaX=None,
                 eSigmaZ=None, eEpsilonX=1., eEpsilonZ=0.01, betaX=9., betaZ=2.,
                 eMin=9000, eMax=9100, distE='eV', xPrimeMax=0.01, zPrimeMax=0.01,
                 R0=None, uniformRayDensity=False, filamentBeam=False, pitch=0,
                 yaw=0, period=50, n=50, K=1, Kx=0, Ky=0, B0x=0, B0y=0, phaseDeg=0,
                 taper=None, targetE=None, xPrimeMaxAutoReduce=True,
                 zPrimeMaxAutoReduce=True, gp=1e-6, gIntervals=2, gNodes=None,
                 targetOpenCL='auto', precisionOpenCL='auto', device=None, eN=51, nx=25,
                 nz=25, **kwargs):
        """Arguments as the reference's ``Undulator`` (synchr.py:1362-1435,
        sybase.py:34-190, 940-1030); *targetOpenCL*/*precisionOpenCL* are
        accepted and ignored (the integral always runs in fp64 on the GPU);
        *device*: torch device for the kernel (default: current)."""
        given = dict(locals())
        if kwargs:
            raise NotImplementedError('unsupported Undulator arguments: %s'
                                      % sorted(kwargs))
        raycing.enrol(self, bl, 'sources', 0, name, 'Source')
        if bl is not None:
            bl.oesDict[self.uuid][1] = 0          # a source, not an optical element
        for key in ('name', 'center', 'R0', 'distE', 'uniformRayDensity', 'filamentBeam',
                    'eEspread', 'gp', 'device', 'phaseDeg'):
            setattr(self, key, given[key])
        for key in ('eE', 'eI', 'eMin', 'eMax'):
            setattr(self, key, float(given[key]))
        self.eN, self.nx, self.nz = eN, nx, nz         # the default meshes of meshes.py
        self.pitch, self.yaw = (raycing.auto_units_angle(v) for v in (pitch, yaw))
        self.nrays = np.int64(int(nrays))
        # Lorentz factor of the electrons, and its square
        self.gamma = self.eE * 1e9 * EV2ERG / (M0 * C**2)
        self.gamma2 = self.gamma * self.gamma
        self._set_electron_beam(eSigmaX, eSigmaZ, eEpsilonX, eEpsilonZ, betaX, betaZ)
        self._xPrimeMin, self._xPrimeMax = self._angular_range(xPrimeMax)
        self._zPrimeMin, self._zPrimeMax = self._angular_range(zPrimeMax)
        # quadrature: a given number of nodes per interval, or searched for at reset()
        self.gIntervals = int(gIntervals)
        self.needConvergence = gNodes is None
        self.quadm = 0 if gNodes is None else int(gNodes)
        self.maxIntegrationNodes = int(6e5)
        self.convergenceSearchFlag = self._useGauLeg = False
        # the magnet: period [mm], number of periods, gap taper, phase between the fields
        self.L0, self.Np = period, n
        self._set_taper(taper)
        self.phase = np.radians(phaseDeg)
        self.targetE = None
        if targetE is not None:
            self._set_targetE(targetE)           # K from the wanted harmonic energy
        elif Kx != 0 or Ky != 0:
            self.Kx, self.Ky = float(Kx), float(Ky)
        elif abs(K) > 0:
            self.Kx, self.Ky = 0., float(K)
        elif B0x != 0 or B0y != 0:               # peak fields [T]
            self.Kx, self.Ky = (float(b) * self.L0 / K2B for b in (B0x, B0y))
        else:
            raise ValueError("Please define either K or B0!")
        # a near-field calculation always narrows the angular range to the radiation cone
        self.xPrimeMaxAutoReduce = xPrimeMaxAutoReduce or R0 is not None
        self.zPrimeMaxAutoReduce = zPrimeMaxAutoReduce or R0 is not None
        self.report_E1()
        self.needReset = True

    # ---- parameter bookkeeping (host, mirrors the property setters) ---------
    def _set_electron_beam(self, eSigmaX, eSigmaZ, epsX, epsZ, betaX, betaZ):
        """dx, dz [mm], dxprime, dzprime [rad] from emittance [nm rad] and beta
        [m] or explicit sizes [um] (sybase.py:160-175, 219-348)."""
        out = []
        for sig, eps, beta in ((eSigmaX, epsX, betaX), (eSigmaZ, epsZ, betaZ)):
            eps = None if eps is None else eps * 1e-6
            if sig is not None:
                d = sig * 1e-3
            elif eps is not None and beta is not None:
                d = np.sqrt(eps * (beta * 1e3))
            else:
                d = 0
            dp = (eps / d if d > 0 else 0) if eps is not None else 0
            out += [d, dp]
        self.dx, self.dxprime, self.dz, self.dzprime = out

    @staticmethod
    def _angular_range(v):
        """mrad input -> (min, max) in rad (sybase.py:393-409)."""
        if isinstance(v, (tuple, list)):
            lim = [raycing.auto_units_angle(v[0], defaultFactor=1e-3),
                   raycing.auto_units_angle(v[-1], defaultFactor=1e-3)]
            return min(lim), max(lim)
        if isinstance(v, str):
            m = abs(raycing.auto_units_angle(v))
            return -m, m
        m = abs(v) * 1e-3
        return -m, m

    def _set_taper(self, taper):
        """synchr.py:1566-1594."""
        self.taper, self._taperVal = taper, None
        if taper is not None and np.ndim(taper) == 0:
            self._taperVal = float(taper)
        elif taper is not None:
            t = np.asarray(taper, dtype=float).ravel()
            if len(t) == 1:
                dgap, gap = 0., t[0]
            elif len(t) == 2:
                dgap, gap = t
            else:
                raise ValueError('taper must be (dgap, gap)')
            self.gap, self._taperVal = gap, \
                (None if dgap == 0 else dgap / self.Np / self.L0 / gap)

    def _set_targetE(self, targetE):
        """(energy, harmonic[, isElliptical]) -> Kx, Ky (synchr.py:1499-1560)."""
        energy, harmonic = float(targetE[0]), float(targetE[1])
        Ky = np.sqrt(harmonic * 8 * PI * self.gamma2 / self.L0 / energy / E2WC - 2)
        Kx = 0
        if Ky != Ky:                   # below the harmonic's lowest energy
            raise ValueError('Cannot calculate K, try to increase the undulator '
                             'harmonic number')
        if len(targetE) > 2 and targetE[2]:
            if isinstance(targetE[2], float):
                Kx = Ky * np.cos(targetE[2])
                Ky = Ky * np.sin(targetE[2])
            else:
                Kx = Ky = Ky / 2**0.5          # helical
        self.targetE, self.Kx, self.Ky = targetE, Kx, Ky

    @property
    def K(self):
        return self.Ky

    @property
    def B0x(self):
        return K2B * self.Kx / self.L0

    @property
    def B0y(self):
        return K2B * self.Ky / self.L0

    def report_E1(self):
        """First-harmonic energy (synchr.py:1658-1674)."""
        g2, kk = self.gamma2, (0.5*self.Kx**2, 0.5*self.Ky**2)
        wu = PI / self.L0 / g2 * (2*g2 - 1. - kk[0] - kk[1]) / E2WC
        self.E1 = 2*wu*g2 / (1 + kk[0] + kk[1])
        return self.E1

    def _prime_max_mrad(self, lo, hi, reduce_by):
        """What the reference's xPrimeMax / zPrimeMax *getters* return
        (sybase.py:369-392, 410-434): mrad, reduced to K/gamma if asked."""
        if reduce_by is not None:
            tmp = reduce_by / self.gamma
            if abs(hi) > abs(tmp):
                hi_new = tmp
            else:
                hi_new = hi
            if abs(lo) > abs(tmp):
                lo = np.sign(lo) * tmp
            hi = hi_new
        if abs(lo) == abs(hi):
            return hi * 1e3
        return [lo * 1e3, hi * 1e3]

    def _reset_limits(self):
        """sybase.py:479-515. (The reference's z auto-reduction tests for an
        attribute `_gamma` that never exists, sybase.py:417, so only the
        horizontal range is ever reduced; same here.)"""
        kx = (self.Ky if abs(self.Ky) > 0 else 2.) if self.xPrimeMaxAutoReduce \
            else None
        xp = self._prime_max_mrad(self._xPrimeMin, self._xPrimeMax, kx)
        zp = self._prime_max_mrad(self._zPrimeMin, self._zPrimeMax, None)
        lims = []
        for raw, v in ((self._xPrimeMax, xp), (self._zPrimeMax, zp)):
            if not raw:
                lims.append((-1e-3, 1e-3))
            elif isinstance(v, (tuple, list)):
                lims.append((v[0] * 1e-3, v[-1] * 1e-3))
            else:
                lims.append((-v * 1e-3, v * 1e-3))
        (xpMin, xpMax), (zpMin, zpMax) = lims
        self.Theta_min = float(xpMin - self.dxprime)
        self.Theta_max = float(xpMax + self.dxprime)
        self.Psi_min = float(zpMin - self.dzprime)
        self.Psi_max = float(zpMax + self.dzprime)
        self.E_min, self.E_max = (float(pick(self.eMin, self.eMax)) for pick in (min, max))
        # steps of the default meshes (meshes.py): eN energies, 2 nx by 2 nz angles
        for step, lo, hi, count in (('dE', self.E_min, self.E_max, self.eN),
                                    ('dTheta', self.Theta_min, self.Theta_max, 2*self.nx),
                                    ('dPsi', self.Psi_min, self.Psi_max, 2*self.nz)):
            setattr(self, step, (hi - lo) / float(count))

    # ---- integration grid ----------------------------------------------------
    def _build_integration_grid(self):
        """Node tables of one period, uploaded once per grid
        (synchr.py:1787-1801)."""
        rule = np.polynomial.legendre.leggauss if self._useGauLeg else \
            clenshaw_curtis
        tg_n, ag_n = rule(self.quadm)
        self.dstep = dstep = 2 * PI / float(self.gIntervals)
        mid = np.arange(-PI + 0.5 * dstep, PI, dstep)           # interval centres
        self.tg = (mid[:, None] + 0.5 * dstep * tg_n).ravel()
        self.ag = (mid[:, None] * 0 + ag_n).ravel()
        shifted = self.tg + self.phase                          # the horizontal field's phase
        self.sintg, self.costg = np.sin(self.tg), np.cos(self.tg)
        self.sintgph, self.costgph = np.sin(shifted), np.cos(shifted)
        self._tables = {}            # per device, uploaded on first use there

    def _device_tables(self):
        # (run_ray_tracing(threads=N) puts its workers on different GPUs: one copy per device,
        # filled under a lock)
        dev = self._device()
        key = str(dev)
        with _TABLE_LOCK:
            tables = self._tables.get(key) if isinstance(self._tables, dict) else None
            if tables is None:
                if not isinstance(self._tables, dict):
                    self._tables = {}
                tables = self._tables[key] = [
                    torch.from_numpy(np.ascontiguousarray(t)).to(dev)
                    for t in (self.tg, self.ag, self.sintg, self.costg, self.sintgph,
                              self.costgph)]
        return tables

    def _device(self):
        if self.device is not None:
            return torch.device(self.device)
        from ... import _lib
        _lib.require_gpu()
        return torch.device('cuda', torch.cuda.current_device())

    def _get_mad(self):
        """Median spread of the on-edge intensity over 5 neighbouring grid
        sizes (sybase.py:1245-1285)."""
        keep = self.quadm
        k = self.quadm - 2
        sE = self.E_max * np.ones(1)
        sT = self.Theta_max * np.ones(1)
        sP = self.Psi_max * np.ones(1)
        vals, dvals = [], []
        old = None
        for m in range(5):
            k += 1
            self.quadm = k
            self._build_integration_grid()
            new = self.build_I_map(sE, sT, sP)[0]
            if m > 0:
                vals.append(new)
                dvals.append(np.abs(new - old) / new)
            old = new
        v = np.abs(np.array(vals))
        mad = np.median(np.abs(v - np.median(v)))
        dimad = np.median(dvals)
        self.quadm = keep
        return mad, dimad

    def _find_convergence_mixed(self):
        """Doubling then bisection on the number of nodes until the on-edge
        intensity is stable to *gp* (sybase.py:1190-1243)."""
        def stable(nodes):
            self.quadm = nodes
            spread, drift = self._get_mad()
            return drift < self.gp or spread < self.gp
        power = 4
        while not stable(int(2**power)) and self.quadm <= self.maxIntegrationNodes and \
                power < 10000:
            power += 1
        lo, hi = int(2**(power - 1)), self.quadm
        for _ in range(int(np.log2((hi - lo) / 20.))):
            half = int(0.5 * (hi + lo))
            if stable(half):
                hi = half
            else:
                lo = half
        self.quadm = hi

    def _reset_integration_grid(self):
        """sybase.py:1452-1467."""
        if self.needConvergence:
            # searched without energy spread, on the bare field of one probe ray
            spread, self.eEspread, self.quadm = self.eEspread, 0, 0
            self.convergenceSearchFlag = True
            try:
                self._find_convergence_mixed()
            finally:
                self.convergenceSearchFlag = False
                self.eEspread = spread
        self._build_integration_grid()

    def reset(self):
        """Limits, quadrature grid and -- for a filament beam -- the intensity ceiling of
        the rejection sampling, estimated from one batch of uniformly drawn (E, theta, psi)
        without energy spread (three uniform draws, then one rand for the acceptance
        count: the reference's order, sybase.py:521-547)."""
        self.needReset = False
        self._reset_limits()
        self._reset_integration_grid()
        self.Imax = 0.
        if self.filamentBeam and not hasattr(self, 'dimExy'):
            count = self.nrays
            trial = [np.random.uniform(lo, hi, count) for lo, hi in (
                (self.E_min, self.E_max), (self.Theta_min, self.Theta_max),
                (self.Psi_min, self.Psi_max))]
            spread, self.eEspread = self.eEspread, 0
            intensity = self.build_I_map(*trial)[0]
            self.eEspread = spread
            self.Imax = np.max(intensity) * 1.2
            accepted = np.where(self.Imax * np.random.rand(count) < intensity)[0]
            self.nrepmax = np.floor(count / len(accepted))
        self.xzE = (self.E_max - self.E_min) * (self.Theta_max - self.Theta_min) * \
            (self.Psi_max - self.Psi_min)
        self.fluxConst = self.Imax * self.xzE

    # ---- the field integral: one HIP launch ---------------------------------
    @property
    def mode(self):
        return UND_TAPER if self._taperVal is not None else \
            UND_NF if self.R0 is not None else UND_FAR

    def build_I_map_device(self, w, ddtheta, ddpsi, harmonic=None, gamma=None):
        """(I, Es, Ep) as device tensors for device (or host) arrays of photon
        energy [eV] and observation angles [rad]; *gamma*: optional per-ray
        electron gamma (energy spread)."""
        if not hasattr(self, '_tables') and self.needReset:
            self.reset()
        tables = self._device_tables()
        dev = tables[0].device

        def up(a):
            if isinstance(a, torch.Tensor):
                return a.to(dev, torch.float64).contiguous()
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)

        mode = self.mode
        return hipcalls.undulator_imap(
            mode, self.Kx, self.Ky, tables, up(w), up(ddtheta), up(ddpsi),
            self.L0, self.Np, self.gamma, self.eI, self.dstep, self.distE == 'BW',
            gamma=None if gamma is None else up(gamma), harmonic=harmonic,
            alpha_s=self._taperVal / E2WC if mode == UND_TAPER else 0.,
            r0z=self.R0 * np.pi * 2 / self.L0 if mode == UND_NF else 0.)

    def build_I_map(self, w, ddtheta, ddpsi, harmonic=None, dg=None):
        """Host-array form with the reference's return values
        (synchr.py:2035-2108): (I, Es, Ep) numpy arrays, or the bare
        |field|·dstep/2 during the convergence search."""
        w = np.atleast_1d(np.asarray(w, dtype=float))
        n = len(w)
        th = np.atleast_1d(np.asarray(ddtheta, dtype=float)) * np.ones(n)
        ps = np.atleast_1d(np.asarray(ddpsi, dtype=float)) * np.ones(n)
        gamma = None
        if self.eEspread > 0:
            g = self.gamma
            if dg is not None:
                g = g + dg
            else:
                sz = 1 if self.filamentBeam else n
                g = g + g * self.eEspread * np.random.normal(size=sz)
            gamma = g * np.ones(n)
        if self.convergenceSearchFlag:
            return self._bare_field(w, th, ps, gamma)
        if harmonic is not None and np.ndim(harmonic):
            # a harmonic per ray (mesh functions): one launch per distinct value
            harmonic = np.asarray(harmonic) * np.ones(n)
            out = (np.zeros(n), np.zeros(n, dtype=complex), np.zeros(n, dtype=complex))
            for h in np.unique(harmonic):
                sel = harmonic == h
                part = self.build_I_map_device(w[sel], th[sel], ps[sel], float(h),
                                               None if gamma is None else gamma[sel])
                for whole, piece in zip(out, part):
                    whole[sel] = piece.cpu().numpy()
            return out
        I, Es, Ep = self.build_I_map_device(w, th, ps, harmonic, gamma)
        return I.cpu().numpy(), Es.cpu().numpy(), Ep.cpu().numpy()

    def _bare_field(self, w, th, ps, gamma):
        """|Is|,|Ip| -> sqrt(|Is|²+|Ip|²)·dstep/2 (synchr.py:2103-2104), from the
        raw device sums."""
        g = self.gamma * np.ones(len(w)) if gamma is None else gamma
        g2 = g**2
        wu = PI / self.L0 / g2 * np.ones_like(w) * \
            (2*g2 - 1 - 0.5*self.Kx**2 - 0.5*self.Ky**2) / E2WC
        ww1 = w * ((1. + 0.5*self.Kx**2 + 0.5*self.Ky**2) +
                   g2 * (th**2 + ps**2)) / (2. * g2 * wu)
        tables = self._device_tables()
        dev = tables[0].device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        mode = self.mode
        Is, Ip = hipcalls.undulator(
            mode, self.Kx, self.Ky, tables, up(g), up(wu), up(w), up(ww1),
            up(th), up(ps), nper=self.Np if mode else 1,
            alpha_s=self._taperVal / E2WC if mode == UND_TAPER else 0.,
            r0z=self.R0 * np.pi * 2 / self.L0 if mode == UND_NF else 0.)
        f = np.abs(Is.cpu().numpy())**2 + np.abs(Ip.cpu().numpy())**2
        return np.abs(np.sqrt(f) * 0.5 * self.dstep)

    # ---- photon source size (synchr.py:2200-2260, sybase.py:664-674) --------
    def get_sigma_r02(self, E):
        return 2 * CHeVcm/E*10 * self.L0*self.Np / PI2**2

    def get_sigmaP_r02(self, E):
        return CHeVcm/E*10 / (2 * self.L0*self.Np)

    @staticmethod
    def tanaka_kitamura_Qa2(x, eps=1e-6):
        """Tanaka & Kitamura's energy-spread factor Q_a(x)^2 (1 below *eps*)."""
        x = np.asarray(x, dtype=float)
        out = np.ones_like(x)
        big = x > eps
        y = SQ2 * x[big]
        y2 = y**2
        out[big] = y2 / (np.exp(-y2) + SQPI*y*special.erf(y) - 1)
        return out

    def _harmonic_of(self, E, onlyOddHarmonics):
        harmonic = np.floor_divide(E, self.E1)
        if onlyOddHarmonics:
            harmonic += harmonic % 2 - 1
        return harmonic

    def get_sigma_r2(self, E, onlyOddHarmonics=True, with0eSpread=False):
        s = self.get_sigma_r02(E)
        if self.eEspread == 0 or with0eSpread:
            return s
        norm = PI2 * self._harmonic_of(E, onlyOddHarmonics) * self.Np * self.eEspread
        return s * self.tanaka_kitamura_Qa2(norm/4.)**(2/3.)

    def get_sigmaP_r2(self, E, onlyOddHarmonics=True, with0eSpread=False):
        s = self.get_sigmaP_r02(E)
        if self.eEspread == 0 or with0eSpread:
            return s
        norm = PI2 * self._harmonic_of(E, onlyOddHarmonics) * self.Np * self.eEspread
        return s * self.tanaka_kitamura_Qa2(norm)

    def get_SIGMA(self, E, onlyOddHarmonics=True, with0eSpread=False):
        s = self.get_sigma_r2(E, onlyOddHarmonics, with0eSpread)
        return (self.dx**2 + s)**0.5, (self.dz**2 + s)**0.5

    def get_SIGMAP(self, E, onlyOddHarmonics=True, with0eSpread=False):
        s = self.get_sigmaP_r2(E, onlyOddHarmonics, with0eSpread)
        return (self.dxprime**2 + s)**0.5, (self.dzprime**2 + s)**0.5

    # ---- shine ----------------------------------------------------------------
    # The numpy RNG is the only shared state with the reference: for one seed the
    # same rays must come out. The order of draws (documented per helper) is the
    # contract; it follows sybase.py:1470-1810.
    def _draw_filament(self, accuBeam):
        """One electron of the beam: (E, x, z, x', z', dgamma, seeded, seededI).
        Draws: 1 uniform, 4 normals (+1 normal with energy spread)."""
        if accuBeam is not None:
            return dict(E=accuBeam.E[0], x=accuBeam.filamentDX, z=accuBeam.filamentDZ,
                        xp=accuBeam.filamentDtheta, zp=accuBeam.filamentDpsi,
                        dgamma=accuBeam.filamentDgamma, seeded=accuBeam.seeded,
                        seededI=accuBeam.seededI)
        el = dict(seeded=np.int64(0), seededI=0., dgamma=None)
        el['E'] = np.random.random_sample() * float(self.E_max - self.E_min) + \
            self.E_min
        el['x'] = self.dx * np.random.standard_normal()
        el['z'] = self.dz * np.random.standard_normal()
        el['xp'] = self.dxprime * np.random.standard_normal()
        el['zp'] = self.dzprime * np.random.standard_normal()
        if self.eEspread > 0:
            el['dgamma'] = self.gamma * self.eEspread * np.random.standard_normal()
        return el

    def _draw_observation(self, n, el, fixedEnergy, wave):
        """Photon energies and observation angles of one batch. Draws: energies
        (unless filament / fixed energy); for a wave the electron offsets and
        divergences (4 normal arrays, each only if its sigma > 0); for rays the
        two uniform angle arrays."""
        if el is not None or fixedEnergy:
            E = (fixedEnergy if fixedEnergy else el['E']) * np.ones(n)
        else:
            E = np.random.rand(n) * float(self.E_max - self.E_min) + self.E_min
        if wave is None:
            theta = np.random.rand(n) * (self.Theta_max - self.Theta_min) + \
                self.Theta_min
            psi = np.random.rand(n) * (self.Psi_max - self.Psi_min) + self.Psi_min
            return E, theta, psi
        self.xzE = (self.E_max - self.E_min)
        if el is not None:
            sx, sz = el['x'], el['z']
        else:
            sx = np.random.normal(0, self.dx, n) if self.dx > 0 else 0
            sz = np.random.normal(0, self.dz, n) if self.dz > 0 else 0
        x = wave.xDiffr + sx
        y = wave.yDiffr
        z = wave.zDiffr + sz
        r = np.sqrt((x**2 + y**2 + z**2))
        theta = x / r
        psi = z / r
        if el is not None:
            theta += el['xp']
            psi += el['zp']
        else:
            if self.dxprime > 0:
                theta += np.random.normal(0, self.dxprime, n)
            if self.dzprime > 0:
                psi += np.random.normal(0, self.dzprime, n)
        return E, theta, psi

    def _accept(self, intensity, n):
        """Rejection sampling against the running maximum (1 uniform array); a
        slice for uniform ray density."""
        top = np.max(intensity)
        if top > self.Imax:
            self.Imax = top
            self.fluxConst = self.Imax * self.xzE
        if self.uniformRayDensity:
            return slice(None), n
        keep = np.where(self.Imax * np.random.rand(n) < intensity)[0]
        return keep, len(keep)

    def _source_points(self, bot, npassed, el):
        """Emission points: the electron position or the photon source size
        convolved with the electron beam (2 normal arrays)."""
        if el is not None:
            return el['x'], el['z']
        bot.sourceSIGMAx, bot.sourceSIGMAz = self.get_SIGMA(
            bot.E, onlyOddHarmonics=False)
        return (np.random.normal(0, bot.sourceSIGMAx, npassed),
                np.random.normal(0, bot.sourceSIGMAz, npassed))

    def _set_polarisation(self, bot, fs, fp, withAmplitudes):
        """Coherency matrix (normalised per ray unless uniform ray density) and
        amplitudes from the two field components."""
        s2 = (fs * np.conj(fs)).real
        p2 = (fp * np.conj(fp)).real
        tot = 1. if self.uniformRayDensity else s2 + p2
        with np.errstate(invalid='ignore', divide='ignore'):
            bot.Jsp[:] = np.where(tot, fs * np.conj(fp) / tot, 0)
            bot.Jss[:] = np.where(tot, s2 / tot, 0)
            bot.Jpp[:] = np.where(tot, p2 / tot, 0)
            if withAmplitudes:
                if self.uniformRayDensity:
                    bot.Es[:] = fs
                    bot.Ep[:] = fp
                else:
                    bot.Es[:] = fs / s2**0.5
                    bot.Ep[:] = fp / p2**0.5

    def _map_on_device(self):
        """Does this class implement ``build_I_map_device`` (Undulator does; subclasses that
        replace it with a refusal say so with ``_no_device_map``)?"""
        return not getattr(type(self).build_I_map_device, '_no_device_map', False)

    def shine(self, toGlobal=True, withAmplitudes=True, fixedEnergy=False,
              wave=None, accuBeam=None):
        """The source beam (rays sampled by rejection on the intensity map) or,
        with *wave* (a beam from ``prepare_wave``), the undulator field on the
        wave's points (reference: sybase.py:1470-1810)."""
        self._ready_to_shine()
        if wave is not None:
            if 'rDiffr' not in wave.array_fields():
                raise ValueError("If you want to use a `wave`, run a "
                                 "`prepare_wave` before shine!")
            self.uniformRayDensity = True
            # (the device path needs this class's map on the device: SourceFromField and the
            # bending magnets keep theirs on the host and take the host path below -- ADVICE r3)
            if all(k in wave._d for k in ('xDiffr', 'yDiffr', 'zDiffr')) and \
                    not (self.pitch or self.yaw) and self._map_on_device():
                return self._shine_wave_on_device(wave, toGlobal, fixedEnergy, accuBeam)
        batch = len(wave.a) if wave is not None else self.nrays
        if self.uniformRayDensity:
            withAmplitudes = True
        el = self._draw_filament(accuBeam) if self.filamentBeam else None
        seeded = el['seeded'] if el is not None else np.int64(0)
        seededI = el['seededI'] if el is not None else 0.
        parts, length, nrep = [], 0, 0
        while True:
            seeded += batch
            E, theta, psi = self._draw_observation(batch, el, fixedEnergy, wave)
            intensity, fs, fp = self.build_I_map(
                E, theta, psi, dg=None if el is None else el['dgamma'])
            if self.uniformRayDensity:
                seededI += batch * self.xzE
                sourceWeight = self.xzE
            else:
                seededI += intensity.sum() * self.xzE
                sourceWeight = seededI / seeded
            keep, npassed = self._accept(intensity, batch)
            if npassed == 0:
                continue
            bot = wave if wave is not None else \
                Beam(npassed, withAmplitudes=withAmplitudes)
            bot.state[:] = 1
            bot.E[:] = E[keep]
            px, pz = self._source_points(bot, npassed, el)
            fs, fp = fs[keep], fp[keep]
            if wave is not None:
                wave.rDiffr = np.sqrt(((wave.xDiffr - px)**2 + wave.yDiffr**2 +
                                       (wave.zDiffr - pz)**2))
                wave.path[:] = 0
                wave.a[:] = (wave.xDiffr - px) / wave.rDiffr
                wave.b[:] = wave.yDiffr / wave.rDiffr
                wave.c[:] = (wave.zDiffr - pz) / wave.rDiffr
                area = wave.areaNormal if hasattr(wave, 'areaNormal') else wave.area
                spread = area**0.5 / wave.rDiffr     # field per sample area
                fs *= spread
                fp *= spread
            else:
                bot.x[:] = px
                bot.z[:] = pz
                bot.a[:] = theta[keep]
                bot.c[:] = psi[keep]
                if el is not None:
                    bot.a[:] += el['xp']
                    bot.c[:] += el['zp']
                else:                       # electron divergence: 2 normal arrays
                    if self.dxprime > 0:
                        bot.a[:] += np.random.normal(0, self.dxprime, npassed)
                    if self.dzprime > 0:
                        bot.c[:] += np.random.normal(0, self.dzprime, npassed)
            self._set_polarisation(bot, fs, fp, withAmplitudes)
            parts.append(bot)
            length += npassed
            if self.uniformRayDensity:
                break
            if self.filamentBeam:
                nrep += 1
                if nrep >= self.nrepmax:
                    break
            elif length >= self.nrays:
                break

        bo = parts[0] if len(parts) == 1 else _concatenate(parts, withAmplitudes)
        self._book_flux(bo, length, seeded, seededI,
                        sourceWeight / (self.nrays if wave is None else len(wave.a)))
        if length > self.nrays and not self.filamentBeam and wave is None:
            bo.filter_by_index(slice(0, int(self.nrays)))
        if el is not None:
            bo.filamentDtheta, bo.filamentDpsi = el['xp'], el['zp']
            bo.filamentDX, bo.filamentDZ = el['x'], el['z']
            bo.filamentDgamma = el['dgamma']
        self._unit_directions(bo, (bo.a**2 + bo.b**2 + bo.c**2)**0.5)
        out = Beam(copyFrom=bo)
        if wave is not None:
            out.x[:] = px
            out.y[:] = 0.
            out.z[:] = pz
            if self.R0 is None:     # far field: carry the spherical-wave phase
                out.path[:] = 0.
                phase = np.exp(1e7j * wave.E/CHBAR * wave.rDiffr)
                wave.Es *= phase
                wave.Ep *= phase
        out.parentId = self.uuid
        if toGlobal:
            raycing.virgin_local_to_global(self.bl, out, self.center)
        return out


def _shine_wave_on_device(self, wave, toGlobal, fixedEnergy, accuBeam):
    """``shine(wave=...)`` for a wave whose samples live on the GPU (what the apertures' and
    elements' ``prepare_wave`` make): the random numbers are drawn on the host in the
    reference's order and uploaded, everything else -- observation angles, the field map,
    distances and directions, coherency matrix, the spherical phase, the source beam -- is
    the host form's arithmetic as tensor operations in the same order (+, -, *, / and sqrt
    round the same way on both sides; sybase.py:1470-1810)."""
    dev = wave._d['xDiffr'].device
    n = wave.nrays
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
    full = lambda v: torch.full((n,), float(v), dtype=torch.float64, device=dev)  # noqa: E731
    el = self._draw_filament(accuBeam) if self.filamentBeam else None
    seeded = (el['seeded'] if el is not None else np.int64(0)) + n
    seededI = el['seededI'] if el is not None else 0.
    # -- _draw_observation
    if el is not None or fixedEnergy:
        E_host = (fixedEnergy if fixedEnergy else el['E']) * np.ones(n)
        E = full(E_host[0])
    else:
        E_host = np.random.rand(n) * float(self.E_max - self.E_min) + self.E_min
        E = up(E_host)
    self.xzE = (self.E_max - self.E_min)
    xD, yD, zD = (wave._d[k] for k in ('xDiffr', 'yDiffr', 'zDiffr'))
    if el is not None:
        sx, sz = el['x'], el['z']
    else:
        sx = up(np.random.normal(0, self.dx, n)) if self.dx > 0 else 0
        sz = up(np.random.normal(0, self.dz, n)) if self.dz > 0 else 0
    x, z = xD + sx, zD + sz
    r = torch.sqrt((x * x + yD * yD) + z * z)
    theta, psi = x / r, z / r
    if el is not None:
        theta, psi = theta + el['xp'], psi + el['zp']
    else:
        if self.dxprime > 0:
            theta = theta + up(np.random.normal(0, self.dxprime, n))
        if self.dzprime > 0:
            psi = psi + up(np.random.normal(0, self.dzprime, n))
    # -- build_I_map (its energy-spread draw comes here in the order of calls)
    gamma = None
    if self.eEspread > 0:
        g = self.gamma
        if el is not None and el['dgamma'] is not None:
            g = g + el['dgamma']
        else:
            g = g + g * self.eEspread * np.random.normal(size=1 if self.filamentBeam else n)
        gamma = up(g * np.ones(n))
    intensity, fs, fp = self.build_I_map_device(E, theta, psi, None, gamma)
    seededI += n * self.xzE
    top = float(intensity.max())
    if top > self.Imax:
        self.Imax = top
        self.fluxConst = self.Imax * self.xzE
    # -- the wave's samples as seen from the emission points
    wave.state = torch.ones(n, dtype=torch.int32, device=dev)
    wave.E = E
    if el is not None:
        px, pz = el['x'], el['z']
    else:
        wave.sourceSIGMAx, wave.sourceSIGMAz = self.get_SIGMA(E_host, onlyOddHarmonics=False)
        px = up(np.random.normal(0, wave.sourceSIGMAx, n))
        pz = up(np.random.normal(0, wave.sourceSIGMAz, n))
    dx_, dz_ = xD - px, zD - pz
    dist = torch.sqrt((dx_ * dx_ + yD * yD) + dz_ * dz_)
    wave.rDiffr = dist
    wave.path = torch.zeros(n, dtype=torch.float64, device=dev)
    a, b, c = dx_ / dist, yD / dist, dz_ / dist
    area = wave.areaNormal if hasattr(wave, 'areaNormal') else wave.area
    spread = area**0.5 / dist                # field per sample area
    fs, fp = fs * spread, fp * spread
    # -- _set_polarisation with uniform ray density (no normalisation per ray)
    wave.Jsp = fs * torch.conj(fp)
    wave.Jss = (fs * torch.conj(fs)).real.contiguous()
    wave.Jpp = (fp * torch.conj(fp)).real.contiguous()
    wave.Es, wave.Ep = fs, fp
    self._book_flux(wave, n, seeded, seededI, self.xzE / n, energy_sum=E_host.sum())
    if el is not None:
        wave.filamentDtheta, wave.filamentDpsi = el['xp'], el['zp']
        wave.filamentDX, wave.filamentDZ = el['x'], el['z']
        wave.filamentDgamma = el['dgamma']
    length = torch.sqrt((a * a + b * b) + c * c)
    wave.a, wave.b, wave.c = a / length, b / length, c / length
    out = Beam(copyFrom=wave)
    out.x = px if isinstance(px, torch.Tensor) else full(px)
    out.y = torch.zeros(n, dtype=torch.float64, device=dev)
    out.z = pz if isinstance(pz, torch.Tensor) else full(pz)
    if self.R0 is None:     # far field: carry the spherical-wave phase
        out.path = torch.zeros(n, dtype=torch.float64, device=dev)
        # numpy's exp(1e7j E / CHBAR r): the argument is ((1e7 E) (1 / CHBAR)) r in its complex
        # arithmetic, exp of a pure imaginary number is (cos, sin)
        phi = ((1e7 * E) * (1. / CHBAR)) * dist
        phase = torch.complex(torch.cos(phi), torch.sin(phi))
        wave.Es, wave.Ep = fs * phase, fp * phase
    out.parentId = self.uuid
    if toGlobal:
        undo = [(2, self.bl.cosAzimuth, -self.bl.sinAzimuth)] if self.bl.sinAzimuth != 0 else []
        raycing.turn([out._d['a'], out._d['b'], out._d['c']], undo)
        position = raycing.turn([out._d['x'], out._d['y'], out._d['z']], undo)
        for coordinate, c0 in zip(position, self.center):
            coordinate += c0
    return out


def _ready_to_shine(self):
    """Pending reset; the beamline's alignment energy defaults to the middle of the range."""
    if self.needReset:
        self.reset()
    if self.bl is not None:
        try:
            self.bl._alignE = float(self.bl.alignE)
        except (ValueError, AttributeError, TypeError):
            self.bl._alignE = 0.5 * (self.eMin + self.eMax)


def _book_flux(self, bo, length, seeded, seededI, weight, energy_sum=None):
    """What the beam says about the flux it represents (the rays accepted of those seeded)."""
    bo.accepted, bo.acceptedE = (length * self.fluxConst,
                                 (bo.E.sum() if energy_sum is None else energy_sum) *
                                 self.fluxConst * SIE0)
    bo.seeded, bo.seededI, bo.sourceWeight = seeded, seededI, weight


def _unit_directions(self, bo, length):
    """Direction cosines from the (a, b, c) drawn so far, turned by the source's pitch / yaw."""
    for comp in (bo.a, bo.b, bo.c):
        comp /= length
    if self.pitch or self.yaw:
        raycing.rotate_beam(bo, pitch=self.pitch, yaw=self.yaw)


Undulator._shine_wave_on_device = _shine_wave_on_device
Undulator._ready_to_shine = _ready_to_shine
Undulator._book_flux = _book_flux
Undulator._unit_directions = _unit_directions


def _concatenate(parts, withAmplitudes):
    """All batches of one shine() in one Beam (the reference's
    Beam.concatenate, beams.py:230-252, does not carry Es/Ep along; here they
    are kept consistent with the other fields)."""
    n = sum(len(p.x) for p in parts)
    bo = Beam(n, withAmplitudes=withAmplitudes)
    fields = ['state', 'x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp']
    if withAmplitudes:
        fields += ['Es', 'Ep']
    for f in fields:
        setattr(bo, f, np.concatenate([getattr(p, f) for p in parts]))
    last = parts[-1]
    for k in ('sourceSIGMAx', 'sourceSIGMAz'):
        if k in parts[0].__dict__:
            object.__setattr__(bo, k, parts[0].__dict__[k])
    del last
    return bo
