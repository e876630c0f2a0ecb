"""``BendingMagnet`` and ``Wiggler`` sources (reference:
xrt/backends/raycing/sources/synchr.py:69-610 on the source base, sources/sybase.py:29-560).

The reference computes these on the host only ("reasonably fast and thus a GPU is not
required"); here the intensity / amplitude map of a batch — the Schwinger formula with
K_{1/3} and K_{2/3} per (E, theta, psi) — is one HIP launch (``xrt_hip_bend_imap_f64_dev``),
so that the product has no numpy physics of its own, and the sampling around it is host
code that draws from numpy's global generator in the reference's order: a seed gives the
reference's rays.
"""
import numpy as np
import torch

from .. import raycing
from ... import hipcalls
from .physconsts import C, CHeVcm, E0, EV2ERG, K2B, M0, PI, PI2
from .sources import Beam
from .undulator import Undulator, _concatenate


class BendingMagnet(Undulator):
    isMPW = False

    def __init__(self, *args, **kwargs):
        """*B0* [T] or *rho* [m] (bending radius) define the magnet; the other arguments
        are the electron-beam, energy-range and angular-range arguments of the source
        base (see ``Undulator``)."""
        field, radius = kwargs.pop('B0', 1.), kwargs.pop('rho', None)
        self._base_init(args, kwargs)
        self.Np = 0.5
        self.B, self.ro = field, radius
        if self.ro:
            if not self.B:
                self.B = self._bend(self.ro)
        elif self.B:
            self.ro = self._bend(self.B)

    def _base_init(self, args, kwargs, K=1., period=50, n=50):
        kwargs.update(K=K, period=period, n=n, gNodes=1, xPrimeMaxAutoReduce=False,
                      zPrimeMaxAutoReduce=False)
        Undulator.__init__(self, *args, **kwargs)
        for undulator_only in ('Kx', 'Ky', 'L0', 'Np', 'phase', 'targetE', 'quadm',
                               'gIntervals'):
            self.__dict__.pop(undulator_only, None)
        self.xPrimeMaxAutoReduce = self.zPrimeMaxAutoReduce = False

    def _bend(self, other):
        """Bending radius [m] for a field [T] or the field for a radius: B rho = m c^2
        gamma / e (the division by *other* sits where the reference has it: rho enters the
        ray positions bit for bit)."""
        return M0 * C**2 * self.gamma / other / E0 / 1e6

    B0 = property(lambda self: self.B)
    rho = property(lambda self: self.ro)

    def report_E1(self):
        pass

    def _reset_limits(self):
        self.Kx, self.Ky = 0., getattr(self, '_K', 0.)
        try:
            Undulator._reset_limits(self)
        finally:
            del self.Kx, self.Ky

    def _reset_integration_grid(self):
        pass

    # ---- the map: one launch ----------------------------------------------------------
    def build_I_map(self, dde, ddtheta, ddpsi, harmonic=None, dg=None):
        """(I, Es, Ep) per ray (synchr.py:185-227); with energy spread every call of more
        than one ray draws one normal array for the electrons' gamma."""
        if self.needReset:
            self.reset()
        E = np.atleast_1d(np.asarray(dde, dtype=float))
        n = len(E)
        gamma = None
        if self.eEspread > 0 and np.ndim(dde) and n > 1:
            gamma = self.gamma + np.random.normal(0, self.gamma*self.eEspread, E.shape)
        dev = self._device()

        def up(a):
            a = np.array(np.broadcast_to(np.asarray(a, dtype=float), (n,)), order='C')
            return torch.from_numpy(a).to(dev)
        I, Es, Ep = hipcalls.bend_imap(
            up(E), up(ddtheta), up(ddpsi), self.gamma, self.B, self.eI, poles=2 * self.Np,
            K=getattr(self, '_K', 0.), wiggler=self.isMPW,
            per_bandwidth=self.distE == 'BW', gamma=None if gamma is None else up(gamma))
        return I.cpu().numpy(), Es.cpu().numpy(), Ep.cpu().numpy()

    def build_I_map_device(self, *args, **kwargs):
        raise NotImplementedError('use build_I_map')
    build_I_map_device._no_device_map = True

    # ---- sampling -----------------------------------------------------------------------
    def _filament_electron(self, accuBeam):
        """The one electron of a filament beam: energy, emission point, angular offsets.
        Draws (no accuBeam): E uniform; wiggler: theta0 uniform, pole (integer),
        x normal, z normal; magnet: z normal, theta0 uniform, radius normal; then the two
        angular offsets."""
        if accuBeam is not None:
            return dict(E=accuBeam.E[0], x=accuBeam.x[0], y=accuBeam.y[0], z=accuBeam.z[0],
                        dtheta=accuBeam.filamentDtheta, dpsi=accuBeam.filamentDpsi,
                        theta0=getattr(accuBeam, 'filamentTheta0', None))
        el = dict(E=np.random.random_sample() * float(self.E_max - self.E_min) + self.E_min)
        span = self.Theta_max - self.Theta_min
        if self.isMPW:
            el['theta0'] = np.random.random_sample() * span + self.Theta_min
            lean = np.clip(el['theta0'] * self.gamma / self.K, -1., 1.)
            along = 0.5 * self.L0 * (np.arccos(lean) / PI) + \
                0.5 * self.L0 * np.random.randint(0, int(2*self.Np - 1) + 1)   # = random_integers
            y = along - 0.5*self.L0*self.Np
            if along - 0.25*self.L0 <= 0:
                y += self.L0*self.Np
            el['x'] = self.X0 * np.sin(PI2 * y / self.L0) + \
                self.dx * np.random.standard_normal()
            el['y'] = y - 0.25 * self.L0
            el['z'] = self.dz * np.random.standard_normal()
        else:
            el['z'] = self.dz * np.random.standard_normal()
            el['theta0'] = np.random.random_sample() * span + self.Theta_min
            radius = self.dx * np.random.standard_normal() + self.ro * 1000.
            el['x'] = -radius * np.cos(el['theta0']) + self.ro*1000.
            el['y'] = radius * np.sin(el['theta0'])
        el['dtheta'] = self.dxprime * np.random.standard_normal()
        el['dpsi'] = self.dzprime * np.random.standard_normal()
        return el

    def _emission_points(self, bot, theta0, count):
        """Positions of the rays of a non-filament batch on the electron's arc.
        Draws: z normal (if dz > 0), radius normal (if dx > 0)."""
        if self.dz > 0:
            bot.z[:] = np.random.normal(0., self.dz, count)
        radius = np.random.normal(self.ro*1e3, self.dx, count) if self.dx > 0 else self.ro * 1e3
        bot.x[:] = -radius * np.cos(theta0) + self.ro*1000.
        bot.y[:] = radius * np.sin(theta0)

    def shine(self, toGlobal=True, withAmplitudes=True, fixedEnergy=False, accuBeam=None):
        """The source beam: rejection sampling of (E, theta, psi) on the intensity map,
        batches of 1.2 nrays until nrays are accepted (reference synchr.py:229-508)."""
        self._ready_to_shine()
        if self.uniformRayDensity:
            withAmplitudes = True
        batch = self.nrays if self.uniformRayDensity else np.int64(self.nrays * 1.2)
        el = self._filament_electron(accuBeam) if self.filamentBeam else None
        energy = fixedEnergy if (fixedEnergy and el is not None) else \
            (el['E'] if el is not None else None)
        parts, length, seeded, seededI, nrep = [], 0, np.int64(0), 0., 0
        while True:
            draw = np.random.rand(batch, 4)      # energy, theta, psi, acceptance
            seeded += batch
            if el is not None:
                lo = np.max((self.Theta_min, el['theta0'] - 1. / self.gamma))
                hi = np.min((self.Theta_max, el['theta0'] + 1. / self.gamma))
                theta = draw[:, 1] * (hi - lo) + lo
                E = energy * np.ones(batch)
            else:
                E = draw[:, 0] * float(self.E_max - self.E_min) + self.E_min
                theta = draw[:, 1] * (self.Theta_max - self.Theta_min) + self.Theta_min
            psi = draw[:, 2] * (self.Psi_max - self.Psi_min) + self.Psi_min
            intensity, fs, fp = self.build_I_map(E, theta, psi)
            if self.uniformRayDensity:
                seededI += self.nrays * self.xzE
                sourceWeight = self.xzE
            else:
                seededI += intensity.sum() * self.xzE
                sourceWeight = seededI / seeded
            top = np.max(intensity)
            if top > self.Imax:
                self.Imax = top
                self.fluxConst = self.Imax * self.xzE
            if self.uniformRayDensity:
                keep, count = slice(None), batch
            else:
                keep = np.where(self.Imax * draw[:, 3] < intensity)[0]
                count = len(keep)
            if count == 0:
                continue
            bot = Beam(count, withAmplitudes=withAmplitudes)
            bot.state[:] = 1
            bot.E[:] = E[keep]
            theta0, psi0 = theta[keep], psi[keep]
            if el is not None:
                dtheta, dpsi = el['dtheta'], el['dpsi']
            else:   # electron divergence (+ the 1/gamma cone of a magnet): normal arrays
                dtheta = np.random.normal(0, self.dxprime, count) if self.dxprime > 0 else 0
                if not self.isMPW:
                    dtheta += np.random.normal(0, 1/self.gamma, count)
                dpsi = np.random.normal(0, self.dzprime, count) if self.dzprime > 0 else 0
            bot.a[:] = np.tan(theta0 + dtheta)
            bot.c[:] = np.tan(psi0 + dpsi)
            fs, fp = fs[keep], fp[keep]
            s2, p2 = (fs * np.conj(fs)).real, (fp * np.conj(fp)).real
            total = 1. if self.uniformRayDensity else s2 + p2
            if el is not None:
                bot.x[:], bot.y[:], bot.z[:] = el['x'], el['y'], el['z']
            if self.isMPW:
                self._wiggler_points(bot, theta0, count, el)
                bot.Jsp[:] = np.zeros(count)
            else:
                if el is None:
                    self._emission_points(bot, theta0, count)
                with np.errstate(invalid='ignore', divide='ignore'):
                    bot.Jsp[:] = np.array(np.where(total, fs * np.conj(fp) / total, total),
                                          dtype=complex)
            with np.errstate(invalid='ignore', divide='ignore'):
                bot.Jss[:] = np.where(total, s2 / total, total)
                bot.Jpp[:] = np.where(total, p2 / total, total)
            if withAmplitudes:
                bot.Es[:] = fs
                bot.Ep[:] = fp
            parts.append(bot)
            length += count
            if self.uniformRayDensity:
                break
            if el is not None:
                nrep += 1
                if nrep >= self.nrepmax:
                    break
            elif length >= self.nrays:
                break
        bo = parts[0] if len(parts) == 1 else _concatenate(parts, withAmplitudes)
        for name in ('sourceSIGMAx', 'sourceSIGMAz'):
            if hasattr(parts[-1], name) and not hasattr(bo, name):
                setattr(bo, name, getattr(parts[-1], name))
        if length >= self.nrays:      # (a filament beam may deliver fewer: no flux figures)
            self._book_flux(bo, length, seeded, seededI, sourceWeight / self.nrays)
        if length > self.nrays and el is None:
            bo.filter_by_index(slice(0, int(self.nrays)))
        if el is not None:
            bo.filamentDtheta, bo.filamentDpsi = el['dtheta'], el['dpsi']
            bo.filamentTheta0 = el['theta0']
        self._unit_directions(bo, np.sqrt(bo.a**2 + 1.0 + bo.c**2))    # (a, 1, c) = tangents
        bo.parentId = self.uuid
        if toGlobal:
            raycing.virgin_local_to_global(self.bl, bo, self.center)
        return bo

    def _wiggler_points(self, bot, theta0, count, el):
        raise NotImplementedError


class Wiggler(BendingMagnet):
    """Multipole wiggler: *K*, *period* [mm], *n* periods; the field follows from K, the
    critical energy varies with the horizontal angle, the source points lie on the
    sinusoidal orbit."""
    isMPW = True

    def __init__(self, *args, **kwargs):
        K, period, n = kwargs.pop('K', 8.446), kwargs.pop('period', 50), kwargs.pop('n', 40)
        kwargs.setdefault('name', 'wiggler')
        self._base_init(args, kwargs, K=K, period=period, n=n)
        self._K, self.L0, self.Np = K, period, n
        self.B = K2B * K / self.L0
        self.ro = self._bend(self.B)
        self.X0 = 0.5 * K * self.L0 / self.gamma / PI
        self.xPrimeMaxAutoReduce = True      # to the K / gamma fan

    @property
    def K(self):
        return self._K

    @K.setter
    def K(self, value):
        """As the reference's setter (synchr.py:568-576): K, the orbit amplitude and the
        radius follow; the field B that the critical energy is computed from stays."""
        self._K = float(value)
        self.ro = self._bend(self.B)
        self.X0 = 0.5 * value * self.L0 / self.gamma / PI
        self.needReset = True

    def power_vs_K(self, energy, theta, psi, Ks):
        """Total power [W] through the aperture per K of *Ks* (synchr.py:581-609)."""
        energy = np.asarray(energy)
        volume = (theta[1] - theta[0]) * (psi[1] - psi[0]) * (energy[1] - energy[0]) \
            if np.ndim(theta) else 1
        keep, powers = self.K, []
        for K in Ks:
            self.K = K
            self.reset()
            flux = self.intensities_on_mesh(energy, theta, psi)[0]
            flux = flux * 1e3 if self.distE == 'BW' else flux * energy[:, None, None]
            powers.append(flux.sum() * volume * EV2ERG * 1e-7)
        self.K = keep
        return np.array(powers)
    period = property(lambda self: self.L0)
    n = property(lambda self: self.Np)

    def _wiggler_points(self, bot, theta0, count, el):
        """Photon source size of the whole device convolved with the electron beam; the
        rays start where the orbit points along their theta. Draws (non-filament): pole
        (randint), x normal, z normal."""
        spot = 2 * (CHeVcm/bot.E*10 * self.L0*self.Np) / PI2**2
        bot.sourceSIGMAx = np.sqrt(self.dx**2 + spot)
        bot.sourceSIGMAz = np.sqrt(self.dz**2 + spot)
        if el is not None:
            return
        lean = np.clip(theta0*self.gamma/self.K, -1., 1.)
        bot.y[:] = ((np.arccos(lean) / PI) +
                    np.random.randint(-int(self.Np), int(self.Np), count) - 0.5) * 0.5 * self.L0
        bot.x[:] = self.X0 * np.sin(PI2 * bot.y / self.L0) + \
            np.random.normal(0., bot.sourceSIGMAx, count)
        bot.z[:] = np.random.normal(0., bot.sourceSIGMAz, count)
