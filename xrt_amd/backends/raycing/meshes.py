"""Mesh functions of the synchrotron sources (reference sources/sybase.py:676-932,
sources/synchr.py:1710-1785): intensities / Stokes parameters on an (energy, theta, psi[,
harmonic]) mesh with incoherent averaging over the electron beam's energy spread and
divergence, the stack of macro-electron fields, tuning and power curves. The field on the
mesh is ``build_I_map`` (a HIP launch per harmonic); the bookkeeping around it is a few
numpy reductions on the mesh-sized result, as in the reference."""
import numpy as np

from .physconsts import EV2ERG


class MeshFunctions(object):
    def _default_mesh(self, energy, theta, psi):
        if self.needReset:
            self.reset()
        if isinstance(energy, str):
            energy = np.mgrid[self.E_min:self.E_max + 0.5*self.dE:self.dE]
        if isinstance(theta, str):
            theta = np.mgrid[self.Theta_min:self.Theta_max + 0.5*self.dTheta:self.dTheta]
        if isinstance(psi, str):
            psi = np.mgrid[self.Psi_min:self.Psi_max + 0.5*self.dPsi:self.dPsi]
        return energy, theta, psi

    def multi_electron_stack(self, energy='auto', theta='auto', psi='auto', harmonic=None,
                             withElectronDivergence=True):
        """(Es, Ep) in the shape (energy, theta, psi[, harmonic]): along the first axis
        one macro-electron per entry of *energy*, each with its own random angular offset
        (one normal array per plane with a non-zero divergence) and its own gamma within
        the energy spread (one more normal array)."""
        energy, theta, psi = self._default_mesh(energy, theta, psi)
        electrons = 1 if np.ndim(energy) == 0 else len(energy)
        axes = (energy, theta, psi) if harmonic is None else (energy, theta, psi, harmonic)
        mesh = list(np.meshgrid(*axes, indexing='ij'))
        along_first = (slice(None),) + (None,) * (len(axes) - 1)
        for plane, sigma in ((1, self.dxprime), (2, self.dzprime)):
            if withElectronDivergence and sigma > 0:
                mesh[plane] = mesh[plane] + np.random.normal(0, sigma, electrons)[along_first]
        dgamma = 0
        if self.eEspread > 0:
            offsets = np.random.normal(0, self.eEspread, electrons) * self.gamma
            dgamma = (np.zeros_like(mesh[0]) + offsets[along_first]).ravel()
        shape = tuple(len(np.atleast_1d(a)) for a in axes)
        res = self.build_I_map(mesh[0].ravel(), mesh[1].ravel(), mesh[2].ravel(),
                               None if harmonic is None else mesh[3].ravel(), dgamma)
        return res[1].reshape(shape), res[2].reshape(shape)

    def intensities_on_mesh(self, energy='auto', theta='auto', psi='auto', harmonic=None,
                            eSpreadSigmas=3.5, eSpreadNSamples=36, mode='constant',
                            resultKind='Stokes'):
        """resultKind 'Stokes': [s0, s1/s0, s2/s0, s3/s0]; 'vortex': [Is, Ip, OAMs, OAMp,
        Es, Ep] (orbital angular momentum densities); arrays of the shape (energy, theta,
        psi[, harmonic]). Energy spread: an extra mesh axis of *eSpreadNSamples* gammas
        within +-*eSpreadSigmas*, averaged with normal weights; divergence: a Gaussian
        filter of the angular planes (scipy.ndimage, border *mode*). Phases are lost."""
        if resultKind not in ('Stokes', 'vortex'):
            raise ValueError("Unknown resultKind {0}".format(resultKind))
        energy, theta, psi = self._default_mesh(energy, theta, psi)
        axes = [energy, theta, psi]
        if harmonic is not None:
            axes.append(harmonic)
        weights = None
        if self.eEspread > 0:
            steps = np.linspace(-eSpreadSigmas, eSpreadSigmas, eSpreadNSamples)
            weights = np.exp(-0.5 * steps**2)
            weights /= weights.sum()
            axes.append(self.gamma * steps * self.eEspread)
        mesh = np.meshgrid(*axes, indexing='ij')
        shape = [len(a) for a in axes]
        spread_axis = len(axes) - 1 if weights is not None else None
        res = self.build_I_map(
            mesh[0].ravel(), mesh[1].ravel(), mesh[2].ravel(),
            None if harmonic is None else mesh[3].ravel(),
            None if weights is None else mesh[spread_axis].ravel())
        Es, Ep = res[1].reshape(shape), res[2].reshape(shape)
        Is = (Es*np.conj(Es)).real.astype(float)
        Ip = (Ep*np.conj(Ep)).real.astype(float)
        if resultKind == 'Stokes':
            parts = [Is, Ip, Es*np.conj(Ep).astype(complex)]
        else:
            ds_dtheta, ds_dpsi = np.gradient(Es, theta, psi, axis=(1, 2))
            dp_dtheta, dp_dpsi = np.gradient(Ep, theta, psi, axis=(1, 2))
            grid = [1] * Es.ndim
            grid[1] = len(theta)
            th = np.asarray(theta, dtype=float).reshape(grid)
            grid[1], grid[2] = 1, len(psi)
            ps = np.asarray(psi, dtype=float).reshape(grid)
            parts = [Is, Ip, (Es.conj()*(1j*(ds_dtheta*ps - ds_dpsi*th))).real.astype(float),
                     (Ep.conj()*(1j*(dp_dtheta*ps - dp_dpsi*th))).real.astype(float), Es, Ep]
        if weights is not None:
            w = weights.reshape([1] * spread_axis + [-1])
            parts = [(p * w).sum(axis=spread_axis) for p in parts]
        self.Is, self.Ip = parts[0], parts[1]
        if resultKind == 'Stokes':
            self.Isp = parts[2]
            out = [parts[0] + parts[1], parts[0] - parts[1], 2. * np.real(parts[2]),
                   -2. * np.imag(parts[2])]
        else:
            out = parts
        if (self.dxprime > 0 or self.dzprime > 0) and len(theta) > 1 and len(psi) > 1:
            from scipy.ndimage import gaussian_filter
            blur = [self.dxprime / (theta[1] - theta[0]), self.dzprime / (psi[1] - psi[0])]
            for name, sigma, axis in (('theta', blur[0], theta), ('psi', blur[1], psi)):
                if sigma > len(axis)//4:
                    print('Warning: the %s mesh is narrower than the electron beam '
                          'divergence it is convolved with' % name)
            for ie in range(len(energy)):
                for arr in out:
                    if harmonic is None:
                        arr[ie, :, :] = gaussian_filter(arr[ie, :, :], blur, mode=mode)
                    else:
                        for ih in range(len(harmonic)):
                            arr[ie, :, :, ih] = gaussian_filter(arr[ie, :, :, ih], blur,
                                                                mode=mode)
        if resultKind == 'vortex':
            return out
        s0 = out[0]
        with np.errstate(divide='ignore', invalid='ignore'):
            return [s0] + [np.where(s0, s/s0, s0) for s in out[1:]]

    def _scan_K(self, Ks, job):
        """*job()* for every deflection parameter of *Ks* (the vertical-field K)."""
        keep, results = self.Ky, []
        try:
            for K in Ks:
                self.Ky = K
                self.needReset = True
                results.append(job())
        finally:
            self.Ky = keep
            self.needReset = True
        return results

    def tuning_curves(self, energy, theta, psi, harmonics, Ks):
        """-> (energies [keV], fluxes) of the flux maxima of *harmonics* through the
        aperture *theta* x *psi*, rows = *Ks*... transposed as the reference returns them:
        rows = harmonics, columns = Ks."""
        energy = np.asarray(energy)
        cell = (theta[1] - theta[0]) * (psi[1] - psi[0]) if np.ndim(theta) else 1

        def curve():
            flux = np.vstack([self.intensities_on_mesh([e], theta, psi, harmonics)[0]
                              .sum(axis=(1, 2)) * cell for e in energy]) \
                if len(energy) > 1 else \
                self.intensities_on_mesh([energy[0]], theta, psi, harmonics)[0].sum(
                    axis=(1, 2)) * cell
            return energy[np.argmax(flux, axis=0)] / 1000., np.max(flux, axis=0)
        rows = self._scan_K(Ks, curve)
        return np.array([r[0] for r in rows]).T, np.array([r[1] for r in rows]).T

    def power_vs_K(self, energy, theta, psi, harmonics, Ks):
        """Total power [W] through the aperture within the energy range, per K of *Ks*."""
        energy = np.asarray(energy)
        volume = (theta[1] - theta[0]) * (psi[1] - psi[0]) * (energy[1] - energy[0]) \
            if np.ndim(theta) else 1

        def power():
            flux = self.intensities_on_mesh(energy, theta, psi, harmonics)[0]
            if self.distE == 'BW':
                flux = flux * 1e3
            else:
                flux = flux * energy.reshape([-1] + [1] * (flux.ndim - 1))
            return flux.sum() * volume * EV2ERG * 1e-7
        return np.array(self._scan_K(Ks, power))
