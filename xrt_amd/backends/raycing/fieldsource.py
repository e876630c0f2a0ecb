"""``SourceFromField`` — a synchrotron source defined by a tabulated magnetic field
(reference: xrt/backends/raycing/sources/synchr.py:612-1347 on top of the integrated-source
base, sources/sybase.py:933-1810).

What runs where: every ``build_I_map`` of the reference first integrates the electron's
trajectory through the field (a Python Runge-Kutta loop over ~10 grid points per mm of
device — seconds per call — or, with OpenCL, the one-work-item kernels ``get_trajectory`` /
``get_trajectory_filament``) and then sums the field of all rays over the integration nodes
(``_sp_sum`` / the ``custom_field`` kernels). Both are HIP kernels here
(``xrt_hip_trajectory_f64_dev``, ``xrt_hip_custom_field_f64_dev``); the trajectory depends
on the field table and (filament beam) the electron energy only, so it is integrated once
per such pair, not once per batch of rays. Cubic splines (field table -> grids, trajectory
-> integration nodes) are scipy's, as in the reference; sampling, rejection and
polarisation are the host code shared with ``Undulator`` (numpy RNG in the reference's
order, so a seed gives the reference's rays).

Calls of ten rays or fewer (the probe rays of the automatic search for the number of
nodes, ``gNodes=None``) take the carrier frequency of the reference's second, vectorised
formulation of the integrand, as the reference does (synchr.py:1327-1335; the two differ
in nothing else). Not mirrored: the built-in periodic test field (``customField=None``),
Excel input.
"""
import numpy as np
import torch
from scipy.interpolate import interp1d

from ... import hipcalls
from .physconsts import C, CHeVcm, PI2, SIE0, FINE_STR
from .undulator import Undulator, clenshaw_curtis, _TABLE_LOCK

SIM0 = 9.109383701528e-31       # electron mass [kg], reference physconsts.py:17
EMC = 0.5866791802416487        # e / (m c) in the units of the field tables, physconsts.py:24
_CUBIC = dict(kind='cubic', bounds_error=False, fill_value='extrapolate')


class SourceFromField(Undulator):
    def __init__(self, *args, **kwargs):
        """*customField*: the field table as an array or the name of a text file (or a
        pair (name, keyword dictionary for ``numpy.loadtxt``)): columns = longitudinal
        coordinate [mm], then B_ver, or B_hor and B_ver, or B_hor, B_ver and B_long [T].
        The other arguments are the electron-beam, energy-range, angular-range and
        quadrature (*gNodes*, *gIntervals*, *gp*) arguments of ``Undulator``."""
        table = kwargs.pop('customField', None)
        if table is None:
            raise NotImplementedError('SourceFromField needs a customField table (the '
                                      "reference's periodic test field is not mirrored)")
        kwargs.update(K=1., xPrimeMaxAutoReduce=False, zPrimeMaxAutoReduce=False)
        Undulator.__init__(self, *args, **kwargs)
        for undulator_only in ('Kx', 'Ky', 'L0', 'Np', 'phase', 'targetE'):
            self.__dict__.pop(undulator_only, None)
        # no radiation-cone narrowing of the angular range for a general field
        self.xPrimeMaxAutoReduce = self.zPrimeMaxAutoReduce = False
        self.deviceLength = 0
        self._trajectories = {}
        self.customField = table

    # ---- the field table ---------------------------------------------------------------
    @property
    def customField(self):
        return self._customField

    @customField.setter
    def customField(self, table):
        self._customField = table
        self._grid_paths = {}
        if isinstance(table, np.ndarray):
            self.customFieldData = table
        else:
            name, readkw = table if isinstance(table, (tuple, list)) else (table, {})
            self.customFieldData = self.read_custom_field(name, readkw)
        self._trajectories = {}
        self.needReset = True

    def read_custom_field(self, fname, kwargs={}):
        """Text table -> array; the device length is the FWHM of the largest field
        component along the table (synchr.py:695-710)."""
        if fname.endswith(('.xls', '.xlsx')):
            raise NotImplementedError('Excel field tables')
        data = np.loadtxt(fname, **kwargs)
        z, peak = data[:, 0], np.abs(data[:, 1:]).max(axis=1)
        strong = np.argwhere(peak >= peak.max()*0.5)
        self.deviceLength = z[np.max(strong)] - z[np.min(strong)] + (z[1] - z[0])
        return data

    def _magnetic_field(self, grid=None):
        """(Bx, By, Bz) on *grid* [mm], or on the half-step grid of the trajectory
        integration, which this call lays out: 10 points per mm of table
        (synchr.py:727-765)."""
        table = self.customFieldData
        z0, z1 = table[0, 0], table[-1, 0]
        if grid is None:
            self.wtGrid = np.linspace(z0, z1, int(np.abs(z1 - z0)*10))
            self.BGrid = grid = np.linspace(z0, z1, 2*len(self.wtGrid) - 1)

        def column(c):
            return interp1d(table[:, 0], table[:, c], **_CUBIC)(grid)
        count = table.shape[1]
        if count not in (2, 3, 4):
            raise ValueError('field table with %d columns' % count)
        By = column(1 if count == 2 else 2)
        Bx = column(1) if count > 2 else np.zeros_like(By)
        Bz = column(3) if count > 3 else np.zeros_like(By)
        return Bx, By, Bz

    # ---- photon source size: device length in place of N periods (synchr.py:712-725) ---
    def get_sigma_r02(self, E):
        return 2 * CHeVcm/E*10 * self.deviceLength / PI2**2

    def get_sigmaP_r02(self, E):
        return CHeVcm/E*10 / (2 * self.deviceLength)

    def get_SIGMA(self, E, onlyOddHarmonics=True, with0eSpread=False):
        spot = self.get_sigma_r02(E)
        return (self.dx**2 + spot)**0.5, (self.dz**2 + spot)**0.5

    def report_E1(self):
        pass

    # ---- limits and grid ---------------------------------------------------------------
    def _reset_limits(self):
        self.Kx = self.Ky = 0.          # read by the shared limit code only
        try:
            Undulator._reset_limits(self)
        finally:
            del self.Kx, self.Ky

    def _build_integration_grid(self):
        """Nodes along the device: *gIntervals* equal stretches of the table, *quadm*
        nodes in each (synchr.py:988-1003)."""
        rule = np.polynomial.legendre.leggauss if self._useGauLeg else clenshaw_curtis
        nodes, weights = rule(self.quadm)
        z = self.customFieldData[:, 0]
        dstep = (z[-1] - z[0]) / float(self.gIntervals)
        mid = np.arange(0.5 * dstep + z[0], z[-1], dstep)
        self.tg = (mid[:, None] + 0.5*dstep*nodes).ravel()
        self.ag = (mid[:, None]*0 + weights).ravel()
        self.dstep = dstep
        self._trajectories = {}


    # ---- trajectory: one kernel, cached ---------------------------------------------------
    def build_trajectory(self, Bx, By, Bz, gamma=None):
        """-> (betax, betay, [betam], trajx, trajy, trajz) on the integration nodes
        (synchr.py:1005-1147). Runge-Kutta on the GPU, splines here."""
        dev = self._device()

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        g = float(self.gamma if gamma is None else gamma) if self.filamentBeam else None
        key = (id(self.customFieldData), g)
        on_grid = self._grid_paths.get(key)      # the grid does not depend on the nodes
        if on_grid is None:
            if g is not None:
                out = hipcalls.trajectory(up(self.wtGrid), up(Bx), up(By), up(Bz), gamma=g,
                                          emcg=SIE0 / SIM0 / C / 10. / g)
            else:
                out = hipcalls.trajectory(up(self.wtGrid), up(Bx), up(By), up(Bz))
            on_grid = self._grid_paths[key] = [t.cpu().numpy() for t in out]
        on_nodes = [interp1d(self.wtGrid, a, **_CUBIC)(self.tg) for a in on_grid[:5]]
        return on_nodes[0], on_nodes[1], [float(on_grid[5][0])], on_nodes[2], on_nodes[3], \
            on_nodes[4]

    def _node_tables(self):
        """The ten node tables of the field sum on the device + betam, for the present
        field table, grid and (filament beam) electron energy."""
        dev = self._device()
        key = (id(self.customFieldData), bool(self.filamentBeam),
               float(self.gamma) if self.filamentBeam else None, len(self.tg), str(dev))
        with _TABLE_LOCK:       # (workers of run_ray_tracing(threads=N) share this source)
            hit = self._trajectories.get(key)
        if hit is None:
            Bx, By, Bz = self._magnetic_field()
            betax, betay, betazav, trajx, trajy, trajz = self.build_trajectory(Bx, By, Bz)
            Bxt, Byt, Bzt = self._magnetic_field(self.tg)
            host = dict(tg=self.tg, ag=self.ag, Bx=Bxt, By=Byt, Bz=Bzt, betax=betax,
                        betay=betay, trajx=trajx, trajy=trajy, trajz=trajz)
            tables = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(dev)
                      for k, v in host.items()}
            hit = (tables, host, betazav[-1])
            with _TABLE_LOCK:
                self._trajectories[key] = hit
        return hit

    # ---- the field integral ---------------------------------------------------------------
    def build_I_map(self, w, ddtheta, ddpsi, harmonic=None, dg=None):
        """(I, Es, Ep) per ray (synchr.py:975-986, 1274-1346): trajectory tables, then the
        sum over the nodes on the GPU, then the flux scaling."""
        if self.needReset:
            self.reset()
        w = np.atleast_1d(np.asarray(w, dtype=float))
        n = len(w)
        theta = np.atleast_1d(np.asarray(ddtheta, dtype=float)) * np.ones(n)
        psi = np.atleast_1d(np.asarray(ddpsi, dtype=float)) * np.ones(n)
        gamma = self.gamma
        if self.eEspread > 0:
            if dg is not None:
                gamma = gamma + dg
            else:
                gamma = gamma + gamma * self.eEspread * np.random.normal(
                    size=1 if self.filamentBeam else n)
        gamma = gamma * np.ones(n)
        tables, host, betam = self._node_tables()
        if self.filamentBeam:
            ab = 0.5 / np.pi / betam
        else:
            ab = 0.5 / np.pi / (1. - 0.5/gamma**2 + betam*EMC**2/gamma**2)
        emcg = SIE0 / SIM0 / C / 10. / gamma
        # what the reference keeps for inspection
        if self.filamentBeam:
            self.beta = [host['betax'], host['betay']]
            self.trajectory = [host['trajx'], host['trajy'], host['trajz']]
        else:
            self.beta = [host['betax']*emcg[0], host['betay']*emcg[0]]
            self.trajectory = [host['trajx']*emcg[0], host['trajy']*emcg[0],
                               self.tg*(1.-0.5/gamma[0]**2) +
                               host['trajz']*EMC**2/gamma[0]**2]
        dev = tables['tg'].device

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        Is, Ip = hipcalls.custom_field(
            tables, up(emcg), up(gamma), up(w), up(theta), up(psi), betam,
            filament=bool(self.filamentBeam), R0=self.R0 if self.R0 else None,
            carrier_form=0 if n > 10 else 1)
        Is, Ip = Is.cpu().numpy(), Ip.cpu().numpy()
        bandwidth = 0.001 if self.distE == 'BW' else 1./w
        to_flux = FINE_STR * bandwidth * self.eI / SIE0
        power = np.abs(Is)**2 + np.abs(Ip)**2
        if self.convergenceSearchFlag:
            return np.abs(np.sqrt(power) * 0.5 * self.dstep)
        return (to_flux * 0.25 * self.dstep**2 * ab**2 * power,
                np.sqrt(to_flux) * Is * 0.5 * self.dstep * ab,
                np.sqrt(to_flux) * Ip * 0.5 * self.dstep * ab)

    def build_I_map_device(self, *args, **kwargs):
        raise NotImplementedError('SourceFromField: use build_I_map')
    build_I_map_device._no_device_map = True
