"""Materials on the accelerated path — host-side mirror of
xrt/backends/raycing/materials/{element,material,crystal,crystals_basic}.py.

The objects carry the parameters (tables, density, lattice constants ...); the
amplitudes themselves — Fresnel rs/rp/ts/tp (material.py:415-493) and the
Belyakov-Dmitrienko Bragg/Laue amplitudes (crystal.py:492-645) — are evaluated
per ray inside the HIP kernels. ``get_amplitude`` / ``get_refractive_index``
keep xrt's signatures and run the same device functions on arrays.
"""
import ctypes
import os

import numpy as np
import torch

from ... import hipcalls as _hipcalls

from ... import _lib, _structs
from .physconsts import AVOGADRO, CH, CHBAR, PI, PI2, R0

ch = CH        # names user scripts use (materials/__init__.py:97-98)
chbar = CHBAR

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))), 'data', 'elements.npz')
_tables = None

elementsList = (
    'none', 'H', 'He', 'Li', 'Be', 'B', 'C', 'N', 'O', 'F', 'Ne', 'Na', 'Mg',
    'Al', 'Si', 'P', 'S', 'Cl', 'Ar', 'K', 'Ca', 'Sc', 'Ti', 'V', 'Cr', 'Mn',
    'Fe', 'Co', 'Ni', 'Cu', 'Zn', 'Ga', 'Ge', 'As', 'Se', 'Br', 'Kr', 'Rb',
    'Sr', 'Y', 'Zr', 'Nb', 'Mo', 'Tc', 'Ru', 'Rh', 'Pd', 'Ag', 'Cd', 'In',
    'Sn', 'Sb', 'Te', 'I', 'Xe', 'Cs', 'Ba', 'La', 'Ce', 'Pr', 'Nd', 'Pm',
    'Sm', 'Eu', 'Gd', 'Tb', 'Dy', 'Ho', 'Er', 'Tm', 'Yb', 'Lu', 'Hf', 'Ta',
    'W', 'Re', 'Os', 'Ir', 'Pt', 'Au', 'Hg', 'Tl', 'Pb', 'Bi', 'Po', 'At',
    'Rn', 'Fr', 'Ra', 'Ac', 'Th', 'Pa', 'U')


def _load_tables():
    global _tables
    if _tables is None:
        _tables = np.load(_DATA)
    return _tables


class Element(object):
    """Chemical element with f0 coefficients and tabulated f1, f2
    (element.py:76-263; table 'Chantler total' shipped in xrt_amd/data)."""

    def __init__(self, elem=None, table='Chantler total'):
        if isinstance(elem, str):
            self.name = elem
            self.Z = elementsList.index(elem)
        elif isinstance(elem, (int, np.integer)):
            self.name = elementsList[int(elem)]
            self.Z = int(elem)
        else:
            raise NameError('Wrong chemical element')
        if table != 'Chantler total':
            raise ValueError("only the 'Chantler total' table is shipped")
        self.table = table
        tb = _load_tables()
        if self.name + '_E' not in tb.files:
            raise ValueError('no tabulated data for ' + self.name)
        self.f0coeffs = [float(v) for v in tb[self.name + '_f0']]
        self.mass = float(tb[self.name + '_mass'])
        self.E = np.array(tb[self.name + '_E'], dtype=np.float64)
        self.f1 = np.array(tb[self.name + '_f1'], dtype=np.float64)
        self.f2 = np.array(tb[self.name + '_f2'], dtype=np.float64)
        self._dev = {}

    def get_f0(self, qOver4pi=0):
        """Waasmaier-Kirfel f0 (element.py:203-207); a per-crystal constant on
        this path, evaluated once on the host."""
        c = self.f0coeffs
        gauss = [amp * np.exp(-width * qOver4pi**2) for amp, width in zip(c[:5], c[6:])]
        return c[5] + sum(gauss)

    def device_tables(self, device):
        """E, f1, f2 and the coarse index of E (xrt_hip.h: tab_bucket) on *device*."""
        key = str(device)
        if key not in self._dev:
            edges = ((np.arange(_structs.BUCKETS + 1, dtype=np.int64) + _structs.BUCKET_KEY0)
                     << _structs.BUCKET_SHIFT).view(np.float64)
            coarse = np.searchsorted(self.E, edges, side='right').astype(np.int32)
            self._dev[key] = tuple(
                torch.from_numpy(a).to(device) for a in (self.E, self.f1, self.f2, coarse))
        return self._dev[key]


_KINDS = {'mirror': _structs.MAT_MIRROR, 'thin mirror': _structs.MAT_THIN_MIRROR,
          'plate': _structs.MAT_PLATE, 'lens': _structs.MAT_PLATE,
          'crystal': _structs.MAT_CRYSTAL,
          # a 'grating' reflects like a mirror (material.py:476); the grating
          # equation itself is a property of the element (xrt_hip_pass.grating)
          'grating': _structs.MAT_MIRROR,
          # the transparent zones of a zone plate pass the ray unchanged (material.py:457-459)
          'FZP': _structs.MAT_NONE}


def _dev_f64(a, device):
    return torch.from_numpy(np.array(a, dtype=np.float64, order='C')).to(device)


_SPLINE = dict(kind='cubic', bounds_error=False, fill_value='extrapolate')   # material.py:11


def _read_index_file(path):
    """E, n, k of a refractive-index file the way the reference reads it (material.py:284-330):
    a spreadsheet with the three columns, or comma-separated text whose rows are 'E, n' or
    'E, n, k' (k on a sparser or denser grid than n: it is splined onto the energies of n)."""
    from scipy.interpolate import interp1d
    if path.endswith(('.xls', '.xlsx')):
        from pandas import read_excel
        data = read_excel(path).values
        return data[:, 0], data[:, 1], data[:, 2]
    e_n, e_k, n, k = [], [], [], []
    with open(path) as f:
        for line in f:
            fields = line.split(',')
            try:
                energy = float(fields[0])
            except ValueError:
                continue
            if len(fields) < 3:
                e_n.append(energy)
                n.append(float(fields[-1]))
            else:
                e_k.append(energy)
                k.append(float(fields[-1]))
                if fields[1].strip():
                    e_n.append(energy)
                    n.append(float(fields[1]))
    e_n = np.array(e_n)
    return e_n, np.array(n), interp1d(np.array(e_k), np.array(k), **_SPLINE)(e_n)


def _given_index(spec):
    """None, a complex number, or [energies, interp1d] of a tabulated index."""
    import os
    from scipy.interpolate import interp1d
    if spec is None:
        return None
    if isinstance(spec, (int, float, complex)) and not isinstance(spec, bool):
        return complex(spec)
    if isinstance(spec, str):
        if not os.path.exists(spec):
            print(os.path.abspath(spec), "not found! Using refractive index of 1")
            return complex(1.)
        energies, n, k = _read_index_file(spec)
    else:
        table = np.asarray(spec)
        if table.ndim != 2 or table.shape[1] < 3:
            raise ValueError('refractiveIndex: a number, a file, or an array with the columns '
                             'E, n, k')
        energies, n, k = table[:, 0], table[:, 1], table[:, 2]
    energies = np.asarray(energies, dtype=float)
    return [energies, interp1d(energies, np.complex128(np.asarray(n) + 1j*np.asarray(k)),
                               **_SPLINE)]


class Material(object):
    """Amorphous material given by its chemical formula and density
    (material.py:23-157)."""

    def __init__(self, elements=None, quantities=None, kind='auto', rho=0, t=None,
                 table='Chantler total', efficiency=None, efficiencyFile=None, name='',
                 refractiveIndex=None, **kwargs):
        if isinstance(elements, str):
            elements = elements,
        self.table = table
        self.elements = [e if isinstance(e, Element) else Element(e, table)
                         for e in (elements or ())]
        if quantities is None:
            self.quantities = [1. for _ in self.elements]
        elif not isinstance(quantities, (list, tuple)):
            self.quantities = [quantities]
        else:
            self.quantities = list(quantities)
        self.kind = kind
        self.rho = rho
        self.t = t
        # the index given instead of taken from the element tables (visible light, IR, VUV:
        # outside the tables), material.py:240-262: a number, or n(E) + i k(E) as an array with
        # the columns E, n, k or as a file of them -> [energies, cubic spline] like the reference
        self.refractiveIndex = _given_index(refractiveIndex)
        self._index_poly = {}
        # gratings / zone plates: [order, efficiency] pairs used in place of the Fresnel
        # amplitudes (material.py:78-95, 391-413)
        # with *efficiencyFile* the second number of a pair is a column of that file: efficiency
        # against energy, interpolated per ray (material.py:335-346, 403-410)
        self.efficiency, self.efficiencyFile = efficiency, efficiencyFile
        self._efficiency_dev = {}
        if efficiencyFile is not None:
            self.read_efficiency_file()
        self.geom = ''
        self.mass = 0.
        for elem, xi in zip(self.elements, self.quantities):
            self.mass += xi * elem.mass
        self.name = name or ''.join(e.name for e in self.elements)
        self.uuid = kwargs.get('uuid')

    def read_efficiency_file(self):
        """material.py:335-346: a pickle of (energies, table[energy, column]) or a text file
        whose first column is the energy."""
        cols = [int(c[1]) for c in self.efficiency]
        if self.efficiencyFile.endswith('.pickle'):
            import pickle
            with open(self.efficiencyFile, 'rb') as f:
                res = pickle.load(f)
            es, eff = np.asarray(res[0]), np.asarray(res[1]).T[cols, :]
        else:
            es = np.loadtxt(self.efficiencyFile, usecols=(0,), unpack=True)
            eff = np.loadtxt(self.efficiencyFile, usecols=cols,
                             unpack=True).reshape(len(cols), -1)
        self.efficiency_E = np.ascontiguousarray(es, dtype=np.float64)
        self.efficiency_I = np.ascontiguousarray(eff, dtype=np.float64)
        self._efficiency_dev.clear()

    def efficiency_on_device(self, device):
        """(energies, rows) of the efficiency table in HBM, one row per entry of
        *efficiency* (xrt_hip_pass.eff_tab_E / eff_tab_I)."""
        import torch
        held = self._efficiency_dev.get(str(device))
        if held is None or held[0] is not self.efficiency_I:
            held = self._efficiency_dev[str(device)] = (
                self.efficiency_I,
                torch.from_numpy(self.efficiency_E.copy()).to(device),
                torch.from_numpy(self.efficiency_I.copy()).to(device).contiguous())
        return held[1], held[2]

    # ---- struct for the kernels ---------------------------------------------
    def _fill_elements(self, s, device):
        if len(self.elements) > _structs.MAX_ELEM:
            raise NotImplementedError('the kernels take at most %d different elements per '
                                      'material (%s has %d)' % (
                                          _structs.MAX_ELEM, self.name, len(self.elements)))
        keep = []
        s.nelem = len(self.elements)
        for i, (e, xi) in enumerate(zip(self.elements, self.quantities)):
            tE, t1, t2, coarse = e.device_tables(device)
            keep += [tE, t1, t2, coarse]
            s.tab_bucket[i] = coarse.data_ptr()
            s.Z[i] = e.Z
            s.tab_n[i] = tE.numel()
            s.quantity[i] = float(xi)
            s.tab_E[i] = tE.data_ptr()
            s.tab_f1[i] = t1.data_ptr()
            s.tab_f2[i] = t2.data_ptr()
        return keep

    def _table_covers(self, E):
        """min(E) and max(E) strictly inside the tabulated energies (material.py:366-367)."""
        energies = self.refractiveIndex[0]
        if isinstance(E, torch.Tensor):
            if E.numel() == 0:
                return True
            lo, hi = float(E.min()), float(E.max())
        else:
            lo, hi = float(np.min(E)), float(np.max(E))
        return lo > energies[0] and hi < energies[-1]

    def _index_on_device(self, E):
        """The spline of the tabulated index at the energies *E* (device tensor) -> complex128
        tensor: the cubic pieces of scipy's interpolant (its B-spline turned into a piecewise
        polynomial once), found by a bisection and evaluated by Horner's rule, all on the GPU."""
        # (keyed on the interpolant too: assigning another table to refractiveIndex must not
        # leave the old spline on the GPU -- ADVICE r4)
        key = (E.device, id(self.refractiveIndex[1]))
        if key not in self._index_poly:
            for stale in [k for k in self._index_poly if k[1] != key[1]]:
                del self._index_poly[stale]
            from scipy.interpolate import BSpline, PPoly
            spline = self.refractiveIndex[1]._spline      # complex coefficients: two real ones
            flat = np.asarray(spline.c).reshape(len(spline.c), -1)[:, 0]
            parts = [PPoly.from_spline(BSpline(spline.t, np.ascontiguousarray(c), spline.k))
                     for c in (flat.real, flat.imag)]
            knots = np.asarray(parts[0].x, dtype=float)
            coef = np.asarray(parts[0].c) + 1j * np.asarray(parts[1].c)      # [4, pieces]
            # (the first and last three pieces of the B-spline's knot vector are empty)
            keep = np.flatnonzero(np.diff(knots) > 0)
            self._index_poly[key] = (
                torch.from_numpy(knots[keep].copy()).to(E.device),
                torch.from_numpy(np.ascontiguousarray(coef[:, keep]).astype(complex)).to(E.device))
        starts, coef = self._index_poly[key]
        piece = torch.clamp(torch.searchsorted(starts, E, right=True) - 1, 0, starts.numel() - 1)
        dx = E - starts[piece]
        value = coef[0][piece]
        for order in range(1, coef.shape[0]):
            value = value * dx + coef[order][piece]
        return value.contiguous()

    def to_struct(self, fromVacuum=True, device=None, E=None):
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        kind = 'mirror' if self.kind == 'auto' else self.kind
        if kind not in _KINDS:
            raise NotImplementedError('material kind %r is not on the GPU path' % kind)
        s = _structs.Material()
        s.kind = _KINDS[kind]
        s.from_vacuum = 1 if fromVacuum else 0
        s._keep = self._fill_elements(s, device)
        s.rho = float(self.rho)
        s.mass = float(self.mass)
        s.t = float(self.t) if self.t is not None else 0.
        if isinstance(self.refractiveIndex, complex):
            s.n_fixed, s.n_re, s.n_im = 1, self.refractiveIndex.real, self.refractiveIndex.imag
        elif self.refractiveIndex is not None:
            # tabulated: the spline at every ray's energy (device tensor *E*), if the whole batch
            # lies inside the table -- else the element tables, as the reference decides
            # (material.py:364-373, one decision per call)
            if E is None:
                raise NotImplementedError(
                    '%s: a tabulated refractive index is evaluated per ray of a beam (mirrors, '
                    'plates, gratings); not inside a layered material' % self.name)
            if self._table_covers(E):
                index = self._index_on_device(E)
                s._keep.append(index)
                s.n_fixed, s.n_ray = 2, index.data_ptr()
            elif not self.elements:
                raise ValueError('%s: energies outside the tabulated refractive index and no '
                                 'elements to fall back to' % self.name)
            else:
                print("Cannot calculate refractive index. Energy outside of the range. "
                      "Using atomic scattering factors")
        elif not self.elements:
            raise ValueError('a material needs elements or a refractiveIndex')
        if kind == 'thin mirror' and self.t is None:
            raise ValueError('thin mirror needs a thickness t')
        return s

    # ---- xrt API, evaluated by the device functions ---------------------------
    def get_amplitude(self, E, beamInDotNormal, fromVacuum=True):
        """(rs, rp, mu [1/cm], Re(n) k [1/cm]) per ray, material.py:415-493."""
        if self.kind == 'FZP':
            return 1, 1, 0
        _lib.require_gpu()
        lib = _lib.load()
        dev = torch.device('cuda', torch.cuda.current_device())
        E = np.atleast_1d(np.asarray(E, dtype=np.float64))
        bdn = np.broadcast_to(np.asarray(beamInDotNormal, dtype=np.float64),
                              E.shape)
        n = E.size
        dE, db = _dev_f64(E, dev), _dev_f64(bdn, dev)
        s = self.to_struct(fromVacuum, dev, E=dE)
        rs = torch.empty(n, dtype=torch.complex128, device=dev)
        rp = torch.empty(n, dtype=torch.complex128, device=dev)
        mu = torch.empty(n, dtype=torch.float64, device=dev)
        nk = torch.empty(n, dtype=torch.float64, device=dev)
        _lib.check(lib.xrt_hip_material_amplitude_f64_dev(
            ctypes.byref(s), n, dE.data_ptr(), db.data_ptr(), rs.data_ptr(),
            rp.data_ptr(), mu.data_ptr(), nk.data_ptr(),
            _hipcalls.stream_ptr()),
            'xrt_hip_material_amplitude_f64_dev')
        return (rs.cpu().numpy(), rp.cpu().numpy(), mu.cpu().numpy(),
                nk.cpu().numpy())

    def get_refractive_index(self, E):
        """n(E) from (mu, Re(n) k) of the device function (material.py:348-378):
        n = nk*CHBAR/(E*1e8) + i*mu*CHBAR/(E*2e8); sign of Im(n) as tabulated
        (f2 > 0 -> Im(n) < 0)."""
        if isinstance(self.refractiveIndex, complex):
            return self.refractiveIndex
        if self.refractiveIndex is not None and self._table_covers(E):
            return self.refractiveIndex[1](E)
        E = np.atleast_1d(np.asarray(E, dtype=np.float64))
        saved = self.kind
        try:
            if saved not in ('mirror', 'thin mirror', 'plate', 'lens'):   # incl. 'FZP', 'grating'
                self.kind = 'mirror'
            _, _, mu, nk = self.get_amplitude(E, -np.ones_like(E) * 0.5)
        finally:
            self.kind = saved
        return nk * CHBAR / (E * 1e8) - 1j * mu * CHBAR / (E * 2e8)

    def get_absorption_coefficient(self, E):
        E = np.atleast_1d(np.asarray(E, dtype=np.float64))
        return self.get_amplitude(E, -np.ones_like(E) * 0.5)[2]


def _depth_profile(at_top, at_substrate, periods, power):
    """Thickness of one layer kind in every period, vacuum side first: constant, or the
    power law d_n = A / (B + n)**power through the two end values
    (multilayer.py:167-191)."""
    if not at_substrate:
        return np.full(periods, float(at_top))
    ratio = (at_top / at_substrate) ** (1. / power)
    shift = (periods - ratio) / (ratio - 1.)
    scale = at_top * (shift + 1.) ** power
    return scale * (shift + np.arange(1, periods + 1)) ** (-power)


class Multilayer(object):
    """Periodic or depth-graded multilayer on a substrate (materials/multilayer.py:10-566):
    *nPairs* periods of *tLayer* (towards vacuum, *tThickness* [A]) over *bLayer*
    (*bThickness*), thicknesses running to *tThicknessLow* / *bThicknessLow* at the
    substrate by a power law when those are given; *idThickness*: rms interdiffusion /
    roughness of every interface; *geom* 'reflected' or 'transmitted' (then through a
    substrate of *substThickness*).

    The reflectivity (Parratt's recursion with Nevot-Croce factors) is evaluated per ray
    inside the reflect kernels, and by ``get_amplitude`` on arrays through the same device
    function. Laterally graded thicknesses -- the reference's way is to override
    ``get_t_thickness`` / ``get_b_thickness`` in Python -- are not on the GPU path."""

    def __init__(self, tLayer=None, tThickness=0., bLayer=None, bThickness=0., nPairs=0,
                 substrate=None, tThicknessLow=0., bThicknessLow=0., idThickness=0., power=2.,
                 substRoughness=0., substThickness=np.inf, name='', geom='reflected', **kwargs):
        self.tLayer, self.bLayer, self.substrate = tLayer, bLayer, substrate
        self._profile = dict(nPairs=int(nPairs), power=power, tThickness=float(tThickness),
                             bThickness=float(bThickness), tThicknessLow=float(tThicknessLow),
                             bThicknessLow=float(bThicknessLow))
        self.kind = 'multilayer'
        self.geom = geom or 'reflected'
        self.idThickness = idThickness
        self.substRoughness = float(substRoughness)
        self.substThickness = substThickness
        self.name = name
        self.uuid = kwargs.get('uuid')
        self._layout()

    # the six numbers the thickness profile depends on; changing one lays it out again
    def _layout(self):
        q = self._profile
        self.dti = _depth_profile(q['tThickness'], q['tThicknessLow'], q['nPairs'], q['power'])
        self.dbi = _depth_profile(q['bThickness'], q['bThicknessLow'], q['nPairs'], q['power'])
        self._records = {}

    def __getattr__(self, name):
        q = self.__dict__.get('_profile')
        if q is not None and name in q:
            return q[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        q = self.__dict__.get('_profile')
        if q is not None and name in q:
            q[name] = int(value) if name == 'nPairs' else float(value)
            self._layout()
        else:
            object.__setattr__(self, name, value)
            if name in ('tLayer', 'bLayer', 'substrate', 'idThickness', 'substRoughness',
                        'substThickness', 'geom') and '_records' in self.__dict__:
                self._records = {}

    @property
    def d(self):
        return float(self.tThickness + self.bThickness)

    # ---- alignment helpers (host scalars) ---------------------------------------------
    def get_sin_Bragg_angle(self, E, order=1):
        s = order * CH / (2 * self.d * np.asarray(E, dtype=float))
        return np.clip(s, -1 + 1e-16, 1 - 1e-16)

    def get_Bragg_angle(self, E, order=1):
        return np.arcsin(self.get_sin_Bragg_angle(E, order))

    def get_dtheta_symmetric_Bragg(self, E, order=1):
        """Bragg angle minus the angle of the refraction-corrected Bragg law with the
        period average of delta (multilayer.py:222-240)."""
        def decrement(layer, thickness):
            return 0. if layer is None else \
                (layer.get_refractive_index(E).real - 1) * thickness
        mean = abs(decrement(self.tLayer, self.tThickness) +
                   decrement(self.bLayer, self.bThickness)) / self.d
        corrected = ((order * CH / np.asarray(E, dtype=float))**2 +
                     self.d**2 * 8 * mean)**0.5 / (2 * self.d)
        return self.get_Bragg_angle(E, order) - np.arcsin(corrected)

    def get_dtheta(self, E, order=1):
        return self.get_dtheta_symmetric_Bragg(E, order=order)

    def get_t_thickness(self, x, y, iPair):
        return self.dti[iPair]

    def get_b_thickness(self, x, y, iPair):
        return self.dbi[iPair]

    # ---- the record the kernels read ----------------------------------------------------
    def _layer_struct(self, layer, device, keep):
        s = _structs.Material()
        if layer is not None:
            if isinstance(getattr(layer, 'refractiveIndex', None), list):
                raise NotImplementedError(
                    '%s: a tabulated refractive index is evaluated per ray of a beam (mirrors, '
                    'plates, gratings); not inside a layered material' % layer.name)
            keep += layer._fill_elements(s, device)
            s.rho, s.mass = float(layer.rho), float(layer.mass)
        return s

    def to_struct(self, fromVacuum=True, device=None):
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        cls = type(self)
        if cls.get_t_thickness is not Multilayer.get_t_thickness or \
                cls.get_b_thickness is not Multilayer.get_b_thickness:
            raise NotImplementedError('laterally graded multilayers (get_t_thickness / '
                                      'get_b_thickness overridden) are not on the GPU path')
        if self.nPairs < 1:
            raise ValueError('a multilayer needs at least one period')
        key = str(device)
        if key not in self._records:
            keep = []
            rec = _structs.Multilayer()
            rec.top = self._layer_struct(self.tLayer, device, keep)
            rec.bottom = self._layer_struct(self.bLayer, device, keep)
            rec.substrate = self._layer_struct(self.substrate, device, keep)
            rec.npairs = int(self.nPairs)
            rec.transmitted = 1 if 'tran' in self.geom else 0
            rec.uniform = int(np.all(self.dti == self.dti[0]) and np.all(self.dbi == self.dbi[0]))
            dti, dbi = _dev_f64(self.dti, device), _dev_f64(self.dbi, device)
            rec.dti, rec.dbi = dti.data_ptr(), dbi.data_ptr()
            rec.id2 = float(self.idThickness)**2
            rec.bs_rough2 = rec.id2 if self.tLayer is not None else self.substRoughness**2
            rec.subst_thickness = float(self.substThickness)
            image = torch.frombuffer(bytearray(bytes(rec)), dtype=torch.uint8).to(device)
            self._records[key] = (image, keep + [dti, dbi])
        image, keep = self._records[key]
        s = _structs.Material()
        s.kind = _structs.MAT_MULTILAYER
        s.from_vacuum = 1 if fromVacuum else 0
        # kind 'multilayer' deflects like a Bragg crystal of spacing d (reflect.py:865-872);
        # Coated is a 'mirror'
        s.geom_bragg = 1 if self.kind == 'multilayer' else 0
        s.geom_transmitted = 1 if (self.kind == 'multilayer' and
                                   self.geom.endswith('transmitted')) else 0
        s.d = self.d
        s.layers = image.data_ptr()
        s._keep = [image] + keep
        return s

    def get_amplitude(self, E, beamInDotNormal, x=None, y=None, ucl=None):
        """(r_s, r_p) -- or (t_s, t_p) for geom 'transmitted' -- per ray,
        multilayer.py:257-566."""
        _lib.require_gpu()
        lib = _lib.load()
        dev = torch.device('cuda', torch.cuda.current_device())
        bdn = np.atleast_1d(np.asarray(beamInDotNormal, dtype=np.float64))
        E = np.broadcast_to(np.asarray(E, dtype=np.float64), bdn.shape) \
            if np.ndim(E) == 0 else np.asarray(E, dtype=np.float64)
        bdn = np.broadcast_to(bdn, E.shape)
        n = E.size
        s = self.to_struct(True, dev)
        dE, db = _dev_f64(E, dev), _dev_f64(bdn, dev)
        rs = torch.empty(n, dtype=torch.complex128, device=dev)
        rp = torch.empty(n, dtype=torch.complex128, device=dev)
        _lib.check(lib.xrt_hip_multilayer_amplitude_f64_dev(
            ctypes.byref(s), n, dE.data_ptr(), db.data_ptr(), rs.data_ptr(), rp.data_ptr(),
            _hipcalls.stream_ptr()),
            'xrt_hip_multilayer_amplitude_f64_dev')
        return rs.cpu().numpy(), rp.cpu().numpy()


class GradedMultilayer(Multilayer):
    """The reference keeps this name for depth-graded stacks (multilayer.py:569-574)."""


class Coated(Multilayer):
    """A mirror coating of *cThickness* [A] on a *substrate*, with *surfaceRoughness* and
    *substRoughness* [A rms] (multilayer.py:577-625): one period without a top layer; the
    element treats it as a mirror."""

    def __init__(self, *args, **kwargs):
        coating = kwargs.pop('coating', None)
        thickness = kwargs.pop('cThickness', 0)
        rough = kwargs.pop('surfaceRoughness', 0)
        Multilayer.__init__(self, *args, bLayer=coating, bThickness=thickness,
                            idThickness=rough, nPairs=1, **kwargs)
        self.kind = 'mirror'

    coating = property(lambda self: self.bLayer,
                       lambda self, m: setattr(self, 'bLayer', m))
    cThickness = property(lambda self: self.bThickness,
                          lambda self, t: setattr(self, 'bThickness', t))
    surfaceRoughness = property(lambda self: self.idThickness,
                                lambda self, t: setattr(self, 'idThickness', t))


class EmptyMaterial(object):
    """A material without reflectivity that still tells the element what it is -- by default
    a 'grating': the rays take the grating equation, their amplitudes stay (reference
    materials/__init__.py:101-113)."""

    def __init__(self, kind='grating', **kwargs):
        self.kind, self.geom, self.name = kind, '', ''
        self.uuid = kwargs.get('uuid')
        self.efficiency = None

    def to_struct(self, fromVacuum=True, device=None):
        s = _structs.Material()
        s.kind, s.from_vacuum, s._keep = _structs.MAT_NONE, int(bool(fromVacuum)), []
        return s


def parse_hkl(hkl):
    return tuple(int(i) for i in hkl)


class Crystal(Material):
    """Perfect crystal (crystal.py:30-226): hkl, d spacing, unit-cell volume,
    geometry 'Bragg|Laue reflected|transmitted', thickness t [mm] or None."""

    structure = 0      # 0: fcc-like structure factor; 1: diamond

    def __init__(self, hkl=(1, 1, 1), d=0, V=None, elements='Si', quantities=None,
                 rho=0, t=None, factDW=1., geom='Bragg reflected',
                 table='Chantler total', name='', **kwargs):
        super(Crystal, self).__init__(elements, quantities, rho=rho, table=table,
                                      name=name, **kwargs)
        self.hkl = parse_hkl(hkl) if len(hkl) else (1, 1, 1)
        self.sqrthkl2 = (sum(i**2 for i in self.hkl))**0.5
        self.d = d
        self.V = V if V is not None else (d * self.sqrthkl2)**3
        self.chiToF = -R0 / PI / self.V          # crystal.py:201
        self.chiToFd2 = abs(self.chiToF) * d**2
        if len(geom) < 6:
            geom = geom.strip() + ' reflected'
        self.geom = geom
        self.factDW = factDW
        self.kind = 'crystal'
        self.t = t
        self.mosaicity = 0
        self.useTT = False
        self.volumetricDiffraction = False

    def get_Bragg_angle(self, E, order=1):
        a = order * CH / (2*self.d*np.asarray(E, dtype=float))
        a = np.clip(a, -1 + 1e-16, 1 - 1e-16)
        return np.arcsin(a)

    def get_structure_factor_f0(self):
        return self.elements[0].get_f0(0.5 / self.d)

    def to_struct(self, fromVacuum=True, device=None):
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        if self.mosaicity or self.useTT or self.volumetricDiffraction:
            raise NotImplementedError('mosaic / bent (TT) / volumetric crystals '
                                      'are outside the accelerated path')
        if self.geom.endswith('Fresnel'):
            raise NotImplementedError("'Fresnel' crystal geometry is not on the GPU path")
        s = _structs.Material()
        s.kind = _structs.MAT_CRYSTAL
        s.from_vacuum = 1 if fromVacuum else 0
        s._keep = self._fill_elements(s, device)
        s.rho = float(self.rho)
        s.mass = float(self.mass)
        s.t = 0.
        s.structure = int(self.structure)
        for i in range(3):
            s.hkl[i] = int(self.hkl[i])
        s.geom_bragg = 1 if self.geom.startswith('Bragg') else 0
        s.geom_transmitted = 1 if self.geom.endswith('transmitted') else 0
        s.thick = 1 if self.t is None else 0
        s.t_crystal = 0. if self.t is None else float(self.t)
        s.d = float(self.d)
        s.chi_to_f = float(self.chiToF)
        s.fact_dw = float(self.factDW)
        s.f0_hkl = float(self.get_structure_factor_f0())
        d2f = 1 + np.exp(0.5j * PI * sum(self.hkl))   # crystals_basic.py:77
        s.d2f_re = float(d2f.real)
        s.d2f_im = float(d2f.imag)
        return s

    def get_amplitude(self, E, beamInDotNormal, beamOutDotNormal=None,
                      beamInDotHNormal=None, xd=None, yd=None):
        """(curveS, curveP) complex amplitudes per ray, crystal.py:492-645."""
        _lib.require_gpu()
        lib = _lib.load()
        dev = torch.device('cuda', torch.cuda.current_device())
        E = np.atleast_1d(np.asarray(E, dtype=np.float64))
        g0 = np.broadcast_to(np.asarray(beamInDotNormal, dtype=np.float64), E.shape)
        gh = -g0 if beamOutDotNormal is None else np.broadcast_to(
            np.asarray(beamOutDotNormal, dtype=np.float64), E.shape)
        hn = g0 if beamInDotHNormal is None else np.broadcast_to(
            np.asarray(beamInDotHNormal, dtype=np.float64), E.shape)
        n = E.size
        s = self.to_struct(True, dev)
        S = torch.empty(n, dtype=torch.complex128, device=dev)
        P = torch.empty(n, dtype=torch.complex128, device=dev)
        args = [_dev_f64(a, dev) for a in (E, g0, gh, hn)]
        _lib.check(lib.xrt_hip_crystal_amplitude_f64_dev(
            ctypes.byref(s), n, *[a.data_ptr() for a in args], S.data_ptr(),
            P.data_ptr(),
            _hipcalls.stream_ptr()),
            'xrt_hip_crystal_amplitude_f64_dev')
        return S.cpu().numpy(), P.cpu().numpy()

    def get_dtheta_symmetric_Bragg(self, E):
        """chi0 / sin(2 theta_B), crystal.py:1125-1139 (host scalar helper used
        for alignment; needs only F0 = 8 or 4 (Z + f1 + i f2) at E)."""
        E = np.atleast_1d(np.asarray(E, dtype=float))
        e = self.elements[0]
        f1 = np.interp(E, e.E, e.f1)
        F0re = 4 * (e.Z + f1) * self.factDW * (2 if self.structure == 1 else 1)
        chi0 = F0re * self.chiToF * (CH / E)**2
        return chi0 / np.sin(2*self.get_Bragg_angle(E))

    def get_dtheta(self, E, alpha=None):
        """Refraction correction of the Bragg angle for an asymmetric cut,
        crystal.py:1141-1171 (Authier eq. 8.3); alignment helper."""
        cut = 0 if alpha is None else alpha
        theta = self.get_Bragg_angle(E)
        sign = 1 if not self.geom.startswith('Bragg') else -1
        g_in = np.sin(theta + cut)
        g_out = sign * np.sin(theta - cut)
        cos_in = np.sqrt(1. - g_in**2)
        under = g_in**2 + sign*(g_in - g_out) * cos_in * self.get_dtheta_symmetric_Bragg(E)
        return -((sign*g_in - sign*np.sqrt(under)) / cos_in)


class CrystalFcc(Crystal):
    structure = 0


class CrystalDiamond(CrystalFcc):
    structure = 1

    def __init__(self, *args, **kwargs):
        a = kwargs.pop('a', None)
        if a is not None:
            hkl = kwargs.get('hkl', args[0] if args else (1, 1, 1))
            kwargs['d'] = a / (sum(i**2 for i in hkl))**0.5
        kwargs.setdefault('name', 'Diamond')
        super(CrystalDiamond, self).__init__(*args, **kwargs)
        self.a = self.d * self.sqrthkl2


class CrystalSi(CrystalDiamond):
    """Silicon with the temperature-dependent lattice parameter of
    crystals_basic.py:83-142 (Swenson's thermal expansion)."""

    def __init__(self, *args, **kwargs):
        self.a0 = 5.430710
        self.tK = kwargs.pop('tK', 297.15)
        hkl = kwargs.get('hkl', args[0] if args else (1, 1, 1))
        kwargs.pop('a', None)
        sqrthkl2 = (sum(i**2 for i in hkl))**0.5
        kwargs['d'] = self.get_a() / sqrthkl2
        kwargs['elements'] = 'Si'
        kwargs['hkl'] = hkl
        kwargs.setdefault('name', 'Si')
        super(CrystalSi, self).__init__(*args[1:], **kwargs)

    # Swenson's relative thermal expansion of silicon: [t_from, t_to) in K -> coefficients
    # of t^4 ... t^0 (summed in that order)
    _EXPANSION = (
        (0.0, 30.0, (0., 0., 0., 0., -2.154537e-004)),
        (30.0, 130.0, (-2.303956e-014, 7.834799e-011, -1.724143e-008, 8.396104e-007,
                       -2.276144e-004)),
        (130.0, 293.0, (0., -1.223001e-011, 1.532991e-008, -3.263667e-006, -5.217231e-005)),
        (293.0, 1000.0 + 1e-9, (0., -1.161022e-012, 3.311476e-009, 1.124129e-006,
                                -5.844535e-004)),
    )

    @classmethod
    def dl_l(cls, t):
        for t_from, t_to, coefs in cls._EXPANSION:
            if t_from <= t < t_to:
                terms = [c * t**(4 - k) for k, c in enumerate(coefs) if c != 0.]
                total = terms[0]
                for term in terms[1:]:
                    total = total + term
                return total
        return 1.0e+100

    def get_a(self):
        return self.a0 * (self.dl_l(self.tK) - self.dl_l(273.15 + 19.9) + 1)


_DIAMOND_CELL = [[0., 0., 0.], [0., 0.5, 0.5], [0.5, 0.5, 0.], [0.5, 0., 0.5],
                 [0.25, 0.25, 0.25], [0.25, 0.75, 0.75], [0.75, 0.25, 0.75],
                 [0.75, 0.75, 0.25]]


class CrystalFromCell(Crystal):
    """Crystal given by its unit cell (crystals_basic.py:157-440): edges *a*, *b*, *c* [A]
    (*b*, *c* default to *a*), angles *alpha*, *beta*, *gamma* [deg], ALL atoms of the cell
    (*atoms*: Z or symbol each, *atomsXYZ*: fractional coordinates, *atomsFraction*:
    occupancies). d spacing, cell volume, density and the structure factor follow.

    On the GPU the sums over the atoms of one element are constants of the reflection
    (``xrt_hip_material.cell_*``); the ray-dependent part is f1 + i f2 of each element --
    at most four different elements per crystal. ``elements`` / ``quantities`` hold the
    DISTINCT elements and their summed occupancies (the reference repeats them per atom)."""
    structure = 2

    def __init__(self, name='', hkl=(1, 1, 1), a=5.430710, b=None, c=None, alpha=90, beta=90,
                 gamma=90, atoms=(14,)*8, atomsXYZ=_DIAMOND_CELL, atomsFraction=None, tK=0,
                 t=None, factDW=1., geom='Bragg reflected', table='Chantler total',
                 volumetricDiffraction=False, useTT=False, nu=0, mosaicity=0, **kwargs):
        self.a, self.b, self.c = a, b or a, c or a
        self.alpha, self.beta, self.gamma = alpha, beta, gamma
        self.atoms = list(atoms)
        self.atomsXYZ = [list(r) for r in atomsXYZ]
        self.atomsFraction = [1 for _ in self.atoms] if atomsFraction is None \
            else list(atomsFraction)
        if not len(self.atoms) == len(self.atomsXYZ) == len(self.atomsFraction):
            raise ValueError('atoms, atomsXYZ and atomsFraction differ in length')
        distinct, per_atom = [], []
        for atom in self.atoms:
            e = Element(atom, table)
            if e.Z not in [q.Z for q in distinct]:
                distinct.append(e)
            per_atom.append([q.Z for q in distinct].index(e.Z))
        self._atom_element = per_atom
        summed = [sum(f for f, k in zip(self.atomsFraction, per_atom) if k == i)
                  for i in range(len(distinct))]
        cosines = np.cos(np.radians((alpha, beta, gamma)))
        sines = np.sin(np.radians((alpha, beta, gamma)))
        ca, cb, cg = cosines
        sa, sb, sg = sines
        a, b, c = self.a, self.b, self.c
        V = a * b * c * (1 - ca**2 - cb**2 - cg**2 + 2*ca*cb*cg)**0.5
        h, k, l = hkl   # noqa: E741
        # 1/d^2 of a triclinic lattice (crystals_basic.py:416-421, same operation order:
        # d sets the Bragg angle the user aligns with)
        d = V / (a * b * c) * \
            ((h*sa/a)**2 + (k*sb/b)**2 + (l*sg/c)**2 + 2*h*k * (ca*cb - cg) / (a*b) +
             2*h*l * (ca*cg - cb) / (a*c) + 2*k*l * (cb*cg - ca) / (b*c))**(-0.5)
        Crystal.__init__(self, hkl=hkl, d=float(d), V=float(V), elements=distinct,
                         quantities=summed, t=t, factDW=factDW, geom=geom, table=table,
                         name=name, **kwargs)
        self.mass = 0.
        for f, idx in zip(self.atomsFraction, per_atom):     # per atom, as the reference sums
            self.mass += f * distinct[idx].mass
        self.rho = self.mass / AVOGADRO / self.V * 1e24
        self.tK, self.nu = tK, nu
        self.volumetricDiffraction, self.useTT, self.mosaicity = \
            volumetricDiffraction, useTT, mosaicity

    def _cell_sums(self):
        """Per distinct element: (sum w, sum w e^{+i phase}, sum w e^{-i phase})."""
        n = len(self.elements)
        w, s, sm = np.zeros(n), np.zeros(n, complex), np.zeros(n, complex)
        for f, idx, xyz in zip(self.atomsFraction, self._atom_element, self.atomsXYZ):
            turn = np.exp(2j * np.pi * np.dot(xyz, self.hkl))
            w[idx] += f
            s[idx] += f * turn
            sm[idx] += f / turn
        return w, s, sm

    def get_structure_factor(self, E, sinThetaOverLambda=0, needFhkl=True):
        """(F0, F_hkl, F_-h-k-l) on the host (alignment helper)."""
        E = np.asarray(E, dtype=float)
        w, s, sm = self._cell_sums()
        F0 = Fh = Fhm = 0
        for e, we, se, sme in zip(self.elements, w, s, sm):
            anomalous = np.interp(E, e.E, e.f1) + 1j * np.interp(E, e.E, e.f2)
            f0 = e.get_f0(sinThetaOverLambda) if needFhkl else 0
            F0 = F0 + we * (e.Z + anomalous) * self.factDW
            Fh = Fh + (f0 + anomalous) * se * self.factDW
            Fhm = Fhm + (f0 + anomalous) * sme * self.factDW
        return F0, Fh, Fhm

    def get_dtheta_symmetric_Bragg(self, E):
        E = np.atleast_1d(np.asarray(E, dtype=float))
        F0 = self.get_structure_factor(E, needFhkl=False)[0]
        chi0 = F0.real * self.chiToF * (CH / E)**2
        return chi0 / np.sin(2*self.get_Bragg_angle(E))

    def to_struct(self, fromVacuum=True, device=None):
        s = Crystal.to_struct(self, fromVacuum, device)
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        w, cs, csm = self._cell_sums()
        rec = _structs.Cell()
        for i, e in enumerate(self.elements):
            rec.w[i] = float(w[i])
            rec.f0[i] = float(e.get_f0(0.5 / self.d))
            rec.s[i][0], rec.s[i][1] = float(cs[i].real), float(cs[i].imag)
            rec.sm[i][0], rec.sm[i][1] = float(csm[i].real), float(csm[i].imag)
        image = torch.frombuffer(bytearray(bytes(rec)), dtype=torch.uint8).to(device)
        s.cell = image.data_ptr()
        s._keep = list(s._keep) + [image]
        return s


# ---- predefined materials: materials.elemental / .compounds / .crystals ------------------
def _predefined_modules():
    """The reference's catalogues of ready-made materials (materials/elemental.py: one class
    per element at its ambient density; compounds.py: common compounds and polymers;
    crystals.py: crystal structures from their unit cells) as modules of classes built from
    the data extract xrt_amd/data/materials.json (oracle/gen_material_data.py). A class takes
    the keyword arguments of its base (``Material``, ``CrystalDiamond``, ``CrystalFromCell``)
    on top of its defaults: ``xcryst.Ge(hkl=(2, 2, 0))``, ``xcomp.Silica(kind='plate')``."""
    import json
    import sys
    import types
    with open(os.path.join(os.path.dirname(_DATA), 'materials.json')) as f:
        catalogue = json.load(f)

    def make(base, defaults, label):
        def __init__(self, *args, **kwargs):
            for key, value in defaults.items():
                kwargs.setdefault(key, value)
            base.__init__(self, *args, **kwargs)
        return type(label, (base,), {'__init__': __init__, '__doc__': '%s (%s)' % (
            label, ', '.join('%s=%r' % kv for kv in sorted(defaults.items())
                             if kv[0] not in ('atomsXYZ', 'atoms', 'atomsFraction')))})

    modules = {}
    for section in ('elemental', 'compounds', 'crystals'):
        mod = types.ModuleType(__name__ + '.' + section, 'predefined materials: ' + section)
        for label, d in catalogue[section].items():
            if section != 'crystals':
                defaults = dict(elements=tuple(d['elements']), quantities=tuple(d['quantities']),
                                rho=d['rho'], name=d['name'])
                cls = make(Material, defaults, label)
            elif d['base'] == 'diamond':
                cls = make(CrystalDiamond, dict(elements=d['elements'][0], a=d['a'],
                                                name=d['name']), label)
            else:
                defaults = {k: d[k] for k in ('name', 'a', 'b', 'c', 'alpha', 'beta', 'gamma',
                                              'atoms', 'atomsXYZ', 'atomsFraction')}
                cls = make(CrystalFromCell, defaults, label)
            cls.__module__ = mod.__name__
            setattr(mod, label, cls)
        mod.__all__ = tuple(catalogue[section])
        sys.modules[mod.__name__] = mod
        modules[section] = mod
    return modules


elemental, compounds, crystals = (_predefined_modules()[k]
                                  for k in ('elemental', 'compounds', 'crystals'))
