# -*- coding: utf-8 -*-
"""Figure errors of optical surfaces: height maps that ``OE(figureError=...)`` adds to its
surface (reference: xrt/backends/raycing/figure_error.py; the hooks on the ray path are
``OE.local_z_distorted`` inside ``find_dz``, oes/base.py:826-830, and ``OE.local_n_distorted``
on the normal at the hit point, oes/reflect.py:767-775).

The map lives where the reference keeps it: a ``scipy.interpolate.RectBivariateSpline`` through
heights [nm] on a regular (x, y) grid, built on the host once per change of a parameter. What is
new is where it is EVALUATED: the spline's knots and coefficients, and the coefficients of its
two partial derivatives (formed here the way FITPACK's ``parder`` forms them), go to HBM
(``device_record``) and the ray kernels evaluate them per ray inside the intersection search and
at the hit point (csrc/reflect_impl.h: fe_spline, figure_height, figure_turn_normal; pass
record fields ``xrt_hip_pass.fe_*``). The numpy methods below (``local_z_distorted``,
``local_n_distorted``) are the reference's, for scripts that look at a map themselves.

Classes as in the reference: RandomRoughness, GaussianBump, Waviness, PlanarRidge,
FigureErrorImported; maps add up through *baseFE*.
"""
import os

import numpy as np
from scipy import interpolate

__all__ = ('RandomRoughness', 'GaussianBump', 'Waviness', 'FigureErrorImported')

maxFeHalfSize = 100      # [mm] default half size of a map (figure_error.py:42)


def _rebuilding(name):
    """A constructor argument kept as ``_name`` whose assignment rebuilds the spline."""
    def get(self):
        return getattr(self, '_' + name)

    def put(self, value):
        setattr(self, '_' + name, value)
        self.build_spline()
    return property(get, put)


class FigureErrorBase(object):
    """Common part: the grid, the spline, its evaluation (figure_error.py:45-268).

    *baseFE*: another figure error whose map this one is added to. *limPhysX*, *limPhysY*:
    extent of the map [mm] (default +-100). *gridStep* [mm]: the number of nodes per axis is
    the next power of two of extent / gridStep, at least 128."""

    def __init__(self, name='', baseFE=None, limPhysX=None, limPhysY=None, gridStep=0.5,
                 **kwargs):
        self.name = name
        self.bl = kwargs.get('bl')
        self._baseFE = baseFE
        self._gridStep = gridStep
        self._splineOrder = 3
        self.xShift = 0.
        self.yShift = 0.
        self._limPhysX = self._limits(limPhysX)
        self._limPhysY = self._limits(limPhysY)
        self._device = {}
        if 'skip_build_spline' not in kwargs:
            self.build_spline()

    @staticmethod
    def _limits(lim):
        return [-maxFeHalfSize, maxFeHalfSize] if lim is None else list(lim)

    baseFE = _rebuilding('baseFE')
    gridStep = _rebuilding('gridStep')
    splineOrder = _rebuilding('splineOrder')

    @property
    def limPhysX(self):
        return self._limPhysX

    @limPhysX.setter
    def limPhysX(self, limPhysX):
        self._limPhysX = self._limits(limPhysX)
        self.build_spline()

    @property
    def limPhysY(self):
        return self._limPhysY

    @limPhysY.setter
    def limPhysY(self, limPhysY):
        self._limPhysY = self._limits(limPhysY)
        self.build_spline()

    # ---- diagnostics (figure_error.py:157-200) ------------------------------------------
    def get_rms(self):
        """rms height of the map [nm]."""
        z = self.local_z_distorted(self.x2d, self.y2d) * 1e6
        return np.sqrt(((z - z.mean())**2).mean())

    def get_rms_slope(self):
        """(rms pitch, rms roll) slope errors [rad]."""
        d_pitch, d_roll = self.local_n_distorted(self.x2d, self.y2d)
        return np.sqrt((d_pitch**2).mean()), np.sqrt((d_roll**2).mean())

    def next_pow2(self, n):
        return 1 << int(np.ceil(np.log2(n)))

    def get_dimensions(self):
        xlength = np.abs(self.limPhysX[-1] - self.limPhysX[0])
        ylength = np.abs(self.limPhysY[-1] - self.limPhysY[0])
        self.nx = max(self.next_pow2(xlength / self.gridStep), 128)
        self.ny = max(self.next_pow2(ylength / self.gridStep), 128)
        self.dx = xlength / self.nx
        self.dy = ylength / self.ny

    def get_grids(self):
        self.get_dimensions()
        self.x1d = np.linspace(min(self.limPhysX), max(self.limPhysX), self.nx)
        self.y1d = np.linspace(min(self.limPhysY), max(self.limPhysY), self.ny)
        self.x2d, self.y2d = np.meshgrid(self.x1d, self.y1d)

    def get_angles(self):
        self.a2d, self.b2d = np.gradient(self.z2d * 1e-6, self.y1d, self.x1d)
        self.a2d = np.arctan(self.a2d)
        self.b2d = np.arctan(self.b2d)

    def get_psd(self):
        """(KX, KY, PSD) of the map."""
        z = self.local_z_distorted(self.x2d, self.y2d) * 1e6
        nrow, ncol = z.shape
        H = np.fft.fftshift(np.fft.fft2(z - z.mean()))
        PSD = np.abs(H)**2 / (nrow * ncol)
        dx = np.abs(self.limPhysX[-1] - self.limPhysX[0]) / nrow
        dy = np.abs(self.limPhysY[-1] - self.limPhysY[0]) / ncol
        kx = 2 * np.pi * np.fft.fftshift(np.fft.fftfreq(nrow, d=dx))
        ky = 2 * np.pi * np.fft.fftshift(np.fft.fftfreq(ncol, d=dy))
        KX, KY = np.meshgrid(kx, ky, indexing='xy')
        return KX, KY, PSD

    # ---- the map ----------------------------------------------------------------------
    def _base_profile(self):
        """The map of *baseFE* on this one's grid [nm] (zeros without one)."""
        if self.baseFE is not None and hasattr(self.baseFE, 'local_z_distorted'):
            return self.baseFE.local_z_distorted(self.x2d, self.y2d) * 1e6
        return np.zeros_like(self.x2d)

    def generate_profile(self):
        """Heights [nm] on the (y, x) grid; overridden by the subclasses."""
        self.get_grids()
        return np.zeros_like(self.x2d)

    def build_spline(self):
        z = self.generate_profile()
        self.local_z_spline = interpolate.RectBivariateSpline(
            self.y1d, self.x1d, z, kx=self.splineOrder, ky=self.splineOrder)
        self._device = {}               # (the copies in HBM are of the old spline)
        self.z2d = self.local_z_distorted(self.x2d, self.y2d) * 1e6       # [nm]
        self.get_angles()

    def _flat_args(self, x, y):
        x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
        return x.shape, x.ravel(), y.ravel()

    def local_z_distorted(self, x, y):
        """Height of the map at (x, y) [mm] (figure_error.py:214-235)."""
        shape, x, y = self._flat_args(x, y)
        z = self.local_z_spline.ev(y + self.yShift, x + self.xShift)
        return z.reshape(shape) * 1e-6

    def local_n_distorted(self, x, y):
        """[d_pitch, d_roll]: the two angles the local normal is turned by
        (figure_error.py:237-265)."""
        shape, x, y = self._flat_args(x, y)
        a = self.local_z_spline.ev(y + self.yShift, x + self.xShift, dx=0, dy=1) * 1e-6
        b = self.local_z_spline.ev(y + self.yShift, x + self.xShift, dx=1, dy=0) * 1e-6
        return [np.arctan(b.reshape(shape)), -np.arctan(a.reshape(shape))]

    # ---- what the kernels read ---------------------------------------------------------
    def spline_arrays(self):
        """(k, ty, tx, c, cy, cx): the spline as FITPACK holds it -- first axis = y -- and the
        coefficients of its partial derivatives along y and x as ``parder`` forms them (one
        degree less on that axis, knots without the first and the last one):
        c'[i] = (c[i + 1] - c[i]) * k / (t[i + k + 1] - t[i + 1])."""
        ty, tx, c = self.local_z_spline.tck
        k = int(self.local_z_spline.degrees[0])
        if self.local_z_spline.degrees[1] != k:
            raise ValueError('figure error: one spline order for both axes')
        c = np.asarray(c, dtype=float).reshape(len(ty) - k - 1, len(tx) - k - 1)
        span_y = ty[k + 1:len(ty) - 1] - ty[1:len(ty) - k - 1]
        span_x = tx[k + 1:len(tx) - 1] - tx[1:len(tx) - k - 1]
        with np.errstate(divide='ignore', invalid='ignore'):
            cy = (c[1:, :] - c[:-1, :]) * float(k) / span_y[:, None]
            cx = (c[:, 1:] - c[:, :-1]) * float(k) / span_x[None, :]
        cy[span_y <= 0, :] = c[:-1, :][span_y <= 0, :]          # (parder leaves those as they are)
        cx[:, span_x <= 0] = c[:, :-1][:, span_x <= 0]
        return k, np.ascontiguousarray(ty, dtype=float), np.ascontiguousarray(tx, dtype=float), \
            np.ascontiguousarray(c), np.ascontiguousarray(cy), np.ascontiguousarray(cx)

    def device_record(self, device):
        """The spline in HBM (one block, kept per device until the map changes) ->
        dict(k, nty, ntx, ty, tx, c, cy, cx: device addresses, shift)."""
        import torch
        key = str(device)
        if key not in self._device:
            k, ty, tx, c, cy, cx = self.spline_arrays()
            parts = [ty, tx, c.ravel(), cy.ravel(), cx.ravel()]
            block = torch.from_numpy(np.concatenate(parts)).to(device)
            offsets = np.cumsum([0] + [len(p) for p in parts[:-1]])
            base = block.data_ptr()
            addr = [base + 8 * int(o) for o in offsets]
            self._device[key] = dict(k=k, nty=len(ty), ntx=len(tx), ty=addr[0], tx=addr[1],
                                     c=addr[2], cy=addr[3], cx=addr[4], _keep=block)
        rec = dict(self._device[key])
        rec['shift'] = (float(self.xShift), float(self.yShift))
        return rec


class RandomRoughness(FigureErrorBase):
    """A random map of given rms height [nm] (*rmsKind* 'height') or rms slope [urad] ('slope':
    one number, or (pitch, roll)), smoothed to the correlation length *corrLength* [mm] by a
    Gaussian filter in the spatial-frequency domain; *seed* makes it reproducible
    (figure_error.py:463-620)."""

    def __init__(self, rms=1., rmsKind='height', corrLength=5., seed=None, **kwargs):
        self._rmsKind = rmsKind
        self._rms = rms
        self._corrLength = corrLength
        self._seed = np.random.SeedSequence().entropy if seed is None else seed
        kwargs.setdefault('name', 'random roughness')
        super().__init__(**kwargs)

    def _directional_height(self):
        return self._rmsKind == 'height' and isinstance(self._rms, (tuple, list))

    @property
    def rms(self):
        return self._rms

    @rms.setter
    def rms(self, rms):
        self._rms = rms
        if not self._directional_height():
            self.build_spline()

    @property
    def rmsKind(self):
        return self._rmsKind

    @rmsKind.setter
    def rmsKind(self, rmsKind):
        self._rmsKind = rmsKind
        if not self._directional_height():
            self.build_spline()

    corrLength = _rebuilding('corrLength')

    @property
    def seed(self):
        return self._seed

    @seed.setter
    def seed(self, seed):
        self._seed = np.random.SeedSequence().entropy if seed is None else seed
        self.build_spline()

    def generate_profile(self):
        rng = np.random.default_rng(self.seed)
        self.get_grids()
        base_z = self._base_profile()
        z = rng.normal(loc=0.0, scale=1.0, size=(self.ny, self.nx))
        if self.corrLength is not None:
            spectrum = np.fft.rfft2(z)
            kx = 2 * np.pi * np.fft.rfftfreq(self.nx, d=self.dx)
            ky = 2 * np.pi * np.fft.fftfreq(self.ny, d=self.dy)
            KX, KY = np.meshgrid(kx, ky, indexing='xy')
            if isinstance(self.rms, (tuple, list)):      # the smaller rms, the longer
                corrY = self.corrLength                                      # pitch
                corrX = self.corrLength * self.rms[0] / self.rms[1]          # roll
            else:
                corrX = corrY = self.corrLength
            z = np.fft.irfft2(spectrum * np.exp(-0.5*(KX**2*corrX**2 + KY**2*corrY**2)),
                              s=(self.ny, self.nx))
        z -= z.mean()
        if self.rmsKind == 'height':
            current = np.sqrt((z**2).mean())
            if current > 0:
                z *= (self.rms / current)
        elif self.rmsKind == 'slope':
            a2d, b2d = np.gradient(z*1e-6, self.y1d, self.x1d)
            rms_pitch = np.sqrt((np.arctan(a2d)**2).mean())
            rms_roll = np.sqrt((np.arctan(b2d)**2).mean())
            if isinstance(self.rms, (list, tuple)):
                scale_y = self.rms[0] * 1e-6 / rms_pitch
                scale_x = self.rms[1] * 1e-6 / rms_roll
                spectrum = np.fft.rfft2(z)
                spectrum *= np.sqrt((scale_x * KX)**2 + (scale_y * KY)**2) /\
                    np.sqrt(KX**2 + KY**2 + 1e-30)
                z = np.fft.irfft2(spectrum, s=(self.ny, self.nx))
            else:
                z *= self.rms * 1e-6 / np.sqrt(0.5 * (rms_pitch**2 + rms_roll**2))
        return z + base_z


class GaussianBump(FigureErrorBase):
    """A Gaussian bump of *bumpHeight* [nm] at (*cX*, *cY*) with widths *sigmaX*, *sigmaY*
    [mm] (figure_error.py:623-700)."""

    def __init__(self, bumpHeight=10., cX=0., cY=0., sigmaX=10., sigmaY=10., **kwargs):
        self._bumpHeight, self._sigmaX, self._sigmaY = bumpHeight, sigmaX, sigmaY
        self._cX, self._cY = cX, cY
        kwargs.setdefault('name', 'gaussian bump')
        super().__init__(**kwargs)

    bumpHeight = _rebuilding('bumpHeight')
    sigmaX = _rebuilding('sigmaX')
    sigmaY = _rebuilding('sigmaY')
    cX = _rebuilding('cX')
    cY = _rebuilding('cY')

    def generate_profile(self):
        self.get_grids()
        base_z = self._base_profile()
        z = self.bumpHeight *\
            np.exp(-(self.x2d-self.cX)**2/self.sigmaX**2
                   - (self.y2d-self.cY)**2/self.sigmaY**2)
        return z + base_z


class Waviness(FigureErrorBase):
    """A product of two cosines of *amplitude* [nm] and periods *xWaveLength*, *yWaveLength*
    [mm] (figure_error.py:703-760)."""

    def __init__(self, amplitude=10., xWaveLength=20., yWaveLength=50., **kwargs):
        self._amplitude = amplitude
        self._xWaveLength, self._yWaveLength = xWaveLength, yWaveLength
        kwargs.setdefault('name', 'waviness')
        super().__init__(**kwargs)

    amplitude = _rebuilding('amplitude')
    xWaveLength = _rebuilding('xWaveLength')
    yWaveLength = _rebuilding('yWaveLength')

    def generate_profile(self):
        self.get_grids()
        base_z = self._base_profile()
        z = self.amplitude * np.cos(2*np.pi*self.x2d/self.xWaveLength) *\
            np.cos(2*np.pi*self.y2d/self.yWaveLength)
        return z + base_z


class PlanarRidge(FigureErrorBase):
    """Two planes meeting in a ridge of height *amplitude* [mm] through the centre, each
    inclined by *slopeAngle* [rad], the ridge turned by *orientationAngle* from the x axis (the
    reference's own test shape, figure_error.py:763-840)."""

    def __init__(self, amplitude=1., slopeAngle=1e-3, orientationAngle=0, **kwargs):
        self._amplitude = amplitude
        self._slopeAngle, self._orientationAngle = slopeAngle, orientationAngle
        kwargs.setdefault('name', 'ridge')
        super().__init__(**kwargs)

    amplitude = _rebuilding('amplitude')
    slopeAngle = _rebuilding('slopeAngle')
    orientationAngle = _rebuilding('orientationAngle')

    def generate_profile(self):
        self.get_grids()
        base_z = self._base_profile()
        across = -self.x2d*np.sin(self.orientationAngle) +\
            self.y2d*np.cos(self.orientationAngle)
        z = self.amplitude - np.tan(self.slopeAngle)*np.abs(across)
        return z*1e6 + base_z


class FigureErrorImported(FigureErrorBase):
    """A measured map from a text file of three columns on a full grid: coordinates [mm] and
    height [nm] after *columnFactors*, the columns in the order *orientation* names;
    *recenter* moves the middle of the map to (0, 0) (figure_error.py:271-460)."""

    def __init__(self, fileName=None, recenter=False, orientation='XYZ',
                 columnFactors=[1, 1, 1], **kwargs):
        self.surfArrays = {}
        kwargs.setdefault('name', 'NOM surface')
        self._recenter, self._orientation = recenter, orientation
        self._fileName = None
        self.columnFactors = columnFactors
        kwargs['skip_build_spline'] = True
        super().__init__(**kwargs)
        self._baseFE = None
        self.fileName = fileName

    def _realign(self):
        if self.surfArrays:
            self.align_arrays()
            self.build_spline()

    @property
    def orientation(self):
        return self._orientation

    @orientation.setter
    def orientation(self, orientation):
        self._orientation = orientation
        self._realign()

    @property
    def recenter(self):
        return self._recenter

    @recenter.setter
    def recenter(self, recenter):
        self._recenter = recenter
        self._realign()

    @property
    def columnFactors(self):
        return self._columnFactors

    @columnFactors.setter
    def columnFactors(self, columnFactors):
        try:
            self._columnFactors = [cf*1.0 for cf in columnFactors[:3]]
        except Exception:  # noqa: BLE001 -- not three numbers
            self._columnFactors = [1, 1, 1]
        if self._fileName is not None:
            self.read_file(self.fileName)
            self._realign()

    @property
    def fileName(self):
        return self._fileName

    @fileName.setter
    def fileName(self, fileName):
        self._fileName = fileName
        if fileName is None:
            self._init_empty()
        else:
            self.read_file(fileName)
            self._realign()

    def read_file(self, fileName):
        if not os.path.isfile(str(fileName)):
            raise ValueError('The figure error file does not exist')
        data = np.loadtxt(str(fileName))
        if data.ndim != 2 or data.shape[1] < 3:
            raise ValueError('Invalid figure error file')
        x, y, z = [col * f for col, f in zip(data.T[:3], self.columnFactors)]
        self.surfArrays = {'x': x, 'y': y, 'z': z}

    def align_arrays(self):
        order = str(self.orientation).lower()
        x, y, z = (self.surfArrays.get(order[0]), self.surfArrays.get(order[1]),
                   self.surfArrays.get(order[-1]))
        if self.recenter:
            x -= 0.5*(np.min(x) + np.max(x))
            y -= 0.5*(np.min(y) + np.max(y))
        self._limPhysX = [np.min(x), np.max(x)]
        self._limPhysY = [np.min(y), np.max(y)]
        x1d, y1d = np.unique(x), np.unique(y)
        nx, ny = len(x1d), len(y1d)
        if nx*ny != len(x):
            print('Input data does not form a grid')
            return
        self.nx, self.ny = nx, ny
        by_rows = np.all(np.diff(x[:nx]) > 0) and np.all(y[:nx] == y[0])
        z2d = z.reshape((ny, nx)) if by_rows else z.reshape((nx, ny)).T
        self.x1d, self.y1d = x1d, y1d
        self.x2d, self.y2d = np.meshgrid(self.x1d, self.y1d)
        self.z2d = z2d
        self.get_angles()

    def _init_empty(self):
        self.x1d = np.array(np.linspace(-1, 1, 5))
        self.y1d = np.array(np.linspace(-1, 1, 5))
        self.nx, self.ny = len(self.x1d), len(self.y1d)
        self.x2d, self.y2d = np.meshgrid(self.x1d, self.y1d)
        self.z2d = np.zeros((self.ny, self.nx))
        self.local_z_spline = interpolate.RectBivariateSpline(
            self.y1d, self.x1d, self.z2d, kx=self.splineOrder, ky=self.splineOrder)
        self._device = {}
        self.get_angles()

    def get_grids(self):
        pass

    def get_dimensions(self):
        pass

    def generate_profile(self):
        base_z = self._base_profile()
        z = self.z2d if (self.surfArrays and self.z2d is not None) else np.zeros_like(self.x2d)
        return z + base_z
