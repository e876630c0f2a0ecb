# -*- coding: utf-8 -*-
"""Figure errors of optical surfaces: height maps that ``OE(figureError=...)`` adds to its
surface (reference: xrt/backends/raycing/figure_error.py; the hooks on the ray path are
``OE.local_z_distorted`` inside ``find_dz``, oes/base.py:826-830, and ``OE.local_n_distorted``
on the normal at the hit point, oes/reflect.py:767-775).

The map lives where the reference keeps it: a ``scipy.interpolate.RectBivariateSpline`` through
heights [nm] on a regular (x, y) grid, rebuilt on the host whenever a parameter is assigned.
What is new is where it is EVALUATED: the spline's knots and coefficients, and the coefficients
of its two partial derivatives (formed here the way FITPACK's ``parder`` forms them), go to HBM
(``device_record``) and the ray kernels evaluate them per ray inside the intersection search and
at the hit point (csrc/reflect_impl.h: fe_spline, figure_height, figure_turn_normal; pass
record fields ``xrt_hip_pass.fe_*``). ``local_z_distorted`` / ``local_n_distorted`` below are
the reference's numpy methods, for scripts that look at a map themselves.

The classes and their arguments are the reference's (RandomRoughness, GaussianBump, Waviness,
PlanarRidge, FigureErrorImported; maps add up through *baseFE*), and so is every map: the
generators keep the reference's order of floating-point operations, and
tests/test_figure_error.py holds their splines against splines made inside the reference.
"""
import os

import numpy as np
from scipy import interpolate

__all__ = ('RandomRoughness', 'GaussianBump', 'Waviness', 'FigureErrorImported')

maxFeHalfSize = 100      # [mm] default half size of a map (figure_error.py:42)


class _Param(object):
    """A constructor argument of a map, kept on the instance under ``_<name>``: assigning it
    rebuilds the spline (what the reference does with one property pair per argument).
    *convert*: applied to an assigned value; *when*: a predicate of the instance that must
    hold for the rebuild."""

    def __init__(self, convert=None, when=None):
        self.convert, self.when = convert, when

    def __set_name__(self, owner, name):
        self.slot = '_' + name

    def __get__(self, obj, owner=None):
        return self if obj is None else getattr(obj, self.slot)

    def __set__(self, obj, value):
        setattr(obj, self.slot, value if self.convert is None else self.convert(value))
        if getattr(obj, '_built', False) and (self.when is None or self.when(obj)):
            obj.build_spline()


def _extent(lim):
    return [-maxFeHalfSize, maxFeHalfSize] if lim is None else list(lim)


def _nodes(lim, step):
    """(number of nodes, cell size, nodes) along one axis of a generated map: the next power of
    two of extent / step, at least 128 (figure_error.py:167-181)."""
    length = np.abs(lim[-1] - lim[0])
    n = max(1 << int(np.ceil(np.log2(length / step))), 128)
    return n, length / n, np.linspace(min(lim), max(lim), n)


class FigureErrorBase(object):
    """Common part: the grid, the spline, its evaluation (figure_error.py:45-268).

    *baseFE*: another figure error whose map this one is added to. *limPhysX*, *limPhysY*:
    extent of the map [mm] (default +-100). *gridStep* [mm]: the number of nodes per axis is
    the next power of two of extent / gridStep, at least 128."""

    baseFE = _Param()
    gridStep = _Param()
    splineOrder = _Param()
    limPhysX = _Param(_extent)
    limPhysY = _Param(_extent)

    def __init__(self, name='', baseFE=None, limPhysX=None, limPhysY=None, gridStep=0.5,
                 **kwargs):
        self._built = False
        self.name, self.bl = name, kwargs.get('bl')
        self.xShift = self.yShift = 0.
        self.baseFE, self.gridStep, self.splineOrder = baseFE, gridStep, 3
        self.limPhysX, self.limPhysY = limPhysX, limPhysY
        self._device = {}
        self._built = True
        if 'skip_build_spline' not in kwargs:
            self.build_spline()

    # ---- diagnostics (figure_error.py:157-200) ------------------------------------------
    def _heights_nm(self):
        return self.local_z_distorted(self.x2d, self.y2d) * 1e6

    def get_rms(self):
        """rms height of the map [nm]."""
        centred = self._heights_nm()
        centred = centred - centred.mean()
        return np.sqrt((centred**2).mean())

    def get_rms_slope(self):
        """(rms pitch, rms roll) slope errors [rad]."""
        return tuple(np.sqrt((angle**2).mean())
                     for angle in self.local_n_distorted(self.x2d, self.y2d))

    def next_pow2(self, n):
        return 1 << int(np.ceil(np.log2(n)))

    def get_dimensions(self):
        self.nx, self.dx, _ = _nodes(self.limPhysX, self.gridStep)
        self.ny, self.dy, _ = _nodes(self.limPhysY, self.gridStep)

    def get_grids(self):
        self.nx, self.dx, self.x1d = _nodes(self.limPhysX, self.gridStep)
        self.ny, self.dy, self.y1d = _nodes(self.limPhysY, self.gridStep)
        self.x2d, self.y2d = np.meshgrid(self.x1d, self.y1d)

    def get_angles(self):
        """Slope angles of the map on its grid (a2d along y, b2d along x)."""
        along_y, along_x = np.gradient(self.z2d * 1e-6, self.y1d, self.x1d)
        self.a2d, self.b2d = np.arctan(along_y), np.arctan(along_x)

    def get_psd(self):
        """(KX, KY, PSD) of the map."""
        z = self._heights_nm()
        rows, cols = z.shape
        PSD = np.abs(np.fft.fftshift(np.fft.fft2(z - z.mean())))**2 / (rows * cols)
        step_x = np.abs(self.limPhysX[-1] - self.limPhysX[0]) / rows
        step_y = np.abs(self.limPhysY[-1] - self.limPhysY[0]) / cols
        KX, KY = np.meshgrid(2 * np.pi * np.fft.fftshift(np.fft.fftfreq(rows, d=step_x)),
                             2 * np.pi * np.fft.fftshift(np.fft.fftfreq(cols, d=step_y)),
                             indexing='xy')
        return KX, KY, PSD

    # ---- the map ----------------------------------------------------------------------
    def _base_profile(self):
        """The map of *baseFE* on this one's grid [nm] (zeros without one)."""
        under = self.baseFE
        if under is not None and hasattr(under, 'local_z_distorted'):
            return under.local_z_distorted(self.x2d, self.y2d) * 1e6
        return np.zeros_like(self.x2d)

    def _own_profile(self):
        """This class's heights [nm] on the (y, x) grid (the subclasses' part)."""
        return np.zeros_like(self.x2d)

    def generate_profile(self):
        """Heights [nm] on the (y, x) grid: the class's own map on top of *baseFE*'s."""
        self.get_grids()
        under = self._base_profile()
        return self._own_profile() + under

    def _fit(self, heights):
        self.local_z_spline = interpolate.RectBivariateSpline(
            self.y1d, self.x1d, heights, kx=self.splineOrder, ky=self.splineOrder)
        self._device = {}               # (the copies in HBM are of the old spline)

    def build_spline(self):
        self._fit(self.generate_profile())
        self.z2d = self._heights_nm()
        self.get_angles()

    def _ev(self, x, y, **derivative):
        x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
        values = self.local_z_spline.ev(y.ravel() + self.yShift, x.ravel() + self.xShift,
                                        **derivative)
        return values.reshape(x.shape)

    def local_z_distorted(self, x, y):
        """Height of the map at (x, y) [mm] (figure_error.py:214-235)."""
        return self._ev(x, y) * 1e-6

    def local_n_distorted(self, x, y):
        """[d_pitch, d_roll]: the two angles the local normal is turned by
        (figure_error.py:237-265)."""
        slope_x = self._ev(x, y, dx=0, dy=1) * 1e-6
        slope_y = self._ev(x, y, dx=1, dy=0) * 1e-6
        return [np.arctan(slope_y), -np.arctan(slope_x)]

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

    @staticmethod
    def paired_rows(c):
        """[rows][cols] -> [rows][cols][2] with (c[i][j], c[i + 1][j]), zeros under the last row:
        the form the kernels read coefficients in (two rows per 16-byte load)."""
        below = np.vstack([c[1:], np.zeros((1, c.shape[1]))])
        return np.ascontiguousarray(np.stack([c, below], axis=-1))

    @staticmethod
    def linspace_of(knots, k):
        """(lo, step, hi) if the knots are those of an interpolating cubic spline through a
        numpy linspace -- x[0] four times, x[2] .. x[N - 3], x[N - 1] four times, x[j] = j * step +
        lo in linspace's two roundings -- bit for bit, else None. The kernels then compute the
        knots instead of loading them."""
        n = len(knots) - 4
        if k != 3 or n < 8:
            return None
        lo, hi = float(knots[0]), float(knots[-1])
        step = (hi - lo) / (n - 1)
        nodes = np.arange(0, n) * step + lo
        nodes[-1] = hi
        built = np.concatenate([[lo] * 4, nodes[2:n - 2], [hi] * 4])
        if step > 0 and np.array_equal(built, knots):
            return lo, step, hi
        return None

    def device_record(self, device):
        """The spline in HBM (one block, kept per device until the map changes) ->
        dict(k, nty, ntx; ty, tx, c, cy, cx: device addresses; grid, shift): the knots, the
        coefficients and those of the two partial derivatives as paired rows, and per axis
        (y, x) the linspace its knots follow, if they do."""
        import torch
        key = str(device)
        if key not in self._device:
            k, ty, tx, c, cy, cx = self.spline_arrays()
            parts = [ty, tx, self.paired_rows(c).ravel(), self.paired_rows(cy).ravel(),
                     self.paired_rows(cx).ravel()]
            parts = [np.concatenate([p, np.zeros(len(p) % 2)]) for p in parts]   # 16-B starts
            block = torch.from_numpy(np.concatenate(parts)).to(device)
            offsets = np.cumsum([0] + [len(p) for p in parts[:-1]])
            addr = [block.data_ptr() + 8 * int(o) for o in offsets]
            self._device[key] = dict(k=k, nty=len(ty), ntx=len(tx), ty=addr[0], tx=addr[1],
                                     c=addr[2], cy=addr[3], cx=addr[4], _keep=block,
                                     grid=(self.linspace_of(ty, k), self.linspace_of(tx, k)))
        rec = dict(self._device[key])
        rec['shift'] = (float(self.xShift), float(self.yShift))
        return rec


def _fresh_seed(seed):
    return np.random.SeedSequence().entropy if seed is None else seed


def _scalar_rms(fe):
    """(a height rms must be one number: a pair is kept but builds nothing,
    figure_error.py:531-549)"""
    return not (fe._rmsKind == 'height' and isinstance(fe._rms, (tuple, list)))


class RandomRoughness(FigureErrorBase):
    """A random map of given rms height [nm] (*rmsKind* 'height') or rms slope [urad] ('slope':
    one number, or (pitch, roll)), smoothed to the correlation length *corrLength* [mm] by a
    Gaussian filter in the spatial-frequency domain; *seed* makes it reproducible
    (figure_error.py:463-620)."""

    rms = _Param(when=_scalar_rms)
    rmsKind = _Param(when=_scalar_rms)
    corrLength = _Param()
    seed = _Param(_fresh_seed)

    def __init__(self, rms=1., rmsKind='height', corrLength=5., seed=None, **kwargs):
        self._built = False
        self.rmsKind, self.rms, self.corrLength, self.seed = rmsKind, rms, corrLength, seed
        super().__init__(**dict(kwargs, name=kwargs.get('name', 'random roughness')))

    def _own_profile(self):
        # (numbers are drawn and combined in the reference's order: the map is the same bits)
        noise = np.random.default_rng(self.seed).normal(loc=0.0, scale=1.0,
                                                        size=(self.ny, self.nx))
        pair = isinstance(self.rms, (tuple, list))
        KX = KY = None
        if self.corrLength is not None:
            KX, KY = np.meshgrid(2 * np.pi * np.fft.rfftfreq(self.nx, d=self.dx),
                                 2 * np.pi * np.fft.fftfreq(self.ny, d=self.dy), indexing='xy')
            # two rms slopes: the smaller one gets the longer correlation
            along_y = self.corrLength                                            # pitch
            along_x = self.corrLength * self.rms[0] / self.rms[1] if pair else along_y   # roll
            window = np.exp(-0.5*(KX**2*along_x**2 + KY**2*along_y**2))
            noise = np.fft.irfft2(np.fft.rfft2(noise) * window, s=(self.ny, self.nx))
        noise -= noise.mean()
        if self.rmsKind == 'height':
            now = np.sqrt((noise**2).mean())
            if now > 0:
                noise *= (self.rms / now)
        elif self.rmsKind == 'slope':
            along_y, along_x = np.gradient(noise*1e-6, self.y1d, self.x1d)
            pitch_now = np.sqrt((np.arctan(along_y)**2).mean())
            roll_now = np.sqrt((np.arctan(along_x)**2).mean())
            if pair:
                gain_y = self.rms[0] * 1e-6 / pitch_now
                gain_x = self.rms[1] * 1e-6 / roll_now
                spectrum = np.fft.rfft2(noise)
                spectrum *= np.sqrt((gain_x * KX)**2 + (gain_y * KY)**2) /\
                    np.sqrt(KX**2 + KY**2 + 1e-30)
                noise = np.fft.irfft2(spectrum, s=(self.ny, self.nx))
            else:
                noise *= self.rms * 1e-6 / np.sqrt(0.5 * (pitch_now**2 + roll_now**2))
        return noise


class GaussianBump(FigureErrorBase):
    """A Gaussian bump of *bumpHeight* [nm] at (*cX*, *cY*) with widths *sigmaX*, *sigmaY*
    [mm] (figure_error.py:623-700)."""

    bumpHeight = _Param()
    sigmaX = _Param()
    sigmaY = _Param()
    cX = _Param()
    cY = _Param()

    def __init__(self, bumpHeight=10., cX=0., cY=0., sigmaX=10., sigmaY=10., **kwargs):
        self._built = False
        self.bumpHeight, self.cX, self.cY = bumpHeight, cX, cY
        self.sigmaX, self.sigmaY = sigmaX, sigmaY
        super().__init__(**dict(kwargs, name=kwargs.get('name', 'gaussian bump')))

    def _own_profile(self):
        exponent = -(self.x2d - self.cX)**2 / self.sigmaX**2 \
            - (self.y2d - self.cY)**2 / self.sigmaY**2
        return self.bumpHeight * np.exp(exponent)


class Waviness(FigureErrorBase):
    """A product of two cosines of *amplitude* [nm] and periods *xWaveLength*, *yWaveLength*
    [mm] (figure_error.py:703-760)."""

    amplitude = _Param()
    xWaveLength = _Param()
    yWaveLength = _Param()

    def __init__(self, amplitude=10., xWaveLength=20., yWaveLength=50., **kwargs):
        self._built = False
        self.amplitude, self.xWaveLength, self.yWaveLength = amplitude, xWaveLength, yWaveLength
        super().__init__(**dict(kwargs, name=kwargs.get('name', 'waviness')))

    def _own_profile(self):
        across = np.cos(2*np.pi*self.x2d/self.xWaveLength)
        along = np.cos(2*np.pi*self.y2d/self.yWaveLength)
        return self.amplitude * across * along


class PlanarRidge(FigureErrorBase):
    """Two planes meeting in a ridge of height *amplitude* [mm] through the centre, each
    inclined by *slopeAngle* [rad], the ridge turned by *orientationAngle* from the x axis (the
    reference's own test shape, figure_error.py:763-840)."""

    amplitude = _Param()
    slopeAngle = _Param()
    orientationAngle = _Param()

    def __init__(self, amplitude=1., slopeAngle=1e-3, orientationAngle=0, **kwargs):
        self._built = False
        self.amplitude, self.slopeAngle = amplitude, slopeAngle
        self.orientationAngle = orientationAngle
        super().__init__(**dict(kwargs, name=kwargs.get('name', 'ridge')))

    def _own_profile(self):
        turn = self.orientationAngle
        off_ridge = -self.x2d*np.sin(turn) + self.y2d*np.cos(turn)
        return (self.amplitude - np.tan(self.slopeAngle)*np.abs(off_ridge)) * 1e6


def _three_factors(factors):
    try:
        return [f*1.0 for f in factors[:3]]
    except Exception:  # noqa: BLE001 -- not three numbers
        return [1, 1, 1]


class FigureErrorImported(FigureErrorBase):
    """A measured map from a text file of three columns on a full grid: coordinates [mm] and
    height [nm] after *columnFactors*, the columns in the order *orientation* names;
    *recenter* moves the middle of the map to (0, 0) (figure_error.py:271-460)."""

    def __init__(self, fileName=None, recenter=False, orientation='XYZ',
                 columnFactors=[1, 1, 1], **kwargs):
        self._built = False
        self.surfArrays, self.z2d = {}, None
        self._recenter, self._orientation, self._fileName = recenter, orientation, None
        self._columnFactors = _three_factors(columnFactors)
        # (nothing may reset baseFE after this call: the constructor argument is honoured as in
        # the reference, figure_error.py:306-309 -- ADVICE r4)
        super().__init__(**dict(kwargs, name=kwargs.get('name', 'NOM surface'),
                                skip_build_spline=True))
        self.fileName = fileName

    def _realign(self):
        if self.surfArrays:
            self.align_arrays()
            self.build_spline()

    def _reload(self):
        if self._fileName is not None:
            self.read_file(self._fileName)
            self._realign()

    orientation = property(lambda self: self._orientation)
    recenter = property(lambda self: self._recenter)
    columnFactors = property(lambda self: self._columnFactors)
    fileName = property(lambda self: self._fileName)

    @orientation.setter
    def orientation(self, value):
        self._orientation = value
        self._realign()

    @recenter.setter
    def recenter(self, value):
        self._recenter = value
        self._realign()

    @columnFactors.setter
    def columnFactors(self, value):
        self._columnFactors = _three_factors(value)
        self._reload()

    @fileName.setter
    def fileName(self, value):
        self._fileName = value
        if value is None:
            self._init_empty()
        else:
            self._reload()

    def read_file(self, fileName):
        if not os.path.isfile(str(fileName)):
            raise ValueError('The figure error file does not exist')
        table = np.loadtxt(str(fileName))
        if table.ndim != 2 or table.shape[1] < 3:
            raise ValueError('Invalid figure error file')
        self.surfArrays = {axis: column * factor for axis, column, factor in
                           zip('xyz', table.T[:3], self._columnFactors)}

    def align_arrays(self):
        """File columns -> the (y, x) grid of the map (figure_error.py:384-431)."""
        names = str(self.orientation).lower()
        x, y, z = (self.surfArrays.get(names[0]), self.surfArrays.get(names[1]),
                   self.surfArrays.get(names[-1]))
        if self.recenter:
            x -= 0.5*(np.min(x) + np.max(x))
            y -= 0.5*(np.min(y) + np.max(y))
        self._limPhysX, self._limPhysY = [np.min(x), np.max(x)], [np.min(y), np.max(y)]
        xs, ys = np.unique(x), np.unique(y)
        if len(xs) * len(ys) != len(x):
            print('Input data does not form a grid')
            return
        self.nx, self.ny = len(xs), len(ys)
        x_runs_first = np.all(np.diff(x[:self.nx]) > 0) and np.all(y[:self.nx] == y[0])
        self.z2d = z.reshape((self.ny, self.nx)) if x_runs_first else \
            z.reshape((self.nx, self.ny)).T
        self.x1d, self.y1d = xs, ys
        self.x2d, self.y2d = np.meshgrid(xs, ys)
        self.get_angles()

    def _init_empty(self):
        """No file: a flat map on a 5 x 5 grid."""
        self.x1d = self.y1d = np.array(np.linspace(-1, 1, 5))
        self.nx = self.ny = 5
        self.x2d, self.y2d = np.meshgrid(self.x1d, self.y1d)
        self.z2d = np.zeros((5, 5))
        self._fit(self.z2d)
        self.get_angles()

    def get_grids(self):            # (the grid is the file's)
        pass

    def get_dimensions(self):
        pass

    def _own_profile(self):
        if self.surfArrays and self.z2d is not None:
            return self.z2d
        return np.zeros_like(self.x2d)
