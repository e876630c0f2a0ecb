"""Minimal, matplotlib-free mirror of the part of ``xrt.plotter`` that the job
runner needs: ``XYCAxis`` / ``XYCPlot`` as accumulators of the 2-D histogram,
its 1-D projections and the ray counters (xrt/plotter.py:227-330, 684-900;
xrt/multipro.py:53-177). Drawing, colour (hue) histograms, KDE and persistence
are out of scope (plotter OOS).

The accumulators of a plot live ON THE DEVICE between iterations (the histogram kernels add
into them, nothing is copied or waited for per iteration); ``total2D``, ``total2D_RGB``, the
axes' ``total1D4`` and the ray counters are read through properties that bring the device part
home first. On a machine without a GPU they are plain numpy arrays."""
import numpy as np

_UNIT_FACTORS = {'mm': 1., 'm': 1e-3, 'um': 1e3, u'µm': 1e3, 'nm': 1e6,
                 'rad': 1., 'mrad': 1e3, 'urad': 1e6, u'µrad': 1e6, 'eV': 1.,
                 'keV': 1e-3, '': 1.}
_FLUX = {'total': 0, 's': 1, 'p': 2, '+/-45': 3, 'left-right': 4, 'power': 5}


class XYCAxis(object):
    def __init__(self, label='', unit='mm', factor=None, data='auto', limits=None,
                 offset=0, bins=128, ppb=2, density='histogram', **kwargs):
        given = dict(locals())
        for key in ('label', 'unit', 'data', 'limits', 'offset', 'ppb', 'density'):
            setattr(self, key, given[key])
        self.factor = _UNIT_FACTORS.get(unit, 1.) if factor is None else factor
        self.bins = int(bins)
        self._plot, self._slot = None, None      # the accumulator this axis belongs to
        self._total1D4 = np.zeros((self.bins, 4))

    @property
    def total1D4(self):
        """[bin][flux, R, G, B] of the 1-D histogram along this axis."""
        if self._plot is not None:
            return self._plot._part(self._slot)
        return self._total1D4

    @total1D4.setter
    def total1D4(self, value):
        if self._plot is not None:
            self._plot._part(self._slot)[...] = value
        else:
            self._total1D4 = np.array(value, dtype=float)

    def field(self):
        """Which beam quantity this axis shows (label convention of xrt:
        x, y, z, x', z', energy, path)."""
        lab = self.label.strip('$').replace(' ', '').lower()
        return {'x': 'x', 'y': 'y', 'z': 'z', "x'": 'xprime', "z'": 'zprime',
                'energy': 'E', 'e': 'E', 'path': 'path'}.get(lab, lab)


class XYCPlot(object):
    def __init__(self, beam=None, rayFlag=(1,), xaxis=None, yaxis=None, caxis=None,
                 aspect='equal', title='', fluxKind='total', beamState=None,
                 ePos=1, colorFactor=0.85, colorSaturation=0.85, **kwargs):
        self.beam, self.rayFlag = beam, tuple(rayFlag)
        self.xaxis = xaxis if xaxis is not None else XYCAxis('x', 'mm')
        self.yaxis = yaxis if yaxis is not None else XYCAxis('z', 'mm')
        # colour axis (xrt/plotter.py: caxis='category' colours by ray state and
        # is not mirrored): default = energy in eV
        if caxis == 'category':
            raise NotImplementedError("caxis='category'")
        self.caxis = caxis if caxis is not None else XYCAxis('energy', 'eV', bins=128)
        self.ePos, self.colorFactor, self.colorSaturation = ePos, colorFactor, colorSaturation
        self.title = title or str(beam)
        if not any(fluxKind.startswith(k) for k in _FLUX):
            raise NotImplementedError('fluxKind %r' % fluxKind)
        self.fluxKind, self.beamState = fluxKind, beamState
        self.reset_bins2D()

    # layout of the accumulator: what the histogram kernels fill in one call
    _PARTS = ('total2D', 'total2D_RGB', 'x', 'y', 'c', 'counters')
    # counters[k]: selected, flux, flux inside the limits, alive, good, out, over, dead
    _COUNTER_SLOTS = {'nRaysSelected': 0, 'intensity': 1, 'intensityInRange': 2,
                      'nRaysAlive': 3, 'nRaysGood': 4, 'nRaysOut': 5, 'nRaysOver': 6,
                      'nRaysDead': 7}

    def part_sizes(self):
        nx, ny, nc = self.xaxis.bins, self.yaxis.bins, self.caxis.bins
        return (ny * nx, ny * nx * 3, nx * 4, ny * 4, nc * 4, 8)

    def reset_bins2D(self):
        nx, ny, nc = self.xaxis.bins, self.yaxis.bins, self.caxis.bins
        sizes = self.part_sizes()
        self._flat = np.zeros(sum(sizes))                  # host part of the accumulator
        self._device_flat = None                           # device part (a torch tensor)
        cuts = np.cumsum((0,) + sizes)
        shapes = ((ny, nx), (ny, nx, 3), (nx, 4), (ny, 4), (nc, 4), (8,))
        self._views = {name: self._flat[cuts[k]:cuts[k + 1]].reshape(shapes[k])
                       for k, name in enumerate(self._PARTS)}
        # 1-D histograms, accumulated like xrt/plotter.py's *axis.total1D* and
        # *total1D_RGB*: column 0 = flux weights, columns 1..3 = R, G, B
        for axis, slot in ((self.xaxis, 'x'), (self.yaxis, 'y'), (self.caxis, 'c')):
            axis._plot, axis._slot = self, slot
        self.nRaysAll = 0
        self.iteration = 0

    def device_accumulator(self, device):
        """The device part of the accumulator (zeros when new): the kernels ADD into it."""
        import torch
        if self._device_flat is not None and self._device_flat.device != device:
            self.bring_home()
        if self._device_flat is None:
            self._device_flat = torch.zeros(self._flat.size, dtype=torch.float64, device=device)
        return self._device_flat

    def bring_home(self):
        """Adds what the device holds to the host arrays (one copy, one sync)."""
        if self._device_flat is not None:
            import torch
            # (the kernels may have run on any stream of that device)
            torch.cuda.synchronize(self._device_flat.device)
            self._flat += self._device_flat.cpu().numpy()
            self._device_flat = None

    def _part(self, name):
        self.bring_home()
        return self._views[name]

    total2D = property(lambda self: self._part('total2D'),
                       lambda self, v: self._part('total2D').__setitem__(Ellipsis, v))
    total2D_RGB = property(lambda self: self._part('total2D_RGB'),
                           lambda self, v: self._part('total2D_RGB').__setitem__(Ellipsis, v))

    _COUNTERS = ('nRaysAll', 'iteration')

    def spawn(self):
        """An empty accumulator with this plot's settings (axes copied, limits as they are
        now): what a worker of run_ray_tracing fills."""
        import copy
        twin = copy.copy(self)
        twin.xaxis, twin.yaxis, twin.caxis = (copy.copy(a) for a in
                                              (self.xaxis, self.yaxis, self.caxis))
        twin.reset_bins2D()
        return twin

    def absorb(self, other):
        """Adds a worker's histograms and counters to this plot."""
        other.bring_home()
        self.bring_home()
        self._flat += other._flat
        for name in self._COUNTERS:
            setattr(self, name, getattr(self, name) + getattr(other, name))

    @property
    def flux_kind_code(self):
        # same precedence as raycing.get_output: 'power' before 'p'
        for k in ('power', 's', 'p', '+/-45', 'left-right', 'total'):
            if self.fluxKind.startswith(k):
                return _FLUX[k]
        return 0

    @property
    def ray_flag_mask(self):
        m = 0
        for f in self.rayFlag:
            if f == 1:
                m |= 1
            elif f == 2:
                m |= 2
            elif f == 3:
                m |= 4
            elif f == 4:
                m |= 16
            elif f < 0:
                m |= 8
        return m

    @property
    def total1D_x(self):
        """1-D histogram of x over ALL selected rays (np.histogram on x alone,
        multipro.py:338): not the marginal of the 2-D histogram."""
        return self.xaxis.total1D4[:, 0]

    @property
    def total1D_y(self):
        return self.yaxis.total1D4[:, 0]

    @property
    def total1D_c(self):
        return self.caxis.total1D4[:, 0]

    def edges(self):
        return (np.linspace(self.xaxis.limits[0], self.xaxis.limits[1],
                            self.xaxis.bins + 1),
                np.linspace(self.yaxis.limits[0], self.yaxis.limits[1],
                            self.yaxis.bins + 1))


def _counter_property(slot, integer):
    def get(self):
        v = self._part('counters')[slot]
        return int(round(v)) if integer else float(v)

    def put(self, value):
        self._part('counters')[slot] = value
    return property(get, put)


for _name, _slot in XYCPlot._COUNTER_SLOTS.items():
    setattr(XYCPlot, _name, _counter_property(_slot, _name.startswith('nRays')))
