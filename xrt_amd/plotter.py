"""Minimal, matplotlib-free mirror of the part of ``xrt.plotter`` that the job
runner needs: ``XYCAxis`` / ``XYCPlot`` as accumulators of the 2-D histogram,
its 1-D projections and the ray counters (xrt/plotter.py:227-330, 684-900;
xrt/multipro.py:53-177). Drawing, colour (hue) histograms, KDE and persistence
are out of scope (plotter OOS)."""
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

    def reset_bins2D(self):
        self.total2D = np.zeros((self.yaxis.bins, self.xaxis.bins))
        self.total2D_RGB = np.zeros((self.yaxis.bins, self.xaxis.bins, 3))
        # 1-D histograms, accumulated like xrt/plotter.py's *axis.total1D* and
        # *total1D_RGB*: column 0 = flux weights, columns 1..3 = R, G, B
        self.xaxis.total1D4 = np.zeros((self.xaxis.bins, 4))
        self.yaxis.total1D4 = np.zeros((self.yaxis.bins, 4))
        self.caxis.total1D4 = np.zeros((self.caxis.bins, 4))
        self.nRaysAll = 0
        self.nRaysSelected = 0
        self.nRaysAlive = 0
        self.nRaysGood = 0
        self.nRaysOut = 0
        self.nRaysOver = 0
        self.nRaysDead = 0
        self.intensity = 0.          # sum of weights of the selected rays
        self.intensityInRange = 0.   # ... of those inside the plot limits
        self.iteration = 0

    _COUNTERS = ('nRaysAll', 'nRaysSelected', 'nRaysAlive', 'nRaysGood', 'nRaysOut',
                 'nRaysOver', 'nRaysDead', 'intensity', 'intensityInRange', 'iteration')

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
        self.total2D += other.total2D
        self.total2D_RGB += other.total2D_RGB
        for mine, theirs in ((self.xaxis, other.xaxis), (self.yaxis, other.yaxis),
                             (self.caxis, other.caxis)):
            mine.total1D4 += theirs.total1D4
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
