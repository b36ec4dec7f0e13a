"""
Result containers of the KDE path, API-compatible with getdist/densities.py:19-301 (Density1D,
Density2D, GridDensity, getContourLevels).  The grids themselves are produced on the GPU; these
classes only hold them and provide the cheap host-side post-processing (normalisation, trapezoid
integrals, spline look-ups, credible limits, contour levels) on grid-sized data.
"""

import numpy as np
from scipy.interpolate import RectBivariateSpline, splev, splrep


class DensitiesError(Exception):
    pass


defaultContours = (0.68, 0.95)


def getContourLevels(inbins, contours=defaultContours, missing_norm=0, half_edge=True):
    """
    Density levels enclosing the given probability fractions (densities.py:19-56).

    Edge bins count half along every axis when ``half_edge``; the ordering comes from the *un-halved*
    grid while the running sum uses the halved one, and the level is linearly interpolated between the
    straddling sorted entries -- all as in the reference.
    """
    inbins = np.asarray(inbins)
    if half_edge:
        mass = inbins.copy()
        for axis in range(mass.ndim):
            first = [slice(None)] * mass.ndim
            last = [slice(None)] * mass.ndim
            first[axis] = 0
            last[axis] = -1
            mass[tuple(last)] /= 2
            mass[tuple(first)] /= 2
    else:
        mass = inbins
    total = np.sum(mass)
    targets = (1 - np.array(contours)) * total - missing_norm
    order = inbins.reshape(-1).argsort()
    ordered = mass.reshape(-1)[order]
    running = np.cumsum(ordered)
    levels = np.zeros(len(contours))
    for i, ix in enumerate(np.searchsorted(running, targets)):
        if ix == 0:
            raise DensitiesError("Contour level outside plotted ranges")
        step = running[ix] - running[ix - 1]
        d = (running[ix] - targets[i]) / step
        levels[i] = ordered[ix] * (1 - d) + d * ordered[ix - 1]
    return levels


class GridDensity:
    """Base class for density grids (densities.py:59-129)."""

    P = None
    axes = ()
    view_ranges = None
    spl = None

    def normalize(self, by="integral", in_place=False):
        if by == "integral":
            norm = self.norm_integral()
        elif by == "max":
            norm = np.max(self.P)
            if norm == 0:
                raise DensitiesError("no samples in bin")
        else:
            raise DensitiesError("Density: unknown normalization")
        if in_place:
            self.P /= norm
        else:
            self.setP(self.P / norm)
        self.spl = None
        return self

    def setP(self, P=None):
        if P is not None:
            for size, ax in zip(P.shape, self.axes):
                if size != ax.size:
                    raise DensitiesError(f"Array size mismatch in Density arrays: P {size}, axis {ax.size}")
            self.P = P
        else:
            self.P = np.zeros([ax.size for ax in self.axes])
        self.spl = None

    def bounds(self):
        if self.view_ranges is not None:
            return self.view_ranges
        b = [(ax[0], ax[-1]) for ax in self.axes]
        b.reverse()
        return b

    def getContourLevels(self, contours=defaultContours):
        return getContourLevels(self.P, contours)


class _LimitGrid:
    __slots__ = ("factor", "bign", "grid", "norm", "sortgrid", "cumsum")


class Density1D(GridDensity):
    """1D marginalised density on a regular grid (densities.py:132-248); callable like a spline."""

    def __init__(self, x, P=None, view_ranges=None):
        self.n = x.size
        self.axes = [x]
        self.x = x
        self.view_ranges = view_ranges
        self.spacing = x[1] - x[0]
        self.likes = None
        self.setP(P)

    def bounds(self):
        if self.view_ranges is not None:
            return self.view_ranges
        return self.x[0], self.x[-1]

    def _initSpline(self):
        self.spl = splrep(self.x, self.P, s=0)

    def Prob(self, x, derivative=0):
        if self.spl is None:
            self._initSpline()
        if isinstance(x, (np.ndarray, list, tuple)):
            return splev(x, self.spl, derivative, ext=1)
        return splev([x], self.spl, derivative, ext=1)[0]

    __call__ = Prob

    def integrate(self, P):
        return ((P[0] + P[-1]) / 2 + np.sum(P[1:-1])) * self.spacing

    def norm_integral(self):
        return self.integrate(self.P)

    def initLimitGrids(self, factor=None):
        """Fine spline-resampled grid, sorted, with running sum (densities.py:186-205)."""
        if self.spl is None:
            self._initSpline()
        g = _LimitGrid()
        g.factor = max(2, 20000 // self.n) if factor is None else factor
        g.bign = (self.n - 1) * g.factor + 1
        g.grid = splev(self.x[0] + np.arange(g.bign) * self.spacing / g.factor, self.spl)
        g.norm = np.sum(g.grid) - (0.5 * self.P[-1]) - (0.5 * self.P[0])
        g.sortgrid = np.sort(g.grid)
        g.cumsum = np.cumsum(g.sortgrid)
        return g

    def getLimits(self, p, interpGrid=None, accuracy_factor=None):
        """Equal-density credible limits (densities.py:207-248): (min, max, has_min, has_top) per p."""
        g = interpGrid or self.initLimitGrids(accuracy_factor)
        parr = np.atleast_1d(p)
        targets = (1 - parr) * g.norm
        out = []
        fine = self.spacing / g.factor
        for ix, target in zip(np.searchsorted(g.cumsum, targets), targets):
            level = g.sortgrid[ix]
            if ix > 0:
                step = g.cumsum[ix] - g.cumsum[ix - 1]
                frac = (g.cumsum[ix] - target) / step
                level = (1 - frac) * level + frac * g.sortgrid[ix + 1]  # sic: ix+1 (densities.py:227)
            lim_bot = g.grid[0] >= level
            if lim_bot:
                mn = self.x[0]
            else:
                i = np.argmax(g.grid > level)
                d = (g.grid[i] - level) / (g.grid[i] - g.grid[i - 1])
                mn = self.x[0] + (i - d) * fine
            lim_top = g.grid[-1] >= level
            if lim_top:
                mx = self.x[-1]
            else:
                i = g.bign - np.argmax(g.grid[::-1] > level) - 1
                d = (g.grid[i] - level) / (g.grid[i] - g.grid[i + 1])
                mx = self.x[0] + (i + d) * fine
            if parr is not p:
                return mn, mx, lim_bot, lim_top
            out.append((mn, mx, lim_bot, lim_top))
        return out


class Density2D(GridDensity):
    """2D marginalised density, P indexed [y, x] (densities.py:251-301); callable like RectBivariateSpline."""

    def __init__(self, x, y, P=None, view_ranges=None, mask=None):
        self.x = x
        self.y = y
        self.axes = [y, x]
        self.view_ranges = view_ranges
        self.mask = mask
        self.spacing = (self.x[1] - self.x[0]) * (self.y[1] - self.y[0])
        self.likes = None
        self.contours = None
        self.setP(P)

    @classmethod
    def _wrap(cls, x, y, P, view_ranges, spacing):
        """Constructor without the shape checks, for batches of grids whose axes were built with them."""
        d = cls.__new__(cls)
        d.x, d.y, d.axes, d.view_ranges, d.mask, d.spacing = x, y, [y, x], view_ranges, None, spacing
        d.likes = d.contours = d.spl = None
        d.P = P
        return d

    def integrate(self, P):
        inner = np.sum(P[1:-1, 1:-1])
        corners = (P[0, 0] + P[0, -1] + P[-1, 0] + P[-1, -1]) / 4.0
        edges = (np.sum(P[1:-1, 0]) + np.sum(P[0, 1:-1]) + np.sum(P[1:-1, -1]) + np.sum(P[-1, 1:-1])) / 2.0
        return (inner + corners + edges) * self.spacing

    def norm_integral(self):
        return self.integrate(self.P)

    def _initSpline(self):
        self.spl = RectBivariateSpline(self.x, self.y, self.P.T, s=0)

    def Prob(self, x, y, grid=False):
        return self.__call__(x, y, grid=grid)

    def __call__(self, *args, **kwargs):
        if self.spl is None:
            self._initSpline()
        return self.spl(*args, **kwargs)
