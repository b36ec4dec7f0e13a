"""
Result containers of the KDE path: ``Density1D``, ``Density2D``, ``GridDensity`` and ``getContourLevels`` with the
call contract of getdist/densities.py (names, arguments, ``P[y, x]`` orientation, error class and messages); the
implementation is this package's own.

* Interpolation of a 1D density is a not-a-knot cubic spline (``NotAKnotSpline`` below) -- mathematically the
  interpolant ``splrep(x, P, s=0)`` builds in the reference (densities.py:155-170), written in the second-derivative
  form that the device kernel ``k_limits1d`` (csrc/limits1d.hip) also uses, so host and device limits agree.
* Equal-density credible limits (densities.py:186-248) and contour levels (densities.py:19-56) both look for the density
  level under which a given amount of (weighted) grid mass lies, on a ranked grid.  Batches of 1D limits run on the GPU (``MCSamples.getMargeStats`` -> ``gd_limits1d``); batches of 2D contour levels
  run on the GPU (``gd_contour_levels``); the functions here serve single objects and user-constructed grids.
"""

import numpy as np

defaultContours = (0.68, 0.95)


class DensitiesError(Exception):
    pass


# ---- shared numerics -----------------------------------------------------------------------------------------------
def _trapezoid_weights(shape):
    """Half weight on the first and last sample of every axis (the trapezoid rule on a regular grid)."""
    w = np.ones(shape)
    for axis, size in enumerate(shape):
        edge = np.ones(size)
        edge[0] = edge[-1] = 0.5
        w *= edge.reshape([-1 if a == axis else 1 for a in range(len(shape))])
    return w


def _ascending_mass(values, mass):
    """(values sorted ascending is implied) the masses in ascending order of ``values`` and their running sum."""
    ordered = mass[np.argsort(values)]
    return ordered, np.cumsum(ordered)


def getContourLevels(inbins, contours=defaultContours, missing_norm=0, half_edge=True):
    """
    Density levels enclosing the fractions ``contours`` of the probability, for a grid of any dimension
    (contract of densities.py:19-56).  With ``half_edge`` the outermost samples of every axis carry half their mass;
    the cells are ranked by their *un-halved* value, as the reference ranks them.  ``missing_norm`` is probability
    known to lie outside the grid.  The level is interpolated linearly in the cumulative mass between the two ranked
    cells around the crossing.
    """
    grid = np.asarray(inbins)
    mass = grid * _trapezoid_weights(grid.shape) if half_edge else grid
    wanted = (1 - np.asarray(contours, dtype=float)) * np.sum(mass) - missing_norm
    ranked, running = _ascending_mass(grid.reshape(-1), mass.reshape(-1))
    at = np.searchsorted(running, wanted)
    if np.any(at == 0):
        raise DensitiesError("Contour level outside plotted ranges")
    back = (running[at] - wanted) / (running[at] - running[at - 1])
    return ranked[at] * (1 - back) + back * ranked[at - 1]


class NotAKnotSpline:
    """
    Cubic spline through (x_i, y_i) with the not-a-knot end condition (the third derivative is continuous at x_1 and
    x_{n-2}), held as the second derivatives M_i at the nodes.  Outside [x_0, x_{n-1}] it evaluates to zero, which is
    what the reference's ``splev(..., ext=1)`` returns.
    """

    def __init__(self, x, y):
        from scipy.linalg import solve_banded

        x = np.asarray(x, dtype=float)
        y = np.asarray(y, dtype=float)
        n = x.size
        if n < 4 or y.size != n:
            raise DensitiesError("a cubic spline needs at least 4 grid points")
        h = np.diff(x)
        slope = np.diff(y) / h
        ab = np.zeros((5, n))  # banded storage, two sub- and two super-diagonals: ab[2 + i - j, j] = A[i, j]
        rhs = np.zeros(n)
        # interior rows: h_{i-1} M_{i-1} + 2 (h_{i-1} + h_i) M_i + h_i M_{i+1} = 6 (slope_i - slope_{i-1})
        i = np.arange(1, n - 1)
        ab[3, i - 1] = h[i - 1]
        ab[2, i] = 2 * (h[i - 1] + h[i])
        ab[1, i + 1] = h[i]
        rhs[i] = 6 * (slope[i] - slope[i - 1])
        # not-a-knot rows: (M_1 - M_0) / h_0 = (M_2 - M_1) / h_1 and its mirror image
        ab[2, 0], ab[1, 1], ab[0, 2] = h[1], -(h[0] + h[1]), h[0]
        ab[4, n - 3], ab[3, n - 2], ab[2, n - 1] = h[-1], -(h[-2] + h[-1]), h[-2]
        self.x, self.y, self.h, self.slope = x, y, h, slope
        self.M = solve_banded((2, 2), ab, rhs)

    def __call__(self, xq, derivative=0, extrapolate=False):
        """Values (or derivatives) at ``xq``; beyond the end nodes zero, or the end cubic continued if ``extrapolate``."""
        xq = np.asarray(xq, dtype=float)
        x, y, h, M = self.x, self.y, self.h, self.M
        k = np.clip(np.searchsorted(x, xq, side="right") - 1, 0, x.size - 2)
        t = xq - x[k]
        c1 = self.slope[k] - h[k] * (2 * M[k] + M[k + 1]) / 6
        c2 = M[k] / 2
        c3 = (M[k + 1] - M[k]) / (6 * h[k])
        if derivative == 0:
            out = y[k] + t * (c1 + t * (c2 + t * c3))
        elif derivative == 1:
            out = c1 + t * (2 * c2 + 3 * t * c3)
        elif derivative == 2:
            out = 2 * c2 + 6 * t * c3
        elif derivative == 3:
            out = 6 * c3
        else:
            out = np.zeros_like(t)
        return out if extrapolate else np.where((xq < x[0]) | (xq > x[-1]), 0.0, out)


# ---- containers ----------------------------------------------------------------------------------------------------
class GridDensity:
    """A density sampled on a regular grid, normalised or not.  ``axes`` lists the coordinate arrays in the index
    order of ``P``; ``view_ranges`` (optional) are the plot bounds in x, y order."""

    P = None
    axes = ()
    view_ranges = None
    spl = None

    def _normalizer(self, by):
        if by == "integral":
            return self.norm_integral()
        if by == "max":
            top = np.max(self.P)
            if top == 0:
                raise DensitiesError("no samples in bin")
            return top
        raise DensitiesError("Density: unknown normalization")

    def normalize(self, by="integral", in_place=False):
        """Divide the grid by its integral (``by='integral'``) or its maximum (``by='max'``); returns self."""
        scale = self._normalizer(by)
        if in_place:
            self.P /= scale
            self.spl = None
        else:
            self.setP(self.P / scale)
        return self

    def setP(self, P=None):
        """Install a grid of values (zeros when omitted); its shape must match the axes."""
        shape = tuple(ax.size for ax in self.axes)
        if P is None:
            P = np.zeros(shape)
        else:
            for got, want in zip(np.shape(P), shape):
                if got != want:
                    raise DensitiesError(f"Array size mismatch in Density arrays: P {got}, axis {want}")
        self.P = P
        self.spl = None

    def bounds(self):
        """Bounds in x, y, ... order: the view ranges if set, else the extent of the axes."""
        if self.view_ranges is not None:
            return self.view_ranges
        return [(ax[0], ax[-1]) for ax in reversed(self.axes)]

    def integrate(self, P):
        """Trapezoid-rule integral of a grid of this shape."""
        return np.sum(P * _trapezoid_weights(np.shape(P))) * self.spacing

    def norm_integral(self):
        return self.integrate(self.P)

    def getContourLevels(self, contours=defaultContours):
        return getContourLevels(self.P, contours)


class LimitGrid:
    """The spline-refined density of a Density1D, ranked for credible-limit searches (cf. InterpGridCache)."""

    __slots__ = ("factor", "bign", "grid", "norm", "sortgrid", "cumsum")


class Density1D(GridDensity):
    """1D marginalised density on a regular grid ``x``; calling it (or ``Prob``) interpolates with a cubic spline."""

    def __init__(self, x, P=None, view_ranges=None):
        self.x = x
        self.n = x.size
        self.axes = [x]
        self.spacing = x[1] - x[0]
        self.view_ranges = view_ranges
        self.likes = None
        self.setP(P)

    def bounds(self):
        """(min, max): the view range if set, else the grid extent."""
        return self.view_ranges if self.view_ranges is not None else (self.x[0], self.x[-1])

    def _initSpline(self):
        self.spl = NotAKnotSpline(self.x, self.P)

    def Prob(self, x, derivative=0):
        """Interpolated density (or its ``derivative``-th derivative) at ``x``: an array for sequences, a scalar for
        a scalar; zero outside the grid."""
        if self.spl is None:
            self._initSpline()
        if isinstance(x, (np.ndarray, list, tuple)):
            return self.spl(x, derivative)
        return self.spl(np.array([x]), derivative)[0]

    __call__ = Prob

    def initLimitGrids(self, factor=None):
        """
        Refine the density ``factor`` times (default: to about 20000 points, at least 2 per bin) with the spline and
        rank the refined values; ``norm`` is their trapezoid mass in refined-grid units.
        """
        if self.spl is None:
            self._initSpline()
        g = LimitGrid()
        g.factor = max(2, 20000 // self.n) if factor is None else factor
        g.bign = (self.n - 1) * g.factor + 1
        # the last refined abscissa can land an ulp beyond x[-1]: like the reference's plain splev call, continue the
        # end cubic there instead of returning zero
        g.grid = self.spl(self.x[0] + np.arange(g.bign) * self.spacing / g.factor, extrapolate=True)
        g.norm = np.sum(g.grid) - 0.5 * self.P[-1] - 0.5 * self.P[0]
        g.sortgrid = np.sort(g.grid)
        g.cumsum = np.cumsum(g.sortgrid)
        return g

    def _equal_density_level(self, g, p):
        """Density level enclosing probability p on the refined grid.  Above the lowest ranked value the level is
        interpolated towards the NEXT ranked value (densities.py:227 uses ix + 1 where getContourLevels uses ix - 1)."""
        wanted = (1 - p) * g.norm
        at = int(np.searchsorted(g.cumsum, wanted))
        if at == 0:
            return g.sortgrid[0]
        back = (g.cumsum[at] - wanted) / (g.cumsum[at] - g.cumsum[at - 1])
        return (1 - back) * g.sortgrid[at] + back * g.sortgrid[at + 1]

    def _crossings(self, g, level):
        """(lower, upper, lower_is_grid_edge, upper_is_grid_edge) where the refined density crosses ``level``."""
        step = self.spacing / g.factor
        above = np.flatnonzero(g.grid > level)
        open_bot, open_top = g.grid[0] >= level, g.grid[-1] >= level
        lower, upper = self.x[0], self.x[-1]
        if not open_bot:
            i = above[0] if above.size else 0
            lower = self.x[0] + (i - (g.grid[i] - level) / (g.grid[i] - g.grid[i - 1])) * step
        if not open_top:
            i = above[-1] if above.size else g.bign - 1
            upper = self.x[0] + (i + (g.grid[i] - level) / (g.grid[i] - g.grid[i + 1])) * step
        return lower, upper, open_bot, open_top

    def getLimits(self, p, interpGrid=None, accuracy_factor=None):
        """
        Equal-density credible interval(s) holding probability ``p`` (a number or a sequence): (min, max, has_min,
        has_top) -- has_min / has_top are True where the interval runs into the edge of the grid, i.e. the density is
        still above the level there and only a one-tail (or no) limit exists.  A sequence gives a list of tuples.
        """
        g = interpGrid or self.initLimitGrids(accuracy_factor)
        if np.ndim(p) == 0:
            return self._crossings(g, self._equal_density_level(g, p))
        return [self._crossings(g, self._equal_density_level(g, q)) for q in p]


class Density2D(GridDensity):
    """2D marginalised density with ``P[iy, ix]``; calling it (or ``Prob``) interpolates like a
    scipy RectBivariateSpline over (x, y)."""

    def __init__(self, x, y, P=None, view_ranges=None, mask=None):
        self._set_axes(x, y, (x[1] - x[0]) * (y[1] - y[0]), view_ranges)
        self.mask = mask
        self.setP(P)

    def _set_axes(self, x, y, spacing, view_ranges):
        self.x, self.y = x, y
        self.axes = [y, x]
        self.spacing = spacing
        self.view_ranges = view_ranges
        self.mask = self.likes = self.contours = self.spl = None

    @classmethod
    def _wrap(cls, x, y, P, view_ranges, spacing):
        """Batch constructor: axes, cell area and grid are already consistent (no checks, no copies)."""
        d = cls.__new__(cls)
        d._set_axes(x, y, spacing, view_ranges)
        d.P = P
        return d

    # ``P`` may still be in flight from the device when a batched call returns (the copy of one triangle's grids
    # overlaps whatever the caller does next): the first read waits for the copies of that call and surfaces its
    # per-grid status.  Grids set by the user or by the synchronous paths have no waiter.
    @property
    def P(self):
        d = self.__dict__
        waiter = d.get("_wait")
        if waiter is not None:
            waiter()  # raises DensitiesError for an empty grid -- on every read of it: the waiter stays in place
            d["_wait"] = None
        return d.get("_P")

    @P.setter
    def P(self, value):
        self.__dict__["_P"] = value

    def __getstate__(self):
        """Pickling / copying completes the grid first; the waiter (bound to a device context) is not part of the state."""
        self.P
        state = dict(self.__dict__)
        state.pop("_wait", None)
        state["spl"] = None
        return state

    @classmethod
    def _from_fields(cls, fields):
        """Batch constructor used for whole triangles: ``fields`` becomes the instance dictionary (the caller supplies
        every attribute _set_axes / setP would)."""
        d = cls.__new__(cls)
        d.__dict__ = fields
        return d

    def _initSpline(self):
        from scipy.interpolate import RectBivariateSpline

        self.spl = RectBivariateSpline(self.x, self.y, self.P.T, s=0)

    def __call__(self, *args, **kwargs):
        if self.spl is None:
            self._initSpline()
        return self.spl(*args, **kwargs)

    def Prob(self, x, y, grid=False):
        """Interpolated density at the points (x, y), or on their outer product with ``grid=True``."""
        return self(x, y, grid=grid)
