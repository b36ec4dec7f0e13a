"""
``MCSamples``: the host-side mirror of GetDist's analysis object for the KDE / weighted-statistics hot
path, driving libgdhip.so (HIP kernels on one MI355X) through the ctypes C ABI.

Same method names, keyword arguments, defaults (analysis_defaults.ini) and result types as
getdist/mcsamples.py + getdist/chains.py for that path; every O(N) step and every O(F^2) grid step runs
on the GPU, including the root finders of the bandwidth selection (MINPACK hybrd as scipy's fsolve runs it for the
1D Botev fixed point, Brent's method for the 2D one: csrc/solvers.hpp).  What stays in Python is the reference's
*scalar* logic (ranges and limits, bin edges, bandwidth branch selection, fallbacks), evaluated in the reference's
expression order so that bin edges are bit-identical.

Additive API (not in the reference): ``get1DDensities``, ``get2DDensities`` and ``triangleDensities``
compute many densities in batched kernel launches; the per-name/per-pair methods are thin views over
them.  There is no CPU fallback: without the library or a GPU, construction raises.
"""

import logging
import os
import threading
import time

import numpy as np

from ._lib import Context, GdhipError
from .densities import Density1D, Density2D, DensitiesError

# analysis_defaults.ini:1-76 (the ini always overrides the class literals, mcsamples.py:491-492)
DEFAULT_SETTINGS = dict(
    ignore_rows=0.0, min_weight_ratio=1e-30, contours=[0.68, 0.95, 0.99], credible_interval_threshold=0.05,
    range_ND_contour=-1, range_confidence=0.001, corr_length_thin=0, corr_length_steps=15, converge_test_limit=0.95, fine_bins=1024, smooth_scale_1D=-1.0,
    boundary_correction_order=1, mult_bias_correction_order=1, smooth_scale_2D=-1.0, max_corr_2D=0.99,
    fine_bins_2D=256, use_effective_samples_2D=False, max_scatter_points=2000, num_bins=100, num_bins_2D=40)


class WeightedSampleError(Exception):
    pass


def _where_rows(where, numrows):
    """Row indices of a ``where=`` index array with numpy's semantics of x[where]: negative indices count from the end,
    anything outside [-numrows, numrows) raises IndexError (it used to wrap around silently)."""
    ix = np.asarray(where).astype(np.int64).ravel()
    if ix.size and (ix.min() < -numrows or ix.max() >= numrows):
        bad = ix[(ix < -numrows) | (ix >= numrows)][0]
        raise IndexError("index %d is out of bounds for axis 0 with size %d" % (int(bad), int(numrows)))
    return np.where(ix < 0, ix + numrows, ix)


class MCSamplesError(WeightedSampleError):
    pass


class SettingError(MCSamplesError):
    pass


class BandwidthError(MCSamplesError):
    pass


class ParamError(MCSamplesError):
    pass


try:
    # GetDist's plotting layer -- the caller this package is a drop-in for -- recognises a parameter object by
    # isinstance(param, getdist.paramnames.ParamInfo) (plots.py:607,1981,2027).  Where GetDist is installed beside this
    # package our parameter objects therefore derive from its class; without it they stand alone.
    from getdist.paramnames import ParamInfo as _PlotParamInfo
except Exception:  # noqa: BLE001 -- not installed (or not importable): nothing of the path needs it
    _PlotParamInfo = object


class ParamInfo(_PlotParamInfo):
    """Per-parameter state bag; the attributes the hot path reads and writes (paramnames.py:69-154)."""

    def __init__(self, name, label=None):
        if _PlotParamInfo is not object:
            super().__init__(name=name, label=label or name)
        self.name = name
        self.label = label or name
        self.isDerived = False
        self.limmin = self.limmax = None
        self.has_limits_bot = self.has_limits_top = self.has_limits = False
        self.periodic = False
        self.N_eff_kde = None
        self.kde_h = None
        self.renames = []   # alternative names a caller may use for this parameter (paramnames.py:86)
        self.comment = ""

    def getLabel(self):
        """paramnames.py:120-124"""
        return self.label if self.label else self.name

    def latexLabel(self):
        """paramnames.py:126-130: what the plotting layer writes on an axis"""
        return "$" + self.label + "$" if self.label else self.name

    def __repr__(self):
        return "ParamInfo(%s)" % self.name


class ParamLimit:
    """A marginalised parameter limit (types.py:652-716): lower, upper and which tails are constrained."""

    def __init__(self, minmax, tag="two"):
        self.lower, self.upper = minmax[0], minmax[1]
        self.twotail = tag == "two"
        self.onetail_upper = tag == ">"
        self.onetail_lower = tag == "<"

    def limitTag(self):
        return "two" if self.twotail else (">" if self.onetail_upper else ("<" if self.onetail_lower else "none"))

    def __str__(self):
        return f"{self.lower:g} {self.upper:g} {self.limitTag()}"


class MargeStats:
    """The numbers of types.MargeStats (types.py:718-800): per-parameter mean, err and limits per contour."""

    def __init__(self, names, limits):
        self.names = names
        self.limits = limits
        self.hasBestFit = False

    def parWithName(self, name):
        for p in self.names:
            if p.name == name:
                return p
        return None


class LikeStats:
    """The numbers of types.LikeStats (types.py:900-939): posterior statistics of the sample log-likelihoods; the N-D
    confidence-region limits and the best-fit sample live on ``names[i]`` (ND_limit_bot / ND_limit_top / bestfit_sample)."""

    def __init__(self):
        self.logLike_sample = self.logMeanInvLike = self.meanLogLike = self.logMeanLike = None
        self.complexity = self.varLogLike = None
        self.names = []

    def likeSummary(self):
        text = "Best fit sample -log(Like) = %f\n" % self.logLike_sample
        if self.logMeanInvLike:
            text += "Ln(mean 1/like) = %f\n" % self.logMeanInvLike
        text += "mean(-Ln(like)) = %f\n" % self.meanLogLike
        text += "-Ln(mean like)  = %f\n" % self.logMeanLike
        text += "2*Var(Ln(like)) = %f\n" % (self.varLogLike * 2.0)
        return text


class ParamConfidenceData:
    """Handle returned by initParamConfidenceData (chains.py:176-178 namedtuple in the reference)."""

    def __init__(self, col, start, end, weights=None, vec=None):
        self.col, self.start, self.end = col, start, end
        self.weights = weights  # alternative weights (host, full length) or None
        self.vec = vec          # host vector the handle refers to when it is not a resident column


class ParamNames:
    def __init__(self, names, labels=None):
        labels = labels or [None] * len(names)
        self.names = [ParamInfo(n, lab) for n, lab in zip(names, labels)]

    def parWithName(self, name, error=False, renames=None):
        """paramnames.py:232-255: the parameter called ``name`` -- by its own name, by one of its ``renames``, or through the
        optional ``renames`` mapping {name: alternative name(s)} the plotting layer passes along."""
        if not isinstance(name, str):
            raise ValueError('"name" must be a parameter name string not %s: %s' % (type(name), name))

        def alts(key):
            v = renames.get(key, []) if renames else []
            return [v] if isinstance(v, str) else list(v)

        asked = {name, *alts(name)}
        for p in self.names:
            if asked & {p.name, *getattr(p, "renames", []), *alts(p.name)}:
                return p
        if error:
            raise ParamError("parameter name not found: %s" % name)
        return None

    def hasParam(self, name):
        return self.numberOfName(name) != -1

    def getMatches(self, pattern, strings=False):
        """paramnames.py:299-307: parameters whose name matches a shell-style pattern"""
        import fnmatch

        return [(p.name if strings else p) for p in self.names if fnmatch.fnmatchcase(p.name, pattern)]

    def parsWithNames(self, names, error=False, renames=None):
        """paramnames.py:273-297: ParamInfo per name (None where a name is unknown and ``error`` is false for it); names
        holding * or ? expand to every match; ``error`` may be one flag or one per name."""
        if isinstance(names, str):
            names = [names]
        flags = list(error) if isinstance(error, (list, tuple)) else [error]
        if len(flags) < len(names):
            flags = len(names) * flags
        out = []
        for nm, flag in zip(names, flags):
            if isinstance(nm, ParamInfo):
                out.append(nm)
            elif "?" in nm or "*" in nm:
                out += self.getMatches(nm)
            else:
                out.append(self.parWithName(nm, flag, renames))
        return out

    def getRenames(self, keep_empty=False):
        """paramnames.py:324-332"""
        return {p.name: list(getattr(p, "renames", [])) for p in self.names if keep_empty or getattr(p, "renames", [])}

    def numParams(self):
        return len(self.names)

    def labels(self):
        return [p.label for p in self.names]

    def numberOfName(self, name):
        for i, p in enumerate(self.names):
            if p.name == name:
                return i
        return -1

    def list(self):
        return [p.name for p in self.names]

    def numNonDerived(self):
        return len([p for p in self.names if not p.isDerived])

    def deleteIndices(self, indices):
        gone = set(indices)
        self.names = [p for i, p in enumerate(self.names) if i not in gone]


class ParamBounds:
    """Hard prior ranges and periodic flags (parampriors.py:6-139, the parts the hot path uses)."""

    def __init__(self):
        self.lower, self.upper, self.periodic = {}, {}, set()

    def setRange(self, name, rng):
        lo, hi = rng[0], rng[1]
        if len(rng) > 2 and rng[2] in (True, "periodic"):
            self.periodic.add(name)
        elif name in self.periodic:
            self.periodic.discard(name)
        for store, v in ((self.lower, lo), (self.upper, hi)):
            if v is None or (isinstance(v, str) and v in ("N", "None")):
                store.pop(name, None)
            else:
                store[name] = float(v)

    def setFixed(self, name, value):
        """parampriors.py:78-79: a fixed parameter is a zero-width range"""
        self.lower[name] = self.upper[name] = float(value)

    def fixedValue(self, name):
        lo, hi = self.lower.get(name), self.upper.get(name)
        return lo if lo is not None and lo == hi else None

    def getLower(self, name):
        return self.lower.get(name)

    def getUpper(self, name):
        return self.upper.get(name)


# keys of a reference analysis .ini that are settings of this path although they are not in analysis_defaults.ini by
# that name: the contour list in its numbered form and the limit-type overrides (mcsamples.py:417-433)
_INI_EXTRA = ("num_contours", "force_twotail")


def _read_ini_settings(ini):
    """
    The analysis settings of a GetDist .ini file (``key = value`` lines, ``#`` comments; inifile.py:100-180) that this
    path knows; other keys (plot options, file lists) are ignored like the reference ignores what it does not read.
    ``num_contours`` + ``contour1..N`` become ``contours`` (unless ``contours`` itself is given, mcsamples.py:420-424),
    ``force_twotail`` and ``max_frac_twotailN`` are kept (mcsamples.py:417,426-431).  A dict is passed through.
    """
    if isinstance(ini, dict):
        raw = {str(k): v for k, v in ini.items()}
    else:
        raw = {}
        with open(ini, encoding="utf-8-sig") as f:
            for line in f:
                line = line.split("#", 1)[0].strip()
                if "=" not in line:
                    continue
                key, value = (t.strip() for t in line.split("=", 1))
                if value != "":
                    raw[key] = value
    out = {}
    for key, value in raw.items():
        if key in DEFAULT_SETTINGS:
            out[key] = [float(v) for v in value.split()] if (key == "contours" and isinstance(value, str)) else value
        elif key == "force_twotail" or key.startswith("max_frac_twotail"):
            out[key] = value
    if "contours" not in out and "num_contours" in raw:
        n = int(raw["num_contours"])
        missing = [i + 1 for i in range(n) if "contour%d" % (i + 1) not in raw]
        if missing:
            raise SettingError("num_contours = %d but contour%d is not set" % (n, missing[0]))
        out["contours"] = [float(raw["contour%d" % (i + 1)]) for i in range(n)]
    return out


def _stack_rows(parts):
    """
    np.vstack / np.hstack of per-chain arrays.  Chains that are consecutive row ranges of one column-major block (the
    binary chain cache, chainfiles.read_soa_cache) are returned as a view of that block -- no host copy, and the
    page-locked columns go to the device as they are; anything else is copied into a new column-major array.
    """
    first = parts[0]
    if len(parts) == 1:
        return first
    adjacent = all(p.dtype == np.float64 and p.strides == first.strides and p.shape[1:] == first.shape[1:] for p in parts)
    if adjacent:
        addr = first.ctypes.data
        for p in parts:
            if p.ctypes.data != addr:
                adjacent = False
                break
            addr += p.shape[0] * p.strides[0]
        adjacent = adjacent and first.strides[0] == 8
    total = sum(p.shape[0] for p in parts)
    if adjacent:
        shape = (total,) + first.shape[1:]
        return np.lib.stride_tricks.as_strided(first, shape=shape, strides=first.strides, writeable=False)
    out = np.empty((total,) + first.shape[1:], dtype=np.float64, order="F")
    a = 0
    for p in parts:
        out[a:a + p.shape[0]] = p
        a += p.shape[0]
    return out


_HOSTLOG = [] if os.environ.get("GETDIST_AMD_HOSTLOG") else None


def _hostlog(what):
    """Host-side timeline of a batched call (GETDIST_AMD_HOSTLOG=1; read by scripts and tests only)."""
    if _HOSTLOG is not None:
        import time

        _HOSTLOG.append((time.perf_counter(), what))


class _Plan(list):
    """The per-pair bandwidth records (dicts) of _bandwidth_plan plus the same fields as arrays over the pairs
    (``arr``: branch code 0/1/2 = A/B/C, has_limits, corr, rangex, rangey and -- once the effective sample numbers are
    known -- neff, fallback_t): what sits between two kernels of a batched call is evaluated on the arrays."""

    arr = None


class _Done:
    """A finished future (the lag probe that came back with the quantile select)."""

    def __init__(self, value):
        self._value = value

    def result(self):
        return self._value


class _FastThreadSwitch:
    """While a helper thread drives a second stream, a thread returning from a C call would wait for the GIL up to the
    interpreter's switch interval (5 ms by default) whenever the other thread is in Python scalar code -- longer than
    most kernels here.  100 us for the duration of a batched call; restored on exit."""

    def __init__(self, active=True):
        self.active = active

    def __enter__(self):
        if self.active:
            import sys

            self.old = sys.getswitchinterval()
            sys.setswitchinterval(float(os.environ.get("GETDIST_AMD_SWITCH_INTERVAL", 1e-4)))
        return self

    def __exit__(self, *exc):
        if self.active:
            import sys

            sys.setswitchinterval(self.old)
        return False


class _Phase:
    """Wall-clock phase accounting (enabled by GETDIST_AMD_TIMING=1; syncs the stream at phase edges)."""

    def __init__(self, mc, name):
        self.mc, self.name = mc, name

    def __enter__(self):
        if self.mc._timing:
            self.mc.ctx.sync()
            self.t0 = time.perf_counter()

    def __exit__(self, *a):
        if self.mc._timing:
            self.mc.ctx.sync()
            self.mc.timings[self.name] = self.mc.timings.get(self.name, 0.0) + time.perf_counter() - self.t0


def _g2_markov_vs_second_order(tran):
    """
    Likelihood-ratio statistic G^2 of the first-order Markov model against the second-order one for the transition
    counts tran[a, b, c] of a binary chain (mcsamples.py:1072-1089): the Markov fit of cell (a, b, c) is
    n(a,b,.) n(.,b,c) / n(.,b,.); empty cells drop out.  Terms are added in C order, like the reference's loops.
    """
    import math

    n_ab = tran.sum(axis=2)
    n_bc = tran.sum(axis=0)
    n_b = tran.sum(axis=(0, 2))
    g2 = 0.0
    for (a, b, c), focus in np.ndenumerate(tran):
        if focus != 0:
            fitted = float(n_ab[a, b] * n_bc[b, c]) / float(n_b[b])
            g2 += math.log(float(focus) / fitted) * float(focus)
    return 2 * g2


def _g2_independence_vs_markov(tran2, thin_rows):
    """G^2 of independence against first-order Markov for pair counts tran2[a, b] (mcsamples.py:1124-1139); None where the
    reference gives up ("Raftery and Lewis estimator had problems")."""
    rows, cols = tran2.sum(axis=1), tran2.sum(axis=0)
    g2 = 0
    for (a, b), focus in np.ndenumerate(tran2):
        if focus != 0:
            fitted = float(rows[a] * cols[b]) / float(thin_rows - 1)
            if fitted <= 0 or focus <= 0:
                return None
            g2 += np.log(float(focus) / fitted) * float(focus)
    return 2 * g2


def _set_edge_mask_2d(parx, pary, prior_mask, winw):
    """mcsamples.py:1688-1703: half weight on a limit's edge bins, zero beyond -- on non-periodic axes only"""
    if not parx.periodic:
        if parx.has_limits_bot:
            prior_mask[:, winw] /= 2
            prior_mask[:, :winw] = 0
        if parx.has_limits_top:
            prior_mask[:, -(winw + 1)] /= 2
            prior_mask[:, -winw:] = 0
    if not pary.periodic:
        if pary.has_limits_bot:
            prior_mask[winw, :] /= 2
            prior_mask[:winw] = 0
        if pary.has_limits_top:
            prior_mask[-(winw + 1), :] /= 2
            prior_mask[-winw:, :] = 0


def _set_all_edge_mask_2d(prior_mask, winw, periodic_x=False, periodic_y=False):
    """mcsamples.py:1705-1712: zero the padding margins along non-periodic axes"""
    if not periodic_x:
        prior_mask[:, :winw] = 0
        prior_mask[:, -winw:] = 0
    if not periodic_y:
        prior_mask[:winw] = 0
        prior_mask[-winw:, :] = 0


class MCSamples:
    """
    Weighted samples resident in HBM + the KDE hot path.  Constructor arguments follow
    mcsamples.py:149-161 (``samples`` may be an (N, n) array or a list of per-chain arrays;
    ``ranges`` maps name -> (lower, upper[, True|'periodic'])).  ``device`` selects the GPU.
    """

    def __init__(self, root=None, ini=None, settings=None, ranges=None, samples=None, weights=None, loglikes=None,
                 temperature=None, names=None, labels=None, label=None, name_tag=None, sampler=None, device=0,
                 _context_factory=None, **kwargs):
        self.sampler = sampler or "mcmc"
        self.temperature, self.cooled = temperature, 1
        self._likeStats = None
        self._loglikes_col = None
        self.label, self.name_tag = label, name_tag
        self.root = root
        self.raise_on_bandwidth_errors = False
        self.no_warning_params = []          # mcsamples.py:259-260,438-439: parameters whose 1D bandwidth fallback is silent
        self.no_warning_chi2_params = True
        self.chain_offsets = None
        self.chains = None
        self.ctx = None
        for k, v in DEFAULT_SETTINGS.items():
            setattr(self, k, v)
        if "ignore_rows" in kwargs:  # mcsamples.py:246-250: the keyword is a setting
            settings = dict(settings or {})
            settings["ignore_rows"] = kwargs.pop("ignore_rows")
        chain_exclude, no_cache = kwargs.pop("_chain_exclude", None), kwargs.pop("_no_cache", False)
        # multi-rank jobs: a parallel.ColumnShare -- this rank uploads only its block of columns over PCIe and the ranks
        # broadcast their blocks to one another over xGMI (gd_upload_shard / gd_comm_share_columns); additive keyword
        self._column_share = kwargs.pop("column_share", None)
        if kwargs:
            raise TypeError("unexpected keyword arguments: %s" % ", ".join(kwargs))
        if ini is not None:
            settings = dict(_read_ini_settings(ini), **(settings or {}))  # the dict takes preference (:472-499)
        if settings:
            self.updateSettings(settings, doUpdate=False)
        if self.sampler == "nested" and not np.isclose(self.ignore_rows, 0):
            raise ValueError("Should not remove burn-in from Nested Sampler samples.")
        self.ranges = ParamBounds()
        derived = None
        if root is not None and samples is None:
            # mcsamples.py:47-146, chains.py:1368-1405: the chain files, side files and (when fresh) the binary cache
            from . import chainfiles

            self.ctx = (_context_factory or Context)(device)  # page-locked landing buffer for the binary cache
            loaded = chainfiles.read_root(root, chain_exclude, no_cache,
                                          alloc=getattr(self.ctx, "pinned_block", None))
            samples, weights, loglikes = loaded["samples"], loaded["weights"], loaded["loglikes"]
            names = names or loaded["names"]
            labels = labels or loaded["labels"]
            derived = loaded["derived"]
            for nm, rng in loaded["ranges"].items():
                self.ranges.setRange(nm, rng)
            self.name_tag = self.name_tag or os.path.basename(root)
        ignore_lines = int(self.ignore_rows)
        if samples is None:
            raise MCSamplesError("samples are required")
        for nm, rng in (ranges or {}).items():
            self.ranges.setRange(nm, rng)
        samples, weights, loglikes, fixed = self._read_chains(samples, weights, loglikes, ignore_lines)
        self.samples = samples
        self.loglikes = loglikes
        self.numrows, n_all = samples.shape[0], samples.shape[1] + len(fixed)
        self._user_weights = weights is not None
        self.weights = weights
        if names is None:
            names = ["param%d" % (i + 1) for i in range(n_all)]
        if len(names) != n_all:
            raise MCSamplesError("names do not match the number of sample columns")
        self.paramNames = ParamNames(list(names), labels)
        if derived is not None:
            for par, d in zip(self.paramNames.names, derived):
                par.isDerived = bool(d)
        for ix, value in fixed:  # chains.py:1555-1559: a parameter that never moves becomes a zero-width range
            self.ranges.setFixed(self.paramNames.names[ix].name, value)
        self.paramNames.deleteIndices([ix for ix, _ in fixed])
        self.n = samples.shape[1]
        self.index = {p.name: i for i, p in enumerate(self.paramNames.names)}
        # _context_factory is a TEST hook (tests/fake_ctx.py drives the host logic on CPU); the product always uses
        # the HIP library and raises if it or a GPU is missing
        if getattr(self, "ctx", None) is None:
            self.ctx = (_context_factory or Context)(device)
        self._timing = os.environ.get("GETDIST_AMD_TIMING", "0") == "1"
        self.timings = {}
        self.density1D = {}
        self._idx_cols = {}
        self.shade_likes_is_mean_loglikes = False  # mcsamples.py:233
        self._context_factory = _context_factory or Context
        self._device = device
        self._lane, self._nlanes = 0, 1
        # set up front: helper threads assign these while another thread may be iterating this object's __dict__
        self._lag_prefetch = self._pending_results = self._parked = None
        self._helper_exec = None
        self._lane_exec = None
        self._twin = None
        self._chain_stats_cache = {}
        self.needs_update = True
        self._upload(filter_weights=False)  # the per-chain filter already ran (makeSingle passes min_weight_ratio=-1)
        self.updateBaseStatistics()

    def _read_chains(self, samples, weights, loglikes, ignore_lines):
        """
        loadChains + readChains for array input (chains.py:1405-1443, mcsamples.py:501-528): per chain, drop
        ``ignore_lines`` leading rows, drop rows below min_weight_ratio of THAT chain's maximum weight
        (chains.py:1017-1027), drop the burn-in fraction (chains.py:1047-1061); delete the parameters that do not move
        (decided on the first chain, chains.py:1029-1045,1548-1555); stack the chains and record their offsets
        (makeSingle, chains.py:1488-1503).  Returns (samples, weights, loglikes, [(fixed column, value)]).
        """
        is_list = isinstance(samples, (list, tuple)) and len(samples) and np.ndim(samples[0]) == 2
        if is_list:
            chains = [[np.asarray(c), None if weights is None else np.asarray(weights[i], dtype=np.float64),
                       None if loglikes is None else np.asarray(loglikes[i], dtype=np.float64)]
                      for i, c in enumerate(samples)]
        else:
            if isinstance(samples, (list, tuple)):  # a list of parameter vectors (chains.py:287-288)
                samples = np.hstack([np.asarray(x).reshape(-1, 1) for x in samples])
            samples = np.asarray(samples)
            if samples.ndim == 1:
                samples = samples.reshape(-1, 1)
            chains = [[samples, None if weights is None else np.asarray(weights, dtype=np.float64),
                       None if loglikes is None else np.asarray(loglikes, dtype=np.float64)]]
        ignore_frac = 0 if int(self.ignore_rows) else self.ignore_rows
        mwr = self.min_weight_ratio
        for ch in chains:
            if ignore_lines:
                ch[:] = [None if v is None else v[ignore_lines:] for v in ch]
            w = ch[1]
            if w is not None and mwr is not None and mwr >= 0 and w.size:
                mx = np.max(w)
                if np.min(w) < mx * mwr:
                    keep = w > mx * mwr
                    ch[:] = [None if v is None else v[keep] for v in ch]
            if ignore_frac:
                ix = int(ignore_frac) if ignore_frac >= 1 else int(round(ch[0].shape[0] * ignore_frac))
                ch[:] = [None if v is None else v[ix:] for v in ch]
        first = chains[0][0]
        fixed = []
        if first.shape[0]:
            for i in range(first.shape[1]):
                if np.isclose(first[0, i], first[-1, i], equal_nan=True):
                    mean = np.average(first[:, i])
                    if np.allclose(first[:, i], mean, rtol=1e-12, atol=0, equal_nan=True):
                        fixed.append((i, mean))
        if fixed:
            gone = [i for i, _ in fixed]
            for ch in chains:
                ch[0] = np.delete(ch[0], gone, 1)
        if is_list:
            self.chain_offsets = np.cumsum(np.array([0] + [ch[0].shape[0] for ch in chains]))
            samples = _stack_rows([ch[0] for ch in chains])
            weights = None if chains[0][1] is None else _stack_rows([ch[1] for ch in chains])
            loglikes = None if chains[0][2] is None else _stack_rows([ch[2] for ch in chains])
        else:
            samples, weights, loglikes = chains[0]
        weights = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        return samples, weights, loglikes, fixed

    # ---- state -----------------------------------------------------------------------------------------
    def _second_lane(self):
        """
        A shallow twin of this object on a second context (= second stream) that borrows the resident sample set:
        independent pairs dealt to the two lanes overlap one lane's host-side scalar work, launch gaps and result
        copies with the other lane's kernels.  Parameter state is shared (read-only while densities are made).
        """
        own = ("ctx", "_idx_cols", "density1D", "timings", "_helper_exec", "_lane_exec", "_twin", "_lane")
        if self._twin is None:
            import copy

            twin = copy.copy(self)
            twin.ctx = self._context_factory(self._device)
            twin.ctx.attach(self.ctx)
            twin._idx_cols, twin.density1D, twin.timings = {}, {}, {}
            twin._helper_exec = twin._lane_exec = twin._twin = None
            twin._lane = 1
            self._twin = twin
        self._nlanes = 2
        # settings and statistics may have changed since the twin was made: everything but the lane's own state follows
        # (a snapshot: a helper thread may be adding attributes to this object at the same time)
        self._twin.__dict__.update({k: v for k, v in list(self.__dict__.items()) if k not in own})
        return self._twin

    def _lane_thread(self, twin):
        """The thread that drives the second lane (its own helper thread stays free for the lane's inner overlap)."""
        if self._lane_exec is None:
            from concurrent.futures import ThreadPoolExecutor

            self._lane_exec = ThreadPoolExecutor(max_workers=1, thread_name_prefix="gdhip-lane",
                                                 initializer=twin.ctx.bind_thread)
        return self._lane_exec

    def _finish_pending(self):
        """Complete a lazily delivered batched call (its grids view page-locked memory of this object's contexts)."""
        pend, self._pending_results = getattr(self, "_pending_results", None), None
        if pend is not None:
            pend.wait()
        if getattr(self.ctx, "h", None) is not None and hasattr(self.ctx, "batch2d_finish"):
            self.ctx.batch2d_finish()  # the library hands the call's device blocks back (before a second context goes away)
        twin = getattr(self, "_twin", None)
        if twin is not None and getattr(twin, "_pending_results", None) is not None:
            pend, twin._pending_results = twin._pending_results, None
            pend.wait()

    def _drop_second_lane(self):
        self._finish_pending()
        if getattr(self, "_lane_exec", None) is not None:
            self._lane_exec.shutdown(wait=True)
        self._lane_exec = None
        if self._twin is not None:
            if self._twin._helper_exec is not None:
                self._twin._helper_exec.shutdown(wait=True)
            self._twin.ctx.close()
            self._twin = None
        self._nlanes = 1

    def _upload(self, filter_weights=True):
        """(Re)build the device mirror of samples/weights (chains.py:276-323 funnel).  The min-weight filter of
        setSamples applies to a single sample array only: chain lists are filtered per chain before they are stacked."""
        self._drop_second_lane()
        self._chain_stats_cache = {}
        self._loglikes_col = None  # (an extra-column slot of the old sample set)
        w = self.weights
        if (filter_weights and self.chain_offsets is None and w is not None and self.min_weight_ratio is not None
                and self.min_weight_ratio >= 0):
            mx, mn = np.max(w), np.min(w)  # chains.py:1017-1027
            if mn < mx * self.min_weight_ratio:
                keep = w > mx * self.min_weight_ratio
                self.samples = self.samples[keep]
                self.weights = w = w[keep]
                if self.loglikes is not None:
                    self.loglikes = self.loglikes[keep]
                self.numrows = self.samples.shape[0]
        share = getattr(self, "_column_share", None)
        if share is not None:
            share.upload(self.ctx, self.samples, w)  # (collective: every rank of the job uploads at the same point)
        else:
            self.ctx.upload(self.samples, w)
        self._idx_cols = {}
        self.mean_loglike = None  # chains.py:317; recomputed on the device when a mean-likelihood is asked for
        self._like_mode = None

    def setSamples(self, samples, weights=None, loglikes=None, min_weight_ratio=None):
        """chains.py:276-300: replace samples / weights; drops the device mirror and every derived cache."""
        samples = np.asarray(samples)
        if samples.ndim == 1:
            samples = samples.reshape(-1, 1)
        if samples.shape[1] != self.n:
            raise WeightedSampleError("setSamples: number of parameters changed")
        self.samples = samples
        self.numrows = samples.shape[0]
        self.weights = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        self.loglikes = None if loglikes is None else np.asarray(loglikes, dtype=np.float64)
        if min_weight_ratio is not None:
            self.min_weight_ratio = min_weight_ratio
        self.chain_offsets = None
        self._weightsChanged()

    def changeSamples(self, samples):
        """chains.py:302-308"""
        self.setSamples(samples, self.weights, self.loglikes)

    def _weightsChanged(self, filter_weights=True):
        """chains.py:310-323: everything derived from samples/weights is stale; re-upload and recompute.  The
        min-weight filter belongs to setSamples (chains.py:296-299), not to the reference's _weightsChanged: mutators
        that call that directly (reweightAddingLogLikes, cool) pass filter_weights=False."""
        self.means = self.vars = self.sddev = self.fullcov = self.correlationMatrix = None
        self._upload(filter_weights=filter_weights)
        self.needs_update = True
        self.updateBaseStatistics()

    # ---- mutators of the sample set (state invalidation): every one funnels into a re-upload -------
    def _replace_samples(self, samples, weights, loglikes, chain_offsets=None):
        """setSamples(..., min_weight_ratio=-1) of the reference's mutators (no weight filter), for a sample array whose
        row AND column counts may have changed; the device mirror and every derived cache are rebuilt."""
        samples = np.asarray(samples)
        if samples.ndim == 1:
            samples = samples.reshape(-1, 1)
        if samples.shape[1] != len(self.paramNames.names):
            raise WeightedSampleError("number of sample columns does not match the parameter names")
        self.samples = samples
        self.numrows, self.n = samples.shape
        self.weights = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        self.loglikes = None if loglikes is None else np.ascontiguousarray(loglikes, dtype=np.float64)
        self.chain_offsets = None if chain_offsets is None else np.asarray(chain_offsets, dtype=np.int64)
        self.index = {p.name: i for i, p in enumerate(self.paramNames.names)}
        self._weightsChanged(filter_weights=False)

    def _host_weights(self):
        return self.weights if self.weights is not None else np.ones(self.numrows)

    def thin(self, factor):
        """chains.py:941-952: thin by ``factor`` to unit-weight samples (integer weights).  The thinned row list comes
        from the device (gd_thin_rows); chain boundaries follow the rows that survive."""
        thin_ix = self.thin_indices(factor)
        offsets = None if self.chain_offsets is None else np.searchsorted(thin_ix, self.chain_offsets)
        self._replace_samples(self.samples[thin_ix, :], None,
                              None if self.loglikes is None else self.loglikes[thin_ix], offsets)

    def weighted_thin(self, factor):
        """chains.py:954-966,1188-1206: thin by ``factor`` keeping integer multiplicities; separate chains are thinned
        one by one (the cumulative weight restarts with every chain), as the reference does."""
        if not self.ctx.weights_integral():
            raise WeightedSampleError("Can only thin with integer weights")
        ranges = [(0, self.numrows)] if self.chain_offsets is None else self._chain_ranges()
        rows, counts, lens = [], [], [0]
        for lo, hi in ranges:
            buf, K = self._thin_rows(factor, lo, hi)
            ix = buf.to_host((K,), dtype=np.int32).astype(np.int64) if K else np.zeros(0, dtype=np.int64)
            buf.free()
            u, c = np.unique(ix, return_counts=True)
            rows.append(u), counts.append(c), lens.append(len(u))
        rows, counts = np.concatenate(rows), np.concatenate(counts)
        offsets = None if self.chain_offsets is None else np.cumsum(lens)
        self._replace_samples(self.samples[rows, :], counts.astype(np.float64),
                              None if self.loglikes is None else self.loglikes[rows], offsets)

    def filter(self, where):
        """chains.py:968-979,1174-1186: keep the rows ``where`` (boolean mask or row indices)"""
        where = np.asarray(where)
        offsets = None
        if self.chain_offsets is not None:
            if where.dtype == bool:
                offsets = np.cumsum([0] + [int(np.count_nonzero(where[a:b])) for a, b in self._chain_ranges()])
            elif where.size == 0 or np.all(np.diff(where) > 0):
                offsets = np.searchsorted(where, self.chain_offsets)
        self._replace_samples(self.samples[where, :], None if self.weights is None else self.weights[where],
                              None if self.loglikes is None else self.loglikes[where], offsets)

    def deleteZeros(self):
        """chains.py:1010-1015"""
        self.filter(self._host_weights() > 0)

    def setMinWeightRatio(self, min_weight_ratio=1e-30):
        """chains.py:1017-1027"""
        if self.weights is not None and min_weight_ratio >= 0:
            mx, mn = np.max(self.weights), np.min(self.weights)
            if mn < mx * min_weight_ratio:
                self.filter(self.weights > mx * min_weight_ratio)

    def reweightAddingLogLikes(self, logLikes):
        """chains.py:981-993: importance-sample by adding ``logLikes`` (-log likelihood per sample)"""
        logLikes = np.asarray(logLikes, dtype=np.float64)
        if logLikes.shape != (self.numrows,):
            raise WeightedSampleError("logLikes must have one entry per sample")
        scale = np.min(logLikes)
        if self.loglikes is not None:
            self.loglikes = self.loglikes + logLikes
        self.weights = self._host_weights() * np.exp(-(logLikes - scale))
        self._weightsChanged(filter_weights=False)

    def cool(self, cool=None):
        """mcsamples.py:533-550 + chains.py:995-1008: multiply the log-likelihoods by ``cool`` and re-weight"""
        if cool is None:
            if self.temperature is None:
                raise ValueError("Pass a cooling temperature, since the sample does not have one specified")
            cool = float(self.temperature)
        if cool == 1:
            return
        if self.cooled != 1:
            logging.warning("Chain has already been cooled by %s", self.cooled)
        if self.loglikes is None:
            raise WeightedSampleError("Samples have no likelihood values, required to cool")
        MaxL = np.min(self.loglikes)
        newL = self.loglikes * cool
        self.weights = self._host_weights() * np.exp(-(newL - self.loglikes) - (MaxL * (1 - cool)))
        self.loglikes = newL
        self._weightsChanged(filter_weights=False)
        self.cooled = cool
        if self.temperature is not None:
            self.temperature = float(self.temperature) / cool

    def removeBurn(self, remove=0.3):
        """chains.py:1047-1061: drop the first ``remove`` fraction of the rows (or that many rows if >= 1)"""
        ix = int(remove) if remove >= 1 else int(round(self.numrows * remove))
        offsets = None
        if self.chain_offsets is not None:
            # the rows go from the front of the stacked array (the reference, chains.py:1047-1061, knows no chains here);
            # chains that lose all their rows are dropped from the chain list instead of staying behind with zero length
            offsets = np.unique(np.maximum(self.chain_offsets - ix, 0))
            if len(offsets) < 2:
                offsets = None
        self._replace_samples(self.samples[ix:, :], None if self.weights is None else self.weights[ix:],
                              None if self.loglikes is None else self.loglikes[ix:], offsets)

    def deleteFixedParams(self):
        """chains.py:1029-1045,1544-1559: remove the parameters that do not vary (they become zero-width ranges).
        Returns (indices removed, their values)."""
        fixed, values = [], []
        for i in range(self.samples.shape[1]):
            if np.isclose(self.samples[0, i], self.samples[-1, i], equal_nan=True):
                mean = np.average(self.samples[:, i])
                if np.allclose(self.samples[:, i], mean, rtol=1e-12, atol=0, equal_nan=True):
                    fixed.append(i)
                    values.append(mean)
        if fixed:
            for ix, value in zip(fixed, values):
                self.ranges.setFixed(self.paramNames.names[ix].name, value)
            self.paramNames.deleteIndices(fixed)
            self._replace_samples(np.delete(self.samples, fixed, 1), self.weights, self.loglikes, self.chain_offsets)
        return fixed, values

    def addDerived(self, paramVec, name, label="", comment="", range=None):
        """mcsamples.py:2560-2575 + chains.py:1354-1366: append a derived parameter column.  Returns its ParamInfo."""
        if self.paramNames.parWithName(name):
            raise ValueError("Parameter with name %s already exists" % name)
        vec = np.asarray(paramVec, dtype=np.float64).reshape(-1)
        if vec.shape != (self.numrows,):
            raise WeightedSampleError("derived parameter vector must have one entry per sample")
        if range is not None:
            self.ranges.setRange(name, range)
        new = np.empty((self.numrows, self.n + 1), dtype=np.float64, order="F")  # the device layout: no transpose
        new[:, :self.n] = self.samples
        new[:, self.n] = vec
        par = ParamInfo(name, label or None)
        par.isDerived, par.comment = True, comment
        self.paramNames.names.append(par)
        self._replace_samples(new, self.weights, self.loglikes, self.chain_offsets)
        return par

    # ---- what GetDist's plotting layer asks a sample set for besides densities (plots.py:655-690,933-955,2262-2290) ------
    def getParamNames(self):
        """chains.py:1221-1225"""
        return self.paramNames

    def getRenames(self):
        return self.paramNames.getRenames()

    def getName(self):
        return self.name_tag

    def getLabel(self):
        """chains.py:260-266: the samples' label for legends (the name tag with LaTeX specials escaped when there is none)"""
        if self.label:
            return self.label
        name = self.getName()
        return None if name is None else "".join("\\" + ch if ch in "_&%$#{}" else ch for ch in name)

    def getUpper(self, name):
        """mcsamples.py:2311-2321: the hard upper bound in force for the parameter (None: unbounded / unknown name)"""
        par = self.paramNames.parWithName(name)
        return getattr(par, "limmax", None) if par else None

    def getLower(self, name):
        par = self.paramNames.parWithName(name)
        return getattr(par, "limmin", None) if par else None

    def getParams(self):
        """chains.py:1252-1268 in spirit: an object with one attribute per parameter name holding its sample vector"""

        class ParSamples:
            pass

        out = ParSamples()
        for j, par in enumerate(self.paramNames.names):
            setattr(out, par.name, self.samples[:, j])
        return out

    # ---- likelihood statistics (mcsamples.py:2216-2261, 2369-2378) ----------------------------------------------
    def _setLikeStats(self):
        """Best-fit sample, posterior likelihood statistics and the N-D confidence-region limits.  Two passes over the
        loglikes column on the device (gd_like_stats) give every weighted mean the reference forms from exp / square
        of that vector; the N-D limits come from _setNDLimits (weighted quantile + conditional min / max)."""
        if self.loglikes is None:
            self._likeStats = None
            return None
        ctx = self.ctx
        col = ctx.set_extra_column(ctx.EXTRA_COLS - 1, self.loglikes)
        self._loglikes_col = col  # (_setNDLimits below reads the same resident copy instead of uploading it again)
        try:  # whatever happens below, the column id must not outlive this call: the slot is rewritten by later uploads
            st = ctx.like_stats(col)
            norm = self.norm
            maxlike = st["min"]
            m = LikeStats()
            m.logLike_sample = maxlike
            m.logMeanInvLike = (np.log(st["sum_w_exp_plus"] / norm) + maxlike) if st["max"] - maxlike < 30 else None
            self.mean_loglike = st["sum_wl"] / norm  # chains.py:380-383
            m.meanLogLike = self.mean_loglike
            m.logMeanLike = -np.log(st["sum_w_exp_minus"] / norm) + maxlike
            m.complexity = 2 * (self.mean_loglike - maxlike)
            m.varLogLike = st["sum_wl2"] / norm - self.mean_loglike**2
            m.names = self.paramNames.names
            self._setNDLimits()
        finally:
            self._loglikes_col = None
        best = self.samples[st["argmin"]]
        for j, par in enumerate(self.paramNames.names):
            par.bestfit_sample = best[j]
        self._likeStats = m
        return m

    @property
    def likeStats(self):
        """The reference sets this attribute inside updateBaseStatistics (mcsamples.py:552-576); here it is computed when first
        read after a change of the samples (three passes over the sample set) -- reading the attribute and calling
        getLikeStats() are the same thing, as are the parameters' ``bestfit_sample`` values it leaves."""
        return self._likeStats if self._likeStats is not None else self._setLikeStats()

    @likeStats.setter
    def likeStats(self, value):
        self._likeStats = value

    def getLikeStats(self):
        """mcsamples.py:2369-2378 (computed on first use after a change of the samples, not inside every
        updateBaseStatistics: it costs three passes over the sample set)"""
        return self.likeStats

    def _use_like_weights(self, mode):
        """
        Make the mean-likelihood weights resident on the device: mode 0 = weights*exp(mean_loglike - loglikes)
        (mcsamples.py:1560,1830), mode 1 = weights*loglikes (:1558).  mean_loglike (chains.py:380-383) falls out of
        the mode-1 pass as sum/norm.
        """
        if self.loglikes is None:
            raise MCSamplesError("mean likelihoods need the loglikes column")
        if self.mean_loglike is None:
            self.mean_loglike = self.ctx.like_weights(self.loglikes, 1, 0.0) / self.norm
            self._like_mode = 1
        if self._like_mode != mode:
            self.ctx.like_weights(self.loglikes, mode, self.mean_loglike)
            self._like_mode = mode

    def _like_histograms(self, mode, fn):
        """Run the histogram call ``fn`` with the like weights selected (np.bincount(..., weights=w) of :1561,1831)."""
        self._use_like_weights(mode)
        self.ctx.select_weights(1)
        try:
            return fn()
        finally:
            self.ctx.select_weights(0)

    def mean_diff(self, paramVec, where=None):
        """chains.py:744-761 (host vector p_i - mean; the device path never materialises it)"""
        vec = self._host_vector(paramVec)
        if vec is None:
            vec = self.samples[:, self._col(paramVec)]
        if where is None:
            return vec - self.mean(paramVec)
        return vec[where] - self.mean(paramVec, where)

    def mean_diffs(self, pars=None, where=None):
        """chains.py:763-780"""
        cols = range(self.n) if pars is None else (range(pars) if isinstance(pars, (int, np.integer)) else pars)
        return [self.mean_diff(j, where) for j in cols]

    def initParamConfidenceData(self, paramVec, start=0, end=None, weights=None):
        """
        chains.py:793-812.  The reference caches argsort + cumulative weights here; the device path selects
        quantiles without sorting, so the "cache" is just the (column, row range) handle confidence() accepts.
        """
        vec = self._host_vector(paramVec)
        return ParamConfidenceData(None if vec is not None else self._col(paramVec), start,
                                   self.numrows if end is None else end, weights=weights, vec=vec)

    def getFractionIndices(self, weights, n):
        """mcsamples.py:668-680: row indices splitting the total weight into n equal parts"""
        if weights is None:
            weights = np.ones(self.numrows)
        cumsum = np.cumsum(weights)
        return np.append(np.searchsorted(cumsum, np.linspace(0, 1, n, endpoint=False) * self.norm), self.numrows)

    def getCorrLengths(self, min_corr=0.05):
        """
        The numbers of the CorrLengths block of getConvergeTests (mcsamples.py:941-962): per parameter, the weight-unit
        autocorrelation length from the chain-averaged autocovariance (each chain about its own mean), summed up to the
        first lag at or below 5 %.  Lag sums per chain run on the GPU in 32-lag chunks with early exit.
        """
        if self.chain_offsets is None:
            raise WeightedSampleError("Samples were not combined from separate chains")
        if self.needs_update:
            self.updateBaseStatistics()
        ranges = list(zip(self.chain_offsets[:-1], self.chain_offsets[1:]))
        stats = self.getSeparateChainStats(self.n)
        maxoff = int(min((b - a) // 10 for a, b in ranges))
        cols = list(range(self.n))
        corr_rows = [[] for _ in cols]
        result = [None] * self.n
        k0 = 0
        while k0 <= maxoff and any(r is None for r in result):
            nl = min(32, maxoff + 1 - k0)
            chunk = np.zeros((self.n, nl))
            for (a, b), (cmeans, _, _) in zip(ranges, stats):
                nc = int(b - a)
                lags = self.ctx.autocov_lags_range_batch(cols, cmeans, int(a), int(b), k0, nl)
                chunk += lags / (nc - np.arange(k0, k0 + nl)) * nc  # normalize=True, weight_units, times chain.norm
            chunk /= (self.norm * self.vars)[:, None]
            for j in cols:
                if result[j] is not None:
                    continue
                corr_rows[j].extend(chunk[j].tolist())
                c = np.array(corr_rows[j])
                below = np.nonzero(~(c > min_corr * c[0]))[0]
                if below.size:
                    result[j] = c[0] + 2 * float(np.sum(c[1:int(below[0])]))
            k0 += nl
        for j in cols:
            if result[j] is None:
                result[j] = corr_rows[j][0]  # argmin of an all-True mask is 0
        self.indep_thin = max(result)
        return np.array(result)

    def getSplitTests(self, test_confidence=0.95, max_split_tests=4):
        """
        The numbers of the SplitTest block of getConvergeTests (mcsamples.py:1005-1034): for n = 2..max_split_tests
        splits of the rows, rms over the splits of the change in the upper / lower quantile, in units of the standard
        deviation.  Returns an array (nparam, max_split_tests-1, 2) ordered [upper, lower] like the reference's table.
        Every (row range) needs one batched quantile-select launch over all parameters.
        """
        if self.needs_update:
            self.updateBaseStatistics()
        limits = np.array([1 - (1 - test_confidence) / 2, (1 - test_confidence) / 2])
        cols = list(range(self.n))

        def conf(lo, hi):
            norm = self.norm if (lo == 0 and hi == self.numrows) else self.ctx.weight_stats(int(lo), int(hi))["norm"]
            return self.ctx.quantiles(cols, np.tile(norm * limits, (self.n, 1)), lo=int(lo), hi=int(hi))

        confids = conf(0, self.numrows)
        out = np.zeros((self.n, max_split_tests - 1, 2))
        for ix in range(max_split_tests - 1):
            split_n = 2 + ix
            frac = self.getFractionIndices(self.weights, split_n)
            for f1, f2 in zip(frac[:-1], frac[1:]):
                out[:, ix, :] += (conf(f1, f2) - confids) ** 2
            out[:, ix, :] = np.sqrt(out[:, ix, :] / split_n) / self.sddev[:, None]
        return out

    def getConvergeTests(self, test_confidence=0.95, writeDataToFile=False,
                         what=("MeanVar", "GelmanRubin", "SplitTest", "RafteryLewis", "CorrLengths"), filename=None,
                         feedback=False):
        """
        mcsamples.py:904-1221: the report text of the convergence tests (same defaults as the reference; "CorrSteps" is
        opt-in there too).  Sets self.GelmanRubin, self.indep_thin, self.RL_indep_thin like the reference.  Every
        N-sized pass behind the numbers (chain covariances, quantiles on row sub-ranges, lag sums, thinning, transition
        counts) runs on the GPU; this function only formats.
        """
        if writeDataToFile or filename:
            raise NotImplementedError("file output is outside the accelerated path")
        for w in what:
            if w not in ("MeanVar", "GelmanRubin", "SplitTest", "CorrLengths", "RafteryLewis", "CorrSteps"):
                raise NotImplementedError("convergence test %s is outside the accelerated path" % w)
        lines = ""
        nchains = 0 if self.chain_offsets is None else len(self.chain_offsets) - 1
        if "CorrLengths" in what:
            lines += ("Parameter autocorrelation lengths (effective number of samples N_eff = tot weight/weight length)\n\n"
                      + "%-20s%15s %15s %15s\n" % ("", "Weight Length", "Sample length", "N_eff"))
            for nm, Nw in zip(self.paramNames.list(), self.getCorrLengths()):
                form = "%15.2f" if self.mean_mult > 1 else "%15.2E"
                lines += "%-20s" % nm + form % Nw + " %15.2f %15i\n" % (Nw / self.mean_mult, self.norm / Nw)
            lines += "\n"
        if nchains > 1 and "MeanVar" in what:
            lines += "\nmean convergence stats using remaining chains\nparam sqrt(var(chain mean)/mean(chain var))\n\n"
            for nm, v in zip(self.paramNames.list(), self.getMeanVarTest()):
                lines += "%-20s%10.4f\n" % (nm, v)
            lines += "\n"
        if nchains > 1 and "GelmanRubin" in what:
            D = self.getGelmanRubinEigenvalues()
            if D is not None:
                self.GelmanRubin = np.max(D)
                lines += "var(mean)/mean(var) for eigenvalues of covariance of y of orthonormalized parameters\n"
                for jj, Di in enumerate(D):
                    lines += "%3i%13.5f\n" % (jj + 1, Di)
                summary = " var(mean)/mean(var), remaining chains, worst e-value: R-1 = %13.5F" % self.GelmanRubin
            else:
                self.GelmanRubin = None
                summary = "Gelman-Rubin covariance not invertible (parameter not moved?)"
                logging.warning(summary)
            if feedback:
                print(summary)
            lines += "\n"
        if "SplitTest" in what:
            lines += "Split tests: rms_n([delta(upper/lower quantile)]/sd) n={2,3,4}, limit=%.0f%%:\n" % (
                100 * self.converge_test_limit)
            lines += "i.e. mean sample splitting change in the quantiles in units of the st. dev.\n\n"
            st = self.getSplitTests(test_confidence)
            for j, nm in enumerate(self.paramNames.list()):
                for endb, typestr in enumerate(["upper", "lower"]):
                    lines += "%-20s" % nm + "".join("%9.4f" % st[j, ix, endb] for ix in range(st.shape[1])) + " %s\n" % typestr
            lines += "\n"
        # mcsamples.py:1039: the remaining two tests need integer weights (raw MCMC multiplicities)
        if ("RafteryLewis" in what or "CorrSteps" in what) and self.ctx.weights_integral():
            if "RafteryLewis" in what:
                rl = self.getRafteryLewis(test_confidence)
                if rl is None:
                    print("Raftery and Lewis estimator had problems")
                    return None
                lines += "Raftery&Lewis statistics\n\nchain  markov_thin  indep_thin    nburn\n"
                for ix in range(len(rl["thin_fac"])):
                    if rl["thin_fac"][ix] == 0:
                        lines += "%4i      Failed/not enough samples\n" % ix
                    else:
                        lines += "%4i%12i%12i%12i\n" % (ix, rl["markov_thin"][ix], rl["thin_fac"][ix], rl["nburn"][ix])
                if feedback:
                    if not np.all(rl["thin_fac"] != 0):
                        print("RL: Not enough samples to estimate convergence stats")
                    else:
                        print("RL: Thin for Markov: ", np.max(rl["markov_thin"]))
                        print("RL: Thin for indep samples:  ", str(self.RL_indep_thin))
                        print("RL: Estimated burn in steps: ", np.max(rl["nburn"]), " (",
                              int(round(np.max(rl["nburn"]) / self.mean_mult)), " rows)")
                lines += "\n"
            if "CorrSteps" in what:
                lines += "Parameter auto-correlations as function of step separation\n\n"
                thin, corrs = self.getCorrSteps()
                if corrs is not None:
                    lines += "%-20s" % "" + "".join("%8i" % ((i + 1) * thin) for i in range(corrs.shape[0])) + "\n"
                    for j, nm in enumerate(self.paramNames.list()):
                        lines += "%-20s" % nm + "".join("%8.3f" % corrs[i][j] for i in range(corrs.shape[0])) + " \n"
        return lines

    # ---- thinned-chain diagnostics (mcsamples.py:1039-1221; chains.py:853-916) ---------------------------------
    def _chain_ranges(self):
        if self.chain_offsets is None:
            raise WeightedSampleError("Samples were not combined from separate chains")
        return [(int(a), int(b)) for a, b in zip(self.chain_offsets[:-1], self.chain_offsets[1:])]

    def _thin_rows(self, factor, lo=0, hi=None):
        """
        Device row list of the weight-one thinning of rows [lo,hi) (chains.py:878-916): (buffer, count).  Which of the
        reference's two branches applies is decided by factor >= max weight of that chain, as there.
        """
        hi = self.numrows if hi is None else hi
        return self._thin_rows_on(self.ctx, factor, lo, hi)

    @staticmethod
    def _thin_rows_on(ctx, factor, lo, hi):
        if factor != int(factor):
            raise WeightedSampleError("Thin factor must be integer")
        ws = ctx.weight_stats(lo, hi)
        unique_mode = int(factor) >= ws["max_w"]
        capacity = int(ws["norm"]) // int(factor) + 2
        return ctx.thin_rows(lo, hi, int(factor), unique_mode, capacity)

    def thin_indices(self, factor, weights=None):
        """chains.py:853-863: indices that make single-weight samples (the device list copied to the host).  ``weights``:
        thin THAT weight vector instead of the resident one (any length, as the reference's static
        thin_indices_single_samples does): it is uploaded to a short-lived context of its own -- the cached prefix sum and the
        thinning kernels belong to a context's sample weights -- and thinned by the same kernels."""
        if weights is not None:
            w = np.ascontiguousarray(weights, dtype=np.float64).ravel()
            if w.size == 0:
                return np.zeros(0, dtype=np.int64)
            tmp = self._context_factory(self._device)
            try:
                tmp.upload(np.zeros((w.size, 1)), w)
                if not tmp.weights_integral():
                    raise WeightedSampleError("Can only thin with integer weights")
                buf, K = self._thin_rows_on(tmp, factor, 0, w.size)
                out = buf.to_host((K,), dtype=np.int32).astype(np.int64) if K else np.zeros(0, dtype=np.int64)
                buf.free()
            finally:
                tmp.close()
            return out
        if not self.ctx.weights_integral():
            raise WeightedSampleError("Can only thin with integer weights")
        buf, K = self._thin_rows(factor)
        out = buf.to_host((K,), dtype=np.int32).astype(np.int64) if K else np.zeros(0, dtype=np.int64)
        buf.free()
        return out

    def getRafteryLewis(self, test_confidence=0.95, nparam=None):
        """
        The Raftery-Lewis block of getConvergeTests (mcsamples.py:1039-1165): per chain the thinning needed for the
        thinned binary chains (parameter above/below a tail quantile) to be first-order Markov, then independent, and
        the burn-in estimate.  Returns dict(markov_thin, thin_fac, nburn) (arrays over chains; thin_fac = indep_thin,
        0 = failed) or None where the reference gives up.  The quantiles, the thinning and the transition counts of
        every (parameter, tail) at the current thin factor come from the GPU in one launch each; only the BIC logic
        on 8 / 4 integers runs here.
        """
        import math

        if not self.ctx.weights_integral():
            raise WeightedSampleError("Raftery-Lewis needs integer weights")
        ctx = self.ctx
        ranges = self._chain_ranges()
        nparamMC = nparam or self.paramNames.numNonDerived()
        cols = list(range(nparamMC))
        limits = np.array([1 - (1 - test_confidence) / 2, (1 - test_confidence) / 2])
        nc = len(ranges)
        thin_fac = np.zeros(nc, dtype=int)
        nburn = np.zeros(nc, dtype=int)
        markov_thin = np.zeros(nc, dtype=int)
        epsilon = 0.001
        hardest, hardestend = -1, 0  # carried over from chain to chain, as in the reference

        class Failed(Exception):
            pass

        for ix, (lo, hi) in enumerate(ranges):
            ws = ctx.weight_stats(lo, hi)
            thin_fac[ix] = int(round(ws["max_w"]))
            targets = np.tile(ws["norm"] * limits, (nparamMC, 1))
            confids = ctx.quantiles(cols, targets, lo=lo, hi=hi)  # (param, upper/lower)
            cache = {}

            def counts(f, columns=cols, thr=confids, key="all"):
                """(thin_rows, transition counts of every column/threshold) at thin factor f, cached per factor."""
                if (key, f) not in cache:
                    rows, K = self._thin_rows(f, lo, hi)
                    cache[(key, f)] = (K, ctx.binary_transitions(columns, rows, K, thr) if K >= 2 else None)
                    rows.free()
                return cache[(key, f)]

            thin_rows = None
            try:
                for j in range(nparamMC):
                    for endb in (0, 1):
                        tran = None
                        while True:
                            thin_rows, c = counts(int(thin_fac[ix]))
                            if thin_rows < 2:
                                break
                            tran = c[j, endb, :8].reshape(2, 2, 2)
                            g2 = _g2_markov_vs_second_order(tran)
                            if g2 - math.log(float(thin_rows - 2)) * 2 < 0:
                                break
                            thin_fac[ix] += 1
                        if tran is None:
                            raise ValueError("not enough thinned samples")  # the reference's NameError -> bare except
                        if np.sum(tran[:, 0, 1]) == 0 or np.sum(tran[:, 1, 0]) == 0:
                            thin_fac[ix] = 0
                            raise Failed()
                        alpha = np.sum(tran[:, 0, 1]) / float(np.sum(tran[:, 0, 0]) + np.sum(tran[:, 0, 1]))
                        beta = np.sum(tran[:, 1, 0]) / float(np.sum(tran[:, 1, 0]) + np.sum(tran[:, 1, 1]))
                        probsum = alpha + beta
                        tmp1 = math.log(probsum * epsilon / max(alpha, beta)) / math.log(abs(1.0 - probsum))
                        if int(tmp1 + 1) * thin_fac[ix] > nburn[ix]:
                            nburn[ix] = int(tmp1 + 1) * thin_fac[ix]
                            hardest, hardestend = j, endb
                markov_thin[ix] = thin_fac[ix]
                hardest = max(hardest, 0)
                u = self.confidence(hardest, (1 - test_confidence) / 2, hardestend == 0)  # over ALL samples (:1113)
                while True:
                    thin_rows, c = counts(int(thin_fac[ix]), [hardest], [[u]], key=("indep", hardest, hardestend))
                    if thin_rows < 2:
                        break
                    tran2 = c[0, 0, 8:].reshape(2, 2)
                    g2 = _g2_independence_vs_markov(tran2, thin_rows)
                    if g2 is None:
                        return None
                    if g2 - np.log(float(thin_rows - 1)) < 0:
                        break
                    thin_fac[ix] += 1
            except Failed:
                pass
            except (ValueError, ZeroDivisionError, OverflowError, FloatingPointError):
                thin_fac[ix] = 0  # the arithmetic failures the reference's bare `except:` swallows (:1146-1147)
            if thin_fac[ix] and thin_rows is not None and thin_rows < 2:
                thin_fac[ix] = 0
        self.RL_indep_thin = np.max(thin_fac)
        return dict(markov_thin=markov_thin, thin_fac=thin_fac, nburn=nburn)

    def getCorrSteps(self):
        """
        The CorrSteps block (mcsamples.py:1183-1210): auto-correlation of every parameter in the thinned chains
        (each about its own mean) at step separations 1..maxoff thinned rows.  Returns (autocorr_thin, corrs[maxoff, n])
        or (autocorr_thin, None).  Thinning and the gathered lag sums run on the GPU.
        """
        if self.needs_update:
            self.updateBaseStatistics()
        ranges = self._chain_ranges()
        if self.corr_length_thin != 0:
            autocorr_thin = self.corr_length_thin
        else:
            indep_thin = getattr(self, "indep_thin", 0)
            if indep_thin == 0:
                autocorr_thin = 20
            elif indep_thin <= 30:
                autocorr_thin = 5
            else:
                autocorr_thin = int(5 * (indep_thin / 30))
        rows, K = self._thin_rows(autocorr_thin)
        rows.free()
        maxoff = int(min(self.corr_length_steps, K // (2 * len(ranges))))
        if maxoff <= 0:
            return autocorr_thin, None
        cols = list(range(self.n))
        corrs = np.zeros((maxoff, self.n))
        for (lo, hi), (cmeans, _, _) in zip(ranges, self.getSeparateChainStats(self.n)):
            rows, K = self._thin_rows(autocorr_thin, lo, hi)
            maxoff = min(maxoff, K // autocorr_thin)
            if maxoff > 0:
                lags = self.ctx.thinned_lag_sums(cols, cmeans, rows, K, maxoff)  # (n, maxoff)
                corrs[:maxoff] += (lags / (K - np.arange(1, maxoff + 1))).T / self.vars
            rows.free()
        corrs /= len(ranges)
        return autocorr_thin, corrs[:maxoff]

    def updateSettings(self, settings=None, ini=None, doUpdate=True):
        """mcsamples.py:472-499: analysis settings from a dict and / or a .ini file (the dict takes preference)"""
        if ini is not None:
            settings = dict(_read_ini_settings(ini), **(settings or {}))
        for k, v in (settings or {}).items():
            if k == "force_twotail":
                self.force_twotail = v in (True, "T", "t", "True", "true", 1)
                if self.force_twotail:
                    logging.warning("Computing two tail limits")
                continue
            if k.startswith("max_frac_twotail") and k[len("max_frac_twotail"):].isdigit():
                self._max_frac_overrides = dict(getattr(self, "_max_frac_overrides", {}))
                self._max_frac_overrides[int(k[len("max_frac_twotail"):]) - 1] = float(v)
                continue
            if k == "no_warning_params":  # (ini: a space-separated list, mcsamples.py:438)
                self.no_warning_params = v.split() if isinstance(v, str) else list(v)
                continue
            if k == "no_warning_chi2_params":
                self.no_warning_chi2_params = v in (True, "T", "t", "True", "true", 1)
                continue
            if k not in DEFAULT_SETTINGS:
                raise SettingError("unknown setting: %s" % k)
            cur = DEFAULT_SETTINGS[k]
            if isinstance(cur, bool):
                v = v in (True, "T", "t", "True", "true", 1)
            elif isinstance(cur, int) and not isinstance(v, bool):
                v = int(v)
            elif isinstance(cur, float):
                v = float(v)
            setattr(self, k, v)
        if doUpdate:
            self.needs_update = True

    def setRanges(self, ranges):
        """mcsamples.py:324-346"""
        if isinstance(ranges, dict):
            for nm, rng in ranges.items():
                self.ranges.setRange(nm, rng)
        else:
            for nm, rng in zip(self.paramNames.list(), ranges):
                self.ranges.setRange(nm, rng)
        self.needs_update = True

    def _partial_moments(self, lo, hi):
        """One packed vector of the base statistics of rows [lo, hi): [norm, max_w, sum_w2, min(n), max(n), mean(n),
        cov(n x n)] -- what a rank contributes when the rows are split over ranks (three launches over its share)."""
        ws = self.ctx.weight_stats(lo, hi)
        means, cov, norm, mm = self.ctx.cov(list(range(self.n)), lo=lo, hi=hi, minmax=True)
        nrm = norm if self.weights is not None else float(hi - lo)
        return np.concatenate([[nrm, ws["max_w"], ws["sum_w2"]], mm[:, 0], mm[:, 1], means, cov.reshape(-1)])

    def _combine_moments(self, parts):
        """Pool per-share moments: means by weight, covariance as the weighted mean of the shares' covariances plus the
        spread of their means (the identity behind chains.py:1456-1466), minima / maxima / sums directly."""
        n = self.n
        parts = np.asarray(parts, dtype=np.float64)
        norms = parts[:, 0]
        norm = float(np.sum(norms))
        means = norms @ parts[:, 3 + 2 * n:3 + 3 * n] / norm
        cov = np.zeros((n, n))
        for p in parts:
            d = p[3 + 2 * n:3 + 3 * n] - means
            cov += p[0] * (p[3 + 3 * n:].reshape(n, n) + np.outer(d, d))
        cov /= norm
        return dict(norm=norm, max_w=float(np.max(parts[:, 1])), sum_w2=float(np.sum(parts[:, 2])),
                    col_min=np.min(parts[:, 3:3 + n], axis=0), col_max=np.max(parts[:, 3 + n:3 + 2 * n], axis=0),
                    means=means, cov=cov)

    def updateBaseStatistics(self, row_share=None, exchange=None):
        """
        chains.py:1340-1352 + mcsamples.py:552-576, with the column scans on the GPU.

        Multi-GPU (samples replicated): ``row_share=(rank, world)`` makes this process reduce only its
        contiguous share of the rows; ``exchange(vector) -> (world, len)`` (an all-gather of n^2 + 3n + 3 doubles over
        RCCL) pools the shares, so the O(N n^2) covariance pass costs 1/world per rank instead of being repeated.
        """
        if row_share is not None:
            rank, world = row_share
            per = (self.numrows + world - 1) // world
            lo, hi = min(rank * per, self.numrows), min((rank + 1) * per, self.numrows)
            if hi > lo:
                mine = self._partial_moments(lo, hi)
            else:
                # a rank without rows (N < world, or the last rank after the ceiling division) still contributes a
                # vector of the full length, so the all-gather's shapes agree: zero norm, +inf / -inf extrema
                n = self.n
                mine = np.zeros(3 + 3 * n + n * n)
                mine[3:3 + n], mine[3 + n:3 + 2 * n] = np.inf, -np.inf
            parts = exchange(mine)
            pooled = self._combine_moments([p for p in parts if p is not None and p[0] > 0])
            self.norm = np.float64(pooled["norm"]) if self.weights is not None else np.float64(self.numrows)
            self._col_min, self._col_max = pooled["col_min"], pooled["col_max"]
            self.means = pooled["means"]
            self.fullcov = pooled["cov"]
            self.vars = np.diag(self.fullcov).copy()
            self.sddev = np.sqrt(self.vars)
            self.mean_mult = self.norm / self.numrows
            self.max_mult = pooled["max_w"]
            self._sum_w2 = pooled["sum_w2"]
            if self.weights is not None:  # mcsamples.py:559-562, from the pooled sums (each rank counts its own rows)
                mult_max = (self.mean_mult * self.numrows) / min(self.numrows // 2, 500)
                if self.max_mult > mult_max:
                    outliers = self.ctx.weight_stats(thresh=mult_max)["n_above"]
                    if outliers != 0:
                        logging.warning("outlier fraction %s ", float(outliers) / self.numrows)
            self.correlationMatrix = None
            self._after_base_statistics()
            return self
        ws = self.ctx.weight_stats()
        if self.weights is not None:
            self.norm = ws["norm"]
        else:
            self.norm = np.float64(self.numrows)  # chains.py:315
        # one statistics pass (min, max, weighted mean) + one covariance pass; the variances are its diagonal
        # (chains.py:409-410 and :729 are the same sum)
        means, cov, _, mm = self.ctx.cov(list(range(self.n)), minmax=True)
        self._col_min, self._col_max = mm[:, 0].copy(), mm[:, 1].copy()
        self.means = means
        self.vars = np.diag(cov).copy()
        self.sddev = np.sqrt(self.vars)
        self.mean_mult = self.norm / self.numrows
        self.max_mult = ws["max_w"]
        self._sum_w2 = ws["sum_w2"]
        mult_max = (self.mean_mult * self.numrows) / min(self.numrows // 2, 500)
        if self.weights is not None:
            outliers = self.ctx.weight_stats(thresh=mult_max)["n_above"]
            if outliers != 0:
                logging.warning("outlier fraction %s ", float(outliers) / self.numrows)
        self.fullcov = cov
        self.correlationMatrix = None
        self._after_base_statistics()
        return self

    def _after_base_statistics(self):
        self.density1D = {}
        self._initLimits()
        for par in self.paramNames.names:
            par.N_eff_kde = None
            par._ranges_done = False
        self._nd_limits_done = False
        self.likeStats = None
        self.needs_update = False

    def _initLimits(self):
        """mcsamples.py:442-470"""
        for par in self.paramNames.names:
            par.limmin = self.ranges.getLower(par.name)
            par.limmax = self.ranges.getUpper(par.name)
            par.has_limits_bot = par.limmin is not None
            par.has_limits_top = par.limmax is not None
            par.periodic = par.name in self.ranges.periodic

    def _parAndNumber(self, name):
        """chains.py:1235-1250"""
        if isinstance(name, ParamInfo):
            name = name.name
        if isinstance(name, str):
            name = self.index.get(name, None)
            if name is None:
                return None, None
        if isinstance(name, (int, np.integer)):
            return int(name), self.paramNames.names[int(name)]
        raise ParamError("Unknown parameter type %s" % name)

    # ---- moments (chains.py:339-412, 636-780) ------------------------------------------------------------
    # Vectors, row filters and alternative weights (chains.py:325-337, 636-780): a host vector goes into one of the
    # device's spare columns; `where=` / `weights=` become an auxiliary weight vector (weights*mask) that is swapped in
    # for the duration of the call, which gives exactly the reference's x[where], w[where] sums.
    def _host_vector(self, par):
        """The host vector behind a non-column argument of _makeParamvec (chains.py:325-337), else None."""
        if isinstance(par, np.ndarray):
            if par.shape != (self.numrows,):
                raise WeightedSampleError("parameter vector must have one entry per sample")
            return par
        if isinstance(par, (int, np.integer)) and not isinstance(par, bool):
            if par == -1:
                if self.loglikes is None:
                    raise WeightedSampleError("Samples do not have logLikes (par=-1)")
                return self.loglikes
            if par == -2:
                return self.weights if self.weights is not None else np.ones(self.numrows)
            if not 0 <= par < self.n:
                raise WeightedSampleError("Parameter %i does not exist" % par)
        return None

    def _vec_col(self, par, slot=0):
        """Device column index for a parameter reference or a host vector (uploaded into spare column ``slot``)."""
        vec = par.vec if isinstance(par, ParamConfidenceData) else self._host_vector(par)
        if vec is not None:
            if slot >= self.ctx.EXTRA_COLS:
                raise WeightedSampleError("at most %d vector arguments per call" % self.ctx.EXTRA_COLS)
            return self.ctx.set_extra_column(slot, vec)
        return par.col if isinstance(par, ParamConfidenceData) else self._col(par)

    def _where_weights(self, where):
        """weights*mask for a boolean mask or an index array (numpy semantics of x[where])."""
        where = np.asarray(where)
        w = self.weights if self.weights is not None else np.ones(self.numrows)
        if where.dtype == bool:
            if where.shape != (self.numrows,):
                raise WeightedSampleError("where must have one entry per sample")
            return w * where
        return w * np.bincount(_where_rows(where, self.numrows), minlength=self.numrows)

    def _with_weights(self, w_host, fn):
        """Run ``fn`` with the auxiliary weight vector ``w_host`` selected on the device."""
        self.ctx.aux_weights(w_host)
        self._like_mode = None  # the auxiliary buffer is shared with the like weights
        self.ctx.select_weights(1)
        try:
            return fn()
        finally:
            self.ctx.select_weights(0)

    def get_norm(self, where=None):
        if where is None:
            return self.norm
        return self._with_weights(self._where_weights(where), lambda: self.ctx.weight_stats()["norm"])

    def weighted_sum(self, paramVec, where=None):
        """chains.py:636-649"""
        return self.mean(paramVec, where) * self.get_norm(where)

    def getMeans(self, pars=None):
        return self.means if pars is None else np.array([self.means[i] for i in pars])

    def getVars(self):
        return self.vars

    def _setCov(self):
        _, cov, _ = self.ctx.cov()
        self.fullcov = cov
        return cov

    def getCov(self, nparam=None, pars=None):
        if self.fullcov is None:
            self._setCov()
        if pars is not None:
            return self.fullcov[np.ix_(pars, pars)]
        return self.fullcov[:nparam, :nparam]

    def _moments(self, pars, where):
        """(means, cov, norm) of parameter references / vectors, optionally over a row filter: one gd_cov call."""
        slot = 0
        cols = []
        for p in pars:
            if self._host_vector(p) is not None:
                cols.append(self._vec_col(p, slot))
                slot += 1
            else:
                cols.append(self._col(p))
        if where is None:
            return self.ctx.cov(cols)
        return self._with_weights(self._where_weights(where), lambda: self.ctx.cov(cols))

    def cov(self, pars=None, where=None):
        """chains.py:709-733"""
        if isinstance(pars, (int, np.integer)):
            pars = range(pars)
        return self._moments(list(range(self.n)) if pars is None else list(pars), where)[1]

    def corr(self, pars=None):
        return covToCorr(self.cov(pars))

    def getCorrelationMatrix(self):
        if self.correlationMatrix is None:
            self.correlationMatrix = covToCorr(self.getCov())
        return self.correlationMatrix

    def _col(self, par):
        if type(par) is int and 0 <= par < self.n:
            return par
        j = self._parAndNumber(par)[0]
        if j is None:
            raise ParamError("unknown parameter %s" % par)
        return j

    def _is_plain_column(self, par):
        return self._host_vector(par) is None

    def mean(self, paramVec, where=None):
        """chains.py:665-677"""
        if isinstance(paramVec, (list, tuple)):
            if where is None and all(self._is_plain_column(p) for p in paramVec):
                return np.array([self.means[self._col(p)] for p in paramVec])
            return np.array([self.mean(p, where) for p in paramVec])
        if where is None and self._is_plain_column(paramVec):
            return self.means[self._col(paramVec)]
        return self._moments([paramVec], where)[0][0]

    def var(self, paramVec, where=None):
        """chains.py:679-693 (like the reference, a list ignores ``where``)"""
        if isinstance(paramVec, (list, tuple)):
            return np.array([self.var(p) for p in paramVec])
        if where is None and self._is_plain_column(paramVec):
            return self.vars[self._col(paramVec)]
        return self._moments([paramVec], where)[1][0, 0]

    def std(self, paramVec, where=None):
        return np.sqrt(self.var(paramVec, where))

    # ---- weighted quantiles (chains.py:782-838) ----------------------------------------------------------
    def confidence(self, paramVec, limfrac, upper=False, start=0, end=None, weights=None):
        """chains.py:814-838: sort-free weighted quantile selection on the device (gd_quantiles)."""
        if isinstance(paramVec, ParamConfidenceData):
            start, end = paramVec.start, paramVec.end
            weights = paramVec.weights if weights is None else weights
        vec_arg = paramVec.vec if isinstance(paramVec, ParamConfidenceData) else self._host_vector(paramVec)
        j = self._vec_col(paramVec)
        limfrac = np.atleast_1d(np.asarray(limfrac, dtype=np.float64))
        end = self.numrows if end is None else end

        def select():
            full = start == 0 and end == self.numrows and weights is None
            norm = self.norm if full else self.ctx.weight_stats(start, end)["norm"]
            targets = norm * limfrac if not upper else norm * (1 - limfrac)
            mm = self._minmax_of([j]) if (vec_arg is None and j < self.n) else None
            return self.ctx.quantiles([j], targets[None, :], lo=start, hi=end, minmax=mm)[0]

        if weights is None:
            out = select()
        else:
            weights = np.asarray(weights, dtype=np.float64)
            if weights.shape != (self.numrows,):
                raise WeightedSampleError("weights must have one entry per sample")
            out = self._with_weights(weights, select)
        return out if out.size > 1 else out[0]

    def twoTailLimits(self, paramVec, confidence):
        limits = np.array([(1 - confidence) / 2, 1 - (1 - confidence) / 2])
        return self.confidence(paramVec, limits)

    # ---- autocorrelation / effective samples (chains.py:423-574) -------------------------------------------
    # batched 2D calls of this many pairs convolve their batches on two streams (below: not worth a second context's
    # plans and scratch; above: one batch fills the chip)
    CONV_TWO_STREAMS_PAIRS = (64, 400)
    # large calls: the share of the base grid's pairs whose bandwidths are optimised first, so that their convolution
    # (second stream) runs beside the optimisation of the rest
    KOPT_FIRST_FRACTION = 0.5
    KOPT_SPLIT_MIN = 256  # ... when the launch has at least this many pairs
    DIRECT_LAGS_MAX = 512  # beyond this many lags the length-2N FFT (gd_autoconvolve) is cheaper than lag sums

    def _autocov(self, col, mean, k0, nlags):
        """Un-normalised autocovariance lag sums sum_i d_i d_{i+k}, d = (x - mean) w, for k0 <= k < k0 + nlags: direct
        lag sums for a few lags, the reference's FFT route (convolve.py:458-478 on the device) for many."""
        if nlags <= self.DIRECT_LAGS_MAX:
            return self.ctx.autocov_lags(col, mean, k0, nlags)
        from .convolve import nearestFFTnumber

        s = int(nearestFFTnumber(2 * self.numrows))
        return self.ctx.autoconvolve(s, k0 + nlags, False, col=col, mean=mean, use_weights=self.weights is not None)[k0:]

    def getAutocorrelation(self, paramVec, maxOff=None, weight_units=True, normalized=True):
        """chains.py:423-447; ``paramVec`` may be a parameter or a vector of one value per sample"""
        j = self._vec_col(paramVec)
        if maxOff is None:
            maxOff = self.n - 1
        lags = self._autocov(j, self.mean(paramVec), 0, maxOff + 1)
        corr = lags / np.arange(self.numrows, self.numrows - (maxOff + 1), -1)
        if normalized:
            corr /= self.var(paramVec)
        if weight_units:
            return corr * self.numrows / self.norm
        return corr

    def getCorrelationLength(self, j, weight_units=True, min_corr=0.05, corr=None):
        """chains.py:449-466.  Without ``corr``: direct lag sums in growing chunks with early exit; a
        chain whose correlation has not dropped below ``min_corr`` within DIRECT_LAGS_MAX lags takes the FFT route for
        all N/10 lags at once."""
        if corr is not None:
            corr = np.asarray(corr)
            ix = int(np.argmin(corr > min_corr * corr[0]))
            return corr[0] + 2 * np.sum(corr[1:ix])
        col = self._vec_col(j)
        mean, var = self.mean(j), self.var(j)
        max_off = self.numrows // 10
        scale = (self.numrows / self.norm) if weight_units else 1.0
        vals = np.zeros(0)
        k0, chunk = 0, 32
        while k0 <= max_off:
            nl = min(chunk, max_off + 1 - k0)
            if k0 + nl > self.DIRECT_LAGS_MAX:
                nl = max_off + 1 - k0  # everything that is left, in one transform
            lags = self._autocov(col, mean, k0, nl)
            c = lags / (self.numrows - np.arange(k0, k0 + nl)) / var * scale
            vals = np.concatenate([vals, c])
            below = np.nonzero(~(vals > min_corr * vals[0]))[0]
            if below.size:
                return vals[0] + 2 * float(np.sum(vals[1:int(below[0])]))
            k0 += nl
            chunk *= 2
        return vals[0]  # argmin of an all-True mask is 0 (chains.py:464-465)

    def getEffectiveSamples(self, j=0, min_corr=0.05):
        return self.norm / self.getCorrelationLength(j, min_corr=min_corr)

    def getEffectiveSamplesGaussianKDE(self, paramVec, h=0.2, scale=None, maxoff=None, min_corr=0.05):
        """chains.py:477-574; the lag sums with the Gaussian kernel run on the GPU."""
        if self.sampler in ("nested", "uncorrelated"):
            return self.norm**2 / self._sum_w2
        j = self._col(paramVec)
        kernel_std = (scale or self.sddev[j]) * h
        if maxoff is None:
            maxoff = int(self.getCorrelationLength(j, weight_units=False) * 1.5) + 4
        return self._neff_from_lags(j, kernel_std, maxoff, min_corr, None)

    def _neff_lag_list(self, tail=2):
        """The lags of the batched kernel-sum launch: the five of the uncorrelated term (chains.py:514-519) and the first
        ``tail`` of the scan (corr_k(1), corr_k(2): :541-545).  corr_k(2) is needed only by a chain that is still correlated
        at lag 1; _neff_batch asks for it up front (tail = 2) unless the autocorrelation probe shows every column of the
        batch below the threshold already at lag 1 -- then the launch carries six exponentials per sample instead of seven,
        and a column the probe misjudged fetches its lag 2 by itself (same value, one more launch)."""
        uncorr_len = self.numrows // 2
        return list(range(uncorr_len, uncorr_len + 5)) + [k for k in (1, 2)[:tail] if k <= self.numrows // 10]

    def _neff_from_lags(self, j, kernel_std, maxoff, min_corr, seed_sums):
        """The scalar part of chains.py:509-574 given (optionally pre-computed) Gaussian-kernel lag sums."""
        maxoff = min(maxoff, self.numrows // 10)
        uncorr_len = self.numrows // 2
        inv4s2 = 1.0 / (4 * kernel_std**2)
        lags = self._neff_lag_list() if seed_sums is None else self._neff_lag_list(len(seed_sums) - 5)
        sums = self.ctx.kde_lag_sums(j, inv4s2, lags) if seed_sums is None else seed_sums
        nav = sum(self.numrows - k for k in range(uncorr_len, uncorr_len + 5))
        uncorr_term = float(np.sum(sums[:5])) / nav
        n = float(self.numrows)
        cache = {k: sums[5 + i] for i, k in enumerate(lags[5:])}

        def corr_k(k):
            if k not in cache:
                cache[k] = self.ctx.kde_lag_sums(j, inv4s2, [k])[0]
            return cache[k] - (n - k) * uncorr_term

        corr0 = self._sum_w2
        threshold = min_corr * corr0
        c1 = corr_k(1)
        if c1 < threshold:
            N = corr0
        else:
            c2 = corr_k(2)
            if c2 > threshold:
                max_k = maxoff
                while max_k > 10:
                    if corr_k(max_k // 3) >= threshold:
                        break
                    max_k //= 3
                step_size = 1 if max_k < 20 else max_k // 10
                cum_sum = c1 + c2
                for k in range(3, maxoff + 1, step_size):
                    test_val = corr_k(k)
                    if test_val < threshold:
                        break
                    cum_sum += test_val * step_size if k > 3 else (test_val * step_size) / 2
                N = corr0 + 2 * cum_sum
            else:
                N = corr0 + 2 * c1
        return self.norm**2 / N

    def _probe_lags(self, todo, nl):
        """The first ``nl`` autocovariance lag sums of columns ``todo``: taken from the prefetch started by
        prepareParams on the second context (they need the means only, so they ran beside the quantile select), else
        computed now."""
        pre = getattr(self, "_lag_prefetch", None)
        self._lag_prefetch = None
        if pre is not None:
            cols, pnl, fut = pre
            try:
                lags = fut.result()
            except Exception:
                lags = None
            if lags is not None and pnl == nl and set(todo) <= set(cols):
                row = {c: k for k, c in enumerate(cols)}
                return lags[[row[c] for c in todo]]
        return self.ctx.autocov_lags_batch(todo, self.means[todo], 0, nl)

    def _neff_batch(self, js, min_corr=0.05):
        """_get1DNeff for many parameters with two batched launches (32 autocovariance lags, 7 kernel lag sums)."""
        todo = [j for j in js if self.paramNames.names[j].N_eff_kde is None]
        if not todo:
            return
        share = getattr(self, "_neff_share", None)
        if share is not None:
            # multi-rank runs (parallel.NeffShare): this rank computes the parameters it owns; the others arrive by an
            # exchange that is always issued from the main thread (collectives of a process stay on one thread), i.e.
            # here, or -- when this runs on the helper thread beside the binning -- by the caller's _neff_complete
            self._neff_share = None
            try:
                self._neff_batch([j for j in todo if j in share.params], min_corr)
            finally:
                self._neff_share = share
            if threading.current_thread() is threading.main_thread():
                self._neff_complete(js, min_corr)
            return
        if self.sampler in ("nested", "uncorrelated"):
            for j in todo:
                self.paramNames.names[j].N_eff_kde = self.norm**2 / self._sum_w2
            return
        max_off = self.numrows // 10
        nl = min(8, max_off + 1)  # short probe first; correlated chains continue in getCorrelationLength
        lag0 = self._probe_lags(todo, nl)
        kstd, maxoffs = [], []
        # the probe of all columns at once: c[row, k] = autocovariance at lag k over the variance (chains.py:449-466)
        C_all = np.asarray(lag0) / (self.numrows - np.arange(nl)) / np.asarray(self.vars)[todo][:, None]
        below_all = ~(C_all > min_corr * C_all[:, :1])
        first_below = np.where(below_all.any(axis=1), below_all.argmax(axis=1), -1).tolist()
        for row, j in enumerate(todo):
            par = self.paramNames.names[j]
            c = C_all[row]
            if first_below[row] >= 0:
                corrlen = c[0] + 2 * float(np.sum(c[1:first_below[row]]))
            elif nl == max_off + 1:
                corrlen = c[0]
            else:
                corrlen = self.getCorrelationLength(j, weight_units=False, min_corr=min_corr)
            kstd.append((par.sigma_range or self.sddev[j]) * 0.2)
            maxoffs.append(int(corrlen * 1.5) + 4)
        # (the rule of csrc/batch2d.hpp neff_batch: lag 2 rides along unless every column is uncorrelated at lag 1 by the probe)
        tail = 1 if all(fb == 1 for fb in first_below) else 2
        sums = self.ctx.kde_lag_sums_batch(todo, [1.0 / (4 * k**2) for k in kstd], self._neff_lag_list(tail))
        for row, j in enumerate(todo):
            self.paramNames.names[j].N_eff_kde = self._neff_from_lags(j, kstd[row], maxoffs[row], min_corr, sums[row])

    def _neff_complete(self, js, min_corr=0.05):
        """Multi-rank runs: fetch the N_eff values of the parameters other ranks own (parallel.NeffShare.exchange), then
        compute whatever nobody owned.  Main thread only."""
        share = getattr(self, "_neff_share", None)
        if share is None:
            return
        if not getattr(share, "exchanged", False):
            # unconditional, once per step on every rank: a rank whose own parameters cover its pairs must still enter
            # the collective the other ranks are waiting in
            share.exchanged = True
            share.exchange(self)
        self._neff_share = None
        try:
            self._neff_batch(js, min_corr)  # owned by nobody: computed here
        finally:
            self._neff_share = share

    def getEffectiveSamplesGaussianKDE_2d(self, i, j, h=0.3, maxoff=None, min_corr=0.05):
        """chains.py:576-635 (used when use_effective_samples_2D is set); lag sums on the GPU, 8 lags per launch."""
        if self.sampler in ("nested", "uncorrelated"):
            return self.norm**2 / self._sum_w2
        i, j = self._col(i), self._col(j)
        cov = self.getCov(pars=[i, j])
        if abs(cov[0, 1]) > np.sqrt(cov[0, 0] * cov[1, 1]) * 0.999:
            return self.getEffectiveSamplesGaussianKDE(i, h=h, min_corr=min_corr)  # totally correlated: 1D estimate
        kernel_inv = np.linalg.inv(cov) / h**2
        kinv3 = [kernel_inv[0, 0], kernel_inv[0, 1] + kernel_inv[1, 0], kernel_inv[1, 1]]
        if maxoff is None:
            maxoff = int(max(self.getCorrelationLength(i, weight_units=False),
                             self.getCorrelationLength(j, weight_units=False)) * 1.5) + 4
        maxoff = min(maxoff, self.numrows // 10)
        uncorr_len = self.numrows // 2
        sums = self.ctx.kde_lag_sums_2d(i, j, kinv3, list(range(uncorr_len, uncorr_len + 5)))
        nav = sum(self.numrows - k for k in range(uncorr_len, uncorr_len + 5))
        uncorr_term = float(np.sum(sums)) / nav
        corr0 = self._sum_w2
        n = float(self.numrows)
        total = 0.0
        k = 1
        done = False
        while k <= maxoff and not done:
            lags = list(range(k, min(k + 8, maxoff + 1)))
            vals = self.ctx.kde_lag_sums_2d(i, j, kinv3, lags)
            for kk, v in zip(lags, vals):
                c = v - (n - kk) * uncorr_term
                if c < min_corr * corr0:
                    done = True
                    break
                total += c
            k += len(lags)
        N = corr0 + 2 * total
        return self.norm**2 / N

    def _get1DNeff(self, par, param):
        """mcsamples.py:1230-1235"""
        if par.N_eff_kde is None:
            par.N_eff_kde = self.getEffectiveSamplesGaussianKDE(param, scale=par.sigma_range)
        return par.N_eff_kde

    # ---- ranges and limits (mcsamples.py:1421-1498) --------------------------------------------------------
    def _initParamRanges(self, j, paramConfid=None):
        self._init_params([self._col(j)])
        return self.paramNames.names[self._col(j)]

    def _setNDLimits(self):
        """
        The N-dimensional confidence-region limits of _setLikeStats (mcsamples.py:2263-2274): per contour, min / max of
        every parameter over the best-likelihood samples holding that fraction of the weight.  The reference argsorts
        loglikes; here the weighted quantile of the loglikes column (gd_quantiles) is the likelihood of the first
        sample outside the region, and one conditional min/max pass per contour (gd_col_minmax) does the rest.  Rows
        tied with that threshold are all excluded (the reference keeps an arbitrary subset of them).
        """
        if self._nd_limits_done:
            return
        ctx = self.ctx
        col = getattr(self, "_loglikes_col", None)  # (resident already when _setLikeStats is the caller)
        if col is None:
            col = ctx.set_extra_column(ctx.EXTRA_COLS - 1, self.loglikes)
        contours = np.asarray(self.contours, dtype=np.float64)
        thr = ctx.quantiles([col], (self.norm * contours)[None, :])[0]
        lims = np.empty((len(contours), self.n, 2))
        for i, (c, t) in enumerate(zip(contours, thr)):
            lims[i] = ctx.col_minmax(list(range(self.n)), cond_col=-1 if c >= 1 else col, cond_below=t)
        for j, par in enumerate(self.paramNames.names):
            par.ND_limit_bot = lims[:, j, 0].copy()
            par.ND_limit_top = lims[:, j, 1].copy()
        self._nd_limits_done = True

    def _minmax_of(self, js):
        """(len(js), 2) minima / maxima of resident columns from the base statistics: with them the quantile select
        needs two reads of a column instead of four (gd_quantiles_mm)."""
        if getattr(self, "_col_min", None) is None:
            return None
        js = np.asarray(js, dtype=np.int64)
        return np.stack([np.asarray(self._col_min)[js], np.asarray(self._col_max)[js]], axis=1)

    def _init_params(self, js, lag_probe=False):
        """_initParam for several parameters with ONE batched quantile-select launch.  ``lag_probe``: the select's counting
        pass also delivers the autocovariance probe of the N_eff estimate for the same columns (one read of the samples
        serves both; kept in ``_lag_prefetch`` for _probe_lags / the batched 2D entry)."""
        todo = [j for j in dict.fromkeys(js) if not getattr(self.paramNames.names[j], "_ranges_done", False)]
        if not todo:
            return
        rc = self.range_confidence
        fracs = np.array([rc, 1 - rc] + list(np.linspace(0.1, 0.9, 9)))
        targets = np.tile(self.norm * fracs, (len(todo), 1))
        if lag_probe:
            q, lags = self.ctx.quantiles_probe(todo, targets, self._minmax_of(todo), self.means[todo])
            q = np.asarray(q)
            if lags is not None:
                self._lag_prefetch = (todo, 8, _Done(lags))
        else:
            q = np.asarray(self.ctx.quantiles(todo, targets, minmax=self._minmax_of(todo)))
        # mcsamples.py:1440-1452 for all parameters at once: [param_min, deciles 0.1..0.9, param_max], spans of four
        err_v = np.asarray(self.sddev)[todo]
        confids = np.empty_like(q)
        confids[:, 0] = np.asarray(self._col_min)[todo]
        confids[:, 1:-1] = q[:, 2:]
        confids[:, -1] = np.asarray(self._col_max)[todo]
        diffs = confids[:, 4:] - confids[:, :-4]
        scale_v = np.min(diffs, axis=1) / 1.049
        flat_v = (np.all(diffs > (err_v * 1.049)[:, None], axis=1) & np.all(diffs < (scale_v * 1.5)[:, None], axis=1)).tolist()
        for row, j in enumerate(todo):
            par = self.paramNames.names[j]
            par.err = self.sddev[j]
            par.mean = self.means[j]
            par.param_min = self._col_min[j]
            par.param_max = self._col_max[j]
            par.range_min, par.range_max = q[row, 0], q[row, 1]
            scale = scale_v[row]
            if flat_v[row]:
                par.sigma_range = scale  # very flat
            else:
                par.sigma_range = min(par.err, scale)
            if self.range_ND_contour >= 0 and self.loglikes is not None:  # mcsamples.py:1455-1459
                self._setNDLimits()
                if self.range_ND_contour >= par.ND_limit_bot.size:
                    raise SettingError("range_ND_contour should be -1 (off), or an index into the computed contour levels")
                par.range_min = min(max(par.range_min - par.err, par.ND_limit_bot[self.range_ND_contour]), par.range_min)
                par.range_max = max(max(par.range_max + par.err, par.ND_limit_top[self.range_ND_contour]), par.range_max)
            smooth_1D = par.sigma_range * 0.4
            if par.has_limits_bot:
                if par.range_min - par.limmin > 2 * smooth_1D and par.param_min - par.limmin > smooth_1D:
                    par.has_limits_bot = False  # long way from limit
                else:
                    par.range_min = par.limmin
            if par.has_limits_top:
                if par.limmax - par.range_max > 2 * smooth_1D and par.limmax - par.param_max > smooth_1D:
                    par.has_limits_top = False
                else:
                    par.range_max = par.limmax
            if not par.has_limits_bot:
                par.range_min -= smooth_1D * 2
            if not par.has_limits_top:
                par.range_max += smooth_1D * 2
            par.has_limits = par.has_limits_top or par.has_limits_bot
            par._ranges_done = True

    def prepareParams(self, params=None, neff=True):
        """
        Additive API: compute the per-parameter state every density needs (ranges, limits, sigma_range and, if
        ``neff``, the KDE effective sample number) for ``params`` (default all).  The reference recomputes this inside
        every density call (mcsamples.py:1786-1787); it is a pure function of the column, so doing it once is
        result-preserving.
        """
        if self.needs_update:
            self.updateBaseStatistics()
        js = list(range(self.n)) if params is None else [self._col(p) for p in params]
        todo = [j for j in js if self.paramNames.names[j].N_eff_kde is None]
        # the autocovariance probe of the N_eff estimate needs the means only.  Round 6: it rides on the counting pass of the
        # quantile select (gd_quantiles_mm_probe: one read of the columns for both) when every parameter that needs it is
        # about to go through that select; else, as before, it runs on the second context beside the select
        fused = (bool(todo) and self.sampler not in ("nested", "uncorrelated") and self.numrows // 10 + 1 >= 8
                 and hasattr(self.ctx, "quantiles_probe") and os.environ.get("GETDIST_AMD_FUSED_PROBE", "1") == "1"
                 and all(not getattr(self.paramNames.names[j], "_ranges_done", False) for j in todo))
        if fused:
            with _Phase(self, "prep.ranges"):
                self._init_params(js, lag_probe=True)
        elif (todo and self._lane == 0 and not self._timing and self.sampler not in ("nested", "uncorrelated")
                and len(todo) >= 2 and os.environ.get("GETDIST_AMD_OVERLAP_NEFF", "1") == "1"):
            # the autocovariance probe of the N_eff estimate depends on the means only: start it on the second context
            # (own stream) while this one runs the quantile select
            twin = self._second_lane()
            self._nlanes = 1
            nl = min(8, self.numrows // 10 + 1)
            self._lag_prefetch = (todo, nl, self._lane_thread(twin).submit(
                twin.ctx.autocov_lags_batch, todo, self.means[todo], 0, nl))
        with _Phase(self, "prep.ranges"), _FastThreadSwitch(getattr(self, "_lag_prefetch", None) is not None):
            self._init_params(js)
        _hostlog("prep: ranges done")
        if neff:
            with _Phase(self, "prep.neff"):
                self._neff_batch(js)
            _hostlog("prep: N_eff done")
        return js

    def _bin_edge_arrays(self, js, borderfrac=0.1):
        """binmin, binmax of _bin_edges for the parameters ``js``, as arrays indexed by parameter number (the same
        fp64 operations, element by element)."""
        names = self.paramNames.names
        n = max(js) + 1
        rmin, rmax, pmin, pmax = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
        lim_b, lim_t = np.zeros(n, dtype=bool), np.zeros(n, dtype=bool)
        for j in js:
            p = names[j]
            rmin[j], rmax[j], pmin[j], pmax[j] = p.range_min, p.range_max, p.param_min, p.param_max
            lim_b[j], lim_t[j] = bool(p.has_limits_bot), bool(p.has_limits_top)
        border = (rmax - rmin) * borderfrac
        binmin = np.minimum(pmin, rmin)
        binmin = np.where(lim_b, binmin, binmin - border)
        binmax = np.maximum(pmax, rmax)
        binmax = np.where(lim_t, binmax, binmax + border)
        return binmin, binmax

    @staticmethod
    def _bin_edges(par, num_fine_bins, borderfrac=0.1):
        """The scalar half of _binSamples (mcsamples.py:1486-1496); the index half runs fused in the kernels."""
        border = (par.range_max - par.range_min) * borderfrac
        binmin = min(par.param_min, par.range_min)
        if not par.has_limits_bot:
            binmin -= border
        binmax = max(par.param_max, par.range_max)
        if not par.has_limits_top:
            binmax += border
        fine_width = (binmax - binmin) / (num_fine_bins - 1)
        return fine_width, binmin, binmax

    # ---- 1D densities (mcsamples.py:1237-1283, 1500-1686) ---------------------------------------------------
    def getAutoBandwidth1D(self, bins, par, param, mult_bias_correction_order=None, kernel_order=1, N_eff=None):
        if N_eff is None:
            N_eff = self._get1DNeff(par, param)
        h, status = self.ctx.isj1d(np.asarray(bins, dtype=np.float64)[None, :], [N_eff])
        return self._bandwidth_1d(None if status[0] else h[0], par, N_eff, mult_bias_correction_order, kernel_order)

    def _no_bandwidth_warning(self, par):
        """mcsamples.py:1259-1261: parameters for which a failed / very small 1D bandwidth neither warns nor raises."""
        return par.name in self.no_warning_params or (
            bool(self.no_warning_chi2_params) and ("chi2_" in par.name or "minuslog" in par.name))

    def _bandwidth_1d(self, h, par, N_eff, mult_bias_correction_order, kernel_order):
        """The scalar tail of getAutoBandwidth1D (mcsamples.py:1256-1283) given the device's ISJ solution ``h`` (None
        where the solver failed): rule-of-thumb fallback when it failed or is very small, higher-order rescaling."""
        if h is None:
            logging.warning("1D auto bandwidth failed. Using fallback: zero f in _bandwidth_fixed_point (non-convergence)")
        bin_range = max(par.param_max, par.range_max) - min(par.param_min, par.range_min)
        if h is None or h < 0.01 * N_eff ** (-1.0 / 5) * (par.range_max - par.range_min) / bin_range:
            hnew = 1.06 * par.sigma_range * N_eff ** (-1.0 / 5) / bin_range
            if not self._no_bandwidth_warning(par):
                msg = f"auto bandwidth for {par.name} very small or failed (h={h},N_eff={N_eff}). Using fallback (h={hnew})"
                if self.raise_on_bandwidth_errors:
                    raise BandwidthError(msg)
                logging.warning(msg)
            h = hnew
        par.kde_h = h
        m = self.mult_bias_correction_order if mult_bias_correction_order is None else mult_bias_correction_order
        if kernel_order > 1:
            m = max(m, 1)
        if m:
            return h * N_eff ** (1.0 / 5 - 1.0 / (4 * m + 5))
        return h

    def get1DDensity(self, name, **kwargs):
        if self.needs_update:
            self.updateBaseStatistics()
        if not kwargs:
            j, par = self._parAndNumber(name)
            if par is not None and par.name in self.density1D:
                return self.density1D[par.name]
        return self.get1DDensityGridData(name, **kwargs)

    def get1DDensityGridData(self, j, paramConfid=None, meanlikes=False, **kwargs):
        if self.needs_update:
            self.updateBaseStatistics()
        j = self._parAndNumber(j)[0]
        if j is None:
            return None
        return self.get1DDensities([j], meanlikes=meanlikes, **kwargs)[0]

    def get1DDensities(self, params=None, meanlikes=False, **kwargs):
        """
        Batched 1D KDEs (additive API): a list of Density1D, one per entry of ``params`` (default: all); with
        ``meanlikes`` each carries the mean-likelihood profile ``likes`` (mcsamples.py:1556-1561,1672-1682).
        """
        if self.needs_update:
            self.updateBaseStatistics()
        for k in kwargs:
            if k not in ("smooth_scale_1D", "boundary_correction_order", "mult_bias_correction_order", "fine_bins",
                         "num_bins"):
                raise SettingError("unknown 1D density argument %s" % k)
        js = list(range(self.n)) if params is None else [self._col(p) for p in params]
        num_bins = kwargs.get("num_bins", self.num_bins)
        smooth_scale_1D = kwargs.get("smooth_scale_1D", self.smooth_scale_1D)
        bco = kwargs.get("boundary_correction_order", self.boundary_correction_order)
        mbc = kwargs.get("mult_bias_correction_order", self.mult_bias_correction_order)
        fine_bins = kwargs.get("fine_bins", self.fine_bins)
        if bco > 2:
            raise SettingError("Unknown boundary_correction_order (expected 0, 1, 2)")
        self._init_params(js)
        pars = [self.paramNames.names[j] for j in js]
        if hasattr(self.ctx, "density1d_batch") and os.environ.get("GETDIST_AMD_NATIVE_BATCH", "1") == "1":
            # ONE native call (csrc/batch1d.hpp) for every case: in a multi-rank job this rank's share of the N_eff values and
            # the exchange happen first (the call then finds every value), a parameter listed twice is computed once
            from . import batch1d

            if getattr(self, "_neff_share", None) is not None and smooth_scale_1D <= 0:
                self._neff_batch(js)
            uniq = list(dict.fromkeys(js))
            P, hist, meta = batch1d.run(self, uniq, fine_bins, num_bins, smooth_scale_1D, bco, mbc, want_hist=meanlikes)
            if len(uniq) != len(js):
                at = {j: b for b, j in enumerate(uniq)}
                rows = [at[j] for j in js]
                P, meta = P[rows], meta[rows]
                hist = None if hist is None else hist[rows]
            edges = [((meta[b, 1] - meta[b, 0]) / (fine_bins - 1), meta[b, 0], meta[b, 1]) for b in range(len(js))]
            smooth, winw = meta[:, 3].tolist(), meta[:, 4].astype(np.int64).tolist()
            flags = [(1 if par.has_limits_bot else 0) | (2 if par.has_limits_top else 0) | (4 if par.periodic else 0)
                     for par in pars]
            return self._finish_1d(js, pars, edges, P, hist, smooth, winw, flags, fine_bins, meanlikes, kwargs)
        # (no native entry on this context: the Python-planned sequence is tests/planned_route.py, as for the 2D path)
        route = getattr(MCSamples, "_planned_route_1d", None)
        if route is None:
            raise MCSamplesError("get1DDensities needs a device context with gd_density1d_batch (libgdhip)")
        return route(self, js, pars, fine_bins, num_bins, smooth_scale_1D, bco, mbc, meanlikes, kwargs)

    def _finish_1d(self, js, pars, edges, P, hist, smooth, winw, flags, fine_bins, meanlikes, kwargs):
        """Mean-likelihood profiles (mcsamples.py:1556-1561,1672-1682) and the Density1D objects of get1DDensities."""
        likes = None
        if meanlikes:
            shade = bool(self.shade_likes_is_mean_loglikes)
            likehist = self._like_histograms(1 if shade else 0, lambda: self.ctx.hist1d(
                js, [e[1] for e in edges], [e[0] for e in edges], fine_bins))
            likes, _ = self.ctx.likes1d(hist, likehist, P, smooth, winw, flags, shade)
        out = []
        for b, (j, par) in enumerate(zip(js, pars)):
            fine_width, binmin, binmax = edges[b]
            d = Density1D(np.linspace(binmin, binmax, fine_bins), P=P[b].copy(), view_ranges=[par.range_min, par.range_max])
            d.likes = None if likes is None else likes[b].copy()
            if not kwargs:
                self.density1D[par.name] = d
            out.append(d)
        return out

    # ---- marginalised limits (mcsamples.py:2353-2367, 2442-2531) -----------------------------------------------
    def _max_frac_twotail(self):
        """mcsamples.py:427-433: how small the end bin must be relative to the maximum to use a two-tail limit"""
        from scipy.stats import norm
        import math

        over = getattr(self, "_max_frac_overrides", {})  # max_frac_twotailN of the .ini (mcsamples.py:429-430)
        return [over.get(i, np.exp(-1.0 * math.pow(norm.ppf((1 - c) / 2), 2) / 2)) for i, c in enumerate(self.contours)]

    def _marge_limit_inputs(self, js, densities):
        """
        Everything the marginalised limits of parameters ``js`` need from the device, in three batched calls:
        equal-density credible intervals of every (density, contour) (gd_limits1d), and the one- and two-tail sample
        quantiles of every (column, contour) (gd_quantiles; targets per contour: f, 1-f, f/2, 1-f/2 with f = 1-contour).
        Returns (credible[len(js), nc, 4], tails[len(js), nc, 4]).
        """
        contours = np.asarray(self.contours, dtype=np.float64)
        nc = len(contours)
        credible = np.zeros((len(js), nc, 4))
        by_F = {}
        for b, d in enumerate(densities):
            by_F.setdefault(d.P.size, []).append(b)
        for F, members in by_F.items():
            for c0 in range(0, nc, 8):
                lims, status = self.ctx.limits1d(np.stack([densities[b].P for b in members]),
                                                 [densities[b].x[0] for b in members],
                                                 [densities[b].spacing for b in members], contours[c0:c0 + 8])
                if np.any(status != 0):
                    raise DensitiesError("credible limit outside the refined density grid")
                credible[members, c0:c0 + 8] = lims
        tails = np.zeros((len(js), nc, 4))
        f = 1 - contours
        for c0 in range(0, nc, 4):
            fr = f[c0:c0 + 4]
            fracs = np.stack([fr, 1 - fr, fr / 2, 1 - fr / 2], axis=1).reshape(-1)
            q = self.ctx.quantiles(js, np.tile(self.norm * fracs, (len(js), 1)), minmax=self._minmax_of(js))
            tails[:, c0:c0 + 4] = q.reshape(len(js), -1, 4)
        return credible, tails

    def _assign_marge_limits(self, par, density, credible, tails, max_frac_twotail):
        """
        The limit type and values per contour (contract of mcsamples.py:2460-2531).  A hard prior edge where the
        density is still high (above max_frac_twotail of the peak) makes that side unconstrained; otherwise the
        equal-density interval of the smoothed density decides which sides are open, open sides take the prior range,
        a single closed side takes the one-tail sample quantile, and two closed sides take the equal-density interval
        -- or the two-tail quantiles when the density is nearly equal at them (credible_interval_threshold).
        """
        force_twotail = getattr(self, "force_twotail", False)
        par.limits = []
        for c in range(len(self.contours)):
            open_bot = bool(par.has_limits_bot and not force_twotail and density.P[0] > max_frac_twotail[c])
            open_top = bool(par.has_limits_top and not force_twotail and density.P[-1] > max_frac_twotail[c])
            lower, upper = par.range_min, par.range_max
            if not (open_bot and open_top):
                cred_lo, cred_hi, open_bot, open_top = credible[c]
                open_bot, open_top = bool(open_bot), bool(open_top)
                one_lo, one_hi, two_lo, two_hi = tails[c]
                if not open_bot and not open_top:
                    lower, upper = cred_lo, cred_hi
                    if abs(density.Prob(two_hi) - density.Prob(two_lo)) < self.credible_interval_threshold:
                        lower, upper = two_lo, two_hi
                elif not open_bot:
                    lower = one_lo
                elif not open_top:
                    upper = one_hi
            tag = {(True, True): "none", (True, False): ">", (False, True): "<", (False, False): "two"}[(open_bot, open_top)]
            par.limits.append(ParamLimit([lower, upper], tag))

    def _setMargeLimits(self, par, paramConfid=None, max_frac_twotail=None, density1D=None):
        """Marginalised limits of ONE parameter (the reference's per-parameter entry point, mcsamples.py:2460)."""
        j = self._col(par.name)
        density1D = density1D or self.get1DDensity(par.name)
        credible, tails = self._marge_limit_inputs([j], [density1D])
        self._assign_marge_limits(par, density1D, credible[0], tails[0], max_frac_twotail or self._max_frac_twotail())

    def getMargeStats(self, include_bestfit=False):
        """mcsamples.py:2353-2367: marginalised 1D constraints (numbers only; text tables are out of scope).  All
        densities, all equal-density intervals and all tail quantiles come from batched device calls."""
        if include_bestfit:
            raise NotImplementedError("best-fit files are outside the accelerated path")
        if self.needs_update:
            self.updateBaseStatistics()
        dens = self.get1DDensities()  # one batched launch, cached per name
        mft = self._max_frac_twotail()
        js = list(range(self.n))
        credible, tails = self._marge_limit_inputs(js, dens)
        for j, par in enumerate(self.paramNames.names):
            self._assign_marge_limits(par, dens[j], credible[j], tails[j], mft)
        return MargeStats(self.paramNames.names, self.contours)

    # ---- 2D densities (mcsamples.py:1285-1419, 1730-2010) ---------------------------------------------------
    def get2DDensity(self, x, y, normalized=False, **kwargs):
        if self.needs_update:
            self.updateBaseStatistics()
        density = self.get2DDensityGridData(x, y, get_density=True, **kwargs)
        if density is not None and normalized:
            density.normalize(in_place=True)
        return density

    def get2DDensityGridData(self, j, j2, num_plot_contours=None, get_density=False, meanlikes=False,
                             mask_function=None, **kwargs):
        if self.needs_update:
            self.updateBaseStatistics()
        j = self._parAndNumber(j)[0]
        j2 = self._parAndNumber(j2)[0]
        if j is None or j2 is None:
            return None
        density = self.get2DDensities([(j, j2)], num_plot_contours=num_plot_contours, get_density=get_density,
                                      meanlikes=meanlikes, mask_function=mask_function, **kwargs)[0]
        density.P  # a single-pair call delivers (and raises "no samples in bin") here, like the reference
        return density

    def triangleDensities(self, params=None, **kwargs):
        """All lower-triangle pairs (x=params[i], y=params[i2>i]) in triangle-plot order; returns (pairs, densities)."""
        names = list(range(self.n)) if params is None else [self._col(p) for p in params]
        pairs = [(names[i], names[i2]) for i in range(len(names)) for i2 in range(i + 1, len(names))]
        return pairs, self.get2DDensities(pairs, **kwargs)

    def getAutoBandwidth2D(self, bins, parx, pary, paramx, paramy, corr, rangex, rangey, base_fine_bins_2D,
                           mult_bias_correction_order=None, min_corr=0.2, N_eff=None, use_2D_Neff=False):
        """Per-pair entry point with the reference's signature (mcsamples.py:1285-1419); ``bins`` is a host F x F grid."""
        bins = np.ascontiguousarray(bins, dtype=np.float64)
        F = bins.shape[0]
        d = self.ctx.alloc(bins.nbytes)
        d.from_host(bins)
        plan = self._bandwidth_plan([(paramx, paramy)], [corr], [(rangex, rangey)], base_fine_bins_2D, N_eff=N_eff)
        res = self._bandwidth_2d(plan, {F: (d, [0])}, [F], base_fine_bins_2D, mult_bias_correction_order)
        return tuple(res[0].tolist())

    def _bandwidth_plan(self, pairs, corrs, ranges_xy, base_F, min_corr=0.2, N_eff=None, defer_neff=False):
        """Branch selection per pair (mcsamples.py:1325-1409), scalars only.  The classification is evaluated on arrays
        over the pairs (a triangle has thousands); powers stay Python-float operations so that every scalar is the one the
        reference computes.  With ``defer_neff`` the effective sample numbers are not touched yet (their kernels may
        still be running on a helper thread): returns (plan, fill) and ``fill()`` completes the entries later."""
        names = self.paramNames.names
        npairs = len(pairs)
        if npairs == 0:
            return (_Plan(), lambda: None) if defer_neff else _Plan()
        if N_eff is None and not defer_neff:
            self._neff_batch(list(dict.fromkeys([j for p in pairs for j in p])))
        jx = [p[0] for p in pairs]
        jy = [p[1] for p in pairs]
        used = sorted(set(jx) | set(jy))
        at = np.full(max(used) + 1, -1, dtype=np.int64)
        at[used] = np.arange(len(used))
        ix, iy = at[jx], at[jy]
        upar = [names[j] for j in used]
        lim_u = np.array([bool(p.has_limits) for p in upar])
        sig_u = np.array([np.nan if p.sigma_range is None else p.sigma_range for p in upar], dtype=np.float64)
        corr_v = np.asarray(corrs, dtype=np.float64)

        def effective_samples():
            if N_eff is not None:
                return np.full(npairs, float(N_eff))
            if self.use_effective_samples_2D:
                return np.array([self.getEffectiveSamplesGaussianKDE_2d(a, b) if abs(c) < 0.999  # mcsamples.py:1326-1328
                                 else min(self._get1DNeff(names[a], a), self._get1DNeff(names[b], b))
                                 for a, b, c in zip(jx, jy, corr_v.tolist())], dtype=np.float64)
            neff_u = np.array([self._get1DNeff(p, j) for p, j in zip(upar, used)], dtype=np.float64)
            return np.minimum(neff_u[ix], neff_u[iy])

        limx, limy = lim_u[ix], lim_u[iy]
        has_limits = limx | limy
        do_correlated = ~limx | ~limy
        absc = np.abs(corr_v)
        is_A = (min_corr < absc) & (absc <= self.max_corr_2D) & do_correlated
        is_B = ~is_A & ((absc > self.max_corr_2D) | (~do_correlated & (corr_v > 0.8)))
        rng = np.asarray(ranges_xy, dtype=np.float64).reshape(npairs, 2)
        with np.errstate(all="ignore"):
            ratio = np.minimum(sig_u[iy] / rng[:, 1], sig_u[ix] / rng[:, 0]).tolist()
        branch = np.where(is_A, "A", np.where(is_B, "B", "C")).tolist()
        plan = _Plan(dict(jx=a, jy=b, parx=names[a], pary=names[b], corr=c, neff=None, has_limits=hl, rangex=rx_, rangey=ry_,
                          branch=br)
                     for a, b, c, hl, rx_, ry_, br in zip(jx, jy, corr_v.tolist(), has_limits.tolist(),
                                                          rng[:, 0].tolist(), rng[:, 1].tolist(), branch))
        plan.arr = dict(branch=np.where(is_A, 0, np.where(is_B, 1, 2)).astype(np.int8), has_limits=has_limits, corr=corr_v,
                        rangex=rng[:, 0].copy(), rangey=rng[:, 1].copy(), neff=None, fallback_t=None)
        is_C = np.nonzero(~is_A & ~is_B)[0].tolist()

        def fill():
            neff_v = effective_samples()
            neff_l = neff_v.tolist()
            for e, ne in zip(plan, neff_l):
                e["neff"] = ne
            fb = np.full(npairs, np.nan)
            for k in is_C:
                fb[k] = plan[k]["fallback_t"] = (ratio[k] / neff_l[k] ** (1.0 / 6)) ** 2
            plan.arr["neff"], plan.arr["fallback_t"] = neff_v, fb

        if not defer_neff:
            fill()
        for k in np.nonzero(is_A)[0].tolist():
            e = plan[k]
            parx, pary = e["parx"], e["pary"]
            i, j = e["jx"], e["jy"]
            imax, imin = None, None
            if parx.has_limits_bot:
                imin = parx.range_min
            if parx.has_limits_top:
                imax = parx.range_max
            if pary.has_limits:
                i, j = j, i
                if pary.has_limits_bot:
                    imin = pary.range_min
                if pary.has_limits_top:
                    imax = pary.range_max
            cov = self.getCov(pars=[i, j])
            S = np.linalg.cholesky(cov)
            ichol = np.linalg.inv(S)
            S *= ichol[0, 0]
            r = ichol[1, :] / ichol[0, 0]
            e.update(i=i, j=j, imin=imin, imax=imax, S=S, r=r)
        return (plan, fill) if defer_neff else plan

    def _fallback_widths(self, e, ex):
        parx, pary, corr, neff = e["parx"], e["pary"], e["corr"], e["neff"]
        msg = f"2D kernel density bandwidth optimizer failed for {parx.name}, {pary.name}. Using fallback width: {ex}"
        if self.raise_on_bandwidth_errors:
            raise BandwidthError(msg)
        logging.warning(msg)
        _hx = parx.sigma_range / neff ** (1.0 / 6)
        _hy = pary.sigma_range / neff ** (1.0 / 6)
        return _hx, _hy, max(min(corr, self.max_corr_2D), -self.max_corr_2D)

    def _shear_histograms(self, plan, base_F):
        """Branch A of getAutoBandwidth2D (mcsamples.py:1347-1378): min/max of the sheared coordinate and the re-binned
        base_F x base_F histograms of every sheared pair, two batched launches.  Independent of the pairs' own histograms,
        so the caller may run it while those are still being made on the second stream.  None if there is no such pair."""
        A = [k for k, e in enumerate(plan) if e["branch"] == "A"]
        if not A:
            return None
        ctx = self.ctx
        mm = ctx.minmax_affine([plan[k]["i"] for k in A], [plan[k]["j"] for k in A], [plan[k]["r"][0] for k in A],
                               [plan[k]["r"][1] for k in A])
        xmin, dx, ymin, dy, r1s, r2s = [], [], [], [], [], []
        for row, k in enumerate(A):
            e = plan[k]
            # kde.bin_samples(p1, nbins, range_min=imin, range_max=imax) (kde_bandwidth.py:76-87)
            mn, mx = self._col_min[e["i"]], self._col_max[e["i"]]
            delta = mx - mn
            rmin = e["imin"] if e["imin"] is not None else mn - delta * 0.1
            rmax = e["imax"] if e["imax"] is not None else mx + delta * 0.1
            R1 = rmax - rmin
            mn2, mx2 = mm[row]
            delta2 = mx2 - mn2
            rmin2 = mn2 - delta2 * 0.1
            R2 = (mx2 + delta2 * 0.1) - rmin2
            xmin.append(rmin), dx.append(R1 / (base_F - 1)), ymin.append(rmin2), dy.append(R2 / (base_F - 1))
            r1s.append(R1), r2s.append(R2)
        d_rot = ctx.hist2d_sheared([plan[k]["i"] for k in A], [plan[k]["j"] for k in A],
                                   [plan[k]["r"][0] for k in A], [plan[k]["r"][1] for k in A], xmin, dx, ymin, dy,
                                   base_F)
        return dict(d_rot=d_rot, r1s=r1s, r2s=r2s)

    def _bandwidth_2d(self, plan, hists_by_F, pair_F, base_F, mult_bias_correction_order, shear=None, deferred=None,
                      on_chunk=None, first_fraction=None, more_deferred=None):
        """
        getAutoBandwidth2D for a batch (mcsamples.py:1325-1419).  ``hists_by_F``: F -> (device buffer of that class's
        histograms, list of plan indices in buffer order); ``pair_F[k]`` the fine grid size of plan entry k.  Returns the
        (hx, hy, corr) triples in parameter units as an (npair, 3) array.  The whole KernelOptimizer2D -- fixed point,
        functionals, get_h with its TNC minimisations -- runs on the device (gd_kopt2d); here only branch bookkeeping and
        unit conversions, on arrays over the pairs: this code sits between the optimiser's kernels and the convolution's.
        The per-pair records (``plan[k]["kopt"]``) are written by a callable appended to ``deferred`` (the caller runs it
        once the next kernels are enqueued), or at once when ``deferred`` is None.

        ``on_chunk(ks, last, W)`` is called after every optimiser launch with the plan indices whose triples are final
        (rule-of-thumb pairs ride with the first launch); with ``first_fraction`` the base grid's launch is cut in two at
        that fraction of its pairs, so that the caller can convolve the first part (on another stream) while the second
        part is being optimised.
        """
        npair = len(plan)
        ctx = self.ctx
        arr = plan.arr
        W = np.full((npair, 3), np.nan)
        kopt_rows = []  # (plan indices, optimiser output rows) per launch
        m = self.mult_bias_correction_order if mult_bias_correction_order is None else mult_bias_correction_order
        branch = arr["branch"]
        if m:  # higher-order bias correction widens the kernel (mcsamples.py:1412-1416); few distinct N_eff values, each
            # scale a Python-float power as before
            uniq, inv = np.unique(arr["neff"], return_inverse=True)
            widen = np.array([1.1 * ne ** (1.0 / 6 - 1.0 / (2 + 4 * (1 + m))) for ne in uniq.tolist()])[inv]
        else:
            widen = None

        # -- branch A: sheared re-binning at base_F, optimiser with corr=0 and no fallback_t
        A = np.nonzero(branch == 0)[0]
        if shear is None:
            shear = self._shear_histograms(plan, base_F)
        # -- branch B: rule of thumb
        kB = np.nonzero(branch == 1)[0]
        for k in kB.tolist():
            e = plan[k]
            c = max(min(e["corr"], self.max_corr_2D), -self.max_corr_2D)
            W[k] = (e["parx"].sigma_range / e["neff"] ** (1.0 / 6), e["pary"].sigma_range / e["neff"] ** (1.0 / 6), c)
        waiting = [kB]  # final triples not yet reported to on_chunk

        def report(ks, last):
            ks = np.concatenate(waiting + [ks]) if waiting else ks
            waiting.clear()
            if widen is not None:
                W[ks, 0] *= widen[ks]
                W[ks, 1] *= widen[ks]
            if on_chunk is not None and (len(ks) or last):
                on_chunk(ks, last, W)

        def optimise(F, d_batch, ks, row_A, r1, r2):
            """One device-optimiser launch over plan entries ``ks`` (batch order); ``row_A`` marks the sheared rows,
            whose sheared ranges are r1, r2."""
            fb = np.where(row_A, -1.0, arr["fallback_t"][ks])
            corr_in = np.where(row_A, 0.0, arr["corr"][ks])
            out = ctx.kopt2d(d_batch, len(ks), F, arr["neff"][ks], (~arr["has_limits"][ks]).astype(np.int32), fb, corr_in)
            no_root = out[:, 7] != 0
            if np.any(~no_root & (out[:, 11] != 0)):
                raise Exception("bias not positive definite")  # kde_bandwidth.py:229-230, raised out of get_h
            # branch C (the bulk): parameter units = the fractions times the ranges, evaluated on the whole batch
            hx = out[:, 8] * arr["rangex"][ks]
            hy = out[:, 9] * arr["rangey"][ks]
            c = out[:, 10].copy()
            # branch A: de-rotate the sheared kernels (mcsamples.py:1379-1390), kernelC = S K S^T written out for the
            # 2x2 case and evaluated on all sheared pairs at once
            if np.any(row_A):
                ra = np.nonzero(row_A)[0]
                hxa = out[ra, 8] * r1
                hya = out[ra, 9] * r2
                ca = out[ra, 10]
                S = np.array([plan[k]["S"] for k in ks[ra].tolist()])  # (nA, 2, 2)
                k00, k01, k11 = hxa**2, hxa * hya * ca, hya**2
                t00 = S[:, 0, 0] * k00 + S[:, 0, 1] * k01
                t01 = S[:, 0, 0] * k01 + S[:, 0, 1] * k11
                t10 = S[:, 1, 0] * k00 + S[:, 1, 1] * k01
                t11 = S[:, 1, 0] * k01 + S[:, 1, 1] * k11
                c00 = t00 * S[:, 0, 0] + t01 * S[:, 0, 1]
                c01 = t00 * S[:, 1, 0] + t01 * S[:, 1, 1]
                c11 = t10 * S[:, 1, 0] + t11 * S[:, 1, 1]
                sx, sy = np.sqrt(c00), np.sqrt(c11)
                swap = np.array([bool(plan[k]["pary"].has_limits) for k in ks[ra].tolist()])
                hx[ra], hy[ra], c[ra] = np.where(swap, sy, sx), np.where(swap, sx, sy), c01 / np.sqrt(c00 * c11)
            W[ks, 0], W[ks, 1], W[ks, 2] = hx, hy, c
            for row in np.nonzero(no_root)[0].tolist():
                k = int(ks[row])
                plan[k]["kopt"] = out[row]
                W[k] = self._fallback_widths(plan[k], "2D fixed point: no root in [0, 0.1]")
            kopt_rows.append((ks, out))

        # -- branches A and C share the device optimiser: the sheared histograms ride in the same launch as the
        #    base-grid pairs' own histograms (one block per pair; a short extra launch would cost a full block latency)
        item = base_F * base_F * 8
        nA = len(A)
        r1s = np.asarray(shear["r1s"], dtype=np.float64) if nA else None
        r2s = np.asarray(shear["r2s"], dtype=np.float64) if nA else None
        launches = []  # callables (last) -> None, in launch order

        def book():
            for e in plan:
                e["kopt"] = None
            for ks, out in kopt_rows:
                for k, row in zip(ks.tolist(), out):
                    plan[k]["kopt"] = row

        if deferred is not None:  # queued before on_chunk enqueues anything: it runs when the caller drains the list
            deferred.append(book)
            if more_deferred is not None:
                deferred.append(more_deferred)

        def add_launch(F, build, ks, row_A, r1, r2):
            def go(last):
                d_batch, own = build()
                try:
                    optimise(F, d_batch, ks, row_A, r1, r2)
                finally:
                    if own:
                        d_batch.free()
                report(ks, last)

            launches.append(go)

        merged = False
        for F, (d_hist, members) in hists_by_F.items():
            mem = np.asarray(members, dtype=np.int64)
            pos_C = np.nonzero(branch[mem] == 2)[0]
            if F == base_F and len(pos_C):
                cuts = [0, len(pos_C)]
                if first_fraction and len(pos_C) >= self.KOPT_SPLIT_MIN:
                    cuts = [0, int(len(pos_C) * first_fraction), len(pos_C)]
                for part in range(len(cuts) - 1):
                    pc = pos_C[cuts[part]:cuts[part + 1]]
                    na = nA if part == 0 else 0  # the sheared pairs ride with the first part

                    def build(pc=pc, na=na, d_hist=d_hist, whole=len(mem)):
                        if na == 0 and len(pc) == whole:
                            return d_hist, False  # the class's buffer as it is
                        d_all = ctx.alloc((na + len(pc)) * item)
                        if na:
                            ctx.gather_items(d_all, shear["d_rot"], np.arange(na, dtype=np.int32), item)
                        ctx.gather_items(d_all, d_hist, pc, item, dst_offset=na)
                        return d_all, True

                    add_launch(F, build, np.concatenate([A[:na], mem[pc]]), np.arange(na + len(pc)) < na, r1s if na else None,
                               r2s if na else None)
                merged = True
                continue
            if not len(pos_C):
                continue

            def build(pos_C=pos_C, mem=mem, d_hist=d_hist, F=F):
                if len(pos_C) == len(mem):
                    return d_hist, False
                d_sub = ctx.alloc(len(pos_C) * F * F * 8)
                self._gather_device(d_hist, d_sub, pos_C, F * F * 8)
                return d_sub, True

            add_launch(F, build, mem[pos_C], np.zeros(len(pos_C), dtype=bool), None, None)
        if nA and not merged:
            add_launch(base_F, lambda: (shear["d_rot"], False), A, np.ones(nA, dtype=bool), r1s, r2s)
        try:
            for q, go in enumerate(launches):
                go(q == len(launches) - 1)
            if not launches:
                report(np.zeros(0, dtype=np.int64), True)
        finally:
            if shear is not None:
                shear["d_rot"].free()
        if deferred is None:
            book()
        return W

    def _gather_device(self, d_src, d_dst, positions, item_bytes, ctx=None):
        """Copy selected fixed-size items of one device buffer into another (one gather kernel)."""
        (ctx or self.ctx).gather_items(d_dst, d_src, positions, item_bytes)

    def _index_columns8(self, wanted):
        """Byte index columns of the 256-bin grid (wanted: j -> (binmin, width)) in ONE launch; False if any sample of
        any column falls outside the grid (then the u16 path, which marks such samples, must be used)."""
        todo = [j for j, bw in wanted.items() if self._idx_cols.get((j, 256, "u8"), (None, None))[1] != bw]
        if todo:
            bufs = []
            for j in todo:
                hit = self._idx_cols.get((j, 256, "u8"))
                bufs.append(hit[0] if hit is not None else self.ctx.alloc(self.numrows + 64))
            bad = self.ctx.prebin8_batch(todo, [wanted[j][0] for j in todo], [wanted[j][1] for j in todo], 256, bufs)
            for j, buf, nb in zip(todo, bufs, bad):
                self._idx_cols[(j, 256, "u8")] = (buf, wanted[j] if nb == 0 else None)
        return all(self._idx_cols[(j, 256, "u8")][1] == bw for j, bw in wanted.items())

    def _index_column(self, j, F, binmin, width):
        key = (j, F)
        hit = self._idx_cols.get(key)
        if hit is None or hit[1] != (binmin, width):
            buf = hit[0] if hit is not None else None
            buf = self.ctx.prebin(j, binmin, width, F, buf)
            self._idx_cols[key] = (buf, (binmin, width))
        return self._idx_cols[key][0]

    def get2DDensities(self, pairs, num_plot_contours=None, get_density=True, _bandwidths=None, meanlikes=False,
                       mask_function=None, **kwargs):
        """
        Batched 2D KDEs (additive API): a list of Density2D, one per (x, y) entry of ``pairs``.
        ``mask_function(minx, miny, stepx, stepy, mask)`` may zero parts of each pair's prior mask in place
        (mcsamples.py:1767-1770); those pairs take the explicit-mask entry point one at a time.
        With ``meanlikes`` each result carries the mean-likelihood grid ``likes`` (mcsamples.py:1829-1831,1886-1903).
        Each result carries ``bandwidth`` = (hx, hy, corr) in parameter units, ``bandwidth_branch`` and
        ``kopt`` (the device optimiser's {t*, psi_02, psi_20, psi_11, psi_00, psi_13, psi_31, status}).
        ``_bandwidths`` (tests only) injects the (hx, hy, corr) triples instead of optimising.

        One implementation: every branch goes through gd_density2d_batch (getdist_amd/batch2d.py, csrc/batch2d.hpp).
        """
        if self.needs_update:
            self.updateBaseStatistics()
        for k in kwargs:
            if k not in ("fine_bins_2D", "boundary_correction_order", "mult_bias_correction_order", "smooth_scale_2D"):
                raise SettingError("unknown 2D density argument %s" % k)
        pa = None
        if len(pairs) > 8:
            try:  # a triangle's worth of integer indices: one conversion instead of a name lookup per entry
                pa = np.asarray(pairs)
                if pa.ndim != 2 or pa.shape[1] != 2 or pa.dtype.kind != "i" or pa.min() < 0 or pa.max() >= self.n:
                    pa = None
            except (ValueError, TypeError):
                pa = None
        pairs = pa.astype(np.int64, copy=False) if pa is not None else [(self._col(a), self._col(b)) for a, b in pairs]
        if hasattr(self.ctx, "density2d_batch") and os.environ.get("GETDIST_AMD_NATIVE_BATCH", "1") == "1":
            # ONE native call: every decision between the kernels is taken inside the library (csrc/batch2d.hpp), for the
            # optional branches too -- injected bandwidths and the 2D effective sample numbers are inputs of the call, the
            # mean-likelihood grids are a pass of their own over its per-pair table, and a mask callback gets the call's
            # bandwidths (bandwidths_only) and then the explicit-mask entry point pair by pair.
            from . import batch2d

            base_F = kwargs.get("fine_bins_2D", self.fine_bins_2D)
            bco = kwargs.get("boundary_correction_order", self.boundary_correction_order)
            mbc = kwargs.get("mult_bias_correction_order", self.mult_bias_correction_order)
            ss = float(kwargs.get("smooth_scale_2D", self.smooth_scale_2D))
            if abs(self.max_corr_2D) > 1:
                raise SettingError("max_corr_2D cannot be >=1")
            if bco > 1:
                raise SettingError("unknown boundary_correction_order (expected 0 or 1)")
            pa64 = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
            pair_neff = None
            if self.use_effective_samples_2D and ss < 0 and _bandwidths is None and len(pa64):
                pair_neff = self._pair_neff_2d(pa64)
            if mask_function is not None and len(pa64):
                return self._densities_with_mask_callback(pa64, base_F, bco, mbc, ss, num_plot_contours, get_density, _bandwidths,
                                                          pair_neff, meanlikes, mask_function)
            out = batch2d.run(self, pa64, base_F, bco, mbc, ss, num_plot_contours, get_density, bandwidths=_bandwidths,
                              pair_neff=pair_neff)
            if meanlikes and len(pa64):
                self._attach_mean_likelihoods(out, pa64, mbc)
            return out
        # No native entry on this context: the product has no second orchestration of the kernels.  The Python-planned
        # pipeline of rounds 2-5 lives in tests/planned_route.py as the comparison the native route is held bit-equal to;
        # it registers itself here for the numpy context double of the CPU tests (and for GETDIST_AMD_NATIVE_BATCH=0).
        route = getattr(MCSamples, "_planned_route", None)
        if route is None:
            raise MCSamplesError("get2DDensities needs a device context with gd_density2d_batch (libgdhip)")
        with _FastThreadSwitch(len(pairs) >= 64):
            return route(self, pairs, num_plot_contours, get_density, _bandwidths, meanlikes, mask_function=mask_function,
                         **kwargs)

    # ---- the optional branches of the native route -------------------------------------------------------------------
    def _pair_neff_2d(self, pa):
        """use_effective_samples_2D (mcsamples.py:1322-1328): the 2D estimate per pair, the smaller 1D one for a pair that is
        correlated to 0.999."""
        used = list(dict.fromkeys(pa.ravel().tolist()))
        self._init_params(used)
        names = self.paramNames.names
        corr = np.asarray(self.getCorrelationMatrix())[pa[:, 1], pa[:, 0]]
        return np.array([self.getEffectiveSamplesGaussianKDE_2d(a, b) if abs(c) < 0.999
                         else min(self._get1DNeff(names[a], a), self._get1DNeff(names[b], b))
                         for (a, b), c in zip(pa.tolist(), corr.tolist())], dtype=np.float64)

    def _pair_flags(self, pa, with_prior_mask=False):
        """Flag bits per pair (mcsamples.py:1688-1703, 1794): 0/1 = x bot/top, 2/3 = y bot/top, 4/5 = x/y periodic, 6 = has_prior."""
        names = self.paramNames.names
        lim_bits, per_bit, has_lim = np.zeros(self.n, np.int64), np.zeros(self.n, np.int64), np.zeros(self.n, bool)
        for j in np.unique(pa).tolist():
            p_ = names[j]
            lim_bits[j] = 0 if p_.periodic else (1 if p_.has_limits_bot else 0) | (2 if p_.has_limits_top else 0)
            per_bit[j] = 1 if p_.periodic else 0
            has_lim[j] = bool(p_.has_limits)
        jx, jy = pa[:, 0], pa[:, 1]
        has_prior = has_lim[jx] | has_lim[jy] | bool(with_prior_mask)
        return lim_bits[jx] | (per_bit[jx] << 4) | (lim_bits[jy] << 2) | (per_bit[jy] << 5) | (has_prior.astype(np.int64) << 6)

    def _class_histograms(self, pa, members, F, meta, likes=False):
        """Histograms (and, with ``likes``, the like-weighted ones: mcsamples.py:1829-1831) of the pairs ``members`` of one
        grid-size class from the bin edges the native call used (meta[23..26])."""
        ctx = self.ctx
        ix = [self._index_column(int(pa[k, 0]), F, meta[k, 23], (meta[k, 24] - meta[k, 23]) / (F - 1)) for k in members]
        iy = [self._index_column(int(pa[k, 1]), F, meta[k, 25], (meta[k, 26] - meta[k, 25]) / (F - 1)) for k in members]
        d_hist = ctx.hist2d_prebinned(ix, iy, F)
        d_like = self._like_histograms(0, lambda: ctx.hist2d_prebinned(ix, iy, F)) if likes else None
        return d_hist, d_like

    def _attach_mean_likelihoods(self, out, pa, mbc):
        """``likes`` of every result of a native call (mcsamples.py:1829-1831, 1886-1903): the like-weighted histogram of
        each pair convolved with the pair's own window (the call's per-pair table holds its scales), grid class by grid
        class."""
        ctx = self.ctx
        meta, F_v = out._meta, np.asarray(out._F)
        flags = self._pair_flags(pa)
        for F, periodic_bits in sorted(set(zip(F_v.tolist(), (flags & 48).tolist()))):  # (a launch holds one kind of axes)
            members = np.nonzero((F_v == F) & ((flags & 48) == periodic_bits))[0]
            step = max(1, min(int(24e9 // (F * F * 8 * 30)), 320))
            for s0 in range(0, len(members), step):
                mem = members[s0:s0 + step]
                d_hist, d_like = self._class_histograms(pa, mem.tolist(), F, meta, likes=True)
                d_L, lstatus = ctx.likes2d(d_hist, d_like, len(mem), F, meta[mem, 18], meta[mem, 19], meta[mem, 20],
                                           meta[mem, 21].astype(np.int64), flags[mem], mbc)
                d_hist.free()
                d_like.free()
                if np.any(lstatus != 0):
                    d_L.free()
                    raise DensitiesError("no likelihood weight in any bin")
                L = d_L.to_host((len(mem), F, F))
                d_L.free()
                for row, k in enumerate(mem.tolist()):
                    out[k].likes = L[row]

    def _densities_with_mask_callback(self, pa, base_F, bco, mbc, ss, num_plot_contours, get_density, bandwidths, pair_neff,
                                      meanlikes, mask_function):
        """get2DDensities with ``mask_function`` (mcsamples.py:1767-1770, 1905-1919, 1973-1979): the callback edits each pair's
        prior mask on the padded frame, whose size follows from the pair's window -- so the native call runs up to the
        bandwidths (bandwidths_only) and every pair then goes through the explicit-mask entry point."""
        from . import batch2d

        ctx = self.ctx
        names = self.paramNames.names
        meta, F_v = batch2d.run(self, pa, base_F, bco, mbc, ss, None, True, bandwidths=bandwidths, pair_neff=pair_neff,
                                bandwidths_only=True)
        F_v = np.asarray(F_v)
        flags = self._pair_flags(pa, with_prior_mask=True)
        ncontours = len(self.contours)
        if num_plot_contours:
            ncontours = min(num_plot_contours, ncontours)
        out = [None] * len(pa)
        for F in np.unique(F_v).tolist():
            members = np.nonzero(F_v == F)[0].tolist()
            d_hist, d_like = self._class_histograms(pa, members, F, meta, likes=meanlikes)
            try:
                for pos, k in enumerate(members):
                    j, j2 = int(pa[k, 0]), int(pa[k, 1])
                    parx, pary = names[j], names[j2]
                    w_ = int(meta[k, 21])
                    fwx, fwy = (meta[k, 24] - meta[k, 23]) / (F - 1), (meta[k, 26] - meta[k, 25]) / (F - 1)
                    prior_mask = np.ones((F + 2 * w_, F + 2 * w_))
                    mask_function(meta[k, 23] - w_ * fwx, meta[k, 25] - w_ * fwy, fwx, fwy, prior_mask)
                    bool_mask = prior_mask[w_:-w_, w_:-w_] < 1e-8
                    mask_bc = mask_mbc = None
                    if bco >= 0:
                        _set_edge_mask_2d(parx, pary, prior_mask, w_)
                        mask_bc = prior_mask.copy()
                    if mbc:
                        _set_all_edge_mask_2d(prior_mask, w_, parx.periodic, pary.periodic)
                        mask_mbc = prior_mask
                    d_P, status = ctx.density2d_masked(d_hist, pos, F, float(meta[k, 18]), float(meta[k, 19]), float(meta[k, 20]),
                                                       w_, int(flags[k]), bco, mbc, mask_bc, mask_mbc, bool_mask)
                    contours = None
                    if not get_density:
                        lev, lev_state = ctx.contour_levels(d_P, 1, F, self.contours[:ncontours])
                        state = int(np.asarray(lev_state)[0])
                        if state == -4:
                            raise DensitiesError("Contour level outside plotted ranges")
                        contours = lev[0].copy() if state == 0 else None
                    L = None
                    if meanlikes:
                        # the mean-likelihood grid does not see the mask (mcsamples.py:1886-1903 precede it)
                        d_one, d_lone = ctx.alloc(F * F * 8), ctx.alloc(F * F * 8)
                        self._gather_device(d_hist, d_one, [pos], F * F * 8)
                        self._gather_device(d_like, d_lone, [pos], F * F * 8)
                        d_L, lstatus = ctx.likes2d(d_one, d_lone, 1, F, meta[[k], 18], meta[[k], 19], meta[[k], 20],
                                                   meta[[k], 21].astype(np.int64), flags[[k]], mbc)
                        d_one.free()
                        d_lone.free()
                        if np.any(lstatus != 0):
                            raise DensitiesError("no likelihood weight in any bin")
                        L = d_L.to_host((1, F, F))[0]
                        d_L.free()
                    P = d_P.to_host((1, F, F))[0]
                    d_P.free()
                    if np.any(np.asarray(status) != 0):
                        raise DensitiesError("no samples in bin")
                    ax = np.arange(F, dtype=np.float64) * fwx + meta[k, 23]
                    ay = np.arange(F, dtype=np.float64) * fwy + meta[k, 25]
                    ax[-1], ay[-1] = meta[k, 24], meta[k, 26]
                    auto = ss < 0
                    dens = Density2D._from_fields(dict(
                        x=ax, y=ay, axes=[ay, ax], spacing=(ax[1] - ax[0]) * (ay[1] - ay[0]),
                        view_ranges=[(parx.range_min, parx.range_max), (pary.range_min, pary.range_max)], mask=bool_mask, likes=L,
                        contours=contours, spl=None, _P=P, _wait=None,
                        bandwidth=tuple(meta[k, 2:5].tolist()) if auto else None,
                        bandwidth_branch="ABC"[int(meta[k, 5])] if auto and meta[k, 5] >= 0 else None,
                        kopt=None if np.isnan(meta[k, 13]) else meta[k, 6:18].copy()))
                    if contours is None and not get_density:
                        dens.contours = dens.getContourLevels(self.contours[:ncontours])
                    out[k] = dens
            finally:
                d_hist.free()
                if d_like is not None:
                    d_like.free()
        return out

    # ---- convergence (chains.py:1446-1527; mcsamples.py:964-1003) ------------------------------------------
    def getSeparateChainStats(self, nparam=None):
        """Per-chain (means, cov, norm) over the first nparam parameters: one covariance launch per chain over ALL
        columns, cached until the samples change, so Gelman-Rubin, MeanVar, CorrLengths and CorrSteps share one pass."""
        if self.chain_offsets is None:
            raise WeightedSampleError("Samples were not combined from separate chains")
        nparam = nparam or self.paramNames.numNonDerived()
        if "all" not in self._chain_stats_cache:
            cols = list(range(self.n))
            self._chain_stats_cache["all"] = [self.ctx.cov(cols, lo=int(a), hi=int(b))
                                              for a, b in zip(self.chain_offsets[:-1], self.chain_offsets[1:])]
        return [(m[:nparam], c[:nparam, :nparam], nrm) for m, c, nrm in self._chain_stats_cache["all"]]

    def getSeparateChains(self):
        """
        chains.py:1505-1527: one object per chain.  The reference slices the host arrays into WeightedSamples; here each
        is a ChainView -- a row range [lo, hi) of the resident device columns with the WeightedSamples statistics API
        (getMeans / getVars / getCov / mean / var / std / cov / corr / confidence / twoTailLimits / norm), no copy.
        """
        if self.chain_offsets is None:
            raise WeightedSampleError("Samples were not combined from separate chains")
        return [ChainView(self, int(a), int(b)) for a, b in zip(self.chain_offsets[:-1], self.chain_offsets[1:])]

    def makeSingle(self):
        """chains.py:1488-1503.  The constructor already stacks a list of chains into one resident array (recording
        chain_offsets), after which the reference's ``chains`` attribute is None and this call raises there too."""
        if not self.chains:
            raise ValueError("There are no separated chains for makeSingle()")
        return self

    def getGelmanRubinEigenvalues(self, nparam=None, chainlist=None):
        """chains.py:1446-1474: var(mean)/mean(var) in the orthogonalised parameters; ``chainlist`` may be any
        sub-list of getSeparateChains() (or objects with getMeans() / getCov(nparam))."""
        from .parallel import gelman_rubin_from_chain_stats

        nparam = nparam or self.paramNames.numNonDerived()
        if chainlist is None:
            stats = self.getSeparateChainStats(nparam)
        else:
            stats = [(np.asarray(ch.getMeans())[:nparam], np.asarray(ch.getCov(nparam)), None) for ch in chainlist]
        return gelman_rubin_from_chain_stats(stats, self.getMeans())

    def getGelmanRubin(self, nparam=None, chainlist=None):
        return np.max(self.getGelmanRubinEigenvalues(nparam, chainlist))

    def getMeanVarTest(self, nparam=None):
        """The MeanVar block of getConvergeTests (mcsamples.py:964-985): sqrt(var(chain mean)/mean(chain var))."""
        nparam = nparam or self.n
        stats = self.getSeparateChainStats(nparam)
        between = np.zeros(nparam)
        within = np.zeros(nparam)
        for cmeans, ccov, cnorm in stats:
            between += (cmeans - self.means[:nparam]) ** 2
            within += np.diag(ccov) * cnorm
        between /= len(stats) - 1
        within /= self.norm
        return np.sqrt(between / within)


class ChainView:
    """
    One chain of a combined sample set: rows [lo, hi) of the parent's device-resident columns, with the statistics
    interface of chains.WeightedSamples (chains.py:339-412, 636-838) evaluated on that row range by the same kernels
    (every entry point of the C ABI takes a row range).  ``samples`` / ``weights`` / ``loglikes`` are host views.
    """

    def __init__(self, parent, lo, hi):
        self.parent, self.lo, self.hi = parent, lo, hi
        self.numrows = hi - lo
        self.n = parent.n
        self.paramNames = parent.paramNames
        self._stats = self._cov = self._ws = None

    samples = property(lambda self: self.parent.samples[self.lo:self.hi])
    weights = property(lambda self: (np.ones(self.numrows) if self.parent.weights is None
                                     else self.parent.weights[self.lo:self.hi]))
    loglikes = property(lambda self: None if self.parent.loglikes is None else self.parent.loglikes[self.lo:self.hi])

    def _weight_stats(self):
        if self._ws is None:
            self._ws = self.parent.ctx.weight_stats(self.lo, self.hi)
        return self._ws

    @property
    def norm(self):
        return self._weight_stats()["norm"] if self.parent.weights is not None else np.float64(self.numrows)

    def _where_global(self, where):
        """A chain-relative ``where`` (boolean mask or row indices of THIS chain) as full-length weights*mask of the
        parent: the kernels then see x[where], w[where] of the chain inside its row range."""
        where = np.asarray(where)
        p = self.parent
        w = np.zeros(p.numrows)
        base = p.weights[self.lo:self.hi] if p.weights is not None else np.ones(self.numrows)
        if where.dtype == bool:
            if where.shape != (self.numrows,):
                raise WeightedSampleError("where must have one entry per sample of the chain")
            w[self.lo:self.hi] = base * where
        else:
            w[self.lo:self.hi] = base * np.bincount(_where_rows(where, self.numrows), minlength=self.numrows)
        return w

    def _moments(self, pars, where):
        p = self.parent
        cols = [p._col(q) for q in pars]
        return p._with_weights(self._where_global(where), lambda: p.ctx.cov(cols, lo=self.lo, hi=self.hi))

    def get_norm(self, where=None):
        if where is not None:
            p = self.parent
            return p._with_weights(self._where_global(where), lambda: p.ctx.weight_stats(self.lo, self.hi)["norm"])
        return self.norm

    def _col_stats(self):
        if self._stats is None:
            self._stats = self.parent.ctx.col_stats(self.lo, self.hi)
        return self._stats

    def getMeans(self, pars=None):
        means = self._col_stats()[:, 2]
        return means.copy() if pars is None else np.array([means[self.parent._col(p)] for p in pars])

    def getVars(self):
        return self._col_stats()[:, 3].copy()

    def getCov(self, nparam=None, pars=None):
        if self._cov is None:
            self._cov = self.parent.ctx.cov(list(range(self.n)), lo=self.lo, hi=self.hi)[1]
        if pars is not None:
            return self._cov[np.ix_(pars, pars)]
        return self._cov[:nparam, :nparam]

    def getCorrelationMatrix(self):
        return covToCorr(self.getCov())

    def cov(self, pars=None, where=None):
        if isinstance(pars, (int, np.integer)):
            pars = range(pars)
        cols = list(range(self.n)) if pars is None else [self.parent._col(p) for p in pars]
        if where is not None:
            return self._moments(cols, where)[1]
        return self.parent.ctx.cov(cols, lo=self.lo, hi=self.hi)[1]

    def corr(self, pars=None):
        return covToCorr(self.cov(pars))

    def mean(self, paramVec, where=None):
        if isinstance(paramVec, (list, tuple)):
            return np.array([self.mean(p, where) for p in paramVec])
        if where is not None:
            return self._moments([paramVec], where)[0][0]
        return self._col_stats()[self.parent._col(paramVec), 2]

    def var(self, paramVec, where=None):
        if isinstance(paramVec, (list, tuple)):
            return np.array([self.var(p) for p in paramVec])  # like the reference, a list ignores ``where``
        if where is not None:
            return self._moments([paramVec], where)[1][0, 0]
        return self._col_stats()[self.parent._col(paramVec), 3]

    def std(self, paramVec, where=None):
        return np.sqrt(self.var(paramVec, where))

    def confidence(self, paramVec, limfrac, upper=False):
        return self.parent.confidence(paramVec, limfrac, upper, start=self.lo, end=self.hi)

    def twoTailLimits(self, paramVec, confidence):
        limits = np.array([(1 - confidence) / 2, 1 - (1 - confidence) / 2])
        return self.confidence(paramVec, limits)


_FFT_SIZE_CACHE = {}


def next_fft_size(n):
    """Smallest 2^a * {1,3,5,9,15} (a >= 4) >= n: the FFT frame ladder density2d.hip plans for."""
    if n in _FFT_SIZE_CACHE:
        return _FFT_SIZE_CACHE[n]
    _FFT_SIZE_CACHE[n] = v = _next_fft_size(n)
    return v


def _next_fft_size(n):
    best = None
    for a in range(4, 28):
        for odd in (1, 3, 5, 9, 15):
            v = (1 << a) * odd
            if v >= n and (best is None or v < best):
                best = v
    return best


def covToCorr(cov, copy=True):
    """chains.py:155-169: for i in order, row i and then column i are divided by sqrt(cov[i, i]) -- so element (a, b) is
    divided by the standard deviation of min(a, b) FIRST and by that of max(a, b) second (zero deviations are skipped).
    The same two divisions per element, on the whole matrix at once."""
    cov = np.array(cov, dtype=np.float64) if copy else cov
    d = np.sqrt(cov.diagonal())
    d = np.where(d != 0, d, 1.0)
    idx = np.arange(len(d))
    first, second = np.minimum(idx[:, None], idx[None, :]), np.maximum(idx[:, None], idx[None, :])
    cov[...] = (cov / d[first]) / d[second]
    return cov
