"""
Drives the REAL caller of the path -- getdist.plots.GetDistPlotter.triangle_plot (plots.py:2845-2878) through
plots.MCSampleAnalysis.get_density / get_density_grid (plots.py:594-645) -- with a getdist_amd sample set whose plot caches were
filled by the batched entry (getdist_amd.plotting.prefill_plot_caches), and with the per-parameter / per-pair methods patched
to RAISE: the whole figure must be drawn from the two batched calls.  Build container only (imports the reference from
/root/reference; never shipped); the sample set runs on the numpy context double (tests/fake_ctx.py), so no GPU is needed.

    python scripts/drive_real_caller.py            # check against tests/golden/triangle_plot_levels.npz
    python scripts/drive_real_caller.py --write    # (re)write that golden file

What is stored: for every 2D panel the contour levels matplotlib drew (QuadContourSet.levels: what get_density_grid's
Density2D.contours delivered), the axis limits the plotter chose, and for every 1D panel the curve it drew; the GPU test
(tests/test_gpu_mutators.py::test_triangle_plot_levels_on_the_device) compares the HIP path's caches with the same file.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REFERENCE = os.environ.get("GETDIST_REFERENCE", "/root/reference")
# GetDist "installed beside" the package: getdist_amd's parameter objects then derive from getdist.paramnames.ParamInfo, which
# is how the plotting layer recognises them (must be importable BEFORE getdist_amd is imported)
sys.path.insert(0, REFERENCE)
GOLD = os.path.join(ROOT, "tests", "golden", "triangle_plot_levels.npz")
FIXTURE, NPAR, ROOTNAME = "c1_bounded", 4, "amd_chain"


def make_samples(context_factory=None):
    from getdist_amd.mcsamples import MCSamples
    from oracle.fixtures import fixture_zoo

    fx = {f["name"]: f for f in fixture_zoo()}[FIXTURE]
    kw = {} if context_factory is None else dict(_context_factory=context_factory)
    mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"], label="getdist_amd", **kw)
    return mc, fx["names"][:NPAR]


def draw(mc, params):
    """The reference's triangle plot of ``mc`` from prefilled caches; returns {key: array} of what was drawn."""
    import matplotlib

    matplotlib.use("Agg")
    from getdist import plots
    from matplotlib.contour import QuadContourSet

    from getdist_amd.plotting import prefill_plot_caches

    g = plots.get_subplot_plotter()
    g.sample_analyser.mcsamples[ROOTNAME] = mc  # what samples_for_root(ROOTNAME) returns (plots.py:517-518)
    n1, n2 = prefill_plot_caches(g.sample_analyser, ROOTNAME, mc, params=params, conts=g.settings.num_plot_contours)
    assert (n1, n2) == (len(params), len(params) * (len(params) - 1) // 2)
    calls = []

    def refuse(*a, **k):
        calls.append(a)
        raise AssertionError("a per-parameter / per-pair density call was issued: the caches were not used")

    mc.get1DDensityGridData = mc.get2DDensityGridData = mc.get1DDensity = mc.get2DDensity = refuse
    g.triangle_plot([ROOTNAME], params, filled=True)
    assert not calls
    out = {}
    n = len(params)
    for i in range(n):
        ax = g.subplots[i, i]
        line = ax.get_lines()[0]
        out["1d/%s/x" % params[i]], out["1d/%s/y" % params[i]] = np.asarray(line.get_xdata(), float), np.asarray(line.get_ydata(), float)
        out["1d/%s/xlim" % params[i]] = np.asarray(ax.get_xlim(), float)
        for i2 in range(i + 1, n):
            ax = g.subplots[i2, i]
            sets = [c for c in ax.get_children() if isinstance(c, QuadContourSet)]
            assert sets, (params[i], params[i2])
            levels = np.unique(np.concatenate([np.asarray(c.levels, float) for c in sets]))
            key = "2d/%s/%s" % (params[i], params[i2])
            out[key + "/levels"] = levels
            out[key + "/lims"] = np.asarray(list(ax.get_xlim()) + list(ax.get_ylim()), float)
    # the cache entries the plotter read, for the device comparison
    for (a, b, likes, conts), d in g.sample_analyser.densities_2D[ROOTNAME].items():
        out["cache2d/%s/%s/contours" % (a, b)] = np.asarray(d.contours, float)
        out["cache2d/%s/%s/P16" % (a, b)] = np.asarray(d.P[::16, ::16], float)
    for (a, likes), d in g.sample_analyser.densities_1D[ROOTNAME].items():
        out["cache1d/%s/P8" % a] = np.asarray(d.P[::8], float)
    import matplotlib.pyplot as plt

    plt.close("all")
    return out


def draw_reference(params):
    """The same figure from the reference's own MCSamples on the same arrays (its per-pair calls, its numpy / scipy path)."""
    import matplotlib

    matplotlib.use("Agg")
    from getdist import MCSamples as RefSamples
    from getdist import plots
    from matplotlib.contour import QuadContourSet
    from oracle.fixtures import fixture_zoo

    fx = {f["name"]: f for f in fixture_zoo()}[FIXTURE]
    ref = RefSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"], label="reference")
    g = plots.get_subplot_plotter()
    g.triangle_plot([ref], params, filled=True)
    out = {}
    n = len(params)
    for i in range(n):
        line = g.subplots[i, i].get_lines()[0]
        out["1d/%s/y" % params[i]] = np.asarray(line.get_ydata(), float)
        for i2 in range(i + 1, n):
            ax = g.subplots[i2, i]
            sets = [c for c in ax.get_children() if isinstance(c, QuadContourSet)]
            out["2d/%s/%s/levels" % (params[i], params[i2])] = np.unique(np.concatenate([np.asarray(c.levels, float) for c in sets]))
            out["2d/%s/%s/lims" % (params[i], params[i2])] = np.asarray(list(ax.get_xlim()) + list(ax.get_ylim()), float)
    import matplotlib.pyplot as plt

    plt.close("all")
    return out


def main():
    import fake_ctx

    mc, params = make_samples(fake_ctx.FakeContext)
    got = draw(mc, params)
    # the figure the reference draws of the same samples by itself: same contour levels, axis limits and 1D curves
    ref = draw_reference(params)
    # (levels of a pair whose bandwidth goes through TNC follow the grid to ~1e-5: the numpy double is not the reference bit
    # for bit there; everything else to 1e-6)
    for k, v in ref.items():
        tol = 1e-4 if k.endswith("/levels") else 1e-6
        assert got[k].shape == v.shape and np.allclose(got[k], v, rtol=tol, atol=1e-9), (k, got[k], v)
    if "--write" in sys.argv:
        np.savez_compressed(GOLD, **got)
        print("wrote", GOLD, len(got), "arrays")
        return
    want = np.load(GOLD)
    assert sorted(want.files) == sorted(got), (sorted(set(want.files) ^ set(got)))
    for k in want.files:
        assert got[k].shape == want[k].shape and np.allclose(got[k], want[k], rtol=1e-9, atol=1e-12), k
    print("triangle plot drawn by getdist.plots from the prefilled caches: %d panels; levels / limits / curves equal the golden "
          "file and the figure the reference draws of the same samples by itself" % (len(params) * (len(params) + 1) // 2))


if __name__ == "__main__":
    main()
