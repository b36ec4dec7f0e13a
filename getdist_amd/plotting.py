"""
Glue for GetDist's plotting layer.  ``plots.MCSampleAnalysis`` keeps per-root density caches,
``densities_1D[root][(name, likes)]`` and ``densities_2D[root][(xname, yname, likes, conts)]`` (plots.py:594-645), and
computes a missing entry with one ``get1DDensityGridData`` / ``get2DDensityGridData`` call.  A triangle plot therefore
issues n + n(n-1)/2 separate calls; filling the caches from the batched device path first turns that into two.
"""


def prefill_plot_caches(analysis, root, samples, params=None, conts=2, likes=False, lower_triangle=True):
    """
    Fill ``analysis.densities_1D[root]`` and ``analysis.densities_2D[root]`` (any object with those two dicts, e.g.
    getdist.plots.MCSampleAnalysis) for the parameters ``params`` of the getdist_amd ``samples``:
    every 1D density in one batched call and every pair the triangle plot will ask for (x = params[i],
    y = params[i2 > i]; plots.py:2845-2878) in another, contour levels included (``conts`` as in
    ``get_density_grid``).  Returns (n_1d, n_2d) = the number of cache entries written.
    """
    names = samples.paramNames.list() if params is None else [samples.paramNames.names[samples._col(p)].name for p in params]
    cols = [samples._col(nm) for nm in names]
    d1 = analysis.densities_1D.setdefault(root, {})
    for nm, dens in zip(names, samples.get1DDensities(cols, meanlikes=likes)):
        d1.pop((nm, not likes), None)
        d1[(nm, likes)] = dens
    pairs = [(i, i2) for i in range(len(cols)) for i2 in range(i + 1, len(cols))]
    if not lower_triangle:
        pairs += [(i2, i) for i, i2 in pairs]
    d2 = analysis.densities_2D.setdefault(root, {})
    grids = samples.get2DDensities([(cols[i], cols[i2]) for i, i2 in pairs], num_plot_contours=conts, get_density=False,
                                   meanlikes=likes) if pairs else []
    for (i, i2), dens in zip(pairs, grids):
        d2[(names[i], names[i2], likes, conts)] = dens
    return len(names), len(pairs)
