"""
getdist_amd -- MI355X-native weighted-sample statistics and 1D/2D kernel density estimation behind the GetDist API.

    from getdist_amd import MCSamples
    s = MCSamples(samples=x, weights=w, names=names, ranges={...})
    s.get1DDensity("a"); s.get2DDensity("a", "b"); s.triangleDensities(); s.getMargeStats(); s.getGelmanRubin()

Everything with O(N) or O(F^2) data runs in hand-written HIP kernels (getdist_amd/csrc, C ABI in include/gdhip.h);
there is no CPU fallback.
"""

__version__ = "0.1.0"

_EXPORTS = {
    "MCSamples": "mcsamples", "MCSamplesError": "mcsamples", "SettingError": "mcsamples", "BandwidthError": "mcsamples",
    "ParamError": "mcsamples", "WeightedSampleError": "mcsamples", "MargeStats": "mcsamples", "ParamLimit": "mcsamples",
    "covToCorr": "mcsamples", "Density1D": "densities", "Density2D": "densities", "GridDensity": "densities",
    "DensitiesError": "densities", "getContourLevels": "densities", "nearestFFTnumber": "convolve",
    "loadMCSamples": "chainfiles", "chainFiles": "chainfiles", "prefill_plot_caches": "plotting",
}


def __getattr__(name):  # lazy: importing the package must not need the native library
    if name in _EXPORTS:
        import importlib

        return getattr(importlib.import_module("." + _EXPORTS[name], __name__), name)
    raise AttributeError(name)


__all__ = sorted(_EXPORTS)
