"""
-m gpu: the sample-set mutators, getLikeStats, the FFT autocorrelation route and the public functions of
getdist_amd.convolve on the HIP path, against the reference's own outputs (tests/golden/mutators.npz).
"""

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


def test_mutators_like_stats_autocorrelation_and_convolve_on_the_device():
    gu.mutator_checks(None)


def test_autoconvolve_fft_route_equals_direct_lag_sums_at_full_size():
    """SURVEY.md A.9 at N = 4e6: the length-2N FFT (gd_autoconvolve) and the direct lag kernel give the same lag sums,
    weighted and unit; getCorrelationLength is the same through either route."""
    from getdist_amd.convolve import nearestFFTnumber
    from getdist_amd.mcsamples import MCSamples

    rng = np.random.default_rng(3)
    N = 4_000_000
    e = rng.standard_normal(N)
    x = np.empty(N)
    x[0] = e[0]
    rho = 0.9
    # AR(1) by blocks (vectorised): x_t = rho x_{t-1} + e_t
    from scipy.signal import lfilter

    x = lfilter([1.0], [1.0, -rho], e)
    for w in (None, rng.integers(1, 5, N).astype(np.float64)):
        mc = MCSamples(samples=x.reshape(-1, 1), weights=w, names=["x"])
        s = int(nearestFFTnumber(2 * N))
        direct = mc.ctx.autocov_lags(0, mc.means[0], 0, 256)
        fft = mc.ctx.autoconvolve(s, 256, False, col=0, mean=mc.means[0], use_weights=w is not None)
        assert np.max(np.abs(fft - direct)) <= 1e-9 * abs(direct[0])
        a = mc.getCorrelationLength(0)
        mc.DIRECT_LAGS_MAX = 8  # force the FFT route
        b = mc.getCorrelationLength(0)
        assert abs(a - b) < 1e-8 * a
        mc.ctx.close()


def test_prefill_plot_caches_on_the_device(zoo):
    """SURVEY.md 8f rank 3 on the HIP path: the plot-cache glue over the batched calls and gd_contour_levels."""
    gu.prefill_plot_caches_checks(zoo, None)


def test_triangle_plot_levels_on_the_device(zoo):
    """The HIP path's plot caches against the record of GetDist's real plotter (tests/golden/triangle_plot_levels.npz,
    scripts/drive_real_caller.py): contour levels drawn, grids, 1D curves, axis limits."""
    gu.triangle_plot_golden_checks(zoo, None)


def test_root_constructor_and_binary_cache_on_the_device(tmp_path):
    """SURVEY.md 8f rank 4 on the HIP path: text chains -> MCSamples(root=), the .gdamd_soa cache read into page-locked
    memory and uploaded from there, reload equality, statistics from the reloaded object."""
    gu.root_constructor_checks(tmp_path, None)
