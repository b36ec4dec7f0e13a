"""A/B of the two fixed-point kernels of gd_kopt2d on C3-like histograms (F = 256): k_kopt2d_res (matrix resident on the
CU) against k_kopt2d (matrix streamed through LDS, GDHIP_KOPT_STREAMED=1): result rows side by side and wall time of the
whole call.  Prints one JSON object."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from getdist_amd import synth
from getdist_amd._lib import Context


def main():
    N, n, F = 2_000_000, 16, 256
    samples = synth.block_recipe(n, N)[0]
    ctx = Context(0)
    ctx.upload(samples, None)
    pairs = [(i, j) for i in range(n) for j in range(i)] * 10
    mn, mx = samples.min(0), samples.max(0)
    ix = [ctx.prebin(j, mn[j], (mx[j] - mn[j]) / (F - 1), F) for j in range(n)]
    hp = ctx.hist2d_prebinned([ix[a] for a, b in pairs], [ix[b] for a, b in pairs], F)
    B = len(pairs)
    rng = np.random.default_rng(5)
    neff = list(10.0 ** rng.uniform(3.5, 6.5, B))
    res = {"pairs": B, "F": F}
    rows = {}
    for dc in (0, 1):
        for mode in ("resident", "streamed"):
            if mode == "streamed":
                os.environ["GDHIP_KOPT_STREAMED"] = "1"
            else:
                os.environ.pop("GDHIP_KOPT_STREAMED", None)
            ts = []
            for r in range(5):
                t0 = time.perf_counter()
                out = ctx.kopt2d(hp, B, F, neff, [dc] * B, [1e-4] * B, [0.3] * B)
                ts.append(time.perf_counter() - t0)
            rows[(dc, mode)] = np.array(out, copy=True)
            res["ms_whole_call_do_corr%d_%s" % (dc, mode)] = round(1e3 * min(ts[1:]), 3)
        a, b = rows[(dc, "resident")], rows[(dc, "streamed")]
        ncmp = 7  # t*, psi_02, psi_20, psi_11, psi_00, psi_13, psi_31
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = np.abs(a[:, :ncmp] - b[:, :ncmp]) / np.abs(b[:, :ncmp])
        same_nan = np.array_equal(np.isnan(a[:, :ncmp]), np.isnan(b[:, :ncmp]))
        res["do_corr%d" % dc] = dict(max_rel_diff_tstar=float(np.nanmax(rel[:, 0])), max_rel_diff_psi=float(np.nanmax(rel[:, 1:])),
                                     nan_pattern_equal=bool(same_nan), status_equal=bool(np.array_equal(a[:, 7], b[:, 7])),
                                     max_rel_diff_h=float(np.nanmax(np.abs(a[:, 8:11] - b[:, 8:11]) / np.abs(b[:, 8:11]))),
                                     tstar_first=[float(a[0, 0]), float(b[0, 0])])
    print(json.dumps(res))


main()
