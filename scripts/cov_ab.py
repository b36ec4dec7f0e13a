"""A/B of the covariance entry on the MI355X: one pass (provisional shift + ones column, round 6) against the two passes
(means, then the slab kernel) -- time and agreement.  python scripts/cov_ab.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from getdist_amd._lib import Context  # noqa: E402


def timed(ctx, fn, reps=5):
    fn()
    ctx.sync()
    ms = []
    for _ in range(reps):
        ctx.timer_start()
        fn()
        ms.append(ctx.timer_stop_ms())
    return round(float(np.median(ms)), 4)


res = {}
for (N, m, weighted, offset) in ((10_000_000, 50, False, 0.0), (5_000_000, 100, True, 0.0), (2_000_000, 200, False, 0.0),
                                 (5_000_000, 64, True, 1e6), (3_000_000, 30, False, 1e4)):
    rng = np.random.default_rng(m)
    A = rng.standard_normal((m, m)) / np.sqrt(m) + 0.3 * np.eye(m)
    s = np.asfortranarray(rng.standard_normal((N, m)) @ A.T + offset + rng.standard_normal(m))
    w = rng.exponential(1.0, N) if weighted else None
    c = Context(0)
    c.upload(s, w)
    cols = list(range(m))
    tag = "N%d_m%d_%s_offset%g" % (N, m, "w" if weighted else "u", offset)
    one = c.cov(cols, minmax=True)
    t1 = timed(c, lambda: c.cov(cols, minmax=True))
    os.environ["GDHIP_COV_TWOPASS"] = "1"
    two = c.cov(cols, minmax=True)
    t2 = timed(c, lambda: c.cov(cols, minmax=True))
    os.environ.pop("GDHIP_COV_TWOPASS")
    wn = w if weighted else np.ones(N)
    mean = (wn[:, None] * s).sum(axis=0) / wn.sum()
    d = s - mean
    ref = (d * wn[:, None]).T @ d / wn.sum()
    scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
    res[tag] = dict(ms_one_pass=t1, ms_two_pass=t2,
                    cov_one_vs_numpy=float(np.max(np.abs(one[1] - ref) / scale)), cov_two_vs_numpy=float(np.max(np.abs(two[1] - ref) / scale)),
                    mean_one_vs_numpy=float(np.max(np.abs(one[0] - mean) / np.sqrt(np.diag(ref)))),
                    mean_two_vs_numpy=float(np.max(np.abs(two[0] - mean) / np.sqrt(np.diag(ref)))),
                    minmax_equal=bool(np.array_equal(one[3], two[3])), norm_equal=bool(one[2] == two[2]),
                    symmetric=bool(np.array_equal(one[1], one[1].T)))
    c.close()
print(json.dumps(res, indent=1))
