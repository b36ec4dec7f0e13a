"""Process-pool workers of the full-size parity tests (importable for the 'spawn' start method): one oracle task each."""
import os
import time

import numpy as np


def oracle_pair(task):
    """The oracle's 2D density of one pair at full size, from the two sample columns saved under /dev/shm."""
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    import logging
    import warnings

    from oracle import kde_oracle as ko

    warnings.simplefilter("ignore")
    logging.disable(logging.WARNING)
    cols = np.load(task["path"])
    t0 = time.perf_counter()
    orc = ko.OracleSamples(cols, names=task["names"], ranges=task["ranges"])
    tr = {}
    o = orc.density_2d(0, 1, trace=tr)
    out = dict(pair=task["pair"], P=o["P"], branch=tr.get("branch"), bw=(tr.get("hx"), tr.get("hy"), tr.get("c")),
               seconds=time.perf_counter() - t0, tnc="p_13" in tr)
    if out["tnc"]:
        psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr["p_00"], tr["p_13"], tr["p_31"])
        out["t_star"], out["psi"], out["opt_N"] = tr["t_star"], psi, tr["opt_N"]
        out["ensembles"] = ko.get_h_ensembles(psi, tr["opt_N"], tr["opt_corr"])  # one per scale of ko.ENSEMBLE_SCALES
        out["ensemble"] = out["ensembles"][0]
    return out
