"""The optimiser's DCT through LDS transforms (k_dct_pass) against the two-GEMM route it replaces: t*, the psi functionals
and the bandwidth triples of gd_kopt2d on smooth random histograms, every grid size of a triangle; timings of both."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from getdist_amd._lib import Context

ctx = Context(0)
rng = np.random.default_rng(3)
x = rng.normal(size=(200000, 2))
ctx.upload(x)
out = {}
for F, B in ((128, 40), (256, 600), (384, 13), (768, 6), (960, 6)):
    yy, xx = np.mgrid[0:F, 0:F] / F
    H = np.empty((B, F, F))
    for b in range(B):
        cx, cy, sx, sy, rho = rng.uniform(0.3, 0.7), rng.uniform(0.3, 0.7), rng.uniform(0.05, 0.15), rng.uniform(0.05, 0.15), rng.uniform(-0.6, 0.6)
        u, v = (xx - cx) / sx, (yy - cy) / sy
        lam = 4e5 * np.exp(-(u * u - 2 * rho * u * v + v * v) / (2 * (1 - rho * rho)))
        H[b] = rng.poisson(lam / lam.sum() * 1e6)
    d = ctx.alloc(H.nbytes)
    d.from_host(H)
    neff = np.full(B, 1e6)
    do_corr = (np.arange(B) % 2).astype(np.int32)
    fb = np.full(B, 1e-3)
    corr = rng.uniform(-0.5, 0.5, size=B)
    res = {}
    for route in ("fft", "gemm"):
        if route == "gemm":
            os.environ["GDHIP_KOPT_DCT_GEMM"] = "1"
        else:
            os.environ.pop("GDHIP_KOPT_DCT_GEMM", None)
        ctx.kopt2d(d, B, F, neff, do_corr, fb, corr)
        t0 = time.perf_counter()
        for _ in range(3):
            o = ctx.kopt2d(d, B, F, neff, do_corr, fb, corr)
        res[route] = (o, (time.perf_counter() - t0) / 3 * 1e3)
    a, b_ = res["fft"][0], res["gemm"][0]
    ok = a[:, 7] == 0
    assert np.array_equal(a[:, 7], b_[:, 7]) and ok.sum() > 0.9 * B, (F, a[:, 7], b_[:, 7])
    rel = np.abs(a[ok, :7] - b_[ok, :7]) / np.maximum(np.abs(b_[ok, :7]), 1e-300)
    rel = np.where(np.isnan(rel), 0, rel)
    hrel = np.abs(a[ok, 8:11] - b_[ok, 8:11]) / np.maximum(np.abs(b_[ok, 8:11]), 1e-300)
    out[F] = dict(pairs=B, max_rel_t_psi=float(rel.max()), frac_bandwidths_within_1e_6=float(np.mean(np.nanmax(hrel, axis=1) < 1e-6)),
                  ms_fft=res["fft"][1], ms_gemm=res["gemm"][1])
    print(F, out[F])
    assert rel.max() < 1e-10
    d.free()
os.environ.pop("GDHIP_KOPT_DCT_GEMM", None)
os.makedirs("gpurun_out/r04", exist_ok=True)
json.dump(out, open("gpurun_out/r04/dct_route_check.json", "w"), indent=1)
