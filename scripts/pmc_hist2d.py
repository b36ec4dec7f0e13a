"""
Launch ONLY the 2D binning kernels (for rocprofv3 --pmc passes; see profiles/README.md):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o h -- python scripts/pmc_hist2d.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o h -- python scripts/pmc_hist2d.py
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES \
              --output-format csv -d gpurun_out/pmc_sq -o h -- python scripts/pmc_hist2d.py
Three launches each, over the base-grid pairs of the C3 triangle (F=256, unit weights): the byte-index batched kernel
(k_hist2d_u8, the one bench.py's roofline block times), the u16 batched kernel, the fused-fp64 packed kernel, and the
sheared kernel on 79 pairs.  scripts/summarise_pmc.py turns the CSVs into profiles/r02_pmc_hist2d.json.
"""

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from getdist_amd import synth  # noqa: E402
from getdist_amd.mcsamples import MCSamples  # noqa: E402


def main():
    only_u8 = "--only-u8" in sys.argv  # bench.py's live counter pass: the roofline kernel alone, three launches
    N, n, F = 10_000_000, 50, 256
    for a in sys.argv[1:]:
        if a.startswith("--nsamples="):
            N = int(a.split("=")[1])
        if a.startswith("--nparams="):
            n = int(a.split("=")[1])
    s, w, names, ranges = synth.config_c3(N, n)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    mc.prepareParams(neff=False)
    ctx = mc.ctx
    par = mc.paramNames.names
    e = [mc._bin_edges(p, F) for p in par]
    corr = mc.getCorrelationMatrix()
    pairs = [p for p in synth.triangle_pairs(n) if abs(corr[p[1]][p[0]]) <= 0.866]  # the 1200 base-grid pairs
    assert mc._index_columns8({j: (e[j][1], e[j][0]) for j in range(n)})
    i8 = [mc._idx_cols[(j, 256, "u8")][0] for j in range(n)]
    idx = [] if only_u8 else [mc._index_column(j, F, e[j][1], e[j][0]) for j in range(n)]
    out = ctx.alloc(len(pairs) * F * F * 8)
    for _ in range(3):
        ctx.hist2d_prebinned8([i8[a] for a, b in pairs], [i8[b] for a, b in pairs], out=out)
    if only_u8:
        ctx.sync()
        print("done: %d pairs (byte-index kernel only)" % len(pairs))
        return
    for _ in range(3):
        ctx.hist2d_prebinned([idx[a] for a, b in pairs], [idx[b] for a, b in pairs], F, out=out)
    for _ in range(3):
        ctx.hist2d([a for a, b in pairs], [b for a, b in pairs], [e[a][1] for a, b in pairs], [e[a][0] for a, b in pairs],
                   [e[b][1] for a, b in pairs], [e[b][0] for a, b in pairs], F, out=out)
    sh = pairs[:79]
    for _ in range(3):
        ctx.hist2d_sheared([a for a, b in sh], [b for a, b in sh], [1.0] * 79, [-0.4] * 79, [e[a][1] for a, b in sh],
                           [e[a][0] for a, b in sh], [-12.0] * 79, [24.0 / 255] * 79, F, out=out)
    ctx.sync()
    print("done: %d pairs; streaming model per launch = %.3f GB (16N + 8F^2 per density, unit weights)"
          % (len(pairs), len(pairs) * (16.0 * N + 8 * F * F) / 1e9))


if __name__ == "__main__":
    main()
