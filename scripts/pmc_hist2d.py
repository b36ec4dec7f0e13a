"""
Launch ONLY the 2D binning kernels (for rocprofv3 --pmc passes; see profiles/README.md):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -o h -- python scripts/pmc_hist2d.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -o h -- python scripts/pmc_hist2d.py
Three launches each of the pre-binned and of the fused-fp64 kernel over the full C3 triangle at F=256.
"""

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from getdist_amd import synth  # noqa: E402
from getdist_amd.mcsamples import MCSamples  # noqa: E402


def main():
    N, n, F = 10_000_000, 50, 256
    s, w, names, ranges = synth.config_c3(N, n)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges)
    mc.prepareParams(neff=False)
    ctx = mc.ctx
    par = mc.paramNames.names
    e = [mc._bin_edges(p, F) for p in par]
    pairs = synth.triangle_pairs(n)
    idx = [mc._index_column(j, F, e[j][1], e[j][0]) for j in range(n)]
    out = ctx.alloc(len(pairs) * F * F * 8)
    for _ in range(3):
        ctx.hist2d_prebinned([idx[a] for a, b in pairs], [idx[b] for a, b in pairs], F, out=out)
    for _ in range(3):
        ctx.hist2d([a for a, b in pairs], [b for a, b in pairs], [e[a][1] for a, b in pairs], [e[a][0] for a, b in pairs],
                   [e[b][1] for a, b in pairs], [e[b][0] for a, b in pairs], F, out=out)
    ctx.sync()
    print("done: %d pairs, algorithmic bytes per launch = %.3f GB" % (len(pairs), len(pairs) * (24.0 * N + 8 * F * F) / 1e9))


if __name__ == "__main__":
    main()
