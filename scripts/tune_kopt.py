"""Time gd_kopt2d (DCT GEMMs + fixed point + get_h) on C3-like histograms; used to compare build variants on the GPU box."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from getdist_amd._lib import Context
from getdist_amd import synth

def main():
    N, n, F = 4_000_000, 16, 256
    samples = synth.block_recipe(n, N)[0]
    ctx = Context(0)
    ctx.upload(samples, None)
    pairs = [(i, j) for i in range(n) for j in range(i)] * 10
    mn, mx = samples.min(0), samples.max(0)
    ix = [ctx.prebin(j, mn[j], (mx[j] - mn[j]) / (F - 1), F) for j in range(n)]
    hp = ctx.hist2d_prebinned([ix[a] for a, b in pairs], [ix[b] for a, b in pairs], F)
    B = len(pairs)
    for dc in (0, 1):
        ts = []
        for r in range(4):
            t0 = time.perf_counter()
            out = ctx.kopt2d(hp, B, F, [1e6] * B, [dc] * B, [1e-4] * B, [0.3] * B)
            ts.append(time.perf_counter() - t0)
        print("kopt2d B=%d do_corr=%d: %.3f ms (min of 3), t*[0]=%.17g" % (B, dc, 1e3 * min(ts[1:]), out[0, 0]))

main()
