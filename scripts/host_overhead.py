"""Host-side (Python) cost of one bench step, measured without a GPU: the numpy test double of the C ABI with its
heavy entries (optimiser, convolution, histograms) replaced by constant-time stubs, so what is left is this package's
own per-step Python.  Usage: python scripts/host_overhead.py [--profile]"""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "1"); os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
import bench
from fake_ctx import FakeContext, FakeBuf
from getdist_amd import synth
from getdist_amd.mcsamples import MCSamples


class StubBuf(FakeBuf):
    def to_host(self, shape, dtype=np.float64, offset_bytes=0, pinned=False):
        return self.a.reshape(shape)

    def to_host_async(self, shape, dtype=np.float64):
        return self.a.reshape(shape)


_ZERO = {}


def zeros(B, F):
    if (B, F) not in _ZERO:
        _ZERO[(B, F)] = np.zeros((B, F, F))
    return _ZERO[(B, F)]


class StubContext(FakeContext):
    def kopt2d(self, d_hist, B, F, neff, do_corr, fallback_t, corr):
        out = np.zeros((B, 12))
        out[:, 0] = 1e-4
        out[:, 1:7] = 1.0
        out[:, 8:10] = 0.02
        out[:, 10] = np.asarray(corr) * 0.5
        return out

    def density2d(self, d_hist, B, F, rx, ry, corr, winw, flags, bco, mbc, out=None):
        return StubBuf(zeros(B, F), B * F * F * 8), np.zeros(B, dtype=np.int32)

    def hist2d_prebinned(self, idx_x, idx_y, F, out=None):
        return StubBuf(zeros(len(idx_x), F), 0)

    def hist2d_prebinned8(self, idx_x, idx_y):
        return StubBuf(zeros(len(idx_x), 256), 0)

    def prebin8_batch(self, cols, binmin, width, F, bufs):
        return np.zeros(len(cols), dtype=np.int64)

    def prebin_batch(self, cols, binmin, width, F, bufs):
        pass

    def hist2d_sheared(self, coli, colj, r0, r1, xmin, dx, ymin, dy, F, out=None):
        return StubBuf(zeros(len(coli), F), 0)

    def gather_items(self, dst, src, index, item_bytes, dst_offset=0):
        pass


def main():
    s, w, names, ranges = synth.block_recipe(50, 20000, weighted=False, stream=1)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges, _context_factory=StubContext)
    pairs = synth.triangle_pairs(50)
    bench.one_step(mc, pairs, None, 0, 1, None)
    t0 = time.perf_counter()
    for _ in range(3):
        bench.one_step(mc, pairs, None, 0, 1, None)
    print("step (stubbed kernels): %.1f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
    if mc.timings:
        print({k: round(v / 4 * 1e3, 2) for k, v in sorted(mc.timings.items())})
    if "--profile" in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(3):
            bench.one_step(mc, pairs, None, 0, 1, None)
        pr.disable()
        st = io.StringIO()
        pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(35)
        print(st.getvalue()[:7000])


main()
