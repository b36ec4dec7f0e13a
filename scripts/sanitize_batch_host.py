"""
Race / memory evidence for the host side of gd_density2d_batch (getdist_amd/csrc/batch2d.hpp: four host threads over
three contexts -- N_eff / binning / shear chains, the optimiser's staging thread and its finisher): the CPU harness built
with -fsanitize=thread or -fsanitize=address, driven through the staged large-call route and the small-call route on the
numpy context double (whose entry points are called back from those threads).

    LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS="report_signal_unsafe=0 exitcode=66" \\
        python scripts/sanitize_batch_host.py /tmp/san/libbatch_harness_tsan.so
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS="detect_leaks=0 exitcode=67" \\
        python scripts/sanitize_batch_host.py /tmp/san/libbatch_harness_asan.so
(build lines in profiles/README.md).  Test infrastructure; prints the number of calls made and exits 0 when the sanitizer
stayed silent (its reports go to stderr and change the exit code).
"""
import os
import sys

os.environ.setdefault("GDHIP_BATCH_SHEAR_DEFERRED_MIN", "16")  # the deferred shear chain and the two-launch main binning (round 6)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")  # (OpenBLAS's own pool is not instrumented: its hand-offs read as races)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "native"))
import ctypes

import numpy as np

import build

so = sys.argv[1]
build.load_batch = lambda: ctypes.CDLL(so)  # the instrumented harness instead of the plain one
import native_batch_util as nb
from getdist_amd.mcsamples import MCSamples
from oracle.fixtures import fixture_zoo

fx = [f for f in fixture_zoo() if f["name"] == "block50"][0]


def make(factory):
    return MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"], _context_factory=factory)


pairs = [(i, j) for i in range(13) for j in range(i + 1, 13)]
plain = make(nb.PlainContext).get2DDensities(pairs)
calls = 0
for rep in range(3):
    mc = make(nb.HarnessContext)
    mc.CONV_TWO_STREAMS_PAIRS = (8, 20)
    mc.KOPT_SPLIT_MIN = 8
    for sel in (pairs, pairs[:30], pairs[:5]):
        out = mc.get2DDensities(sel)
        calls += 1
        assert all(np.array_equal(a.P, b.P) for a, b in zip(out, plain))
        mc.ctx.batch2d_invalidate()
        for p in mc.paramNames.names:
            p.N_eff_kde = None
    mc.ctx.batch2d_finish()
print("sanitized host run: %d batched calls (staged three-stream route, two-stream route, small route), grids equal to the "
      "Python-planned route" % calls)
