"""Do page-locked result blocks differ in D2H bandwidth?  Eight 642-MB blocks from gd_host_alloc (the block size of a C3
triangle's grids), the same device buffer copied into each three times: GB/s per block.  (The delivered-triangle loop of
bench.py alternates between two such blocks; in some processes every other step is ~1.8 ms slower.)
    python scripts/micro/pinned_block_bw.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from getdist_amd._lib import Context

ctx = Context(0)
n = 1225 * 65536
dev = ctx.alloc(n * 8)
blocks = [ctx.pinned_array((n,), np.float64) for _ in range(8)]  # all alive: eight distinct blocks
assert len({b.ctypes.data for b in blocks}) == 8
for rep in range(3):
    row = []
    for b in blocks:
        ctx.sync()
        t0 = time.perf_counter()
        ctx._check(ctx.lib.gd_memcpy_d2h(ctx.h, b.ctypes.data, dev.ptr, b.nbytes))
        row.append(b.nbytes / (time.perf_counter() - t0) / 1e9)
    print("pass %d GB/s per block: %s" % (rep, " ".join("%.1f" % v for v in row)))
print("addresses mod 2 MiB:", [hex(b.ctypes.data % (1 << 21)) for b in blocks])
