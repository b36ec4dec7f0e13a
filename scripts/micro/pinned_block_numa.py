"""Where do the page-locked result blocks live?  NUMA nodes of the host, the GPU's node, and the per-node page counts of
three 642-MB blocks from gd_host_alloc (/proc/self/numa_maps).
    python scripts/micro/pinned_block_numa.py"""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from getdist_amd._lib import Context

print("nodes online:", open("/sys/devices/system/node/online").read().strip())
for p in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")):
    try:
        vendor = open(os.path.join(os.path.dirname(p), "vendor")).read().strip()
    except OSError:
        vendor = "?"
    print(p, "numa_node", open(p).read().strip(), "vendor", vendor)
print("this process may run on cpus:", open("/proc/self/status").read().split("Cpus_allowed_list:")[1].split()[0],
      " mems:", open("/proc/self/status").read().split("Mems_allowed_list:")[1].split()[0])
ctx = Context(0)
n = 1225 * 65536
blocks = [ctx.pinned_array((n,), np.float64) for _ in range(3)]
maps = open("/proc/self/numa_maps").read().splitlines()
for b in blocks:
    a = b.ctypes.data
    hit = [l for l in maps if l.split()[0] == "%x" % a]
    print("block at %x:" % a, hit[0][:200] if hit else "(no numa_maps line starts at this address)")
if not any(l.split()[0] == "%x" % blocks[0].ctypes.data for l in maps):
    big = [l for l in maps if any(t.startswith("N") and "=" in t and int(t.split("=")[1]) > 100000 for t in l.split())]
    print("large mappings:")
    for l in big[:12]:
        print("   ", l[:200])
