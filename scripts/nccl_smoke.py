"""One-GPU smoke of what a multi-rank bench process does besides the densities: torch's HIP runtime, RCCL (backend "nccl")
and libgdhip.so in ONE process -- a single-rank communicator, the all-gathers of parallel.py on GPU tensors, a batched call
between them.  (The multi-rank logic itself is covered by the gloo tests; 8-GPU runs are the driver's.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

from getdist_amd import synth
from getdist_amd.mcsamples import MCSamples

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
s, w, names, ranges = synth.config_c3(200_000, 12)
mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges, device=0)
t = torch.from_numpy(np.arange(8, dtype=np.float64)).to(dev)
out = [torch.empty_like(t)]
dist.all_gather(out, t)
assert np.array_equal(out[0].cpu().numpy(), np.arange(8.0))
d1 = mc.get2DDensities(synth.triangle_pairs(12))
from concurrent.futures import ThreadPoolExecutor

with ThreadPoolExecutor(1) as ex:  # a collective while another thread of the process is inside the library
    f = ex.submit(mc.getMargeStats)
    dist.all_gather(out, t)
    f.result()
dist.barrier()
torch.cuda.synchronize()
d2 = mc.get2DDensities(synth.triangle_pairs(12))
assert all(np.array_equal(a.P, b.P) for a, b in zip(d1, d2))
# RCCL through the C ABI (gd_comm_*): a single-rank communicator made from an id that travelled through torch.distributed,
# the two collectives on the library's stream, and a batched call that exchanges its N_eff values over it
from getdist_amd import parallel

comm = parallel.init_library_comm(mc.ctx, dist, 0, 1)
v = np.arange(12, dtype=np.float64) * 1.5
assert comm.world == 1 and np.array_equal(comm.allgather(v), v[None, :]) and np.array_equal(comm.allreduce_sum(v), v)
for p in mc.paramNames.names:
    p.N_eff_kde = None
mc.ctx.batch2d_invalidate()
share = parallel.NeffShare(list(range(mc.n)), lambda mc_: parallel.allgather_neff(mc_, list(range(mc_.n)), mc_.n, comm=comm))
share.library_comm = True
mc._neff_share = share
d3 = mc.get2DDensities(synth.triangle_pairs(12))
mc._neff_share = None
assert share.exchanged and all(np.array_equal(a.P, b.P) for a, b in zip(d1, d3))
parallel.allgather_param_state(mc, list(range(mc.n)), mc.n, comm=comm)
# ONE RCCL in the process: the library resolved its collectives from the copy torch had already mapped
path, preloaded = mc.ctx.comm_rccl_path()
maps = [ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln]
assert preloaded and len(set(maps)) == 1 and os.path.samefile(path, maps[0]), (path, preloaded, sorted(set(maps)))
# the convergence configuration's all-gather through the library communicator (one chain per rank; here one rank)
res = parallel.convergence_chain_per_rank(mc, comm=comm)
assert res["D"] is None and np.allclose(res["pooled_means"], mc.means[:len(res["pooled_means"])], rtol=1e-13)
# sample distribution by column shards: this rank's block over PCIe, the broadcast group over the communicator; the set
# on the device is the one a full upload leaves (densities bit-equal)
share2 = parallel.ColumnShare(dist, 0, 1)  # (a single rank: its block is everything, the broadcast goes to itself)
mc2 = MCSamples(samples=s, weights=w, names=names, ranges=ranges, device=0, column_share=share2)
assert share2.comm is not None and share2.bytes_uploaded == s.size * 8
d4 = mc2.get2DDensities(synth.triangle_pairs(12))
assert all(np.array_equal(a.P, b.P) for a, b in zip(d1, d4))
mc2.ctx.comm_destroy()
mc.ctx.comm_destroy()
# the watchdog of gd_comm_init: a rank that takes longer than GDHIP_COMM_TIMEOUT_S to join is an error return, not a hang
# (GDHIP_COMM_INJECT_HANG_MS delays the helper thread that calls ncclCommInitRank), and the all-or-nothing set-up then
# leaves the job on torch.distributed's collectives
import time

os.environ["GDHIP_COMM_TIMEOUT_S"], os.environ["GDHIP_COMM_INJECT_HANG_MS"] = "1", "4000"
t0 = time.time()
assert parallel.init_library_comm(mc.ctx, dist, 0, 1) is None and time.time() - t0 < 3.5
del os.environ["GDHIP_COMM_INJECT_HANG_MS"]
os.environ["GDHIP_COMM_TIMEOUT_S"] = "120"
time.sleep(3.5)  # (the abandoned helper finishes and aborts the communicator it obtained)
comm = parallel.init_library_comm(mc.ctx, dist, 0, 1)  # and the next set-up works
assert comm is not None and np.array_equal(comm.allreduce_sum(v), v)
# a host-vector collective whose wait gives up (GDHIP_COMM_INJECT_WAIT_TIMEOUT: as after GDHIP_COMM_[STEADY_]TIMEOUT_S with a
# missing peer): a distinct status (GD_ERR_TIMEOUT -> parallel.CommTimeout), the communicator aborted and dropped, the stream
# drained and the pending result deliveries forgotten -- the caller's vector (released by the binding when the error is
# raised) is not written afterwards, and the NEXT synchronising call on the same context (an upload) works
import gc

os.environ["GDHIP_COMM_INJECT_WAIT_TIMEOUT"] = "1"
try:
    comm.allgather(v)
    raise AssertionError("the injected timeout did not surface")
except parallel.CommTimeout as exc:
    assert exc.code == -8 and "aborted" in str(exc)
del os.environ["GDHIP_COMM_INJECT_WAIT_TIMEOUT"]
gc.collect()
world_after, _ = mc.ctx.comm_info()
assert world_after == 0  # dropped by the library
mc.ctx.upload(s, w)  # gd_upload -> gd_stream_sync on the same context: nothing left to deliver into freed memory
mc.ctx.sync()
comm = parallel.init_library_comm(mc.ctx, dist, 0, 1)  # and a fresh communicator can be made
assert comm is not None and np.array_equal(comm.allreduce_sum(v), v)
mc.ctx.comm_destroy()
dist.destroy_process_group()
print("nccl smoke ok (torch.distributed + gd_comm_*): torch %s, %d densities four times, bit-equal; RCCL = %s" % (torch.__version__, len(d1), path))
