"""
Convergence configuration C4 with ONE CHAIN PER GPU: each rank generates / uploads its own chain,
computes the chain's weighted means, covariance and norm with one gd_cov launch, the ranks all-gather n^2+n+1
doubles -- through the library communicator (gd_comm_allgather: ncclAllGather inside the C ABI) with backend nccl,
torch.distributed otherwise -- and evaluate Gelman-Rubin + MeanVar.  --emulate-world W (one process, one GPU): this rank's
share of a W-chain job -- its own chain's moments, the other chains' contributions replayed -- for the scaling table.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
      scripts/gelman_rubin_multi_gpu.py --rows 5000000 --params 100

--backend gloo --share-device runs all ranks on GPU 0 (for 1-GPU boxes); with no launcher it is a 1-chain run.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=5_000_000, help="rows per chain")
    ap.add_argument("--params", type=int, default=100)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--share-device", action="store_true")
    ap.add_argument("--emulate-world", type=int, default=0)
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    dist = device = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            device = torch.device("cuda", local_rank)
        dist_mod.init_process_group(backend=args.backend, rank=rank, world_size=world)
        dist = dist_mod

    from getdist_amd import parallel, synth
    from getdist_amd.mcsamples import MCSamples

    s, w, names = synth.config_c4_chain(rank, args.rows, args.params)
    t0 = time.perf_counter()
    mc = MCSamples(samples=s, weights=w, names=names, device=local_rank)
    t_ctor = time.perf_counter() - t0
    comm = None
    if dist is not None and args.backend == "nccl" and os.environ.get("GETDIST_AMD_COMM", "lib") == "lib":
        comm = parallel.init_library_comm(mc.ctx, dist, rank, world, device)  # None on every rank if any rank cannot
    if args.emulate_world:
        # what the other W - 1 ranks would all-gather: their chains' moments, computed once on this GPU from their seeds
        W = args.emulate_world
        rows = []
        for r in range(1, W):
            s_r, w_r, _ = synth.config_c4_chain(r, args.rows, args.params)
            other = MCSamples(samples=s_r, weights=w_r, names=names, device=local_rank)
            m_, c_, n_ = other.ctx.cov(list(range(args.params)))
            rows.append(np.concatenate([m_, np.asarray(c_).ravel(), [n_]]))
            other.ctx.close()

        class Replay:
            world = W

            @staticmethod
            def allgather(buf):
                return np.stack([buf] + rows)

        comm = Replay()
    res = parallel.convergence_chain_per_rank(mc, dist, device, comm=comm)  # warm-up
    times = []
    for _ in range(args.reps):
        mc.ctx.sync()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        res = parallel.convergence_chain_per_rank(mc, dist, device, comm=comm)
        mc.ctx.sync()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    if dist is not None:
        import torch

        tt = torch.tensor([t], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    if rank == 0:
        chains = args.emulate_world or world
        print(json.dumps(dict(config="C4 chain-per-GPU", chains=chains, rows_per_chain=args.rows, params=args.params,
                              emulated_world=args.emulate_world or None,
                              collectives=("replayed" if args.emulate_world else "libgdhip gd_comm_allgather" if comm is not None
                                           else "torch.distributed" if world > 1 else None),
                              construct_upload_s=round(t_ctor, 3), gr_meanvar_ms=round(t * 1e3, 3),
                              R_minus_1=res["R_minus_1"], meanvar_max=float(np.max(res["meanvar"])) if chains > 1 else None,
                              backend=args.backend if world > 1 else None)))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
