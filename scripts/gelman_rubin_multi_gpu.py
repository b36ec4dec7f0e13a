"""
Convergence configuration C4 with ONE CHAIN PER GPU: each rank generates / uploads its own chain,
computes the chain's weighted means, covariance and norm with one gd_cov launch, the ranks all-gather n^2+n+1
doubles (torch.distributed; backend nccl = RCCL over xGMI) and evaluate Gelman-Rubin + MeanVar.

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
    res = parallel.convergence_chain_per_rank(mc, dist, device)  # warm-up
    times = []
    for _ in range(args.reps):
        mc.ctx.sync()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        res = parallel.convergence_chain_per_rank(mc, dist, device)
        mc.ctx.sync()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    if dist is not None:
        import torch

        tt = torch.tensor([t], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    if rank == 0:
        print(json.dumps(dict(config="C4 chain-per-GPU", chains=world, rows_per_chain=args.rows, params=args.params,
                              construct_upload_s=round(t_ctor, 3), gr_meanvar_ms=round(t * 1e3, 3),
                              R_minus_1=res["R_minus_1"], meanvar_max=float(np.max(res["meanvar"])) if world > 1 else None,
                              backend=args.backend if world > 1 else None)))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
