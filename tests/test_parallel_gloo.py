"""
CPU tests (gloo, world_size 2) of the multi-GPU decomposition in getdist_amd/parallel.py: the per-parameter
state all-gather and the cost-class pair partition.  The compute itself needs a GPU and is covered by the
-m gpu tests; here the per-parameter state is faked.
"""

import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from getdist_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_pairs_covers_everything_once():
    pairs = [(i, j) for i in range(12) for j in range(i + 1, 12)]
    key = lambda p: (p[0] % 3, p[1] % 2)  # noqa: E731
    for world in (1, 2, 3, 8):
        seen = []
        sizes = []
        for rank in range(world):
            idx, mine = parallel.partition_pairs(pairs, key, world, rank)
            assert [pairs[i] for i in idx] == mine
            seen += idx
            sizes.append(len(idx))
        assert sorted(seen) == list(range(len(pairs)))
        assert max(sizes) - min(sizes) <= 1
        # every rank gets (nearly) the same number of pairs of each cost class
        for cls in set(map(key, pairs)):
            per_rank = [sum(1 for p in parallel.partition_pairs(pairs, key, world, r)[1] if key(p) == cls)
                        for r in range(world)]
            assert max(per_rank) - min(per_rank) <= 1


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch.distributed as dist
    from getdist_amd import parallel

    class Par:  # stand-in for ParamInfo
        pass

    class FakeMC:
        def __init__(self, n):
            class PN: pass
            self.paramNames = PN()
            self.paramNames.names = [Par() for _ in range(n)]
            self.n = n

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 7
    mc = FakeMC(n)
    mine = parallel.partition_round_robin(list(range(n)), world, rank)
    for j in mine:  # "prepare" my parameters
        p = mc.paramNames.names[j]
        for c, a in enumerate(parallel.PARAM_STATE):
            setattr(p, a, (j %% 2 == 0) if a.startswith("has_limits") else 100.0 * j + c)
        if j == 3:
            p.N_eff_kde = None
    parallel.allgather_param_state(mc, mine, n, dist)
    for j in range(n):
        p = mc.paramNames.names[j]
        assert p._ranges_done
        for c, a in enumerate(parallel.PARAM_STATE):
            v = getattr(p, a)
            if a.startswith("has_limits"):
                assert v is (j %% 2 == 0), (j, a, v)
            elif a == "N_eff_kde" and j == 3:
                assert v is None
            else:
                assert v == 100.0 * j + c, (j, a, v)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_allgather_param_state_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out


def test_pack_unpack_roundtrip():
    class Par:
        pass

    class MC:
        pass

    mc = MC()
    mc.paramNames = MC()
    mc.paramNames.names = [Par() for _ in range(3)]
    for j, p in enumerate(mc.paramNames.names):
        for c, a in enumerate(parallel.PARAM_STATE):
            setattr(p, a, bool(j % 2) if a.startswith("has_limits") else np.float64(j + 0.25 * c))
    rows = parallel.pack_param_state(mc, [0, 2])
    mc2 = MC()
    mc2.paramNames = MC()
    mc2.paramNames.names = [Par() for _ in range(3)]
    parallel.unpack_param_state(mc2, rows)
    for j in (0, 2):
        for a in parallel.PARAM_STATE:
            assert getattr(mc2.paramNames.names[j], a) == getattr(mc.paramNames.names[j], a)


def test_gelman_rubin_from_chain_stats_matches_golden():
    """Host half of the C4 path: numpy per-chain moments -> eigenvalues, against the reference's golden values."""
    import golden_util as gu
    from getdist_amd import synth

    g = np.load(gu.GOLDEN_DIR + "/convergence.npz")
    samples, weights, names, offsets = synth.config_c4(nchains=4, N=20000, n=8)
    stats = []
    for a, b in zip(offsets[:-1], offsets[1:]):
        x, w = samples[a:b], weights[a:b]
        m = w.dot(x) / w.sum()
        d = x - m
        stats.append((m, (d * w[:, None]).T @ d / w.sum(), w.sum()))
    st2, pooled = parallel.allgather_chain_stats(*stats[0])  # single-process path returns the local chain only
    assert len(st2) == 1 and np.allclose(pooled, stats[0][0])
    tot = sum(s_[2] for s_ in stats)
    pooled = sum(s_[2] * s_[0] for s_ in stats) / tot
    assert np.allclose(pooled, g["means"], rtol=1e-12)
    D = parallel.gelman_rubin_from_chain_stats(stats, pooled)
    assert gu.relerr(D, g["gr_eigenvalues"]) < 1e-9


CHAIN_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np
    import torch.distributed as dist
    from fake_ctx import FakeContext
    from getdist_amd import parallel, synth
    from getdist_amd.mcsamples import MCSamples

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, w, names = synth.config_c4_chain(rank, 6000, 6)
    mc = MCSamples(samples=s, weights=w, names=names, _context_factory=FakeContext)   # this rank's chain only
    res = parallel.convergence_chain_per_rank(mc, dist)
    # single-process comparator: all chains in one sample set, through the list-of-chains API
    chains = [synth.config_c4_chain(c, 6000, 6) for c in range(world)]
    allmc = MCSamples(samples=[c[0] for c in chains], weights=[c[1] for c in chains], names=names,
                      _context_factory=FakeContext)
    D = allmc.getGelmanRubinEigenvalues()
    assert np.allclose(res["D"], D, rtol=1e-10), (res["D"], D)
    assert np.allclose(res["meanvar"], allmc.getMeanVarTest(), rtol=1e-10)
    assert np.allclose(res["pooled_means"], allmc.means, rtol=1e-12)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_chain_per_rank_convergence_gloo_world2(tmp_path):
    """one chain per rank, all-gather of the chain moments, Gelman-Rubin + MeanVar on every rank."""
    script = tmp_path / "chain_worker.py"
    script.write_text(CHAIN_WORKER % dict(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out


def test_bench_launcher_spawns_the_ranks_it_was_asked_for():
    """`python bench.py --gpus 2` without a launcher around it re-launches itself under torch.distributed.run: the
    JSON line reports two ranks from the communicator (gloo and the numpy context double here, RCCL on the GPU box).
    17 parameters = 68 pairs per rank: enough for the overlapped pipeline (N_eff kernels on the helper thread beside the
    binning, their exchange from the main thread once they are through)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(root, "tests"), root, env.get("PYTHONPATH", "")])
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--nsamples", "6000",
           "--nparams", "17", "--backend", "gloo", "--share-device", "--no-cpu-baseline", "--context-factory",
           "fake_ctx:FakeContext"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and len(line["ms_per_step_by_rank"]) == 2
    assert line["value"] > 0 and line["scaling"] == "strong"
    # a communicator that does not have --gpus ranks is refused, not silently measured
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    bad = subprocess.run(cmd, env=env2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert bad.returncode != 0 and "--gpus 2" in (bad.stderr + bad.stdout)


COMM_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch
    import torch.distributed as dist
    from getdist_amd import parallel

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class FakeLibCtx:  # the gd_comm_* surface of _lib.Context, its collectives carried by gloo
        def __init__(self, fail=None):
            self.fail, self.comm_world, self.destroyed = fail, 0, 0
        def comm_unique_id(self):
            if self.fail == "load":
                raise RuntimeError("librccl.so: cannot open shared object file")
            return b"x" * 128
        def comm_init(self, world, rank, id128):
            assert id128 == b"x" * 128
            if self.fail == "init":
                raise RuntimeError("ncclCommInitRank failed")
            self.comm_world = world
        def comm_allreduce_sum(self, vec):
            t = torch.from_numpy(np.array(vec, dtype=np.float64))
            dist.all_reduce(t)
            out = t.numpy()
            if self.fail == "sum":
                out = out + 1.0
            return out
        def comm_destroy(self):
            self.destroyed += 1
            self.comm_world = 0

    # every rank healthy: the library's communicator on both
    ctx = FakeLibCtx()
    comm = parallel.init_library_comm(ctx, dist, rank, world)
    assert isinstance(comm, parallel.LibraryComm) and comm.world == world and ctx.destroyed == 0
    assert np.array_equal(comm.allreduce_sum(np.array([1.0, rank])), [world, world * (world - 1) / 2])
    # one rank cannot: None on EVERY rank, whatever stage fails, and nobody is left inside a collective
    for stage, bad_rank in (("load", 1), ("init", 0), ("sum", 1)):
        ctx = FakeLibCtx(stage if rank == bad_rank else None)
        assert parallel.init_library_comm(ctx, dist, rank, world) is None, (stage, rank)
        if stage == "init":
            assert ctx.destroyed == (0 if rank == bad_rank else 1)
        if stage == "sum":
            assert ctx.destroyed == 1
    dist.barrier()
    print("rank", rank, "ok")
""")


def test_library_communicator_is_all_or_nothing_gloo_world2(tmp_path):
    """parallel.init_library_comm: the ranks agree after each stage (library loadable, communicator created, a test
    all-reduce right) whether the step's collectives go through the C ABI or stay with torch.distributed -- a failure on one
    rank at any stage gives None on every rank, never a mismatched collective."""
    script = tmp_path / "comm_worker.py"
    script.write_text(COMM_WORKER % dict(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out
