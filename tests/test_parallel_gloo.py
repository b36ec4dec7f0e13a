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

    import datetime, time
    # the "library's" collectives travel on a group of their own, as RCCL's do beside torch.distributed's
    lib_group = dist.new_group(timeout=datetime.timedelta(seconds=6))

    class FakeLibCtx:  # the gd_comm_* surface of _lib.Context, its collectives carried by gloo
        def __init__(self, fail=None):
            self.fail, self.comm_world, self.destroyed = fail, 0, 0
        def comm_unique_id(self):
            if self.fail == "load":
                raise RuntimeError("librccl.so: cannot open shared object file")
            if self.fail == "hang_load":
                time.sleep(3600)
            return b"x" * 128
        def comm_init(self, world, rank, id128):
            assert id128 == b"x" * 128
            if self.fail == "init":
                raise RuntimeError("ncclCommInitRank failed")
            if self.fail == "hang_init":
                time.sleep(3600)  # a rank that never joins
            self.comm_world = world
        def comm_allreduce_sum(self, vec):
            if self.fail == "hang_sum":
                time.sleep(3600)  # never enters the collective: the peer's all-reduce has no partner
            t = torch.from_numpy(np.array(vec, dtype=np.float64))
            dist.all_reduce(t, group=lib_group)
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
    # a HANG at any stage (a rank that never arrives, a collective without its partner) ends the same way: every stage runs
    # under a watchdog, the rank that timed out reports failure, and every rank keeps torch.distributed's collectives
    for stage, bad_rank in (("hang_load", 0), ("hang_init", 1), ("hang_sum", 0)):
        ctx = FakeLibCtx(stage if rank == bad_rank else None)
        t0 = time.time()
        assert parallel.init_library_comm(ctx, dist, rank, world, stage_timeout=1.5) is None, (stage, rank)
        assert time.time() - t0 < 30.0, (stage, rank)
        # a gd_comm_* call that outlived its watchdog may still be inside the library on that context: the context is
        # flagged, and the fallback upload refuses to run on it (round 6, advisor finding)
        # (hang_sum: the peer's all-reduce has no partner and outlives its watchdog too)
        assert parallel.context_is_stuck(ctx) == (stage == "hang_sum" or (rank == bad_rank and stage == "hang_init")), (stage, rank)
        if parallel.context_is_stuck(ctx):
            share = parallel.ColumnShare(dist, rank, world)
            share.tried = True
            try:
                share.upload(ctx, np.zeros((4, 2)), None)
                raise AssertionError("a stuck context was reused")
            except parallel.StageTimeout:
                pass
    # ... and torch.distributed itself is still in step afterwards
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    assert t.item() == world * (world + 1) / 2
    dist.barrier()
    print("rank", rank, "ok")
    sys.stdout.flush()
    os._exit(0)  # (helper threads left inside the injected hangs do not keep the worker alive)
""")


def test_library_communicator_is_all_or_nothing_gloo_world2(tmp_path):
    """parallel.init_library_comm: the ranks agree after each stage (library loadable, communicator created, a test
    all-reduce right) whether the step's collectives go through the C ABI or stay with torch.distributed -- a failure on one
    rank at any stage gives None on every rank, never a mismatched collective.  Round 5: so does a HANG injected at each
    stage (unique id, communicator creation, test all-reduce): the stages run under watchdogs (parallel.call_with_watchdog
    here, GDHIP_COMM_TIMEOUT_S inside the C ABI)."""
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


SHARE_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np
    import torch
    import torch.distributed as dist
    from fake_ctx import FakeContext
    from getdist_amd import parallel, synth
    from getdist_amd.mcsamples import MCSamples

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class CommCtx(FakeContext):  # the numpy double plus the gd_comm_* surface, carried by gloo
        comm_world = 0
        def comm_unique_id(self):
            return b"y" * 128
        def comm_init(self, world, rank, id128):
            self.comm_world, self.comm_rank = world, rank
        def comm_destroy(self):
            self.comm_world = 0
        def comm_allreduce_sum(self, vec):
            t = torch.from_numpy(np.array(vec, dtype=np.float64))
            dist.all_reduce(t)
            return t.numpy()
        def comm_allgather(self, vec):
            t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
            out = [torch.empty_like(t) for _ in range(self.comm_world)]
            dist.all_gather(out, t)
            return np.stack([o.numpy() for o in out])
        def comm_broadcast(self, arr, root):
            t = torch.from_numpy(np.ascontiguousarray(arr))
            dist.broadcast(t, src=root)
            return t.numpy()

    s, w, names, ranges = synth.config_c1(20000, bounded=True)
    w = np.random.default_rng(3).exponential(size=len(s))
    # a rank whose broadcasts fail sends EVERY rank back to the full upload and the torch collectives
    class Broken(CommCtx):
        def comm_share_columns(self, first_by_rank):
            if rank == 1:
                raise RuntimeError("injected: ncclBroadcast failed")
            return None  # (rank 0 believes its part went through)
    bad = parallel.ColumnShare(dist, rank, world)
    mcb = MCSamples(samples=s, weights=w, names=names, ranges=ranges, _context_factory=Broken, column_share=bad)
    assert bad.comm is None and bad.bytes_uploaded == s.nbytes and mcb.ctx.comm_world == 0
    assert np.array_equal(mcb.ctx.s[:, :s.shape[1]], s)
    share = parallel.ColumnShare(dist, rank, world)
    mc = MCSamples(samples=s, weights=w, names=names, ranges=ranges, _context_factory=CommCtx, column_share=share)
    first = parallel.column_blocks(s.shape[1], world)
    # this rank sent only its own block over "PCIe" ...
    assert share.comm is not None and share.bytes_uploaded == len(s) * int(first[rank + 1] - first[rank]) * 8
    # ... and holds the whole set afterwards, bit for bit
    assert np.array_equal(mc.ctx.s[:, :s.shape[1]], s)
    ref = MCSamples(samples=s, weights=w, names=names, ranges=ranges, _context_factory=FakeContext)
    assert np.array_equal(mc.means, ref.means) and np.array_equal(mc.fullcov, ref.fullcov)
    d, r = mc.get2DDensity(names[0], names[1]), ref.get2DDensity(names[0], names[1])
    assert np.array_equal(d.P, r.P)
    # the convergence configuration through the library communicator: one chain per rank
    chain = s[rank::world][:, :3]
    wc = w[rank::world]
    mc1 = MCSamples(samples=chain, weights=wc, names=names[:3], _context_factory=CommCtx)
    mc1.ctx.comm_init(world, rank, b"y" * 128)
    out = parallel.convergence_chain_per_rank(mc1, comm=parallel.LibraryComm(mc1.ctx))
    out_t = parallel.convergence_chain_per_rank(mc1, dist=dist)
    assert np.array_equal(out["D"], out_t["D"]) and np.array_equal(out["meanvar"], out_t["meanvar"]) and out["total_norm"] == out_t["total_norm"]
    dist.barrier()
    print("rank", rank, "ok")
""")


def test_column_shards_and_chain_statistics_over_the_library_communicator_gloo_world2(tmp_path):
    """Sample distribution for multi-rank jobs (parallel.ColumnShare: every rank uploads its block of columns only and the
    ranks broadcast their blocks to one another -- gd_upload_shard / gd_comm_share_columns in the C ABI) leaves every rank
    with the full set bit for bit, and the convergence configuration's all-gather (chains.py:1446-1474 per-chain moments)
    runs through the library communicator with the same result as through torch.distributed."""
    script = tmp_path / "share_worker.py"
    script.write_text(SHARE_WORKER % dict(root=ROOT))
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


def test_partition_by_column_blocks_covers_every_pair_once_and_touches_few_columns():
    """parallel.partition_pairs_by_column_blocks: whole tiles of the triangle go to one rank, so a rank's pairs touch a
    fraction of the columns (what it pre-bins), every pair is dealt exactly once, and the shares stay balanced."""
    from getdist_amd import parallel, synth

    n = 50
    pairs = synth.triangle_pairs(n)
    rng = np.random.default_rng(0)
    klass = rng.choice([0, 1, 10, 11, 21, 101], size=len(pairs), p=[0.55, 0.25, 0.08, 0.06, 0.04, 0.02])
    for world in (2, 4, 8):
        seen, touched, sizes = [], [], []
        for rank in range(world):
            mine, mp = parallel.partition_pairs_by_column_blocks(pairs, klass, world, rank, n)
            assert mp == [pairs[i] for i in mine]
            seen += mine
            touched.append(len({c for pr in mp for c in pr}))
            sizes.append(len(mine))
        assert sorted(seen) == list(range(len(pairs)))
        assert max(sizes) <= 1.35 * len(pairs) / world, (world, sizes)
        if world >= 4:
            assert max(touched) <= (0.9 if world == 4 else 0.65) * n, (world, touched)
    # a triangle too small to tile falls back to the class deal
    small = synth.triangle_pairs(4)
    got = sorted(i for r in range(2) for i in parallel.partition_pairs_by_column_blocks(small, [0] * len(small), 2, r, 4)[0])
    assert got == list(range(len(small)))
