"""
Multi-GPU decomposition of the batched 2D-density path: one process per GPU, every rank
holds a full replica of the sample columns, the per-parameter preparation is split round-robin over ranks and
its ~12 scalars per parameter are all-gathered (the only collective: RCCL when the backend is "nccl", gloo in
the CPU tests), and the independent parameter pairs are partitioned by cost class with no data-path collective.
"""

import numpy as np

PARAM_STATE = ("err", "mean", "param_min", "param_max", "range_min", "range_max", "sigma_range", "has_limits_bot",
               "has_limits_top", "has_limits", "N_eff_kde")


def partition_round_robin(items, world, rank):
    return [it for i, it in enumerate(items) if i % world == rank]


def partition_pairs(pairs, cost_key, world, rank):
    """
    Deal pairs to ranks so every rank gets the same mix of cost classes: sort by class, then round-robin.
    ``cost_key(pair)`` returns a sortable class id (e.g. (F, bounded?, branch)).  Returns (indices, pairs).
    """
    order = sorted(range(len(pairs)), key=lambda i: (cost_key(pairs[i]), i))
    mine = [order[i] for i in range(len(order)) if i % world == rank]
    mine.sort()
    return mine, [pairs[i] for i in mine]


def partition_pairs_by_class(pairs, class_ids, world, rank):
    """partition_pairs for integer class ids given as an array (one per pair): the same deal, without a Python call per
    pair -- at 8 ranks the dealing of a 1225-pair triangle is otherwise a visible part of a rank's step."""
    class_ids = np.asarray(class_ids)
    order = np.argsort(class_ids, kind="stable")
    mine = np.sort(order[rank::world]).tolist()
    return mine, [pairs[i] for i in mine]


class NeffShare:
    """
    Who computes which parameter's KDE effective sample number in a multi-rank run.  Set as ``mc._neff_share``:
    ``MCSamples._neff_batch`` then computes only the parameters in ``params`` (this rank's share) -- on the helper thread,
    beside the 2D binning, exactly as in the single-GPU pipeline -- and ``exchange(mc)``, called from the main thread once
    that has finished (``_neff_complete``), must leave ``N_eff_kde`` of every other parameter filled in (an all-gather of
    one double per parameter).  Every collective of a process is issued from its main thread.
    """

    def __init__(self, params, exchange):
        self.params = set(params)
        self.exchange = exchange
        self.exchanged = False  # the collective is entered exactly once per step (complete())
        self.library_comm = False  # True: gd_density2d_batch exchanges the values itself over the context's communicator

    def complete(self, mc):
        """Enter the exchange if the step has not done so yet (a rank without pairs, a call with a fixed smoothing scale,
        a call that raised): every rank issues the same collectives in the same order, whatever its share needed."""
        if not self.exchanged:
            self.exchanged = True
            self.exchange(mc)


class LibraryComm:
    """
    The collectives of a step through the C ABI (gd_comm_* in include/gdhip.h: RCCL on the context's own stream, device
    buffers inside the library) instead of torch.distributed.  The 128-byte RCCL id travels once, at start-up, by whatever
    the host application has -- here torch.distributed's broadcast.
    """

    def __init__(self, ctx):
        self.ctx = ctx

    @property
    def world(self):
        return self.ctx.comm_world

    def allgather(self, vec):
        return self.ctx.comm_allgather(vec)

    def allreduce_sum(self, vec):
        return self.ctx.comm_allreduce_sum(vec)

    def share_columns(self, first_by_rank):
        return self.ctx.comm_share_columns(first_by_rank)


class StageTimeout(RuntimeError):
    pass


from ._lib import GdhipTimeout as CommTimeout  # noqa: E402,F401 -- GD_ERR_TIMEOUT of a gd_comm_* call (communicator dropped)


def call_with_watchdog(fn, seconds, what):
    """Run ``fn()`` on a helper thread and wait at most ``seconds`` for it: the set-up stages of the library communicator
    are collectives, and a rank that never arrives must end in the all-or-nothing fallback, not in a hung job.  (The C ABI
    has its own watchdog around ncclCommInitRank -- GDHIP_COMM_TIMEOUT_S -- which normally fires first; this one also
    covers a binding whose call never returns.)  On a timeout the helper thread is left behind (daemon) and StageTimeout
    is raised."""
    import threading

    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as exc:  # noqa: BLE001 -- delivered to the caller below
            box["error"] = exc

    th = threading.Thread(target=run, name="gdamd-comm-%s" % what, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        exc = StageTimeout("%s did not finish within %.0f s" % (what, seconds))
        exc.thread = th  # still inside fn(): whatever fn works on must not be touched by anybody else
        raise exc
    if "error" in box:
        raise box["error"]
    return box.get("value")


def comm_stage_timeout():
    import os

    try:
        lib_limit = float(os.environ.get("GDHIP_COMM_TIMEOUT_S", "") or 120.0)
    except ValueError:
        lib_limit = 120.0
    # the library's own watchdog must come first: a stage that times out HERE means a call that did not return even after the
    # library gave up on it, and the context it runs on is then treated as lost (mark_stuck).  A value from the environment
    # below the library's limit would let this watchdog fire while the library is still legitimately waiting -- and the
    # fallback would then use a context another thread is still working on -- so it is raised to the library's limit + 10 s
    try:
        return max(float(os.environ["GETDIST_AMD_COMM_STAGE_TIMEOUT_S"]), lib_limit + 10.0)
    except (KeyError, ValueError):
        return lib_limit + 30.0


def mark_stuck(ctx):
    """A gd_comm_* call on ``ctx`` outlived its watchdog: its helper thread may still be inside the C ABI on this context,
    which is documented as not thread-safe.  Tell the library not to install a communicator should that call come back
    (gd_comm_abandon -- the one call allowed from another thread) and flag the context so that no other work is put on it
    (ColumnShare.upload and bench.py raise instead of falling back onto it)."""
    ctx._comm_helper_stuck = True
    abandon = getattr(ctx, "comm_abandon", None)
    if abandon is not None:
        try:
            abandon()
        except Exception:  # noqa: BLE001
            pass


def context_is_stuck(ctx):
    return bool(getattr(ctx, "_comm_helper_stuck", False))


def _everyone(dist, device, flag):
    """True on every rank iff ``flag`` is true on every rank (the agreement step between the stages of the set-up)."""
    import torch

    if device is None and dist.get_backend() == "nccl":  # RCCL reduces device tensors only
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([1 if flag else 0], dtype=torch.int32)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def init_library_comm(ctx, dist, rank, world, device=None, stage_timeout=None):
    """
    Rank 0 makes the RCCL id, every rank joins (collective).  Returns a LibraryComm -- or None ON EVERY RANK when any rank
    cannot take part (librccl not loadable, the communicator not created or not created IN TIME, a test all-reduce with
    the wrong answer or none at all): the ranks agree over torch.distributed after each stage, so that either all of them
    use the library's collectives or all of them keep torch.distributed's, and a half-initialised job neither crashes nor
    deadlocks in a mismatched collective.  Every stage runs under a watchdog (call_with_watchdog): a hang counts as a
    failure of that stage on the rank that saw it.
    """
    import logging

    import torch

    if stage_timeout is None:
        stage_timeout = comm_stage_timeout()
    if device is None and dist.get_backend() == "nccl":  # RCCL reduces device tensors only
        device = torch.device("cuda", torch.cuda.current_device())

    def everyone(flag):
        return _everyone(dist, device, flag)

    def give_up(stage):
        if rank == 0:
            logging.warning("library communicator not available (%s): the step's collectives stay with torch.distributed", stage)
        return None

    def destroy_quietly():
        try:
            call_with_watchdog(ctx.comm_destroy, stage_timeout, "comm_destroy")
        except Exception:  # noqa: BLE001 -- (a communicator whose peers are gone: the library aborts it on its own timeout)
            pass

    try:  # stage 1, local: the RCCL library loads and resolves on this rank (gd_comm_unique_id needs nothing else)
        my_id = call_with_watchdog(ctx.comm_unique_id, stage_timeout, "comm_unique_id")
        ok = True
    except Exception:
        my_id, ok = None, False
    if not everyone(ok):
        return give_up("librccl could not be loaded on a rank")
    box = [my_id if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    hung = False
    try:  # stage 2, collective
        call_with_watchdog(lambda: ctx.comm_init(world, rank, box[0]), stage_timeout, "comm_init")
        ok = True
    except StageTimeout:
        ok, hung = False, True
        mark_stuck(ctx)
    except Exception:
        ok = False
    if not everyone(ok):
        if ok:
            destroy_quietly()
        return give_up("communicator creation failed or timed out on a rank")
    comm = LibraryComm(ctx)
    try:  # stage 3: a sum whose answer every rank knows
        out = call_with_watchdog(lambda: comm.allreduce_sum(np.array([rank + 1.0, 1.0])), stage_timeout, "test all-reduce")
        ok = bool(out[0] == world * (world + 1) / 2 and out[1] == world)
    except StageTimeout:
        ok, hung = False, True
        mark_stuck(ctx)
    except Exception:
        ok = False
    if not everyone(ok):
        if not hung:  # (a context whose helper thread is still inside the library is left alone)
            destroy_quietly()
        return give_up("test all-reduce failed or timed out on a rank")
    return comm


def column_blocks(n, world):
    """first_by_rank (world + 1 entries): contiguous blocks of columns, sizes differing by at most one."""
    base, extra = divmod(int(n), int(world))
    sizes = [base + (1 if r < extra else 0) for r in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


class ColumnShare:
    """
    Sample distribution of a multi-rank job over xGMI instead of W full uploads over PCIe ("broadcast once"):
    rank r uploads only its block of columns (gd_upload_shard; the weights, 1 / n of the data, by every rank) and the
    ranks broadcast their blocks to one another over the library communicator (gd_comm_share_columns: W ncclBroadcast in one
    group on the context's stream).  Pass as ``MCSamples(..., column_share=ColumnShare(dist, rank, world))``: the first
    upload creates the communicator (all-or-nothing, init_library_comm) and keeps it in ``.comm`` for the step's other
    exchanges; when no rank can use the library's collectives every rank uploads the full array as before.  A host process
    then needs only its own columns in memory: ``samples`` may be the full (N, n) array or a callable ``cols(first, last)``
    returning the (N, last - first) block.
    """

    def __init__(self, dist, rank, world, device=None, comm=None):
        self.dist, self.rank, self.world, self.device = dist, int(rank), int(world), device
        self.comm = comm
        self.tried = comm is not None
        self.bytes_uploaded = 0

    def upload(self, ctx, samples, weights, n=None):
        if not self.tried:
            self.tried = True
            self.comm = init_library_comm(ctx, self.dist, self.rank, self.world, self.device)
        if context_is_stuck(ctx):
            # a set-up stage hung on THIS rank beyond the library's own watchdog: its helper thread may still be inside the
            # C ABI on this context (it is not thread-safe), so the fallback upload must not run on it either
            raise StageTimeout("a gd_comm_* set-up call never returned on rank %d: the context cannot be reused" % self.rank)
        full = samples if not callable(samples) else None
        if self.comm is None:
            if full is None:
                nn = int(n)
                full = samples(0, nn)
            ctx.upload(full, weights)
            self.bytes_uploaded = int(np.asarray(full).nbytes)
            return
        if full is not None:
            N, nn = np.shape(full)
        else:
            nn = int(n)
        first = column_blocks(nn, self.world)
        a, b = int(first[self.rank]), int(first[self.rank + 1])
        block = np.asarray(full)[:, a:b] if full is not None else np.asarray(samples(a, b))
        N = block.shape[0]
        # stage 4 of the all-or-nothing set-up (init_library_comm has done 1-3): the broadcasts themselves, under the same
        # watchdog and followed by the same agreement -- a rank whose exchange failed or timed out (the library aborts a
        # communicator it has waited GDHIP_COMM_TIMEOUT_S for) sends every rank back to the full upload, and the step's
        # other exchanges back to torch.distributed
        hung = False
        try:
            ctx.upload_shard(block, N, nn, a, weights)
            call_with_watchdog(lambda: self.comm.share_columns(first), comm_stage_timeout(), "share_columns")
            ok = True
        except StageTimeout:
            ok, hung = False, True
        except Exception:  # noqa: BLE001 -- any failure of this rank is a failure of the stage
            ok = False
        if _everyone(self.dist, self.device, ok):
            self.bytes_uploaded = int(block.shape[0]) * (b - a) * 8
            return
        import logging

        if self.rank == 0:
            logging.warning("column shards could not be exchanged over the library communicator on a rank: every rank uploads "
                            "the full sample array, the step's collectives stay with torch.distributed")
        if hung:
            mark_stuck(ctx)
            raise StageTimeout("gd_comm_share_columns never returned on rank %d: the context cannot be reused" % self.rank)
        try:
            call_with_watchdog(ctx.comm_destroy, comm_stage_timeout(), "comm_destroy")
        except Exception:  # noqa: BLE001 -- (already aborted by the library)
            pass
        self.comm = None
        if full is None:
            full = samples(0, nn)
        ctx.upload(full, weights)
        self.bytes_uploaded = int(np.asarray(full).nbytes)


_TILE_DEALS = {}


def partition_pairs_by_column_blocks(pairs, class_ids, world, rank, n):
    """
    Deal the pairs of a triangle so that a rank touches FEW columns: the columns are cut into G contiguous groups, pairs
    fall into the G (G + 1) / 2 group pairs ("tiles"), and whole tiles are dealt to the ranks, heaviest first to the least
    loaded rank with a preference for the rank that already holds the tile's column groups (weights: the pair count by
    cost class -- a sheared or bounded pair costs more than a plain one).  A rank's pairs then touch a part of the columns: the
    per-rank pre-binning (all n columns at 0.83 ms of a 6.8-ms step at 8 ranks, round 4) shrinks with the world size.
    Falls back to the class deal (partition_pairs_by_class) when the triangle is too small to tile.
    Returns (indices, pairs) like the other partitioners; deterministic, the same on every rank.
    """
    pa = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    P = len(pa)
    if world <= 1 or P == 0:
        return list(range(P)), [pairs[i] for i in range(P)]
    # the deal is a pure function of (pairs, classes, world, n): a job that repeats its triangle (every step of a bench,
    # every refresh of a plot grid) pays the ~0.5 ms of tile bookkeeping once -- it sits between the quantile kernels and the
    # first binning launch of a rank's step
    class_ids = np.ascontiguousarray(class_ids)
    key = (pa.tobytes(), class_ids.tobytes(), int(world), int(n))
    owner_of_pair = _TILE_DEALS.get(key)
    if owner_of_pair is not None:
        if owner_of_pair is False:
            return partition_pairs_by_class(pairs, class_ids, world, rank)
        mine = np.nonzero(owner_of_pair == rank)[0].tolist()
        return mine, [pairs[i] for i in mine]
    if len(_TILE_DEALS) >= 16:
        _TILE_DEALS.clear()
    G = 1
    while G * (G + 1) // 2 < 6 * world and G < n:  # >= 6 tiles per rank: shares within ~3 % of one another
        G += 1
    if G < 2 or n < 2 * G:
        _TILE_DEALS[key] = False
        return partition_pairs_by_class(pairs, class_ids, world, rank)
    group_of = (np.arange(n) * G) // n  # column -> group
    ga, gb = group_of[pa[:, 0]], group_of[pa[:, 1]]
    lo, hi = np.minimum(ga, gb), np.maximum(ga, gb)
    tile = lo * G + hi
    class_ids = np.asarray(class_ids)
    # cost of a pair by class: upscaled grid (x6), sheared / optimiser (x1.5), per bounded parameter (+0.5)
    cost = 1.0 + 5.0 * (class_ids >= 100) + 0.5 * ((class_ids // 10) % 10) + 0.5 * (class_ids % 10 > 0)
    tiles = np.unique(tile)
    tcost = np.array([cost[tile == t].sum() for t in tiles])
    order = np.argsort(-tcost, kind="stable")
    load = np.zeros(world)
    cols_of = [set() for _ in range(world)]
    owner = {}
    for k in order:
        t = int(tiles[k])
        tl, th = divmod(t, G)
        # least loaded rank; ties go to the rank that already holds this tile's column groups
        # (tile counts / affinity weight swept on the C3 and C5 triangles: 6 tiles per rank and 0.2 give 30 of 50 columns
        # at 8 ranks with the heaviest share 3 % above the mean; heavier affinity buys nothing, fewer tiles unbalance)
        best = min(range(world), key=lambda r: (load[r] + 0.2 * tcost[k] * len({tl, th} - cols_of[r]), r))
        owner[t] = best
        load[best] += tcost[k]
        cols_of[best].update((tl, th))
    owner_of_pair = np.array([owner[int(t)] for t in tile])
    _TILE_DEALS[key] = owner_of_pair
    mine = np.nonzero(owner_of_pair == rank)[0].tolist()
    return mine, [pairs[i] for i in mine]


def allgather_neff(mc, my_js, n_params, dist=None, device=None, comm=None):
    """The second, small exchange of a step: N_eff_kde of the parameters each rank owns.  With a library communicator:
    ONE sum all-reduce of n doubles (a rank contributes the values it owns, zero elsewhere) -- the same collective
    gd_density2d_batch issues when it exchanges them itself, so ranks may mix the two routes."""
    names = mc.paramNames.names
    if comm is not None:
        v = np.zeros(n_params)
        for j in my_js:
            if names[j].N_eff_kde is not None:
                v[j] = float(names[j].N_eff_kde)
        out = comm.allreduce_sum(v)
        for j in np.nonzero(out > 0)[0].tolist():
            if names[j].N_eff_kde is None:
                names[j].N_eff_kde = float(out[j])
        return
    mine = np.array([[j, np.nan if names[j].N_eff_kde is None else float(names[j].N_eff_kde)] for j in my_js]).reshape(-1, 2)
    if dist is None or dist.get_world_size() == 1:
        return
    import torch

    world = dist.get_world_size()
    per = (n_params + world - 1) // world
    buf = np.full((per, 2), -1.0)
    buf[:len(mine)] = mine
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    rows = np.concatenate([g.cpu().numpy() for g in gathered])
    for j, v in rows[rows[:, 0] >= 0]:
        names[int(j)].N_eff_kde = None if np.isnan(v) else float(v)


def pack_param_state(mc, js):
    """Rows of PARAM_STATE values for the parameters ``js`` this rank prepared."""
    out = np.zeros((len(js), 1 + len(PARAM_STATE)))
    for r, j in enumerate(js):
        par = mc.paramNames.names[j]
        out[r, 0] = j
        for c, a in enumerate(PARAM_STATE):
            v = getattr(par, a, None)
            out[r, 1 + c] = np.nan if v is None else float(v)
    return out


def unpack_param_state(mc, rows):
    for row in rows:
        par = mc.paramNames.names[int(row[0])]
        for c, a in enumerate(PARAM_STATE):
            v = row[1 + c]
            if a.startswith("has_limits"):
                setattr(par, a, bool(v))
            elif a == "N_eff_kde":
                par.N_eff_kde = None if np.isnan(v) else float(v)
            else:
                setattr(par, a, np.float64(v))
        par._ranges_done = True


def allgather_param_state(mc, my_js, n_params, dist=None, device=None, comm=None):
    """
    All-gather the prepared per-parameter scalars so every rank knows every parameter.
    ``dist`` is torch.distributed (initialised) or None for single-process runs; ``comm`` a LibraryComm (RCCL through
    the C ABI) takes precedence.
    """
    if comm is not None:
        mine = pack_param_state(mc, my_js)
        per = (n_params + comm.world - 1) // comm.world
        buf = np.full((per, mine.shape[1]), -1.0)
        buf[:len(mine)] = mine
        rows = comm.allgather(buf.ravel()).reshape(-1, mine.shape[1])
        unpack_param_state(mc, rows[rows[:, 0] >= 0])
        return
    if dist is None or dist.get_world_size() == 1:
        return  # one rank prepared every parameter: nothing to exchange
    mine = pack_param_state(mc, my_js)
    import torch

    world = dist.get_world_size()
    per = (n_params + world - 1) // world
    buf = np.full((per, mine.shape[1]), -1.0)
    buf[:len(mine)] = mine
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    rows = np.concatenate([g.cpu().numpy() for g in gathered])
    unpack_param_state(mc, rows[rows[:, 0] >= 0])


def allgather_vector(vec, dist=None, device=None, comm=None):
    """All-gather one equal-length fp64 vector per rank -> list of numpy vectors.  Every rank must pass a vector of the
    same length (a rank with nothing to contribute passes its neutral element, see updateBaseStatistics)."""
    if vec is None:
        raise ValueError("allgather_vector: every rank contributes a vector of the common length")
    if comm is not None:
        return list(comm.allgather(np.ascontiguousarray(vec, dtype=np.float64)))
    if dist is None or dist.get_world_size() == 1:
        return [vec]
    import torch

    t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, t)
    return [g.cpu().numpy() for g in gathered]


# ---- convergence config: one chain per GPU ---------------------------------------------------
def gelman_rubin_from_chain_stats(stats, total_means):
    """
    Brooks-Gelman var(mean)/mean(var) eigenvalues in the orthogonalised parameters (chains.py:1446-1474) from
    per-chain (means, cov, norm) triples and the pooled weighted means.  Returns None when the mean covariance is
    not positive definite, as the reference does.
    """
    if len(stats) < 2:
        return None  # the between-chain variance needs at least two chains
    nparam = len(stats[0][0])
    means = np.asarray(total_means)[:nparam]
    meanscov = np.zeros((nparam, nparam))
    meancov = np.zeros((nparam, nparam))
    for cmeans, ccov, _ in stats:
        diff = np.asarray(cmeans) - means
        meanscov += np.outer(diff, diff)
        meancov += ccov
    meanscov /= len(stats) - 1
    meancov /= len(stats)
    w, U = np.linalg.eigh(meancov)
    if np.min(w) > 0:
        U /= np.sqrt(w)
        return np.linalg.eigvalsh(np.dot(U.T, meanscov).dot(U))
    return None


def allgather_chain_stats(local_means, local_cov, local_norm, dist=None, device=None, comm=None):
    """
    Every rank holds ONE chain and has computed its weighted means / covariance / norm on its GPU (gd_cov);
    exchange the n^2+n+1 doubles per rank (RCCL all-gather over xGMI; latency-bound, one fused buffer) and return
    the list of per-chain (means, cov, norm) plus the pooled means  sum_c norm_c mean_c / sum_c norm_c.
    ``comm``: a LibraryComm -- the all-gather runs inside the C ABI (gd_comm_allgather: ncclAllGather on the context's
    stream), so a binding that is not Python has the same path; otherwise torch.distributed.
    """
    n = len(local_means)
    buf = np.concatenate([np.asarray(local_means, dtype=np.float64), np.asarray(local_cov, dtype=np.float64).ravel(),
                          [float(local_norm)]])
    if comm is not None:
        rows = list(np.asarray(comm.allgather(buf)).reshape(comm.world, -1))
    elif dist is None or dist.get_world_size() == 1:
        rows = [buf]
    else:
        import torch

        t = torch.from_numpy(buf)
        if device is not None:
            t = t.to(device)
        gathered = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, t)
        rows = [g.cpu().numpy() for g in gathered]
    stats = [(r[:n], r[n:n + n * n].reshape(n, n), r[-1]) for r in rows]
    tot = sum(st[2] for st in stats)
    pooled = sum(st[2] * st[0] for st in stats) / tot
    return stats, pooled


def convergence_chain_per_rank(mc, dist=None, device=None, nparam=None, comm=None):
    """
    SURVEY.md 8e, convergence configuration: every rank holds ONE chain in ``mc`` (an MCSamples on that rank's GPU).
    The local weighted means / covariance / norm come from one gd_cov launch, the n^2+n+1 doubles per rank are
    all-gathered (RCCL over xGMI: through the library communicator ``comm`` -- gd_comm_allgather, inside the C ABI -- or
    torch.distributed with backend "nccl"), and every rank evaluates the Gelman-Rubin eigenvalues
    (chains.py:1446-1474) and the per-parameter MeanVar statistic (mcsamples.py:964-985) of the pooled set.
    Returns dict(D, R_minus_1, meanvar, pooled_means, total_norm).
    """
    nparam = nparam or mc.paramNames.numNonDerived()
    means, cov, norm = mc.ctx.cov(list(range(nparam)))
    stats, pooled = allgather_chain_stats(means, cov, norm, dist, device, comm)
    D = gelman_rubin_from_chain_stats(stats, pooled)
    total = sum(st[2] for st in stats)
    between = sum((st[0] - pooled) ** 2 for st in stats) / (len(stats) - 1) if len(stats) > 1 else np.zeros(nparam)
    within = sum(np.diag(st[1]) * st[2] for st in stats) / total
    return dict(D=D, R_minus_1=None if D is None else float(np.max(D)), meanvar=np.sqrt(between / within),
                pooled_means=pooled, total_norm=total)
