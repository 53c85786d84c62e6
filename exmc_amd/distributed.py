"""Chain sharding across the GPUs of one node.

Chains are independent given the shared tuning (sampler.ex:1082-1130), so the sampling loop has
no collective: rank r owns the contiguous chain block [r*C/N, (r+1)*C/N) and chain i keeps seed
base + 7919*i whatever the shard. The only exchange is after sampling: one all-gather of the
finished traces (RCCL over xGMI on the GPUs; gloo in the CPU tests) for split R-hat, and one
all-reduce of per-parameter ESS sums. This replaces Exmc.NUTS.Distributed's :erpc fan-out
(lib/exmc/nuts/distributed.ex:56-101).
"""
import os

import numpy as np
import torch


def shard_range(n_chains_total, rank, world):
    """Contiguous block of chain indices for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_chains_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def _skip(dist, force):
    """No collective to make: no process group, or a group of one rank (unless `force`: a one-rank
    group still executes every collective on the real backend -- how the RCCL calls are exercised on
    a one-GPU box, tests/test_gpu_rccl_one_rank.py, `bench.py --force-dist`)."""
    return dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force)


def gather_traces(draws_local, dist=None, force=False):
    """All-gather [S][D][C_local] trace blocks into [S][D][C_total] in chain order.
    Every rank must hold the same C_local (pad the last shard otherwise)."""
    if _skip(dist, force):
        return draws_local
    world = dist.get_world_size()
    S, D, Cl = draws_local.shape
    src = draws_local.contiguous()
    if dist.get_backend() == "gloo":
        # gloo has no all_gather_into_tensor; the list form is the same exchange
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src)
        gathered = torch.stack(parts, dim=0)
    else:
        gathered = torch.empty((world, S, D, Cl), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(gathered, src)   # RCCL: one collective over xGMI; errors propagate
    return gathered.permute(1, 2, 0, 3).reshape(S, D, world * Cl)


def reduce_sum(t, dist=None, force=False):
    if not _skip(dist, force):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def reduce_max(t, dist=None, force=False):
    if not _skip(dist, force):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def half_chain_stats(draws):
    """Sufficient statistics of split R-hat for the local chains: mean and unbiased variance of
    each half of each chain, [2][D][C_local] each (half 0 = draws [0, n), half 1 = [mid, mid+n))."""
    S = draws.shape[0]
    mid = S // 2
    n = min(mid, S - mid)
    a, b = draws[:n], draws[mid:mid + n]
    means = torch.stack([a.mean(dim=0), b.mean(dim=0)], dim=0)
    variances = torch.stack([a.var(dim=0, unbiased=True), b.var(dim=0, unbiased=True)], dim=0)
    return means, variances, n


def gather_chain_stats(stat_local, dist=None, force=False):
    """All-gather [2][D][C_local] per-chain statistics into [2][D][C_total] in chain order: a few
    hundred KB instead of the [S][D][C] traces (2.6 GB per rank at 8 x 4096 x 1000 x 10)."""
    if _skip(dist, force):
        return stat_local
    world = dist.get_world_size()
    src = stat_local.contiguous()
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src)
    return torch.cat(parts, dim=2)


def split_rhat_from_stats(means, variances, n):
    """split_rhat from half-chain statistics [2][D][C]: the same B/W ratio, the 2C half-chains
    taken in the order split_rhat uses (all first halves, then all second halves)."""
    mh = torch.cat([means[0], means[1]], dim=1)        # [D][2C]
    vh = torch.cat([variances[0], variances[1]], dim=1)
    m = mh.shape[1]
    gm = mh.mean(dim=1, keepdim=True)
    b = n / (m - 1) * ((mh - gm) ** 2).sum(dim=1)
    w = vh.mean(dim=1)
    return torch.sqrt(((n - 1) / n * w + b / n) / w)


def split_rhat(draws):
    """Exmc.Diagnostics.rhat (lib/exmc/diagnostics.ex:80-115) per parameter on a [S][D][C]
    trace: each chain split in half, B/W variance ratio."""
    S = draws.shape[0]
    mid = S // 2
    n = min(mid, S - mid)
    halves = torch.cat([draws[:n], draws[mid:mid + n]], dim=2)   # [n][D][2C]
    m = halves.shape[2]
    means = halves.mean(dim=0)
    var = halves.var(dim=0, unbiased=True)
    gm = means.mean(dim=1, keepdim=True)
    b = n / (m - 1) * ((means - gm) ** 2).sum(dim=1)
    w = var.mean(dim=1)
    return torch.sqrt(((n - 1) / n * w + b / n) / w)


# ------------------------------------------------------------------------------------------
# The fan-out as an API: Exmc.NUTS.Distributed.sample_chains/2 (lib/exmc/nuts/distributed.ex:56-101)
# re-designed for one node of GPUs. The reference compiles and warms up on the coordinator, ships
# the tuning to peer nodes over :erpc and retries failed chains locally; here every worker process
# owns one GPU, repeats the (deterministic) shared warmup with the same seed instead of receiving
# it, samples its contiguous block of chains and hands its finished traces back -- no exchange
# while sampling. `sampler.sample_chains(ir, n, devices=[0, 1, ...])` routes here.
# ------------------------------------------------------------------------------------------
def run_shard(engine, spec, num_chains, opts, device, rank, world):
    """What one rank does (also the body bench.py's ranks follow): compile on `device`, the
    shared warmup (every rank with the same seed: identical tuning, no broadcast), then chains
    [lo, hi) of num_chains with seeds seed + 7919 * i. Returns the rank's traces: host arrays
    [C][S][...], or with opts["raw_on_device"] the device buffers themselves ([S][...][C] tensors)."""
    o = dict(opts or {})
    o["device"] = device
    lo, hi = shard_range(num_chains, rank, world)
    compiled = engine.compile(spec, {"device": device})
    init_values = o.get("init_values") or {}
    tuning = engine.warmup(compiled, init_values, o)
    if hi > lo:
        _, stats, extra = engine.sample_compiled_tuned(compiled, tuning, init_values, o,
                                                       num_chains=num_chains, chain_lo=lo, chain_hi=hi)
        raw = extra["raw"]
        leapfrogs = int(extra["total_leapfrogs"])
    else:           # more ranks than chains: an empty shard
        raw, leapfrogs = None, 0
    out = dict(rank=rank, lo=lo, hi=hi, raw=raw, leapfrogs=leapfrogs, epsilon=float(tuning["epsilon"]),
               inv_mass=np.asarray(tuning.get("inv_mass_diag", tuning["inv_mass"]), dtype=np.float64))
    if tuning.get("chol_cov") is not None:      # opts["dense_mass"]
        out["cov"], out["chol_cov"] = np.asarray(tuning["cov"]), np.asarray(tuning["chol_cov"])
    return out


_TRACE_KEYS = ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy")


def _gather_shards_to_root(local, cmax, dist, device, rank, world, counts):
    """Collect every rank's trace arrays on rank 0, one key at a time: `local[k]` is this rank's
    block in the device layout the kernels write -- torch tensors [S][...][C_local] on `device`, straight
    from the sampler's buffers (no host round trip before the collective) -- padded to cmax chains,
    gathered to rank 0 over the process group (RCCL between GPUs; gloo moves host copies), and turned
    into the host layout [C_total][S][...] there, rank block by rank block. Only rank 0 ever holds more
    than its own shard, and on the device never more than world x ONE key at a time; the other ranks
    get None."""
    import torch
    out = {} if rank == 0 else None
    use_host = dist.get_backend() == "gloo"
    for k in _TRACE_KEYS:
        a = local[k]
        if a.shape[-1] != cmax:
            pad = torch.zeros(a.shape[:-1] + (cmax,), dtype=a.dtype, device=a.device)
            pad[..., :a.shape[-1]] = a
            a = pad
        src = (a.cpu() if use_host else a).contiguous()
        parts = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
        dist.gather(src, gather_list=parts, dst=0)
        if rank == 0:
            blocks = []
            for r in range(world):
                h = parts[r][..., :counts[r]].cpu().numpy()          # [S][...][C_r]
                blocks.append(np.ascontiguousarray(np.moveaxis(h, -1, 0)))   # [C_r][S][...]
                parts[r] = None                                      # free the device block as we go
            out[k] = np.concatenate(blocks, axis=0)
        del src, parts
    return out


def _device_layout(raw, device):
    """Host arrays [C][S][...] (an engine without device buffers: the CPU checker) as tensors in the
    device layout [S][...][C]."""
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(np.moveaxis(np.asarray(raw[k]), 0, -1))).to(device)
            for k in _TRACE_KEYS}


def _shard_worker(rank, world, engine_name, spec, num_chains, opts, devices, port, backend, queue, received):
    import importlib

    import torch
    import torch.distributed as dist
    engine = importlib.import_module(engine_name)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the parent holds the rendezvous store on a port it bound itself and still owns
    store = dist.TCPStore("127.0.0.1", port, world, is_master=False)
    device = torch.device("cpu")
    on_gpu = engine_name == "exmc_amd.sampler"
    if on_gpu:
        torch.cuda.set_device(devices[rank])
        device = torch.device("cuda", devices[rank])
    if backend == "nccl":
        dist.init_process_group("nccl", store=store, rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group("gloo", store=store, rank=rank, world_size=world)
    try:
        o = dict(opts or {})
        o["device"] = devices[rank]
        o["raw_on_device"] = on_gpu        # the sampler leaves the finished trace in HBM ([S][...][C])
        res = run_shard(engine, spec, num_chains, o, devices[rank], rank, world)
        S = int(engine._merge_opts(opts)["num_samples"])
        counts = [shard_range(num_chains, r, world)[1] - shard_range(num_chains, r, world)[0] for r in range(world)]
        cmax = max(counts)
        local = res["raw"]
        if local is None:           # an empty shard still takes part in the collective
            local = dict(draws=np.zeros((0, S, spec.d)), logp=np.zeros((0, S)), tree_depth=np.zeros((0, S), np.int32),
                         n_steps=np.zeros((0, S), np.int32), divergent=np.zeros((0, S), np.int32),
                         accept_prob=np.zeros((0, S)), energy=np.zeros((0, S)))
        if not isinstance(local["draws"], torch.Tensor):
            local = _device_layout(local, device)
        allraw = _gather_shards_to_root(local, cmax, dist, device, rank, world, counts)
        # the shared tuning, cross-checked where it was computed: every rank must have derived the same
        tun = np.concatenate([[res["epsilon"]], res["inv_mass"].ravel(),
                              np.asarray(res.get("cov", np.zeros(0)), dtype=np.float64).ravel(),
                              np.asarray(res.get("chol_cov", np.zeros(0)), dtype=np.float64).ravel()])
        t = torch.from_numpy(tun).to(device if backend == "nccl" else "cpu")
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        same = all(torch.equal(parts[0], p_) for p_ in parts)
        lf = torch.tensor([float(res["leapfrogs"])], dtype=torch.float64, device=t.device)
        dist.all_reduce(lf, op=dist.ReduceOp.SUM)
        if rank == 0:
            # the traces travel as shared-memory tensors (torch.multiprocessing hands over the
            # segments, nothing is pickled or written to a file); this process must outlive the hand-over
            shared = {k: torch.from_numpy(v).share_memory_() for k, v in allraw.items()}
            queue.put(dict(raw=shared, same_tuning=bool(same), leapfrogs=int(lf.item()), epsilon=res["epsilon"],
                           inv_mass=res["inv_mass"], cov=res.get("cov"), chol_cov=res.get("chol_cov")))
            received.wait(timeout=600.0)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def sample_chains_sharded(spec, num_chains, opts=None, devices=None, engine="exmc_amd.sampler"):
    """sample_chains over several GPUs of one node: one spawned process per entry of `devices`
    (started before it touches a GPU; the parent needs none), chain blocks by shard_range, the
    finished traces gathered from the ranks' device buffers to rank 0 over their process group (RCCL
    when every rank has a GPU of its own, gloo otherwise -- the CPU tests, or several ranks sharing
    one GPU) and handed to the parent as shared-memory tensors: no file is written, nothing is
    pickled. The returned ({name: draws}[], stats[]) equal the single-device sample_chains whatever
    the number of devices. `engine` names the module that provides compile / warmup /
    sample_compiled_tuned (the tests substitute the CPU checker). A rank OTHER than the coordinator's
    (rank 0, devices[0]) that fails ends the fan-out -- the gather is a collective, so its peers cannot
    finish without it -- and the chains are run again on the coordinator's device alone, as the
    reference retries a failed chain on the coordinator (distributed.ex:158-180: `catch kind, reason ->
    ... run_chain_local`); chain i keeps its seed, so the answer is the one the ranks would have
    given, and stats[*]["extra"]["retried_on_coordinator"] names the rank that failed and why.
    opts["retry_on_coordinator"] = False turns that off (the failure raises). A failure of rank 0
    itself raises (there is no other coordinator to fall back to), and so does a rank that neither
    finishes nor fails within opts["shard_timeout_s"] (default six hours): which rank hangs is not
    known, its device may be the coordinator's, so the processes are terminated and the call raises."""
    import importlib
    import pickle
    import time

    import torch
    import torch.distributed as tdist
    import torch.multiprocessing as mp
    if num_chains < 1:
        raise ValueError("num_chains must be >= 1")
    devices = list(devices) if devices is not None else [0]
    world = len(devices)
    if world < 1:
        raise ValueError("devices must name at least one GPU")
    eng = importlib.import_module(engine)
    o = eng._merge_opts(opts)
    if world == 1:
        res = run_shard(eng, spec, num_chains, opts, devices[0], 0, 1)
        raw, total_lf, first = res["raw"], res["leapfrogs"], res
        shards = [(res["lo"], res["hi"])]
    else:
        try:
            pickle.dumps((spec, opts))
        except Exception as e:   # e.g. a spec that still holds a Custom distribution's closure
            raise ValueError("the model spec / opts cannot be sent to the rank processes: %s" % e) from e
        # the rendezvous store lives in this process on a port the kernel picked for it and that stays
        # bound for the whole call (no bind-then-close window for another process to take)
        store = tdist.TCPStore("127.0.0.1", 0, world, is_master=True, wait_for_workers=False)
        gpu_each = (engine == "exmc_amd.sampler" and len(set(devices)) == world and
                    torch.cuda.device_count() >= world)
        backend = "nccl" if gpu_each else "gloo"
        ctx = mp.get_context("spawn")
        queue = ctx.SimpleQueue()
        received = ctx.Event()
        procs = mp.spawn(_shard_worker, args=(world, engine, spec, num_chains, opts, devices, store.port, backend,
                                              queue, received),
                         nprocs=world, join=False)
        deadline = time.monotonic() + float(o.get("shard_timeout_s", 6 * 3600.0))

        def stop_ranks():
            """End the rank processes (never re-exec): SIGTERM, a short join, SIGKILL for whatever is
            left -- a rank blocked in a driver call or a collective can sit out the first signal -- and
            a last join so that no child stays unreaped."""
            alive = [p_ for p_ in procs.processes if p_.is_alive()]
            for p_ in alive:
                p_.terminate()
            for p_ in alive:
                p_.join(timeout=5.0)
            for p_ in alive:
                if p_.is_alive():
                    p_.kill()
            for p_ in alive:
                p_.join(timeout=5.0)

        def expired():
            if time.monotonic() > deadline:
                stop_ranks()
                raise TimeoutError("sample_chains_sharded: the rank processes did not finish within "
                                   "shard_timeout_s; they have been terminated")
        first = None
        try:
            try:
                while first is None:
                    if not queue.empty():
                        first = queue.get()
                        first["raw"] = {k: v.numpy().copy() for k, v in first["raw"].items()}
                    elif procs.join(timeout=0.05):      # every rank has exited (join raises if one failed)
                        if queue.empty():
                            raise RuntimeError("the rank processes ended without a result")
                    expired()
            finally:
                received.set()
            while not procs.join(timeout=0.05):
                expired()
        except BaseException as e:
            stop_ranks()     # a failed rank (join raised) must not leave its peers holding their GPUs
            # the reference's fallback (distributed.ex:172-180): a failed peer's chains run on the coordinator.
            # torch's spawn says which rank raised or died (error_index); rank 0 IS the coordinator.
            failed_rank = getattr(e, "error_index", None)
            if not (isinstance(e, Exception) and not isinstance(e, TimeoutError) and failed_rank not in (None, 0)
                    and o.get("retry_on_coordinator", True)):
                raise
            retried = dict(rank=int(failed_rank), device=devices[failed_rank],
                           error=("%s: %s" % (type(e).__name__, e))[:2000])
            import logging
            logging.getLogger("exmc_amd.distributed").warning(
                "rank %d (device %r) failed (%s); retrying all %d chains on the coordinator's device %r",
                retried["rank"], retried["device"], type(e).__name__, num_chains, devices[0])
            first = None
        finally:
            del store        # the rendezvous store goes with the call, whichever way it ends
        if first is None:    # the retry: one shard, the coordinator's device, this process
            res = run_shard(eng, spec, num_chains, opts, devices[0], 0, 1)
            raw, total_lf, first = res["raw"], res["leapfrogs"], res
            shards = [(res["lo"], res["hi"])]
        else:
            retried = None
            if not first["same_tuning"]:
                raise RuntimeError("ranks disagree on the shared tuning: the warmup is not deterministic")
            shards = [shard_range(num_chains, r, world) for r in range(world)]
            raw = first["raw"]
            total_lf = first["leapfrogs"]
    traces, stats = [], []
    for c in range(num_chains):
        traces.append(eng._build_trace(spec, raw["draws"][c]))
        st = dict(step_size=float(first["epsilon"]), inv_mass_diag=np.array(first["inv_mass"], copy=True),
                  divergences=int(raw["divergent"][c].sum()), num_warmup=o["num_warmup"],
                  num_samples=o["num_samples"], sample_stats=eng.SampleStats(raw, c))
        if first.get("cov") is not None:      # opts["dense_mass"]: as the single-device path returns it
            st["cov"], st["chol_cov"] = first["cov"], first["chol_cov"]
        stats.append(st)
    extra = dict(total_leapfrogs=int(total_lf), raw=raw, shards=[(int(a), int(b)) for a, b in shards],
                 devices=devices)
    if world > 1 and retried is not None:
        extra["retried_on_coordinator"] = retried
    for s_ in stats:
        s_["extra"] = extra
    return traces, stats
