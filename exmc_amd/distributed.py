"""Chain sharding across the GPUs of one node.

Chains are independent given the shared tuning (sampler.ex:1082-1130), so the sampling loop has
no collective: rank r owns the contiguous chain block [r*C/N, (r+1)*C/N) and chain i keeps seed
base + 7919*i whatever the shard. The only exchange is after sampling: one all-gather of the
finished traces (RCCL over xGMI on the GPUs; gloo in the CPU tests) for split R-hat, and one
all-reduce of per-parameter ESS sums. This replaces Exmc.NUTS.Distributed's :erpc fan-out
(lib/exmc/nuts/distributed.ex:56-101).
"""
import os
import tempfile

import numpy as np
import torch


def shard_range(n_chains_total, rank, world):
    """Contiguous block of chain indices for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_chains_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_traces(draws_local, dist=None):
    """All-gather [S][D][C_local] trace blocks into [S][D][C_total] in chain order.
    Every rank must hold the same C_local (pad the last shard otherwise)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
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


def reduce_sum(t, dist=None):
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def reduce_max(t, dist=None):
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
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


def gather_chain_stats(stat_local, dist=None):
    """All-gather [2][D][C_local] per-chain statistics into [2][D][C_total] in chain order: a few
    hundred KB instead of the [S][D][C] traces (2.6 GB per rank at 8 x 4096 x 1000 x 10)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
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
    [lo, hi) of num_chains with seeds seed + 7919 * i. Returns the rank's host traces."""
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
    return dict(rank=rank, lo=lo, hi=hi, raw=raw, leapfrogs=leapfrogs, epsilon=float(tuning["epsilon"]),
                inv_mass=np.asarray(tuning["inv_mass"], dtype=np.float64))


def _shard_worker(rank, world, engine_name, spec, num_chains, opts, devices, out_dir):
    import importlib
    engine = importlib.import_module(engine_name)
    res = run_shard(engine, spec, num_chains, opts, devices[rank], rank, world)
    arrays = dict(lo=res["lo"], hi=res["hi"], leapfrogs=res["leapfrogs"], epsilon=res["epsilon"],
                  inv_mass=res["inv_mass"])
    if res["raw"] is not None:
        arrays.update({"raw_" + k: v for k, v in res["raw"].items()})
    np.savez(os.path.join(out_dir, "shard%d.npz" % rank), **arrays)


def sample_chains_sharded(spec, num_chains, opts=None, devices=None, engine="exmc_amd.sampler"):
    """sample_chains over several GPUs of one node: one spawned process per entry of `devices`
    (started before it touches a GPU; the parent needs none), chain blocks by shard_range, results
    concatenated in chain order -- so the returned ({name: draws}[], stats[]) equal the
    single-device sample_chains whatever the number of devices. `engine` names the module that
    provides compile / warmup / sample_compiled_tuned (the tests substitute the CPU checker)."""
    import importlib

    import torch.multiprocessing as mp
    if num_chains < 1:
        raise ValueError("num_chains must be >= 1")
    devices = list(devices) if devices is not None else [0]
    world = len(devices)
    if world < 1:
        raise ValueError("devices must name at least one GPU")
    eng = importlib.import_module(engine)
    with tempfile.TemporaryDirectory(prefix="exmc_shards_") as out_dir:
        if world == 1:
            _shard_worker(0, 1, engine, spec, num_chains, opts, devices, out_dir)
        else:
            mp.spawn(_shard_worker, args=(world, engine, spec, num_chains, opts, devices, out_dir),
                     nprocs=world, join=True)
        shards = [dict(np.load(os.path.join(out_dir, "shard%d.npz" % r))) for r in range(world)]
    eps = {float(z["epsilon"]) for z in shards}
    if len(eps) != 1 or any(not np.array_equal(z["inv_mass"], shards[0]["inv_mass"]) for z in shards):
        raise RuntimeError("ranks disagree on the shared tuning: the warmup is not deterministic")
    keys = [k[4:] for k in shards[0] if k.startswith("raw_")]
    filled = [z for z in shards if int(z["hi"]) > int(z["lo"])]
    raw = {k: np.concatenate([z["raw_" + k] for z in filled], axis=0) for k in keys}
    o = eng._merge_opts(opts)
    traces, stats = [], []
    for c in range(num_chains):
        traces.append(eng._build_trace(spec, raw["draws"][c]))
        stats.append(dict(step_size=float(shards[0]["epsilon"]), inv_mass_diag=shards[0]["inv_mass"].copy(),
                          divergences=int(raw["divergent"][c].sum()), num_warmup=o["num_warmup"],
                          num_samples=o["num_samples"], sample_stats=eng.SampleStats(raw, c)))
    extra = dict(total_leapfrogs=int(sum(int(z["leapfrogs"]) for z in shards)), raw=raw,
                 shards=[(int(z["lo"]), int(z["hi"])) for z in shards], devices=devices)
    for s_ in stats:
        s_["extra"] = extra
    return traces, stats
