"""Chain sharding across the GPUs of one node.

Chains are independent given the shared tuning (sampler.ex:1082-1130), so the sampling loop has
no collective: rank r owns the contiguous chain block [r*C/N, (r+1)*C/N) and chain i keeps seed
base + 7919*i whatever the shard. The only exchange is after sampling: one all-gather of the
finished traces (RCCL over xGMI on the GPUs; gloo in the CPU tests) for split R-hat, and one
all-reduce of per-parameter ESS sums. This replaces Exmc.NUTS.Distributed's :erpc fan-out
(lib/exmc/nuts/distributed.ex:56-101).
"""
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
    gathered = torch.empty((world, S, D, Cl), dtype=src.dtype, device=src.device)
    try:
        dist.all_gather_into_tensor(gathered, src)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src)
        gathered = torch.stack(parts, dim=0)
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
