"""Host-side mirror of Exmc.Diagnostics (lib/exmc/diagnostics.ex) over the device kernels.

    ess(compiled, draws)       -> [d][C]   Geyer initial-positive-sequence ESS   diagnostics.ex:42-52
    ess_bulk(compiled, draws)  -> [d][C]   the same on rank-normalised series    diagnostics.ex:60-72
    rhat(compiled, draws)      -> [d]      split R-hat over the chains           diagnostics.ex:80-115
    summary(compiled, draws, names) -> {name: {mean, std, q5, q50, q95, ess, ess_bulk, rhat}}
                                                                                 diagnostics.ex:14-40

`draws` is the device trace the sampling kernels write, a float64 CUDA tensor [S][d][C], or a host
array [C][S][d] (the layout of the `_host` entry points and of `stats["raw"]["draws"]`), which is
uploaded. The statistics are computed on the GPU by ess_kernel / rhat_kernel in the reference's
summation order; ESS values are per chain, as Diagnostics.ess is per series -- sum over chains for
a pooled figure. torch is used for the device buffers only. No CPU fallback.
"""
import numpy as np


def _device_trace(compiled, draws):
    import torch
    dev = torch.device("cuda", compiled.device)
    if isinstance(draws, torch.Tensor):
        if draws.dtype != torch.float64 or draws.dim() != 3 or not draws.is_cuda:
            raise ValueError("device trace must be a float64 CUDA tensor [S][d][C]")
        return draws.contiguous()
    a = np.asarray(draws, dtype=np.float64)
    if a.ndim != 3 or a.shape[2] != compiled.d:
        raise ValueError("host trace must be [C][S][d]")
    return torch.from_numpy(np.ascontiguousarray(a.transpose(1, 2, 0))).to(dev)


def _per_series(compiled, draws, fn_name):
    import torch
    x = _device_trace(compiled, draws)
    S, d, C = x.shape
    out = torch.empty((d, C), dtype=torch.float64, device=x.device)
    fn = getattr(compiled.L, fn_name)
    compiled.check(fn(compiled.h, x.data_ptr(), S, d, C, out.data_ptr()))
    torch.cuda.synchronize(x.device)
    return out.cpu().numpy()


def ess(compiled, draws):
    return _per_series(compiled, draws, "exmc_hip_ess")


def ess_bulk(compiled, draws):
    return _per_series(compiled, draws, "exmc_hip_ess_bulk")


def rhat(compiled, draws):
    import torch
    x = _device_trace(compiled, draws)
    S, d, C = x.shape
    out = torch.empty((d,), dtype=torch.float64, device=x.device)
    compiled.check(compiled.L.exmc_hip_rhat(compiled.h, x.data_ptr(), S, d, C, out.data_ptr()))
    torch.cuda.synchronize(x.device)
    return out.cpu().numpy()


def summary(compiled, draws, names=None):
    """Diagnostics.summary: per variable mean, std (n - 1), 5 / 50 / 95 % quantiles of the pooled
    draws plus ESS (summed over chains), bulk ESS and split R-hat. `draws` are whatever values the
    caller wants summarised (constrain them first for constrained-space summaries)."""
    import torch
    x = _device_trace(compiled, draws)
    S, d, C = x.shape
    names = list(names) if names is not None else list(compiled.spec.var_names)
    e, eb = ess(compiled, x).sum(axis=1), ess_bulk(compiled, x).sum(axis=1)
    r = rhat(compiled, x) if S >= 4 else np.full(d, np.nan)
    pooled = x.permute(1, 0, 2).reshape(d, S * C)
    qs = torch.quantile(pooled, torch.tensor([0.05, 0.5, 0.95], dtype=torch.float64, device=x.device),
                        dim=1).cpu().numpy()
    mean, std = pooled.mean(dim=1).cpu().numpy(), pooled.std(dim=1).cpu().numpy()
    return {names[i]: dict(mean=float(mean[i]), std=float(std[i]), q5=float(qs[0, i]),
                           q50=float(qs[1, i]), q95=float(qs[2, i]), ess=float(e[i]),
                           ess_bulk=float(eb[i]), rhat=float(r[i])) for i in range(d)}
