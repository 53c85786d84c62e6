"""Host-side mirror of Exmc.Diagnostics (lib/exmc/diagnostics.ex) over the device kernels.

    ess(compiled, draws)       -> [d][C]   Geyer initial-positive-sequence ESS   diagnostics.ex:42-52
    ess_bulk(compiled, draws)  -> [d][C]   the same on rank-normalised series    diagnostics.ex:60-72
    rhat(compiled, draws)      -> [d]      split R-hat over the chains           diagnostics.ex:80-115
    summary(compiled, draws, names) -> {name: {mean, std, q5, q25, q50, q75, q95} + ess, ess_bulk,
                                        rhat}                                     diagnostics.ex:14-34

`draws` is the device trace the sampling kernels write, a float64 CUDA tensor [S][d][C], or a host
array [C][S][d] (the layout of the `_host` entry points and of `stats["raw"]["draws"]`), which is
uploaded. The statistics are computed on the GPU by ess_series_kernel / rank_scores_kernel /
rhat_kernel in the reference's summation order; ESS values are per chain, as Diagnostics.ess is per series -- sum over chains for
a pooled figure. torch is used for the device buffers only. No CPU fallback.
"""
import numpy as np


def _device_trace(compiled, draws):
    import torch
    dev = torch.device("cuda", compiled.device)
    if isinstance(draws, torch.Tensor):
        if draws.dtype != torch.float64 or draws.dim() != 3 or not draws.is_cuda:
            raise ValueError("device trace must be a float64 CUDA tensor [S][d][C]")
        if draws.device.index != compiled.device:
            raise ValueError("the trace lives on cuda:%d, the compiled model on cuda:%d"
                             % (draws.device.index, compiled.device))
        return draws.contiguous()
    a = np.asarray(draws, dtype=np.float64)
    if a.ndim != 3 or a.shape[2] != compiled.d:
        raise ValueError("host trace must be [C][S][d]")
    return torch.from_numpy(np.ascontiguousarray(a.transpose(1, 2, 0))).to(dev)


def _ordered_after_torch(x):
    """The library launches on its own non-blocking stream: whatever torch has queued on its
    current stream for this tensor (a copy made by contiguous(), the producer of a tensor the
    caller has just computed, the upload) must have finished before the kernels read it."""
    import torch
    torch.cuda.current_stream(x.device).synchronize()


def _per_series(compiled, draws, fn_name):
    import torch
    x = _device_trace(compiled, draws)
    S, d, C = x.shape
    out = torch.empty((d, C), dtype=torch.float64, device=x.device)
    _ordered_after_torch(x)
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
    _ordered_after_torch(x)
    compiled.check(compiled.L.exmc_hip_rhat(compiled.h, x.data_ptr(), S, d, C, out.data_ptr()))
    torch.cuda.synchronize(x.device)
    return out.cpu().numpy()


def summary(compiled, draws, names=None):
    """Diagnostics.summary (diagnostics.ex:14-34): per variable the mean, the population standard
    deviation (divisor n, as the reference's `variance = sum((x - mean)^2) / n`) and the 5 / 25 /
    50 / 75 / 95 % linear-interpolation quantiles (diagnostics.ex:169-181) of the pooled draws.
    ess (summed over chains), ess_bulk and rhat ride along as extra keys. `draws` are whatever
    values the caller wants summarised (constrain them first for constrained-space summaries)."""
    import torch
    x = _device_trace(compiled, draws)
    S, d, C = x.shape
    names = list(names) if names is not None else list(compiled.spec.var_names)
    e, eb = ess(compiled, x).sum(axis=1), ess_bulk(compiled, x).sum(axis=1)
    r = rhat(compiled, x) if S >= 4 else np.full(d, np.nan)
    pooled = x.permute(1, 0, 2).reshape(d, S * C)
    probs = torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95], dtype=torch.float64, device=x.device)
    qs = torch.quantile(pooled, probs, dim=1, interpolation="linear").cpu().numpy()
    mean = pooled.mean(dim=1).cpu().numpy()
    std = pooled.std(dim=1, unbiased=False).cpu().numpy()
    return {names[i]: dict(mean=float(mean[i]), std=float(std[i]), q5=float(qs[0, i]),
                           q25=float(qs[1, i]), q50=float(qs[2, i]), q75=float(qs[3, i]),
                           q95=float(qs[4, i]), ess=float(e[i]), ess_bulk=float(eb[i]),
                           rhat=float(r[i])) for i in range(d)}
