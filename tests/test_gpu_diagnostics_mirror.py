"""exmc_amd.diagnostics (the host mirror of Exmc.Diagnostics) over a sampled trace: host and
device inputs agree, per-series values equal the checker's, the summary is sane."""
import numpy as np
import pytest
import torch

import oracle as O
from exmc_amd import diagnostics, models, sampler

pytestmark = pytest.mark.gpu


def test_diagnostics_mirror_on_a_sampled_trace(hip):
    spec = models.eight_schools()
    comp = sampler.compile(spec)
    opts = dict(num_warmup=200, num_samples=300, seed=8, init_values=spec.default_init)
    traces, stats = sampler.sample_chains_compiled(comp, 12, opts)
    raw = stats[0]["extra"]["raw"]["draws"]                       # [C][S][d]
    e_host = diagnostics.ess(comp, raw)
    dev = torch.from_numpy(np.ascontiguousarray(raw.transpose(1, 2, 0))).to("cuda:0")
    assert np.array_equal(e_host, diagnostics.ess(comp, dev))
    eb = diagnostics.ess_bulk(comp, raw)
    rh = diagnostics.rhat(comp, raw)
    L = O.lib()
    for i in range(spec.d):
        for c in (0, 5, 11):
            series = np.ascontiguousarray(raw[c, :, i])
            assert L.exo_ess(O.dptr(series), 300) == e_host[i, c]
            assert L.exo_ess_bulk_mode(O.dptr(series), 300, 1) == eb[i, c]
        chains = np.ascontiguousarray(raw[:, :, i])
        assert L.exo_rhat(O.dptr(chains), 12, 300) == rh[i]
    cons = spec.constrain(raw)
    summ = diagnostics.summary(comp, cons)
    assert set(summ) == set(spec.var_names)
    # Diagnostics.summary's key set and conventions (diagnostics.ex:14-34): divisor-n std, linear
    # interpolation quantiles h = (n - 1) p
    for i, name in enumerate(spec.var_names):
        v = summ[name]
        assert set(v) == {"mean", "std", "q5", "q25", "q50", "q75", "q95", "ess", "ess_bulk", "rhat"}
        flat = cons[:, :, i].ravel()
        assert abs(v["std"] - np.std(flat)) <= 1e-12 * (1 + np.std(flat))        # ddof = 0
        assert abs(v["mean"] - np.mean(flat)) <= 1e-12 * (1 + abs(np.mean(flat)))
        for key, pr in (("q5", 0.05), ("q25", 0.25), ("q50", 0.5), ("q75", 0.75), ("q95", 0.95)):
            assert abs(v[key] - np.quantile(flat, pr)) <= 1e-12 * (1 + abs(v[key]))
    # a trace that is a non-contiguous view (torch makes the copy on its own stream)
    view = dev.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    assert not view.is_contiguous()
    assert np.array_equal(e_host, diagnostics.ess(comp, view))
    assert 0.0 < summ["tau"]["q5"] < summ["tau"]["q50"] < summ["tau"]["q95"]
    assert all(0.95 < v["rhat"] < 1.1 and v["ess"] > 100 for v in summ.values())
