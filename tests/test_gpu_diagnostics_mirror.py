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
    summ = diagnostics.summary(comp, spec.constrain(raw))
    assert set(summ) == set(spec.var_names)
    assert 0.0 < summ["tau"]["q5"] < summ["tau"]["q50"] < summ["tau"]["q95"]
    assert all(0.95 < v["rhat"] < 1.1 and v["ess"] > 100 for v in summ.values())
