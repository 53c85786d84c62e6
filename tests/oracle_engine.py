"""The CPU checker behind the interface exmc_amd.distributed expects of an engine (compile /
warmup / sample_compiled_tuned with the signatures of exmc_amd.sampler), so that the multi-process
sharding path of the product runs in the CPU tests, where no GPU exists. Test infrastructure only."""
import numpy as np

import oracle as O
from exmc_amd.sampler import SampleStats, _build_trace, _merge_opts  # noqa: F401  (re-exported)


class Compiled:
    def __init__(self, spec, device=0):
        self.spec, self.device, self.d = spec, device, spec.d
        self.om = O.model_for(spec)


def compile(spec, opts=None):  # noqa: A001
    return Compiled(spec, (opts or {}).get("device", 0))


def _q0(spec, init_values):
    return None if not init_values else spec.to_unconstrained(init_values)


def warmup(compiled, init_values=None, opts=None):
    o = _merge_opts(opts)
    st = O.warmup(compiled.om, _q0(compiled.spec, init_values), num_warmup=o["num_warmup"],
                  max_tree_depth=o["max_tree_depth"], target_accept=o["target_accept"], seed=o["seed"],
                  cfg=O.Cfg(1, 1))
    return dict(epsilon=st.step_size, inv_mass=np.array(st.inv_mass[:compiled.d]), chol_cov=None,
                warmup_divergences=st.divergences)


def sample_compiled_tuned(compiled, tuning, init_values=None, opts=None, num_chains=1, chain_lo=0,
                          chain_hi=None):
    o = _merge_opts(opts)
    chain_hi = num_chains if chain_hi is None else chain_hi
    q0 = _q0(compiled.spec, init_values)
    keys = ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy")
    out = {k: [] for k in keys}
    lf = 0
    for c in range(chain_lo, chain_hi):
        t, st = O.sample_tuned(compiled.om, tuning["epsilon"], tuning["inv_mass"], q0,
                               num_samples=o["num_samples"], max_tree_depth=o["max_tree_depth"],
                               seed=o["seed"] + 7919 * c, cfg=O.Cfg(1, 1))
        for k in keys:
            out[k].append(t[k])
        lf += st.total_leapfrogs
    raw = {k: np.stack(v) for k, v in out.items()}
    return None, None, dict(total_leapfrogs=lf, raw=raw)
