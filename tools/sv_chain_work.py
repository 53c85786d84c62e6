"""development: spread of the per-chain leapfrog totals of sv (what the launch's tail is made of)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from exmc_amd import sampler
spec = bench.make_spec("sv")[0]
comp = sampler.compile(spec)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
opts = dict(num_warmup=1000, num_samples=S, seed=42, lanes_per_chain=64)
tuning = sampler.warmup(comp, spec.default_init, opts)
_, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=2048)
n = extra["raw"]["n_steps"].astype(np.int64)          # [C][S]
tot = n.sum(axis=1)
print("chains", n.shape, "mean per chain %.0f" % tot.mean(), "min %.0f max %.0f" % (tot.min(), tot.max()),
      "max/mean %.3f" % (tot.max() / tot.mean()), "p99/mean %.3f" % (np.percentile(tot, 99) / tot.mean()),
      "p90/mean %.3f" % (np.percentile(tot, 90) / tot.mean()), "std/mean %.3f" % (tot.std() / tot.mean()))
# how persistent is a chain's cost: correlation of the first and second half totals
h = S // 2
a, b = n[:, :h].sum(axis=1), n[:, h:].sum(axis=1)
print("corr(first half, second half) %.3f" % np.corrcoef(a, b)[0, 1])
q = max(S // 10, 1)
first = n[:, :q].sum(axis=1)
rest = n[:, q:].sum(axis=1)
print("corr(first tenth, rest) %.3f" % np.corrcoef(first, rest)[0, 1])
