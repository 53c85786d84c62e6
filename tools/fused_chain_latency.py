"""Round 6: what one call of the fused-chain hook costs on the GPU box (B2', tree.ex:613-653) -- the hook's whole point
is the cost of a dispatch. Per call of exmc_hip_leapfrog_chain_normal_host with K = 32 (tree.ex:515): wall clock over
N calls, for a few d and for small batches; beside it what the reference states for its own backends (one XLA dispatch
~250 us, batched_leapfrog.ex:6; the Vulkan chain shader ~50 us per step amortised at K = 32, i.e. ~1.6 ms per dispatch,
docs/VULKAN_KNOWN_ISSUES.md:108-111).   python tools/fused_chain_latency.py [calls]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exmc_amd import fused_chain  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
out = {"k": 32, "calls": N, "cases": []}
for C, d in ((1, 1), (1, 10), (1, 102), (1, 256), (16, 256), (64, 10)):
    rng = np.random.default_rng(d)
    q, p, im = rng.normal(size=(C, d)), rng.normal(size=(C, d)), np.ones(d)
    if C == 1:
        q, p = q[0], p[0]
    for _ in range(20):
        fused_chain.leapfrog_chain_normal(q, p, im, 32, 0.1, 0.0, 1.0)
    t0 = time.perf_counter()
    for _ in range(N):
        fused_chain.leapfrog_chain_normal(q, p, im, 32, 0.1, 0.0, 1.0)
    us = (time.perf_counter() - t0) / N * 1e6
    out["cases"].append({"n_chains": C, "d": d, "us_per_call": round(us, 2), "us_per_step": round(us / 32, 3),
                         "us_per_chain_step": round(us / 32 / C, 4)})
out["reference_states"] = {"xla_dispatch_us": 250, "vulkan_chain_us_per_step_at_k32": 50,
                           "source": "lib/exmc/nuts/batched_leapfrog.ex:6; docs/VULKAN_KNOWN_ISSUES.md:108-111"}
print(json.dumps(out))
