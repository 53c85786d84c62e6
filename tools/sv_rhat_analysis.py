"""Why is split R-hat > 1.05 for sv at the 1000-draw protocol? Samples sv (2048 chains x 1000
draws after the shared warmup, as bench.py does), then reports per-parameter split R-hat of the
full trace, of the trace without its first 200 / 500 draws, and the per-chain ESS of the worst
parameters. Run on the GPU box: python tools/sv_rhat_analysis.py > gpurun_out/<tag>/sv_rhat.txt"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from exmc_amd import _lib, sampler  # noqa: E402
from exmc_amd import distributed as xd  # noqa: E402

spec, _ = bench.make_spec("sv")
comp = sampler.compile(spec)
Cn, S, d = 2048, 1000, spec.d
opts = sampler._merge_opts(dict(num_warmup=1000, num_samples=S, seed=42))
tuning = sampler.warmup(comp, spec.default_init, opts)
print("eps %.5f  inv_mass min/median/max %.3g %.3g %.3g" % (tuning["epsilon"], tuning["inv_mass"].min(),
      np.median(tuning["inv_mass"]), tuning["inv_mass"].max()))
tun = sampler._tuning_struct(tuning, d)
dev = torch.device("cuda:0")
draws = torch.empty((S, d, Cn), dtype=torch.float64, device=dev)
depth = torch.empty((S, Cn), dtype=torch.int32, device=dev)
tr = _lib.Trace(draws.data_ptr(), None, depth.data_ptr(), None, None, None, None)
iq = np.ascontiguousarray(spec.to_unconstrained(spec.default_init))
lf, dv = C.c_int64(), C.c_int32()
comp.check(comp.L.exmc_hip_sample_chains(comp.h, C.byref(tun), iq.ctypes.data_as(C.POINTER(C.c_double)),
                                         Cn, 0, Cn, sampler._c_opts(opts), tr, C.byref(lf), C.byref(dv)))
torch.cuda.synchronize()
names = spec.var_names
for skip in (0, 200, 500):
    r = xd.split_rhat(draws[skip:]).cpu().numpy()
    order = np.argsort(-r)[:4]
    print("skip %3d draws: rhat max %.4f, > 1.05 on %d of %d parameters; worst: %s"
          % (skip, r.max(), int((r > 1.05).sum()), d, ", ".join("%s %.3f" % (names[i], r[i]) for i in order)))
x = draws.cpu().numpy()
for name in ("sigma", "nu", "s_1", "s_50"):
    i = names.index(name)
    m = x[:, i, :]
    print("%-6s chain-mean of draws   0-100: %.3f  100-200: %.3f  500-1000: %.3f   (init %.3f); between-chain sd of chain means %.3f, within sd %.3f"
          % (name, m[:100].mean(), m[100:200].mean(), m[500:].mean(), iq[i], m[500:].mean(axis=0).std(), m[500:].std(axis=0).mean()))
print("mean tree depth %.2f, divergent %d" % (depth.float().mean().item(), dv.value))
