#!/usr/bin/env python3
"""BENCH_r03's sv leg printed split R-hat 6.58 (torch, gathered traces) and 1.19
(chain statistics) for the same launch, and an ESS total (22927.14) that no builder run shows
(22896.21). This repeats the sv leg of the driver's command in ONE process -- the eight_schools
leg first (same allocator history), then N times [5 warm launches, the timed launch] -- and reads
the finished trace through every route several times: the library's ESS and R-hat kernels, torch's
half-chain statistics, torch's cat + var over the whole trace. Any value that differs from the
others of its kind is a read (or reduction) fault, not a sampling fault: the trace is written once.

    python3 tools/r4_rhat_stress.py [N]  > gpurun_out/<tag>/stress.txt
"""
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    # the eight_schools leg of the default run, including its 2 GB multi_step buffers
    spec0, _ = bench.make_spec("eight_schools")
    comp0 = sampler.compile(spec0, {"device": 0})
    bench.multi_step_roofline(comp0, spec0, dev)
    comp0.close()
    spec, _ = bench.make_spec("sv")
    comp = sampler.compile(spec, {"device": 0})
    L = comp.L
    d, S, Cper, B, W = spec.d, 1000, 2048, 50, 5
    opts = sampler._merge_opts(dict(num_warmup=1000, num_samples=S, seed=42, lanes_per_chain=comp.default_lanes))
    tuning = sampler.warmup(comp, spec.default_init, dict(opts, warmup_lanes=comp.default_warmup_lanes))
    tun = sampler._tuning_struct(tuning, d)
    iq = np.ascontiguousarray(spec.to_unconstrained(spec.default_init))
    iqp = iq.ctypes.data_as(C.POINTER(C.c_double))
    bad = 0
    for it in range(n):
        draws = torch.empty((S, d, Cper), dtype=torch.float64, device=dev)
        n_steps = torch.empty((S, Cper), dtype=torch.int32, device=dev)
        depth = torch.empty((S, Cper), dtype=torch.int32, device=dev)
        diverg = torch.empty((S, Cper), dtype=torch.int32, device=dev)
        accept = torch.empty((S, Cper), dtype=torch.float64, device=dev)
        tr = _lib.Trace(draws.data_ptr(), None, depth.data_ptr(), n_steps.data_ptr(), diverg.data_ptr(),
                        accept.data_ptr(), None)
        lf, dv = C.c_int64(), C.c_int32()
        comp.check(L.exmc_hip_chains_init(comp.h, C.byref(tun), iqp, Cper, 0, Cper, sampler._c_opts(opts)))
        for k in range(W):
            comp.check(L.exmc_hip_chains_advance(comp.h, B, (k % 20) * B, tr, C.byref(lf), C.byref(dv)))
        torch.cuda.synchronize()
        comp.check(L.exmc_hip_chains_init(comp.h, C.byref(tun), iqp, Cper, 0, Cper, sampler._c_opts(opts)))
        comp.check(L.exmc_hip_chains_advance(comp.h, S, 0, tr, C.byref(lf), C.byref(dv)))
        torch.cuda.synchronize()
        vals = {"ess": [], "rhat_lib": [], "rhat_stats": [], "rhat_torch": [], "sum": []}
        for rep in range(3):
            ess = torch.empty((d, Cper), dtype=torch.float64, device=dev)
            comp.check(L.exmc_hip_ess(comp.h, draws.data_ptr(), S, d, Cper, ess.data_ptr()))
            vals["ess"].append(float(ess.sum(dim=1).min()))
            hm, hv, hn = xd.half_chain_stats(draws)
            vals["rhat_stats"].append(float(xd.split_rhat_from_stats(hm, hv, hn).max()))
            vals["rhat_torch"].append(float(xd.split_rhat(draws).max()))
            rk = torch.empty((d,), dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            comp.check(L.exmc_hip_rhat(comp.h, draws.data_ptr(), S, d, Cper, rk.data_ptr()))
            vals["rhat_lib"].append(float(rk.max()))
            vals["sum"].append(float(draws.sum()))
        ok = all(len(set(v)) == 1 for v in vals.values())
        bad += 0 if ok else 1
        print("iter %d lf=%d %s %s" % (it, lf.value, "consistent" if ok else "INCONSISTENT",
                                       {k: (v[0] if len(set(v)) == 1 else v) for k, v in vals.items()}), flush=True)
        del draws, n_steps, depth, diverg, accept
    print("inconsistent iterations: %d of %d" % (bad, n))


if __name__ == "__main__":
    main()
