#!/usr/bin/env python3
"""Cost of a model's log-density + gradient alone: the B2 multi_step kernel (one leapfrog = one
logp_grad call plus 3 d flops) over a batch of chains, per model and layout:
    python tools/model_cost.py sv gen_sv logistic gen_logistic radon gen_radon [--steps 64]
    python tools/model_cost.py logistic:4:32768 logistic:16:32768      (model:lanes:chains)
Prints ns per leapfrog per chain and leapfrogs/s; under rocprofv3 --pmc SQ_INSTS_VALU ... the
multi_step_kernel rows give instructions per leapfrog."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from exmc_amd import sampler  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 64
    dev = torch.device("cuda", 0)
    for arg in args:
        name, *over = arg.split(":")
        spec, _ = bench.make_spec(name)
        comp = sampler.compile(spec)
        lanes = int(over[0]) if over else comp.default_lanes
        n = int(over[1]) if len(over) > 1 else bench.DEFAULT_CHAINS_PER_GPU[name]
        d = spec.d
        g = torch.Generator(device=dev).manual_seed(1)
        q0 = torch.tensor(spec.to_unconstrained(spec.default_init), dtype=torch.float64, device=dev)
        q = (q0[:, None] + 0.05 * torch.randn((d, n), dtype=torch.float64, device=dev, generator=g)).contiguous()
        p = 0.1 * torch.randn((d, n), dtype=torch.float64, device=dev, generator=g)
        gr = torch.zeros((d, n), dtype=torch.float64, device=dev)
        aq = torch.empty((steps, d, n), dtype=torch.float64, device=dev)
        ap, ag = torch.empty_like(aq), torch.empty_like(aq)
        al = torch.empty((steps, n), dtype=torch.float64, device=dev)
        im = np.ones(d)
        imp = im.ctypes.data_as(C.POINTER(C.c_double))
        torch.cuda.synchronize()
        ts = []
        for i in range(4):
            comp.check(comp.L.exmc_hip_multi_step(comp.h, q.data_ptr(), p.data_ptr(), gr.data_ptr(), 1e-3, imp,
                                                  steps, n, lanes, aq.data_ptr(), ap.data_ptr(), al.data_ptr(),
                                                  ag.data_ptr()))
            if i:
                ts.append(comp.last_kernel_ms)
        ms = min(ts)
        waves = n * lanes / 64
        name = arg
        print("%-14s lanes %2d chains %5d (%5d waves): %8.3f ms / %d steps = %7.1f ns per leapfrog-chain, "
              "%.3e leapfrog/s, %.2f us per wave-leapfrog" % (name, lanes, n, waves, ms, steps, ms * 1e6 / steps / n,
                                                              n * steps / (ms * 1e-3), ms * 1e3 / steps / max(1.0, waves / 1024.0) / (1 if waves >= 1024 else 1)))
        comp.close()


if __name__ == "__main__":
    main()
