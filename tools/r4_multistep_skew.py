#!/usr/bin/env python3
"""multi_step_kernel (the B2 batched leapfrog, 262144 chains x 32 steps, 2.14 GB written) measures 0.66 to 0.80 of HBM
peak from run to run. Its three [n][d][C] outputs and the [n][C] one are written at the same relative offsets at the
same time: does the relative placement of the four arrays decide it? One allocation, the arrays carved out of it with a
skew of `s` bytes between consecutive ones."""
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
    dev = torch.device("cuda", 0)
    spec, _ = bench.make_spec("eight_schools")
    comp = sampler.compile(spec, {"device": 0})
    L = comp.L
    d, n_chains, n_steps = spec.d, 262144, 32
    g = torch.Generator(device=dev).manual_seed(1)
    q = 0.3 * torch.randn((d, n_chains), dtype=torch.float64, device=dev, generator=g)
    p = torch.randn((d, n_chains), dtype=torch.float64, device=dev, generator=g)
    gr = torch.zeros((d, n_chains), dtype=torch.float64, device=dev)
    big = n_steps * d * n_chains          # doubles of one [n][d][C] output
    small = n_steps * n_chains
    im = np.ones(d)
    imp = im.ctypes.data_as(C.POINTER(C.c_double))
    nbytes = 3 * d * 8 * n_chains + (3 * d + 1) * 8 * n_steps * n_chains
    for skew in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, (2 << 20) + 65536 + 256, 3 * 4096 + 256, 77 * 4096):
        sk = skew // 8
        pool = torch.empty((3 * big + small + 4 * sk + 1024,), dtype=torch.float64, device=dev)
        base = pool.data_ptr()
        aq = base
        ap = aq + (big + sk) * 8
        ag = ap + (big + sk) * 8
        al = ag + (big + sk) * 8
        ts = []
        for i in range(6):
            comp.check(L.exmc_hip_multi_step(comp.h, q.data_ptr(), p.data_ptr(), gr.data_ptr(), 0.05, imp,
                                             n_steps, n_chains, 1, aq, ap, al, ag))
            if i:
                ts.append(comp.last_kernel_ms)
        print("skew %9d B: %s ms  best %.1f %% mean %.1f %% of 8 TB/s" % (
            skew, " ".join("%.3f" % t for t in ts), 100 * nbytes / (min(ts) * 1e-3) / 8e12, 100 * nbytes / (np.mean(ts) * 1e-3) / 8e12), flush=True)
        del pool
    # separate torch allocations, as bench.py makes them
    for rep in range(3):
        aq = torch.empty((n_steps, d, n_chains), dtype=torch.float64, device=dev)
        ap = torch.empty_like(aq)
        ag = torch.empty_like(aq)
        al = torch.empty((n_steps, n_chains), dtype=torch.float64, device=dev)
        ts = []
        for i in range(6):
            comp.check(L.exmc_hip_multi_step(comp.h, q.data_ptr(), p.data_ptr(), gr.data_ptr(), 0.05, imp,
                                             n_steps, n_chains, 1, aq.data_ptr(), ap.data_ptr(), al.data_ptr(), ag.data_ptr()))
            if i:
                ts.append(comp.last_kernel_ms)
        print("separate tensors (%#x %#x %#x %#x): %s ms  mean %.1f %%" % (aq.data_ptr(), ap.data_ptr(), ag.data_ptr(), al.data_ptr(),
              " ".join("%.3f" % t for t in ts), 100 * nbytes / (np.mean(ts) * 1e-3) / 8e12), flush=True)
        del aq, ap, ag, al
        torch.cuda.empty_cache()


def by_chain_count():
    """The row pitch of the [n][d][C] outputs is C * 8 bytes: 2 MB at C = 262144, so the thirty streams a wavefront writes
    per step are congruent modulo 2 MB. Other chain counts, separate torch tensors."""
    dev = torch.device("cuda", 0)
    spec, _ = bench.make_spec("eight_schools")
    comp = sampler.compile(spec, {"device": 0})
    for n_chains in (262144, 262144 + 64, 262144 + 1024, 254 * 1024, 264 * 1024, 249856 + 192, 262144):
        for rep in range(2):
            r = bench.multi_step_roofline(comp, spec, dev, n_chains=n_chains, reps=5)
            print("chains %7d (pitch %% 2 MB = %7d B): %.3f ms, %.1f %% of 8 TB/s" % (n_chains, (n_chains * 8) % (2 << 20), r["kernel_ms"], 100 * r["frac"]), flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "chains":
        by_chain_count()
    else:
        main()
