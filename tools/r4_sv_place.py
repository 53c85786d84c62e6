#!/usr/bin/env python3
"""Which allocation makes the sv launch 6 % slower after 2 GB of torch buffers were allocated and freed
(profiles/r4_driver_cmd/README.md)? One process per case:
    none        the sv leg alone
    cached      3 x 671 MB + 67 MB torch tensors written once and freed (they stay in torch's cache), then sv
    emptied     the same, then torch.cuda.empty_cache(), then sv
    handle1st   the sv handle compiled and warmed up BEFORE the 2 GB (its own hipMallocs come first), then sv
    msleg       bench.multi_step_roofline on an eight_schools handle, the handle closed, then sv
    msleg_keep  the same with the eight_schools handle left open
Prints the kernel time of the timed launch and where the trace landed."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def case(mode):
    import numpy as np
    import torch
    import bench
    from exmc_amd import _lib, sampler
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    spec, _ = bench.make_spec("sv")
    comp = None
    d, S, Cper, B, W = spec.d, 1000, 2048, 50, 5

    def prep():
        c = sampler.compile(spec, {"device": 0})
        opts = sampler._merge_opts(dict(num_warmup=1000, num_samples=S, seed=42, lanes_per_chain=c.default_lanes))
        tuning = sampler.warmup(c, spec.default_init, dict(opts, warmup_lanes=c.default_warmup_lanes))
        return c, opts, tuning
    if mode == "handle1st":
        comp, opts, tuning = prep()
    if mode in ("msleg", "msleg_keep"):
        # the batched-leapfrog roofline leg itself, on an eight_schools handle (closed, or kept open)
        spec0, _ = bench.make_spec("eight_schools")
        comp0 = sampler.compile(spec0, {"device": 0})
        bench.multi_step_roofline(comp0, spec0, dev)
        if mode == "msleg":
            comp0.close()
    elif mode == "svms":
        # a batched-leapfrog launch of sv's own kernel (sv-only development builds have no other kind)
        comp0 = sampler.compile(spec, {"device": 0})
        bench.multi_step_roofline(comp0, spec, dev, n_chains=8192, n_steps=8, lanes=64)
        comp0.close()
    elif mode != "none":
        bufs = [torch.empty((32, 10, 262144), dtype=torch.float64, device=dev) for _ in range(3)]
        bufs.append(torch.empty((32, 262144), dtype=torch.float64, device=dev))
        for b in bufs:
            b.fill_(1.0)
        torch.cuda.synchronize()
        del bufs, b
        if mode == "emptied":
            torch.cuda.empty_cache()
    if comp is None:
        comp, opts, tuning = prep()
    L = comp.L
    tun = sampler._tuning_struct(tuning, d)
    draws = torch.empty((S, d, Cper), dtype=torch.float64, device=dev)
    n_steps = torch.empty((S, Cper), dtype=torch.int32, device=dev)
    depth = torch.empty((S, Cper), dtype=torch.int32, device=dev)
    diverg = torch.empty((S, Cper), dtype=torch.int32, device=dev)
    accept = torch.empty((S, Cper), dtype=torch.float64, device=dev)
    tr = _lib.Trace(draws.data_ptr(), None, depth.data_ptr(), n_steps.data_ptr(), diverg.data_ptr(), accept.data_ptr(), None)
    iq = np.ascontiguousarray(spec.to_unconstrained(spec.default_init))
    iqp = iq.ctypes.data_as(C.POINTER(C.c_double))
    lf, dv = C.c_int64(), C.c_int32()
    comp.check(L.exmc_hip_chains_init(comp.h, C.byref(tun), iqp, Cper, 0, Cper, sampler._c_opts(opts)))
    for k in range(W):
        comp.check(L.exmc_hip_chains_advance(comp.h, B, (k % 20) * B, tr, C.byref(lf), C.byref(dv)))
    comp.check(L.exmc_hip_chains_init(comp.h, C.byref(tun), iqp, Cper, 0, Cper, sampler._c_opts(opts)))
    comp.check(L.exmc_hip_chains_advance(comp.h, S, 0, tr, C.byref(lf), C.byref(dv)))
    print("%-10s kernel %.1f ms  draws @ %#x (mod 2 MB %#x)  n_steps @ %#x  reserved %.2f GB" % (
        mode, comp.last_kernel_ms, draws.data_ptr(), draws.data_ptr() % (2 << 20), n_steps.data_ptr(),
        torch.cuda.memory_reserved() / 2 ** 30), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        case(sys.argv[1])
    else:
        for rep in range(3):
            for mode in (sys.argv[2:] if False else ("none", "msleg", "msleg_keep", "cached")):
                subprocess.call([sys.executable, os.path.abspath(__file__), mode], stderr=subprocess.DEVNULL)
