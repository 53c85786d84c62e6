#!/usr/bin/env python3
"""bench.py — leapfrog steps/s and ESS/s of the NUTS hot path on MI355X.

A "step" is one pass of the hot path over one batch: --draws-per-step (default 50) NUTS
transitions for every resident chain, so the driver's `--steps 20` is SURVEY 8(d)'s protocol --
1000 draws per chain after one shared 1000-iteration warmup. After the shared adaptation warmup
(sampler.ex:1053-1080, run once, untimed for `value`, wall time reported and counted in ESS/s) W
untimed steps are taken on throw-away chains, the chains are initialised again, and exactly K steps
(one launch of the NUTS kernel, one trace of K * draws-per-step rows) are timed between
barrier + synchronize pairs. value = useful leapfrogs (sum of n_steps, tree.ex:1612) per second
over all GPUs; ESS/s (adaptation + sampling + the ESS kernel + the gather), the roofline of the
NUTS kernel and the CPU checker's numbers ride along. Order of the default run: the eight_schools
sampling leg, the sv sampling leg (BASELINE.json's metric names both), then -- after every sampling leg
-- the batched-leapfrog roofline leg and the CPU checker's legs (profiles/r4_driver_cmd/README.md says
why the order matters). Split R-hat is computed on two independent routes that must agree to 1e-9,
or every rank exits with code 3 (finish_model).

    python bench.py --gpus N --steps K --warmup W      (N > 1 without a launcher: spawns the ranks)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from exmc_amd import _lib, models, sampler  # noqa: E402
from exmc_amd import distributed as xd  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def synthetic_sv_returns():
    """returns simulated with sigma*=0.15, nu*=10 (STANDARD_BENCHMARKS.md:79) -- the
    committed series (exmc_amd/data/workloads.npz, digest-pinned), not a fresh draw."""
    return models.sv_returns()


def make_spec(name):
    if name == "eight_schools":
        return models.eight_schools(), 488  # (6d+1)*8 bytes per leapfrog per chain, SURVEY 8d
    if name == "sv":
        return models.sv(synthetic_sv_returns()), 4904
    if name == "logistic":
        return models.logistic(), 1016
    if name == "radon":
        return models.radon(), 4328
    if name == "gen_eight_schools":
        # the same posterior written as 26 Builder nodes and compiled by exmc_amd/codegen.py
        # (one lane per chain, generated value + gradient); not a BASELINE config
        from exmc_amd import codegen
        init = {n: 0.0 for n in ["mu"] + ["theta_%d" % j for j in range(8)]}
        init["tau"] = 1.0
        return codegen.compile_ir(codegen.eight_schools_ir(), name=name, default_init=init), 488
    if name in ("gen_sv", "gen_radon", "gen_logistic"):
        # the BASELINE configs above d = 20 compiled from Builder node lists (the lane layout of
        # exmc_amd/codegen_lanes.py) instead of their hand-written kinds: same data, same names
        from exmc_amd import codegen
        hand, nbytes = make_spec(name[4:])
        if name == "gen_sv":
            ir, ncp, lanes = codegen.sv_ir(hand.data), False, 64
        elif name == "gen_logistic":
            n = hand.n_obs
            ir, ncp, lanes = codegen.logistic_ir(hand.data[:n * 20].reshape(n, 20), hand.data[n * 20:]), True, 16
        else:
            J = 85
            start = hand.data[J:2 * J + 1].astype(int)
            nobs = int(start[-1])
            cty = [v for v in hand.var_names if v.startswith("alpha_raw")]
            ir = codegen.radon_ir(hand.data[:J], start, hand.data[2 * J + 1:2 * J + 1 + nobs],
                                  hand.data[2 * J + 1 + nobs:], names=cty)
            ncp, lanes = False, 64
        # radon's 1024 chains are 1024 wavefronts, one per SIMD: no second wave to make room for.
        # sv and logistic launch 2048: compiled for two resident waves per SIMD like their
        # hand-written kinds (the plug-in sizes its LDS so that eight workgroups fit a CU;
        # gen_sv 1.9 -> 2.9e8, gen_logistic 1.4 -> 1.7e8 leapfrog/s; EXMC_GEN_WPS=1 / 2 for A/B runs)
        wps = int(os.environ.get("EXMC_GEN_WPS", "1" if name == "gen_radon" else "2"))
        return codegen.compile_ir(ir, ncp=ncp, name=name, default_init=hand.default_init, lanes=lanes,
                                  waves_per_simd=wps), nbytes
    raise SystemExit("unknown model %s" % name)


DEFAULT_CHAINS_PER_GPU = {"eight_schools": 4096, "logistic": 8192, "sv": 2048, "radon": 1024,
                          "gen_eight_schools": 4096, "gen_sv": 2048, "gen_radon": 1024, "gen_logistic": 8192}


def kernel_source_sha16():
    """sha256[:16] over the kernel sources (exmc_amd/csrc/*, include/*.h): the counter entries of
    profiles/pmc_traffic.json are stamped with it when they are collected (tools/pmc_table_update.py), so
    that a line can say whether the counters it quotes were taken on the kernels it timed."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "exmc_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def counters_current(entry):
    """True: the entry was collected on this tree's kernel sources; False: on other sources (the same
    instruction stream is then an assumption, the equal leapfrog count its only check); None: unstamped."""
    sha = (entry or {}).get("csrc_sha16")
    return None if sha is None else (sha == kernel_source_sha16())


def measured_traffic(model, chains, steps, lanes):
    """HBM bytes of one timed launch from the PMC passes committed under profiles/ (rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE on this same command, see profiles/README.md); None when no
    measurement of this exact workload is on file. bench.py cannot collect PMC counters itself."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return None
    return table.get("%s:%d:%d:%d" % (model, chains, steps, lanes), {}).get("hbm_bytes")


def traffic_entry(model, chains, steps, lanes):
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return {}
    return table.get("%s:%d:%d:%d" % (model, chains, steps, lanes), {})


SIMDS = 1024               # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9           # MI355X_MICROARCH.md: max clock


def issue_roofline(model, chains, steps, lanes, kernel_ms, leapfrogs):
    """The bound that binds nuts_kernel: one wavefront per SIMD, so a SIMD retires at
    most one vector instruction per 4 clocks (the f64 pipe: 16 lanes per clock) and a lone wave
    issues one instruction of any kind per ~5.2 clocks (tools/probe/exec_mask_rate_probe.hip).
    Instruction counts of the timed launch come from the committed SQ counter pass of this exact
    workload (same seeds => same instruction stream); the time is the one measured in this run."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return None
    e = table.get("%s:%d:%d:%d" % (model, chains, steps, lanes), {})
    sq = e.get("sq")
    if not sq or e.get("leapfrogs") != leapfrogs:
        return None           # another workload (or another instruction stream) than the one profiled
    t = kernel_ms * 1e-3
    valu = sq["SQ_INSTS_VALU"]
    f64 = sq["SQ_INSTS_VALU_ADD_F64"] + sq["SQ_INSTS_VALU_MUL_F64"] + sq["SQ_INSTS_VALU_FMA_F64"]
    every = valu + sq["SQ_INSTS_SALU"] + sq["SQ_INSTS_BRANCH"] + sq["SQ_INSTS_LDS"]
    peak = SIMDS * CLOCK_HZ / 4.0 / 1e9
    lone = SIMDS * CLOCK_HZ / 5.2 / 1e9
    out = {"bound": "valu_issue", "achieved": valu / t / 1e9, "peak": peak, "unit": "G wave-instr/s",
           "frac": valu / t / 1e9 / peak, "kernel": "nuts_kernel",
           "valu_per_leapfrog": valu / leapfrogs, "f64_arith_share_of_valu": f64 / valu,
           "instructions_per_leapfrog": every / leapfrogs,
           "source": e.get("source"), "counters_on_these_sources": counters_current(e)}
    if chains * lanes <= SIMDS * 64:
        # the bound of a LONE wave per SIMD; a launch with two waves per SIMD (sv, logistic) issues
        # past it by design, so the figure is only emitted where it applies
        out["lone_wave_issue"] = {"achieved": every / t / 1e9, "peak": lone, "frac": every / t / 1e9 / lone,
                                  "note": "all instructions against one issue per 5.2 clocks per SIMD"}
    return out


def multi_step_roofline(comp, spec, dev, n_chains=262144, n_steps=32, lanes=1, reps=3):
    """The B2 `multi_step_fn` contract at scale (batched_leapfrog.ex:50-101): every chain takes
    n_steps leapfrogs and every intermediate (q, p, grad, logp) is written to HBM, [step][dim][chain].
    Bytes per launch are algorithmic: reads 3*d*8*C, writes (3*d+1)*8*n*C. Kernel time from the HIP
    events the library records on its own stream. (The figure moves between 0.63 and 0.80 of peak with
    the box and with where the four output arrays land: tools/r4_multistep_skew.py)"""
    d = spec.d
    L = comp.L
    g = torch.Generator(device=dev).manual_seed(1)
    q = 0.3 * torch.randn((d, n_chains), dtype=torch.float64, device=dev, generator=g)
    p = torch.randn((d, n_chains), dtype=torch.float64, device=dev, generator=g)
    gr = torch.zeros((d, n_chains), dtype=torch.float64, device=dev)
    aq = torch.empty((n_steps, d, n_chains), dtype=torch.float64, device=dev)
    ap = torch.empty_like(aq)
    ag = torch.empty_like(aq)
    al = torch.empty((n_steps, n_chains), dtype=torch.float64, device=dev)
    im = np.ones(d)
    imp = im.ctypes.data_as(C.POINTER(C.c_double))
    torch.cuda.synchronize()
    times = []
    for i in range(reps + 1):                      # first launch is untimed warmup
        comp.check(L.exmc_hip_multi_step(comp.h, q.data_ptr(), p.data_ptr(), gr.data_ptr(), 0.05, imp,
                                         n_steps, n_chains, lanes, aq.data_ptr(), ap.data_ptr(),
                                         al.data_ptr(), ag.data_ptr()))
        if i:
            times.append(comp.last_kernel_ms)
    best = sum(times) / len(times)                 # average launch duration
    nbytes = 3 * d * 8 * n_chains + (3 * d + 1) * 8 * n_steps * n_chains
    achieved = nbytes / (best * 1e-3) / 1e9
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        table = {}
    traffic = table.get("multi_step:%s:%d:%d:%d" % (spec.name, n_chains, n_steps, lanes), {}).get("hbm_bytes")
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_on_these_sources": counters_current(table.get("multi_step:%s:%d:%d:%d" % (spec.name, n_chains, n_steps, lanes))),
            "kernel": "multi_step_kernel",
            "kernel_ms": best, "chains": n_chains, "steps": n_steps, "lanes_per_chain": lanes,
            "leapfrog_steps_per_s": n_chains * n_steps / (best * 1e-3),
            "bytes_per_launch": nbytes}


RHAT_ROUTE_TOL = 1e-9

# What the reference itself publishes for these models (1 chain, 1000 + 1000,
# 5-seed medians, CPU of unstated make, EXLA-JIT'd gradients): quoted beside cpu_baseline, which is
# this repository's C port of the same sampler on the GPU box's host cores -- not eXMC on the BEAM.
REFERENCE_PUBLISHED = {
    "eight_schools": {"ess_per_s": 12.0, "pymc_ess_per_s": 5.0, "source": "README.md:44",
                      "note": "centered parameterisation (STANDARD_BENCHMARKS.md:30); this line runs the non-centered "
                              "posteriordb variant BASELINE.json names"},
    "logistic": {"ess_per_s": 69.0, "pymc_ess_per_s": 336.0, "source": "README.md:46",
                 "wall_s_per_run": "14-16 (STANDARD_BENCHMARKS.md:185-187)"},
    "sv": {"ess_per_s": 1.0, "pymc_ess_per_s": 1.0, "source": "README.md:47",
           "note": "ratio 1.20x; 83-95 s per run, median min-ESS 52 (STANDARD_BENCHMARKS.md:171-175)"},
    "radon": None,   # no published number (notebooks/09_radon_bhm.livemd prints wall time at run time)
}
F64_PEAK_TFLOPS = 78.6     # 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz: vector FMA = matrix f64 peak


def reference_published(model):
    base = model[4:] if model.startswith("gen_") else model
    r = REFERENCE_PUBLISHED.get(base)
    if r is None:
        return {"ess_per_s": None, "note": "the reference publishes no number for this model (BASELINE.md section 1)"}
    return dict(r, protocol="1 chain, 1000 warmup + 1000 draws, 5-seed median, min-over-parameters ESS / wall s; "
                            "eXMC on the BEAM with EXLA-JIT'd gradients, CPU, hardware unstated",
                comparable="no: different hardware, one chain, and not the implementation cpu_baseline times")


def finish_model(*, model, d, K, W, B, adapt, Cper, world, rank, dist, draws, ess, leap_local, div_local,
                 elapsed_local, kernel_ms, adapt_s, ess_s, ess_ms, epsilon, lanes, warm_lanes,
                 bytes_per_leapfrog, gather_traces, rhat_fn, dense_mass=False, sync=lambda: None,
                 traffic=None, force=False, ess_bulk=None, ess_bulk_s=None):
    """Everything after the timed launch, on whatever device the tensors live (the GPUs of the ranks;
    CPU tensors over gloo in tests/test_bench_multirank_gloo.py): the max / sum reductions of the
    ranks' clocks and counters, the ESS sums, both gather routes (the finished [S][d][C] traces;
    the per-chain half-chain statistics), split R-hat (diagnostics.ex:80-115) by both and the bench
    line. Every rank makes the same collective calls; returns (line on rank 0 / None, routes_agree).

      draws   [S][d][Cper] this rank's finished trace      ess  [d][Cper] per-chain ESS of it
      rhat_fn [S][d][C] -> [d] split R-hat of a whole trace (the library's rhat_kernel on the GPU)

    R-hat is computed twice on independent code paths -- rhat_fn on the traces (this rank's shard
    always; the gathered whole when gather_traces), torch on the gathered half-chain statistics --
    and the two must agree to RHAT_ROUTE_TOL, else the caller exits non-zero: BENCH_r03.json carried
    6.58 and 1.19 for one launch and nothing noticed.

    force: make every collective even in a process group of ONE rank (`--force-dist`,
    tests/test_gpu_rccl_one_rank.py: the RCCL calls executed on the real device of a one-GPU box; the
    line must equal the dist=None line). ess_bulk: [d][Cper] rank-normalised bulk ESS
    (diagnostics.ex:60-72) with its wall clock, or None."""
    S = K * B
    Ctot = Cper * world
    dev = draws.device
    stats = torch.tensor([elapsed_local, float(leap_local), float(div_local)], dtype=torch.float64, device=dev)
    if dist is not None:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, leapfrogs, divs = float(mx[0]), float(sm[1]), float(sm[2])
    else:
        elapsed, leapfrogs, divs = float(elapsed_local), float(leap_local), float(div_local)

    ess_sum = ess.sum(dim=1)
    # the sufficient-statistics route is always timed (it is what R-hat needs); sv also runs the
    # trace all-gather BASELINE.json's config names, and that is the one its ESS/s wall clock counts
    hm, hv, hn = xd.half_chain_stats(draws)
    # this rank's shard by both routes: the statistics this rank contributes are checked against
    # the library's kernel before they travel
    rhat_local_lib = rhat_fn(draws)
    rhat_local_stats = xd.split_rhat_from_stats(hm, hv, hn)
    local_gap = float((rhat_local_lib - rhat_local_stats).abs().max())
    sync()
    t0 = time.perf_counter()
    xd.reduce_sum(ess_sum, dist, force)
    ess_bulk_sum = None
    if ess_bulk is not None:
        ess_bulk_sum = ess_bulk.sum(dim=1)
        xd.reduce_sum(ess_bulk_sum, dist, force)
    gather_stats_s = gather_traces_s = None
    rhat_traces = None
    if gather_traces:
        all_draws = xd.gather_traces(draws, dist, force)
        sync()
        gather_s = gather_traces_s = time.perf_counter() - t0
        rhat_traces = rhat_local_lib if all_draws is draws else rhat_fn(all_draws)
        del all_draws
        t1 = time.perf_counter()
        ghm, ghv = xd.gather_chain_stats(hm, dist, force), xd.gather_chain_stats(hv, dist, force)
        sync()
        gather_stats_s = time.perf_counter() - t1
    else:
        # the exchange: per-chain sufficient statistics of the finished traces give
        # the same split R-hat with ~1 MB per rank on the wire instead of the [S][d][C] draws
        ghm, ghv = xd.gather_chain_stats(hm, dist, force), xd.gather_chain_stats(hv, dist, force)
        sync()
        gather_s = gather_stats_s = time.perf_counter() - t0
        if world == 1:
            rhat_traces = rhat_local_lib     # one rank holds every chain
    rhat_stats = xd.split_rhat_from_stats(ghm, ghv, hn)
    rhat = rhat_traces if rhat_traces is not None else rhat_stats
    gap = torch.tensor([local_gap, float((rhat - rhat_stats).abs().max())], dtype=torch.float64, device=dev)
    gap = torch.nan_to_num(gap, nan=float("inf"))
    xd.reduce_max(gap, dist, force)
    ok = bool(float(gap.max()) <= RHAT_ROUTE_TOL)
    if not ok:
        log("%s rank %d: split R-hat routes disagree: shard |lib - stats| %.3e, whole |%s - stats| %.3e\n"
            "  lib   %s\n  stats %s" % (model, rank, float(gap[0]), "traces" if rhat_traces is not None else "stats",
                                       float(gap[1]), rhat.tolist(), rhat_stats.tolist()))
        # read the finished trace once more by both routes: values that change between two reads of
        # the same buffer are a read (or reduction) fault, values that stay are in the data
        again_lib = rhat_fn(draws)
        hm2, hv2, _ = xd.half_chain_stats(draws)
        again_stats = xd.split_rhat_from_stats(hm2, hv2, hn)
        log("  second evaluation of this rank's shard: lib moved %.3e, stats moved %.3e, |lib - stats| %.3e"
            % (float((again_lib - rhat_local_lib).abs().max()), float((again_stats - rhat_local_stats).abs().max()),
               float((again_lib - again_stats).abs().max())))
    ess_min = float(ess_sum.min())
    total_s = adapt_s + elapsed + ess_s + gather_s
    value = leapfrogs / elapsed
    if rank != 0:
        return None, ok
    local_lf = float(leap_local)
    achieved = bytes_per_leapfrog * local_lf / (kernel_ms * 1e-3) / 1e9
    out = {
        "metric": "leapfrog_steps_per_s",
        "value": value,
        "unit": "leapfrog_steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed * 1e3 / K,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s d=%d, %d chains/GPU (%d total), %d draws/chain (%d steps of %d "
                               "draws) after one shared %d-iteration warmup, max_tree_depth 10, "
                               "target_accept 0.8%s"
                               % (model, d, Cper, Ctot, S, K, B, adapt,
                                  ", dense mass matrix" if dense_mass else ""),
                   "draws_per_step": B, "draws_per_chain": S,
                   "lanes_per_chain": lanes, "warmup_lanes_per_chain": warm_lanes, "seed": 42},
        "ess_per_s": ess_min / total_s,
        "ess_min_total": ess_min,
        "ess_wall_s": {"adaptation": adapt_s, "sampling": elapsed, "ess_kernel": ess_s,
                       "gather": gather_s},
        "gather": {"counted": "traces" if gather_traces else "chain_stats",
                   "traces_s": gather_traces_s, "chain_stats_s": gather_stats_s,
                   "bytes_per_rank": int(draws.numel() * 8) if gather_traces else int(2 * hm.numel() * 8)},
        # split R-hat of ALL chains: the library's rhat_kernel on the whole trace where one place
        # holds it (one rank, or the gathered traces), else the gathered half-chain statistics
        "rhat_max": float(rhat.max()),
        "rhat_route": "rhat_kernel(traces)" if rhat_traces is not None else "chain_stats",
        "rhat_max_from_chain_stats": float(rhat_stats.max()),
        "rhat_routes_agree": ok,
        "rhat_routes_max_gap": float(gap.max()),
        "divergent_transitions": divs,
        "mean_leapfrogs_per_draw": leapfrogs / (S * Ctot),
        "step_size": epsilon,
        "ess_kernel_ms": ess_ms,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "kernel": "nuts_kernel", "launches": 1, "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_leapfrog": bytes_per_leapfrog,
                     "leapfrogs_per_launch": local_lf},
    }
    if ess_bulk_sum is not None:
        # Diagnostics.ess_bulk (diagnostics.ex:60-72): the same minimum over parameters on the
        # rank-normalised traces, with ITS kernels' wall clock in place of the Geyer kernel's
        bulk_min = float(ess_bulk_sum.min())
        out["ess_bulk_min_total"] = bulk_min
        out["ess_bulk_per_s"] = bulk_min / (adapt_s + elapsed + ess_bulk_s + gather_s)
        out["ess_wall_s"]["ess_bulk_kernels"] = ess_bulk_s
    if force:
        out["collectives"] = {"forced": True, "backend": dist.get_backend() if dist is not None else None,
                              "world": world}
    return out, ok


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children through
    torch.distributed.run (before anything in this process touches a GPU; never exec in place, a
    profiler may already have initialised the device) and relay their output and exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline(spec, init, K, n_chains_total, budget_s=15.0, max_chains=None):
    """The CPU checker (oracle/, libm mode = the reference's own arithmetic, one chain per host
    thread) on a bounded sample of the same workload. A reported baseline, not the target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    if hasattr(spec, "gen"):
        # a generated model's CPU form is its own generated text compiled for the host; it only
        # exists in the deterministic-math spelling
        import gen_checker
        gl = spec.gen.lanes if getattr(spec.gen, "lane_layout", None) is not None else 1
        om, cfg = gen_checker.model(spec.gen, gl), O.Cfg(1, gl)
    else:
        om, cfg = O.model_for(spec), O.Cfg(0, 1)
    q0 = spec.to_unconstrained(init)
    cores = min(os.cpu_count() or 1, 64)
    t0 = time.perf_counter()
    O.sample_chains(om, 1, init_q=q0, num_warmup=1000, num_samples=K, seed=42, cfg=cfg)
    one = time.perf_counter() - t0            # warmup + one chain
    t0 = time.perf_counter()
    O.sample_chains(om, 2, init_q=q0, num_warmup=1000, num_samples=K, seed=42, cfg=cfg)
    per_chain = max(time.perf_counter() - t0 - one, 1e-6)
    n = int(max(cores, min(n_chains_total, budget_s * cores / per_chain)))
    if max_chains:
        n = min(n, max(cores, max_chains))
    n -= n % cores
    n = max(n, cores)
    t0 = time.perf_counter()
    t, st = O.sample_chains(om, n, init_q=q0, num_warmup=1000, num_samples=K, seed=42,
                            n_threads=cores, cfg=cfg)
    wall = time.perf_counter() - t0
    L = O.lib()
    ess = np.zeros(spec.d)
    sub = min(n, 4 * cores)
    for c in range(sub):
        for i in range(spec.d):
            x = np.ascontiguousarray(t["draws"][c, :, i])
            ess[i] += L.exo_ess(O.dptr(x), K)
    ess_min = float(ess.min()) * n / sub
    # The sample is smaller than the GPU's batch whenever a chain costs about a second (sv), and its
    # wall clock holds the serial warmup once: per chain that overhead is larger than in a run of
    # all n_chains_total chains. The same legs scaled to the GPU's chain count (serial warmup once,
    # chains at the sample's measured rate per thread) give the ratio at EQUAL chains per run.
    warm_s = max(one - per_chain, 0.0)
    sample_rate = max(wall - warm_s, 1e-9) / n                       # seconds per chain with `cores` threads busy
    wall_eq = warm_s + sample_rate * n_chains_total
    ess_eq = ess_min * n_chains_total / n
    return {
        "value": st.total_leapfrogs / wall,
        "unit": "leapfrog_steps/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d chains x %d draws after the shared 1000-iteration warmup, oracle %s mode, "
                  "%d host threads, wall %.2f s (includes the serial warmup)"
                  % (n, K, "libm" if cfg.math_mode == 0 else "deterministic-math", cores, wall),
        "ess_per_s": ess_min / wall,
        "chains": n,
        "wall_s": wall,
        "ess_per_s_at_equal_chains": ess_eq / wall_eq,
        "equal_chains": {"chains": n_chains_total, "wall_s_extrapolated": wall_eq, "serial_warmup_s": warm_s,
                         "note": "the sample's per-chain rate scaled to the GPU leg's chain count, serial warmup once"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--draws-per-step", type=int, default=50,
                    help="NUTS transitions per chain in one step (one kernel launch); 20 steps of "
                         "50 = the 1000-draw protocol of SURVEY 8(d)")
    ap.add_argument("--model", default="eight_schools")
    ap.add_argument("--chains-per-gpu", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--warmup-lanes", type=int, default=0,
                    help="lanes per chain of the shared one-chain warmup (default: --lanes if given, else the "
                         "library's choice for the model)")
    ap.add_argument("--adapt", type=int, default=1000, help="NUTS adaptation iterations")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dense-mass", action="store_true",
                    help="opts[:dense_mass]: dense adaptation windows and a dense mass matrix while sampling "
                         "(layouts that carry it; not the default protocol, the CPU leg is skipped)")
    ap.add_argument("--no-multi-step", action="store_true",
                    help="skip the batched-leapfrog roofline leg (development builds without that layout)")
    ap.add_argument("--gather-traces", action="store_true",
                    help="all-gather the full [S][d][C] traces for split R-hat instead of the "
                         "per-chain half-chain statistics (sv does by default: BASELINE.json's config)")
    ap.add_argument("--no-sv-leg", action="store_true",
                    help="default run only: skip the sv(d=102) leg that rides on the eight_schools line")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="default run only: skip the logistic and radon legs (BASELINE configs 3 and 5)")
    ap.add_argument("--independent-warmup", action="store_true",
                    help="sample_chains(ir, n, vectorized: false), sampler.ex:1139-1176: every chain runs its OWN "
                         "adaptation and then its draws, all chains in one launch (exmc_hip_sample_independent); "
                         "the timed region is that launch, adaptation included. Not the default protocol.")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1 only: create a ONE-rank nccl (RCCL) process group on the device and make "
                         "every collective of the multi-GPU path anyway (all_reduce, all_gather_into_tensor, "
                         "all_gather, barrier); the line must equal the ordinary one")
    args = ap.parse_args()
    # before the first torch.cuda call: the HIP / HSA runtime reads its environment when it is
    # initialised (the host driver supports dmabuf IPC only; RCCL needs this for more than one rank,
    # and --force-dist is to run under the same runtime configuration as the ranks it stands for)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (no CPU fallback)")
    # The contract is ONE JSON line on stdout. Libraries write there too -- RCCL prints a five-line
    # version banner to stdout when the first communicator is created (seen the first time any nccl
    # branch of this file ran: profiles/r5_rccl) -- so from here on file descriptor 1 IS stderr, and
    # the line goes out through a private duplicate of the real stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # EXMC_BENCH_SHARED_GPU=1 is a REHEARSAL of the multi-rank control flow on a box with one GPU (the builder's):
    # every rank uses cuda:0 and the ranks talk over gloo (RCCL refuses two ranks on one device). The line says so
    # ("rehearsal") and is never a measurement; the driver's launch does not set the variable.
    shared_gpu = os.environ.get("EXMC_BENCH_SHARED_GPU") == "1" and world > 1
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    elif args.force_dist:
        import torch.distributed as dist
        store = dist.TCPStore("127.0.0.1", 0, 1, is_master=True, wait_for_workers=False)
        dist.init_process_group("nccl", store=store, rank=0, world_size=1, device_id=dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out, ok = run_model(args, args.model, rank, local_rank, world, dev, dist, barrier, primary=True)
    if args.model == "eight_schools" and not args.no_sv_leg and not args.dense_mass:
        # BASELINE.json's metric names two models: the line the driver records carries the sv(d=102)
        # leg too -- 2048 chains per GPU (16384 over 8), the same 1000-draw protocol, the finished
        # traces all-gathered over RCCL as the config says, its CPU leg on a smaller sample
        sv, ok_sv = run_model(args, "sv", rank, local_rank, world, dev, dist, barrier, primary=False)
        ok = ok and ok_sv
        if rank == 0:
            out["models"] = {"sv": sv}
    if args.model == "eight_schools" and not args.no_extra_legs and not args.dense_mass:
        # BASELINE configs 3 and 5 in the same record: logistic 8192 and radon 1024 chains per GPU (their
        # GPU legs are a few hundred milliseconds each; their CPU legs are bounded to a few seconds)
        for extra in ("logistic", "radon"):
            eo, ok_e = run_model(args, extra, rank, local_rank, world, dev, dist, barrier, primary=False)
            ok = ok and ok_e
            if rank == 0:
                out.setdefault("models", {})[extra] = eo
    if rank == 0:
        # the batched-leapfrog roofline leg and the CPU checker's legs, after the sampling legs of the line
        # (see run_model)
        legs = [out] + list(out.get("models", {}).values())
        for key in ("_gpu_leg", "_cpu_leg"):
            for o in legs:
                leg = o.pop(key, None)
                if leg is not None:
                    leg()
        if shared_gpu:
            out["rehearsal"] = ("%d ranks sharing ONE GPU over gloo (EXMC_BENCH_SHARED_GPU=1): the multi-rank control "
                                "flow executed, not a measurement" % world)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        # the line above is printed for the record, but a run whose two R-hat routes disagree is not
        # a measurement (finish_model): every rank leaves with the same non-zero code
        log("bench.py: split R-hat routes disagree by more than %.0e -- failing the run" % RHAT_ROUTE_TOL)
        raise SystemExit(3)


def run_model(args, model, rank, local_rank, world, dev, dist, barrier, primary):
    """One model through the protocol; returns (the result object on rank 0 / None elsewhere,
    whether the two split R-hat routes agreed -- the same flag on every rank)."""
    spec, bytes_per_leapfrog = make_spec(model)
    gather_traces = args.gather_traces or model in ("sv", "gen_sv")
    K, W, d = args.steps, args.warmup, spec.d
    B = args.draws_per_step
    if K < 1 or B < 1 or W < 0:
        raise SystemExit("--steps and --draws-per-step must be >= 1, --warmup >= 0")
    S = K * B                      # draws per chain in the timed region
    Cper = (args.chains_per_gpu if primary else 0) or DEFAULT_CHAINS_PER_GPU[model]
    Ctot = Cper * world
    comp = sampler.compile(spec, {"device": local_rank})
    lanes = (args.lanes if primary else 0) or comp.default_lanes
    opts = sampler._merge_opts(dict(num_warmup=args.adapt, num_samples=S, seed=42,
                                    lanes_per_chain=lanes))
    if args.dense_mass:
        opts["dense_mass"] = True
        args.no_cpu = True
    init = spec.default_init
    L = comp.L

    if getattr(args, "independent_warmup", False):
        return run_model_independent(args, model, spec, comp, opts, lanes, bytes_per_leapfrog, gather_traces,
                                     rank, world, dev, dist, barrier, Cper)
    # --- shared adaptation warmup: every rank runs it with the same seed (deterministic, so no
    # broadcast is needed) ---
    # (a 2-iteration throwaway call first, untimed like the W warmup steps of the sampling region:
    # it pays for loading the code object and the first-launch setup, not for adaptation)
    warm_lanes = ((args.warmup_lanes or args.lanes) if primary else 0) or comp.default_warmup_lanes
    sampler.warmup(comp, init, dict(opts, num_warmup=2, warmup_lanes=warm_lanes))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tuning = sampler.warmup(comp, init, dict(opts, warmup_lanes=warm_lanes))
    adapt_s = time.perf_counter() - t0
    tun = sampler._tuning_struct(tuning, d)
    if rank == 0:
        log("%s adaptation: eps=%.5f, %d iterations in %.3f s" % (model, tuning["epsilon"], args.adapt, adapt_s))

    # --- resident chains + trace buffers in HBM ---
    draws = torch.empty((S, d, Cper), dtype=torch.float64, device=dev)
    n_steps = torch.empty((S, Cper), dtype=torch.int32, device=dev)
    depth = torch.empty((S, Cper), dtype=torch.int32, device=dev)
    diverg = torch.empty((S, Cper), dtype=torch.int32, device=dev)
    accept = torch.empty((S, Cper), dtype=torch.float64, device=dev)
    tr = _lib.Trace(draws.data_ptr(), None, depth.data_ptr(), n_steps.data_ptr(), diverg.data_ptr(),
                    accept.data_ptr(), None)
    iq = np.ascontiguousarray(spec.to_unconstrained(init))
    iqp = iq.ctypes.data_as(C.POINTER(C.c_double))
    lo, hi = rank * Cper, (rank + 1) * Cper
    lf, dv = C.c_int64(), C.c_int32()
    if W > 0:
        # untimed steps on throw-away chains (they pay for the code object and first-touch of the
        # trace buffers); the timed region starts from freshly initialised chains
        comp.check(L.exmc_hip_chains_init(comp.h, C.byref(tun), iqp, Ctot, lo, hi, sampler._c_opts(opts)))
        for k in range(W):
            comp.check(L.exmc_hip_chains_advance(comp.h, B, (k % K) * B, tr, C.byref(lf), C.byref(dv)))

    # --- timed region: chain initialisation, then exactly K steps of B draws for every chain ---
    barrier()
    t0 = time.perf_counter()
    comp.check(L.exmc_hip_chains_init(comp.h, C.byref(tun), iqp, Ctot, lo, hi, sampler._c_opts(opts)))
    # the K steps are one launch of the NUTS kernel (K * B transitions per chain): the chains stay in
    # registers from the first draw to the last
    comp.check(L.exmc_hip_chains_advance(comp.h, S, 0, tr, C.byref(lf), C.byref(dv)))
    leap_local, div_local = lf.value, dv.value
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = comp.last_kernel_ms                   # HIP events on the library's stream

    # --- diagnostics: per-chain Geyer ESS on device, summed over chains; RCCL all-gather of the
    # finished traces for split R-hat (the only collective on the path) ---
    ess = torch.empty((d, Cper), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    comp.check(L.exmc_hip_ess(comp.h, draws.data_ptr(), S, d, Cper, ess.data_ptr()))
    ess_s = time.perf_counter() - t0      # the call returns when the kernel has finished
    ess_ms = comp.last_kernel_ms
    # Diagnostics.ess_bulk (diagnostics.ex:60-72): rank-normalise (rank_scores_kernel) into a second
    # trace, then the same ESS kernels on it -- BASELINE.md section 2 asks for ESS/s by both
    essb = torch.empty((d, Cper), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    comp.check(L.exmc_hip_ess_bulk(comp.h, draws.data_ptr(), S, d, Cper, essb.data_ptr()))
    ess_bulk_s = time.perf_counter() - t0

    def rhat_lib(x):
        """Diagnostics.rhat (diagnostics.ex:80-115) of a [S][d][C] device trace by the library's own
        rhat_kernel -- the reference's summation order, bit-identical to the checker
        (tests/test_gpu_diagnostics.py). The library launches on its own stream: whatever torch has
        queued for x (the all-gather, the permute copy) must have finished first."""
        x = x.contiguous()
        rk = torch.empty((x.shape[1],), dtype=torch.float64, device=x.device)
        torch.cuda.synchronize()
        comp.check(L.exmc_hip_rhat(comp.h, x.data_ptr(), x.shape[0], x.shape[1], x.shape[2], rk.data_ptr()))
        return rk

    out, ok = finish_model(model=model, d=d, K=K, W=W, B=B, adapt=args.adapt, Cper=Cper, world=world,
                           rank=rank, dist=dist, draws=draws, ess=ess, leap_local=leap_local,
                           div_local=div_local, elapsed_local=elapsed, kernel_ms=kernel_ms,
                           adapt_s=adapt_s, ess_s=ess_s, ess_ms=ess_ms, epsilon=tuning["epsilon"],
                           lanes=lanes, warm_lanes=warm_lanes, bytes_per_leapfrog=bytes_per_leapfrog,
                           gather_traces=gather_traces, rhat_fn=rhat_lib, dense_mass=args.dense_mass,
                           sync=torch.cuda.synchronize,
                           traffic=measured_traffic(model, Cper, S, lanes),
                           force=bool(getattr(args, "force_dist", False)) and world == 1,
                           ess_bulk=essb, ess_bulk_s=ess_bulk_s)
    if rank == 0:
        out["roofline"]["traffic_on_these_sources"] = counters_current(traffic_entry(model, Cper, S, lanes))
        value, local_lf = out["value"], float(leap_local)
        if model == "eight_schools":
            # the HBM fraction of this kernel has a ceiling that is not HBM
            out["roofline"]["note"] = ("issue-bound, not HBM-bound: chain state never leaves registers / LDS (counter traffic = the "
                                       "trace write); 4096 chains x 16 lanes are one wave per SIMD, which issues at 0.88 of a lone "
                                       "wave's rate; 0.40 of HBM peak would need <= 61 instructions per leapfrog, the kernel "
                                       "executes 198 (DESIGN.md section 5)")
        ri = issue_roofline(model, Cper, S, lanes, kernel_ms, local_lf) if world == 1 else None
        if ri:
            out["roofline_issue"] = ri
        out["reference_published"] = reference_published(model)
        if model in ("logistic", "gen_logistic"):
            # "MFMA utilisation instead of HBM fraction for the logistic logp kernel".
            # The two contractions of a leapfrog (X beta and X^T r over N x (K + 1)) are 2 * 2 * N * (K + 1)
            # flops; the f64 matrix peak equals the vector FMA peak on this chip, and the kernel runs them
            # on the vector pipe (the per-observation exp / quotient / log bind it)
            n_obs = int(getattr(spec, "n_obs", 500))
            flop = 4.0 * n_obs * spec.d
            tf = flop * local_lf / (kernel_ms * 1e-3) / 1e12
            out["roofline_f64"] = {"bound": "f64_flops", "achieved": tf, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": tf / F64_PEAK_TFLOPS, "flop_per_leapfrog": flop,
                                   "note": "contraction flops only (2 * 2 * N * (K + 1)); the specials are not counted"}
        if world == 1 and model == "eight_schools" and not args.no_multi_step:
            # the B2 batched-leapfrog contract at a batch that fills the chip (not the timed path). It runs
            # after every sampling leg of the line: its 2 GB of buffers, allocated and freed between the
            # two models' legs, made the sv launch that followed 6 % slower in two runs of three (1693 /
            # 1697 / 1602 ms against 1593-1605 without it, tools/r4_sv_variance.sh, profiles/r4_driver_cmd)
            def multi_step_leg():
                c2 = sampler.compile(spec, {"device": local_rank})
                try:
                    out["roofline_multi_step"] = multi_step_roofline(c2, spec, dev)
                finally:
                    c2.close()
                    torch.cuda.empty_cache()
            out["_gpu_leg"] = multi_step_leg
        if world == 1 and not args.no_cpu:
            # The CPU legs run after EVERY GPU leg of the line (main calls these): nothing host-side
            # sits between the two models' GPU legs
            def cpu_leg():
                # the second model's CPU leg is bounded tighter (sv: about one second per chain per core)
                # (logistic, radon: a chain is ~1 s of one core; their legs stay within ~6 s each so that the
                # whole default command stays under a minute)
                budget = 15.0 if primary else (8.0 if model == "sv" else 3.0)
                cb = cpu_baseline(spec, init, S, Ctot, budget_s=budget, max_chains=None if primary else 256)
                out["cpu_baseline"] = cb
                out["gpu_over_cpu"] = {"leapfrog_steps_per_s": value / cb["value"],
                                       "ess_per_s": out["ess_per_s"] / cb["ess_per_s"],
                                       "ess_per_s_at_equal_chains": out["ess_per_s"] / cb["ess_per_s_at_equal_chains"]}
            out["_cpu_leg"] = cpu_leg
        comp.close()
        return out, ok
    comp.close()
    return None, ok


def run_model_independent(args, model, spec, comp, opts, lanes, bytes_per_leapfrog, gather_traces, rank, world,
                          dev, dist, barrier, Cper):
    """`--independent-warmup`: sample_chains_parallel (sampler.ex:1139-1176) as ONE launch per rank --
    chain i = sample/3 with seed 42 + 7919 i: its own step-size search, dual averaging and Welford
    windows, then its draws from the adapted position. The timed region is the whole launch; `value`
    counts every leapfrog of it (warmup and sampling), ESS/s has no separate adaptation term."""
    K, W, d = args.steps, args.warmup, spec.d
    B = args.draws_per_step
    S = K * B
    Ctot = Cper * world
    L = comp.L
    lo, hi = rank * Cper, (rank + 1) * Cper
    draws = torch.empty((S, d, Cper), dtype=torch.float64, device=dev)
    n_steps = torch.empty((S, Cper), dtype=torch.int32, device=dev)
    diverg = torch.empty((S, Cper), dtype=torch.int32, device=dev)
    tr = _lib.Trace(draws.data_ptr(), None, None, n_steps.data_ptr(), diverg.data_ptr(), None, None)
    tune = np.zeros((Cper, 3 + d))
    tp = tune.ctypes.data_as(C.POINTER(C.c_double))
    lf, dv = C.c_int64(), C.c_int32()
    init = spec.default_init
    iq = np.ascontiguousarray(spec.to_unconstrained(init))
    iqp = iq.ctypes.data_as(C.POINTER(C.c_double))
    if W > 0:   # a throw-away launch pays for the code object and the first touch of the buffers
        o2 = sampler._c_opts(dict(opts, num_warmup=2, num_samples=min(S, 2)))
        comp.check(L.exmc_hip_sample_independent(comp.h, iqp, Ctot, lo, hi, o2, tr, tp, C.byref(lf), C.byref(dv)))
    barrier()
    t0 = time.perf_counter()
    comp.check(L.exmc_hip_sample_independent(comp.h, iqp, Ctot, lo, hi, sampler._c_opts(opts), tr, tp,
                                             C.byref(lf), C.byref(dv)))
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = comp.last_kernel_ms
    warm_lf = int(tune[:, 2].sum())
    ess = torch.empty((d, Cper), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    comp.check(L.exmc_hip_ess(comp.h, draws.data_ptr(), S, d, Cper, ess.data_ptr()))
    ess_s = time.perf_counter() - t0
    ess_ms = comp.last_kernel_ms
    essb = torch.empty((d, Cper), dtype=torch.float64, device=dev)
    t0 = time.perf_counter()
    comp.check(L.exmc_hip_ess_bulk(comp.h, draws.data_ptr(), S, d, Cper, essb.data_ptr()))
    ess_bulk_s = time.perf_counter() - t0

    def rhat_lib(x):
        x = x.contiguous()
        rk = torch.empty((x.shape[1],), dtype=torch.float64, device=x.device)
        torch.cuda.synchronize()
        comp.check(L.exmc_hip_rhat(comp.h, x.data_ptr(), x.shape[0], x.shape[1], x.shape[2], rk.data_ptr()))
        return rk
    eps = np.sort(tune[:, 0])
    out, ok = finish_model(model=model, d=d, K=K, W=W, B=B, adapt=args.adapt, Cper=Cper, world=world, rank=rank,
                           dist=dist, draws=draws, ess=ess, leap_local=lf.value + warm_lf, div_local=dv.value,
                           elapsed_local=elapsed, kernel_ms=kernel_ms, adapt_s=0.0, ess_s=ess_s, ess_ms=ess_ms,
                           epsilon=float(eps[len(eps) // 2]), lanes=lanes, warm_lanes=lanes,
                           bytes_per_leapfrog=bytes_per_leapfrog, gather_traces=gather_traces, rhat_fn=rhat_lib,
                           sync=torch.cuda.synchronize, ess_bulk=essb, ess_bulk_s=ess_bulk_s)
    if rank == 0:
        out["config"]["workload"] = out["config"]["workload"].replace(
            "after one shared %d-iteration warmup" % args.adapt,
            "each after its OWN %d-iteration warmup (vectorized: false), one launch" % args.adapt)
        out["independent_warmup"] = {
            "sampling_leapfrogs": int(lf.value), "warmup_leapfrogs": warm_lf,
            "warmup_divergences": int(tune[:, 1].sum()),
            "step_size_min_median_max": [float(eps[0]), float(eps[len(eps) // 2]), float(eps[-1])],
            "note": "value and roofline count the leapfrogs of both phases; ess_wall_s.adaptation is 0 because the "
                    "adaptation is inside the timed launch"}
        out["reference_published"] = reference_published(model)
        out["roofline"]["kernel"] = "indep_kernel"
        if world == 1:
            out["mean_leapfrogs_per_draw"] = lf.value / float(S * Ctot)   # the draws' own trees
    comp.close()
    return (out if rank == 0 else None), ok


if __name__ == "__main__":
    main()
