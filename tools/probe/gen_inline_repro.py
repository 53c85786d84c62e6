#!/usr/bin/env python3
"""Reproducer for the miscompare noted in DESIGN.md ("Generated models"): the d = 9 `zoo` body with
its ~30 exp / log calls INLINED at -O3 (-DEXMC_GEN_INLINE_MATH) against the called form. Every
chain of a batch gets the same input, so any lane that disagrees with lane 0, or any run that
disagrees with the previous one, is visible without a reference. Run on the GPU box:
    python tools/probe/gen_inline_repro.py [extra hipcc flags for the inline build ...]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_models as GM  # noqa: E402
from exmc_amd import codegen as cg, sampler  # noqa: E402


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def build(flags):
    os.environ["EXMC_GEN_EXTRA_FLAGS"] = " ".join(flags)
    spec = cg.compile_ir(GM.zoo_ir(), default_init=GM.ZOO_INIT)
    os.environ["EXMC_GEN_EXTRA_FLAGS"] = ""
    return spec


def resources(so):
    """VGPR / SGPR / scratch of the logp_grad and multi_step kernels from the code object notes."""
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", so], capture_output=True, text=True).stdout
    except OSError:
        return "llvm-readelf missing"
    rows, cur = [], {}
    for ln in out.splitlines():
        ln = ln.strip()
        for key in (".name:", ".vgpr_count:", ".sgpr_count:", ".vgpr_spill_count:", ".sgpr_spill_count:",
                    ".private_segment_fixed_size:"):
            if ln.startswith(key):
                cur[key] = ln.split(":", 1)[1].strip()
        if ln.startswith(".wavefront_size:") and cur:
            rows.append(cur); cur = {}
    keep = [r for r in rows if any(k in r.get(".name:", "") for k in ("logp_grad_kernel", "multi_step_kernel", "warmup_kernel"))]
    return "\n".join("   %s vgpr %s sgpr %s vgpr_spill %s sgpr_spill %s scratch %s" % (
        r.get(".name:", "?")[:60], r.get(".vgpr_count:"), r.get(".sgpr_count:"), r.get(".vgpr_spill_count:"),
        r.get(".sgpr_spill_count:"), r.get(".private_segment_fixed_size:")) for r in keep)


def probe(tag, spec):
    comp = sampler.compile(spec)
    d, Cn = spec.d, 256
    q0 = spec.to_unconstrained(GM.ZOO_INIT)
    rng = np.random.default_rng(3)
    print("== %s  (%s)" % (tag, spec.lib_path))
    print(resources(spec.lib_path))
    bad_lane = bad_run = 0
    ref = None
    for trial in range(6):
        q = np.ascontiguousarray(np.tile(q0 + 0.05 * rng.normal(size=d) * (trial > 0), (Cn, 1)))
        lp, g = np.zeros(Cn), np.zeros((Cn, d))
        for rep in range(2):
            comp.check(comp.L.exmc_hip_logp_grad_host(comp.h, _dp(q), Cn, 1, _dp(lp), _dp(g)))
            lane_diff = int(np.sum(lp != lp[0]) + np.sum(np.any(g != g[0], axis=1)))
            bad_lane += lane_diff
            if rep == 0:
                first = (lp.copy(), g.copy())
            else:
                bad_run += int(not (np.array_equal(first[0], lp) and np.array_equal(first[1], g)))
        if trial == 0:
            ref = (lp[0], g[0].copy())
    print("   logp_grad: chains differing from chain 0 (same input): %d ; repeat runs that differ: %d" % (bad_lane, bad_run))
    # leapfrog chains: 64 identical chains x 24 steps
    n, Cm = 24, 64
    p = np.ascontiguousarray(np.tile(rng.normal(size=d), (Cm, 1)))
    q = np.ascontiguousarray(np.tile(q0, (Cm, 1)))
    g = np.ascontiguousarray(np.tile(ref[1], (Cm, 1)))
    outs = []
    for rep in range(3):
        hq = np.zeros((Cm, n, d)); hp = np.zeros((Cm, n, d)); hg = np.zeros((Cm, n, d)); hl = np.zeros((Cm, n))
        comp.check(comp.L.exmc_hip_multi_step_host(comp.h, _dp(q), _dp(p), _dp(g), 0.3, _dp(np.ones(d)), n, Cm, 1,
                                                   _dp(hq), _dp(hp), _dp(hl), _dp(hg)))
        outs.append((hq, hl))
        print("   multi_step rep %d: chains differing from chain 0: %d" % (rep, int(np.sum(np.any(hq != hq[0], axis=(1, 2))))))
    print("   multi_step repeat runs identical: %s" % all(np.array_equal(outs[0][0], o[0]) for o in outs[1:]))
    eps = [sampler.warmup(comp, GM.ZOO_INIT, dict(num_warmup=nw, seed=17))["epsilon"] for nw in (0, 0, 20, 20, 80, 80)]
    print("   warmup eps (0,0,20,20,80,80 iterations):", ["%.9g" % e for e in eps])
    return ref, outs[0], eps


def main():
    extra = sys.argv[1:]
    called = probe("called form (shipped)", build([]))
    inl = probe("inlined exp/log, -O3", build(["-DEXMC_GEN_INLINE_MATH"] + extra))
    print("inline == called: logp %s grad %s leapfrog %s warmup-eps %s" % (
        called[0][0] == inl[0][0], np.array_equal(called[0][1], inl[0][1]),
        np.array_equal(called[1][0], inl[1][0]), called[2] == inl[2]))
    inl1 = probe("inlined exp/log, -O1", build(["-DEXMC_GEN_INLINE_MATH", "-O1"] + extra))
    print("inline -O1 == called: logp %s grad %s leapfrog %s warmup-eps %s" % (
        called[0][0] == inl1[0][0], np.array_equal(called[0][1], inl1[0][1]),
        np.array_equal(called[1][0], inl1[1][0]), called[2] == inl1[2]))


if __name__ == "__main__":
    main()
