#!/usr/bin/env python3
"""Summarise the PMC passes of tools/gpu_pmc_session.sh into profiles/<name>/:
    python tools/pmc_summary.py gpurun_out/<tag> profiles/<name> [--kernel nuts_kernel]
Writes pmc_<kernel>.json (FETCH/WRITE -> HBM bytes with the gfx950 FETCH_SIZE x2 correction),
pmc_sq_<kernel>.json (SQ counters of the timed launch, also per wave-pass when the bench line
carries the pass count), kernel_stats.csv, bench.json, bench_under_rocprof.json."""
import csv
import glob
import json
import os
import shutil
import sys


def one(pattern):
    m = glob.glob(pattern, recursive=True)
    return m[0] if m else None


def timed_launch(dirname, kernel):
    """counter -> value for the largest dispatch of `kernel` (the timed launch)."""
    f = one(os.path.join(dirname, "**", "*counter_collection.csv"))
    if not f:
        return {}
    rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
    by_disp = {}
    for r in rows:
        by_disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    if not by_disp:
        return {}
    return max(by_disp.values(), key=lambda c: sum(c.values()))


def main():
    src, dst = sys.argv[1], sys.argv[2]
    kernel = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "nuts_kernel"
    os.makedirs(dst, exist_ok=True)
    fe = timed_launch(os.path.join(src, "fetch"), kernel).get("FETCH_SIZE")
    wr = timed_launch(os.path.join(src, "write"), kernel).get("WRITE_SIZE")
    if fe is not None and wr is not None:
        json.dump({"kernel": kernel, "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr,
                   "hbm_bytes_timed_launch": (2.0 * fe + wr) * 1024.0,
                   "note": "separate --pmc passes with --kernel-trace only; FETCH_SIZE doubled per "
                           "MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)"},
                  open(os.path.join(dst, "pmc_%s.json" % kernel), "w"), indent=1)
    sq = {}
    for name in ("insts", "cycles", "f64"):
        sq.update(timed_launch(os.path.join(src, name), kernel))
    bench = None
    if os.path.exists(os.path.join(src, "bench.json")):
        bench = json.load(open(os.path.join(src, "bench.json")))
        shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench.json"))
    if os.path.exists(os.path.join(src, "bench_under_rocprof.json")):
        shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, "bench_under_rocprof.json"))
    st = one(os.path.join(src, "stats", "**", "*kernel_stats.csv"))
    if st:
        shutil.copy(st, os.path.join(dst, "kernel_stats.csv"))
    if sq:
        doc = {"kernel": kernel + ", timed launch", "counters": sq}
        lf = bench["roofline"]["leapfrogs_per_launch"] if bench else None
        if lf:
            doc["leapfrogs_per_launch"] = lf
            doc["per_leapfrog"] = {k: v / lf for k, v in sq.items()}
            waves = sq.get("SQ_WAVES")
            if waves:
                doc["note"] = ("per_leapfrog = counter / useful leapfrogs of the launch; a wave-pass serves the "
                               "chain groups of one wavefront (4 for eight_schools at 16 lanes per chain)")
        json.dump(doc, open(os.path.join(dst, "pmc_sq_%s.json" % kernel), "w"), indent=1)
    print("saved", dst, sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
