#!/usr/bin/env python3
"""Copy rocprofv3 summaries from gpurun_out/ (scratch) into profiles/<name>/ (tracked).

    python tools/save_profile.py <name> --stats gpurun_out/prof_x --bench gpurun_out/b.json \
        [--bench-prof gpurun_out/prof_bench.json] [--fetch gpurun_out/pmc_fetch] \
        [--write gpurun_out/pmc_write] [--kernel nuts_kernel]

Writes kernel_stats.csv, bench.json, bench_under_rocprof.json and pmc_<kernel>.json: the PMC
values of the largest dispatch of <kernel> (the timed launch), FETCH_SIZE / WRITE_SIZE in KB as
rocprofv3 reports them, plus hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 — the x2 on the read
side is the gfx950 correction MI355X_MICROARCH.md prescribes for wide coalesced streaming reads.
"""
import argparse
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    m = glob.glob(pattern, recursive=True)
    if not m:
        raise SystemExit("nothing matches %s" % pattern)
    return m[0]


def pmc_max(dirname, kernel, counter):
    f = one(os.path.join(dirname, "**", "*_counter_collection.csv"))
    rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    big = max(rows, key=lambda r: float(r["Counter_Value"]))
    return dict(value=float(big["Counter_Value"]), dispatches=len(rows), vgpr=big["VGPR_Count"],
                sgpr=big["SGPR_Count"], lds=big["LDS_Block_Size"], scratch=big["Scratch_Size"],
                grid=big["Grid_Size"], workgroup=big["Workgroup_Size"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--stats")
    ap.add_argument("--bench")
    ap.add_argument("--bench-prof")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--kernel", default="nuts_kernel")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles", a.name)
    os.makedirs(out, exist_ok=True)
    if a.stats:
        shutil.copy(one(os.path.join(a.stats, "**", "*_kernel_stats.csv")), os.path.join(out, "kernel_stats.csv"))
    if a.bench:
        shutil.copy(a.bench, os.path.join(out, "bench.json"))
    if a.bench_prof:
        shutil.copy(a.bench_prof, os.path.join(out, "bench_under_rocprof.json"))
    if a.fetch and a.write:
        fe = pmc_max(a.fetch, a.kernel, "FETCH_SIZE")
        wr = pmc_max(a.write, a.kernel, "WRITE_SIZE")
        pm = {"kernel": a.kernel, "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr,
              "hbm_bytes_timed_launch": (2.0 * fe["value"] + wr["value"]) * 1024.0,
              "note": "separate --pmc passes with --kernel-trace only; FETCH_SIZE doubled per "
                      "MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)"}
        json.dump(pm, open(os.path.join(out, "pmc_%s.json" % a.kernel), "w"), indent=1)
        print(json.dumps(pm))
    print("saved", out, sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
