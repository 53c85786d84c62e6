#!/bin/bash
# The measurement set of round 6, everything under gpurun_out/$1/:
#   pytest.log               the whole GPU suite
#   driver_cmd_run<i>.json   the DRIVER'S command (python3 bench.py --gpus 1 --steps 20 --warmup 5): eight_schools + sv +
#                            logistic + radon legs, twice
#   force_dist.json          the same command with a one-rank nccl group and every collective made (--force-dist)
#   kernel_stats.csv         rocprofv3 --kernel-trace --stats of the driver's command
tag=${1:-r6_end}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for i in 1 2; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/driver_cmd_run$i.json 2> $out/driver_cmd_run$i.err || { tail -5 $out/driver_cmd_run$i.err; exit 1; }
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --force-dist > $out/force_dist.json 2> $out/force_dist.err || { tail -5 $out/force_dist.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $out/bench_under_rocprof.json 2> $out/stats.err
cp $(find $out/stats -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv 2>/dev/null
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$out/driver_cmd_run*.json")) + ["$out/force_dist.json"]:
    d = json.load(open(f))
    def show(d):
        print("%-18s %.3e lf/s  kernel %.1f ms  frac %.3f  issue %s  adapt %.3f s  ess/s %.3e bulk %.3e  rhat %.6f (%s)  traffic %s  gpu/cpu %s %s" % (
            d["config"]["workload"][:18], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
            round(d.get("roofline_issue", {}).get("frac", 0), 3), d["ess_wall_s"]["adaptation"], d["ess_per_s"], d["ess_bulk_per_s"],
            d["rhat_max"], d["rhat_routes_agree"], d["roofline"]["traffic"],
            {k: round(v, 1) for k, v in d.get("gpu_over_cpu", {}).items()}, d.get("collectives", "")))
    print(f)
    show(d)
    for v in d.get("models", {}).values():
        show(v)
PY
