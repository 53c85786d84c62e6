#!/bin/bash
# The measurement set of round 4, everything under gpurun_out/$1/:
#   pmc_es/, pmc_sv/        the five counter passes + kernel stats + bench line (tools/gpu_pmc_session.sh)
#                           of `bench.py --no-sv-leg` and `bench.py --model sv`
#   driver_cmd_run<i>.json  the DRIVER'S command, `python3 bench.py --gpus 1 --steps 20 --warmup 5`, three times
#   bench_<model>.json      radon, logistic (CPU legs), the generated forms of the four models
tag=${1:-r4_end}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
bash tools/gpu_pmc_session.sh $tag/pmc_es --no-sv-leg > $out/pmc_es.log 2>&1 || { tail -5 $out/pmc_es.log; exit 1; }
bash tools/gpu_pmc_session.sh $tag/pmc_sv --model sv > $out/pmc_sv.log 2>&1 || { tail -5 $out/pmc_sv.log; exit 1; }
for i in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/driver_cmd_run$i.json 2> $out/driver_cmd_run$i.err || { tail -5 $out/driver_cmd_run$i.err; exit 1; }
done
for m in radon logistic gen_sv gen_radon gen_logistic gen_eight_schools; do
  python3 bench.py --model $m > $out/bench_$m.json 2> $out/bench_$m.err || { tail -5 $out/bench_$m.err; exit 1; }
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$out/driver_cmd_run*.json")) + sorted(glob.glob("$out/bench_*.json")):
    d = json.load(open(f))
    def show(d):
        print("%-18s %.3e lf/s  kernel %.1f ms  frac %.3f  adapt %.3f s  ess/s %.3e  rhat %.6f (%s, stats route %.6f)  gpu/cpu %s" % (
            d["config"]["workload"][:18], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
            d["ess_wall_s"]["adaptation"], d["ess_per_s"], d["rhat_max"], d["rhat_routes_agree"], d["rhat_max_from_chain_stats"],
            {k: round(v, 1) for k, v in d.get("gpu_over_cpu", {}).items()}))
    show(d)
    for v in d.get("models", {}).values():
        show(v)
PY
