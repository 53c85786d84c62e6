#!/bin/bash
# The measurement set of a round's end, everything under gpurun_out/$1/:
#   pmc_es/, pmc_sv/   the five counter passes + kernel stats + bench line (tools/gpu_pmc_session.sh)
#   bench.json         the driver's command (eight_schools line with the sv leg, CPU legs)
#   bench_<model>.json radon, logistic (CPU legs), the generated forms of sv / radon / logistic
tag=${1:-r3_end}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
bash tools/gpu_pmc_session.sh $tag/pmc_es --no-sv-leg > $out/pmc_es.log 2>&1 || { tail -5 $out/pmc_es.log; exit 1; }
bash tools/gpu_pmc_session.sh $tag/pmc_sv --model sv > $out/pmc_sv.log 2>&1 || { tail -5 $out/pmc_sv.log; exit 1; }
python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
for m in radon logistic gen_sv gen_radon gen_logistic gen_eight_schools; do
  python bench.py --model $m > $out/bench_$m.json 2> $out/bench_$m.err || { tail -5 $out/bench_$m.err; exit 1; }
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bench*.json")):
    d = json.load(open(f))
    def show(d):
        print("%-18s %.3e lf/s  kernel %.1f ms  frac %.3f  adapt %.3f s  ess/s %.3e  rhat %.3f  gpu/cpu %s" % (
            d["config"]["workload"][:18], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
            d["ess_wall_s"]["adaptation"], d["ess_per_s"], d["rhat_max"],
            {k: round(v, 1) for k, v in d.get("gpu_over_cpu", {}).items()}))
    show(d)
    for v in d.get("models", {}).values():
        show(v)
PY
