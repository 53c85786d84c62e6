#!/bin/bash
# bench lines of the other BASELINE models at the 1000-draw protocol: gpurun -- 'bash tools/gpu_models_session.sh <tag> [models...]'
tag=${1:-models}; shift; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -2 $out/pytest.log
for m in "$@"; do
  python bench.py --model $m > $out/bench_$m.json 2> $out/bench_$m.err || tail -3 $out/bench_$m.err
  python - <<PY
import json
d=json.load(open("$out/bench_$m.json"))
print("$m", "lf/s %.3e" % d["value"], "kernel_ms %.1f" % d["roofline"]["kernel_ms"], "adapt %.3f s" % d["ess_wall_s"]["adaptation"],
      "ess/s %.3e" % d["ess_per_s"], "rhat_max %.4f" % d["rhat_max"], "frac %.4f" % d["roofline"]["frac"],
      "gpu/cpu", d.get("gpu_over_cpu"))
PY
done
