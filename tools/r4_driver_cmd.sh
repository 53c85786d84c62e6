#!/bin/bash
# The driver's exact bench command, N times, lines kept under gpurun_out/$1/ (VERDICT r3 item 1a).
tag=${1:-r4_driver_cmd}; n=${2:-3}; out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $n); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/run$i.json 2> $out/run$i.err || { tail -5 $out/run$i.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open("$out/run$i.json"))
sv = d["models"]["sv"]
print("run $i: es %.4e lf/s rhat %.6f | sv %.4e lf/s kernel %.1f ms rhat %.6f stats-route %.6f lf %d" % (
    d["value"], d["rhat_max"], sv["value"], sv["roofline"]["kernel_ms"], sv["rhat_max"],
    sv["rhat_max_from_chain_stats"], sv["roofline"]["leapfrogs_per_launch"]))
PY
done
