#!/bin/bash
# sv at the protocol with chain migration off / on (same box): gpurun_out/$1/
out=gpurun_out/${1:-sv_mig}; mkdir -p $out
for mig in 0 1 1; do
  EXMC_HIP_MIGRATE_STATS=1 EXMC_HIP_MIGRATE=$mig timeout -k 10 300 python bench.py --model sv --no-multi-step --no-cpu > $out/bench_mig$mig.json 2> $out/bench_mig$mig.err || exit 1
  grep migrate $out/bench_mig$mig.err | tail -1
  python - <<PY
import json
d=json.loads(open("$out/bench_mig$mig.json").read().strip().splitlines()[-1])
print("mig=$mig %.4g lf/s, sampling %.3f s" % (d["value"], d["ess_wall_s"]["sampling"]))
PY
done
