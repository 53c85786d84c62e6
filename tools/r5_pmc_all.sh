#!/bin/bash
# Counter passes (tools/gpu_pmc_session.sh) for every workload profiles/pmc_traffic.json keys, on ONE tree:
#   gpurun --timeout 1100 -- 'bash tools/r5_pmc_all.sh <tag>'
# then, on the same tree:  for m in es sv logistic radon: python tools/pmc_summary.py gpurun_out/<tag>/pmc_<m> profiles/<name>/<m>
# (es also with --kernel multi_step_kernel) and python tools/pmc_table_update.py <key> profiles/<name>/<m>
tag=${1:-r5_pmc}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
bash tools/gpu_pmc_session.sh $tag/pmc_es --no-sv-leg --no-extra-legs > $out/pmc_es.log 2>&1 || { tail -5 $out/pmc_es.log; exit 1; }
echo es done
for m in radon logistic sv; do
  bash tools/gpu_pmc_session.sh $tag/pmc_$m --model $m > $out/pmc_$m.log 2>&1 || { tail -5 $out/pmc_$m.log; exit 1; }
  echo $m done
done
