set -e
mkdir -p gpurun_out/r2_mig
for steps in 4 20; do
for mig in 0 1; do
  EXMC_HIP_MIGRATE_STATS=1 EXMC_HIP_MIGRATE=$mig timeout -k 10 300 python bench.py --model sv --no-multi-step --no-cpu --steps $steps > gpurun_out/r2_mig/b_sv_mig${mig}_$steps.json 2> gpurun_out/r2_mig/b_sv_mig${mig}_$steps.err
  grep migrate gpurun_out/r2_mig/b_sv_mig${mig}_$steps.err | tail -1 || true
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2_mig/b_sv_mig${mig}_$steps.json").read().strip().splitlines()[-1])
print("steps=$steps mig=$mig", d["value"], d["ess_wall_s"]["sampling"], d["roofline"]["kernel_ms"])
PY
done
done
