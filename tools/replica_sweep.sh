mkdir -p gpurun_out/r2_rep
for r in 8 32 96 256; do
  for k in 1 2; do
  EXMC_HIP_WARMUP_REPLICAS=$r python bench.py --no-cpu --no-multi-step --steps 2 > gpurun_out/r2_rep/b_$r.json 2> gpurun_out/r2_rep/b_$r.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2_rep/b_$r.json").read().strip().splitlines()[-1])
print("replicas=$r adaptation %.3f ms" % (d["ess_wall_s"]["adaptation"]*1e3))
PY
  done
done
