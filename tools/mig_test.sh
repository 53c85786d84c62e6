set -e
mkdir -p gpurun_out/r2_mig
# parity first at a size where migration triggers: forced on, small chain count is not enough to share SIMDs, so use 2048 x 30
EXMC_HIP_MIGRATE=1 timeout -k 10 300 python - <<'PY' > gpurun_out/r2_mig/parity.log 2>&1
import sys, numpy as np
sys.path.insert(0, "tests")
import oracle as O, test_golden_traces as TG
from exmc_amd import models, sampler
spec = models.sv(TG.GOLD["sv_returns"]); comp = sampler.compile(spec); om = O.model_for(spec)
opts = dict(num_warmup=150, num_samples=30, seed=5)
tuning = sampler.warmup(comp, spec.default_init, opts)
import os
res = {}
for mig in ("0", "1"):
    os.environ["EXMC_HIP_MIGRATE"] = mig
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=2048)
    res[mig] = extra["raw"]
for k in ("draws", "n_steps", "tree_depth", "energy", "accept_prob", "divergent"):
    assert np.array_equal(res["0"][k], res["1"][k]), k
print("migrate on == off over 2048 chains x 30 draws:", res["1"]["n_steps"].sum(), "leapfrogs")
q0 = spec.to_unconstrained(spec.default_init)
for c in (0, 7, 2047):
    t, _ = O.sample_tuned(om, tuning["epsilon"], tuning["inv_mass"], q0, num_samples=30, seed=5 + 7919 * c, cfg=O.Cfg(1, 64))
    assert np.array_equal(t["draws"], res["1"]["draws"][c]), c
print("checker parity ok")
PY
cat gpurun_out/r2_mig/parity.log
for mig in 0 1; do
  EXMC_HIP_MIGRATE=$mig timeout -k 10 300 python bench.py --model sv --no-multi-step --no-cpu > gpurun_out/r2_mig/bench_sv_mig$mig.json 2> gpurun_out/r2_mig/bench_sv_mig$mig.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2_mig/bench_sv_mig$mig.json").read().strip().splitlines()[-1])
print("mig=$mig", d["value"], d["ess_wall_s"]["sampling"], d["rhat_max"], d["roofline"]["kernel_ms"])
PY
done
