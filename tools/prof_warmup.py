import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from exmc_amd import models, sampler
spec = models.eight_schools()
comp = sampler.compile(spec)
for nw in (2, 1000):
    print("== warmup", nw, file=sys.stderr, flush=True)
    t = sampler.warmup(comp, spec.default_init, dict(num_warmup=nw, seed=42, lanes_per_chain=16))
    print(t["epsilon"], comp.last_kernel_ms, file=sys.stderr, flush=True)
