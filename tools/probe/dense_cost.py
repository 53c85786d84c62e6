"""Cost of the dense-mass operations in isolation: sv with M^-1 = I as a dense matrix against the
same chains under the diagonal identity mass (identical trajectories, so identical leapfrog counts)."""
import sys, time
import numpy as np
sys.path.insert(0, "tests")
import test_golden_traces as TG
from exmc_amd import models, sampler

spec = models.sv(TG.GOLD["sv_returns"])
comp = sampler.compile(spec)
d = spec.d
opts = dict(num_warmup=0, num_samples=int(sys.argv[1]) if len(sys.argv) > 1 else 40, seed=5)
eps = 0.02
for name, tuning in (("diag", dict(epsilon=eps, inv_mass=np.ones(d), chol_cov=None)),
                     ("dense", dict(epsilon=eps, inv_mass=np.eye(d), cov=np.eye(d), chol_cov=np.eye(d)))):
    for rep in range(2):
        t0 = time.perf_counter()
        _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=2048)
        dt = time.perf_counter() - t0
    lf = int(extra["raw"]["n_steps"].sum())
    ms = comp.L.exmc_hip_last_kernel_ms(comp.h)
    print("%s: %d leapfrogs, kernel %.1f ms, %.3g lf/s (wall %.2f s), mean lf/draw %.1f" % (name, lf, ms, lf / ms * 1e3, dt, lf / 2048 / opts["num_samples"]))
