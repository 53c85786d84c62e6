"""The workgroup form of a generated lane layout's sampling kernel (exmc_nuts.hpp nuts_kernel_wg around one LDS image of
the layout's tables; EXMC_GEN_WG, decided by the generator) and the pair layout of wide per-unit rows
(codegen_lanes.py PAIR_MIN_COLS: columns 2p, 2p + 1 interleaved over the units, one ds_read_b128 per pair): the 500 x 20
logistic regression and smaller / ragged regressions compiled from Builder nodes (more than 20 dimensions each: below that a
model also gets a one-lane layout, whose straight-line body over hundreds of observations takes minutes to compile), chain counts that leave wavefronts and
lane groups of a workgroup empty, both forms forced (EXMC_HIP_NUTS_WG) and the dispatcher's own choice -- against the
checker running the same generated text on the CPU, bit for bit. The shared warmup of these models runs in the one-chain
form, which keeps its own LDS image of the tables (CustomSplit::stage): covered by the sample/3 case."""
import numpy as np
import pytest

import gen_checker as GC
import oracle as O
from exmc_amd import codegen as cg, models, sampler

pytestmark = pytest.mark.gpu

KEYS = ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy")
_cache = {}


def _compiled(n_obs, k):
    key = (n_obs, k)
    if key not in _cache:
        X, y = models.logistic_data(seed=300 + n_obs + k, n=n_obs, k=k)
        hand = models.logistic(X, y) if k == 20 else None
        ir = cg.logistic_ir(X, y)
        init = hand.default_init if hand is not None else None
        if init is None:
            gen0 = cg.generate(ir, ncp=True, lanes=16, waves_per_simd=2)
            init = {n: 0.0 for n in gen0.var_names}
        spec = cg.compile_ir(ir, ncp=True, name="gen_lg_%d_%d" % key, default_init=init, lanes=16, waves_per_simd=2)
        _cache[key] = (spec, sampler.compile(spec), GC.model(spec.gen, 16))
    return _cache[key]


@pytest.mark.parametrize("n_obs,k,n_chains,wg_expected", [(500, 20, 37, 1), (500, 20, 5, 1), (333, 24, 70, 1), (48, 20, 9, 0)])
def test_generated_workgroup_form_equals_the_checker(monkeypatch, hip, n_obs, k, n_chains, wg_expected):
    spec, comp, om = _compiled(n_obs, k)
    assert ("#define EXMC_GEN_WG %d" % wg_expected) in spec.gen.header
    assert ("EXMC_GEN_LT2(" in spec.gen.header) == (k >= 8)
    q0 = spec.to_unconstrained(spec.default_init)
    rng = np.random.default_rng(n_obs)
    im = np.ascontiguousarray(rng.uniform(0.5, 2.0, size=spec.d))
    eps, n_draws = 0.15, 10
    want = {key: [] for key in KEYS}
    for c in range(n_chains):
        t, _ = O.sample_tuned(om, eps, im, q0, num_samples=n_draws, max_tree_depth=6, seed=21 + 7919 * c, cfg=O.Cfg(1, 16))
        for key in KEYS:
            want[key].append(t[key])
    want = {key: np.stack(v) for key, v in want.items()}
    opts = dict(num_warmup=0, num_samples=n_draws, seed=21, lanes_per_chain=16, max_tree_depth=6)
    for form in ("1", "0", None):
        if form is None:
            monkeypatch.delenv("EXMC_HIP_NUTS_WG", raising=False)
        else:
            monkeypatch.setenv("EXMC_HIP_NUTS_WG", form)
        _, _, extra = sampler.sample_compiled_tuned(comp, dict(epsilon=eps, inv_mass=im, chol_cov=None), spec.default_init,
                                                    opts, num_chains=n_chains)
        for key in KEYS:
            assert np.array_equal(want[key], extra["raw"][key], equal_nan=True), (form, n_obs, k, n_chains, key)
        assert int(want["n_steps"].sum()) == extra["total_leapfrogs"]


def test_generated_wide_rows_warmup_bit_exact(hip, monkeypatch):
    """The shared warmup of the 333 x 24 regression: the one-chain form reads its tables from the LDS image it stages
    (CustomSplit::stage) -- tuning against the checker's wave_split model, whichever sampling form is forced."""
    spec, comp, om = _compiled(333, 24)
    q0 = spec.to_unconstrained(spec.default_init)
    # the shared warmup of a 16-lane layout runs in the one-chain form (its own summation order: the checker's
    # wave_split model); the draws that follow run in the sampling layout
    oms = GC.model(spec.gen, 16, wave_split=True)
    ost = O.warmup(oms, init_q=q0, num_warmup=150, seed=5, cfg=O.Cfg(1, 16))
    for form in ("0", "1"):
        monkeypatch.setenv("EXMC_HIP_NUTS_WG", form)
        tuning = sampler.warmup(comp, spec.default_init, dict(num_warmup=150, seed=5))
        assert tuning["epsilon"] == ost.step_size
        assert np.array_equal(tuning["inv_mass"], np.array(ost.inv_mass[:spec.d]))
