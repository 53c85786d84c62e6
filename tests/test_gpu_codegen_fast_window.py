"""The fast window of the generated layouts (exmc_models.hpp, EXMC_GEN_FAST_WINDOW): the lane function is
evaluated with the main paths of exp / log / log1p and a watch per argument; a wavefront in which ANY lane
saw an argument outside the domain (|x| > 700 or NaN for exp; anything but a positive normal number for
log) evaluates the exact form again. The other GPU tests of the generated layouts stay inside the domain
almost everywhere (the transforms clamp at +-200); here the positions are chosen to LEAVE it -- NaN,
infinities, 1e308, denormal scales -- in some chains of a wavefront and not in others, so that both the
re-evaluation and the lanes that did not need it are compared with the checker (the same generated text on
the CPU, general functions only), bit for bit and NaN for NaN."""
import ctypes as C

import numpy as np
import pytest

import gen_checker as GC
import gen_models as GM
import oracle as O
from exmc_amd import codegen as cg, sampler

pytestmark = pytest.mark.gpu


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


_cache = {}


def _compiled(which):
    if which not in _cache:
        if which == "schools_plate":
            init = {n: 0.0 for n in ["mu"] + ["theta_%d" % j for j in range(8)]}
            init["tau"] = 1.0
            spec = cg.compile_ir(GM.eight_schools_ir(), name="gen_eight_schools", default_init=init)
        elif which == "zoo_plate":
            spec = cg.compile_ir(GM.zoo_ir(), name="gen_zoo", default_init=GM.ZOO_INIT)
        elif which in ("zoo16", "walk16"):
            ir = GM.zoo_ir() if which == "zoo16" else GM.walk_ir()
            init = GM.ZOO_INIT if which == "zoo16" else GM.WALK_INIT
            spec = cg.compile_ir(ir, name=which, default_init=init, lanes=16)
        else:
            ir, ncp, hand, lanes = GM.baseline_pair(which)
            spec = cg.compile_ir(ir, ncp=ncp, name="gen_" + which, default_init=hand.default_init, lanes=lanes)
        lanes = spec.gen.lanes
        _cache[which] = (spec, sampler.compile(spec), GC.model(spec.gen, lanes), lanes)
    return _cache[which]


SPECIALS = [np.nan, np.inf, -np.inf, 1e308, -1e308, 1e5, -1e5, 705.0, -705.0, 745.2, -745.2, 5e-324, -5e-324,
            2.2250738585072014e-308, 1e-310, 199.99999, -199.99999, 200.0, -200.0, 200.00001, -200.00001]


@pytest.mark.parametrize("which", ["schools_plate", "zoo_plate", "sv", "radon", "logistic", "zoo16", "walk16"])
def test_positions_that_leave_the_fast_window_bit_exact(which, hip):
    spec, comp, om, lanes = _compiled(which)
    rng = np.random.default_rng(23)
    n = 512
    q0 = spec.to_unconstrained(spec.default_init)
    q = np.ascontiguousarray(q0[None, :] + rng.normal(size=(n, spec.d)) * 0.3)
    # three chains of every four get special values in one to three components: with 64 / lanes chains per
    # wavefront a wave holds both kinds (one chain per wave at 64 lanes: whole waves of either kind)
    hit = 0
    for c in range(n):
        if c % 4 == 0:
            continue
        k = int(rng.integers(1, 4))
        for i in rng.choice(spec.d, size=min(k, spec.d), replace=False):
            q[c, i] = SPECIALS[int(rng.integers(len(SPECIALS)))]
        hit += 1
    assert hit > n // 2
    lp = np.zeros(n)
    g = np.zeros((n, spec.d))
    comp.check(comp.L.exmc_hip_logp_grad_host(comp.h, _dp(q), n, lanes, _dp(lp), _dp(g)))
    cfg = O.Cfg(1, lanes)
    n_nonfinite = 0
    for c in range(n):
        olp, og = om.logp_grad(q[c], cfg)
        assert olp == lp[c] or (np.isnan(olp) and np.isnan(lp[c])), (which, c, olp, lp[c], q[c])
        assert np.array_equal(og, g[c], equal_nan=True), (which, c, q[c])
        n_nonfinite += not np.isfinite(olp)
    # the special values did reach the functions: some log-densities are not finite, and not all of them
    assert 0 < n_nonfinite < n


@pytest.mark.parametrize("spread", [6.0, 40.0])
@pytest.mark.parametrize("which", ["schools_plate", "zoo_plate", "sv", "radon", "logistic", "walk16"])
def test_generated_sample_from_a_hostile_start_bit_exact(which, spread, hip, monkeypatch):
    """Sampler.sample/3 on the generated kernels started far from the mode (every coordinate 6 or 40 units away
    on the unconstrained scale): first-leaf divergences, clamps, a step-size search that halves dozens of times
    -- the fast window is left and re-entered throughout. Bit for bit against the checker."""
    spec, comp, om, lanes = _compiled(which)
    rng = np.random.default_rng(41)
    q0 = spec.to_unconstrained(spec.default_init) + rng.normal(size=spec.d) * spread
    nw, ns = (40, 12) if which == "sv" else (60, 25)
    opts = dict(num_warmup=nw, num_samples=ns, seed=9, lanes_per_chain=lanes, max_tree_depth=8)
    # (the host side takes constrained init values by name; the position is handed over as it is instead of
    # inverting transforms, non-centred pairs and vector entries for a made-up point)
    monkeypatch.setattr(spec, "to_unconstrained", lambda _iv: q0.copy())
    _, stats = sampler.sample_compiled(comp, {"hostile": True}, opts)
    monkeypatch.undo()
    t, st = O.sample(om, init_q=q0, num_warmup=nw, num_samples=ns, seed=9, cfg=O.Cfg(1, lanes), max_tree_depth=8)
    assert stats["step_size"] == st.step_size or (np.isnan(stats["step_size"]) and np.isnan(st.step_size))
    raw = stats["raw"]
    for k in ("tree_depth", "n_steps", "divergent", "draws", "energy"):
        assert np.array_equal(raw[k][0], t[k], equal_nan=True), (which, spread, k)
