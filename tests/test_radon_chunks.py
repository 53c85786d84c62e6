"""CPU: the checker's 64-lane radon order (observation i on lane i mod 64, the likelihood / floor /
z^2 totals per lane in slot order and then the lanes; a county's sum in index order --
exmc_oracle.c logp_radon) against the plain left-to-right order and under a reordering of the
counties. Sum order changes roundings, never values: everything agrees to ~1e-13 relative.
(The file keeps its name from rounds 1-2, when the 64-lane layout walked counties in chunks.)"""
import numpy as np

from exmc_amd import models

import oracle as O


def _survey_like(seed=17):
    rng = np.random.default_rng(seed)
    J, N = 85, 919
    sizes = np.concatenate([rng.integers(2, 7, size=J - 6), [14, 25, 46, 52, 105, 116]])
    while sizes.sum() > N:
        sizes[int(np.argmax(sizes[:J - 6]))] -= 1
    while sizes.sum() < N:
        sizes[int(rng.integers(0, J - 6))] += 1
    start = np.concatenate([[0], np.cumsum(sizes)])
    u = rng.normal(0.0, 0.5, size=J)
    county = np.repeat(np.arange(J), sizes)
    floor = (rng.uniform(size=N) < 0.2).astype(float)
    y = 1.4 + 0.7 * u[county] - 0.7 * floor + 0.7 * rng.normal(size=N)
    return u, start, floor, y


def test_chunked_sums_equal_plain_sums_to_rounding():
    data = _survey_like()
    assert np.diff(data[1]).max() == 116 and -(-919 // 64) == 15    # 15 slots per lane; one county of 116
    for sort_counties in (True, False):
        spec = models.radon(data, sort_counties=sort_counties)
        om = O.model_for(spec)
        rng = np.random.default_rng(2)
        for _ in range(5):
            q = rng.normal(size=spec.d) * 0.4
            lp1, g1 = om.logp_grad(q, O.Cfg(1, 1))
            lp32, g32 = om.logp_grad(q, O.Cfg(1, 32))
            lp64, g64 = om.logp_grad(q, O.Cfg(1, 64))
            assert abs(lp64 - lp1) <= 1e-12 * abs(lp1) and abs(lp32 - lp1) <= 1e-12 * abs(lp1)
            assert np.allclose(g64, g1, rtol=1e-11, atol=1e-11) and np.allclose(g32, g1, rtol=1e-11, atol=1e-11)


def test_county_order_is_only_a_relabelling():
    data = _survey_like()
    a = models.radon(data, sort_counties=True)
    b = models.radon(data, sort_counties=False)
    oa, ob = O.model_for(a), O.model_for(b)
    rng = np.random.default_rng(4)
    qb = rng.normal(size=b.d) * 0.3
    # the same point expressed in a's variable order
    qa = np.array([qb[b.var_names.index(n)] for n in a.var_names])
    la, ga = oa.logp_grad(qa, O.Cfg(1, 64))
    lb, gb = ob.logp_grad(qb, O.Cfg(1, 64))
    assert abs(la - lb) <= 1e-12 * abs(lb)
    gb_in_a = np.array([gb[b.var_names.index(n)] for n in a.var_names])
    assert np.allclose(ga, gb_in_a, rtol=1e-11, atol=1e-11)


def test_more_than_1024_observations_go_through_the_generator():
    """The built-in radon kind holds at most 1024 observations (16 slots per lane); models.radon
    routes larger data through the generator (codegen.radon_ir -> a lane layout of its own) instead
    of refusing it. The generated text equals the hand-written checker model to 1e-12."""
    import gen_checker as GC
    import gen_models as GM
    data = models.radon_data(seed=3, n_counties=85, n_obs=1500)
    spec = models.radon(data)
    hand = models.radon(data, builtin=True)
    assert hasattr(spec, "gen") and spec.d == hand.d == 90
    assert sorted(spec.var_names) == sorted(hand.var_names) and spec.default_init == hand.default_init
    om = O.model_for(hand)
    idx = GM.to_spec_order(spec.gen, hand)
    rng = np.random.default_rng(8)
    q0 = hand.to_unconstrained(hand.default_init)
    for t in range(20):
        q = q0 + 0.3 * rng.normal(size=hand.d) * (1 + t % 3)
        lp_o, g_o = om.logp_grad(q, O.Cfg(0, 1))
        lp_g, g_g = GC.logp_grad(spec.gen, q[idx], lanes=64)
        assert abs(lp_g - lp_o) <= 1e-12 * max(1.0, abs(lp_o))
        assert np.all(np.abs(g_g - g_o[idx]) <= 1e-12 * max(1.0, np.max(np.abs(g_o))))
