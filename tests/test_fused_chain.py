"""B2' -- the fused-chain hook of the reference's speculative path (lib/exmc/nuts/tree.ex:613-653,
`leapfrog_chain_normal(q, p, inv_mass, k, signed_eps, mu, sigma)`): the checker's statement of it against
(i) a third, plain-Python statement written from batched_leapfrog.ex:50-101 and dist/normal.ex:15-24, (ii) the
checker's own multi_step on its N(0, 1) kind ("Output contract is identical in both branches", tree.ex:620), (iii) the
properties the reference tests for multi_step (prefix, nuts_test.exs:417-474; reversibility, :39-81), and (iv) the
acceptance band of the reference's test of the hook itself (test/nuts/fused_chain_diag_test.exs). CPU only; the HIP
entry point against this checker: tests/test_gpu_fused_chain.py."""
import math

import numpy as np
import pytest

import fused_chain_model as FCM
import oracle as O
import py_sampler as PS

F32 = lambda x: float(np.float32(x))  # noqa: E731


def plain_chain(q, p, im, k, eps, mu, sigma):
    """Third statement: python floats, left to right (Nx.sum on the BinaryBackend), libm log."""
    q, p = [float(x) for x in q], [float(x) for x in p]
    d = len(q)
    ss = max(sigma, F32(1.0e-30))                                   # normal.ex:18
    log_term = F32(math.log(F32(2.0 * math.pi))) + 2.0 * math.log(ss)   # normal.ex:19,22
    def density(q):
        tot, g = 0.0, []
        for x in q:
            z = (x - mu) / ss
            tot = tot + (-0.5 * (z * z + log_term))
            g.append((-0.5 * z + -0.5 * z) / ss)                    # reverse mode through z * z
        return tot, g
    _, g = density(q)                                               # the hook is handed no gradient (tree.ex:637)
    half = eps / 2.0                                                # batched_leapfrog.ex:64
    rows = ([], [], [], [])
    for _ in range(k):
        ph = [p[i] + half * g[i] for i in range(d)]
        q = [q[i] + eps * (im[i] * ph[i]) for i in range(d)]
        lp, g = density(q)
        p = [ph[i] + half * g[i] for i in range(d)]
        for r, v in zip(rows, (q, p, lp, g)):
            r.append(v)
    return tuple(np.array(r, dtype=np.float64).reshape((k, d) if i != 2 else (k,)) for i, r in enumerate(rows))


def _case(d, seed):
    rng = np.random.default_rng(seed)
    return rng.normal(size=d), rng.normal(size=d), rng.uniform(0.3, 3.0, size=d)


@pytest.mark.parametrize("d", [1, 2, 10, 63, 64, 65, 200, 256])
@pytest.mark.parametrize("mu,sigma,eps", [(0.0, 1.0, 0.1), (1.5, 0.7, -0.23), (-3.0, 12.0, 0.9)])
def test_checker_equals_the_plain_statement(d, mu, sigma, eps):
    q, p, im = _case(d, d)
    got = O.leapfrog_chain_normal(q, p, im, 9, eps, mu, sigma, O.Cfg(0, 1))
    exp = plain_chain(q, p, im, 9, eps, mu, sigma)
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("d", [1, 7, 64, 256])
@pytest.mark.parametrize("cfg", [(0, 1), (1, 1), (1, 16), (1, 64)])
def test_same_rows_as_multi_step_of_the_normal_kind(d, cfg):
    """tree.ex:620-621: both branches of do_dispatch return the same thing. At mu = 0, sigma = 1 the chain IS
    multi_step_fn of the checker's N(0, 1) kind started from that kind's own gradient, bit for bit, in both
    numeric modes and every lane layout."""
    q, p, im = _case(d, 100 + d)
    c = O.Cfg(*cfg)
    m = O.std_normal(d)
    _, g = m.logp_grad(q, c)
    for eps in (0.17, -0.17):
        a = m.multi_step(q, p, g, eps, im, 12, c)
        b = O.leapfrog_chain_normal(q, p, im, 12, eps, 0.0, 1.0, c)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_lane_layout_and_deterministic_log_agree_to_rounding():
    """The mode the GPU is compared with bit for bit (deterministic log, 64-lane sums) against the reference's order:
    the rows are elementwise and do not move at all; logp moves by the rounding of one sum and of log(sigma)."""
    q, p, im = _case(200, 5)
    a = O.leapfrog_chain_normal(q, p, im, 20, 0.05, 0.4, 2.5, O.Cfg(0, 1))
    b = O.leapfrog_chain_normal(q, p, im, 20, 0.05, 0.4, 2.5, O.Cfg(1, 64))
    for i in (0, 1, 3):
        assert np.array_equal(a[i], b[i])
    assert np.max(np.abs(a[2] - b[2]) / np.abs(a[2])) < 1e-13


def test_prefix_property_and_zero_steps():
    """nuts_test.exs:417-474 for the hook: 32 steps start with the 16 steps."""
    q, p, im = _case(10, 3)
    a = O.leapfrog_chain_normal(q, p, im, 32, 0.1, 0.0, 1.0)
    b = O.leapfrog_chain_normal(q, p, im, 16, 0.1, 0.0, 1.0)
    for x, y in zip(a, b):
        assert np.array_equal(x[:16], y)
    z = O.leapfrog_chain_normal(q, p, im, 0, 0.1, 0.0, 1.0)
    assert z[0].shape == (0, 10) and z[2].shape == (0,)


def test_signed_step_size_walks_back():
    """nuts_test.exs:39-81 (reversibility): the chain continued with -signed_eps from its last row returns to the start."""
    q, p, im = _case(10, 4)
    fq, fp, _, _ = O.leapfrog_chain_normal(q, p, im, 25, 0.08, 0.5, 1.3)
    bq, bp, _, _ = O.leapfrog_chain_normal(fq[-1], fp[-1], im, 25, -0.08, 0.5, 1.3)
    assert np.allclose(bq[-1], q, atol=1e-10) and np.allclose(bp[-1], p, atol=1e-10)
    assert np.allclose(bq[0], fq[-2], atol=1e-10)


def test_energy_is_conserved_along_the_chain():
    """nuts_test.exs:39-81: |H(step k) - H(step 1)| stays small at a small step size."""
    q, p, im = _case(10, 6)
    aq, ap, al, _ = O.leapfrog_chain_normal(q, p, im, 50, 0.01, 0.0, 1.0)
    h = -al + 0.5 * np.sum(ap * ap * im, axis=1)
    assert np.max(np.abs(h - h[0])) < 1e-3


def test_sigma_guard_and_bad_sizes():
    """normal.ex:18: sigma below 1e-30 (an f32 literal) is replaced, not divided by."""
    q, p, im = _case(3, 8)
    a = O.leapfrog_chain_normal(q * 1e-31, p * 1e-31, im, 4, 1e-62, 0.0, 0.0)
    b = plain_chain(q * 1e-31, p * 1e-31, im, 4, 1e-62, 0.0, 0.0)
    for x, y in zip(a, b):
        assert np.array_equal(x, y) and np.all(np.isfinite(x))
    with pytest.raises(ValueError):
        O.leapfrog_chain_normal(np.zeros(257), np.zeros(257), np.ones(257), 1, 0.1, 0.0, 1.0)     # tree.ex:636
    with pytest.raises(ValueError):
        O.leapfrog_chain_normal(q, p, im, -1, 0.1, 0.0, 1.0)


def test_fused_chain_diag_band():
    """test/nuts/fused_chain_diag_test.exs:54-58,85-90,123-140: x ~ N(0, 1), seed 42, 200 warmup + 1000 draws through
    the fused chain; posterior variance in [0.7, 1.3]. Here additionally: the run whose leapfrog steps come out of the
    hook in dispatches of 32 is the run of the plain step function, draw for draw."""
    fused = FCM.FusedChainModel(1, 0.0, 1.0, lambda *a: O.leapfrog_chain_normal(*a))
    tr, st = PS.sample(fused, num_warmup=200, num_samples=1000, seed=42)
    xs = tr["draws"][:, 0]
    var = float(np.mean((xs - xs.mean()) ** 2))
    assert 0.7 <= var <= 1.3, var
    assert abs(xs.mean()) < 0.2
    assert fused.dispatches < fused.steps          # the point of the hook: fewer dispatches than steps
    plain, st2 = PS.sample(O.std_normal(1), num_warmup=200, num_samples=1000, seed=42)
    assert np.array_equal(tr["draws"], plain["draws"]) and np.array_equal(tr["n_steps"], plain["n_steps"])
    assert st["step_size"] == st2["step_size"]


def test_product_entry_point_has_no_cpu_fallback():
    """the HIP entry point refuses without a device (no route through the checker): EXMC_ERR_NO_DEVICE, loudly."""
    from exmc_amd import _lib, fused_chain
    if _lib.load().exmc_hip_device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(_lib.ExmcHipError, match="no HIP device"):
        fused_chain.leapfrog_chain_normal(np.zeros(3), np.zeros(3), np.ones(3), 2, 0.1, 0.0, 1.0)
    import inspect
    assert "oracle" not in inspect.getsource(fused_chain)
