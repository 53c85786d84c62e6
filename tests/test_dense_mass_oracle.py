"""opts[:dense_mass] in the CPU checker: the dense Welford and its finalisation
(lib/exmc/nuts/mass_matrix.ex:27-35, 56-72, 105-140), the momentum p = L^-T z
(sampler.ex:412-427), M^-1 p by the dense product (leapfrog.ex:57-61) and the U-turn rule through
v = M^-1 rho. The reference has no test of this mode and, for d >= 2, raises inside its first dense
transition (see the note in oracle/exmc_oracle.c); these tests pin the documented intent: closed
forms, and d = 1 where dense and diagonal are the same sampler."""
import ctypes as C

import numpy as np
import pytest

import oracle as O


def test_dense_welford_and_finalize_closed_form():
    rng = np.random.default_rng(0)
    n, d = 40, 4
    A = rng.normal(size=(d, d))
    x = rng.normal(size=(n, d)) @ A
    cov, chol = np.zeros((d, d)), np.zeros((d, d))
    O.lib().exo_welford_dense_finalize(O.dptr(np.ascontiguousarray(x)), n, d, O.dptr(cov), O.dptr(chol))
    s = np.cov(x, rowvar=False, ddof=1)                       # m2 / (n - 1), mass_matrix.ex:112
    alpha = 5.0 / (n + 5.0)
    want = (1 - alpha) * s + alpha * np.diag(np.maximum(np.diag(s), 1e-6))   # :117-135
    assert np.allclose(cov, want, rtol=1e-12, atol=1e-14)
    assert np.allclose(chol @ chol.T, cov, rtol=1e-12) and np.allclose(chol, np.tril(chol))
    assert np.allclose(chol, np.linalg.cholesky(cov), rtol=1e-11)            # Nx.LinAlg.cholesky :138
    # fewer than three samples: identity (mass_matrix.ex:105-109)
    O.lib().exo_welford_dense_finalize(O.dptr(np.ascontiguousarray(x[:2])), 2, d, O.dptr(cov), O.dptr(chol))
    assert np.array_equal(cov, np.eye(d)) and np.array_equal(chol, np.eye(d))


def test_dense_momentum_solves_lt_p_equals_z():
    rng = np.random.default_rng(1)
    d = 10
    m = O.eight_schools()
    A = rng.normal(size=(d, d))
    cov = A @ A.T + d * np.eye(d)
    chol = np.ascontiguousarray(np.linalg.cholesky(cov))
    r1, r2 = O.Rng(), O.Rng()
    O.lib().exo_rng_seed(C.byref(r1), 11)
    O.lib().exo_rng_seed(C.byref(r2), 11)
    z = np.array([O.lib().exo_rng_normal(C.byref(r1), 0) for _ in range(d)])
    p = np.zeros(d)
    O.lib().exo_dense_momentum(m.h, O.dptr(chol), C.byref(r2), O.dptr(p), 0)
    assert np.allclose(chol.T @ p, z, rtol=1e-12, atol=1e-13)      # triangular_solve(L^T, z), sampler.ex:424
    assert (r1.a, r1.b) == (r2.a, r2.b)                             # exactly d draws consumed


def test_dense_product_and_uturn_rule():
    rng = np.random.default_rng(2)
    d = 6
    A = rng.normal(size=(d, d))
    cov = np.ascontiguousarray(A @ A.T + np.eye(d))
    x = rng.normal(size=d)
    out = np.zeros(d)
    O.lib().exo_dense_mass_times(O.dptr(cov), O.dptr(x), d, O.dptr(out))
    assert np.allclose(out, cov @ x, rtol=1e-13)
    for _ in range(200):
        rho, pl, pr = rng.normal(size=(3, d))
        v = cov @ rho
        want = (v @ pr < 0) or (v @ pl < 0)
        got = O.lib().exo_dense_check_uturn(O.dptr(cov), O.dptr(rho), O.dptr(pl), O.dptr(pr), d, O.Cfg(0, 1))
        assert bool(got) == bool(want)
    # a diagonal covariance reduces to the diagonal rule of tree.ex:1578-1588
    im = rng.uniform(0.5, 2.0, size=d)
    for _ in range(50):
        rho, pl, pr = rng.normal(size=(3, d))
        a = O.lib().exo_dense_check_uturn(O.dptr(np.ascontiguousarray(np.diag(im))), O.dptr(rho), O.dptr(pl),
                                          O.dptr(pr), d, O.Cfg(0, 1))
        b = O.lib().exo_check_uturn(O.dptr(rho), O.dptr(pl), O.dptr(pr), O.dptr(im), d, O.Cfg(0, 1))
        assert a == b


def test_dense_warmup_and_sampling_on_a_correlated_posterior():
    """eight_schools, dense: the window length is max(25, 10 d) = 100 (sampler.ex:682); the factor
    reproduces the covariance; sampling under it recovers the posterior the diagonal sampler finds."""
    m = O.eight_schools()
    q0 = np.zeros(10)
    st, cov, chol = O.warmup_dense(m, q0, num_warmup=1000, seed=42)
    assert 0.05 < st.step_size < 2.0
    assert np.allclose(chol @ chol.T, cov, rtol=1e-12) and np.all(np.diag(cov) > 0)
    assert np.array_equal(np.array(st.inv_mass[:10]), np.diag(cov))      # inv_mass_diag_out, sampler.ex:236-240
    assert np.abs(cov - np.diag(np.diag(cov))).max() > 1e-3              # off-diagonals were estimated
    draws = []
    for c in range(6):
        t, s2 = O.sample_tuned_dense(m, st.step_size, cov, chol, q0, num_samples=400, seed=42 + 7919 * c)
        draws.append(t["draws"])
        assert t["divergent"].mean() < 0.1
    x = np.concatenate(draws)
    td, _ = O.sample_chains(m, 6, init_q=q0, num_warmup=1000, num_samples=400, seed=42)
    xd = td["draws"].reshape(-1, 10)
    assert abs(x[:, 0].mean() - xd[:, 0].mean()) < 0.6           # mu
    assert abs(x[:, 1].mean() - xd[:, 1].mean()) < 0.35          # log tau


def test_dense_equals_diagonal_in_one_dimension():
    """d = 1: L = sqrt(var), p = z / L = z / sqrt(inv_mass), M^-1 p = var * p -- the same sampler
    (and the one case in which the reference's dense mode runs)."""
    m = O.std_normal(1)
    var = 1.7
    cov, chol = np.array([[var]]), np.array([[np.sqrt(var)]])
    t1, _ = O.sample_tuned_dense(m, 0.8, cov, chol, np.zeros(1), num_samples=200, seed=3)
    t2, _ = O.sample_tuned(m, 0.8, np.array([var]), np.zeros(1), num_samples=200, seed=3)
    assert np.allclose(t1["draws"], t2["draws"], rtol=1e-12, atol=1e-14)
    assert np.array_equal(t1["tree_depth"], t2["tree_depth"])


def test_dense_mass_lives_on_the_flat_vector():
    """The reference's covariance is that of its flat vector (sampler.ex:682-705), whose order is the
    string sort of the ids (point_map.ex:30-60): for sv that is nu, s_1, s_10, s_100, s_11, ...,
    sigma while the kernel order is s_1..s_100, sigma, nu. cov / chol are indexed by flat entries;
    the momentum solves L^T p = z on the flat vector, draw r belonging to flat entry r."""
    import test_golden_traces as TG
    from exmc_amd import models
    spec = models.sv(TG.GOLD["sv_returns"])
    m = O.model_for(spec)
    d = spec.d
    flat = np.asarray(m.flat_order())            # flat entry -> kernel dimension
    assert flat[0] == d - 1 and flat[-1] == d - 2 and flat[1] == 0     # nu first, sigma last, then s_1
    rng = np.random.default_rng(4)
    A = rng.normal(size=(d, d)) * 0.1
    cov = np.ascontiguousarray(A @ A.T + np.eye(d))
    chol = np.ascontiguousarray(np.linalg.cholesky(cov))
    r1, r2 = O.Rng(), O.Rng()
    O.lib().exo_rng_seed(C.byref(r1), 5)
    O.lib().exo_rng_seed(C.byref(r2), 5)
    z = np.array([O.lib().exo_rng_normal(C.byref(r1), 0) for _ in range(d)])
    p = np.zeros(d)
    O.lib().exo_dense_momentum(m.h, O.dptr(chol), C.byref(r2), O.dptr(p), 0)
    assert np.allclose(chol.T @ p[flat], z, rtol=1e-11, atol=1e-12)   # p in kernel order, the solve in flat order
    # a warmup window's covariance: its diagonal is the flat-order variance of the chain
    q0 = spec.to_unconstrained(spec.default_init)
    st, wcov, wchol = O.warmup_dense(m, q0, num_warmup=200, seed=3, cfg=O.Cfg(1, 64))
    assert np.array_equal(np.diag(wcov), np.asarray(st.inv_mass[:d])[flat])
    assert np.allclose(wchol @ wchol.T, wcov, rtol=1e-10)
    # nu and sigma (flat entries 0 and d-1) are the scale parameters: far larger posterior variance
    # in the window than any latent volatility; in kernel order they are the last two dimensions
    dg = np.diag(wcov)
    assert dg[0] != dg[1] and np.argmax(np.asarray(st.inv_mass[:d])) in (d - 1, d - 2)
