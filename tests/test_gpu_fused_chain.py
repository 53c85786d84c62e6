"""GPU parity for B2' -- the fused-chain hook of the reference's speculative path (lib/exmc/nuts/tree.ex:613-653):
`exmc_hip_leapfrog_chain_normal_host` through the Python mirror, the raw C ABI and the NIF function, against the
checker's statement of it in the mode the kernels are compared in (deterministic log, 64-lane sums), bit for bit;
and the acceptance band of the reference's own test of the hook (test/nuts/fused_chain_diag_test.exs) with the
leapfrog steps coming off the GPU in dispatches of 32."""
import ctypes as C

import numpy as np
import pytest

import fused_chain_model as FCM
import nif_harness as H
import oracle as O
import py_sampler as PS
from exmc_amd import _lib, fused_chain

pytestmark = pytest.mark.gpu

CFG = O.Cfg(1, 64)


def _same(a, b):
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def _check_against_checker(q, p, im, k, eps, mu, sigma):
    got = fused_chain.leapfrog_chain_normal(q, p, im, k, eps, mu, sigma)
    q2 = np.atleast_2d(q)
    p2 = np.atleast_2d(p)
    for c in range(q2.shape[0]):
        exp = O.leapfrog_chain_normal(q2[c], p2[c], im, k, eps, mu, sigma, CFG)
        for g, e, what in zip(got, exp, ("q_chain", "p_chain", "logp_chain", "grad_chain")):
            gc = g if np.ndim(q) == 1 else g[c]
            assert _same(gc, e), (what, c, q2.shape, k, eps, mu, sigma)


@pytest.mark.parametrize("d", [1, 2, 10, 63, 64, 65, 128, 129, 200, 255, 256])
def test_one_chain_bit_exact(hip, d):
    rng = np.random.default_rng(d)
    q, p, im = rng.normal(size=d), rng.normal(size=d), rng.uniform(0.3, 3.0, size=d)
    for k, eps, mu, sigma in ((32, 0.1, 0.0, 1.0), (7, -0.23, 1.5, 0.7), (1, 0.9, -3.0, 12.0), (100, 0.01, 0.25, 1e-3)):
        _check_against_checker(q, p, im, k, eps, mu, sigma)


@pytest.mark.parametrize("n_chains,d", [(2, 5), (7, 64), (300, 10), (64, 256), (1025, 3), (2000, 64)])
def test_batches_of_independent_chains_bit_exact(hip, n_chains, d):
    """the batched form: every chain of a launch equals the one-chain call of the checker."""
    rng = np.random.default_rng(1000 + n_chains)
    q, p = rng.normal(size=(n_chains, d)), rng.normal(size=(n_chains, d))
    im = rng.uniform(0.5, 2.0, size=d)
    got = fused_chain.leapfrog_chain_normal(q, p, im, 12, -0.17, 0.3, 1.9)
    assert got[0].shape == (n_chains, 12, d) and got[2].shape == (n_chains, 12)
    for c in sorted({0, 1, n_chains // 2, n_chains - 1}):
        exp = O.leapfrog_chain_normal(q[c], p[c], im, 12, -0.17, 0.3, 1.9, CFG)
        for g, e in zip(got, exp):
            assert _same(g[c], e)
    # and the whole batch against its own one-chain launches
    for c in (0, n_chains - 1):
        one = fused_chain.leapfrog_chain_normal(q[c], p[c], im, 12, -0.17, 0.3, 1.9)
        for g, o in zip(got, one):
            assert _same(g[c], o)


def test_same_rows_as_the_batched_leapfrog_kernel_of_a_generated_normal_model(hip):
    """tree.ex:620-621: "Output contract is identical in both branches". The other branch here is multi_step_fn of
    the model compiled from its Builder node (x ~ Normal(0, 1), exmc_amd/codegen.py): same rows to rounding (the
    generated text folds constants its own way), identical where the arithmetic is elementwise."""
    from exmc_amd import codegen, sampler
    ir = codegen.IR()
    ir.rv("x", "normal", {"mu": 0.0, "sigma": 1.0})
    comp = sampler.compile(codegen.compile_ir(ir, name="gen_fused_chain_normal", default_init={"x": 0.0}))
    rng = np.random.default_rng(7)
    q, p = rng.normal(size=1), rng.normal(size=1)
    lp = np.zeros(1); g = np.zeros(1)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    L = comp.L
    comp.check(L.exmc_hip_logp_grad_host(comp.h, dp(q), 1, 0, dp(lp), dp(g)))
    aq = np.zeros((16, 1)); ap = np.zeros((16, 1)); ag = np.zeros((16, 1)); al = np.zeros(16)
    im = np.ones(1)
    comp.check(L.exmc_hip_multi_step_host(comp.h, dp(q), dp(p), dp(g), 0.2, dp(im), 16, 1, 0, dp(aq), dp(ap), dp(al), dp(ag)))
    fq, fp, fl, fg = fused_chain.leapfrog_chain_normal(q, p, im, 16, 0.2, 0.0, 1.0)
    assert np.allclose(fq, aq, rtol=0, atol=1e-14) and np.allclose(fp, ap, rtol=0, atol=1e-14)
    assert np.allclose(fg, ag, rtol=0, atol=1e-14) and np.allclose(fl, al, rtol=1e-14, atol=0)
    comp.close()


def test_hostile_inputs_bit_exact(hip):
    """positions and momenta with NaN, infinities, 1e300 and denormals; a step size far too large; sigma below the
    guard of normal.ex:18, zero, negative and NaN; an inverse mass of zero and of 1e300."""
    d = 70
    rng = np.random.default_rng(11)
    q, p, im = rng.normal(size=d), rng.normal(size=d), rng.uniform(0.5, 2.0, size=d)
    q[[3, 64]] = [np.nan, np.inf]
    q[[5, 66]] = [1e300, -1e300]
    q[7] = 5e-324
    p[[9, 69]] = [-np.inf, 1e308]
    im[[11, 12]] = [0.0, 1e300]
    for eps in (0.1, 1e6, -1e-300):
        for sigma in (1.0, 1e-40, 0.0, -2.0, np.nan, np.inf):
            _check_against_checker(q, p, im, 5, eps, 0.5, sigma)
    _check_against_checker(q, p, im, 3, np.nan, np.nan, 1.0)


def test_prefix_and_zero_steps_on_the_device(hip):
    rng = np.random.default_rng(2)
    q, p, im = rng.normal(size=10), rng.normal(size=10), np.ones(10)
    a = fused_chain.leapfrog_chain_normal(q, p, im, 32, 0.1, 0.0, 1.0)
    b = fused_chain.leapfrog_chain_normal(q, p, im, 16, 0.1, 0.0, 1.0)
    for x, y in zip(a, b):
        assert _same(x[:16], y)
    z = fused_chain.leapfrog_chain_normal(q, p, im, 0, 0.1, 0.0, 1.0)
    assert z[0].shape == (0, 10) and z[2].shape == (0,)


def test_arguments_are_checked(hip):
    """d above the hook's own bound (tree.ex:636), negative k, null pointers, a device that does not exist: refused
    with EXMC_ERR_BADARG, nothing launched."""
    L = hip
    dp = C.POINTER(C.c_double)
    z = np.zeros(300)
    zp = z.ctypes.data_as(dp)
    out = np.zeros(300 * 4)
    op = out.ctypes.data_as(dp)
    f = L.exmc_hip_leapfrog_chain_normal_host
    assert f(0, 1, 257, zp, zp, zp, 1, 0.1, 0.0, 1.0, op, op, op, op) == _lib.ERR_BADARG
    assert b"256" in L.exmc_hip_last_error()
    assert f(0, 1, 0, zp, zp, zp, 1, 0.1, 0.0, 1.0, op, op, op, op) == _lib.ERR_BADARG
    assert f(0, 0, 4, zp, zp, zp, 1, 0.1, 0.0, 1.0, op, op, op, op) == _lib.ERR_BADARG
    assert f(0, 1, 4, zp, zp, zp, -1, 0.1, 0.0, 1.0, op, op, op, op) == _lib.ERR_BADARG
    assert f(0, 1, 4, None, zp, zp, 1, 0.1, 0.0, 1.0, op, op, op, op) == _lib.ERR_BADARG
    assert f(99, 1, 4, zp, zp, zp, 1, 0.1, 0.0, 1.0, op, op, op, op) == _lib.ERR_BADARG
    # outputs are optional one by one
    assert f(0, 1, 4, zp, zp, zp, 2, 0.1, 0.0, 1.0, None, None, None, op) == _lib.OK
    with pytest.raises(ValueError):
        fused_chain.leapfrog_chain_normal(np.zeros(257), np.zeros(257), np.ones(257), 1, 0.1, 0.0, 1.0)
    with pytest.raises(ValueError):
        fused_chain.leapfrog_chain_normal(np.zeros(4), np.zeros(5), np.ones(4), 1, 0.1, 0.0, 1.0)


def test_fused_chain_diag_band_on_the_device(hip):
    """test/nuts/fused_chain_diag_test.exs:123-140 ("fused chain: leapfrog_chain_normal produces var ~ 1.0"): x ~ N(0, 1),
    seed 42, 200 warmup + 1000 draws, the tree's leapfrog steps served from GPU dispatches of 32; variance in
    [0.7, 1.3] -- and the run equals the checker-fed run draw for draw (for d = 1 the 64-lane sum is the plain sum)."""
    dev = FCM.FusedChainModel(1, 0.0, 1.0, lambda *a: fused_chain.leapfrog_chain_normal(*a))
    tr, st = PS.sample(dev, num_warmup=200, num_samples=1000, seed=42)
    xs = tr["draws"][:, 0]
    var = float(np.mean((xs - xs.mean()) ** 2))
    assert 0.7 <= var <= 1.3, var
    assert dev.dispatches < dev.steps
    cpu = FCM.FusedChainModel(1, 0.0, 1.0, lambda *a: O.leapfrog_chain_normal(*a))
    tr2, st2 = PS.sample(cpu, num_warmup=200, num_samples=1000, seed=42)
    assert np.array_equal(tr["draws"], tr2["draws"]) and np.array_equal(tr["tree_depth"], tr2["tree_depth"])
    assert np.array_equal(tr["divergent"], tr2["divergent"]) and st["step_size"] == st2["step_size"]


def test_through_the_nif_function(hip, tmp_path):
    """HipNative.leapfrog_chain_normal/7 as elixir/patches/tree.ex.diff calls it: binaries in, {:ok, {q_chain, p_chain,
    grad_chain, logp_chain}} out (the shape of Nx.Vulkan.leapfrog_chain_normal/7, tree.ex:641-647); badarg above 256
    dimensions and on ragged binaries."""
    hn = H.build(str(tmp_path))[1]["HipNative"]
    rng = np.random.default_rng(5)
    q, p, im = rng.normal(size=6), rng.normal(size=6), rng.uniform(0.5, 2.0, size=6)
    ok, (aq, ap, ag, al) = hn.call("leapfrog_chain_normal", q, p, im, 9, -0.3, 0.25, 1.75)
    assert ok == H.Atom("ok")
    eq, ep, el, eg = O.leapfrog_chain_normal(q, p, im, 9, -0.3, 0.25, 1.75, CFG)
    assert _same(H.f64(aq).reshape(9, 6), eq) and _same(H.f64(ap).reshape(9, 6), ep)
    assert _same(H.f64(ag).reshape(9, 6), eg) and _same(H.f64(al), el)
    ok, (aq, _, _, al) = hn.call("leapfrog_chain_normal", q, p, im, 0, 0.1, 0, 1)       # integers for mu, sigma; k = 0
    assert ok == H.Atom("ok") and H.f64(aq).size == 0 and H.f64(al).size == 0
    with pytest.raises(H.BadArg):
        hn.call("leapfrog_chain_normal", np.zeros(257), np.zeros(257), np.ones(257), 1, 0.1, 0.0, 1.0)
    with pytest.raises(H.BadArg):
        hn.call("leapfrog_chain_normal", q, p[:5], im, 1, 0.1, 0.0, 1.0)
    with pytest.raises(H.BadArg):
        hn.call("leapfrog_chain_normal", q, p, im, -1, 0.1, 0.0, 1.0)
