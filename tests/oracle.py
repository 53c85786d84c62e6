"""ctypes binding of the CPU parity checker (oracle/). Test infrastructure only.

The product package (exmc_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# EXMC_ORACLE_LIB: another build of the same checker (tools/sanitize_cpu.sh: -fsanitize=address,undefined)
LIB_PATH = os.environ.get("EXMC_ORACLE_LIB") or os.path.join(ROOT, "oracle", "build", "libexmc_oracle.so")

MAX_D = 256
STD_NORMAL, SIMPLE, EIGHT_SCHOOLS, SV, LOGISTIC, RADON = range(6)
EXO_MODEL_CUSTOM = 6


class Cfg(C.Structure):
    _fields_ = [("math_mode", C.c_int), ("lanes", C.c_int)]


class Rng(C.Structure):
    _fields_ = [("a", C.c_uint64), ("b", C.c_uint64)]


class TreeResult(C.Structure):
    _fields_ = [("logp", C.c_double), ("n_steps", C.c_int), ("divergent", C.c_int),
                ("accept_sum", C.c_double), ("depth", C.c_int)]


class DA(C.Structure):
    _fields_ = [("log_epsilon", C.c_double), ("log_epsilon_bar", C.c_double),
                ("h_bar", C.c_double), ("mu", C.c_double), ("m", C.c_int),
                ("gamma", C.c_double), ("t0", C.c_double), ("kappa", C.c_double),
                ("target_accept", C.c_double), ("math_mode", C.c_int)]


class Welford(C.Structure):
    _fields_ = [("n", C.c_int), ("d", C.c_int), ("mean", C.c_double * MAX_D),
                ("m2", C.c_double * MAX_D)]


class Opts(C.Structure):
    _fields_ = [("num_warmup", C.c_int), ("num_samples", C.c_int), ("max_tree_depth", C.c_int),
                ("target_accept", C.c_double), ("seed", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("step_size", C.c_double), ("inv_mass", C.c_double * MAX_D),
                ("divergences", C.c_int), ("total_leapfrogs", C.c_long)]


class Trace(C.Structure):
    _fields_ = [("draws", C.c_void_p), ("logp", C.c_void_p), ("tree_depth", C.c_void_p),
                ("n_steps", C.c_void_p), ("divergent", C.c_void_p), ("accept_prob", C.c_void_p),
                ("energy", C.c_void_p)]


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    L.exo_rng_seed.argtypes = [C.POINTER(Rng), C.c_uint64]
    L.exo_rng_next.argtypes = [C.POINTER(Rng)]
    L.exo_rng_next.restype = C.c_uint64
    L.exo_rng_uniform.argtypes = [C.POINTER(Rng)]
    L.exo_rng_uniform.restype = C.c_double
    L.exo_rng_normal.argtypes = [C.POINTER(Rng), C.c_int]
    L.exo_rng_normal.restype = C.c_double
    L.exo_splitmix64.argtypes = [C.POINTER(C.c_uint64)]
    L.exo_splitmix64.restype = C.c_uint64
    L.exo_xoshiro_seed_from_u64.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
    L.exo_xoshiro_next.argtypes = [C.POINTER(C.c_uint64)]
    L.exo_xoshiro_next.restype = C.c_uint64
    L.exo_xoshiro_f64.argtypes = [C.POINTER(C.c_uint64)]
    L.exo_xoshiro_f64.restype = C.c_double
    for name in ("exo_exp", "exo_log", "exo_log1p", "exo_lgamma_lanczos"):
        f = getattr(L, name)
        f.argtypes = [C.c_double, C.c_int]
        f.restype = C.c_double
    for name in ("exo_det_exp", "exo_det_log", "exo_det_log1p", "exo_det_erf"):
        f = getattr(L, name)
        f.argtypes = [C.c_double]
        f.restype = C.c_double
    L.exo_log_sum_exp.argtypes = [C.c_double, C.c_double, C.c_int]
    L.exo_log_sum_exp.restype = C.c_double
    L.exo_model_create.argtypes = [C.c_int, C.c_int, dp, C.c_int]
    L.exo_model_create.restype = C.c_void_p
    L.exo_model_free.argtypes = [C.c_void_p]
    L.exo_model_set_custom.argtypes = [C.c_void_p, C.c_void_p]
    L.exo_model_set_custom.restype = None
    L.exo_model_dim.argtypes = [C.c_void_p]
    L.exo_model_set_flat_order.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.exo_model_set_flat_order.restype = C.c_int
    L.exo_model_get_flat_order.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.exo_model_get_flat_order.restype = None
    L.exo_logp_grad.argtypes = [C.c_void_p, dp, dp, Cfg]
    L.exo_logp_grad.restype = C.c_double
    L.exo_constrain.argtypes = [C.c_void_p, dp, dp]
    L.exo_dist_normal.argtypes = [C.c_double] * 3 + [C.c_int]
    L.exo_dist_normal.restype = C.c_double
    L.exo_dist_half_cauchy.argtypes = [C.c_double] * 2 + [C.c_int]
    L.exo_dist_half_cauchy.restype = C.c_double
    L.exo_dist_exponential.argtypes = [C.c_double] * 2 + [C.c_int]
    L.exo_dist_exponential.restype = C.c_double
    L.exo_dist_half_normal.argtypes = [C.c_double] * 2 + [C.c_int]
    L.exo_dist_half_normal.restype = C.c_double
    L.exo_dist_bernoulli.argtypes = [C.c_double] * 2 + [C.c_int]
    L.exo_dist_bernoulli.restype = C.c_double
    L.exo_dist_student_t.argtypes = [C.c_double] * 4 + [C.c_int]
    L.exo_dist_student_t.restype = C.c_double
    L.exo_kinetic_energy.argtypes = [dp, dp, C.c_int, Cfg]
    L.exo_kinetic_energy.restype = C.c_double
    L.exo_leapfrog.argtypes = [C.c_void_p, dp, dp, dp, C.c_double, dp, dp, Cfg]
    L.exo_leapfrog.restype = C.c_double
    L.exo_multi_step.argtypes = [C.c_void_p, dp, dp, dp, C.c_double, dp, C.c_int, dp, dp, dp, dp,
                                 Cfg]
    L.exo_leapfrog_chain_normal.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                            dp, dp, dp, dp, Cfg]
    L.exo_leapfrog_chain_normal.restype = C.c_int
    L.exo_tree_build.argtypes = [C.c_void_p, dp, dp, C.c_double, dp, C.c_double, dp, C.c_int, Rng,
                                 C.c_double, dp, dp, C.POINTER(TreeResult), Cfg]
    L.exo_check_uturn.argtypes = [dp, dp, dp, dp, C.c_int, Cfg]
    L.exo_check_uturn.restype = C.c_int
    L.exo_da_init.argtypes = [C.POINTER(DA), C.c_double, C.c_double]
    L.exo_da_init_mode.argtypes = [C.POINTER(DA), C.c_double, C.c_double, C.c_int]
    L.exo_da_update.argtypes = [C.POINTER(DA), C.c_double]
    L.exo_da_finalize.argtypes = [C.POINTER(DA)]
    L.exo_da_finalize.restype = C.c_double
    L.exo_welford_init.argtypes = [C.POINTER(Welford), C.c_int]
    L.exo_welford_update.argtypes = [C.POINTER(Welford), dp]
    L.exo_welford_finalize.argtypes = [C.POINTER(Welford), dp]
    ip = C.POINTER(C.c_int)
    L.exo_build_windows.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, C.c_int]
    L.exo_build_windows.restype = C.c_int
    L.exo_sample.argtypes = [C.c_void_p, dp, Opts, Trace, C.POINTER(Stats), Cfg]
    L.exo_sample_chains.argtypes = [C.c_void_p, dp, C.c_int, C.c_int, C.c_int, Opts, Trace,
                                    C.POINTER(Stats), C.c_int, Cfg]
    L.exo_warmup.argtypes = [C.c_void_p, dp, Opts, C.POINTER(Stats), Cfg]
    L.exo_sample_tuned.argtypes = [C.c_void_p, dp, C.c_double, dp, Opts, Trace, C.POINTER(Stats),
                                   Cfg]
    L.exo_sample_warm.argtypes = [C.c_void_p, dp, C.c_double, dp, Opts, Trace, C.POINTER(Stats), Cfg]
    L.exo_warmup_dense.argtypes = [C.c_void_p, dp, Opts, C.POINTER(Stats), dp, dp, Cfg]
    L.exo_sample_tuned_dense.argtypes = [C.c_void_p, dp, C.c_double, dp, dp, Opts, Trace, C.POINTER(Stats), Cfg]
    L.exo_cholesky_lower.argtypes = [dp, C.c_int, dp]
    L.exo_dense_mass_times.argtypes = [dp, dp, C.c_int, dp]
    L.exo_dense_mass_times.restype = None
    L.exo_dense_check_uturn.argtypes = [dp, dp, dp, dp, C.c_int, Cfg]
    L.exo_dense_momentum.argtypes = [C.c_void_p, dp, C.POINTER(Rng), dp, C.c_int]
    L.exo_dense_momentum.restype = None
    L.exo_welford_dense_finalize.argtypes = [dp, C.c_int, C.c_int, dp, dp]
    L.exo_welford_dense_finalize.restype = None
    L.exo_ess.argtypes = [dp, C.c_int]
    L.exo_ess.restype = C.c_double
    L.exo_ess_bulk.argtypes = [dp, C.c_int]
    L.exo_ess_bulk.restype = C.c_double
    L.exo_ess_bulk_mode.argtypes = [dp, C.c_int, C.c_int]
    L.exo_ess_bulk_mode.restype = C.c_double
    L.exo_rhat.argtypes = [dp, C.c_int, C.c_int]
    L.exo_rhat.restype = C.c_double
    L.exo_nt_init_trajectory.argtypes = [dp, dp, dp, C.c_double, C.c_int]
    L.exo_nt_init_trajectory.restype = C.c_void_p
    L.exo_nt_free.argtypes = [C.c_void_p]
    L.exo_nt_is_terminated.argtypes = [C.c_void_p]
    L.exo_nt_is_terminated.restype = C.c_int
    L.exo_nt_get_endpoint.argtypes = [C.c_void_p, C.c_int, dp, dp, dp]
    L.exo_nt_build_and_merge.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, C.c_double, C.c_int,
                                         C.c_int, C.c_int, C.c_uint64]
    L.exo_nt_get_result.argtypes = [C.c_void_p, dp, dp, C.POINTER(TreeResult)]
    L.exo_nt_build_subtree.argtypes = [dp, dp, dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int,
                                       C.c_uint64, dp, dp, C.POINTER(C.c_int)]
    L.exo_nt_build_full_tree.argtypes = [dp, dp, dp, C.c_double, dp, dp, dp, dp, C.c_int, dp, dp,
                                         dp, dp, C.c_int, dp, C.c_double, C.c_int, C.c_int,
                                         C.c_uint64, dp, dp, C.POINTER(TreeResult)]
    _lib = L
    return L


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


class Model:
    """Owns an exo_model*."""

    def __init__(self, kind, d=0, data=()):
        data = arr(data)
        self.kind = kind
        self.h = lib().exo_model_create(kind, d, dptr(data), int(data.size))
        if not self.h:
            raise ValueError("exo_model_create failed")
        self.d = lib().exo_model_dim(self.h)

    def __del__(self):
        try:
            if self.h:
                lib().exo_model_free(self.h)
                self.h = None
        except Exception:
            pass

    def set_flat_order(self, order):
        """order[r] = kernel dimension of the r-th entry of the reference's flat vector."""
        a = np.ascontiguousarray(order, dtype=np.int32)
        assert a.size == self.d
        if lib().exo_model_set_flat_order(self.h, a.ctypes.data_as(C.POINTER(C.c_int))) != 0:
            raise ValueError("flat order is not a permutation")

    def flat_order(self):
        a = np.zeros(self.d, dtype=np.int32)
        lib().exo_model_get_flat_order(self.h, a.ctypes.data_as(C.POINTER(C.c_int)))
        return [int(v) for v in a]

    def logp_grad(self, q, cfg=None):
        cfg = cfg or Cfg(0, 1)
        q = arr(q)
        g = np.zeros(self.d)
        lp = lib().exo_logp_grad(self.h, dptr(q), dptr(g), cfg)
        return lp, g

    def constrain(self, q):
        q = arr(q)
        x = np.zeros(self.d)
        lib().exo_constrain(self.h, dptr(q), dptr(x))
        return x

    def leapfrog(self, q, p, g, eps, inv_mass, cfg=None):
        cfg = cfg or Cfg(0, 1)
        q, p, g, im = arr(q).copy(), arr(p).copy(), arr(g).copy(), arr(inv_mass)
        jlp = C.c_double()
        lp = lib().exo_leapfrog(self.h, dptr(q), dptr(p), dptr(g), eps, dptr(im),
                                C.cast(C.byref(jlp), C.POINTER(C.c_double)), cfg)
        return q, p, lp, g, jlp.value

    def multi_step(self, q, p, g, eps, inv_mass, n, cfg=None):
        cfg = cfg or Cfg(0, 1)
        q, p, g, im = arr(q), arr(p), arr(g), arr(inv_mass)
        aq = np.zeros((n, self.d)); ap = np.zeros((n, self.d)); ag = np.zeros((n, self.d))
        alp = np.zeros(n)
        lib().exo_multi_step(self.h, dptr(q), dptr(p), dptr(g), eps, dptr(im), n, dptr(aq),
                             dptr(ap), dptr(alp), dptr(ag), cfg)
        return aq, ap, alp, ag

    def tree_build(self, q, p, logp, g, eps, inv_mass, max_depth, rng, jlp0, cfg=None):
        cfg = cfg or Cfg(0, 1)
        q, p, g, im = arr(q), arr(p), arr(g), arr(inv_mass)
        qo = np.zeros(self.d); go = np.zeros(self.d)
        res = TreeResult()
        lib().exo_tree_build(self.h, dptr(q), dptr(p), logp, dptr(g), eps, dptr(im), max_depth,
                             rng, jlp0, dptr(qo), dptr(go), C.byref(res), cfg)
        return qo, go, res


def alloc_trace(n, d):
    t = dict(draws=np.zeros((n, d)), logp=np.zeros(n), tree_depth=np.zeros(n, np.int32),
             n_steps=np.zeros(n, np.int32), divergent=np.zeros(n, np.int32),
             accept_prob=np.zeros(n), energy=np.zeros(n))
    tr = Trace(*[t[k].ctypes.data for k in
                 ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy")])
    return t, tr


def sample(model, init_q=None, num_warmup=1000, num_samples=1000, max_tree_depth=10,
           target_accept=0.8, seed=0, cfg=None):
    cfg = cfg or Cfg(0, 1)
    t, tr = alloc_trace(num_samples, model.d)
    st = Stats()
    iq = None if init_q is None else dptr(arr(init_q))
    lib().exo_sample(model.h, iq, Opts(num_warmup, num_samples, max_tree_depth, target_accept, seed),
                     tr, C.byref(st), cfg)
    return t, st


def warmup(model, init_q=None, num_warmup=1000, max_tree_depth=10, target_accept=0.8, seed=0,
           cfg=None):
    cfg = cfg or Cfg(0, 1)
    st = Stats()
    iq = None if init_q is None else dptr(arr(init_q))
    lib().exo_warmup(model.h, iq, Opts(num_warmup, 0, max_tree_depth, target_accept, seed),
                     C.byref(st), cfg)
    return st


def sample_tuned(model, epsilon, inv_mass, init_q=None, num_samples=1000, max_tree_depth=10, seed=0,
                 cfg=None):
    cfg = cfg or Cfg(0, 1)
    t, tr = alloc_trace(num_samples, model.d)
    st = Stats()
    iq = None if init_q is None else dptr(arr(init_q))
    im = arr(inv_mass)
    lib().exo_sample_tuned(model.h, iq, epsilon, dptr(im),
                           Opts(0, num_samples, max_tree_depth, 0.8, seed), tr, C.byref(st), cfg)
    return t, st


def sample_warm(model, prev_epsilon, prev_inv_mass, init_q=None, num_warmup=1000, num_samples=1000,
                max_tree_depth=10, target_accept=0.8, seed=0, cfg=None):
    """opts[:warm_start] of Sampler.sample (sampler.ex:167-197)."""
    cfg = cfg or Cfg(0, 1)
    t, tr = alloc_trace(num_samples, model.d)
    st = Stats()
    iq = None if init_q is None else dptr(arr(init_q))
    im = arr(prev_inv_mass)
    lib().exo_sample_warm(model.h, iq, prev_epsilon, dptr(im),
                          Opts(num_warmup, num_samples, max_tree_depth, target_accept, seed), tr,
                          C.byref(st), cfg)
    return t, st


def warmup_dense(model, init_q=None, num_warmup=1000, max_tree_depth=10, target_accept=0.8, seed=0,
                 cfg=None):
    """opts[:dense_mass] warmup: returns (stats, cov [d][d], chol_cov [d][d])."""
    cfg = cfg or Cfg(0, 1)
    st = Stats()
    d = model.d
    cov, chol = np.zeros((d, d)), np.zeros((d, d))
    iq = None if init_q is None else dptr(arr(init_q))
    rc = lib().exo_warmup_dense(model.h, iq, Opts(num_warmup, 0, max_tree_depth, target_accept, seed),
                                C.byref(st), dptr(cov), dptr(chol), cfg)
    if rc != 0:
        raise ArithmeticError("the window covariance is not positive definite")
    return st, cov, chol


def sample_tuned_dense(model, epsilon, cov, chol, init_q=None, num_samples=1000, max_tree_depth=10,
                       seed=0, cfg=None):
    cfg = cfg or Cfg(0, 1)
    t, tr = alloc_trace(num_samples, model.d)
    st = Stats()
    iq = None if init_q is None else dptr(arr(init_q))
    cov, chol = arr(cov), arr(chol)
    lib().exo_sample_tuned_dense(model.h, iq, epsilon, dptr(cov), dptr(chol),
                                 Opts(0, num_samples, max_tree_depth, 0.8, seed), tr, C.byref(st), cfg)
    return t, st


def sample_chains(model, n_chains, init_q=None, num_warmup=1000, num_samples=1000,
                  max_tree_depth=10, target_accept=0.8, seed=0, chain_lo=0, chain_hi=None,
                  n_threads=1, cfg=None):
    cfg = cfg or Cfg(0, 1)
    chain_hi = n_chains if chain_hi is None else chain_hi
    nc = chain_hi - chain_lo
    t, tr = alloc_trace(nc * num_samples, model.d)
    st = Stats()
    iq = None if init_q is None else dptr(arr(init_q))
    lib().exo_sample_chains(model.h, iq, n_chains, chain_lo, chain_hi,
                            Opts(num_warmup, num_samples, max_tree_depth, target_accept, seed), tr,
                            C.byref(st), n_threads, cfg)
    for k in t:
        t[k] = t[k].reshape((nc, num_samples) + t[k].shape[1:])
    return t, st


def model_for(spec):
    """The checker's model for an exmc_amd ModelSpec, drawing in the spec's flat order
    (ModelSpec.flat_order = PointMap.build's sorted ids, point_map.ex:30-60)."""
    m = Model(spec.kind, spec.d, spec.data)
    m.set_flat_order(spec.flat_order())
    return m


EIGHT_SCHOOLS_Y = [28.0, 8.0, -3.0, 7.0, -1.0, 1.0, 18.0, 12.0]
EIGHT_SCHOOLS_SIGMA = [15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0]
SIMPLE_Y = [float(np.float32(v)) for v in (2.1, 1.8, 2.5, 2.0, 1.9, 2.3, 2.2, 1.7, 2.4, 2.6)]


def eight_schools():
    return Model(EIGHT_SCHOOLS, 10, EIGHT_SCHOOLS_Y + EIGHT_SCHOOLS_SIGMA)


def simple():
    return Model(SIMPLE, 2, SIMPLE_Y)


def leapfrog_chain_normal(q, p, inv_mass, k, signed_eps, mu, sigma, cfg=None):
    """B2' (tree.ex:613-653): the checker's statement of the fused-chain hook -> (q_chain [k][d], p_chain,
    logp_chain [k], grad_chain) -- the order of do_dispatch's result tuple {all_q, all_p, all_logp, all_grad}."""
    cfg = cfg or Cfg(0, 1)
    q, p, im = arr(q), arr(p), arr(inv_mass)
    d = q.shape[0]
    aq = np.zeros((k, d)); ap = np.zeros((k, d)); ag = np.zeros((k, d)); alp = np.zeros(k)
    rc = lib().exo_leapfrog_chain_normal(dptr(q), dptr(p), dptr(im), d, k, signed_eps, mu, sigma, dptr(aq), dptr(ap),
                                         dptr(ag), dptr(alp), cfg)
    if rc:
        raise ValueError("exo_leapfrog_chain_normal: bad sizes")
    return aq, ap, alp, ag


def std_normal(d):
    return Model(STD_NORMAL, d, ())
