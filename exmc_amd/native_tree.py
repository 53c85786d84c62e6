"""Host-side mirror of Exmc.NUTS.NativeTree (lib/exmc/nuts/native_tree.ex) for the entry point the
GPU library provides: build_full_tree_bin/17, batched over chains.

The reference NIF takes raw native-endian f64 binaries (Nx.to_binary) for one chain; here every
argument gains a leading chain axis and the result map's fields become arrays over chains.
"""
import ctypes as C

import numpy as np

from . import _lib


def _f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None and a.shape != shape:
        raise ValueError("expected shape %s, got %s" % (shape, a.shape))   # NIF badarg
    return a


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def build_full_tree_bin(q0, p0, grad0, logp0, fwd_q, fwd_p, fwd_logp, fwd_grad, bwd_q, bwd_p,
                        bwd_logp, bwd_grad, inv_mass, joint_logp_0, max_depth, d, rng_seed,
                        device=0):
    """NativeTree.build_full_tree_bin/17 (native_tree.ex:55-75; lib.rs:219-302) for C chains.

    q0, p0, grad0: [C, d]; logp0, joint_logp_0, rng_seed: [C]; fwd_* / bwd_*: [C, n, d] and
    [C, n]; inv_mass: [d]. Returns the NIF's result map with a chain axis:
    {q_bin [C,d], logp [C], grad_bin [C,d], n_steps, divergent, accept_sum, depth}."""
    q0 = _f64(q0)
    if q0.ndim != 2 or q0.shape[1] != d:
        raise ValueError("q0 must be [C, d]")
    Cn = q0.shape[0]
    p0, grad0 = _f64(p0, (Cn, d)), _f64(grad0, (Cn, d))
    logp0, jlp0 = _f64(logp0, (Cn,)), _f64(joint_logp_0, (Cn,))
    fwd_q = _f64(fwd_q)
    bwd_q = _f64(bwd_q)
    n_fwd, n_bwd = fwd_q.shape[1], bwd_q.shape[1]
    fwd_q, fwd_p, fwd_grad = (_f64(x, (Cn, n_fwd, d)) for x in (fwd_q, fwd_p, fwd_grad))
    bwd_q, bwd_p, bwd_grad = (_f64(x, (Cn, n_bwd, d)) for x in (bwd_q, bwd_p, bwd_grad))
    fwd_logp, bwd_logp = _f64(fwd_logp, (Cn, n_fwd)), _f64(bwd_logp, (Cn, n_bwd))
    inv_mass = _f64(inv_mass, (d,))
    seeds = np.ascontiguousarray(np.asarray(rng_seed, dtype=np.uint64))
    if seeds.shape != (Cn,):
        raise ValueError("rng_seed must be [C]")
    q = np.zeros((Cn, d)); g = np.zeros((Cn, d)); lp = np.zeros(Cn); acc = np.zeros(Cn)
    ns = np.zeros(Cn, np.int32); dv = np.zeros(Cn, np.int32); dep = np.zeros(Cn, np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
    _lib.check(_lib.load().exmc_hip_build_full_tree_host(
        int(device), Cn, int(d), _dp(q0), _dp(p0), _dp(grad0), _dp(logp0), _dp(fwd_q), _dp(fwd_p),
        _dp(fwd_logp), _dp(fwd_grad), n_fwd, _dp(bwd_q), _dp(bwd_p), _dp(bwd_logp), _dp(bwd_grad),
        n_bwd, _dp(inv_mass), _dp(jlp0), int(max_depth),
        seeds.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(q), _dp(lp), _dp(g), ip(ns), ip(dv),
        _dp(acc), ip(dep)))
    return dict(q_bin=q, logp=lp, grad_bin=g, n_steps=ns, divergent=dv.astype(bool),
                accept_sum=acc, depth=dep)
