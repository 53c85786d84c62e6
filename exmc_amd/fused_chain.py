"""Host-side mirror of the fused-chain hook of the reference's speculative path (lib/exmc/nuts/tree.ex:613-653):
`dispatch_multi_step` sends a direction's next K leapfrog steps to `leapfrog_chain_normal(q, p, inv_mass, k,
signed_eps, mu, sigma)` -- one dispatch for K steps -- when the application has set
`:fused_leapfrog_normal_meta = {mu, sigma}` (the model is d independent Normal(mu, sigma) coordinates) and
d <= 256, and to `multi_step_fn` otherwise; both return `{all_q, all_p, all_logp, all_grad}`.

Here the dispatch is one launch of `leapfrog_chain_normal_kernel` (exmc_amd/csrc/exmc_kernels.hpp) through
`exmc_hip_leapfrog_chain_normal_host`; f64 throughout (the reference's hook moves f32 binaries because its
Vulkan device computes in f32, tree.ex:655-669). No CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib

MAX_D = 256    # tree.ex:636 `d <= 256`


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def leapfrog_chain_normal(q, p, inv_mass, k, signed_eps, mu, sigma, device=0):
    """tree.ex:632-653 for one chain (q, p: [d]) or a batch of independent chains ([C, d]).

    Returns (q_chain, p_chain, logp_chain, grad_chain) -- the order of do_dispatch's result tuple,
    `{all_q, all_p, all_logp, all_grad}` -- of shapes [k, d] / [k] (one chain) or [C, k, d] / [C, k]."""
    q = np.ascontiguousarray(np.asarray(q, dtype=np.float64))
    single = q.ndim == 1
    if single:
        q = q[None, :]
    if q.ndim != 2:
        raise ValueError("q must be [d] or [C, d]")
    Cn, d = q.shape
    p = np.ascontiguousarray(np.asarray(p, dtype=np.float64))
    if single and p.ndim == 1:
        p = p[None, :]
    if p.shape != q.shape:
        raise ValueError("p must have q's shape")                      # NIF badarg
    im = np.ascontiguousarray(np.asarray(inv_mass, dtype=np.float64))
    if im.shape != (d,):
        raise ValueError("inv_mass must be [d]")
    k = int(k)
    if not (1 <= d <= MAX_D) or k < 0:
        # the reference's function head does not match above 256 dimensions and falls through to multi_step_fn
        raise ValueError("leapfrog_chain_normal: 1 <= d <= %d and k >= 0" % MAX_D)
    aq = np.empty((Cn, k, d)); ap = np.empty((Cn, k, d)); ag = np.empty((Cn, k, d)); al = np.empty((Cn, k))
    L = _lib.load()
    _lib.check(L.exmc_hip_leapfrog_chain_normal_host(int(device), Cn, d, _dp(q), _dp(p), _dp(im), k, float(signed_eps),
                                                     float(mu), float(sigma), _dp(aq), _dp(ap), _dp(ag), _dp(al)))
    if single:
        return aq[0], ap[0], al[0], ag[0]
    return aq, ap, al, ag
