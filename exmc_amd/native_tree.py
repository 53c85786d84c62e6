"""Host-side mirror of Exmc.NUTS.NativeTree (lib/exmc/nuts/native_tree.ex) for the entry point the
GPU library provides, each batched over chains: build_full_tree_bin/17, build_subtree_bin/10 and
the trajectory resource (init_trajectory_bin, get_endpoint_bin, build_and_merge_bin, is_terminated,
get_result_bin).

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


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _i32(a, n):
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.int32), (n,)))
    return a


def _subtree_args(all_q, all_p, all_logp, all_grad, inv_mass, joint_logp_0, depth, d, go_right,
                  rng_seed):
    all_q = _f64(all_q)
    if all_q.ndim != 3 or all_q.shape[2] != d:
        raise ValueError("all_q must be [C, n, d]")
    Cn, n = all_q.shape[0], all_q.shape[1]
    all_p, all_grad = _f64(all_p, (Cn, n, d)), _f64(all_grad, (Cn, n, d))
    all_logp = _f64(all_logp, (Cn, n))
    inv_mass = _f64(inv_mass, (d,))
    jlp0 = _f64(np.broadcast_to(np.asarray(joint_logp_0, dtype=np.float64), (Cn,)))
    depth, go_right = _i32(depth, Cn), _i32(go_right, Cn)
    seeds = np.ascontiguousarray(np.broadcast_to(np.asarray(rng_seed, dtype=np.uint64), (Cn,)))
    return Cn, n, all_q, all_p, all_logp, all_grad, inv_mass, jlp0, depth, go_right, seeds


class Trajectories:
    """The NIF's trajectory resource (native_tree.ex:19-40, 79-110; lib.rs:37-112, 305-343) for a
    batch of chains, resident on the GPU between calls. Method names and argument order are the
    NIF's, every argument with a leading chain axis."""

    def __init__(self, q, p, grad, logp, device=0):
        """init_trajectory_bin/4."""
        q = _f64(q)
        if q.ndim != 2:
            raise ValueError("q must be [C, d]")
        self.C, self.d = q.shape
        p, grad, logp = _f64(p, q.shape), _f64(grad, q.shape), _f64(logp, (self.C,))
        h = C.c_void_p()
        _lib.check(_lib.load().exmc_hip_traj_create(int(device), self.C, self.d, _dp(q), _dp(p),
                                                    _dp(grad), _dp(logp), C.byref(h)))
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            _lib.load().exmc_hip_traj_destroy(h)

    def get_endpoint_bin(self, go_right):
        """get_endpoint_bin/2 -> (q, p, grad), each [C, d]."""
        gr = _i32(go_right, self.C)
        q, p, g = (np.zeros((self.C, self.d)) for _ in range(3))
        _lib.check(_lib.load().exmc_hip_traj_get_endpoint_host(self.h, _ip(gr), _dp(q), _dp(p), _dp(g)))
        return q, p, g

    def build_and_merge_bin(self, all_q, all_p, all_logp, all_grad, inv_mass, joint_logp_0, depth,
                            d, go_right, rng_seed):
        """build_and_merge_bin/11: depth, go_right, rng_seed, joint_logp_0 scalars or [C]; a chain
        with depth < 0 is left untouched."""
        Cn, n, aq, ap, alp, ag, im, jlp0, depth, gr, seeds = _subtree_args(
            all_q, all_p, all_logp, all_grad, inv_mass, joint_logp_0, depth, d, go_right, rng_seed)
        if Cn != self.C or d != self.d:
            raise ValueError("batch does not match the trajectories")
        _lib.check(_lib.load().exmc_hip_traj_build_and_merge_host(
            self.h, _dp(aq), _dp(ap), _dp(alp), _dp(ag), n, _dp(im), _dp(jlp0), _ip(depth), _ip(gr),
            seeds.ctypes.data_as(C.POINTER(C.c_uint64))))
        return "ok"

    def is_terminated(self):
        """is_terminated/1 -> bool [C]."""
        out = np.zeros(self.C, np.int32)
        _lib.check(_lib.load().exmc_hip_traj_is_terminated_host(self.h, _ip(out)))
        return out.astype(bool)

    def get_result_bin(self):
        """get_result_bin/1 -> the NIF's result map with a chain axis."""
        q, g = np.zeros((self.C, self.d)), np.zeros((self.C, self.d))
        lp, acc = np.zeros(self.C), np.zeros(self.C)
        ns, dv, dep = (np.zeros(self.C, np.int32) for _ in range(3))
        _lib.check(_lib.load().exmc_hip_traj_get_result_host(self.h, _dp(q), _dp(lp), _dp(g), _ip(ns),
                                                             _ip(dv), _dp(acc), _ip(dep)))
        return dict(q_bin=q, logp=lp, grad_bin=g, n_steps=ns, divergent=dv.astype(bool),
                    accept_sum=acc, depth=dep)


def build_subtree_bin(all_q, all_p, all_logp, all_grad, inv_mass, joint_logp_0, depth, d,
                      going_right, rng_seed, device=0):
    """NativeTree.build_subtree_bin/10 (native_tree.ex:42-55; lib.rs:114-212) for C chains: the
    subtree record as the NIF's map, every field with a chain axis."""
    Cn, n, aq, ap, alp, ag, im, jlp0, depth, gr, seeds = _subtree_args(
        all_q, all_p, all_logp, all_grad, inv_mass, joint_logp_0, depth, d, going_right, rng_seed)
    v = [np.zeros((Cn, d)) for _ in range(9)]   # qL pL gL qR pR gR qP gP rho
    lpp, lsw, acc = np.zeros(Cn), np.zeros(Cn), np.zeros(Cn)
    ns, dv, tn, dep = (np.zeros(Cn, np.int32) for _ in range(4))
    _lib.check(_lib.load().exmc_hip_build_subtree_host(
        int(device), Cn, int(d), _dp(aq), _dp(ap), _dp(alp), _dp(ag), n, _dp(im), _dp(jlp0),
        _ip(depth), _ip(gr), seeds.ctypes.data_as(C.POINTER(C.c_uint64)),
        _dp(v[0]), _dp(v[1]), _dp(v[2]), _dp(v[3]), _dp(v[4]), _dp(v[5]), _dp(v[6]), _dp(lpp),
        _dp(v[7]), _dp(lsw), _ip(ns), _ip(dv), _dp(acc), _ip(tn), _ip(dep), _dp(v[8])))
    return dict(q_left_bin=v[0], p_left_bin=v[1], grad_left_bin=v[2], q_right_bin=v[3],
                p_right_bin=v[4], grad_right_bin=v[5], q_prop_bin=v[6], logp_prop=lpp,
                grad_prop_bin=v[7], log_sum_weight=lsw, n_steps=ns, divergent=dv.astype(bool),
                accept_sum=acc, turning=tn.astype(bool), depth=dep, rho_bin=v[8])
