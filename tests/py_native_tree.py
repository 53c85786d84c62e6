"""A third statement of the Rust crate's tree (native/exmc_tree/src/tree.rs build_subtree / build_leaf :16-95,
merge_subtrees :103-189, merge_into_trajectory :194-265, build_full_tree :276-326; uturn.rs:8-24; math.rs:3-10;
types.rs Trajectory::new / is_terminated) in plain Python, as the crate writes it (a recursion over pre-computed
states). TEST INFRASTRUCTURE. Shared with the C checker: only the Xoshiro256** stream (exo_xoshiro_*, pinned by
published vectors). The crate's own differences from tree.ex are all here: a divergent leaf keeps the NEW state, the
kinetic energy is sum(0.5 * p * m * p), log_sum_exp returns -inf, the sub-trajectory checks are skipped when the merge
already diverged / turned, the full tree stops early when a chain's budget is exhausted."""
import ctypes as C
import math

import numpy as np

import oracle as O


class Rng:
    def __init__(self, seed):
        self.s = (C.c_uint64 * 4)()
        O.lib().exo_xoshiro_seed_from_u64(self.s, seed)

    def f64(self):
        return O.lib().exo_xoshiro_f64(self.s)


def uturn(rho, pl, pr, im):                                     # uturn.rs:8-24
    dr = dl = 0.0
    for r, a, b, m in zip(rho, pl, pr, im):
        v = r * m
        dr += v * b
        dl += v * a
    return dr < 0.0 or dl < 0.0


def lse(a, b):                                                  # math.rs:3-10
    m = max(a, b)
    return -math.inf if m == -math.inf else m + math.log(math.exp(a - m) + math.exp(b - m))


def _exp(x):
    try:
        return math.exp(x)
    except OverflowError:
        return math.inf


def leaf(st, im, jlp0, counter):                                # tree.rs:44-95
    i = counter[0]
    counter[0] += 1
    q, p, lp, g = [float(v) for v in st["q"][i]], [float(v) for v in st["p"][i]], float(st["logp"][i]), [float(v) for v in st["g"][i]]
    ke = 0.0
    for pi, mi in zip(p, im):
        ke += 0.5 * pi * float(mi) * pi             # (Python floats: an overflow is +inf, as in Rust)
    j = lp - ke
    if math.isfinite(j):
        d = j - jlp0
        div, lw, acc = d < -1000.0, d, min(_exp(min(d, 0.0)), 1.0)
    else:
        div, lw, acc = True, -1001.0, 0.0
    return dict(ql=q, pl=p, gl=g, qr=q, pr=p, gr=g, qp=q, lpp=lp, gp=g, rho=list(p), lsw=lw, n=1, div=div, acc=acc, turn=False,
                depth=0)


def _sub(left, right, im):
    r2 = [a + b for a, b in zip(left["rho"], right["pl"])]
    if uturn(r2, left["pl"], right["pl"], im):
        return True
    r3 = [a + b for a, b in zip(left["pr"], right["rho"])]
    return uturn(r3, left["pr"], right["pr"], im)


def merge_subtrees(a, b, right, im, rng):                       # tree.rs:103-189
    lsw = lse(a["lsw"], b["lsw"])
    div = a["div"] or b["div"]
    u = rng.f64()
    src = b if u < _exp(b["lsw"] - lsw) else a
    rho = [x + y for x, y in zip(a["rho"], b["rho"])]
    sub = False
    if not div and not b["turn"] and a["depth"] > 0:
        sub = _sub(*((a, b) if right else (b, a)), im)
    left, rgt = (a, b) if right else (b, a)
    turn = div or b["turn"] or sub or uturn(rho, left["pl"], rgt["pr"], im)
    return dict(ql=left["ql"], pl=left["pl"], gl=left["gl"], qr=rgt["qr"], pr=rgt["pr"], gr=rgt["gr"], qp=src["qp"], lpp=src["lpp"],
                gp=src["gp"], rho=rho, lsw=lsw, n=a["n"] + b["n"], div=div, acc=a["acc"] + b["acc"], turn=turn,
                depth=max(a["depth"], b["depth"]) + 1)


def subtree(st, im, jlp0, depth, right, counter, rng):          # tree.rs:16-41
    if depth == 0:
        return leaf(st, im, jlp0, counter)
    first = subtree(st, im, jlp0, depth - 1, right, counter, rng)
    if first["div"] or first["turn"]:
        return first
    second = subtree(st, im, jlp0, depth - 1, right, counter, rng)
    return merge_subtrees(first, second, right, im, rng)


def merge_into_trajectory(t, s, right, im, rng):                # tree.rs:194-265
    lsw = lse(t["lsw"], s["lsw"])
    div = t["div"] or s["div"]
    sub = False
    if not div and not s["turn"]:
        sub = _sub(*((t, s) if right else (s, t)), im)
    u = rng.f64()
    if (math.log(u) if u > 0.0 else -math.inf) < (s["lsw"] - t["lsw"]):
        t["qp"], t["lpp"], t["gp"] = s["qp"], s["lpp"], s["gp"]
    t["rho"] = [x + y for x, y in zip(t["rho"], s["rho"])]
    if right:
        t["qr"], t["pr"], t["gr"] = s["qr"], s["pr"], s["gr"]
    else:
        t["ql"], t["pl"], t["gl"] = s["ql"], s["pl"], s["gl"]
    t["turn"] = div or s["turn"] or sub or uturn(t["rho"], t["pl"], t["pr"], im)
    t["lsw"], t["n"], t["acc"], t["div"] = lsw, t["n"] + s["n"], t["acc"] + s["acc"], div
    t["depth"] += 1


def _slice(st, off, n):
    return {k: v[off:off + n] for k, v in st.items()}


def build_full_tree(q0, p0, g0, logp0, fwd, bwd, im, jlp0, max_depth, seed):   # tree.rs:276-326
    rng = Rng(seed)
    t = dict(ql=list(q0), pl=list(p0), gl=list(g0), qr=list(q0), pr=list(p0), gr=list(g0), qp=list(q0), lpp=logp0, gp=list(g0),
             rho=list(p0), lsw=0.0, n=0, div=False, acc=0.0, turn=False, depth=0)
    cur = {True: 0, False: 0}
    for _ in range(max_depth):
        if t["div"] or t["turn"]:
            break
        right = rng.f64() > 0.5
        n = 1 << t["depth"]
        chain = fwd if right else bwd
        if cur[right] + n > len(chain["logp"]):
            break
        s = subtree(_slice(chain, cur[right], n), im, jlp0, t["depth"], right, [0], rng)
        cur[right] += n
        merge_into_trajectory(t, s, right, im, rng)
    return dict(q=t["qp"], logp=t["lpp"], grad=t["gp"], n_steps=t["n"], divergent=t["div"], accept_sum=t["acc"], depth=t["depth"])
