"""A third statement of Tree.build's plain path (lib/exmc/nuts/tree.ex: build_speculative / do_build :266-500,
build_subtree :1011-1203, merge_subtrees :1390-1476, merge_trajectories :1479-1568, check_uturn_rho :1578-1588,
log_sum_exp :1597-1605, result :1607-1618), written as the reference writes it -- a recursion over maps -- in
plain Python. TEST INFRASTRUCTURE: small cases only. It shares two things with the C checker, on purpose, so
that what is compared is the TREE LOGIC and nothing else: the leapfrog step (one call of exo_leapfrog per leaf,
libm mode = step_fn) and the random stream (exo_rng_uniform = :rand.uniform_s on exsss). Everything between --
leaf classification, the first-half short circuit, both merges with their draws, the three U-turn checks, the
endpoints by direction, the proposal selection, the doubling loop and its stops -- is restated here from the
reference's text, not from oracle/exmc_oracle.c."""
import ctypes as C
import math

import numpy as np

import oracle as O


class Tree:
    def __init__(self, model, inv_mass, jlp0, rng):
        self.m, self.im, self.jlp0, self.rng = model, np.asarray(inv_mass, dtype=np.float64), jlp0, rng
        self.L = O.lib()

    def uniform(self):
        return self.L.exo_rng_uniform(C.byref(self.rng))

    # tree.ex:1578-1588
    def uturn(self, rho, pl, pr):
        dr = dl = 0.0
        for r, a, b, im in zip(rho, pl, pr, self.im):
            v = r * im
            dr = dr + v * b
            dl = dl + v * a
        return dr < 0.0 or dl < 0.0

    # tree.ex:1597-1605
    @staticmethod
    def lse(a, b):
        m = max(a, b)
        if m == -math.inf or m == -1.0e300:
            return -1.0e300
        return m + math.log(math.exp(a - m) + math.exp(b - m))

    # tree.ex:1011-1141 (depth 0)
    def leaf(self, q, p, g, eps):
        qn, pn, lpn, gn, jlp = self.m.leapfrog(q, p, g, eps, self.im)
        if math.isfinite(jlp):
            d = jlp - self.jlp0
            div, lw, acc = d < -1000.0, d, min(1.0, math.exp(min(d, 0.0)))
        else:
            div, lw, acc = True, -1001.0, 0.0
        if div:     # the incoming state, logp_prop -1e30, accept_sum 0 (tree.ex:1052-1079)
            return dict(ql=q, pl=p, gl=g, qr=q, pr=p, gr=g, qp=q, lpp=-1.0e30, gp=g, rho=list(p), depth=0, lsw=lw, n=1,
                        div=True, acc=0.0, turn=False)
        return dict(ql=qn, pl=pn, gl=gn, qr=qn, pr=pn, gr=gn, qp=qn, lpp=lpn, gp=gn, rho=list(pn), depth=0, lsw=lw, n=1,
                    div=False, acc=acc, turn=False)

    # tree.ex:1144-1203
    def subtree(self, q, p, g, eps, depth):
        if depth == 0:
            return self.leaf(q, p, g, eps)
        first = self.subtree(q, p, g, eps, depth - 1)
        if first["div"] or first["turn"]:
            return first                                         # no draw, the second half is never built
        nq, np_, ng = (first["qr"], first["pr"], first["gr"]) if eps > 0 else (first["ql"], first["pl"], first["gl"])
        second = self.subtree(nq, np_, ng, eps, depth - 1)
        return self.merge_subtrees(first, second, eps)

    def _sub_checks(self, left, right):
        r2 = [a + b for a, b in zip(left["rho"], right["pl"])]
        if self.uturn(r2, left["pl"], right["pl"]):
            return True
        r3 = [a + b for a, b in zip(left["pr"], right["rho"])]
        return self.uturn(r3, left["pr"], right["pr"])

    # tree.ex:1390-1476
    def merge_subtrees(self, a, b, eps):
        lsw = self.lse(a["lsw"], b["lsw"])
        div = a["div"] or b["div"]
        u = self.uniform()
        src = b if u < math.exp(b["lsw"] - lsw) else a
        rho = [x + y for x, y in zip(a["rho"], b["rho"])]
        left, right = (a, b) if eps > 0 else (b, a)
        turn = div or b["turn"] or self.uturn(rho, left["pl"], right["pr"])
        if not turn and a["depth"] > 0:
            turn = self._sub_checks(left, right)
        return dict(ql=left["ql"], pl=left["pl"], gl=left["gl"], qr=right["qr"], pr=right["pr"], gr=right["gr"],
                    qp=src["qp"], lpp=src["lpp"], gp=src["gp"], rho=rho, depth=max(a["depth"], b["depth"]) + 1, lsw=lsw,
                    n=a["n"] + b["n"], div=div, acc=a["acc"] + b["acc"], turn=turn)

    # tree.ex:1479-1568
    def merge_trajectories(self, t, s, go_right):
        lsw = self.lse(t["lsw"], s["lsw"])
        div = t["div"] or s["div"]
        u = self.uniform()
        use_sub = (math.log(u) if u > 0.0 else -math.inf) < (s["lsw"] - t["lsw"])
        src = s if use_sub else t
        rho = [x + y for x, y in zip(t["rho"], s["rho"])]
        left, right = (t, s) if go_right else (s, t)
        turn = div or s["turn"] or self.uturn(rho, left["pl"], right["pr"])
        if not turn:
            turn = self._sub_checks(left, right)
        return dict(ql=left["ql"], pl=left["pl"], gl=left["gl"], qr=right["qr"], pr=right["pr"], gr=right["gr"],
                    qp=src["qp"], lpp=src["lpp"], gp=src["gp"], rho=rho, depth=t["depth"] + 1, lsw=lsw, n=t["n"] + s["n"],
                    div=div, acc=t["acc"] + s["acc"], turn=turn)


def build(model, q, p, logp, g, eps, inv_mass, max_depth, rng, jlp0):
    """Tree.build, plain path: -> dict(q, logp, grad, n_steps, divergent, accept_sum, depth)."""
    t = Tree(model, inv_mass, jlp0, rng)
    q, p, g = (np.asarray(x, dtype=np.float64) for x in (q, p, g))
    traj = dict(ql=q, pl=p, gl=g, qr=q, pr=p, gr=g, qp=q, lpp=logp, gp=g, rho=list(p), depth=0, lsw=0.0, n=0, div=False,
                acc=0.0, turn=False)
    depth = 0
    while not (depth >= max_depth or traj["div"] or traj["turn"]):              # tree.ex:340-387
        go_right = t.uniform() > 0.5                                             # tree.ex:403-405
        e = eps if go_right else -eps
        sq, sp, sg = (traj["qr"], traj["pr"], traj["gr"]) if go_right else (traj["ql"], traj["pl"], traj["gl"])
        sub = t.subtree(sq, sp, sg, e, depth)
        traj = t.merge_trajectories(traj, sub, go_right)
        depth += 1
    return dict(q=traj["qp"], logp=traj["lpp"], grad=traj["gp"], n_steps=traj["n"], divergent=traj["div"],
                accept_sum=traj["acc"], depth=depth)
