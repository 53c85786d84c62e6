"""A third statement of Exmc.NUTS.Sampler.sample/3's control flow (lib/exmc/nuts/sampler.ex: sample_from_compiled
:126-257, init_position :339-356, sample_momentum_fast :393-403, find_reasonable_epsilon_with_rng / search_epsilon
:451-530, run_warmup / run_phase / run_phase_ii / build_windows :537-785, nuts_step_warmup / nuts_step_with_stats
:794-925, run_sampling :929-973; step_size.ex:13-50; mass_matrix.ex:40-97) in plain Python, on top of tests/py_tree.py.
TEST INFRASTRUCTURE, small runs only. Shared with the C checker, on purpose: the model's value / gradient, the
leapfrog step, the kinetic energy and the random stream (all through ctypes) -- so that what is compared is the
sampler's LOGIC: the order of the draws, the discarded tree draws, the three warmup phases and their windows, the
depth cap of the first 200 iterations, dual averaging, Welford with Stan's regularisation, the step-size search."""
import ctypes as C
import math

import numpy as np

import oracle as O
import py_tree as PT

CFG = O.Cfg(0, 1)


class DA:                                                      # step_size.ex:13-50
    def __init__(self, eps, target):
        self.log_eps = self.log_eps_bar = math.log(eps)
        self.h_bar, self.mu, self.m, self.target = 0.0, math.log(10.0 * eps), 0, target

    def update(self, accept):
        self.m += 1
        eta = 1.0 / (self.m + 10.0)
        self.h_bar = (1.0 - eta) * self.h_bar + eta * (self.target - accept)
        self.log_eps = self.mu - math.sqrt(self.m) / 0.05 * self.h_bar
        mk = math.pow(self.m, -0.75)
        self.log_eps_bar = mk * self.log_eps + (1.0 - mk) * self.log_eps_bar

    eps = property(lambda s: math.exp(s.log_eps))
    final = property(lambda s: math.exp(s.log_eps_bar))


class Chain:
    def __init__(self, model, seed):
        self.m, self.L = model, O.lib()
        self.rng = O.Rng()
        self.L.exo_rng_seed(C.byref(self.rng), seed)
        self.div = 0

    def normal(self):
        return self.L.exo_rng_normal(C.byref(self.rng), 0)

    def uniform(self):
        return self.L.exo_rng_uniform(C.byref(self.rng))

    def momentum(self, im):                                    # sampler.ex:393-403
        return np.array([self.normal() / math.sqrt(v) for v in im])

    def jlp(self, logp, p, im):                                # leapfrog.ex:39-51
        return logp - self.L.exo_kinetic_energy(O.dptr(np.ascontiguousarray(p)), O.dptr(np.ascontiguousarray(im)), self.m.d, CFG)

    def find_eps(self, q, logp, g, im):                        # sampler.ex:451-530
        p = self.momentum(im)
        j0 = self.jlp(logp, p, im)

        def la(eps):
            j = self.m.leapfrog(q, p, g, eps, im)[4]
            return (j - j0) if (math.isfinite(j0) and math.isfinite(j)) else -1000.0
        eps = 1.0
        direction = 1.0 if la(eps) > math.log(0.5) else -1.0
        for _ in range(100):
            eps = eps * math.pow(2.0, direction)
            a = la(eps)
            crossed = a < math.log(0.5) if direction > 0 else a > math.log(0.5)
            if crossed or not math.isfinite(a):
                return max(eps, 1.0e-10)
        return max(eps, 1.0e-10)

    def step(self, st, eps, im, max_depth):                    # sampler.ex:794-925
        p = self.momentum(im)
        j0 = self.jlp(st["logp"], p, im)
        tree_rng = O.Rng()
        C.memmove(C.byref(tree_rng), C.byref(self.rng), C.sizeof(O.Rng))      # the tree's draws are not kept
        r = PT.build(self.m, st["q"], p, st["logp"], st["g"], eps, im, max_depth, tree_rng, j0)
        accept = r["accept_sum"] / r["n_steps"] if r["n_steps"] > 0 else 0.0
        self.uniform()                                          # {_, rng} = :rand.uniform_s(rng), sampler.ex:897
        self.div += bool(r["divergent"])
        return dict(q=r["q"], logp=r["logp"], g=r["grad"]), accept, dict(depth=r["depth"], n_steps=r["n_steps"],
                                                                        divergent=r["divergent"], energy=-j0)


def windows(frm, to, base=25):                                 # sampler.ex:765-785
    out, cur, w = [], frm, base
    while cur < to:
        rem = to - cur
        size = rem if rem <= w * 1.5 else w
        out.append((cur, cur + size))
        cur, w = cur + size, w * 2
    return out


def sample(model, init_q=None, num_warmup=1000, num_samples=1000, max_tree_depth=10, target_accept=0.8, seed=0,
           warm_start=None):
    ch = Chain(model, seed)
    d = model.d
    q = np.array([0.1 * ch.normal() for _ in range(d)]) if init_q is None else np.asarray(init_q, dtype=np.float64)
    lp, g = model.logp_grad(q, CFG)
    st = dict(q=q, logp=lp, g=g)
    if warm_start is not None:                                 # sampler.ex:167-197: the previous tuning, at most 50 iterations
        eps, im = float(warm_start[0]), np.asarray(warm_start[1], dtype=np.float64).copy()
        num_warmup = min(num_warmup, 50)
    else:
        im = np.ones(d)
        eps = ch.find_eps(st["q"], st["logp"], st["g"], im)

    def phase(st, da, n, depth_of=lambda i: max_tree_depth, frm=0, welford=None):
        for i in range(frm, frm + n):
            before = ch.div
            st, acc, _ = ch.step(st, da.eps, im, depth_of(i))
            da.update(acc)
            if welford is not None and ch.div == before:      # mass_matrix.ex:40-54, skipped on a divergent step
                welford["n"] += 1
                delta = st["q"] - welford["mean"]
                welford["mean"] = welford["mean"] + delta / float(welford["n"])
                welford["m2"] = welford["m2"] + delta * (st["q"] - welford["mean"])
        return st

    if num_warmup > 0:                                         # sampler.ex:537-621
        init_buffer, adapt_end = min(75, num_warmup // 3), num_warmup - 50
        da = DA(eps, target_accept)
        st = phase(st, da, init_buffer)
        eps = da.eps
        if adapt_end <= init_buffer:
            eps = da.final
        else:
            for a, b in windows(init_buffer, adapt_end):       # sampler.ex:663-762
                w = dict(n=0, mean=np.zeros(d), m2=np.zeros(d))
                da = DA(eps, target_accept)
                st = phase(st, da, b - a, depth_of=lambda i: min(max_tree_depth, 8) if i < 200 else max_tree_depth,
                           frm=a, welford=w)
                if w["n"] < 3:                                 # mass_matrix.ex:77-97
                    im = np.ones(d)
                else:
                    var = np.maximum(w["m2"] / float(w["n"] - 1), 1.0e-6)
                    alpha = 5.0 / (w["n"] + 5.0)
                    im = (1.0 - alpha) * var + alpha * 1.0e-3
                eps = ch.find_eps(st["q"], st["logp"], st["g"], im)
            da = DA(eps, target_accept)
            st = phase(st, da, num_warmup - adapt_end)
            eps = da.final
    out = dict(draws=[], tree_depth=[], n_steps=[], divergent=[], accept_prob=[], energy=[], logp=[])
    for _ in range(num_samples):                               # sampler.ex:929-973
        st, acc, info = ch.step(st, eps, im, max_tree_depth)
        out["draws"].append(st["q"])
        out["logp"].append(st["logp"])
        out["tree_depth"].append(info["depth"])
        out["n_steps"].append(info["n_steps"])
        out["divergent"].append(int(info["divergent"]))
        out["accept_prob"].append(acc)
        out["energy"].append(info["energy"])
    return {k: np.array(v) for k, v in out.items()}, dict(step_size=eps, inv_mass=im, divergences=ch.div)
