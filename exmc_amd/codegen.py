"""Builder IR -> HIP kernel source (restricted first cut).

The reference turns a Builder IR into a logp closure by walking the nodes (compiler.ex:200-269,
one term per free RV and per obs node, summed by sum_logps :394-395) and leaves the gradient to
Nx.Defn reverse-mode AD (compiler.ex:131-141); EXLA then JIT-compiles the graph. Here the same
walk emits straight-line C for value + gradient, hipcc compiles it into a copy of the NUTS
kernels (one lane per chain, the whole position in registers), and the result is loaded as a
plug-in library with the same C ABI as libexmc_hip.so (include/exmc_hip.h, kind
EXMC_MODEL_CUSTOM).

What is covered (everything else raises CodegenError -- there is no interpreter fallback):
  * scalar free RVs (a vector free RV with an ELEMENTWISE distribution does not compile in the
    reference either: sum_logps reshapes every term to {});
  * Normal, HalfNormal, HalfCauchy, Exponential, Cauchy, Laplace, Lognormal, StudentT and
    (as an observation) Bernoulli, each restated operation by operation from lib/exmc/dist/;
  * transforms nil, :log, :softplus, :logit (transform.ex:15-66);
  * params that are numbers, vectors (observation targets only) or string refs to free RVs
    (resolve_params_constrained, compiler.ex:436-463), the non-centred rewrite of
    Normal(ref, ref) RVs included (rewrite/non_centered_parameterization.ex:50-55);
  * obs nodes with scalar or vector values and reduce :sum (builder.ex:97-102);
  * (round 2) vector free RVs whose distribution returns a scalar: GaussianRandomWalk
    (dist/gaussian_random_walk.ex:21-57) and MvNormal with constant mu / cov
    (dist/mv_normal.ex:20-47: precision and log-determinant prepared eagerly on the host); they
    occupy `length` consecutive flat entries (point_map.ex:30-60) and may be referenced as vector
    params (e.g. as the mean of a vector obs);
  * (round 2) Custom distributions (dist/custom.ex): the closure is a Python callable over a
    declarative op set (`Ops`: add sub mul div neg exp log log1p abs max min lit sum), called with
    the value and the resolved params -- validate_posteriordb.exs:279-295 writes eight schools'
    likelihood that way;
  * (round 2) meas_obs nodes with {:affine, a, b} and {:matmul, A} (compiler.ex:258-266, 342-369):
    eager, i.e. constants of the data -- they shift the log-density and leave the gradient alone.

Numeric contract of the generated code:
  * forward value: the reference's Nx operation sequence with its f32-rounded literals
    (Nx.tensor(<float>) is f32), sums left to right from the first element as Nx.sum on the
    BinaryBackend, exp/log/log1p from include/exmc_detmath.h;
  * terms in Map.values order of the node map = ids sorted as strings (exact for <= 32 nodes,
    where Erlang maps are sorted; a larger map iterates in its hash order: the exporter on the
    BEAM writes that order into the document ("term_order", IR.order) and the terms follow it --
    without one, sorted ids);
  * gradient: reverse-mode accumulation over the same graph with the local rules listed at
    `_Grad` below. Nx's own AD rewrites are not restated ("parity unpinned"), so the
    checker for a generated model is the same generated C compiled for the host
    (tests/gen_checker.py), not a hand-derived gradient.
  * everything that depends only on the model's data is evaluated once on the host at
    exmc_hip_model_create (exmc_gen_fold) and read by the kernels through uniform scalar loads.
"""
import hashlib
import math
import os
import subprocess
import sys
import time

import numpy as np

from . import build as _build
from .models import ModelSpec

CUSTOM = 6
MAX_D = 20   # one lane per chain: 5*D+3 doubles per tree node, one level must fit LDS
MAX_D_LANES = 256   # EXMC_HIP_MAX_D (include/exmc_hip.h): the lane layout, DPL = ceil(D / lanes)
MAX_NODES_SORTED = 32


class CodegenError(ValueError):
    pass


class F32(float):
    """A param written Nx.tensor(<float>) without a type: an f32 tensor. Its value is the f32
    rounding, and an operation whose operands are all such constants yields an f32 tensor again
    (Nx's type inference; the BinaryBackend computes in Erlang floats and stores f32), e.g.
    Nx.log(lambda) of exponential.ex:16 for lambda = Nx.tensor(50.0)."""

    def __new__(cls, x):
        return super().__new__(cls, float(np.float32(x)))


def _f32(x):
    return float(np.float32(x))


LOG_2PI_F32 = _f32(math.log(_f32(2.0 * math.pi)))   # Nx.log(Nx.tensor(2*pi)), normal.ex:19,22
LOG_2_OVER_PI_F32 = _f32(math.log(2.0 / math.pi))   # half_cauchy.ex:22
NEG_LOG_PI_F32 = _f32(-math.log(math.pi))           # cauchy.ex:21
LOG_2_F32 = _f32(math.log(_f32(2.0)))               # half_normal.ex:21
PI_F32 = _f32(math.pi)                              # student_t.ex:27
TINY_F32 = _f32(1.0e-30)
HALF_LOG_2PI_F32 = _f32(0.5 * math.log(2.0 * math.pi))   # math.ex:30
BERN_LO = _f32(1.0e-7)                              # bernoulli.ex:20-21 (f32 arithmetic)
BERN_HI = _f32(np.float32(1.0) - np.float32(1.0e-7))
LANCZOS = [0.99999999999980993, 676.5203681218851, -1259.1392167224028, 771.32342877765313,
           -176.61502916214059, 12.507343278686905, -0.13857109526572012, 9.9843695780195716e-6,
           1.5056327351493116e-7]                   # math.ex:10-20 (Nx.tensor(c) is f32)


# ---------------------------------------------------------------------------------------------
# Builder mirror (lib/exmc/builder.ex:34-67)
# ---------------------------------------------------------------------------------------------
class IR:
    def __init__(self):
        self.nodes = {}
        self.data_tensor = None
        self.term_order = None   # the ids in the BEAM's own Map.values order, when an exporter sent it

    def order(self, ids):
        """Map.keys(ir.nodes) as the VM that exported the IR iterates it (elixir/.../hip_export.ex:
        "term_order"). compiler.ex:176-180 builds the terms in Map.values order and sum_logps adds
        them in that order; up to 32 keys an Erlang map is sorted, a larger one iterates in the order
        of its internal hash, which only the VM knows -- so the exporter writes it down."""
        ids = list(ids)
        if sorted(ids) != sorted(self.nodes) or len(set(ids)) != len(ids):
            raise CodegenError("term_order must name every node id exactly once")
        self.term_order = ids
        return self

    def data(self, tensor):
        """Builder.data (builder.ex:19-21): the model's observation tensor (rank 0, 1 or 2). A
        param written as the string "__obs_data" resolves to it (compiler.ex:103-118). Like every
        other constant of a generated model it travels in the data array exmc_hip_model_create
        receives, not in the generated text: the same plug-in serves any data of the same shape."""
        t = np.asarray(tensor, dtype=np.float64)
        if t.ndim > 2:
            raise CodegenError("Builder.data takes a scalar, a vector or a matrix")
        self.data_tensor = t
        return self

    def _add(self, id_, node):
        if not isinstance(id_, str):
            raise CodegenError("node ids are strings")
        if id_ in self.nodes:
            raise CodegenError("duplicate node id %r" % id_)
        self.nodes[id_] = node
        return self

    def rv(self, id_, dist, params, transform=None):
        return self._add(id_, dict(op="rv", dist=dist, params=dict(params), transform=transform))

    def obs(self, id_, rv_id, value, **opts):
        """Builder.obs (builder.ex:38-102) with its meta: reduce (:sum | :mean | :logsumexp),
        weight (scalar or vector), mask (vector of booleans), censored (:right | :left, or
        :interval with value = {"lower": a, "upper": b})."""
        for k in opts:
            if k not in ("reduce", "weight", "mask", "censored", "likelihood"):
                raise CodegenError("obs option %r is not covered by the generator" % k)
        cens = opts.get("censored")
        if cens not in (None, "right", "left", "interval"):
            raise CodegenError("censored: %r is not covered" % (cens,))
        if cens == "interval":
            if not (isinstance(value, dict) and set(value) == {"lower", "upper"}):
                raise CodegenError("an interval-censored obs takes value = {'lower': a, 'upper': b}")
            lo = np.asarray(value["lower"], dtype=np.float64)
            hi = np.asarray(value["upper"], dtype=np.float64)
            if lo.shape != hi.shape:
                raise CodegenError("interval bounds differ in shape")
            v = np.stack([lo, hi], axis=-1)       # [..., 2]
            shape = lo.shape
        else:
            v = np.asarray(value, dtype=np.float64)
            shape = v.shape
        reduce_ = opts.get("reduce")
        if len(shape) > 0 and reduce_ is None:
            reduce_ = "sum"                      # builder.ex:97-102
        if len(shape) > 1:
            raise CodegenError("obs values are scalars or vectors")
        if reduce_ not in (None, "sum", "mean", "logsumexp"):
            raise CodegenError("reduce %r is not covered" % (reduce_,))
        if len(shape) == 1 and reduce_ is None:
            raise CodegenError("vector obs needs a reduce")
        meta = dict(reduce=reduce_, censored=cens)
        if opts.get("likelihood") is False:
            meta["likelihood"] = False           # compiler.ex:244-245: the node contributes no term
        if opts.get("weight") is not None:
            w = np.asarray(opts["weight"], dtype=np.float64)
            if w.ndim > 1 or (w.ndim == 1 and w.shape != shape):
                raise CodegenError("weight must be a scalar or match the value")
            meta["weight"] = w
        if opts.get("mask") is not None:
            mk = np.asarray(opts["mask"]).astype(bool)
            if mk.shape != shape or len(shape) > 1:
                raise CodegenError("mask must match the value (a scalar for a scalar, a vector for a vector)")
            meta["mask"] = mk                    # (a scalar mask on a scalar obs: test/exmc_test.exs:188-209)
        return self._add(id_, dict(op="obs", target=rv_id, value=v, meta=meta))

    def det(self, id_, fun, args):
        """Builder.det (builder.ex:80-83): a deterministic node. It contributes no term
        (compiler.ex:269); an obs of det("affine", [a, b, rv]) or det("matmul", [A, rv]) becomes a
        meas_obs under `rewrite` (rewrite/lift_measurable_affine.ex, lift_measurable_matmul.ex)."""
        return self._add(id_, dict(op="det", fun=fun, args=list(args)))

    def meas_obs(self, id_, rv_id, value, op_info, meta=None):
        """Builder.meas_obs (builder.ex:69-82): an observation of a measurable function of an rv;
        op_info = ("affine", a, b) for y = a x + b or ("matmul", A) for y = A x. `meta` is an obs
        node's meta carried over by the lifting passes (weight / mask / reduce)."""
        v = np.asarray(value, dtype=np.float64)
        if v.ndim > 1:
            raise CodegenError("meas_obs values are scalars or vectors")
        kind = op_info[0]
        if kind == "affine" and len(op_info) == 3:
            # a, b scalars, or vectors that broadcast against the value (compiler.ex:363-369 divides and
            # subtracts tensors; test/exmc_test.exs:348-372)
            a_, b_ = np.asarray(op_info[1], dtype=np.float64), np.asarray(op_info[2], dtype=np.float64)
            for c in (a_, b_):
                if c.ndim > 1 or (c.ndim == 1 and (v.ndim != 1 or c.shape != v.shape)):
                    raise CodegenError("affine meas_obs: a and b are scalars or vectors matching the value")
            info = ("affine", float(a_) if a_.ndim == 0 else a_, float(b_) if b_.ndim == 0 else b_)
        elif kind == "matmul" and len(op_info) == 2:
            a = np.asarray(op_info[1], dtype=np.float64)
            if a.ndim != 2 or a.shape[0] != a.shape[1] or v.ndim != 1 or a.shape[0] != v.shape[0]:
                raise CodegenError("matmul meas_obs needs a square matrix matching the value")
            info = ("matmul", a)
        else:
            raise CodegenError("meas_obs op %r is not covered" % (kind,))
        return self._add(id_, dict(op="meas_obs", target=rv_id, value=v, info=info, meta=meta))


def simple_ir(y=None):
    """The reference README's first model: mu ~ N(0,5), sigma ~ Exponential(1) [:log],
    y ~ N(mu, sigma) observed."""
    from .models import SIMPLE_Y
    ir = IR()
    ir.rv("mu", "normal", dict(mu=0.0, sigma=5.0))
    ir.rv("sigma", "exponential", {"lambda": 1.0}, transform="log")
    ir.rv("y", "normal", dict(mu="mu", sigma="sigma"))
    ir.obs("y_obs", "y", SIMPLE_Y if y is None else y)
    return ir


def eight_schools_ir(y=None, sigma=None):
    """Eight schools as plain Builder nodes (26 of them): theta_j ~ N("mu","tau") is what the
    non-centred rewrite turns into theta_j_raw ~ N(0,1) with theta_j = mu + tau*raw
    (rewrite/non_centered_parameterization.ex:50-55)."""
    from .models import EIGHT_SCHOOLS_SIGMA, EIGHT_SCHOOLS_Y
    y = EIGHT_SCHOOLS_Y if y is None else y
    sigma = EIGHT_SCHOOLS_SIGMA if sigma is None else sigma
    ir = IR()
    ir.rv("mu", "normal", dict(mu=0.0, sigma=5.0))
    ir.rv("tau", "half_cauchy", dict(scale=5.0), transform="log")
    for j in range(len(y)):
        ir.rv("theta_%d" % j, "normal", dict(mu="mu", sigma="tau"))
        ir.rv("y_%d" % j, "normal", dict(mu="theta_%d" % j, sigma=sigma[j]))
        ir.obs("y_obs_%d" % j, "y_%d" % j, y[j])
    return ir


def sv_ir(returns):
    """Stochastic volatility as STANDARD_BENCHMARKS.md:51-61 describes it ("100 separate Normal
    random variables connected by string references"): sigma ~ Exponential(50), nu ~
    Exponential(0.1) (both Nx.tensor(<float>) = f32, both :log), s_1 ~ N(0, sigma), s_t ~ N(s_{t-1},
    sigma), r_t ~ StudentT(nu, 0, exp(s_t)) observed. The benchmark's script is not in the
    reference repository; the likelihood is written here the way its sibling scripts write theirs
    (a Custom closure over refs, validate_posteriordb.exs:279-295): the sum of student_t.ex's
    logpdf over the returns. Compile with ncp=False (the hand-written kind is centred too)."""
    r = [float(v) for v in returns]
    T = len(r)
    ir = IR()
    ir.rv("sigma", "exponential", {"lambda": F32(50.0)}, transform="log")
    ir.rv("nu", "exponential", {"lambda": F32(0.1)}, transform="log")
    ir.rv("s_1", "normal", dict(mu=0.0, sigma="sigma"))
    for t in range(2, T + 1):
        ir.rv("s_%d" % t, "normal", dict(mu="s_%d" % (t - 1), sigma="sigma"))

    def lik(o, _x, p):
        return o.sum([o.logpdf("student_t", o.data(r[t - 1]),
                               dict(df=p["nu"], loc=o.lit(0.0), scale=o.exp(p["s_%d" % t])))
                      for t in range(1, T + 1)])
    params = {"s_%d" % t: "s_%d" % t for t in range(1, T + 1)}
    params.update(nu="nu", logpdf=lik)
    ir.rv("returns", "custom", params)
    ir.obs("returns_obs", "returns", 0.0)
    return ir


def logistic_ir(X, y):
    """Logistic regression (STANDARD_BENCHMARKS.md:41-49, :187 "20-way Nx.stack + 500 x 20 matvec"):
    alpha, beta_j ~ N(0, 10); y_n ~ Bernoulli(sigmoid(alpha + X beta)) as a Custom closure over the
    21 refs -- Nx.sigmoid, bernoulli.ex's clipped logpdf, Nx.sum over the observations."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n, k = X.shape
    ir = IR()
    ir.rv("alpha", "normal", dict(mu=0.0, sigma=10.0))
    for j in range(1, k + 1):
        ir.rv("beta_%d" % j, "normal", dict(mu=0.0, sigma=10.0))

    def lik(o, _x, p):
        beta = [p["beta_%d" % j] for j in range(1, k + 1)]
        terms = []
        for i in range(n):
            eta = p["alpha"]
            for j in range(k):
                eta = o.add(eta, o.mul(o.data(X[i, j]), beta[j]))
            terms.append(o.logpdf("bernoulli", o.data(y[i]), dict(p=o.sigmoid(eta))))
        return o.sum(terms)
    params = {"beta_%d" % j: "beta_%d" % j for j in range(1, k + 1)}
    params.update(alpha="alpha", logpdf=lik)
    ir.rv("y", "custom", params)
    ir.obs("y_obs", "y", 0.0)
    return ir


def radon_ir(u, start, floor, y, names=None):
    """The radon varying-intercept model of notebooks/09_radon_bhm.livemd ("The Radon Model"; its
    benchmark/radon_model.exs is not in the reference repository): five hyper-parameters, 85
    alpha_raw_j ~ N(0, 1), alpha_j = mu_alpha + gamma_u u_j + sigma_alpha alpha_raw_j "reconstructed
    inside a Custom dist closure", y_ij ~ N(alpha_j + beta floor_ij, sigma_y). `start` are the
    county offsets into floor / y; names[j] = id of county j's intercept."""
    u = np.asarray(u, dtype=np.float64)
    start = np.asarray(start).astype(int)
    J = len(u)
    names = ["alpha_raw_%d" % j for j in range(J)] if names is None else list(names)
    ir = IR()
    ir.rv("mu_alpha", "normal", dict(mu=0.0, sigma=10.0))
    ir.rv("gamma_u", "normal", dict(mu=0.0, sigma=5.0))
    ir.rv("sigma_alpha", "half_cauchy", dict(scale=2.5), transform="log")
    ir.rv("sigma_y", "half_cauchy", dict(scale=2.5), transform="log")
    ir.rv("beta", "normal", dict(mu=0.0, sigma=5.0))
    for nm in names:
        ir.rv(nm, "normal", dict(mu=0.0, sigma=1.0))

    def lik(o, _x, p):
        terms = []
        for j in range(J):
            alpha = o.add(o.add(p["mu_alpha"], o.mul(p["gamma_u"], o.data(u[j]))),
                          o.mul(p["sigma_alpha"], p[names[j]]))
            for i in range(start[j], start[j + 1]):
                mean = o.add(alpha, o.mul(p["beta"], o.data(floor[i])))
                terms.append(o.logpdf("normal", o.data(y[i]), dict(mu=mean, sigma=p["sigma_y"])))
        return o.sum(terms)
    params = {nm: nm for nm in names}
    params.update(mu_alpha="mu_alpha", gamma_u="gamma_u", sigma_alpha="sigma_alpha", sigma_y="sigma_y",
                  beta="beta", logpdf=lik)
    ir.rv("radon", "custom", params)
    ir.obs("radon_obs", "radon", 0.0)
    return ir


# ---------------------------------------------------------------------------------------------
# expression graph (hash-consed; a vector is a python list of scalar nodes)
# ---------------------------------------------------------------------------------------------
class _Graph:
    # leaves and whether they depend on the data only; `red j` is reduced sum j of the lane layout
    # (codegen_lanes.py): a run-time value the differentiation does not look into
    LEAF_CONST = {"lit": True, "data": True, "q": False, "red": False}

    def __init__(self):
        self.ops = []      # (op, args) with args = tuple of node indices / payload
        self.const = []    # depends on data only
        self.key = {}
        self.data = []     # raw model data, in first-use order
        self.sums = {}     # result of a left-to-right sum -> its summands (_sum_left)
        self.f32 = set()   # data nodes that are f32 tensors (class F32)

    _F32_FOLD = {"add": lambda a, b: a + b, "sub": lambda a, b: a - b, "mul": lambda a, b: a * b,
                 "div": lambda a, b: a / b, "neg": lambda a: -a, "log": math.log, "exp": math.exp}

    def datum32(self, x):
        i = self.datum(float(np.float32(x)))
        self.f32.add(i)
        return i

    def _node(self, op, *args):
        if self.f32 and op in self._F32_FOLD and all(a in self.f32 for a in args):
            vals = [self.data[self.ops[a][1]] for a in args]
            return self.datum32(self._F32_FOLD[op](*vals))
        k = (op,) + args
        i = self.key.get(k)
        if i is None:
            i = len(self.ops)
            self.ops.append(k)
            c = self.LEAF_CONST[op] if op in self.LEAF_CONST else all(self.const[a] for a in args)
            self.const.append(c)
            self.key[k] = i
        return i

    def lit(self, x):
        return self._node("lit", float(x).hex())

    def lit_value(self, i):
        return float.fromhex(self.ops[i][1]) if self.ops[i][0] == "lit" else None

    def datum(self, x):
        self.data.append(float(x))
        return self._node("data", len(self.data) - 1)

    def q(self, i):
        return self._node("q", i)

    def add(self, a, b): return self._node("add", a, b)
    def sub(self, a, b): return self._node("sub", a, b)
    def div(self, a, b): return self._node("div", a, b)
    def exp(self, a): return self._node("exp", a)
    def log(self, a): return self._node("log", a)
    def log1p(self, a): return self._node("log1p", a)
    def erf(self, a): return self._node("erf", a)
    def max(self, a, b): return self._node("max", a, b)
    def min(self, a, b): return self._node("min", a, b)
    def abs(self, a): return self._node("abs", a)
    def sel_gt(self, a, b, x, y): return self._node("sel_gt", a, b, x, y)

    def neg(self, a):
        if self.ops[a][0] == "neg":
            return self.ops[a][1]
        return self._node("neg", a)

    def mul(self, a, b):
        # only rewrites that are exact in IEEE arithmetic
        for x, y in ((a, b), (b, a)):
            v = self.lit_value(x)
            if v == 1.0:
                return y
            if v == -1.0:
                return self.neg(y)
        return self._node("mul", a, b)

    def recip(self, a):
        return self.div(self.lit(1.0), a)


class _Grad:
    """Reverse-mode accumulation. Local rules (g = adjoint of the result y):
         a+b: g, g        a-b: g, -g        a*b: g*b, g*a        -a: -g
         a/b: g*r, -((g*r)*y) with r = 1/b (one reciprocal per distinct denominator)
         exp a: g*y       log a: g*(1/a)    log1p a: g*(1/(1+a))
         max(a,b): g where a > b / g where b > a (nothing on a tie, as the clamps in
                   exmc_models.hpp); min likewise; |a|: g, -g or 0 by the sign of a
       Contributions to one node are added in decreasing order of the consumer's index."""

    def __init__(self, g, root, const_num_ok=None):
        self.g = g
        self.adj = {root: g.lit(1.0)}
        # lane layout only: c / b with a constant numerator c that is finite and non-zero (the
        # callback decides, it knows the per-unit tables) takes d/db = -(y * y) * (1 / c) with 1 / c
        # folded into the tables -- no second quotient at run time
        self.const_num_ok = const_num_ok

    def _acc(self, node, contrib):
        if self.g.const[node]:
            return
        cur = self.adj.get(node)
        self.adj[node] = contrib if cur is None else self.g.add(cur, contrib)

    def run(self, n_forward):
        g = self.g
        zero = g.lit(0.0)
        for y in range(n_forward - 1, -1, -1):
            gy = self.adj.get(y)
            if gy is None or g.const[y]:
                continue
            op = g.ops[y][0]
            a = g.ops[y][1:]
            if op == "add":
                self._acc(a[0], gy); self._acc(a[1], gy)
            elif op == "sub":
                self._acc(a[0], gy); self._acc(a[1], g.neg(gy))
            elif op == "mul":
                self._acc(a[0], g.mul(gy, a[1])); self._acc(a[1], g.mul(gy, a[0]))
            elif op == "neg":
                self._acc(a[0], g.neg(gy))
            elif op == "div":
                if (self.const_num_ok is not None and g.const[a[0]] and not g.const[a[1]]
                        and self.const_num_ok(a[0])):
                    y2 = g.mul(y, y)
                    if g.lit_value(a[0]) != 1.0:
                        y2 = g.mul(y2, g.recip(a[0]))
                    self._acc(a[1], g.neg(g.mul(gy, y2)))
                    continue
                gr = g.mul(gy, g.recip(a[1]))
                self._acc(a[0], gr)
                if not g.const[a[1]]:
                    self._acc(a[1], g.neg(g.mul(gr, y)))
            elif op == "exp":
                self._acc(a[0], g.mul(gy, y))
            elif op == "log":
                self._acc(a[0], g.mul(gy, g.recip(a[0])))
            elif op == "log1p":
                self._acc(a[0], g.mul(gy, g.recip(g.add(g.lit(1.0), a[0]))))
            elif op == "erf":     # 2/sqrt(pi) * exp(-a^2)
                self._acc(a[0], g.mul(gy, g.mul(g.lit(2.0 / math.sqrt(math.pi)), g.exp(g.neg(g.mul(a[0], a[0]))))))
            elif op == "max":
                self._acc(a[0], g.sel_gt(a[0], a[1], gy, zero))
                self._acc(a[1], g.sel_gt(a[1], a[0], gy, zero))
            elif op == "min":
                self._acc(a[0], g.sel_gt(a[1], a[0], gy, zero))
                self._acc(a[1], g.sel_gt(a[0], a[1], gy, zero))
            elif op == "abs":
                self._acc(a[0], g.sel_gt(a[0], zero, gy, g.sel_gt(zero, a[0], g.neg(gy), zero)))
            elif op == "sel_gt":
                self._acc(a[2], g.sel_gt(a[0], a[1], gy, zero))
                self._acc(a[3], g.sel_gt(a[0], a[1], zero, gy))
            elif op in ("q", "qown", "ext", "gat", "red"):
                pass
            else:
                raise CodegenError("no gradient rule for %s" % op)


# ---------------------------------------------------------------------------------------------
# distributions and transforms, operation by operation
# ---------------------------------------------------------------------------------------------
def _softplus(g, x):
    # transform.ex:293-296
    return g.add(g.max(x, g.lit(0.0)), g.log1p(g.exp(g.neg(g.abs(x)))))


def _clamp200(g, z):
    # transform.ex:27,53: max(lo, min(z, hi)), f64 range (jit.ex precision :f64)
    return g.max(g.lit(-200.0), g.min(z, g.lit(200.0)))


def _apply_transform(g, t, z):
    if t is None:
        return z
    if t == "log":
        return g.exp(_clamp200(g, z))
    if t == "softplus":
        return _softplus(g, z)
    if t == "logit":
        return g.exp(g.neg(_softplus(g, g.neg(z))))
    raise CodegenError("transform %r is not covered" % (t,))


def _log_abs_det_jacobian(g, t, z):
    if t is None:
        return g.lit(0.0)
    if t == "log":
        return _clamp200(g, z)
    if t == "softplus":
        return g.neg(_softplus(g, g.neg(z)))
    if t == "logit":
        return g.add(g.neg(_softplus(g, g.neg(z))), g.neg(_softplus(g, z)))
    raise CodegenError("transform %r is not covered" % (t,))


def _lgamma(g, x):
    # math.ex:27-52
    t = g.add(x, g.lit(6.5))
    ag = g.lit(_f32(LANCZOS[0]))
    terms = [ag]
    for i, c in enumerate(LANCZOS[1:]):
        terms.append(g.div(g.lit(_f32(c)), g.add(x, g.lit(float(i)))))
        ag = g.add(ag, terms[-1])
    if hasattr(g, "sums"):
        g.sums[ag] = terms          # the lane layout evaluates the series one quotient per lane
    r = g.add(g.lit(HALF_LOG_2PI_F32), g.mul(g.sub(x, g.lit(0.5)), g.log(t)))
    return g.add(g.sub(r, t), g.log(ag))


def _erfc(g, x):
    # censored.ex:55-68 (Abramowitz & Stegun 7.1.26; every Nx.tensor(<float>) literal is f32)
    ax = g.abs(x)
    t_ = g.div(g.lit(1.0), g.add(g.lit(1.0), g.mul(g.lit(_f32(0.3275911)), ax)))
    poly = g.lit(0.0)
    for c in reversed([0.254829592, -0.284496736, 1.421413741, -1.453152027, 1.061405429]):
        poly = g.add(g.lit(_f32(c)), g.mul(t_, poly))
    res = g.mul(g.mul(t_, poly), g.exp(g.neg(g.mul(ax, ax))))
    return g.sel_gt(g.lit(0.0), x, g.sub(g.lit(2.0), res), res)     # select(x < 0, 2 - r, r)


def _normal_cdf(g, z):
    # censored.ex:44-46
    return g.mul(g.lit(0.5), _erfc(g, g.neg(g.div(z, g.lit(_f32(math.sqrt(2.0)))))))


def _censored_loglik(g, kind, value, dist, p):
    # censored.ex:14-42
    tiny = g.lit(TINY_F32)
    if dist == "weibull" and kind == "right":
        k, lam = _need(p, dist, "k", "lambda")
        z = g.sub(g.log(value), g.log(lam))              # weibull.ex:51-54 log_survival
        return g.neg(g.exp(g.mul(k, z)))
    if dist == "normal":
        mu, sigma = _need(p, dist, "mu", "sigma")
        ss = g.max(sigma, tiny)
        if kind == "interval":
            lo, hi = value
            z_lo = g.div(g.sub(lo, mu), ss)
            z_hi = g.div(g.sub(hi, mu), ss)
            return g.log(g.sub(_normal_cdf(g, z_hi), _normal_cdf(g, z_lo)))
        z = g.div(g.sub(value, mu), ss)
        if kind == "left":
            return g.log(_normal_cdf(g, z))
        return g.log(_normal_cdf(g, g.neg(z)))             # log_sf(z) = log_cdf(-z)
    raise CodegenError("censored: %s is not covered for distribution %r" % (kind, dist))


def _logsumexp(g, xs):
    # Nx.logsumexp (third-party Nx 0.10): the maximum, then log(sum(exp(x - max))) + max
    m = xs[0]
    for e in xs[1:]:
        m = g.max(m, e)
    return g.add(g.log(_sum_left(g, [g.exp(g.sub(e, m)) for e in xs])), m)


def _need(params, dist, *names):
    if sorted(params) != sorted(names):
        raise CodegenError("%s takes params %s, got %s" % (dist, sorted(names), sorted(params)))
    return [params[n] for n in names]


def _logpdf(g, dist, x, p):
    tiny = g.lit(TINY_F32)
    if dist == "normal":          # normal.ex:15-24
        mu, sigma = _need(p, dist, "mu", "sigma")
        ss = g.max(sigma, tiny)
        z = g.div(g.sub(x, mu), ss)
        log_term = g.add(g.lit(LOG_2PI_F32), g.mul(g.lit(2.0), g.log(ss)))
        return g.mul(g.lit(-0.5), g.add(g.mul(z, z), log_term))
    if dist == "half_normal":     # half_normal.ex:15-22
        (sigma,) = _need(p, dist, "sigma")
        ss = g.max(sigma, tiny)
        z = g.div(x, ss)
        base = g.mul(g.lit(-0.5), g.add(g.mul(z, z), g.lit(LOG_2PI_F32)))
        return g.add(base, g.sub(g.lit(LOG_2_F32), g.log(ss)))
    if dist == "half_cauchy":     # half_cauchy.ex:17-25
        (scale,) = _need(p, dist, "scale")
        ss = g.max(scale, tiny)
        z = g.div(x, ss)
        r = g.sub(g.lit(LOG_2_OVER_PI_F32), g.log(ss))
        return g.sub(r, g.log(g.add(g.lit(1.0), g.mul(z, z))))
    if dist == "exponential":     # exponential.ex:15-17
        (lam,) = _need(p, dist, "lambda")
        return g.sub(g.log(lam), g.mul(lam, x))
    if dist == "cauchy":          # cauchy.ex:15-24
        loc, scale = _need(p, dist, "loc", "scale")
        ss = g.max(scale, tiny)
        z = g.div(g.sub(x, loc), ss)
        r = g.sub(g.lit(NEG_LOG_PI_F32), g.log(ss))
        return g.sub(r, g.log(g.add(g.lit(1.0), g.mul(z, z))))
    if dist == "laplace":         # laplace.ex:15-21
        mu, b = _need(p, dist, "mu", "b")
        sb = g.max(b, tiny)
        return g.sub(g.neg(g.log(g.mul(g.lit(2.0), sb))), g.div(g.abs(g.sub(x, mu)), sb))
    if dist == "lognormal":       # lognormal.ex:15-25
        mu, sigma = _need(p, dist, "mu", "sigma")
        ss = g.max(sigma, tiny)
        lx = g.log(x)
        z = g.div(g.sub(lx, mu), ss)
        log_term = g.add(g.lit(LOG_2PI_F32), g.mul(g.lit(2.0), g.log(ss)))
        return g.sub(g.mul(g.lit(-0.5), g.add(g.mul(z, z), log_term)), lx)
    if dist == "student_t":       # student_t.ex:15-29
        df, loc, scale = _need(p, dist, "df", "loc", "scale")
        ss = g.max(scale, tiny)
        sdf = g.max(df, tiny)
        z = g.div(g.sub(x, loc), ss)
        z2 = g.mul(z, z)
        hp1 = g.div(g.add(sdf, g.lit(1.0)), g.lit(2.0))
        h = g.div(sdf, g.lit(2.0))
        r = g.sub(_lgamma(g, hp1), _lgamma(g, h))
        r = g.sub(r, g.mul(g.lit(0.5), g.log(g.mul(sdf, g.lit(PI_F32)))))
        r = g.sub(r, g.log(ss))
        return g.sub(r, g.mul(hp1, g.log(g.add(g.lit(1.0), g.div(z2, sdf)))))
    if dist == "bernoulli":       # bernoulli.ex:17-27
        (pp,) = _need(p, dist, "p")
        pc = g.min(g.max(pp, g.lit(BERN_LO)), g.lit(BERN_HI))
        return g.add(g.mul(x, g.log(pc)),
                     g.mul(g.sub(g.lit(1.0), x), g.log(g.sub(g.lit(1.0), pc))))
    if dist == "gamma":           # gamma.ex:15-27 (shape alpha, rate beta)
        alpha, beta = _need(p, dist, "alpha", "beta")
        a = g.add(g.mul(g.sub(alpha, g.lit(1.0)), g.log(x)), g.mul(alpha, g.log(beta)))
        return g.sub(a, g.add(g.mul(beta, x), _lgamma(g, alpha)))
    if dist == "beta":            # beta.ex:15-24, math.ex:57-61
        alpha, beta = _need(p, dist, "alpha", "beta")
        a = g.add(g.mul(g.sub(alpha, g.lit(1.0)), g.log(x)),
                  g.mul(g.sub(beta, g.lit(1.0)), g.log(g.sub(g.lit(1.0), x))))
        lbeta = g.sub(g.add(_lgamma(g, alpha), _lgamma(g, beta)), _lgamma(g, g.add(alpha, beta)))
        return g.add(a, g.neg(lbeta))
    if dist == "weibull":         # weibull.ex:17-27 (shape k, scale lambda)
        k, lam = _need(p, dist, "k", "lambda")
        log_lam, log_k = g.log(lam), g.log(k)
        z = g.sub(g.log(x), log_lam)
        a = g.add(g.sub(log_k, log_lam), g.mul(g.sub(k, g.lit(1.0)), z))
        return g.sub(a, g.exp(g.mul(k, z)))
    if dist == "poisson":         # poisson.ex:16-20
        (mu,) = _need(p, dist, "mu")
        return g.sub(g.sub(g.mul(x, g.log(mu)), mu), _lgamma(g, g.add(x, g.lit(1.0))))
    if dist == "truncated_normal":   # truncated_normal.ex:16-40 (f32 literals like normal.ex; Nx.erf = exmc_erf)
        _need(p, dist, "mu", "sigma", "lower", "upper")
        ss = g.max(p["sigma"], tiny)
        z = g.div(g.sub(x, p["mu"]), ss)
        log_term = g.add(g.lit(LOG_2PI_F32), g.mul(g.lit(2.0), g.log(ss)))
        normal = g.mul(g.lit(-0.5), g.add(g.mul(z, z), log_term))

        def cdf(v):
            return g.mul(g.lit(0.5), g.add(g.lit(1.0), g.erf(g.div(v, g.lit(_f32(math.sqrt(2.0)))))))
        alpha = g.div(g.sub(p["lower"], p["mu"]), ss)
        beta = g.div(g.sub(p["upper"], p["mu"]), ss)
        return g.sub(normal, g.log(g.sub(cdf(beta), cdf(alpha))))
    if dist == "mixture":         # mixture.ex:13-27
        comps, cps, ws = _need(p, dist, "components", "params", "weights")
        if not (isinstance(ws, list) and len(comps) == len(cps) == len(ws) and len(ws) >= 1):
            raise CodegenError("mixture needs components, params and weights of equal length")
        lps = [g.add(_logpdf(g, comps[k], x, cps[k]), g.log(ws[k])) for k in range(len(ws))]
        return _logsumexp(g, lps)
    if dist == "uniform01":       # uniform01.ex:14-16
        _need(p, dist)
        return g.lit(0.0)
    raise CodegenError("distribution %r is not covered" % (dist,))


VECTOR_DISTS = ("gaussian_random_walk", "mv_normal", "dirichlet")


def _sigmoid(g, z):
    # transform.ex:274-276
    return g.exp(g.neg(_softplus(g, g.neg(z))))


def _stick_breaking_forward(g, zs):
    # transform.ex:125-143: y_i = sigmoid(z_i), x_i = y_i * remaining, remaining *= 1 - y_i
    xs, rem = [], g.lit(1.0)
    for z in zs:
        y = _sigmoid(g, z)
        xs.append(g.mul(y, rem))
        rem = g.mul(rem, g.sub(g.lit(1.0), y))
    return xs + [rem]


def _stick_breaking_ladj(g, zs):
    # transform.ex:186-203
    lj, rem = g.lit(0.0), g.lit(1.0)
    for z in zs:
        y = _sigmoid(g, z)
        log_dy = g.add(g.neg(_softplus(g, g.neg(z))), g.neg(_softplus(g, z)))
        lj = g.add(lj, g.add(g.log(rem), log_dy))
        rem = g.mul(rem, g.sub(g.lit(1.0), y))
    return lj
LOG_2PI_OF_F64_F32 = _f32(math.log(2.0 * math.pi))   # Nx.tensor(:math.log(2 pi)), mv_normal.ex:27


def _vector_length(id_, n):
    p = n["params"]
    if n["dist"] == "gaussian_random_walk":
        if "steps" not in p or int(p["steps"]) < 1:
            raise CodegenError("gaussian_random_walk %r needs steps >= 1" % id_)
        return int(p["steps"])
    if n["dist"] == "dirichlet":
        # Transform.unconstrained_length(:stick_breaking, {K}) = K - 1 (transform.ex:80-83)
        alpha = np.asarray(p.get("alpha"), dtype=np.float64)
        if alpha.ndim != 1 or alpha.shape[0] < 2:
            raise CodegenError("dirichlet %r needs a constant concentration vector of length >= 2" % id_)
        if n["transform"] != "stick_breaking":
            raise CodegenError("a free dirichlet %r lives on the simplex: transform must be 'stick_breaking'" % id_)
        return int(alpha.shape[0]) - 1
    mu = np.asarray(p.get("mu"), dtype=np.float64)
    if mu.ndim != 1:
        raise CodegenError("mv_normal %r needs a vector mu" % id_)
    return int(mu.shape[0])


def _sum_left(g, xs):
    acc = xs[0]                                   # Nx.sum on the BinaryBackend: left to right
    for e in xs[1:]:
        acc = g.add(acc, e)
    if len(xs) > 1:
        g.sums[acc] = list(xs)                    # the lane layout spreads the summands over lanes
    return acc


def _logpdf_vector(g, dist, xs, p, resolve_value):
    if dist == "gaussian_random_walk":            # gaussian_random_walk.ex:21-57
        sigma = resolve_value(p["sigma"])
        if isinstance(sigma, list):
            raise CodegenError("gaussian_random_walk takes a scalar sigma")
        ss = g.max(sigma, g.lit(TINY_F32))
        log_sigma = g.log(ss)
        norm = g.add(g.lit(LOG_2PI_F32), g.mul(g.lit(2.0), log_sigma))

        def step(d):
            z = g.div(d, ss)
            return g.mul(g.lit(-0.5), g.add(g.mul(z, z), norm))
        logp_init = step(xs[0])
        if len(xs) == 1:
            return logp_init
        steps = [step(g.sub(xs[t], xs[t - 1])) for t in range(1, len(xs))]
        return g.add(logp_init, _sum_left(g, steps))
    if dist == "mv_normal":                       # mv_normal.ex:20-47
        mu = np.asarray(p["mu"], dtype=np.float64)
        cov = np.asarray(p["cov"], dtype=np.float64)
        d = mu.shape[0]
        if cov.shape != (d, d) or len(xs) != d:
            raise CodegenError("mv_normal needs mu {d} and cov {d, d}")
        # prepare_params (eager, host): Nx.LinAlg.cholesky / invert are third-party LinAlg, taken
        # from numpy here (a rounding-level difference in constants of the data)
        chol = np.linalg.cholesky(cov)
        log_det = 2.0 * float(np.sum(np.log(np.diag(chol))))
        prec = np.linalg.inv(cov)
        diff = [g.sub(xs[i], g.datum(mu[i])) for i in range(d)]
        inner = [_sum_left(g, [g.mul(g.datum(prec[i, j]), diff[j]) for j in range(d)]) for i in range(d)]
        mahal = _sum_left(g, [g.mul(diff[i], inner[i]) for i in range(d)])
        base = g.add(g.mul(g.lit(_f32(d * 1.0)), g.lit(LOG_2PI_OF_F64_F32)), g.datum(log_det))
        return g.mul(g.lit(-0.5), g.add(base, mahal))
    if dist == "dirichlet":                       # dirichlet.ex:19-27; xs on the simplex
        alpha = np.asarray(p["alpha"], dtype=np.float64)
        if alpha.ndim != 1 or len(xs) != alpha.shape[0]:
            raise CodegenError("dirichlet needs alpha {K} and a value of K elements")
        al = [g.datum(float(a)) for a in alpha]
        kernel = _sum_left(g, [g.mul(g.sub(al[i], g.lit(1.0)), g.log(xs[i])) for i in range(len(xs))])
        log_norm = g.sub(_lgamma(g, _sum_left(g, al)), _sum_left(g, [_lgamma(g, a) for a in al]))
        return g.add(kernel, log_norm)
    raise CodegenError("vector distribution %r is not covered" % (dist,))


class Ops:
    """The declarative op set a Custom distribution's closure may use (dist/custom.ex: the
    reference's closure composes Nx ops; here the same composition builds expression-graph nodes,
    so value and gradient are generated like any other term). Scalars are opaque handles; vectors
    are python lists of them."""

    def __init__(self, g):
        self._g = g

    def lit(self, x): return self._g.lit(float(x))
    def f32(self, x): return self._g.lit(_f32(x))          # Nx.tensor(<float>) defaults to f32
    def data(self, x): return self._g.datum(float(x))      # a datum captured by the closure
    def sigmoid(self, a):                                  # Nx.sigmoid: 1 / (1 + exp(-x))
        g = self._g
        return g.div(g.lit(1.0), g.add(g.lit(1.0), g.exp(g.neg(a))))
    def add(self, a, b): return self._g.add(a, b)
    def sub(self, a, b): return self._g.sub(a, b)
    def mul(self, a, b): return self._g.mul(a, b)
    def div(self, a, b): return self._g.div(a, b)
    def neg(self, a): return self._g.neg(a)
    def exp(self, a): return self._g.exp(a)
    def log(self, a): return self._g.log(a)
    def log1p(self, a): return self._g.log1p(a)
    def abs(self, a): return self._g.abs(a)
    def max(self, a, b): return self._g.max(a, b)
    def min(self, a, b): return self._g.min(a, b)
    def sum(self, xs): return _sum_left(self._g, list(xs))   # Nx.sum: left to right

    def logpdf(self, dist, x, params):
        """Exmc.Dist.<dist>.logpdf(x, params) called from inside a closure (the reference's closures
        call the distribution modules the same way, e.g. benchmark/reliability_model.exs)."""
        return _logpdf(self._g, dist, x, dict(params))

    def dot(self, xs, ys):
        """Nx.dot of two vectors on the BinaryBackend: products summed left to right."""
        return _sum_left(self._g, [self._g.mul(a, b) for a, b in zip(xs, ys)])


# ---------------------------------------------------------------------------------------------
# the term walk
# ---------------------------------------------------------------------------------------------
class Generated:
    """d, var_names (flat order, point_map.ex:37), transforms, ncp_info, data (what
    exmc_hip_model_create receives), header (the generated source) and its digest."""


def _observed_targets(ir):
    return {n["target"] for n in ir.nodes.values() if n["op"] in ("obs", "meas_obs")}


# Dist.transform/1 of each module under lib/exmc/dist (the default a 3-tuple rv node receives)
DEFAULT_TRANSFORMS = dict(bernoulli="logit", beta="logit", uniform01="logit", dirichlet="stick_breaking",
                          exponential="log", gamma="log", half_cauchy="log", lognormal="log", poisson="log",
                          weibull="log", half_normal="softplus")


def _default_transform(dist, params):
    if dist == "mixture":                         # mixture.ex:34-36: the first component's
        comps, cps = params.get("components") or [None], params.get("params") or [{}]
        return _default_transform(comps[0], cps[0])
    if dist == "custom":                          # custom.ex:92-95: the closure bundle's own field
        return params.get("transform")
    return DEFAULT_TRANSFORMS.get(dist)


def rewrite(ir):
    """Exmc.Rewrite.apply (rewrite.ex:13-34) up to the non-centred pass, which `generate` applies
    itself: AttachDefaultTransforms (an rv written without a transform gets its distribution's
    default -- observed rvs too, which moves their obs term to the unconstrained value of the
    datum, compiler.ex:284-291), LiftMeasurableMatmul, LiftMeasurableAffine (obs of a det node ->
    meas_obs, the obs meta carried over). NormalizeObs / PopulateObsMetadata only fill defaults
    this IR already has (a weight of 1.0 multiplies exactly)."""
    out = IR()
    for id_, n in ir.nodes.items():
        n = dict(n)
        if n["op"] == "rv" and n["transform"] is None:
            n["transform"] = _default_transform(n["dist"], n["params"])
        elif n["op"] == "obs":
            tgt = ir.nodes.get(n["target"])
            if tgt is not None and tgt["op"] == "det":
                fun, args = tgt["fun"], tgt["args"]
                lifted = None
                if fun == "matmul" and len(args) == 2 and isinstance(args[1], str):
                    lifted = (args[1], ("matmul", args[0]))
                elif fun == "affine" and len(args) == 3 and isinstance(args[2], str):
                    lifted = (args[2], ("affine", args[0], args[1]))
                if lifted is not None:
                    if (n.get("meta") or {}).get("censored"):
                        raise CodegenError("obs %r: a censored observation of a det node is not covered" % id_)
                    tmp = IR().meas_obs(id_, lifted[0], n["value"], lifted[1], meta=n.get("meta"))
                    n = tmp.nodes[id_]
        out.nodes[id_] = n
    out.data_tensor = ir.data_tensor
    return out


def _apply_ncp(ir, ncp):
    """rewrite/non_centered_parameterization.ex:26-55"""
    nodes, info = dict(ir.nodes), {}
    if not ncp:
        return nodes, info
    observed = _observed_targets(ir)
    for id_, n in ir.nodes.items():
        if (n["op"] == "rv" and n["dist"] == "normal" and n["transform"] is None
                and id_ not in observed and isinstance(n["params"].get("mu"), str)
                and isinstance(n["params"].get("sigma"), str)):
            info[id_] = dict(mu=n["params"]["mu"], sigma=n["params"]["sigma"])
            nodes[id_] = dict(op="rv", dist="normal", params=dict(mu=0.0, sigma=1.0), transform=None)
    return nodes, info


def generate(ir, ncp=True, vectorize=True, rewrite_passes=False, lanes=None, waves_per_simd=1):
    """Compiler.compile_for_sampling (compiler.ex:46-58) as source text. `rewrite_passes` runs the
    reference's IR passes first (`rewrite`); without it the IR is taken as already rewritten
    (transforms explicit), which is what an exporter on the Elixir side sends. `lanes` = 16 / 32 /
    64 asks for the lane layout of codegen_lanes.py (a chain over that many lanes, several
    dimensions per lane); models above MAX_D free dimensions get it by themselves.

    Term order: Map.values of the node map (compiler.ex:176-180) = ids sorted as strings for up to
    MAX_NODES_SORTED nodes. A larger Erlang map iterates in the order of its internal hash, which
    is not restated here: an IR exported from a BEAM carries that order (`ir.term_order`, written by
    HipExport.to_json) and the terms follow it; an IR built in Python has no VM to ask and takes
    sorted-id order there too -- a reordering of the final sum (a few ulp of the log-density; the
    gradient's entries are sums over the same terms)."""
    if rewrite_passes:
        ir = rewrite(ir)
    nodes, ncp_info = _apply_ncp(ir, ncp)
    for id_, n in nodes.items():
        if n["op"] == "obs" and n["target"] not in nodes:
            raise CodegenError("obs %r targets unknown node %r" % (id_, n["target"]))
    observed = _observed_targets(ir)
    free = sorted(i for i, n in nodes.items() if n["op"] == "rv" and i not in observed)
    if not free:
        raise CodegenError("no free random variables")
    # PointMap.build (point_map.ex:30-60): entries sorted by id, each `length` flat slots
    offset, length, flat_names, vector_entries = {}, {}, [], {}
    for id_ in free:
        n = nodes[id_]
        ln = _vector_length(id_, n) if n["dist"] in VECTOR_DISTS else 1
        if n["dist"] in VECTOR_DISTS and n["transform"] is not None and n["dist"] != "dirichlet":
            raise CodegenError("a transformed vector rv is not covered")
        offset[id_], length[id_] = len(flat_names), ln
        if n["dist"] in VECTOR_DISTS:
            vector_entries[id_] = (len(flat_names), ln)
            flat_names.extend("%s[%d]" % (id_, i) for i in range(ln))
        else:
            flat_names.append(id_)
    if len(flat_names) > MAX_D_LANES:
        raise CodegenError("%d free dimensions; the kernels take at most %d" % (len(flat_names), MAX_D_LANES))
    if lanes is None and len(flat_names) > MAX_D:
        lanes = 64 if len(flat_names) > 32 else 16
    if lanes is not None and lanes not in (16, 32, 64):
        raise CodegenError("lanes must be 16, 32 or 64")
    one_lane = len(flat_names) <= MAX_D
    g = _Graph()
    custom_roots = set()      # terms that are the result of a Custom closure (a hand-written reduction)

    def resolve_ref(id_, stack=()):
        # compiler.ex:447-463
        if id_ not in offset:
            raise CodegenError("param ref %r is not a free random variable" % id_)
        if id_ in stack:
            raise CodegenError("cyclic non-centred reference through %r" % id_)
        if id_ in vector_entries:
            zs = [g.q(offset[id_] + i) for i in range(length[id_])]
            return _stick_breaking_forward(g, zs) if nodes[id_]["dist"] == "dirichlet" else zs
        z = g.q(offset[id_])
        if id_ in ncp_info:
            mu = resolve_value(ncp_info[id_]["mu"], stack + (id_,))
            sigma = resolve_value(ncp_info[id_]["sigma"], stack + (id_,))
            return g.add(mu, g.mul(sigma, z))
        return _apply_transform(g, nodes[id_]["transform"], z)

    def resolve_value(v, stack=()):
        if isinstance(v, str) and v == "__obs_data":       # compiler.ex:114-118: the IR's data tensor
            t = ir.data_tensor
            if t is None:
                raise CodegenError('"__obs_data" is referenced but the IR has no data (Builder.data)')
            if t.ndim == 2:
                return [[g.datum(float(x)) for x in row] for row in t]
            return g.datum(float(t)) if t.ndim == 0 else [g.datum(float(x)) for x in t]
        if isinstance(v, str):
            return resolve_ref(v, stack)
        if isinstance(v, F32):
            return g.datum32(v)
        a = np.asarray(v, dtype=np.float64)
        if a.ndim == 0:
            return g.datum(float(a))
        if a.ndim == 1:
            return [g.datum(float(x)) for x in a]
        raise CodegenError("params are scalars, vectors or refs")

    def resolve_params(dist, params):
        if dist == "mixture":     # nested: one params map per component, a weight vector
            cps = [resolve_params(c, pp) for c, pp in zip(params.get("components", []), params.get("params", []))]
            ws = resolve_value(params.get("weights"))
            return dict(components=list(params.get("components", [])), params=cps,
                        weights=ws if isinstance(ws, list) else [ws])
        return {k: resolve_value(v) for k, v in params.items()}

    def elementwise(dist, x, params):
        if dist == "mixture":     # scalar params per component; only the value may be a vector
            if isinstance(x, list):
                return [_logpdf(g, dist, xi, params) for xi in x], True
            return _logpdf(g, dist, x, params), False
        n = max([len(v) for v in [x] + list(params.values()) if isinstance(v, list)] + [0])
        if n == 0:
            return _logpdf(g, dist, x, params), False
        for v in [x] + list(params.values()):
            if isinstance(v, list) and len(v) != n:
                raise CodegenError("vector lengths differ")
        pick = lambda v, i: v[i] if isinstance(v, list) else v   # noqa: E731
        return [_logpdf(g, dist, pick(x, i), {k: pick(v, i) for k, v in params.items()})
                for i in range(n)], True

    ops = Ops(g)

    def custom_logpdf(params, x):
        fn = params.get("logpdf")
        if not callable(fn):
            raise CodegenError("a custom distribution needs a callable 'logpdf'")
        rest = {k: resolve_value(v) for k, v in params.items() if k != "logpdf"}
        t = fn(ops, x, rest)
        if isinstance(t, list):
            raise CodegenError("a custom logpdf must return a scalar (reduce inside the closure)")
        custom_roots.add(t)
        return t

    def const_x(tr, v, id_):
        # compiler.ex:284-291, 327-334, 350-357: z = inverse_transform(value), x = Transform.apply(z),
        # the log-Jacobian at z joins the term (all constants of the data). v: a float or a node.
        d = v if isinstance(v, int) else g.datum(float(v))
        if tr is None:
            return d, None
        if tr == "log":
            z = g.log(d)
        elif tr == "softplus":
            if isinstance(v, int):
                raise CodegenError("%r: a softplus-transformed target of a computed value is not covered" % id_)
            z = g.datum(math.log(math.expm1(float(v))))      # Nx.log(Nx.expm1(x)), host libm
        elif tr == "logit":
            z = g.sub(g.log(d), g.log1p(g.neg(d)))
        else:
            raise CodegenError("%r: an observation of a %r-transformed rv is not covered" % (id_, tr))
        return _apply_transform(g, tr, z), _log_abs_det_jacobian(g, tr, z)

    def apply_obs_meta(elems, is_vec, meta, id_):
        # compiler.ex:401-418: weight, mask, reduce
        w = meta.get("weight")
        if w is not None:
            if is_vec:
                elems = [g.mul(e, g.datum(float(w if w.ndim == 0 else w[i]))) for i, e in enumerate(elems)]
            else:
                if w.ndim != 0:
                    raise CodegenError("obs %r: a vector weight on a scalar term" % id_)
                elems = g.mul(elems, g.datum(float(w)))
        mk = meta.get("mask")
        if mk is not None:
            mk = np.asarray(mk)
            if not is_vec:                        # Nx.select(mask, logp, 0.0) with a scalar mask
                if mk.ndim != 0:
                    raise CodegenError("obs %r: a vector mask on a scalar term" % id_)
                elems = elems if bool(mk) else g.lit(0.0)
            else:
                if mk.ndim != 1 or len(mk) != len(elems):
                    raise CodegenError("obs %r: mask does not match the term" % id_)
                elems = [e if bool(mk[i]) else g.lit(0.0) for i, e in enumerate(elems)]
        red = meta.get("reduce")
        if not is_vec:
            return elems
        if red == "sum":
            return _sum_left(g, elems)            # Nx.sum on the BinaryBackend: left to right
        if red == "mean":
            return g.div(_sum_left(g, elems), g.lit(float(len(elems))))
        if red == "logsumexp":
            return _logsumexp(g, elems)
        raise CodegenError("obs %r: a vector-valued term needs a reduce" % id_)

    terms = []
    term_order = getattr(ir, "term_order", None)
    if term_order is not None and (sorted(term_order) != sorted(nodes) or len(term_order) != len(nodes)):
        raise CodegenError("term_order does not match the node map")   # (a rewrite pass changed the key set)
    if term_order is not None and len(nodes) <= MAX_NODES_SORTED and list(term_order) != sorted(nodes):
        raise CodegenError("term_order of a map of <= %d keys must be the sorted ids (Erlang flatmaps "
                           "are sorted)" % MAX_NODES_SORTED)
    for id_ in (term_order if term_order is not None else sorted(nodes)):   # Map.values order, compiler.ex:176-180
        n = nodes[id_]
        if n["op"] == "det":                      # compiler.ex:268-269
            continue
        if n["op"] in ("obs", "meas_obs") and (n.get("meta") or {}).get("likelihood") is False:
            continue                              # compiler.ex:244-245
        if n["op"] == "rv":
            if id_ not in offset:
                continue
            if n["dist"] in VECTOR_DISTS:
                xs = [g.q(offset[id_] + i) for i in range(length[id_])]
                if n["dist"] == "dirichlet":      # logpdf on the simplex + log|J| (compiler.ex:222-229)
                    t = _logpdf_vector(g, "dirichlet", _stick_breaking_forward(g, xs), n["params"], resolve_value)
                    terms.append(g.add(t, _stick_breaking_ladj(g, xs)))
                    continue
                terms.append(_logpdf_vector(g, n["dist"], xs, n["params"], resolve_value))
                continue
            z = g.q(offset[id_])
            x = _apply_transform(g, n["transform"], z)
            if n["dist"] == "custom":
                t = custom_logpdf(n["params"], x)
            else:
                params = resolve_params(n["dist"], n["params"])
                if n["dist"] != "mixture" and any(isinstance(v, list) for v in params.values()):
                    raise CodegenError("free RV %r has a vector param" % id_)
                t = _logpdf(g, n["dist"], x, params)
            if n["transform"] is not None:
                t = g.add(t, _log_abs_det_jacobian(g, n["transform"], z))   # compiler.ex:222-229
            terms.append(t)
        elif n["op"] == "meas_obs":
            # compiler.ex:258-266, 342-369: eager -- the target's params are used as written (no
            # refs), so the whole term is a constant of the data
            tgt = nodes[n["target"]]
            if tgt["op"] != "rv" or tgt["dist"] in VECTOR_DISTS + ("custom",):
                raise CodegenError("meas_obs %r: target not covered" % id_)
            if any(isinstance(v, str) for v in tgt["params"].values()):
                raise CodegenError("meas_obs %r: the reference evaluates it eagerly, params must be constants" % id_)
            params = resolve_params(tgt["dist"], tgt["params"])
            val, info, tr = n["value"], n["info"], tgt["transform"]
            if info[0] == "affine":
                av, bv = np.asarray(info[1], dtype=np.float64), np.asarray(info[2], dtype=np.float64)
                at = lambda c, i: g.datum(float(c if c.ndim == 0 else c[i]))   # noqa: E731
                conv = lambda v, i: g.div(g.sub(g.datum(float(v)), at(bv, i)), at(av, i))   # noqa: E731
                x = conv(val, 0) if val.ndim == 0 else [conv(v, i) for i, v in enumerate(val)]
                if av.ndim == 0:
                    jac = g.neg(g.log(g.abs(g.datum(float(av)))))
                else:                             # -log|a| element by element: added to the vector term below
                    jac = [g.neg(g.log(g.abs(g.datum(float(c))))) for c in av]
            else:
                sol = np.linalg.solve(info[1], val)                  # jit_solve: third-party LinAlg
                x = [g.datum(float(v)) for v in sol]
                jac = g.datum(-math.log(abs(float(np.linalg.det(info[1])))))
            tjac = None
            if tr is not None:                    # compiler.ex:350-359, 371-382: (logp + jac) + meas_jac
                pairs = [const_x(tr, v, id_) for v in (x if isinstance(x, list) else [x])]
                x = [p_[0] for p_ in pairs] if isinstance(x, list) else pairs[0][0]
                tjac = [p_[1] for p_ in pairs]
            t, vec = elementwise(tgt["dist"], x, params)
            if vec:
                # logpdf of a vector value is a vector; combined = logp + jac broadcasts, the meta's
                # reduce (or, without one, sum_logps' Nx.sum, compiler.ex:396-397) folds it
                if tjac is not None:
                    t = [g.add(e, tjac[i if len(tjac) > 1 else 0]) for i, e in enumerate(t)]
                t = [g.add(e, jac[i] if isinstance(jac, list) else jac) for i, e in enumerate(t)]
            else:
                if tjac is not None:
                    t = g.add(t, tjac[0])
                t = g.add(t, jac)
            meta = dict(n.get("meta") or {})
            if vec and meta.get("reduce") is None:
                meta["reduce"] = "sum"
            terms.append(apply_obs_meta(t, vec, meta, id_))
        else:
            tgt = nodes[n["target"]]
            if tgt["op"] != "rv":
                # the reference gives such a node no term (compiler.ex:293); a det target that the
                # lifting passes did not take is a model error here
                raise CodegenError("obs %r does not target an rv (run the rewrite passes for det targets)" % id_)
            val, meta = n["value"], n.get("meta") or dict(reduce="sum" if n["value"].ndim else None, censored=None)
            cens = meta.get("censored")
            tr = tgt["transform"]
            if tr is not None and (cens == "interval" or tgt["dist"] in ("custom", "mv_normal", "gaussian_random_walk")):
                raise CodegenError("obs %r: a transformed target with this distribution / censoring is not covered" % id_)
            if tr is not None and cens:
                # compiler.ex:274, 298-311 match the 3-tuple rv node only: on a target that carries a
                # transform the 4-tuple clauses (:285, :325) run and the censoring is not applied
                cens = None

            vec = val.ndim > (1 if cens == "interval" else 0)
            if tgt["dist"] == "custom":
                x = [g.datum(float(v)) for v in val] if vec else g.datum(float(val))
                t = custom_logpdf(tgt["params"], x)
                elems, is_vec = t, False
            elif tgt["dist"] in VECTOR_DISTS:
                if val.ndim != 1 or cens:
                    raise CodegenError("obs %r of a vector distribution needs a plain vector value" % id_)
                xs, jac = [g.datum(float(v)) for v in val], None
                if tr == "stick_breaking":        # compiler.ex:424: z from the datum, the point rebuilt from z
                    zs = [g.datum(float(v)) for v in inverse_stick_breaking(val)]
                    xs, jac = _stick_breaking_forward(g, zs), _stick_breaking_ladj(g, zs)
                elif tr is not None:
                    raise CodegenError("obs %r: transform %r on a vector distribution is not covered" % (id_, tr))
                elems = _logpdf_vector(g, tgt["dist"], xs, tgt["params"], resolve_value)
                if jac is not None:
                    elems = g.add(elems, jac)
                is_vec = False
            elif cens:
                params = resolve_params(tgt["dist"], tgt["params"])
                if any(isinstance(v, list) for v in params.values()):
                    raise CodegenError("censored obs %r: vector params are not covered" % id_)
                rows = val if vec else val[None, ...]
                out = []
                for r in rows:
                    value = (g.datum(float(r[0])), g.datum(float(r[1]))) if cens == "interval" else g.datum(float(r))
                    out.append(_censored_loglik(g, cens, value, tgt["dist"], params))
                elems, is_vec = (out, True) if vec else (out[0], False)
            else:
                params = resolve_params(tgt["dist"], tgt["params"])
                if vec:
                    xs, jacs = zip(*[const_x(tr, v, id_) for v in val])
                    elems, is_vec = elementwise(tgt["dist"], list(xs), params)
                    if tr is not None:
                        elems = [g.add(e, j) for e, j in zip(elems, jacs)]
                else:
                    x, jac = const_x(tr, val, id_)
                    elems, is_vec = elementwise(tgt["dist"], x, params)
                    if tr is not None:
                        elems = [g.add(e, jac) for e in elems] if is_vec else g.add(elems, jac)
            t = apply_obs_meta(elems, is_vec, meta, id_)
            terms.append(t)
    total = terms[0]                              # sum_logps, compiler.ex:394-395
    for t in terms[1:]:
        total = g.add(t, total)
    if g.const[total]:
        raise CodegenError("the log-density does not depend on the free variables")

    out = Generated()
    out.d = len(flat_names)
    out.var_names = flat_names
    out.vector_entries = vector_entries
    # simplex entries: K - 1 flat slots each, a K-vector in the trace
    out.simplex_entries = {i: vector_entries[i] for i in vector_entries if nodes[i]["dist"] == "dirichlet"}
    out.transforms = {i: nodes[i]["transform"] for i in free if nodes[i]["transform"]}
    out.ncp_info = ncp_info
    out.lanes = 1
    out.vec = out.lane_layout = None
    if one_lane:
        n_fwd = len(g.ops)
        ad = _Grad(g, total)
        ad.run(n_fwd)
        grads = [ad.adj.get(g.key.get(("q", k))) for k in range(len(flat_names))]
        out.data = np.asarray(g.data, dtype=np.float64)
        out.header = _emit(g, total, grads, out.d)
    else:
        out.data = np.zeros(0)
        out.header = _emit_no_one_lane(out.d)
    # the 16-lane layout, when the model has plates to spread over lanes (codegen_vec.py): the
    # plug-in then carries Custom<16> next to Custom<1> and defaults to it
    from . import codegen_vec
    def _plain_node(n):
        if n["op"] == "rv":
            return n["dist"] not in VECTOR_DISTS + ("custom", "mixture")
        if n["op"] == "obs":
            m = n.get("meta") or {}
            return (m.get("reduce") in (None, "sum") and m.get("censored") is None and m.get("weight") is None
                    and m.get("mask") is None and nodes[n["target"]].get("transform") is None)
        return False
    plain = all(_plain_node(n) for n in nodes.values())
    if lanes is None and len(nodes) <= MAX_NODES_SORTED:
        out.vec = codegen_vec.generate(ir, ncp=ncp) if (vectorize and plain) else None
    if out.vec is not None:
        # (the plate layout's lane function is a section its users may include again, codegen_vec._emit)
        out.header = "#ifndef EXMC_GEN_VEC_SECTION\n" + out.header + "\n" + out.vec["text"]
        out.data = np.concatenate([out.data, out.vec["vdata"]])
        out.lanes = codegen_vec.G
    if lanes is not None:
        # several dimensions per lane (codegen_lanes.py): any d, the repeated terms over the lanes
        from . import codegen_lanes
        out.lane_layout = codegen_lanes.generate(g, terms, custom_roots, out.d, lanes, waves_per_simd)
        # (the lane function is a section of its own that its users include once per table placement)
        out.header = "#ifndef EXMC_GEN_LANES_SECTION\n" + out.header + \
                     "\n#define EXMC_GEN_LOFF %d   /* where the lane layout's table starts in data */\n" \
                     % out.data.size + out.lane_layout["text"]
        out.data = np.concatenate([out.data, out.lane_layout["data"]])
        out.lanes = lanes
    out.digest = hashlib.sha256(out.header.encode()).hexdigest()[:16]
    out.n_ops = out.header.count("\n")
    return out


# ---------------------------------------------------------------------------------------------
# emission
# ---------------------------------------------------------------------------------------------
_BIN = {"add": "+", "sub": "-", "mul": "*", "div": "/"}
_FN1 = {"exp": "EXMC_GEN_EXP", "log": "EXMC_GEN_LOG", "log1p": "EXMC_GEN_LOG1P", "erf": "EXMC_GEN_ERF",
        "abs": "fabs"}
_FN2 = {"max": "fmax", "min": "fmin"}


def _lds_levels(d):
    per_level = (5 * d + 3) * 64 * 8
    return max(1, min(6, (56 * 1024) // per_level))


def _emit_no_one_lane(d):
    """Header of a model that exists in the lane layout only (d > MAX_D)."""
    L = ["/* generated by exmc_amd/codegen.py -- do not edit. Included twice over: by",
         " * exmc_amd/csrc/exmc_models.hpp (device functor, -DEXMC_CUSTOM_HEADER) and by the host",
         " * checker tests build from the same text. */",
         "#define EXMC_GEN_D %d" % d, "#define EXMC_GEN_NDATA 0", "#define EXMC_GEN_NCONST 1",
         "#define EXMC_GEN_LDS_LEVELS 1", ""]
    return "\n".join(L) + "\n"


def _emit(g, total, grads, d):
    outputs = [total] + [x for x in grads if x is not None]
    # liveness: everything reachable from the outputs
    live = set()
    stack = list(outputs)
    while stack:
        i = stack.pop()
        if i in live:
            continue
        live.add(i)
        op = g.ops[i]
        if op[0] not in ("lit", "data", "q"):
            stack.extend(op[1:])
    # folded constants: const non-literal nodes read by a dynamic node (or being an output)
    slot = {}
    for a in outputs:
        if g.const[a] and g.ops[a][0] != "lit" and a not in slot:
            slot[a] = len(slot)
    for i in sorted(live):
        if g.const[i]:
            continue
        for a in (g.ops[i][1:] if g.ops[i][0] != "q" else ()):
            if g.const[a] and g.ops[a][0] != "lit" and a not in slot:
                slot[a] = len(slot)
    # host-side nodes needed for the slots
    host = set()
    stack = list(slot)
    while stack:
        i = stack.pop()
        if i in host:
            continue
        host.add(i)
        op = g.ops[i]
        if op[0] not in ("lit", "data"):
            stack.extend(op[1:])

    def ref(i, dyn):
        op = g.ops[i]
        if op[0] == "lit":
            s = repr(float.fromhex(op[1]))
            if "inf" in s or "nan" in s:
                raise CodegenError("non-finite literal")
            return "(%s)" % s if s.startswith("-") else s
        if dyn and i in slot:
            return "c[%d]" % slot[i]
        if op[0] == "data":
            return "data[%d]" % op[1]
        if op[0] == "q":
            return "q[%d]" % op[1]
        return "t%d" % i

    def stmt(i, dyn):
        op = g.ops[i]
        a = [ref(x, dyn) for x in op[1:]] if op[0] not in ("lit", "data", "q") else []
        if op[0] in _BIN:
            e = "%s %s %s" % (a[0], _BIN[op[0]], a[1])
        elif op[0] == "neg":
            e = "-%s" % a[0]
        elif op[0] in _FN1:
            e = "%s(%s)" % (_FN1[op[0]], a[0])
        elif op[0] in _FN2:
            e = "%s(%s, %s)" % (_FN2[op[0]], a[0], a[1])
        elif op[0] == "sel_gt":
            e = "(%s > %s) ? %s : %s" % tuple(a)
        else:
            raise CodegenError("cannot emit %s" % op[0])
        return "  const double t%d = %s;" % (i, e)

    L = []
    L.append("/* generated by exmc_amd/codegen.py -- do not edit. Included twice over: by")
    L.append(" * exmc_amd/csrc/exmc_models.hpp (device functor Custom<1>, -DEXMC_CUSTOM_HEADER) and by")
    L.append(" * the host checker tests build from the same text. */")
    L.append("#define EXMC_GEN_D %d" % d)
    L.append("#define EXMC_GEN_ONE_LANE 1")
    L.append("#define EXMC_GEN_NDATA %d" % len(g.data))
    L.append("#define EXMC_GEN_NCONST %d" % max(1, len(slot)))
    L.append("#define EXMC_GEN_LDS_LEVELS %d" % _lds_levels(d))
    L.append("")
    L.append("EXMC_GEN_HOST void exmc_gen_fold(const double* data, double* c) {")
    for i in sorted(host):
        if g.ops[i][0] not in ("lit", "data"):
            L.append(stmt(i, False))
    for i, k in sorted(slot.items(), key=lambda kv: kv[1]):
        L.append("  c[%d] = %s;" % (k, ref(i, False)))
    if not slot:
        L.append("  c[0] = 0.0;")
    L.append("  (void)data;")
    L.append("}")
    L.append("")
    L.append("EXMC_GEN_FN double exmc_gen_logp_grad(const double* c, const double* q, double* g) {")
    for i in sorted(live):
        if not g.const[i] and g.ops[i][0] != "q":
            L.append(stmt(i, True))
    for k, x in enumerate(grads):
        L.append("  g[%d] = %s;" % (k, "0.0" if x is None else ref(x, True)))
    L.append("  (void)c;")
    L.append("  return %s;" % ref(total, True))
    L.append("}")
    return "\n".join(L) + "\n"


# ---------------------------------------------------------------------------------------------
# plug-in build + ModelSpec
# ---------------------------------------------------------------------------------------------
GEN_DIR = os.path.join(_build.OUT_DIR, "gen")


def _extra_flags():
    return os.environ.get("EXMC_GEN_EXTRA_FLAGS", "").split()


def _build_form():
    """How build_plugin links the model-dependent kernels: device-only code objects launched through
    hipModuleLaunchKernel (the default), host stubs (EXMC_PLUGIN_STUBS=1), or one translation unit
    (EXMC_PLUGIN_ONE_TU=1). Part of the cache tag: a library of one form is never handed out for another."""
    if os.environ.get("EXMC_PLUGIN_ONE_TU") == "1":
        return "onetu"
    return "stubs" if os.environ.get("EXMC_PLUGIN_STUBS") == "1" else "modules"


def plugin_paths(gen):
    tag = gen.digest
    if _extra_flags():
        tag += "_" + hashlib.sha256(" ".join(_extra_flags()).encode()).hexdigest()[:8]
    if _build_form() != "modules":
        tag += "_" + _build_form()
    d = os.path.join(GEN_DIR, tag)
    return d, os.path.join(d, "exmc_gen_model.h"), os.path.join(d, "libexmc_hip_gen.so")


def build_plugin(gen, force=False, verbose=False):
    """hipcc the NUTS kernels around the generated functor (the analogue of the EXLA JIT step,
    jit.ex). Cached by the digest of the generated text and the kernel sources' mtimes."""
    d, hdr, so = plugin_paths(gen)
    os.makedirs(d, exist_ok=True)
    if not os.path.exists(hdr) or open(hdr).read() != gen.header:
        with open("%s.%d.tmp" % (hdr, os.getpid()), "w") as f:
            f.write(gen.header)
        os.replace(f.name, hdr)
    if not force and os.path.exists(so):
        t = os.path.getmtime(so)
        if all(os.path.getmtime(p) <= t for p in _build.DEPS + [hdr]):
            return so
    tmp = "%s.%d.tmp" % (so, os.getpid())       # two processes may build the same model: publish atomically
    hipcc = _build.hipcc()
    defs = ["-DEXMC_ONLY_CUSTOM", '-DEXMC_CUSTOM_HEADER="%s"' % hdr]
    flags = [f for f in _build.FLAGS if f != "-shared"] + _extra_flags()
    cwd = os.path.dirname(_build.SRC)
    objs = []
    try:
        if os.environ.get("EXMC_PLUGIN_ONE_TU") == "1":
            cmd = [hipcc] + _build.FLAGS + _extra_flags() + defs + ["-o", tmp, _build.SRC]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=cwd)
        else:
            # The model-dependent kernels (sampling, its stream form, the two warmup forms, the four
            # auxiliary kernels) are compiled DEVICE-ONLY, one code object per part, side by side; the
            # rest of the library (host code, the C ABI) is one more unit next to them, compiled with
            # -DEXMC_PLUGIN_MODULES: it launches those kernels by their device-side names through
            # hipModuleLaunchKernel from the code objects embedded in the library as data
            # (exmc_hip.hip "launching the model-dependent kernels"). The plug-in is ready in the time of
            # its slowest device pass plus a link -- no host pass, no host stubs for the parts (the
            # analogue of the EXLA JIT step, jit.ex; EXMC_PLUGIN_STUBS=1 builds the round-3 form, every
            # part a host + device unit launched through its stub).
            part_src = os.path.join(cwd, "exmc_plugin_part.hip")
            modules = os.environ.get("EXMC_PLUGIN_STUBS") != "1"
            host_flags = flags + ["-Xarch_host", "-O0"]   # launch code, nothing numeric
            common = _build.build_common()
            main_defs = ["-DEXMC_PLUGIN_SPLIT", "-DEXMC_COMMON_DECL_ONLY"] + (["-DEXMC_PLUGIN_MODULES"] if modules else [])
            jobs = [([hipcc] + host_flags + defs + main_defs + ["-c", "-o", "%s.main.o" % tmp, _build.SRC])]
            layouts = [k for k, macro in ((1, "EXMC_GEN_ONE_LANE"), (2, "EXMC_GEN_VEC"), (3, "EXMC_GEN_LANES "))
                       if ("#define " + macro) in gen.header]
            parts = [(k, lay) for k in (3, 4, 7, 1, 2, 5) for lay in layouts]
            if 3 in layouts and gen.lanes < 64:     # the one-chain warmup form of the lane layout (CustomSplit)
                parts.append((6, 3))
            if 3 in layouts and "#define EXMC_GEN_WG 1" in gen.header:   # the workgroup form of the sampling kernel
                parts.append((8, 3))
            part_flags = ([f for f in flags if f != "-fPIC"] + ["--cuda-device-only"]) if modules else host_flags
            ext = "hsaco" if modules else "o"
            jobs += [([hipcc] + part_flags + defs + ["-DEXMC_PLUGIN_PART=%d" % k, "-DEXMC_PLUGIN_LAYOUT=%d" % lay, "-c", "-o",
                                                    "%s.p%d_%d.%s" % (tmp, k, lay, ext), part_src]) for k, lay in parts]
            objs = [j[j.index("-o") + 1] for j in jobs]
            if verbose:
                for j in jobs:
                    print(" ".join(j))
            t_start = time.time()
            part_objs = objs[1:]
            if modules:
                blob_s, blob_o = "%s.blobs.S" % tmp, "%s.blobs.o" % tmp
                objs = objs + [blob_s, blob_o]      # on the clean-up list before anything can fail
            procs = [subprocess.Popen(j, cwd=cwd) for j in jobs]
            if modules:
                # while the compilers run: the table of embedded code objects, as assembler source
                with open(blob_s, "w") as f:
                    f.write('\t.section .rodata.exmc_blobs,"a",@progbits\n')
                    for i, o in enumerate(part_objs):
                        f.write("\t.balign 4096\nexmc_blob_%d:\n\t.incbin \"%s\"\nexmc_blob_%d_end:\n" % (i, o, i))
                    f.write('\t.section .data.rel.ro.exmc_blob_table,"aw",@progbits\n\t.balign 8\n'
                            "\t.globl exmc_blob_table\n\t.type exmc_blob_table,@object\nexmc_blob_table:\n")
                    for i in range(len(part_objs)):
                        f.write("\t.quad exmc_blob_%d\n\t.quad exmc_blob_%d_end\n" % (i, i))
                    f.write("\t.quad 0\n\t.quad 0\n\t.size exmc_blob_table, .-exmc_blob_table\n"
                            '\t.section .note.GNU-stack,"",@progbits\n')
            rcs = [p_.wait() for p_ in procs]
            if any(rcs):
                raise subprocess.CalledProcessError(max(rcs), jobs[rcs.index(max(rcs))])
            t_compiled = time.time()
            link_objs = objs
            if modules:
                # a kernel the host unit can launch but no part defines used to be a link error; in this
                # form it would be hipErrorInvalidDeviceFunction at that kernel's first launch
                _check_module_kernels(objs[0], part_objs, common)
                # (.incbin reads the code objects now; assembled by the compiler driver already in use:
                # no second toolchain in the JIT path)
                subprocess.check_call([hipcc, "-x", "assembler", "-c", "-o", blob_o, blob_s], cwd=cwd)
                link_objs = [objs[0], blob_o]
            link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + link_objs + [common]
            if verbose:
                print(" ".join(link))
            subprocess.check_call(link, cwd=cwd)
            if os.environ.get("EXMC_PLUGIN_TIMING"):
                print("[exmc plug-in] %d units compiled side by side in %.2f s, link %.2f s" % (
                    len(jobs), t_compiled - t_start, time.time() - t_compiled), file=sys.stderr)
        os.replace(tmp, so)
    finally:
        for f in [tmp] + objs:
            if os.path.exists(f):
                os.remove(f)
    return so


def _elf_symbol_names(path):
    """Names in the symbol tables of a little-endian ELF64 file (no tool needed)."""
    import struct
    b = open(path, "rb").read()
    if b[:24] == b"__CLANG_OFFLOAD_BUNDLE__":      # what --cuda-device-only -c writes: the code object inside
        n, = struct.unpack_from("<Q", b, 24)
        at, names = 32, set()
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", b, at)
            triple = b[at + 24:at + 24 + tl]
            at += 24 + tl
            if size and triple.startswith(b"hip"):
                names |= _elf_symbols(b[off:off + size], path)
        return names
    return _elf_symbols(b, path)


def _elf_symbols(b, path):
    import struct
    if b[:4] != b"\x7fELF" or b[4] != 2:
        raise RuntimeError("%s is not an ELF64 object" % path)
    shoff, = struct.unpack_from("<Q", b, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", b, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", b, shoff + i * shentsize) for i in range(shnum)]
    names = set()
    for sh in secs:
        if sh[1] not in (2, 11):          # SHT_SYMTAB, SHT_DYNSYM
            continue
        off, size, link, entsize = sh[4], sh[5], sh[6], sh[9] or 24
        stroff = secs[link][4]
        for i in range(size // entsize):
            st_name, = struct.unpack_from("<I", b, off + i * entsize)
            if st_name:
                end = b.index(b"\0", stroff + st_name)
                names.add(b[stroff + st_name:end].decode("ascii", "replace"))
    return names


def _check_module_kernels(host_obj, part_objs, common_obj):
    """Every device-side kernel name the host unit carries (the operands of its EXMC_KLAUNCH calls,
    stored as strings) must be defined by one of the parts' code objects -- or, for the
    model-independent kernels it launches through their host stubs, by the prebuilt common object."""
    import re
    wanted = set(m.decode() for m in re.findall(rb"_ZN4exmc\d+\w*kernel\w*", open(host_obj, "rb").read()))
    have = set()
    for o in list(part_objs) + [common_obj]:
        have |= _elf_symbol_names(o)
    missing = sorted(n for n in wanted if n not in have)
    if missing:
        raise RuntimeError("plug-in build: the host unit can launch %d kernel(s) that no part defines "
                           "(exmc_plugin_kernels.inc out of step with exmc_hip.hip?):\n  %s"
                           % (len(missing), "\n  ".join(missing)))


def inverse_stick_breaking(x):
    """Transform.inverse_stick_breaking (transform.ex:234-249): a point of the K-simplex to its
    K - 1 unconstrained coordinates (host side: initial values only)."""
    x = np.asarray(x, dtype=np.float64)
    z, rem = [], 1.0
    for i in range(x.shape[0] - 1):
        y = min(max(x[i] / rem, 1.0e-10), 1.0 - 1.0e-10)
        z.append(math.log(y) - math.log(1.0 - y))
        rem = rem - x[i]
    return np.array(z)


def stick_breaking(z):
    """Transform.apply(:stick_breaking, z) on the last axis (transform.ex:125-165): {.., K-1} -> {.., K}."""
    z = np.asarray(z, dtype=np.float64)
    sp = lambda v: np.maximum(v, 0.0) + np.log1p(np.exp(-np.abs(v)))   # noqa: E731
    y = np.exp(-sp(-z))
    rem = np.ones(z.shape[:-1])
    cols = []
    for i in range(z.shape[-1]):
        cols.append(y[..., i] * rem)
        rem = rem * (1.0 - y[..., i])
    cols.append(rem)
    return np.stack(cols, axis=-1)


class GeneratedSpec(ModelSpec):
    """ModelSpec of a generated model; `lib_path` names the plug-in library sampler.Compiled
    loads instead of libexmc_hip.so."""

    def __init__(self, gen, lib_path, name="generated", default_init=None):
        super().__init__(CUSTOM, name, gen.data, gen.var_names, gen.transforms, default_init)
        self.gen = gen
        self.lib_path = lib_path
        self.vector_entries = dict(getattr(gen, "vector_entries", {}))   # id -> (offset, length)
        self.simplex_entries = dict(getattr(gen, "simplex_entries", {}))  # id -> (offset, K - 1)

    def flat_order(self):
        # the generator lays the entries out in PointMap order already (ids sorted, a vector entry's
        # elements consecutive: point_map.ex:30-60); sorting the per-element names would not be it
        return list(range(self.d))

    def to_unconstrained(self, init_values):
        # invert_ncp_init + PointMap.to_unconstrained (sampler.ex:351-392)
        vals = {}
        for k, v in init_values.items():
            if k in self.simplex_entries:         # a point of the K-simplex -> K - 1 unconstrained slots
                z = inverse_stick_breaking(v)
                if z.shape != (self.simplex_entries[k][1],):
                    raise ValueError("init value of %r must have %d elements" % (k, self.simplex_entries[k][1] + 1))
                vals.update({"%s[%d]" % (k, i): float(x) for i, x in enumerate(z)})
            elif k in self.vector_entries:        # a vector rv's init is a sequence of its elements
                a = np.asarray(v, dtype=np.float64)
                if a.shape != (self.vector_entries[k][1],):
                    raise ValueError("init value of %r must have %d elements" % (k, self.vector_entries[k][1]))
                vals.update({"%s[%d]" % (k, i): float(x) for i, x in enumerate(a)})
            else:
                vals[k] = float(v)
        raw = {}
        for id_, src in self.gen.ncp_info.items():
            if id_ in vals:
                res = lambda s: vals[s] if isinstance(s, str) else float(s)   # noqa: E731
                raw[id_] = (vals[id_] - res(src["mu"])) / res(src["sigma"])
        vals.update(raw)
        q = np.zeros(self.d)
        for i, name in enumerate(self.var_names):
            x = vals[name]
            t = self.transforms.get(name)
            if t == "log":
                q[i] = math.log(x)
            elif t == "softplus":
                q[i] = math.log(math.expm1(x))
            elif t == "logit":
                q[i] = math.log(x) - math.log1p(-x)
            else:
                q[i] = x
        return q

    def constrain(self, q):
        # build_trace: Transform.apply per entry, then reconstruct_ncp (sampler.ex:1281-1313)
        x = np.array(q, dtype=np.float64, copy=True)
        sp = lambda v: np.maximum(v, 0.0) + np.log1p(np.exp(-np.abs(v)))   # noqa: E731
        for i, name in enumerate(self.var_names):
            t = self.transforms.get(name)
            if t == "log":
                x[..., i] = np.exp(np.clip(x[..., i], -200.0, 200.0))
            elif t == "softplus":
                x[..., i] = sp(x[..., i])
            elif t == "logit":
                x[..., i] = np.exp(-sp(-x[..., i]))
        idx = {n: i for i, n in enumerate(self.var_names)}
        done, pending = set(), dict(self.gen.ncp_info)
        while pending:
            ready = [i for i, s in pending.items()
                     if all(not (isinstance(v, str) and v in pending) for v in s.values())]
            if not ready:
                raise CodegenError("cyclic non-centred references")
            for id_ in ready:
                s = pending.pop(id_)
                val = lambda v: x[..., idx[v]] if isinstance(v, str) else float(v)   # noqa: E731
                x[..., idx[id_]] = val(s["mu"]) + val(s["sigma"]) * x[..., idx[id_]]
                done.add(id_)
        return x


def compile_ir(ir, ncp=True, name="generated", default_init=None, verbose=False, rewrite_passes=False,
               lanes=None, waves_per_simd=1):
    """IR -> GeneratedSpec with its plug-in library built. Needs hipcc (no fallback). `lanes`,
    `waves_per_simd`: see `generate` / codegen_lanes.generate."""
    gen = generate(ir, ncp=ncp, rewrite_passes=rewrite_passes, lanes=lanes, waves_per_simd=waves_per_simd)
    so = build_plugin(gen, verbose=verbose)
    return GeneratedSpec(gen, so, name=name, default_init=default_init)


# ---------------------------------------------------------------------------------------------
# language-neutral front door: python -m exmc_amd.codegen model.json out_dir [--no-build]
#   model.json: {"ncp": true, "nodes": {"mu": {"op": "rv", "dist": "normal",
#                "params": {"mu": 0.0, "sigma": 5.0}, "transform": null}, ...,
#                "y_obs": {"op": "obs", "target": "y", "value": [2.1, 1.8]}}}
#   ("rewrite": true runs the reference's IR passes first; obs nodes take reduce / weight / mask /
#   censored / likelihood, {"op": "det", "fun": "affine", "args": [a, b, "x"]} and
#   {"op": "meas_obs", "target": "x", "value": [...], "info": ["affine", a, b]} are accepted)
#   out_dir gets exmc_gen_model.h, libexmc_hip_gen.so and model.json (d, var_names = the flat
#   order, transforms, ncp_info, data = what exmc_hip_model_create takes with EXMC_MODEL_CUSTOM).
# ---------------------------------------------------------------------------------------------
def ir_from_json(doc):
    ir = IR()
    if doc.get("data") is not None:
        ir.data(doc["data"])
    def param(v):      # {"f32": x}: an untyped Nx.tensor(<float>) (class F32)
        return F32(v["f32"]) if isinstance(v, dict) and set(v) == {"f32"} else v
    for id_, n in doc["nodes"].items():
        if n.get("op") == "rv":
            ir.rv(id_, n["dist"], {k: param(v) for k, v in n["params"].items()}, transform=n.get("transform"))
        elif n.get("op") == "obs":
            opts = {k: n[k] for k in ("reduce", "weight", "mask", "censored", "likelihood") if n.get(k) is not None}
            ir.obs(id_, n["target"], n["value"], **opts)
        elif n.get("op") == "det":
            ir.det(id_, n["fun"], n["args"])
        elif n.get("op") == "meas_obs":
            ir.meas_obs(id_, n["target"], n["value"], tuple(n["info"]))
        else:
            raise CodegenError("node %r: op %r is not covered" % (id_, n.get("op")))
    if doc.get("term_order") is not None:
        ir.order(doc["term_order"])
    return ir


def main(argv=None):
    import json
    import shutil
    import sys
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) < 2:
        raise SystemExit("usage: python -m exmc_amd.codegen model.json out_dir [--no-build]")
    doc = json.load(open(argv[0]))
    gen = generate(ir_from_json(doc), ncp=doc.get("ncp", True), rewrite_passes=doc.get("rewrite", False),
                   lanes=doc.get("lanes"), waves_per_simd=doc.get("waves_per_simd", 1))
    os.makedirs(argv[1], exist_ok=True)
    with open(os.path.join(argv[1], "exmc_gen_model.h"), "w") as f:
        f.write(gen.header)
    meta = dict(kind=CUSTOM, d=gen.d, var_names=gen.var_names, transforms=gen.transforms,
                ncp_info=gen.ncp_info, data=gen.data.tolist(), digest=gen.digest, lanes_per_chain=gen.lanes)
    if "--no-build" not in argv:
        shutil.copyfile(build_plugin(gen), os.path.join(argv[1], "libexmc_hip_gen.so"))
        meta["library"] = "libexmc_hip_gen.so"
    with open(os.path.join(argv[1], "model.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(os.path.join(argv[1], "model.json"))


if __name__ == "__main__":
    main()
