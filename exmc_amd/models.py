"""Model descriptions for the BASELINE configs.

The reference describes models as Builder/DSL IR (lib/exmc/builder.ex); the HIP library takes a
model *kind* plus its data. A ModelSpec carries what Exmc.PointMap carries for the kind: the free
variables in kernel order, their transforms (lib/exmc/transform.ex) and the reference's flat
(alphabetical, point_map.ex:37) order.
"""
import os

import numpy as np

STD_NORMAL, SIMPLE, EIGHT_SCHOOLS, SV, LOGISTIC, RADON = range(6)

EIGHT_SCHOOLS_Y = [28.0, 8.0, -3.0, 7.0, -1.0, 1.0, 18.0, 12.0]
EIGHT_SCHOOLS_SIGMA = [15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0]
# README.md:72-74 (Nx.tensor([...]) defaults to f32)
SIMPLE_Y = [float(np.float32(v)) for v in (2.1, 1.8, 2.5, 2.0, 1.9, 2.3, 2.2, 1.7, 2.4, 2.6)]


class ModelSpec:
    def __init__(self, kind, name, data, var_names, transforms, default_init=None):
        self.kind = kind
        self.name = name
        self.data = np.ascontiguousarray(np.asarray(data, dtype=np.float64))
        self.var_names = list(var_names)          # kernel order
        self.transforms = dict(transforms)        # var name -> "log" | None
        self.d = len(self.var_names)
        self.default_init = default_init

    # reference flat layout = ids sorted as strings (point_map.ex:37)
    def flat_order(self):
        return sorted(range(self.d), key=lambda i: self.var_names[i])

    def to_unconstrained(self, init_values):
        """PointMap.to_unconstrained + pack (sampler.ex:351-356); missing names are an error, as
        Map.fetch! is in the reference."""
        q = np.zeros(self.d)
        for i, name in enumerate(self.var_names):
            x = float(init_values[name])
            q[i] = np.log(x) if self.transforms.get(name) == "log" else x
        return q

    def constrain(self, q):
        """Transform.apply per entry (transform.ex:15-29): q [..., d] -> constrained."""
        x = np.array(q, dtype=np.float64, copy=True)
        for i, name in enumerate(self.var_names):
            if self.transforms.get(name) == "log":
                x[..., i] = np.exp(np.clip(x[..., i], -200.0, 200.0))
        return x


def eight_schools(y=EIGHT_SCHOOLS_Y, sigma=EIGHT_SCHOOLS_SIGMA):
    """Non-centered eight schools (benchmark/posteriordb/validate_posteriordb.exs:246-324)."""
    names = ["mu", "tau"] + ["theta_trans_%d" % j for j in range(8)]
    init = {n: 0.0 for n in names}
    init["tau"] = 1.0   # validate_posteriordb.exs:310-313
    return ModelSpec(EIGHT_SCHOOLS, "eight_schools", list(y) + list(sigma), names, {"tau": "log"},
                     init)


def simple(y=SIMPLE_Y):
    """d=2 plumbing model: mu ~ N(0,5), sigma ~ Exponential(1), y ~ N(mu, sigma)."""
    return ModelSpec(SIMPLE, "simple", list(y), ["mu", "sigma"], {"sigma": "log"},
                     {"mu": 2.0, "sigma": 1.0})


WORKLOADS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "workloads.npz")
# sha256 of the arrays' bytes: the benchmark workloads are DATA, committed once.
# They were drawn with numpy.random.default_rng(42) by the generators below, whose streams numpy does
# not promise to keep across versions -- a numpy upgrade must not change leapfrogs_per_launch, the
# tuned step sizes or the key of profiles/pmc_traffic.json. tests/test_workloads_pinned.py checks them.
WORKLOAD_SHA256 = {
    "sv_returns": "f26a618726c851925f71c3dcd5904dac163ffde50c37b7abce27c95ef69fbf1a",
    "logistic_X": "18510110199a0ab7f28508b21a092e5b625229648e34719c04400e4feffbbe5a",
    "logistic_y": "ae9f41b6fc0248642fc2f0ff932beab0b1c47162d102afcbd1e649ce3d761a6e",
    "radon_u": "ae11c72c3b8cfc8d02d7ceb59b8f14fc755dd8266875702b809e778a6c943355",
    "radon_start": "a0e2c5b0c07d935ea59ad70f31321d0b91d9d6c495dc3d6ab0753d7dae899325",
    "radon_floor": "bdeb12a77e34d2231b9f8dbd9edcd5c11caaac808d17cb7ad2a65223ca93cac5",
    "radon_y": "e6ec7e7275278656cb54b122c3f6ff6ae5aae1dfbd1ba2346c979414afb6b757",
}
_workloads = None


def workload(name):
    """One of the committed benchmark arrays (exmc_amd/data/workloads.npz), digest-checked on load."""
    global _workloads
    if _workloads is None:
        import hashlib
        z = np.load(WORKLOADS)
        got = {k: np.ascontiguousarray(z[k]) for k in z.files}
        for k, want in WORKLOAD_SHA256.items():
            if hashlib.sha256(got[k].tobytes()).hexdigest() != want:
                raise ValueError("exmc_amd/data/workloads.npz: array %s does not match its pinned digest" % k)
        _workloads = got
    return _workloads[name].copy()


def sv_returns(seed=42, T=100, sigma=0.15, nu=10.0):
    """returns simulated with sigma* = 0.15, nu* = 10 (STANDARD_BENCHMARKS.md:79). The
    default arguments are the committed benchmark series; anything else is generated (numpy
    default_rng, not stable across numpy versions)."""
    if (seed, T, sigma, nu) == (42, 100, 0.15, 10.0):
        return workload("sv_returns").tolist()
    rng = np.random.default_rng(seed)
    s = np.cumsum(rng.normal(0, sigma, T))
    return (np.exp(s) * rng.standard_t(nu, T)).tolist()


def logistic_data(seed=42, n=500, k=20):
    """X iid N(0,1), beta* = 0.5*N(0,1), alpha* = 0.5 (STANDARD_BENCHMARKS.md:77),
    y ~ Bernoulli(sigmoid(alpha* + X beta*)). The reference's generator is not in its
    repository; this one is build-defined. The default arguments return the committed benchmark
    arrays (see WORKLOAD_SHA256); other arguments generate (numpy default_rng(seed))."""
    if (seed, n, k) == (42, 500, 20):
        return workload("logistic_X"), workload("logistic_y")
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, k))
    beta = 0.5 * rng.normal(size=k)
    p = 1.0 / (1.0 + np.exp(-(0.5 + X @ beta)))
    y = (rng.uniform(size=n) < p).astype(np.float64)
    return X, y


def logistic(X=None, y=None):
    """Logistic regression, d = K+1 (STANDARD_BENCHMARKS.md:41-49). Kernel order alpha,
    beta_1..beta_K; the reference's flat order is the string sort (alpha, beta_1, beta_10, ...)."""
    if X is None:
        X, y = logistic_data()
    X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    y = np.asarray(y, dtype=np.float64)
    n, k = X.shape
    names = ["alpha"] + ["beta_%d" % j for j in range(1, k + 1)]
    spec = ModelSpec(LOGISTIC, "logistic", np.concatenate([X.ravel(), y]), names, {},
                     {nm: 0.0 for nm in names})
    spec.n_obs = n
    return spec


def radon_data(seed=42, n_counties=85, n_obs=919):
    """notebooks/09_radon_bhm.livemd: 85 counties, ~919 observations, truth mu_alpha 1.4,
    gamma_u 0.7, sigma_alpha 0.4, sigma_y 0.7, beta -0.7. benchmark/radon_data.exs is absent from
    the reference repository, so this generator is build-defined. The default arguments return the
    committed benchmark arrays (see WORKLOAD_SHA256); other arguments generate (numpy default_rng(seed))."""
    if (seed, n_counties, n_obs) == (42, 85, 919):
        return workload("radon_u"), workload("radon_start"), workload("radon_floor"), workload("radon_y")
    rng = np.random.default_rng(seed)
    w = rng.gamma(1.5, 1.0, size=n_counties) + 0.15
    counts = np.maximum(2, np.floor(w / w.sum() * n_obs)).astype(int)
    while counts.sum() > n_obs:
        counts[np.argmax(counts)] -= 1
    while counts.sum() < n_obs:
        counts[np.argmin(counts)] += 1
    u = rng.normal(0.0, 0.5, size=n_counties)
    alpha = 1.4 + 0.7 * u + 0.4 * rng.normal(size=n_counties)
    start = np.concatenate([[0], np.cumsum(counts)])
    county = np.repeat(np.arange(n_counties), counts)
    floor = (rng.uniform(size=n_obs) < 0.2).astype(np.float64)
    y = alpha[county] - 0.7 * floor + 0.7 * rng.normal(size=n_obs)
    return u, start, floor, y


def radon(data=None, sort_counties=True, builtin=False):
    """Hierarchical radon, d = J+5 = 90. Kernel order: the county intercepts alpha_raw_j by
    DESCENDING observation count, then mu_alpha, gamma_u, sigma_alpha, sigma_y, beta (the
    reference's flat order is the string sort; names carry the original county index). In the
    64-lane layout the observations are spread over the lanes whatever their county (observation i
    on lane i mod 64) and a county's owner lane adds up its observations' contributions
    (exmc_models.hpp Radon); at most 1024 observations -- larger data is compiled by the generator into
    a lane layout of its own (a GeneratedSpec comes back). sort_counties=False keeps the file order
    (any order is valid input for the library; with the sorted one the longest owner sums sit in
    the first dimension slot)."""
    u, start, floor, y = data if data is not None else radon_data()
    J = len(u)
    u, start, floor, y = (np.asarray(u, float), np.asarray(start).astype(int),
                          np.asarray(floor, float), np.asarray(y, float))
    sizes = np.diff(start)
    order = np.argsort(-sizes, kind="stable") if sort_counties else np.arange(J)
    obs = np.concatenate([np.arange(start[j], start[j + 1]) for j in order])
    u, floor, y = u[order], floor[obs], y[obs]
    start = np.concatenate([[0], np.cumsum(sizes[order])])
    names = ["alpha_raw_%d" % j for j in order] + ["mu_alpha", "gamma_u", "sigma_alpha",
                                                     "sigma_y", "beta"]
    init = {nm: 0.0 for nm in names}
    init.update(sigma_alpha=1.0, sigma_y=1.0)
    if len(y) > 1024 and not builtin:   # builtin=True: the kind's own spec whatever the size (the CPU checker takes any)
        # The built-in kind's 64-lane layout gives a lane 16 observation slots (exmc_hip.h:
        # EXMC_ERR_UNSUPPORTED above 1024 observations). Larger data goes through the generator: the
        # same model as Builder nodes (codegen.radon_ir), compiled into a lane layout whose observation
        # walk is as long as the data needs -- same names, same kernel order, same default init.
        from . import codegen
        ir = codegen.radon_ir(u, start, floor, y, names=names[:J])
        return codegen.compile_ir(ir, ncp=False, name="radon_n%d" % len(y), default_init=init, lanes=64,
                                  waves_per_simd=1)
    blob = np.concatenate([np.asarray(u, float), np.asarray(start, float), np.asarray(floor, float),
                           np.asarray(y, float)])
    return ModelSpec(RADON, "radon", blob, names, {"sigma_alpha": "log", "sigma_y": "log"}, init)


def sv(returns):
    """Stochastic volatility, T = 100 (STANDARD_BENCHMARKS.md:51-61). Kernel order s_1..s_T,
    sigma, nu; the reference's flat order is the string sort (nu, s_1, s_10, s_100, ...)."""
    r = list(returns)
    if len(r) != 100:
        raise ValueError("libexmc_hip compiles sv for T = 100")
    names = ["s_%d" % t for t in range(1, 101)] + ["sigma", "nu"]
    init = {n: 0.0 for n in names}
    init["sigma"] = 0.1
    init["nu"] = 10.0
    return ModelSpec(SV, "sv", r, names, {"sigma": "log", "nu": "log"}, init)
