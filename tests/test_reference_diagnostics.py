"""The reference's own test/diagnostics_test.exs, case by case, with the reference's generator (OTP :rand
exsss seeded 42, normal_s -- restated in the checker): the acceptance bands are the reference's literals.
CPU: the checker's Diagnostics restatement (oracle/, diagnostics.ex:42-52, 80-115). GPU (-m gpu): the
device kernels behind exmc_amd.diagnostics on the same series -- the same bands, and the checker's bits."""
import ctypes as C

import numpy as np
import pytest

import oracle as O


def _normals(n, seed=42):
    """Enum.map_reduce(1..n, :rand.seed_s(:exsss, seed), &:rand.normal_s/1) -- libm mode, as the BEAM."""
    L = O.lib()
    r = O.Rng()
    L.exo_rng_seed(C.byref(r), seed)
    return np.array([L.exo_rng_normal(C.byref(r), 0) for _ in range(n)])


def _ar1(n, rho, seed=42):
    z = _normals(n, seed)
    x = np.zeros(n)
    prev = 0.0
    for i in range(n):
        prev = rho * prev + np.sqrt(1.0 - rho * rho) * z[i]
        x[i] = prev
    return x


CASES = {
    "iid_1000": lambda: _normals(1000),                       # diagnostics_test.exs:27-40
    "ar1_099_1000": lambda: _ar1(1000, 0.99),                 # :42-57
}


def _ess(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return O.lib().exo_ess(O.dptr(x), x.size)


def _rhat(chains):
    ch = np.ascontiguousarray(chains, dtype=np.float64)
    return O.lib().exo_rhat(O.dptr(ch), ch.shape[0], ch.shape[1])


def test_ess_bands_of_the_reference():
    n = 1000
    assert _ess(CASES["iid_1000"]()) > n * 0.5                # "ESS should be close to N for iid samples"
    assert _ess(CASES["ar1_099_1000"]()) < n * 0.2            # "much less than N for highly correlated samples"


def test_rhat_bands_of_the_reference():
    z = _normals(1000)                                        # chain2 continues chain1's generator (:61-78)
    c1, c2 = z[:500], z[500:]
    assert abs(_rhat([c1, c2]) - 1.0) <= 0.1
    assert _rhat([c1, c2 + 10.0]) > 1.5                       # :80-97


def test_summary_of_one_to_a_hundred():
    """diagnostics_test.exs:8-23 through the formulas of diagnostics.ex:14-34, 169-181 (divisor-n standard
    deviation, linear-interpolation quantiles): mean 50.5, std 28.87, q50 50.5, ordered quantiles."""
    x = np.arange(1, 101, dtype=np.float64)
    mean = x.mean()
    std = np.sqrt(((x - mean) ** 2).sum() / x.size)
    q = np.quantile(x, [0.05, 0.25, 0.5, 0.75, 0.95], method="linear")
    assert abs(mean - 50.5) <= 0.01 and abs(std - 28.87) <= 0.1 and abs(q[2] - 50.5) <= 1.0
    assert np.all(np.diff(q) > 0)


def test_autocorrelation_bands_of_the_reference():
    """diagnostics_test.exs:101-140 with Diagnostics.autocorrelation restated (diagnostics.ex:123-143: lag k =
    sum_{i < n-k} (x_i - mean)(x_{i+k} - mean) / sum (x_i - mean)^2)."""
    def acf(x, k):
        c = x - x.mean()
        v = (c * c).sum()
        return np.array([(c[:x.size - j] * c[j:]).sum() / v for j in range(k + 1)])
    a = acf(_normals(1000), 10)
    assert abs(a[0] - 1.0) <= 1e-10 and np.all(np.abs(a[1:]) < 0.1)
    b = acf(_ar1(5000, 0.8), 5)
    assert abs(b[0] - 1.0) <= 1e-10 and abs(b[1] - 0.8) <= 0.1 and abs(b[2] - 0.64) <= 0.15
    assert np.all(np.diff(b) < 0)


@pytest.mark.gpu
def test_device_diagnostics_on_the_reference_series(hip):
    """The same series through ess_series_kernel / rhat_kernel / the summary mirror: the reference's bands, and
    bit for bit what the checker computes."""
    from exmc_amd import diagnostics, models, sampler
    comp = sampler.compile(models.simple())              # d = 2: two series side by side
    try:
        iid, ar = CASES["iid_1000"](), CASES["ar1_099_1000"]()
        tr = np.stack([iid, ar], axis=1)[None, :, :]     # [C = 1][S = 1000][d = 2]
        e = diagnostics.ess(comp, tr)
        assert e[0, 0] > 500 and e[1, 0] < 200
        assert e[0, 0] == _ess(iid) and e[1, 0] == _ess(ar)
        z = _normals(1000)
        two = np.stack([np.stack([z[:500], z[:500]], axis=1), np.stack([z[500:], z[500:] + 10.0], axis=1)])   # [2][500][2]
        r = diagnostics.rhat(comp, two)
        assert abs(r[0] - 1.0) <= 0.1 and r[1] > 1.5
        assert r[0] == _rhat([z[:500], z[500:]]) and r[1] == _rhat([z[:500], z[500:] + 10.0])
        x = np.arange(1, 101, dtype=np.float64)
        s = diagnostics.summary(comp, np.stack([x, x[::-1]], axis=1)[None, :, :], names=["x", "y"])["x"]
        assert abs(s["mean"] - 50.5) <= 0.01 and abs(s["std"] - 28.87) <= 0.1 and abs(s["q50"] - 50.5) <= 1.0
        assert s["q5"] < s["q25"] < s["q50"] < s["q75"] < s["q95"]
    finally:
        comp.close()
