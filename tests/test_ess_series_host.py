"""The device's per-series ESS routine (exmc_amd/csrc/exmc_ess.hpp: lags in blocks of eight,
stopping at Geyer's first non-positive pair) compiled for the host, against the checker's direct
restatement of diagnostics.ex:123-167 (all lags, then the pair rule): identical bits."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ess") / "libess_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", out,
                           os.path.join(ROOT, "tests", "host", "ess_host_shim.cpp")])
    L = C.CDLL(out)
    L.ess_series_host.argtypes = [C.POINTER(C.c_double), C.c_long, C.c_int]
    L.ess_series_host.restype = C.c_double
    return L


def ar1(rng, n, phi):
    e = rng.normal(size=n)
    x = np.zeros(n)
    for i in range(1, n):
        x[i] = phi * x[i - 1] + e[i]
    return x


def both(shim, x, stride=1):
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = x.size // stride if stride > 1 else x.size
    got = shim.ess_series_host(O.dptr(x), stride, n)
    col = np.ascontiguousarray(x[::stride][:n])
    want = O.lib().exo_ess(O.dptr(col), n)
    return got, want


@pytest.mark.parametrize("n", [1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 100, 999, 1000, 1003])
@pytest.mark.parametrize("phi", [-0.6, 0.0, 0.5, 0.95, 0.999])
def test_ar1_series_bit_exact(shim, n, phi):
    rng = np.random.default_rng(1000 * n + int(100 * phi))
    got, want = both(shim, ar1(rng, n, phi))
    assert got == want


def test_edge_series(shim):
    for x in (np.zeros(50), np.full(33, 2.5), np.arange(64.0), np.array([1.0, -1.0] * 40),
              np.array([1.0, 2.0, 3.0, 4.0]), np.r_[np.zeros(20), 1.0, np.zeros(20)]):
        got, want = both(shim, x)
        assert got == want, x[:5]
    # a series whose positive pairs run to the very last lag (monotone ramp) and a non-finite one
    x = np.linspace(-1, 1, 41) ** 3
    assert both(shim, x)[0] == both(shim, x)[1]
    x = np.r_[np.ones(10), np.inf, np.ones(10)]
    got, want = both(shim, x)
    assert got == want == 21.0


def test_strided_trace_layout(shim):
    """[S][D][C] layout: series (dim, chain) is every (D*C)-th element."""
    rng = np.random.default_rng(5)
    S, DC = 257, 6
    tr = np.stack([ar1(rng, S, 0.3 + 0.1 * k) for k in range(DC)], axis=1)   # [S][DC]
    flat = np.ascontiguousarray(tr).ravel()
    for k in range(DC):
        sub = np.ascontiguousarray(flat[k:])
        got = shim.ess_series_host(O.dptr(sub), DC, S)
        col = np.ascontiguousarray(tr[:, k])
        assert got == O.lib().exo_ess(O.dptr(col), S)
