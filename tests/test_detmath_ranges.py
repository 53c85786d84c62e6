"""include/exmc_detmath.h: the range-restricted exp / log variants the kernels call where the call
site proves the argument's range (no special-case branches) equal the general functions bit for
bit over their stated domains, special values included. Host build of the shared header; the
device spellings are compared by tools/probe/detmath_probe.hip (tests/test_gpu_detmath.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("detmath") / "libdetmath_host.so")
    fma = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared"]
                          + fma + ["-o", out, os.path.join(ROOT, "tests", "host", "detmath_host_shim.c"), "-lm"])
    L = C.CDLL(out)
    for n in ("h_exp", "h_log", "h_exp_pm200", "h_exp_le0", "h_log_ge1", "h_log_unit"):
        getattr(L, n).argtypes = [C.c_double]
        getattr(L, n).restype = C.c_double
    L.h_compare.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_long]
    L.h_compare.restype = C.c_long
    return L


def _bad(lib, which, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return lib.h_compare(which, x.ctypes.data_as(C.POINTER(C.c_double)), x.size)


def test_exp_pm200(lib):
    rng = np.random.default_rng(1)
    x = np.r_[rng.uniform(-200, 200, 2_000_000), [-200.0, 200.0, 0.0, -0.0, 1e-300, -1e-300]]
    assert _bad(lib, 0, x) == 0


def test_exp_le0(lib):
    rng = np.random.default_rng(2)
    x = np.r_[-rng.uniform(0, 800, 1_000_000), -745.2 + rng.uniform(0, 1.5, 1_000_000),
              -np.exp(rng.uniform(-700, 700, 200_000)),
              [0.0, -0.0, -np.inf, -746.0, -745.1332191019412, -745.13321910194, -1e300, -5e-324, np.nan]]
    assert _bad(lib, 1, x) == 0
    assert np.isnan(lib.h_exp_le0(np.nan)) and lib.h_exp_le0(-np.inf) == 0.0 and lib.h_exp_le0(0.0) == 1.0


def test_log_ge1(lib):
    rng = np.random.default_rng(3)
    x = np.r_[1.0 + rng.uniform(0, 1, 1_000_000), np.exp(rng.uniform(0, 709, 1_000_000)),
              [1.0, 2.0, 1.0000000000000002, 1e308, 1.7976931348623157e308, np.nan]]
    assert _bad(lib, 2, x) == 0
    assert lib.h_log_ge1(1.0) == 0.0 and np.isnan(lib.h_log_ge1(np.nan))


def test_log_unit(lib):
    rng = np.random.default_rng(4)
    x = np.r_[np.floor(rng.uniform(0, 1, 2_000_000) * 2.0 ** 53) / 2.0 ** 53, [0.0, 2.0 ** -53, 0.5, 1 - 2.0 ** -53]]
    assert _bad(lib, 3, x) == 0
    assert lib.h_log_unit(0.0) == -np.inf
