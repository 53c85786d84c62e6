"""CPU tests: the deterministic math contract (include/exmc_detmath.h) and the RNG restatements."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import oracle as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                   "reference_known_answers.json")))


def _ulps(got, ref):
    return np.abs(got - ref) / np.spacing(np.maximum(np.abs(ref), 5e-324))


def test_det_exp_within_one_ulp_of_libm():
    L = O.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-745, 709.7, 100000), rng.normal(0, 3, 100000),
                         -np.exp(rng.uniform(-30, 7, 50000))])
    got = np.array([L.exo_det_exp(float(x)) for x in xs])
    assert _ulps(got, np.exp(xs)).max() <= 1.0


def test_det_log_within_one_ulp_of_libm():
    L = O.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([np.exp(rng.uniform(-700, 700, 100000)), rng.uniform(0.4, 2.5, 100000),
                         1.0 + rng.normal(0, 1e-6, 20000), [5e-324, 1e-310, 1.0, 2.0]])
    got = np.array([L.exo_det_log(float(x)) for x in xs])
    ref = np.log(xs)
    nz = ref != 0
    assert _ulps(got[nz], ref[nz]).max() <= 1.0
    assert np.all(got[~nz] == 0.0)


def test_det_log1p_within_one_ulp_of_libm():
    L = O.lib()
    rng = np.random.default_rng(2)
    xs = np.concatenate([np.exp(rng.uniform(-40, 3, 50000)), -np.exp(rng.uniform(-40, -0.7, 50000))])
    got = np.array([L.exo_det_log1p(float(x)) for x in xs])
    assert _ulps(got, np.log1p(xs)).max() <= 1.0


def test_det_erf_against_scipy():
    from scipy import special
    L = O.lib()
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-7, 7, 60000), rng.normal(0, 1, 40000), np.exp(rng.uniform(-40, 1.1, 20000)),
                         [2.9999999999, 3.0, 5.9999999, 6.0, -3.0, -6.0]])
    got = np.array([L.exo_det_erf(float(x)) for x in xs])
    ref = special.erf(xs)
    assert np.abs(got - ref).max() <= 1e-15
    small = np.abs(xs) < 3
    assert (np.abs(got[small] / ref[small] - 1.0)).max() <= 1e-15
    assert np.array_equal(got, -np.array([L.exo_det_erf(float(-x)) for x in xs]))     # odd, exactly
    assert L.exo_det_erf(0.0) == 0.0 and L.exo_det_erf(40.0) == 1.0 and L.exo_det_erf(-math.inf) == -1.0
    assert math.isnan(L.exo_det_erf(math.nan))


def test_det_math_special_values():
    L = O.lib()
    assert L.exo_det_exp(0.0) == 1.0
    assert L.exo_det_exp(-0.0) == 1.0
    assert L.exo_det_exp(710.0) == math.inf
    assert L.exo_det_exp(-746.0) == 0.0
    assert L.exo_det_exp(-math.inf) == 0.0
    assert math.isnan(L.exo_det_exp(math.nan))
    assert L.exo_det_log(1.0) == 0.0
    assert L.exo_det_log(0.0) == -math.inf
    assert math.isnan(L.exo_det_log(-1.0))
    assert L.exo_det_log(math.inf) == math.inf
    assert math.isnan(L.exo_det_log(math.nan))
    assert L.exo_det_log1p(0.0) == 0.0
    assert L.exo_det_log1p(1e-300) == 1e-300


def test_splitmix64_published_vector():
    L = O.lib()
    x = C.c_uint64(1234567)
    got = [str(L.exo_splitmix64(C.byref(x))) for _ in range(5)]
    assert got == GOLD["third_party_vectors"]["splitmix64_seed_1234567"]["expect"]


def test_xoshiro256starstar_published_vector():
    L = O.lib()
    s = (C.c_uint64 * 4)(1, 2, 3, 4)
    exp = GOLD["third_party_vectors"]["xoshiro256starstar_state_1_2_3_4"]["expect"]
    assert [str(L.exo_xoshiro_next(s)) for _ in range(len(exp))] == exp


def test_xoshiro_seed_from_u64_is_splitmix_fill():
    L = O.lib()
    s = (C.c_uint64 * 4)()
    L.exo_xoshiro_seed_from_u64(s, 1234567)
    x = C.c_uint64(1234567)
    assert list(s) == [L.exo_splitmix64(C.byref(x)) for _ in range(4)]
    u = L.exo_xoshiro_f64(s)
    assert 0.0 <= u < 1.0


def test_exsss_structure():
    """58-bit words, seed words = first two SplitMix64 outputs masked to 58 bits, uniform in
    [0,1) on a 2^-53 grid, state update = xorshift116 recurrence."""
    L = O.lib()
    m58 = (1 << 58) - 1
    for seed in (0, 1, 42, 42 + 7919 * 5, 2 ** 63 + 11):
        r = O.Rng()
        L.exo_rng_seed(C.byref(r), seed)
        x = C.c_uint64(seed)
        assert r.a == L.exo_splitmix64(C.byref(x)) & m58
        assert r.b == L.exo_splitmix64(C.byref(x)) & m58
        for _ in range(200):
            s1, s0 = r.a, r.b
            w = L.exo_rng_next(C.byref(r))
            assert 0 <= w <= m58
            v1 = (s0 * 5) & m58
            v2 = ((v1 << 7) & m58) | (v1 >> 51)
            assert w == (v2 * 9) & m58
            s1b = s1 ^ ((s1 << 24) & m58)
            assert r.a == s0
            assert r.b == s1b ^ s0 ^ (s1b >> 11) ^ (s0 >> 41)
        r2 = O.Rng(r.a, r.b)
        u = L.exo_rng_uniform(C.byref(r))
        assert 0.0 <= u < 1.0
        assert u == (L.exo_rng_next(C.byref(r2)) >> 5) * 2.0 ** -53


@pytest.mark.parametrize("mode", [0, 1])
def test_normal_s_is_standard_normal(mode):
    from scipy import stats
    L = O.lib()
    r = O.Rng()
    L.exo_rng_seed(C.byref(r), 2024)
    z = np.array([L.exo_rng_normal(C.byref(r), mode) for _ in range(200000)])
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert stats.kstest(z, "norm").pvalue > 1e-3
    assert 0.0020 < (np.abs(z) > 3.0).mean() < 0.0034      # 2*(1-Phi(3)) = 0.0027
    assert (np.abs(z) > 3.6541528853610088).sum() > 0        # the tail branch is reached


def test_normal_modes_share_the_fast_path():
    """libm and deterministic math only differ on the ~1.5% slow path; streams agree until then."""
    L = O.lib()
    a, b = O.Rng(), O.Rng()
    L.exo_rng_seed(C.byref(a), 7)
    L.exo_rng_seed(C.byref(b), 7)
    za = [L.exo_rng_normal(C.byref(a), 0) for _ in range(2000)]
    zb = [L.exo_rng_normal(C.byref(b), 1) for _ in range(2000)]
    assert np.allclose(za, zb, rtol=1e-14, atol=0)
    assert (a.a, a.b) == (b.a, b.b)


def test_ziggurat_tables_consistent():
    """Marsaglia-Tsang construction: x increasing, f = exp(-x^2/2), k_i = floor(2^51 x_{i-1}/x_i),
    equal-area layers."""
    import re
    txt = open(os.path.join(O.ROOT, "include", "exmc_zig_tables.h")).read()

    def grab(name):
        body = txt.split("#define %s {" % name)[1].split("}")[0]
        return [t.strip().rstrip("\\").strip().rstrip(",") for t in body.split("\n") if t.strip(" \\")]

    ki = [int(t.replace("ULL", "")) for t in grab("EXMC_ZIG_KI_INIT")]
    wi = [float.fromhex(t) for t in grab("EXMC_ZIG_WI_INIT")]
    fi = [float.fromhex(t) for t in grab("EXMC_ZIG_FI_INIT")]
    assert len(ki) == len(wi) == len(fi) == 256
    R = float.fromhex(re.search(r"#define EXMC_NOR_R (\S+)", txt).group(1))
    assert abs(R - 3.6541528853610088) < 1e-15
    x = np.array(wi) * 2.0 ** 51
    assert abs(x[255] - R) < 1e-15
    assert np.all(np.diff(x[1:]) > 0)
    assert ki[1] == 0 and fi[0] == 1.0
    assert np.allclose(fi[1:], np.exp(-0.5 * x[1:] ** 2), rtol=1e-14)
    for i in range(2, 256):
        assert abs(ki[i] - x[i - 1] / x[i] * 2.0 ** 51) <= 2.0
    # equal areas: x_i * (f_{i-1} - f_i) = V for the rectangular layers
    f = np.array(fi)
    areas = x[2:] * (f[1:-1] - f[2:])
    assert np.allclose(areas, areas[0], rtol=1e-10)
