"""The checker's Exmc.Diagnostics (oracle/exmc_oracle.c: exo_ess, exo_ess_bulk, exo_rhat) against a third statement in
plain Python lists, written from lib/exmc/diagnostics.ex:42-115, 123-167, 186-237 as the reference writes it (Enum.sum /
Enum.reduce left to right, the Geyer pair rule with its -1.0 start, average ranks for ties, the rational probit): bit for
bit on white noise, AR(1) series, short series, constants and series with ties."""
import math

import numpy as np
import pytest

import oracle as O


def acf(values, max_lag):                                       # diagnostics.ex:123-143
    n = len(values)
    mean = sum(values) / n                                      # Enum.sum: left to right
    c = [x - mean for x in values]
    var = sum(x * x for x in c)
    if var == 0.0:
        return [0.0] * (max_lag + 1)
    out = []
    for lag in range(max_lag + 1):
        s = 0.0
        for i in range(n - lag):
            s = s + c[i] * c[i + lag]
        out.append(s / var)
    return out


def ess_from_acf(a, n):                                         # diagnostics.ex:147-167
    tau = -1.0
    for k in range((len(a) - 1) // 2 + 1):
        r0 = a[2 * k] if 2 * k < len(a) else 0.0
        r1 = a[2 * k + 1] if 2 * k + 1 < len(a) else 0.0
        if r0 + r1 > 0:
            tau = tau + 2 * (r0 + r1)
        else:
            break
    return n / max(tau, 1.0)


def ess(values):                                                # diagnostics.ex:42-52
    n = len(values)
    return n * 1.0 if n < 4 else ess_from_acf(acf(values, min(n - 1, n)), n)


def probit(p):                                                  # diagnostics.ex:221-237
    def inner(p):
        t = math.sqrt(-2.0 * math.log(p))
        return t - (2.515517 + 0.802853 * t + 0.010328 * t * t) / (1.0 + 1.432788 * t + 0.189269 * t * t + 0.001308 * t * t * t)
    return -inner(p) if p < 0.5 else inner(1.0 - p)


def rank_normalize(values):                                     # diagnostics.ex:186-219
    n = len(values)
    order = sorted(range(n), key=lambda i: values[i])           # Enum.sort_by is stable
    ranks = [0.0] * n
    pos, i = 1, 0
    while i < n:
        j = i
        while j + 1 < n and values[order[j + 1]] == values[order[i]]:
            j += 1
        avg = pos + (j - i) / 2.0
        for k in range(i, j + 1):
            ranks[order[k]] = avg
        pos += j - i + 1
        i = j + 1
    return [probit((r - 0.375) / (n + 0.25)) for r in ranks]


def ess_bulk(values):                                           # diagnostics.ex:60-72
    n = len(values)
    return n * 1.0 if n < 4 else ess_from_acf(acf(rank_normalize(values), min(n - 1, n)), n)


def rhat(chains):                                               # diagnostics.ex:80-115
    split = []
    for c in chains:
        mid = len(c) // 2
        split += [c[:mid], c[mid:]]
    m = len(split)
    n = min(len(c) for c in split)
    tr = [c[:n] for c in split]
    means = [sum(c) / n for c in tr]
    grand = sum(means) / m
    b = n / (m - 1) * sum((cm - grand) ** 2 for cm in means)
    vs = [sum((x - cm) ** 2 for x in c) / (n - 1) for c, cm in zip(tr, means)]
    w = sum(vs) / m
    return math.sqrt(((n - 1) / n * w + b / n) / w)


def _series():
    rng = np.random.default_rng(17)
    out = {"white_200": rng.normal(size=200), "short_3": rng.normal(size=3), "short_5": rng.normal(size=5),
           "const_40": np.ones(40), "ties_120": np.round(rng.normal(size=120), 1), "odd_157": rng.normal(size=157)}
    for rho in (0.5, 0.95):
        x = np.zeros(300)
        for i in range(1, 300):
            x[i] = rho * x[i - 1] + rng.normal()
        out["ar_%g" % rho] = x
    return out


@pytest.mark.parametrize("name,x", sorted(_series().items()))
def test_ess_and_bulk_ess_bit_for_bit(name, x):
    L = O.lib()
    xs = [float(v) for v in x]
    a = np.ascontiguousarray(x, dtype=np.float64)
    assert L.exo_ess(O.dptr(a), a.size) == ess(xs)
    if name != "const_40":                          # (rank-normalising a constant series: every rank tied -- also equal)
        assert L.exo_ess_bulk_mode(O.dptr(a), a.size, 0) == ess_bulk(xs)
    else:
        assert L.exo_ess_bulk_mode(O.dptr(a), a.size, 0) == ess_bulk(xs) == 40.0


def test_split_rhat_bit_for_bit():
    L = O.lib()
    rng = np.random.default_rng(23)
    for C_, n in ((2, 100), (4, 251), (3, 64)):
        ch = np.ascontiguousarray(rng.normal(size=(C_, n)) + np.arange(C_)[:, None] * 0.3)
        assert L.exo_rhat(O.dptr(ch), C_, n) == rhat([[float(v) for v in row] for row in ch])
