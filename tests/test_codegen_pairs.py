"""The lane layout's table for wide per-unit rows (exmc_amd/codegen_lanes.py, round 6): a family of at least
PAIR_MIN_COLS columns stores them in interleaved pairs -- columns 2p and 2p + 1 of unit u at base + 2 p npad + 2 u, + 1, an
odd last column alone -- with the first pair on a 128-byte boundary of the table, and reads a pair through EXMC_GEN_LT2;
narrower families stay row-major. Checked on the generated text and data of regressions of several widths: every (unit,
column) entry of the design matrix is where the text reads it, and the decision about the workgroup form (EXMC_GEN_WG)
follows the sizes."""
import re

import numpy as np
import pytest

from exmc_amd import codegen as cg, codegen_lanes as cl, models


def _gen(n_obs, k):
    X, y = models.logistic_data(seed=11 + n_obs + k, n=n_obs, k=k)
    return cg.generate(cg.logistic_ir(X, y), ncp=True, lanes=16, waves_per_simd=2), X, y


@pytest.mark.parametrize("n_obs,k", [(500, 20), (333, 12), (100, 9), (64, 8)])
def test_wide_rows_are_stored_in_pairs_where_the_text_reads_them(n_obs, k):
    gen, X, y = _gen(n_obs, k)
    text = gen.lane_layout["text"]
    lt = np.asarray(gen.lane_layout["data"], dtype=np.float64)
    assert k >= cl.PAIR_MIN_COLS and "EXMC_GEN_LT2(" in text
    # the (offset, stride) of every column read in the text, family by family: "cc0 = EXMC_GEN_LT2(off + un * 2)" reads
    # columns c and c + 1, "c8 = EXMC_GEN_LT(off + un * s)" one
    fams = re.split(r"/\* family \d+:", text)[1:]
    seen_rows = 0
    for ftxt in fams:
        m = re.match(r" (\d+) units", ftxt)
        n_units = int(m.group(1))
        pairs = re.findall(r"(?:cc(\d+)|cc_) = EXMC_GEN_LT2\((\d+) \+ (?:un|uc) \* 2\)", ftxt)
        singles = re.findall(r"c(?:_\[j\]\[)?(\d+)\]? = EXMC_GEN_LT\((\d+) \+ (?:un|uc) \* (\d+)\)", ftxt)
        if not pairs:
            continue
        assert len(pairs) == k // 2 and len(singles) == (k & 1)
        cols = np.zeros((n_units, k))
        offs = sorted(int(o) for _, o in pairs)
        assert offs[0] % 16 == 0 or seen_rows > 0          # the first pair of the table on a 128-byte boundary
        for p, off in enumerate(offs):
            assert off % 2 == 0                            # 16-byte aligned pairs
            for u in range(n_units):
                cols[u, 2 * p] = lt[off + 2 * u]
                cols[u, 2 * p + 1] = lt[off + 2 * u + 1]
        for _, off, stride in singles:
            assert int(stride) == 1
            for u in range(n_units):
                cols[u, k - 1] = lt[int(off) + u]
        # the family's units are rows of X (the generator splits the observations by their response): every row of cols is
        # a row of the design matrix, and all of them together are all of it
        Xs = {tuple(r) for r in np.asarray(X, dtype=np.float64)}
        assert all(tuple(r) in Xs for r in cols)
        seen_rows += n_units
    assert seen_rows == n_obs


def test_narrow_rows_stay_row_major_and_small_tables_do_not_ask_for_the_workgroup_form():
    spec = models.radon()
    from gen_models import baseline_pair
    ir, ncp, _, lanes = baseline_pair("radon")
    gen = cg.generate(ir, ncp=ncp, lanes=lanes)
    assert "EXMC_GEN_LT2(" not in gen.header and "#define EXMC_GEN_WG 0" in gen.header
    g500, _, _ = _gen(500, 20)
    assert "#define EXMC_GEN_WG 1" in g500.header and g500.lane_layout["wg"] == 1
    g96, _, _ = _gen(48, 20)                                # 48 x 20: the table fits beside a one-wave workgroup
    assert "#define EXMC_GEN_WG 0" in g96.header
    g1, _, _ = _gen(500, 20)
    assert cg.generate(cg.logistic_ir(*models.logistic_data(seed=11 + 500 + 20, n=500, k=20)), ncp=True, lanes=16,
                       waves_per_simd=1).lane_layout["wg"] == 0   # one wave per SIMD: no second wave to share an image with
    assert spec.d == 90
