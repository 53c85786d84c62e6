"""The plate layout's lane function is a section of the generated header that a build may include a
second time under another name, with other exp / log and one more parameter (the device's fast window:
exmc_models.hpp EXMC_GEN_VEC_SECTION / EXMC_GENV_NAME / EXMC_GENV_CTX_DECL). Checked here with gcc:
the header once as the host checker includes it, then the section again with COUNTING exp / log -- two
functions from one text, equal outputs on random positions, and the second one really went through the
macros it was given (the count equals the calls in the text)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from exmc_amd import codegen as cg
import gen_models as GM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "oracle", "build")

SRC = r"""
#include <math.h>
#include <string.h>
#include "exmc_detmath.h"
#define EXMC_GEN_HOST static inline
#define EXMC_GEN_FN static inline
#define EXMC_GEN_EXP exmc_exp
#define EXMC_GEN_LOG exmc_log
#define EXMC_GEN_LOG1P exmc_log1p
#define EXMC_GEN_ERF exmc_erf
#define EXMC_GENV_EXP exmc_exp
#define EXMC_GENV_LOG exmc_log
#define EXMC_GENV_LOG1P exmc_log1p
#define EXMC_GENV_ERF exmc_erf
#include "%(header)s"
#ifndef EXMC_GEN_VEC
#error "this test needs the plate layout"
#endif
/* the section once more */
#define EXMC_GEN_VEC_SECTION
#undef EXMC_GENV_NAME
#undef EXMC_GENV_CTX_DECL
#undef EXMC_GENV_EXP
#undef EXMC_GENV_LOG
#undef EXMC_GENV_LOG1P
#define EXMC_GENV_NAME lane_again
#define EXMC_GENV_CTX_DECL , int* calls
#define EXMC_GENV_EXP(x) (++*calls, exmc_exp(x))
#define EXMC_GENV_LOG(x) (++*calls, exmc_log(x))
#define EXMC_GENV_LOG1P(x) (++*calls, exmc_log1p(x))
#include "%(header)s"
#undef EXMC_GEN_VEC_SECTION

/* -> number of differing bytes over all 16 lanes' outputs; *calls = exp/log/log1p calls of ONE lane */
int both(const double* data, const double* q, int* calls) {
  const double* vdata = data + EXMC_GEN_NDATA;
  double vc[EXMC_GEN_NVC];
  int diff = 0;
  exmc_gen_vfold(vdata, vc);
  for (int l = 0; l < 16; l++) {
    double lc[EXMC_GEN_NLC];
    double s1[EXMC_GEN_NS], sg1[EXMC_GEN_D], go1, lp1, s2[EXMC_GEN_NS], sg2[EXMC_GEN_D], go2, lp2;
    exmc_gen_vfold_lane(vc, vdata + EXMC_GEN_NVU + l * EXMC_GEN_NLR, lc);
    const double qo = l < EXMC_GEN_D ? q[l] : 0.0;
    exmc_gen_lane(vc, lc, q, qo, s1, &go1, sg1, &lp1);
    *calls = 0;
    lane_again(vc, lc, q, qo, s2, &go2, sg2, &lp2, calls);
    diff += memcmp(s1, s2, sizeof s1) != 0;
    diff += memcmp(sg1, sg2, sizeof sg1) != 0;
    diff += memcmp(&go1, &go2, 8) != 0;
    diff += memcmp(&lp1, &lp2, 8) != 0;
  }
  return diff;
}
"""


def _build(gen, tag):
    os.makedirs(OUT, exist_ok=True)
    hdr = os.path.join(OUT, "vecsec_%s.h" % tag)
    src = os.path.join(OUT, "vecsec_%s.c" % tag)
    so = os.path.join(OUT, "vecsec_%s_%d.so" % (tag, os.getpid()))
    open(hdr, "w").write(gen.header)
    open(src, "w").write(SRC % dict(header=hdr))
    subprocess.check_call(["gcc", "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                           "-o", so, src, "-lm"])
    lib = C.CDLL(so)
    os.remove(so)
    lib.both.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    lib.both.restype = C.c_int
    return lib


@pytest.mark.parametrize("which", ["eight_schools", "zoo"])
def test_plate_lane_function_compiles_twice_from_one_text(which):
    if which == "eight_schools":
        gen = cg.generate(cg.eight_schools_ir(), ncp=True)
    else:
        gen = cg.generate(GM.zoo_ir(), ncp=False)
    if gen.vec is None:
        pytest.skip("no plate layout for this model")
    text = gen.header
    # the structure the device build relies on
    assert text.startswith("#ifndef EXMC_GEN_VEC_SECTION\n")
    assert text.count("#endif   /* !EXMC_GEN_VEC_SECTION */") == 1
    tail = text.split("#endif   /* !EXMC_GEN_VEC_SECTION */")[1]
    assert "EXMC_GEN_FN void EXMC_GENV_NAME(" in tail and "exmc_gen_vfold" not in tail
    n_calls = len(re.findall(r"EXMC_GENV_(?:EXP|LOG|LOG1P)\(", tail))
    lib = _build(gen, which)
    rng = np.random.default_rng(4)
    data = np.ascontiguousarray(gen.data, dtype=np.float64)
    for _ in range(20):
        q = np.ascontiguousarray(rng.normal(size=gen.d) * 0.7)
        calls = C.c_int(-1)
        assert lib.both(data.ctypes.data, q.ctypes.data, C.byref(calls)) == 0
        assert calls.value == n_calls
