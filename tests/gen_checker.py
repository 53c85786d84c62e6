"""Host checker for a generated model (TEST INFRASTRUCTURE, like everything under oracle/).

The text exmc_amd/codegen.py emits is compiled here with gcc into a small shared object and
hooked into the CPU oracle as EXO_MODEL_CUSTOM, so the oracle's leapfrog / tree / sampler run
over exactly the expression list the HIP functor Custom<1> was compiled from (deterministic
math, no contraction). Product code never imports this file.
"""
import ctypes as C
import os
import subprocess

import numpy as np

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_DIR = os.path.join(ROOT, "oracle", "build")

WRAPPER = """
#include <math.h>
#include "exmc_detmath.h"
#define EXMC_GEN_HOST static inline
#define EXMC_GEN_FN static inline
#define EXMC_GEN_EXP exmc_exp
#define EXMC_GEN_LOG exmc_log
#define EXMC_GEN_LOG1P exmc_log1p
#include "%(header)s"
int exmc_gen_check_dim(void) { return EXMC_GEN_D; }
int exmc_gen_check_ndata(void) { return EXMC_GEN_NDATA; }
double exmc_gen_check(const double* data, const double* q, double* g) {
  double c[EXMC_GEN_NCONST];
  exmc_gen_fold(data, c);
  return exmc_gen_logp_grad(c, q, g);
}
"""

_keep = []


def build(gen):
    os.makedirs(OUT_DIR, exist_ok=True)
    hdr = os.path.join(OUT_DIR, "gen_%s.h" % gen.digest)
    src = os.path.join(OUT_DIR, "gen_%s.c" % gen.digest)
    so = os.path.join(OUT_DIR, "gen_%s.so" % gen.digest)
    if not os.path.exists(so):
        with open(hdr, "w") as f:
            f.write(gen.header)
        with open(src, "w") as f:
            f.write(WRAPPER % dict(header=hdr))
        fma = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-ffp-contract=off",
                               "-fno-fast-math"] + fma +
                              ["-I", os.path.join(ROOT, "include"), "-shared", "-o", so, src, "-lm"])
    return so


def model(gen):
    """oracle Model running the generated value+gradient."""
    L = C.CDLL(build(gen))
    assert L.exmc_gen_check_dim() == gen.d and L.exmc_gen_check_ndata() == gen.data.size
    m = O.Model(O.EXO_MODEL_CUSTOM, gen.d, gen.data)
    fn = C.cast(L.exmc_gen_check, C.c_void_p)
    O.lib().exo_model_set_custom(m.h, fn)
    _keep.append(L)
    m.gen_lib = L
    return m


def logp_grad(gen, q):
    L = C.CDLL(build(gen))
    L.exmc_gen_check.restype = C.c_double
    L.exmc_gen_check.argtypes = [C.POINTER(C.c_double)] * 3
    q = np.ascontiguousarray(q, dtype=np.float64)
    g = np.zeros(gen.d)
    data = np.ascontiguousarray(gen.data)
    lp = L.exmc_gen_check(O.dptr(data), O.dptr(q), O.dptr(g))
    return lp, g
