"""Host checker for a generated model (TEST INFRASTRUCTURE, like everything under oracle/).

The text exmc_amd/codegen.py emits is compiled here with gcc into a small shared object and
hooked into the CPU oracle as EXO_MODEL_CUSTOM, so the oracle's leapfrog / tree / sampler run
over exactly the expression list the HIP functors Custom<1> / Custom<16> were compiled from
(deterministic math, no contraction). For the 16-lane layout the per-lane function runs in a loop
over 16 virtual lanes and the partials are added in the order of the device's xor butterfly.
Product code never imports this file.
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
#include <string.h>
#include "exmc_detmath.h"
#define EXMC_GEN_HOST static inline
#define EXMC_GEN_FN static inline
#define EXMC_GEN_EXP exmc_exp
#define EXMC_GEN_LOG exmc_log
#define EXMC_GEN_LOG1P exmc_log1p
#define EXMC_GENV_EXP exmc_exp
#define EXMC_GENV_LOG exmc_log
#define EXMC_GENV_LOG1P exmc_log1p
#define EXMC_GEN_ERF exmc_erf
#define EXMC_GENV_ERF exmc_erf
#define EXMC_GENL_EXP exmc_exp
#define EXMC_GENL_LOG exmc_log
#define EXMC_GENL_LOG1P exmc_log1p
#define EXMC_GENL_ERF exmc_erf
/* the lane layout (exmc_amd/codegen_lanes.py) on virtual lanes: every lane runs the function up to
 * the butterfly (pass 0: its partial sums are collected), the sums are added in the order of the
 * device's xor butterfly, every lane runs it again to the end (pass 1) */
typedef struct { double x, y; } exmc_gen_d2;
typedef struct { double* sh; int pass; double* S; const double* R; double* SW; const double* RW; int g0, ng; } exmc_gen_ctx;
/* the one-chain warmup spreads the units of a family over the 64 / G lane groups of the wavefront
 * (Custom...Split of exmc_models.hpp): group g0 takes the slots g0, g0 + ng, ...; the groups' sums are
 * added in group order after each group's butterfly (the emulation does that outside the function) */
#define EXMC_GEN_G0 ctx->g0
#define EXMC_GEN_NG ctx->ng
#define EXMC_GEN_XGROUP(s)
#define EXMC_GEN_CTX_DECL , exmc_gen_ctx* ctx
#define EXMC_GEN_SH(i) ctx->sh[i]
/* pass 0 ends at the butterfly of the spread sums of the uniform part (when the model has any),
 * pass 1 at the butterfly of the families, pass 2 runs to the end */
#define EXMC_GEN_ALLSUM_W(w) do { \
    if (ctx->pass == 0) { memcpy(ctx->SW + (size_t)l * EXMC_GEN_NW, w, sizeof(double) * EXMC_GEN_NW); return 0.0; } \
    memcpy(w, ctx->RW, sizeof(double) * EXMC_GEN_NW); } while (0)
#define EXMC_GEN_ALLSUM(s) do { \
    if (ctx->pass == 1) { memcpy(ctx->S + (size_t)l * EXMC_GEN_NS, s, sizeof(double) * EXMC_GEN_NS); return 0.0; } \
    memcpy(s, ctx->R, sizeof(double) * EXMC_GEN_NS); } while (0)
#define EXMC_GEN_FENCE()
#define EXMC_GEN_FMA(a, b, c) __builtin_fma(a, b, c)
/* lane_batch of exmc_device.hpp: every value by the operations of the plain call */
#define EXMC_GEN_BATCH_LOG(n, b) do { for (int i_ = 0; i_ < n; i_++) b[i_] = EXMC_GENL_LOG(b[i_]); } while (0)
#define EXMC_GEN_BATCH_EXP(n, b) do { for (int i_ = 0; i_ < n; i_++) b[i_] = EXMC_GENL_EXP(b[i_]); } while (0)
#define EXMC_GEN_BATCH_LOG1P(n, b) do { for (int i_ = 0; i_ < n; i_++) b[i_] = EXMC_GENL_LOG1P(b[i_]); } while (0)
#define EXMC_GEN_BATCH_RCP(n, b) do { for (int i_ = 0; i_ < n; i_++) b[i_] = 1.0 / b[i_]; } while (0)
#include "%(header)s"
#ifdef EXMC_GEN_LANES
#define EXMC_GEN_LANES_SECTION
#define EXMC_GEN_LANES_NAME exmc_gen_lanes
#define EXMC_GEN_LT(i) lt[i]
#define EXMC_GEN_LT2(i) (*(const exmc_gen_d2*)&EXMC_GEN_LT(i))   /* a pair of columns, one load on the device */
#define EXMC_GEN_IT(i) ((const int*)(lt + EXMC_GEN_IOFF))[i]
#include "%(header)s"
#endif
int exmc_gen_check_dim(void) { return EXMC_GEN_D; }
int exmc_gen_check_ndata(void) {
  int n = EXMC_GEN_NDATA;
#ifdef EXMC_GEN_VEC
  n += EXMC_GEN_NVU + 16 * EXMC_GEN_NLR;
#endif
#ifdef EXMC_GEN_LANES
  n += EXMC_GEN_NLT;
#endif
  return n;
}
#ifdef EXMC_GEN_ONE_LANE
double exmc_gen_check(const double* data, const double* q, double* g) {
  double c[EXMC_GEN_NCONST];
  exmc_gen_fold(data, c);
  return exmc_gen_logp_grad(c, q, g);
}
#endif
#ifdef EXMC_GEN_LANES
/* Custom<EXMC_GEN_LANES> of exmc_amd/csrc/exmc_models.hpp: dimension i in slot i / G of lane i mod G */
int exmc_gen_check_lanes(void) { return EXMC_GEN_LANES; }
static double exmc_gen_check_groups(const double* data, const double* q, double* g, int ng) {
  enum { G = EXMC_GEN_LANES };
  const double* lt = data + EXMC_GEN_LOFF;
  double sh[EXMC_GEN_LSH], S[G * EXMC_GEN_NS], R[EXMC_GEN_NS], gl[EXMC_GEN_DPL], lp = 0.0;
  double SW[G * (EXMC_GEN_NW + 1)], RW[EXMC_GEN_NW + 1];
  exmc_gen_ctx ctx = {sh, 0, S, R, SW, RW, 0, ng};
  for (int i = 0; i < EXMC_GEN_LSH; i++) sh[i] = 0.0;
  for (int i = 0; i < EXMC_GEN_D; i++) sh[i] = q[i];
  const int* ell = (const int*)lt + EXMC_GEN_ELL_OFF;
  for (int stage = (EXMC_GEN_NW > 0) ? 0 : 1; stage < 2; stage++) {
    const int n = stage ? EXMC_GEN_NS : EXMC_GEN_NW;
    double* const from = stage ? S : SW;
    double* const to = stage ? R : RW;
    ctx.pass = stage;
    /* the spread sums of the uniform part are not split: every group computes them alike */
    for (int grp = 0; grp < (stage ? ng : 1); grp++) {
      ctx.g0 = grp;
      for (int l = 0; l < G; l++) (void)exmc_gen_lanes(lt, ell + l * EXMC_GEN_NELL, l, gl, &ctx);
      for (int k = 0; k < n; k++) {   /* group_allsum_n: xor butterfly over the G lanes */
        double part[G], nxt[G];
        for (int l = 0; l < G; l++) part[l] = from[l * n + k];
        for (int m = 1; m < G; m <<= 1) {
          for (int l = 0; l < G; l++) nxt[l] = part[l] + part[l ^ m];
          memcpy(part, nxt, sizeof part);
        }
        to[k] = grp ? to[k] + part[0] : part[0];   /* xgroup_sum_n: group 0 first */
      }
    }
  }
  ctx.pass = 2;
  ctx.g0 = 0;   /* every group ends with the same sums and the same strips: group 0's gradient */
  for (int l = 0; l < G; l++) {
    const double v = exmc_gen_lanes(lt, ell + l * EXMC_GEN_NELL, l, gl, &ctx);
    if (l == 0) lp = v;
    for (int k = 0; k < EXMC_GEN_DPL; k++)
      if (l + k * G < EXMC_GEN_D) g[l + k * G] = gl[k];
  }
  return lp;
}
double exmc_gen_checkL(const double* data, const double* q, double* g) {
  return exmc_gen_check_groups(data, q, g, 1);
}
/* the layout of the one-chain warmup: the units over all 64 / G groups of a wavefront */
double exmc_gen_checkLW(const double* data, const double* q, double* g) {
  return exmc_gen_check_groups(data, q, g, 64 / EXMC_GEN_LANES);
}
#endif
#ifdef EXMC_GEN_VEC
/* Custom<16> of exmc_amd/csrc/exmc_models.hpp on 16 virtual lanes */
double exmc_gen_check16(const double* data, const double* q, double* g) {
  const double* vdata = data + EXMC_GEN_NDATA;
  double vc[EXMC_GEN_NVC];
  double S[16][EXMC_GEN_NS], gown[16], sg[EXMC_GEN_D], slp = 0.0, R[EXMC_GEN_NS];
  static const int smap[EXMC_GEN_D] = EXMC_GEN_SMAP;
  exmc_gen_vfold(vdata, vc);
  for (int l = 0; l < 16; l++) {
    double lc[EXMC_GEN_NLC], sgl[EXMC_GEN_D], slpl;
    exmc_gen_vfold_lane(vc, vdata + EXMC_GEN_NVU + l * EXMC_GEN_NLR, lc);
    exmc_gen_lane(vc, lc, q, l < EXMC_GEN_D ? q[l] : 0.0, S[l], &gown[l], sgl, &slpl);
    if (l == 0) { memcpy(sg, sgl, sizeof sg); slp = slpl; }
  }
  for (int k = 0; k < EXMC_GEN_NS; k++) {   /* group_allsum_n: xor butterfly over the 16 lanes */
    double part[16], nxt[16];
    for (int l = 0; l < 16; l++) part[l] = S[l][k];
    for (int m = 1; m < 16; m <<= 1) {
      for (int l = 0; l < 16; l++) nxt[l] = part[l] + part[l ^ m];
      memcpy(part, nxt, sizeof part);
    }
    R[k] = part[0];
  }
  for (int i = 0; i < EXMC_GEN_D; i++) {
    const double tot = smap[i] > 0 ? sg[i] + R[smap[i]] : sg[i];
    g[i] = tot + gown[i];
  }
  return slp + R[0];
}
#endif
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


def _lib(gen):
    L = C.CDLL(build(gen))
    for name in ("exmc_gen_check", "exmc_gen_check16", "exmc_gen_checkL", "exmc_gen_checkLW"):
        if hasattr(L, name):
            f = getattr(L, name)
            f.restype = C.c_double
            f.argtypes = [C.POINTER(C.c_double)] * 3
    _keep.append(L)
    return L


def model(gen, lanes=None, wave_split=False):
    """oracle Model running the generated value+gradient; lanes 1 or 16 (default: the layout the
    plug-in defaults to). Use with O.Cfg(1, lanes). wave_split: the lane layout as the one-chain
    warmup runs it (exmc_hip_model_default_warmup_lanes = 64 for a layout of fewer lanes: the units
    of a family over all lane groups of the wavefront)."""
    lanes = gen.lanes if lanes is None else lanes
    L = _lib(gen)
    assert L.exmc_gen_check_dim() == gen.d and L.exmc_gen_check_ndata() == gen.data.size
    m = O.Model(O.EXO_MODEL_CUSTOM, gen.d, gen.data)
    fn = C.cast(_entry(gen, L, lanes, wave_split), C.c_void_p)
    O.lib().exo_model_set_custom(m.h, fn)
    m.gen_lib = L
    m.lanes = lanes
    return m


def _entry(gen, L, lanes, wave_split=False):
    """The checker function of a layout: one lane, the 16-lane plate layout, or the lane layout."""
    if getattr(gen, "lane_layout", None) is not None and lanes == gen.lane_layout["lanes"]:
        return L.exmc_gen_checkLW if wave_split else L.exmc_gen_checkL
    if wave_split:
        raise ValueError("only the lane layout has a wave-split warmup form")
    if lanes == 16:
        return L.exmc_gen_check16
    if lanes != 1:
        raise ValueError("the generated model has no %d-lane layout" % lanes)
    return L.exmc_gen_check


def logp_grad(gen, q, lanes=1, wave_split=False):
    L = _lib(gen)
    q = np.ascontiguousarray(q, dtype=np.float64)
    g = np.zeros(gen.d)
    data = np.ascontiguousarray(gen.data)
    f = _entry(gen, L, lanes, wave_split)
    return f(O.dptr(data), O.dptr(q), O.dptr(g)), g
