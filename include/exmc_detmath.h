/* exmc_detmath.h — bit-reproducible f64 exp / log / log1p / erf.
 *
 * Why this exists: the reference sampler (lib/exmc/nuts/tree.ex:1045,1398,1490,1603
 * and the model log-densities, lib/exmc/dist/<name>.ex, lib/exmc/transform.ex:17-29)
 * calls the platform libm through Erlang's :math / Nx.BinaryBackend. libm results
 * differ by an ulp between platforms (glibc vs the ROCm device library), and NUTS
 * is chaotic: one ulp in a log-weight eventually flips a tree decision. To make
 * "GPU == CPU checker, bit for bit, over whole chains" a testable statement, both
 * the HIP kernels and the oracle's deterministic mode evaluate exp/log through the
 * functions below, which use only IEEE-754 correctly-rounded operations
 * (+, -, *, /, fma, rint) in a fixed order. Compile every translation unit that
 * includes this header with -ffp-contract=off (fma appears only where written).
 *
 * Accuracy (tests/test_detmath.py): <= 1 ulp vs glibc over the sampled ranges.
 * Algorithms: exp = Cody-Waite reduction by ln2 (two fma) + degree-13 Taylor
 * polynomial (Horner, fma) + two-step power-of-two scaling; log = the classic
 * s = f/(2+f) atanh-series form with the 7 published fdlibm/musl coefficients, its two
 * coefficient chains and the k*ln2 terms evaluated with fma (one rounding less per step
 * than the libm form, and one issue slot less on the GPU);
 * log1p = log(1+x) with the (x-(u-1))/u correction term.
 *
 * Device code has two spellings of the same arithmetic: exmc_exp / exmc_log leave instruction
 * selection to the compiler; exmc_exp_v / exmc_log_v run the polynomial chains as one inline-asm
 * block of three-address v_fma_f64 with the coefficients pinned in vector registers (fewer issue
 * slots and no scalar-register pressure, at the price of ~40 VGPRs: for kernels with registers
 * to spare). The bits are identical by construction (same IEEE operations in the same order).
 */
#ifndef EXMC_DETMATH_H
#define EXMC_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define EXMC_HD __host__ __device__ __forceinline__
#else
#define EXMC_HD static inline
#endif

EXMC_HD double exmc_from_bits(uint64_t u) {
  double d;
  __builtin_memcpy(&d, &u, 8);
  return d;
}
EXMC_HD uint64_t exmc_to_bits(double d) {
  uint64_t u;
  __builtin_memcpy(&u, &d, 8);
  return u;
}

#define EXMC_INF_BITS 0x7FF0000000000000ULL
#define EXMC_NAN_BITS 0x7FF8000000000000ULL

EXMC_HD int exmc_isfinite(double x) {
  return (exmc_to_bits(x) & EXMC_INF_BITS) != EXMC_INF_BITS;
}

#if defined(__HIPCC__)
/* a / b, correctly rounded, for operands that need no range scaling (device only). The compiler
 * expands an f64 division into v_div_scale x2, v_rcp_f64 (a 16-cycle instruction), two Newton
 * steps, quotient, residual, v_div_fmas, v_div_fixup; scale / fmas / fixup only act when an
 * operand or the quotient is zero, denormal, huge, infinite or NaN. These two functions are that
 * sequence without the three range instructions, and exmc_div_core alone (3 operations) when the
 * refined reciprocal of a reused divisor is kept. For in-range operands the bits are the
 * hardware division's, i.e. IEEE (tests/test_gpu_fastdiv.py checks them against `/` on the
 * device). Callers own the range argument. */
static __device__ __forceinline__ double exmc_rcp_refined(double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
static __device__ __forceinline__ double exmc_div_core(double a, double b, double r) {
  const double q = a * r;
  const double e = __builtin_fma(-b, q, a);
  return __builtin_fma(e, r, q);
}
#endif

/* log's s = f / (2 + f): f is 0 or 2^-53 <= |f| < 0.42, the denominator lies in [1.7, 2.42] */
#if defined(__HIP_DEVICE_COMPILE__)
#define EXMC_DIV_LOG(f, d) exmc_div_core((f), (d), exmc_rcp_refined(d))
#else
#define EXMC_DIV_LOG(f, d) ((f) / (d))
#endif

/* ---- polynomial cores, portable spelling ---- */
EXMC_HD void exmc_exp_core(double kf, double x, double* r_out, double* p_out) {
  double r = __builtin_fma(kf, -0x1.62e42fefa39efp-1, x);      /* - k*ln2_hi */
  r = __builtin_fma(kf, -0x1.abc9e3b39803fp-56, r);            /* - k*ln2_lo */
  double p = __builtin_fma(0x1.6124613a86d09p-33, r, 0x1.1eed8eff8d898p-29); /* 1/13!, 1/12! */
  p = __builtin_fma(p, r, 0x1.ae64567f544e4p-26);              /* 1/11! */
  p = __builtin_fma(p, r, 0x1.27e4fb7789f5cp-22);              /* 1/10! */
  p = __builtin_fma(p, r, 0x1.71de3a556c734p-19);              /* 1/9!  */
  p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-16);              /* 1/8!  */
  p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-13);              /* 1/7!  */
  p = __builtin_fma(p, r, 0x1.6c16c16c16c17p-10);              /* 1/6!  */
  p = __builtin_fma(p, r, 0x1.1111111111111p-7);               /* 1/5!  */
  p = __builtin_fma(p, r, 0x1.5555555555555p-5);               /* 1/4!  */
  p = __builtin_fma(p, r, 0x1.5555555555555p-3);               /* 1/3!  */
  *r_out = r;
  *p_out = p;
}
EXMC_HD void exmc_log_core(double w, double* u1_out, double* u2_out) {
  double u1 = __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01);
  double u2 = __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01);
  u1 = __builtin_fma(w, u1, 3.999999999940941908e-01);
  u2 = __builtin_fma(w, u2, 2.857142874366239149e-01);
  u2 = __builtin_fma(w, u2, 6.666666666666735130e-01);
  *u1_out = u1;
  *u2_out = u2;
}

/* ---- polynomial cores, gfx950 spelling: the same fma sequence as ONE asm block each. Left to the
 * compiler a Horner step acc*r + C becomes v_mov_b64 tmp, C; v_fmac_f64 tmp, acc, r (two issue
 * slots) and the hoisted coefficients fill the scalar registers until loop state spills to
 * v_writelane/v_readlane; one-instruction asm statements are each padded with an s_nop. The result
 * of a block is only consumed by ordinary arithmetic in this file: a function never returns an asm
 * output, so a DPP or lane-read consumer always sees a plain VALU producer. ---- */
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ void exmc_exp_core_v(double kf, double x, double* r_out, double* p_out) {
  double p, r;
  __asm__("v_fma_f64 %1, %2, %4, %3\n\t"
          "v_fma_f64 %1, %2, %5, %1\n\t"
          "v_fma_f64 %0, %6, %1, %7\n\t"
          "v_fma_f64 %0, %0, %1, %8\n\t"
          "v_fma_f64 %0, %0, %1, %9\n\t"
          "v_fma_f64 %0, %0, %1, %10\n\t"
          "v_fma_f64 %0, %0, %1, %11\n\t"
          "v_fma_f64 %0, %0, %1, %12\n\t"
          "v_fma_f64 %0, %0, %1, %13\n\t"
          "v_fma_f64 %0, %0, %1, %14\n\t"
          "v_fma_f64 %0, %0, %1, %15\n\t"
          "v_fma_f64 %0, %0, %1, %16"
          : "=&v"(p), "=&v"(r)
          : "v"(kf), "v"(x), "v"(-0x1.62e42fefa39efp-1), "v"(-0x1.abc9e3b39803fp-56),
            "v"(0x1.6124613a86d09p-33), "v"(0x1.1eed8eff8d898p-29), "v"(0x1.ae64567f544e4p-26),
            "v"(0x1.27e4fb7789f5cp-22), "v"(0x1.71de3a556c734p-19), "v"(0x1.a01a01a01a01ap-16),
            "v"(0x1.a01a01a01a01ap-13), "v"(0x1.6c16c16c16c17p-10), "v"(0x1.1111111111111p-7),
            "v"(0x1.5555555555555p-5), "v"(0x1.5555555555555p-3));
  *r_out = r;
  *p_out = p;
}
static __device__ __forceinline__ void exmc_log_core_v(double w, double* u1_out, double* u2_out) {
  double u1, u2;
  __asm__("v_fma_f64 %0, %2, %3, %4\n\t"
          "v_fma_f64 %1, %2, %6, %7\n\t"
          "v_fma_f64 %0, %2, %0, %5\n\t"
          "v_fma_f64 %1, %2, %1, %8\n\t"
          "v_fma_f64 %1, %2, %1, %9"
          : "=&v"(u1), "=&v"(u2)
          : "v"(w), "v"(1.531383769920937332e-01), "v"(2.222219843214978396e-01),
            "v"(3.999999999940941908e-01), "v"(1.479819860511658591e-01),
            "v"(1.818357216161805012e-01), "v"(2.857142874366239149e-01),
            "v"(6.666666666666735130e-01));
  *u1_out = u1;
  *u2_out = u2;
}
/* Third spelling: the same chains with the coefficients in SCALAR registers (a VOP3 instruction of
 * gfx950 reads one scalar operand: every Horner step acc*r + C is one v_fma_f64 with C scalar; the
 * first step of a chain has two constants, so one of them rides in a vector register). For kernels
 * that have neither the ~40 vector registers of the _v form nor issue slots for the v_mov of the
 * compiler's form (the 16-lane logistic sampling kernel: 256 registers, two waves per SIMD). */
static __device__ __forceinline__ void exmc_exp_core_s(double kf, double x, double* r_out, double* p_out) {
  double p, r;
  __asm__("v_fma_f64 %1, %2, %4, %3\n\t"
          "v_fma_f64 %1, %2, %5, %1\n\t"
          "v_fma_f64 %0, %1, %6, %7\n\t"
          "v_fma_f64 %0, %0, %1, %8\n\t"
          "v_fma_f64 %0, %0, %1, %9\n\t"
          "v_fma_f64 %0, %0, %1, %10\n\t"
          "v_fma_f64 %0, %0, %1, %11\n\t"
          "v_fma_f64 %0, %0, %1, %12\n\t"
          "v_fma_f64 %0, %0, %1, %13\n\t"
          "v_fma_f64 %0, %0, %1, %14\n\t"
          "v_fma_f64 %0, %0, %1, %15\n\t"
          "v_fma_f64 %0, %0, %1, %16"
          : "=&v"(p), "=&v"(r)
          : "v"(kf), "v"(x), "s"(-0x1.62e42fefa39efp-1), "s"(-0x1.abc9e3b39803fp-56),
            "s"(0x1.6124613a86d09p-33), "v"(0x1.1eed8eff8d898p-29), "s"(0x1.ae64567f544e4p-26),
            "s"(0x1.27e4fb7789f5cp-22), "s"(0x1.71de3a556c734p-19), "s"(0x1.a01a01a01a01ap-16),
            "s"(0x1.a01a01a01a01ap-13), "s"(0x1.6c16c16c16c17p-10), "s"(0x1.1111111111111p-7),
            "s"(0x1.5555555555555p-5), "s"(0x1.5555555555555p-3));
  *r_out = r;
  *p_out = p;
}
static __device__ __forceinline__ void exmc_log_core_s(double w, double* u1_out, double* u2_out) {
  double u1, u2;
  __asm__("v_fma_f64 %0, %2, %3, %4\n\t"
          "v_fma_f64 %1, %2, %6, %7\n\t"
          "v_fma_f64 %0, %2, %0, %5\n\t"
          "v_fma_f64 %1, %2, %1, %8\n\t"
          "v_fma_f64 %1, %2, %1, %9"
          : "=&v"(u1), "=&v"(u2)
          : "v"(w), "s"(1.531383769920937332e-01), "v"(2.222219843214978396e-01),
            "s"(3.999999999940941908e-01), "s"(1.479819860511658591e-01),
            "v"(1.818357216161805012e-01), "s"(2.857142874366239149e-01),
            "s"(6.666666666666735130e-01));
  *u1_out = u1;
  *u2_out = u2;
}
#elif defined(__HIPCC__)
/* host pass of hipcc: the names must exist; the bodies are the portable ones */
static __device__ __forceinline__ void exmc_exp_core_v(double kf, double x, double* r_out, double* p_out) {
  exmc_exp_core(kf, x, r_out, p_out);
}
static __device__ __forceinline__ void exmc_log_core_v(double w, double* u1_out, double* u2_out) {
  exmc_log_core(w, u1_out, u2_out);
}
static __device__ __forceinline__ void exmc_exp_core_s(double kf, double x, double* r_out, double* p_out) {
  exmc_exp_core(kf, x, r_out, p_out);
}
static __device__ __forceinline__ void exmc_log_core_s(double w, double* u1_out, double* u2_out) {
  exmc_log_core(w, u1_out, u2_out);
}
#endif

/* ---- exp ---- */
#define EXMC_EXP_BODY(CORE)                                                                  \
  if (!(x == x)) return x;                       /* NaN */                                   \
  if (x > 709.782712893384) return exmc_from_bits(EXMC_INF_BITS);                           \
  if (x < -745.1332191019412) return 0.0;                                                   \
  double kf = __builtin_rint(x * 0x1.71547652b82fep+0);      /* x * log2(e) */              \
  double r, p;                                                                               \
  CORE(kf, x, &r, &p);                                       /* reduction + 1/13! .. 1/3! */ \
  p = __builtin_fma(p, r, 0.5);                                                              \
  p = __builtin_fma(p, r, 1.0);                                                              \
  p = __builtin_fma(p, r, 1.0);                                                              \
  int k = (int)kf;                                                                           \
  int k1 = k >> 1;                                                                           \
  int k2 = k - k1;                                                                           \
  p *= exmc_from_bits((uint64_t)(k1 + 1023) << 52);                                          \
  p *= exmc_from_bits((uint64_t)(k2 + 1023) << 52);                                          \
  return p;

/* ---- log ---- */
#define EXMC_LOG_BODY(CORE)                                                                  \
  uint64_t ix = exmc_to_bits(x);                                                             \
  int e = 0;                                                                                 \
  if (ix < 0x0010000000000000ULL || (ix >> 63)) {                                            \
    if ((ix << 1) == 0) return -exmc_from_bits(EXMC_INF_BITS); /* +-0 */                     \
    if (ix >> 63) return exmc_from_bits(EXMC_NAN_BITS);        /* negative */                \
    x *= 0x1p54;                                               /* subnormal */               \
    ix = exmc_to_bits(x);                                                                    \
    e = -54;                                                                                 \
  } else if (ix >= EXMC_INF_BITS) {                                                          \
    return x;                                                  /* +inf, NaN */               \
  }                                                                                          \
  /* normalise mantissa into [sqrt(2)/2, sqrt(2)) */                                         \
  uint64_t t = ix + (0x3FF0000000000000ULL - 0x3FE6A09E667F3BCDULL);                         \
  e += (int)(t >> 52) - 1023;                                                                \
  ix = (t & 0x000FFFFFFFFFFFFFULL) + 0x3FE6A09E667F3BCDULL;                                  \
  double f = exmc_from_bits(ix) - 1.0;                                                       \
  double hfsq = 0.5 * f * f;                                                                 \
  double dn = 2.0 + f;                                                                       \
  double s = EXMC_DIV_LOG(f, dn);                                                            \
  double z = s * s;                                                                          \
  double w = z * z;                                                                          \
  double u1, u2;                                                                             \
  CORE(w, &u1, &u2);                                                                         \
  double t1 = w * u1;                                                                        \
  double t2 = z * u2;                                                                        \
  double R = t2 + t1;                                                                        \
  double dk = (double)e;                                                                     \
  double acc = __builtin_fma(dk, 1.90821492927058770002e-10, s * (hfsq + R)); /* k*ln2_lo */ \
  acc = (acc - hfsq) + f;                                                                    \
  return __builtin_fma(dk, 6.93147180369123816490e-01, acc);                  /* k*ln2_hi */

EXMC_HD double exmc_exp(double x) { EXMC_EXP_BODY(exmc_exp_core) }
EXMC_HD double exmc_log(double x) { EXMC_LOG_BODY(exmc_log_core) }

EXMC_HD double exmc_log1p(double x) {
  double u = 1.0 + x;
  if (u == 1.0) return x;
  return exmc_log(u) + (x - (u - 1.0)) / u;
}

/* erf, for the generated TruncatedNormal term (lib/exmc/dist/truncated_normal.ex:27-40 calls Nx.erf,
 * which is :math.erf on BinaryBackend and XLA's own polynomial under EXLA: backend-defined like exp
 * and log, so this contract fixes one evaluation). No memorised coefficient table:
 *   |x| < 3 : erf = 2/sqrt(pi) * exp(-y) * |x| * S(y), y = x^2,
 *             S = 1 + (2y/3)(1 + (2y/5)(1 + ... (1 + 2y/(2N+1)))), every term positive (no
 *             cancellation), N = 56 leaves a truncation below 2^-60 at y = 9;
 *   3 <= |x| < 6 : erf = 1 - exp(-y)/sqrt(pi) / (|x| + (1/2)/(|x| + 1/(|x| + (3/2)/(|x| + ...)))),
 *             Laplace's continued fraction for erfc cut at 24 partial quotients (relative error
 *             below 1e-15 at |x| = 3, and erfc itself is below 2.3e-5 there);
 *   |x| >= 6 : 1 (erfc(6) = 2e-17); NaN propagates; the sign is restored at the end.
 * tests/test_detmath.py: within 1e-15 of scipy.special.erf, absolute and (below 3) relative. */
EXMC_HD double exmc_erf(double x) {
  if (x != x) return x;
  const double ax = x < 0.0 ? -x : x;
  double r;
  if (ax >= 6.0) {
    r = 1.0;
  } else {
    const double y = ax * ax;
    const double e = exmc_exp(-y);
    if (ax < 3.0) {
      const double y2 = y + y;
      double s = 1.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int n = 56; n >= 1; --n) s = __builtin_fma(s * y2, 1.0 / (double)(2 * n + 1), 1.0);
      r = ((1.12837916709551257390e+00 * e) * ax) * s;                          /* 2/sqrt(pi) */
    } else {
      double f = ax;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int k = 24; k >= 1; --k) f = ax + (0.5 * (double)k) / f;
      r = 1.0 - (5.64189583547756286948e-01 * e) / f;                           /* 1/sqrt(pi) */
    }
  }
  return x < 0.0 ? -r : r;
}

/* ---- the same functions for arguments whose range the call site proves ----
 * exmc_exp / exmc_log guard every special case (NaN, overflow, underflow, zero, negative,
 * subnormal, infinity) with a branch; on the GPU each guard is a compare, an exec-mask save, a
 * branch and a restore -- a dozen issue slots per call that a lone wavefront pays in full. The
 * variants below are the main path of the same algorithm (same operations, same order, same bits)
 * plus the one or two fix-ups their stated domain still needs. Outside the domain they are
 * undefined; tests/test_detmath_rng.py compares each with the general function over its domain on
 * the host and tools/probe/detmath_probe.hip does the same for the device spellings.
 *
 *   exmc_exp_pm200   |x| <= 200                          (a clamp200'ed unconstrained scale)
 *   exmc_exp_le0     x <= 0 (including -inf), or NaN     (weights: exp(min(d, 0)), exp(b - lse))
 *   exmc_log_ge1     1 <= x < +inf, or NaN               (log(1 + y), y >= 0)
 *   exmc_log_unit    0 <= x < 1, x zero or normal        (log of a uniform_s variate)
 *
 * Scaling by 2^k: the general form multiplies by 2^(k>>1) and 2^(k - (k>>1)); the first product is
 * exact (its exponent stays normal), so only the second rounds -- the single rounding ldexp
 * performs. EXMC_EXP_SCALE is that ldexp (v_ldexp_f64 on the device, one issue slot). */
#if defined(__HIP_DEVICE_COMPILE__)
#define EXMC_EXP_SCALE(p, k) __builtin_amdgcn_ldexp(p, k)
#else
#define EXMC_EXP_SCALE(p, k) \
  (((p) * exmc_from_bits((uint64_t)(((k) >> 1) + 1023) << 52)) * exmc_from_bits((uint64_t)((k) - ((k) >> 1) + 1023) << 52))
#endif

#define EXMC_EXP_MAIN(CORE, xc)                                                              \
  double kf = __builtin_rint((xc) * 0x1.71547652b82fep+0);                                   \
  double r, p;                                                                               \
  CORE(kf, (xc), &r, &p);                                                                    \
  p = __builtin_fma(p, r, 0.5);                                                              \
  p = __builtin_fma(p, r, 1.0);                                                              \
  p = __builtin_fma(p, r, 1.0);                                                              \
  int k = (int)kf;                                                                           \
  return EXMC_EXP_SCALE(p, k);

/* the reduction, the series and the reconstruction of EXMC_LOG_BODY for a normal positive x */
#define EXMC_LOG_MAIN(CORE, x, res)                                                          \
  uint64_t ix_ = exmc_to_bits(x);                                                            \
  uint64_t t_ = ix_ + (0x3FF0000000000000ULL - 0x3FE6A09E667F3BCDULL);                       \
  int e_ = (int)(t_ >> 52) - 1023;                                                           \
  ix_ = (t_ & 0x000FFFFFFFFFFFFFULL) + 0x3FE6A09E667F3BCDULL;                                \
  double f_ = exmc_from_bits(ix_) - 1.0;                                                     \
  double hfsq_ = 0.5 * f_ * f_;                                                              \
  double dn_ = 2.0 + f_;                                                                     \
  double s_ = EXMC_DIV_LOG(f_, dn_);                                                         \
  double z_ = s_ * s_;                                                                       \
  double w_ = z_ * z_;                                                                       \
  double u1_, u2_;                                                                           \
  CORE(w_, &u1_, &u2_);                                                                      \
  double t1_ = w_ * u1_;                                                                     \
  double t2_ = z_ * u2_;                                                                     \
  double R_ = t2_ + t1_;                                                                     \
  double dk_ = (double)e_;                                                                   \
  double acc_ = __builtin_fma(dk_, 1.90821492927058770002e-10, s_ * (hfsq_ + R_));           \
  acc_ = (acc_ - hfsq_) + f_;                                                                \
  double res = __builtin_fma(dk_, 6.93147180369123816490e-01, acc_);

#define EXMC_RANGE_FUNCS(SUFFIX, EXPCORE, LOGCORE)                                           \
  EXMC_RHD double exmc_exp_pm200##SUFFIX(double x) { EXMC_EXP_MAIN(EXPCORE, x) }             \
  EXMC_RHD double exmc_exp_le0##SUFFIX(double x) {                                           \
    /* below -746 the result is 0 whatever x is; a NaN fails the comparison and flows through   \
     * (rint, fma and ldexp keep it, (int)NaN = 0) */                                        \
    const double xc = (x < -746.0) ? -746.0 : x;                                             \
    EXMC_EXP_MAIN(EXPCORE, xc)                                                               \
  }                                                                                          \
  EXMC_RHD double exmc_log_ge1##SUFFIX(double x) {                                           \
    EXMC_LOG_MAIN(LOGCORE, x, res)                                                           \
    return res + (x - x);   /* + 0.0 for a finite x (res is never -0), NaN for a NaN */       \
  }                                                                                          \
  EXMC_RHD double exmc_log_unit##SUFFIX(double x) {                                          \
    EXMC_LOG_MAIN(LOGCORE, x, res)                                                           \
    return (x == 0.0) ? -exmc_from_bits(EXMC_INF_BITS) : res;                                \
  }

#define EXMC_RHD EXMC_HD
EXMC_RANGE_FUNCS(, exmc_exp_core, exmc_log_core)
#undef EXMC_RHD

#if defined(__HIPCC__)
static __device__ __forceinline__ double exmc_exp_v(double x) { EXMC_EXP_BODY(exmc_exp_core_v) }
static __device__ __forceinline__ double exmc_log_v(double x) { EXMC_LOG_BODY(exmc_log_core_v) }
#define EXMC_RHD static __device__ __forceinline__
EXMC_RANGE_FUNCS(_v, exmc_exp_core_v, exmc_log_core_v)
EXMC_RANGE_FUNCS(_s, exmc_exp_core_s, exmc_log_core_s)
/* log of a NORMAL positive x (the caller proves x is neither zero, subnormal, infinite nor NaN): the
 * main path alone, e.g. a probability clipped into [1e-7, 1 - 1e-7] */
static __device__ __forceinline__ double exmc_log_normal_s(double x) {
  EXMC_LOG_MAIN(exmc_log_core_s, x, res)
  return res;
}
static __device__ __forceinline__ double exmc_log_normal_v(double x) {   /* ... with the vector-register cores */
  EXMC_LOG_MAIN(exmc_log_core_v, x, res)
  return res;
}
#undef EXMC_RHD
#endif

/* ---- table-driven log (round 6) ----
 * log(x) for a NORMAL positive x (the caller proves it: e.g. a probability clipped into
 * [1e-7, 1 - 1e-7]) in 21 operations where the atanh-series form above takes 38 -- it has no
 * quotient. x = 2^e m by the high-word normalisation of fdlibm with the base moved from sqrt(2)/2 to
 * 0x3fe6b000 (m in [0.7090, 1.4180)) so that 1.0 is the MIDDLE of its segment; the top seven bits of
 * m's normalised mantissa pick (c, l) = (~1/m, -log c) from include/exmc_logtab.h;
 * r = fma(m, c, -1) (|r| <= 2^-7, abs. error 2^-61); log m = l + log1p(r), log1p by its Taylor
 * series to r^8 (truncation below 2^-66); log x = (e ln2_hi + l) + (e ln2_lo + log1p r). The segment
 * around 1.0, [1 - 2^-9, 1 + 2^-8), has c = 1, l = 0: log keeps its relative accuracy as x -> 1.
 * Accuracy (tests/test_detmath_logtab.py, against the long double logarithm): <= 1.6 ulp where
 * |log x| >= 0.02, <= 2.5 ulp in the segments next to the one around 1.
 * A DIFFERENT rounding contract from exmc_log (the two differ in the last bit for about a fifth of
 * the arguments): a model uses one of them throughout, in the kernels and in the checker
 * (logp_logistic: this one). The split / finish halves are separate so that a kernel can fetch
 * (c, l) from wherever it keeps the table (LDS, or global memory). */
#include "exmc_logtab.h"
EXMC_HD int exmc_logtab_split(double x, double* m_out, int* e_out) {
  const uint64_t ix = exmc_to_bits(x);
  uint32_t hx = (uint32_t)(ix >> 32);
  hx += 0x3FF00000u - EXMC_LOGTAB_BASE;
  *e_out = (int)(hx >> 20) - 0x3FF;
  const int i = (int)((hx >> 13) & 0x7Fu);
  hx = (hx & 0x000FFFFFu) + EXMC_LOGTAB_BASE;
  *m_out = exmc_from_bits(((uint64_t)hx << 32) | (ix & 0xFFFFFFFFull));
  return i;
}
EXMC_HD double exmc_logtab_finish(double m, int e, double c, double l) {
  const double r = __builtin_fma(m, c, -1.0);
  double q = __builtin_fma(r, -0.125, 0x1.2492492492492p-3);   /* -1/8, 1/7 */
  q = __builtin_fma(q, r, -0x1.5555555555555p-3);              /* -1/6 */
  q = __builtin_fma(q, r, 0x1.999999999999ap-3);               /* 1/5 */
  q = __builtin_fma(q, r, -0.25);
  q = __builtin_fma(q, r, 0x1.5555555555555p-2);               /* 1/3 */
  q = __builtin_fma(q, r, -0.5);
  const double r2 = r * r;
  const double lp = __builtin_fma(q, r2, r);                   /* log1p(r) */
  const double dk = (double)e;
  const double hi = __builtin_fma(dk, 6.93147180369123816490e-01, l);   /* e ln2_hi is exact */
  const double lo = __builtin_fma(dk, 1.90821492927058770002e-10, lp);
  return hi + lo;
}
#if defined(__HIP_DEVICE_COMPILE__)
/* exmc_logtab_finish with the series as one asm block, its coefficients in scalar registers (the _s
 * spelling of the cores above: left to the compiler every Horner step with a non-inline constant is a
 * v_mov_b64 + v_fmac_f64 pair). Same operations in the same order: same bits. */
static __device__ __forceinline__ double exmc_logtab_finish_s(double m, int e, double c, double l) {
  const double r = __builtin_fma(m, c, -1.0);
  double q, r2, lp;
  __asm__("v_fma_f64 %0, %3, %4, %5\n\t"
          "v_fma_f64 %0, %0, %3, %6\n\t"
          "v_fma_f64 %0, %0, %3, %7\n\t"
          "v_fma_f64 %0, %0, %3, %8\n\t"
          "v_fma_f64 %0, %0, %3, %9\n\t"
          "v_fma_f64 %0, %0, %3, -0.5\n\t"
          "v_mul_f64 %1, %3, %3\n\t"
          "v_fma_f64 %2, %0, %1, %3"
          : "=&v"(q), "=&v"(r2), "=&v"(lp)
          : "v"(r), "s"(-0.125), "v"(0x1.2492492492492p-3), "s"(-0x1.5555555555555p-3), "s"(0x1.999999999999ap-3),
            "s"(-0.25), "s"(0x1.5555555555555p-2));
  const double dk = (double)e;
  const double hi = __builtin_fma(dk, 6.93147180369123816490e-01, l);
  const double lo = __builtin_fma(dk, 1.90821492927058770002e-10, lp);
  return hi + lo;
}
#elif defined(__HIPCC__)
static __device__ __forceinline__ double exmc_logtab_finish_s(double m, int e, double c, double l) {
  return exmc_logtab_finish(m, e, c, l);
}
#endif
/* the host's function (a hipcc device pass parses host code too, so it is declared there as well) */
static const double EXMC_LOGTAB_HOST[2 * EXMC_LOGTAB_ENTRIES] = {EXMC_LOGTAB_VALUES};
#if defined(__HIPCC__)
static __host__ __forceinline__
#else
static inline
#endif
double exmc_log_tab(double x) {
  double m;
  int e;
  const int i = exmc_logtab_split(x, &m, &e);
  return exmc_logtab_finish(m, e, EXMC_LOGTAB_HOST[2 * i], EXMC_LOGTAB_HOST[2 * i + 1]);
}

#endif /* EXMC_DETMATH_H */
