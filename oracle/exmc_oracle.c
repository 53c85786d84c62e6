/* exmc_oracle.c — CPU restatement of eXMC's NUTS hot path. TEST INFRASTRUCTURE ONLY.
 * See exmc_oracle.h for the scope, the "parity unpinned" statement and the two numeric modes.
 * Citations are path:line under /root/reference (read as text; nothing is copied or executed).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no SIMD intrinsics, scalar f64).
 */
#include "exmc_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/exmc_detmath.h"
#include "../include/exmc_zig_tables.h"

#define M58 ((1ULL << 58) - 1)

/* ======================================================================================
 * RNG: OTP :rand `exsss` = Xorshift116** on two 58-bit words (third-party, un-vendored;
 * restated from the published algorithm — PARITY UNPINNED).
 * ==================================================================================== */

uint64_t exo_splitmix64(uint64_t* x) {
  /* SplitMix64 (Steele, Lea, Flood 2014); OTP seeds exs* generators with it. */
  uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

void exo_rng_seed(exo_rng* r, uint64_t seed) {
  /* :rand.seed_s(:exsss, Integer) — two successive SplitMix64 words masked to 58 bits,
   * redrawn when zero; first word is the list head (call site sampler.ex:154,1055,1087). */
  uint64_t x = seed;
  uint64_t w[2];
  for (int i = 0; i < 2; i++) {
    uint64_t z;
    do {
      z = exo_splitmix64(&x) & M58;
    } while (z == 0);
    w[i] = z;
  }
  r->a = w[0];
  r->b = w[1];
}

static inline uint64_t rotl58(uint64_t x, int n) { return ((x << n) & M58) | (x >> (58 - n)); }

uint64_t exo_rng_next(exo_rng* r) {
  /* state [S1|S0]: output = starstar scrambler of S0; S1 is the word that is shifted. */
  uint64_t s1 = r->a, s0 = r->b;
  uint64_t v1 = (s0 + ((s0 << 2) & M58)) & M58;           /* S0 * 5 mod 2^58 */
  uint64_t v2 = rotl58(v1, 7);
  uint64_t out = (v2 + ((v2 << 3) & M58)) & M58;          /* * 9 mod 2^58 */
  uint64_t s1b = s1 ^ ((s1 << 24) & M58);
  uint64_t nw = s1b ^ s0 ^ (s1b >> 11) ^ (s0 >> 41);
  r->a = s0;
  r->b = nw;
  return out;
}

double exo_rng_uniform(exo_rng* r) {
  /* uniform_s: 53 high bits of the 58-bit word, [0,1). */
  return (double)(exo_rng_next(r) >> 5) * 0x1p-53;
}

static const uint64_t ZIG_KI[256] = EXMC_ZIG_KI_INIT;
static const double ZIG_WI[256] = EXMC_ZIG_WI_INIT;
static const double ZIG_FI[256] = EXMC_ZIG_FI_INIT;

double exo_exp(double x, int mm) { return mm ? exmc_exp(x) : exp(x); }
double exo_log(double x, int mm) { return mm ? exmc_log(x) : log(x); }
double exo_log1p(double x, int mm) { return mm ? exmc_log1p(x) : log1p(x); }
double exo_det_exp(double x) { return exmc_exp(x); }
double exo_det_log(double x) { return exmc_log(x); }
double exo_det_log1p(double x) { return exmc_log1p(x); }
double exo_det_erf(double x) { return exmc_erf(x); }

double exo_rng_normal(exo_rng* r, int mm) {
  /* normal_s: 256-layer ziggurat on one 58-bit word: bit 6 = sign, bits 7..57 = 51-bit R,
   * low 8 bits of R = layer. Fast accept R < KI; layer 0 = tail; else wedge test. */
  for (;;) {
    uint64_t w = exo_rng_next(r);
    int sign = (int)((w >> 6) & 1);
    uint64_t R = w >> 7;
    int idx = (int)(R & 255);
    double x = (double)R * ZIG_WI[idx];
    if (R < ZIG_KI[idx]) return sign ? -x : x;
    if (sign) x = -x;
    if (idx == 0) {
      for (;;) {
        double u0 = exo_rng_uniform(r);
        double xt = (-(1.0 / EXMC_NOR_R)) * exo_log(u0, mm);
        double u1 = exo_rng_uniform(r);
        double y = -exo_log(u1, mm);
        if (y + y > xt * xt) return sign ? (-EXMC_NOR_R - xt) : (EXMC_NOR_R + xt);
      }
    }
    double fi2 = ZIG_FI[idx];
    double u0 = exo_rng_uniform(r);
    if ((ZIG_FI[idx - 1] - fi2) * u0 + fi2 < exo_exp(-0.5 * x * x, mm)) return x;
  }
}

/* rand_xoshiro 0.6.0: Xoshiro256StarStar::seed_from_u64 fills the state with SplitMix64;
 * rand 0.8.5 Standard f64 = (next_u64 >> 11) * 2^-53. (lib.rs:96,137,262,385) */
void exo_xoshiro_seed_from_u64(uint64_t s[4], uint64_t seed) {
  uint64_t x = seed;
  for (int i = 0; i < 4; i++) s[i] = exo_splitmix64(&x);
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t exo_xoshiro_next(uint64_t s[4]) {
  uint64_t result = rotl64(s[1] * 5, 7) * 9;
  uint64_t t = s[1] << 17;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl64(s[3], 45);
  return result;
}
double exo_xoshiro_f64(uint64_t s[4]) { return (double)(exo_xoshiro_next(s) >> 11) * 0x1p-53; }

/* ======================================================================================
 * Reductions. G = 1: left-to-right from 0.0 (tree.ex:1583-1588, Nx.sum on BinaryBackend).
 * G > 1: lane l owns slots l, l+G, ...; lane partials then xor-butterfly. `init0` seeds
 * lane 0's partial (used so that G = 1 reproduces compiler.ex:396-397's term fold).
 * ==================================================================================== */
static double lane_sum(const double* v, int n, int G, double init0) {
  /* A 16-lane chain group that holds one dimension per lane and at most 12 of them sums in lane
   * order on the GPU (v_fmac_f64_dpp row_newbcast chains, exmc_device.hpp kSeqSum): that is the
   * left-to-right sum seeded with init0, i.e. the reference's own order. */
  if (G == 16 && n <= 12) G = 1;
  if (G <= 1) {
    double acc = init0;
    for (int i = 0; i < n; i++) acc = acc + v[i];
    return acc;
  }
  double part[64];
  for (int l = 0; l < G; l++) {
    double acc = (l == 0) ? init0 : 0.0;
    for (int i = l; i < n; i += G) acc = acc + v[i];
    part[l] = acc;
  }
  for (int m = 1; m < G; m <<= 1) {
    double nxt[64];
    for (int l = 0; l < G; l++) nxt[l] = part[l] + part[l ^ m];
    memcpy(part, nxt, sizeof(double) * G);
  }
  return part[0];
}

/* ======================================================================================
 * Distributions (lib/exmc/dist/<name>.ex) with the reference's f32-rounded literals:
 * Nx.tensor(<float>) defaults to f32, so e.g. log(2*pi) is computed and stored in f32.
 * ==================================================================================== */
static double f32r(double x) { return (double)(float)x; }
static double LOG_2PI_F32(void) { return f32r(log(f32r(2.0 * M_PI))); }       /* normal.ex:19,22 */
static double LOG_2_OVER_PI_F32(void) { return f32r(log(2.0 / M_PI)); }      /* half_cauchy.ex:22 */
static double PI_F32(void) { return f32r(M_PI); }                            /* student_t.ex:27 */
static double TINY_F32(void) { return f32r(1.0e-30); }                       /* Nx.tensor(1.0e-30) */

double exo_dist_normal(double x, double mu, double sigma, int mm) {
  /* normal.ex:15-24 */
  double ss = fmax(sigma, TINY_F32());
  double z = (x - mu) / ss;
  double log_term = LOG_2PI_F32() + 2.0 * exo_log(ss, mm);
  return -0.5 * (z * z + log_term);
}
double exo_dist_half_cauchy(double x, double scale, int mm) {
  /* half_cauchy.ex:17-25 */
  double ss = fmax(scale, TINY_F32());
  double z = x / ss;
  return (LOG_2_OVER_PI_F32() - exo_log(ss, mm)) - exo_log(1.0 + z * z, mm);
}
double exo_dist_exponential(double x, double lambda, int mm) {
  /* exponential.ex:15-17 */
  return exo_log(lambda, mm) - lambda * x;
}
double exo_dist_half_normal(double x, double sigma, int mm) {
  /* half_normal.ex:15-22: -0.5*(z^2 + log(2pi)) + (log(2) - log(sigma)), literals f32 */
  double ss = fmax(sigma, TINY_F32());
  double z = x / ss;
  double base = -0.5 * (z * z + LOG_2PI_F32());
  return base + (f32r(log(2.0)) - exo_log(ss, mm));
}
double exo_dist_bernoulli(double x, double p, int mm) {
  /* bernoulli.ex:17-27: p clipped to [1e-7, 1-1e-7] */
  double lo = f32r(1.0e-7), hi = 1.0 - f32r(1.0e-7);
  double pc = fmin(fmax(p, lo), hi);
  return x * exo_log(pc, mm) + (1.0 - x) * exo_log(1.0 - pc, mm);
}

static const double LANCZOS[9] = {0.99999999999980993,  676.5203681218851,     -1259.1392167224028,
                                  771.32342877765313,   -176.61502916214059,   12.507343278686905,
                                  -0.13857109526572012, 9.9843695780195716e-6, 1.5056327351493116e-7};

/* math.ex:27-52; value and d/dx of the same expression (coefficients are f32 tensors there) */
static double lanczos_val_d(double x, int mm, double* dx) {
  double half_log_2pi = f32r(0.5 * log(2.0 * M_PI));
  double t = x + 6.5;
  double ag = f32r(LANCZOS[0]);
  double dag = 0.0;
  for (int i = 1; i < 9; i++) {
    double c = f32r(LANCZOS[i]);
    double den = x + (double)(i - 1) * 1.0;
    double term = c / den;
    ag = ag + term;
    dag = dag - term / den;
  }
  double lt = exo_log(t, mm);
  double val = ((half_log_2pi + (x - 0.5) * lt) - t) + exo_log(ag, mm);
  if (dx) *dx = ((lt + (x - 0.5) / t) - 1.0) + dag / ag;
  return val;
}
double exo_lgamma_lanczos(double x, int mm) { return lanczos_val_d(x, mm, 0); }

double exo_dist_student_t(double x, double df, double loc, double scale, int mm) {
  /* student_t.ex:15-29 */
  double ss = fmax(scale, TINY_F32());
  double sdf = fmax(df, TINY_F32());
  double z = (x - loc) / ss;
  double z2 = z * z;
  double hp1 = (sdf + 1.0) / 2.0;
  double h = sdf / 2.0;
  double r = exo_lgamma_lanczos(hp1, mm) - exo_lgamma_lanczos(h, mm);
  r = r - 0.5 * exo_log(sdf * PI_F32(), mm);
  r = r - exo_log(ss, mm);
  r = r - hp1 * exo_log(1.0 + z2 / sdf, mm);
  return r;
}

/* ======================================================================================
 * Models: log-density + gradient in unconstrained space, "kernel order" (see DESIGN.md).
 * Gradients are hand-derived (the reference uses Nx.Defn reverse-mode AD, compiler.ex:131-141)
 * and are checked against central differences in tests, as the reference's tests do.
 * ==================================================================================== */
struct exo_model {
  int kind, d, n;
  double* data;     /* model-specific */
  double* aux;      /* derived constants */
  exo_custom_fn custom;
  /* flat[r] = kernel dimension of the r-th entry of the reference's flat vector: PointMap.build
   * sorts the free RVs by id as strings (point_map.ex:30-60) and init_position /
   * sample_momentum_fast consume one normal_s per flat entry in that order (sampler.ex:339-349,
   * 393-403). Identity when the kernel order already is the sorted one. */
  int flat[EXO_MAX_D];
};

/* ids of a kind whose names the kind fixes, sorted as the reference sorts them (byte order) */
static int cmp_names(const void* a, const void* b) {
  return strcmp(((const char* const*)a)[0], ((const char* const*)b)[0]);
}
static void flat_from_names(exo_model* m, char (*names)[24]) {
  const char* ptr[EXO_MAX_D];
  for (int i = 0; i < m->d; i++) ptr[i] = names[i];
  qsort(ptr, (size_t)m->d, sizeof(ptr[0]), cmp_names);
  for (int r = 0; r < m->d; r++) m->flat[r] = (int)((ptr[r] - names[0]) / 24);
}
static void default_flat_order(exo_model* m) {
  char names[EXO_MAX_D][24];
  for (int i = 0; i < EXO_MAX_D; i++) m->flat[i] = i;
  if (m->kind == EXO_MODEL_SV) {
    /* kernel order s_1..s_T, sigma, nu (STANDARD_BENCHMARKS.md:51-61 names) */
    int T = m->d - 2;
    for (int t = 0; t < T; t++) snprintf(names[t], 24, "s_%d", t + 1);
    snprintf(names[T], 24, "sigma");
    snprintf(names[T + 1], 24, "nu");
    flat_from_names(m, names);
  } else if (m->kind == EXO_MODEL_LOGISTIC) {
    /* kernel order alpha, beta_1..beta_K */
    snprintf(names[0], 24, "alpha");
    for (int j = 1; j < m->d; j++) snprintf(names[j], 24, "beta_%d", j);
    flat_from_names(m, names);
  }
  /* radon: the county order of the kernel layout depends on the data (descending county size);
   * its caller passes the order with exo_model_set_flat_order. eight_schools (mu, tau,
   * theta_trans_0..7) and simple (mu, sigma) are sorted already. */
}

exo_model* exo_model_create(int kind, int d, const double* data, int n_data) {
  exo_model* m = (exo_model*)calloc(1, sizeof(exo_model));
  m->kind = kind;
  m->d = d;
  m->n = n_data;
  if (n_data > 0) {
    m->data = (double*)malloc(sizeof(double) * n_data);
    memcpy(m->data, data, sizeof(double) * n_data);
  }
  switch (kind) {
    case EXO_MODEL_STD_NORMAL:
      break;
    case EXO_MODEL_SIMPLE:
      m->d = 2;
      break;
    case EXO_MODEL_EIGHT_SCHOOLS: {
      /* data = y[8], sigma[8]; aux = log(sigma_j) (validate_posteriordb.exs:291: Nx.log(s_j),
       * a constant of the data; libm on the host in both modes) */
      m->d = 10;
      m->aux = (double*)malloc(sizeof(double) * 8);
      for (int j = 0; j < 8; j++) m->aux[j] = log(m->data[8 + j]);
      break;
    }
    case EXO_MODEL_SV:
      m->d = n_data + 2; /* data = r[T] */
      break;
    case EXO_MODEL_LOGISTIC:
      /* data = X[N][K] row-major, y[N]; d = K + 1 */
      if (d < 2 || n_data % d != 0) { exo_model_free(m); return 0; }
      m->d = d;
      break;
    case EXO_MODEL_CUSTOM:
      m->d = d;
      break;
    case EXO_MODEL_RADON:
      /* data = u[J], county_start[J+1], floor[N], y[N] (observations sorted by county); d = J+5 */
      if (d < 6 || (n_data - (2 * (d - 5) + 1)) % 2 != 0) { exo_model_free(m); return 0; }
      m->d = d;
      break;
    default:
      break;
  }
  if (m->d > EXO_MAX_D) {
    exo_model_free(m);
    return 0;
  }
  default_flat_order(m);
  return m;
}
int exo_model_set_flat_order(exo_model* m, const int* flat) {
  char seen[EXO_MAX_D];
  memset(seen, 0, sizeof(seen));
  for (int r = 0; r < m->d; r++) {
    if (flat[r] < 0 || flat[r] >= m->d || seen[flat[r]]) return -1;   /* not a permutation */
    seen[flat[r]] = 1;
  }
  for (int r = 0; r < m->d; r++) m->flat[r] = flat[r];
  return 0;
}
void exo_model_get_flat_order(const exo_model* m, int* flat) {
  for (int r = 0; r < m->d; r++) flat[r] = m->flat[r];
}
void exo_model_free(exo_model* m) {
  if (!m) return;
  free(m->data);
  free(m->aux);
  free(m);
}
int exo_model_dim(const exo_model* m) { return m->d; }
void exo_model_set_custom(exo_model* m, exo_custom_fn fn) { m->custom = fn; }

static double clamp200(double z) { return fmax(-200.0, fmin(z, 200.0)); } /* transform.ex:17-29 */

static double logp_std_normal(const exo_model* m, const double* q, double* g, exo_cfg c) {
  double T[EXO_MAX_D];
  double c1 = LOG_2PI_F32() + 2.0 * 0.0; /* log(1.0) = 0 */
  for (int i = 0; i < m->d; i++) {
    T[i] = -0.5 * (q[i] * q[i] + c1);
    g[i] = -q[i];
  }
  return lane_sum(T, m->d, c.lanes, 0.0);
}

static double logp_simple(const exo_model* m, const double* q, double* g, exo_cfg c) {
  /* build-defined: mu ~ N(0,5); sigma ~ Exponential(1) [:log]; y_i ~ N(mu, sigma);
   * observations are f32 literals (README.md:72-74). Term fold: mu, sigma, obs. */
  int mm = c.math_mode;
  double mu = q[0], zc = clamp200(q[1]);
  double sigma = exo_exp(zc, mm);
  double ss = fmax(sigma, TINY_F32());
  double ls = exo_log(ss, mm);
  double zmu = (mu - 0.0) / 5.0;
  double t_mu = -0.5 * (zmu * zmu + (LOG_2PI_F32() + 2.0 * log(5.0)));
  double t_sig = (0.0 - 1.0 * sigma) + zc; /* log(1) - lambda*x + log|J| */
  double cn = LOG_2PI_F32() + 2.0 * ls;
  double ll[64], a[64], b[64];
  int n = m->n;
  for (int i = 0; i < n; i++) {
    double z = (m->data[i] - mu) / ss;
    ll[i] = -0.5 * (z * z + cn);
    a[i] = z / ss;          /* d ll / d mu */
    b[i] = z * z - 1.0;     /* d ll / d log sigma */
  }
  double obs = lane_sum(ll, n, 1, 0.0);
  double sa = lane_sum(a, n, 1, 0.0);
  double sb = lane_sum(b, n, 1, 0.0);
  int in = (q[1] > -200.0) && (q[1] < 200.0);
  g[0] = (-(zmu / 5.0)) + sa;
  g[1] = in ? ((sb - sigma) + 1.0) : 0.0;
  return (t_mu + t_sig) + obs;
}

static double logp_eight_schools(const exo_model* m, const double* q, double* g, exo_cfg c) {
  /* validate_posteriordb.exs:246-324; order mu, tau, theta_trans_0..7 (point_map.ex:37).
   * Term fold (compiler.ex:174-178,396-397; node keys sorted): lik_obs, mu, tau, theta_trans_j. */
  int mm = c.math_mode;
  const double* y = m->data;
  const double* sg = m->data + 8;
  const double* lsg = m->aux;
  double mu = q[0], zc = clamp200(q[1]);
  double tau = exo_exp(zc, mm);
  double L[10], A[10], B[10], T[10];
  L[0] = L[1] = A[0] = A[1] = B[0] = B[1] = 0.0;
  double c1 = LOG_2PI_F32() + 2.0 * 0.0;
  for (int j = 0; j < 8; j++) {
    double th = q[2 + j];
    double theta = mu + tau * th;
    double z = (y[j] - theta) / sg[j];
    L[2 + j] = (-0.5 * (z * z)) - lsg[j];
    double a = z / sg[j];
    A[2 + j] = a;
    B[2 + j] = a * th;
    T[2 + j] = -0.5 * (th * th + c1);
    g[2 + j] = (-th) + a * tau;
  }
  double lik = lane_sum(L, 10, c.lanes, 0.0);
  double sa = lane_sum(A, 10, c.lanes, 0.0);
  double sb = lane_sum(B, 10, c.lanes, 0.0);
  double zmu = (mu - 0.0) / 5.0;
  T[0] = -0.5 * (zmu * zmu + (LOG_2PI_F32() + 2.0 * log(5.0)));
  double zt = tau / 5.0;
  double zt2 = zt * zt;
  T[1] = ((LOG_2_OVER_PI_F32() - log(5.0)) - exo_log(1.0 + zt2, mm)) + zc;
  g[0] = (-(zmu / 5.0)) + sa;
  double dhc = -(((2.0 * zt) / 5.0) / (1.0 + zt2));
  int in = (q[1] > -200.0) && (q[1] < 200.0);
  g[1] = in ? ((dhc + sb) * tau + 1.0) : 0.0;
  return lane_sum(T, 10, c.lanes, lik);
}

static double logp_sv(const exo_model* m, const double* q, double* g, exo_cfg c) {
  /* STANDARD_BENCHMARKS.md:51-61. Kernel order: s_1..s_T, log sigma, log nu (the reference's
   * flat layout is the string sort nu, s_1, s_10, s_100, s_11, ..., sigma: exo_flat_index).
   * sigma ~ Exponential(50), nu ~ Exponential(0.1) (f32 literals), s_1 ~ N(0,sigma),
   * s_t ~ N(s_{t-1}, sigma), r_t ~ StudentT(nu, 0, exp(s_t)) with log(scale) taken as s_t. */
  int mm = c.math_mode;
  int T = m->n;
  const double* r = m->data;
  double zs = clamp200(q[T]), zn = clamp200(q[T + 1]);
  double sigma = exo_exp(zs, mm), nu = exo_exp(zn, mm);
  double ss = fmax(sigma, TINY_F32());
  double sdf = fmax(nu, TINY_F32());
  double lam_s = 50.0, lam_n = f32r(0.1);
  double t_sigma = (f32r(log(lam_s)) - lam_s * sigma) + zs;
  double t_nu = (f32r(log(lam_n)) - lam_n * nu) + zn;
  double hp1 = (sdf + 1.0) / 2.0, h = sdf / 2.0;
  double d1, d0;
  double lg1 = lanczos_val_d(hp1, mm, &d1), lg0 = lanczos_val_d(h, mm, &d0);
  double An = (lg1 - lg0) - 0.5 * exo_log(sdf * PI_F32(), mm);
  double dAn = (0.5 * d1 - 0.5 * d0) - 0.5 / sdf;
  double cn = LOG_2PI_F32() + 2.0 * exo_log(ss, mm);
  double P[EXO_MAX_D], LL[EXO_MAX_D], E2[EXO_MAX_D], DN[EXO_MAX_D], de[EXO_MAX_D];
  for (int t = 0; t < T; t++) {
    double prev = (t == 0) ? 0.0 : q[t - 1];
    double e = (q[t] - prev) / ss;
    P[t] = -0.5 * (e * e + cn);
    E2[t] = e * e - 1.0;
    de[t] = -(e / ss);                       /* dP_t / d s_t */
    double z = r[t] * exo_exp(-q[t], mm);
    double w = (z * z) / sdf;
    double l = exo_log(1.0 + w, mm);
    double wr = w / (1.0 + w);
    LL[t] = (An - q[t]) - hp1 * l;
    DN[t] = (dAn - 0.5 * l) + (hp1 * wr) / sdf;
    g[t] = -1.0 + (sdf + 1.0) * wr;          /* likelihood part; prior added below */
  }
  for (int t = 0; t < T; t++) {
    double nxt = (t + 1 < T) ? de[t + 1] : 0.0;
    g[t] = g[t] + (de[t] - nxt);
  }
  P[T] = P[T + 1] = LL[T] = LL[T + 1] = E2[T] = E2[T + 1] = DN[T] = DN[T + 1] = 0.0;
  int n = T + 2;
  double sp = lane_sum(P, n, c.lanes, 0.0);
  double sl = lane_sum(LL, n, c.lanes, 0.0);
  double se = lane_sum(E2, n, c.lanes, 0.0);
  double sn = lane_sum(DN, n, c.lanes, 0.0);
  int in_s = (q[T] > -200.0) && (q[T] < 200.0);
  int in_n = (q[T + 1] > -200.0) && (q[T + 1] < 200.0);
  g[T] = in_s ? ((se - lam_s * sigma) + 1.0) : 0.0;
  g[T + 1] = in_n ? ((sn * nu - lam_n * nu) + 1.0) : 0.0;
  return ((t_sigma + t_nu) + sp) + sl;
}

/* multiply-accumulate of the dense contractions: fused in the GPU contract, mul + add in the
 * reference's arithmetic (Nx.dot on BinaryBackend) */
static double mac(double a, double b, double acc, int mm) {
  return mm ? __builtin_fma(a, b, acc) : (a * b + acc);
}

static double logp_logistic(const exo_model* m, const double* q, double* g, exo_cfg c) {
  /* STANDARD_BENCHMARKS.md:41-49: alpha, beta_j ~ N(0,10); y_n ~ Bernoulli(sigmoid(alpha + X beta))
   * with p clipped to [1e-7, 1-1e-7] (bernoulli.ex:17-27). Kernel order alpha, beta_1..beta_K.
   * Observation n belongs to lane n mod G; each lane walks its observations in increasing n. */
  int mm = c.math_mode, d = m->d, K = d - 1, N = m->n / d;
  int G = c.lanes < 1 ? 1 : c.lanes;
  const double* X = m->data;
  const double* y = m->data + (size_t)N * K;
  double lo = f32r(1.0e-7), hi = 1.0 - f32r(1.0e-7);
  double lik_part[64], gp[64][EXO_MAX_D];
  /* G == 4 is the matrix-core layout (16 chains per wavefront, v_mfma_f64_16x16x4_f64 = fma chain
   * over k): the gradient contraction X^T r runs over ALL observations in increasing n in one
   * chain; the log-likelihood keeps four row-group partials (n mod 4) + butterfly. */
  int mfma = (G == 4);
  double* rbuf = (double*)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
  for (int l = 0; l < G; l++) {
    double lik = 0.0;
    double* acc = gp[l];
    for (int j = 0; j < d; j++) acc[j] = 0.0;
    for (int n = l; n < N; n += G) {
      const double* x = X + (size_t)n * K;
      double eta = mac(1.0, q[0], 0.0, mm);
      for (int j = 0; j < K; j++) eta = mac(x[j], q[1 + j], eta, mm);
      double p = 1.0 / (1.0 + exo_exp(-eta, mm));
      double pc = fmin(fmax(p, lo), hi);
      /* deterministic mode: the table-driven logarithm of include/exmc_detmath.h (exmc_log_tab; round 6 --
       * the argument is a probability clipped into [1e-7, 1 - 1e-7]: normal and positive), which is what
       * every layout of the kernels evaluates here; libm mode: the reference's own log */
      double la = (y[n] == 1.0) ? pc : (1.0 - pc);
      double ll = mm ? exmc_log_tab(la) : log(la);
      double r = (p > lo && p < hi) ? (y[n] - p) : 0.0;
      lik = lik + ll;
      rbuf[n] = r;
      if (!mfma) {
        acc[0] = mac(1.0, r, acc[0], mm);
        for (int j = 0; j < K; j++) acc[1 + j] = mac(x[j], r, acc[1 + j], mm);
      }
    }
    lik_part[l] = lik;
  }
  if (mfma) {
    double* acc = gp[0];
    for (int n = 0; n < N; n++) {
      const double* x = X + (size_t)n * K;
      acc[0] = mac(1.0, rbuf[n], acc[0], mm);
      for (int j = 0; j < K; j++) acc[1 + j] = mac(x[j], rbuf[n], acc[1 + j], mm);
    }
  }
  free(rbuf);
  /* butterflies over the lane partials (lane_sum with one slot per lane) */
  double lik = lane_sum(lik_part, G, G, 0.0);
  double T[EXO_MAX_D];
  double c10 = LOG_2PI_F32() + 2.0 * log(10.0);
  for (int j = 0; j < d; j++) {
    double col[64];
    for (int l = 0; l < G; l++) col[l] = gp[l][j];
    double gj = mfma ? gp[0][j] : lane_sum(col, G, G, 0.0);
    double z = (q[j] - 0.0) / 10.0;
    T[j] = -0.5 * (z * z + c10);
    g[j] = (-(z / 10.0)) + gj;
  }
  return lane_sum(T, d, G, lik);
}

static double half_cauchy_d(double x, double s, double* dx, int mm) {
  /* half_cauchy.ex:17-25 value and d/dx */
  double z = x / s;
  double z2 = z * z;
  *dx = -(((2.0 * z) / s) / (1.0 + z2));
  return (LOG_2_OVER_PI_F32() - log(s)) - exo_log(1.0 + z2, mm);
}

static double logp_radon(const exo_model* m, const double* q, double* g, exo_cfg c) {
  /* notebooks/09_radon_bhm.livemd "The Radon Model": alpha_raw_j ~ N(0,1), mu_alpha ~ N(0,10),
   * gamma_u ~ N(0,5), sigma_alpha, sigma_y ~ HalfCauchy(2.5) [:log], beta ~ N(0,5),
   * alpha_j = mu_alpha + gamma_u*u_j + sigma_alpha*alpha_raw_j, y ~ N(alpha_j + beta*floor, sigma_y).
   * Kernel order: alpha_raw_0..J-1, mu_alpha, gamma_u, log sigma_alpha, log sigma_y, beta. */
  int mm = c.math_mode, d = m->d, J = d - 5;
  int G = c.lanes < 1 ? 1 : c.lanes;
  const double* u = m->data;
  const double* cs = m->data + J;           /* county_start[J+1] */
  int N = (int)cs[J];
  const double* fl = m->data + J + (J + 1);
  const double* y = fl + N;
  double mu = q[J], gam = q[J + 1], zsa = clamp200(q[J + 2]), zsy = clamp200(q[J + 3]), beta = q[J + 4];
  double sa = exo_exp(zsa, mm), sy = exo_exp(zsy, mm);
  double ssy = fmax(sy, TINY_F32());
  double cn = LOG_2PI_F32() + 2.0 * exo_log(ssy, mm);
  double c1 = LOG_2PI_F32() + 2.0 * 0.0;
  double rinv = 1.0 / ssy;   /* deterministic mode: the one reciprocal of a leapfrog */
  double LIK[EXO_MAX_D], S[EXO_MAX_D], SU[EXO_MAX_D], SA[EXO_MAX_D], F[EXO_MAX_D], Z2[EXO_MAX_D];
  double T[EXO_MAX_D];
  for (int j = 0; j < d; j++) LIK[j] = S[j] = SU[j] = SA[j] = F[j] = Z2[j] = T[j] = 0.0;
  /* 64 lanes per chain (exmc_models.hpp Radon<64>): observation i sits on lane i mod 64 in slot
   * i / 64 whatever its county, so the likelihood, floor and z^2 totals are each lane's terms in
   * slot order and then the lanes; a county's sum of a_i = z_i / sigma_y is its observations in
   * index order from 0.0 (its owner lane adds them up from the wavefront's LDS strip). Other
   * layouts: the per-county totals over the lanes that own the counties. */
  double* OL = NULL; double* OF = NULL; double* OZ = NULL;
  if (G == 64) {
    OL = (double*)malloc(sizeof(double) * 3 * (size_t)(N > 0 ? N : 1));
    OF = OL + (N > 0 ? N : 1);
    OZ = OF + (N > 0 ? N : 1);
  }
  for (int j = 0; j < J; j++) {
    double ar = q[j];
    double alpha = (mu + gam * u[j]) + sa * ar;
    int i0 = (int)cs[j], i1 = (int)cs[j + 1];
    double lik = 0.0, s = 0.0, f = 0.0, z2s = 0.0;
    for (int i = i0; i < i1; i++) {
      if (mm) {
        /* deterministic mode = the kernels' unit arithmetic since round 5 (exmc_models.hpp Radon::eval,
         * taken from the generated radon): fused multiply-adds and the two quotients by sigma_y as
         * products with one correctly rounded reciprocal. 64 lanes: the three totals are accumulated
         * per lane in slot order (below), so only the per-observation values are kept here. */
        double mean = __builtin_fma(beta, fl[i], alpha);
        double z = (y[i] - mean) * rinv;
        double a = z * rinv;
        double t = __builtin_fma(z, z, cn);
        double zz1 = __builtin_fma(z, z, -1.0);
        lik = __builtin_fma(-0.5, t, lik);
        s = s + a;
        f = __builtin_fma(a, fl[i], f);
        z2s = z2s + zz1;
        if (G == 64) { OL[i] = t; OF[i] = a; OZ[i] = zz1; }
        continue;
      }
      double mean = alpha + beta * fl[i];
      double z = (y[i] - mean) / ssy;
      double a = z / ssy;
      lik = lik + (-0.5 * (z * z + cn));
      s = s + a;
      f = f + a * fl[i];
      z2s = z2s + (z * z - 1.0);
      if (G == 64) { OL[i] = -0.5 * (z * z + cn); OF[i] = a * fl[i]; OZ[i] = z * z - 1.0; }
    }
    LIK[j] = lik; S[j] = s; SU[j] = s * u[j]; SA[j] = s * ar;
    F[j] = f; Z2[j] = z2s;
    T[j] = -0.5 * (ar * ar + c1);
    g[j] = (-ar) + s * sa;
  }
  double lik, sf, sz2;
  if (G == 64 && mm) {
    /* lane l, slots in order: lik = fma(-0.5, t_i, lik), f = fma(a_i, floor_i, f), z2s = z2s + (z_i^2 - 1);
     * then the butterfly over the 64 lanes */
    double PL[64], PF[64], PZ[64];
    for (int l = 0; l < 64; l++) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      for (int i = l; i < N; i += 64) {
        a0 = __builtin_fma(-0.5, OL[i], a0);
        a1 = __builtin_fma(OF[i], fl[i], a1);
        a2 = a2 + OZ[i];
      }
      PL[l] = a0; PF[l] = a1; PZ[l] = a2;
    }
    lik = lane_sum(PL, 64, 64, 0.0);
    sf = lane_sum(PF, 64, 64, 0.0);
    sz2 = lane_sum(PZ, 64, 64, 0.0);
  } else {
    lik = (G == 64) ? lane_sum(OL, N, 64, 0.0) : lane_sum(LIK, d, G, 0.0);
    sf = (G == 64) ? lane_sum(OF, N, 64, 0.0) : lane_sum(F, d, G, 0.0);
    sz2 = (G == 64) ? lane_sum(OZ, N, 64, 0.0) : lane_sum(Z2, d, G, 0.0);
  }
  double ss = lane_sum(S, d, G, 0.0);
  double su = lane_sum(SU, d, G, 0.0);
  double sar = lane_sum(SA, d, G, 0.0);
  if (OL) free(OL);
  double zmu = (mu - 0.0) / 10.0, zg = (gam - 0.0) / 5.0, zb = (beta - 0.0) / 5.0;
  T[J] = -0.5 * (zmu * zmu + (LOG_2PI_F32() + 2.0 * log(10.0)));
  T[J + 1] = -0.5 * (zg * zg + (LOG_2PI_F32() + 2.0 * log(5.0)));
  double dsa, dsy;
  T[J + 2] = half_cauchy_d(sa, 2.5, &dsa, mm) + zsa;
  T[J + 3] = half_cauchy_d(sy, 2.5, &dsy, mm) + zsy;
  T[J + 4] = -0.5 * (zb * zb + (LOG_2PI_F32() + 2.0 * log(5.0)));
  int in_a = (q[J + 2] > -200.0) && (q[J + 2] < 200.0);
  int in_y = (q[J + 3] > -200.0) && (q[J + 3] < 200.0);
  g[J] = (-(zmu / 10.0)) + ss;
  g[J + 1] = (-(zg / 5.0)) + su;
  g[J + 2] = in_a ? ((dsa + sar) * sa + 1.0) : 0.0;
  g[J + 3] = in_y ? ((dsy * sy + sz2) + 1.0) : 0.0;
  g[J + 4] = (-(zb / 5.0)) + sf;
  return lane_sum(T, d, G, lik);
}

double exo_logp_grad(const exo_model* m, const double* q, double* grad, exo_cfg cfg) {
  switch (m->kind) {
    case EXO_MODEL_LOGISTIC: return logp_logistic(m, q, grad, cfg);
    case EXO_MODEL_RADON: return logp_radon(m, q, grad, cfg);
    case EXO_MODEL_STD_NORMAL: return logp_std_normal(m, q, grad, cfg);
    case EXO_MODEL_SIMPLE: return logp_simple(m, q, grad, cfg);
    case EXO_MODEL_EIGHT_SCHOOLS: return logp_eight_schools(m, q, grad, cfg);
    case EXO_MODEL_SV: return logp_sv(m, q, grad, cfg);
    case EXO_MODEL_CUSTOM: return m->custom ? m->custom(m->data, q, grad) : NAN;
    default: return NAN;
  }
}

void exo_constrain(const exo_model* m, const double* q, double* x) {
  /* sampler.ex:1281-1298: forward transform per PointMap entry */
  memcpy(x, q, sizeof(double) * m->d);
  switch (m->kind) {
    case EXO_MODEL_SIMPLE: x[1] = exp(clamp200(q[1])); break;
    case EXO_MODEL_EIGHT_SCHOOLS: x[1] = exp(clamp200(q[1])); break;
    case EXO_MODEL_SV:
      x[m->n] = exp(clamp200(q[m->n]));
      x[m->n + 1] = exp(clamp200(q[m->n + 1]));
      break;
    case EXO_MODEL_RADON:
      x[m->d - 3] = exp(clamp200(q[m->d - 3]));
      x[m->d - 2] = exp(clamp200(q[m->d - 2]));
      break;
    default: break;
  }
}

/* ======================================================================================
 * Leapfrog (leapfrog.ex:14-51; batched_leapfrog.ex:79-85; compiler.ex:143-170)
 * ==================================================================================== */
/* ---- dense inverse mass (opts[:dense_mass], mass_matrix.ex:27-35,56-72,105-140) ----
 * tl_cov: the row-major d x d covariance (= M^-1, leapfrog.ex:57-61 `Nx.dot(inv_mass, p)`) in
 * force on this thread, or NULL for the diagonal mass the `im` arguments carry. Thread-local, so
 * the chain workers of exo_sample_chains each set their own.
 *
 * NOTE on the reference: with dense_mass: true its tree hands check_uturn_rho the FLATTENED d x d
 * matrix as the per-dimension list (tree.ex:1425,1517 `Nx.to_flat_list(inv_mass_diag)`; the zip in
 * tree.ex:1583-1588 then has no clause for lists of unequal length) and find_reasonable_epsilon
 * draws d*d momenta (sampler.ex:452 `sample_momentum_fast(rng, inv_mass_diag)`), so for d >= 2 the
 * reference raises before the first dense transition completes and no reference test covers the
 * mode. What is restated here is the documented intent (DECISIONS.md 37; Betancourt's criterion
 * rho . (M^-1 p+-) < 0 as the comment at tree.ex:1572-1577 states it): v = M^-1 rho by the dense
 * product, momentum p = L^-T z (sampler.ex:412-427), kinetic energy and position update through
 * M^-1 p (leapfrog.ex:39-61). Products accumulate with fma in ascending index, as the device does.
 *
 * Order: the reference's covariance is that of its flat vector (PointMap order, sampler.ex:682-705
 * feeds Welford the flat q), so cov / chol are indexed by FLAT entries here as well, and every
 * contraction runs in ascending flat index. tl_flat (m->flat, or NULL = identity) maps flat entry r
 * to its kernel dimension; the vectors of this file stay in kernel order. */
static __thread const double* tl_cov = NULL;
static __thread const int* tl_flat = NULL;

static void mass_times(const double* im, const double* x, int d, double* out) {
  if (!tl_cov) {
    for (int i = 0; i < d; i++) out[i] = im[i] * x[i];
    return;
  }
  for (int r = 0; r < d; r++) {
    double acc = 0.0;
    for (int s = 0; s < d; s++) acc = fma(x[tl_flat ? tl_flat[s] : s], tl_cov[(size_t)r * d + s], acc);
    out[tl_flat ? tl_flat[r] : r] = acc;
  }
}

double exo_kinetic_energy(const double* p, const double* im, int d, exo_cfg c) {
  double v[EXO_MAX_D], mp[EXO_MAX_D];
  mass_times(im, p, d, mp);
  for (int i = 0; i < d; i++) v[i] = p[i] * mp[i];
  return 0.5 * lane_sum(v, d, c.lanes, 0.0);
}

double exo_leapfrog(const exo_model* m, double* q, double* p, double* g, double eps,
                    const double* im, double* jlp, exo_cfg c) {
  int d = m->d;
  double h = eps / 2.0;
  double mp[EXO_MAX_D];
  for (int i = 0; i < d; i++) p[i] = p[i] + h * g[i];
  mass_times(im, p, d, mp);
  for (int i = 0; i < d; i++) q[i] = q[i] + eps * mp[i];
  double logp = exo_logp_grad(m, q, g, c);
  for (int i = 0; i < d; i++) p[i] = p[i] + h * g[i];
  if (jlp) *jlp = logp - exo_kinetic_energy(p, im, d, c);
  return logp;
}

void exo_multi_step(const exo_model* m, const double* q0, const double* p0, const double* g0,
                    double eps, const double* im, int n_steps, double* all_q, double* all_p,
                    double* all_logp, double* all_g, exo_cfg c) {
  int d = m->d;
  double q[EXO_MAX_D], p[EXO_MAX_D], g[EXO_MAX_D];
  memcpy(q, q0, sizeof(double) * d);
  memcpy(p, p0, sizeof(double) * d);
  memcpy(g, g0, sizeof(double) * d);
  for (int s = 0; s < n_steps; s++) {
    double lp = exo_leapfrog(m, q, p, g, eps, im, 0, c);
    memcpy(all_q + (size_t)s * d, q, sizeof(double) * d);
    memcpy(all_p + (size_t)s * d, p, sizeof(double) * d);
    memcpy(all_g + (size_t)s * d, g, sizeof(double) * d);
    all_logp[s] = lp;
  }
}

/* B2' -- the fused-chain hook of the speculative path (tree.ex:613-653, dispatch_multi_step / do_dispatch):
 * `leapfrog_chain_normal(q, p, inv_mass, k, signed_eps, mu, sigma)` = K leapfrog steps of ONE chain whose d
 * coordinates are independent Normal(mu, sigma) terms, in one dispatch. The function itself lives in
 * nx_vulkan@718d80a (mix.lock; absent from the reference tree), so what is restated is what its call site
 * fixes: "Output contract is identical in both branches: {all_q, all_p, all_logp, all_grad}" (tree.ex:620-621)
 * = multi_step's rows (batched_leapfrog.ex:50-101: the step of leapfrog.ex:14-31 iterated, RAW logp stored,
 * :87), with the density of dist/normal.ex:15-24 per coordinate. The call passes no gradient (tree.ex:637 binds
 * `_grad`): the chain's first half-kick uses the gradient of the density at q. signed_eps = dir_sign * epsilon
 * (tree.ex:639). Rows [k][d] row-major, logp [k]. */
static double chain_normal_logp_grad(const double* q, int d, double mu, double ss, double log_term,
                                     double* g, exo_cfg c) {
  double T[EXO_MAX_D];
  for (int i = 0; i < d; i++) {
    double z = (q[i] - mu) / ss;
    T[i] = -0.5 * (z * z + log_term);
    g[i] = (-z) / ss;   /* reverse mode of -0.5 * (z * z + c): (-0.5 z) + (-0.5 z), then / sigma */
  }
  return lane_sum(T, d, c.lanes, 0.0);
}

int exo_leapfrog_chain_normal(const double* q0, const double* p0, const double* im, int d, int k,
                              double signed_eps, double mu, double sigma, double* q_chain,
                              double* p_chain, double* grad_chain, double* logp_chain, exo_cfg c) {
  if (d < 1 || d > EXO_MAX_D || k < 0) return -1;
  double q[EXO_MAX_D], p[EXO_MAX_D], g[EXO_MAX_D];
  const double ss = fmax(sigma, TINY_F32());                               /* normal.ex:18 */
  const double log_term = LOG_2PI_F32() + 2.0 * exo_log(ss, c.math_mode);  /* normal.ex:19,22 */
  const double h = signed_eps / 2.0;                                       /* batched_leapfrog.ex:64 */
  memcpy(q, q0, sizeof(double) * d);
  memcpy(p, p0, sizeof(double) * d);
  chain_normal_logp_grad(q, d, mu, ss, log_term, g, c);
  for (int s = 0; s < k; s++) {
    for (int i = 0; i < d; i++) {
      p[i] = p[i] + h * g[i];
      q[i] = q[i] + signed_eps * (im[i] * p[i]);
    }
    double lp = chain_normal_logp_grad(q, d, mu, ss, log_term, g, c);
    for (int i = 0; i < d; i++) p[i] = p[i] + h * g[i];
    memcpy(q_chain + (size_t)s * d, q, sizeof(double) * d);
    memcpy(p_chain + (size_t)s * d, p, sizeof(double) * d);
    memcpy(grad_chain + (size_t)s * d, g, sizeof(double) * d);
    logp_chain[s] = lp;
  }
  return 0;
}

/* ======================================================================================
 * Tree (tree.ex). Recursive, as the reference; nodes live on a bump arena.
 * ==================================================================================== */
typedef struct {
  double *qL, *pL, *gL, *qR, *pR, *gR, *qP, *gP, *rho;
  double logpP, lsw, acc;
  int n, div, turn, depth;
} node;

typedef struct {
  const exo_model* m;
  const double* im;
  double jlp0;
  exo_cfg c;
  int d;
  exo_rng rng;
  double* arena;
  size_t top, cap;
} tctx;

static void node_alloc(tctx* t, node* n) {
  int d = t->d;
  double* b = t->arena + t->top;
  t->top += (size_t)9 * d;
  n->qL = b; n->pL = b + d; n->gL = b + 2 * d; n->qR = b + 3 * d; n->pR = b + 4 * d;
  n->gR = b + 5 * d; n->qP = b + 6 * d; n->gP = b + 7 * d; n->rho = b + 8 * d;
}
static void vcp(double* dst, const double* src, int d) { memcpy(dst, src, sizeof(double) * d); }
static void node_copy(node* dst, const node* src, int d) {
  vcp(dst->qL, src->qL, d); vcp(dst->pL, src->pL, d); vcp(dst->gL, src->gL, d);
  vcp(dst->qR, src->qR, d); vcp(dst->pR, src->pR, d); vcp(dst->gR, src->gR, d);
  vcp(dst->qP, src->qP, d); vcp(dst->gP, src->gP, d); vcp(dst->rho, src->rho, d);
  dst->logpP = src->logpP; dst->lsw = src->lsw; dst->acc = src->acc;
  dst->n = src->n; dst->div = src->div; dst->turn = src->turn; dst->depth = src->depth;
}

double exo_log_sum_exp(double a, double b, int mm) {
  /* tree.ex:1597-1605 */
  double mx = (a > b) ? a : b;
  if (mx == -INFINITY || mx == -1.0e300) return -1.0e300;
  return mx + exo_log(exo_exp(a - mx, mm) + exo_exp(b - mx, mm), mm);
}

int exo_check_uturn(const double* rho, const double* pl, const double* pr, const double* im, int d,
                    exo_cfg c) {
  /* tree.ex:1578-1588 */
  double vr[EXO_MAX_D], vl[EXO_MAX_D], v[EXO_MAX_D];
  if (tl_cov) mass_times(im, rho, d, v);                 /* v = M^-1 rho */
  else for (int i = 0; i < d; i++) v[i] = rho[i] * im[i];
  for (int i = 0; i < d; i++) {
    vr[i] = v[i] * pr[i];
    vl[i] = v[i] * pl[i];
  }
  double dr = lane_sum(vr, d, c.lanes, 0.0);
  double dl = lane_sum(vl, d, c.lanes, 0.0);
  return (dr < 0.0) || (dl < 0.0);
}

static void build_leaf(tctx* t, const double* q, const double* p, const double* g, double eps,
                       node* out) {
  /* tree.ex:1011-1141 */
  int d = t->d;
  double qn[EXO_MAX_D], pn[EXO_MAX_D], gn[EXO_MAX_D], jlp;
  vcp(qn, q, d); vcp(pn, p, d); vcp(gn, g, d);
  double logp = exo_leapfrog(t->m, qn, pn, gn, eps, t->im, &jlp, t->c);
  int div;
  double lw, acc;
  if (isfinite(jlp)) {
    double dl = jlp - t->jlp0;
    div = dl < -1000.0;
    lw = dl;
    double e = exo_exp(fmin(dl, 0.0), t->c.math_mode);
    acc = fmin(1.0, e);
  } else {
    div = 1; lw = -1001.0; acc = 0.0;
  }
  const double *sq = qn, *sp = pn, *sg = gn;
  if (div) { sq = q; sp = p; sg = g; logp = -1.0e30; acc = 0.0; }
  vcp(out->qL, sq, d); vcp(out->pL, sp, d); vcp(out->gL, sg, d);
  vcp(out->qR, sq, d); vcp(out->pR, sp, d); vcp(out->gR, sg, d);
  vcp(out->qP, sq, d); vcp(out->gP, sg, d); vcp(out->rho, sp, d);
  out->logpP = logp; out->lsw = lw; out->acc = acc;
  out->n = 1; out->div = div; out->turn = 0; out->depth = 0;
}

static void merge_common(tctx* t, const node* L, const node* R, const node* take_prop,
                         double lsw, node* out) {
  /* endpoints: L's left, R's right; rho summed; proposal from take_prop */
  int d = t->d;
  vcp(out->qL, L->qL, d); vcp(out->pL, L->pL, d); vcp(out->gL, L->gL, d);
  vcp(out->qR, R->qR, d); vcp(out->pR, R->pR, d); vcp(out->gR, R->gR, d);
  vcp(out->qP, take_prop->qP, d); vcp(out->gP, take_prop->gP, d);
  out->logpP = take_prop->logpP;
  out->lsw = lsw;
}

static int sub_uturn(tctx* t, const node* L, const node* R) {
  /* checks 2 and 3 (tree.ex:1437-1446) */
  int d = t->d;
  double pr[EXO_MAX_D];
  for (int i = 0; i < d; i++) pr[i] = L->rho[i] + R->pL[i];
  if (exo_check_uturn(pr, L->pL, R->pL, t->im, d, t->c)) return 1;
  for (int i = 0; i < d; i++) pr[i] = L->pR[i] + R->rho[i];
  return exo_check_uturn(pr, L->pR, R->pR, t->im, d, t->c);
}

static void merge_subtrees(tctx* t, const node* a, const node* b, double eps, node* out) {
  /* tree.ex:1390-1476 */
  int d = t->d, mm = t->c.math_mode;
  double lsw = exo_log_sum_exp(a->lsw, b->lsw, mm);
  double u = exo_rng_uniform(&t->rng);
  int use_b = u < exo_exp(b->lsw - lsw, mm);
  const node* L = (eps > 0) ? a : b;
  const node* R = (eps > 0) ? b : a;
  double rho[EXO_MAX_D];
  for (int i = 0; i < d; i++) rho[i] = a->rho[i] + b->rho[i];
  int divg = a->div || b->div;
  int turning = divg || b->turn || exo_check_uturn(rho, L->pL, R->pR, t->im, d, t->c);
  if (!turning && a->depth > 0) turning = sub_uturn(t, L, R);
  int n = a->n + b->n;
  double acc = a->acc + b->acc;
  int depth = (a->depth > b->depth ? a->depth : b->depth) + 1;
  merge_common(t, L, R, use_b ? b : a, lsw, out);
  vcp(out->rho, rho, d);
  out->n = n; out->acc = acc; out->div = divg; out->turn = turning; out->depth = depth;
}

static void merge_trajectories(tctx* t, node* traj, const node* sub, int go_right) {
  /* tree.ex:1479-1568; result replaces traj */
  int d = t->d, mm = t->c.math_mode;
  double lsw = exo_log_sum_exp(traj->lsw, sub->lsw, mm);
  double u = exo_rng_uniform(&t->rng);
  int use_sub = exo_log(u, mm) < (sub->lsw - traj->lsw);
  const node* L = go_right ? traj : sub;
  const node* R = go_right ? sub : traj;
  double rho[EXO_MAX_D];
  for (int i = 0; i < d; i++) rho[i] = traj->rho[i] + sub->rho[i];
  int divg = traj->div || sub->div;
  int turning = divg || sub->turn || exo_check_uturn(rho, L->pL, R->pR, t->im, d, t->c);
  if (!turning) turning = sub_uturn(t, L, R);
  size_t mark = t->top;
  node out;
  node_alloc(t, &out);
  merge_common(t, L, R, use_sub ? sub : traj, lsw, &out);
  vcp(out.rho, rho, d);
  out.n = traj->n + sub->n;
  out.acc = traj->acc + sub->acc;
  out.div = divg; out.turn = turning; out.depth = traj->depth + 1;
  node_copy(traj, &out, d);
  t->top = mark;
}

static void build_subtree(tctx* t, const double* q, const double* p, const double* g, double eps,
                          int level, node* out) {
  /* tree.ex:1011-1203 */
  if (level == 0) {
    build_leaf(t, q, p, g, eps, out);
    return;
  }
  size_t mark = t->top;
  node first, second;
  node_alloc(t, &first);
  build_subtree(t, q, p, g, eps, level - 1, &first);
  if (first.div || first.turn) {
    node_copy(out, &first, t->d);
    t->top = mark;
    return;
  }
  node_alloc(t, &second);
  if (eps > 0) build_subtree(t, first.qR, first.pR, first.gR, eps, level - 1, &second);
  else build_subtree(t, first.qL, first.pL, first.gL, eps, level - 1, &second);
  merge_subtrees(t, &first, &second, eps, out);
  t->top = mark;
}

void exo_tree_build(const exo_model* m, const double* q, const double* p, double logp,
                    const double* g, double eps, const double* im, int max_depth, exo_rng rng,
                    double jlp0, double* q_out, double* g_out, exo_tree_result* res, exo_cfg c) {
  /* tree.ex:266-500 (do_build), :1607-1618 (result) */
  int d = m->d;
  tctx t;
  t.m = m; t.im = im; t.jlp0 = jlp0; t.c = c; t.d = d; t.rng = rng;
  t.cap = (size_t)9 * d * (3 * (EXO_MAX_DEPTH + 2) + 4);
  t.arena = (double*)malloc(sizeof(double) * t.cap);
  t.top = 0;
  node traj, sub;
  node_alloc(&t, &traj);
  node_alloc(&t, &sub);
  vcp(traj.qL, q, d); vcp(traj.pL, p, d); vcp(traj.gL, g, d);
  vcp(traj.qR, q, d); vcp(traj.pR, p, d); vcp(traj.gR, g, d);
  vcp(traj.qP, q, d); vcp(traj.gP, g, d); vcp(traj.rho, p, d);
  traj.logpP = logp; traj.lsw = 0.0; traj.acc = 0.0;
  traj.n = 0; traj.div = 0; traj.turn = 0; traj.depth = 0;
  int depth = 0;
  while (depth < max_depth && !traj.div && !traj.turn) {
    double u = exo_rng_uniform(&t.rng);
    int go_right = u > 0.5;
    double de = go_right ? eps : -eps;
    if (go_right) build_subtree(&t, traj.qR, traj.pR, traj.gR, de, depth, &sub);
    else build_subtree(&t, traj.qL, traj.pL, traj.gL, de, depth, &sub);
    merge_trajectories(&t, &traj, &sub, go_right);
    depth++;
  }
  vcp(q_out, traj.qP, d);
  vcp(g_out, traj.gP, d);
  res->logp = traj.logpP;
  res->n_steps = traj.n;
  res->divergent = traj.div;
  res->accept_sum = traj.acc;
  res->depth = depth;
  free(t.arena);
}

/* ======================================================================================
 * Adaptation
 * ==================================================================================== */
void exo_da_init_mode(exo_da* s, double epsilon, double target_accept, int math_mode) {
  /* step_size.ex:13-30. math_mode 1: exp/log through exmc_detmath.h and m^-kappa as
   * exp(-kappa*log(m)) — the arithmetic the device-side warmup performs. */
  s->math_mode = math_mode;
  s->log_epsilon = exo_log(epsilon, math_mode);
  s->log_epsilon_bar = exo_log(epsilon, math_mode);
  s->h_bar = 0.0;
  s->mu = exo_log(10.0 * epsilon, math_mode);
  s->m = 0;
  s->gamma = 0.05; s->t0 = 10.0; s->kappa = 0.75;
  s->target_accept = target_accept;
}
void exo_da_init(exo_da* s, double epsilon, double target_accept) {
  exo_da_init_mode(s, epsilon, target_accept, 0);
}
void exo_da_update(exo_da* s, double accept_stat) {
  /* step_size.ex:35-44 */
  int m = s->m + 1;
  double eta = 1.0 / (m + s->t0);
  double h_bar = (1.0 - eta) * s->h_bar + eta * (s->target_accept - accept_stat);
  double le = s->mu - sqrt((double)m) / s->gamma * h_bar;
  double mk = s->math_mode ? exmc_exp(-s->kappa * exmc_log((double)m)) : pow((double)m, -s->kappa);
  double leb = mk * le + (1.0 - mk) * s->log_epsilon_bar;
  s->m = m; s->h_bar = h_bar; s->log_epsilon = le; s->log_epsilon_bar = leb;
}
double exo_da_finalize(const exo_da* s) { return exo_exp(s->log_epsilon_bar, s->math_mode); }
static double da_current(const exo_da* s) { return exo_exp(s->log_epsilon, s->math_mode); }

void exo_welford_init(exo_welford* w, int d) {
  w->n = 0; w->d = d;
  for (int i = 0; i < d; i++) w->mean[i] = w->m2[i] = 0.0;
}
void exo_welford_update(exo_welford* w, const double* q) {
  /* mass_matrix.ex:40-54 */
  int n = w->n + 1;
  for (int i = 0; i < w->d; i++) {
    double delta = q[i] - w->mean[i];
    double nm = w->mean[i] + delta / ((double)n * 1.0);
    double d2 = q[i] - nm;
    w->m2[i] = w->m2[i] + delta * d2;
    w->mean[i] = nm;
  }
  w->n = n;
}
void exo_welford_finalize(const exo_welford* w, double* im) {
  /* mass_matrix.ex:77-97 */
  if (w->n < 3) {
    for (int i = 0; i < w->d; i++) im[i] = 1.0;
    return;
  }
  double alpha = 5.0 / (w->n + 5.0);
  for (int i = 0; i < w->d; i++) {
    double var = w->m2[i] / ((double)(w->n - 1) * 1.0);
    var = fmax(var, 1.0e-6);
    im[i] = (1.0 - alpha) * var + alpha * 1.0e-3;
  }
}
int exo_build_windows(int from, int to, int base, int* starts, int* ends, int maxw) {
  /* sampler.ex:764-785 */
  int n = 0, cur = from;
  double w = base;
  if (to - from <= 0) return 0;
  while (cur < to && n < maxw) {
    int remaining = to - cur;
    int actual = ((double)remaining <= w * 1.5) ? remaining : (int)w;
    starts[n] = cur;
    ends[n] = cur + actual;
    n++;
    cur += actual;
    w *= 2;
  }
  return n;
}

/* ======================================================================================
 * Sampler (sampler.ex)
 * ==================================================================================== */
typedef struct {
  double q[EXO_MAX_D], g[EXO_MAX_D];
  double logp;
  exo_rng rng;
  int divergences;
} cstate;

static __thread const double* tl_chol = NULL;   /* lower Cholesky factor of tl_cov, row-major */

static void sample_momentum(const exo_model* m, exo_rng* rng, const double* im, double* p, int mm) {
  /* sampler.ex:393-403: one normal_s per entry of the flat vector, in flat (sorted-id) order */
  int d = m->d;
  if (!tl_chol) {
    for (int r = 0; r < d; r++) {
      int i = m->flat[r];
      double z = exo_rng_normal(rng, mm);
      p[i] = z / sqrt(im[i]);
    }
    return;
  }
  /* sampler.ex:412-427: z_1..z_d, then solve L^T p = z (L^T upper triangular) by back substitution,
   * all of it on the flat vector (the factor is that of the flat covariance); entry r of the
   * solution is the momentum of kernel dimension flat[r] */
  double z[EXO_MAX_D], pf[EXO_MAX_D];
  for (int r = 0; r < d; r++) z[r] = exo_rng_normal(rng, mm);
  for (int i = d - 1; i >= 0; i--) {
    double acc = z[i];
    for (int j = d - 1; j > i; j--) acc = fma(-tl_chol[(size_t)j * d + i], pf[j], acc);
    pf[i] = acc / tl_chol[(size_t)i * d + i];
  }
  for (int r = 0; r < d; r++) p[m->flat[r]] = pf[r];
}

typedef struct {
  int depth, n_steps, divergent;
  double accept, energy;
} step_info;

static void nuts_step(const exo_model* m, cstate* s, double eps, const double* im, int max_depth,
                      step_info* info, exo_cfg c) {
  /* sampler.ex:794-925 */
  int d = m->d;
  double p[EXO_MAX_D], qn[EXO_MAX_D], gn[EXO_MAX_D];
  sample_momentum(m, &s->rng, im, p, c.math_mode);
  double jlp0 = s->logp - exo_kinetic_energy(p, im, d, c);
  exo_tree_result r;
  exo_tree_build(m, s->q, p, s->logp, s->g, eps, im, max_depth, s->rng, jlp0, qn, gn, &r, c);
  (void)exo_rng_uniform(&s->rng); /* sampler.ex:836,897: tree's own draws are discarded */
  if (r.divergent) s->divergences++;
  vcp(s->q, qn, d); vcp(s->g, gn, d);
  s->logp = r.logp;
  info->depth = r.depth; info->n_steps = r.n_steps; info->divergent = r.divergent;
  info->accept = (r.n_steps > 0) ? r.accept_sum / r.n_steps : 0.0;
  info->energy = -jlp0;
}

static double find_reasonable_epsilon(const exo_model* m, cstate* s, const double* im, exo_cfg c) {
  /* sampler.ex:451-530 */
  int d = m->d;
  double p[EXO_MAX_D];
  sample_momentum(m, &s->rng, im, p, c.math_mode);
  double jlp0 = s->logp - exo_kinetic_energy(p, im, d, c);
  double eps = 1.0;
  double q[EXO_MAX_D], pp[EXO_MAX_D], g[EXO_MAX_D], jlp;
  vcp(q, s->q, d); vcp(pp, p, d); vcp(g, s->g, d);
  exo_leapfrog(m, q, pp, g, eps, im, &jlp, c);
  double la = (isfinite(jlp0) && isfinite(jlp)) ? jlp - jlp0 : -1000.0;
  double dir = (la > log(0.5)) ? 1.0 : -1.0;
  for (int count = 0; count < 100; count++) {
    double ne = eps * pow(2.0, dir);
    vcp(q, s->q, d); vcp(pp, p, d); vcp(g, s->g, d);
    exo_leapfrog(m, q, pp, g, ne, im, &jlp, c);
    la = (isfinite(jlp0) && isfinite(jlp)) ? jlp - jlp0 : -1000.0;
    int crossed = (dir > 0) ? (la < log(0.5)) : (la > log(0.5));
    if (crossed || !isfinite(la)) return fmax(ne, 1.0e-10);
    eps = ne;
  }
  return fmax(eps, 1.0e-10);
}

static void init_chain(const exo_model* m, const double* init_q, uint64_t seed, cstate* s,
                       exo_cfg c) {
  /* sampler.ex:154-165, 339-356 */
  int d = m->d;
  exo_rng_seed(&s->rng, seed);
  if (init_q) {
    vcp(s->q, init_q, d);
  } else {
    /* sampler.ex:339-349: d normal_s draws fill the flat vector front to back */
    for (int r = 0; r < d; r++) s->q[m->flat[r]] = exo_rng_normal(&s->rng, c.math_mode) * 0.1;
  }
  s->logp = exo_logp_grad(m, s->q, s->g, c);
  s->divergences = 0;
}

static void run_phase(const exo_model* m, cstate* s, const double* im, int max_depth, exo_da* da,
                      int from, int to, exo_cfg c) {
  /* sampler.ex:623-666 */
  for (int i = from; i < to; i++) {
    step_info info;
    nuts_step(m, s, da_current(da), im, max_depth, &info, c);
    exo_da_update(da, info.accept);
  }
}

/* ---- dense Welford / finalize (mass_matrix.ex:27-35, 56-72, 105-140) ---- */
typedef struct {
  int n, d;
  double mean[EXO_MAX_D];
  double* m2;   /* d x d */
} welford_dense;

static void wd_init(welford_dense* w, int d, double* m2_store) {
  w->n = 0; w->d = d; w->m2 = m2_store;
  for (int i = 0; i < d; i++) w->mean[i] = 0.0;
  for (int i = 0; i < d * d; i++) m2_store[i] = 0.0;
}
static void wd_update(welford_dense* w, const double* q, const int* flat) {
  /* mass_matrix.ex:56-72: m2 += outer(delta, delta2), accumulated with fma; rows and columns are
   * flat entries (flat[r] = the kernel dimension of entry r, NULL = identity) */
  int d = w->d, nn = w->n + 1;
  double delta[EXO_MAX_D], delta2[EXO_MAX_D];
  for (int i = 0; i < d; i++) {
    double qi = q[flat ? flat[i] : i];
    delta[i] = qi - w->mean[i];
    double nm = w->mean[i] + delta[i] / ((double)nn * 1.0);
    delta2[i] = qi - nm;
    w->mean[i] = nm;
  }
  for (int i = 0; i < d; i++)
    for (int j = 0; j < d; j++) w->m2[(size_t)i * d + j] = fma(delta[i], delta2[j], w->m2[(size_t)i * d + j]);
  w->n = nn;
}
int exo_cholesky_lower(const double* a, int d, double* l) {
  /* Nx.LinAlg.cholesky (mass_matrix.ex:138): lower factor, row by row (Cholesky-Banachiewicz),
   * sums accumulated with fma in ascending k. Returns -1 if a pivot is not positive. */
  for (int i = 0; i < d * d; i++) l[i] = 0.0;
  for (int i = 0; i < d; i++)
    for (int j = 0; j <= i; j++) {
      double acc = a[(size_t)i * d + j];
      for (int k = 0; k < j; k++) acc = fma(-l[(size_t)i * d + k], l[(size_t)j * d + k], acc);
      if (i == j) {
        if (!(acc > 0.0)) return -1;
        l[(size_t)i * d + i] = sqrt(acc);
      } else {
        l[(size_t)i * d + j] = acc / l[(size_t)j * d + j];
      }
    }
  return 0;
}
static int wd_finalize(const welford_dense* w, double* cov, double* chol) {
  /* mass_matrix.ex:105-140 */
  int d = w->d, n = w->n;
  if (n < 3) {
    for (int i = 0; i < d * d; i++) cov[i] = chol[i] = 0.0;
    for (int i = 0; i < d; i++) cov[(size_t)i * d + i] = chol[(size_t)i * d + i] = 1.0;
    return 0;
  }
  double alpha = 5.0 / (n + 5.0);
  for (int i = 0; i < d * d; i++) cov[i] = w->m2[i] / ((double)(n - 1) * 1.0);
  for (int i = 0; i < d; i++)
    for (int j = 0; j < d; j++) {
      /* shrink toward the floored sample diagonal: (1 - alpha) cov + alpha max(diag(cov), 1e-6 I) */
      double dg = (i == j) ? fmax(cov[(size_t)i * d + i], 1.0e-6) : 0.0;
      cov[(size_t)i * d + j] = (1.0 - alpha) * cov[(size_t)i * d + j] + alpha * dg;
    }
  return exo_cholesky_lower(cov, d, chol);
}

static double run_warmup_dense(const exo_model* m, cstate* s, double eps, double* im, double* cov,
                               double* chol, int* dense_on, exo_opts o, exo_cfg c) {
  /* sampler.ex:537-762 with use_dense: Phase I on the identity diagonal, windows of base
   * max(25, 10 d) (sampler.ex:682), after each window the dense covariance and its factor, then
   * the step-size search and the remaining phases under the dense mass */
  int d = m->d, W = o.num_warmup;
  *dense_on = 0;
  if (W == 0) return eps;
  int init_buffer = (75 < W / 3) ? 75 : W / 3;
  int adapt_end = W - 50;
  exo_da da;
  exo_da_init_mode(&da, eps, o.target_accept, c.math_mode);
  run_phase(m, s, im, o.max_tree_depth, &da, 0, init_buffer, c);
  eps = da_current(&da);
  if (adapt_end <= init_buffer) return exo_da_finalize(&da);
  int ws[32], we[32];
  int base = 10 * d > 25 ? 10 * d : 25;
  int nw = exo_build_windows(init_buffer, adapt_end, base, ws, we, 32);
  double* m2 = (double*)malloc(sizeof(double) * d * d);
  for (int k = 0; k < nw; k++) {
    welford_dense wf;
    wd_init(&wf, d, m2);
    exo_da_init_mode(&da, eps, o.target_accept, c.math_mode);
    for (int i = ws[k]; i < we[k]; i++) {
      int cap = (i < 200) ? (o.max_tree_depth < 8 ? o.max_tree_depth : 8) : o.max_tree_depth;
      int div_before = s->divergences;
      step_info info;
      nuts_step(m, s, da_current(&da), im, cap, &info, c);
      exo_da_update(&da, info.accept);
      if (s->divergences == div_before) wd_update(&wf, s->q, m->flat);
    }
    tl_cov = NULL; tl_chol = NULL; tl_flat = NULL;
    if (wd_finalize(&wf, cov, chol) != 0) { free(m2); return -1.0; }   /* not positive definite */
    for (int r = 0; r < d; r++) im[m->flat[r]] = cov[(size_t)r * d + r];
    tl_cov = cov; tl_chol = chol; tl_flat = m->flat;
    *dense_on = 1;
    eps = find_reasonable_epsilon(m, s, im, c);
  }
  free(m2);
  exo_da_init_mode(&da, eps, o.target_accept, c.math_mode);
  run_phase(m, s, im, o.max_tree_depth, &da, adapt_end, W, c);
  return exo_da_finalize(&da);
}

static double run_warmup(const exo_model* m, cstate* s, double eps, double* im, exo_opts o,
                         exo_cfg c) {
  /* sampler.ex:537-762; diagonal mass only */
  int d = m->d, W = o.num_warmup;
  if (W == 0) return eps;
  int init_buffer = (75 < W / 3) ? 75 : W / 3;
  int adapt_end = W - 50;
  exo_da da;
  exo_da_init_mode(&da, eps, o.target_accept, c.math_mode);
  run_phase(m, s, im, o.max_tree_depth, &da, 0, init_buffer, c);
  eps = da_current(&da);
  if (adapt_end <= init_buffer) return exo_da_finalize(&da);
  int ws[32], we[32];
  int nw = exo_build_windows(init_buffer, adapt_end, 25, ws, we, 32);
  for (int k = 0; k < nw; k++) {
    exo_welford wf;
    exo_welford_init(&wf, d);
    exo_da_init_mode(&da, eps, o.target_accept, c.math_mode);
    for (int i = ws[k]; i < we[k]; i++) {
      int cap = (i < 200) ? (o.max_tree_depth < 8 ? o.max_tree_depth : 8) : o.max_tree_depth;
      int div_before = s->divergences;
      step_info info;
      nuts_step(m, s, da_current(&da), im, cap, &info, c);
      exo_da_update(&da, info.accept);
      if (s->divergences == div_before) exo_welford_update(&wf, s->q);
    }
    exo_welford_finalize(&wf, im);
    eps = find_reasonable_epsilon(m, s, im, c);
  }
  exo_da_init_mode(&da, eps, o.target_accept, c.math_mode);
  run_phase(m, s, im, o.max_tree_depth, &da, adapt_end, W, c);
  return exo_da_finalize(&da);
}

static void run_sampling(const exo_model* m, cstate* s, double eps, const double* im, exo_opts o,
                         exo_trace tr, size_t base, long* leapfrogs, exo_cfg c) {
  /* sampler.ex:929-973 */
  int d = m->d;
  for (int i = 0; i < o.num_samples; i++) {
    step_info info;
    nuts_step(m, s, eps, im, o.max_tree_depth, &info, c);
    size_t k = base + (size_t)i;
    if (tr.draws) vcp(tr.draws + k * d, s->q, d);
    if (tr.logp) tr.logp[k] = s->logp;
    if (tr.tree_depth) tr.tree_depth[k] = info.depth;
    if (tr.n_steps) tr.n_steps[k] = info.n_steps;
    if (tr.divergent) tr.divergent[k] = info.divergent;
    if (tr.accept_prob) tr.accept_prob[k] = info.accept;
    if (tr.energy) tr.energy[k] = info.energy;
    if (leapfrogs) *leapfrogs += info.n_steps;
  }
}

static double warmup_chain0(const exo_model* m, const double* init_q, exo_opts o, cstate* s,
                            double* im, exo_cfg c) {
  int d = m->d;
  init_chain(m, init_q, o.seed, s, c);
  for (int i = 0; i < d; i++) im[i] = 1.0;
  double eps = find_reasonable_epsilon(m, s, im, c);
  return run_warmup(m, s, eps, im, o, c);
}

int exo_warmup(const exo_model* m, const double* init_q, exo_opts o, exo_stats* st, exo_cfg c) {
  cstate s;
  st->step_size = warmup_chain0(m, init_q, o, &s, st->inv_mass, c);
  st->divergences = s.divergences;
  st->total_leapfrogs = 0;
  return 0;
}

int exo_sample(const exo_model* m, const double* init_q, exo_opts o, exo_trace tr, exo_stats* st,
               exo_cfg c) {
  /* sampler.ex:126-257 (cold start, diagonal mass) */
  cstate s;
  st->step_size = warmup_chain0(m, init_q, o, &s, st->inv_mass, c);
  st->total_leapfrogs = 0;
  run_sampling(m, &s, st->step_size, st->inv_mass, o, tr, 0, &st->total_leapfrogs, c);
  st->divergences = s.divergences;
  return 0;
}

int exo_sample_warm(const exo_model* m, const double* init_q, double prev_epsilon,
                    const double* prev_inv_mass, exo_opts o, exo_trace tr, exo_stats* st, exo_cfg c) {
  /* sampler.ex:167-197 (opts[:warm_start]): the previous run's inverse mass and step size, a short
   * warmup of min(num_warmup, 50) iterations without the initial step-size search, then sampling */
  cstate s;
  init_chain(m, init_q, o.seed, &s, c);
  vcp(st->inv_mass, prev_inv_mass, m->d);
  exo_opts ow = o;
  ow.num_warmup = o.num_warmup < 50 ? o.num_warmup : 50;
  st->step_size = run_warmup(m, &s, prev_epsilon, st->inv_mass, ow, c);
  st->total_leapfrogs = 0;
  run_sampling(m, &s, st->step_size, st->inv_mass, o, tr, 0, &st->total_leapfrogs, c);
  st->divergences = s.divergences;
  return 0;
}

int exo_sample_tuned(const exo_model* m, const double* init_q, double epsilon, const double* im,
                     exo_opts o, exo_trace tr, exo_stats* st, exo_cfg c) {
  /* sampler.ex:260-335 */
  cstate s;
  init_chain(m, init_q, o.seed, &s, c);
  st->step_size = epsilon;
  vcp(st->inv_mass, im, m->d);
  st->total_leapfrogs = 0;
  run_sampling(m, &s, epsilon, im, o, tr, 0, &st->total_leapfrogs, c);
  st->divergences = s.divergences;
  return 0;
}

/* ---- opts[:dense_mass] entry points (cov, chol: row-major d x d, caller-owned) ---- */
int exo_warmup_dense(const exo_model* m, const double* init_q, exo_opts o, exo_stats* st, double* cov,
                     double* chol, exo_cfg c) {
  cstate s;
  int d = m->d, on = 0;
  tl_cov = NULL; tl_chol = NULL; tl_flat = NULL;
  init_chain(m, init_q, o.seed, &s, c);
  for (int i = 0; i < d; i++) st->inv_mass[i] = 1.0;
  double eps = find_reasonable_epsilon(m, &s, st->inv_mass, c);
  st->step_size = run_warmup_dense(m, &s, eps, st->inv_mass, cov, chol, &on, o, c);
  tl_cov = NULL; tl_chol = NULL; tl_flat = NULL;
  if (!on) {   /* no window ran: identity covariance, as finalize of an empty Welford would give */
    for (int i = 0; i < d * d; i++) cov[i] = chol[i] = 0.0;
    for (int i = 0; i < d; i++) cov[(size_t)i * d + i] = chol[(size_t)i * d + i] = 1.0;
  }
  st->divergences = s.divergences;
  st->total_leapfrogs = 0;
  return st->step_size < 0.0 ? -1 : 0;
}

int exo_sample_tuned_dense(const exo_model* m, const double* init_q, double epsilon, const double* cov,
                           const double* chol, exo_opts o, exo_trace tr, exo_stats* st, exo_cfg c) {
  /* sample_compiled_tuned with tuning.chol_cov (sampler.ex:260-335) for one chain */
  cstate s;
  int d = m->d;
  tl_cov = NULL; tl_chol = NULL; tl_flat = NULL;
  init_chain(m, init_q, o.seed, &s, c);
  for (int r = 0; r < d; r++) st->inv_mass[m->flat[r]] = cov[(size_t)r * d + r];
  st->step_size = epsilon;
  st->total_leapfrogs = 0;
  tl_cov = cov; tl_chol = chol; tl_flat = m->flat;
  run_sampling(m, &s, epsilon, st->inv_mass, o, tr, 0, &st->total_leapfrogs, c);
  tl_cov = NULL; tl_chol = NULL; tl_flat = NULL;
  st->divergences = s.divergences;
  return 0;
}

/* one product / solve each, for the unit tests */
void exo_dense_mass_times(const double* cov, const double* x, int d, double* out) {
  tl_cov = cov; tl_flat = NULL;
  mass_times(NULL, x, d, out);
  tl_cov = NULL;
}
int exo_dense_check_uturn(const double* cov, const double* rho, const double* pl, const double* pr, int d,
                          exo_cfg c) {
  tl_cov = cov; tl_flat = NULL;
  int r = exo_check_uturn(rho, pl, pr, NULL, d, c);
  tl_cov = NULL;
  return r;
}
void exo_dense_momentum(const exo_model* m, const double* chol, exo_rng* rng, double* p, int math_mode) {
  tl_chol = chol;
  sample_momentum(m, rng, NULL, p, math_mode);
  tl_chol = NULL;
}
void exo_welford_dense_finalize(const double* draws, int n, int d, double* cov, double* chol) {
  /* n draws [n][d] through the dense Welford, then finalize_dense */
  double* m2 = (double*)malloc(sizeof(double) * d * d);
  welford_dense w;
  wd_init(&w, d, m2);
  for (int i = 0; i < n; i++) wd_update(&w, draws + (size_t)i * d, NULL);
  wd_finalize(&w, cov, chol);
  free(m2);
}

typedef struct {
  const exo_model* m;
  const double* init_q;
  exo_opts o;
  exo_trace tr;
  exo_cfg c;
  double eps;
  const double* im;
  int lo, hi, chain_lo;
  long leapfrogs;
  int divergences;
} chain_job;

static void* chain_worker(void* arg) {
  chain_job* j = (chain_job*)arg;
  for (int ch = j->lo; ch < j->hi; ch++) {
    cstate s;
    init_chain(j->m, j->init_q, j->o.seed + (uint64_t)ch * 7919ULL, &s, j->c);
    size_t base = (size_t)(ch - j->chain_lo) * (size_t)j->o.num_samples;
    run_sampling(j->m, &s, j->eps, j->im, j->o, j->tr, base, &j->leapfrogs, j->c);
    j->divergences += s.divergences;
  }
  return 0;
}

int exo_sample_chains(const exo_model* m, const double* init_q, int n_chains, int chain_lo,
                      int chain_hi, exo_opts o, exo_trace tr, exo_stats* st, int n_threads,
                      exo_cfg c) {
  /* sampler.ex:1020-1136 */
  cstate s0;
  st->step_size = warmup_chain0(m, init_q, o, &s0, st->inv_mass, c);
  if (chain_hi > n_chains) chain_hi = n_chains;
  int nc = chain_hi - chain_lo;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > nc) n_threads = nc > 0 ? nc : 1;
  chain_job* jobs = (chain_job*)calloc(n_threads, sizeof(chain_job));
  pthread_t* th = (pthread_t*)calloc(n_threads, sizeof(pthread_t));
  for (int t = 0; t < n_threads; t++) {
    chain_job* j = &jobs[t];
    j->m = m; j->init_q = init_q; j->o = o; j->tr = tr; j->c = c;
    j->eps = st->step_size; j->im = st->inv_mass; j->chain_lo = chain_lo;
    j->lo = chain_lo + (int)((long)nc * t / n_threads);
    j->hi = chain_lo + (int)((long)nc * (t + 1) / n_threads);
    if (n_threads > 1) pthread_create(&th[t], 0, chain_worker, j);
    else chain_worker(j);
  }
  st->total_leapfrogs = 0;
  st->divergences = 0;
  for (int t = 0; t < n_threads; t++) {
    if (n_threads > 1) pthread_join(th[t], 0);
    st->total_leapfrogs += jobs[t].leapfrogs;
    st->divergences += jobs[t].divergences;
  }
  free(jobs);
  free(th);
  return 0;
}

/* ======================================================================================
 * Diagnostics (diagnostics.ex)
 * ==================================================================================== */
static double ess_from_acf_direct(const double* x, int n) {
  /* diagnostics.ex:123-167: direct ACF, Geyer initial positive sequence */
  double mean = 0.0;
  for (int i = 0; i < n; i++) mean += x[i];
  mean /= n;
  double* c = (double*)malloc(sizeof(double) * n);
  double var = 0.0;
  for (int i = 0; i < n; i++) {
    c[i] = x[i] - mean;
  }
  for (int i = 0; i < n; i++) var += c[i] * c[i];
  double tau = -1.0;
  if (var != 0.0) {
    int max_lag = n - 1;
    int max_k = max_lag / 2;
    for (int k = 0; k <= max_k; k++) {
      double r[2] = {0.0, 0.0};
      for (int j = 0; j < 2; j++) {
        int lag = 2 * k + j;
        if (lag > max_lag) { r[j] = 0.0; continue; }
        double sum = 0.0;
        for (int i = 0; i < n - lag; i++) sum += c[i] * c[i + lag];
        r[j] = sum / var;
      }
      double pair = r[0] + r[1];
      if (pair > 0) tau += 2 * pair;
      else break;
    }
  }
  free(c);
  return n / fmax(tau, 1.0);
}
double exo_ess(const double* x, int n) {
  if (n < 4) return n * 1.0;
  return ess_from_acf_direct(x, n);
}
static int g_probit_mode = 0;   /* 1: log from exmc_detmath.h, as the device kernel */
static double probit_inner(double p) {
  double t = sqrt(-2.0 * exo_log(p, g_probit_mode));
  return t - (2.515517 + 0.802853 * t + 0.010328 * t * t) /
                 (1.0 + 1.432788 * t + 0.189269 * t * t + 0.001308 * t * t * t);
}
static double probit(double p) { return p < 0.5 ? -probit_inner(p) : probit_inner(1.0 - p); }
typedef struct { double v; int i; } vi;
static int vi_cmp(const void* a, const void* b) {
  double x = ((const vi*)a)->v, y = ((const vi*)b)->v;
  if (x < y) return -1;
  if (x > y) return 1;
  return ((const vi*)a)->i - ((const vi*)b)->i;
}
double exo_ess_bulk(const double* x, int n) {
  /* diagnostics.ex:60-72, 186-219 */
  if (n < 4) return n * 1.0;
  vi* s = (vi*)malloc(sizeof(vi) * n);
  double* z = (double*)malloc(sizeof(double) * n);
  for (int i = 0; i < n; i++) { s[i].v = x[i]; s[i].i = i; }
  qsort(s, n, sizeof(vi), vi_cmp);
  int pos = 0;
  while (pos < n) {
    int e = pos;
    while (e + 1 < n && s[e + 1].v == s[pos].v) e++;
    double avg = (pos + 1) + (e - pos) / 2.0;
    for (int k = pos; k <= e; k++) z[s[k].i] = probit((avg - 0.375) / (n + 0.25));
    pos = e + 1;
  }
  double r = ess_from_acf_direct(z, n);
  free(s);
  free(z);
  return r;
}
double exo_ess_bulk_mode(const double* x, int n, int math_mode) {
  g_probit_mode = math_mode;
  double r = exo_ess_bulk(x, n);
  g_probit_mode = 0;
  return r;
}
double exo_rhat(const double* chains, int nch, int n) {
  /* diagnostics.ex:80-115: split R-hat */
  int mid = n / 2;
  int m = 2 * nch;
  int len = mid < (n - mid) ? mid : (n - mid);
  double* means = (double*)malloc(sizeof(double) * m);
  double* vars = (double*)malloc(sizeof(double) * m);
  for (int c = 0; c < nch; c++)
    for (int hlf = 0; hlf < 2; hlf++) {
      const double* x = chains + (size_t)c * n + (hlf ? mid : 0);
      double sum = 0.0;
      for (int i = 0; i < len; i++) sum += x[i];
      double cm = sum / len;
      double ss = 0.0;
      for (int i = 0; i < len; i++) ss += (x[i] - cm) * (x[i] - cm);
      means[2 * c + hlf] = cm;
      vars[2 * c + hlf] = ss / (len - 1);
    }
  double gm = 0.0;
  for (int k = 0; k < m; k++) gm += means[k];
  gm /= m;
  double b = 0.0, w = 0.0;
  for (int k = 0; k < m; k++) { b += (means[k] - gm) * (means[k] - gm); w += vars[k]; }
  b = (double)len / (m - 1) * b;
  w /= m;
  double var_hat = (double)(len - 1) / len * w + b / len;
  free(means);
  free(vars);
  return sqrt(var_hat / w);
}

/* ======================================================================================
 * NativeTree NIF semantics (native/exmc_tree/src/tree.rs, lib.rs). Differences from the
 * Elixir path that are kept on purpose: a divergent leaf keeps the NEW
 * q,p,g; KE is summed as 0.5*p*m*p per element; sub-trajectory checks run before check 1;
 * log_sum_exp returns -inf; RNG is Xoshiro256** seeded with seed_from_u64.
 * ==================================================================================== */
struct exo_nt_traj {
  int d;
  double *qL, *pL, *gL, *qR, *pR, *gR, *qP, *gP, *rho;
  double logpP, lsw, acc;
  int n, div, turn, depth;
};

typedef struct {
  const double *all_q, *all_p, *all_logp, *all_g, *im;
  int d;
  double jlp0;
  uint64_t* rng;
  double* arena;
  size_t top;
} ntctx;

static int nt_uturn(const double* rho, const double* pl, const double* pr, const double* im, int d) {
  /* uturn.rs:8-24 */
  double dr = 0.0, dl = 0.0;
  for (int i = 0; i < d; i++) {
    double v = rho[i] * im[i];
    dr += v * pr[i];
    dl += v * pl[i];
  }
  return dr < 0.0 || dl < 0.0;
}
/* 0 = libm (what the Rust crate calls); 1 = exmc_detmath.h (the GPU-batched entry point's contract) */
static int g_nt_math_mode = 0;
void exo_nt_set_math_mode(int mode) { g_nt_math_mode = mode; }
static double nt_exp(double x) { return exo_exp(x, g_nt_math_mode); }
static double nt_log(double x) { return exo_log(x, g_nt_math_mode); }
static double nt_lse(double a, double b) {
  /* math.rs:3-10 */
  double m = fmax(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + nt_log(nt_exp(a - m) + nt_exp(b - m));
}
static void nt_node_alloc(ntctx* t, node* n) {
  int d = t->d;
  double* b = t->arena + t->top;
  t->top += (size_t)9 * d;
  n->qL = b; n->pL = b + d; n->gL = b + 2 * d; n->qR = b + 3 * d; n->pR = b + 4 * d;
  n->gR = b + 5 * d; n->qP = b + 6 * d; n->gP = b + 7 * d; n->rho = b + 8 * d;
}
static int nt_sub_uturn(const node* L, const node* R, const double* im, int d) {
  double pr[EXO_MAX_D];
  for (int i = 0; i < d; i++) pr[i] = L->rho[i] + R->pL[i];
  if (nt_uturn(pr, L->pL, R->pL, im, d)) return 1;
  for (int i = 0; i < d; i++) pr[i] = L->pR[i] + R->rho[i];
  return nt_uturn(pr, L->pR, R->pR, im, d);
}
static void nt_build_subtree(ntctx* t, int depth, int going_right, int* counter, node* out) {
  int d = t->d;
  if (depth == 0) {
    /* tree.rs:44-95 */
    int idx = (*counter)++;
    const double* q = t->all_q + (size_t)idx * d;
    const double* p = t->all_p + (size_t)idx * d;
    const double* g = t->all_g + (size_t)idx * d;
    double logp = t->all_logp[idx];
    double ke = 0.0;
    for (int i = 0; i < d; i++) ke += 0.5 * p[i] * t->im[i] * p[i];
    double jlp = logp - ke;
    int div;
    double lw, acc;
    if (isfinite(jlp)) {
      double dl = jlp - t->jlp0;
      div = dl < -1000.0;
      lw = dl;
      acc = fmin(nt_exp(fmin(dl, 0.0)), 1.0);
    } else {
      div = 1; lw = -1001.0; acc = 0.0;
    }
    vcp(out->qL, q, d); vcp(out->pL, p, d); vcp(out->gL, g, d);
    vcp(out->qR, q, d); vcp(out->pR, p, d); vcp(out->gR, g, d);
    vcp(out->qP, q, d); vcp(out->gP, g, d); vcp(out->rho, p, d);
    out->logpP = logp; out->lsw = lw; out->acc = acc;
    out->n = 1; out->div = div; out->turn = 0; out->depth = 0;
    return;
  }
  size_t mark = t->top;
  node a, b;
  nt_node_alloc(t, &a);
  nt_build_subtree(t, depth - 1, going_right, counter, &a);
  if (a.div || a.turn) {
    node_copy(out, &a, d);
    t->top = mark;
    return;
  }
  nt_node_alloc(t, &b);
  nt_build_subtree(t, depth - 1, going_right, counter, &b);
  /* merge_subtrees, tree.rs:103-189 */
  double lsw = nt_lse(a.lsw, b.lsw);
  int divg = a.div || b.div;
  double u = exo_xoshiro_f64(t->rng);
  int use_b = u < nt_exp(b.lsw - lsw);
  double rho[EXO_MAX_D];
  for (int i = 0; i < d; i++) rho[i] = a.rho[i] + b.rho[i];
  const node* L = going_right ? &a : &b;
  const node* R = going_right ? &b : &a;
  int sub_turning = 0;
  if (!divg && !b.turn && a.depth > 0) sub_turning = nt_sub_uturn(L, R, t->im, d);
  int turning = divg || b.turn || sub_turning || nt_uturn(rho, L->pL, R->pR, t->im, d);
  const node* pr = use_b ? &b : &a;
  vcp(out->qL, L->qL, d); vcp(out->pL, L->pL, d); vcp(out->gL, L->gL, d);
  vcp(out->qR, R->qR, d); vcp(out->pR, R->pR, d); vcp(out->gR, R->gR, d);
  vcp(out->qP, pr->qP, d); vcp(out->gP, pr->gP, d); vcp(out->rho, rho, d);
  out->logpP = pr->logpP; out->lsw = lsw; out->acc = a.acc + b.acc;
  out->n = a.n + b.n; out->div = divg; out->turn = turning;
  out->depth = (a.depth > b.depth ? a.depth : b.depth) + 1;
  t->top = mark;
}
static void nt_merge_into(exo_nt_traj* tr, const node* sub, int go_right, const double* im,
                          uint64_t* rng) {
  /* tree.rs:194-265 */
  int d = tr->d;
  double lsw = nt_lse(tr->lsw, sub->lsw);
  int divg = tr->div || sub->div;
  int sub_turning = 0;
  if (!divg && !sub->turn) {
    node T;
    T.rho = tr->rho; T.pL = tr->pL; T.pR = tr->pR;
    const node* L = go_right ? &T : sub;
    const node* R = go_right ? sub : &T;
    sub_turning = nt_sub_uturn(L, R, im, d);
  }
  double u = exo_xoshiro_f64(rng);
  if (nt_log(u) < (sub->lsw - tr->lsw)) {
    vcp(tr->qP, sub->qP, d); vcp(tr->gP, sub->gP, d);
    tr->logpP = sub->logpP;
  }
  for (int i = 0; i < d; i++) tr->rho[i] += sub->rho[i];
  if (go_right) { vcp(tr->qR, sub->qR, d); vcp(tr->pR, sub->pR, d); vcp(tr->gR, sub->gR, d); }
  else { vcp(tr->qL, sub->qL, d); vcp(tr->pL, sub->pL, d); vcp(tr->gL, sub->gL, d); }
  int turning = divg || sub->turn || sub_turning || nt_uturn(tr->rho, tr->pL, tr->pR, im, d);
  tr->lsw = lsw; tr->n += sub->n; tr->acc += sub->acc;
  tr->div = divg; tr->turn = turning; tr->depth += 1;
}

exo_nt_traj* exo_nt_init_trajectory(const double* q, const double* p, const double* g, double logp,
                                    int d) {
  /* types.rs:129-152 */
  exo_nt_traj* t = (exo_nt_traj*)calloc(1, sizeof(exo_nt_traj));
  double* b = (double*)malloc(sizeof(double) * 9 * d);
  t->d = d;
  t->qL = b; t->pL = b + d; t->gL = b + 2 * d; t->qR = b + 3 * d; t->pR = b + 4 * d;
  t->gR = b + 5 * d; t->qP = b + 6 * d; t->gP = b + 7 * d; t->rho = b + 8 * d;
  vcp(t->qL, q, d); vcp(t->pL, p, d); vcp(t->gL, g, d);
  vcp(t->qR, q, d); vcp(t->pR, p, d); vcp(t->gR, g, d);
  vcp(t->qP, q, d); vcp(t->gP, g, d); vcp(t->rho, p, d);
  t->logpP = logp; t->lsw = 0.0; t->acc = 0.0;
  return t;
}
void exo_nt_free(exo_nt_traj* t) {
  if (!t) return;
  free(t->qL);
  free(t);
}
int exo_nt_is_terminated(const exo_nt_traj* t) { return t->div || t->turn; }
void exo_nt_get_endpoint(const exo_nt_traj* t, int go_right, double* q, double* p, double* g) {
  int d = t->d;
  vcp(q, go_right ? t->qR : t->qL, d);
  vcp(p, go_right ? t->pR : t->pL, d);
  vcp(g, go_right ? t->gR : t->gL, d);
}
static void nt_run_subtree(exo_nt_traj* tr, const double* aq, const double* ap, const double* alp,
                           const double* ag, const double* im, double jlp0, int depth, int d,
                           int go_right, uint64_t* rng) {
  ntctx t;
  t.all_q = aq; t.all_p = ap; t.all_logp = alp; t.all_g = ag; t.im = im;
  t.d = d; t.jlp0 = jlp0; t.rng = rng;
  t.arena = (double*)malloc(sizeof(double) * 9 * d * (3 * (EXO_MAX_DEPTH + 2) + 2));
  t.top = 0;
  node sub;
  nt_node_alloc(&t, &sub);
  int counter = 0;
  nt_build_subtree(&t, depth, go_right, &counter, &sub);
  nt_merge_into(tr, &sub, go_right, im, rng);
  free(t.arena);
}
void exo_nt_build_and_merge(exo_nt_traj* tr, const double* aq, const double* ap, const double* alp,
                            const double* ag, const double* im, double jlp0, int depth, int d,
                            int go_right, uint64_t seed) {
  /* lib.rs:114-147 */
  uint64_t rng[4];
  exo_xoshiro_seed_from_u64(rng, seed);
  nt_run_subtree(tr, aq, ap, alp, ag, im, jlp0, depth, d, go_right, rng);
}
void exo_nt_build_subtree(const double* aq, const double* ap, const double* alp, const double* ag,
                          const double* im, double jlp0, int depth, int d, int going_right,
                          uint64_t seed, double* vecs, double* scalars, int* ints) {
  /* lib.rs:114-212: the subtree record. vecs = qL,pL,gL,qR,pR,gR,qP,gP,rho (9*d);
   * scalars = logp_prop, log_sum_weight, accept_sum; ints = n_steps, divergent, turning, depth */
  uint64_t rng[4];
  exo_xoshiro_seed_from_u64(rng, seed);
  ntctx t;
  t.all_q = aq; t.all_p = ap; t.all_logp = alp; t.all_g = ag; t.im = im;
  t.d = d; t.jlp0 = jlp0; t.rng = rng;
  t.arena = (double*)malloc(sizeof(double) * 9 * d * (3 * (EXO_MAX_DEPTH + 2) + 2));
  t.top = 0;
  node sub;
  nt_node_alloc(&t, &sub);
  int counter = 0;
  nt_build_subtree(&t, depth, going_right, &counter, &sub);
  vcp(vecs, sub.qL, d); vcp(vecs + d, sub.pL, d); vcp(vecs + 2 * d, sub.gL, d);
  vcp(vecs + 3 * d, sub.qR, d); vcp(vecs + 4 * d, sub.pR, d); vcp(vecs + 5 * d, sub.gR, d);
  vcp(vecs + 6 * d, sub.qP, d); vcp(vecs + 7 * d, sub.gP, d); vcp(vecs + 8 * d, sub.rho, d);
  scalars[0] = sub.logpP; scalars[1] = sub.lsw; scalars[2] = sub.acc;
  ints[0] = sub.n; ints[1] = sub.div; ints[2] = sub.turn; ints[3] = sub.depth;
  free(t.arena);
}
void exo_nt_get_result(const exo_nt_traj* t, double* q, double* g, exo_tree_result* res) {
  vcp(q, t->qP, t->d);
  vcp(g, t->gP, t->d);
  res->logp = t->logpP; res->n_steps = t->n; res->divergent = t->div;
  res->accept_sum = t->acc; res->depth = t->depth;
}
void exo_nt_build_full_tree(const double* q0, const double* p0, const double* g0, double logp0,
                            const double* fq, const double* fp, const double* flp, const double* fg,
                            int n_fwd, const double* bq, const double* bp, const double* blp,
                            const double* bg, int n_bwd, const double* im, double jlp0,
                            int max_depth, int d, uint64_t seed, double* q_out, double* g_out,
                            exo_tree_result* res) {
  /* tree.rs:276-326 */
  uint64_t rng[4];
  exo_xoshiro_seed_from_u64(rng, seed);
  exo_nt_traj* tr = exo_nt_init_trajectory(q0, p0, g0, logp0, d);
  size_t fc = 0, bc = 0;
  for (int it = 0; it < max_depth; it++) {
    if (exo_nt_is_terminated(tr)) break;
    int go_right = exo_xoshiro_f64(rng) > 0.5;
    size_t n = (size_t)1 << tr->depth;
    if (go_right && fc + n > (size_t)n_fwd) break;
    if (!go_right && bc + n > (size_t)n_bwd) break;
    if (go_right) {
      nt_run_subtree(tr, fq + fc * d, fp + fc * d, flp + fc, fg + fc * d, im, jlp0, tr->depth, d,
                     1, rng);
      fc += n;
    } else {
      nt_run_subtree(tr, bq + bc * d, bp + bc * d, blp + bc, bg + bc * d, im, jlp0, tr->depth, d,
                     0, rng);
      bc += n;
    }
  }
  exo_nt_get_result(tr, q_out, g_out, res);
  exo_nt_free(tr);
}
