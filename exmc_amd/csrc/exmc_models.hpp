// exmc_models.hpp — per-model log-posterior + gradient device functors.
//
// Each model M<G> exposes:
//   D, DPL                 dimension and per-lane slots
//   Consts                 chain-invariant data passed by value in the kernel arguments
//   Lane                   per-lane registers loaded once per kernel (data for the lane's dims)
//   load(c, l, lane)       fill Lane
//   logp_grad(c, lane, l, q, g) -> logp   (every lane of the group returns the same logp)
// Arithmetic order is the numeric contract shared with the CPU checker (DESIGN.md "Arithmetic
// order"); f32-rounded literals reproduce the reference's Nx.tensor(<float>) defaults.
#pragma once

#include "exmc_device.hpp"

namespace exmc {

__device__ __forceinline__ double clamp200(double z) { return fmax(-200.0, fmin(z, 200.0)); }

// kCoop: logp_grad is wave-cooperative (MFMA) and needs every lane of the wavefront active;
// kExtraLdsDoubles: LDS scratch the functor wants per wavefront (Lane::sh points at it).
// kStageDoubles: model data a single-workgroup kernel (the adaptation warmup) may copy into LDS
// (Lane::xs); with one wavefront on the whole chip every global load is an exposed round trip.
struct ModelDefaults {
  static constexpr bool kVregMath = false;   // exp / log cores with VGPR-pinned coefficients
  static constexpr bool kCoop = false;
  static constexpr int kExtraLdsDoubles = 0;
  static constexpr int kStageDoubles = 0;
  static constexpr int kStageRowOffset = 0;   // where Lane::xs points inside the image (doubles)
  // kLdsDataDoubles: observations every NUTS workgroup keeps in LDS for the whole kernel
  // (stage_data fills the image, Lane::xoff = its offset in the dynamic LDS array or -1)
  static constexpr int kLdsDataDoubles = 0;
  // single-chain warmup as two waves (tree + integrator, exmc_nuts.hpp PipeBox). Pays when the
  // tree bookkeeping is a real share of a leaf pass: sv 900 -> 695 ms, radon 194 -> 169 ms,
  // eight_schools ~20 -> ~17 ms; not for logistic, whose pass is nearly all model (158 -> 162 ms)
  static constexpr bool kPipeWarmup = true;
  static constexpr bool kHasPipeWarmup = true;   // false: the two-wave warmup form is not even built
  // 64-lane sums: the cross-row stages through ds_bpermute instead of v_readlane (exmc_device.hpp
  // group_allsum_n): for kernels bound by vector issue with two waves per SIMD
  static constexpr bool kXRowLds = false;
  // opts[:dense_mass] in the row layout (16 lanes, one dimension per lane): see RowDenseModel
  static constexpr bool kRowDense = false;
  // opts[:dense_mass] in a lane layout (a chain over G lanes, DPL dimensions per lane): see
  // LaneDenseModel; kDenseLdsDoubles = the LDS strips its contractions exchange vectors through
  static constexpr bool kLaneDense = false;
  static constexpr bool kDenseImage = false;
  static constexpr int kDenseLdsDoubles = 0;
  // Resident waves per SIMD the sampling kernel's register allocation must allow (the second
  // launch bound). 2 caps the kernel at 256 vector registers: what the allocator would have kept
  // in accumulator registers goes to scratch instead. Worth it when the configuration launches
  // more waves than SIMDs and the spilled values are per-transition state, not leaf-pass state.
  static constexpr int kNutsWavesPerSimd = 1;
  // 64 lanes per chain: the momentum draw advances the generator on the scalar unit into an orbit held
  // across the lanes (exmc_nuts.hpp draw_momentum_orbit) -- ten unrolled instructions per word of code, so
  // only where the draw is a visible share of a transition (radon: 90 draws against ~15 leapfrogs)
  static constexpr bool kRngOrbit = false;
  // one chain per wave, two waves per SIMD: a chain may move to a SIMD that has run empty while
  // its own SIMD still holds two chains (exmc_nuts.hpp "chain migration")
  static constexpr bool kMigrate = false;
  // leapfrogs per ms of a wave that shares its SIMD with one other, holding the issue priority and
  // not holding it (tools/sv_probe.sh): the time-sliced priority of the migration loop gives the
  // chain with more left to do the larger share by these; equal (unmeasured): even shares
  static constexpr double kPrioRateWith = 1.0, kPrioRateWithout = 1.0;
  // > 0: the sampling kernel runs as wave pairs too, with this many tree-stack levels in LDS
  static constexpr int kPipeNutsLevels = 0;
  // false: the sampling kernel reads the ziggurat tables from global memory instead of staging
  // them in LDS (6 KB per workgroup) -- for a model that draws a momentum once per few hundred
  // leapfrogs and spends the space on another tree-stack level (exmc_nuts.hpp nuts_lds_data_offset)
  static constexpr bool kNutsZigInLds = true;
  // true: tree nodes carry the proposal as (q, logp) only; its gradient is evaluated once more at the
  // end of the transition (exmc_nuts.hpp nuts_run) -- for deep-tree models, where one more model
  // evaluation per transition is cheaper than two more doubles in every node
  static constexpr bool kRegradProposal = false;
  // > 0: the sampling kernel also exists as workgroups of this many wavefronts that share one LDS
  // image of the model's data (stage / kStageDoubles), with kWgLdsLevels tree-stack levels per
  // wavefront in LDS (exmc_nuts.hpp nuts_kernel_wg); wg_ok(consts): the data fit the image
  static constexpr int kWgWaves = 0;
  static constexpr int kWgLdsLevels = 1;
  static constexpr int kWgImageDoubles = 0;   // the image's size; wg_stage fills it, wg_attach points a lane at it
  template <class C>
  __host__ __device__ static bool wg_ok(const C&) { return false; }
};

// the dynamic LDS of the running kernel (every extern __shared__ array names the same base)
extern __shared__ double exmc_dyn_lds[];
// two doubles of LDS behind a pointer that says so: ds_read_b128 of a 16-byte aligned pair
typedef double exmc_v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) exmc_v2d lds_v2d;
typedef __attribute__((address_space(3))) double lds_f64_m;
// the table of exmc_log_tab (include/exmc_logtab.h) in global memory: the copy a kernel stages into LDS,
// and what the layouts without an LDS image read (128 x 16 bytes: resident in the vector L1)
static __device__ const double exmc_logtab_dev[2 * EXMC_LOGTAB_ENTRIES] = {EXMC_LOGTAB_VALUES};
// exmc_log_tab from the global copy (a normal positive argument)
__device__ static __forceinline__ double exmc_log_tab_global(double x) {
  double m;
  int e;
  const int i = exmc_logtab_split(x, &m, &e);
  return exmc_logtab_finish(m, e, exmc_logtab_dev[2 * i], exmc_logtab_dev[2 * i + 1]);
}

// ------------------------------------------------------------------------------------------
// eight_schools, non-centered (benchmark/posteriordb/validate_posteriordb.exs:246-324).
// dims: 0 mu, 1 log tau, 2..9 theta_trans_0..7 (point_map.ex:37 alphabetical order).
// ------------------------------------------------------------------------------------------
struct EightSchoolsConsts {
  double y[8], sg[8], lsg[8];
  double c_mu;   // f32(log(f32(2pi))) + 2*log(5)        dist/normal.ex:19-23
  double c_hc;   // f32(log(2/pi)) - log(5)              dist/half_cauchy.ex:22-24
  double c1;     // f32(log(f32(2pi))) + 2*log(1)
};

template <int G>
struct EightSchools : ModelDefaults {
  static constexpr int D = 10;
  static constexpr int DPL = (D + G - 1) / G;
  static constexpr bool kVregMath = (DPL <= 2);
  // wave pairs in the sampling kernel: stack levels in LDS such that FOUR workgroups (eight waves:
  // two per SIMD) fit a CU -- levels x 4 KB + 6 KB of ziggurat tables + the 23 KB mailbox <= 40 KB
#ifndef EXMC_ES_PIPE_LEVELS
#define EXMC_ES_PIPE_LEVELS 2
#endif
  static constexpr int kPipeNutsLevels = (G == 16) ? EXMC_ES_PIPE_LEVELS : 0;
  using MM = Math<kVregMath>;
  using Consts = EightSchoolsConsts;
  struct Lane {
    double y[DPL], lsg[DPL];
    Recip sg[DPL];   // sigma_j as a reusable divisor
    Recip five;      // the prior scales Normal(0, 5), HalfCauchy(5)
    double cc;       // row form: the Normal prior's constant of this lane's dimension (c_mu | c1)
  };

  __device__ static __forceinline__ void load(const Consts& c, int l, Lane& ln) {
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      const int j = (i >= 2 && i < D) ? (i - 2) : 0;
      ln.y[k] = c.y[j];
      ln.sg[k] = make_recip(c.sg[j]);
      ln.lsg[k] = c.lsg[j];
    }
    ln.five = make_recip_literal(5.0);
    ln.cc = (l == 0) ? c.c_mu : c.c1;
  }

  __device__ static __forceinline__ double logp_grad(const Consts& c, const Lane& ln, int l,
                                                     const double (&q)[DPL], double (&g)[DPL]) {
    return with_fast_div([&](auto& dv) -> double { return eval(c, ln, l, q, g, dv); });
  }

  // Quotients: (y - theta) / sigma_j and mu / 5 have watched numerators (sigma_j is watched at
  // load); their quotients by the same moderate constants stay in range; tau = exp(clamp200(.))
  // lies in [e^-200, e^200], so tau / 5, 2 zt / 5 and the final quotient by 1 + zt^2 in
  // [1, e^400] have operands within 2^+-580 of each other and quotients >= e^-200: in range.
  template <class DV>
  __device__ static __forceinline__ double eval(const Consts& c, const Lane& ln, int l,
                                                const double (&q)[DPL], double (&g)[DPL], DV& dv) {
    if constexpr (kSeqSum<G, D>) return eval_row(c, ln, l, q, g, dv);
    else return eval_slots(c, ln, l, q, g, dv);
  }

  // One dimension per lane of a 16-lane row (lane 0 mu, lane 1 log tau, lanes 2..9 theta_trans_j),
  // straight-line: no lane of the row takes a branch of its own. The two hyper-parameter terms
  // share their instruction stream -- x / 5, its square, the quotient by 5 again are (mu - 0) / 5,
  // zmu^2, zmu / 5 on lane 0 and tau / 5, zt^2, zt / 5 on lane 1 ((2 zt) / 5 = 2 (zt / 5) exactly) --
  // and each lane then selects the term and the gradient entry of its own dimension. Same
  // operations on the same operands as the slot form below, so the same bits.
  template <class DV>
  __device__ static __forceinline__ double eval_row(const Consts& c, const Lane& ln, int l,
                                                    const double (&q)[DPL], double (&g)[DPL], DV& dv) {
    static_assert(DPL == 1, "one dimension per lane");
    const double mu = group_bcast_c<G, 0>(q[0]);
    const double zraw = group_bcast_c<G, 1>(q[0]);
    const double zc = clamp200(zraw);
    const double tau = MM::exp_pm200(zc);   // zc = clamp200(.)
    const bool isth = (l >= 2) && (l < D);
    const bool l0 = (l == 0), l1 = (l == 1);
    // theta lanes
    const double th = q[0];
    const double theta = mu + tau * th;
    const double resid = ln.y[0] - theta;
    dv.watch_if(isth, resid);
    dv.watch_if(isth, ln.sg[0].b);
    const double z = dv(resid, ln.sg[0]);
    const double a = dv(z, ln.sg[0]);
    const double Lk = isth ? ((-0.5 * (z * z)) - ln.lsg[0]) : 0.0;
    const double Ak = isth ? a : 0.0;
    const double Bk = isth ? (a * th) : 0.0;
    double lik = 0.0, sa = 0.0, sb = 0.0;
    row_seqsum<D>(lik, sa, sb, Lk, Ak, Bk);
    // hyper-parameter lanes
    const double x = l0 ? (mu - 0.0) : tau;
    dv.watch_if(l0, x);                      // tau = exp(clamp200(.)) is in range by construction
    const double zh = dv(x, ln.five);        // zmu | zt
    const double zh2 = zh * zh;
    const double w = dv(zh, ln.five);        // zmu / 5 | zt / 5
    const double den = 1.0 + zh2;
    const double lg = MM::log_ge1(den);      // lane 1: zt2 in [0, e^400 / 25]; elsewhere unused
    const double dhc = -dv(2.0 * w, den);
    const bool in = (zraw > -200.0) && (zraw < 200.0);
    double g_tau = in ? ((dhc + sb) * tau + 1.0) : 0.0;
    // keep the lane-1 values in the straight line: left alone, the compiler sinks them into an
    // exec-masked region of their own (five scalar instructions and a branch per leapfrog)
    __asm__("" : "+v"(g_tau));
    const double g_mu = (-w) + sa;
    const double g_th = (-th) + a * tau;
    // T = -0.5 (u^2 + cc): u = zmu with cc = c_mu on lane 0, u = theta_trans with cc = c1 on its lanes
    const double u2 = l0 ? zh2 : (th * th);
    double T = -0.5 * (u2 + ln.cc);
    double t_tau = (c.c_hc - lg) + zc;
    __asm__("" : "+v"(t_tau));
    T = l1 ? t_tau : T;
    g[0] = l0 ? g_mu : (l1 ? g_tau : g_th);
    const double Tv[1] = {T};
    const bool valid[1] = {l < D};
    return group_sum_slots<G, DPL, D>(Tv, valid, l, lik);
  }

  template <class DV>
  __device__ static __forceinline__ double eval_slots(const Consts& c, const Lane& ln, int l,
                                                      const double (&q)[DPL], double (&g)[DPL], DV& dv) {
    const double mu = group_bcast_c<G, 0 % G>(q[0 / G]);
    const double zraw = group_bcast_c<G, 1 % G>(q[1 / G]);
    const double zc = clamp200(zraw);
    const double tau = MM::exp_pm200(zc);   // zc = clamp200(.)
    dv.watch(mu);
    double L[DPL], A[DPL], B[DPL], T[DPL];
    bool valid[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      valid[k] = i < D;
      const bool isth = (i >= 2) && (i < D);
      const double th = q[k];
      const double theta = mu + tau * th;
      const double resid = ln.y[k] - theta;
      dv.watch_if(isth, resid);
      dv.watch_if(isth, ln.sg[k].b);
      const double z = dv(resid, ln.sg[k]);
      const double a = dv(z, ln.sg[k]);
      L[k] = isth ? ((-0.5 * (z * z)) - ln.lsg[k]) : 0.0;
      A[k] = isth ? a : 0.0;
      B[k] = isth ? (a * th) : 0.0;
      T[k] = -0.5 * (th * th + c.c1);
      g[k] = (-th) + a * tau;
    }
    // three independent group sums in one pass: in lane order for a kSeqSum group (0 + L[0] +
    // L[1] + ..., the reference's left-to-right sum), else lane partials and one butterfly
    double s3[3] = {0.0, 0.0, 0.0};
    if constexpr (kSeqSum<G, D>) {
      row_seqsum<D>(s3[0], s3[1], s3[2], L[0], A[0], B[0]);
    } else {
#pragma unroll
      for (int k = 0; k < DPL; k++) {
        s3[0] = valid[k] ? (s3[0] + L[k]) : s3[0];
        s3[1] = valid[k] ? (s3[1] + A[k]) : s3[1];
        s3[2] = valid[k] ? (s3[2] + B[k]) : s3[2];
      }
      group_allsum_n<G, 3>(s3);
    }
    const double lik = s3[0], sa = s3[1], sb = s3[2];
    const double zmu = dv(mu - 0.0, ln.five);
    const double t_mu = -0.5 * (zmu * zmu + c.c_mu);
    const double zt = dv(tau, ln.five);
    const double zt2 = zt * zt;
    const double t_tau = (c.c_hc - MM::log_ge1(1.0 + zt2)) + zc;   // zt2 in [0, e^400 / 25]
    const double g_mu = (-dv(zmu, ln.five)) + sa;
    const double dhc = -dv(dv(2.0 * zt, ln.five), 1.0 + zt2);
    const bool in = (zraw > -200.0) && (zraw < 200.0);
    const double g_tau = in ? ((dhc + sb) * tau + 1.0) : 0.0;
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      if (i == 0) { T[k] = t_mu; g[k] = g_mu; }
      if (i == 1) { T[k] = t_tau; g[k] = g_tau; }
    }
    return group_sum_slots<G, DPL, D>(T, valid, l, lik);
  }
};

// ------------------------------------------------------------------------------------------
// simple (d=2): mu ~ N(0,5); sigma ~ Exponential(1) [:log]; y_i ~ N(mu, sigma) (SURVEY 8d;
// data README.md:72-74, f32 literals). One lane per chain only.
// ------------------------------------------------------------------------------------------
struct SimpleConsts {
  double y[64];
  int n;
  double c_mu;      // f32(log(f32(2pi))) + 2*log(5)
  double log2pi32;  // f32(log(f32(2pi)))
  double tiny32;    // f32(1e-30)
};

template <int G>
struct Simple : ModelDefaults {
  static_assert(G == 1, "simple model is one lane per chain");
  static constexpr int D = 2;
  static constexpr int DPL = 2;
  using Consts = SimpleConsts;
  struct Lane {};
  __device__ static __forceinline__ void load(const Consts&, int, Lane&) {}
  __device__ static __forceinline__ double logp_grad(const Consts& c, const Lane&, int,
                                                     const double (&q)[DPL], double (&g)[DPL]) {
    const double mu = q[0], zc = clamp200(q[1]);
    const double sigma = exmc_exp(zc);
    const double ss = fmax(sigma, c.tiny32);
    const double ls = exmc_log(ss);
    const double zmu = (mu - 0.0) / 5.0;
    const double t_mu = -0.5 * (zmu * zmu + c.c_mu);
    const double t_sig = (0.0 - 1.0 * sigma) + zc;
    const double cn = c.log2pi32 + 2.0 * ls;
    double obs = 0.0, sa = 0.0, sb = 0.0;
    for (int i = 0; i < c.n; i++) {
      const double z = (c.y[i] - mu) / ss;
      obs = obs + (-0.5 * (z * z + cn));
      sa = sa + z / ss;
      sb = sb + (z * z - 1.0);
    }
    const bool in = (q[1] > -200.0) && (q[1] < 200.0);
    g[0] = (-(zmu / 5.0)) + sa;
    g[1] = in ? ((sb - sigma) + 1.0) : 0.0;
    return (t_mu + t_sig) + obs;
  }
};

// ------------------------------------------------------------------------------------------
// stochastic volatility, T = 100 (STANDARD_BENCHMARKS.md:51-61). Kernel order: dims 0..T-1 =
// s_1..s_T, T = log sigma, T+1 = log nu. Neighbouring s_t live on neighbouring lanes.
// ------------------------------------------------------------------------------------------
struct SVConsts {
  double r[100];
  double lanczos[9];     // f32-rounded (math.ex:10-20 as Nx.tensor)
  double half_log_2pi32; // f32(0.5*log(2pi))
  double log2pi32, pi32, tiny32;
  double lam_s, lam_n, log_lam_s32, log_lam_n32;
};

// Quotients (caller watches x in [2^-101, 2^100)): den = x + j >= x, the f32 Lanczos
// coefficients lie in [2^-20, 2^11), so term in 2^(-121..112) and term / den in 2^(-221..213);
// x - 0.5, ag and dag are watched here; t = x + 6.5 in [6.5, 2^101).
// Both lgamma evaluations of a leapfrog (x1 = (nu+1)/2, x0 = nu/2) at once. The 16 series terms
// c_i / (x + i - 1) and their derivatives are one quotient pair on 16 different lanes (lane l:
// call l / 8, term l % 8 + 1; lzc / lzj are that lane's coefficient and offset), summed in the
// reference's order from broadcasts; the four logarithms of the two calls plus the two the
// caller needs (extra[0..1] in, their logs out) are two lane-batched evaluations.
// Every 16-lane row of the wavefront holds the same sixteen (term, tdd) pairs (the coefficient
// depends on l & 7, the argument on l & 8), so the four sums run inside each row as
// v_fmac_f64_dpp row_newbcast chains -- acc + t = fma(t, 1, acc), acc - t = fma(t, -1, acc),
// one instruction per term, in the reference's order -- and every lane ends up with all four
// without a cross-row broadcast (the v_readlane form cost three instructions per term).
#define EXMC_LZ1(I) \
  "v_fmac_f64_dpp %[a1], %[t], %[one] row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f64_dpp %[d1], %[d], %[mone] row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
#define EXMC_LZ0(I) \
  "v_fmac_f64_dpp %[a0], %[t], %[one] row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f64_dpp %[d0], %[d], %[mone] row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
template <class MM, int G, int... I>
__device__ __forceinline__ void lanczos_sums(double term, double tdd, double c0, double& ag1,
                                             double& dag1, double& ag0, double& dag0,
                                             std::integer_sequence<int, I...>) {
  ag1 = c0; dag1 = 0.0; ag0 = c0; dag0 = 0.0;
  const double one = seq_one();
  double mone = -1.0;
  __asm__("" : "+v"(mone));
  __asm__("s_nop 1\n\t"
          EXMC_LZ1(0) EXMC_LZ1(1) EXMC_LZ1(2) EXMC_LZ1(3) EXMC_LZ1(4) EXMC_LZ1(5) EXMC_LZ1(6) EXMC_LZ1(7)
          EXMC_LZ0(8) EXMC_LZ0(9) EXMC_LZ0(10) EXMC_LZ0(11) EXMC_LZ0(12) EXMC_LZ0(13) EXMC_LZ0(14) EXMC_LZ0(15)
          : [a1] "+v"(ag1), [d1] "+v"(dag1), [a0] "+v"(ag0), [d0] "+v"(dag0)
          : [t] "v"(term), [d] "v"(tdd), [one] "v"(one), [mone] "v"(mone));
}

template <class MM, int G, class DV>
__device__ __forceinline__ void lanczos_pair_d(double lz0, double half_log_2pi, int l, double lzc, double lzj,
                                               double x1, double x0, double (&extra)[2],
                                               double& lg1, double& d1, double& lg0, double& d0,
                                               DV& dv) {
  static_assert(G >= 16, "one lane per series term of the two calls");
  const double xs = ((l & 8) == 0) ? x1 : x0;
  const Recip den = make_recip(xs + lzj * 1.0);
  const double term = dv(lzc, den);
  const double tdd = dv(term, den);
  double ag1, dag1, ag0, dag0;
  lanczos_sums<MM, G>(term, tdd, lz0, ag1, dag1, ag0, dag0,
                      std::make_integer_sequence<int, 8>{});
  // The six logarithms in their short form while the wavefront is inside the fast window: every
  // argument is then positive, finite and normal -- t = x + 6.5 >= 6.5; the caller's two are a
  // watched scale and a watched degree of freedom times pi; the series sums are watched below and
  // checked to be positive -- which is the domain of the main path (exmc_log_ge1 is that path plus a
  // NaN fix-up; its name states the caller it was written for). Same bits as exmc_log there.
  constexpr bool kFast = std::is_same_v<DV, Div<true>>;
  const double t1 = x1 + 6.5, t0 = x0 + 6.5;
  double la[4] = {t1, t0, extra[0], extra[1]};
  if constexpr (kFast) lane_batch<G, 4>(la, l, [](double v) { return MM::log_ge1(v); });
  else lane_batch<G, 4>(la, l, [](double v) { return MM::log(v); });
  extra[0] = la[2];
  extra[1] = la[3];
  double lb[2] = {ag1, ag0};
  if constexpr (kFast) {
    dv.ok = dv.ok && (ag1 > 0.0) && (ag0 > 0.0);
    lane_batch<G, 2>(lb, l, [](double v) { return MM::log_ge1(v); });
  } else {
    lane_batch<G, 2>(lb, l, [](double v) { return MM::log(v); });
  }
  const double xm1 = x1 - 0.5, xm0 = x0 - 0.5;
  dv.watch(xm1); dv.watch(xm0);
  dv.watch(ag1); dv.watch(ag0);
  dv.watch(dag1); dv.watch(dag0);
  d1 = ((la[0] + dv(xm1, t1)) - 1.0) + dv(dag1, ag1);
  d0 = ((la[1] + dv(xm0, t0)) - 1.0) + dv(dag0, ag0);
  lg1 = ((half_log_2pi + xm1 * la[0]) - t1) + lb[0];
  lg0 = ((half_log_2pi + xm0 * la[1]) - t0) + lb[1];
}

template <int G>
struct SV : ModelDefaults {
  static constexpr int T = 100;
  static constexpr int D = T + 2;
  static constexpr int DPL = (D + G - 1) / G;
  static_assert(G >= 16, "sv spreads a chain over >= 16 lanes (lane-batched lgamma series)");
  // G = 64: 2048 chains are 2048 waves on 1024 SIMDs. At 304 registers one wave is resident per
  // SIMD and issues every ~5 clocks; capped at 256 (49 dwords of per-transition state in scratch,
  // one scratch access inside the leaf loop) two are, and the pair issues every ~3.5
  // (4096 chains x 200 draws: 1307 -> 950 ms)
  static constexpr int kNutsWavesPerSimd = (G == 64) ? 2 : 1;
#ifndef EXMC_SV_XROW_LDS
#define EXMC_SV_XROW_LDS 1
#endif
  static constexpr bool kXRowLds = (G == 64) && (EXMC_SV_XROW_LDS != 0);
  static constexpr bool kMigrate = (G == 64);
  // G = 64: a transition is ~340 leapfrogs and one momentum draw; the 6 KB of the tables buy the
  // third stack level in LDS (nodes of 11 doubles since kRegradProposal: 3 x 5.5 KB = 16.5 KB, still
  // eight workgroups per CU), which halves
  // the visits to the spill stack in global memory (a quarter of the leaves merge or park at level
  // >= 2, an eighth at level >= 3)
  static constexpr bool kNutsZigInLds = (G != 64);
  static constexpr bool kRegradProposal = (G == 64);
  static constexpr double kPrioRateWith = 291.0, kPrioRateWithout = 169.0;   // profiles/r3_sv_prio
  using Consts = SVConsts;
  struct Lane {
    double r[DPL];
    double lzc, lzj;   // this lane's Lanczos coefficient and offset (lanczos_pair_d)
    // the model's scalar constants as vector registers (the same value in every lane): as scalar
    // registers they are spilled in 16-dword tuples around the leaf loop and every use pays a
    // v_readlane per dword (400 of the kernel's 1400 vector instructions per leapfrog)
    double k[9];
  };
  enum { kTiny = 0, kLogLamS, kLamS, kLogLamN, kLamN, kPi, kLog2Pi, kLz0, kHalfLog2Pi };
  __device__ static __forceinline__ double pin(double x) {
    asm volatile("" : "+v"(x));
    return x;
  }
  __device__ static __forceinline__ void load(const Consts& c, int l, Lane& ln) {
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      ln.r[k] = c.r[i < T ? i : 0];
    }
    ln.lzc = c.lanczos[(l & 7) + 1];
    ln.lzj = (double)(l & 7);
    ln.k[kTiny] = pin(c.tiny32); ln.k[kLogLamS] = pin(c.log_lam_s32); ln.k[kLamS] = pin(c.lam_s);
    ln.k[kLogLamN] = pin(c.log_lam_n32); ln.k[kLamN] = pin(c.lam_n); ln.k[kPi] = pin(c.pi32);
    ln.k[kLog2Pi] = pin(c.log2pi32); ln.k[kLz0] = pin(c.lanczos[0]); ln.k[kHalfLog2Pi] = pin(c.half_log_2pi32);
  }
  static constexpr bool kVregMath = true;
  using MM = Math<kVregMath>;

  __device__ static __forceinline__ double logp_grad(const Consts& c, const Lane& ln, int l,
                                                     const double (&q)[DPL], double (&g)[DPL]) {
    return with_fast_div([&](auto& dv) -> double { return eval(c, ln, l, q, g, dv); });
  }

  // Quotients: sigma and nu are watched in [2^-100, 2^100), which puts the Lanczos arguments
  // nu/2, (nu+1)/2 in [2^-101, 2^100); per slot the AR(1) residual and z^2 are watched in
  // 2^+-250, so e = resid / sigma, w = z^2 / nu stay within 2^+-350, 1 + w in [1, 2^351),
  // wr = w / (1 + w) in (2^-351, 1], hp1 * wr in (2^-352, 2^100): every operand of every
  // quotient is inside the 2^+-380 window.
  template <class DV>
  __device__ static __forceinline__ double eval(const Consts& c, const Lane& ln, int l,
                                                const double (&q)[DPL], double (&g)[DPL], DV& dv) {
    const double zs_raw = group_bcast_c<G, T % G>(q[T / G]);
    const double zn_raw = group_bcast_c<G, (T + 1) % G>(q[(T + 1) / G]);
    constexpr bool kFast = std::is_same_v<DV, Div<true>>;
    const double zs = clamp200(zs_raw), zn = clamp200(zn_raw);
    double ez[2] = {zs, zn};
    lane_batch<G, 2>(ez, l, [](double v) { return MM::exp_pm200(v); });   // clamp200'ed: always in range
    const double sigma = ez[0], nu = ez[1];
    const double ss = fmax(sigma, ln.k[kTiny]);
    const double sdf = fmax(nu, ln.k[kTiny]);
    dv.template watch_exp_if<-100, 100>(true, ss);
    dv.template watch_exp_if<-100, 100>(true, sdf);
    const Recip rss = make_recip(ss), rsdf = make_recip(sdf);
    const double t_sigma = (ln.k[kLogLamS] - ln.k[kLamS] * sigma) + zs;
    const double t_nu = (ln.k[kLogLamN] - ln.k[kLamN] * nu) + zn;
    const double hp1 = (sdf + 1.0) / 2.0, h = sdf / 2.0;
    double d1, d0, lg1, lg0;
    double lx[2] = {sdf * ln.k[kPi], ss};   // in: arguments, out: their logarithms
    lanczos_pair_d<MM, G>(ln.k[kLz0], ln.k[kHalfLog2Pi], l, ln.lzc, ln.lzj,
                          hp1, h, lx, lg1, d1, lg0, d0, dv);
    const double An = (lg1 - lg0) - 0.5 * lx[0];
    const double dAn = (0.5 * d1 - 0.5 * d0) - dv(0.5, rsdf);
    const double cn = ln.k[kLog2Pi] + 2.0 * lx[1];
    double P[DPL], LL[DPL], E2[DPL], DN[DPL], de[DPL];
    bool valid[DPL];
    // previous-state values: dim i-1 is (lane l-1, slot k) or (lane G-1, slot k-1) when l == 0:
    // one rotate of the group per slot delivers both (lane 0 receives lane G-1's value)
    double qrot[DPL], qprev[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) qrot[k] = group_rot_prev<G>(q[k]);
#pragma unroll
    for (int k = 0; k < DPL; k++) qprev[k] = (l > 0) ? qrot[k] : ((k > 0) ? qrot[k > 0 ? k - 1 : 0] : 0.0);
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      valid[k] = i < D;
      const bool ist = i < T;
      const double qi = q[k];
      const double prev = (i == 0) ? 0.0 : qprev[k];
      const double resid = qi - prev;
      dv.template watch_exp_if<-250, 250>(ist, resid);
      const double e = dv(resid, rss);
      // the two specials of a slot in their short forms while the wavefront is inside the fast
      // window (a state below 2^-380 or above 2^7 in magnitude re-evaluates the pass with the
      // general forms, like an out-of-range quotient; same bits on the domain): exp(-s_t) without
      // its special-case guards, and log(1 + w) with 1 + w in [1, 2^351) by the watches above
      dv.template watch_exp_if<-380, 7>(ist, qi);
      const double z = ln.r[k] * (kFast ? MM::exp_pm200(-qi) : MM::exp(-qi));
      const double zz = z * z;
      dv.template watch_exp_if<-250, 250>(ist, zz);
      const double w = dv(zz, rsdf);
      const double lg = kFast ? MM::log_ge1(1.0 + w) : MM::log(1.0 + w);
      const double wr = dv(w, 1.0 + w);
      P[k] = ist ? (-0.5 * (e * e + cn)) : 0.0;
      E2[k] = ist ? (e * e - 1.0) : 0.0;
      de[k] = ist ? (-dv(e, rss)) : 0.0;
      LL[k] = ist ? ((An - qi) - hp1 * lg) : 0.0;
      DN[k] = ist ? ((dAn - 0.5 * lg) + dv(hp1 * wr, rsdf)) : 0.0;
      g[k] = -1.0 + (sdf + 1.0) * wr;
    }
    // dP_{t+1}/ds_{t+1} from dim i+1: (lane l+1, slot k) or (lane 0, slot k+1) when l == G-1
    double drot[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) drot[k] = group_rot_next<G>(de[k]);
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      const double same = drot[k];
      const double upper = (k + 1 < DPL) ? drot[k + 1 < DPL ? k + 1 : k] : 0.0;
      double nxt = (l < G - 1) ? same : upper;
      nxt = (i + 1 < T) ? nxt : 0.0;
      g[k] = g[k] + (de[k] - nxt);
    }
    double sp, sl, se, sn;
    if constexpr (G == 64 && kXRowLds) {
      // the four sums as ONE reduce-scatter (exmc_device.hpp rs64_allsum4): the lane partials of
      // group_sum_slots, then the same xor-butterfly additions with a lane carrying two, then one
      // value instead of four -- the same totals, bit for bit, for about half the instructions
      double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < DPL; k++) {
        s4[0] = valid[k] ? (s4[0] + P[k]) : s4[0];
        s4[1] = valid[k] ? (s4[1] + LL[k]) : s4[1];
        s4[2] = valid[k] ? (s4[2] + E2[k]) : s4[2];
        s4[3] = valid[k] ? (s4[3] + DN[k]) : s4[3];
      }
      rs64_allsum4(s4);
      sp = s4[0]; sl = s4[1]; se = s4[2]; sn = s4[3];
    } else {
      sp = group_sum_slots<G, DPL, G * DPL, kXRowLds>(P, valid, l, 0.0);
      sl = group_sum_slots<G, DPL, G * DPL, kXRowLds>(LL, valid, l, 0.0);
      se = group_sum_slots<G, DPL, G * DPL, kXRowLds>(E2, valid, l, 0.0);
      sn = group_sum_slots<G, DPL, G * DPL, kXRowLds>(DN, valid, l, 0.0);
    }
    const bool in_s = (zs_raw > -200.0) && (zs_raw < 200.0);
    const bool in_n = (zn_raw > -200.0) && (zn_raw < 200.0);
    const double g_s = in_s ? ((se - ln.k[kLamS] * sigma) + 1.0) : 0.0;
    const double g_n = in_n ? ((sn * nu - ln.k[kLamN] * nu) + 1.0) : 0.0;
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      if (i == T) g[k] = g_s;
      if (i == T + 1) g[k] = g_n;
    }
    return ((t_sigma + t_nu) + sp) + sl;
  }
};

// ------------------------------------------------------------------------------------------
// logistic regression, K = 20 covariates, N observations (STANDARD_BENCHMARKS.md:41-49;
// Bernoulli clip dist/bernoulli.ex:17-27). dims: 0 alpha, 1..20 beta_j. Observation n is
// handled by lane n mod G of the chain's group (row n of X stays in registers for both the
// linear predictor and the gradient accumulation); contractions are fma chains.
// ------------------------------------------------------------------------------------------
struct LogisticConsts {
  const double* X;   // dev [N][K] row-major
  const double* y;   // dev [N]
  int N;
  double c10;        // f32(log(f32(2pi))) + 2*log(10)
  double lo, hi;     // f32(1e-7), 1 - f32(1e-7)
  // MFMA path (G = 4): design matrix augmented with a ones column (feature 0), zero padded
  const double* XaT;   // dev [24][Npad]  feature-major: A operand of eta = Xa @ beta
  const double* Xa32;  // dev [Npad][32]  observation-major: A operand of grad = Xa^T @ r
  const double* ypad;  // dev [Npad]
  int Npad;            // N rounded up to a multiple of 16
};

// 1: the next step's row is read into a second buffer before this step's arithmetic. Measured on
// the workgroup form (profiles/r5_lg_wg/ab_row_ahead.log): 156.0 against 155.2 ms -- the second wave
// of the SIMD already covers the LDS round trip, the kernel is bound by vector issue. Off.
#ifndef EXMC_LG_ROW_AHEAD
#define EXMC_LG_ROW_AHEAD 0
#endif
// 1: the sampling kernel reads the (c, l) pair of exmc_log_tab from the LDS image; 0: from the global copy
// (vector L1). Measured (profiles/r6_lg_logtab): 2.578 ns per leapfrog before the table, 2.547 with the
// global copy, 2.456 with the LDS copy.
#ifndef EXMC_LG_LOGTAB_LDS
#define EXMC_LG_LOGTAB_LDS 1
#endif
// 1: the 16-lane layout with an LDS image checks the short forms' domain per PASS (a sticky lane mask, one
// branch at the end of the pass); 0: per step of sixteen observations (a compare and two branches each).
#ifndef EXMC_LG_PASS_GUARD
#define EXMC_LG_PASS_GUARD 1
#endif
template <int G>
struct Logistic : ModelDefaults {
  static constexpr bool kPipeWarmup = false;
  // G = 16: 8192 chains are 2048 waves; two resident waves per SIMD at 256 registers + scratch
  // beat one at 364 (8192 x 200 draws: 63.5 -> 53.8 ms)
  static constexpr int kNutsWavesPerSimd = (G == 16) ? 2 : 1;
  static constexpr int K = 20;
  static constexpr int D = K + 1;
  static constexpr int DPL = (D + G - 1) / G;
  using Consts = LogisticConsts;
  // LDS image of X and y: row n = x[n][0..19], y[n], one pad double. 22 doubles = 176 bytes, so
  // every row is 16-byte aligned (ds_read_b128) and sixteen consecutive rows start 44 dwords apart:
  // in sixteen different groups of four banks, the 16-lane groups of a b128 read are conflict-free.
  // Rows N..kObsCap-1 are zero. For single-workgroup kernels (the adaptation warmup: with one wave
  // on the whole chip every row of X is otherwise an exposed L2 round trip, 32 times per leapfrog)
  // and for the sampling kernel in its workgroup form (kWgWaves, exmc_nuts.hpp nuts_kernel_wg).
  static constexpr int kObsCap = 512;
  static constexpr int kRowStride = K + 2;
  // ... behind the 128 (c, l) pairs of the table-driven logarithm (exmc_detmath.h exmc_log_tab): the
  // table comes FIRST so that in the workgroup form of the sampling kernel it lies below 64 KB and the
  // pair's ds_read_b128 takes its base as the instruction's 16-bit offset. Lane::xs points at the rows.
  static constexpr int kRowDoubles = kObsCap * kRowStride;
  static constexpr int kStageRowOffset = 2 * EXMC_LOGTAB_ENTRIES;
  static constexpr int kStageDoubles = kStageRowOffset + kRowDoubles;
  // G = 16: the sampling kernel as workgroups of eight wavefronts (two per SIMD, one workgroup per
  // compute unit) that share ONE image: 8192 chains x 16 lanes are exactly 256 such workgroups.
  // 88 KB of image + 6 KB of ziggurat tables leave one tree-stack level per wavefront in LDS
  // (8 x 6.5 KB); a 7-leapfrog tree parks one node per transition on level 1 (L2-resident).
  static constexpr int kWgWaves = (G == 16) ? 8 : 0;
  static constexpr int kWgLdsLevels = 1;
  __host__ __device__ static bool wg_ok(const Consts& c) { return c.N <= kObsCap; }
  static constexpr int kWgImageDoubles = (G == 16) ? kStageDoubles : 0;
  struct Lane {
    const double* xs;   // LDS image, or null (rows stream from L2)
  };
  __device__ static __forceinline__ void wg_stage(const Consts& c, double* image) { stage(c, image); }
  __device__ static __forceinline__ void wg_attach(Lane& ln, double* image, int) { ln.xs = image + kStageRowOffset; }
  __device__ static __forceinline__ void load(const Consts&, int, Lane& ln) { ln.xs = nullptr; }
  // cooperative (whole workgroup); the caller synchronises afterwards and guarantees N <= kObsCap
  __device__ static __forceinline__ void stage(const Consts& c, double* image) {
    for (int i = threadIdx.x; i < 2 * EXMC_LOGTAB_ENTRIES; i += blockDim.x) image[i] = exmc_logtab_dev[i];
    double* const dst = image + kStageRowOffset;
    for (int i = threadIdx.x; i < kRowDoubles; i += blockDim.x) {
      const int n = i / kRowStride, j = i % kRowStride;
      double v = 0.0;
      if (n < c.N) v = (j < K) ? c.X[(size_t)n * K + j] : ((j == K) ? c.y[n] : 0.0);
      dst[i] = v;
    }
  }
  // log of a clipped probability (normal, positive): exmc_log_tab with the pair fetched from the LDS
  // image (img != null) or from the global copy
  template <bool kStaged>
  __device__ static __forceinline__ double log_tab(const lds_v2d* img, double x) {
    double m;
    int e;
    const int i = exmc_logtab_split(x, &m, &e);
    exmc_v2d cl;
    if constexpr (kStaged && EXMC_LG_LOGTAB_LDS != 0) cl = img[i - EXMC_LOGTAB_ENTRIES];   // the table sits in front of the rows
    else cl = ((const exmc_v2d*)exmc_logtab_dev)[i];
    return exmc_logtab_finish_s(m, e, cl[0], cl[1]);
  }

  // every lane needs the whole coefficient vector: dim i lives in slot i / G of lane i % G
  template <int... I>
  __device__ static __forceinline__ void bcast_all(const double (&q)[DPL], double (&qf)[D],
                                                   std::integer_sequence<int, I...>) {
    ((qf[I] = group_bcast_c<G, I % G>(q[I / G])), ...);
  }

  __device__ static __forceinline__ double logp_grad(const Consts& c, const Lane& ln, int l,
                                                     const double (&q)[DPL], double (&g)[DPL]) {
    // the short forms (below) where the kernel has registers for the two copies of the pass: the
    // one-chain-per-wave layout of the shared warmup (61.8 -> 54.2 ms). The 16-lane sampling layout
    // decides per observation step instead (eval_row16)
    if constexpr (G == 64) {
      return with_fast_div([&](auto& dv) -> double { return eval(c, ln, l, q, g, dv); });
    } else if constexpr (G == 16) {
      return eval_row16(c, ln, l, q, g);
    } else {
      Div<false> exact;
      return eval(c, ln, l, q, g, exact);
    }
  }

  // ---- the 16-lane layout: a chain is one DPP row (round 5) ----
  // Same operations in the same order as eval() below (the numeric contract of logp_logistic in the
  // checker is untouched); what differs is how they are issued:
  //  * the linear predictor reads the coefficients where they live -- dimension i in lane i % 16,
  //    slot i / 16 of the chain's row -- through v_fmac_f64_dpp row_newbcast: eta = fma(beta_j[lane],
  //    x_j, eta) is one instruction per feature and no lane keeps a copy of all 21 coefficients
  //    (42 registers, and the 42 DPP moves that filled them every leapfrog);
  //  * the three specials of an observation in their short forms (exp without its guards, the
  //    quotient through the refined reciprocal, log of a clipped probability: the main path alone)
  //    with the polynomial coefficients in scalar registers (exmc_detmath.h, _s cores), decided PER
  //    STEP of sixteen observations: a wavefront whose predictors all lie in [-200, 200] takes the
  //    short forms, otherwise that step alone takes the general ones. Same bits on the domain;
  //  * every lane runs every step (the row's DPP reads need their source lanes active); a lane
  //    whose observation index is past N reads a zero row and adds r = 0 and ll = 0, which leave
  //    the (never negative-zero) partial sums as they are.
  __device__ static __forceinline__ double eta_row16(double q0, double q1, const double (&x)[K]) {
    static_assert(G != 16 || (K == 20 && DPL == 2), "dims 0..15 in slot 0, 16..20 in slot 1");
    double e = 0.0;
    const double one = seq_one();
    __asm__("s_nop 1\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[one] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x0] row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x1] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x2] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x3] row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x4] row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x5] row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x6] row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x7] row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x8] row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x9] row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x10] row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x11] row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x12] row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x13] row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q0], %[x14] row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q1], %[x15] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q1], %[x16] row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q1], %[x17] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q1], %[x18] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[e], %[q1], %[x19] row_newbcast:4 row_mask:0xf bank_mask:0xf"
            : [e] "+v"(e)
            : [q0] "v"(q0), [q1] "v"(q1), [one] "v"(one), [x0] "v"(x[0]), [x1] "v"(x[1]),
              [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]),
              [x7] "v"(x[7]), [x8] "v"(x[8]), [x9] "v"(x[9]), [x10] "v"(x[10]), [x11] "v"(x[11]),
              [x12] "v"(x[12]), [x13] "v"(x[13]), [x14] "v"(x[14]), [x15] "v"(x[K > 15 ? 15 : 0]),
              [x16] "v"(x[K > 16 ? 16 : 0]), [x17] "v"(x[K > 17 ? 17 : 0]),
              [x18] "v"(x[K > 18 ? 18 : 0]), [x19] "v"(x[K > 19 ? 19 : 0]));
    return e;
  }

  // a row of the design matrix and its response, as a step holds them
  struct Row {
    double x[K], y;
  };
  template <bool kStaged>
  __device__ static __forceinline__ void load_row(const Consts& c, const lds_v2d* img, int n, bool live, Row& r) {
    if constexpr (kStaged) {
      const lds_v2d* const row = img + n * (kRowStride / 2);
#pragma unroll
      for (int j = 0; j < K / 2; j++) {
        const exmc_v2d v = row[j];
        r.x[2 * j] = v[0];
        r.x[2 * j + 1] = v[1];
      }
      // (8 bytes, not the pair with the pad: a dead half would be a register the allocator hands to
      // something else while the read is still in flight -- a wait in front of that something)
      r.y = *(const lds_f64_m*)(row + K / 2);
    } else {
      const int nr = live ? n : (c.N - 1);
      const double* x = c.X + (size_t)nr * K;
#pragma unroll
      for (int j = 0; j < K; j++) r.x[j] = x[j];
      r.y = c.y[nr];
    }
  }
  // one step: observations it * 16 + l of the row's sixteen lanes, from the row already in registers.
  // kTail: the last, partly filled step (N mod 16 lanes carry an observation): `live` masks the two
  // contributions; the full steps carry no mask at all.
  // kMode 0: short forms or general forms decided in this step (one compare and two branches per step);
  // 1: short forms whatever the predictors are, the lanes outside [-200, 200] reported in `out` (the caller
  // evaluates the pass again in mode 2 if any was: same bits, decided once per pass); 2: general forms.
  template <bool kTail, bool kStaged, int kMode = 0>
  __device__ static __forceinline__ void step_row16(const Row& row, const lds_v2d* img, bool live, double q0, double q1,
                                                    double (&s)[D + 1], unsigned long long* out = nullptr) {
    constexpr double kLo = (double)1.0e-7f, kHi = 1.0 - (double)1.0e-7f;   // = Consts::lo, hi
    const double yn = row.y;
    const double eta = eta_row16(q0, q1, row.x);
    double p, ll;
    const bool inr = fabs(eta) <= 200.0;   // false for a NaN
    bool fast;
    if constexpr (kMode == 0) fast = __builtin_expect(__builtin_amdgcn_ballot_w64(!inr) == 0, 1);
    else fast = kMode == 1;
    if constexpr (kMode == 1) *out |= __builtin_amdgcn_ballot_w64(!inr);
    if (fast) {
      const double u = 1.0 + exmc_exp_pm200_s(-eta);   // [1, e^200 + 1): inside the division window
      p = exmc_div_core(1.0, u, exmc_rcp_refined(u));
      const double pc = fmin(fmax(p, kLo), kHi);
      // y = 1: pc, y = 0: 1 - pc, as ONE fma(pc, 2y - 1, 1 - y): pc * 1 + 0 and pc * (-1) + 1 are the
      // two values exactly (a compare and two selects less); [1e-7, 1 - 1e-7]: normal, positive
      const double sgn = __builtin_fma(yn, 2.0, -1.0), off = 1.0 - yn;
      ll = log_tab<kStaged>(img, __builtin_fma(pc, sgn, off));
    } else {
      p = 1.0 / (1.0 + exmc_exp(-eta));
      const double pc = fmin(fmax(p, kLo), kHi);
      ll = log_tab<kStaged>(img, (yn == 1.0) ? pc : (1.0 - pc));
    }
    const bool in = p > kLo && p < kHi;
    const double r = (kTail ? (live && in) : in) ? (yn - p) : 0.0;
    s[D] = s[D] + ((!kTail || live) ? ll : 0.0);
    s[0] = __builtin_fma(1.0, r, s[0]);
#pragma unroll
    for (int j = 0; j < K; j++) s[1 + j] = __builtin_fma(row.x[j], r, s[1 + j]);
  }
  // all steps of a pass; kAhead: the next step's row is requested before this step's arithmetic (the
  // LDS image: a ds_read_b128 burst whose latency the specials cover; from L2 the second row buffer
  // costs more in registers than the wait it hides)
  template <bool kStaged, bool kAhead, int kMode = 0>
  __device__ static __forceinline__ void steps_row16(const Consts& c, const lds_v2d* img, int l, double q0,
                                                     double q1, double (&s)[D + 1], unsigned long long* out = nullptr) {
    const int steps = (c.N + 15) >> 4;
    if constexpr (kAhead) {
      // two row buffers taking turns (a rotating copy would be 21 register moves per step): the row of
      // step it + 1 is requested before the arithmetic of step it
      const int full = c.N >> 4;
      Row a, b;
      load_row<kStaged>(c, img, l, l < c.N, a);
      int it = 0;
      for (; it + 2 <= full; it += 2) {                      // steps it and it + 1 are full
        const int n = l + (it << 4);
        load_row<kStaged>(c, img, n + 16, true, b);
        step_row16<false, kStaged, kMode>(a, img, true, q0, q1, s, out);
        if (it + 2 < steps) load_row<kStaged>(c, img, n + 32, n + 32 < c.N, a);   // wave-uniform
        step_row16<false, kStaged, kMode>(b, img, true, q0, q1, s, out);
      }
      const int n = l + (it << 4);
      if (it < full) {                                       // one full step left, then perhaps the tail
        if (it + 1 < steps) load_row<kStaged>(c, img, n + 16, n + 16 < c.N, b);
        step_row16<false, kStaged, kMode>(a, img, true, q0, q1, s, out);
        if (it + 1 < steps) step_row16<true, kStaged, kMode>(b, img, n + 16 < c.N, q0, q1, s, out);
      } else if (it < steps) {                               // the tail step's row is in a
        step_row16<true, kStaged, kMode>(a, img, n < c.N, q0, q1, s, out);
      }
    } else {
      const int full = c.N >> 4;                           // steps in which every lane has an observation
      int n = l;
      for (int it = 0; it < full; it++, n += 16) {
        Row cur;
        load_row<kStaged>(c, img, n, true, cur);
        step_row16<false, kStaged, kMode>(cur, img, true, q0, q1, s, out);
      }
      if (full < steps) {                                  // wave-uniform
        Row cur;
        load_row<kStaged>(c, img, n, n < c.N, cur);
        step_row16<true, kStaged, kMode>(cur, img, n < c.N, q0, q1, s, out);
      }
    }
  }

  // kImage: the caller guarantees the LDS image (the workgroup form of the sampling kernel)
  __device__ static __forceinline__ double logp_grad_staged(const Consts& c, const Lane& ln, int l,
                                                            const double (&q)[DPL], double (&g)[DPL]) {
    return eval_row16<true>(c, ln, l, q, g);
  }
  template <bool kImage = false>
  __device__ static __forceinline__ double eval_row16(const Consts& c, const Lane& ln, int l,
                                                      const double (&q)[DPL], double (&g)[DPL]) {
    double s[D + 1];   // s[0..D-1] gradient partials, s[D] likelihood partial
#pragma unroll
    for (int j = 0; j <= D; j++) s[j] = 0.0;
    const double q0 = q[0], q1 = q[DPL > 1 ? 1 : 0];
    if (kImage || ln.xs != nullptr) {   // wave-uniform
      if constexpr (EXMC_LG_PASS_GUARD == 0) {
        steps_row16<true, EXMC_LG_ROW_AHEAD != 0>(c, (const lds_v2d*)ln.xs, l, q0, q1, s);
      } else {
        // the short forms for the whole pass, the domain checked lane by lane on the way and decided ONCE:
        // a pass that saw a predictor outside [-200, 200] (or a NaN) is evaluated again in the general forms
        unsigned long long out = 0;
        steps_row16<true, EXMC_LG_ROW_AHEAD != 0, 1>(c, (const lds_v2d*)ln.xs, l, q0, q1, s, &out);
        if (__builtin_expect(out != 0, 0)) {
#pragma unroll
          for (int j = 0; j <= D; j++) s[j] = 0.0;
          steps_row16<true, false, 2>(c, (const lds_v2d*)ln.xs, l, q0, q1, s);   // (rare: no row buffers ahead)
        }
      }
    } else {
      steps_row16<false, false>(c, nullptr, l, q0, q1, s);
    }
    // the 22 sums as a reduce-scatter over the row (exmc_device.hpp row16_reduce_scatter: the butterfly's
    // own additions): lane l ends with the gradient totals of ITS dimensions l and 16 + l -- no select
    // chain over 21 totals per slot -- and lane 5 with the likelihood total (quantity 21 = 16 + 5),
    // which the final sum wants as lane 0's seed: one row broadcast
    double tot[DPL];
    row16_reduce_scatter<D + 1>(s, tot);
    const double lik = row_bcast_f64<(D % 16)>(tot[D / 16]);
    double T[DPL];
    bool valid[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      valid[k] = i < D;
      const double z = (q[k] - 0.0) / 10.0;
      T[k] = -0.5 * (z * z + c.c10);
      g[k] = valid[k] ? ((-(z / 10.0)) + tot[k]) : 0.0;
    }
    return group_sum_slots<G, DPL>(T, valid, l, lik);
  }

  // (the layouts below decide at run time whether an image is there: `staged` is wave-uniform)
  __device__ static __forceinline__ double log_tab_any(bool staged, int xoff, double x) {
    double m;
    int e;
    const int i = exmc_logtab_split(x, &m, &e);
    double c, lv;
    if (staged) {
      const double* t = exmc_dyn_lds + xoff - kStageRowOffset + 2 * i;   // in front of the rows
      c = t[0];
      lv = t[1];
    } else {
      c = exmc_logtab_dev[2 * i];
      lv = exmc_logtab_dev[2 * i + 1];
    }
    return exmc_logtab_finish(m, e, c, lv);
  }

  // The three per-observation specials in their short forms while every linear predictor of the
  // wavefront lies in [-200, 200]: exp without its special-case guards (exmc_exp_pm200), the
  // quotient 1 / (1 + e) through the refined reciprocal (1 + e in [1, e^200 + 1): inside the
  // division window), log of a clipped probability through exmc_log_unit (its domain (0, 1) holds
  // always: the clip bounds are 1e-7 and 1 - 1e-7). Same bits on that domain; a wavefront that
  // sees a predictor outside it (or a NaN) re-evaluates the pass with the general forms.
  template <class DV>
  __device__ static __forceinline__ double eval(const Consts& c, const Lane& ln, int l,
                                                const double (&q)[DPL], double (&g)[DPL], DV& dv) {
    constexpr bool kFast = std::is_same_v<DV, Div<true>>;
    double qf[D];
    bcast_all(q, qf, std::make_integer_sequence<int, D>{});
    double s[D + 1];   // s[0..D-1] gradient partials, s[D] likelihood partial
#pragma unroll
    for (int j = 0; j <= D; j++) s[j] = 0.0;
    const bool staged = ln.xs != nullptr;                     // wave-uniform
    const int xoff = staged ? (int)(ln.xs - exmc_dyn_lds) : 0;
    for (int n = l; n < c.N; n += G) {
      double xr[K], yn;
      if (staged) {
        const double* x = exmc_dyn_lds + xoff + n * kRowStride;
#pragma unroll
        for (int j = 0; j < K; j++) xr[j] = x[j];
        yn = x[K];
      } else {
        const double* x = c.X + (size_t)n * K;
#pragma unroll
        for (int j = 0; j < K; j++) xr[j] = x[j];
        yn = c.y[n];
      }
      double eta = __builtin_fma(1.0, qf[0], 0.0);
#pragma unroll
      for (int j = 0; j < K; j++) eta = __builtin_fma(xr[j], qf[1 + j], eta);
      double p, ll;
      if constexpr (kFast) {
        dv.ok = dv.ok && (fabs(eta) <= 200.0);   // false for a NaN
        // (the scalar-register cores of the 16-lane layout were measured here too: 54-55 against 53 ms for
        // the 1000-iteration warmup -- this kernel already spills scalar registers, 163 -> 328)
        p = dv(1.0, 1.0 + exmc_exp_pm200(-eta));
        const double pc = fmin(fmax(p, c.lo), c.hi);
        ll = log_tab_any(staged, xoff, (yn == 1.0) ? pc : (1.0 - pc));
      } else {
        p = 1.0 / (1.0 + exmc_exp(-eta));
        const double pc = fmin(fmax(p, c.lo), c.hi);
        ll = log_tab_any(staged, xoff, (yn == 1.0) ? pc : (1.0 - pc));
      }
      const double r = (p > c.lo && p < c.hi) ? (yn - p) : 0.0;
      s[D] = s[D] + ll;
      s[0] = __builtin_fma(1.0, r, s[0]);
#pragma unroll
      for (int j = 0; j < K; j++) s[1 + j] = __builtin_fma(xr[j], r, s[1 + j]);
    }
    if constexpr (G == 64) {
      // Round 5: the 22 sums of the one-chain warmup layout as a reduce-scatter -- over each 16-lane row
      // first (row16_reduce_scatter: a lane ends with the row partials of quantities l % 16 and 16 + l % 16),
      // then those two values through the two cross-row stages. The butterfly's own additions on the lanes
      // that need them (same bits: the pairs, quads, eights ... of group_allsum_n<64, 22>), about 180
      // instructions where the full butterfly took 400 and the per-lane pick of a total another 42 -- and
      // lanes 0 .. 20 already hold the gradient entry of their own dimension.
      static_assert(DPL == 1 && D + 1 <= 32, "two totals per lane");
      double tot[2];
      row16_reduce_scatter<D + 1>(s, tot);
      tot[0] = sum_xor32(sum_xor16(tot[0]));
      tot[1] = sum_xor32(sum_xor16(tot[1]));
      const double gi = (l < 16) ? tot[0] : tot[1];
      const double lik = readlane_f64(tot[1], D - 16);
      const bool valid[1] = {l < D};
      const double z = (q[0] - 0.0) / 10.0;
      const double T[1] = {-0.5 * (z * z + c.c10)};
      g[0] = valid[0] ? ((-(z / 10.0)) + gi) : 0.0;
      return group_sum_slots<G, DPL>(T, valid, l, lik);
    }
    group_allsum_n<G, D + 1>(s);
    double T[DPL];
    bool valid[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      valid[k] = i < D;
      double gi = 0.0;
#pragma unroll
      for (int j = 0; j < D; j++) gi = (i == j) ? s[j] : gi;
      const double z = (q[k] - 0.0) / 10.0;
      T[k] = -0.5 * (z * z + c.c10);
      g[k] = valid[k] ? ((-(z / 10.0)) + gi) : 0.0;
    }
    return group_sum_slots<G, DPL>(T, valid, l, s[D]);
  }
};

// ------------------------------------------------------------------------------------------
// logistic regression on the matrix cores: G = 4 lanes per chain => 16 chains per wavefront, the
// N dimension of v_mfma_f64_16x16x4_f64. Per 16-observation tile:
//     eta[16 obs][16 chains]   = Xa[16 obs][24] @ beta[24][16 chains]      6 MFMAs (K = 4 each)
//     p, log-lik, r = y - p    on the C/D fragment: 4 (obs, chain) pairs per lane, all 64 lanes
//     grad[32 feats][16 chains] += Xa^T[32][16 obs] @ r[16 obs][16 chains]  8 MFMAs
// The C/D register r of the eta tile (rows 4r..4r+3 across the four 16-lane rows) is exactly the
// B fragment of the K-step that consumes observations 4r..4r+3, so r never leaves registers.
// The f64 MFMA is an fma chain over k = 0..3 from C (tools/probe/mfma_f64_probe.hip: 256/256
// bitwise), so the CPU checker reproduces it with fma loops: eta over features 0..23 in order,
// grad over observations 0..Npad-1 in order, log-lik per row group then xor-butterfly.
// Needs every lane of the wavefront active (kCoop) and 912 doubles of LDS to move beta and the
// gradient between the chain-group layout and the MFMA fragment layout.
// ------------------------------------------------------------------------------------------
typedef double exmc_v4d __attribute__((ext_vector_type(4)));

template <>
struct Logistic<4> : ModelDefaults {
  static constexpr bool kPipeWarmup = false;
  static constexpr int G = 4;
  static constexpr int K = 20;
  static constexpr int D = K + 1;
  static constexpr int DPL = 6;
  static constexpr bool kCoop = true;
  static constexpr int kExtraLdsDoubles = 24 * 16 + 32 * 16 + 16;
  // LDS image of the design matrix for single-workgroup kernels: Xa feature-major with row
  // stride 514 doubles (the gradient fragment walks 16 features at a fixed observation: stride
  // 514 = 2 mod 32 keeps those ds_read_b64 conflict-free; the eta fragment sees 2-way), then y.
  static constexpr int kStageStride = 514;
  static constexpr int kStageDoubles = 24 * kStageStride + 512;
  using Consts = LogisticConsts;
  struct Lane {
    double* sh;         // wavefront-shared LDS scratch: beta [24][16], grad [32][16], lik [16]
    const double* xs;   // LDS image of Xa / y, or null (operands then stream from L2)
  };
  __device__ static __forceinline__ void load(const Consts&, int, Lane& ln) { ln.xs = nullptr; }
  // cooperative (whole workgroup): fill the LDS image; the caller synchronises afterwards
  __device__ static __forceinline__ void stage(const Consts& c, double* dst) {
    for (int i = threadIdx.x; i < kStageDoubles; i += blockDim.x) {
      double v = 0.0;
      if (i < 24 * kStageStride) {
        const int f = i / kStageStride, n = i % kStageStride;
        if (n < c.Npad) v = c.XaT[(size_t)f * c.Npad + n];
      } else {
        const int n = i - 24 * kStageStride;
        if (n < c.Npad) v = c.ypad[n];
      }
      dst[i] = v;
    }
  }

  __device__ static __forceinline__ double logp_grad(const Consts& c, const Lane& ln, int l,
                                                     const double (&q)[DPL], double (&g)[DPL]) {
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 2;        // chain of this lane within the wavefront
    const int col = lane & 15;        // MFMA column (= chain) this lane's fragment belongs to
    const int rowg = lane >> 4;       // MFMA row group
    double* bsh = ln.sh;
    double* gsh = ln.sh + 24 * 16;
    double* lsh = ln.sh + 24 * 16 + 32 * 16;
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;        // 0..23: rows 21..23 are the zero padding of K
      bsh[i * 16 + grp] = (i < D) ? q[k] : 0.0;
    }
    double bfrag[6];
#pragma unroll
    for (int ks = 0; ks < 6; ks++) bfrag[ks] = bsh[(4 * ks + rowg) * 16 + col];
    exmc_v4d g0 = {0.0, 0.0, 0.0, 0.0}, g1 = {0.0, 0.0, 0.0, 0.0};
    double likp = 0.0;
    const int Npad = c.Npad;
    // software pipeline: the A fragments of the next tile's eta product and of this tile's
    // gradient product are requested before the exp/log section, whose branches would otherwise
    // pin the loads right in front of the MFMAs that consume them (one wave per SIMD: nothing
    // else hides an L2 round trip)
    const double* xs = ln.xs;   // wave-uniform: LDS image or null
    auto ld_eta = [&](int ks, int n) -> double {
      return xs ? xs[(4 * ks + rowg) * kStageStride + n + col]
                : c.XaT[(size_t)(4 * ks + rowg) * Npad + n + col];
    };
    auto ld_g = [&](int ft, int n) -> double {
      if (xs) {
        const int f = 16 * ft + col;
        return (f < 24) ? xs[f * kStageStride + n] : 0.0;
      }
      return c.Xa32[(size_t)n * 32 + 16 * ft + col];
    };
    auto ld_y = [&](int n) -> double { return xs ? xs[24 * kStageStride + n] : c.ypad[n]; };
    double a_eta[6];
#pragma unroll
    for (int ks = 0; ks < 6; ks++) a_eta[ks] = ld_eta(ks, 0);
    for (int n0 = 0; n0 < Npad; n0 += 16) {
      double a_g0[4], a_g1[4], yv[4], a_next[6];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        a_g0[r] = ld_g(0, n0 + 4 * r + rowg);
        a_g1[r] = ld_g(1, n0 + 4 * r + rowg);
        yv[r] = ld_y(n0 + rowg + 4 * r);
      }
      const int nn = (n0 + 16 < Npad) ? (n0 + 16) : n0;
#pragma unroll
      for (int ks = 0; ks < 6; ks++) a_next[ks] = ld_eta(ks, nn);
      exmc_v4d eta = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 6; ks++)
        eta = __builtin_amdgcn_mfma_f64_16x16x4f64(a_eta[ks], bfrag[ks], eta, 0, 0, 0);
      exmc_v4d rr;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = n0 + rowg + 4 * r;
        const double yn = yv[r];
        const double p = 1.0 / (1.0 + exmc_exp(-eta[r]));
        const double pc = fmin(fmax(p, c.lo), c.hi);
        const double ll = exmc_log_tab_global((yn == 1.0) ? pc : (1.0 - pc));   // (the contract of logp_logistic)
        const double rv = (p > c.lo && p < c.hi) ? (yn - p) : 0.0;
        const bool live = n < c.N;
        likp = live ? (likp + ll) : likp;
        rr[r] = live ? rv : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_g0[r], rr[r], g0, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_g1[r], rr[r], g1, 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < 6; ks++) a_eta[ks] = a_next[ks];
    }
    // log-likelihood: the four row groups of a column hold partial sums of the same chain
    likp = likp + __shfl_xor(likp, 16, 64);
    likp = likp + __shfl_xor(likp, 32, 64);
    if (rowg == 0) lsh[col] = likp;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      gsh[(rowg + 4 * r) * 16 + col] = g0[r];
      gsh[(16 + rowg + 4 * r) * 16 + col] = g1[r];
    }
    const double lik = lsh[grp];
    double T[DPL];
    bool valid[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      valid[k] = i < D;
      const double gi = gsh[i * 16 + grp];
      const double z = (q[k] - 0.0) / 10.0;
      T[k] = -0.5 * (z * z + c.c10);
      g[k] = valid[k] ? ((-(z / 10.0)) + gi) : 0.0;
    }
    return group_sum_slots<G, DPL>(T, valid, l, lik);
  }
};

// ------------------------------------------------------------------------------------------
// hierarchical radon, J = 85 counties (notebooks/09_radon_bhm.livemd "The Radon Model").
// dims: 0..J-1 alpha_raw_j, J mu_alpha, J+1 gamma_u, J+2 log sigma_alpha, J+3 log sigma_y,
// J+4 beta; dimension i in slot i / G of lane i % G.
// G = 64 (the layout that ships): the observations are spread over the lanes whatever their
// county -- observation i on lane i % 64 in slot i / 64, 15 balanced slots for 919 observations
// (round 2 walked each county on its owner lane in chunks: 35 lock-step iterations, 43 % of them
// carrying an observation; round 3's generated radon, which derives this spread from the node
// list, ran 19 % faster than that kernel). Per leapfrog a lane fetches the y, floor and county of
// its slots in one batch of buffer loads from zero-padded copies (held in registers across the
// kernel they cost more in accumulator-register moves than the loads do; as flat loads from clamped
// addresses they cost 51 parked address pairs), the owner lanes publish their counties' intercepts
// alpha_j in an LDS strip, every observation reads its county's, and leaves a_i = z_i / sigma_y in
// a second strip from which the owner lane of a county adds its observations in index order.
// Sums: the likelihood / floor / z^2 totals = each lane's terms in slot order, then the lanes; a
// county's sum = its observations in index order from 0.0. The CPU checker restates the same rule.
// Other G (not dispatched): every county is walked by its owner lane.
// ------------------------------------------------------------------------------------------
struct RadonConsts {
  const double* u;      // dev [J]
  const double* cs;     // dev [J+1] county start offsets (as doubles)
  const double* fl;     // dev [N]
  const double* y;      // dev [N]
  const double* pobs;   // dev [3][kObsCap + 64]: y | floor | 8 * county (an integer in the low word), each zero-padded:
                        // the 64-lane layout's copies, slot s of lane l at entry 64 s + l whatever N
  double log2pi32, tiny32;
  double c_mu10;        // log2pi32 + 2*log(10)
  double c_n5;          // log2pi32 + 2*log(5)
  double c1;            // log2pi32 + 2*log(1)
  double c_hc;          // f32(log(2/pi)) - log(2.5)
};

template <int G>
struct Radon : ModelDefaults {
  static constexpr int J = 85;
  static constexpr int D = J + 5;
  static constexpr int DPL = (D + G - 1) / G;
  using Consts = RadonConsts;
  static constexpr bool kSpread = (G == 64);           // observations over the lanes (see above)
  static constexpr int kObsCap = 1024;                 // observations the spread layout holds
  static constexpr int kSlots = kSpread ? kObsCap / 64 : 1;
  static constexpr int kObsPad = kObsCap + 64;         // entries of each padded copy (Consts::pobs)
  static constexpr bool kRngOrbit = kSpread;
  // the tree's own exp / log (nuts_run) as the one-instruction-per-step asm cores: the coefficients sit in
  // vector registers either way, the compiler's spelling spends a v_mov per Horner step on top (68.8 ->
  // 68.4 ms, adaptation 75.4 -> 74.8 ms)
  static constexpr bool kVregMath = true;
  static constexpr int kAlphaOff = kObsCap;            // strip: [a_i (kObsCap)] [alpha_j (DPL * 64 cells, 0.0 from J on)]
  static constexpr int kZeroRun = kObsCap + 88;        // eight cells of 0.0 in a row (past the J = 85 intercepts)
  static constexpr int kExtraLdsDoubles = kSpread ? kObsCap + 128 : 0;
  struct Lane {
    double u[DPL];
    int i0[DPL], i1[DPL];   // own counties: their observations [i0, i1)
    int nobs;
    int maxc[DPL];          // kSpread: the largest county of each dimension slot (the same on every lane)
    Recip ten, five, c25;   // prior scales Normal(0, 10), Normal(0, 5), HalfCauchy(2.5)
    double* sh;             // kSpread: the wavefront's strips (attach_scratch / lane_setup)
    double kc[6];           // the model's scalar constants as (uniform) vector registers, see SV::Lane::k
  };
  enum { kTiny = 0, kLog2Pi, kHc, kC1, kMu10, kN5 };
  __device__ static __forceinline__ void load(const Consts& c, int l, Lane& ln) {
    ln.sh = nullptr;
    ln.nobs = (int)c.cs[J];
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int j = l + k * G;
      const bool cty = j < J;
      ln.u[k] = cty ? c.u[j] : 0.0;
      ln.i0[k] = cty ? (int)c.cs[j] : 0;
      ln.i1[k] = cty ? (int)c.cs[j + 1] : 0;
      ln.maxc[k] = 0;
    }
    if constexpr (kSpread) {
      for (int j = 0; j < J; j++) {
        const int nj = (int)c.cs[j + 1] - (int)c.cs[j];
#pragma unroll
        for (int k = 0; k < DPL; k++)
          if (j / G == k) ln.maxc[k] = nj > ln.maxc[k] ? nj : ln.maxc[k];
      }
    }
    ln.ten = make_recip_literal(10.0);
    ln.five = make_recip_literal(5.0);
    ln.c25 = make_recip_literal(2.5);
    const double src[6] = {c.tiny32, c.log2pi32, c.c_hc, c.c1, c.c_mu10, c.c_n5};
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double v = src[i];
      asm volatile("" : "+v"(v));
      ln.kc[i] = v;
    }
  }
  __device__ static __forceinline__ double logp_grad(const Consts& c, const Lane& ln, int l,
                                                     const double (&q)[DPL], double (&g)[DPL]) {
    return with_fast_div([&](auto& dv) -> double { return eval(c, ln, l, q, g, dv); });
  }

  // Quotients: sigma_y is watched in [2^-100, 2^100) and every residual y - mean in 2^+-250, so
  // z = resid / sigma_y stays within 2^+-350 and is a legal numerator again; mu, gamma and beta
  // are watched before their two quotients by the prior scale. The HalfCauchy arguments
  // x = exp(clamp200(.)) lie in [e^-200, e^200]: x / 2.5, 2z / 2.5 and the quotient by 1 + z^2
  // in [1, e^400] have operands within 2^+-580 of each other and quotients >= e^-201 (in range).
  template <class DV>
  __device__ static __forceinline__ double eval(const Consts& c, const Lane& ln, int l,
                                                const double (&q)[DPL], double (&g)[DPL], DV& dv) {
    // this lane's observations (64 lanes: y, floor and county of all its slots) are requested FIRST: the
    // addresses depend on nothing computed here, and a wave that has its SIMD to itself sits out every
    // L2 round trip it starts late -- this one now runs beside the five transcendentals below
    // (round 5; same values, same bits)
    // The slots every lane fills (nobs / 64 of them) and the one partly filled slot behind them are
    // fetched apart, the latter from its own (run-time) index: the unrolled walk below then tests a
    // full slot with ONE scalar compare, where a slot that could be either cost ten scalar
    // instructions of lane-mask bookkeeping -- and a wave alone on its SIMD pays an issue slot for
    // every instruction, whatever unit executes it.
    [[maybe_unused]] double oy[kSlots], ofl[kSlots], oc[kSlots];
    [[maybe_unused]] double oy_t = 0.0, ofl_t = 0.0, oc_t = 0.0;
    [[maybe_unused]] int nfull = 0;
    if constexpr (kSpread) {
      // (the padded copies: slot s of lane l is entry 64 s + l whatever the observation count. Buffer
      // loads -- one descriptor, the lane's byte offset in ONE register, slot and array as an immediate
      // plus a scalar offset -- where flat addresses were 51 register pairs, parked in accumulator
      // registers and read back at every evaluation)
      nfull = __builtin_amdgcn_readfirstlane(ln.nobs >> 6);
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(c.pobs), 0, 3 * kObsPad * 8, 0x00020000);
      const int l8 = l * 8;
      auto fetch = [&](int arr, int voff, int soff) -> double {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, arr * kObsPad * 8 + soff, 0));
      };
#pragma unroll
      for (int sl = 0; sl < kSlots; sl++) {
        const int vo = l8 + (sl & 7) * 512, so = (sl >> 3) * 4096;
        oy[sl] = fetch(0, vo, so);
        ofl[sl] = fetch(1, vo, so);
        oc[sl] = fetch(2, vo, so);
      }
      oy_t = fetch(0, l8, nfull * 512);           // entry 64 nfull + l <= kObsCap + 63
      ofl_t = fetch(1, l8, nfull * 512);
      oc_t = fetch(2, l8, nfull * 512);
      // (the third copy holds 8 * county as an integer in the low word of each entry: the byte offset of the
      // county's intercept in the alpha strip, used as it comes)
      __builtin_amdgcn_sched_barrier(0);   // the scheduler would sink the loads back towards their uses
    }
    const double mu = group_bcast_c<G, J % G>(q[J / G]);
    const double gam = group_bcast_c<G, (J + 1) % G>(q[(J + 1) / G]);
    const double zsa_raw = group_bcast_c<G, (J + 2) % G>(q[(J + 2) / G]);
    const double zsy_raw = group_bcast_c<G, (J + 3) % G>(q[(J + 3) / G]);
    const double beta = group_bcast_c<G, (J + 4) % G>(q[(J + 4) / G]);
    const double zsa = clamp200(zsa_raw), zsy = clamp200(zsy_raw);
    // the chain-scalar transcendentals as lane-batched evaluations: 2 exps, then the 3 logs
    constexpr bool kFast = std::is_same_v<DV, Div<true>>;
    double ez[2] = {zsa, zsy};
    lane_batch<G, 2>(ez, l, [](double v) { return exmc_exp_pm200(v); });   // clamp200'ed: always in range
    const double sa = ez[0], sy = ez[1];
    const double ssy = fmax(sy, ln.kc[kTiny]);
    dv.template watch_exp_if<-100, 100>(true, ssy);
    dv.watch(mu);
    dv.watch(gam);
    dv.watch(beta);
    const Recip rsy = make_recip(ssy);
    // HalfCauchy(2.5) on sigma_alpha, sigma_y (half_cauchy.ex:19-27): z = x / 2.5,
    // logp = c_hc - log(1 + z^2), d/dx = -((2z / 2.5) / (1 + z^2))
    const double za = dv(sa, ln.c25), zy = dv(sy, ln.c25);
    const double za2 = za * za, zy2 = zy * zy;
    // inside the fast window ssy is a watched positive scale and 1 + z^2 lies in [1, e^400]: the main
    // path of the logarithm alone (same bits)
    double lx[3] = {ssy, 1.0 + za2, 1.0 + zy2};
    if constexpr (kFast) lane_batch<G, 3>(lx, l, [](double v) { return exmc_log_ge1(v); });
    else lane_batch<G, 3>(lx, l, [](double v) { return exmc_log(v); });
    const double cn = ln.kc[kLog2Pi] + 2.0 * lx[0];
    const double dsa = -dv(dv(2.0 * za, ln.c25), 1.0 + za2);
    const double dsy = -dv(dv(2.0 * zy, ln.c25), 1.0 + zy2);
    const double t_sa = (ln.kc[kHc] - lx[1]) + zsa;
    const double t_sy = (ln.kc[kHc] - lx[2]) + zsy;
    double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // lik, S, S*u, S*alpha_raw, F, Z2 partials
    double T[DPL];
    bool valid[DPL];
    // one observation against its county's intercept: -> a = z / sigma_y
    // (rounds 2-4 watched the residuals' magnitudes for the short division; since round 5 z and a are
    // PRODUCTS with the reciprocal of sigma_y, exact for any operand, so nothing is left to watch)
    // Round 5: the unit arithmetic of the generated radon (codegen_lanes.py), which measured 6 % faster
    // with it: fused multiply-adds, and the two quotients by sigma_y as products with ONE correctly
    // rounded reciprocal per leapfrog (1 / sigma_y by the IEEE division, so the checker reproduces it
    // as 1.0 / ssy) -- 12 operations per observation instead of 18. Inside the 1e-12 tolerance to the
    // reference's own arithmetic (tests/test_oracle_sampler.py); restated in the checker's logp_radon.
    const double rinv = dv(1.0, rsy);
    auto obs = [&](double alpha, double fi, double yi, double& lik, double& f, double& z2s) -> double {
      const double mean = __builtin_fma(beta, fi, alpha);
      const double resid = yi - mean;
      const double z = resid * rinv;
      const double a = z * rinv;
      lik = __builtin_fma(-0.5, __builtin_fma(z, z, cn), lik);
      f = __builtin_fma(a, fi, f);
      z2s = z2s + __builtin_fma(z, z, -1.0);
      return a;
    };
    double alpha_own[DPL], sj_own[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      valid[k] = (l + k * G) < D;
      alpha_own[k] = (mu + gam * ln.u[k]) + sa * q[k];
    }
    if constexpr (kSpread) {
      double* const cell = ln.sh;               // a_i of observation i
      double* const al = ln.sh + kAlphaOff;     // alpha_j of county j
      // every lane writes every slot, 0.0 from county J on: no lane mask, and the cells from J on are the
      // zeros a lane past the end of its county adds (below)
#pragma unroll
      for (int k = 0; k < DPL; k++) al[l + k * G] = (l + k * G < J) ? alpha_own[k] : 0.0;
      wave_lds_fence();
      // (this lane's observations were requested at the top) the counties' intercepts
      auto alpha_at = [&](double packed) -> double {
        const int off = (int)__builtin_bit_cast(unsigned long long, packed);   // 8 * county
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(al) + off);
      };
      double av[kSlots];
#pragma unroll
      for (int sl = 0; sl < kSlots; sl++) av[sl] = alpha_at(oc[sl]);
      const double av_t = alpha_at(oc_t);
      // ONE wait for the seventeen reads (the walk needs the last of them anyway): left to the compiler every
      // observation carries its own s_waitcnt, and a wave alone on its SIMD pays a slot for each
      exmc_wait_lgkm0();
      double lik = 0.0, f = 0.0, z2s = 0.0;
#pragma unroll
      for (int sl = 0; sl < kSlots; sl++)
        if (__builtin_expect(sl < nfull, 1))                                         // wave-uniform test
          cell[sl * 64 + l] = obs(av[sl], ofl[sl], oy[sl], lik, f, z2s);
      {
        const int i = nfull * 64 + l;         // the partly filled slot: the last terms of every lane's sums
        if (i < ln.nobs) cell[i] = obs(av_t, ofl_t, oy_t, lik, f, z2s);
      }
      s[0] = lik;
      s[4] = f;
      s[5] = z2s;
      wave_lds_fence();
      // a county's sum: its cells in index order, several reads in flight at a time; the loop runs to
      // the slot's largest county on every lane (a wave-uniform bound: scalar loop control), a lane
      // past the end of its own county keeps its sum
#pragma unroll
      for (int k = 0; k < DPL; k++) {
        double sj = 0.0;
        const int i0 = ln.i0[k], i1 = ln.i1[k];
        const int nb = __builtin_amdgcn_readfirstlane(ln.maxc[k]);
        // (eight reads in flight: a wave that has its SIMD to itself waits out every LDS round trip,
        // and the additions behind them are one dependent chain in index order whatever the batch)
        constexpr int kAhead = 8;
        // (a lane past the end of its own county reads the strip's zeros: sj + 0.0 is sj -- the sum
        // starts at +0.0 and a sum of doubles is never -0.0 unless every term is -- so the addition
        // needs no mask: two selects per cell less, same bits)
        // Round 5: whole batches of eight cells first -- ONE select per lane and batch between this lane's run
        // and the run of eight zeros, the cell's place in the batch as the read's immediate offset -- then the
        // (up to seven) cells a county has beyond its last whole batch, selected cell by cell. A lane adds its
        // own cells in index order with zeros in between, which leaves every sum as it was (was: index,
        // compare, select and shift for every cell of every batch up to the largest county).
        const double* const pz = ln.sh + kZeroRun;
        const int n = i1 - i0;
        const int nwhole = n >> 3;                       // this lane's whole batches
        const int rest = n & 7;
        const double* pa = cell + i0;
        const double* const prest = pa + (n & ~7);
        const int nbw = nb >> 3;                         // wave-uniform: whole batches of the largest county
        for (int b = 0; b < nbw; b++) {
          const double* const base = (b < nwhole) ? pa : pz;
          double v[kAhead];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < kAhead; j++) v[j] = base[j];
          __builtin_amdgcn_sched_barrier(0);   // all the reads in flight before the first addition waits
          exmc_wait_lgkm0();                   // (one wait per batch, not one per pair of cells)
#pragma unroll
          for (int j = 0; j < kAhead; j++) sj = sj + v[j];
          pa += kAhead;
        }
        {
          const double* ad[kAhead - 1];
          double v[kAhead - 1];
#pragma unroll
          for (int j = 0; j < kAhead - 1; j++) ad[j] = ((rest > j) ? prest : pz) + j;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < kAhead - 1; j++) v[j] = *ad[j];
          __builtin_amdgcn_sched_barrier(0);
          exmc_wait_lgkm0();
#pragma unroll
          for (int j = 0; j < kAhead - 1; j++) sj = sj + v[j];
        }
        sj_own[k] = sj;
      }
    } else {
#pragma unroll
      for (int k = 0; k < DPL; k++) {
        double lik = 0.0, f = 0.0, z2s = 0.0, sj = 0.0;
        for (int i = ln.i0[k]; i < ln.i1[k]; i++) sj = sj + obs(alpha_own[k], c.fl[i], c.y[i], lik, f, z2s);
        sj_own[k] = sj;
        if (valid[k]) {
          s[0] = s[0] + lik;
          s[4] = s[4] + f;
          s[5] = s[5] + z2s;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const double ar = q[k];
      const double sj = sj_own[k];
      if (valid[k]) {
        s[1] = s[1] + sj;
        s[2] = s[2] + sj * ln.u[k];
        s[3] = s[3] + sj * ar;
      }
      T[k] = -0.5 * (ar * ar + ln.kc[kC1]);
      g[k] = (-ar) + sj * sa;
    }
    // 64 lanes: the six totals as a reduce-scatter (exmc_device.hpp rs64_reduce6: the butterfly's own
    // additions, half its instructions) and six scalar broadcasts -- the same bits
    if constexpr (G == 64) rs64_allsum6(s);
    else group_allsum_n<G, 6>(s);
    const bool in_a = (zsa_raw > -200.0) && (zsa_raw < 200.0);
    const bool in_y = (zsy_raw > -200.0) && (zsy_raw < 200.0);
    if constexpr (kSpread) {
      // The five scalar parameters sit on five lanes of the last slot. Written as five lane-specific
      // branches (below, the other layouts) the wavefront walks all five bodies one after another with one
      // lane live in each; here the three Normal priors are ONE evaluation with per-lane operands and the
      // two HalfCauchy derivatives another -- the same operations on the same values, lane by lane.
      constexpr int ks = J / G;
      const int j = l + ks * G;
      const bool is_mu = j == J, is_g = j == J + 1, is_a = j == J + 2, is_y = j == J + 3, is_b = j == J + 4;
      const double v = is_mu ? mu : (is_g ? gam : beta);
      Recip rc;
      rc.b = is_mu ? ln.ten.b : ln.five.b;
      rc.r = is_mu ? ln.ten.r : ln.five.r;
      const double cst = is_mu ? ln.kc[kMu10] : ln.kc[kN5];
      const double sx = is_mu ? s[1] : (is_g ? s[2] : s[4]);
      const double zn = dv(v - 0.0, rc);
      const double Tn = -0.5 * (zn * zn + cst);
      const double gn = (-dv(zn, rc)) + sx;
      const double zc = is_a ? za : zy, zc2 = is_a ? za2 : zy2;
      const double dc = -dv(dv(2.0 * zc, ln.c25), 1.0 + zc2);
      const double ga = in_a ? ((dc + s[3]) * sa + 1.0) : 0.0;
      const double gy = in_y ? ((dc * sy + s[5]) + 1.0) : 0.0;
      const bool is_n = is_mu || is_g || is_b;
      T[ks] = is_n ? Tn : (is_a ? t_sa : (is_y ? t_sy : T[ks]));
      g[ks] = is_n ? gn : (is_a ? ga : (is_y ? gy : g[ks]));
#pragma unroll
      for (int k = 0; k < DPL; k++)
        if (l + k * G >= D) g[k] = 0.0;
      return group_sum_slots<G, DPL>(T, valid, l, s[0]);
    }
    const double zmu = dv(mu - 0.0, ln.ten), zg = dv(gam - 0.0, ln.five), zb = dv(beta - 0.0, ln.five);
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int j = l + k * G;
      if (j == J) { T[k] = -0.5 * (zmu * zmu + ln.kc[kMu10]); g[k] = (-dv(zmu, ln.ten)) + s[1]; }
      if (j == J + 1) { T[k] = -0.5 * (zg * zg + ln.kc[kN5]); g[k] = (-dv(zg, ln.five)) + s[2]; }
      if (j == J + 2) { T[k] = t_sa; g[k] = in_a ? ((dsa + s[3]) * sa + 1.0) : 0.0; }
      if (j == J + 3) { T[k] = t_sy; g[k] = in_y ? ((dsy * sy + s[5]) + 1.0) : 0.0; }
      if (j == J + 4) { T[k] = -0.5 * (zb * zb + ln.kc[kN5]); g[k] = (-dv(zb, ln.five)) + s[4]; }
      if (j >= D) g[k] = 0.0;
    }
    return group_sum_slots<G, DPL>(T, valid, l, s[0]);
  }
};

// The same model with a dense mass matrix in the row layout: a compile-time switch of the
// mass-dependent operations of nuts_run (exmc_nuts.hpp mass_*), so the diagonal kernels carry no
// trace of the mode. Valid for 16-lane groups that hold one dimension per lane with D <= 12.
template <class M>
struct RowDenseModel : M {
  static constexpr bool kRowDense = true;
  static constexpr bool kPipeWarmup = false;   // the dense warmup is the one-wave form
  static_assert(M::DPL == 1 && M::D <= 12, "row layout: one dimension per lane, D <= 12");
};

// The lane layouts (sv, radon, logistic: d = 102, 90, 21 over 64 or 16 lanes) with a dense mass
// matrix: the same compile-time switch, the contractions of exmc_device.hpp lane_dense_* (operands
// permuted on the host so that flat-order chains read coalesced rows). One wave per SIMD: the
// sweeps over the d x d operands are what the kernel does most, not the model.
template <class M, int G>
struct LaneDenseModel : M {
  static constexpr bool kLaneDense = true;
  static constexpr bool kPipeWarmup = false;   // the dense warmup is the one-wave form
  static constexpr int kNutsWavesPerSimd = 1;
  static constexpr int kPipeNutsLevels = 0;
  // M^-1 goes to LDS when that costs no resident wave (logistic: 3.5 KB); sv / radon (83 / 65 KB)
  // would leave one wave per CU, measured 2x slower than four waves per CU sweeping it from L2
  static constexpr bool kDenseImage = M::D * M::D * 8 <= 16 * 1024;
  static constexpr int kDenseLdsDoubles = (64 / G) * 3 * M::D + (kDenseImage ? M::D * M::D + 64 : 0);
  static_assert(G >= 16 && M::DPL * G >= M::D && !M::kCoop, "a lane layout that holds the whole chain");
};

}  // namespace exmc

// ------------------------------------------------------------------------------------------
// A model generated from Builder IR (exmc_amd/codegen.py; the reference's compiler.ex term walk
// + Nx.Defn value_and_grad, compiler.ex:131-141,200-269). The generated header is straight-line
// code over the whole position, so a chain is one lane and q / g stay in registers; c[] (the
// data-only subexpressions, folded on the host at model create) is read with scalar loads.
// ------------------------------------------------------------------------------------------
#ifdef EXMC_CUSTOM_HEADER
#define EXMC_GEN_HOST static inline
#define EXMC_GEN_FN static __host__ __device__ __forceinline__
// exp / log / log1p are real calls here, not inlined. A generated body holds tens of them; inlined
// at -O3, each brings its special-case branches, the folded constants already fill the scalar
// registers, and the resulting scalar-register spill code produced run-to-run different results
// on gfx950 (ROCm 7.2; a d = 9 model with 30 calls: step-size search off by 2x, lanes 1.. of a
// wave differing from lane 0, while -O1 and the called form agree with the CPU checker bit for
// bit -- tools/gen_debug.py). The call form also cuts the VGPR spills of the NUTS kernel 12x.
static __host__ __device__ __noinline__ double exmc_gen_exp_call(double x) { return exmc_exp(x); }
static __host__ __device__ __noinline__ double exmc_gen_log_call(double x) { return exmc_log(x); }
static __host__ __device__ __noinline__ double exmc_gen_log1p_call(double x) { return exmc_log1p(x); }
static __host__ __device__ __noinline__ double exmc_gen_erf_call(double x) { return exmc_erf(x); }
#define EXMC_GEN_ERF exmc_gen_erf_call    // two unrolled loops: a call in either layout
#define EXMC_GENV_ERF exmc_gen_erf_call
#ifdef EXMC_GEN_INLINE_MATH   // tools/probe/gen_inline_repro.py: the inlined form under investigation
#define EXMC_GEN_EXP exmc_exp
#define EXMC_GEN_LOG exmc_log
#define EXMC_GEN_LOG1P exmc_log1p
#else
#define EXMC_GEN_EXP exmc_gen_exp_call
#define EXMC_GEN_LOG exmc_gen_log_call
#define EXMC_GEN_LOG1P exmc_gen_log1p_call
#endif
// the 16-lane plate layout's lane function: inlined unless the build asks for calls (the fence of
// tests/test_gpu_codegen_inline.py covers the inlined form of the one-lane body as well)
#ifdef EXMC_GENV_CALLED_MATH
#define EXMC_GENV_EXP exmc_gen_exp_call
#define EXMC_GENV_LOG exmc_gen_log_call
#define EXMC_GENV_LOG1P exmc_gen_log1p_call
#else
#define EXMC_GENV_EXP exmc_exp
#define EXMC_GENV_LOG exmc_log
#define EXMC_GENV_LOG1P exmc_log1p
#endif
// the lane layout (exmc_amd/codegen_lanes.py): its exp / log are inlined unless the build asks for
// calls; its LDS strip is the chain group's part of the functor's scratch (EXMC_GEN_SH), the
// butterfly of its partial sums group_allsum_n, its fence the wave-level LDS fence
#ifdef EXMC_GENL_CALLED_MATH
#define EXMC_GENL_EXP exmc_gen_exp_call
#define EXMC_GENL_LOG exmc_gen_log_call
#define EXMC_GENL_LOG1P exmc_gen_log1p_call
#else   // the general functions with their coefficients pinned in vector registers (same bits)
#define EXMC_GENL_EXP exmc::Math<true>::exp
#define EXMC_GENL_LOG exmc::Math<true>::log
#define EXMC_GENL_LOG1P exmc_log1p
#endif
#define EXMC_GENL_ERF exmc_gen_erf_call
#define EXMC_GEN_CTX_DECL , int shoff
#define EXMC_GEN_SH(i) exmc::exmc_dyn_lds[shoff + (i)]
#define EXMC_GEN_XROW (EXMC_GEN_LANES == 64 && EXMC_GEN_WAVES_PER_SIMD == 2)
#define EXMC_GEN_ALLSUM(s) exmc::group_allsum_n<EXMC_GEN_LANES, EXMC_GEN_NS, EXMC_GEN_XROW>(s)
#define EXMC_GEN_ALLSUM_W(w) exmc::group_allsum_n<EXMC_GEN_LANES, EXMC_GEN_NW, EXMC_GEN_XROW>(w)
// n chain-scalar calls as one: argument i on lane i of the group, results broadcast back
#define EXMC_GEN_BATCH_LOG(n, b) exmc::lane_batch<EXMC_GEN_LANES, n>(b, l, [](double a_) { return EXMC_GENL_LOG(a_); })
#define EXMC_GEN_BATCH_EXP(n, b) exmc::lane_batch<EXMC_GEN_LANES, n>(b, l, [](double a_) { return EXMC_GENL_EXP(a_); })
#define EXMC_GEN_BATCH_LOG1P(n, b) exmc::lane_batch<EXMC_GEN_LANES, n>(b, l, [](double a_) { return EXMC_GENL_LOG1P(a_); })
#define EXMC_GEN_BATCH_RCP(n, b) exmc::lane_batch<EXMC_GEN_LANES, n>(b, l, [](double a_) { return 1.0 / a_; })
#define EXMC_GEN_FENCE() exmc::wave_lds_fence()
// a pair of per-unit columns (codegen_lanes.py: interleaved over the units, 16-byte aligned in every placement of the
// table): one global_load_dwordx4 / ds_read_b128 -- the alignment is what lets the compiler choose ds_read_b128 over
// ds_read2_b64, a quarter of its LDS cycles
struct alignas(16) exmc_gen_d2 { double x, y; };
namespace exmc {
// (the LDS placements say so in the pointer's type: the compiler then reads the pair with ds_read_b128; behind a generic
// pointer it chose ds_read2_b64, four times the LDS cycles)
__device__ __forceinline__ exmc_gen_d2 gen_lt2_lds(int k) {
  // (k is even and the image 16-byte aligned: said out loud, because the compiler sees exmc_dyn_lds, an array of doubles)
  const exmc_v2d v = *(const lds_v2d*)__builtin_assume_aligned(exmc_dyn_lds + k, 16);
  return exmc_gen_d2{v.x, v.y};
}
}  // namespace exmc
#define EXMC_GEN_LT2_GLOBAL(i) (*(const exmc_gen_d2*)&lt[i])
#define EXMC_GEN_LT2_LDS(i) exmc::gen_lt2_lds(ltoff + (i))
#define EXMC_GEN_FMA(a, b, c) __builtin_fma(a, b, c)
#include EXMC_CUSTOM_HEADER
#ifdef EXMC_GEN_LANES
// LDS of a sampling workgroup of the lane layout: tree-stack levels, the ziggurat tables, the
// chains' strips and (when it fits) an image of the model's tables. Two waves per SIMD are eight
// workgroups per CU -- 20 KB each of the 160 KB, or they are not resident together and the launch
// bound bought nothing (gen_sv: 31.5 KB with two stack levels and the table image = five per CU).
// A second stack level is worth more than the table image (a wave pair hides the L2 round trips of
// table reads; stack traffic is on the tree's critical path).
#define EXMC_GEN_LDS_BUDGET ((EXMC_GEN_WAVES_PER_SIMD == 2) ? 160 * 1024 / 8 : 160 * 1024 / 4)
#define EXMC_GEN_LDS_BYTES(levels, table)                                                      \
  ((levels) * (5 * EXMC_GEN_DPL + 3) * 64 * 8 + 768 * 8 + (64 / EXMC_GEN_LANES) * EXMC_GEN_LSH * 8 + \
   (table) * EXMC_GEN_NLT * 8)
#if EXMC_GEN_DPL == 1
#define EXMC_GEN_LDSL 6
#elif EXMC_GEN_LDS_BYTES(2, 0) <= EXMC_GEN_LDS_BUDGET
#define EXMC_GEN_LDSL 2
#else
#define EXMC_GEN_LDSL 1
#endif
#if EXMC_GEN_NLT <= 2048 && EXMC_GEN_LDS_BYTES(EXMC_GEN_LDSL, 1) <= EXMC_GEN_LDS_BUDGET
#define EXMC_GEN_TABLE_IN_LDS 1
#else
#define EXMC_GEN_TABLE_IN_LDS 0
#endif
// (Round 5 measured the workgroup form of the sampling kernel -- eight wavefronts around ONE LDS image of
// the tables, exmc_nuts.hpp nuts_kernel_wg -- for the generated 500 x 20 logistic regression, whose
// 86 KB of tables do not fit beside a one-wave workgroup: 267.9 against 269.5 ms, bit-identical. The
// generated pass waits on its LDS strips and fences, not on its table reads, so the form is not built
// for generated layouts: no ninth translation unit in every plug-in for it.)
namespace exmc {
constexpr bool kGenLdsTable = EXMC_GEN_TABLE_IN_LDS != 0;
}
#endif
#ifndef EXMC_GEN_FAST_WINDOW
#define EXMC_GEN_FAST_WINDOW 1
#endif
#ifndef EXMC_GEN_WG     // (headers generated before round 6, and the layouts that do not ask for the workgroup form)
#define EXMC_GEN_WG 0
#endif
#if EXMC_GEN_FAST_WINDOW
// the watched main paths of the fast window (the lane layouts and the plate layout below)
namespace exmc {
__device__ __forceinline__ double genf_exp(double x, bool& ok) {
  ok = ok && (fabs(x) <= 700.0);                       // false for a NaN
  return exmc_exp_pm200_v(x);                          // the main path: valid while the result is a normal number
}
__device__ __forceinline__ double genf_log(double x, bool& ok) {
  ok = ok && __builtin_amdgcn_class(x, 0x100);         // a positive normal number
  return exmc_log_normal_v(x);
}
__device__ __forceinline__ double genf_log1p(double x, bool& ok) {
  const double u = 1.0 + x;                            // exmc_log1p: log(u) + (x - (u - 1)) / u, x itself when u == 1
  ok = ok && __builtin_amdgcn_class(u, 0x100);
  const double r = exmc_log_normal_v(u) + (x - (u - 1.0)) / u;
  return (u == 1.0) ? x : r;
}
}  // namespace exmc
#if defined(EXMC_GEN_VEC) && !defined(EXMC_GENV_CALLED_MATH)
#define EXMC_GENV_FAST 1
// the plate layout's lane function once more, as the lane layouts' below: the same text with watched
// main-path exp / log / log1p (Custom<16>::logp_grad evaluates the exact one again when a lane's
// argument left the domain)
#define EXMC_GEN_VEC_SECTION
#undef EXMC_GENV_NAME
#undef EXMC_GENV_CTX_DECL
#undef EXMC_GENV_EXP
#undef EXMC_GENV_LOG
#undef EXMC_GENV_LOG1P
#define EXMC_GENV_NAME exmc_gen_lane_fast
#define EXMC_GENV_CTX_DECL , bool& exmc_gen_ok
#define EXMC_GENV_EXP(x) exmc::genf_exp((x), exmc_gen_ok)
#define EXMC_GENV_LOG(x) exmc::genf_log((x), exmc_gen_ok)
#define EXMC_GENV_LOG1P(x) exmc::genf_log1p((x), exmc_gen_ok)
#include EXMC_CUSTOM_HEADER
#undef EXMC_GEN_VEC_SECTION
#endif
#endif
#ifdef EXMC_GEN_LANES
// the lane function twice: its tables in global memory, and in an LDS image the NUTS and warmup
// workgroups stage once per kernel (a lone wave per SIMD would otherwise sit out an L2 round trip
// per family and leapfrog; ModelDefaults::kLdsDataDoubles). ltoff = the image's offset in doubles.
#define EXMC_GEN_LANES_SECTION
#define EXMC_GEN_G0 0       // a chain's units on its own lane group (sampling: 64 / G chains per wavefront)
#define EXMC_GEN_NG 1
#define EXMC_GEN_XGROUP(s)
#define EXMC_GEN_LANES_NAME exmc_gen_lanes_global
#define EXMC_GEN_LT(i) lt[i]
#define EXMC_GEN_LT2(i) EXMC_GEN_LT2_GLOBAL(i)
#define EXMC_GEN_IT(i) ((const int*)(lt + EXMC_GEN_IOFF))[i]
#include EXMC_CUSTOM_HEADER
#undef EXMC_GEN_LANES_NAME
#undef EXMC_GEN_LT
#undef EXMC_GEN_LT2
#undef EXMC_GEN_IT
#undef EXMC_GEN_CTX_DECL
#define EXMC_GEN_CTX_DECL , int shoff, int ltoff
#define EXMC_GEN_LANES_NAME exmc_gen_lanes_lds
#define EXMC_GEN_LT(i) exmc::exmc_dyn_lds[ltoff + (i)]
#define EXMC_GEN_LT2(i) EXMC_GEN_LT2_LDS(i)
#define EXMC_GEN_IT(i) ((const int*)(exmc::exmc_dyn_lds + ltoff + EXMC_GEN_IOFF))[i]
#include EXMC_CUSTOM_HEADER
#if EXMC_GEN_FAST_WINDOW
// Round 6 -- the fast window of the hand-written kinds, generically: the SAME generated text compiled once more
// with exp / log / log1p as the main paths of the same algorithms (no special-case branches, v_ldexp_f64
// scaling; the bits of the general functions on their domain, exmc_detmath.h) and every argument WATCHED:
// a lane whose argument leaves the domain (an exp argument beyond +-700, a log argument that is not a
// positive normal number, NaN) clears exmc_gen_ok, and a wavefront that saw one evaluates the exact form
// again (Custom::logp_grad below; the function only stores to its strips, so a second evaluation simply
// overwrites the first). A general exp / log is three to four compare-and-branch guards per call and a
// two-product scaling; a generated body holds one to a dozen calls per unit.
#undef EXMC_GENL_EXP
#undef EXMC_GENL_LOG
#undef EXMC_GENL_LOG1P
#undef EXMC_GEN_BATCH_LOG
#undef EXMC_GEN_BATCH_EXP
#undef EXMC_GEN_BATCH_LOG1P
#define EXMC_GENL_EXP(x) exmc::genf_exp((x), exmc_gen_ok)
#define EXMC_GENL_LOG(x) exmc::genf_log((x), exmc_gen_ok)
#define EXMC_GENL_LOG1P(x) exmc::genf_log1p((x), exmc_gen_ok)
#define EXMC_GEN_BATCH_LOG(n, b) exmc::lane_batch<EXMC_GEN_LANES, n>(b, l, [&](double a_) { return EXMC_GENL_LOG(a_); })
#define EXMC_GEN_BATCH_EXP(n, b) exmc::lane_batch<EXMC_GEN_LANES, n>(b, l, [&](double a_) { return EXMC_GENL_EXP(a_); })
#define EXMC_GEN_BATCH_LOG1P(n, b) exmc::lane_batch<EXMC_GEN_LANES, n>(b, l, [&](double a_) { return EXMC_GENL_LOG1P(a_); })
#undef EXMC_GEN_LANES_NAME
#undef EXMC_GEN_LT
#undef EXMC_GEN_LT2
#undef EXMC_GEN_IT
#undef EXMC_GEN_CTX_DECL
#define EXMC_GEN_CTX_DECL , int shoff, bool& exmc_gen_ok
#define EXMC_GEN_LANES_NAME exmc_gen_lanes_global_fast
#define EXMC_GEN_LT(i) lt[i]
#define EXMC_GEN_LT2(i) EXMC_GEN_LT2_GLOBAL(i)
#define EXMC_GEN_IT(i) ((const int*)(lt + EXMC_GEN_IOFF))[i]
#include EXMC_CUSTOM_HEADER
#undef EXMC_GEN_LANES_NAME
#undef EXMC_GEN_LT
#undef EXMC_GEN_LT2
#undef EXMC_GEN_IT
#undef EXMC_GEN_CTX_DECL
#define EXMC_GEN_CTX_DECL , int shoff, int ltoff, bool& exmc_gen_ok
#define EXMC_GEN_LANES_NAME exmc_gen_lanes_lds_fast
#define EXMC_GEN_LT(i) exmc::exmc_dyn_lds[ltoff + (i)]
#define EXMC_GEN_LT2(i) EXMC_GEN_LT2_LDS(i)
#define EXMC_GEN_IT(i) ((const int*)(exmc::exmc_dyn_lds + ltoff + EXMC_GEN_IOFF))[i]
#include EXMC_CUSTOM_HEADER
// (back to the exact forms for whatever is included below)
#undef EXMC_GENL_EXP
#undef EXMC_GENL_LOG
#undef EXMC_GENL_LOG1P
#undef EXMC_GEN_BATCH_LOG
#undef EXMC_GEN_BATCH_EXP
#undef EXMC_GEN_BATCH_LOG1P
#undef EXMC_GEN_CTX_DECL
#define EXMC_GEN_CTX_DECL , int shoff, int ltoff
#ifdef EXMC_GENL_CALLED_MATH
#define EXMC_GENL_EXP exmc_gen_exp_call
#define EXMC_GENL_LOG exmc_gen_log_call
#define EXMC_GENL_LOG1P exmc_gen_log1p_call
#else
#define EXMC_GENL_EXP exmc::Math<true>::exp
#define EXMC_GENL_LOG exmc::Math<true>::log
#define EXMC_GENL_LOG1P exmc_log1p
#endif
#define EXMC_GEN_BATCH_LOG(n, b) exmc::lane_batch<EXMC_GEN_LANES, n>(b, l, [](double a_) { return EXMC_GENL_LOG(a_); })
#define EXMC_GEN_BATCH_EXP(n, b) exmc::lane_batch<EXMC_GEN_LANES, n>(b, l, [](double a_) { return EXMC_GENL_EXP(a_); })
#define EXMC_GEN_BATCH_LOG1P(n, b) exmc::lane_batch<EXMC_GEN_LANES, n>(b, l, [](double a_) { return EXMC_GENL_LOG1P(a_); })
#endif
#if EXMC_GEN_LANES < 64
// a third time for the one-chain warmup of a layout with fewer than 64 lanes per chain: the units of
// a family over ALL 64 / G lane groups of the wavefront (group g takes the slots g, g + NG, ...), the
// groups' sums added in group order after each group's butterfly, one set of strips for the wave
// (CustomSplit below). Tables where the warmup kernel keeps them (kGenLdsTable).
#undef EXMC_GEN_LANES_NAME
#undef EXMC_GEN_G0
#undef EXMC_GEN_NG
#undef EXMC_GEN_XGROUP
#define EXMC_GEN_G0 ((int)(threadIdx.x & 63) / EXMC_GEN_LANES)
#define EXMC_GEN_NG (64 / EXMC_GEN_LANES)
#define EXMC_GEN_XGROUP(s) exmc::xgroup_sum_n<EXMC_GEN_LANES>(s)
#if EXMC_GEN_TABLE_IN_LDS || EXMC_GEN_WG
#define EXMC_GEN_LANES_NAME exmc_gen_lanes_split_lds
#include EXMC_CUSTOM_HEADER
#undef EXMC_GEN_LANES_NAME
#endif
#undef EXMC_GEN_LT
#undef EXMC_GEN_LT2
#undef EXMC_GEN_IT
#define EXMC_GEN_LT(i) lt[i]
#define EXMC_GEN_LT2(i) EXMC_GEN_LT2_GLOBAL(i)
#define EXMC_GEN_IT(i) ((const int*)(lt + EXMC_GEN_IOFF))[i]
#define EXMC_GEN_LANES_NAME exmc_gen_lanes_split_global
#include EXMC_CUSTOM_HEADER
#endif
#undef EXMC_GEN_LANES_SECTION
#endif


namespace exmc {

struct CustomConsts {
  const double* c;    // folded constants of the one-lane form
  const double* vc;   // 16-lane form: [EXMC_GEN_NVC uniform][16][EXMC_GEN_NLC per lane]
  const double* lt;   // lane layout: [uniform constants][per-unit columns][int32 index tables]
};

template <int G>
struct Custom;

#ifdef EXMC_GEN_ONE_LANE
template <>
struct Custom<1> : ModelDefaults {
  static constexpr int D = EXMC_GEN_D;
  static constexpr int DPL = D;
  using Consts = CustomConsts;
  struct Lane {};
  __device__ static __forceinline__ void load(const Consts&, int, Lane&) {}
  __device__ static __forceinline__ double logp_grad(const Consts& c, const Lane&, int,
                                                     const double (&q)[DPL], double (&g)[DPL]) {
    return exmc_gen_logp_grad(c.c, q, g);
  }
};
#endif

#ifdef EXMC_GEN_LANES
// Several dimensions per lane (exmc_amd/codegen_lanes.py): a chain over G lanes, dimension i in
// slot i / G of lane i % G. The position goes to the chain's LDS strip; the generated function
// walks the model's families of repeated terms (unit u on lane u % G), reduces the partial sums in
// one butterfly and gathers this lane's gradient entries from the adjoint strips.
template <>
struct Custom<EXMC_GEN_LANES> : ModelDefaults {
  static constexpr int G = EXMC_GEN_LANES;
  static constexpr int D = EXMC_GEN_D;
  static constexpr int DPL = EXMC_GEN_DPL;
  static_assert(DPL * G >= D, "every dimension has a slot");
  static constexpr bool kVregMath = true;   // the tree's own exp / log (nuts_run)
  // the two waves of the pipelined warmup never evaluate the model at the same time (the tree wave
  // only in the step-size searches, while the integrator waits at a barrier): one strip serves both
  static constexpr bool kPipeWarmup = true;
  static constexpr int kNutsWavesPerSimd = EXMC_GEN_WAVES_PER_SIMD;
  // one chain per wave and two waves per SIMD: what the hand-written sv kind runs with -- cross-row
  // sums through ds_bpermute, chain migration and the time-sliced issue priority (exmc_nuts.hpp)
  static constexpr bool kXRowLds = (G == 64 && EXMC_GEN_WAVES_PER_SIMD == 2);
  static constexpr bool kMigrate = (G == 64 && EXMC_GEN_WAVES_PER_SIMD == 2);
  static constexpr int kExtraLdsDoubles = (64 / G) * EXMC_GEN_LSH;
  // a short owner list (the strip cells whose sum is this lane's gradient entries) lives in registers
  static constexpr bool kEllRegs = EXMC_GEN_NELL <= 16;
  // tables up to 16 KB are staged in LDS by the kernels that run a lone wave per SIMD for long
  static constexpr bool kLdsTable = kGenLdsTable;
  static constexpr int kLdsDataDoubles = kLdsTable ? EXMC_GEN_NLT : 0;
  using Consts = CustomConsts;
  struct Lane {
    double* sh;   // the wavefront's scratch (attach_scratch / lane_setup)
    int xoff;     // the LDS image of the tables (offset in doubles), or -1: read from global memory
    const double* xs;   // the one-chain warmup's image of tables too large for xoff's budget (CustomSplit::stage), or null
    int ell[kEllRegs ? EXMC_GEN_NELL : 1];
  };
  // cooperative (whole workgroup); the caller synchronises afterwards
  __device__ static __forceinline__ bool stage_data(const Consts& c, double* dst) {
    for (int i = threadIdx.x; i < EXMC_GEN_NLT; i += blockDim.x) dst[i] = c.lt[i];
    return true;
  }
  // Round 6: the workgroup form of the sampling kernel (exmc_nuts.hpp nuts_kernel_wg) for a layout whose tables do
  // not fit beside a one-wave workgroup -- eight wavefronts around ONE image of the tables (EXMC_GEN_WG, decided by
  // the generator from the same sizes). The generated 500 x 20 regression read ten 16-byte rows per unit and lane
  // from L2 at every leapfrog, the same sixteen rows for the four chains of a wavefront.
  static constexpr int kWgWaves = EXMC_GEN_WG ? 8 : 0;
  static constexpr int kWgLdsLevels = 1;
  static constexpr int kWgImageDoubles = EXMC_GEN_WG ? EXMC_GEN_NLT : 0;
  __host__ __device__ static bool wg_ok(const Consts&) { return EXMC_GEN_WG != 0; }
  __device__ static __forceinline__ void wg_stage(const Consts& c, double* image) {
    for (int i = threadIdx.x; i < EXMC_GEN_NLT; i += blockDim.x) image[i] = c.lt[i];
  }
  __device__ static __forceinline__ void wg_attach(Lane& ln, double*, int image_offset) { ln.xoff = image_offset; }
  __device__ static __forceinline__ void load(const Consts& c, int l, Lane& ln) {
    ln.sh = nullptr;
    ln.xoff = -1;
    ln.xs = nullptr;
    if constexpr (kEllRegs) {
      const int* e = (const int*)c.lt + EXMC_GEN_ELL_OFF + l * EXMC_GEN_NELL;
#pragma unroll
      for (int j = 0; j < EXMC_GEN_NELL; j++) ln.ell[j] = e[j];
    }
  }
  __device__ static __forceinline__ double logp_grad(const Consts& c, const Lane& ln, int l,
                                                     const double (&q)[DPL], double (&g)[DPL]) {
    const int shoff = (int)(ln.sh - exmc_dyn_lds) + (int)((threadIdx.x & 63) / G) * EXMC_GEN_LSH;
#pragma unroll
    for (int k = 0; k < DPL; k++)
      if (l + k * G < D) exmc_dyn_lds[shoff + l + k * G] = q[k];
    wave_lds_fence();
    const int* el = kEllRegs ? ln.ell : ((const int*)c.lt + EXMC_GEN_ELL_OFF + l * EXMC_GEN_NELL);
#if EXMC_GEN_FAST_WINDOW
    // the fast window first; the exact forms again if any lane of the wavefront left it (see above)
    bool ok = true;
    double r;
    if (kLdsTable && ln.xoff >= 0) r = exmc_gen_lanes_lds_fast(c.lt, el, l, g, shoff, ln.xoff, ok);   // wave-uniform
    else r = exmc_gen_lanes_global_fast(c.lt, el, l, g, shoff, ok);
    if (__builtin_expect(__any(ok ? 0 : 1) == 0, 1)) return r;
    wave_lds_fence();
#endif
    if constexpr (kLdsTable) {
      if (ln.xoff >= 0) return exmc_gen_lanes_lds(c.lt, el, l, g, shoff, ln.xoff);   // wave-uniform
    }
    return exmc_gen_lanes_global(c.lt, el, l, g, shoff);
  }
  // the workgroup form: always from the image wg_attach pointed the lane at
  __device__ static __forceinline__ double logp_grad_staged(const Consts& c, const Lane& ln, int l,
                                                            const double (&q)[DPL], double (&g)[DPL]) {
    const int shoff = (int)(ln.sh - exmc_dyn_lds) + (int)((threadIdx.x & 63) / G) * EXMC_GEN_LSH;
#pragma unroll
    for (int k = 0; k < DPL; k++)
      if (l + k * G < D) exmc_dyn_lds[shoff + l + k * G] = q[k];
    wave_lds_fence();
    const int* el = kEllRegs ? ln.ell : ((const int*)c.lt + EXMC_GEN_ELL_OFF + l * EXMC_GEN_NELL);
#if EXMC_GEN_FAST_WINDOW
    bool ok = true;
    const double r = exmc_gen_lanes_lds_fast(c.lt, el, l, g, shoff, ln.xoff, ok);
    if (__builtin_expect(__any(ok ? 0 : 1) == 0, 1)) return r;
    wave_lds_fence();
#endif
    return exmc_gen_lanes_lds(c.lt, el, l, g, shoff, ln.xoff);
  }
};

#if EXMC_GEN_LANES < 64
// The same layout for ONE chain on a whole wavefront -- the shared warmup (exmc_hip.hip: warmup
// lanes 64 for a generated layout of fewer lanes per chain, the default). All 64 / G lane groups
// carry the chain redundantly (kCoop: they stay in lockstep, group 0 writes) and share group 0's
// strips; in logp_grad group g evaluates the slots g, g + 64 / G, ... of every family, and the
// groups' reduced sums are added in group order (exmc_gen_lanes_split). A 500-observation
// likelihood at 16 lanes per chain: 8 slots per lane instead of 32. The sums differ from the
// sampling layout's in their order, so this is a layout of its own for the checker too
// (tests/gen_checker.py model(..., wave_split=True)), like the hand-written logistic kind's 64-lane
// warmup layout.
struct CustomSplit : Custom<EXMC_GEN_LANES> {
  using Base = Custom<EXMC_GEN_LANES>;
  static constexpr bool kCoop = true;
  static constexpr int kWgWaves = 0;          // (a warmup form: no sampling kernel of its own)
  static constexpr int kWgImageDoubles = 0;
  // tables too large for the sampling layout's LDS budget (EXMC_GEN_WG): the one-workgroup warmup has a compute
  // unit's whole LDS to itself and keeps an image of them (warmup_kernel: P.stage_model, Lane::xs)
  static constexpr int kStageDoubles = EXMC_GEN_WG ? EXMC_GEN_NLT : 0;
  __device__ static __forceinline__ void stage(const Consts& c, double* dst) {
    for (int i = threadIdx.x; i < EXMC_GEN_NLT; i += blockDim.x) dst[i] = c.lt[i];
  }
  // one wave: where this form pays the pass is nearly all model (like the hand-written logistic
  // kind, ModelDefaults::kPipeWarmup), and a plug-in build is spared its heaviest kernel
  static constexpr bool kPipeWarmup = false;
  static constexpr bool kHasPipeWarmup = false;
  __device__ static __forceinline__ double logp_grad(const Consts& c, const Lane& ln, int l,
                                                     const double (&q)[DPL], double (&g)[DPL]) {
    const int shoff = (int)(ln.sh - exmc_dyn_lds);   // group 0's strips for every group
#pragma unroll
    for (int k = 0; k < DPL; k++)
      if (l + k * G < D) exmc_dyn_lds[shoff + l + k * G] = q[k];   // the same value from every group
    wave_lds_fence();
    const int* el = kEllRegs ? ln.ell : ((const int*)c.lt + EXMC_GEN_ELL_OFF + l * EXMC_GEN_NELL);
#if EXMC_GEN_TABLE_IN_LDS || EXMC_GEN_WG
    const int xo = (EXMC_GEN_WG && ln.xs) ? (int)(ln.xs - exmc_dyn_lds) : ln.xoff;
    if (xo >= 0) return exmc_gen_lanes_split_lds(c.lt, el, l, g, shoff, xo);   // wave-uniform
#endif
    return exmc_gen_lanes_split_global(c.lt, el, l, g, shoff, -1);
  }
};
#endif
#endif

#ifdef EXMC_GEN_VEC
// Plates across lanes (exmc_amd/codegen_vec.py): lane l owns dimension l and evaluates the units
// of the vectorised families assigned to it with its own row of constants; the partial
// log-density and the partial adjoints of the shared variables cross the group in one butterfly.
template <>
struct Custom<16> : ModelDefaults {
  static constexpr int G = 16;
  static constexpr int D = EXMC_GEN_D;
  static constexpr int DPL = 1;
  static constexpr bool kVregMath = true;   // the tree's own exp / log (nuts_run), as EightSchools<16>
  static_assert(D <= G, "one dimension per lane");
  using Consts = CustomConsts;
  struct Lane {
    double lc[EXMC_GEN_NLC];
  };
  __device__ static __forceinline__ void load(const Consts& c, int l, Lane& ln) {
#pragma unroll
    for (int k = 0; k < EXMC_GEN_NLC; k++) ln.lc[k] = c.vc[EXMC_GEN_NVC + l * EXMC_GEN_NLC + k];
  }
  template <int... I>
  __device__ static __forceinline__ void bcast_all(double x, double (&qs)[D],
                                                   std::integer_sequence<int, I...>) {
    // only the dimensions the generated code reads as shared variables are broadcast
    ((qs[I] = ((EXMC_GEN_QMASK >> I) & 1u) ? group_bcast_c<G, I>(x) : 0.0), ...);
  }
  __device__ static __forceinline__ double logp_grad(const Consts& c, const Lane& ln, int l,
                                                     const double (&q)[DPL], double (&g)[DPL]) {
    double qs[D], s[EXMC_GEN_NS], sg[D], gown, slp;
    bcast_all(q[0], qs, std::make_integer_sequence<int, D>{});
#ifdef EXMC_GENV_FAST
    bool ok = true;
    exmc_gen_lane_fast(c.vc, ln.lc, qs, q[0], s, &gown, sg, &slp, ok);
    if (__builtin_expect(__any(!ok), 0))   // wave-uniform: some lane's exp / log argument left the main path's domain
#endif
      exmc_gen_lane(c.vc, ln.lc, qs, q[0], s, &gown, sg, &slp);
    group_allsum_n<G, EXMC_GEN_NS>(s);
    constexpr int smap[D] = EXMC_GEN_SMAP;
    double gi = 0.0;
#pragma unroll
    for (int i = 0; i < D; i++) {
      const double tot = (smap[i] > 0) ? (sg[i] + s[smap[i] > 0 ? smap[i] : 0]) : sg[i];
      gi = (l == i) ? tot : gi;
    }
    g[0] = gi + gown;
    return slp + s[0];
  }
};
#endif

}  // namespace exmc
#endif
