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
struct EightSchools {
  static constexpr int D = 10;
  static constexpr int DPL = (D + G - 1) / G;
  using Consts = EightSchoolsConsts;
  struct Lane {
    double y[DPL], sg[DPL], lsg[DPL];
  };

  __device__ static __forceinline__ void load(const Consts& c, int l, Lane& ln) {
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      const int j = (i >= 2 && i < D) ? (i - 2) : 0;
      ln.y[k] = c.y[j];
      ln.sg[k] = c.sg[j];
      ln.lsg[k] = c.lsg[j];
    }
  }

  __device__ static __forceinline__ double logp_grad(const Consts& c, const Lane& ln, int l,
                                                     const double (&q)[DPL], double (&g)[DPL]) {
    const double mu = group_bcast<G>(q[0 / G], 0 % G);
    const double zraw = group_bcast<G>(q[1 / G], 1 % G);
    const double zc = clamp200(zraw);
    const double tau = exmc_exp(zc);
    double L[DPL], A[DPL], B[DPL], T[DPL];
    bool valid[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      valid[k] = i < D;
      const bool isth = (i >= 2) && (i < D);
      const double th = q[k];
      const double theta = mu + tau * th;
      const double z = (ln.y[k] - theta) / ln.sg[k];
      const double a = z / ln.sg[k];
      L[k] = isth ? ((-0.5 * (z * z)) - ln.lsg[k]) : 0.0;
      A[k] = isth ? a : 0.0;
      B[k] = isth ? (a * th) : 0.0;
      T[k] = -0.5 * (th * th + c.c1);
      g[k] = (-th) + a * tau;
    }
    // three independent group sums in one butterfly pass
    double s3[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      s3[0] = valid[k] ? (s3[0] + L[k]) : s3[0];
      s3[1] = valid[k] ? (s3[1] + A[k]) : s3[1];
      s3[2] = valid[k] ? (s3[2] + B[k]) : s3[2];
    }
    group_allsum_n<G, 3>(s3);
    const double lik = s3[0], sa = s3[1], sb = s3[2];
    const double zmu = (mu - 0.0) / 5.0;
    const double t_mu = -0.5 * (zmu * zmu + c.c_mu);
    const double zt = tau / 5.0;
    const double zt2 = zt * zt;
    const double t_tau = (c.c_hc - exmc_log(1.0 + zt2)) + zc;
    const double g_mu = (-(zmu / 5.0)) + sa;
    const double dhc = -(((2.0 * zt) / 5.0) / (1.0 + zt2));
    const bool in = (zraw > -200.0) && (zraw < 200.0);
    const double g_tau = in ? ((dhc + sb) * tau + 1.0) : 0.0;
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      if (i == 0) { T[k] = t_mu; g[k] = g_mu; }
      if (i == 1) { T[k] = t_tau; g[k] = g_tau; }
    }
    return group_sum_slots<G, DPL>(T, valid, l, lik);
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
struct Simple {
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

__device__ __forceinline__ double lanczos_val_d(const SVConsts& c, double x, double& dx) {
  const double t = x + 6.5;
  double ag = c.lanczos[0];
  double dag = 0.0;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    const double den = x + (double)(i - 1) * 1.0;
    const double term = c.lanczos[i] / den;
    ag = ag + term;
    dag = dag - term / den;
  }
  const double lt = exmc_log(t);
  dx = ((lt + (x - 0.5) / t) - 1.0) + dag / ag;
  return ((c.half_log_2pi32 + (x - 0.5) * lt) - t) + exmc_log(ag);
}

template <int G>
struct SV {
  static constexpr int T = 100;
  static constexpr int D = T + 2;
  static constexpr int DPL = (D + G - 1) / G;
  static_assert(G >= 2, "sv spreads a chain over >= 2 lanes");
  using Consts = SVConsts;
  struct Lane {
    double r[DPL];
  };
  __device__ static __forceinline__ void load(const Consts& c, int l, Lane& ln) {
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      ln.r[k] = c.r[i < T ? i : 0];
    }
  }
  __device__ static __forceinline__ double logp_grad(const Consts& c, const Lane& ln, int l,
                                                     const double (&q)[DPL], double (&g)[DPL]) {
    const double zs_raw = group_bcast<G>(q[T / G], T % G);
    const double zn_raw = group_bcast<G>(q[(T + 1) / G], (T + 1) % G);
    const double zs = clamp200(zs_raw), zn = clamp200(zn_raw);
    const double sigma = exmc_exp(zs), nu = exmc_exp(zn);
    const double ss = fmax(sigma, c.tiny32);
    const double sdf = fmax(nu, c.tiny32);
    const double t_sigma = (c.log_lam_s32 - c.lam_s * sigma) + zs;
    const double t_nu = (c.log_lam_n32 - c.lam_n * nu) + zn;
    const double hp1 = (sdf + 1.0) / 2.0, h = sdf / 2.0;
    double d1, d0;
    const double lg1 = lanczos_val_d(c, hp1, d1);
    const double lg0 = lanczos_val_d(c, h, d0);
    const double An = (lg1 - lg0) - 0.5 * exmc_log(sdf * c.pi32);
    const double dAn = (0.5 * d1 - 0.5 * d0) - 0.5 / sdf;
    const double cn = c.log2pi32 + 2.0 * exmc_log(ss);
    const int lane = threadIdx.x & 63;
    const int base = lane & ~(G - 1);
    const int prev_lane = base | ((l + G - 1) & (G - 1));
    const int next_lane = base | ((l + 1) & (G - 1));
    double P[DPL], LL[DPL], E2[DPL], DN[DPL], de[DPL];
    bool valid[DPL];
    // previous-state values: dim i-1 is (lane l-1, slot k) or (lane G-1, slot k-1) when l == 0
    double qprev[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const double same = __shfl(q[k], prev_lane, 64);
      const double lower = (k > 0) ? __shfl(q[k > 0 ? k - 1 : 0], prev_lane, 64) : 0.0;
      qprev[k] = (l > 0) ? same : lower;
    }
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      valid[k] = i < D;
      const bool ist = i < T;
      const double qi = q[k];
      const double prev = (i == 0) ? 0.0 : qprev[k];
      const double e = (qi - prev) / ss;
      const double z = ln.r[k] * exmc_exp(-qi);
      const double w = (z * z) / sdf;
      const double lg = exmc_log(1.0 + w);
      const double wr = w / (1.0 + w);
      P[k] = ist ? (-0.5 * (e * e + cn)) : 0.0;
      E2[k] = ist ? (e * e - 1.0) : 0.0;
      de[k] = ist ? (-(e / ss)) : 0.0;
      LL[k] = ist ? ((An - qi) - hp1 * lg) : 0.0;
      DN[k] = ist ? ((dAn - 0.5 * lg) + (hp1 * wr) / sdf) : 0.0;
      g[k] = -1.0 + (sdf + 1.0) * wr;
    }
    // dP_{t+1}/ds_{t+1} from dim i+1: (lane l+1, slot k) or (lane 0, slot k+1) when l == G-1
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      const double same = __shfl(de[k], next_lane, 64);
      const double upper = (k + 1 < DPL) ? __shfl(de[k + 1 < DPL ? k + 1 : k], next_lane, 64) : 0.0;
      double nxt = (l < G - 1) ? same : upper;
      nxt = (i + 1 < T) ? nxt : 0.0;
      g[k] = g[k] + (de[k] - nxt);
    }
    const double sp = group_sum_slots<G, DPL>(P, valid, l, 0.0);
    const double sl = group_sum_slots<G, DPL>(LL, valid, l, 0.0);
    const double se = group_sum_slots<G, DPL>(E2, valid, l, 0.0);
    const double sn = group_sum_slots<G, DPL>(DN, valid, l, 0.0);
    const bool in_s = (zs_raw > -200.0) && (zs_raw < 200.0);
    const bool in_n = (zn_raw > -200.0) && (zn_raw < 200.0);
    const double g_s = in_s ? ((se - c.lam_s * sigma) + 1.0) : 0.0;
    const double g_n = in_n ? ((sn * nu - c.lam_n * nu) + 1.0) : 0.0;
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      if (i == T) g[k] = g_s;
      if (i == T + 1) g[k] = g_n;
    }
    return ((t_sigma + t_nu) + sp) + sl;
  }
};

}  // namespace exmc
