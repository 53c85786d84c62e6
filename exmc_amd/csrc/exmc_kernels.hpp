// exmc_kernels.hpp — the other gfx950 kernels of the hot path (the NUTS transition kernel and
// the adaptation warmup are in exmc_nuts.hpp).
//
//   multi_step_kernel  B2 `multi_step_fn` contract, chain-batched (batched_leapfrog.ex:50-101).
//   init_chains_kernel seed :rand, init position, first logp/grad (sampler.ex:154-165,339-349).
//   find_eps_kernel    find_reasonable_epsilon_with_rng (sampler.ex:451-530).
//   logp_grad_kernel   vag_fn batched (compiler.ex:131-141).
//   ess_series_kernel  Diagnostics.ess (diagnostics.ex:42-52,123-167); rank_scores_kernel the
//                      rank-normalisation of Diagnostics.ess_bulk (diagnostics.ex:60-72,186-219).
//
// All are launched with one wavefront per workgroup (64 threads) and M::kExtraLdsDoubles * 8
// bytes of dynamic LDS. For wave-cooperative models (M::kCoop, the MFMA logistic) lane groups
// without a chain shadow the last chain instead of leaving, so that logp_grad sees a full wave.
#pragma once

#include "exmc_ess.hpp"
#include "exmc_nuts.hpp"

namespace exmc {

constexpr int kAuxBlock = 64;

template <class M>
__host__ __device__ constexpr size_t aux_lds_bytes() { return (size_t)M::kExtraLdsDoubles * 8; }

template <class M>
__device__ __forceinline__ void attach_scratch(typename M::Lane& ln, double* sh) {
  if constexpr (M::kExtraLdsDoubles > 0) ln.sh = sh;
}

// ------------------------------------------------------------------------------------------
// B2: chain-batched multi_step (batched_leapfrog.ex:50-101). all_* rows are [step][dim][chain].
// ------------------------------------------------------------------------------------------
struct MultiStepParams {
  const double* q;  // [D][C]
  const double* p;
  const double* g;
  double eps;
  const double* inv_mass;  // dev [D]
  int n_steps;
  int n_chains;
  double* all_q;     // [n][D][C]
  double* all_p;
  double* all_g;
  double* all_logp;  // [n][C]
};

template <class M, int G>
__global__ void __launch_bounds__(kAuxBlock) multi_step_kernel(MultiStepParams P,
                                                               typename M::Consts mc) {
  constexpr int D = M::D, DPL = M::DPL;
  extern __shared__ double xlds[];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = threadIdx.x & (G - 1);
  const int C = P.n_chains;
  const bool has_chain = (tid / G) < C;
  const int chain = has_chain ? (tid / G) : (C - 1);
  if (!M::kCoop && !has_chain) return;
  typename M::Lane ln;
  M::load(mc, l, ln);
  attach_scratch<M>(ln, xlds);
  double q[DPL], p[DPL], g[DPL], im[DPL];
  bool valid[DPL];
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = l + k * G;
    valid[k] = i < D;
    const size_t o = (size_t)i * C + chain;
    q[k] = valid[k] ? P.q[o] : 0.0;
    p[k] = valid[k] ? P.p[o] : 0.0;
    g[k] = valid[k] ? P.g[o] : 0.0;
    im[k] = valid[k] ? P.inv_mass[i] : 1.0;
  }
  const double eps = P.eps;
  const double h = eps / 2.0;
  for (int s = 0; s < P.n_steps; s++) {
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const double ph = p[k] + h * g[k];
      p[k] = ph;
      q[k] = q[k] + eps * (im[k] * ph);
    }
    const double logp = M::logp_grad(mc, ln, l, q, g);
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      p[k] = p[k] + h * g[k];
      if (valid[k] && has_chain) {
        const size_t o = ((size_t)s * D + (l + k * G)) * C + chain;
        P.all_q[o] = q[k];
        P.all_p[o] = p[k];
        P.all_g[o] = g[k];
      }
    }
    if (l == 0 && has_chain) P.all_logp[(size_t)s * C + chain] = logp;
  }
}

// vag_fn batched: q [D][C] -> logp [C], grad [D][C]
template <class M, int G>
__global__ void __launch_bounds__(kAuxBlock)
logp_grad_kernel(const double* qin, int C, double* logp, double* grad, typename M::Consts mc) {
  constexpr int D = M::D, DPL = M::DPL;
  extern __shared__ double xlds[];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = threadIdx.x & (G - 1);
  const bool has_chain = (tid / G) < C;
  const int chain = has_chain ? (tid / G) : (C - 1);
  if (!M::kCoop && !has_chain) return;
  typename M::Lane ln;
  M::load(mc, l, ln);
  attach_scratch<M>(ln, xlds);
  double q[DPL], g[DPL];
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = l + k * G;
    q[k] = (i < D) ? qin[(size_t)i * C + chain] : 0.0;
    g[k] = 0.0;
  }
  const double lp = M::logp_grad(mc, ln, l, q, g);
  if (!has_chain) return;
  if (l == 0) logp[chain] = lp;
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = l + k * G;
    if (i < D) grad[(size_t)i * C + chain] = g[k];
  }
}

// sampler.ex:154-165, 339-349, 1083-1093: seed, initial position, first logp/grad
struct InitParams {
  ChainState st;
  int n_chains;
  int chain_lo;
  uint64_t base_seed;
  const double* init_q;  // dev [D] or null
  const uint64_t* zig_ki;
  const double* zig_wi;
  const double* zig_fi;
  double nor_r;
  FlatOrder flat;
};

template <class M, int G>
__global__ void __launch_bounds__(kAuxBlock) init_chains_kernel(InitParams P,
                                                                typename M::Consts mc) {
  constexpr int D = M::D, DPL = M::DPL;
  extern __shared__ double xlds[];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = threadIdx.x & (G - 1);
  const int C = P.n_chains;
  const bool has_chain = (tid / G) < C;
  const int chain = has_chain ? (tid / G) : (C - 1);
  if (!M::kCoop && !has_chain) return;
  typename M::Lane ln;
  M::load(mc, l, ln);
  attach_scratch<M>(ln, xlds);
  const ZigTables zt{P.zig_ki, P.zig_wi, P.zig_fi};
  Rng rng;
  rng_seed(rng, P.base_seed + 7919ULL * (uint64_t)(P.chain_lo + chain));
  double q[DPL], g[DPL];
#pragma unroll
  for (int k = 0; k < DPL; k++) q[k] = g[k] = 0.0;
  if (P.init_q) {
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      if (i < D) q[k] = P.init_q[i];
    }
  } else {
    // sampler.ex:339-349: variate r fills entry r of the reference's flat vector
    int rank[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const int i = l + k * G;
      rank[k] = (i < D) ? (P.flat.rank ? P.flat.rank[i] : i) : D;
    }
    for (int r = 0; r < D; r++) {
      const double z = rng_normal(rng, zt, P.nor_r);
#pragma unroll
      for (int k = 0; k < DPL; k++)
        if (rank[k] == r) q[k] = z * 0.1;
    }
  }
  const double lp = M::logp_grad(mc, ln, l, q, g);
  if (!has_chain) return;
  if (l == 0) {
    P.st.logp[chain] = lp;
    P.st.rng[chain] = rng.a;
    P.st.rng[(size_t)C + chain] = rng.b;
  }
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = l + k * G;
    if (i < D) {
      P.st.q[(size_t)i * C + chain] = q[k];
      P.st.g[(size_t)i * C + chain] = g[k];
    }
  }
}

// find_reasonable_epsilon_with_rng (sampler.ex:451-530) for chain 0 of the state buffers (the
// host-driven warmup path; the device warmup calls find_eps_dev directly). Launched with
// nuts_lds_bytes<M, 0>() of dynamic LDS.
struct FindEpsParams {
  ChainState st;
  int n_chains;
  const double* inv_mass;
  const double* sqrt_inv_mass;
  double log_half;  // libm log(0.5), computed on the host as :math.log(0.5)
  double* eps_out;
  const uint64_t* zig_ki;
  const double* zig_wi;
  const double* zig_fi;
  double nor_r;
  FlatOrder flat;
};

template <class M, int G>
__global__ void __launch_bounds__(kNutsBlock) find_eps_kernel(FindEpsParams P,
                                                              typename M::Consts mc) {
  constexpr int DPL = M::DPL;
  constexpr int NSLOT = nuts_nslot<M>();
  extern __shared__ double lds[];
  const ZigTables zt = stage_zig_tables<0, NSLOT>(lds, P.zig_ki, P.zig_wi, P.zig_fi);
  const bool writer = threadIdx.x < G;
  if (blockIdx.x != 0 || (!M::kCoop && !writer)) return;
  NutsLane<M, G> L;
  lane_setup<M, G, 0>(L, mc, lds, nullptr, P.inv_mass, P.sqrt_inv_mass, zt, P.nor_r, P.flat);
  ChainRegs<DPL> st;
  chain_load<M, G>(P.st, P.n_chains, 0, L.l, st);
  const double eps = find_eps_dev<M, G>(mc, L, st, P.log_half);
  if (writer && L.l == 0) {
    *P.eps_out = eps;
    P.st.rng[0] = st.rng.a;
    P.st.rng[(size_t)P.n_chains] = st.rng.b;
  }
}

#ifndef EXMC_PLUGIN_PART   // the model-independent kernels live in exmc_hip.hip's translation unit only
// Diagnostics.ess (diagnostics.ex:42-52, 123-167) over a [S][D][C] trace, one lane per (dim, chain)
// series (lanes sweep chains first, so every load of a wavefront is one coalesced row segment);
// the per-series routine is exmc_ess.hpp.
constexpr int kEssBlock = 64;

// One lane per series for the mean and the first sixteen lags. A series that has not met Geyer's
// cut by then goes on a worklist for ess_tail_kernel; `work` = [count | items...], null = finish
// every series here.
struct EssTailItem {
  double mean, var, tau;
  int series, next_lag;
};

__global__ void __launch_bounds__(kEssBlock)
ess_series_kernel(const double* draws, int S, int D, int C, double* ess_out, int* work_count,
                  EssTailItem* work)
#ifdef EXMC_COMMON_DECL_ONLY
;   // defined in the prebuilt exmc_common object (exmc_common.hip)
#else
{
  const size_t series = (size_t)blockIdx.x * kEssBlock + threadIdx.x;   // dim * C + chain
  const size_t stride = (size_t)D * C;
  if (series >= stride) return;
  if (work == nullptr) {
    ess_out[series] = ess_series(draws + series, stride, S);
    return;
  }
  EssPartial part;
  const double e = ess_series(draws + series, stride, S, 1, &part);
  if (part.finished) {
    ess_out[series] = e;
  } else {
    const int slot = atomicAdd(work_count, 1);
    work[slot] = EssTailItem{part.mean, part.var, part.tau, (int)series, part.next_lag};
  }
}
#endif

// The rest of a series' lags, a wavefront per series: the centred series in LDS, lane k sums lag
// l0 + k left to right over i exactly as ess_series does (`acc + c[i] * c[i + lag]`,
// diagnostics.ex:137-141), 64 lags per sweep; then Geyer's pairs in order (diagnostics.ex:147-167).
__global__ void __launch_bounds__(64)
ess_tail_kernel(const double* draws, int S, int D, int C, double* ess_out, const int* work_count,
                const EssTailItem* work)
#ifdef EXMC_COMMON_DECL_ONLY
;   // defined in the prebuilt exmc_common object (exmc_common.hip)
#else
{
  extern __shared__ double cen[];   // S centred values
  const int lane = threadIdx.x;
  const size_t stride = (size_t)D * C;
  const int count = *work_count;
  const int max_k = (S - 1) / 2;
  for (int item = blockIdx.x; item < count; item += gridDim.x) {
    const EssTailItem it = work[item];
    const double* x = draws + it.series;
    __syncthreads();   // the previous item's readers are done with cen[]
    for (int i = lane; i < S; i += 64) cen[i] = x[(size_t)i * stride] - it.mean;
    __syncthreads();
    double tau = it.tau;
    bool done = false;
    for (int l0 = it.next_lag; !done; l0 += 64) {
      const int lag = l0 + lane;
      const int n = S - lag;   // terms of this lane's sum; <= 0: the empty sum 0.0
      double acc = 0.0;
      for (int i = 0; i < S - l0; i++) {   // the longest sum of the sweep (lane 0)
        const double prod = cen[i] * cen[(i + lag < S) ? (i + lag) : (S - 1)];
        acc = (i < n) ? (acc + prod) : acc;
      }
      // pair kk of this sweep: lags l0 + 2 kk and l0 + 2 kk + 1, on lanes 2 kk and 2 kk + 1
      const double r = acc / it.var;
      const double rn = __shfl_down(r, 1, 64);
      const double pair = r + rn;   // meaningful on even lanes
#pragma unroll 1
      for (int kk = 0; kk < 32 && !done; kk++) {
        const int k = l0 / 2 + kk;
        if (k > max_k) {
          done = true;
        } else {
          const double pk = __shfl(pair, 2 * kk, 64);
          if (pk > 0) tau += 2 * pk;
          else done = true;
        }
      }
    }
    if (lane == 0) {
      const double t = tau > 1.0 ? tau : 1.0;   // max(tau, 1.0), diagnostics.ex:165
      ess_out[it.series] = S / t;
    }
  }
}
#endif

// Diagnostics.ess_bulk (diagnostics.ex:60-72, 186-219), first half: every series is replaced by
// the normal scores of its ranks -- average rank for ties, (r - 3/8) / (n + 1/4), the reference's
// rational probit approximation with sqrt IEEE and log from exmc_detmath.h. One workgroup per
// series; the ranks are counted (every thread compares its elements with the whole series in
// LDS). The scores go to a second [S][D][C] array on which ess_series_kernel then runs.
__device__ __forceinline__ double probit_inner_dev(double p) {
  const double t = __dsqrt_rn(-2.0 * exmc_log(p));
  return t - (2.515517 + 0.802853 * t + 0.010328 * t * t) /
                 (1.0 + 1.432788 * t + 0.189269 * t * t + 0.001308 * t * t * t);
}

__global__ void __launch_bounds__(256)
rank_scores_kernel(const double* draws, int S, int D, int C, double* scores)
#ifdef EXMC_COMMON_DECL_ONLY
;   // defined in the prebuilt exmc_common object (exmc_common.hip)
#else
{
  extern __shared__ double raw[];  // S values
  const size_t series = blockIdx.x;  // dim * C + chain
  const size_t stride = (size_t)D * C;
  const double* x = draws + series;
  double* z = scores + series;
  for (int i = threadIdx.x; i < S; i += blockDim.x) raw[i] = x[(size_t)i * stride];
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    const double xi = raw[i];
    int lo = 0, eq = 0;
    for (int j = 0; j < S; j++) {
      const double xj = raw[j];
      lo += (xj < xi) ? 1 : 0;
      eq += (xj == xi) ? 1 : 0;
    }
    const double avg = (double)(lo + 1) + (double)(eq - 1) / 2.0;
    const double pr = (avg - 0.375) / ((double)S + 0.25);
    z[(size_t)i * stride] = (pr < 0.5) ? -probit_inner_dev(pr) : probit_inner_dev(1.0 - pr);
  }
}
#endif

// The same scores by SORTING the series (round 5): ranks are integers, so how they are found is free.
// One workgroup per series: (value, index) pairs in LDS, padded with (+inf, index >= S) to a power
// of two P, a bitonic sort ascending by value then index (P/2 compare-exchanges per stage over 256
// threads, log2(P) (log2(P) + 1) / 2 stages), then every position walks to the ends of its run of
// equal values: lo = elements below the run, eq = its length -- the numbers the counting kernel gets
// from S comparisons per element. 1000 draws: ~1.1 k instructions per thread instead of ~16 k. A
// series that holds a NaN is ranked by the counting rule (every comparison with a NaN is false, which
// the sort cannot reproduce); the worst case of the walks (a constant series) costs what the counting
// kernel always costs. P <= 4096 (48 KB of LDS); the host takes the counting kernel above that.
constexpr int kRankSortMaxP = 4096;
__global__ void __launch_bounds__(256)
rank_scores_sort_kernel(const double* draws, int S, int P, int D, int C, double* scores)
#ifdef EXMC_COMMON_DECL_ONLY
;   // defined in the prebuilt exmc_common object (exmc_common.hip)
#else
{
  extern __shared__ double raw[];            // P keys, then P indices
  int* const idx = (int*)(raw + P);
  const size_t series = blockIdx.x;          // dim * C + chain
  const size_t stride = (size_t)D * C;
  const double* x = draws + series;
  double* z = scores + series;
  int nan_here = 0;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    const double v = (i < S) ? x[(size_t)i * stride] : __longlong_as_double(0x7FF0000000000000LL);
    nan_here |= (v != v) ? 1 : 0;
    raw[i] = v;
    idx[i] = i;
  }
  const bool any_nan = __syncthreads_or(nan_here) != 0;
  if (any_nan) {                             // workgroup-uniform: the counting rule of rank_scores_kernel
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
      const double xi = raw[i];
      int lo = 0, eq = 0;
      for (int j = 0; j < S; j++) {
        const double xj = raw[j];
        lo += (xj < xi) ? 1 : 0;
        eq += (xj == xi) ? 1 : 0;
      }
      const double avg = (double)(lo + 1) + (double)(eq - 1) / 2.0;
      const double pr = (avg - 0.375) / ((double)S + 0.25);
      z[(size_t)i * stride] = (pr < 0.5) ? -probit_inner_dev(pr) : probit_inner_dev(1.0 - pr);
    }
    return;
  }
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
        const int a = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // the lower index of pair t at distance j
        const int b = a + j;
        const bool up = (a & k) == 0;
        const double ka = raw[a], kb = raw[b];
        const int ia = idx[a], ib = idx[b];
        const bool a_after_b = (ka > kb) || (ka == kb && ia > ib);
        if (a_after_b == up) {
          raw[a] = kb; raw[b] = ka;
          idx[a] = ib; idx[b] = ia;
        }
      }
      __syncthreads();
    }
  }
  for (int p = threadIdx.x; p < S; p += blockDim.x) {   // the S real elements are the first S positions
    const double v = raw[p];
    int a = p, b = p + 1;
    while (a > 0 && raw[a - 1] == v) a--;
    while (b < S && raw[b] == v) b++;
    const int lo = a, eq = b - a;
    const double avg = (double)(lo + 1) + (double)(eq - 1) / 2.0;
    const double pr = (avg - 0.375) / ((double)S + 0.25);
    z[(size_t)idx[p] * stride] = (pr < 0.5) ? -probit_inner_dev(pr) : probit_inner_dev(1.0 - pr);
  }
}
#endif

// Diagnostics.rhat (diagnostics.ex:80-115), split R-hat of one dimension per workgroup over a
// [S][D][C] trace: each chain split at S/2, both halves trimmed to the shorter length; half-chain
// means and variances summed left to right over draws (one thread per half-chain, coalesced over
// chains), then the between / within sums over the 2C half-chains in the reference's order
// (chain 0 first half, chain 0 second half, chain 1 ...) by one thread. stats: scratch [D][2][2C].
// Two launches: `phase` 0 fills the half-chain statistics on a (dimension, slice of half-chains)
// grid -- with one workgroup per dimension the 328 MB trace of the bench was read by 10 CUs in
// 8.9 ms --, `phase` 1 is the serial tail of one thread per dimension.
__global__ void __launch_bounds__(256)
rhat_kernel(const double* draws, int S, int D, int C, double* stats, double* rhat_out, int phase)
#ifdef EXMC_COMMON_DECL_ONLY
;   // defined in the prebuilt exmc_common object (exmc_common.hip)
#else
{
  const int dim = blockIdx.x;
  const int mid = S / 2;
  const int len = mid < (S - mid) ? mid : (S - mid);
  const int m = 2 * C;
  double* means = stats + (size_t)dim * 2 * m;
  double* vars = means + m;
  const size_t stride = (size_t)D * C;
  for (int h = blockIdx.y * blockDim.x + threadIdx.x; phase == 0 && h < m;
       h += gridDim.y * blockDim.x) {
    const int c = h % C, half = h / C;     // threads sweep chains first: coalesced loads
    const double* x = draws + (size_t)(half ? mid : 0) * stride + (size_t)dim * C + c;
    double sum = 0.0;
    for (int i = 0; i < len; i++) sum += x[(size_t)i * stride];
    const double cm = sum / len;
    double ss = 0.0;
    for (int i = 0; i < len; i++) {
      const double dv = x[(size_t)i * stride] - cm;
      ss += dv * dv;
    }
    means[2 * c + half] = cm;
    vars[2 * c + half] = ss / (len - 1);
  }
  if (phase == 1 && threadIdx.x == 0) {
    double gm = 0.0;
    for (int k = 0; k < m; k++) gm += means[k];
    gm /= m;
    double b = 0.0, w = 0.0;
    for (int k = 0; k < m; k++) {
      const double dm = means[k] - gm;
      b += dm * dm;
      w += vars[k];
    }
    b = (double)len / (m - 1) * b;
    w /= m;
    const double var_hat = (double)(len - 1) / len * w + b / len;
    rhat_out[dim] = __dsqrt_rn(var_hat / w);
  }
}
#endif

#endif  // EXMC_PLUGIN_PART

// ------------------------------------------------------------------------------------------
// B2': the fused-chain hook of the speculative path (tree.ex:613-653, do_dispatch) --
// leapfrog_chain_normal(q, p, inv_mass, k, signed_eps, mu, sigma): K leapfrog steps of a chain of d
// independent Normal(mu, sigma) coordinates in ONE launch, the rows of multi_step_fn out
// (batched_leapfrog.ex:50-101; raw logp, :87). The caller hands over no gradient (tree.ex:637): the
// first half-kick takes it at q. One wavefront per chain; lane l owns dimensions l, l + 64, l + 128,
// l + 192 (d <= 256, the hook's own bound, tree.ex:636); the density's sum over the dimensions is the
// 64-lane group sum of every other kernel here (lane partials in ascending dimension, then the
// butterfly) = the checker's lane_sum(.., 64, ..). Rows per chain [k][d] row-major as the hook
// returns them: a wavefront writes a row in runs of 512 contiguous bytes.
// ------------------------------------------------------------------------------------------
struct ChainNormalParams {
  const double* q;         // [C][d]
  const double* p;         // [C][d]
  const double* inv_mass;  // [d]
  int d, k, n_chains;
  double eps;              // signed
  double mu, sigma;
  double tiny32, log2pi32; // Nx.tensor(1.0e-30), log(Nx.tensor(2 pi)): f32 literals (normal.ex:18-19)
  double* q_chain;         // [C][k][d]
  double* p_chain;
  double* g_chain;
  double* logp_chain;      // [C][k]
};

constexpr int kChainNormalMaxD = 256;

__global__ void __launch_bounds__(64) leapfrog_chain_normal_kernel(ChainNormalParams P)
#ifdef EXMC_COMMON_DECL_ONLY
;   // defined in the prebuilt exmc_common object (exmc_common.hip)
#else
{
  constexpr int G = 64, DPL = kChainNormalMaxD / G;
  const int chain = blockIdx.x;
  const int l = threadIdx.x;
  const int d = P.d;
  double q[DPL], p[DPL], g[DPL], im[DPL], t[DPL];
  bool valid[DPL];
#pragma unroll
  for (int j = 0; j < DPL; j++) {
    const int i = l + j * G;
    valid[j] = i < d;
    const size_t o = (size_t)chain * d + i;
    q[j] = valid[j] ? P.q[o] : 0.0;
    p[j] = valid[j] ? P.p[o] : 0.0;
    im[j] = valid[j] ? P.inv_mass[i] : 1.0;
  }
  // normal.ex:18-22: safe_sigma, log(2 pi) + 2 log(safe_sigma)
  const double mu = P.mu;
  const double ss = fmax(P.sigma, P.tiny32);
  const double log_term = P.log2pi32 + 2.0 * exmc_log(ss);
  const double eps = P.eps;
  const double h = eps / 2.0;
  // density and gradient at the lane's coordinates (normal.ex:20-23 and its reverse mode)
  auto density = [&]() {
#pragma unroll
    for (int j = 0; j < DPL; j++) {
      const double z = (q[j] - mu) / ss;
      t[j] = -0.5 * (z * z + log_term);
      g[j] = (-z) / ss;
    }
  };
  density();
  for (int s = 0; s < P.k; s++) {
#pragma unroll
    for (int j = 0; j < DPL; j++) {
      p[j] = p[j] + h * g[j];
      q[j] = q[j] + eps * (im[j] * p[j]);
    }
    density();
    const double logp = group_sum_slots<G, DPL>(t, valid, l, 0.0);
    const size_t row = ((size_t)chain * P.k + s) * d;
#pragma unroll
    for (int j = 0; j < DPL; j++) {
      p[j] = p[j] + h * g[j];
      if (valid[j]) {
        const size_t o = row + l + j * G;
        P.q_chain[o] = q[j];
        P.p_chain[o] = p[j];
        P.g_chain[o] = g[j];
      }
    }
    if (l == 0) P.logp_chain[(size_t)chain * P.k + s] = logp;
  }
}
#endif

}  // namespace exmc
