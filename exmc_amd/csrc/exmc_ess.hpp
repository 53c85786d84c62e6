// exmc_ess.hpp -- Diagnostics.ess (lib/exmc/diagnostics.ex:42-52, 123-167) of one strided series.
//
// The reference computes every lag of the autocorrelation (O(S^2) per series) and then keeps only
// the lags in front of Geyer's first non-positive pair (diagnostics.ex:147-167). Here the lags are
// computed sixteen at a time, one sweep over the series per block of sixteen, and the sweeps stop at
// that pair. The values that are used are the same bits: each lag sum runs left to right over i as
// `acc + c[i] * c[i + lag]` (diagnostics.ex:137-141), the centred values are `x - mean` with
// mean = (left-to-right sum) / S, and lag 0 of the first block is the variance.
// A sweep for lags [l0, l0 + 16) walks j = i + lag from l0 up, reading c[j] and c[j - l0]; the last
// sixteen values of the second stream sit in a register ring (ring[(j - l0) & 15], static indices
// after unrolling), so lag l0 + k multiplies c[j] with ring[(u - k) & 15]. Typical chains stop
// inside the first block (Geyer's cut at a few lags), so a wavefront whose 64 series all do costs
// two sweeps: the mean and one block.
// On the device the series that are not finished after the first block (a percent or so: pairs of
// pure noise stay positive with probability 1/2 each, so the longest run over 40 000 series is
// ~15 pairs) are handed to a second kernel instead of holding their wavefront: ess_series(...,
// max_blocks = 1, &partial) returns with `finished = false` and the state after lag 15.
// Plain C++ (no HIP types): the device kernel calls it per lane, and tests/test_ess_series_host.py
// compiles it for the host to check it against the CPU checker. Build with -ffp-contract=off.
#pragma once

#include <stddef.h>

#if defined(__HIPCC__)
#define EXMC_ESS_HD __host__ __device__ __forceinline__
#else
#define EXMC_ESS_HD static inline
#endif

namespace exmc {

constexpr int kEssLags = 16;        // lags per sweep (the register ring)
constexpr int kEssLoadBlock = 32;   // loads in flight per stream (a multiple of kEssLags)

// where a series stands after max_blocks blocks of lags without having met Geyer's cut
struct EssPartial {
  double mean, var, tau;
  int next_lag;     // first lag not yet summed
  bool finished;    // the returned value is the ESS
};

EXMC_ESS_HD double ess_series(const double* x, size_t stride, int S, int max_blocks = 1 << 30,
                              EssPartial* part = nullptr) {
  if (part) part->finished = true;
  if (S < 4) return S * 1.0;   // diagnostics.ex:46
  double sum = 0.0;
#pragma unroll 32
  for (int i = 0; i < S; i++) sum += x[(size_t)i * stride];
  const double mean = sum / S;
  const int max_k = (S - 1) / 2;
  double tau = -1.0, var = 0.0;
  bool done = false;
  for (int l0 = 0; !done; l0 += kEssLags) {
    if (l0 / kEssLags >= max_blocks) {   // the caller continues from here (ess_tail_kernel)
      part->mean = mean;
      part->var = var;
      part->tau = tau;
      part->next_lag = l0;
      part->finished = false;
      return 0.0;
    }
    double acc[kEssLags], ring[kEssLags];
#pragma unroll
    for (int k = 0; k < kEssLags; k++) acc[k] = ring[k] = 0.0;
    // ring entries that are not filled yet stand for indices i < 0: they are zeros and add 0.0 to
    // their lag sum, which leaves it unchanged
    for (int j0 = l0; j0 < S; j0 += kEssLoadBlock) {
      // all loads of a block first (clamped, so they need no guard and their latencies overlap:
      // a lane walks its series with a stride of D * C doubles, every load is a cache miss), then
      // the arithmetic
      double xa[kEssLoadBlock], xb[kEssLoadBlock];
#pragma unroll
      for (int u = 0; u < kEssLoadBlock; u++) {
        const int j = (j0 + u < S) ? (j0 + u) : (S - 1);
        xa[u] = x[(size_t)j * stride];
      }
#pragma unroll
      for (int u = 0; u < kEssLoadBlock; u++) {
        const int j = (j0 + u < S) ? (j0 + u) : (S - 1);
        xb[u] = (l0 != 0) ? x[(size_t)(j - l0) * stride] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < kEssLoadBlock; u++) {
        if (j0 + u < S) {
          const double cj = xa[u] - mean;
          ring[u & (kEssLags - 1)] = (l0 == 0) ? cj : (xb[u] - mean);
#pragma unroll
          for (int k = 0; k < kEssLags; k++) acc[k] = acc[k] + ring[(u - k) & (kEssLags - 1)] * cj;
        }
      }
    }
    if (l0 == 0) {
      var = acc[0];
      if (var == 0.0) break;   // diagnostics.ex:130-131: all-zero ACF, tau stays -1
    }
#pragma unroll
    for (int kk = 0; kk < kEssLags / 2; kk++) {
      const int k = l0 / 2 + kk;
      if (!done) {
        if (k > max_k) {
          done = true;
        } else {
          // lags past S - 1 have empty sums: 0.0 / var = 0.0 = Enum.at(acf, lag, 0.0)
          const double pair = acc[2 * kk] / var + acc[2 * kk + 1] / var;
          if (pair > 0) tau += 2 * pair;
          else done = true;
        }
      }
    }
  }
  const double t = tau > 1.0 ? tau : 1.0;   // max(tau, 1.0), diagnostics.ex:165
  return S / t;
}

}  // namespace exmc
