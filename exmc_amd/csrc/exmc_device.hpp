// exmc_device.hpp — device-side building blocks for the gfx950 NUTS kernels.
//
// Execution model: one chain is owned by a group of G consecutive lanes of a 64-wide
// wavefront (G in {1,2,4,...,64}); lane l of the group owns dimensions l, l+G, l+2G, ...
// (DPL = ceil(D/G) register slots). Per-chain reductions (kinetic energy, the U-turn dot
// products of tree.ex:1578-1588, model sums) are lane-partial sums followed by an
// xor-butterfly over the group, so every lane of the group ends with the same bits.
// Control flow is uniform inside a group and may diverge between groups.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/exmc_detmath.h"

namespace exmc {

constexpr uint64_t kMask58 = (1ULL << 58) - 1;

template <int G>
__device__ __forceinline__ double group_allsum(double v) {
#pragma unroll
  for (int m = 1; m < G; m <<= 1) v = v + __shfl_xor(v, m, 64);
  return v;
}

// value held by lane `src` (0..G-1) of this lane's group
template <int G>
__device__ __forceinline__ double group_bcast(double v, int src) {
  if (G == 1) return v;
  const int lane = threadIdx.x & 63;
  return __shfl(v, (lane & ~(G - 1)) | src, 64);
}

// ---- OTP :rand exsss (Xorshift116**, 58-bit words); call sites sampler.ex:154,343,396,836,897,
// tree.ex:403,1397,1489. Restated from the published algorithm (parity with a BEAM unpinned).
struct Rng {
  uint64_t a, b;  // OTP state [a|b]
};

__device__ __forceinline__ uint64_t rotl58(uint64_t x, int n) {
  return ((x << n) & kMask58) | (x >> (58 - n));
}

__device__ __forceinline__ uint64_t rng_next(Rng& r) {
  const uint64_t s1 = r.a, s0 = r.b;
  const uint64_t v1 = (s0 + ((s0 << 2) & kMask58)) & kMask58;
  const uint64_t v2 = rotl58(v1, 7);
  const uint64_t out = (v2 + ((v2 << 3) & kMask58)) & kMask58;
  const uint64_t s1b = s1 ^ ((s1 << 24) & kMask58);
  r.a = s0;
  r.b = s1b ^ s0 ^ (s1b >> 11) ^ (s0 >> 41);
  return out;
}

__device__ __forceinline__ double rng_uniform(Rng& r) {
  return (double)(rng_next(r) >> 5) * 0x1p-53;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t& x) {
  uint64_t z = (x += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

__device__ __forceinline__ void rng_seed(Rng& r, uint64_t seed) {
  uint64_t x = seed, w[2];
  for (int i = 0; i < 2; i++) {
    uint64_t z;
    do {
      z = splitmix64(x) & kMask58;
    } while (z == 0);
    w[i] = z;
  }
  r.a = w[0];
  r.b = w[1];
}

struct ZigTables {
  const uint64_t* ki;
  const double* wi;
  const double* fi;
};

// :rand.normal_s — 256-layer ziggurat on one 58-bit word (bit 6 sign, bits 7..57 the
// 51-bit R whose low 8 bits select the layer).
__device__ inline double rng_normal(Rng& r, const ZigTables& zt, double nor_r) {
  for (;;) {
    const uint64_t w = rng_next(r);
    const int sign = (int)((w >> 6) & 1);
    const uint64_t R = w >> 7;
    const int idx = (int)(R & 255);
    double x = (double)R * zt.wi[idx];
    if (R < zt.ki[idx]) return sign ? -x : x;
    if (sign) x = -x;
    if (idx == 0) {
      for (;;) {
        const double u0 = rng_uniform(r);
        const double xt = (-(1.0 / nor_r)) * exmc_log(u0);
        const double u1 = rng_uniform(r);
        const double y = -exmc_log(u1);
        if (y + y > xt * xt) return sign ? (-nor_r - xt) : (nor_r + xt);
      }
    }
    const double fi2 = zt.fi[idx];
    const double u0 = rng_uniform(r);
    if ((zt.fi[idx - 1] - fi2) * u0 + fi2 < exmc_exp(-0.5 * x * x)) return x;
  }
}

// tree.ex:1597-1605
__device__ __forceinline__ double log_sum_exp(double a, double b) {
  const double mx = (a > b) ? a : b;
  if (mx == -exmc_from_bits(EXMC_INF_BITS) || mx == -1.0e300) return -1.0e300;
  return mx + exmc_log(exmc_exp(a - mx) + exmc_exp(b - mx));
}

// lane-partial sum over this lane's valid slots, lane 0 seeded with init0, then butterfly
template <int G, int DPL>
__device__ __forceinline__ double group_sum_slots(const double (&v)[DPL], const bool (&valid)[DPL],
                                                  int l, double init0) {
  double acc = (l == 0) ? init0 : 0.0;
#pragma unroll
  for (int k = 0; k < DPL; k++) acc = valid[k] ? (acc + v[k]) : acc;
  return group_allsum<G>(acc);
}

// leapfrog.ex:39-42: 0.5 * sum(p * (M^-1 * p))
template <int G, int DPL>
__device__ __forceinline__ double kinetic_energy(const double (&p)[DPL], const double (&im)[DPL],
                                                 const bool (&valid)[DPL]) {
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < DPL; k++) acc = valid[k] ? (acc + p[k] * (im[k] * p[k])) : acc;
  return 0.5 * group_allsum<G>(acc);
}

// tree.ex:1578-1588: rho-based U-turn test against two endpoint momenta
template <int G, int DPL>
__device__ __forceinline__ bool uturn(const double (&rho)[DPL], const double (&pa)[DPL],
                                      const double (&pb)[DPL], const double (&im)[DPL],
                                      const bool (&valid)[DPL]) {
  double sa = 0.0, sb = 0.0;
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const double v = rho[k] * im[k];
    sa = valid[k] ? (sa + v * pa[k]) : sa;
    sb = valid[k] ? (sb + v * pb[k]) : sb;
  }
  sa = group_allsum<G>(sa);
  sb = group_allsum<G>(sb);
  return (sa < 0.0) || (sb < 0.0);
}

}  // namespace exmc
