// exmc_device.hpp — device-side building blocks for the gfx950 NUTS kernels.
//
// Execution model: one chain is owned by a group of G consecutive lanes of a 64-wide
// wavefront (G in {1,2,4,...,64}); lane l of the group owns dimensions l, l+G, l+2G, ...
// (DPL = ceil(D/G) register slots). Per-chain reductions (kinetic energy, the U-turn dot
// products of tree.ex:1578-1588, model sums) are lane-partial sums followed by an
// xor-butterfly over the group, so every lane of the group ends with the same bits.
// Control flow is uniform inside a group and may diverge between groups.
//
// The butterfly runs on DPP (quad_perm, row_half_mirror, row_mirror: VALU-speed cross-lane
// moves inside a 16-lane row) and, above 16 lanes, on v_readlane: after the row stages every
// lane of a row holds the row sum, so mirror moves deliver exactly the operands the xor pattern
// would, and the result is bit-identical to `v[l] + v[l ^ m]` for m = 1, 2, 4, ...
// Also here: group broadcasts and rotates on DPP / v_readlane, lane-batched evaluation of
// chain-scalar functions, the division policy (IEEE quotients without the range instructions),
// the two spellings of exp / log, OTP's exsss generator and its ziggurat normal.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include <utility>

#include "../../include/exmc_detmath.h"

namespace exmc {

constexpr uint64_t kMask58 = (1ULL << 58) - 1;

// ---- IEEE f64 division without the range instructions (exmc_detmath.h: exmc_div_core) ----
// The short sequence is bit-identical to the compiler's full f64 division when both operands
// have magnitudes in [2^-380, 2^380): neither operand nor quotient is then zero, denormal or
// huge, and the exponent difference stays below v_div_scale's 768 threshold. A model evaluates
// its quotients through a Div<kFast> policy: with kFast the quotients are exmc_div_core and the
// policy records whether every operand the model declared with `watch` was in range; the caller
// re-evaluates with Div<false> (the ordinary `/`) when any lane of the wavefront saw an operand
// outside it (zeros, infinities, NaN, extreme magnitudes: the first leapfrog from an all-zero
// init, a diverging trajectory). Operands that are not watched carry a range argument at the
// call site. A Recip keeps the refined reciprocal of a reused divisor, so a quotient by a model
// constant costs 3 VALU operations instead of ~25 issue slots.
__device__ __forceinline__ bool div_in_range(double x) {
  const uint32_t h = (uint32_t)__double2hiint(x) & 0x7fffffffu;
  return (h - 0x28300000u) < (0x57C00000u - 0x28300000u);   // biased exponent in [643, 1404)
}

struct Recip {
  double b, r;
};

__device__ __forceinline__ Recip make_recip(double b) { return Recip{b, exmc_rcp_refined(b)}; }

// Recip of a literal: the value is hidden from the optimiser so that the reciprocal comes from
// v_rcp_f64 and the Newton steps at run time, like every other divisor (LLVM would otherwise fold
// rcp(constant) to a correctly rounded 1/b, a different starting point than the hardware's).
__device__ __forceinline__ Recip make_recip_literal(double b) {
  __asm__ volatile("" : "+v"(b));
  return make_recip(b);
}

template <bool kFast>
struct Div {
  bool ok = true;
  __device__ __forceinline__ void watch(double x) {
    if (kFast) ok = ok && div_in_range(x);
  }
  __device__ __forceinline__ void watch_if(bool relevant, double x) {
    if (kFast) ok = ok && (!relevant || div_in_range(x));
  }
  // |x| in [2^ELO, 2^EHI): a tighter declaration, for operands whose products or quotients are
  // used as operands again and must themselves stay inside the 2^+-380 window
  template <int ELO, int EHI>
  __device__ __forceinline__ void watch_exp_if(bool relevant, double x) {
    static_assert(ELO >= -380 && EHI <= 380 && ELO < EHI, "inside the division window");
    if (kFast) {
      const uint32_t h = (uint32_t)__double2hiint(x) & 0x7fffffffu;
      const bool in = (h - ((uint32_t)(1023 + ELO) << 20)) < ((uint32_t)(EHI - ELO) << 20);
      ok = ok && (!relevant || in);
    }
  }
  __device__ __forceinline__ double operator()(double a, const Recip& c) const {
    return kFast ? exmc_div_core(a, c.b, c.r) : (a / c.b);
  }
  __device__ __forceinline__ double operator()(double a, double b) const {
    return kFast ? exmc_div_core(a, b, exmc_rcp_refined(b)) : (a / b);
  }
};

// evaluate f(Div<true>&); if any lane of the wavefront left the fast range, evaluate f(Div<false>&)
template <class F>
__device__ __forceinline__ double with_fast_div(F&& f) {
  Div<true> fast;
  double r = f(fast);
  if (__builtin_expect(__any(fast.ok ? 0 : 1) != 0, 0)) {
    Div<false> exact;
    r = f(exact);
  }
  return r;
}

// DPP controls (gfx9 encoding)
constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // lane i <- lane 7-i within 8
constexpr int kDppRowMirror = 0x140;  // lane i <- lane 15-i within 16

template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// The sums of the 64 / G lane groups of a wavefront, added in group order (group 0 first); every
// value is uniform inside its group (after the group's butterfly), so a group's value is a
// v_readlane of its first lane. Every lane of the wavefront ends with the same totals. For a chain
// whose model terms are spread over all groups of the wavefront (the one-chain warmup of generated
// lane layouts, exmc_models.hpp CustomSplit).
template <int G, int N>
__device__ __forceinline__ void xgroup_sum_n(double (&s)[N]) {
  static_assert(G == 16 || G == 32, "two or four lane groups per wavefront");
#pragma unroll
  for (int k = 0; k < N; k++) {
    double t = readlane_f64(s[k], 0);
#pragma unroll
    for (int j = 1; j < 64 / G; j++) t = t + readlane_f64(s[k], j * G);
    s[k] = t;
  }
}

// ---- cross-row exchanges without the LDS crossbar (gfx950) ----
// v_permlane16_swap_b32 / v_permlane32_swap_b32 (CDNA4) swap the odd 16-lane rows of one register
// with the even rows of another / the upper half of one with the lower half of another, in the
// vector ALU. Fed two copies of a value v they leave (a, b) with a + b = v[l] + v[l ^ 16] (resp.
// l ^ 32) in EVERY lane -- in half the lanes as own + partner, in the other half as partner + own,
// which are the same bits: the butterfly stage of group_allsum_n and of the reduce-scatter below
// without the round trip through LDS (two swaps, one move and the add per stage and value, against
// two ds_bpermute and the add; a tenth of the latency, which is what a wave that has the issue
// priority -- or its SIMD to itself -- waits for). EXMC_XROW_PERMLANE=0 builds the LDS forms.
#ifndef EXMC_XROW_PERMLANE
#define EXMC_XROW_PERMLANE 1
#endif
__device__ __forceinline__ double sum_xor16(double v) {
#if EXMC_XROW_PERMLANE
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
#else
  const int a = (((int)threadIdx.x & 63) ^ 16) << 2;
  return v + __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v)),
                              __builtin_amdgcn_ds_bpermute(a, __double2loint(v)));
#endif
}
__device__ __forceinline__ double sum_xor32(double v) {
#if EXMC_XROW_PERMLANE
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
#else
  const int a = (((int)threadIdx.x & 63) ^ 32) << 2;
  return v + __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v)),
                              __builtin_amdgcn_ds_bpermute(a, __double2loint(v)));
#endif
}
// lane l <- lane l ^ 4: two masked DPP moves per dword (row_shr:4 into the banks whose lanes have
// bit 2 set, row_shl:4 into the others) instead of a ds_swizzle
__device__ __forceinline__ int xor4_b32(int x) {
  int t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xA, false);   // row_shr:4 -> lanes 4-7, 12-15
  return __builtin_amdgcn_update_dpp(t, x, 0x104, 0xF, 0x5, false);    // row_shl:4 -> lanes 0-3, 8-11
}

// N independent all-reduce sums over the G-lane group, stage by stage (the N moves of a stage
// are independent, so their latencies overlap). kLds (G = 64): the two cross-row stages go
// through the LDS crossbar (ds_bpermute) -- two vector instructions per value instead of the
// eleven of the v_readlane form, for kernels that are bound by vector issue and have a second
// wave on the SIMD to cover the LDS round trips (a lone wave would sit them out).
template <int N>
__device__ __forceinline__ void rs64_allsum_generic(double (&v)[N]);   // below, behind row16_reduce_scatter

template <int G, int N, bool kLds = false>
__device__ __forceinline__ void group_allsum_butterfly_n(double (&v)[N]);

template <int G, int N, bool kLds = false>
__device__ __forceinline__ void group_allsum_n(double (&v)[N]) {
#if EXMC_XROW_PERMLANE
  if constexpr (G == 64 && N >= 3 && N <= 32) {
    // a whole wavefront, several sums: the reduce-scatter computes the butterfly's own totals (same
    // bits) in about half the instructions and hands them back as scalars (round 5; the generated
    // 64-lane layouts' six to eight sums, the dense lane layouts)
    rs64_allsum_generic<N>(v);
    return;
  }
#endif
  group_allsum_butterfly_n<G, N, kLds>(v);
}

template <int G, int N, bool kLds>
__device__ __forceinline__ void group_allsum_butterfly_n(double (&v)[N]) {
  static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "G");
  if (G >= 2) {
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = v[j] + dpp_move<kDppXor1>(v[j]);
  }
  if (G >= 4) {
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = v[j] + dpp_move<kDppXor2>(v[j]);
  }
  if (G >= 8) {
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = v[j] + dpp_move<kDppHalfMirror>(v[j]);
  }
  if (G >= 16) {
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = v[j] + dpp_move<kDppRowMirror>(v[j]);
  }
#if EXMC_XROW_PERMLANE
  // the cross-row stages in the vector ALU (sum_xor16 / sum_xor32 above), whatever kLds says: rows
  // (0, 1) and (2, 3) first, then the halves -- (r0 + r1) + (r2 + r3) in every lane, the expression
  // the v_readlane form and the LDS form both compute
  if (G >= 32) {
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = sum_xor16(v[j]);
  }
  if (G == 64) {
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = sum_xor32(v[j]);
  }
#else
  if constexpr (G == 64 && kLds) {
    // v + v[l ^ 16], then + v[l ^ 32]: the same butterfly
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = sum_xor16(v[j]);
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = sum_xor32(v[j]);
    return;
  }
  if (G >= 32) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const double r0 = readlane_f64(v[j], 0), r1 = readlane_f64(v[j], 16);
      const double r2 = readlane_f64(v[j], 32), r3 = readlane_f64(v[j], 48);
      const double h0 = r0 + r1, h1 = r2 + r3;   // == row + (row ^ 1) for every row
      if (G == 32) v[j] = (lane < 32) ? h0 : h1;
      else v[j] = h0 + h1;                        // == half + (half ^ 1)
    }
  }
#endif
}

template <int G, bool kLds = false>
__device__ __forceinline__ double group_allsum(double v) {
  double a[1] = {v};
  group_allsum_n<G, 1, kLds>(a);
  return a[0];
}

// ---- sums over a 16-lane chain group in lane order (v_fmac_f64_dpp row_newbcast) ---------------
// The f64 ALU of gfx950 takes one DPP control, row_newbcast:i (operand 0 is read from lane i of the
// row), and v_fmac_f64 is the one f64 arithmetic instruction with a DPP form. So
//     acc = fma(v[lane i], 1.0, acc)   for i = 0 .. NL-1
// is ONE instruction per term, leaves the sum in every lane of the row, and adds the terms left
// to right -- the reference's own order (Enum.sum / Nx.sum over the flat vector), where the xor
// butterfly costs 3 instructions per stage (two 32-bit DPP moves and the add) plus the selects
// that mask the lanes past D. It wins while NL <= 12: chain groups of 16 lanes that hold one
// dimension per lane with D <= 12 (eight_schools, generated models) sum this way, seeded with the
// caller's start value; every other layout keeps the butterfly. ("Reductions": for
// such a group the lane order IS the left-to-right sum of the numeric contract.)
// One asm block per batch of sums: a block is not padded with hazard nops per instruction, and the
// leading s_nop covers a DPP read of a register the preceding VALU instruction wrote.
template <int G, int D>
inline constexpr bool kSeqSum = (G == 16 && D <= 12);

#define EXMC_FMAC_B(A, V, I) \
  "v_fmac_f64_dpp %[" #A "], %[" #V "], %[one] row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
#define EXMC_SEQ1(I) EXMC_FMAC_B(a0, v0, I)
#define EXMC_SEQ2(I) EXMC_SEQ1(I) EXMC_FMAC_B(a1, v1, I)
#define EXMC_SEQ3(I) EXMC_SEQ2(I) EXMC_FMAC_B(a2, v2, I)
#define EXMC_SEQ6(I) EXMC_SEQ3(I) EXMC_FMAC_B(a3, v3, I) EXMC_FMAC_B(a4, v4, I) EXMC_FMAC_B(a5, v5, I)
#define EXMC_LANES_1(S) S(0)
#define EXMC_LANES_2(S) EXMC_LANES_1(S) S(1)
#define EXMC_LANES_3(S) EXMC_LANES_2(S) S(2)
#define EXMC_LANES_4(S) EXMC_LANES_3(S) S(3)
#define EXMC_LANES_5(S) EXMC_LANES_4(S) S(4)
#define EXMC_LANES_6(S) EXMC_LANES_5(S) S(5)
#define EXMC_LANES_7(S) EXMC_LANES_6(S) S(6)
#define EXMC_LANES_8(S) EXMC_LANES_7(S) S(7)
#define EXMC_LANES_9(S) EXMC_LANES_8(S) S(8)
#define EXMC_LANES_10(S) EXMC_LANES_9(S) S(9)
#define EXMC_LANES_11(S) EXMC_LANES_10(S) S(10)
#define EXMC_LANES_12(S) EXMC_LANES_11(S) S(11)
// the asm statement for NL lanes, chosen at compile time
#define EXMC_SEQSUM_ASM(SEQ, OUTS, INS)                                                        \
  if constexpr (NL == 1) __asm__("s_nop 1\n\t" EXMC_LANES_1(SEQ) : OUTS : INS);                \
  else if constexpr (NL == 2) __asm__("s_nop 1\n\t" EXMC_LANES_2(SEQ) : OUTS : INS);           \
  else if constexpr (NL == 3) __asm__("s_nop 1\n\t" EXMC_LANES_3(SEQ) : OUTS : INS);           \
  else if constexpr (NL == 4) __asm__("s_nop 1\n\t" EXMC_LANES_4(SEQ) : OUTS : INS);           \
  else if constexpr (NL == 5) __asm__("s_nop 1\n\t" EXMC_LANES_5(SEQ) : OUTS : INS);           \
  else if constexpr (NL == 6) __asm__("s_nop 1\n\t" EXMC_LANES_6(SEQ) : OUTS : INS);           \
  else if constexpr (NL == 7) __asm__("s_nop 1\n\t" EXMC_LANES_7(SEQ) : OUTS : INS);           \
  else if constexpr (NL == 8) __asm__("s_nop 1\n\t" EXMC_LANES_8(SEQ) : OUTS : INS);           \
  else if constexpr (NL == 9) __asm__("s_nop 1\n\t" EXMC_LANES_9(SEQ) : OUTS : INS);           \
  else if constexpr (NL == 10) __asm__("s_nop 1\n\t" EXMC_LANES_10(SEQ) : OUTS : INS);         \
  else if constexpr (NL == 11) __asm__("s_nop 1\n\t" EXMC_LANES_11(SEQ) : OUTS : INS);         \
  else __asm__("s_nop 1\n\t" EXMC_LANES_12(SEQ) : OUTS : INS);
#define EXMC_COMMA ,

// 1.0 as a vector-register operand (the DPP form of v_fmac_f64 takes no literal): a plain constant,
// so the compiler keeps ONE copy live across the sampling loop instead of a v_mov_b64 per sum
__device__ __forceinline__ double seq_one() { return 1.0; }

// a_j <- a_j + v_j[lane 0] + v_j[lane 1] + ... + v_j[lane NL-1] (left to right), in every lane
template <int NL>
__device__ __forceinline__ void row_seqsum(double& a0, double v0) {
  static_assert(NL >= 1 && NL <= 12, "lanes of one DPP row, where the chain beats the butterfly");
  const double one = seq_one();
  EXMC_SEQSUM_ASM(EXMC_SEQ1, [a0] "+v"(a0), [v0] "v"(v0) EXMC_COMMA [one] "v"(one))
}
template <int NL>
__device__ __forceinline__ void row_seqsum(double& a0, double& a1, double v0, double v1) {
  const double one = seq_one();
  EXMC_SEQSUM_ASM(EXMC_SEQ2, [a0] "+v"(a0) EXMC_COMMA [a1] "+v"(a1),
                  [v0] "v"(v0) EXMC_COMMA [v1] "v"(v1) EXMC_COMMA [one] "v"(one))
}
template <int NL>
__device__ __forceinline__ void row_seqsum(double& a0, double& a1, double& a2, double v0, double v1,
                                           double v2) {
  const double one = seq_one();
  EXMC_SEQSUM_ASM(EXMC_SEQ3, [a0] "+v"(a0) EXMC_COMMA [a1] "+v"(a1) EXMC_COMMA [a2] "+v"(a2),
                  [v0] "v"(v0) EXMC_COMMA [v1] "v"(v1) EXMC_COMMA [v2] "v"(v2) EXMC_COMMA [one] "v"(one))
}
// ---- a dense D x D matrix against a vector held one entry per lane (row layout of a 16-lane
// chain group): lane i keeps row i of the matrix in registers and
//     acc = fma(x[lane j], c[j], acc)   for j = 0 .. NL-1, from 0.0
// is the i-th entry of the product with the fma chain in ascending j (dense_times' order) ----
#define EXMC_MV1(I) \
  "v_fmac_f64_dpp %[acc], %[x], %[c" #I "] row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
template <int NL>
__device__ __forceinline__ double row_matvec(double x, const double (&c)[NL]) {
  static_assert(NL >= 1 && NL <= 12, "lanes of one DPP row");
  double acc = 0.0;
#define EXMC_MV_C(I) [c##I] "v"(c[(I) < NL ? (I) : 0])
  EXMC_SEQSUM_ASM(EXMC_MV1, [acc] "+v"(acc),
                  [x] "v"(x) EXMC_COMMA EXMC_MV_C(0) EXMC_COMMA EXMC_MV_C(1) EXMC_COMMA EXMC_MV_C(2) EXMC_COMMA
                  EXMC_MV_C(3) EXMC_COMMA EXMC_MV_C(4) EXMC_COMMA EXMC_MV_C(5) EXMC_COMMA EXMC_MV_C(6) EXMC_COMMA
                  EXMC_MV_C(7) EXMC_COMMA EXMC_MV_C(8) EXMC_COMMA EXMC_MV_C(9) EXMC_COMMA EXMC_MV_C(10) EXMC_COMMA
                  EXMC_MV_C(11))
#undef EXMC_MV_C
  return acc;
}
// m[j] = fma(x[lane j], c, m[j]) for j = 0 .. NL-1: lane i's row of the rank-one update c x^T
#define EXMC_OUTER1(I) \
  "v_fmac_f64_dpp %[m" #I "], %[x], %[c] row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
template <int NL>
__device__ __forceinline__ void row_outer_acc(double (&m)[NL], double x, double c) {
  static_assert(NL >= 1 && NL <= 12, "lanes of one DPP row");
  double pad[12];
#pragma unroll
  for (int j = 0; j < 12; j++) pad[j] = (j < NL) ? m[j < NL ? j : 0] : 0.0;
#define EXMC_OUTER_M(I) [m##I] "+v"(pad[I])
  EXMC_SEQSUM_ASM(EXMC_OUTER1,
                  EXMC_OUTER_M(0) EXMC_COMMA EXMC_OUTER_M(1) EXMC_COMMA EXMC_OUTER_M(2) EXMC_COMMA EXMC_OUTER_M(3) EXMC_COMMA
                  EXMC_OUTER_M(4) EXMC_COMMA EXMC_OUTER_M(5) EXMC_COMMA EXMC_OUTER_M(6) EXMC_COMMA EXMC_OUTER_M(7) EXMC_COMMA
                  EXMC_OUTER_M(8) EXMC_COMMA EXMC_OUTER_M(9) EXMC_COMMA EXMC_OUTER_M(10) EXMC_COMMA EXMC_OUTER_M(11),
                  [x] "v"(x) EXMC_COMMA [c] "v"(c))
#undef EXMC_OUTER_M
#pragma unroll
  for (int j = 0; j < NL; j++) m[j] = pad[j];
}
// x[lane J] in every lane of the row (one fmac from 0.0: fma(x_J, 1, 0) = x_J)
template <int J>
__device__ __forceinline__ double row_bcast_f64(double x) {
  double acc = 0.0;
  const double one = seq_one();
  __asm__("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
          : "+v"(acc) : "v"(x), "v"(one), "n"(J));
  return acc;
}

template <int NL>
__device__ __forceinline__ void row_seqsum6(double (&a)[6], const double (&v)[6]) {
  const double one = seq_one();
  EXMC_SEQSUM_ASM(EXMC_SEQ6,
                  [a0] "+v"(a[0]) EXMC_COMMA [a1] "+v"(a[1]) EXMC_COMMA [a2] "+v"(a[2]) EXMC_COMMA
                  [a3] "+v"(a[3]) EXMC_COMMA [a4] "+v"(a[4]) EXMC_COMMA [a5] "+v"(a[5]),
                  [v0] "v"(v[0]) EXMC_COMMA [v1] "v"(v[1]) EXMC_COMMA [v2] "v"(v[2]) EXMC_COMMA
                  [v3] "v"(v[3]) EXMC_COMMA [v4] "v"(v[4]) EXMC_COMMA [v5] "v"(v[5]) EXMC_COMMA
                  [one] "v"(one))
}

// value held by lane `src` (0..G-1) of this lane's group
template <int G>
__device__ __forceinline__ double group_bcast(double v, int src) {
  if (G == 1) return v;
  const int lane = threadIdx.x & 63;
  return __shfl(v, (lane & ~(G - 1)) | src, 64);
}

// the same for a compile-time source lane, without the LDS round trip of ds_bpermute wherever
// the lane group maps onto a DPP pattern: quad_perm for G <= 4, row_newbcast for G = 16 (a group
// is one DPP row), v_readlane for G >= 32.
template <int G, int SRC>
__device__ __forceinline__ double group_bcast_c(double v) {
  static_assert(SRC >= 0 && SRC < G, "source lane outside the group");
  if constexpr (G == 1) {
    return v;
  } else if constexpr (G == 2) {
    return dpp_move<SRC | (SRC << 2) | ((2 + SRC) << 4) | ((2 + SRC) << 6)>(v);
  } else if constexpr (G == 4) {
    return dpp_move<SRC | (SRC << 2) | (SRC << 4) | (SRC << 6)>(v);
  } else if constexpr (G == 16) {
    return dpp_move<0x150 + SRC>(v);   // row_newbcast:SRC
  } else if constexpr (G == 32) {
    const double lo = readlane_f64(v, SRC), hi = readlane_f64(v, 32 + SRC);
    return ((threadIdx.x & 63) < 32) ? lo : hi;
  } else if constexpr (G == 64) {
    return readlane_f64(v, SRC);
  } else {
    return group_bcast<G>(v, SRC);
  }
}

// the value held by the previous / next lane of the chain group, wrapping inside the group:
// DPP row rotates for G = 16, wavefront rotates for G = 64, ds_bpermute otherwise
template <int G>
__device__ __forceinline__ double group_rot_prev(double v) {   // lane l <- lane (l - 1) mod G
  if constexpr (G == 64) return dpp_move<0x13C>(v);             // wave_ror:1
  else if constexpr (G == 16) return dpp_move<0x121>(v);        // row_ror:1
  else {
    const int lane = threadIdx.x & 63;
    return __shfl(v, (lane & ~(G - 1)) | ((lane + G - 1) & (G - 1)), 64);
  }
}
template <int G>
__device__ __forceinline__ double group_rot_next(double v) {   // lane l <- lane (l + 1) mod G
  if constexpr (G == 64) return dpp_move<0x134>(v);             // wave_rol:1
  else if constexpr (G == 16) return dpp_move<0x12F>(v);        // row_ror:15
  else {
    const int lane = threadIdx.x & 63;
    return __shfl(v, (lane & ~(G - 1)) | ((lane + 1) & (G - 1)), 64);
  }
}

// One evaluation of f for N different group-uniform arguments: lane i of the group evaluates
// x[i] (the other lanes x[0]) and the results are broadcast back. Scalar per-chain work is the same
// instruction stream whatever its argument, so N calls of a 45-instruction log cost one call plus
// the selects and broadcasts; each value is computed by exactly the operations of f (same bits).
template <int G, class F, int... I>
__device__ __forceinline__ void lane_batch_impl(double (&x)[sizeof...(I)], int l, F&& f,
                                                std::integer_sequence<int, I...>) {
  double a = x[0];
  ((a = (l == I) ? x[I] : a), ...);
  const double v = f(a);
  ((x[I] = group_bcast_c<G, I>(v)), ...);
}
template <int G, int N, class F>
__device__ __forceinline__ void lane_batch(double (&x)[N], int l, F&& f) {
  static_assert(N <= G, "one lane per argument");
  lane_batch_impl<G>(x, l, f, std::make_integer_sequence<int, N>{});
}

// ---- OTP :rand exsss (Xorshift116**, 58-bit words); call sites sampler.ex:154,343,396,836,897,
// tree.ex:403,1397,1489. Restated from the published algorithm (parity with a BEAM unpinned).
struct Rng {
  uint64_t a, b;  // OTP state [a|b]
};

__device__ __forceinline__ uint64_t rotl58(uint64_t x, int n) {
  return ((x << n) & kMask58) | (x >> (58 - n));
}

// output word of a step = ** scrambler of the state's tail word; the state update is separate
__device__ __forceinline__ uint64_t rng_scramble(uint64_t s0) {
  const uint64_t v1 = (s0 + ((s0 << 2) & kMask58)) & kMask58;
  const uint64_t v2 = rotl58(v1, 7);
  return (v2 + ((v2 << 3) & kMask58)) & kMask58;
}

__device__ __forceinline__ void rng_advance(Rng& r) {
  const uint64_t s1 = r.a, s0 = r.b;
  const uint64_t s1b = s1 ^ ((s1 << 24) & kMask58);
  r.a = s0;
  r.b = s1b ^ s0 ^ (s1b >> 11) ^ (s0 >> 41);
}

__device__ __forceinline__ uint64_t rng_next(Rng& r) {
  const uint64_t out = rng_scramble(r.b);
  rng_advance(r);
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
// (words: incremented by the number of generator words the draw consumed -- one when the first word is
// accepted at once)
template <bool kCount = false>
__device__ inline double rng_normal_impl(Rng& r0, const ZigTables& zt, double nor_r, int& words) {
  struct Counted {
    Rng& r;
    int& n;
  } cr{r0, words};
  auto next = [&]() -> uint64_t {
    if constexpr (kCount) cr.n++;
    return rng_next(cr.r);
  };
  auto uniform = [&]() -> double { return (double)(next() >> 5) * 0x1p-53; };
  for (;;) {
    const uint64_t w = next();
    const int sign = (int)((w >> 6) & 1);
    const uint64_t R = w >> 7;
    const int idx = (int)(R & 255);
    double x = (double)R * zt.wi[idx];
    if (R < zt.ki[idx]) return sign ? -x : x;
    if (sign) x = -x;
    if (idx == 0) {
      for (;;) {
        const double u0 = uniform();
        const double xt = (-(1.0 / nor_r)) * exmc_log(u0);
        const double u1 = uniform();
        const double y = -exmc_log(u1);
        if (y + y > xt * xt) return sign ? (-nor_r - xt) : (nor_r + xt);
      }
    }
    const double fi2 = zt.fi[idx];
    const double u0 = uniform();
    if ((zt.fi[idx - 1] - fi2) * u0 + fi2 < exmc_exp(-0.5 * x * x)) return x;
  }
}
__device__ inline double rng_normal_counted(Rng& r, const ZigTables& zt, double nor_r, int& words) {
  return rng_normal_impl<true>(r, zt, nor_r, words);
}

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

// The fast-accept branch of normal_s on a word already drawn: returns true and the variate when
// the word is accepted without a second draw (~98.5 % of words).
__device__ __forceinline__ bool normal_fast(uint64_t w, const ZigTables& zt, double& z) {
  const int sign = (int)((w >> 6) & 1);
  const uint64_t R = w >> 7;
  const int idx = (int)(R & 255);
  const double x = (double)R * zt.wi[idx];
  z = sign ? -x : x;
  return R < zt.ki[idx];
}

// Which spelling of exp / log a kernel uses (exmc_detmath.h): kVreg = the asm-block cores with
// coefficients pinned in ~40 vector registers. Same bits either way; models whose chain state
// already fills the 256 architectural VGPRs (DPL >= 2 at G = 64) keep the compiler's spelling.
template <bool kVreg>
struct Math {
  __device__ static __forceinline__ double exp(double x) {
    if constexpr (kVreg) return exmc_exp_v(x);
    else return exmc_exp(x);
  }
  __device__ static __forceinline__ double log(double x) {
    if constexpr (kVreg) return exmc_log_v(x);
    else return exmc_log(x);
  }
  // the same functions where the call site proves the argument's range (exmc_detmath.h): the main
  // path without its special-case branches, v_ldexp_f64 scaling; same bits on the stated domain
  __device__ static __forceinline__ double exp_pm200(double x) {   // |x| <= 200
    if constexpr (kVreg) return exmc_exp_pm200_v(x);
    else return exmc_exp_pm200(x);
  }
  __device__ static __forceinline__ double exp_le0(double x) {     // x <= 0 or NaN
    if constexpr (kVreg) return exmc_exp_le0_v(x);
    else return exmc_exp_le0(x);
  }
  __device__ static __forceinline__ double log_ge1(double x) {     // 1 <= x < inf or NaN
    if constexpr (kVreg) return exmc_log_ge1_v(x);
    else return exmc_log_ge1(x);
  }
  __device__ static __forceinline__ double log_unit(double x) {    // a uniform_s variate, [0, 1)
    if constexpr (kVreg) return exmc_log_unit_v(x);
    else return exmc_log_unit(x);
  }
  // tree.ex:1597-1605. exp(0) = 1 exactly under exmc_exp, so the larger term is not evaluated.
  // mn - mx <= 0 (or NaN), 1 + exp(.) in [1, 2] (or NaN): the range-restricted forms apply.
  __device__ static __forceinline__ double log_sum_exp(double a, double b) {
    const double mx = (a > b) ? a : b;
    if (mx == -exmc_from_bits(EXMC_INF_BITS) || mx == -1.0e300) return -1.0e300;
    const double mn = (a > b) ? b : a;
    return mx + log_ge1(1.0 + exp_le0(mn - mx));
  }
};

// ---- 64-lane sums as a reduce-scatter (one chain per wavefront) ----
// group_allsum_n<64, N> moves every one of its N values through all six butterfly stages: 6 N
// exchanges and adds. The sum tree of a value does not care WHICH lane adds a pair, so the first
// stages can halve the number of values a lane carries instead: at stage xor-m a lane keeps half of
// its values, sends the other half to lane ^ m (which kept those) and adds what it receives to what
// it kept; after log2(N) such stages every lane holds one value, partially reduced over its 2^k-lane
// group, and the remaining stages move that one value. The additions are the butterfly's own --
// v[l] + v[l ^ 1], then (.)[l] + (.)[l ^ 2], ... -- so every total has the bits group_allsum_n gives
// it (tools/probe/allsum_rs_probe.hip); quantity k ends up in the lanes with a known code in their
// low bits. Exchanges: xor 1, 2 by DPP quad_perm, xor 8 by DPP row_ror:8, xor 4 / 16 by ds_swizzle
// (bit mode), xor 32 by ds_bpermute: for kernels with a second wave on the SIMD to cover the LDS
// crossbar's round trips, like the kLds form of group_allsum_n.
template <int CTRL>
__device__ __forceinline__ double xchg_dpp(double v) { return dpp_move<CTRL>(v); }
template <int PATTERN>
__device__ __forceinline__ double xchg_swizzle(double v) {
  return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), PATTERN),
                          __builtin_amdgcn_ds_swizzle(__double2loint(v), PATTERN));
}
__device__ __forceinline__ double xchg_xor4(double v) {   // lane l <- lane l ^ 4
#if EXMC_XROW_PERMLANE
  return __hiloint2double(xor4_b32(__double2hiint(v)), xor4_b32(__double2loint(v)));
#else
  return xchg_swizzle<(4 << 10) | 0x1F>(v);
#endif
}
__device__ __forceinline__ double xchg_xor32(double v) {
  const int a = (((int)threadIdx.x & 63) ^ 32) << 2;
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v)),
                          __builtin_amdgcn_ds_bpermute(a, __double2loint(v)));
}
constexpr int kSwzXor4 = (4 << 10) | 0x1F, kSwzXor16 = (16 << 10) | 0x1F;   // and 0x1f, or 0, xor m
constexpr int kDppRowRor8 = 0x128;

// one value through the stages xor 4 .. xor 32 (FROM = 4) or xor 8 .. xor 32 (FROM = 8)
template <int FROM>
__device__ __forceinline__ double rs64_tail(double z) {
  if (FROM <= 4) z = z + xchg_xor4(z);
  z = z + xchg_dpp<kDppRowRor8>(z);
  z = sum_xor16(z);
  z = sum_xor32(z);
  return z;
}

// two sums: lane 0 (and every lane with bit 0 clear) ends with total 0, lane 1 with total 1
__device__ __forceinline__ double rs64_reduce2(const double (&v)[2]) {
  const bool b0 = (threadIdx.x & 1) != 0;
  const double w = (b0 ? v[1] : v[0]) + xchg_dpp<kDppXor1>(b0 ? v[0] : v[1]);
  return rs64_tail<4>(w + xchg_dpp<kDppXor2>(w));
}
// four sums: total 0 in lane 0, 2 in lane 1, 1 in lane 2, 3 in lane 3 (low two bits)
__device__ __forceinline__ double rs64_reduce4(const double (&v)[4]) {
  const bool b0 = (threadIdx.x & 1) != 0, b1 = (threadIdx.x & 2) != 0;
  const double wa = (b0 ? v[2] : v[0]) + xchg_dpp<kDppXor1>(b0 ? v[0] : v[2]);
  const double wb = (b0 ? v[3] : v[1]) + xchg_dpp<kDppXor1>(b0 ? v[1] : v[3]);
  const double z = (b1 ? wb : wa) + xchg_dpp<kDppXor2>(b1 ? wa : wb);
  return rs64_tail<4>(z);
}
// six sums: totals 0, 3, 2, 5, 1, 4 in lanes 0 .. 5 (1 and 4 again in lanes 6, 7)
__device__ __forceinline__ double rs64_reduce6(const double (&v)[6]) {
  const bool b0 = (threadIdx.x & 1) != 0, b1 = (threadIdx.x & 2) != 0, b2 = (threadIdx.x & 4) != 0;
  const double w0 = (b0 ? v[3] : v[0]) + xchg_dpp<kDppXor1>(b0 ? v[0] : v[3]);
  const double w1 = (b0 ? v[4] : v[1]) + xchg_dpp<kDppXor1>(b0 ? v[1] : v[4]);
  const double w2 = (b0 ? v[5] : v[2]) + xchg_dpp<kDppXor1>(b0 ? v[2] : v[5]);
  const double x = (b1 ? w2 : w0) + xchg_dpp<kDppXor2>(b1 ? w0 : w2);   // quantities 0 / 3 or 2 / 5
  const double y = w1 + xchg_dpp<kDppXor2>(w1);                          // quantities 1 / 4
  const double z = (b2 ? y : x) + xchg_xor4(b2 ? x : y);
  return rs64_tail<8>(z);
}
// the four totals in every lane (they come back as scalars)
__device__ __forceinline__ void rs64_allsum4(double (&v)[4]) {
  const double z = rs64_reduce4(v);
  v[0] = readlane_f64(z, 0);
  v[2] = readlane_f64(z, 1);
  v[1] = readlane_f64(z, 2);
  v[3] = readlane_f64(z, 3);
}

// the six totals in every lane (rs64_reduce6 leaves totals 0, 3, 2, 5, 1, 4 in lanes 0 .. 5)
__device__ __forceinline__ void rs64_allsum6(double (&v)[6]) {
  const double z = rs64_reduce6(v);
  v[0] = readlane_f64(z, 0);
  v[3] = readlane_f64(z, 1);
  v[2] = readlane_f64(z, 2);
  v[5] = readlane_f64(z, 3);
  v[1] = readlane_f64(z, 4);
  v[4] = readlane_f64(z, 5);
}

// N sums over the sixteen lanes of a DPP row as a reduce-scatter (round 5): lane l ends with the totals
// of quantities l, l + 16, ... (out[k] = total of quantity l + 16 k), each total the expression
// ((v0 + v1) + (v2 + v3)) + ... the butterfly of group_allsum_n<16, N> gives it (stage by stage a lane
// keeps the half of its quantities whose destination lane has its own bit, sends the other half to the
// partner and adds what it receives: own + partner, the butterfly's addition on that lane) -- in about
// 7 N instructions instead of 12 N, and with every total already on the lane that uses it. Quantities
// past N are zero padding; their "totals" are never read.
template <int N>
__device__ __forceinline__ void row16_reduce_scatter(const double (&v)[N], double (&out)[(N + 15) / 16]) {
  constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2;
  static_assert(N4 == (N + 15) / 16, "four halvings");
  const int lane = threadIdx.x;
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0, b3 = (lane & 8) != 0;
  double a1[N1], a2[N2], a3[N3];
#pragma unroll
  for (int i = 0; i < N1; i++) {
    const double lo = v[2 * i], hi = (2 * i + 1 < N) ? v[(2 * i + 1 < N) ? 2 * i + 1 : 0] : 0.0;
    a1[i] = (b0 ? hi : lo) + xchg_dpp<kDppXor1>(b0 ? lo : hi);
  }
#pragma unroll
  for (int i = 0; i < N2; i++) {
    const double lo = a1[2 * i], hi = (2 * i + 1 < N1) ? a1[(2 * i + 1 < N1) ? 2 * i + 1 : 0] : 0.0;
    a2[i] = (b1 ? hi : lo) + xchg_dpp<kDppXor2>(b1 ? lo : hi);
  }
#pragma unroll
  for (int i = 0; i < N3; i++) {
    const double lo = a2[2 * i], hi = (2 * i + 1 < N2) ? a2[(2 * i + 1 < N2) ? 2 * i + 1 : 0] : 0.0;
    a3[i] = (b2 ? hi : lo) + xchg_xor4(b2 ? lo : hi);
  }
#pragma unroll
  for (int i = 0; i < N4; i++) {
    const double lo = a3[2 * i], hi = (2 * i + 1 < N3) ? a3[(2 * i + 1 < N3) ? 2 * i + 1 : 0] : 0.0;
    out[i] = (b3 ? hi : lo) + xchg_dpp<kDppRowRor8>(b3 ? lo : hi);
  }
}

// N (3 .. 32) sums over all 64 lanes, every lane ends with every total: the rows' reduce-scatter, the two
// cross-row stages on the one or two values a lane keeps, and a scalar broadcast of total j from lane
// j % 16 -- the additions of group_allsum_n<64, N> (pairs, quads, eights, rows, row pairs, halves), so
// the same bits (tools/probe/allsum_rs_probe.hip).
template <int N>
__device__ __forceinline__ void rs64_allsum_generic(double (&v)[N]) {
  constexpr int NT = (N + 15) / 16;
  double tot[NT];
  row16_reduce_scatter<N>(v, tot);
#pragma unroll
  for (int k = 0; k < NT; k++) tot[k] = sum_xor32(sum_xor16(tot[k]));
#pragma unroll
  for (int j = 0; j < N; j++) v[j] = readlane_f64(tot[j / 16], j % 16);
}

// init0 + the sum over the chain's dimensions: lane-partial sum over this lane's valid slots, lane 0
// seeded with init0, then the butterfly -- or, for a kSeqSum group, init0 + v[dim 0] + v[dim 1] + ...
template <int G, int DPL, int D = G * DPL, bool kLds = false>
__device__ __forceinline__ double group_sum_slots(const double (&v)[DPL], const bool (&valid)[DPL],
                                                  int l, double init0) {
  if constexpr (kSeqSum<G, D>) {
    static_assert(DPL == 1, "one dimension per lane");
    double acc = init0;
    row_seqsum<D>(acc, v[0]);
    return acc;
  } else {
    double acc = (l == 0) ? init0 : 0.0;
#pragma unroll
    for (int k = 0; k < DPL; k++) acc = valid[k] ? (acc + v[k]) : acc;
    return group_allsum<G, kLds>(acc);
  }
}

// leapfrog.ex:39-42: 0.5 * sum(p * (M^-1 * p))
template <int G, int DPL, int D = G * DPL, bool kLds = false>
__device__ __forceinline__ double kinetic_energy(const double (&p)[DPL], const double (&im)[DPL],
                                                 const bool (&valid)[DPL]) {
  if constexpr (kSeqSum<G, D>) {
    double acc = 0.0;
    row_seqsum<D>(acc, p[0] * (im[0] * p[0]));
    return 0.5 * acc;
  } else {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DPL; k++) acc = valid[k] ? (acc + p[k] * (im[k] * p[k])) : acc;
    return 0.5 * group_allsum<G, kLds>(acc);
  }
}

// lane partials of the two dot products of tree.ex:1578-1588 for one (rho, pa, pb) triple
template <int DPL>
__device__ __forceinline__ void uturn_partials(const double (&rho)[DPL], const double (&pa)[DPL],
                                               const double (&pb)[DPL], const double (&im)[DPL],
                                               const bool (&valid)[DPL], double& sa, double& sb) {
  sa = 0.0;
  sb = 0.0;
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const double v = rho[k] * im[k];
    sa = valid[k] ? (sa + v * pa[k]) : sa;
    sb = valid[k] ? (sb + v * pb[k]) : sb;
  }
}

// tree.ex:1578-1588: rho-based U-turn test against two endpoint momenta
template <int G, int DPL, int D = G * DPL, bool kLds = false>
__device__ __forceinline__ bool uturn(const double (&rho)[DPL], const double (&pa)[DPL],
                                      const double (&pb)[DPL], const double (&im)[DPL],
                                      const bool (&valid)[DPL]) {
  double s[2];
  if constexpr (kSeqSum<G, D>) {
    const double v = rho[0] * im[0];
    s[0] = s[1] = 0.0;
    row_seqsum<D>(s[0], s[1], v * pa[0], v * pb[0]);
  } else {
    uturn_partials<DPL>(rho, pa, pb, im, valid, s[0], s[1]);
    if constexpr (G == 64 && (kLds || EXMC_XROW_PERMLANE)) {
      // (since the cross-row stages are permlane swaps the reduce-scatter touches no LDS: also for the
      // models whose waves are alone on their SIMDs -- radon's merges ran the full butterfly of six
      // until round 5: 108 instructions against 45)
      // only the signs of the two totals are needed: they sit in lanes 0 and 1
      const unsigned long long neg = __ballot((rs64_reduce2(s) < 0.0) ? 1 : 0);
      return (neg & 0x3ULL) != 0;
    } else {
      group_allsum_n<G, 2, kLds>(s);
    }
  }
  return (s[0] < 0.0) || (s[1] < 0.0);
}

// The three U-turn tests of one merge (tree.ex:1428-1446) in one pass of six reductions.
//   check 1: rho_all      against (pa1, pb1)
//   check 2: rho2         against (pa2, pb2)
//   check 3: rho3         against (pa3, pb3)
// The tests have no side effects, so evaluating all three (instead of short-circuiting) leaves
// every output unchanged. Returns {c1, c2 || c3}.
template <int G, int DPL, int D = G * DPL, bool kLds = false>
__device__ __forceinline__ void uturn3(const double (&r1)[DPL], const double (&a1)[DPL],
                                       const double (&b1)[DPL], const double (&r2)[DPL],
                                       const double (&a2)[DPL], const double (&b2)[DPL],
                                       const double (&r3)[DPL], const double (&a3)[DPL],
                                       const double (&b3)[DPL], const double (&im)[DPL],
                                       const bool (&valid)[DPL], bool& c1, bool& c23) {
  double s[6];
  if constexpr (kSeqSum<G, D>) {
    const double v1 = r1[0] * im[0], v2 = r2[0] * im[0], v3 = r3[0] * im[0];
    const double t[6] = {v1 * a1[0], v1 * b1[0], v2 * a2[0], v2 * b2[0], v3 * a3[0], v3 * b3[0]};
#pragma unroll
    for (int j = 0; j < 6; j++) s[j] = 0.0;
    row_seqsum6<D>(s, t);
  } else {
    uturn_partials<DPL>(r1, a1, b1, im, valid, s[0], s[1]);
    uturn_partials<DPL>(r2, a2, b2, im, valid, s[2], s[3]);
    uturn_partials<DPL>(r3, a3, b3, im, valid, s[4], s[5]);
    if constexpr (G == 64 && (kLds || EXMC_XROW_PERMLANE)) {
      // signs only: totals 0, 3, 2, 5, 1, 4 sit in lanes 0 .. 5 (rs64_reduce6)
      const unsigned long long neg = __ballot((rs64_reduce6(s) < 0.0) ? 1 : 0);
      c1 = (neg & ((1ULL << 0) | (1ULL << 4))) != 0;
      c23 = (neg & ((1ULL << 1) | (1ULL << 2) | (1ULL << 3) | (1ULL << 5))) != 0;
      return;
    } else {
      group_allsum_n<G, 6, kLds>(s);
    }
  }
  c1 = (s[0] < 0.0) || (s[1] < 0.0);
  c23 = (s[2] < 0.0) || (s[3] < 0.0) || (s[4] < 0.0) || (s[5] < 0.0);
}

// ---- dense inverse mass (opts[:dense_mass]; mass_matrix.ex:27-35,56-72,105-140) for the layouts
// that hold a whole chain in one lane (G = 1: DPL = D). cov = the covariance M^-1 (row-major D x D),
// chol = its lower Cholesky factor; both chain-invariant, so a wavefront reads them through the
// scalar cache. Products accumulate with fma in ascending index (the numeric contract's order for
// this mode). cov == nullptr means the diagonal mass.
struct DenseMass {
  const double* cov = nullptr;
  const double* chol = nullptr;
  const double* covp = nullptr;    // the lane layouts' permuted operands (LaneDense below)
  const double* cholp = nullptr;
};

template <int D>
__device__ __forceinline__ void dense_times(const double* cov, const double (&x)[D], double (&out)[D]) {
#pragma unroll
  for (int i = 0; i < D; i++) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < D; j++) acc = __builtin_fma(x[j], cov[i * D + j], acc);
    out[i] = acc;
  }
}

// leapfrog.ex:39-61: 0.5 * sum(p * (M^-1 p))
template <int D>
__device__ __forceinline__ double kinetic_energy_dense(const double* cov, const double (&p)[D]) {
  double mp[D];
  dense_times<D>(cov, p, mp);
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < D; k++) acc = acc + p[k] * mp[k];
  return 0.5 * acc;
}

// v = M^-1 rho, turn <=> v . pa < 0 or v . pb < 0 (the rule of tree.ex:1572-1588 with a dense M^-1)
template <int D>
__device__ __forceinline__ bool uturn_dense(const double* cov, const double (&rho)[D],
                                            const double (&pa)[D], const double (&pb)[D]) {
  double v[D];
  dense_times<D>(cov, rho, v);
  double sa = 0.0, sb = 0.0;
#pragma unroll
  for (int k = 0; k < D; k++) {
    sa = sa + v[k] * pa[k];
    sb = sb + v[k] * pb[k];
  }
  return (sa < 0.0) || (sb < 0.0);
}

// sampler.ex:412-427: solve L^T p = z by back substitution
template <int D>
__device__ __forceinline__ void dense_momentum(const double* chol, const double (&z)[D], double (&p)[D]) {
#pragma unroll
  for (int i = D - 1; i >= 0; i--) {
    double acc = z[i];
#pragma unroll
    for (int j = D - 1; j > i; j--) acc = __builtin_fma(-chol[j * D + i], p[j], acc);
    p[i] = acc / chol[i * D + i];
  }
}

// ---- the same four operations for the row layout (16 lanes, one dimension per lane, D <= 12):
// lane i holds row i of M^-1 (cr), the negated column i of its Cholesky factor below the diagonal
// (ncc[j] = -L[j][i]) and L[i][i]. Same fma chains and sums as above, in the same order. ----
template <int D>
struct RowDense {
  double cr[D], ncc[D], diag;
};

template <int D>
__device__ __forceinline__ double kinetic_energy_rowdense(const RowDense<D>& m, double p) {
  const double mp = row_matvec<D>(p, m.cr);
  double acc = 0.0;
  row_seqsum<D>(acc, p * mp);
  return 0.5 * acc;
}

template <int D>
__device__ __forceinline__ bool uturn_rowdense(const RowDense<D>& m, double rho, double pa, double pb) {
  const double v = row_matvec<D>(rho, m.cr);
  double sa = 0.0, sb = 0.0;
  row_seqsum<D>(sa, sb, v * pa, v * pb);
  return (sa < 0.0) || (sb < 0.0);
}

// solve L^T p = z by back substitution: at step j lane j's accumulator is final, its quotient by
// L[j][j] is p_j, and every lane takes fma(-L[j][i], p_j, acc) (only the lanes i < j still count)
template <int D, int J = D - 1>
__device__ __forceinline__ void rowdense_backsub(const RowDense<D>& m, int l, double& acc, double& p) {
  const double pj = acc / m.diag;
  p = (l == J) ? pj : p;
  if constexpr (J > 0) {
    const double pb = row_bcast_f64<J>(pj);
    acc = __builtin_fma(m.ncc[J], pb, acc);
    rowdense_backsub<D, J - 1>(m, l, acc, p);
  }
}
template <int D>
__device__ __forceinline__ double momentum_rowdense(const RowDense<D>& m, int l, double z) {
  double acc = z, p = 0.0;
  rowdense_backsub<D>(m, l, acc, p);
  return p;
}

// ---- the same operations for the lane layouts (a chain over G lanes, DPL dimensions per lane,
// kernel dimension i = lane + k G; sv, radon, logistic). cov and chol belong to the reference's
// FLAT vector (PointMap order), and every contraction runs in ascending flat index, so the
// operands are kept as
//   covp [D][GD]: covp[s GD + i]  = cov[rank(i)][s]     (GD = G DPL; columns i >= D are zero)
//   cholp[D][GD]: cholp[j GD + i] = chol[j][rank(i)]
// i.e. row s holds what every kernel dimension needs at step s of its chain, and the G lanes of a
// group read it as one coalesced line. The vector being contracted goes through a D-double strip of
// LDS in flat order (xs; one strip per group and vector), each step one broadcast read. A wave
// executes its LDS instructions in order, so a wave-level fence between the writes and the reads
// is all the exchange needs. ----
struct LaneDense {
  const double* covp = nullptr;
  const double* cholp = nullptr;
  double* xs = nullptr;     // LDS: this group's 3 D doubles
  double* covl = nullptr;   // LDS image of covp without its padding columns, [D][D] (+ 64 doubles
                            // that the slots past D may read), for models whose image is small
                            // enough not to cost resident waves (kImage); null: sweeps read covp
};

// (re)build the LDS image from covp: every group of the wave writes the same values
template <int G, int DPL, int D>
__device__ __forceinline__ void lane_dense_stage(const LaneDense& ld, int l) {
  constexpr int GD = G * DPL;
  for (int e = l; e < D * D; e += G) {
    const int s = e / D, i = e - s * D;
    ld.covl[e] = ld.covp[(size_t)s * GD + i];
  }
  for (int e = l; e < 64; e += G) ld.covl[D * D + e] = 0.0;
}

// s_waitcnt lgkmcnt(0) alone (vector-memory and export counters untouched): behind a batch of LDS reads
// whose values are consumed one after another it replaces the compiler's wait per consumer -- the same
// stall in total, fewer instructions, which is what a wave alone on its SIMD is short of
__device__ __forceinline__ void exmc_wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); }

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// out = M^-1 x for N vectors at once (the three rho of a merge share the sweep over covp)
template <int G, int DPL, int D, int N, bool kImage>
__device__ __forceinline__ void lane_dense_times(const LaneDense& ld, int l, const int (&rank)[DPL],
                                                 const bool (&valid)[DPL], const double (&x)[N][DPL],
                                                 double (&out)[N][DPL]) {
  constexpr int GD = G * DPL;
  wave_lds_fence();   // the previous sweep's reads are behind us
#pragma unroll
  for (int n = 0; n < N; n++)
#pragma unroll
    for (int k = 0; k < DPL; k++)
      if (valid[k]) ld.xs[n * D + rank[k]] = x[n][k];
  wave_lds_fence();
  double acc[N][DPL];
#pragma unroll
  for (int n = 0; n < N; n++)
#pragma unroll
    for (int k = 0; k < DPL; k++) acc[n][k] = 0.0;
  // rows in blocks of kB: the loads of a block are in flight together (a lone wave sits out one
  // L2 round trip per block), the fma chains then run in ascending s
  constexpr int kB = kImage ? 8 : 16;
  const double* row = (kImage ? ld.covl : ld.covp) + l;
  constexpr int kRow = kImage ? D : GD;
  auto block = [&](int s0, auto nb) {
    constexpr int NB = decltype(nb)::value;
    double c[NB][DPL];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int k = 0; k < DPL; k++) c[b][k] = row[(size_t)(s0 + b) * kRow + k * G];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int n = 0; n < N; n++) {
        const double xv = ld.xs[n * D + s0 + b];
#pragma unroll
        for (int k = 0; k < DPL; k++) acc[n][k] = __builtin_fma(xv, c[b][k], acc[n][k]);
      }
  };
#pragma nounroll
  for (int s0 = 0; s0 + kB <= D; s0 += kB) block(s0, std::integral_constant<int, kB>{});
  if constexpr (D % kB != 0) block(D - D % kB, std::integral_constant<int, D % kB>{});
#pragma unroll
  for (int n = 0; n < N; n++)
#pragma unroll
    for (int k = 0; k < DPL; k++) out[n][k] = valid[k] ? acc[n][k] : 0.0;   // slots past D read the next row
}

template <int G, int DPL, int D, bool kLds, bool kImage>
__device__ __forceinline__ double kinetic_energy_lanedense(const LaneDense& ld, int l, const int (&rank)[DPL],
                                                           const bool (&valid)[DPL], const double (&p)[DPL]) {
  static_assert(!kSeqSum<G, D>, "the lane layouts sum through the butterfly");
  double x[1][DPL], mp[1][DPL];
#pragma unroll
  for (int k = 0; k < DPL; k++) x[0][k] = p[k];
  lane_dense_times<G, DPL, D, 1, kImage>(ld, l, rank, valid, x, mp);
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < DPL; k++) acc = valid[k] ? (acc + p[k] * mp[0][k]) : acc;
  return 0.5 * group_allsum<G, kLds>(acc);
}

// the lane partials of the two dot products of a U-turn test, v = M^-1 rho given
template <int DPL>
__device__ __forceinline__ void uturn_partials_v(const double (&v)[DPL], const double (&pa)[DPL],
                                                 const double (&pb)[DPL], const bool (&valid)[DPL],
                                                 double& sa, double& sb) {
  sa = 0.0;
  sb = 0.0;
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    sa = valid[k] ? (sa + v[k] * pa[k]) : sa;
    sb = valid[k] ? (sb + v[k] * pb[k]) : sb;
  }
}

// solve L^T p = z on the flat vector by back substitution: at step j the accumulator of flat entry j
// is final, its quotient by L[j][j] is p_j, and every entry r < j takes fma(-L[j][r], p_j, acc).
// perm: flat entry -> kernel dimension (null = identity); base = first lane of this group.
template <int G, int DPL, int D>
__device__ __forceinline__ void lane_dense_momentum(const LaneDense& ld, const int32_t* perm, int l, int base,
                                                    const int (&rank)[DPL], const double (&z)[DPL],
                                                    double (&p)[DPL]) {
  constexpr int GD = G * DPL;
  double acc[DPL];
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    acc[k] = z[k];
    p[k] = 0.0;
  }
  const double* row = ld.cholp + l;
  double cn[DPL];   // the next row, loaded one step ahead of its use
#pragma unroll
  for (int k = 0; k < DPL; k++) cn[k] = row[(size_t)(D - 1) * GD + k * G];
#pragma nounroll
  for (int j = D - 1; j >= 0; j--) {
    const int dim = perm ? perm[j] : j;
    const int ks = dim / G;
    double c[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      c[k] = cn[k];
      cn[k] = row[(size_t)(j > 0 ? j - 1 : 0) * GD + k * G];
    }
    double a = acc[0], cd = c[0];
#pragma unroll
    for (int k = 1; k < DPL; k++) {
      a = (k == ks) ? acc[k] : a;
      cd = (k == ks) ? c[k] : cd;
    }
    const double pj = __shfl(a / cd, base | (dim & (G - 1)), 64);   // the owner's quotient
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      p[k] = (rank[k] == j) ? pj : p[k];
      acc[k] = (rank[k] < j) ? __builtin_fma(-c[k], pj, acc[k]) : acc[k];
    }
  }
}

}  // namespace exmc
