// exmc_nuts.hpp — the NUTS transition kernel and the on-device adaptation warmup for gfx950
// (SURVEY.md 8a a4-a19).
//
//   nuts_run     device function: n NUTS transitions of this lane group's chain (momentum draw,
//                iterative tree with multinomial proposals, rho-based U-turn checks, divergence
//                guard); replaces Tree.build/12 + nuts_step_with_stats (tree.ex:65-151,
//                sampler.ex:854-925) and the Rust crate native/exmc_tree.
//   nuts_kernel  nuts_run for every resident chain (sampling phase, sampler.ex:929-973).
//   warmup_kernel  run_warmup (sampler.ex:537-762) for chain 0 in ONE launch: dual averaging
//                (step_size.ex:13-50), Welford windows (mass_matrix.ex:40-97),
//                find_reasonable_epsilon (sampler.ex:451-530) all on the device. By default a
//                two-wave workgroup (PipeBox: the integrator wave runs one leaf ahead of the tree
//                wave) launched as several replicas of the same deterministic chain, of which
//                the first to finish publishes the result.
//
// The tree is built iteratively. The reference recursion (tree.ex:1144-1203) is unrolled into a
// per-level stack of pending "first halves": after leaf k completes, one inner merge fires for
// every pending level from the bottom up (tree.ex:1390-1476), each consuming one uniform; a node
// that is divergent or turning and has no pending sibling is returned upward unmerged exactly as
// `if first.divergent or first.turning` does (tree.ex:1175-1177); when the doubling's subtree is
// complete it is merged into the trajectory (tree.ex:1479-1568).
//
// Schedule: a wavefront holds 64/G chains, one per G-lane group, and the groups walk the tree
// skeleton in lock step (see nuts_run). The first schedule built was asynchronous — every pass
// gave each group one leapfrog and whatever bookkeeping that group's own tree phase needed, so
// no chain ever waited — but with four groups per wave nearly every pass paid for the union of
// all paths (momentum draw, direction, inner merges at several levels, outer merge, trace
// write): 27.7 ms for eight_schools 4096 x 1000 against 17.7 ms for the lock-step walk, where
// idle groups cost 18 % more passes and each pass costs ~0.55x. Also measured and rejected on
// the asynchronous loop: capping merges at one per pass (+20 %), one unified inner/outer merge
// body (+22 %), G = 32 with two waves per SIMD (+31 %).
//
// Memory plan per workgroup (one wavefront): the first LDSL stack levels and the ziggurat
// tables live in LDS; deeper levels spill to a global scratch that stays L2-resident. Per-chain
// dot products run on DPP butterflies (exmc_device.hpp), six at a time per merge.
#pragma once

#include <type_traits>

#include "exmc_models.hpp"

// EXMC_ABLATE = 1..4 builds timing-only variants (U-turn reductions / merge transcendentals /
// model gradient / leaf exp stubbed out) for tools/kernel_probe.py; outputs are wrong then.
#ifndef EXMC_ABLATE
#define EXMC_ABLATE 0
#endif

// EXMC_PROFILE_SECTIONS = 1 builds a variant whose main loop accumulates s_memtime cycles per
// section (wave time, so divergence shows up where it is paid); the library prints the totals
// after each timed launch. Development only, never the shipped build.
#ifndef EXMC_PROFILE_SECTIONS
#define EXMC_PROFILE_SECTIONS 0
#endif

namespace exmc {

#ifdef EXMC_XCC_PROBE
__device__ double g_wave_probe[4096 * 5];   // per workgroup of nuts_kernel: placement (xcc, cu, simd), shader
                                            // clocks, leapfrogs, wall-clock start and end (100 MHz ticks)
#endif

#if EXMC_PROFILE_SECTIONS
__device__ unsigned long long g_prof[16];
#define EXMC_PROF_DECL long long prof_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long prof_t_ = clock64();
#define EXMC_PROF(i) { const long long n_ = clock64(); prof_[i] += n_ - prof_t_; prof_t_ = n_; }
#define EXMC_PROF_COUNT(i) { prof_[i] += 1; }
#define EXMC_PROF_FLUSH                                                        \
  if ((threadIdx.x & 63) == 0) {                                               \
    for (int i_ = 0; i_ < 10; i_++) atomicAdd(&g_prof[i_], (unsigned long long)prof_[i_]); \
  }
// integrator wave of the two-wave pipeline: [10] units, [11] unit compute, [12] barrier waits, [13] rest
#define EXMC_IPROF_DECL long long iprof_[4] = {0, 0, 0, 0}; long long iprof_t_ = clock64();
#define EXMC_IPROF(i) { const long long n_ = clock64(); iprof_[i] += n_ - iprof_t_; iprof_t_ = n_; }
#define EXMC_IPROF_COUNT { iprof_[0] += 1; }
#define EXMC_IPROF_FLUSH                                                       \
  if ((threadIdx.x & 63) == 0) {                                               \
    for (int i_ = 0; i_ < 4; i_++) atomicAdd(&g_prof[10 + i_], (unsigned long long)iprof_[i_]); \
  }
#else
#define EXMC_IPROF_DECL
#define EXMC_IPROF(i)
#define EXMC_IPROF_COUNT
#define EXMC_IPROF_FLUSH
#define EXMC_PROF_DECL
#define EXMC_PROF(i)
#define EXMC_PROF_COUNT(i)
#define EXMC_PROF_FLUSH
#endif

// The reference's flat vector is the free RVs sorted by id as strings (point_map.ex:30-60) and
// the RNG-consuming steps fill it front to back (init_position sampler.ex:339-349,
// sample_momentum_fast sampler.ex:393-403). The kernels keep their own compute layout; a model
// whose kernel order is not the sorted one carries the permutation: perm[r] = kernel dimension of
// flat entry r, rank[i] = flat position of kernel dimension i (both null = identity).
struct FlatOrder {
  const int32_t* perm = nullptr;   // dev [D]
  const int32_t* rank = nullptr;   // dev [D]
};

struct ChainState {
  double* q;       // [D][C]
  double* g;       // [D][C]
  double* logp;    // [C]
  uint64_t* rng;   // [2][C]
};

struct TraceDev {
  double* draws;   // [S][D][C]
  double* logp;    // [S][C]
  int32_t* tree_depth;
  int32_t* n_steps;
  int32_t* divergent;
  double* accept_prob;
  double* energy;
};

struct NutsParams {
  ChainState st;
  int n_chains;
  int n_draws;       // transitions to run in this launch
  int draw_offset;   // first trace row written
  double eps;
  int max_depth;
  const double* inv_mass;       // dev [D]
  const double* sqrt_inv_mass;  // dev [D]
  TraceDev tr;
  double* stack;                // dev scratch for levels >= LDSL
  unsigned long long* counters; // [0] leapfrogs, [1] divergent transitions
  void* scratch;                // dev, 8 bytes: where trace outputs the caller left null are written
  const uint64_t* zig_ki;
  const double* zig_wi;
  const double* zig_fi;
  double nor_r;
  FlatOrder flat;
  DenseMass dm;   // opts[:dense_mass] in force (lanes_per_chain = 1 only)
  int prio;       // 1: time-sliced issue priority between the two waves of a SIMD (kNutsWavesPerSimd == 2)
  int simds;      // SIMDs of the device (4 per compute unit): the pairing below holds for 2 * simds waves
  int* mig;       // chain migration (Model::kMigrate): a zeroed MigBoard, or null
  int* progress;  // kStream launches: where chain 0's count of finished draws is published (host memory)
};

// ---- chain migration (sv: one chain per wave, 2048 waves = two per SIMD) ----
// The launch ends with its slowest chain, and the per-chain leapfrog totals spread (sd 13 % of the
// mean): towards the end some SIMDs still hold two chains, at half speed each, while others have
// run empty. A chain is a serial Markov sequence, but WHERE its next transition runs is free: its
// state between transitions is (q, g, logp, rng), 2 d + 3 doubles. So a wave that finishes the
// last chain of its SIMD stays as a host and advertises itself; a wave that starts a transition
// while its SIMD holds two live chains and a host is advertised stores its chain, hands
// (chain, draws done) to the host's mailbox and exits; the host loads the chain and goes on --
// same arithmetic, same generator state, bit-identical draws. All of it is wave-uniform scalar
// work on global words; a host polls with s_sleep on a SIMD nobody else uses.
//   board[0]              remaining chains (set by the host to C before the launch)
//   board[1]              the advertised host: 0 or its workgroup index + 1
//   board[2], board[3]    statistics: chains moved, waves that became hosts
//   board[8 .. 8+16384)   live chains per SIMD, indexed by (XCC_ID, SE, SH, CU, SIMD)
//   board[16392 + 2 b]    mailbox of workgroup b: chain + 1 (0: empty), draws done
//   board[16392 + 2 W + b]  (W workgroups) what workgroup b estimates is left of its chain, in
//                         thousands of leapfrogs + 1 (0: not known yet): its SIMD partner reads it
//   board[16392 + 3 W + 2 u + s]  the occupants of SIMD u: the workgroup that arrived there s-th
//                         (s = 0, 1), + 1 (0: nobody yet) -- how a wave finds its SIMD partner
constexpr int kMigSimds = 16384;
constexpr int kMigLive = 8;
constexpr int kMigMail = kMigLive + kMigSimds;
__host__ __device__ constexpr size_t mig_board_ints(size_t n_workgroups) {
  return kMigMail + 3 * n_workgroups + 2 * (size_t)kMigSimds;
}

__device__ __forceinline__ int mig_simd_uid() {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  return (int)(((xcc & 0xf) << 10) | (((hw >> 8) & 0xff) << 2) | ((hw >> 4) & 0x3));
}
__device__ __forceinline__ int mig_load(int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kMaxLevels = 12;
constexpr int kNutsBlock = 64;      // one wavefront per workgroup
constexpr int kZigLdsBytes = 768 * 8;

// doubles per lane of one pending tree node: rho, p_in, p_out, q_prop, g_prop (DPL each), lsw,
// logp_prop, acc -- without g_prop for a model that recomputes the proposal's gradient at the end of
// the transition instead of carrying it through every node (M::kRegradProposal)
template <class M>
__host__ __device__ constexpr int nuts_nslot() { return (M::kRegradProposal ? 4 : 5) * M::DPL + 3; }

// dynamic LDS of a NUTS workgroup: [tree stack levels < LDSL][ziggurat tables][model scratch]
// [model data image]. kZig = false: the ziggurat tables stay in global memory (6 KB, L2-resident)
// -- the sampling kernel of a model that draws its momentum once per few hundred leapfrogs and
// needs the space for another stack level (M::kNutsZigInLds, sv).
template <bool kZig>
__host__ __device__ constexpr size_t nuts_zig_doubles() { return kZig ? kZigLdsBytes / 8 : 0; }
template <class M, int LDSL, bool kZig = true>
__host__ __device__ constexpr size_t nuts_lds_data_offset() {   // in doubles
  // (an even count: what follows -- a model's data image, the warmup's staged image -- is read in
  // 16-byte pairs)
  return ((size_t)LDSL * nuts_nslot<M>() * kNutsBlock + nuts_zig_doubles<kZig>() +
          (size_t)M::kExtraLdsDoubles + (size_t)M::kDenseLdsDoubles + 1) & ~(size_t)1;
}
template <class M, int LDSL, bool kZig = true>
__host__ __device__ constexpr size_t nuts_lds_bytes() {
  return (nuts_lds_data_offset<M, LDSL, kZig>() + (size_t)M::kLdsDataDoubles) * 8;
}
// doubles of global spill stack per workgroup (levels >= LDSL of every lane of one wavefront,
// contiguous: level, then slot, then lane)
template <class M, int LDSL>
__host__ __device__ constexpr size_t nuts_spill_doubles() {
  return (size_t)((kMaxLevels > LDSL) ? (kMaxLevels - LDSL) : 1) * nuts_nslot<M>() * kNutsBlock;
}

// Models with kLdsDataDoubles > 0 keep their observations in LDS for the whole kernel
// (M::stage_data fills the image, Lane::xoff is its offset in doubles or -1): with one wave per
// SIMD every global load inside logp_grad is an exposed L2 round trip.
template <class M, int G, int LDSL, bool kZig = true>
__device__ __forceinline__ int stage_model_data(const typename M::Consts& mc, double* lds) {
  if constexpr (M::kLdsDataDoubles > 0) {
    const int off = (int)nuts_lds_data_offset<M, LDSL, kZig>();
    const bool ok = M::stage_data(mc, lds + off);
    __syncthreads();
    return ok ? off : -1;
  } else {
    return -1;
  }
}

// the LDS levels of the stack through pointers that SAY they are LDS: where the level is a run-time
// value (eight_schools: one loop over all levels) the compiler otherwise merges the LDS and the global
// branch into one generic pointer and emits flat_load / flat_store for both -- a flat access to LDS
// goes through the vector-memory path and both wait counters
typedef __attribute__((address_space(3))) double lds_f64;
template <int N>
__device__ __forceinline__ void node_load_lds(const double* base, size_t stride, double (&nd)[N]) {
  const lds_f64* b = (const lds_f64*)base;
#pragma unroll
  for (int s = 0; s < N; s++) nd[s] = b[(size_t)s * stride];
}
template <int N>
__device__ __forceinline__ void node_store_lds(double* base, size_t stride, const double (&nd)[N]) {
  lds_f64* b = (lds_f64*)base;
#pragma unroll
  for (int s = 0; s < N; s++) b[(size_t)s * stride] = nd[s];
}
template <int N>
__device__ __forceinline__ void node_load(const double* base, size_t stride, double (&nd)[N]) {
#pragma unroll
  for (int s = 0; s < N; s++) nd[s] = base[(size_t)s * stride];
}
template <int N>
__device__ __forceinline__ void node_store(double* base, size_t stride, const double (&nd)[N]) {
#pragma unroll
  for (int s = 0; s < N; s++) base[(size_t)s * stride] = nd[s];
}

// per-lane constants of a chain group
template <class M, int G>
struct NutsLane {
  static constexpr int DPL = M::DPL;
  typename M::Lane ln;
  double im[DPL], sim[DPL];
  bool valid[DPL];
  int rank[DPL];     // flat (draw) position of this lane's dimensions; M::D for slots past D
  const int32_t* perm;   // flat position -> kernel dimension, null = identity
  DenseMass dm;          // opts[:dense_mass], one-lane-per-chain layouts; cov == null: diagonal
  RowDense<M::kRowDense ? M::D : 1> rd;   // opts[:dense_mass], row layout (M::kRowDense): this lane's
                                          // row of M^-1 and column of its Cholesky factor
  LaneDense ld;          // opts[:dense_mass], lane layouts (M::kLaneDense); covp == null: diagonal
  int l;
  int prio_slot;     // 0 / 1: which of the two waves of its SIMD this is (time-sliced issue priority,
                     // see nuts_run); -1: the wave has its SIMD to itself
  int prio_duty;     // sixteenths of the time slot 0 holds the priority (8: even shares)
  double* lstk;      // this lane's column of the LDS stack
  double* gstk;      // this lane's column of its workgroup's block of the global spill stack
  ZigTables zt;
  double nor_r;
  bool alive;        // false: a lane group kept only so that wave-cooperative models see all lanes
};

// the mass-dependent operations of a transition: diagonal everywhere; dense (L.dm.cov set, a
// wave-uniform choice) for the layouts that keep a whole chain in one lane
template <class M, int G>
__device__ __forceinline__ double mass_ke(const NutsLane<M, G>& L, const double (&p)[M::DPL]) {
  if constexpr (M::kRowDense) return kinetic_energy_rowdense<M::D>(L.rd, p[0]);
  if constexpr (G == 1) {
    if (L.dm.cov) return kinetic_energy_dense<M::D>(L.dm.cov, p);
  }
  if constexpr (M::kLaneDense) {
    if (L.ld.covp) return kinetic_energy_lanedense<G, M::DPL, M::D, M::kXRowLds, M::kDenseImage>(L.ld, L.l, L.rank, L.valid, p);
  }
  return kinetic_energy<G, M::DPL, M::D, M::kXRowLds>(p, L.im, L.valid);
}
// q += eps * (M^-1 p_half)
template <class M, int G>
__device__ __forceinline__ void mass_drift(const NutsLane<M, G>& L, double eps, const double (&ph)[M::DPL],
                                           double (&q)[M::DPL]) {
  if constexpr (M::kRowDense) {
    q[0] = q[0] + eps * row_matvec<M::D>(ph[0], L.rd.cr);
    return;
  }
  if constexpr (G == 1) {
    if (L.dm.cov) {
      double mp[M::DPL];
      dense_times<M::D>(L.dm.cov, ph, mp);
#pragma unroll
      for (int k = 0; k < M::DPL; k++) q[k] = q[k] + eps * mp[k];
      return;
    }
  }
  if constexpr (M::kLaneDense) {
    if (L.ld.covp) {
      double x[1][M::DPL], mp[1][M::DPL];
#pragma unroll
      for (int k = 0; k < M::DPL; k++) x[0][k] = ph[k];
      lane_dense_times<G, M::DPL, M::D, 1, M::kDenseImage>(L.ld, L.l, L.rank, L.valid, x, mp);
#pragma unroll
      for (int k = 0; k < M::DPL; k++) q[k] = q[k] + eps * mp[0][k];
      return;
    }
  }
#pragma unroll
  for (int k = 0; k < M::DPL; k++) q[k] = q[k] + eps * (L.im[k] * ph[k]);
}
template <class M, int G>
__device__ __forceinline__ bool mass_uturn(const NutsLane<M, G>& L, const double (&rho)[M::DPL],
                                           const double (&pa)[M::DPL], const double (&pb)[M::DPL]) {
  if constexpr (M::kRowDense) return uturn_rowdense<M::D>(L.rd, rho[0], pa[0], pb[0]);
  if constexpr (G == 1) {
    if (L.dm.cov) return uturn_dense<M::D>(L.dm.cov, rho, pa, pb);
  }
  if constexpr (M::kLaneDense) {
    if (L.ld.covp) {
      double x[1][M::DPL], v[1][M::DPL], s[2];
#pragma unroll
      for (int k = 0; k < M::DPL; k++) x[0][k] = rho[k];
      lane_dense_times<G, M::DPL, M::D, 1, M::kDenseImage>(L.ld, L.l, L.rank, L.valid, x, v);
      uturn_partials_v<M::DPL>(v[0], pa, pb, L.valid, s[0], s[1]);
      if constexpr (G == 64 && EXMC_XROW_PERMLANE) {   // signs only, as exmc_device.hpp uturn
        const unsigned long long neg = __ballot((rs64_reduce2(s) < 0.0) ? 1 : 0);
        return (neg & 0x3ULL) != 0;
      }
      group_allsum_n<G, 2, M::kXRowLds>(s);
      return (s[0] < 0.0) || (s[1] < 0.0);
    }
  }
  return uturn<G, M::DPL, M::D, M::kXRowLds>(rho, pa, pb, L.im, L.valid);
}
template <class M, int G>
__device__ __forceinline__ void mass_uturn3(const NutsLane<M, G>& L, const double (&r1)[M::DPL],
                                            const double (&a1)[M::DPL], const double (&b1)[M::DPL],
                                            const double (&r2)[M::DPL], const double (&a2)[M::DPL],
                                            const double (&b2)[M::DPL], const double (&r3)[M::DPL],
                                            const double (&a3)[M::DPL], const double (&b3)[M::DPL],
                                            bool& c1, bool& c23) {
  if constexpr (M::kRowDense) {
    c1 = uturn_rowdense<M::D>(L.rd, r1[0], a1[0], b1[0]);
    const bool c2 = uturn_rowdense<M::D>(L.rd, r2[0], a2[0], b2[0]);
    const bool c3 = uturn_rowdense<M::D>(L.rd, r3[0], a3[0], b3[0]);
    c23 = c2 || c3;
    return;
  }
  if constexpr (G == 1) {
    if (L.dm.cov) {
      c1 = uturn_dense<M::D>(L.dm.cov, r1, a1, b1);
      const bool c2 = uturn_dense<M::D>(L.dm.cov, r2, a2, b2);
      const bool c3 = uturn_dense<M::D>(L.dm.cov, r3, a3, b3);
      c23 = c2 || c3;
      return;
    }
  }
  if constexpr (M::kLaneDense) {
    if (L.ld.covp) {
      double x[3][M::DPL], v[3][M::DPL], s[6];
#pragma unroll
      for (int k = 0; k < M::DPL; k++) {
        x[0][k] = r1[k];
        x[1][k] = r2[k];
        x[2][k] = r3[k];
      }
      lane_dense_times<G, M::DPL, M::D, 3, M::kDenseImage>(L.ld, L.l, L.rank, L.valid, x, v);
      uturn_partials_v<M::DPL>(v[0], a1, b1, L.valid, s[0], s[1]);
      uturn_partials_v<M::DPL>(v[1], a2, b2, L.valid, s[2], s[3]);
      uturn_partials_v<M::DPL>(v[2], a3, b3, L.valid, s[4], s[5]);
      if constexpr (G == 64 && EXMC_XROW_PERMLANE) {   // signs only, as exmc_device.hpp uturn3
        const unsigned long long neg = __ballot((rs64_reduce6(s) < 0.0) ? 1 : 0);
        c1 = (neg & ((1ULL << 0) | (1ULL << 4))) != 0;
        c23 = (neg & ((1ULL << 1) | (1ULL << 2) | (1ULL << 3) | (1ULL << 5))) != 0;
        return;
      }
      group_allsum_n<G, 6, M::kXRowLds>(s);
      c1 = (s[0] < 0.0) || (s[1] < 0.0);
      c23 = (s[2] < 0.0) || (s[3] < 0.0) || (s[4] < 0.0) || (s[5] < 0.0);
      return;
    }
  }
  uturn3<G, M::DPL, M::D, M::kXRowLds>(r1, a1, b1, r2, a2, b2, r3, a3, b3, L.im, L.valid, c1, c23);
}

// a chain's state between transitions
template <int DPL>
struct ChainRegs {
  double q[DPL], g[DPL];
  double logp;
  Rng rng;
};

// cooperative: every lane of the workgroup must call this before any lane leaves
template <int LDSL, int NSLOT, bool kZig = true>
__device__ __forceinline__ ZigTables stage_zig_tables(double* lds, const uint64_t* ki,
                                                      const double* wi, const double* fi) {
  if constexpr (!kZig) return ZigTables{ki, wi, fi};   // read where they are (see nuts_lds_data_offset)
  double* lz = lds + (size_t)LDSL * NSLOT * kNutsBlock;
  for (int i = threadIdx.x; i < 256; i += kNutsBlock) {
    lz[i] = __longlong_as_double((long long)ki[i]);
    lz[256 + i] = wi[i];
    lz[512 + i] = fi[i];
  }
  __syncthreads();
  return ZigTables{(const uint64_t*)lz, lz + 256, lz + 512};
}

// smallest `pos` over the chain groups of the wavefront (pos is uniform inside a group)
template <int G>
__device__ __forceinline__ int wave_min_pos(int pos) {
  if constexpr (G == 64) {
    return __builtin_amdgcn_readfirstlane(pos);
  } else if constexpr (G >= 16) {
    // an exited or masked group's register holds whatever it last wrote; 0 is always a safe bound
    int m = __builtin_amdgcn_readlane(pos, 0);
#pragma unroll
    for (int b = G; b < 64; b += G) m = min(m, __builtin_amdgcn_readlane(pos, b));
    return max(m, 0);
  } else {
    return 0;
  }
}

// smallest value over the G-lane chain group (32-bit, DPP inside a row, v_readlane across rows)
template <int G>
__device__ __forceinline__ int group_min_i32(int v) {
  if (G >= 2) v = min(v, __builtin_amdgcn_mov_dpp(v, kDppXor1, 0xF, 0xF, true));
  if (G >= 4) v = min(v, __builtin_amdgcn_mov_dpp(v, kDppXor2, 0xF, 0xF, true));
  if (G >= 8) v = min(v, __builtin_amdgcn_mov_dpp(v, kDppHalfMirror, 0xF, 0xF, true));
  if (G >= 16) v = min(v, __builtin_amdgcn_mov_dpp(v, kDppRowMirror, 0xF, 0xF, true));
  if (G >= 32) {
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int h0 = min(r0, r1), h1 = min(r2, r3);
    if (G == 32) v = ((threadIdx.x & 63) < 32) ? h0 : h1;
    else v = min(h0, h1);
  }
  return v;
}

// sampler.ex:393-403: d sequential normal_s draws -> p = z / sqrt(M^-1), variate r going to the
// r-th entry of the reference's flat vector (L.rank / L.perm, see FlatOrder). normal_s accepts
// 98.5 % of its 58-bit words at once (one word per variate) and otherwise consumes a
// data-dependent number of further words, so the draws are sequential in principle. Here: the
// group advances a copy of the stream over all remaining draws, each lane keeping the generator
// state in front of the draws of its own dimensions, and every lane tests its own word at once.
// Up to the first draw j whose word is not accepted at once the variates are final; draw j is then
// made the long way from the state in front of it (taken from the lane that owns that
// dimension), and the pass repeats from j + 1. Expected passes: 1 + 0.015 d, against d sequential
// draws.
//
// 64 lanes per chain (round 5): the generator state is the same on every lane, so the stream is advanced
// on the SCALAR unit -- eight 64-bit instructions a step where the vector unit needed thirteen plus ten
// for the lanes' selects -- and the tail word b_i of the state in front of word i is written into lane
// i % 64 of a register pair (v_writelane): an orbit of D + 16 words, built once. State i is
// (b_{i-1}, b_i) (the head word of a state is the tail word of the one before), so the orbit holds
// every state. Draw r reads the word at r + shift, shift = the words earlier draws consumed beyond
// their first; after a draw made the long way only that shift changes and the lanes fetch again
// (ds_bpermute) -- no second advance unless the shift runs past the orbit's margin.
template <int NORB>
struct RngOrbit {
  static_assert(NORB <= 128, "two register pairs of 64 lanes");
  uint32_t lo[2], hi[2];   // lane i of pair w: b_{64 w + i}
  uint64_t a0;             // head word of state 0
  // (v_writelane reads its data from a scalar register, so the lane has to be an immediate -- one scalar
  // operand per vector instruction: the steps are unrolled, ten instructions each)
  template <int I>
  __device__ __forceinline__ void step(uint64_t& a, uint64_t& b) {
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(lo[I >> 6]) : "s"((uint32_t)b), "n"(I & 63));
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(hi[I >> 6]) : "s"((uint32_t)(b >> 32)), "n"(I & 63));
    Rng r{a, b};
    rng_advance(r);
    a = r.a;
    b = r.b;
  }
  template <int... I>
  __device__ __forceinline__ void steps(uint64_t& a, uint64_t& b, std::integer_sequence<int, I...>) {
    (step<I>(a, b), ...);
  }
  __device__ __forceinline__ void build(const Rng& rng) {
    uint64_t a = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(rng.a >> 32)) << 32) |
                 (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)rng.a);
    uint64_t b = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(rng.b >> 32)) << 32) |
                 (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)rng.b);
    a0 = a;
    lo[0] = lo[1] = hi[0] = hi[1] = 0;
    steps(a, b, std::make_integer_sequence<int, NORB>{});
  }
  // b_p for a per-lane position p in [0, NORB)
  __device__ __forceinline__ uint64_t at(int p) const {
    const int src = (p & 63) << 2;
    const uint32_t l0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)lo[0]);
    const uint32_t h0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)hi[0]);
    const uint32_t l1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)lo[1]);
    const uint32_t h1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)hi[1]);
    const bool up = p >= 64;
    return ((uint64_t)(up ? h1 : h0) << 32) | (uint64_t)(up ? l1 : l0);
  }
  // state p for a wave-uniform position p in [0, NORB)
  __device__ __forceinline__ Rng state(int p) const {
    auto word = [&](int q) -> uint64_t {
      const int w = q >> 6, i = q & 63;
      const uint32_t l = (w == 0) ? __builtin_amdgcn_readlane((int)lo[0], i) : __builtin_amdgcn_readlane((int)lo[1], i);
      const uint32_t h = (w == 0) ? __builtin_amdgcn_readlane((int)hi[0], i) : __builtin_amdgcn_readlane((int)hi[1], i);
      return ((uint64_t)h << 32) | (uint64_t)l;
    };
    Rng r;
    r.b = word(p);
    r.a = (p == 0) ? a0 : word(p > 0 ? p - 1 : 0);
    return r;
  }
};

#ifndef EXMC_RNG_ORBIT_MARGIN
#define EXMC_RNG_ORBIT_MARGIN 16
#endif

template <class M, int G>
__device__ __forceinline__ void draw_momentum_orbit(const NutsLane<M, G>& L, Rng& rng, double (&z)[M::DPL]) {
  constexpr int D = M::D, DPL = M::DPL;
  constexpr int NORB = D + EXMC_RNG_ORBIT_MARGIN;
#pragma unroll
  for (int k = 0; k < DPL; k++) z[k] = 0.0;
  int pos = 0;   // draws [0, pos) are final and rng stands in front of draw pos
  for (;;) {
    RngOrbit<NORB> orb;
    orb.build(rng);
    const int org = pos;   // orbit position 0 = the state in front of draw org (with shift 0)
    int shift = 0;
    bool rebuild = false;
    for (;;) {
      bool fail[DPL];
#pragma unroll
      for (int k = 0; k < DPL; k++) {
        const bool live = L.valid[k] && (L.rank[k] >= pos);
        const int pp = live ? (L.rank[k] - org + shift) : 0;
        double zz;
        const bool acc = normal_fast(rng_scramble(orb.at(pp)), L.zt, zz);
        fail[k] = live && !acc;
        z[k] = (live && acc) ? zz : z[k];
      }
      int j = D;   // first draw that needs the long way, D if none
      if (L.perm == nullptr) {
#pragma unroll
        for (int k = DPL - 1; k >= 0; k--) {
          const unsigned long long m = __ballot(fail[k] ? 1 : 0);
          j = (m != 0) ? (k * G + __ffsll((long long)m) - 1) : j;
        }
      } else {
#pragma unroll
        for (int k = 0; k < DPL; k++) j = fail[k] ? min(j, L.rank[k]) : j;
        j = group_min_i32<G>(j);
      }
      j = __builtin_amdgcn_readfirstlane(j);
      if (j >= D) {
        rng = orb.state(D - org + shift);
        return;
      }
      Rng rj = orb.state(j - org + shift);
      int words = 0;
      const double zz = rng_normal_counted(rj, L.zt, L.nor_r, words);
#pragma unroll
      for (int k = 0; k < DPL; k++) z[k] = (L.rank[k] == j) ? zz : z[k];
      words = __builtin_amdgcn_readfirstlane(words);
      pos = j + 1;
      shift += words - 1;
      if (D - org + shift > NORB - 1) {   // the remaining draws run past the orbit: build another from here
        rng = rj;
        rebuild = true;
        break;
      }
    }
    if (!rebuild) return;
    if (pos >= D) return;   // (rng = rj already stands behind the last draw)
  }
}

template <class M, int G>
__device__ __forceinline__ void draw_momentum_variates(const NutsLane<M, G>& L, Rng& rng,
                                                       double (&z)[M::DPL]) {
  constexpr int D = M::D, DPL = M::DPL;
  if constexpr (G == 64 && M::kRngOrbit && D + EXMC_RNG_ORBIT_MARGIN <= 128) {
    draw_momentum_orbit<M, G>(L, rng, z);
    return;
  }
  const int base = (threadIdx.x & 63) & ~(G - 1);
#pragma unroll
  for (int k = 0; k < DPL; k++) z[k] = 0.0;
  int pos = 0;   // draws [0, pos) are final and rng stands in front of draw pos
  for (;;) {
    Rng r2 = rng;
    uint64_t sa[DPL], sb[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) { sa[k] = 0; sb[k] = 0; }
    // kept rolled: unrolled, the D lane predicates (rank == i) are hoisted into scalar register
    // pairs for the whole kernel and the sampling loop pays for them in v_readlane spills
    const int i0 = wave_min_pos<G>(pos);
#pragma nounroll
    for (int i = i0; i < D; i++) {
      const bool act = (G == 64) || (i >= pos);
#pragma unroll
      for (int k = 0; k < DPL; k++) {
        const bool mine = act && (L.rank[k] == i);
        sa[k] = mine ? r2.a : sa[k];
        sb[k] = mine ? r2.b : sb[k];
      }
      Rng nx = r2;
      rng_advance(nx);
      r2.a = act ? nx.a : r2.a;
      r2.b = act ? nx.b : r2.b;
    }
    bool fail[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const bool live = L.valid[k] && (L.rank[k] >= pos);
      double zz;
      const bool acc = normal_fast(rng_scramble(sb[k]), L.zt, zz);
      fail[k] = live && !acc;
      z[k] = (live && acc) ? zz : z[k];
    }
    int j = D;   // first draw of this group that needs the long way, D if none
    if (L.perm == nullptr) {
      // identity order: the first failing dimension is the first set bit of the group's ballot
#pragma unroll
      for (int k = DPL - 1; k >= 0; k--) {
        const unsigned long long m = __ballot(fail[k] ? 1 : 0);
        const unsigned long long gm = (G == 64) ? m : ((m >> base) & ((1ULL << (G & 63)) - 1ULL));
        j = (gm != 0) ? (k * G + __ffsll((long long)gm) - 1) : j;
      }
    } else {
#pragma unroll
      for (int k = 0; k < DPL; k++) j = fail[k] ? min(j, L.rank[k]) : j;
      j = group_min_i32<G>(j);
    }
    if (__any((j < D) ? 1 : 0) == 0) {
      rng = r2;
      break;
    }
    if (j < D) {
      const int dim = (L.perm == nullptr) ? j : L.perm[j];   // the dimension draw j belongs to
      const int ks = dim / G;
      uint64_t aj = 0, bj = 0;
#pragma unroll
      for (int k = 0; k < DPL; k++) {
        aj = (k == ks) ? sa[k] : aj;
        bj = (k == ks) ? sb[k] : bj;
      }
      const int src = base | (dim & (G - 1));
      Rng rj;
      rj.a = __shfl(aj, src, 64);
      rj.b = __shfl(bj, src, 64);
      const double zz = rng_normal(rj, L.zt, L.nor_r);
#pragma unroll
      for (int k = 0; k < DPL; k++) z[k] = (L.rank[k] == j) ? zz : z[k];
      rng = rj;
      pos = j + 1;
    } else {
      rng = r2;
      pos = D;
    }
  }
}

// p = z / sqrt(M^-1) (sampler.ex:400), or the dense form p = L^-T z (sampler.ex:412-427)
template <class M, int G>
__device__ __forceinline__ void momentum_from_variates(const NutsLane<M, G>& L, const double (&z)[M::DPL],
                                                       double (&p)[M::DPL]) {
  if constexpr (M::kRowDense) {
    p[0] = momentum_rowdense<M::D>(L.rd, L.l, z[0]);
    return;
  }
  if constexpr (G == 1) {
    if (L.dm.chol) {
      dense_momentum<M::D>(L.dm.chol, z, p);
      return;
    }
  }
  if constexpr (M::kLaneDense) {
    if (L.ld.cholp) {
      lane_dense_momentum<G, M::DPL, M::D>(L.ld, L.perm, L.l, (threadIdx.x & 63) & ~(G - 1), L.rank, z, p);
      return;
    }
  }
#pragma unroll
  for (int k = 0; k < M::DPL; k++) p[k] = z[k] / L.sim[k];
}

template <class M, int G>
__device__ __forceinline__ void draw_momentum(const NutsLane<M, G>& L, Rng& rng,
                                              double (&p)[M::DPL]) {
  double z[M::DPL];
  draw_momentum_variates<M, G>(L, rng, z);
  momentum_from_variates<M, G>(L, z, p);
}

// ------------------------------------------------------------------------------------------
// Two-wave pipeline (the serial warmup; the paired sampling kernel): wave 0 keeps the tree
// (nuts_run with a PipeBox), wave 1 integrates. Both walk the same skeleton -- transition,
// doubling, unit -- and meet at one workgroup barrier per unit. A unit is what the integrator hands
// over: in doubling 0 the single leaf, in every later doubling a pair of leaves (2k, 2k+1). The
// tree wave merges the pair at level 0 in registers (pipe_pair_unit: tree.ex:1390-1476 with
// depth 1 -- or keeps leaf 2k alone when it diverged, tree.ex:1175-1177) and ascends the result
// through levels >= 1, while the integrator computes the next pair into the other of two LDS
// slots: half the barriers of a leaf-per-barrier hand-over and no level-0 traffic on the stack.
// The integrator does not wait for the end of a doubling either: a complete doubling j consumes
// exactly 2^j uniforms after its direction draw (2^j - 1 inner merges, one outer merge;
// tree.ex:403, 1397, 1489), so it advances a copy of the tree's generator, knows the next
// direction, and has the first unit of the next doubling ready while the tree wave is still in
// the outer merge (wasted when the tree stops there). When the tree has stopped it draws the
// standard normals of the NEXT transition's momentum (sampler.ex:393-403; the generator state
// they start from is known: the post-momentum state of this transition advanced by the one
// uniform of sampler.ex:897) while the tree wave runs the adaptation update; the tree wave takes
// them if its generator stands where the draw started (not after a step-size search, which
// consumes draws in between).
// What either side decides (alive) is published in LDS before the barrier that precedes its use,
// double-buffered by the parity of the barrier count, so both waves always take the same number
// of barriers. The arithmetic of a leaf and of a merge is the one nuts_run performs itself:
// results are bit-identical to the one-wave kernel.
// Mailbox rows (each row = 64 doubles, one column per lane of the wave):
//   start  q0[DPL] p0[DPL] g0[DPL] im[DPL] eps jlp0 rng.a rng.b quit
//   ctrl   [parity][alive, go_right]
//   slot   [parity][2 leaves][q[DPL] p[DPL] g[DPL] logp lsw acc div]
//   next   z[DPL] rng_before.a .b rng_after.a .b   (the pre-drawn momentum variates)
// ------------------------------------------------------------------------------------------
struct NoPipe {
  static constexpr bool kOn = false;
};

// a finished subtree as the tree wave continues with it (nuts_run's c_* variables)
template <int DPL>
struct PipeUnit {
  double q[DPL], p[DPL], g[DPL];   // the outward endpoint
  double rho[DPL], pin[DPL], qp[DPL], gp[DPL];
  double logpP, lsw, acc;
  int n;
  bool div, turn;
};

template <int DPL>
struct PipeLeaf {
  double q[DPL], p[DPL], g[DPL];
  double logp, lsw, acc;
  bool div;
};

template <int DPL>
struct PipeBox {
  static constexpr bool kOn = true;
  static constexpr int kQuit = 4 * DPL + 4;   // start row: the tree wave abandons the kernel
  static constexpr int kCtrl = 4 * DPL + 5;
  static constexpr int kSlot = kCtrl + 4;
  static constexpr int kLeafRows = 3 * DPL + 4;
  static constexpr int kSlotRows = 2 * kLeafRows;
  static constexpr int kNext = kSlot + 2 * kSlotRows;
  static constexpr int kRows = kNext + DPL + 4;
  double* box;    // this lane's column
  int seq;        // barriers passed (identical on both waves)
  bool pending;   // tree wave: the barrier behind the integrator's momentum variates is still to come

  __device__ __forceinline__ double& row(int r) const { return box[(size_t)r * 64]; }
  __device__ __forceinline__ void sync() {
    __syncthreads();
    seq++;
  }
  // ---- tree wave ----
  __device__ __forceinline__ void put_start(const double (&q)[DPL], const double (&p)[DPL],
                                            const double (&g)[DPL], const double (&im)[DPL],
                                            double eps, double jlp0, const Rng& trng) const {
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      row(k) = q[k]; row(DPL + k) = p[k]; row(2 * DPL + k) = g[k]; row(3 * DPL + k) = im[k];
    }
    row(4 * DPL) = eps;
    row(4 * DPL + 1) = jlp0;
    row(4 * DPL + 2) = __longlong_as_double((long long)trng.a);
    row(4 * DPL + 3) = __longlong_as_double((long long)trng.b);
    row(kQuit) = 0.0;
  }
  // instead of a transition: tell the integrator wave to leave (it waits at the start barrier)
  __device__ __forceinline__ void quit() {
    if (pending) {
      sync();
      pending = false;
    }
    row(kQuit) = 1.0;
    sync();
  }
  __device__ __forceinline__ void put_ctrl(bool alive, bool go_right) const {
    const int b = kCtrl + 2 * ((seq + 1) & 1);
    row(b) = alive ? 1.0 : 0.0;
    row(b + 1) = go_right ? 1.0 : 0.0;
  }
  __device__ __forceinline__ void put_alive(bool alive) const {
    row(kCtrl + 2 * ((seq + 1) & 1)) = alive ? 1.0 : 0.0;
  }
  __device__ __forceinline__ void get_leaf(int which, PipeLeaf<DPL>& f) const {
    const int b = kSlot + kSlotRows * (seq & 1) + kLeafRows * which;
#pragma unroll
    for (int k = 0; k < DPL; k++) { f.q[k] = row(b + k); f.p[k] = row(b + DPL + k); f.g[k] = row(b + 2 * DPL + k); }
    f.logp = row(b + 3 * DPL);
    f.lsw = row(b + 3 * DPL + 1);
    f.acc = row(b + 3 * DPL + 2);
    f.div = row(b + 3 * DPL + 3) != 0.0;
  }
  // the pre-drawn variates of this transition's momentum, if the generator stands where the
  // integrator wave assumed it would; rng moves behind the draws
  __device__ __forceinline__ bool take_momentum(Rng& rng, double (&z)[DPL]) const {
    const uint64_t a = (uint64_t)__double_as_longlong(row(kNext + DPL));
    const uint64_t b = (uint64_t)__double_as_longlong(row(kNext + DPL + 1));
    const bool ok = (a == rng.a) && (b == rng.b);
    if (__any(ok ? 0 : 1) != 0) return false;
#pragma unroll
    for (int k = 0; k < DPL; k++) z[k] = row(kNext + k);
    rng.a = (uint64_t)__double_as_longlong(row(kNext + DPL + 2));
    rng.b = (uint64_t)__double_as_longlong(row(kNext + DPL + 3));
    return true;
  }
  __device__ __forceinline__ void clear_momentum() const {
    row(kNext + DPL) = __longlong_as_double(-1LL);   // no 58-bit state word looks like this
    row(kNext + DPL + 1) = __longlong_as_double(-1LL);
  }
  // ---- integrator wave ----
  __device__ __forceinline__ bool get_alive() const { return row(kCtrl + 2 * (seq & 1)) != 0.0; }
  __device__ __forceinline__ void put_leaf(int which, const PipeLeaf<DPL>& f) const {
    const int b = kSlot + kSlotRows * ((seq + 1) & 1) + kLeafRows * which;
#pragma unroll
    for (int k = 0; k < DPL; k++) { row(b + k) = f.q[k]; row(b + DPL + k) = f.p[k]; row(b + 2 * DPL + k) = f.g[k]; }
    row(b + 3 * DPL) = f.logp;
    row(b + 3 * DPL + 1) = f.lsw;
    row(b + 3 * DPL + 2) = f.acc;
    row(b + 3 * DPL + 3) = f.div ? 1.0 : 0.0;
  }
  __device__ __forceinline__ void put_momentum(const double (&z)[DPL], const Rng& before, const Rng& after) const {
#pragma unroll
    for (int k = 0; k < DPL; k++) row(kNext + k) = z[k];
    row(kNext + DPL) = __longlong_as_double((long long)before.a);
    row(kNext + DPL + 1) = __longlong_as_double((long long)before.b);
    row(kNext + DPL + 2) = __longlong_as_double((long long)after.a);
    row(kNext + DPL + 3) = __longlong_as_double((long long)after.b);
  }
};

template <int DPL>
__host__ __device__ constexpr size_t pipe_lds_doubles() { return (size_t)PipeBox<DPL>::kRows * 64; }

// The unit the tree wave ascends with, from the integrator's leaves: leaf a alone (doubling 0, or
// a diverged first leaf of a pair), or (a, b) merged at level 0 exactly as nuts_run's own ascent
// does at lvl 0 (tree.ex:1390-1476). q0 / g0: the state leaf a started from (a diverged leaf's
// proposal, tree.ex:1042-1048). The proposal draw comes from trng only when the pair is merged.
template <class M, int G>
__device__ __forceinline__ void pipe_pair_unit(const NutsLane<M, G>& L, const PipeLeaf<M::DPL>& a,
                                               const PipeLeaf<M::DPL>& b, bool pair,
                                               const double (&q0)[M::DPL], const double (&g0)[M::DPL],
                                               Rng& trng, PipeUnit<M::DPL>& u) {
  constexpr int DPL = M::DPL;
  using MM = Math<M::kVregMath>;
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    u.q[k] = a.q[k]; u.p[k] = a.p[k]; u.g[k] = a.g[k];
    u.rho[k] = a.p[k]; u.pin[k] = a.p[k];
    u.qp[k] = a.div ? q0[k] : a.q[k];
    u.gp[k] = a.div ? g0[k] : a.g[k];
  }
  u.logpP = a.div ? -1.0e30 : a.logp;
  u.lsw = a.lsw;
  u.acc = a.acc;
  u.n = 1;
  u.div = a.div;
  u.turn = false;
  if (pair && !a.div) {
    const double lsw = MM::log_sum_exp(a.lsw, b.lsw);
    const double uu = rng_uniform(trng);
    const bool use_b = uu < MM::exp_le0(b.lsw - lsw);   // lsw >= b.lsw
    bool turning = b.div;
    if (!turning) {
      double rho[DPL];
#pragma unroll
      for (int k = 0; k < DPL; k++) rho[k] = a.p[k] + b.p[k];
      turning = mass_uturn<M, G>(L, rho, a.p, b.p);
#pragma unroll
      for (int k = 0; k < DPL; k++) { u.rho[k] = rho[k]; u.pin[k] = a.p[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < DPL; k++) { u.rho[k] = b.p[k]; u.pin[k] = b.p[k]; }
    }
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      // proposal: b's own (its start, leaf a, when it diverged) or a's
      const double bq = b.div ? a.q[k] : b.q[k];
      const double bg = b.div ? a.g[k] : b.g[k];
      u.qp[k] = use_b ? bq : u.qp[k];
      u.gp[k] = use_b ? bg : u.gp[k];
      u.q[k] = b.q[k]; u.p[k] = b.p[k]; u.g[k] = b.g[k];
    }
    u.logpP = use_b ? (b.div ? -1.0e30 : b.logp) : u.logpP;
    u.lsw = lsw;
    u.acc = a.acc + b.acc;
    u.n = 2;
    u.div = b.div;
    u.turn = turning;
  }
}

// the integrator wave's side of one transition (mirror of nuts_run's skeleton)
template <class M, int G>
__device__ __forceinline__ bool pipe_integrate_transition(const typename M::Consts& mc,
                                                          const NutsLane<M, G>& L,
                                                          PipeBox<M::DPL>& pb) {
  constexpr int DPL = M::DPL;
  using MM = Math<M::kVregMath>;
  double q[DPL], p[DPL], g[DPL], qL[DPL], pL[DPL], gL[DPL], qR[DPL], pR[DPL], gR[DPL], im[DPL];
  EXMC_IPROF_DECL
  pb.sync();   // the tree wave has published the start of the transition
  EXMC_IPROF(2)
  if (__any(pb.row(PipeBox<DPL>::kQuit) != 0.0 ? 1 : 0) != 0) return false;
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    qL[k] = qR[k] = q[k] = pb.row(k);
    pL[k] = pR[k] = p[k] = pb.row(DPL + k);
    gL[k] = gR[k] = g[k] = pb.row(2 * DPL + k);
    im[k] = pb.row(3 * DPL + k);
  }
  const double eps = pb.row(4 * DPL);
  const double jlp0 = pb.row(4 * DPL + 1);
  Rng prng;   // copy of the tree's generator, used for the direction draws only
  prng.a = (uint64_t)__double_as_longlong(pb.row(4 * DPL + 2));
  prng.b = (uint64_t)__double_as_longlong(pb.row(4 * DPL + 3));
  Rng next_rng = prng;          // where the next transition's momentum draw starts: the
  rng_advance(next_rng);        // post-momentum state advanced by one uniform (sampler.ex:897)
  // one leapfrog from (q, p, g) in place and the leaf's scalars: batched_leapfrog.ex:79-85,
  // tree.ex:1042-1109, the same operations in the same order as nuts_run's own leaf pass
  auto leap = [&](double eps_dir, int which) {
    PipeLeaf<DPL> f;
    const double h = eps_dir / 2.0;
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const double ph = p[k] + h * g[k];
      p[k] = ph;
      q[k] = q[k] + eps_dir * (im[k] * ph);
    }
    f.logp = M::logp_grad(mc, L.ln, L.l, q, g);
#pragma unroll
    for (int k = 0; k < DPL; k++) p[k] = p[k] + h * g[k];
    const double jlp = f.logp - kinetic_energy<G, DPL, M::D, M::kXRowLds>(p, im, L.valid);
    // tree.ex:1042-1048 without a branch: a non-finite joint log-probability is a divergence with
    // log-weight -1001 and no acceptance; fmin(dl, 0) is 0 for a NaN, so the exponential is defined
    const bool fin = exmc_isfinite(jlp);
    const double dl = jlp - jlp0;
    f.div = fin ? (dl < -1000.0) : true;
    f.lsw = fin ? dl : -1001.0;
    f.acc = f.div ? 0.0 : fmin(1.0, MM::exp_le0(fmin(dl, 0.0)));
#pragma unroll
    for (int k = 0; k < DPL; k++) { f.q[k] = q[k]; f.p[k] = p[k]; f.g[k] = g[k]; }
    pb.put_leaf(which, f);
  };
  // the next unit into the slot: one leaf in doubling 0, afterwards two
  auto unit = [&](double eps_dir, bool pair) {
    leap(eps_dir, 0);
    if (pair) leap(eps_dir, 1);   // wave-uniform
    EXMC_IPROF_COUNT
  };
  bool go_right = rng_uniform(prng) > 0.5;   // tree.ex:403
  EXMC_IPROF(3)
  unit(go_right ? eps : -eps, false);        // the leaf of doubling 0
  EXMC_IPROF(1)
  for (int depth = 0;; depth++) {
    pb.sync();   // the tree wave says whether this doubling happens; its first unit is in the slot
    EXMC_IPROF(2)
    if (__any(pb.get_alive() ? 1 : 0) == 0) break;
    const double eps_dir = go_right ? eps : -eps;
    const int nunit = (depth == 0) ? 1 : (1 << (depth - 1));
    bool stopped = false;
    for (int i = 1; i < nunit; i++) {
      unit(eps_dir, true);
      EXMC_IPROF(1)
      pb.sync();   // unit i is in its slot; the tree wave is done with unit i - 1
      EXMC_IPROF(2)
      if (__any(pb.get_alive() ? 1 : 0) == 0) { stopped = true; break; }
    }
    if (stopped) continue;   // the next barrier carries alive = false
    // this doubling is complete on this side: the new endpoint, the next direction (2^depth
    // uniforms later in the tree's stream) and, ahead of the tree wave, the next first unit
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      if (go_right) { qR[k] = q[k]; pR[k] = p[k]; gR[k] = g[k]; }
      else { qL[k] = q[k]; pL[k] = p[k]; gL[k] = g[k]; }
    }
    const int nleaf = 1 << depth;
    for (int i = 0; i < nleaf; i++) rng_advance(prng);
    go_right = rng_uniform(prng) > 0.5;
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      q[k] = go_right ? qR[k] : qL[k];
      p[k] = go_right ? pR[k] : pL[k];
      g[k] = go_right ? gR[k] : gL[k];
    }
    EXMC_IPROF(3)
    unit(go_right ? eps : -eps, true);
    EXMC_IPROF(1)
  }
  // the tree has stopped: while the tree wave finishes the transition (adaptation update), draw
  // the variates of the next momentum; it reads them after the barrier that starts the next
  // transition's hand-shake (PipeBox::take_momentum)
  {
    double z[DPL];
    const Rng before = next_rng;
    draw_momentum_variates<M, G>(L, next_rng, z);
    pb.put_momentum(z, before, next_rng);
  }
  EXMC_IPROF(3)
  pb.sync();   // the variates are in the box
  EXMC_IPROF(2)
  EXMC_IPROF_FLUSH
  return true;
}

// n_draws NUTS transitions of this group's chain. sink(draw, q, logp, depth, n_steps, divergent,
// accept_sum, jlp0) is called once per finished transition.
// stack slots of one pending node: rho, p_in, p_out, q_prop, g_prop (DPL each), lsw, logp_prop, acc
//
// Lock-step schedule: the chain groups of a wavefront start every transition together and walk
// the same tree skeleton (doubling j, leaf k) in the same pass, so which levels merge after a
// leaf (the trailing ones of k), the doubling and transition boundaries and the trace writes are
// wave-uniform scalar control flow; per group only the direction, the proposal choices and the
// stop flags differ. A group whose tree has ended (or ended early inside a subtree, the rare
// tree.ex:1175-1177 path, handled by the same level loop) idles until the deepest tree of the
// wavefront is finished: for eight_schools E[max of 4 trees] / E[tree] = 1.18 extra passes buy a
// pass that costs ~0.6x of the asynchronous one (no union of divergent paths, no exec juggling).
// idle(): work of the caller that does not depend on this transition's outcome; called once per
// transition where the pipelined tree wave would otherwise wait for the first leaf.
struct NoIdle {
  __device__ __forceinline__ void operator()() const {}
};

template <class M, int G, int LDSL, class Sink, class Pipe = NoPipe, class Idle = NoIdle>
__device__ __forceinline__ void nuts_run(const typename M::Consts& mc, const NutsLane<M, G>& L,
                                         ChainRegs<M::DPL>& st, int n_draws, double eps,
                                         int max_depth, Sink&& sink, Pipe* pipe = nullptr,
                                         Idle&& idle = Idle{}) {
  constexpr int DPL = M::DPL;
  constexpr int NSLOT = nuts_nslot<M>();
  // M::kRegradProposal: the proposal travels as (q, logp) only and its gradient is evaluated once
  // more when the transition is over -- the same function of the same bits, so the same gradient --
  // instead of riding in every node, every park and every merge's selects (sv: one evaluation per
  // ~340 leapfrogs against 2 of 13 doubles of every node and 12 vector registers of the leaf loop)
  constexpr bool kRegrad = M::kRegradProposal;
  constexpr int kNodeScalars = NSLOT - 3;   // offset of lsw, logp_prop, acc
  using MM = Math<M::kVregMath>;
  const int l = L.l;
  const auto& im = L.im;
  double* lstk = L.lstk;
  double* gstk = L.gstk;

  double q[DPL], p[DPL], g[DPL], qold[DPL], gold[DPL];
  double qL[DPL], pL[DPL], gL[DPL], qR[DPL], pR[DPL], gR[DPL];
  double t_rho[DPL], t_qp[DPL], t_gp[DPL];
  double c_rho[DPL], c_pin[DPL], c_qp[DPL], c_gp[DPL];
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    q[k] = p[k] = g[k] = qold[k] = gold[k] = 0.0;
    qL[k] = pL[k] = gL[k] = qR[k] = pR[k] = gR[k] = 0.0;
    t_rho[k] = t_qp[k] = t_gp[k] = c_rho[k] = c_pin[k] = c_qp[k] = c_gp[k] = 0.0;
  }
  double t_logpP = 0.0, t_lsw = 0.0, t_acc = 0.0, jlp0 = 0.0;
  int t_n = 0, t_depth = 0;
  bool t_div = false, t_turn = false, go_right = true;
  double eps_dir = eps;
  Rng trng = st.rng;
  const bool has = L.alive;   // false: a lane group kept only for wave-cooperative models

  EXMC_PROF_DECL
  for (int draw = 0; draw < n_draws; draw++) {
    EXMC_PROF(8)
    // ---- transition start (sampler.ex:393-403, 890-899) ----
    bool alive = has;
    bool have_z = false;
    double z_pre[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) z_pre[k] = 0.0;
    if constexpr (Pipe::kOn) {
      // the integrator wave drew the variates of this momentum while the previous transition was
      // being finished here (valid if the generator has not moved since, see PipeBox)
      if (pipe->pending) {
        pipe->sync();
        pipe->pending = false;
        have_z = pipe->take_momentum(st.rng, z_pre);
      }
    }
    if (has) {
      if (have_z) momentum_from_variates<M, G>(L, z_pre, pL);
      else draw_momentum<M, G>(L, st.rng, pL);
      jlp0 = st.logp - mass_ke<M, G>(L, pL);
      trng = st.rng;  // the tree consumes a copy (sampler.ex:897 discards its draws)
#pragma unroll
      for (int k = 0; k < DPL; k++) {
        qL[k] = qR[k] = t_qp[k] = st.q[k];
        gL[k] = gR[k] = t_gp[k] = st.g[k];
        pR[k] = t_rho[k] = pL[k];
      }
      t_logpP = st.logp;
      t_lsw = 0.0;
      t_acc = 0.0;
      t_n = 0;
      t_div = t_turn = false;
      t_depth = 0;
    }
    if constexpr (Pipe::kOn) {
      pipe->put_start(st.q, pL, st.g, im, eps, jlp0, trng);
      pipe->sync();
    }
    idle();
    EXMC_PROF(0)

    for (int depth = 0; Pipe::kOn || __any(alive ? 1 : 0) != 0; depth++) {   // depth is wave-uniform
      if (alive) {
        // tree.ex:403-413 direction + outward endpoint
        const double u = rng_uniform(trng);
        go_right = u > 0.5;
        eps_dir = go_right ? eps : -eps;
#pragma unroll
        for (int k = 0; k < DPL; k++) {
          q[k] = go_right ? qR[k] : qL[k];
          p[k] = go_right ? pR[k] : pL[k];
          g[k] = go_right ? gR[k] : gL[k];
        }
      }
      if constexpr (Pipe::kOn) {
        pipe->put_ctrl(alive, go_right);
        pipe->sync();
        if (__any(alive ? 1 : 0) == 0) break;
      }
      EXMC_PROF(1)

      // one-wave form: a pass per leaf, ascent from level 0. Pipelined form: a pass per unit of the
      // integrator wave (a leaf in doubling 0, afterwards a leaf pair merged at level 0), so loop
      // level lvl is tree level lvl + kLvl0
      constexpr int kLvl0 = Pipe::kOn ? 1 : 0;
      const int nlev = (Pipe::kOn && depth > 0) ? depth - 1 : depth;
      const int nleaf = 1 << nlev;
      for (int leaf = 0; leaf < nleaf; leaf++) {
        if constexpr (M::kNutsWavesPerSimd == 2 && !Pipe::kOn) {
          // Two waves share the SIMD and its arbiter serves the OLDER one first: left alone, the
          // older wave runs at the speed of a lone wave (sv: 291 leapfrogs/ms), the younger one on
          // what is left (169), so half the chains end after 52 % of the launch and the other
          // half then run alone at 64 % of the pair's throughput (profiles/r3_sv_prio). The
          // priority is traded in sixteenths of a 2^16-clock period (27 us): slot 0 holds it for
          // prio_duty sixteenths, slot 1 for the rest -- even shares by default, the larger share to
          // the chain with more left to do where the kernel knows it (nuts_kernel, migration loop).
          if (L.prio_slot >= 0) {
            const int phase = (int)((clock64() >> 12) & 15);
            const bool mine = (phase < L.prio_duty) == (L.prio_slot == 0);
            if (__builtin_amdgcn_readfirstlane(mine ? 1 : 0)) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(0);
          }
        }
        if constexpr (Pipe::kOn) {
          if (leaf > 0) pipe->sync();   // unit 0 arrived with the doubling's barrier
        }
        if (leaf > 0 && __any(alive ? 1 : 0) == 0) break;
        EXMC_PROF_COUNT(9)
        double logp_new = 0.0, jlp = 0.0;
        if constexpr (!Pipe::kOn) {
          // ---- one leapfrog on every lane (batched_leapfrog.ex:79-85); idle groups integrate
          // scratch registers so that wave-cooperative models see all 64 lanes ----
          const double h = eps_dir / 2.0;
#pragma unroll
          for (int k = 0; k < DPL; k++) {
            qold[k] = q[k];
            gold[k] = g[k];
            p[k] = p[k] + h * g[k];
          }
          mass_drift<M, G>(L, eps_dir, p, q);
#if EXMC_ABLATE == 3
          logp_new = 0.0;
#pragma unroll
          for (int k = 0; k < DPL; k++) { g[k] = -q[k]; logp_new = logp_new - 0.5 * q[k] * q[k]; }
          logp_new = group_allsum<G>(logp_new);
#else
          logp_new = M::logp_grad(mc, L.ln, l, q, g);
#endif
#pragma unroll
          for (int k = 0; k < DPL; k++) p[k] = p[k] + h * g[k];
          jlp = logp_new - mass_ke<M, G>(L, p);
        }
        EXMC_PROF(2)

        if (alive) {
          // ---- leaf (tree.ex:1042-1109) ----
          bool c_div, c_turn = false;
          double c_lsw, c_acc, c_logpP;
          int c_n = 1;
          if constexpr (Pipe::kOn) {
            // the integrator wave computed this unit while the previous one was merged here
            PipeUnit<DPL> pu;
            {
              PipeLeaf<DPL> la, lb;
              pipe->get_leaf(0, la);
              if (depth > 0) pipe->get_leaf(1, lb);
              else lb = la;
              pipe_pair_unit<M, G>(L, la, lb, depth > 0, q, g, trng, pu);
            }
#pragma unroll
            for (int k = 0; k < DPL; k++) {
              q[k] = pu.q[k]; p[k] = pu.p[k]; g[k] = pu.g[k];
              c_rho[k] = pu.rho[k]; c_pin[k] = pu.pin[k]; c_qp[k] = pu.qp[k]; c_gp[k] = pu.gp[k];
            }
            c_div = pu.div;
            c_turn = pu.turn;
            c_lsw = pu.lsw;
            c_acc = pu.acc;
            c_logpP = pu.logpP;
            c_n = pu.n;
          } else {
            // tree.ex:1042-1048 without a branch (see pipe_integrate_transition)
            const bool fin = exmc_isfinite(jlp);
            const double dl = jlp - jlp0;
            c_div = fin ? (dl < -1000.0) : true;
            c_lsw = fin ? dl : -1001.0;
#if EXMC_ABLATE == 4
            c_acc = fmin(1.0, 1.0 + fmin(dl, 0.0));
#else
            c_acc = fmin(1.0, MM::exp_le0(fmin(dl, 0.0)));
#endif
            c_acc = c_div ? 0.0 : c_acc;
            c_logpP = c_div ? -1.0e30 : logp_new;
#pragma unroll
            for (int k = 0; k < DPL; k++) {
              c_qp[k] = c_div ? qold[k] : q[k];
              if constexpr (!kRegrad) c_gp[k] = c_div ? gold[k] : g[k];
              c_rho[k] = p[k];
              c_pin[k] = p[k];
            }
          }
          EXMC_PROF(3)

          // ---- ascend (tree.ex:1144-1203, 1390-1476): level lvl holds a pending first half iff
          // bit lvl of `leaf` is set; the first clear bit is where an unfinished node parks ----
          bool parked = false;
          for (int lvl = 0; lvl < nlev; lvl++) {
            if ((leaf >> lvl) & 1) {
              if (!parked) {
                double nd[NSLOT];
                if (lvl < LDSL) node_load_lds<NSLOT>(lstk + (size_t)lvl * NSLOT * kNutsBlock, kNutsBlock, nd);
                else node_load<NSLOT>(gstk + (size_t)(lvl - LDSL) * NSLOT * kNutsBlock, kNutsBlock, nd);
                const double a_lsw = nd[kNodeScalars + 0];
                const double a_logpP = nd[kNodeScalars + 1];
                const double a_acc = nd[kNodeScalars + 2];
#if EXMC_ABLATE == 2
                const double lsw = a_lsw + c_lsw;
                const double u = rng_uniform(trng);
                const bool use_b = u < 0.5;
#else
                const double lsw = MM::log_sum_exp(a_lsw, c_lsw);
                const double u = rng_uniform(trng);
                const bool use_b = u < MM::exp_le0(c_lsw - lsw);   // lsw >= c_lsw
#endif
                if (!use_b) {
                  c_logpP = a_logpP;
#pragma unroll
                  for (int k = 0; k < DPL; k++) {
                    c_qp[k] = nd[3 * DPL + k];
                    if constexpr (!kRegrad) c_gp[k] = nd[4 * DPL + k];
                  }
                }
                bool turning = c_div || c_turn;
                if (!turning) {
                  double rho[DPL], r2[DPL], r3[DPL], a_pin[DPL], a_pout[DPL];
#pragma unroll
                  for (int k = 0; k < DPL; k++) {
                    a_pin[k] = nd[1 * DPL + k];
                    a_pout[k] = nd[2 * DPL + k];
                    rho[k] = nd[0 * DPL + k] + c_rho[k];
                    r2[k] = nd[0 * DPL + k] + c_pin[k];
                    r3[k] = a_pout[k] + c_rho[k];
                  }
                  bool c1, c23 = false;
#if EXMC_ABLATE == 1
                  c1 = false;
#else
                  // tree.ex:1428-1446: the two sub-span checks apply from depth 2 on; lvl is
                  // wave-uniform, so the level-0 merges (3 of 4 in a 7-leaf tree) reduce 2 sums, not 6
                  if (lvl + kLvl0 == 0) c1 = mass_uturn<M, G>(L, rho, a_pin, p);
                  else mass_uturn3<M, G>(L, rho, a_pin, p, r2, a_pin, c_pin, r3, a_pout, p, c1, c23);
#endif
                  turning = c1 || c23;
#pragma unroll
                  for (int k = 0; k < DPL; k++) { c_rho[k] = rho[k]; c_pin[k] = a_pin[k]; }
                }
                c_lsw = lsw;
                c_acc = a_acc + c_acc;
                c_n = (1 << (lvl + kLvl0)) + c_n;
                c_turn = turning;
              }
            } else {
              // a node that is divergent or turning is returned upward unmerged (tree.ex:1175-1177)
              if (!parked && !(c_div || c_turn)) {
                double nd[NSLOT];
#pragma unroll
                for (int k = 0; k < DPL; k++) {
                  nd[0 * DPL + k] = c_rho[k];
                  nd[1 * DPL + k] = c_pin[k];
                  nd[2 * DPL + k] = p[k];
                  nd[3 * DPL + k] = c_qp[k];
                  if constexpr (!kRegrad) nd[4 * DPL + k] = c_gp[k];
                }
                nd[kNodeScalars + 0] = c_lsw;
                nd[kNodeScalars + 1] = c_logpP;
                nd[kNodeScalars + 2] = c_acc;
                if (lvl < LDSL) node_store_lds<NSLOT>(lstk + (size_t)lvl * NSLOT * kNutsBlock, kNutsBlock, nd);
                else node_store<NSLOT>(gstk + (size_t)(lvl - LDSL) * NSLOT * kNutsBlock, kNutsBlock, nd);
                parked = true;
              }
              if (__any(parked ? 0 : 1) == 0) break;   // every live group has parked
            }
          }
          EXMC_PROF(4)

          // ---- this doubling's subtree is complete (last leaf, or ended early):
          // merge_trajectories (tree.ex:1479-1568) ----
          if (!parked) {
#if EXMC_ABLATE == 2
            const double lsw = t_lsw + c_lsw;
            const double u = rng_uniform(trng);
            const bool use_sub = u < 0.5;
#else
            const double lsw = MM::log_sum_exp(t_lsw, c_lsw);
            const double u = rng_uniform(trng);
            const bool use_sub = MM::log_unit(u) < (c_lsw - t_lsw);
#endif
            if (use_sub) {
              t_logpP = c_logpP;
#pragma unroll
              for (int k = 0; k < DPL; k++) {
                t_qp[k] = c_qp[k];
                if constexpr (!kRegrad) t_gp[k] = c_gp[k];
              }
            }
            const bool divg = t_div || c_div;
            bool turning = divg || c_turn;
            double rho[DPL];
#pragma unroll
            for (int k = 0; k < DPL; k++) rho[k] = t_rho[k] + c_rho[k];
            if (!turning) {
              double nearp[DPL], farp[DPL], r2[DPL], r3[DPL];
#pragma unroll
              for (int k = 0; k < DPL; k++) {
                nearp[k] = go_right ? pR[k] : pL[k];
                farp[k] = go_right ? pL[k] : pR[k];
                r2[k] = t_rho[k] + c_pin[k];
                r3[k] = nearp[k] + c_rho[k];
              }
              bool c1, c23;
#if EXMC_ABLATE == 1
              c1 = c23 = false;
#else
              mass_uturn3<M, G>(L, rho, farp, p, r2, farp, c_pin, r3, nearp, p, c1, c23);
#endif
              turning = c1 || c23;
            }
#pragma unroll
            for (int k = 0; k < DPL; k++) {
              t_rho[k] = rho[k];
              if (go_right) { qR[k] = q[k]; pR[k] = p[k]; gR[k] = g[k]; }
              else { qL[k] = q[k]; pL[k] = p[k]; gL[k] = g[k]; }
            }
            t_lsw = lsw;
            t_n += c_n;
            t_acc = t_acc + c_acc;
            t_div = divg;
            t_turn = turning;
            t_depth = depth + 1;
            // tree.ex:1607-1618: the trajectory stops at max depth, on divergence or on a U-turn
            alive = !(t_depth >= max_depth || t_div || t_turn);
          }
          EXMC_PROF(5)
        }
        if constexpr (Pipe::kOn) pipe->put_alive(alive);
      }
    }

    // ---- transition done for every group of the wavefront (sampler.ex:890-925) ----
    if constexpr (Pipe::kOn) pipe->pending = true;   // the integrator wave is drawing the next momentum
    if constexpr (kRegrad) {
      // every lane of the wavefront takes part (the model's sums span the lane group); what the
      // function returns is the proposal's log-density again, already held in t_logpP
      (void)M::logp_grad(mc, L.ln, l, t_qp, t_gp);
    }
    if (has) {
      (void)rng_uniform(st.rng);
      st.logp = t_logpP;
#pragma unroll
      for (int k = 0; k < DPL; k++) { st.q[k] = t_qp[k]; st.g[k] = t_gp[k]; }
      sink(draw, st.q, st.logp, t_depth, t_n, t_div, t_acc, jlp0);
    }
    EXMC_PROF(6)
  }
  EXMC_PROF_FLUSH
}

// this lane's row of cov and (negated, below the diagonal) column of chol, both row-major [D][D]
// with `stride` doubles between consecutive entries (1: global arrays); null = identity
template <class M, int G>
__device__ __forceinline__ void rowdense_load(NutsLane<M, G>& L, const double* cov, const double* chol,
                                              int stride) {
  constexpr int D = M::D;
  const int i = L.l < D ? L.l : 0;   // lanes past D carry no dimension; any row keeps them finite
#pragma unroll
  for (int j = 0; j < (M::kRowDense ? D : 1); j++) {
    L.rd.cr[j] = cov ? cov[(size_t)(i * D + j) * stride] : ((i == j) ? 1.0 : 0.0);
    L.rd.ncc[j] = (chol && j > i) ? -chol[(size_t)(j * D + i) * stride] : 0.0;
  }
  L.rd.diag = chol ? chol[(size_t)(i * D + i) * stride] : 1.0;
}

// fill the per-lane constants; inv_mass / sqrt_inv_mass may be null (identity mass)
template <class M, int G, int LDSL, bool kZig = true>
__device__ __forceinline__ void lane_setup(NutsLane<M, G>& L, const typename M::Consts& mc,
                                           double* lds, double* stack, const double* inv_mass,
                                           const double* sqrt_inv_mass, const ZigTables& zt,
                                           double nor_r, const FlatOrder& flat = FlatOrder{},
                                           const DenseMass& dm = DenseMass{}) {
  constexpr int D = M::D, DPL = M::DPL;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  L.l = threadIdx.x & (G - 1);
  L.lstk = lds + threadIdx.x;
  // The spill levels of one wavefront are one contiguous block [level][slot][lane]: a node's
  // slots are 512 bytes apart, so its loads and stores share one address register and differ in
  // their immediate offsets (the chain-interleaved layout of rounds 1-3 paid a 64-bit address
  // computation per slot, 27 of the ~500 instructions of a spilled merge).
  L.gstk = stack + (size_t)(tid / kNutsBlock) * nuts_spill_doubles<M, LDSL>() + (tid % kNutsBlock);
  L.zt = zt;
  L.nor_r = nor_r;
  L.alive = true;
  L.prio_slot = -1;
  L.prio_duty = 8;
  M::load(mc, L.l, L.ln);
  if constexpr (M::kExtraLdsDoubles > 0)
    L.ln.sh = lds + (size_t)LDSL * nuts_nslot<M>() * kNutsBlock + nuts_zig_doubles<kZig>();
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = L.l + k * G;
    L.valid[k] = i < D;
    L.im[k] = (L.valid[k] && inv_mass) ? inv_mass[i] : 1.0;
    L.sim[k] = (L.valid[k] && sqrt_inv_mass) ? sqrt_inv_mass[i] : 1.0;
    L.rank[k] = L.valid[k] ? (flat.rank ? flat.rank[i] : i) : D;
  }
  L.perm = flat.perm;
  L.dm = dm;
  if constexpr (M::kRowDense) rowdense_load<M, G>(L, dm.cov, dm.chol, 1);
  if constexpr (M::kLaneDense) {
    L.ld.covp = dm.covp;
    L.ld.cholp = dm.cholp;
    double* dl = lds + (size_t)LDSL * nuts_nslot<M>() * kNutsBlock + nuts_zig_doubles<kZig>() + M::kExtraLdsDoubles;
    L.ld.xs = dl + (size_t)((threadIdx.x & 63) / G) * 3 * D;
    if constexpr (M::kDenseImage) {
      L.ld.covl = dl + (size_t)(64 / G) * 3 * D;
      if (dm.covp) {
        lane_dense_stage<G, DPL, D>(L.ld, L.l);
        wave_lds_fence();
      }
    }
  }
}

template <class M, int G>
__device__ __forceinline__ void chain_load(const ChainState& s, int C, int chain, int l,
                                           ChainRegs<M::DPL>& st) {
  constexpr int D = M::D, DPL = M::DPL;
  st.logp = s.logp[chain];
  st.rng.a = s.rng[chain];
  st.rng.b = s.rng[(size_t)C + chain];
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = l + k * G;
    st.q[k] = (i < D) ? s.q[(size_t)i * C + chain] : 0.0;
    st.g[k] = (i < D) ? s.g[(size_t)i * C + chain] : 0.0;
  }
}

template <class M, int G>
__device__ __forceinline__ void chain_store(const ChainState& s, int C, int chain, int l,
                                            const ChainRegs<M::DPL>& st) {
  constexpr int D = M::D, DPL = M::DPL;
  if (l == 0) {
    s.logp[chain] = st.logp;
    s.rng[chain] = st.rng.a;
    s.rng[(size_t)C + chain] = st.rng.b;
  }
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = l + k * G;
    if (i < D) {
      s.q[(size_t)i * C + chain] = st.q[k];
      s.g[(size_t)i * C + chain] = st.g[k];
    }
  }
}

// kPipe: every workgroup is a pair of waves over the same 64/G chains -- wave 0 keeps the trees,
// wave 1 integrates one leaf ahead (PipeBox above). With two waves per SIMD the barrier waits and
// the bubbles of one wave (taken branches, > 4-clock issues, LDS round trips) are the other's
// issue slots.
// kStream: the trace lives in page-locked host memory and every finished draw is published there
// (sample_stream's per-draw notification, sampler.ex:1186-1277): after the draw's stores a
// system-scope release, then the count of finished draws of chain 0 in P.progress.
template <class M, int G, int LDSL, bool kPipe = false, bool kStream = false>
__global__ void __launch_bounds__(kPipe ? 2 * kNutsBlock : kNutsBlock, M::kNutsWavesPerSimd)
    nuts_kernel(NutsParams P, typename M::Consts mc) {
  constexpr int D = M::D, DPL = M::DPL;
  constexpr int NSLOT = nuts_nslot<M>();
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const int tid = blockIdx.x * kNutsBlock + lane;   // both waves of a pair see the same chains
  const int C = P.n_chains;
  const bool has_chain = (tid / G) < C;
  const int chain = has_chain ? (tid / G) : (C - 1);   // surplus groups shadow the last chain
  constexpr bool kZig = M::kNutsZigInLds || kPipe;
  const ZigTables zt = stage_zig_tables<LDSL, NSLOT, kZig>(lds, P.zig_ki, P.zig_wi, P.zig_fi);
  const int xoff = stage_model_data<M, G, LDSL, kZig>(mc, lds);
  if (!M::kCoop && !has_chain) return;

  NutsLane<M, G> L;
  lane_setup<M, G, LDSL, kZig>(L, mc, lds, P.stack, P.inv_mass, P.sqrt_inv_mass, zt, P.nor_r, P.flat, P.dm);
  if constexpr (M::kLdsDataDoubles > 0) L.ln.xoff = xoff;
  L.alive = has_chain;
  // workgroups i and i + simds USUALLY land on the same SIMD (one wave each, dispatch in order:
  // tools/sv_probe.sh), the second half being the younger waves: good enough for the even shares of
  // launches without a migration board (short launches, logistic); the migration loop below finds
  // the real partner instead (tools/r4_sv_pairs.sh). Any grid other than exactly two waves per SIMD of
  // the whole device keeps the arbiter's order.
  if constexpr (M::kNutsWavesPerSimd == 2 && !kPipe)
    L.prio_slot = (P.prio && P.simds > 0 && (int)gridDim.x == 2 * P.simds) ? (int)(blockIdx.x >= (unsigned)P.simds) : -1;
  using Pipe = std::conditional_t<kPipe, PipeBox<DPL>, NoPipe>;
  Pipe pipe;
  if constexpr (kPipe) {
    L.lstk = lds + lane;
    L.gstk = P.stack + (size_t)blockIdx.x * nuts_spill_doubles<M, LDSL>() + lane;
    pipe.box = lds + nuts_lds_bytes<M, LDSL>() / 8 + lane;
    pipe.seq = 0;
    pipe.pending = false;
    // The dispatcher puts the two waves of a workgroup on different SIMDs and, with four workgroups
    // per CU, a tree wave and an integrator wave of different workgroups on (nearly) every SIMD
    // (1021 of 1023, tools/es_pipe_probe.sh). The tree wave -- the transition's critical path --
    // takes the issue priority: it runs as if alone on its SIMD, the integrator on what is left.
    const bool integrator = threadIdx.x >= kNutsBlock;
#ifdef EXMC_XCC_PROBE
    if (lane == 0 && 2 * blockIdx.x + (threadIdx.x >> 6) < 4096) {
      unsigned xcc, hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      double* const e = g_wave_probe + (2 * blockIdx.x + (threadIdx.x >> 6)) * 5;
      e[0] = (double)((xcc & 0xf) * 10000 + ((hw >> 8) & 0xf) * 100 + ((hw >> 13) & 0x7) * 10 + ((hw >> 4) & 0x3));
      e[1] = integrator ? 1.0 : 0.0;
      e[3] = (double)wall_clock64();
    }
#endif
    if (P.prio && !integrator) __builtin_amdgcn_s_setprio(3);
    if (integrator) {
      for (int i = 0; i < P.n_draws; i++)
        if (!pipe_integrate_transition<M, G>(mc, L, pipe)) break;
#ifdef EXMC_XCC_PROBE
      if (lane == 0 && 2 * blockIdx.x + (threadIdx.x >> 6) < 4096)
        g_wave_probe[(2 * blockIdx.x + (threadIdx.x >> 6)) * 5 + 4] = (double)wall_clock64();
#endif
      return;
    }
  }
  ChainRegs<DPL> st;
  chain_load<M, G>(P.st, C, chain, L.l, st);

  unsigned long long lf_total = 0, div_total = 0;
  int n_published = 0;
  const int l = L.l;
  // Trace cursors: one per-lane pointer per output, stepped by one trace row per draw. An output
  // the caller left null (or that this lane does not own) points at an 8-byte scratch word with
  // step 0, so the per-draw code is straight stores and 64-bit adds in vector registers: no
  // null tests, no row * D * C address products and no kernel-argument pointers kept live in
  // scalar registers across the sampling loop.
  struct Cursor {
    char* p;
    long long step;
  };
  auto cursor = [&](void* base, bool mine, size_t elem, size_t index, size_t row_elems) -> Cursor {
    const bool on = (base != nullptr) && mine;
    return Cursor{on ? (char*)base + index * elem : (char*)P.scratch,
                  on ? (long long)(row_elems * elem) : 0LL};
  };
  const bool own = (l == 0);
  Cursor c_draw[DPL];
  Cursor c_logp, c_depth, c_steps, c_div, c_acc, c_energy;
  // point the cursors at trace row `row` of chain `ch`
  auto bind = [&](int ch, size_t row) {
    const size_t stat0 = row * C + ch;
#pragma unroll
    for (int k = 0; k < DPL; k++)
      c_draw[k] = cursor(P.tr.draws, L.valid[k], 8, (row * D + (l + k * G)) * C + ch, (size_t)D * C);
    c_logp = cursor(P.tr.logp, own, 8, stat0, C);
    c_depth = cursor(P.tr.tree_depth, own, 4, stat0, C);
    c_steps = cursor(P.tr.n_steps, own, 4, stat0, C);
    c_div = cursor(P.tr.divergent, own, 4, stat0, C);
    c_acc = cursor(P.tr.accept_prob, own, 8, stat0, C);
    c_energy = cursor(P.tr.energy, own, 8, stat0, C);
  };
  bind(chain, (size_t)P.draw_offset);
  auto sink = [&](int, const double (&sq)[DPL], double slogp, int depth, int t_n, bool t_div,
                  double t_acc, double jlp0) {
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      if (L.valid[k]) *(double*)c_draw[k].p = sq[k];
      c_draw[k].p += c_draw[k].step;
    }
    if (own) {
      *(double*)c_logp.p = slogp;
      *(int32_t*)c_depth.p = depth;
      *(int32_t*)c_steps.p = t_n;
      *(int32_t*)c_div.p = t_div ? 1 : 0;
      *(double*)c_acc.p = (t_n > 0) ? (t_acc / (double)t_n) : 0.0;
      *(double*)c_energy.p = -jlp0;
    }
    c_logp.p += c_logp.step;
    c_depth.p += c_depth.step;
    c_steps.p += c_steps.step;
    c_div.p += c_div.step;
    c_acc.p += c_acc.step;
    c_energy.p += c_energy.step;
    lf_total += (unsigned long long)t_n;
    div_total += t_div ? 1u : 0u;
    if constexpr (kStream) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: the host may read this draw's row
      n_published++;
      if (tid == 0) __hip_atomic_store(P.progress, n_published, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };
#ifdef EXMC_XCC_PROBE
  const long long wave_c0 = clock64(), wave_w0 = wall_clock64();
  auto probe_out = [&]() {
    if constexpr (kPipe) {   // the pair's entries were opened at the start of the kernel
      if (lane == 0 && 2 * blockIdx.x + (threadIdx.x >> 6) < 4096) {
        double* const e = g_wave_probe + (2 * blockIdx.x + (threadIdx.x >> 6)) * 5;
        e[2] = (double)lf_total;
        e[4] = (double)wall_clock64();
      }
      return;
    }
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) {
      unsigned xcc, hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      g_wave_probe[blockIdx.x * 5 + 0] = (double)((xcc & 0xf) * 10000 + ((hw >> 8) & 0xf) * 100 + ((hw >> 13) & 0x7) * 10 + ((hw >> 4) & 0x3));
      g_wave_probe[blockIdx.x * 5 + 1] = (double)(clock64() - wave_c0);
      g_wave_probe[blockIdx.x * 5 + 2] = (double)lf_total;
      g_wave_probe[blockIdx.x * 5 + 3] = (double)wave_w0;
      g_wave_probe[blockIdx.x * 5 + 4] = (double)wall_clock64();
    }
  };
#endif
  if constexpr (kPipe) {
    nuts_run<M, G, LDSL>(mc, L, st, P.n_draws, P.eps, P.max_depth, sink, &pipe);
    if (pipe.pending) pipe.sync();   // the integrator wave's last barrier (momentum variates nobody takes)
  } else if constexpr (M::kMigrate && G == 64) {
    if (P.mig != nullptr && P.n_draws > 0) {
      // chain migration (see MigBoard above); everything here is wave-uniform
      int* const board = P.mig;
      const int simd_uid = mig_simd_uid();
      int* const live = board + kMigLive + simd_uid;
      int* const mail = board + kMigMail + 2 * (int)blockIdx.x;
      int cur = chain, done = 0;
      // Who shares this SIMD is found out, not assumed. In a launch of exactly two waves per SIMD the
      // dispatcher USUALLY puts workgroups b and b + simds on one SIMD (1024 pairs of 1024 in 31 of 32
      // probed launches) -- and sometimes does not (25 of 1024, tools/r4_sv_pairs.sh), and then a
      // priority schedule negotiated between strangers costs the launch 3 to 7 %: the slow state of
      // round 3's and round 4's driver-command runs. So every wave enters itself in its SIMD's
      // occupant list in arrival order; the first to arrive is slot 0 (it is the older wave, the one
      // the arbiter prefers when priorities are equal), and the partner is whoever holds the other slot.
      int arrival = 0;
      if (lane == 0) arrival = atomicAdd(live, 1);
      arrival = __builtin_amdgcn_readfirstlane(arrival);
      int* const occupants = board + kMigMail + 3 * (int)gridDim.x + 2 * simd_uid;
      const bool two_per_simd = P.prio && P.simds > 0 && (int)gridDim.x == 2 * P.simds;
      const bool listed = two_per_simd && arrival < 2;
      int partner = -1;
      if (listed) {
        if (lane == 0) __hip_atomic_store(occupants + arrival, (int)blockIdx.x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the other occupant enters itself within microseconds of this one (all waves of the launch are
        // dispatched at once); a wave that is alone on its SIMD gives up after ~0.3 ms
        int o = 0;
        for (int spin = 0; spin < 256 && o == 0; spin++) {
          if (lane == 0) o = mig_load(occupants + (1 - arrival));
          o = __builtin_amdgcn_readfirstlane(o);
          if (o == 0) __builtin_amdgcn_s_sleep(32);
        }
        partner = o - 1;
      }
      // slot 0 is the OLDER wave of the two -- the one dispatched first, the lower workgroup index:
      // the arbiter prefers it whenever the two priorities are equal (around every change of the time
      // slice), and the measured rates the shares are computed from are those of that wave
      L.prio_slot = (partner >= 0) ? (((int)blockIdx.x < partner) ? 0 : 1) : -1;
      // what is left of the two chains of this SIMD: the shares of the issue priority follow it, so
      // that both end together instead of the longer one running on alone. remaining = leapfrogs per
      // draw so far x draws to go, known to the partner one transition late; with the pair's rates
      // (sv: 291 with the priority, 169 without, of 460) the share f of slot 0 that makes the rates
      // proportional to rho = left_0 / left_1 is (291 rho - 169) / (122 (1 + rho)).
      int* const pair_word = board + kMigMail + 2 * (int)gridDim.x;
      unsigned long long lf_at_start = 0;
      int done_at_start = 0;
      for (;;) {
        // (a wave that became a host has its SIMD to itself: prio_slot -1 from then on)
        const bool paired = L.prio_slot >= 0;
        // one transition at a time; the words looked at after it are loaded before it
        int host_adv = 0, n_live = 0, partner_left = 0;
        if (lane == 0) {
          host_adv = mig_load(board + 1);
          n_live = mig_load(live);
          if (paired) partner_left = mig_load(pair_word + partner);
        }
        nuts_run<M, G, LDSL>(mc, L, st, 1, P.eps, P.max_depth, sink);
        done++;
        if (paired) {
          const int seen = done - done_at_start;
          const double per_draw = (seen >= 4) ? (double)(lf_total - lf_at_start) / (double)seen : 0.0;
          const int my_left = 1 + (int)(per_draw * (double)(P.n_draws - done) * 1.0e-3);
          if (lane == 0 && seen >= 4) __hip_atomic_store(pair_word + blockIdx.x, my_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int pl = __builtin_amdgcn_readfirstlane(partner_left);
          int duty = 8;
          if (pl > 0 && seen >= 4) {
            const double l0 = (L.prio_slot == 0) ? (double)my_left : (double)pl;
            const double l1 = (L.prio_slot == 0) ? (double)pl : (double)my_left;
            const double rho = l0 / l1;
            // (the rates of a pair with and without the priority, measured for this model; a model
            // without measured rates keeps even shares)
            constexpr double kHi = M::kPrioRateWith, kLo = M::kPrioRateWithout;
            if constexpr (kHi > kLo) {
              const double f = (kHi * rho - kLo) / ((kHi - kLo) * (1.0 + rho));
              duty = (int)(16.0 * fmin(fmax(f, 0.0), 1.0) + 0.5);
            }
          }
          L.prio_duty = duty;
        }
        if (done == P.n_draws) {
          chain_store<M, G>(P.st, C, cur, l, st);
          int last = 0;
          if (lane == 0) {
            atomicSub(board, 1);
            last = (atomicSub(live, 1) == 1) ? 1 : 0;
          }
          last = __builtin_amdgcn_readfirstlane(last);
          if (!last) break;
          if (lane == 0) atomicAdd(board + 3, 1);
          // the SIMD has run empty: stay as a host until a chain arrives or none is left
          int got = 0, advertised = 0;
          // (one poll per ~100 us and per host, by one lane: a thousand hosts polling one word
          // every few microseconds saturate its L2 channel and slow every running chain)
          for (int spin = 0; spin < (1 << 18); spin++) {
            int a = advertised, m0 = 0, left = 1;
            if (lane == 0) {
              left = mig_load(board);
              if (!a && mig_load(board + 1) == 0) a = (atomicCAS(board + 1, 0, (int)blockIdx.x + 1) == 0) ? 1 : 0;
              m0 = mig_load(mail);
            }
            advertised = __builtin_amdgcn_readfirstlane(a);
            got = __builtin_amdgcn_readfirstlane(m0);
            if (got != 0 || __builtin_amdgcn_readfirstlane(left) <= 0) break;
#pragma unroll 1
            for (int z = 0; z < 32; z++) __builtin_amdgcn_s_sleep(127);
          }
          if (got == 0) break;
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          cur = got - 1;
          done = __builtin_amdgcn_readfirstlane(mig_load(mail + 1));
          if (lane == 0) {
            __hip_atomic_store(mail, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(live, 1);
          }
          chain_load<M, G>(P.st, C, cur, l, st);
          bind(cur, (size_t)P.draw_offset + (size_t)done);
          L.prio_slot = -1;   // a host has its SIMD to itself
          continue;
        }
        if (__builtin_amdgcn_readfirstlane((host_adv != 0 && n_live >= 2) ? 1 : 0)) {
          // claim the right to leave (one of the two), then the advertised host
          int h = 0;
          if (lane == 0) {
            if (atomicCAS(live, 2, 1) == 2) {
              h = atomicExch(board + 1, 0);
              if (h == 0) atomicAdd(live, 1);   // somebody else took the host: stay
            }
          }
          h = __builtin_amdgcn_readfirstlane(h);
          if (h != 0) {
            if (lane == 0) atomicAdd(board + 2, 1);
            chain_store<M, G>(P.st, C, cur, l, st);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (lane == 0) {
              int* const hm = board + kMigMail + 2 * (h - 1);
              __hip_atomic_store(hm + 1, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(hm, cur + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            break;
          }
        }
      }
      if (l == 0 && P.counters) {
        atomicAdd(&P.counters[0], lf_total);
        atomicAdd(&P.counters[1], div_total);
      }
#ifdef EXMC_XCC_PROBE
      probe_out();
#endif
      return;
    }
    nuts_run<M, G, LDSL>(mc, L, st, P.n_draws, P.eps, P.max_depth, sink);
  } else {
    nuts_run<M, G, LDSL>(mc, L, st, P.n_draws, P.eps, P.max_depth, sink);
  }
#ifdef EXMC_XCC_PROBE   // development: per-wave clocks, wall-clock span and placement of the sampling kernel
  probe_out();
#endif

  if (!has_chain) return;
  chain_store<M, G>(P.st, C, chain, l, st);
  if (l == 0 && P.counters) {
    atomicAdd(&P.counters[0], lf_total);
    atomicAdd(&P.counters[1], div_total);
  }
}

// ------------------------------------------------------------------------------------------
// The sampling kernel as workgroups of W wavefronts that share ONE LDS image of the model's data
// (M::kWgWaves, M::wg_stage / wg_attach / kWgImageDoubles; logistic at 16 lanes per chain: its
// 500 x 20 design matrix, which every wavefront otherwise streams from L2 at every leapfrog; a
// generated lane layout: its tables). Every wavefront is the one-wave kernel on its own slice of LDS
// -- W tree stacks of LDSL levels, the ziggurat tables once, W strips of model scratch
// (kExtraLdsDoubles), the image once -- and after the staging barrier no wavefront waits for another.
//   LDS: [wave 0 stack][wave 1 stack] ... [ziggurat tables][wave 0 scratch] ... [image]
// Diagonal mass, no stream, no migration (the kinds that have those keep nuts_kernel).
// ------------------------------------------------------------------------------------------
template <class M, int LDSL, int W>
__host__ __device__ constexpr size_t nuts_wg_extra_offset() {   // in doubles: the wavefronts' model scratch
  return (size_t)W * LDSL * nuts_nslot<M>() * kNutsBlock + kZigLdsBytes / 8;
}
template <class M, int LDSL, int W>
__host__ __device__ constexpr size_t nuts_wg_image_offset() {   // in doubles; even
  return (nuts_wg_extra_offset<M, LDSL, W>() + (size_t)W * M::kExtraLdsDoubles + 1) & ~(size_t)1;
}
template <class M, int LDSL, int W>
__host__ __device__ constexpr size_t nuts_wg_lds_bytes() {
  return (nuts_wg_image_offset<M, LDSL, W>() + (size_t)M::kWgImageDoubles) * 8;
}

// the model as the workgroup form evaluates it: always from the image (wg_attach has pointed the lane
// at it), through the model's own entry for that case -- the compiler keeps one copy of the pass
template <class B>
struct WgModel : B {
  __device__ static __forceinline__ double logp_grad(const typename B::Consts& c, const typename B::Lane& ln,
                                                     int l, const double (&q)[B::DPL], double (&g)[B::DPL]) {
    return B::logp_grad_staged(c, ln, l, q, g);
  }
};

template <class B, int G, int LDSL, int W>
__global__ void __launch_bounds__(W * kNutsBlock) nuts_kernel_wg(NutsParams P, typename B::Consts mc) {
  using M = WgModel<B>;
  constexpr int D = M::D, DPL = M::DPL;
  constexpr int NSLOT = nuts_nslot<M>();
  static_assert(M::kWgImageDoubles > 0 && !M::kCoop, "a model with an image for the workgroup form (wg_stage / wg_attach)");
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tid = blockIdx.x * (W * kNutsBlock) + threadIdx.x;
  const int C = P.n_chains;
  const bool has_chain = (tid / G) < C;
  const int chain = has_chain ? (tid / G) : (C - 1);
  // cooperative part: the tables and the image, once per workgroup
  double* const lz = lds + (size_t)W * LDSL * NSLOT * kNutsBlock;
  for (int i = threadIdx.x; i < 256; i += W * kNutsBlock) {
    lz[i] = __longlong_as_double((long long)P.zig_ki[i]);
    lz[256 + i] = P.zig_wi[i];
    lz[512 + i] = P.zig_fi[i];
  }
  double* const image = lds + nuts_wg_image_offset<M, LDSL, W>();
  M::wg_stage(mc, image);
  __syncthreads();
  if (!has_chain) return;

  NutsLane<M, G> L;
  const ZigTables zt{(const uint64_t*)lz, lz + 256, lz + 512};
  lane_setup<M, G, LDSL>(L, mc, lds, P.stack, P.inv_mass, P.sqrt_inv_mass, zt, P.nor_r, P.flat, P.dm);
  L.lstk = lds + (size_t)wave * LDSL * NSLOT * kNutsBlock + lane;   // this wavefront's stack
  if constexpr (M::kExtraLdsDoubles > 0)                             // ... and its model scratch
    L.ln.sh = lds + nuts_wg_extra_offset<M, LDSL, W>() + (size_t)wave * M::kExtraLdsDoubles;
  M::wg_attach(L.ln, image, (int)nuts_wg_image_offset<M, LDSL, W>());
  // two waves per SIMD by construction (W = 8 on four SIMDs); which two share one is the
  // dispatcher's choice, so the priority is left to the arbiter
  L.prio_slot = -1;
  ChainRegs<DPL> st;
  chain_load<M, G>(P.st, C, chain, L.l, st);

  unsigned long long lf_total = 0, div_total = 0;
  const int l = L.l;
  const bool own = (l == 0);
  // (the trace addresses are computed per draw: a draw of this kernel is tens of thousands of
  // instructions, and cursors held across them would only be spilled)
  auto sink = [&](int draw, const double (&sq)[DPL], double slogp, int depth, int t_n, bool t_div,
                  double t_acc, double jlp0) {
    const size_t row = (size_t)P.draw_offset + (size_t)draw;
#pragma unroll
    for (int k = 0; k < DPL; k++)
      if (L.valid[k] && P.tr.draws) P.tr.draws[(row * D + (l + k * G)) * C + chain] = sq[k];
    if (own) {
      const size_t at = row * C + chain;
      if (P.tr.logp) P.tr.logp[at] = slogp;
      if (P.tr.tree_depth) P.tr.tree_depth[at] = depth;
      if (P.tr.n_steps) P.tr.n_steps[at] = t_n;
      if (P.tr.divergent) P.tr.divergent[at] = t_div ? 1 : 0;
      if (P.tr.accept_prob) P.tr.accept_prob[at] = (t_n > 0) ? (t_acc / (double)t_n) : 0.0;
      if (P.tr.energy) P.tr.energy[at] = -jlp0;
    }
    lf_total += (unsigned long long)t_n;
    div_total += t_div ? 1u : 0u;
  };
  nuts_run<M, G, LDSL>(mc, L, st, P.n_draws, P.eps, P.max_depth, sink);
  chain_store<M, G>(P.st, C, chain, l, st);
  if (l == 0 && P.counters) {
    atomicAdd(&P.counters[0], lf_total);
    atomicAdd(&P.counters[1], div_total);
  }
}

// ------------------------------------------------------------------------------------------
// find_reasonable_epsilon_with_rng (sampler.ex:451-530) for this group's chain
// ------------------------------------------------------------------------------------------
template <class M, int G>
__device__ __forceinline__ double find_eps_dev(const typename M::Consts& mc,
                                               const NutsLane<M, G>& L, ChainRegs<M::DPL>& st,
                                               double log_half) {
  constexpr int DPL = M::DPL;
  double p0[DPL];
#pragma unroll
  for (int k = 0; k < DPL; k++) p0[k] = 0.0;
  // plain sequential normal_s draws (as sample_momentum_fast does); identical to draw_momentum
  draw_momentum<M, G>(L, st.rng, p0);
  const double jlp0 = st.logp - mass_ke<M, G>(L, p0);
  auto try_eps = [&](double eps) -> double {
    double q[DPL], p[DPL], g[DPL];
    const double h = eps / 2.0;
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      p[k] = p0[k] + h * st.g[k];
      q[k] = st.q[k];
      g[k] = 0.0;
    }
    mass_drift<M, G>(L, eps, p, q);
    const double lp = M::logp_grad(mc, L.ln, L.l, q, g);
#pragma unroll
    for (int k = 0; k < DPL; k++) p[k] = p[k] + h * g[k];
    const double jlp = lp - mass_ke<M, G>(L, p);
    return (exmc_isfinite(jlp0) && exmc_isfinite(jlp)) ? (jlp - jlp0) : -1000.0;
  };
  double eps = 1.0;
  double la = try_eps(eps);
  const double dir = (la > log_half) ? 1.0 : -1.0;
  const double factor = (dir > 0) ? 2.0 : 0.5;
  double result = 0.0;
  bool done = false;
  for (int count = 0; count < 100 && !done; count++) {
    const double ne = eps * factor;
    la = try_eps(ne);
    const bool crossed = (dir > 0) ? (la < log_half) : (la > log_half);
    if (crossed || !exmc_isfinite(la)) {
      result = fmax(ne, 1.0e-10);
      done = true;
    } else {
      eps = ne;
    }
  }
  if (!done) result = fmax(eps, 1.0e-10);
  return result;
}

// ------------------------------------------------------------------------------------------
// On-device adaptation warmup for chain 0 (run_warmup, sampler.ex:537-762).
// Scalar adaptation math follows step_size.ex / mass_matrix.ex with exp/log from the numeric
// contract (exmc_detmath.h), sqrt IEEE, m^-kappa as exp(-kappa*log(m)).
// ------------------------------------------------------------------------------------------
struct WarmupParams {
  ChainState st;            // single-chain state (C = 1), initialised by init_chains_kernel
  int num_warmup;
  int max_depth;
  double target_accept;
  double log_half;          // log(0.5) as the host computes it (sampler.ex:469)
  int init_buffer, adapt_end;
  int n_windows;
  int win_start[32], win_end[32];   // build_windows (sampler.ex:764-785), computed on the host
  double* stack;
  size_t stack_stride;      // doubles of spill stack per workgroup (replicas, see `race`)
  int* race;                // null, or a zeroed word: the workgroups are replicas of the same
                            // deterministic chain and the first to finish publishes the result
  int stage_model;          // 1: dynamic LDS includes M::kStageDoubles for the model's LDS image
  double eps0;              // > 0: warm start (sampler.ex:167-197) -- start from this step size and
  const double* inv_mass0;  // this inverse mass (dev [D], with sqrt_inv_mass0) instead of the
  const double* sqrt_inv_mass0;   // identity mass and the initial step-size search
  int dense;                // opts[:dense_mass]: dense Welford windows (lanes_per_chain = 1, one-wave form);
                            // the dynamic LDS then ends with 3 D^2 doubles (m2, cov, chol)
  double* dense_ws;         // lane layouts (M::kLaneDense): LaneDenseWs::doubles() of global workspace
  double* out;              // [0] eps_final (< 0: a window covariance was not positive definite),
                            // [1] divergences, [2] leapfrogs, [3..3+D) inv_mass, dense: cov, chol [D][D] each
  const uint64_t* zig_ki;
  const double* zig_wi;
  const double* zig_fi;
  double nor_r;
  FlatOrder flat;
};

struct DualAvgDev {
  double log_epsilon, log_epsilon_bar, h_bar, mu, target;
  int m;
  __device__ __forceinline__ void init(double epsilon, double target_accept) {
    log_epsilon = exmc_log(epsilon);
    log_epsilon_bar = log_epsilon;
    h_bar = 0.0;
    mu = exmc_log(10.0 * epsilon);
    m = 0;
    target = target_accept;
  }
  // The factors of update() that depend on the iteration count alone (step_size.ex:28-40):
  // computed while the transition runs, so that only the accept-statistic part is left between
  // the end of one tree and the step size of the next. Same operations, same values.
  double eta_, one_m_eta_, sq_, mk_, one_m_mk_;
  __device__ __forceinline__ void prepare() {
    const int mm = m + 1;
    eta_ = 1.0 / (mm + 10.0);
    one_m_eta_ = 1.0 - eta_;
    sq_ = __dsqrt_rn((double)mm) / 0.05;
    mk_ = exmc_exp(-0.75 * exmc_log((double)mm));
    one_m_mk_ = 1.0 - mk_;
  }
  __device__ __forceinline__ void update(double accept_stat) {   // after prepare()
    const double hb = one_m_eta_ * h_bar + eta_ * (target - accept_stat);
    const double le = mu - sq_ * hb;
    const double leb = mk_ * le + one_m_mk_ * log_epsilon_bar;
    m = m + 1;
    h_bar = hb;
    log_epsilon = le;
    log_epsilon_bar = leb;
  }
};

// finalize_dense (mass_matrix.ex:105-140) by one lane on [D][D] arrays: cov = m2 / (n - 1), shrunk
// toward its floored diagonal; lower Cholesky factor row by row, sums with fma in ascending k.
// Fewer than 3 samples: identity. Returns false if a pivot is not positive.
template <int D>
__device__ __forceinline__ bool dense_finalize_serial(const double* m2, double* cov, double* chol, int wn) {
  bool ok = true;
  if (wn < 3) {
    for (int e = 0; e < D * D; e++) cov[e] = chol[e] = ((e / D) == (e % D)) ? 1.0 : 0.0;
    return ok;
  }
  const double alpha = 5.0 / (wn + 5.0);
  for (int e = 0; e < D * D; e++) cov[e] = m2[e] / ((double)(wn - 1) * 1.0);
  for (int a = 0; a < D; a++)
    for (int b = 0; b < D; b++) {
      const double dg = (a == b) ? fmax(cov[a * D + a], 1.0e-6) : 0.0;
      cov[a * D + b] = (1.0 - alpha) * cov[a * D + b] + alpha * dg;
    }
  for (int e = 0; e < D * D; e++) chol[e] = 0.0;
  for (int a = 0; a < D; a++)
    for (int b = 0; b <= a; b++) {
      double acc = cov[a * D + b];
      for (int k = 0; k < b; k++) acc = __builtin_fma(-chol[a * D + k], chol[b * D + k], acc);
      if (a == b) {
        ok = ok && (acc > 0.0);
        chol[a * D + a] = __dsqrt_rn(acc);
      } else {
        chol[a * D + b] = acc / chol[b * D + b];
      }
    }
  return ok;
}

// ---- dense Welford windows in a lane layout (M::kLaneDense): the window's co-moment matrix, the
// covariance, its factor and their permuted copies (exmc_device.hpp LaneDense) live in a global
// workspace of 3 D^2 + 2 D GD doubles; rows and columns are flat entries. The chain's G lanes
// share the element-wise work; what one lane writes and another reads is separated by an
// agent-scope fence (a single wave: program order does the rest). ----
__device__ __forceinline__ void wave_global_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <int G, int DPL, int D>
struct LaneDenseWs {
  static constexpr int GD = G * DPL;
  double *m2, *cov, *chol, *covp, *cholp;
  __device__ __forceinline__ void bind(double* ws) {
    m2 = ws;
    cov = m2 + D * D;
    chol = cov + D * D;
    covp = chol + D * D;
    cholp = covp + D * GD;
  }
  static constexpr size_t doubles() { return 3 * (size_t)D * D + 2 * (size_t)D * GD; }
};

// mass_matrix.ex:56-72: m2[a][b] = fma(delta_a, delta2_b, m2[a][b]) over flat entries; lane l owns
// the columns b = l, l + G, ... of every row, so an element is always updated by the same lane
template <int G, int DPL, int D>
__device__ __forceinline__ void lane_dense_welford(double* m2, double* xs, int l, const int (&rank)[DPL],
                                                   const bool (&valid)[DPL], const double (&dl)[DPL],
                                                   const double (&dl2)[DPL]) {
  wave_lds_fence();
#pragma unroll
  for (int k = 0; k < DPL; k++)
    if (valid[k]) {
      xs[rank[k]] = dl[k];
      xs[D + rank[k]] = dl2[k];
    }
  wave_lds_fence();
  for (int a = 0; a < D; a++) {
    const double da = xs[a];
    for (int b = l; b < D; b += G) m2[a * D + b] = __builtin_fma(da, xs[D + b], m2[a * D + b]);
  }
}

// mass_matrix.ex:105-140 as dense_finalize_serial states it, the element-wise parts spread over the
// G lanes and the factor built column by column: entry (a, b) still accumulates
// fma(-L[a][k], L[b][k], .) in ascending k. rank_of: kernel dimension -> flat entry (null = identity).
template <int G, int DPL, int D>
__device__ __forceinline__ bool lane_dense_finalize(const LaneDenseWs<G, DPL, D>& w, double* xs, int l,
                                                    const int32_t* rank_of, int wn) {
  constexpr int GD = G * DPL;
  constexpr int T = (D + G - 1) / G;
  const double alpha = 5.0 / (wn + 5.0);
  for (int e = l; e < D * D; e += G) {
    const int a = e / D, b = e - a * D;
    if (wn < 3) {
      w.cov[e] = w.chol[e] = (a == b) ? 1.0 : 0.0;
    } else {
      const double v = w.m2[e] / ((double)(wn - 1) * 1.0);
      const double dg = (a == b) ? fmax(v, 1.0e-6) : 0.0;
      w.cov[e] = (1.0 - alpha) * v + alpha * dg;
      w.chol[e] = 0.0;
    }
  }
  wave_global_fence();
  bool ok = true;
  if (wn >= 3) {
    for (int b = 0; b < D; b++) {
      double acc[T];
#pragma unroll
      for (int t = 0; t < T; t++) {
        const int a = l + t * G;
        acc[t] = 0.0;
        if (a >= b && a < D) {
          double s = w.cov[a * D + b];
          for (int k = 0; k < b; k++) s = __builtin_fma(-w.chol[a * D + k], w.chol[b * D + k], s);
          acc[t] = s;
          if (a == b) {
            ok = ok && (s > 0.0);
            const double dgl = __dsqrt_rn(s);
            w.chol[b * D + b] = dgl;
            xs[0] = dgl;
          }
        }
      }
      wave_lds_fence();
      const double dgl = xs[0];
#pragma unroll
      for (int t = 0; t < T; t++) {
        const int a = l + t * G;
        if (a > b && a < D) w.chol[a * D + b] = acc[t] / dgl;
      }
      wave_global_fence();   // column b is read by every later column; xs[0] is free again
    }
  }
  for (int e = l; e < D * GD; e += G) {
    const int s_ = e / GD, i = e - s_ * GD;
    const int r = (i < D) ? (rank_of ? rank_of[i] : i) : 0;
    w.covp[e] = (i < D) ? w.cov[r * D + s_] : 0.0;
    w.cholp[e] = (i < D) ? w.chol[s_ * D + r] : 0.0;
  }
  wave_global_fence();
  return ok;
}

// kPipe: two waves, the second one integrating one leaf ahead of the tree (see PipeBox above).
template <class M, int G, int LDSL, bool kPipe = false>
__global__ void __launch_bounds__(kPipe ? 2 * kNutsBlock : kNutsBlock)
    warmup_kernel(WarmupParams P, typename M::Consts mc) {
  constexpr int D = M::D, DPL = M::DPL;
  constexpr int NSLOT = nuts_nslot<M>();
  extern __shared__ double lds[];
  const ZigTables zt = stage_zig_tables<LDSL, NSLOT>(lds, P.zig_ki, P.zig_wi, P.zig_fi);
  // single workgroup on an otherwise idle chip: keep the model data in LDS when it offers an image
  double* stage_ptr = nullptr;
  size_t lds_used = nuts_lds_bytes<M, LDSL>() / 8;
  if constexpr (M::kStageDoubles > 0) {
    if (P.stage_model) {
      stage_ptr = lds + lds_used;
      lds_used += (size_t)M::kStageDoubles;
      M::stage(mc, stage_ptr);
      __syncthreads();
    }
  }
  const int xoff = stage_model_data<M, G, LDSL>(mc, lds);
  // Wave-cooperative models: every lane group of the wave runs the same chain 0 redundantly (they
  // stay in lockstep, so the wave is fully populated at every logp_grad); group 0 writes.
  const int lane = threadIdx.x & 63;
  const bool writer = lane < G;
  if (!M::kCoop && !writer) return;

  NutsLane<M, G> L;
  lane_setup<M, G, LDSL>(L, mc, lds, P.stack, P.inv_mass0, P.sqrt_inv_mass0, zt, P.nor_r, P.flat);
  if constexpr (M::kStageDoubles > 0) L.ln.xs = stage_ptr ? stage_ptr + M::kStageRowOffset : nullptr;
  if constexpr (M::kLdsDataDoubles > 0) L.ln.xoff = xoff;
  using Pipe = std::conditional_t<kPipe, PipeBox<DPL>, NoPipe>;
  Pipe pipe;
  if constexpr (kPipe) {
    // both waves index the stack and the mailbox by their lane, not by threadIdx.x
    L.lstk = lds + lane;
    L.gstk = P.stack + (size_t)blockIdx.x * P.stack_stride + lane;
    pipe.box = lds + lds_used + lane;
    pipe.seq = 0;
    pipe.pending = false;
    if (threadIdx.x >= kNutsBlock) {
      const bool windows = P.adapt_end > P.init_buffer;
      const int n = (P.num_warmup > 0) ? (windows ? P.num_warmup : P.init_buffer) : 0;
      for (int i = 0; i < n; i++)
        if (!pipe_integrate_transition<M, G>(mc, L, pipe)) break;
      return;
    }
  }
  ChainRegs<DPL> st;
  chain_load<M, G>(P.st, 1, 0, L.l, st);
  // dense mode: the window's co-moment matrix, its covariance and factor live behind everything
  // else in LDS; the one lane that carries the chain is their only user
  double* dn_m2 = nullptr;
  double* dn_cov = nullptr;
  double* dn_chol = nullptr;
  bool dense_ok = true;
  // where the dense mode is built: a whole chain in one lane, or the row layout (M::kRowDense: the
  // window's co-moment rows in registers m2r, one row per lane; LDS holds what finalize exchanges)
  constexpr bool kLaneDenseHere = M::kLaneDense && !kPipe;
  LaneDenseWs<G, DPL, D> lw;
  bool lane_dense = false;
  if constexpr (kLaneDenseHere) {
    if (P.dense) {
      lane_dense = true;
      lw.bind(P.dense_ws);
    }
  }
  constexpr bool kDenseHere = (G == 1 || M::kRowDense) && !kPipe && !M::kLaneDense;
  constexpr int kRowN = M::kRowDense ? D : 1;
  double m2r[kRowN];
#pragma unroll
  for (int j = 0; j < kRowN; j++) m2r[j] = 0.0;
  if constexpr (kDenseHere) {
    if (P.dense) {
      dn_m2 = lds + lds_used;
      dn_cov = dn_m2 + D * D;
      dn_chol = dn_cov + D * D;
      for (int i = 0; i < D * D; i++) {   // every active lane writes the same values
        dn_m2[i] = 0.0;
        dn_cov[i] = dn_chol[i] = ((i / D) == (i % D)) ? 1.0 : 0.0;
      }
    }
  }
#ifdef EXMC_XCC_PROBE
  const long long probe_c0 = clock64(), probe_w0 = wall_clock64();
#endif

  double accept = 0.0;
  bool diverged = false;
  unsigned long long leapfrogs = 0;
  int divergences = 0;
  auto sink = [&](int, const double (&)[DPL], double, int, int t_n, bool t_div, double t_acc,
                  double) {
    accept = (t_n > 0) ? (t_acc / (double)t_n) : 0.0;
    diverged = t_div;
    leapfrogs += (unsigned long long)t_n;
  };

  const int W = P.num_warmup;
  bool lost = false;
  double eps = (P.eps0 > 0.0) ? P.eps0 : find_eps_dev<M, G>(mc, L, st, P.log_half);
  double eps_final = eps;
  if (W > 0) {
    const bool has_windows = P.adapt_end > P.init_buffer;
    const int n_iter = has_windows ? W : P.init_buffer;
    DualAvgDev da;
    da.init(eps, P.target_accept);
    int win = 0;
    bool in_window = false;
    int wn = 0;
    double wmean[DPL], wm2[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) wmean[k] = wm2[k] = 0.0;
    int race_seen = 0;
    for (int i = 0; i < n_iter; i++) {
      // another replica of this chain has already finished: leave (the integrator wave is told
      // through the mailbox, at the barrier it is waiting at)
      if (race_seen != 0) {
        lost = true;
        if constexpr (kPipe) pipe.quit();
        break;
      }
      // the flag is loaded here and looked at one transition later: the L2 round trip, which a
      // lone wave would otherwise sit out, returns while the tree is built
      if (P.race) race_seen = __hip_atomic_load(P.race, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (has_windows) {
        if (win < P.n_windows && i == P.win_start[win]) {
          if (win == 0) eps = exmc_exp(da.log_epsilon);   // sampler.ex:578
          wn = 0;
#pragma unroll
          for (int k = 0; k < DPL; k++) wmean[k] = wm2[k] = 0.0;
          if (dn_m2) {
            if constexpr (M::kRowDense) {
#pragma unroll
              for (int j = 0; j < kRowN; j++) m2r[j] = 0.0;
            } else {
              for (int e = 0; e < D * D; e++) dn_m2[e] = 0.0;
            }
          }
          if constexpr (kLaneDenseHere) {
            if (lane_dense)
              for (int e = L.l; e < D * D; e += G) lw.m2[e] = 0.0;   // each element by the lane that updates it
          }
          da.init(eps, P.target_accept);
          in_window = true;
        }
        if (i == P.adapt_end) da.init(eps, P.target_accept);   // Phase III (sampler.ex:601)
      }
      // sampler.ex:709: depth cap 8 for absolute warmup index < 200, Phase II only
      const int cap = (in_window && i < 200) ? (P.max_depth < 8 ? P.max_depth : 8) : P.max_depth;
      auto idle = [&]() { da.prepare(); };
      if constexpr (kPipe) nuts_run<M, G, LDSL>(mc, L, st, 1, exmc_exp(da.log_epsilon), cap, sink, &pipe, idle);
      else nuts_run<M, G, LDSL>(mc, L, st, 1, exmc_exp(da.log_epsilon), cap, sink, (NoPipe*)nullptr, idle);
      divergences += diverged ? 1 : 0;
      da.update(accept);
      if (in_window) {
        if (!diverged) {
          // mass_matrix.ex:40-54 (diagonal), :56-72 (dense: m2 += outer(delta, delta2))
          const int nn = wn + 1;
          double dl[DPL], dl2[DPL];
#pragma unroll
          for (int k = 0; k < DPL; k++) {
            const double delta = st.q[k] - wmean[k];
            const double nm = wmean[k] + delta / ((double)nn * 1.0);
            const double d2 = st.q[k] - nm;
            wm2[k] = wm2[k] + delta * d2;
            wmean[k] = nm;
            dl[k] = delta;
            dl2[k] = d2;
          }
          if constexpr (kDenseHere) {
            if (dn_m2) {
              if constexpr (M::kRowDense) {
                // lane a's row: m2[a][b] = fma(delta_a, delta2_b, m2[a][b]), delta2_b from lane b
                row_outer_acc<kRowN>(m2r, dl2[0], dl[0]);
              } else {
#pragma unroll
                for (int a = 0; a < D; a++)
#pragma unroll
                  for (int b = 0; b < D; b++) dn_m2[a * D + b] = __builtin_fma(dl[a], dl2[b], dn_m2[a * D + b]);
              }
            }
          }
          if constexpr (kLaneDenseHere) {
            if (lane_dense) lane_dense_welford<G, DPL, D>(lw.m2, L.ld.xs, L.l, L.rank, L.valid, dl, dl2);
          }
          wn = nn;
        }
        if (i + 1 == P.win_end[win]) {
          // mass_matrix.ex:77-97, then re-search the step size (sampler.ex:747-756)
          bool dense_done = false;
          if constexpr (kDenseHere) {
            if (dn_m2) {
              if constexpr (M::kRowDense) {
                // the rows go to LDS, lane 0 of the group finalizes, every lane takes back its row
                // of the covariance and its column of the factor
                if (L.l < D) {
#pragma unroll
                  for (int j = 0; j < kRowN; j++) dn_m2[L.l * D + j] = m2r[j];
                }
                __syncthreads();
                if (L.l == 0) {
                  const bool ok = dense_finalize_serial<D>(dn_m2, dn_cov, dn_chol, wn);
                  dn_m2[0] = ok ? 1.0 : 0.0;
                }
                __syncthreads();
                dense_ok = dense_ok && (dn_m2[0] != 0.0);
                rowdense_load<M, G>(L, dn_cov, dn_chol, 1);
                L.im[0] = (L.l < D) ? dn_cov[L.l * D + L.l] : 1.0;   // inv_mass_diag_out
                __syncthreads();   // dn_m2[0] is read before the next window's rows overwrite it
              } else {
                L.dm = DenseMass{};
                dense_ok = dense_finalize_serial<D>(dn_m2, dn_cov, dn_chol, wn) && dense_ok;
#pragma unroll
                for (int k = 0; k < DPL; k++) L.im[k] = dn_cov[k * D + k];   // inv_mass_diag_out
                L.dm.cov = dn_cov;
                L.dm.chol = dn_chol;
              }
              dense_done = true;
            }
          }
          if constexpr (kLaneDenseHere) {
            if (lane_dense) {
              L.ld.covp = L.ld.cholp = nullptr;
              const bool ok = lane_dense_finalize<G, DPL, D>(lw, L.ld.xs, L.l, P.flat.rank, wn);
              dense_ok = dense_ok && (__all(ok ? 1 : 0) != 0);
#pragma unroll
              for (int k = 0; k < DPL; k++)
                L.im[k] = L.valid[k] ? lw.cov[L.rank[k] * D + L.rank[k]] : 1.0;   // inv_mass_diag_out
              L.ld.covp = lw.covp;
              L.ld.cholp = lw.cholp;
              if constexpr (M::kDenseImage) {
                lane_dense_stage<G, DPL, D>(L.ld, L.l);
                wave_lds_fence();
              }
              dense_done = true;
            }
          }
          if (dense_done) {
          } else if (wn < 3) {
#pragma unroll
            for (int k = 0; k < DPL; k++) L.im[k] = 1.0;
          } else {
            const double alpha = 5.0 / (wn + 5.0);
#pragma unroll
            for (int k = 0; k < DPL; k++) {
              double var = wm2[k] / ((double)(wn - 1) * 1.0);
              var = fmax(var, 1.0e-6);
              L.im[k] = L.valid[k] ? ((1.0 - alpha) * var + alpha * 1.0e-3) : 1.0;
            }
          }
#pragma unroll
          for (int k = 0; k < DPL; k++) L.sim[k] = __dsqrt_rn(L.im[k]);
          eps = find_eps_dev<M, G>(mc, L, st, P.log_half);
          win++;
          in_window = false;
        }
      }
    }
    eps_final = exmc_exp(da.log_epsilon_bar);
    if constexpr (kPipe) {
      // the integrator wave has served its n_iter transitions and waits at its last barrier
      // (a lost race has been through quit(), which clears `pending`)
      if (pipe.pending) pipe.sync();
    }
  }

  if (!writer) return;
  if (P.race) {
    // the first replica to get here owns the outputs
    int won = 0;
    if (!lost && lane == 0) won = (atomicCAS(P.race, 0, 1) == 0) ? 1 : 0;
    won = __builtin_amdgcn_readfirstlane(won);
    if (!won) return;
  }
  chain_store<M, G>(P.st, 1, 0, L.l, st);
  if (L.l == 0) {
    P.out[0] = eps_final;
    P.out[1] = (double)divergences;
    P.out[2] = (double)leapfrogs;
#ifdef EXMC_XCC_PROBE   // development: which XCD / CU / SE the single workgroup ran on
    {
      unsigned xcc, hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      P.out[2] = (double)((xcc & 0xf) * 1000 + ((hw >> 8) & 0xf) * 10 + ((hw >> 13) & 0x7));
      P.out[3 + D] = (double)(clock64() - probe_c0);
      P.out[4 + D] = (double)(wall_clock64() - probe_w0);
    }
#endif
  }
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = L.l + k * G;
    if (i < D) P.out[3 + i] = L.im[k];
  }
  if constexpr (kDenseHere) {
    if (dn_m2 && L.l == 0) {
      for (int e = 0; e < D * D; e++) {
        P.out[3 + D + e] = dn_cov[e];
        P.out[3 + D + D * D + e] = dn_chol[e];
      }
      if (!dense_ok) P.out[0] = -1.0;
    }
  }
  if constexpr (kLaneDenseHere) {
    if (lane_dense) {
      const bool have = L.ld.covp != nullptr;   // no window ran: the identity
      for (int e = L.l; e < D * D; e += G) {
        const double id = ((e / D) == (e % D)) ? 1.0 : 0.0;
        P.out[3 + D + e] = have ? lw.cov[e] : id;
        P.out[3 + D + D * D + e] = have ? lw.chol[e] : id;
      }
      if (!dense_ok && L.l == 0) P.out[0] = -1.0;
    }
  }
}

// ------------------------------------------------------------------------------------------
// sample_chains(ir, n, vectorized: false) -- sample_chains_parallel, sampler.ex:1139-1176: every
// chain is Sampler.sample/3 with seed base + 7919 i: its OWN adaptation (step-size search, dual
// averaging, Welford windows, sampler.ex:537-762) and then its draws, continuing from the adapted
// position with the same generator. One launch: every lane group of the grid is one such chain
// from its first warmup transition to its last draw. The adaptation state of a chain (dual
// averaging, window moments, step size, inverse mass) is per-lane data, so the groups of a
// wavefront adapt side by side; the window schedule is the same for all of them (it depends on
// num_warmup alone), which keeps the phase boundaries wave-uniform, and nuts_run walks their
// trees in lock step as the sampling kernel does. Each chain executes the operations of the
// one-chain warmup_kernel followed by those of nuts_kernel on its own registers: every output equals
// the checker's sample(seed = base + 7919 i) bit for bit. Diagonal mass, one-wave form.
// ------------------------------------------------------------------------------------------
struct IndepParams {
  ChainState st;            // C chains, initialised by init_chains_kernel (seed base + 7919 i)
  int n_chains;
  int num_warmup, num_samples, max_depth;
  double target_accept, log_half;
  int init_buffer, adapt_end, n_windows;
  int win_start[32], win_end[32];
  double* stack;
  TraceDev tr;              // [S][D][C] / [S][C]
  double* tune_out;         // [C][3 + D]: final step size, warmup divergences, warmup leapfrogs, inv_mass[D]
  unsigned long long* counters;   // [0] leapfrogs, [1] divergent transitions of the SAMPLING phase
  const uint64_t* zig_ki;
  const double* zig_wi;
  const double* zig_fi;
  double nor_r;
  FlatOrder flat;
};

// Launch bounds as nuts_kernel's: where the model asks for two waves per SIMD (sv at 64 lanes, logistic at
// 16) the 2048 waves of the bench protocol are one resident round, not two -- sv 2048 x (1000 + 1000)
// 4.98 s -> 4.48 s on one box, same bits (profiles/r5_svi); the spills this buys (272 B per lane for sv)
// sit in the window updates, outside the tree.
template <class M, int G, int LDSL>
__global__ void __launch_bounds__(kNutsBlock, M::kNutsWavesPerSimd) indep_kernel(IndepParams P, typename M::Consts mc) {
  constexpr int D = M::D, DPL = M::DPL;
  constexpr int NSLOT = nuts_nslot<M>();
  static_assert(!M::kCoop, "wave-cooperative models evaluate every group at the same point of the same chain");
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const int tid = blockIdx.x * kNutsBlock + lane;
  const int C = P.n_chains;
  const bool has_chain = (tid / G) < C;
  const int chain = has_chain ? (tid / G) : (C - 1);
  const ZigTables zt = stage_zig_tables<LDSL, NSLOT>(lds, P.zig_ki, P.zig_wi, P.zig_fi);
  const int xoff = stage_model_data<M, G, LDSL>(mc, lds);
  if (!has_chain) return;

  NutsLane<M, G> L;
  lane_setup<M, G, LDSL>(L, mc, lds, P.stack, nullptr, nullptr, zt, P.nor_r, P.flat);   // identity mass
  if constexpr (M::kLdsDataDoubles > 0) L.ln.xoff = xoff;
  ChainRegs<DPL> st;
  chain_load<M, G>(P.st, C, chain, L.l, st);
  const int l = L.l;

  // ---- adaptation (the diagonal path of warmup_kernel, per lane group) ----
  double accept = 0.0;
  bool diverged = false;
  unsigned long long w_leapfrogs = 0;
  int w_divergences = 0;
  auto wsink = [&](int, const double (&)[DPL], double, int, int t_n, bool t_div, double t_acc, double) {
    accept = (t_n > 0) ? (t_acc / (double)t_n) : 0.0;
    diverged = t_div;
    w_leapfrogs += (unsigned long long)t_n;
  };
  const int W = P.num_warmup;
  double eps = find_eps_dev<M, G>(mc, L, st, P.log_half);
  double eps_final = eps;
  if (W > 0) {
    const bool has_windows = P.adapt_end > P.init_buffer;
    const int n_iter = has_windows ? W : P.init_buffer;
    DualAvgDev da;
    da.init(eps, P.target_accept);
    int win = 0;
    bool in_window = false;
    int wn = 0;
    double wmean[DPL], wm2[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) wmean[k] = wm2[k] = 0.0;
    for (int i = 0; i < n_iter; i++) {
      if (has_windows) {
        if (win < P.n_windows && i == P.win_start[win]) {
          if (win == 0) eps = exmc_exp(da.log_epsilon);   // sampler.ex:578
          wn = 0;
#pragma unroll
          for (int k = 0; k < DPL; k++) wmean[k] = wm2[k] = 0.0;
          da.init(eps, P.target_accept);
          in_window = true;
        }
        if (i == P.adapt_end) da.init(eps, P.target_accept);   // Phase III (sampler.ex:601)
      }
      // sampler.ex:709: depth cap 8 for absolute warmup index < 200, Phase II only
      const int cap = (in_window && i < 200) ? (P.max_depth < 8 ? P.max_depth : 8) : P.max_depth;
      auto idle = [&]() { da.prepare(); };
      nuts_run<M, G, LDSL>(mc, L, st, 1, exmc_exp(da.log_epsilon), cap, wsink, (NoPipe*)nullptr, idle);
      w_divergences += diverged ? 1 : 0;
      da.update(accept);
      if (in_window) {
        if (!diverged) {   // mass_matrix.ex:40-54
          const int nn = wn + 1;
#pragma unroll
          for (int k = 0; k < DPL; k++) {
            const double delta = st.q[k] - wmean[k];
            const double nm = wmean[k] + delta / ((double)nn * 1.0);
            const double d2 = st.q[k] - nm;
            wm2[k] = wm2[k] + delta * d2;
            wmean[k] = nm;
          }
          wn = nn;
        }
        if (i + 1 == P.win_end[win]) {
          // mass_matrix.ex:77-97, then re-search the step size (sampler.ex:747-756)
          if (wn < 3) {
#pragma unroll
            for (int k = 0; k < DPL; k++) L.im[k] = 1.0;
          } else {
            const double alpha = 5.0 / (wn + 5.0);
#pragma unroll
            for (int k = 0; k < DPL; k++) {
              double var = wm2[k] / ((double)(wn - 1) * 1.0);
              var = fmax(var, 1.0e-6);
              L.im[k] = L.valid[k] ? ((1.0 - alpha) * var + alpha * 1.0e-3) : 1.0;
            }
          }
#pragma unroll
          for (int k = 0; k < DPL; k++) L.sim[k] = __dsqrt_rn(L.im[k]);
          eps = find_eps_dev<M, G>(mc, L, st, P.log_half);
          win++;
          in_window = false;
        }
      }
    }
    eps_final = exmc_exp(da.log_epsilon_bar);
  }
  double* const to = P.tune_out + (size_t)chain * (3 + D);
  if (l == 0) {
    to[0] = eps_final;
    to[1] = (double)w_divergences;
    to[2] = (double)w_leapfrogs;
  }
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = l + k * G;
    if (i < D) to[3 + i] = L.im[k];
  }

  // ---- the chain's draws (sample_from_compiled: the same chain goes on, sampler.ex:199-257) ----
  unsigned long long lf_total = 0, div_total = 0;
  const bool own = (l == 0);
  auto sink = [&](int draw, const double (&sq)[DPL], double slogp, int depth, int t_n, bool t_div,
                  double t_acc, double jlp0) {
    const size_t row = (size_t)draw;
#pragma unroll
    for (int k = 0; k < DPL; k++)
      if (L.valid[k] && P.tr.draws) P.tr.draws[(row * D + (l + k * G)) * C + chain] = sq[k];
    if (own) {
      const size_t at = row * C + chain;
      if (P.tr.logp) P.tr.logp[at] = slogp;
      if (P.tr.tree_depth) P.tr.tree_depth[at] = depth;
      if (P.tr.n_steps) P.tr.n_steps[at] = t_n;
      if (P.tr.divergent) P.tr.divergent[at] = t_div ? 1 : 0;
      if (P.tr.accept_prob) P.tr.accept_prob[at] = (t_n > 0) ? (t_acc / (double)t_n) : 0.0;
      if (P.tr.energy) P.tr.energy[at] = -jlp0;
    }
    lf_total += (unsigned long long)t_n;
    div_total += t_div ? 1u : 0u;
  };
  nuts_run<M, G, LDSL>(mc, L, st, P.num_samples, eps_final, P.max_depth, sink);
  chain_store<M, G>(P.st, C, chain, l, st);
  if (l == 0 && P.counters) {
    atomicAdd(&P.counters[0], lf_total);
    atomicAdd(&P.counters[1], div_total);
  }
}

}  // namespace exmc
