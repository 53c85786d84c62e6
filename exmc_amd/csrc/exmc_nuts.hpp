// exmc_nuts.hpp — the NUTS transition kernel for gfx950.
//
//   nuts_kernel  whole NUTS transitions (momentum draw, iterative tree with multinomial
//                proposals, rho-based U-turn checks, divergence guard) for every chain;
//                replaces Tree.build/12 + nuts_step_with_stats (tree.ex:65-151,
//                sampler.ex:854-925) and the Rust crate native/exmc_tree.
//
// The tree is built iteratively. The reference recursion (tree.ex:1144-1203) is unrolled into a
// per-level stack of pending "first halves": after leaf k completes, one inner merge fires for
// every pending level from the bottom up (tree.ex:1390-1476), each consuming one uniform; a node
// that is divergent or turning and has no pending sibling is returned upward unmerged exactly as
// `if first.divergent or first.turning` does (tree.ex:1175-1177); when the doubling's subtree is
// complete it is merged into the trajectory (tree.ex:1479-1568).
//
// Schedule: a wavefront holds 64/G chains. Every pass of the main loop gives each chain group
// exactly ONE leapfrog followed by the merges that leaf triggers, so chains with short trees
// never wait for chains with long ones; the bookkeeping between leapfrogs diverges per group.
// Two alternatives were measured and rejected on MI355X (eight_schools, 4096 chains, G = 16):
// capping merges at one per pass (+20 % time: the extra passes cost more than the "max over
// groups" merge trips they remove) and running inner and outer merges through one unified code
// path (+22 %: operand selects and both proposal rules evaluated).
//
// Memory plan per workgroup (one wavefront): the first LDSL stack levels and the ziggurat
// tables live in LDS; deeper levels spill to a global scratch that stays L2-resident. Per-chain
// dot products run on DPP butterflies (exmc_device.hpp), six at a time per merge.
#pragma once

#include "exmc_models.hpp"

// EXMC_ABLATE = 1..4 builds timing-only variants (U-turn reductions / merge transcendentals /
// model gradient / leaf exp stubbed out) for tools/kernel_probe.py; outputs are wrong then.
#ifndef EXMC_ABLATE
#define EXMC_ABLATE 0
#endif

namespace exmc {

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
  const uint64_t* zig_ki;
  const double* zig_wi;
  const double* zig_fi;
  double nor_r;
};

constexpr int kMaxLevels = 12;
constexpr int kNutsBlock = 64;      // one wavefront per workgroup
constexpr int kZigLdsBytes = 768 * 8;

template <class M>
__host__ __device__ constexpr int nuts_nslot() { return 5 * M::DPL + 3; }

template <class M, int LDSL>
__host__ __device__ constexpr size_t nuts_lds_bytes() {
  return (size_t)LDSL * nuts_nslot<M>() * kNutsBlock * 8 + kZigLdsBytes;
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

// stack slots of one pending node: rho, p_in, p_out, q_prop, g_prop (DPL each), lsw, logp_prop, acc
template <class M, int G, int LDSL>
__global__ void __launch_bounds__(kNutsBlock) nuts_kernel(NutsParams P, typename M::Consts mc) {
  constexpr int D = M::D, DPL = M::DPL;
  constexpr int NSLOT = 5 * DPL + 3;
  extern __shared__ double lds[];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const int l = threadIdx.x & (G - 1);
  const int chain = tid / G;
  const int C = P.n_chains;

  // ziggurat tables -> LDS (all lanes help, before any lane leaves)
  double* lz = lds + (size_t)LDSL * NSLOT * kNutsBlock;
  for (int i = threadIdx.x; i < 256; i += kNutsBlock) {
    lz[i] = __longlong_as_double((long long)P.zig_ki[i]);
    lz[256 + i] = P.zig_wi[i];
    lz[512 + i] = P.zig_fi[i];
  }
  __syncthreads();
  if (chain >= C) return;
  const ZigTables zt{(const uint64_t*)lz, lz + 256, lz + 512};

  typename M::Lane ln;
  M::load(mc, l, ln);

  double im[DPL], sim[DPL];
  bool valid[DPL];
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = l + k * G;
    valid[k] = i < D;
    im[k] = valid[k] ? P.inv_mass[i] : 1.0;
    sim[k] = valid[k] ? P.sqrt_inv_mass[i] : 1.0;
  }

  // chain state between transitions
  double sq[DPL], sg[DPL];
  double slogp = P.st.logp[chain];
  Rng rng;
  rng.a = P.st.rng[chain];
  rng.b = P.st.rng[(size_t)C + chain];
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = l + k * G;
    sq[k] = valid[k] ? P.st.q[(size_t)i * C + chain] : 0.0;
    sg[k] = valid[k] ? P.st.g[(size_t)i * C + chain] : 0.0;
  }

  double* lstk = lds + threadIdx.x;
  double* gstk = P.stack + tid;

  // integrator state and tree registers
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
  int t_n = 0, depth = 0;
  bool t_div = false, t_turn = false, go_right = true;
  double eps_dir = P.eps;
  unsigned pending = 0;
  Rng trng = rng;

  unsigned long long lf_total = 0, div_total = 0;
  int draw = 0;
  bool start_transition = true, start_doubling = false;

  while (draw < P.n_draws) {
    if (start_transition) {
      // sampler.ex:393-403: d sequential normal_s draws. Fast path: the group draws d words, each
      // lane tests the word of its own dimension; if every word is accepted at once (~86 % of
      // transitions at d = 10) the stream position is exactly d words further. Otherwise the
      // draws are redone one by one, as normal_s consumes a data-dependent number of words.
      {
        Rng r2 = rng;
        bool ok = true;
        double z[DPL];
        uint64_t s0[DPL];
#pragma unroll
        for (int k = 0; k < DPL; k++) { z[k] = 0.0; s0[k] = 0; }
        // advance the state d times (cheap recurrence); each lane keeps the tail word its own
        // dimension's output is scrambled from, and scrambles only that one
#pragma unroll
        for (int i = 0; i < D; i++) {
#pragma unroll
          for (int k = 0; k < DPL; k++) s0[k] = (l + k * G == i) ? r2.b : s0[k];
          rng_advance(r2);
        }
#pragma unroll
        for (int k = 0; k < DPL; k++) {
          double zz;
          const bool acc = normal_fast(rng_scramble(s0[k]), zt, zz);
          ok = (acc || !valid[k]) && ok;
          z[k] = zz;
        }
        if (group_all<G>(ok)) {
          rng = r2;
#pragma unroll
          for (int k = 0; k < DPL; k++) pL[k] = z[k] / sim[k];
        } else {
          for (int i = 0; i < D; i++) {
            const double zz = rng_normal(rng, zt, P.nor_r);
#pragma unroll
            for (int k = 0; k < DPL; k++)
              if (l + k * G == i) pL[k] = zz / sim[k];
          }
        }
      }
      jlp0 = slogp - kinetic_energy<G, DPL>(pL, im, valid);
      trng = rng;  // the tree consumes a copy (sampler.ex:897 discards its draws)
#pragma unroll
      for (int k = 0; k < DPL; k++) {
        qL[k] = qR[k] = t_qp[k] = sq[k];
        gL[k] = gR[k] = t_gp[k] = sg[k];
        pR[k] = t_rho[k] = pL[k];
      }
      t_logpP = slogp;
      t_lsw = 0.0;
      t_acc = 0.0;
      t_n = 0;
      t_div = t_turn = false;
      depth = 0;
      start_transition = false;
      start_doubling = true;
    }
    if (start_doubling) {
      // tree.ex:403-413 direction + outward endpoint
      const double u = rng_uniform(trng);
      go_right = u > 0.5;
      eps_dir = go_right ? P.eps : -P.eps;
#pragma unroll
      for (int k = 0; k < DPL; k++) {
        q[k] = go_right ? qR[k] : qL[k];
        p[k] = go_right ? pR[k] : pL[k];
        g[k] = go_right ? gR[k] : gL[k];
      }
      pending = 0;
      start_doubling = false;
    }

    // ---- one leapfrog (batched_leapfrog.ex:79-85) ----
    const double h = eps_dir / 2.0;
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      qold[k] = q[k];
      gold[k] = g[k];
      const double ph = p[k] + h * g[k];
      p[k] = ph;
      q[k] = q[k] + eps_dir * (im[k] * ph);
    }
#if EXMC_ABLATE == 3
    double logp_new = 0.0;
#pragma unroll
    for (int k = 0; k < DPL; k++) { g[k] = -q[k]; logp_new = logp_new - 0.5 * q[k] * q[k]; }
    logp_new = group_allsum<G>(logp_new);
#else
    const double logp_new = M::logp_grad(mc, ln, l, q, g);
#endif
#pragma unroll
    for (int k = 0; k < DPL; k++) p[k] = p[k] + h * g[k];
    const double jlp = logp_new - kinetic_energy<G, DPL>(p, im, valid);

    // ---- leaf (tree.ex:1042-1109) ----
    bool c_div, c_turn = false;
    double c_lsw, c_acc, c_logpP;
    int c_n = 1;
    if (exmc_isfinite(jlp)) {
      const double dl = jlp - jlp0;
      c_div = dl < -1000.0;
      c_lsw = dl;
#if EXMC_ABLATE == 4
      c_acc = fmin(1.0, 1.0 + fmin(dl, 0.0));
#else
      c_acc = fmin(1.0, exmc_exp(fmin(dl, 0.0)));
#endif
    } else {
      c_div = true;
      c_lsw = -1001.0;
      c_acc = 0.0;
    }
    c_acc = c_div ? 0.0 : c_acc;
    c_logpP = c_div ? -1.0e30 : logp_new;
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      c_qp[k] = c_div ? qold[k] : q[k];
      c_gp[k] = c_div ? gold[k] : g[k];
      c_rho[k] = p[k];
      c_pin[k] = p[k];
    }

    // ---- ascend: inner merges for every pending level (tree.ex:1144-1203, 1390-1476) ----
    int lvl = 0;
    bool parked = false;
    while (lvl < depth) {
      if (pending & (1u << lvl)) {
        double nd[NSLOT];
        if (lvl < LDSL) node_load<NSLOT>(lstk + (size_t)lvl * NSLOT * kNutsBlock, kNutsBlock, nd);
        else node_load<NSLOT>(gstk + (size_t)(lvl - LDSL) * NSLOT * nthreads, nthreads, nd);
        const double a_lsw = nd[5 * DPL + 0];
        const double a_logpP = nd[5 * DPL + 1];
        const double a_acc = nd[5 * DPL + 2];
#if EXMC_ABLATE == 2
        const double lsw = a_lsw + c_lsw;
        const double u = rng_uniform(trng);
        const bool use_b = u < 0.5;
#else
        const double lsw = log_sum_exp(a_lsw, c_lsw);
        const double u = rng_uniform(trng);
        const bool use_b = u < exmc_exp(c_lsw - lsw);
#endif
        if (!use_b) {
          c_logpP = a_logpP;
#pragma unroll
          for (int k = 0; k < DPL; k++) { c_qp[k] = nd[3 * DPL + k]; c_gp[k] = nd[4 * DPL + k]; }
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
          bool c1, c23;
#if EXMC_ABLATE == 1
          c1 = c23 = false;
#else
          uturn3<G, DPL>(rho, a_pin, p, r2, a_pin, c_pin, r3, a_pout, p, im, valid, c1, c23);
#endif
          turning = c1 || ((lvl > 0) && c23);
#pragma unroll
          for (int k = 0; k < DPL; k++) { c_rho[k] = rho[k]; c_pin[k] = a_pin[k]; }
        }
        c_lsw = lsw;
        c_acc = a_acc + c_acc;
        c_n = (1 << lvl) + c_n;
        c_turn = turning;
        pending &= ~(1u << lvl);
        lvl++;
      } else if (c_div || c_turn) {
        lvl++;  // returned upward unmerged (tree.ex:1175-1177)
      } else {
        double nd[NSLOT];
#pragma unroll
        for (int k = 0; k < DPL; k++) {
          nd[0 * DPL + k] = c_rho[k];
          nd[1 * DPL + k] = c_pin[k];
          nd[2 * DPL + k] = p[k];
          nd[3 * DPL + k] = c_qp[k];
          nd[4 * DPL + k] = c_gp[k];
        }
        nd[5 * DPL + 0] = c_lsw;
        nd[5 * DPL + 1] = c_logpP;
        nd[5 * DPL + 2] = c_acc;
        if (lvl < LDSL) node_store<NSLOT>(lstk + (size_t)lvl * NSLOT * kNutsBlock, kNutsBlock, nd);
        else node_store<NSLOT>(gstk + (size_t)(lvl - LDSL) * NSLOT * nthreads, nthreads, nd);
        pending |= (1u << lvl);
        parked = true;
        break;
      }
    }
    if (parked) continue;

    // ---- subtree for this doubling is complete: merge_trajectories (tree.ex:1479-1568) ----
    {
#if EXMC_ABLATE == 2
      const double lsw = t_lsw + c_lsw;
      const double u = rng_uniform(trng);
      const bool use_sub = u < 0.5;
#else
      const double lsw = log_sum_exp(t_lsw, c_lsw);
      const double u = rng_uniform(trng);
      const bool use_sub = exmc_log(u) < (c_lsw - t_lsw);
#endif
      if (use_sub) {
        t_logpP = c_logpP;
#pragma unroll
        for (int k = 0; k < DPL; k++) { t_qp[k] = c_qp[k]; t_gp[k] = c_gp[k]; }
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
        uturn3<G, DPL>(rho, farp, p, r2, farp, c_pin, r3, nearp, p, im, valid, c1, c23);
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
      depth++;
    }

    if (depth >= P.max_depth || t_div || t_turn) {
      // ---- transition done (tree.ex:1607-1618, sampler.ex:890-925) ----
      (void)rng_uniform(rng);
      slogp = t_logpP;
#pragma unroll
      for (int k = 0; k < DPL; k++) { sq[k] = t_qp[k]; sg[k] = t_gp[k]; }
      const size_t row = (size_t)(P.draw_offset + draw);
      if (P.tr.draws) {
#pragma unroll
        for (int k = 0; k < DPL; k++)
          if (valid[k]) P.tr.draws[(row * D + (l + k * G)) * C + chain] = sq[k];
      }
      if (l == 0) {
        const size_t o = row * C + chain;
        if (P.tr.logp) P.tr.logp[o] = slogp;
        if (P.tr.tree_depth) P.tr.tree_depth[o] = depth;
        if (P.tr.n_steps) P.tr.n_steps[o] = t_n;
        if (P.tr.divergent) P.tr.divergent[o] = t_div ? 1 : 0;
        if (P.tr.accept_prob) P.tr.accept_prob[o] = (t_n > 0) ? (t_acc / (double)t_n) : 0.0;
        if (P.tr.energy) P.tr.energy[o] = -jlp0;
      }
      lf_total += (unsigned long long)t_n;
      div_total += t_div ? 1u : 0u;
      draw++;
      start_transition = true;
    } else {
      start_doubling = true;
    }
  }

  // write back chain state
  P.st.logp[chain] = slogp;
  if (l == 0) {
    P.st.rng[chain] = rng.a;
    P.st.rng[(size_t)C + chain] = rng.b;
    if (P.counters) {
      atomicAdd(&P.counters[0], lf_total);
      atomicAdd(&P.counters[1], div_total);
    }
  }
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    if (valid[k]) {
      const int i = l + k * G;
      P.st.q[(size_t)i * C + chain] = sq[k];
      P.st.g[(size_t)i * C + chain] = sg[k];
    }
  }
}

}  // namespace exmc
