// exmc_kernels.hpp — gfx950 kernels for the NUTS hot path.
//
//   nuts_kernel        whole NUTS transitions (momentum draw, iterative tree with multinomial
//                      proposals, rho-based U-turn checks, divergence guard) for every chain;
//                      replaces Tree.build/12 + nuts_step_with_stats (tree.ex:65-151,
//                      sampler.ex:854-925) and the Rust crate native/exmc_tree.
//   multi_step_kernel  B2 `multi_step_fn` contract, chain-batched (batched_leapfrog.ex:50-101).
//   init_chains_kernel seed :rand, init position, first logp/grad (sampler.ex:154-165,339-349).
//   find_eps_kernel    find_reasonable_epsilon_with_rng (sampler.ex:451-530).
//   logp_grad_kernel   vag_fn batched (compiler.ex:131-141).
//   ess_kernel         Diagnostics.ess (diagnostics.ex:42-52,123-167).
//
// The tree is built iteratively. The reference recursion (tree.ex:1144-1203) is unrolled into a
// per-level stack of pending "first halves": after leaf k completes, one inner merge fires for
// every pending level from the bottom up (tree.ex:1390-1476), each consuming one uniform; a node
// that is divergent or turning and has no pending sibling is returned upward unmerged exactly as
// `if first.divergent or first.turning` does (tree.ex:1175-1177). Main loop = one leapfrog per
// iteration for every chain group in the wave, so chains with short trees never wait for chains
// with long ones; the bookkeeping between leapfrogs diverges per group.
#pragma once

#include "exmc_models.hpp"

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
  double* stack;                // dev scratch, NLEV * NSLOT * nthreads doubles
  unsigned long long* counters; // [0] leapfrogs, [1] divergent transitions
  const uint64_t* zig_ki;
  const double* zig_wi;
  const double* zig_fi;
  double nor_r;
};

constexpr int kMaxLevels = 12;

template <class M>
__host__ __device__ constexpr int nuts_nslot() { return 5 * M::DPL + 3; }

template <class M, int G>
__global__ void __launch_bounds__(64) nuts_kernel(NutsParams P, typename M::Consts mc) {
  constexpr int D = M::D, DPL = M::DPL;
  constexpr int NSLOT = 5 * DPL + 3;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const int l = threadIdx.x & (G - 1);
  const int chain = tid / G;
  const int C = P.n_chains;
  if (chain >= C) return;

  typename M::Lane ln;
  M::load(mc, l, ln);
  const ZigTables zt{P.zig_ki, P.zig_wi, P.zig_fi};

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

  double* stk = P.stack + tid;
#define EXMC_STK(lvl, slot) stk[(size_t)((lvl) * NSLOT + (slot)) * nthreads]

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
      // sampler.ex:393-403 momentum; leapfrog.ex:49-51 joint logp; tree.ex:284-305 initial traj
      for (int i = 0; i < D; i++) {
        const double z = rng_normal(rng, zt, P.nor_r);
#pragma unroll
        for (int k = 0; k < DPL; k++)
          if (l + k * G == i) pL[k] = z / sim[k];
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
    const double logp_new = M::logp_grad(mc, ln, l, q, g);
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
      c_acc = fmin(1.0, exmc_exp(fmin(dl, 0.0)));
    } else {
      c_div = true;
      c_lsw = -1001.0;
      c_acc = 0.0;
    }
    if (c_div) {
      c_acc = 0.0;
      c_logpP = -1.0e30;
#pragma unroll
      for (int k = 0; k < DPL; k++) { c_qp[k] = qold[k]; c_gp[k] = gold[k]; }
    } else {
      c_logpP = logp_new;
#pragma unroll
      for (int k = 0; k < DPL; k++) { c_qp[k] = q[k]; c_gp[k] = g[k]; }
    }
#pragma unroll
    for (int k = 0; k < DPL; k++) { c_rho[k] = p[k]; c_pin[k] = p[k]; }

    // ---- ascend: inner merges for every pending level (tree.ex:1144-1203, 1390-1476) ----
    int lvl = 0;
    bool parked = false;
    while (lvl < depth) {
      if (pending & (1u << lvl)) {
        const double a_lsw = EXMC_STK(lvl, 5 * DPL + 0);
        const double a_logpP = EXMC_STK(lvl, 5 * DPL + 1);
        const double a_acc = EXMC_STK(lvl, 5 * DPL + 2);
        const double lsw = log_sum_exp(a_lsw, c_lsw);
        const double u = rng_uniform(trng);
        const bool use_b = u < exmc_exp(c_lsw - lsw);
        if (!use_b) {
          c_logpP = a_logpP;
#pragma unroll
          for (int k = 0; k < DPL; k++) {
            c_qp[k] = EXMC_STK(lvl, 3 * DPL + k);
            c_gp[k] = EXMC_STK(lvl, 4 * DPL + k);
          }
        }
        bool turning = c_div || c_turn;
        if (!turning) {
          double a_rho[DPL], a_pin[DPL], a_pout[DPL], rho[DPL], tmp[DPL];
#pragma unroll
          for (int k = 0; k < DPL; k++) {
            a_rho[k] = EXMC_STK(lvl, 0 * DPL + k);
            a_pin[k] = EXMC_STK(lvl, 1 * DPL + k);
            a_pout[k] = EXMC_STK(lvl, 2 * DPL + k);
            rho[k] = a_rho[k] + c_rho[k];
          }
          turning = uturn<G, DPL>(rho, a_pin, p, im, valid);
          if (!turning && lvl > 0) {
#pragma unroll
            for (int k = 0; k < DPL; k++) tmp[k] = a_rho[k] + c_pin[k];
            turning = uturn<G, DPL>(tmp, a_pin, c_pin, im, valid);
            if (!turning) {
#pragma unroll
              for (int k = 0; k < DPL; k++) tmp[k] = a_pout[k] + c_rho[k];
              turning = uturn<G, DPL>(tmp, a_pout, p, im, valid);
            }
          }
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
#pragma unroll
        for (int k = 0; k < DPL; k++) {
          EXMC_STK(lvl, 0 * DPL + k) = c_rho[k];
          EXMC_STK(lvl, 1 * DPL + k) = c_pin[k];
          EXMC_STK(lvl, 2 * DPL + k) = p[k];
          EXMC_STK(lvl, 3 * DPL + k) = c_qp[k];
          EXMC_STK(lvl, 4 * DPL + k) = c_gp[k];
        }
        EXMC_STK(lvl, 5 * DPL + 0) = c_lsw;
        EXMC_STK(lvl, 5 * DPL + 1) = c_logpP;
        EXMC_STK(lvl, 5 * DPL + 2) = c_acc;
        pending |= (1u << lvl);
        parked = true;
        break;
      }
    }
    if (parked) continue;

    // ---- subtree for this doubling is complete: merge_trajectories (tree.ex:1479-1568) ----
    {
      const double lsw = log_sum_exp(t_lsw, c_lsw);
      const double u = rng_uniform(trng);
      const bool use_sub = exmc_log(u) < (c_lsw - t_lsw);
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
        double nearp[DPL], farp[DPL], tmp[DPL];
#pragma unroll
        for (int k = 0; k < DPL; k++) {
          nearp[k] = go_right ? pR[k] : pL[k];
          farp[k] = go_right ? pL[k] : pR[k];
        }
        turning = uturn<G, DPL>(rho, farp, p, im, valid);
        if (!turning) {
#pragma unroll
          for (int k = 0; k < DPL; k++) tmp[k] = t_rho[k] + c_pin[k];
          turning = uturn<G, DPL>(tmp, farp, c_pin, im, valid);
          if (!turning) {
#pragma unroll
            for (int k = 0; k < DPL; k++) tmp[k] = nearp[k] + c_rho[k];
            turning = uturn<G, DPL>(tmp, nearp, p, im, valid);
          }
        }
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
#undef EXMC_STK

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
__global__ void __launch_bounds__(256) multi_step_kernel(MultiStepParams P, typename M::Consts mc) {
  constexpr int D = M::D, DPL = M::DPL;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = threadIdx.x & (G - 1);
  const int chain = tid / G;
  const int C = P.n_chains;
  if (chain >= C) return;
  typename M::Lane ln;
  M::load(mc, l, ln);
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
      if (valid[k]) {
        const size_t o = ((size_t)s * D + (l + k * G)) * C + chain;
        P.all_q[o] = q[k];
        P.all_p[o] = p[k];
        P.all_g[o] = g[k];
      }
    }
    if (l == 0) P.all_logp[(size_t)s * C + chain] = logp;
  }
}

// vag_fn batched: q [D][C] -> logp [C], grad [D][C]
template <class M, int G>
__global__ void __launch_bounds__(256)
logp_grad_kernel(const double* qin, int C, double* logp, double* grad, typename M::Consts mc) {
  constexpr int D = M::D, DPL = M::DPL;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = threadIdx.x & (G - 1);
  const int chain = tid / G;
  if (chain >= C) return;
  typename M::Lane ln;
  M::load(mc, l, ln);
  double q[DPL], g[DPL];
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = l + k * G;
    q[k] = (i < D) ? qin[(size_t)i * C + chain] : 0.0;
    g[k] = 0.0;
  }
  const double lp = M::logp_grad(mc, ln, l, q, g);
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
};

template <class M, int G>
__global__ void __launch_bounds__(256) init_chains_kernel(InitParams P, typename M::Consts mc) {
  constexpr int D = M::D, DPL = M::DPL;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = threadIdx.x & (G - 1);
  const int chain = tid / G;
  const int C = P.n_chains;
  if (chain >= C) return;
  typename M::Lane ln;
  M::load(mc, l, ln);
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
    for (int i = 0; i < D; i++) {
      const double z = rng_normal(rng, zt, P.nor_r);
#pragma unroll
      for (int k = 0; k < DPL; k++)
        if (l + k * G == i) q[k] = z * 0.1;
    }
  }
  const double lp = M::logp_grad(mc, ln, l, q, g);
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

// find_reasonable_epsilon_with_rng (sampler.ex:451-530) for chain 0 of the state buffers
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
};

template <class M, int G>
__global__ void __launch_bounds__(64) find_eps_kernel(FindEpsParams P, typename M::Consts mc) {
  constexpr int D = M::D, DPL = M::DPL;
  const int l = threadIdx.x & (G - 1);
  if (threadIdx.x >= G || blockIdx.x != 0) return;
  const int chain = 0;
  const int C = P.n_chains;
  typename M::Lane ln;
  M::load(mc, l, ln);
  const ZigTables zt{P.zig_ki, P.zig_wi, P.zig_fi};
  double q0[DPL], g0[DPL], p0[DPL], im[DPL], sim[DPL];
  bool valid[DPL];
#pragma unroll
  for (int k = 0; k < DPL; k++) {
    const int i = l + k * G;
    valid[k] = i < D;
    q0[k] = valid[k] ? P.st.q[(size_t)i * C + chain] : 0.0;
    g0[k] = valid[k] ? P.st.g[(size_t)i * C + chain] : 0.0;
    im[k] = valid[k] ? P.inv_mass[i] : 1.0;
    sim[k] = valid[k] ? P.sqrt_inv_mass[i] : 1.0;
    p0[k] = 0.0;
  }
  const double logp0 = P.st.logp[chain];
  Rng rng;
  rng.a = P.st.rng[chain];
  rng.b = P.st.rng[(size_t)C + chain];
  for (int i = 0; i < D; i++) {
    const double z = rng_normal(rng, zt, P.nor_r);
#pragma unroll
    for (int k = 0; k < DPL; k++)
      if (l + k * G == i) p0[k] = z / sim[k];
  }
  const double jlp0 = logp0 - kinetic_energy<G, DPL>(p0, im, valid);
  auto try_eps = [&](double eps) -> double {
    double q[DPL], p[DPL], g[DPL];
    const double h = eps / 2.0;
#pragma unroll
    for (int k = 0; k < DPL; k++) {
      const double ph = p0[k] + h * g0[k];
      p[k] = ph;
      q[k] = q0[k] + eps * (im[k] * ph);
      g[k] = 0.0;
    }
    const double lp = M::logp_grad(mc, ln, l, q, g);
#pragma unroll
    for (int k = 0; k < DPL; k++) p[k] = p[k] + h * g[k];
    const double jlp = lp - kinetic_energy<G, DPL>(p, im, valid);
    return (exmc_isfinite(jlp0) && exmc_isfinite(jlp)) ? (jlp - jlp0) : -1000.0;
  };
  double eps = 1.0;
  double la = try_eps(eps);
  const double dir = (la > P.log_half) ? 1.0 : -1.0;
  const double factor = (dir > 0) ? 2.0 : 0.5;
  double result = 0.0;
  bool done = false;
  for (int count = 0; count < 100 && !done; count++) {
    const double ne = eps * factor;
    la = try_eps(ne);
    const bool crossed = (dir > 0) ? (la < P.log_half) : (la > P.log_half);
    if (crossed || !exmc_isfinite(la)) {
      result = fmax(ne, 1.0e-10);
      done = true;
    } else {
      eps = ne;
    }
  }
  if (!done) result = fmax(eps, 1.0e-10);
  if (l == 0) {
    *P.eps_out = result;
    P.st.rng[chain] = rng.a;
    P.st.rng[(size_t)C + chain] = rng.b;
  }
}

// Diagnostics.ess (diagnostics.ex:42-52, 123-167): one workgroup per (dim, chain) series of a
// [S][D][C] trace; lag sums are left-to-right per lag as the reference computes them.
__global__ void __launch_bounds__(256)
ess_kernel(const double* draws, int S, int D, int C, double* ess_out) {
  extern __shared__ double sh[];  // S centred values + S acf values
  double* c = sh;
  double* acf = sh + S;
  const int series = blockIdx.x;  // dim * C + chain
  const double* x = draws + series;
  const size_t stride = (size_t)D * C;
  __shared__ double s_mean, s_var;
  if (threadIdx.x == 0) {
    double sum = 0.0;
    for (int i = 0; i < S; i++) sum += x[(size_t)i * stride];
    s_mean = sum / S;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += blockDim.x) c[i] = x[(size_t)i * stride] - s_mean;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int i = 0; i < S; i++) v += c[i] * c[i];
    s_var = v;
  }
  __syncthreads();
  const double var = s_var;
  for (int lag = threadIdx.x; lag < S; lag += blockDim.x) {
    double sum = 0.0;
    for (int i = 0; i < S - lag; i++) sum += c[i] * c[i + lag];
    acf[lag] = (var != 0.0) ? (sum / var) : 0.0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double result;
    if (S < 4) {
      result = S * 1.0;
    } else {
      double tau = -1.0;
      if (var != 0.0) {
        const int max_k = (S - 1) / 2;
        for (int k = 0; k <= max_k; k++) {
          const double r0 = (2 * k < S) ? acf[2 * k] : 0.0;
          const double r1 = (2 * k + 1 < S) ? acf[2 * k + 1] : 0.0;
          const double pair = r0 + r1;
          if (pair > 0) tau += 2 * pair;
          else break;
        }
      }
      result = S / fmax(tau, 1.0);
    }
    ess_out[series] = result;
  }
}

}  // namespace exmc
