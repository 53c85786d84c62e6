// exmc_native_tree.hpp — the Rust NIF's `build_full_tree` (native/exmc_tree/src/tree.rs:276-326,
// lib.rs:219-302), batched over chains on the GPU: one lane per chain walks pre-computed forward
// and backward leapfrog chains and builds the whole doubling tree.
//
// Semantics kept from the Rust crate where it differs from the Elixir path:
// a divergent leaf keeps the NEW state (tree.rs:44-95); KE = sum 0.5*p*m*p (tree.rs:57-61);
// log_sum_exp returns -inf (math.rs:3-10); the sub-trajectory checks are evaluated before the
// full-trajectory check (no observable difference); the loop stops when a direction's budget is
// exhausted (tree.rs:300-306); RNG = Xoshiro256** seeded by seed_from_u64 (SplitMix64 fill),
// f64 = (next >> 11) * 2^-53. exp/log go through the numeric contract (exmc_detmath.h).
//
// Nodes are index-based: every endpoint / proposal of a subtree is one of the pre-computed
// states, so a node is {first, last, prop indices, lsw, acc, n, depth} plus one rho vector.
// Layout (host = device): per chain row-major [step][dim], as the NIF's binaries (lib.rs:19-24).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/exmc_detmath.h"

namespace exmc {

struct FullTreeParams {
  int n_chains, d, n_fwd, n_bwd, max_depth;
  const double* q0;      // [C][d]
  const double* p0;
  const double* g0;
  const double* logp0;   // [C]
  const double* fwd_q;   // [C][n_fwd][d]
  const double* fwd_p;
  const double* fwd_g;
  const double* fwd_logp;  // [C][n_fwd]
  const double* bwd_q;
  const double* bwd_p;
  const double* bwd_g;
  const double* bwd_logp;
  const double* inv_mass;  // [d]
  const double* jlp0;      // [C]
  const uint64_t* seeds;   // [C]
  double* scratch;         // [C][(kFtLevels + 3)][d] rho vectors
  double* out_q;           // [C][d]
  double* out_g;
  double* out_logp;        // [C]
  double* out_accept_sum;
  int32_t* out_n_steps;
  int32_t* out_divergent;
  int32_t* out_depth;
};

constexpr int kFtLevels = 12;

struct Xoshiro {
  uint64_t s[4];
  __device__ __forceinline__ static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  __device__ __forceinline__ void seed_from_u64(uint64_t seed) {
    uint64_t x = seed;
    for (int i = 0; i < 4; i++) {
      uint64_t z = (x += 0x9e3779b97f4a7c15ULL);
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
      s[i] = z ^ (z >> 31);
    }
  }
  __device__ __forceinline__ uint64_t next() {
    const uint64_t result = rotl(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
  }
  __device__ __forceinline__ double f64() { return (double)(next() >> 11) * 0x1p-53; }
};

// a reference to one pre-computed state of this chain: src 0 = initial, 1 = fwd[idx], 2 = bwd[idx]
struct StateRef {
  int src, idx;
};

struct FtChain {
  int d;
  const double *q0, *p0, *g0;
  const double *fq, *fp, *fg, *flp;
  const double *bq, *bp, *bg, *blp;
  double logp0;
  __device__ __forceinline__ const double* P(StateRef r) const {
    return r.src == 0 ? p0 : (r.src == 1 ? fp + (size_t)r.idx * d : bp + (size_t)r.idx * d);
  }
  __device__ __forceinline__ const double* Q(StateRef r) const {
    return r.src == 0 ? q0 : (r.src == 1 ? fq + (size_t)r.idx * d : bq + (size_t)r.idx * d);
  }
  __device__ __forceinline__ const double* Gd(StateRef r) const {
    return r.src == 0 ? g0 : (r.src == 1 ? fg + (size_t)r.idx * d : bg + (size_t)r.idx * d);
  }
  __device__ __forceinline__ double LP(StateRef r) const {
    return r.src == 0 ? logp0 : (r.src == 1 ? flp[r.idx] : blp[r.idx]);
  }
};

// uturn.rs:8-24 with rho = ra + rb (rb may be null)
__device__ __forceinline__ bool ft_uturn(const double* ra, const double* rb, const double* pl,
                                         const double* pr, const double* im, int d) {
  double dr = 0.0, dl = 0.0;
  for (int i = 0; i < d; i++) {
    const double rho = rb ? (ra[i] + rb[i]) : ra[i];
    const double v = rho * im[i];
    dr += v * pr[i];
    dl += v * pl[i];
  }
  return dr < 0.0 || dl < 0.0;
}

__device__ __forceinline__ double ft_lse(double a, double b) {
  const double m = fmax(a, b);
  if (m == -exmc_from_bits(EXMC_INF_BITS)) return m;
  return m + exmc_log(exmc_exp(a - m) + exmc_exp(b - m));
}

struct FtNode {
  StateRef left, right, prop;   // spatial left / right endpoints, proposal
  double lsw, acc;
  int n, depth;
  bool div, turn;
};

// build_subtree (tree.rs:16-41) over the pre-computed states [base, base + 2^level) of source
// `src`, iteratively: a per-level stack of pending first halves, merge_subtrees (tree.rs:103-189)
// from the bottom up after every leaf, `if first.divergent || first.turning { return first; }`
// (tree.rs:31-33) as skipping upward. c_rho receives the node's rho; tmp and s_rho are scratch.
__device__ __forceinline__ void ft_build_subtree(const FtChain& ch, const double* im, double jlp0,
                                                 int src, int base, int level, bool go_right,
                                                 Xoshiro& rng, double* c_rho, double* tmp,
                                                 double* s_rho, FtNode& cur) {
  const int d = ch.d;
  const int n = 1 << level;
  FtNode stack[kFtLevels];
  unsigned pending = 0;
  cur.div = false; cur.turn = false;
  bool done = false;
  for (int leaf = 0; leaf < n && !done; leaf++) {
    // build_leaf (tree.rs:44-95)
    const StateRef sr{src, base + leaf};
    const double* p = ch.P(sr);
    double ke = 0.0;
    for (int i = 0; i < d; i++) ke += 0.5 * p[i] * im[i] * p[i];
    const double jlp = ch.LP(sr) - ke;
    cur.left = cur.right = cur.prop = sr;
    cur.n = 1; cur.depth = 0; cur.turn = false;
    if (exmc_isfinite(jlp)) {
      const double dl = jlp - jlp0;
      cur.div = dl < -1000.0;
      cur.lsw = dl;
      cur.acc = fmin(exmc_exp(fmin(dl, 0.0)), 1.0);
    } else {
      cur.div = true; cur.lsw = -1001.0; cur.acc = 0.0;
    }
    for (int i = 0; i < d; i++) c_rho[i] = p[i];
    int lvl = 0;
    for (;;) {
      if (lvl == level) { done = true; break; }
      if (pending & (1u << lvl)) {
        // merge_subtrees(first = stack[lvl], second = cur) (tree.rs:103-189)
        const FtNode& a = stack[lvl];
        const double* a_rho = s_rho + (size_t)lvl * d;
        const double lsw = ft_lse(a.lsw, cur.lsw);
        const bool divg = a.div || cur.div;
        const double u = rng.f64();
        const bool use_b = u < exmc_exp(cur.lsw - lsw);
        const FtNode& Ln = go_right ? a : cur;
        const FtNode& Rn = go_right ? cur : a;
        const double* L_rho = go_right ? a_rho : c_rho;
        const double* R_rho = go_right ? c_rho : a_rho;
        bool sub_turning = false;
        if (!divg && !cur.turn && a.depth > 0) {
          // check 2: left rho + first point of right; check 3: last point of left + right rho
          sub_turning = ft_uturn(L_rho, ch.P(Rn.left), ch.P(Ln.left), ch.P(Rn.left), im, d) ||
                        ft_uturn(ch.P(Ln.right), R_rho, ch.P(Ln.right), ch.P(Rn.right), im, d);
        }
        for (int i = 0; i < d; i++) tmp[i] = a_rho[i] + c_rho[i];
        const bool turning = divg || cur.turn || sub_turning ||
                             ft_uturn(tmp, nullptr, ch.P(Ln.left), ch.P(Rn.right), im, d);
        FtNode m;
        m.left = Ln.left; m.right = Rn.right;
        m.prop = use_b ? cur.prop : a.prop;
        m.lsw = lsw; m.acc = a.acc + cur.acc; m.n = a.n + cur.n;
        m.div = divg; m.turn = turning;
        m.depth = (a.depth > cur.depth ? a.depth : cur.depth) + 1;
        cur = m;
        for (int i = 0; i < d; i++) c_rho[i] = tmp[i];
        pending &= ~(1u << lvl);
        lvl++;
      } else if (cur.div || cur.turn) {
        lvl++;   // `if first.divergent || first.turning { return first; }` (tree.rs:31-33)
      } else {
        stack[lvl] = cur;
        for (int i = 0; i < d; i++) s_rho[(size_t)lvl * d + i] = c_rho[i];
        pending |= (1u << lvl);
        break;
      }
    }
  }
}

__global__ void __launch_bounds__(64) full_tree_kernel(FullTreeParams P)
#ifdef EXMC_COMMON_DECL_ONLY
;   // defined in the prebuilt exmc_common object (exmc_common.hip)
#else
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= P.n_chains) return;
  const int d = P.d;
  FtChain ch;
  ch.d = d;
  ch.q0 = P.q0 + (size_t)c * d; ch.p0 = P.p0 + (size_t)c * d; ch.g0 = P.g0 + (size_t)c * d;
  ch.fq = P.fwd_q + (size_t)c * P.n_fwd * d; ch.fp = P.fwd_p + (size_t)c * P.n_fwd * d;
  ch.fg = P.fwd_g + (size_t)c * P.n_fwd * d; ch.flp = P.fwd_logp + (size_t)c * P.n_fwd;
  ch.bq = P.bwd_q + (size_t)c * P.n_bwd * d; ch.bp = P.bwd_p + (size_t)c * P.n_bwd * d;
  ch.bg = P.bwd_g + (size_t)c * P.n_bwd * d; ch.blp = P.bwd_logp + (size_t)c * P.n_bwd;
  ch.logp0 = P.logp0[c];
  const double* im = P.inv_mass;
  const double jlp0 = P.jlp0[c];
  double* scr = P.scratch + (size_t)c * (kFtLevels + 3) * d;
  double* t_rho = scr;                 // trajectory rho
  double* c_rho = scr + d;             // rho of the node being ascended
  double* tmp = scr + 2 * (size_t)d;
  double* s_rho = scr + 3 * (size_t)d; // stack: level l at s_rho + l*d

  Xoshiro rng;
  rng.seed_from_u64(P.seeds[c]);

  // Trajectory::new (types.rs:129-152)
  FtNode T;
  T.left = T.right = T.prop = StateRef{0, 0};
  T.lsw = 0.0; T.acc = 0.0; T.n = 0; T.depth = 0; T.div = false; T.turn = false;
  for (int i = 0; i < d; i++) t_rho[i] = ch.p0[i];
  int fc = 0, bc = 0;

  for (int it = 0; it < P.max_depth; it++) {
    if (T.div || T.turn) break;
    const bool go_right = rng.f64() > 0.5;
    const int level = T.depth;
    const int n = 1 << level;
    if (go_right && fc + n > P.n_fwd) break;
    if (!go_right && bc + n > P.n_bwd) break;
    const int src = go_right ? 1 : 2;
    const int base = go_right ? fc : bc;

    FtNode cur;
    ft_build_subtree(ch, im, jlp0, src, base, level, go_right, rng, c_rho, tmp, s_rho, cur);
    if (go_right) fc += n; else bc += n;

    // ---- merge_into_trajectory (tree.rs:194-265) ----
    {
      const double lsw = ft_lse(T.lsw, cur.lsw);
      const bool divg = T.div || cur.div;
      bool sub_turning = false;
      if (!divg && !cur.turn) {
        const FtNode& Ln = go_right ? T : cur;
        const FtNode& Rn = go_right ? cur : T;
        const double* L_rho = go_right ? t_rho : c_rho;
        const double* R_rho = go_right ? c_rho : t_rho;
        sub_turning = ft_uturn(L_rho, ch.P(Rn.left), ch.P(Ln.left), ch.P(Rn.left), im, d) ||
                      ft_uturn(ch.P(Ln.right), R_rho, ch.P(Ln.right), ch.P(Rn.right), im, d);
      }
      const double u = rng.f64();
      if (exmc_log(u) < (cur.lsw - T.lsw)) T.prop = cur.prop;
      for (int i = 0; i < d; i++) t_rho[i] += c_rho[i];
      if (go_right) T.right = cur.right; else T.left = cur.left;
      const bool turning = divg || cur.turn || sub_turning ||
                           ft_uturn(t_rho, nullptr, ch.P(T.left), ch.P(T.right), im, d);
      T.lsw = lsw; T.n += cur.n; T.acc += cur.acc;
      T.div = divg; T.turn = turning; T.depth += 1;
    }
  }

  // trajectory_to_result (tree.rs:329-339)
  const double* q = ch.Q(T.prop);
  const double* g = ch.Gd(T.prop);
  for (int i = 0; i < d; i++) {
    P.out_q[(size_t)c * d + i] = q[i];
    P.out_g[(size_t)c * d + i] = g[i];
  }
  P.out_logp[c] = ch.LP(T.prop);
  P.out_accept_sum[c] = T.acc;
  P.out_n_steps[c] = T.n;
  P.out_divergent[c] = T.div ? 1 : 0;
  P.out_depth[c] = T.depth;
}
#endif

// ------------------------------------------------------------------------------------------
// The NIF's incremental interface (lib.rs:37-212, 345-434), batched over chains: a trajectory
// resource per chain that lives on the device between calls (init_trajectory, get_endpoint,
// build_and_merge, is_terminated, get_result) and the stateless build_subtree. A call hands in
// the 2^depth pre-computed leapfrog states of one doubling per chain; the subtree is built on
// indices into that chain (ft_build_subtree) and merged into the trajectory's own vectors.
// ------------------------------------------------------------------------------------------
struct TrajDev {
  double *qL, *pL, *gL, *qR, *pR, *gR, *qP, *gP, *rho;   // [C][d] each
  double *logpP, *lsw, *acc;                             // [C]
  int32_t *n, *depth, *div, *turn;                       // [C]
};

struct SubtreeParams {
  int n_chains, d, n_states;      // n_states = row count of the all_* arrays per chain
  const double* all_q;            // [C][n_states][d]
  const double* all_p;
  const double* all_g;
  const double* all_logp;         // [C][n_states]
  const double* inv_mass;         // [d]
  const double* jlp0;             // [C]
  const int32_t* depth;           // [C]; < 0: this chain is skipped
  const int32_t* go_right;        // [C]
  const uint64_t* seeds;          // [C]
  double* scratch;                // [C][kFtLevels + 2][d]
  TrajDev T;                      // build_and_merge: the trajectories to merge into
  TrajDev out;                    // build_subtree: the subtree record (n = n_steps)
};

__device__ __forceinline__ void ft_copy(double* dst, const double* src, int d) {
  for (int i = 0; i < d; i++) dst[i] = src[i];
}

__device__ __forceinline__ FtChain ft_sub_chain(const SubtreeParams& P, int c) {
  FtChain ch;
  const size_t row = (size_t)c * P.n_states;
  ch.d = P.d;
  ch.fq = P.all_q + row * P.d; ch.fp = P.all_p + row * P.d; ch.fg = P.all_g + row * P.d;
  ch.flp = P.all_logp + row;
  // only source 1 (the supplied states) is ever referenced; the other sources alias it
  ch.q0 = ch.bq = ch.fq; ch.p0 = ch.bp = ch.fp; ch.g0 = ch.bg = ch.fg; ch.blp = ch.flp;
  ch.logp0 = 0.0;
  return ch;
}

// build_and_merge_bin (lib.rs:73-112): Xoshiro seeded per call, build_subtree, merge_into_trajectory
__global__ void __launch_bounds__(64) traj_build_and_merge_kernel(SubtreeParams P)
#ifdef EXMC_COMMON_DECL_ONLY
;   // defined in the prebuilt exmc_common object (exmc_common.hip)
#else
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= P.n_chains) return;
  const int level = P.depth[c];
  if (level < 0) return;
  const int d = P.d;
  const FtChain ch = ft_sub_chain(P, c);
  const double* im = P.inv_mass;
  const bool go_right = P.go_right[c] != 0;
  double* scr = P.scratch + (size_t)c * (kFtLevels + 2) * d;
  double* c_rho = scr;
  double* tmp = scr + d;
  double* s_rho = scr + 2 * (size_t)d;
  Xoshiro rng;
  rng.seed_from_u64(P.seeds[c]);
  FtNode cur;
  ft_build_subtree(ch, im, P.jlp0[c], 1, 0, level, go_right, rng, c_rho, tmp, s_rho, cur);

  // merge_into_trajectory (tree.rs:194-265) on the trajectory's own vectors
  const size_t o = (size_t)c * d;
  double *t_rho = P.T.rho + o, *t_pL = P.T.pL + o, *t_pR = P.T.pR + o;
  const bool t_div = P.T.div[c] != 0;
  const double t_lsw = P.T.lsw[c];
  const double lsw = ft_lse(t_lsw, cur.lsw);
  const bool divg = t_div || cur.div;
  bool sub_turning = false;
  if (!divg && !cur.turn) {
    if (go_right)   // left = trajectory, right = subtree
      sub_turning = ft_uturn(t_rho, ch.P(cur.left), t_pL, ch.P(cur.left), im, d) ||
                    ft_uturn(t_pR, c_rho, t_pR, ch.P(cur.right), im, d);
    else            // left = subtree, right = trajectory
      sub_turning = ft_uturn(c_rho, t_pL, ch.P(cur.left), t_pL, im, d) ||
                    ft_uturn(ch.P(cur.right), t_rho, ch.P(cur.right), t_pR, im, d);
  }
  const double u = rng.f64();
  if (exmc_log(u) < (cur.lsw - t_lsw)) {
    ft_copy(P.T.qP + o, ch.Q(cur.prop), d);
    ft_copy(P.T.gP + o, ch.Gd(cur.prop), d);
    P.T.logpP[c] = ch.LP(cur.prop);
  }
  for (int i = 0; i < d; i++) t_rho[i] += c_rho[i];
  if (go_right) {
    ft_copy(P.T.qR + o, ch.Q(cur.right), d); ft_copy(t_pR, ch.P(cur.right), d);
    ft_copy(P.T.gR + o, ch.Gd(cur.right), d);
  } else {
    ft_copy(P.T.qL + o, ch.Q(cur.left), d); ft_copy(t_pL, ch.P(cur.left), d);
    ft_copy(P.T.gL + o, ch.Gd(cur.left), d);
  }
  const bool turning = divg || cur.turn || sub_turning || ft_uturn(t_rho, nullptr, t_pL, t_pR, im, d);
  P.T.lsw[c] = lsw;
  P.T.n[c] += cur.n;
  P.T.acc[c] += cur.acc;
  P.T.div[c] = divg ? 1 : 0;
  P.T.turn[c] = turning ? 1 : 0;
  P.T.depth[c] += 1;
}
#endif

// build_subtree_bin (lib.rs:114-212): the subtree record itself
__global__ void __launch_bounds__(64) build_subtree_kernel(SubtreeParams P)
#ifdef EXMC_COMMON_DECL_ONLY
;   // defined in the prebuilt exmc_common object (exmc_common.hip)
#else
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= P.n_chains) return;
  const int level = P.depth[c];
  if (level < 0) return;
  const int d = P.d;
  const FtChain ch = ft_sub_chain(P, c);
  double* scr = P.scratch + (size_t)c * (kFtLevels + 2) * d;
  Xoshiro rng;
  rng.seed_from_u64(P.seeds[c]);
  FtNode cur;
  ft_build_subtree(ch, P.inv_mass, P.jlp0[c], 1, 0, level, P.go_right[c] != 0, rng, scr, scr + d,
                   scr + 2 * (size_t)d, cur);
  const size_t o = (size_t)c * d;
  ft_copy(P.out.qL + o, ch.Q(cur.left), d); ft_copy(P.out.pL + o, ch.P(cur.left), d);
  ft_copy(P.out.gL + o, ch.Gd(cur.left), d);
  ft_copy(P.out.qR + o, ch.Q(cur.right), d); ft_copy(P.out.pR + o, ch.P(cur.right), d);
  ft_copy(P.out.gR + o, ch.Gd(cur.right), d);
  ft_copy(P.out.qP + o, ch.Q(cur.prop), d); ft_copy(P.out.gP + o, ch.Gd(cur.prop), d);
  ft_copy(P.out.rho + o, scr, d);
  P.out.logpP[c] = ch.LP(cur.prop);
  P.out.lsw[c] = cur.lsw;
  P.out.acc[c] = cur.acc;
  P.out.n[c] = cur.n;
  P.out.depth[c] = cur.depth;
  P.out.div[c] = cur.div ? 1 : 0;
  P.out.turn[c] = cur.turn ? 1 : 0;
}
#endif

}  // namespace exmc
