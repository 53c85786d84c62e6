/* exmc_oracle.h — CPU restatement of eXMC's NUTS hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is the parity checker: a scalar f64 C restatement of the reference's
 * pure-Elixir sampler path (lib/exmc/nuts/{sampler,tree,leapfrog,batched_leapfrog,
 * step_size,mass_matrix}.ex) and of the Rust NIF semantics (native/exmc_tree/src/, all .rs).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (exmc_amd/, libexmc_hip.so) never links, imports or calls anything here.
 *
 * PARITY UNPINNED: the reference needs Elixir/OTP + Rust, neither of
 * which exists in this pipeline, and it commits no golden per-draw traces. The oracle
 * is therefore pinned only by the reference's own known-answer literals and invariants
 * (transcribed under tests/golden/) and by published third-party vectors
 * (SplitMix64, xoshiro256**). OTP's :rand exsss/ziggurat is restated from its published
 * algorithm; its constant tables are regenerated (tools/gen_zig_tables.py), not copied.
 *
 * Two numeric modes (exo_cfg):
 *   math_mode 0 = libm exp/log (what :math / Nx.BinaryBackend call on a BEAM host);
 *   math_mode 1 = include/exmc_detmath.h (the bit-reproducible contract the HIP
 *                 kernels follow; differs from libm by <= 1 ulp per call).
 *   lanes G     = 1: every reduction is a left-to-right sum (the reference's order);
 *               > 1: per-lane partial sums over dims l, l+G, ... then an xor-butterfly,
 *                    which is the order a G-lane chain group uses on the GPU.
 */
#ifndef EXMC_ORACLE_H
#define EXMC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXO_MAX_D 256
#define EXO_MAX_DEPTH 12

enum {
  EXO_MODEL_STD_NORMAL = 0,   /* d independent N(0,1) terms, Normal.logpdf form */
  EXO_MODEL_SIMPLE = 1,       /* d=2 : mu ~ N(0,5), sigma ~ Exponential(1)[:log], y ~ N(mu,sigma) */
  EXO_MODEL_EIGHT_SCHOOLS = 2,/* d=10: non-centered, validate_posteriordb.exs:246-324 */
  EXO_MODEL_SV = 3,           /* d=T+2: stochastic volatility, STANDARD_BENCHMARKS.md:51-61 */
  EXO_MODEL_LOGISTIC = 4,     /* d=K+1: logistic regression, STANDARD_BENCHMARKS.md:41-49 */
  EXO_MODEL_RADON = 5,        /* d=J+5: hierarchical radon, notebooks/09_radon_bhm.livemd */
  EXO_MODEL_CUSTOM = 6        /* a generated model (exmc_amd/codegen.py): logp+grad via a function pointer */
};

typedef struct {
  int math_mode;
  int lanes;
} exo_cfg;

typedef struct { uint64_t a, b; } exo_rng;  /* OTP exsss state [a|b] */

typedef struct exo_model exo_model;

/* ---- RNG: OTP :rand exsss (call sites sampler.ex:154,343,396,836,897; tree.ex:403,1397,1489) */
void exo_rng_seed(exo_rng* r, uint64_t seed);
uint64_t exo_rng_next(exo_rng* r);
double exo_rng_uniform(exo_rng* r);
double exo_rng_normal(exo_rng* r, int math_mode);
uint64_t exo_splitmix64(uint64_t* x);
/* Rust side: rand_xoshiro 0.6.0 Xoshiro256StarStar (lib.rs:96,137,262,385) */
void exo_xoshiro_seed_from_u64(uint64_t s[4], uint64_t seed);
uint64_t exo_xoshiro_next(uint64_t s[4]);
double exo_xoshiro_f64(uint64_t s[4]);

/* ---- math */
double exo_exp(double x, int math_mode);
double exo_log(double x, int math_mode);
double exo_log1p(double x, int math_mode);
double exo_det_exp(double x);
double exo_det_log(double x);
double exo_det_log1p(double x);
double exo_lgamma_lanczos(double x, int math_mode);          /* math.ex:27-52 */
double exo_log_sum_exp(double a, double b, int math_mode);   /* tree.ex:1597-1605 */

/* ---- models */
exo_model* exo_model_create(int kind, int d, const double* data, int n_data);
void exo_model_free(exo_model* m);
int exo_model_dim(const exo_model* m);
/* EXO_MODEL_CUSTOM: fn(data, q, grad) -> logp is the C restatement emitted by the code generator
 * from the same expression graph as the HIP functor (deterministic-math contract only). */
typedef double (*exo_custom_fn)(const double* data, const double* q, double* grad);
void exo_model_set_custom(exo_model* m, exo_custom_fn fn);
/* flat[r] = kernel dimension of the r-th entry of the reference's flat vector (PointMap.build sorts
 * free-RV ids as strings, point_map.ex:30-60); the RNG-consuming steps (init_position,
 * sample_momentum_fast: sampler.ex:339-349, 393-403) draw in that order. Defaults: the string sort
 * of the kind's names for sv and logistic, identity otherwise. Returns -1 if not a permutation. */
int exo_model_set_flat_order(exo_model* m, const int* flat);
void exo_model_get_flat_order(const exo_model* m, int* flat);
double exo_logp_grad(const exo_model* m, const double* q, double* grad, exo_cfg cfg);
void exo_constrain(const exo_model* m, const double* q, double* x); /* Transform.apply per entry */
/* distribution known answers (dist/<name>.ex doctests) */
double exo_dist_normal(double x, double mu, double sigma, int math_mode);
double exo_dist_half_cauchy(double x, double scale, int math_mode);
double exo_dist_exponential(double x, double lambda, int math_mode);
double exo_dist_student_t(double x, double df, double loc, double scale, int math_mode);
double exo_dist_half_normal(double x, double sigma, int math_mode);
double exo_dist_bernoulli(double x, double p, int math_mode);

/* ---- leapfrog (leapfrog.ex:14-61, batched_leapfrog.ex:50-101) */
double exo_kinetic_energy(const double* p, const double* inv_mass, int d, exo_cfg cfg);
/* one step; returns logp'; writes q,p,g in place; *jlp = logp' - KE(p') */
double exo_leapfrog(const exo_model* m, double* q, double* p, double* g, double eps,
                    const double* inv_mass, double* jlp, exo_cfg cfg);
/* B2 multi_step: rows [n][d] row-major, logp [n] raw logp */
void exo_multi_step(const exo_model* m, const double* q, const double* p, const double* g,
                    double eps, const double* inv_mass, int n_steps, double* all_q,
                    double* all_p, double* all_logp, double* all_g, exo_cfg cfg);

/* B2' fused-chain hook (tree.ex:613-653): k leapfrog steps of one chain of d independent Normal(mu, sigma)
 * coordinates from (q, p), the first gradient taken at q; rows [k][d], raw logp [k]. 0, or -1 on bad sizes. */
int exo_leapfrog_chain_normal(const double* q, const double* p, const double* inv_mass, int d, int k,
                              double signed_eps, double mu, double sigma, double* q_chain,
                              double* p_chain, double* grad_chain, double* logp_chain, exo_cfg cfg);

/* ---- one NUTS tree (tree.ex:266-500) from state (q,p,logp,g); rng is COPIED */
typedef struct {
  double logp;
  int n_steps;
  int divergent;
  double accept_sum;
  int depth;
} exo_tree_result;
void exo_tree_build(const exo_model* m, const double* q, const double* p, double logp,
                    const double* g, double eps, const double* inv_mass, int max_depth,
                    exo_rng rng, double jlp0, double* q_out, double* g_out,
                    exo_tree_result* res, exo_cfg cfg);
int exo_check_uturn(const double* rho, const double* pl, const double* pr, const double* inv_mass,
                    int d, exo_cfg cfg);

/* ---- adaptation (step_size.ex:13-50, mass_matrix.ex:40-97, sampler.ex:764-785) */
typedef struct {
  double log_epsilon, log_epsilon_bar, h_bar, mu;
  int m;
  double gamma, t0, kappa, target_accept;
  int math_mode;
} exo_da;
void exo_da_init(exo_da* s, double epsilon, double target_accept);          /* libm */
void exo_da_init_mode(exo_da* s, double epsilon, double target_accept, int math_mode);
void exo_da_update(exo_da* s, double accept_stat);
double exo_da_finalize(const exo_da* s);
typedef struct {
  int n, d;
  double mean[EXO_MAX_D], m2[EXO_MAX_D];
} exo_welford;
void exo_welford_init(exo_welford* w, int d);
void exo_welford_update(exo_welford* w, const double* q);
void exo_welford_finalize(const exo_welford* w, double* inv_mass);
int exo_build_windows(int from, int to, int base, int* starts, int* ends, int max_windows);

/* ---- sampler (sampler.ex:126-257, 1020-1136) */
typedef struct {
  int num_warmup, num_samples, max_tree_depth;
  double target_accept;
  uint64_t seed;
} exo_opts;

typedef struct {
  double step_size;
  double inv_mass[EXO_MAX_D];
  int divergences;          /* warmup + sampling, as stats.divergences (sampler.ex:245) */
  long total_leapfrogs;     /* sum of n_steps over the sampling draws */
} exo_stats;

/* Per-draw outputs; any pointer may be NULL. draws: [num_samples][d] unconstrained. */
typedef struct {
  double* draws;
  double* logp;
  int* tree_depth;
  int* n_steps;
  int* divergent;
  double* accept_prob;
  double* energy;
} exo_trace;

/* init_q NULL => 0.1*normal_s per dim (sampler.ex:339-349) */
int exo_sample(const exo_model* m, const double* init_q, exo_opts o, exo_trace tr, exo_stats* st,
               exo_cfg cfg);
/* shared warmup on chain 0 (plain path), then chains seeded seed+7919*i (sampler.ex:1053-1130).
 * Traces are chain-major: draws [n_chains][num_samples][d] etc. chain_lo..chain_hi selects a
 * sub-range of chains to run (for sharding); n_threads > 1 runs chains on host threads. */
int exo_sample_chains(const exo_model* m, const double* init_q, int n_chains, int chain_lo,
                      int chain_hi, exo_opts o, exo_trace tr, exo_stats* st, int n_threads,
                      exo_cfg cfg);
/* warmup only: returns tuned step size + inv_mass (used to hand the same tuning to the GPU) */
int exo_warmup(const exo_model* m, const double* init_q, exo_opts o, exo_stats* st, exo_cfg cfg);
/* opts[:warm_start] (sampler.ex:167-197): previous inv_mass_diag + step_size, min(num_warmup, 50)
 * warmup iterations with no initial step-size search, then num_samples draws */
int exo_sample_warm(const exo_model* m, const double* init_q, double prev_epsilon,
                    const double* prev_inv_mass, exo_opts o, exo_trace tr, exo_stats* st, exo_cfg cfg);
/* sampling with given tuning (sample_compiled_tuned, sampler.ex:260-335) */
int exo_sample_tuned(const exo_model* m, const double* init_q, double epsilon,
                     const double* inv_mass, exo_opts o, exo_trace tr, exo_stats* st, exo_cfg cfg);

/* ---- opts[:dense_mass] (mass_matrix.ex:27-35,56-72,105-140; sampler.ex:412-427,682): see the note
 * in exmc_oracle.c on what the reference itself does in this mode. cov / chol: row-major d x d. */
int exo_warmup_dense(const exo_model* m, const double* init_q, exo_opts o, exo_stats* st, double* cov,
                     double* chol, exo_cfg cfg);
int exo_sample_tuned_dense(const exo_model* m, const double* init_q, double epsilon, const double* cov,
                           const double* chol, exo_opts o, exo_trace tr, exo_stats* st, exo_cfg cfg);
int exo_cholesky_lower(const double* a, int d, double* l);
void exo_dense_mass_times(const double* cov, const double* x, int d, double* out);
int exo_dense_check_uturn(const double* cov, const double* rho, const double* pl, const double* pr, int d,
                          exo_cfg cfg);
void exo_dense_momentum(const exo_model* m, const double* chol, exo_rng* rng, double* p, int math_mode);
void exo_welford_dense_finalize(const double* draws, int n, int d, double* cov, double* chol);

/* ---- diagnostics (diagnostics.ex:42-167) */
double exo_ess(const double* x, int n);
double exo_ess_bulk(const double* x, int n);
double exo_ess_bulk_mode(const double* x, int n, int math_mode);  /* 1: deterministic log */
double exo_rhat(const double* chains, int n_chains, int n);  /* chains [n_chains][n] */

/* ---- NativeTree NIF semantics (native/exmc_tree/src/{tree,lib}.rs) */
typedef struct exo_nt_traj exo_nt_traj;
/* 0 (default) = libm as the Rust crate; 1 = exmc_detmath.h, the GPU entry point's contract */
void exo_nt_set_math_mode(int mode);
exo_nt_traj* exo_nt_init_trajectory(const double* q, const double* p, const double* g, double logp,
                                    int d);
void exo_nt_free(exo_nt_traj* t);
int exo_nt_is_terminated(const exo_nt_traj* t);
void exo_nt_get_endpoint(const exo_nt_traj* t, int go_right, double* q, double* p, double* g);
void exo_nt_build_and_merge(exo_nt_traj* t, const double* all_q, const double* all_p,
                            const double* all_logp, const double* all_g, const double* inv_mass,
                            double jlp0, int depth, int d, int go_right, uint64_t seed);
void exo_nt_get_result(const exo_nt_traj* t, double* q, double* g, exo_tree_result* res);
/* build_subtree_bin (lib.rs:114-212): vecs = qL,pL,gL,qR,pR,gR,qP,gP,rho (9*d doubles);
 * scalars = logp_prop, log_sum_weight, accept_sum; ints = n_steps, divergent, turning, depth */
void exo_nt_build_subtree(const double* all_q, const double* all_p, const double* all_logp,
                          const double* all_g, const double* inv_mass, double jlp0, int depth,
                          int d, int going_right, uint64_t seed, double* vecs, double* scalars,
                          int* ints);
void exo_nt_build_full_tree(const double* q0, const double* p0, const double* g0, double logp0,
                            const double* fwd_q, const double* fwd_p, const double* fwd_logp,
                            const double* fwd_g, int n_fwd, const double* bwd_q,
                            const double* bwd_p, const double* bwd_logp, const double* bwd_g,
                            int n_bwd, const double* inv_mass, double jlp0, int max_depth, int d,
                            uint64_t seed, double* q_out, double* g_out, exo_tree_result* res);

#ifdef __cplusplus
}
#endif
#endif
