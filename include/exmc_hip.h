/* exmc_hip.h — C ABI of libexmc_hip.so: the MI355X (gfx950) NUTS inner loop for eXMC.
 *
 * Drop-in boundary. Each entry point names the reference interface it replaces
 * (paths under the reference repository). Plain pointers and sizes only; integer error codes;
 * the library never calls back into the host VM; one HIP stream per model handle.
 *
 * Memory conventions
 *   "dev"  pointers are device (HBM) pointers on the handle's GPU.
 *   "host" pointers are ordinary host memory; the call copies in/out (PCIe inclusive).
 *   Chain-batched device layout: vectors are [dim][chain] (chain index fastest, so a 64-lane
 *   wavefront reads 64 consecutive chains), scalars are [chain], traces are
 *   [draw][dim][chain] and [draw][chain].
 *   Host layout follows the reference NIF: native-endian f64, row-major [chain][step][dim]
 *   (native/exmc_tree/src/lib.rs:19-24, types.rs:36-43).
 *
 * Numeric contract: IEEE f64 throughout (lib/exmc/jit.ex:90-98 => :f64); exp/log through
 * include/exmc_detmath.h; reductions over a chain's dimensions in the G-lane order documented
 * in DESIGN.md (G = lanes_per_chain; G = 1 is the reference's left-to-right order).
 *
 * Environment switches (read at call time; every setting gives bit-identical results):
 *   EXMC_HIP_HOST_WARMUP=1       adaptation driven from the host, one launch per transition
 *   EXMC_HIP_WARMUP_PIPE=0|1     one-wave / two-wave (tree + integrator) warmup kernel;
 *                                default: two-wave where the model gains from it
 *   EXMC_HIP_WARMUP_REPLICAS=N   workgroups racing through the same warmup chain (default 32)
 *   EXMC_HIP_NUTS_PIPE=1         wave pairs in the sampling kernel too (eight_schools, 16 lanes)
 *   EXMC_HIP_NUTS_WG=0|1         logistic at 16 lanes per chain: the one-wave sampling kernel / workgroups of
 *                                eight wavefronts around one LDS image of the design matrix; default: the
 *                                workgroup form when a launch has more wavefronts than the device has SIMDs
 *   EXMC_HIP_RANK_SORT=0         exmc_hip_ess_bulk: ranks by counting instead of by sorting the series in LDS
 */
#ifndef EXMC_HIP_H
#define EXMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXMC_HIP_MAX_D 256

/* error codes */
enum {
  EXMC_OK = 0,
  EXMC_ERR_BADARG = 1,      /* NIF badarg equivalent (lib.rs NifResult) */
  EXMC_ERR_NO_DEVICE = 2,   /* no HIP device / extension cannot run: callers must fail loudly */
  EXMC_ERR_HIP = 3,         /* a HIP runtime call failed; see exmc_hip_last_error() */
  EXMC_ERR_UNSUPPORTED = 4  /* model kind / lanes_per_chain combination not compiled in */
};

/* model kinds: the BASELINE.json configs */
enum {
  EXMC_MODEL_STD_NORMAL = 0,
  EXMC_MODEL_SIMPLE = 1,
  EXMC_MODEL_EIGHT_SCHOOLS = 2,
  EXMC_MODEL_SV = 3,
  EXMC_MODEL_LOGISTIC = 4,
  EXMC_MODEL_RADON = 5,
  /* a model generated from Builder IR (compiler.ex:46-58 -> exmc_amd/codegen.py); present only
   * in a plug-in build of this library made for that model; data = the generator's data vector */
  EXMC_MODEL_CUSTOM = 6
};

typedef struct exmc_hip_model exmc_hip_model;

/* Sampler options: Exmc.NUTS.Sampler @default_opts (lib/exmc/nuts/sampler.ex:16-23). */
typedef struct {
  int num_warmup;       /* 1000 */
  int num_samples;      /* 1000 */
  int max_tree_depth;   /* 10   */
  double target_accept; /* 0.8  */
  uint64_t seed;        /* 0    */
  int lanes_per_chain;  /* G: 0 = library default for the model */
} exmc_hip_opts;

/* Tuning hand-off: the `tuning` map of sample_compiled_tuned (sampler.ex:62-71) and
 * Distributed's %{epsilon, inv_mass, chol_cov: nil} (lib/exmc/nuts/distributed.ex:141-146). */
typedef struct {
  double epsilon;
  double inv_mass[EXMC_HIP_MAX_D];
  int warmup_divergences;
} exmc_hip_tuning;

/* Per-draw outputs = stats.sample_stats + draws (sampler.ex:242-250, 960-967).
 * Any pointer may be NULL. */
typedef struct {
  double* draws;       /* unconstrained position q */
  double* logp;
  int32_t* tree_depth;
  int32_t* n_steps;
  int32_t* divergent;
  double* accept_prob;
  double* energy;
} exmc_hip_trace;

const char* exmc_hip_last_error(void);
int exmc_hip_device_count(void);

/* Replaces Compiler.compile_for_sampling/2 for the built-in model kinds
 * (lib/exmc/compiler.ex:46-58): uploads model data, prepares per-model constants.
 * `data`/`n_data` per kind: EIGHT_SCHOOLS y[8],sigma[8]; SIMPLE y[n]; SV r[100];
 * LOGISTIC X[N][20] row-major then y[N]; RADON u[85], county_start[86], floor[N], y[N] with the
 * observations sorted by county, N <= 1024 (EXMC_ERR_UNSUPPORTED above: the 64-lane layout gives a
 * lane 16 observation slots). Free variables are in "kernel order" (DESIGN.md section 2). */
int exmc_hip_model_create(int kind, int d, const double* data, int n_data, int device,
                          exmc_hip_model** out);
void exmc_hip_model_destroy(exmc_hip_model* m);
/* Replaces PointMap.build's layout decision (lib/exmc/point_map.ex:30-60: free RVs sorted by id
 * as strings) for the RNG-consuming steps: init_position (sampler.ex:339-349) and
 * sample_momentum_fast (sampler.ex:393-403) draw one normal_s per entry of that flat vector, front
 * to back. perm[r] = kernel dimension of flat entry r (d entries). The compute layout, traces and
 * inv_mass stay in kernel order. Defaults at create: the string sort of the kind's own names for
 * SV (nu, s_1, s_10, s_100, s_11, ...) and LOGISTIC (alpha, beta_1, beta_10, ...); identity for
 * the kinds whose kernel order is sorted already; RADON's county order depends on its data, so its
 * caller passes the order. Evicts resident chains. */
int exmc_hip_model_set_flat_order(exmc_hip_model* m, const int32_t* perm, int d);
int exmc_hip_model_dim(const exmc_hip_model* m);
int exmc_hip_model_default_lanes(const exmc_hip_model* m);
/* lanes_per_chain that suits the shared one-chain warmup (sampler.ex:1053-1080) when it differs
 * from the sampling layout (logistic: 64; a generated lane layout of fewer than 64 lanes: 64, the
 * chain's model terms over the whole wavefront). exmc_hip_warmup / _warmup_from with
 * lanes_per_chain = 0 run in it. The tuning it returns is layout-independent; chains that continue
 * the warmup chain itself (sample_host, stream) keep one layout for both phases. */
int exmc_hip_model_default_warmup_lanes(const exmc_hip_model* m);
/* lanes_per_chain of the layout that carries a dense mass matrix (opts[:dense_mass]) for this
 * model: 1 where a whole chain fits one lane (eight_schools -- 16 also works --, simple, generated
 * models), the lane layout of the kinds that have no one-lane form (sv 64, radon 64, logistic 16). */
int exmc_hip_model_default_dense_lanes(const exmc_hip_model* m);
/* the model handle's HIP stream (hipStream_t as void*) */
void* exmc_hip_model_stream(const exmc_hip_model* m);

/* vag_fn batched (compiler.ex:131-141): q host [C][d] -> logp host [C], grad host [C][d]. */
int exmc_hip_logp_grad_host(exmc_hip_model* m, const double* q, int n_chains, int lanes,
                            double* logp, double* grad);

/* multi_step_fn, chain-batched (lib/exmc/nuts/batched_leapfrog.ex:21-48, :50-101).
 * Device form: q,p,g dev [d][C]; outputs dev all_q/all_p/all_g [n][d][C], all_logp [n][C]
 * (raw logp, not joint; batched_leapfrog.ex:87). eps may be negative (tree.ex:516-519).
 * Only n_steps rows are written. */
int exmc_hip_multi_step(exmc_hip_model* m, const double* q, const double* p, const double* g,
                        double eps, const double* inv_mass_host, int n_steps, int n_chains,
                        int lanes, double* all_q, double* all_p, double* all_logp, double* all_g);
/* Host form, reference layout: q,p,g [C][d]; outputs [C][n][d], [C][n]. */
int exmc_hip_multi_step_host(exmc_hip_model* m, const double* q, const double* p, const double* g,
                             double eps, const double* inv_mass, int n_steps, int n_chains,
                             int lanes, double* all_q, double* all_p, double* all_logp,
                             double* all_g);

/* One NUTS transition per chain from explicit state, for parity tests of Tree.build/12
 * (lib/exmc/nuts/tree.ex:65-151) + nuts_step_with_stats (sampler.ex:854-925).
 * Host in/out: q [C][d], logp [C], grad [C][d], rng [C][2] (exsss words a,b). */
int exmc_hip_transitions_host(exmc_hip_model* m, double* q, double* logp, double* grad,
                              uint64_t* rng, int n_chains, int n_draws, double eps,
                              const double* inv_mass, int max_depth, int lanes,
                              exmc_hip_trace trace /* host, [C][n_draws][..] */);

/* Shared warmup on chain 0 (sampler.ex:1053-1080 -> run_warmup :537-621): transitions run on
 * the GPU, dual averaging / Welford / window schedule on the host as the reference does
 * (step_size.ex, mass_matrix.ex are plain Erlang floats there too).
 * init_q NULL => 0.1*normal_s per dim (sampler.ex:339-349). */
int exmc_hip_warmup(exmc_hip_model* m, const double* init_q, exmc_hip_opts opts,
                    exmc_hip_tuning* tuning);
/* opts[:dense_mass] (lib/exmc/nuts/mass_matrix.ex:27-35,56-72,105-140; sampler.ex:412-427,682;
 * leapfrog.ex:39-61). Warmup with dense Welford windows of base max(25, 10 d): tuning gets epsilon and
 * inv_mass = diag(cov) (stats.inv_mass_diag, sampler.ex:236-240), cov / chol (caller-owned, row-major
 * d x d) the covariance M^-1 and its lower Cholesky factor (the tuning map's :chol_cov). The dense
 * mass then stays in force on the handle -- momentum p = L^-T z, M^-1 p by the dense product, the
 * U-turn rule through v = M^-1 rho -- for sample_chains / chains_advance / sample_host / stream until
 * exmc_hip_model_clear_dense_mass. exmc_hip_model_set_dense_mass installs a (cov, chol) pair from an
 * earlier run (sample_compiled_tuned with tuning.chol_cov). cov and chol are indexed by the entries
 * of the reference's FLAT vector (the order exmc_hip_model_set_flat_order states; sampler.ex:682-705
 * feeds Welford the flat q), tuning->inv_mass stays in kernel order. Layouts: lanes_per_chain = 1
 * where the kind has a one-lane form, eight_schools with 16 lanes per chain (matrix rows in
 * registers), sv 64, radon 64 and logistic 16 (exmc_hip_model_default_dense_lanes names the kind's
 * layout); EXMC_ERR_UNSUPPORTED otherwise. (The reference raises inside its first dense transition
 * for d >= 2, DESIGN.md "Dense mass"; this is the documented intent of the mode.) */
int exmc_hip_warmup_dense(exmc_hip_model* m, const double* init_q, exmc_hip_opts opts,
                          exmc_hip_tuning* tuning, double* cov, double* chol);
int exmc_hip_model_set_dense_mass(exmc_hip_model* m, const double* cov, const double* chol, int d);
/* Sampler.sample/3 with dense_mass: true: exmc_hip_warmup_dense, then num_samples draws of the same
 * chain under the dense mass (host trace as exmc_hip_sample_host) */
int exmc_hip_sample_dense_host(exmc_hip_model* m, const double* init_q, exmc_hip_opts opts,
                               exmc_hip_trace trace, exmc_hip_tuning* tuning_out, double* cov,
                               double* chol, int32_t* divergences);
int exmc_hip_model_clear_dense_mass(exmc_hip_model* m);
/* opts[:warm_start] of Sampler.sample (lib/exmc/nuts/sampler.ex:167-197): the previous run's
 * inv_mass_diag and step_size instead of the identity mass and the initial step-size search, and
 * a short warmup of min(num_warmup, 50) iterations on top of them. */
int exmc_hip_warmup_from(exmc_hip_model* m, const double* init_q, exmc_hip_opts opts,
                         const exmc_hip_tuning* warm_start, exmc_hip_tuning* tuning);

/* Exmc.NUTS.Sampler.sample_chains vectorized, sampling phase (sampler.ex:1082-1130) for chains
 * [chain_lo, chain_hi) of n_chains: chain i is seeded seed + 7919*i whatever the shard.
 * Device trace layout [draw][dim][chain_local] / [draw][chain_local]; the trace buffers are
 * caller-owned device memory. total_leapfrogs (may be NULL) receives sum of n_steps. */
int exmc_hip_sample_chains(exmc_hip_model* m, const exmc_hip_tuning* tuning, const double* init_q,
                           int n_chains, int chain_lo, int chain_hi, exmc_hip_opts opts,
                           exmc_hip_trace trace_dev, int64_t* total_leapfrogs,
                           int32_t* total_divergences);
/* The two halves of exmc_hip_sample_chains, for callers that keep chains resident in HBM and
 * draw in several launches (run_sampling's Enum.reduce, sampler.ex:946-970, split anywhere):
 * _init seeds / positions chains [chain_lo, chain_hi) and evaluates logp+grad;
 * _advance runs n_draws more transitions of every resident chain, writing trace rows
 * [row_offset, row_offset + n_draws) of buffers that hold `trace_rows` rows in total. */
int exmc_hip_chains_init(exmc_hip_model* m, const exmc_hip_tuning* tuning, const double* init_q,
                         int n_chains, int chain_lo, int chain_hi, exmc_hip_opts opts);
int exmc_hip_chains_advance(exmc_hip_model* m, int n_draws, int row_offset,
                            exmc_hip_trace trace_dev, int64_t* leapfrogs, int32_t* divergences);
/* Same with host trace buffers in the reference's per-chain layout [chain][draw][dim]. */
int exmc_hip_sample_chains_host(exmc_hip_model* m, const exmc_hip_tuning* tuning,
                                const double* init_q, int n_chains, int chain_lo, int chain_hi,
                                exmc_hip_opts opts, exmc_hip_trace trace_host,
                                int64_t* total_leapfrogs, int32_t* total_divergences);

/* Exmc.NUTS.Sampler.sample/3 for one chain (sampler.ex:126-257): warmup + sampling. */
int exmc_hip_sample_host(exmc_hip_model* m, const double* init_q, exmc_hip_opts opts,
                         exmc_hip_trace trace_host, exmc_hip_tuning* tuning_out,
                         int32_t* divergences);
/* the same with opts[:warm_start] (NULL = cold start, i.e. exmc_hip_sample_host) */
int exmc_hip_sample_warm_host(exmc_hip_model* m, const double* init_q, exmc_hip_opts opts,
                              const exmc_hip_tuning* warm_start, exmc_hip_trace trace,
                              exmc_hip_tuning* tuning_out, int32_t* divergences);

/* Exmc.NUTS.Sampler.sample_chains(ir, n, vectorized: false) -- sample_chains_parallel
 * (sampler.ex:992-1000, 1139-1176): chain i of n_chains is Sampler.sample/3 with seed + 7919*i,
 * i.e. it runs its OWN adaptation (step-size search, dual averaging, Welford windows) and then its
 * num_samples draws from the adapted position with the same generator. One launch for chains
 * [chain_lo, chain_hi): every lane group of the grid is one chain from its first warmup transition
 * to its last draw (exmc_nuts.hpp indep_kernel). Trace as exmc_hip_sample_chains (_host: the
 * reference's [chain][draw][dim]); tuning_host (may be NULL) receives, per chain of the shard,
 * 3 + d doubles: final step size, warmup divergences, warmup leapfrogs, inv_mass[d] (kernel order).
 * total_leapfrogs / total_divergences count the sampling phase. Diagonal mass, the kind's default
 * layout (EXMC_ERR_UNSUPPORTED otherwise). */
int exmc_hip_sample_independent(exmc_hip_model* m, const double* init_q, int n_chains, int chain_lo,
                                int chain_hi, exmc_hip_opts opts, exmc_hip_trace trace_dev,
                                double* tuning_host, int64_t* total_leapfrogs,
                                int32_t* total_divergences);
int exmc_hip_sample_independent_host(exmc_hip_model* m, const double* init_q, int n_chains,
                                     int chain_lo, int chain_hi, exmc_hip_opts opts,
                                     exmc_hip_trace trace_host, double* tuning_host,
                                     int64_t* total_leapfrogs, int32_t* total_divergences);

/* Exmc.Diagnostics.rhat (lib/exmc/diagnostics.ex:80-115): split R-hat per dimension across the
 * n_chains chains of a device trace [draw][dim][chain] (n_draws >= 4): rhat_dev [dim]. */
int exmc_hip_rhat(exmc_hip_model* m, const double* draws_dev, int n_draws, int d, int n_chains,
                  double* rhat_dev);

/* Exmc.NUTS.Sampler.sample_stream/4 (sampler.ex:1186-1277), pull style: _begin runs the warmup
 * and keeps the chain resident; each _next call draws the next n_draws transitions of that chain
 * into host buffers [n_draws][..], so the binding can emit {:exmc_sample, i, point, stat} messages
 * while later draws are still to come (the library itself never calls back into the VM). The
 * concatenated draws equal exmc_hip_sample_host's bit for bit. */
int exmc_hip_stream_begin(exmc_hip_model* m, const double* init_q, exmc_hip_opts opts,
                          exmc_hip_tuning* tuning_out);
int exmc_hip_stream_next_host(exmc_hip_model* m, int n_draws, exmc_hip_trace trace_host,
                              int32_t* divergences);
/* sample_stream/4, push style -- the reference sends {:exmc_sample, i, point, stat} after every
 * transition (sampler.ex:1240-1270). _start queues ONE launch for the next n_draws transitions of the
 * resident chain and returns at once; the kernel writes each finished draw straight into page-locked
 * host memory owned by the handle and then publishes, with a system-scope release, the number of
 * finished draws in *progress. `view` receives pointers into that memory ([n_draws][d] draws,
 * [n_draws] stats): rows [0, *progress) are final and may be read while the launch is still running
 * (the binding's thread polls the word and sends the messages; the library never calls back into the
 * VM). _finish waits for the launch, reports the divergences and leaves the memory valid until the
 * next _start or the handle's destruction. The rows equal exmc_hip_stream_next_host's bit for bit.
 * Runs in the kind's default layout under the diagonal mass (EXMC_ERR_UNSUPPORTED otherwise). */
int exmc_hip_stream_start(exmc_hip_model* m, int n_draws, exmc_hip_trace* view,
                          const volatile int32_t** progress);
int exmc_hip_stream_finish(exmc_hip_model* m, int32_t* divergences);

/* Exmc.Diagnostics.ess (lib/exmc/diagnostics.ex:42-52, 123-167) of every (dim, chain) series of
 * a device trace [draw][dim][chain]: ess_dev [dim][chain]. */
int exmc_hip_ess(exmc_hip_model* m, const double* draws_dev, int n_draws, int d, int n_chains,
                 double* ess_dev);
/* Exmc.Diagnostics.ess_bulk (lib/exmc/diagnostics.ex:60-72, 186-219): the same on the
 * rank-normalised series (average ranks, probit of (r - 3/8) / (n + 1/4)). */
int exmc_hip_ess_bulk(exmc_hip_model* m, const double* draws_dev, int n_draws, int d, int n_chains,
                      double* ess_dev);

/* Exmc.NUTS.NativeTree.build_full_tree_bin/17 (lib/exmc/nuts/native_tree.ex:55-75,
 * native/exmc_tree/src/lib.rs:219-302, tree.rs:276-326), batched over n_chains independent
 * trees: pre-computed forward / backward leapfrog chains in, proposal + tree statistics out.
 * Host buffers in the NIF's layout (native-endian f64, row-major [chain][step][dim]):
 *   q0,p0,grad0 [C][d]; logp0 [C]; fwd_* / bwd_* [C][n][d] with logp [C][n]; inv_mass [d];
 *   joint_logp_0 [C]; rng_seed [C] (Xoshiro256** seed_from_u64, lib.rs:262).
 * Outputs q,grad [C][d]; logp, accept_sum [C]; n_steps, divergent, depth int32 [C].
 * Keeps the Rust crate's semantics where they differ from the Elixir path (a divergent leaf
 * keeps the new state; the tree stops when a direction's budget is exhausted). */
int exmc_hip_build_full_tree_host(int device, int n_chains, int d, const double* q0,
                                  const double* p0, const double* grad0, const double* logp0,
                                  const double* fwd_q, const double* fwd_p, const double* fwd_logp,
                                  const double* fwd_grad, int n_fwd, const double* bwd_q,
                                  const double* bwd_p, const double* bwd_logp,
                                  const double* bwd_grad, int n_bwd, const double* inv_mass,
                                  const double* joint_logp_0, int max_depth,
                                  const uint64_t* rng_seed, double* q_out, double* logp_out,
                                  double* grad_out, int32_t* n_steps, int32_t* divergent,
                                  double* accept_sum, int32_t* depth);

/* ---- The NIF's incremental trajectory interface, batched over n_chains trajectories ----
 * (lib/exmc/nuts/native_tree.ex:19-110; native/exmc_tree/src/lib.rs:37-212, 345-434;
 * types.rs:129-172). One handle = the `ResourceArc<TrajectoryResource>` of every chain of a batch,
 * resident on the device between calls. Host buffers in the NIF's binary layout: native-endian
 * f64, row-major [chain][dim] / [chain][state][dim]. */
typedef struct exmc_hip_traj exmc_hip_traj;

/* init_trajectory_bin/4 (lib.rs:37-50, Trajectory::new types.rs:129-152): q, p, grad [C][d];
 * logp [C]. */
int exmc_hip_traj_create(int device, int n_chains, int d, const double* q, const double* p,
                         const double* grad, const double* logp, exmc_hip_traj** out);
void exmc_hip_traj_destroy(exmc_hip_traj* t);

/* get_endpoint_bin/2 (lib.rs:59-71): go_right [C] (0 = left); q, p, grad out [C][d]. */
int exmc_hip_traj_get_endpoint_host(exmc_hip_traj* t, const int32_t* go_right, double* q,
                                    double* p, double* grad);

/* build_and_merge_bin/11 (lib.rs:73-112; build_subtree tree.rs:16-95, merge_into_trajectory
 * tree.rs:194-265): per chain the 2^depth[c] pre-computed leapfrog states of one doubling in
 * all_q, all_p, all_grad [C][n_states][d], all_logp [C][n_states] (n_states >= 2^max depth);
 * inv_mass [d]; joint_logp_0, depth, go_right, rng_seed [C]. A chain with depth[c] < 0 is left
 * untouched (e.g. already terminated). */
int exmc_hip_traj_build_and_merge_host(exmc_hip_traj* t, const double* all_q, const double* all_p,
                                       const double* all_logp, const double* all_grad,
                                       int n_states, const double* inv_mass,
                                       const double* joint_logp_0, const int32_t* depth,
                                       const int32_t* go_right, const uint64_t* rng_seed);

/* is_terminated/1 (lib.rs:52-57): out [C] = divergent || turning. */
int exmc_hip_traj_is_terminated_host(exmc_hip_traj* t, int32_t* out);

/* get_result_bin/1 (lib.rs:305-343, trajectory_to_result tree.rs:329-339): q, grad out [C][d];
 * logp, accept_sum out [C]; n_steps, divergent, depth out int32 [C]. */
int exmc_hip_traj_get_result_host(exmc_hip_traj* t, double* q, double* logp, double* grad,
                                  int32_t* n_steps, int32_t* divergent, double* accept_sum,
                                  int32_t* depth);

/* build_subtree_bin/10 (lib.rs:114-212): the subtree record without a trajectory. Inputs as for
 * build_and_merge; outputs per chain q_left, p_left, grad_left, q_right, p_right, grad_right,
 * q_prop, grad_prop, rho [C][d]; logp_prop, log_sum_weight, accept_sum [C]; n_steps, divergent,
 * turning, depth int32 [C]. */
int exmc_hip_build_subtree_host(int device, int n_chains, int d, const double* all_q,
                                const double* all_p, const double* all_logp,
                                const double* all_grad, int n_states, const double* inv_mass,
                                const double* joint_logp_0, const int32_t* depth,
                                const int32_t* going_right, const uint64_t* rng_seed,
                                double* q_left, double* p_left, double* grad_left,
                                double* q_right, double* p_right, double* grad_right,
                                double* q_prop, double* logp_prop, double* grad_prop,
                                double* log_sum_weight, int32_t* n_steps, int32_t* divergent,
                                double* accept_sum, int32_t* turning, int32_t* subtree_depth,
                                double* rho);

/* B2' -- the fused-chain hook of the speculative path (lib/exmc/nuts/tree.ex:613-653: dispatch_multi_step /
 * do_dispatch call `Nx.Vulkan.leapfrog_chain_normal(q_ref, p_ref, inv_mass_ref, k, signed_eps, mu, sigma)` when
 * the application sets :fused_leapfrog_normal_meta = {mu, sigma} and d <= 256; "Output contract is identical in
 * both branches: {all_q, all_p, all_logp, all_grad}", tree.ex:620-621). K leapfrog steps of a chain whose d
 * coordinates are independent Normal(mu, sigma) terms (lib/exmc/dist/normal.ex:15-24) in one launch, the rows of
 * multi_step_fn out (batched_leapfrog.ex:50-101; raw logp). The hook passes no gradient: the first half-kick
 * uses the gradient at q. signed_eps = dir_sign * epsilon (tree.ex:639). Batched over n_chains independent
 * chains (the reference calls it for one). Host buffers, native-endian f64 where the Vulkan hook moves f32:
 *   q, p [C][d]; inv_mass [d]; q_chain, p_chain, grad_chain out [C][k][d]; logp_chain out [C][k]
 * (any output may be NULL). 1 <= d <= 256 (the hook's own guard, tree.ex:636), k >= 0. */
int exmc_hip_leapfrog_chain_normal_host(int device, int n_chains, int d, const double* q,
                                        const double* p, const double* inv_mass, int k,
                                        double signed_eps, double mu, double sigma, double* q_chain,
                                        double* p_chain, double* grad_chain, double* logp_chain);

/* wall-clock of the last timed kernel region on the handle's stream, HIP events (ms) */
double exmc_hip_last_kernel_ms(const exmc_hip_model* m);

#ifdef __cplusplus
}
#endif
#endif
