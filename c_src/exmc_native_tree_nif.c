/* exmc_native_tree_nif.c -- drop-in NIF for the module `Elixir.Exmc.NUTS.NativeTree` over
 * libexmc_hip.so: the same eleven functions at the same arities, argument order and return shapes
 * as the Rustler crate it replaces (native/exmc_tree/src/lib.rs:37-442; Elixir stubs
 * lib/exmc/nuts/native_tree.ex:20-110), so lib/exmc/nuts/tree.ex works untouched
 * (tree.ex:52-54 probes init_trajectory/4; tree.ex:232-250, 777-788, 1331-1342 are the callers).
 *
 *   init_trajectory_bin/4   lib.rs:37-50     -> exmc_hip_traj_create
 *   is_terminated/1         lib.rs:52-57     -> exmc_hip_traj_is_terminated_host
 *   get_endpoint_bin/2      lib.rs:59-71     -> exmc_hip_traj_get_endpoint_host
 *   build_and_merge_bin/11  lib.rs:73-112    -> exmc_hip_traj_build_and_merge_host
 *   build_subtree_bin/10    lib.rs:114-212   -> exmc_hip_build_subtree_host
 *   build_full_tree_bin/17  lib.rs:219-302   -> exmc_hip_build_full_tree_host
 *   get_result_bin/1        lib.rs:305-343   -> exmc_hip_traj_get_result_host
 *   init_trajectory/4, get_endpoint/2, build_and_merge/11, get_result/1   lib.rs:345-434
 *                                            (the list twins: same calls after list unpacking)
 *
 * Conventions kept: binaries are native-endian f64, row-major [step][dim] (lib.rs:19-24,
 * types.rs:36); the callee copies in and returns fresh binaries (lib.rs:26-32); the trajectory is
 * a GC'd resource; a decode failure is a badarg. One NIF call is one chain (n_chains = 1 of the
 * batched C ABI). The calls the crate flags DirtyCpu wait on the GPU here, so they are flagged
 * dirty IO-bound. The library never calls back into the VM.
 *
 * Build (on a machine with OTP):
 *   cc -O2 -fPIC -shared -DEXMC_USE_SYSTEM_ERL_NIF -I$ERL_ROOT/usr/include -Iinclude \
 *      -o priv/native/libexmc_tree.so c_src/exmc_native_tree_nif.c -Lexmc_amd/lib -lexmc_hip
 * (priv/native/libexmc_tree.so is the path Rustler's `use Rustler, otp_app: :exmc, crate:
 * "exmc_tree"` loads, native_tree.ex:12-15). Here: compiled against erl_nif_decl.h by
 * tests/test_nif_shim.py. */
#include "exmc_nif_util.h"

static ErlNifResourceType* TRAJ_RT;
static int g_device = 0;   /* EXMC_HIP_DEVICE at load */

typedef struct {
  exmc_hip_traj* t;
  int d;
} traj_res;

static void traj_dtor(ErlNifEnv* env, void* obj) {
  (void)env;
  traj_res* r = (traj_res*)obj;
  if (r->t) exmc_hip_traj_destroy(r->t);
  r->t = NULL;
}

static ERL_NIF_TERM make_traj(ErlNifEnv* env, const double* q, const double* p, const double* g,
                              size_t d, double logp) {
  exmc_hip_traj* t = NULL;
  int rc = exmc_hip_traj_create(g_device, 1, (int)d, q, p, g, &logp, &t);
  if (rc != EXMC_OK) return raise_hip(env, rc);
  traj_res* r = (traj_res*)enif_alloc_resource(TRAJ_RT, sizeof(traj_res));
  r->t = t;
  r->d = (int)d;
  ERL_NIF_TERM term = enif_make_resource(env, r);
  enif_release_resource(r);
  return term;
}

/* init_trajectory_bin(q_bin, p_bin, grad_bin, logp) -> resource */
static ERL_NIF_TERM init_trajectory_bin(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  const double *q, *p, *g;
  size_t nq, np, ng;
  double logp;
  (void)argc;
  if (!get_f64_bin(env, argv[0], &q, &nq) || !get_f64_bin(env, argv[1], &p, &np) ||
      !get_f64_bin(env, argv[2], &g, &ng) || !get_f64(env, argv[3], &logp) || nq < 1 ||
      np != nq || ng != nq)
    return enif_make_badarg(env);
  return make_traj(env, q, p, g, nq, logp);
}

/* init_trajectory(q, p, grad :: [float], logp) -> resource */
static ERL_NIF_TERM init_trajectory(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  double *q = NULL, *p = NULL, *g = NULL, logp;
  size_t nq = 0, np = 0, ng = 0;
  (void)argc;
  int ok = get_f64_list(env, argv[0], &q, &nq) && get_f64_list(env, argv[1], &p, &np) &&
           get_f64_list(env, argv[2], &g, &ng) && get_f64(env, argv[3], &logp) && nq >= 1 &&
           np == nq && ng == nq;
  ERL_NIF_TERM out = ok ? make_traj(env, q, p, g, nq, logp) : enif_make_badarg(env);
  if (q) enif_free(q);
  if (p) enif_free(p);
  if (g) enif_free(g);
  return out;
}

static traj_res* get_traj(ErlNifEnv* env, ERL_NIF_TERM t) {
  void* obj;
  if (!enif_get_resource(env, t, TRAJ_RT, &obj)) return NULL;
  return (traj_res*)obj;
}

/* is_terminated(ref) -> boolean */
static ERL_NIF_TERM is_terminated(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  traj_res* r = get_traj(env, argv[0]);
  int32_t out = 0;
  (void)argc;
  if (!r) return enif_make_badarg(env);
  int rc = exmc_hip_traj_is_terminated_host(r->t, &out);
  if (rc != EXMC_OK) return raise_hip(env, rc);
  return make_bool(env, out);
}

static int endpoint(ErlNifEnv* env, const ERL_NIF_TERM argv[], traj_res** r, double** buf) {
  int32_t right;
  *r = get_traj(env, argv[0]);
  if (!*r || !get_bool(env, argv[1], &right)) return EXMC_ERR_BADARG;
  const size_t d = (size_t)(*r)->d;
  *buf = (double*)enif_alloc(3 * d * sizeof(double));
  return exmc_hip_traj_get_endpoint_host((*r)->t, &right, *buf, *buf + d, *buf + 2 * d);
}

/* get_endpoint_bin(ref, go_right) -> {q_bin, p_bin, grad_bin} */
static ERL_NIF_TERM get_endpoint_bin(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  traj_res* r;
  double* b = NULL;
  (void)argc;
  int rc = endpoint(env, argv, &r, &b);
  ERL_NIF_TERM out;
  if (rc != EXMC_OK) out = raise_hip(env, rc);
  else {
    const size_t d = (size_t)r->d;
    out = tuple3(env, make_f64_bin(env, b, d), make_f64_bin(env, b + d, d), make_f64_bin(env, b + 2 * d, d));
  }
  if (b) enif_free(b);
  return out;
}

/* get_endpoint(ref, go_right) -> {q, p, grad} lists */
static ERL_NIF_TERM get_endpoint(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  traj_res* r;
  double* b = NULL;
  (void)argc;
  int rc = endpoint(env, argv, &r, &b);
  ERL_NIF_TERM out;
  if (rc != EXMC_OK) out = raise_hip(env, rc);
  else {
    const size_t d = (size_t)r->d;
    out = tuple3(env, make_f64_list(env, b, d), make_f64_list(env, b + d, d), make_f64_list(env, b + 2 * d, d));
  }
  if (b) enif_free(b);
  return out;
}

/* the pre-computed states of one doubling (PrecomputedStates, types.rs:6-43) */
typedef struct {
  const double *all_q, *all_p, *all_logp, *all_g, *inv_mass;
  size_t n_states;
  double jlp0;
  int depth, d;
  int32_t go_right;
  ErlNifUInt64 seed;
} subtree_args;

static int check_states(const subtree_args* a, size_t nq, size_t np, size_t ng, size_t nim) {
  if (a->d < 1 || a->depth > 30) return 0;
  if (nim != (size_t)a->d || nq != a->n_states * (size_t)a->d || np != nq || ng != nq) return 0;
  if (((size_t)1 << a->depth) > a->n_states) return 0;   /* the crate would index out of bounds */
  return 1;
}

/* argv[0..9] = all_q, all_p, all_logp, all_grad, inv_mass, joint_logp_0, depth, d, go_right, seed */
static int subtree_from_bins(ErlNifEnv* env, const ERL_NIF_TERM argv[], subtree_args* a) {
  size_t nq, np, ng, nim;
  if (!get_f64_bin(env, argv[0], &a->all_q, &nq) || !get_f64_bin(env, argv[1], &a->all_p, &np) ||
      !get_f64_bin(env, argv[2], &a->all_logp, &a->n_states) ||
      !get_f64_bin(env, argv[3], &a->all_g, &ng) || !get_f64_bin(env, argv[4], &a->inv_mass, &nim) ||
      !get_f64(env, argv[5], &a->jlp0) || !get_usize(env, argv[6], &a->depth) ||
      !get_usize(env, argv[7], &a->d) || !get_bool(env, argv[8], &a->go_right) ||
      !enif_get_uint64(env, argv[9], &a->seed))
    return 0;
  return check_states(a, nq, np, ng, nim);
}

/* build_and_merge_bin(ref, all_q_bin, all_p_bin, all_logp_bin, all_grad_bin, inv_mass_bin,
 *                     joint_logp_0, depth, d, go_right, rng_seed) -> :ok */
static ERL_NIF_TERM build_and_merge_bin(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  traj_res* r = get_traj(env, argv[0]);
  subtree_args a;
  (void)argc;
  if (!r || !subtree_from_bins(env, argv + 1, &a) || a.d != r->d) return enif_make_badarg(env);
  int32_t depth = a.depth;
  uint64_t seed = a.seed;
  int rc = exmc_hip_traj_build_and_merge_host(r->t, a.all_q, a.all_p, a.all_logp, a.all_g,
                                              (int)a.n_states, a.inv_mass, &a.jlp0, &depth,
                                              &a.go_right, &seed);
  if (rc != EXMC_OK) return raise_hip(env, rc);
  return enif_make_atom(env, "ok");
}

/* build_and_merge(ref, all_q, all_p, all_logp, all_grad, inv_mass :: lists, joint_logp_0, depth,
 *                 d, go_right, rng_seed) -> :ok */
static ERL_NIF_TERM build_and_merge(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  traj_res* r = get_traj(env, argv[0]);
  double *q = NULL, *p = NULL, *lp = NULL, *g = NULL, *im = NULL;
  size_t nq = 0, np = 0, ng = 0, nim = 0;
  subtree_args a;
  (void)argc;
  int ok = r && get_f64_list(env, argv[1], &q, &nq) && get_f64_list(env, argv[2], &p, &np) &&
           get_f64_list(env, argv[3], &lp, &a.n_states) && get_f64_list(env, argv[4], &g, &ng) &&
           get_f64_list(env, argv[5], &im, &nim) && get_f64(env, argv[6], &a.jlp0) &&
           get_usize(env, argv[7], &a.depth) && get_usize(env, argv[8], &a.d) &&
           get_bool(env, argv[9], &a.go_right) && enif_get_uint64(env, argv[10], &a.seed) &&
           a.d == r->d && check_states(&a, nq, np, ng, nim);
  ERL_NIF_TERM out;
  if (!ok) out = enif_make_badarg(env);
  else {
    int32_t depth = a.depth;
    uint64_t seed = a.seed;
    int rc = exmc_hip_traj_build_and_merge_host(r->t, q, p, lp, g, (int)a.n_states, im, &a.jlp0,
                                                &depth, &a.go_right, &seed);
    out = rc == EXMC_OK ? enif_make_atom(env, "ok") : raise_hip(env, rc);
  }
  if (q) enif_free(q);
  if (p) enif_free(p);
  if (lp) enif_free(lp);
  if (g) enif_free(g);
  if (im) enif_free(im);
  return out;
}

/* build_subtree_bin(all_q_bin, all_p_bin, all_logp_bin, all_grad_bin, inv_mass_bin, joint_logp_0,
 *                   depth, d, going_right, rng_seed) -> the TreeNode map of lib.rs:146-210 */
static ERL_NIF_TERM build_subtree_bin(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  subtree_args a;
  (void)argc;
  if (!subtree_from_bins(env, argv, &a)) return enif_make_badarg(env);
  const size_t d = (size_t)a.d;
  ERL_NIF_TERM tb[9];
  double* v[9];
  for (int i = 0; i < 9; i++) v[i] = new_f64_bin(env, d, &tb[i]);
  double logp_prop, lsw, acc;
  int32_t n_steps, div, turning, sdepth, depth = a.depth;
  uint64_t seed = a.seed;
  int rc = exmc_hip_build_subtree_host(g_device, 1, a.d, a.all_q, a.all_p, a.all_logp, a.all_g,
                                       (int)a.n_states, a.inv_mass, &a.jlp0, &depth, &a.go_right,
                                       &seed, v[0], v[1], v[2], v[3], v[4], v[5], v[6], &logp_prop,
                                       v[7], &lsw, &n_steps, &div, &acc, &turning, &sdepth, v[8]);
  if (rc != EXMC_OK) return raise_hip(env, rc);
  ERL_NIF_TERM m = enif_make_new_map(env);
  m = map_put(env, m, "q_left_bin", tb[0]);
  m = map_put(env, m, "p_left_bin", tb[1]);
  m = map_put(env, m, "grad_left_bin", tb[2]);
  m = map_put(env, m, "q_right_bin", tb[3]);
  m = map_put(env, m, "p_right_bin", tb[4]);
  m = map_put(env, m, "grad_right_bin", tb[5]);
  m = map_put(env, m, "q_prop_bin", tb[6]);
  m = map_put(env, m, "logp_prop", enif_make_double(env, logp_prop));
  m = map_put(env, m, "grad_prop_bin", tb[7]);
  m = map_put(env, m, "log_sum_weight", enif_make_double(env, lsw));
  m = map_put(env, m, "n_steps", enif_make_int(env, n_steps));
  m = map_put(env, m, "divergent", make_bool(env, div));
  m = map_put(env, m, "accept_sum", enif_make_double(env, acc));
  m = map_put(env, m, "turning", make_bool(env, turning));
  m = map_put(env, m, "depth", enif_make_int(env, sdepth));
  m = map_put(env, m, "rho_bin", tb[8]);
  return m;
}

/* build_full_tree_bin(q0_bin, p0_bin, grad0_bin, logp0, fwd_q_bin, fwd_p_bin, fwd_logp_bin,
 *                     fwd_grad_bin, bwd_q_bin, bwd_p_bin, bwd_logp_bin, bwd_grad_bin, inv_mass_bin,
 *                     joint_logp_0, max_depth, d, rng_seed) -> result map (lib.rs:270-300) */
static ERL_NIF_TERM build_full_tree_bin(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  const double *q0, *p0, *g0, *fq, *fp, *fl, *fg, *bq, *bp, *bl, *bg, *im;
  size_t nq0, np0, ng0, nfq, nfp, nfl, nfg, nbq, nbp, nbl, nbg, nim;
  double logp0, jlp0;
  int max_depth, d;
  ErlNifUInt64 seed;
  (void)argc;
  if (!get_f64_bin(env, argv[0], &q0, &nq0) || !get_f64_bin(env, argv[1], &p0, &np0) ||
      !get_f64_bin(env, argv[2], &g0, &ng0) || !get_f64(env, argv[3], &logp0) ||
      !get_f64_bin(env, argv[4], &fq, &nfq) || !get_f64_bin(env, argv[5], &fp, &nfp) ||
      !get_f64_bin(env, argv[6], &fl, &nfl) || !get_f64_bin(env, argv[7], &fg, &nfg) ||
      !get_f64_bin(env, argv[8], &bq, &nbq) || !get_f64_bin(env, argv[9], &bp, &nbp) ||
      !get_f64_bin(env, argv[10], &bl, &nbl) || !get_f64_bin(env, argv[11], &bg, &nbg) ||
      !get_f64_bin(env, argv[12], &im, &nim) || !get_f64(env, argv[13], &jlp0) ||
      !get_usize(env, argv[14], &max_depth) || !get_usize(env, argv[15], &d) ||
      !enif_get_uint64(env, argv[16], &seed))
    return enif_make_badarg(env);
  const size_t dd = (size_t)d;
  if (d < 1 || nq0 != dd || np0 != dd || ng0 != dd || nim != dd || nfq != nfl * dd || nfp != nfq ||
      nfg != nfq || nbq != nbl * dd || nbp != nbq || nbg != nbq)
    return enif_make_badarg(env);
  ERL_NIF_TERM tq, tg;
  double* q = new_f64_bin(env, dd, &tq);
  double* g = new_f64_bin(env, dd, &tg);
  double logp, acc;
  int32_t n_steps, div, depth;
  uint64_t s = seed;
  int rc = exmc_hip_build_full_tree_host(g_device, 1, d, q0, p0, g0, &logp0, fq, fp, fl, fg, (int)nfl,
                                         bq, bp, bl, bg, (int)nbl, im, &jlp0, max_depth, &s, q, &logp,
                                         g, &n_steps, &div, &acc, &depth);
  if (rc != EXMC_OK) return raise_hip(env, rc);
  ERL_NIF_TERM m = enif_make_new_map(env);
  m = map_put(env, m, "q_bin", tq);
  m = map_put(env, m, "logp", enif_make_double(env, logp));
  m = map_put(env, m, "grad_bin", tg);
  m = map_put(env, m, "n_steps", enif_make_int(env, n_steps));
  m = map_put(env, m, "divergent", make_bool(env, div));
  m = map_put(env, m, "accept_sum", enif_make_double(env, acc));
  m = map_put(env, m, "depth", enif_make_int(env, depth));
  return m;
}

static ERL_NIF_TERM result_map(ErlNifEnv* env, const ERL_NIF_TERM argv[], int as_lists) {
  traj_res* r = get_traj(env, argv[0]);
  if (!r) return enif_make_badarg(env);
  const size_t d = (size_t)r->d;
  double* b = (double*)enif_alloc(2 * d * sizeof(double));
  double logp, acc;
  int32_t n_steps, div, depth;
  int rc = exmc_hip_traj_get_result_host(r->t, b, &logp, b + d, &n_steps, &div, &acc, &depth);
  ERL_NIF_TERM m;
  if (rc != EXMC_OK) m = raise_hip(env, rc);
  else {
    m = enif_make_new_map(env);
    m = map_put(env, m, as_lists ? "q" : "q_bin", as_lists ? make_f64_list(env, b, d) : make_f64_bin(env, b, d));
    m = map_put(env, m, "logp", enif_make_double(env, logp));
    m = map_put(env, m, as_lists ? "grad" : "grad_bin",
                as_lists ? make_f64_list(env, b + d, d) : make_f64_bin(env, b + d, d));
    m = map_put(env, m, "n_steps", enif_make_int(env, n_steps));
    m = map_put(env, m, "divergent", make_bool(env, div));
    m = map_put(env, m, "accept_sum", enif_make_double(env, acc));
    m = map_put(env, m, "depth", enif_make_int(env, depth));
  }
  enif_free(b);
  return m;
}
/* get_result_bin(ref) -> %{q_bin, logp, grad_bin, n_steps, divergent, accept_sum, depth} */
static ERL_NIF_TERM get_result_bin(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  (void)argc;
  return result_map(env, argv, 0);
}
/* get_result(ref) -> %{q, logp, grad, n_steps, divergent, accept_sum, depth} */
static ERL_NIF_TERM get_result(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  (void)argc;
  return result_map(env, argv, 1);
}

/* every function reads the trajectory resource from the device (blocking copies): all are dirty
 * IO-bound jobs, including the accessors that were plain memory reads in the Rust crate */
static ErlNifFunc nif_funcs[] = {
    {"init_trajectory_bin", 4, init_trajectory_bin, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"is_terminated", 1, is_terminated, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"get_endpoint_bin", 2, get_endpoint_bin, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"build_and_merge_bin", 11, build_and_merge_bin, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"build_subtree_bin", 10, build_subtree_bin, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"build_full_tree_bin", 17, build_full_tree_bin, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"get_result_bin", 1, get_result_bin, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"init_trajectory", 4, init_trajectory, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"get_endpoint", 2, get_endpoint, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"build_and_merge", 11, build_and_merge, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"get_result", 1, get_result, ERL_NIF_DIRTY_JOB_IO_BOUND},
};

static int on_load(ErlNifEnv* env, void** priv, ERL_NIF_TERM info) {
  (void)priv;
  (void)info;
  const char* dev = getenv("EXMC_HIP_DEVICE");
  g_device = dev ? atoi(dev) : 0;
  TRAJ_RT = enif_open_resource_type(env, NULL, "exmc_hip_traj", traj_dtor, ERL_NIF_RT_CREATE, NULL);
  return TRAJ_RT ? 0 : 1;
}

ERL_NIF_INIT(Elixir.Exmc.NUTS.NativeTree, nif_funcs, on_load, NULL, NULL, NULL)
