/* exmc_hip_nif.c -- NIF module `Elixir.Exmc.NUTS.HipNative`: the chain-batched entry points of
 * libexmc_hip.so that have no counterpart in the Rust crate (the sampler seams B2 / B3 of
 * SURVEY.md 8b). INTEGRATION.md shows where Exmc.NUTS.Sampler calls them.
 *
 *   model_create/2          Compiler.compile_for_sampling/2 for a built model kind (compiler.ex:46-58)
 *   model_create_plugin/2   the same for a model GENERATED from its Builder IR (exmc_amd/codegen.py):
 *                           the plug-in library is dlopen'ed and every later call on the handle goes
 *                           through that library's own copy of the C ABI
 *   model_set_flat_order/2  PointMap.build's sorted-id layout (point_map.ex:30-60)
 *   logp_grad/3             vag_fn, batched (compiler.ex:131-141)
 *   multi_step/8            multi_step_fn, batched (batched_leapfrog.ex:21-48)
 *   leapfrog_chain_normal/7 the fused-chain hook of the speculative path (tree.ex:613-653)
 *   warmup/6                run_warmup of the shared chain (sampler.ex:537-762, 1068-1080)
 *   sample_chains/10        sample_chains_vectorized_compiled's sampling loop (sampler.ex:1082-1130)
 *   sample/7                sample/3 for one chain (sampler.ex:126-257)
 *   sample_warm/9, sample_dense/8   the same with opts[:warm_start] / dense_mass: true (sampler.ex:156, 167-197)
 *   stream_begin/6, stream_next/2   sample_stream/4 (sampler.ex:1186-1277), pulled in chunks
 *   stream_run/3            sample_stream/4's sender: one launch, a message per finished draw
 *
 * Binaries are native-endian f64 (int32 for the integer statistics), row-major
 * [chain][draw][dim]; every call that waits on the GPU is a dirty IO-bound job; errors are
 * {:error, message} from model_create and raised {:exmc_hip_error, code, message} elsewhere.
 * Build as exmc_native_tree_nif.c (one shared object per NIF module, as OTP requires). */
#define _POSIX_C_SOURCE 200809L   /* nanosleep */
#include "exmc_nif_util.h"

#include <dlfcn.h>
#include <time.h>

static ErlNifResourceType* MODEL_RT;
static int g_device = 0;

/* The entry points of include/exmc_hip.h this module calls, as a table: the library this shim is
 * linked against fills one (g_base); a generated model's plug-in library (a build of the same
 * sources around the generated functor, kind EXMC_MODEL_CUSTOM only) fills another through dlsym.
 * A handle keeps the table of the library that created it. */
#define EXMC_NIF_API(X)                                                                                        \
  X(const char*, last_error, (void))                                                                           \
  X(int, model_create, (int, int, const double*, int, int, exmc_hip_model**))                                  \
  X(void, model_destroy, (exmc_hip_model*))                                                                    \
  X(int, model_dim, (const exmc_hip_model*))                                                                   \
  X(int, model_set_flat_order, (exmc_hip_model*, const int32_t*, int))                                         \
  X(int, model_set_dense_mass, (exmc_hip_model*, const double*, const double*, int))                           \
  X(int, model_clear_dense_mass, (exmc_hip_model*))                                                            \
  X(int, logp_grad_host, (exmc_hip_model*, const double*, int, int, double*, double*))                         \
  X(int, multi_step_host, (exmc_hip_model*, const double*, const double*, const double*, double, const double*, \
                           int, int, int, double*, double*, double*, double*))                                 \
  X(int, warmup, (exmc_hip_model*, const double*, exmc_hip_opts, exmc_hip_tuning*))                            \
  X(int, warmup_from, (exmc_hip_model*, const double*, exmc_hip_opts, const exmc_hip_tuning*, exmc_hip_tuning*)) \
  X(int, warmup_dense, (exmc_hip_model*, const double*, exmc_hip_opts, exmc_hip_tuning*, double*, double*))    \
  X(int, sample_chains_host, (exmc_hip_model*, const exmc_hip_tuning*, const double*, int, int, int,           \
                              exmc_hip_opts, exmc_hip_trace, int64_t*, int32_t*))                              \
  X(int, sample_host, (exmc_hip_model*, const double*, exmc_hip_opts, exmc_hip_trace, exmc_hip_tuning*, int32_t*)) \
  X(int, sample_warm_host, (exmc_hip_model*, const double*, exmc_hip_opts, const exmc_hip_tuning*, exmc_hip_trace, \
                            exmc_hip_tuning*, int32_t*))                                                        \
  X(int, sample_dense_host, (exmc_hip_model*, const double*, exmc_hip_opts, exmc_hip_trace, exmc_hip_tuning*,   \
                             double*, double*, int32_t*))                                                       \
  X(int, sample_independent_host, (exmc_hip_model*, const double*, int, int, int, exmc_hip_opts, exmc_hip_trace, \
                                   double*, int64_t*, int32_t*))                                               \
  X(int, stream_begin, (exmc_hip_model*, const double*, exmc_hip_opts, exmc_hip_tuning*))                      \
  X(int, stream_next_host, (exmc_hip_model*, int, exmc_hip_trace, int32_t*))                                   \
  X(int, stream_start, (exmc_hip_model*, int, exmc_hip_trace*, const volatile int32_t**))                      \
  X(int, stream_finish, (exmc_hip_model*, int32_t*))                                                          \
  X(int, leapfrog_chain_normal_host, (int, int, int, const double*, const double*, const double*, int, double,  \
                                      double, double, double*, double*, double*, double*))

typedef struct {
  void* dl;   /* dlopen handle of a plug-in; NULL: the library this shim is linked against */
#define X(ret, name, args) ret (*name) args;
  EXMC_NIF_API(X)
#undef X
} exmc_api;

static const exmc_api g_base = {
    NULL,
#define X(ret, name, args) exmc_hip_##name,
    EXMC_NIF_API(X)
#undef X
};

/* {:exmc_hip_error, code, message} with the message of the library the handle belongs to */
static ERL_NIF_TERM raise_api(ErlNifEnv* env, const exmc_api* A, int rc) {
  if (rc == EXMC_ERR_BADARG) return enif_make_badarg(env);
  return enif_raise_exception(env, tuple3(env, enif_make_atom(env, "exmc_hip_error"), enif_make_int(env, rc),
                                          enif_make_string(env, A->last_error(), ERL_NIF_LATIN1)));
}

/* tid / has_tid: the sender thread of the last stream_run. ERTS threads are joinable and must be
 * joined: by the next stream_run on the handle, by the destructor, or -- when the destructor runs
 * on the sender itself, which dropped the last reference and cannot join itself -- by the reaper:
 * the thread leaves its tid in g_reap and whoever calls into this library next (or unloads it)
 * joins it. tid / has_tid are written and read under g_tid_lock, and the creator holds that lock
 * across enif_thread_create, so the new thread cannot observe a half-published pair. */
typedef struct reap_node { ErlNifTid tid; struct reap_node* next; } reap_node;
/* spare: the list node the sender would park itself in, allocated by stream_run TOGETHER with the
 * thread (has_tid != 0 implies spare != NULL), so that the path on which a thread drops its own last
 * reference never allocates and a tid can never be lost to a failed allocation. */
typedef struct { exmc_hip_model* m; const exmc_api* api; ErlNifTid tid; int has_tid; reap_node* spare; } model_res;

static ErlNifMutex* g_tid_lock;
/* senders that ended on their own last reference, waiting to be joined: a list under g_tid_lock
 * (no bound: every such thread is joined, as ERTS requires) */
static reap_node* g_reap;

/* Join the parked senders. A parked sender may still be inside its handle's destructor (model
 * destruction frees device memory and destroys a stream), so a join here can wait for that to end:
 * every caller is a dirty IO-bound NIF or the unload hook, never a normal scheduler. */
static void reap_senders(void) {
  for (;;) {
    enif_mutex_lock(g_tid_lock);
    reap_node* n = g_reap;
    if (n) g_reap = n->next;
    enif_mutex_unlock(g_tid_lock);
    if (!n) return;
    if (!enif_equal_tids(enif_thread_self(), n->tid)) {
      (void)enif_thread_join(n->tid, NULL);
      enif_free(n);
    } else {                     /* the reaper must not run on a parked thread: put it back and stop */
      enif_mutex_lock(g_tid_lock);
      n->next = g_reap;
      g_reap = n;
      enif_mutex_unlock(g_tid_lock);
      return;
    }
  }
}

static void join_sender(model_res* r) {
  /* (nothing allocates here: the node came with the thread) */
  enif_mutex_lock(g_tid_lock);
  const int has = r->has_tid;
  const ErlNifTid t = r->tid;
  reap_node* n = r->spare;
  r->has_tid = 0;
  r->spare = NULL;
  const int self = has && enif_equal_tids(enif_thread_self(), t);
  if (self) {                    /* joined by the next caller / at unload; n != NULL since has */
    n->tid = t;
    n->next = g_reap;
    g_reap = n;
    n = NULL;
  }
  enif_mutex_unlock(g_tid_lock);
  if (n) enif_free(n);
  if (has && !self) (void)enif_thread_join(t, NULL);
}

static void model_dtor(ErlNifEnv* env, void* obj) {
  (void)env;
  model_res* r = (model_res*)obj;
  join_sender(r);
  if (r->m) r->api->model_destroy(r->m);
  r->m = NULL;
  if (r->api && r->api->dl) {   /* a plug-in's table: the handle was its only user */
    void* dl = r->api->dl;
    enif_free((void*)r->api);
    dlclose(dl);
  }
  r->api = NULL;
}
static model_res* get_res(ErlNifEnv* env, ERL_NIF_TERM t) {
  void* obj;
  if (!enif_get_resource(env, t, MODEL_RT, &obj) || !((model_res*)obj)->m) return NULL;
  return (model_res*)obj;
}
/* every function below starts with: R = the handle, m = its model, A = its library's table */
#define HANDLE(t)                            \
  model_res* R = get_res(env, (t));          \
  exmc_hip_model* m = R ? R->m : NULL;       \
  const exmc_api* A = R ? R->api : &g_base;  \
  (void)A
/* `nil` or an f64 binary of d values */
static int get_init_q(ErlNifEnv* env, ERL_NIF_TERM t, int d, const double** q) {
  char buf[8];
  size_t n;
  if (enif_get_atom(env, t, buf, sizeof buf, ERL_NIF_LATIN1)) {
    *q = NULL;   /* init_values == %{}: 0.1 * normal_s per flat entry, sampler.ex:339-349 */
    return strcmp(buf, "nil") == 0;
  }
  return get_f64_bin(env, t, q, &n) && n == (size_t)d;
}

static ERL_NIF_TERM make_handle(ErlNifEnv* env, const exmc_api* A, int kind, const double* data, size_t n) {
  exmc_hip_model* m = NULL;
  reap_senders();
  int rc = A->model_create(kind, 0, data, (int)n, g_device, &m);
  if (rc != EXMC_OK)
    return tuple2(env, enif_make_atom(env, "error"), enif_make_string(env, A->last_error(), ERL_NIF_LATIN1));
  model_res* r = (model_res*)enif_alloc_resource(MODEL_RT, sizeof(model_res));
  r->m = m;
  r->api = A;
  r->has_tid = 0;
  r->spare = NULL;
  ERL_NIF_TERM ref = enif_make_resource(env, r);
  enif_release_resource(r);
  return tuple2(env, enif_make_atom(env, "ok"), ref);
}

/* model_create(kind, data_bin) -> {:ok, ref} | {:error, message} */
static ERL_NIF_TERM model_create(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int kind;
  const double* data;
  size_t n;
  (void)argc;
  if (!enif_get_int(env, argv[0], &kind) || !get_f64_bin(env, argv[1], &data, &n))
    return enif_make_badarg(env);
  return make_handle(env, &g_base, kind, data, n);
}

/* model_create_plugin(path :: binary, data_bin) -> {:ok, ref} | {:error, message}
 * `path` names the plug-in library a generated model was compiled into (python -m exmc_amd.codegen
 * model.json out_dir -> out_dir/libexmc_hip_gen.so; model.json's "data" is data_bin). Its entry
 * points are bound with dlsym into a table of the handle's own; sampling, streaming, warmup ... are
 * then the functions above, unchanged. */
static ERL_NIF_TERM model_create_plugin(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ErlNifBinary pb;
  const double* data;
  size_t n;
  char path[4096];
  (void)argc;
  if (!enif_inspect_binary(env, argv[0], &pb) || pb.size == 0 || pb.size >= sizeof path ||
      !get_f64_bin(env, argv[1], &data, &n))
    return enif_make_badarg(env);
  memcpy(path, pb.data, pb.size);
  path[pb.size] = 0;
  void* dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!dl) {
    const char* e = dlerror();
    return tuple2(env, enif_make_atom(env, "error"), enif_make_string(env, e ? e : "dlopen failed", ERL_NIF_LATIN1));
  }
  exmc_api* api = (exmc_api*)enif_alloc(sizeof(exmc_api));
  const char* missing = NULL;
  api->dl = dl;
#define X(ret, name, args)                                      \
  *(void**)(&api->name) = dlsym(dl, "exmc_hip_" #name);         \
  if (!api->name && !missing) missing = "exmc_hip_" #name;
  EXMC_NIF_API(X)
#undef X
  if (missing) {
    ERL_NIF_TERM err = tuple2(env, enif_make_atom(env, "error"), enif_make_string(env, missing, ERL_NIF_LATIN1));
    enif_free(api);
    dlclose(dl);
    return err;
  }
  ERL_NIF_TERM r = make_handle(env, api, EXMC_MODEL_CUSTOM, data, n);
  int arity;
  const ERL_NIF_TERM* el;
  char tag[8];
  if (enif_get_tuple(env, r, &arity, &el) && enif_get_atom(env, el[0], tag, sizeof tag, ERL_NIF_LATIN1) &&
      strcmp(tag, "ok") != 0) {   /* no handle took the table over */
    enif_free(api);
    dlclose(dl);
  }
  return r;
}

/* model_set_flat_order(ref, perm :: [non_neg_integer]) -> :ok   (perm[r] = kernel dimension of
 * the r-th id of Enum.sort_by(& &1.id), point_map.ex:37) */
static ERL_NIF_TERM model_set_flat_order(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  unsigned len;
  (void)argc;
  if (!m || !enif_get_list_length(env, argv[1], &len) || (int)len != A->model_dim(m))
    return enif_make_badarg(env);
  int32_t* perm = (int32_t*)enif_alloc((len ? len : 1) * sizeof(int32_t));
  ERL_NIF_TERM head, tail = argv[1];
  int ok = 1;
  for (unsigned i = 0; i < len && ok; i++) {
    int v;
    ok = enif_get_list_cell(env, tail, &head, &tail) && enif_get_int(env, head, &v);
    perm[i] = v;
  }
  int rc = ok ? A->model_set_flat_order(m, perm, (int)len) : EXMC_ERR_BADARG;
  enif_free(perm);
  return rc == EXMC_OK ? enif_make_atom(env, "ok") : raise_api(env, A, rc);
}

/* logp_grad(ref, q_bin [C][d], n_chains) -> {logp_bin [C], grad_bin [C][d]} */
static ERL_NIF_TERM logp_grad(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  const double* q;
  size_t n;
  int c;
  (void)argc;
  if (!m || !get_f64_bin(env, argv[1], &q, &n) || !enif_get_int(env, argv[2], &c) || c < 1 ||
      n != (size_t)c * (size_t)A->model_dim(m))
    return enif_make_badarg(env);
  ERL_NIF_TERM tl, tg;
  double* lp = new_f64_bin(env, (size_t)c, &tl);
  double* g = new_f64_bin(env, n, &tg);
  int rc = A->logp_grad_host(m, q, c, 0, lp, g);
  return rc == EXMC_OK ? tuple2(env, tl, tg) : raise_api(env, A, rc);
}

/* multi_step(ref, q, p, grad :: binary [C][d], eps, inv_mass :: binary [d], n_steps, n_chains)
 *   -> {all_q, all_p, all_logp, all_grad} binaries [C][n][d] / [C][n] */
static ERL_NIF_TERM multi_step(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  const double *q, *p, *g, *im;
  size_t nq, np, ng, nim;
  double eps;
  int n, c;
  (void)argc;
  if (!m || !get_f64_bin(env, argv[1], &q, &nq) || !get_f64_bin(env, argv[2], &p, &np) ||
      !get_f64_bin(env, argv[3], &g, &ng) || !get_f64(env, argv[4], &eps) ||
      !get_f64_bin(env, argv[5], &im, &nim) || !enif_get_int(env, argv[6], &n) ||
      !enif_get_int(env, argv[7], &c) || n < 0 || c < 1)
    return enif_make_badarg(env);
  const size_t d = (size_t)A->model_dim(m);
  if (nq != (size_t)c * d || np != nq || ng != nq || nim != d) return enif_make_badarg(env);
  ERL_NIF_TERM t[4];
  double* aq = new_f64_bin(env, (size_t)c * n * d, &t[0]);
  double* ap = new_f64_bin(env, (size_t)c * n * d, &t[1]);
  double* al = new_f64_bin(env, (size_t)c * n, &t[2]);
  double* ag = new_f64_bin(env, (size_t)c * n * d, &t[3]);
  int rc = A->multi_step_host(m, q, p, g, eps, im, n, c, 0, aq, ap, al, ag);
  return rc == EXMC_OK ? enif_make_tuple_from_array(env, t, 4) : raise_api(env, A, rc);
}

/* leapfrog_chain_normal(q, p, inv_mass :: binary [d], k, signed_eps, mu, sigma)
 *   -> {:ok, {q_chain, p_chain, grad_chain, logp_chain}} binaries [k][d] / [k]
 * The fused-chain hook of the speculative path (tree.ex:613-653): name, argument order and result shape of
 * Nx.Vulkan.leapfrog_chain_normal/7 as do_dispatch calls it (tree.ex:641-647), with f64 binaries where the
 * Vulkan device takes uploaded f32 buffers. No model handle: mu and sigma ARE the model. Always the library
 * this shim is linked against (g_base). */
static ERL_NIF_TERM leapfrog_chain_normal(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  const double *q, *p, *im;
  size_t nq, np, nim;
  int k;
  double eps, mu, sigma;
  (void)argc;
  reap_senders();
  if (!get_f64_bin(env, argv[0], &q, &nq) || !get_f64_bin(env, argv[1], &p, &np) ||
      !get_f64_bin(env, argv[2], &im, &nim) || !enif_get_int(env, argv[3], &k) || !get_f64(env, argv[4], &eps) ||
      !get_f64(env, argv[5], &mu) || !get_f64(env, argv[6], &sigma) || nq < 1 || nq > 256 || np != nq ||
      nim != nq || k < 0)
    return enif_make_badarg(env);   /* d > 256: the reference's function head does not match either (tree.ex:636) */
  const size_t d = nq;
  ERL_NIF_TERM t[4];
  double* aq = new_f64_bin(env, (size_t)k * d, &t[0]);
  double* ap = new_f64_bin(env, (size_t)k * d, &t[1]);
  double* ag = new_f64_bin(env, (size_t)k * d, &t[2]);
  double* al = new_f64_bin(env, (size_t)k, &t[3]);
  int rc = g_base.leapfrog_chain_normal_host(g_device, 1, (int)d, q, p, im, k, eps, mu, sigma, aq, ap, ag, al);
  if (rc != EXMC_OK) return raise_api(env, &g_base, rc);
  return tuple2(env, enif_make_atom(env, "ok"), enif_make_tuple_from_array(env, t, 4));
}

static ERL_NIF_TERM tuning_map(ErlNifEnv* env, const exmc_hip_tuning* tun, int d) {
  ERL_NIF_TERM m = enif_make_new_map(env);
  m = map_put(env, m, "epsilon", enif_make_double(env, tun->epsilon));
  m = map_put(env, m, "inv_mass", make_f64_bin(env, tun->inv_mass, (size_t)d));
  m = map_put(env, m, "warmup_divergences", enif_make_int(env, tun->warmup_divergences));
  return m;
}

/* argv[0..3] = num_warmup, max_tree_depth, target_accept, seed */
static int get_warm_opts(ErlNifEnv* env, const ERL_NIF_TERM argv[], exmc_hip_opts* o) {
  ErlNifUInt64 seed;
  memset(o, 0, sizeof *o);
  if (!enif_get_int(env, argv[0], &o->num_warmup) || !enif_get_int(env, argv[1], &o->max_tree_depth) ||
      !get_f64(env, argv[2], &o->target_accept) || !enif_get_uint64(env, argv[3], &seed))
    return 0;
  o->seed = seed;
  return 1;
}

/* warmup(ref, init_q | nil, num_warmup, max_tree_depth, target_accept, seed)
 *   -> %{epsilon, inv_mass, warmup_divergences}   (the `tuning` map of sampler.ex:62-71) */
static ERL_NIF_TERM warmup(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  const double* iq;
  exmc_hip_opts o;
  exmc_hip_tuning tun;
  (void)argc;
  if (!m || !get_init_q(env, argv[1], A->model_dim(m), &iq) || !get_warm_opts(env, argv + 2, &o))
    return enif_make_badarg(env);
  int rc = A->warmup(m, iq, o, &tun);
  return rc == EXMC_OK ? tuning_map(env, &tun, A->model_dim(m)) : raise_api(env, A, rc);
}

/* warmup_from(ref, init_q | nil, num_warmup, max_tree_depth, target_accept, seed, prev_epsilon,
 *             prev_inv_mass_bin) -> tuning map: opts[:warm_start] of Sampler.sample
 * (sampler.ex:167-197): the previous run's step size and inverse mass, min(num_warmup, 50) iterations */
static ERL_NIF_TERM warmup_from(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  const double *iq, *im;
  size_t nim;
  exmc_hip_opts o;
  exmc_hip_tuning prev, tun;
  (void)argc;
  memset(&prev, 0, sizeof prev);
  if (!m) return enif_make_badarg(env);
  const int d = A->model_dim(m);
  if (!get_init_q(env, argv[1], d, &iq) || !get_warm_opts(env, argv + 2, &o) ||
      !get_f64(env, argv[6], &prev.epsilon) || !get_f64_bin(env, argv[7], &im, &nim) || nim != (size_t)d)
    return enif_make_badarg(env);
  memcpy(prev.inv_mass, im, (size_t)d * 8);
  int rc = A->warmup_from(m, iq, o, &prev, &tun);
  return rc == EXMC_OK ? tuning_map(env, &tun, d) : raise_api(env, A, rc);
}

/* warmup_dense(ref, init_q | nil, num_warmup, max_tree_depth, target_accept, seed, lanes_per_chain)
 *   -> %{epsilon, inv_mass (diagonal), cov, chol_cov (d x d row-major binaries), warmup_divergences}:
 * opts[:dense_mass] (sampler.ex:156, 412-431); the dense mass stays in force on the handle for
 * sample_chains / stream_next until clear_dense_mass */
static ERL_NIF_TERM warmup_dense(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  const double* iq;
  exmc_hip_opts o;
  exmc_hip_tuning tun;
  (void)argc;
  if (!m) return enif_make_badarg(env);
  const int d = A->model_dim(m);
  if (!get_init_q(env, argv[1], d, &iq) || !get_warm_opts(env, argv + 2, &o) ||
      !enif_get_int(env, argv[6], &o.lanes_per_chain) || o.lanes_per_chain < 0)
    return enif_make_badarg(env);
  ERL_NIF_TERM tc, tl;
  double* cov = new_f64_bin(env, (size_t)d * d, &tc);
  double* chol = new_f64_bin(env, (size_t)d * d, &tl);
  int rc = A->warmup_dense(m, iq, o, &tun, cov, chol);
  if (rc != EXMC_OK) return raise_api(env, A, rc);
  ERL_NIF_TERM map = tuning_map(env, &tun, d);
  map = map_put(env, map, "cov", tc);
  map = map_put(env, map, "chol_cov", tl);
  return map;
}

/* set_dense_mass(ref, cov_bin, chol_cov_bin) -> :ok; clear_dense_mass(ref) -> :ok
 * (sample_compiled_tuned with a tuning that carries :chol_cov, sampler.ex:274, 292) */
static ERL_NIF_TERM set_dense_mass(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  const double *cov, *chol;
  size_t nc, nl;
  (void)argc;
  if (!m) return enif_make_badarg(env);
  const size_t d = (size_t)A->model_dim(m);
  if (!get_f64_bin(env, argv[1], &cov, &nc) || !get_f64_bin(env, argv[2], &chol, &nl) || nc != d * d || nl != d * d)
    return enif_make_badarg(env);
  int rc = A->model_set_dense_mass(m, cov, chol, (int)d);
  return rc == EXMC_OK ? enif_make_atom(env, "ok") : raise_api(env, A, rc);
}
static ERL_NIF_TERM clear_dense_mass(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  (void)argc;
  if (!m) return enif_make_badarg(env);
  int rc = A->model_clear_dense_mass(m);
  return rc == EXMC_OK ? enif_make_atom(env, "ok") : raise_api(env, A, rc);
}

/* per-draw outputs as binaries: draws [C][S][d] f64; logp, accept_prob, energy [C][S] f64;
 * tree_depth, n_steps, divergent [C][S] int32 (stats.sample_stats, sampler.ex:960-967) */
typedef struct {
  exmc_hip_trace tr;
  ERL_NIF_TERM t[7];
} trace_bins;
static void new_trace(ErlNifEnv* env, size_t rows, size_t d, trace_bins* b) {
  b->tr.draws = new_f64_bin(env, rows * d, &b->t[0]);
  b->tr.logp = new_f64_bin(env, rows, &b->t[1]);
  b->tr.accept_prob = new_f64_bin(env, rows, &b->t[2]);
  b->tr.energy = new_f64_bin(env, rows, &b->t[3]);
  b->tr.tree_depth = (int32_t*)enif_make_new_binary(env, rows * 4, &b->t[4]);
  b->tr.n_steps = (int32_t*)enif_make_new_binary(env, rows * 4, &b->t[5]);
  b->tr.divergent = (int32_t*)enif_make_new_binary(env, rows * 4, &b->t[6]);
}
static ERL_NIF_TERM trace_map(ErlNifEnv* env, const trace_bins* b) {
  static const char* keys[7] = {"draws", "logp", "accept_prob", "energy", "tree_depth", "n_steps", "divergent"};
  ERL_NIF_TERM m = enif_make_new_map(env);
  for (int i = 0; i < 7; i++) m = map_put(env, m, keys[i], b->t[i]);
  return m;
}

/* sample_chains(ref, epsilon, inv_mass_bin, init_q | nil, n_chains, chain_lo, chain_hi,
 *               num_samples, max_tree_depth, seed) -> {trace_map, leapfrogs, divergences}
 * chains chain_lo..chain_hi-1 of n_chains (chain i is seeded seed + 7919 i, sampler.ex:1083) */
static ERL_NIF_TERM sample_chains(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  exmc_hip_tuning tun;
  const double *im, *iq;
  size_t nim;
  int n_chains, lo, hi;
  ErlNifUInt64 seed;
  exmc_hip_opts o;
  (void)argc;
  memset(&o, 0, sizeof o);
  memset(&tun, 0, sizeof tun);
  if (!m) return enif_make_badarg(env);
  const int d = A->model_dim(m);
  if (!get_f64(env, argv[1], &tun.epsilon) || !get_f64_bin(env, argv[2], &im, &nim) ||
      nim != (size_t)d || !get_init_q(env, argv[3], d, &iq) || !enif_get_int(env, argv[4], &n_chains) ||
      !enif_get_int(env, argv[5], &lo) || !enif_get_int(env, argv[6], &hi) ||
      !enif_get_int(env, argv[7], &o.num_samples) || !enif_get_int(env, argv[8], &o.max_tree_depth) ||
      !enif_get_uint64(env, argv[9], &seed) || lo < 0 || hi <= lo || hi > n_chains || o.num_samples < 0)
    return enif_make_badarg(env);
  memcpy(tun.inv_mass, im, (size_t)d * 8);
  o.seed = seed;
  o.target_accept = 0.8;
  trace_bins b;
  new_trace(env, (size_t)(hi - lo) * (size_t)o.num_samples, (size_t)d, &b);
  int64_t lf = 0;
  int32_t dv = 0;
  int rc = A->sample_chains_host(m, &tun, iq, n_chains, lo, hi, o, b.tr, &lf, &dv);
  if (rc != EXMC_OK) return raise_api(env, A, rc);
  return tuple3(env, trace_map(env, &b), enif_make_uint64(env, (ErlNifUInt64)lf), enif_make_int(env, dv));
}

/* sample_independent(ref, init_q | nil, n_chains, chain_lo, chain_hi, num_warmup, num_samples,
 *                    max_tree_depth, target_accept, seed)
 *   -> {trace_map, tuning_bin, leapfrogs, divergences}
 * sample_chains(ir, n, vectorized: false) -- sample_chains_parallel, sampler.ex:1139-1176: chains
 * [chain_lo, chain_hi) of n_chains, chain i = sample/3 with seed + 7919 i (its own adaptation, then its
 * draws), all in one launch. tuning_bin: per chain 3 + d doubles -- final step size, warmup
 * divergences, warmup leapfrogs, inv_mass (kernel order); leapfrogs / divergences: sampling phase. */
static ERL_NIF_TERM sample_independent(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  const double* iq;
  int n_chains, lo, hi;
  ErlNifUInt64 seed;
  exmc_hip_opts o;
  (void)argc;
  memset(&o, 0, sizeof o);
  if (!m) return enif_make_badarg(env);
  const int d = A->model_dim(m);
  if (!get_init_q(env, argv[1], d, &iq) || !enif_get_int(env, argv[2], &n_chains) ||
      !enif_get_int(env, argv[3], &lo) || !enif_get_int(env, argv[4], &hi) ||
      !enif_get_int(env, argv[5], &o.num_warmup) || !enif_get_int(env, argv[6], &o.num_samples) ||
      !enif_get_int(env, argv[7], &o.max_tree_depth) || !get_f64(env, argv[8], &o.target_accept) ||
      !enif_get_uint64(env, argv[9], &seed) || lo < 0 || hi <= lo || hi > n_chains || o.num_samples < 1 ||
      o.num_warmup < 0)
    return enif_make_badarg(env);
  o.seed = seed;
  const size_t C = (size_t)(hi - lo);
  trace_bins b;
  new_trace(env, C * (size_t)o.num_samples, (size_t)d, &b);
  ERL_NIF_TERM tune_term;
  double* tune = (double*)enif_make_new_binary(env, C * (size_t)(3 + d) * 8, &tune_term);
  int64_t lf = 0;
  int32_t dv = 0;
  int rc = A->sample_independent_host(m, iq, n_chains, lo, hi, o, b.tr, tune, &lf, &dv);
  if (rc != EXMC_OK) return raise_api(env, A, rc);
  return tuple4(env, trace_map(env, &b), tune_term, enif_make_uint64(env, (ErlNifUInt64)lf), enif_make_int(env, dv));
}

/* sample(ref, init_q | nil, num_warmup, num_samples, max_tree_depth, target_accept, seed)
 *   -> {trace_map, tuning_map, divergences} */
static ERL_NIF_TERM sample(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  const double* iq;
  exmc_hip_opts o;
  ErlNifUInt64 seed;
  (void)argc;
  memset(&o, 0, sizeof o);
  if (!m) return enif_make_badarg(env);
  const int d = A->model_dim(m);
  if (!get_init_q(env, argv[1], d, &iq) || !enif_get_int(env, argv[2], &o.num_warmup) ||
      !enif_get_int(env, argv[3], &o.num_samples) || !enif_get_int(env, argv[4], &o.max_tree_depth) ||
      !get_f64(env, argv[5], &o.target_accept) || !enif_get_uint64(env, argv[6], &seed) ||
      o.num_samples < 1)
    return enif_make_badarg(env);
  o.seed = seed;
  trace_bins b;
  new_trace(env, (size_t)o.num_samples, (size_t)d, &b);
  exmc_hip_tuning tun;
  int32_t dv = 0;
  int rc = A->sample_host(m, iq, o, b.tr, &tun, &dv);
  if (rc != EXMC_OK) return raise_api(env, A, rc);
  return tuple3(env, trace_map(env, &b), tuning_map(env, &tun, d), enif_make_int(env, dv));
}

/* argv[0..4] = num_warmup, num_samples, max_tree_depth, target_accept, seed (the options of sample/3) */
static int get_sample_opts(ErlNifEnv* env, const ERL_NIF_TERM argv[], exmc_hip_opts* o) {
  ErlNifUInt64 seed;
  memset(o, 0, sizeof *o);
  if (!enif_get_int(env, argv[0], &o->num_warmup) || !enif_get_int(env, argv[1], &o->num_samples) ||
      !enif_get_int(env, argv[2], &o->max_tree_depth) || !get_f64(env, argv[3], &o->target_accept) ||
      !enif_get_uint64(env, argv[4], &seed) || o->num_samples < 1)
    return 0;
  o->seed = seed;
  return 1;
}

/* sample_warm(ref, init_q | nil, num_warmup, num_samples, max_tree_depth, target_accept, seed,
 *             prev_epsilon, prev_inv_mass_bin) -> {trace_map, tuning_map, divergences}
 * Sampler.sample/3 with opts[:warm_start] (sampler.ex:167-197): the previous run's step size and inverse
 * mass (kernel order, as every inv_mass binary of this module), min(num_warmup, 50) warmup iterations on
 * top of them, then the draws of the SAME chain from where that warmup ended. */
static ERL_NIF_TERM sample_warm(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  const double *iq, *im;
  size_t nim;
  exmc_hip_opts o;
  exmc_hip_tuning prev, tun;
  (void)argc;
  memset(&prev, 0, sizeof prev);
  if (!m) return enif_make_badarg(env);
  const int d = A->model_dim(m);
  if (!get_init_q(env, argv[1], d, &iq) || !get_sample_opts(env, argv + 2, &o) ||
      !get_f64(env, argv[7], &prev.epsilon) || !get_f64_bin(env, argv[8], &im, &nim) || nim != (size_t)d)
    return enif_make_badarg(env);
  memcpy(prev.inv_mass, im, (size_t)d * 8);
  trace_bins b;
  new_trace(env, (size_t)o.num_samples, (size_t)d, &b);
  int32_t dv = 0;
  int rc = A->sample_warm_host(m, iq, o, &prev, b.tr, &tun, &dv);
  if (rc != EXMC_OK) return raise_api(env, A, rc);
  return tuple3(env, trace_map(env, &b), tuning_map(env, &tun, d), enif_make_int(env, dv));
}

/* sample_dense(ref, init_q | nil, num_warmup, num_samples, max_tree_depth, target_accept, seed,
 *              lanes_per_chain) -> {trace_map, tuning_map + cov + chol_cov, divergences}
 * Sampler.sample/3 with dense_mass: true (sampler.ex:156, 412-431): the dense adaptation windows, then the
 * draws of the same chain under the dense mass; cov / chol_cov are d x d row-major binaries in FLAT order
 * (the covariance of the flat vector, sampler.ex:682-705). lanes_per_chain 0 = the kind's dense layout. */
static ERL_NIF_TERM sample_dense(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  const double* iq;
  exmc_hip_opts o;
  exmc_hip_tuning tun;
  (void)argc;
  if (!m) return enif_make_badarg(env);
  const int d = A->model_dim(m);
  if (!get_init_q(env, argv[1], d, &iq) || !get_sample_opts(env, argv + 2, &o) ||
      !enif_get_int(env, argv[7], &o.lanes_per_chain) || o.lanes_per_chain < 0)
    return enif_make_badarg(env);
  trace_bins b;
  new_trace(env, (size_t)o.num_samples, (size_t)d, &b);
  ERL_NIF_TERM tc, tl;
  double* cov = new_f64_bin(env, (size_t)d * d, &tc);
  double* chol = new_f64_bin(env, (size_t)d * d, &tl);
  int32_t dv = 0;
  int rc = A->sample_dense_host(m, iq, o, b.tr, &tun, cov, chol, &dv);
  if (rc != EXMC_OK) return raise_api(env, A, rc);
  ERL_NIF_TERM map = tuning_map(env, &tun, d);
  map = map_put(env, map, "cov", tc);
  map = map_put(env, map, "chol_cov", tl);
  return tuple3(env, trace_map(env, &b), map, enif_make_int(env, dv));
}

/* stream_begin(ref, init_q | nil, num_warmup, max_tree_depth, target_accept, seed) -> tuning_map */
static ERL_NIF_TERM stream_begin(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  const double* iq;
  exmc_hip_opts o;
  exmc_hip_tuning tun;
  (void)argc;
  if (!m || !get_init_q(env, argv[1], A->model_dim(m), &iq) || !get_warm_opts(env, argv + 2, &o))
    return enif_make_badarg(env);
  int rc = A->stream_begin(m, iq, o, &tun);
  return rc == EXMC_OK ? tuning_map(env, &tun, A->model_dim(m)) : raise_api(env, A, rc);
}

/* stream_next(ref, n_draws) -> {trace_map, divergences}: the next n draws of the resident chain;
 * the caller `send`s {:exmc_sample, i, point_map, step_stat} per row (sampler.ex:1270) */
static ERL_NIF_TERM stream_next(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  HANDLE(argv[0]);
  int n;
  (void)argc;
  if (!m || !enif_get_int(env, argv[1], &n) || n < 1) return enif_make_badarg(env);
  trace_bins b;
  new_trace(env, (size_t)n, (size_t)A->model_dim(m), &b);
  int32_t dv = 0;
  int rc = A->stream_next_host(m, n, b.tr, &dv);
  if (rc != EXMC_OK) return raise_api(env, A, rc);
  return tuple2(env, trace_map(env, &b), enif_make_int(env, dv));
}

/* stream_run(ref, n_draws, pid) -> :ok. The sender of sample_stream/4 (sampler.ex:1240-1277): ONE
 * launch draws the next n_draws transitions of the resident chain (exmc_hip_stream_start: the
 * kernel writes every finished draw into page-locked host memory and publishes its count); a thread
 * of this library polls the count and sends
 *   {:exmc_sample, i, q :: f64 binary, {tree_depth, n_steps, divergent, accept_prob, energy}}
 * for i = 1..n as the draws appear, then {:exmc_done, n, divergences}. The handle is busy until
 * that last message: the library refuses every other call on it with {:exmc_hip_error, _, "a stream
 * run is in flight ..."} (the handle is kept alive by the thread; the Elixir side constrains q and
 * builds the point map as it does for stream_next's rows). */
typedef struct {
  model_res* res;
  ErlNifPid pid;
  int n, d;
  exmc_hip_trace view;
  const volatile int32_t* progress;
} stream_job;

static void* stream_sender(void* arg) {
  stream_job* j = (stream_job*)arg;
  ErlNifEnv* env = enif_alloc_env();
  const double* draws = (const double*)j->view.draws;
  const struct timespec nap = {0, 100000};   /* 100 us */
  int sent = 0;
  long idle_naps = 0;
  while (sent < j->n) {
    /* rows [0, ready) are final: the device publishes the count with a system-scope release after
     * the rows' stores; the acquire keeps this thread's reads of the rows behind the count */
    const int ready = __atomic_load_n((const int32_t*)j->progress, __ATOMIC_ACQUIRE);
    if (ready <= sent) {
      if (++idle_naps > 6000000L) break;     /* ten minutes without a draw: give up, report below */
      nanosleep(&nap, NULL);
      continue;
    }
    idle_naps = 0;
    for (; sent < ready; sent++) {
      ERL_NIF_TERM st[5] = {enif_make_int(env, ((const int32_t*)j->view.tree_depth)[sent]),
                            enif_make_int(env, ((const int32_t*)j->view.n_steps)[sent]),
                            make_bool(env, ((const int32_t*)j->view.divergent)[sent] != 0),
                            enif_make_double(env, ((const double*)j->view.accept_prob)[sent]),
                            enif_make_double(env, ((const double*)j->view.energy)[sent])};
      ERL_NIF_TERM msg[4] = {enif_make_atom(env, "exmc_sample"), enif_make_int(env, sent + 1),
                             make_f64_bin(env, draws + (size_t)sent * j->d, (size_t)j->d),
                             enif_make_tuple_from_array(env, st, 5)};
      enif_send(NULL, &j->pid, env, enif_make_tuple_from_array(env, msg, 4));
      enif_clear_env(env);
    }
  }
  int32_t dv = 0;
  const int rc = j->res->api->stream_finish(j->res->m, &dv);
  const int ok = rc == EXMC_OK && sent == j->n;
  enif_send(NULL, &j->pid, env,
            tuple3(env, enif_make_atom(env, ok ? "exmc_done" : "exmc_error"), enif_make_int(env, sent),
                   enif_make_int(env, ok ? dv : rc)));
  enif_free_env(env);
  /* the creator publishes (tid, has_tid) under g_tid_lock right after enif_thread_create: pass
   * through the lock once so that a destructor running below sees the published pair */
  enif_mutex_lock(g_tid_lock);
  enif_mutex_unlock(g_tid_lock);
  enif_release_resource(j->res);
  enif_free(j);
  return NULL;
}

static ERL_NIF_TERM stream_run(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int n;
  ErlNifPid pid;
  (void)argc;
  HANDLE(argv[0]);
  if (!m || !enif_get_int(env, argv[1], &n) || n < 1 || !enif_get_local_pid(env, argv[2], &pid))
    return enif_make_badarg(env);
  stream_job* j = (stream_job*)enif_alloc(sizeof(stream_job));
  reap_node* spare = (reap_node*)enif_alloc(sizeof(reap_node));   /* see model_res.spare */
  if (!j || !spare) {
    if (j) enif_free(j);
    if (spare) enif_free(spare);
    return enif_raise_exception(env, enif_make_atom(env, "enomem"));
  }
  j->res = R;
  j->pid = pid;
  j->n = n;
  j->d = A->model_dim(m);
  int rc = A->stream_start(m, n, &j->view, &j->progress);
  if (rc != EXMC_OK) {          /* includes: the previous run's sender has not finished yet */
    enif_free(j);
    enif_free(spare);
    return raise_api(env, A, rc);
  }
  join_sender(j->res);          /* the previous sender has called stream_finish: it is ending */
  reap_senders();
  enif_keep_resource(j->res);   /* the thread's reference */
  ErlNifTid tid;
  enif_mutex_lock(g_tid_lock);
  const int failed = enif_thread_create((char*)"exmc_hip_stream", &tid, stream_sender, j, NULL) != 0;
  if (!failed) {
    j->res->tid = tid;
    j->res->has_tid = 1;
    j->res->spare = spare;
  }
  enif_mutex_unlock(g_tid_lock);
  if (failed) {
    int32_t dv;
    enif_free(spare);
    (void)A->stream_finish(m, &dv);
    enif_release_resource(j->res);
    enif_free(j);
    return enif_raise_exception(env, enif_make_atom(env, "thread_create_failed"));
  }
  return enif_make_atom(env, "ok");   /* the thread frees its job and ends after :exmc_done; it is joined later */
}

static ErlNifFunc nif_funcs[] = {
    {"model_create", 2, model_create, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"model_create_plugin", 2, model_create_plugin, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"model_set_flat_order", 2, model_set_flat_order, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"logp_grad", 3, logp_grad, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"multi_step", 8, multi_step, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"leapfrog_chain_normal", 7, leapfrog_chain_normal, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"warmup", 6, warmup, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"warmup_from", 8, warmup_from, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"warmup_dense", 7, warmup_dense, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"set_dense_mass", 3, set_dense_mass, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"clear_dense_mass", 1, clear_dense_mass, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"sample_chains", 10, sample_chains, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"sample", 7, sample, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"sample_warm", 9, sample_warm, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"sample_dense", 8, sample_dense, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"sample_independent", 10, sample_independent, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stream_begin", 6, stream_begin, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stream_next", 2, stream_next, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stream_run", 3, stream_run, ERL_NIF_DIRTY_JOB_IO_BOUND},
};

static int on_load(ErlNifEnv* env, void** priv, ERL_NIF_TERM info) {
  (void)priv;
  (void)info;
  const char* dev = getenv("EXMC_HIP_DEVICE");
  g_device = dev ? atoi(dev) : 0;
  g_tid_lock = enif_mutex_create((char*)"exmc_hip_tid");
  g_reap = NULL;
  MODEL_RT = enif_open_resource_type(env, NULL, "exmc_hip_model", model_dtor, ERL_NIF_RT_CREATE, NULL);
  return (MODEL_RT && g_tid_lock) ? 0 : 1;
}

static void on_unload(ErlNifEnv* env, void* priv) {
  (void)env;
  (void)priv;
  reap_senders();               /* no sender thread outlives the library's code */
  if (g_tid_lock) enif_mutex_destroy(g_tid_lock);
  g_tid_lock = NULL;
}

ERL_NIF_INIT(Elixir.Exmc.NUTS.HipNative, nif_funcs, on_load, NULL, NULL, on_unload)
