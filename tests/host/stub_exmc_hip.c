/* stub_exmc_hip.c -- TEST INFRASTRUCTURE ONLY: a test double of libexmc_hip.so for the ThreadSanitizer
 * run of the NIF shim's thread logic (tools/sanitize_cpu.sh). It exports every entry point
 * c_src/exmc_hip_nif.c binds; the streaming pair behaves like the library's -- exmc_hip_stream_start
 * claims the handle with a compare-and-swap, a host thread stands in for the kernel (it writes the rows of
 * the "page-locked" view and then publishes the count of finished draws with a release store),
 * exmc_hip_stream_finish waits for it -- and model destruction takes a little while, like freeing device
 * memory does. No arithmetic, no device. Never linked into the product. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/exmc_hip.h"

struct exmc_hip_model {
  int d;
  int in_flight;            /* claimed by stream_start (CAS), released by stream_finish */
  int n;
  double* draws;
  int32_t *depth, *steps, *div;
  double *acc, *energy;
  int32_t progress;
  pthread_t producer;
  int has_producer;
};

static const char* g_err = "";
const char* exmc_hip_last_error(void) { return g_err; }
int exmc_hip_device_count(void) { return 1; }

int exmc_hip_model_create(int kind, int d, const double* data, int n_data, int device, exmc_hip_model** out) {
  (void)data; (void)device; (void)d;
  if (kind < 0) { g_err = "bad kind"; return EXMC_ERR_BADARG; }
  exmc_hip_model* m = (exmc_hip_model*)calloc(1, sizeof *m);
  m->d = n_data > 0 ? (n_data > 16 ? 16 : n_data) : 2;
  *out = m;
  return EXMC_OK;
}
static void nap_us(long us) {
  struct timespec t = {0, us * 1000L};
  nanosleep(&t, NULL);
}
static void free_view(exmc_hip_model* m) {
  free(m->draws); free(m->depth); free(m->steps); free(m->div); free(m->acc); free(m->energy);
  m->draws = NULL; m->depth = m->steps = m->div = NULL; m->acc = m->energy = NULL;
}
void exmc_hip_model_destroy(exmc_hip_model* m) {
  if (!m) return;
  if (m->has_producer) pthread_join(m->producer, NULL);
  nap_us(2000);             /* hipFree + stream destruction take time: the reaper may have to wait for this */
  free_view(m);
  free(m);
}
int exmc_hip_model_dim(const exmc_hip_model* m) { return m ? m->d : -1; }
int exmc_hip_model_set_flat_order(exmc_hip_model* m, const int32_t* p, int d) { (void)m; (void)p; (void)d; return EXMC_OK; }
int exmc_hip_model_set_dense_mass(exmc_hip_model* m, const double* a, const double* b, int d) { (void)m; (void)a; (void)b; (void)d; return EXMC_OK; }
int exmc_hip_model_clear_dense_mass(exmc_hip_model* m) { (void)m; return EXMC_OK; }

#define NO_DEVICE(...) { g_err = "stub library: no device"; return EXMC_ERR_NO_DEVICE; }
int exmc_hip_logp_grad_host(exmc_hip_model* m, const double* q, int c, int l, double* lp, double* g) NO_DEVICE()
int exmc_hip_multi_step_host(exmc_hip_model* m, const double* q, const double* p, const double* g, double e,
                             const double* im, int n, int c, int l, double* a, double* b, double* cc, double* dd) NO_DEVICE()
int exmc_hip_leapfrog_chain_normal_host(int dev, int c, int d, const double* q, const double* p, const double* im, int k,
                                        double e, double mu, double sg, double* a, double* b, double* cc, double* dd) NO_DEVICE()
int exmc_hip_warmup(exmc_hip_model* m, const double* q, exmc_hip_opts o, exmc_hip_tuning* t) NO_DEVICE()
int exmc_hip_warmup_from(exmc_hip_model* m, const double* q, exmc_hip_opts o, const exmc_hip_tuning* s, exmc_hip_tuning* t) NO_DEVICE()
int exmc_hip_warmup_dense(exmc_hip_model* m, const double* q, exmc_hip_opts o, exmc_hip_tuning* t, double* a, double* b) NO_DEVICE()
int exmc_hip_sample_chains_host(exmc_hip_model* m, const exmc_hip_tuning* t, const double* q, int n, int lo, int hi,
                                exmc_hip_opts o, exmc_hip_trace tr, int64_t* lf, int32_t* dv) NO_DEVICE()
int exmc_hip_sample_host(exmc_hip_model* m, const double* q, exmc_hip_opts o, exmc_hip_trace tr, exmc_hip_tuning* t, int32_t* dv) NO_DEVICE()
int exmc_hip_sample_warm_host(exmc_hip_model* m, const double* q, exmc_hip_opts o, const exmc_hip_tuning* w, exmc_hip_trace tr,
                              exmc_hip_tuning* t, int32_t* dv) NO_DEVICE()
int exmc_hip_sample_dense_host(exmc_hip_model* m, const double* q, exmc_hip_opts o, exmc_hip_trace tr, exmc_hip_tuning* t,
                               double* cov, double* chol, int32_t* dv) NO_DEVICE()
int exmc_hip_sample_independent_host(exmc_hip_model* m, const double* q, int n, int lo, int hi, exmc_hip_opts o,
                                     exmc_hip_trace tr, double* tu, int64_t* lf, int32_t* dv) NO_DEVICE()
int exmc_hip_stream_next_host(exmc_hip_model* m, int n, exmc_hip_trace tr, int32_t* dv) NO_DEVICE()

int exmc_hip_stream_begin(exmc_hip_model* m, const double* q, exmc_hip_opts o, exmc_hip_tuning* t) {
  (void)q; (void)o;
  if (__atomic_load_n(&m->in_flight, __ATOMIC_ACQUIRE)) { g_err = "a stream run is in flight on this handle"; return EXMC_ERR_BADARG; }
  memset(t, 0, sizeof *t);
  t->epsilon = 0.5;
  for (int i = 0; i < m->d; i++) t->inv_mass[i] = 1.0;
  return EXMC_OK;
}

static void* producer(void* arg) {           /* the kernel's stand-in */
  exmc_hip_model* m = (exmc_hip_model*)arg;
  for (int i = 0; i < m->n; i++) {
    for (int k = 0; k < m->d; k++) m->draws[(size_t)i * m->d + k] = (double)(1000 * i + k);
    m->depth[i] = 3; m->steps[i] = 7; m->div[i] = (i % 17 == 16);
    m->acc[i] = 0.75; m->energy[i] = -(double)i;
    __atomic_store_n(&m->progress, i + 1, __ATOMIC_RELEASE);   /* after the row's stores */
    if (i % 8 == 7) nap_us(30);
  }
  return NULL;
}

int exmc_hip_stream_start(exmc_hip_model* m, int n, exmc_hip_trace* view, const volatile int32_t** progress) {
  int expected = 0;
  if (!__atomic_compare_exchange_n(&m->in_flight, &expected, 1, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
    g_err = "a stream run is in flight on this handle: call exmc_hip_stream_finish first";
    return EXMC_ERR_BADARG;
  }
  if (m->has_producer) { pthread_join(m->producer, NULL); m->has_producer = 0; }
  free_view(m);
  m->n = n;
  m->draws = (double*)calloc((size_t)n * m->d, 8);
  m->depth = (int32_t*)calloc(n, 4); m->steps = (int32_t*)calloc(n, 4); m->div = (int32_t*)calloc(n, 4);
  m->acc = (double*)calloc(n, 8); m->energy = (double*)calloc(n, 8);
  __atomic_store_n(&m->progress, 0, __ATOMIC_RELAXED);
  view->draws = m->draws; view->logp = NULL; view->tree_depth = m->depth; view->n_steps = m->steps;
  view->divergent = m->div; view->accept_prob = m->acc; view->energy = m->energy;
  *progress = &m->progress;
  if (pthread_create(&m->producer, NULL, producer, m) != 0) {
    __atomic_store_n(&m->in_flight, 0, __ATOMIC_RELEASE);
    g_err = "thread";
    return EXMC_ERR_HIP;
  }
  m->has_producer = 1;
  return EXMC_OK;
}

int exmc_hip_stream_finish(exmc_hip_model* m, int32_t* divergences) {
  if (!__atomic_load_n(&m->in_flight, __ATOMIC_ACQUIRE)) { g_err = "no stream run in flight"; return EXMC_ERR_BADARG; }
  if (m->has_producer) { pthread_join(m->producer, NULL); m->has_producer = 0; }
  int dv = 0;
  for (int i = 0; i < m->n; i++) dv += m->div[i];
  if (divergences) *divergences = dv;
  __atomic_store_n(&m->in_flight, 0, __ATOMIC_RELEASE);
  return EXMC_OK;
}
