/* fake_erl_nif.c -- TEST INFRASTRUCTURE ONLY: a few hundred lines that implement the enif_*
 * functions declared in c_src/erl_nif_decl.h over a flat term table, so that the NIF shims in
 * c_src/ can be loaded and their functions CALLED from the tests without a BEAM (there is no
 * Erlang/OTP in this image). Terms are indices into a table; nothing is garbage collected
 * (fk_reset clears the table between calls). It is not a VM and is never shipped. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../c_src/erl_nif_decl.h"

enum { T_ATOM = 1, T_DOUBLE, T_INT, T_BIN, T_LIST, T_TUPLE, T_MAP, T_RES, T_STR };

typedef struct {
  int type;
  double d;
  long long i;
  char* s;               /* atom name / string */
  unsigned char* data;   /* binary */
  size_t size;
  ERL_NIF_TERM* items;   /* list / tuple items; map: keys then values */
  unsigned n;
  void* obj;             /* resource */
} term;

/* the table is chunked so that a term's address never moves: a sender thread (enif_send) and the
 * harness thread may create terms at the same time (tools/sanitize_cpu.sh runs both under TSan) */
#define FK_CHUNK 1024
#define FK_MAX_CHUNKS 8192
static term* g_chunks[FK_MAX_CHUNKS];
static size_t g_n;
static ERL_NIF_TERM g_exception;   /* 0 = none */
static int g_badarg;

struct enif_environment_t { int dummy; };
static struct enif_environment_t g_env;

struct enif_resource_type_t { ErlNifResourceDtor* dtor; char name[64]; };
typedef struct { struct enif_resource_type_t* type; int refc; } res_hdr;

/* one lock around the table's growth and the mailbox */
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static ERL_NIF_TERM* g_mail;
static size_t g_mail_n, g_mail_cap;

static term* slot(size_t i) { return &g_chunks[i / FK_CHUNK][i % FK_CHUNK]; }
static ERL_NIF_TERM new_term_locked(int type);
static ERL_NIF_TERM new_term(int type) {
  pthread_mutex_lock(&g_lock);
  ERL_NIF_TERM t = new_term_locked(type);
  pthread_mutex_unlock(&g_lock);
  return t;
}
static ERL_NIF_TERM new_term_locked(int type) {
  if (g_n == 0) g_n = 1;   /* term 0 = invalid */
  if (g_n / FK_CHUNK >= FK_MAX_CHUNKS) { fprintf(stderr, "fake_erl_nif: term table full\n"); abort(); }
  if (!g_chunks[g_n / FK_CHUNK]) g_chunks[g_n / FK_CHUNK] = (term*)calloc(FK_CHUNK, sizeof(term));
  memset(slot(g_n), 0, sizeof(term));
  slot(g_n)->type = type;
  return (ERL_NIF_TERM)g_n++;
}
static size_t table_len(void) {
  pthread_mutex_lock(&g_lock);
  const size_t n = g_n;
  pthread_mutex_unlock(&g_lock);
  return n;
}
static term* T(ERL_NIF_TERM t) { return (t > 0 && t < table_len()) ? slot(t) : NULL; }

/* ---- harness API (called from Python through ctypes) ---- */
void fk_reset(void) {
  for (size_t i = 1; i < g_n; i++) {
    term* t = slot(i);
    free(t->s);
    if (t->type == T_BIN) free(t->data);
    free(t->items);
    if (t->type == T_RES && t->obj) enif_release_resource(t->obj);
  }
  g_n = 0;
  g_exception = 0;
  g_badarg = 0;
  g_mail_n = 0;
}
/* the harness's own reference to a resource term goes away (a variable going out of scope on the BEAM) */
void fk_drop_resource_term(ERL_NIF_TERM t) {
  term* x = T(t);
  if (x && x->type == T_RES && x->obj) {
    void* obj = x->obj;
    x->obj = NULL;
    enif_release_resource(obj);
  }
}
/* the mailbox of the one fake process: messages in arrival order */
size_t fk_mailbox_len(void) {
  pthread_mutex_lock(&g_lock);
  size_t n = g_mail_n;
  pthread_mutex_unlock(&g_lock);
  return n;
}
ERL_NIF_TERM fk_mailbox_get(size_t i) { return i < g_mail_n ? g_mail[i] : 0; }
int fk_mailbox_wait(size_t n, int timeout_ms) {   /* 1 when n messages have arrived */
  for (int waited = 0; waited < timeout_ms; waited++) {
    if (fk_mailbox_len() >= n) return 1;
    usleep(1000);
  }
  return fk_mailbox_len() >= n;
}
ERL_NIF_TERM fk_atom(const char* name) { return enif_make_atom(&g_env, name); }
ERL_NIF_TERM fk_double(double x) { return enif_make_double(&g_env, x); }
ERL_NIF_TERM fk_int(long long v) { ERL_NIF_TERM t = new_term(T_INT); T(t)->i = v; return t; }
ERL_NIF_TERM fk_binary(const void* p, size_t n) {
  ERL_NIF_TERM t;
  unsigned char* dst = enif_make_new_binary(&g_env, n, &t);
  if (n) memcpy(dst, p, n);
  return t;
}
ERL_NIF_TERM fk_list(const ERL_NIF_TERM* items, unsigned n) { return enif_make_list_from_array(&g_env, items, n); }
int fk_type(ERL_NIF_TERM t) { return T(t) ? T(t)->type : 0; }
double fk_get_double(ERL_NIF_TERM t) { return T(t)->d; }
long long fk_get_int(ERL_NIF_TERM t) { return T(t)->i; }
const char* fk_get_str(ERL_NIF_TERM t) { return T(t)->s; }
const void* fk_bin_data(ERL_NIF_TERM t) { return T(t)->data; }
size_t fk_bin_size(ERL_NIF_TERM t) { return T(t)->size; }
unsigned fk_len(ERL_NIF_TERM t) { return T(t)->n; }
ERL_NIF_TERM fk_item(ERL_NIF_TERM t, unsigned i) { return T(t)->items[i]; }
ERL_NIF_TERM fk_map_get(ERL_NIF_TERM m, const char* key) {
  term* t = T(m);
  for (unsigned i = 0; i < t->n; i++) {
    term* k = T(t->items[i]);
    if (k && k->type == T_ATOM && strcmp(k->s, key) == 0) return t->items[t->n + i];
  }
  return 0;
}
ERL_NIF_TERM fk_exception(void) { return g_exception; }
int fk_badarg(void) { return g_badarg; }
/* load: run the module's load callback; call: look a function up in the entry table */
int fk_load(ErlNifEntry* e) { return e->load ? e->load(&g_env, NULL, 0) : 0; }
ERL_NIF_TERM fk_call(ErlNifEntry* e, const char* name, unsigned arity, const ERL_NIF_TERM* argv) {
  g_exception = 0;
  g_badarg = 0;
  for (int i = 0; i < e->num_of_funcs; i++)
    if (strcmp(e->funcs[i].name, name) == 0 && e->funcs[i].arity == arity)
      return e->funcs[i].fptr(&g_env, (int)arity, argv);
  g_badarg = 2;   /* undefined function */
  return 0;
}

/* ---- the declared API ---- */
int enif_get_double(ErlNifEnv* e, ERL_NIF_TERM t, double* dp) {
  (void)e; if (!T(t) || T(t)->type != T_DOUBLE) return 0; *dp = T(t)->d; return 1;
}
int enif_get_int(ErlNifEnv* e, ERL_NIF_TERM t, int* ip) {
  (void)e; if (!T(t) || T(t)->type != T_INT || T(t)->i < -2147483648LL || T(t)->i > 2147483647LL) return 0;
  *ip = (int)T(t)->i; return 1;
}
int enif_get_int64(ErlNifEnv* e, ERL_NIF_TERM t, ErlNifSInt64* ip) {
  (void)e; if (!T(t) || T(t)->type != T_INT) return 0; *ip = T(t)->i; return 1;
}
int enif_get_uint64(ErlNifEnv* e, ERL_NIF_TERM t, ErlNifUInt64* ip) {
  (void)e; if (!T(t) || T(t)->type != T_INT || T(t)->i < 0) return 0; *ip = (ErlNifUInt64)T(t)->i; return 1;
}
int enif_get_atom(ErlNifEnv* e, ERL_NIF_TERM t, char* buf, unsigned len, ErlNifCharEncoding c) {
  (void)e; (void)c;
  if (!T(t) || T(t)->type != T_ATOM || strlen(T(t)->s) + 1 > len) return 0;
  strcpy(buf, T(t)->s);
  return (int)strlen(buf) + 1;
}
int enif_inspect_binary(ErlNifEnv* e, ERL_NIF_TERM t, ErlNifBinary* bin) {
  (void)e; if (!T(t) || T(t)->type != T_BIN) return 0;
  bin->size = T(t)->size; bin->data = T(t)->data; return 1;
}
int enif_get_list_length(ErlNifEnv* e, ERL_NIF_TERM t, unsigned* len) {
  (void)e; if (!T(t) || T(t)->type != T_LIST) return 0; *len = T(t)->n; return 1;
}
int enif_get_list_cell(ErlNifEnv* e, ERL_NIF_TERM list, ERL_NIF_TERM* head, ERL_NIF_TERM* tail) {
  term* l = T(list);
  if (!l || l->type != T_LIST || l->n == 0) return 0;
  *head = l->items[0];
  *tail = enif_make_list_from_array(e, T(list)->items + 1, T(list)->n - 1);
  return 1;
}
int enif_get_resource(ErlNifEnv* e, ERL_NIF_TERM t, ErlNifResourceType* type, void** objp) {
  (void)e; if (!T(t) || T(t)->type != T_RES) return 0;
  res_hdr* h = (res_hdr*)T(t)->obj - 1;
  if (h->type != type) return 0;
  *objp = T(t)->obj; return 1;
}
ERL_NIF_TERM enif_make_badarg(ErlNifEnv* e) { (void)e; g_badarg = 1; return 0; }
ERL_NIF_TERM enif_raise_exception(ErlNifEnv* e, ERL_NIF_TERM reason) { (void)e; g_exception = reason; return 0; }
ERL_NIF_TERM enif_make_atom(ErlNifEnv* e, const char* name) {
  (void)e; ERL_NIF_TERM t = new_term(T_ATOM); T(t)->s = strdup(name); return t;
}
ERL_NIF_TERM enif_make_double(ErlNifEnv* e, double d) { (void)e; ERL_NIF_TERM t = new_term(T_DOUBLE); T(t)->d = d; return t; }
ERL_NIF_TERM enif_make_int(ErlNifEnv* e, int i) { (void)e; return fk_int(i); }
ERL_NIF_TERM enif_make_uint64(ErlNifEnv* e, ErlNifUInt64 i) { (void)e; return fk_int((long long)i); }
ERL_NIF_TERM enif_make_string(ErlNifEnv* e, const char* s, ErlNifCharEncoding c) {
  (void)e; (void)c; ERL_NIF_TERM t = new_term(T_STR); T(t)->s = strdup(s); return t;
}
unsigned char* enif_make_new_binary(ErlNifEnv* e, size_t size, ERL_NIF_TERM* termp) {
  (void)e; ERL_NIF_TERM t = new_term(T_BIN);
  T(t)->data = (unsigned char*)calloc(size ? size : 1, 1); T(t)->size = size; *termp = t;
  return T(t)->data;
}
static ERL_NIF_TERM make_seq(int type, const ERL_NIF_TERM arr[], unsigned cnt) {
  ERL_NIF_TERM* copy = (ERL_NIF_TERM*)malloc((cnt ? cnt : 1) * sizeof(ERL_NIF_TERM));
  if (cnt) memcpy(copy, arr, cnt * sizeof(ERL_NIF_TERM));   /* before new_term may move the table */
  ERL_NIF_TERM t = new_term(type);
  T(t)->items = copy; T(t)->n = cnt;
  return t;
}
ERL_NIF_TERM enif_make_tuple_from_array(ErlNifEnv* e, const ERL_NIF_TERM arr[], unsigned cnt) { (void)e; return make_seq(T_TUPLE, arr, cnt); }
int enif_get_tuple(ErlNifEnv* e, ERL_NIF_TERM tpl, int* arity, const ERL_NIF_TERM** array) {
  (void)e;
  if (!T(tpl) || T(tpl)->type != T_TUPLE) return 0;
  *arity = (int)T(tpl)->n;
  *array = T(tpl)->items;
  return 1;
}
ERL_NIF_TERM enif_make_list_from_array(ErlNifEnv* e, const ERL_NIF_TERM arr[], unsigned cnt) { (void)e; return make_seq(T_LIST, arr, cnt); }
ERL_NIF_TERM enif_make_new_map(ErlNifEnv* e) { (void)e; return new_term(T_MAP); }
int enif_make_map_put(ErlNifEnv* e, ERL_NIF_TERM map_in, ERL_NIF_TERM key, ERL_NIF_TERM value, ERL_NIF_TERM* map_out) {
  (void)e;
  if (!T(map_in) || T(map_in)->type != T_MAP) return 0;
  unsigned n = T(map_in)->n;
  ERL_NIF_TERM* kv = (ERL_NIF_TERM*)malloc(2 * (n + 1) * sizeof(ERL_NIF_TERM));
  for (unsigned i = 0; i < n; i++) { kv[i] = T(map_in)->items[i]; kv[n + 1 + i] = T(map_in)->items[n + i]; }
  kv[n] = key; kv[2 * n + 1] = value;
  ERL_NIF_TERM t = new_term(T_MAP);
  T(t)->items = kv; T(t)->n = n + 1;
  *map_out = t;
  return 1;
}
ErlNifResourceType* enif_open_resource_type(ErlNifEnv* e, const char* module_str, const char* name, ErlNifResourceDtor* dtor,
                                            ErlNifResourceFlags flags, ErlNifResourceFlags* tried) {
  (void)e; (void)module_str; (void)flags;
  struct enif_resource_type_t* t = (struct enif_resource_type_t*)calloc(1, sizeof *t);
  t->dtor = dtor; strncpy(t->name, name, sizeof t->name - 1);
  if (tried) *tried = ERL_NIF_RT_CREATE;
  return t;
}
void* enif_alloc_resource(ErlNifResourceType* type, size_t size) {
  res_hdr* h = (res_hdr*)calloc(1, sizeof(res_hdr) + size);
  h->type = type; h->refc = 1;
  return h + 1;
}
void enif_release_resource(void* obj) {
  res_hdr* h = (res_hdr*)obj - 1;
  if (__atomic_sub_fetch(&h->refc, 1, __ATOMIC_ACQ_REL) == 0) { if (h->type->dtor) h->type->dtor(&g_env, obj); free(h); }
}
ERL_NIF_TERM enif_make_resource(ErlNifEnv* e, void* obj) {
  (void)e; ERL_NIF_TERM t = new_term(T_RES); T(t)->obj = obj; __atomic_add_fetch(&((res_hdr*)obj - 1)->refc, 1, __ATOMIC_RELAXED); return t;
}
int enif_keep_resource(void* obj) { __atomic_add_fetch(&((res_hdr*)obj - 1)->refc, 1, __ATOMIC_RELAXED); return 1; }
ErlNifEnv* enif_alloc_env(void) { return (ErlNifEnv*)calloc(1, sizeof(struct enif_environment_t)); }
void enif_free_env(ErlNifEnv* e) { free(e); }
void enif_clear_env(ErlNifEnv* e) { (void)e; }   /* nothing is collected before fk_reset */
int enif_get_local_pid(ErlNifEnv* e, ERL_NIF_TERM t, ErlNifPid* pid) {
  (void)e;
  if (!T(t) || T(t)->type != T_INT) return 0;    /* the harness names its one process by an integer */
  pid->pid = t;
  return 1;
}
int enif_send(ErlNifEnv* caller, const ErlNifPid* to, ErlNifEnv* msg_env, ERL_NIF_TERM msg) {
  (void)caller; (void)to; (void)msg_env;
  pthread_mutex_lock(&g_lock);
  if (g_mail_n + 1 >= g_mail_cap) {
    g_mail_cap = g_mail_cap ? 2 * g_mail_cap : 256;
    g_mail = (ERL_NIF_TERM*)realloc(g_mail, g_mail_cap * sizeof(ERL_NIF_TERM));
  }
  g_mail[g_mail_n++] = msg;
  pthread_mutex_unlock(&g_lock);
  return 1;
}
int enif_thread_create(char* name, ErlNifTid* tid, void* (*func)(void*), void* args, ErlNifThreadOpts* opts) {
  (void)name; (void)opts;
  pthread_t t;
  int rc = pthread_create(&t, NULL, func, args);
  if (rc == 0) *tid = (ErlNifTid)(size_t)t;
  return rc;
}
int enif_thread_join(ErlNifTid tid, void** exit_value) { return pthread_join((pthread_t)(size_t)tid, exit_value); }
ErlNifTid enif_thread_self(void) { return (ErlNifTid)(size_t)pthread_self(); }
int enif_equal_tids(ErlNifTid a, ErlNifTid b) { return pthread_equal((pthread_t)(size_t)a, (pthread_t)(size_t)b); }
ErlNifMutex* enif_mutex_create(char* name) {
  (void)name;
  pthread_mutex_t* m = (pthread_mutex_t*)malloc(sizeof(pthread_mutex_t));
  pthread_mutex_init(m, NULL);
  return (ErlNifMutex*)m;
}
void enif_mutex_destroy(ErlNifMutex* m) { pthread_mutex_destroy((pthread_mutex_t*)m); free(m); }
void enif_mutex_lock(ErlNifMutex* m) { pthread_mutex_lock((pthread_mutex_t*)m); }
void enif_mutex_unlock(ErlNifMutex* m) { pthread_mutex_unlock((pthread_mutex_t*)m); }
void* enif_alloc(size_t size) { return malloc(size); }
void enif_free(void* ptr) { free(ptr); }
