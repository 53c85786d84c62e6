/* exmc_nif_util.h -- term helpers shared by the two NIF shims (argument decoding with Rustler's
 * conventions: a decode failure is a badarg; native/exmc_tree/src/lib.rs:19-32). */
#ifndef EXMC_NIF_UTIL_H
#define EXMC_NIF_UTIL_H

#ifdef EXMC_USE_SYSTEM_ERL_NIF
#include <erl_nif.h>
#else
#include "erl_nif_decl.h"
#endif

#include <stdlib.h>
#include <string.h>

#include "../include/exmc_hip.h"

/* f64: floats, and integers as Elixir callers sometimes pass `0` for `0.0` */
static __attribute__((unused)) int get_f64(ErlNifEnv* env, ERL_NIF_TERM t, double* out) {
  ErlNifSInt64 i;
  if (enif_get_double(env, t, out)) return 1;
  if (enif_get_int64(env, t, &i)) { *out = (double)i; return 1; }
  return 0;
}
static __attribute__((unused)) int get_usize(ErlNifEnv* env, ERL_NIF_TERM t, int* out) {
  ErlNifUInt64 u;
  if (!enif_get_uint64(env, t, &u) || u > 0x7fffffffULL) return 0;
  *out = (int)u;
  return 1;
}
static __attribute__((unused)) int get_bool(ErlNifEnv* env, ERL_NIF_TERM t, int32_t* out) {
  char buf[8];
  if (!enif_get_atom(env, t, buf, sizeof buf, ERL_NIF_LATIN1)) return 0;
  if (strcmp(buf, "true") == 0) { *out = 1; return 1; }
  if (strcmp(buf, "false") == 0) { *out = 0; return 1; }
  return 0;
}
/* a native-endian f64 binary (Nx.to_binary of an f64 tensor) */
static __attribute__((unused)) int get_f64_bin(ErlNifEnv* env, ERL_NIF_TERM t, const double** p, size_t* n) {
  ErlNifBinary b;
  if (!enif_inspect_binary(env, t, &b) || (b.size & 7) != 0) return 0;
  *p = (const double*)b.data;
  *n = b.size / 8;
  return 1;
}
/* a list of numbers -> enif_alloc'd doubles (the legacy list API, lib.rs:345-434) */
static __attribute__((unused)) int get_f64_list(ErlNifEnv* env, ERL_NIF_TERM t, double** p, size_t* n) {
  unsigned len;
  ERL_NIF_TERM head, tail = t;
  if (!enif_get_list_length(env, t, &len)) return 0;
  double* v = (double*)enif_alloc((len ? len : 1) * sizeof(double));
  if (!v) return 0;
  for (unsigned i = 0; i < len; i++) {
    if (!enif_get_list_cell(env, tail, &head, &tail) || !get_f64(env, head, &v[i])) {
      enif_free(v);
      return 0;
    }
  }
  *p = v;
  *n = len;
  return 1;
}
static __attribute__((unused)) ERL_NIF_TERM make_f64_bin(ErlNifEnv* env, const double* src, size_t n) {
  ERL_NIF_TERM t;
  unsigned char* dst = enif_make_new_binary(env, n * 8, &t);
  if (n) memcpy(dst, src, n * 8);
  return t;
}
static __attribute__((unused)) double* new_f64_bin(ErlNifEnv* env, size_t n, ERL_NIF_TERM* t) {
  return (double*)enif_make_new_binary(env, n * 8, t);
}
static __attribute__((unused)) ERL_NIF_TERM make_f64_list(ErlNifEnv* env, const double* src, size_t n) {
  ERL_NIF_TERM* terms = (ERL_NIF_TERM*)enif_alloc((n ? n : 1) * sizeof(ERL_NIF_TERM));
  for (size_t i = 0; i < n; i++) terms[i] = enif_make_double(env, src[i]);
  ERL_NIF_TERM l = enif_make_list_from_array(env, terms, (unsigned)n);
  enif_free(terms);
  return l;
}
static __attribute__((unused)) ERL_NIF_TERM make_bool(ErlNifEnv* env, int v) { return enif_make_atom(env, v ? "true" : "false"); }
static __attribute__((unused)) ERL_NIF_TERM map_put(ErlNifEnv* env, ERL_NIF_TERM map, const char* key, ERL_NIF_TERM val) {
  ERL_NIF_TERM out = map;
  enif_make_map_put(env, map, enif_make_atom(env, key), val, &out);
  return out;
}
static __attribute__((unused)) ERL_NIF_TERM tuple2(ErlNifEnv* env, ERL_NIF_TERM a, ERL_NIF_TERM b) {
  ERL_NIF_TERM v[2] = {a, b};
  return enif_make_tuple_from_array(env, v, 2);
}
static __attribute__((unused)) ERL_NIF_TERM tuple3(ErlNifEnv* env, ERL_NIF_TERM a, ERL_NIF_TERM b, ERL_NIF_TERM c) {
  ERL_NIF_TERM v[3] = {a, b, c};
  return enif_make_tuple_from_array(env, v, 3);
}
static __attribute__((unused)) ERL_NIF_TERM tuple4(ErlNifEnv* env, ERL_NIF_TERM a, ERL_NIF_TERM b, ERL_NIF_TERM c,
                                                   ERL_NIF_TERM d) {
  ERL_NIF_TERM v[4] = {a, b, c, d};
  return enif_make_tuple_from_array(env, v, 4);
}
/* a failed library call: raise {:exmc_hip_error, code, message} (Rustler turns a NifResult::Err
 * into a raised term the same way) */
static __attribute__((unused)) ERL_NIF_TERM raise_hip(ErlNifEnv* env, int rc) {
  if (rc == EXMC_ERR_BADARG) return enif_make_badarg(env);
  return enif_raise_exception(env, tuple3(env, enif_make_atom(env, "exmc_hip_error"), enif_make_int(env, rc),
                                          enif_make_string(env, exmc_hip_last_error(), ERL_NIF_LATIN1)));
}

#endif
