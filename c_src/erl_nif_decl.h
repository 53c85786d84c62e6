/* erl_nif_decl.h -- declarations of the part of OTP's public NIF API (erts/emulator/beam/erl_nif.h,
 * documented in the erl_nif(3) manual page) that the two shims in this directory use.
 *
 * Why this file exists: the build image has no Erlang/OTP, so the real <erl_nif.h> is absent. These
 * are DECLARATIONS ONLY, written from the public manual page, so that `gcc -c` type-checks the shims
 * and tests/test_nif_shim.py can read the ErlNifEntry tables they export. A maintainer builds with
 * the real header:   cc -DEXMC_USE_SYSTEM_ERL_NIF -I$ERL_ROOT/usr/include ...
 * Nothing here is linked into libexmc_hip.so.
 */
#ifndef EXMC_ERL_NIF_DECL_H
#define EXMC_ERL_NIF_DECL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ERL_NIF_MAJOR_VERSION 2
#define ERL_NIF_MINOR_VERSION 16
#define ERL_NIF_MIN_ERTS_VERSION "erts-12.0"
#define ERL_NIF_VM_VARIANT "beam.vanilla"

typedef unsigned long ERL_NIF_TERM;        /* a machine word on LP64 */
typedef struct enif_environment_t ErlNifEnv;
typedef struct enif_resource_type_t ErlNifResourceType;
typedef void ErlNifResourceDtor(ErlNifEnv*, void*);
typedef int64_t ErlNifSInt64;
typedef uint64_t ErlNifUInt64;

typedef struct {
  size_t size;
  unsigned char* data;
  void* ref_bin;
  void* spare_[2];
} ErlNifBinary;

typedef enum { ERL_NIF_RT_CREATE = 1, ERL_NIF_RT_TAKEOVER = 2 } ErlNifResourceFlags;
typedef enum { ERL_NIF_LATIN1 = 1 } ErlNifCharEncoding;
typedef enum { ERL_NIF_DIRTY_JOB_CPU_BOUND = 1, ERL_NIF_DIRTY_JOB_IO_BOUND = 2 } ErlNifDirtyTaskFlags;

typedef struct enif_func_t {
  const char* name;
  unsigned arity;
  ERL_NIF_TERM (*fptr)(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]);
  unsigned flags;
} ErlNifFunc;

typedef struct enif_entry_t {
  int major;
  int minor;
  const char* name;
  int num_of_funcs;
  ErlNifFunc* funcs;
  int (*load)(ErlNifEnv*, void** priv_data, ERL_NIF_TERM load_info);
  int (*reload)(ErlNifEnv*, void** priv_data, ERL_NIF_TERM load_info);
  int (*upgrade)(ErlNifEnv*, void** priv_data, void** old_priv_data, ERL_NIF_TERM load_info);
  void (*unload)(ErlNifEnv*, void* priv_data);
  const char* vm_variant;
  unsigned options;
  size_t sizeof_ErlNifResourceTypeInit;
  const char* min_erts;
} ErlNifEntry;

/* terms in */
int enif_get_double(ErlNifEnv*, ERL_NIF_TERM, double* dp);
int enif_get_int(ErlNifEnv*, ERL_NIF_TERM, int* ip);
int enif_get_int64(ErlNifEnv*, ERL_NIF_TERM, ErlNifSInt64* ip);
int enif_get_uint64(ErlNifEnv*, ERL_NIF_TERM, ErlNifUInt64* ip);
int enif_get_atom(ErlNifEnv*, ERL_NIF_TERM, char* buf, unsigned len, ErlNifCharEncoding);
int enif_inspect_binary(ErlNifEnv*, ERL_NIF_TERM bin_term, ErlNifBinary* bin);
int enif_get_list_length(ErlNifEnv*, ERL_NIF_TERM, unsigned* len);
int enif_get_list_cell(ErlNifEnv*, ERL_NIF_TERM list, ERL_NIF_TERM* head, ERL_NIF_TERM* tail);
int enif_get_resource(ErlNifEnv*, ERL_NIF_TERM, ErlNifResourceType*, void** objp);
/* terms out */
ERL_NIF_TERM enif_make_badarg(ErlNifEnv*);
ERL_NIF_TERM enif_raise_exception(ErlNifEnv*, ERL_NIF_TERM reason);
ERL_NIF_TERM enif_make_atom(ErlNifEnv*, const char* name);
ERL_NIF_TERM enif_make_double(ErlNifEnv*, double);
ERL_NIF_TERM enif_make_int(ErlNifEnv*, int);
ERL_NIF_TERM enif_make_uint64(ErlNifEnv*, ErlNifUInt64);
ERL_NIF_TERM enif_make_string(ErlNifEnv*, const char*, ErlNifCharEncoding);
unsigned char* enif_make_new_binary(ErlNifEnv*, size_t size, ERL_NIF_TERM* termp);
ERL_NIF_TERM enif_make_tuple_from_array(ErlNifEnv*, const ERL_NIF_TERM arr[], unsigned cnt);
ERL_NIF_TERM enif_make_list_from_array(ErlNifEnv*, const ERL_NIF_TERM arr[], unsigned cnt);
ERL_NIF_TERM enif_make_new_map(ErlNifEnv*);
int enif_make_map_put(ErlNifEnv*, ERL_NIF_TERM map_in, ERL_NIF_TERM key, ERL_NIF_TERM value,
                      ERL_NIF_TERM* map_out);
/* resources */
ErlNifResourceType* enif_open_resource_type(ErlNifEnv*, const char* module_str, const char* name,
                                            ErlNifResourceDtor* dtor, ErlNifResourceFlags flags,
                                            ErlNifResourceFlags* tried);
void* enif_alloc_resource(ErlNifResourceType*, size_t size);
void enif_release_resource(void* obj);
ERL_NIF_TERM enif_make_resource(ErlNifEnv*, void* obj);
/* memory */
void* enif_alloc(size_t size);
void enif_free(void* ptr);

/* process-independent environments, messages and threads (erl_nif: enif_alloc_env, enif_send,
 * enif_thread_create): what a NIF library needs to send from a thread of its own */
typedef struct { ERL_NIF_TERM pid; } ErlNifPid;
typedef struct ErlDrvTid_* ErlNifTid;
typedef struct { int suggested_stack_size; } ErlNifThreadOpts;
ErlNifEnv* enif_alloc_env(void);
void enif_free_env(ErlNifEnv*);
void enif_clear_env(ErlNifEnv*);
int enif_send(ErlNifEnv* caller_env, const ErlNifPid* to_pid, ErlNifEnv* msg_env, ERL_NIF_TERM msg);
int enif_get_local_pid(ErlNifEnv*, ERL_NIF_TERM, ErlNifPid* pid);
int enif_thread_create(char* name, ErlNifTid* tid, void* (*func)(void*), void* args, ErlNifThreadOpts* opts);
int enif_thread_join(ErlNifTid, void** exit_value);
int enif_get_tuple(ErlNifEnv*, ERL_NIF_TERM tpl, int* arity, const ERL_NIF_TERM** array);
ErlNifTid enif_thread_self(void);
int enif_equal_tids(ErlNifTid tid1, ErlNifTid tid2);
int enif_keep_resource(void* obj);
/* mutexes (erl_nif: enif_mutex_create .. enif_mutex_unlock) */
typedef struct ErlDrvMutex_ ErlNifMutex;
ErlNifMutex* enif_mutex_create(char* name);
void enif_mutex_destroy(ErlNifMutex* mtx);
void enif_mutex_lock(ErlNifMutex* mtx);
void enif_mutex_unlock(ErlNifMutex* mtx);

#define ERL_NIF_INIT(NAME, FUNCS, LOAD, RELOAD, UPGRADE, UNLOAD)                              \
  ErlNifEntry* nif_init(void);                                                                \
  ErlNifEntry* nif_init(void) {                                                               \
    static ErlNifEntry entry = {ERL_NIF_MAJOR_VERSION, ERL_NIF_MINOR_VERSION, #NAME,          \
                                (int)(sizeof(FUNCS) / sizeof(*FUNCS)), FUNCS, LOAD, RELOAD,   \
                                UPGRADE, UNLOAD, ERL_NIF_VM_VARIANT, 1, 0,                    \
                                ERL_NIF_MIN_ERTS_VERSION};                                    \
    return &entry;                                                                            \
  }

#ifdef __cplusplus
}
#endif
#endif
