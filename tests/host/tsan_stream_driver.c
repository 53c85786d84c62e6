/* tsan_stream_driver.c -- TEST INFRASTRUCTURE ONLY: drives HipNative.stream_run/3 of c_src/exmc_hip_nif.c
 * (compiled into this program with tests/host/fake_erl_nif.c and tests/host/stub_exmc_hip.c, all under
 * -fsanitize=thread or address) the way a BEAM would, so that the shim's sender thread, its publish /
 * join protocol under g_tid_lock, the compare-and-swap claim of the handle and the reaper of senders
 * that destroy their own handle run without a GPU:
 *   A  one handle, several runs back to back (each run's sender is joined by the next stream_run)
 *   B  a second stream_run while one is in flight is refused by the library, the first completes
 *   C  handles dropped by the caller while their sender still runs: the sender releases the last
 *      reference, the destructor runs ON the sender, its tid is parked and joined by the next caller;
 *      many at once (more than the 64 the reap list used to hold)
 *   D  unload with parked senders left: on_unload joins them
 * Exit code 0 and no sanitizer report = pass. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../c_src/erl_nif_decl.h"

ErlNifEntry* nif_init(void);
/* the harness API of fake_erl_nif.c */
void fk_reset(void);
size_t fk_mailbox_len(void);
ERL_NIF_TERM fk_mailbox_get(size_t i);
int fk_mailbox_wait(size_t n, int timeout_ms);
ERL_NIF_TERM fk_atom(const char* name);
ERL_NIF_TERM fk_double(double x);
ERL_NIF_TERM fk_int(long long v);
ERL_NIF_TERM fk_binary(const void* p, size_t n);
int fk_type(ERL_NIF_TERM t);
long long fk_get_int(ERL_NIF_TERM t);
const char* fk_get_str(ERL_NIF_TERM t);
unsigned fk_len(ERL_NIF_TERM t);
ERL_NIF_TERM fk_item(ERL_NIF_TERM t, unsigned i);
ERL_NIF_TERM fk_exception(void);
int fk_badarg(void);
int fk_load(ErlNifEntry* e);
ERL_NIF_TERM fk_call(ErlNifEntry* e, const char* name, unsigned arity, const ERL_NIF_TERM* argv);
void fk_drop_resource_term(ERL_NIF_TERM t);

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

static ErlNifEntry* E;

static ERL_NIF_TERM new_handle(void) {
  double data[4] = {1, 2, 3, 4};
  ERL_NIF_TERM argv[2] = {fk_int(2), fk_binary(data, sizeof data)};
  ERL_NIF_TERM r = fk_call(E, "model_create", 2, argv);
  CHECK(!fk_badarg() && !fk_exception() && fk_len(r) == 2);
  CHECK(strcmp(fk_get_str(fk_item(r, 0)), "ok") == 0);
  return fk_item(r, 1);
}
static void begin(ERL_NIF_TERM ref) {
  ERL_NIF_TERM argv[6] = {ref, fk_atom("nil"), fk_int(10), fk_int(10), fk_double(0.8), fk_int(1)};
  (void)fk_call(E, "stream_begin", 6, argv);
  CHECK(!fk_badarg() && !fk_exception());
}
static int run(ERL_NIF_TERM ref, int n) {     /* 1: started, 0: refused (the library's "a stream run is in
                                                  flight" is EXMC_ERR_BADARG: a badarg on the BEAM) */
  ERL_NIF_TERM argv[3] = {ref, fk_int(n), fk_int(1)};
  ERL_NIF_TERM r = fk_call(E, "stream_run", 3, argv);
  if (fk_badarg() || fk_exception()) return 0;
  CHECK(strcmp(fk_get_str(r), "ok") == 0);
  return 1;
}
/* the last message must be {:exmc_done, n, divergences} and the draws in order */
static void expect_done(size_t first, int n) {
  CHECK(fk_mailbox_wait(first + (size_t)n + 1, 60000));
  for (int i = 0; i < n; i++) {
    ERL_NIF_TERM m = fk_mailbox_get(first + (size_t)i);
    CHECK(fk_len(m) == 4 && strcmp(fk_get_str(fk_item(m, 0)), "exmc_sample") == 0);
    CHECK(fk_get_int(fk_item(m, 1)) == i + 1);
  }
  ERL_NIF_TERM d = fk_mailbox_get(first + (size_t)n);
  CHECK(fk_len(d) == 3 && strcmp(fk_get_str(fk_item(d, 0)), "exmc_done") == 0);
  CHECK(fk_get_int(fk_item(d, 1)) == n);
}

int main(void) {
  E = nif_init();
  CHECK(fk_load(E) == 0);

  /* A: runs back to back on one handle */
  ERL_NIF_TERM h = new_handle();
  begin(h);
  size_t base = 0;
  for (int k = 0; k < 5; k++) {
    CHECK(run(h, 200));
    expect_done(base, 200);
    base += 201;
  }
  /* B: a second run while one is in flight is refused; the first completes */
  CHECK(run(h, 4000));
  int refused = 0;
  for (int k = 0; k < 3; k++) refused += !run(h, 10);
  CHECK(refused >= 1);
  /* (a refused run must not have started a sender: exactly 4001 more messages arrive) */
  expect_done(base, 4000);
  base += 4001;
  /* the handle's last sender is joined by the next run */
  CHECK(run(h, 50));
  expect_done(base, 50);
  base += 51;

  /* C: handles dropped while their sender runs -- the destructor runs on the sender thread */
  enum { N = 96 };
  for (int k = 0; k < N; k++) {
    ERL_NIF_TERM hk = new_handle();        /* new_handle reaps whatever has parked so far */
    begin(hk);
    CHECK(run(hk, 300));
    fk_drop_resource_term(hk);             /* the caller's reference goes away: the sender holds the last one */
  }
  CHECK(fk_mailbox_wait(base + (size_t)N * 301, 120000));
  /* D: one more caller reaps the parked senders; the rest go at unload */
  ERL_NIF_TERM last = new_handle();
  /* a module is unloaded when no resource of its type is left: the caller's handles go first (their
   * destructors join their own senders), the parked ones are joined by the unload hook */
  fk_drop_resource_term(last);
  fk_drop_resource_term(h);
  if (E->unload) E->unload(NULL, NULL);
  printf("tsan_stream_driver: ok (%zu messages)\n", fk_mailbox_len());
  return 0;
}
