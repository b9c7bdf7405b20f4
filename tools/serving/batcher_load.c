/* Serving-shaped load on the micro-batcher: T caller threads, each issuing ONE CollectiveSignature.Verify per call
 * (bftkv_gpu_batcher_collective_verify) -- the shape of protocol.Server's goroutine-per-request handlers
 * (transport/http/http.go:85,143).  Prints throughput and per-call latency percentiles for a sweep of thread counts.
 *   gcc -O2 -std=gnu99 -I include tools/serving/batcher_load.c -L bftkv_amd -lbftkv_gpu -lpthread -Wl,-rpath,$PWD/bftkv_amd -o batcher_load
 *   ./batcher_load corpus.bin [max_items=256] [max_wait_us=200] [lanes=0 (library default)] [threads,threads,...]
 * Not the bench line (bench.py measures resident batches); numbers feed DESIGN.md section 3.4. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/resource.h>
#include "bftkv_gpu.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

typedef struct {
  uint32_t n_items;
  uint64_t *tb_off, *ss_off;
  uint8_t *tb, *ss, *want_ok;
} corpus;

typedef struct {
  bftkv_gpu_batcher* b; int qh; const corpus* c; int tid, n_threads; double seconds; int* stop;
  uint64_t calls, wrong; double* lat; uint64_t lat_cap, lat_skip;
} worker;

static void* run(void* p) {
  worker* w = (worker*)p;
  uint32_t i = (uint32_t)w->tid;
  while (!__atomic_load_n(w->stop, __ATOMIC_RELAXED)) {
    const uint32_t k = i % w->c->n_items;
    uint8_t err = 0xEE, fenced = 0;
    const double t0 = now_s();
    int rc = bftkv_gpu_batcher_collective_verify(w->b, w->qh, w->c->tb + w->c->tb_off[k], w->c->tb_off[k + 1] - w->c->tb_off[k],
                                                 w->c->ss + w->c->ss_off[k], w->c->ss_off[k + 1] - w->c->ss_off[k], &err, &fenced);
    const double dt = now_s() - t0;
    if (rc != 0 || fenced || (err == 0) != (w->c->want_ok[k] != 0)) ++w->wrong;
    if (w->calls < w->lat_cap) w->lat[w->calls] = dt;       /* (the warm-up's samples are dropped at the end) */
    __atomic_store_n(&w->calls, w->calls + 1, __ATOMIC_RELAXED);     /* (main samples it at the end of the warm-up) */
    i += (uint32_t)w->n_threads;
  }
  return NULL;
}

static int cmp_d(const void* a, const void* b) { double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s corpus.bin [max_items] [max_wait_us] [lanes] [threads,...]\n", argv[0]); return 2; }
  const uint32_t max_items = argc > 2 ? (uint32_t)atoi(argv[2]) : 256, max_wait = argc > 3 ? (uint32_t)atoi(argv[3]) : 200;
  const uint32_t lanes = argc > 4 ? (uint32_t)atoi(argv[4]) : 0;
  int sweep[16] = {1, 8, 64, 256, 1024};
  unsigned n_sweep = 5;
  if (argc > 5) {
    n_sweep = 0;
    for (char* tok = strtok(argv[5], ","); tok && n_sweep < 16; tok = strtok(NULL, ",")) sweep[n_sweep++] = atoi(tok);
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("corpus"); return 2; }
  uint32_t n_keys = 0;
  if (fread(&n_keys, 4, 1, f) != 1) return 2;
  bftkv_gpu_pubkey* keys = calloc(n_keys, sizeof *keys);
  uint64_t* ids = malloc(8 * n_keys);
  uint8_t* mat = malloc((size_t)260 * n_keys);
  for (uint32_t i = 0; i < n_keys; ++i) {
    if (fread(&ids[i], 8, 1, f) != 1 || fread(mat + 260 * i, 260, 1, f) != 1) return 2;
    keys[i].key_id = keys[i].entity_id = ids[i]; keys[i].pk_algo = 1; keys[i].usable_sign = 1;
    keys[i].n = mat + 260 * i; keys[i].n_len = 256; keys[i].e = mat + 260 * i + 256; keys[i].e_len = 4;
  }
  int32_t qn[4];
  corpus c;
  if (fread(qn, 4, 4, f) != 4 || fread(&c.n_items, 4, 1, f) != 1) return 2;
  c.tb_off = malloc(8 * (c.n_items + 1)); c.ss_off = malloc(8 * (c.n_items + 1));
  if (fread(c.tb_off, 8, c.n_items + 1, f) != c.n_items + 1 || fread(c.ss_off, 8, c.n_items + 1, f) != c.n_items + 1) return 2;
  c.tb = malloc(c.tb_off[c.n_items]); c.ss = malloc(c.ss_off[c.n_items]); c.want_ok = malloc(c.n_items);
  if (fread(c.tb, 1, c.tb_off[c.n_items], f) != c.tb_off[c.n_items] || fread(c.ss, 1, c.ss_off[c.n_items], f) != c.ss_off[c.n_items] ||
      fread(c.want_ok, 1, c.n_items, f) != c.n_items) return 2;
  fclose(f);
  bftkv_gpu_ctx* ctx = NULL;
  if (bftkv_gpu_init(0, &ctx)) { fprintf(stderr, "no GPU\n"); return 1; }
  if (bftkv_gpu_keyring_set(ctx, keys, n_keys)) { fprintf(stderr, "keyring: %s\n", bftkv_gpu_last_error(ctx)); return 1; }
  bftkv_gpu_qc qc = {qn[0], qn[1], qn[2], qn[3], ids, n_keys};
  int qh = -1;
  if (bftkv_gpu_quorum_create(ctx, &qc, 1, &qh)) return 1;
  bftkv_gpu_batcher* b = bftkv_gpu_batcher_create_lanes(ctx, max_items, max_wait, lanes);
  if (!b) { fprintf(stderr, "batcher: %s\n", bftkv_gpu_last_error(ctx)); return 1; }
  uint64_t st0[4] = {0, 0, 0, 0};
  bftkv_gpu_batcher_stats(b, st0);
  printf("{\"max_items\": %u, \"lanes\": %llu, \"writes\": %u, \"replicas\": %u, \"runs\": [", max_items,
         (unsigned long long)st0[3], c.n_items, n_keys);
  for (unsigned s = 0; s < n_sweep; ++s) {
    const int T = sweep[s];
    int stop = 0;
    worker* ws = calloc((size_t)T, sizeof *ws);
    pthread_t* th = malloc(sizeof(pthread_t) * (size_t)T);
    for (int t = 0; t < T; ++t) {
      ws[t].b = b; ws[t].qh = qh; ws[t].c = &c; ws[t].tid = t; ws[t].n_threads = T; ws[t].stop = &stop;
      ws[t].lat_cap = 600000 / (uint64_t)T + 64; ws[t].lat = malloc(8 * ws[t].lat_cap);
      pthread_create(&th[t], NULL, run, &ws[t]);
    }
    // warm-up (arenas and pinned buffers grow to this concurrency's batch sizes), then the measured window
    struct timespec warm = {0, 700000000};
    nanosleep(&warm, NULL);
    uint64_t base_calls = 0;
    for (int t = 0; t < T; ++t) { ws[t].lat_skip = __atomic_load_n(&ws[t].calls, __ATOMIC_RELAXED); base_calls += ws[t].lat_skip; }
    struct rusage ru0, ru1;
    getrusage(RUSAGE_SELF, &ru0);
    uint64_t tm0[8] = {0}, tm1[8] = {0}, stw[4] = {0, 0, 0, 0};
    bftkv_gpu_batcher_times(b, tm0);
    bftkv_gpu_batcher_stats(b, stw);
    const double t0 = now_s();
    struct timespec nap = {2, 0};
    nanosleep(&nap, NULL);
    __atomic_store_n(&stop, 1, __ATOMIC_RELAXED);
    const double dt = now_s() - t0;
    for (int t = 0; t < T; ++t) pthread_join(th[t], NULL);
    getrusage(RUSAGE_SELF, &ru1);
    bftkv_gpu_batcher_times(b, tm1);
    st0[0] = stw[0]; st0[1] = stw[1];
    uint64_t calls = 0, wrong = 0, nl = 0;
    for (int t = 0; t < T; ++t) { calls += ws[t].calls; wrong += ws[t].wrong; nl += ws[t].calls < ws[t].lat_cap ? ws[t].calls : ws[t].lat_cap; }
    calls -= base_calls;
    const double usr_s = (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) + 1e-6 * (ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec);
    const double sys_s = (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + 1e-6 * (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec);
    const long csw = (ru1.ru_nvcsw - ru0.ru_nvcsw), icsw = (ru1.ru_nivcsw - ru0.ru_nivcsw);
    double* all = malloc(8 * (nl + 1));
    uint64_t k = 0;
    for (int t = 0; t < T; ++t) {
      uint64_t m = ws[t].calls < ws[t].lat_cap ? ws[t].calls : ws[t].lat_cap;
      const uint64_t sk = ws[t].lat_skip < m ? ws[t].lat_skip : m;
      memcpy(all + k, ws[t].lat + sk, 8 * (m - sk)); k += m - sk;
    }
    nl = k;
    qsort(all, nl, 8, cmp_d);
    uint64_t st[4] = {0, 0, 0, 0};
    bftkv_gpu_batcher_stats(b, st);
    const uint64_t d_calls = st[0] - st0[0], d_batches = st[1] - st0[1];
    st0[0] = st[0]; st0[1] = st[1];
    printf("%s{\"threads\": %d, \"verify_calls_per_s\": %.0f, \"wrong\": %llu, \"latency_ms\": {\"p50\": %.3f, \"p99\": %.3f, \"max\": %.3f}, "
           "\"calls\": %llu, \"device_calls\": %llu, \"largest_batch_so_far\": %llu, \"cpu_cores_busy\": {\"user\": %.2f, \"sys\": %.2f}, "
           "\"ctx_switches_per_call\": {\"voluntary\": %.2f, \"involuntary\": %.2f}, "
           "\"us_per_call\": {\"hash\": %.1f, \"assemble\": %.2f}, \"us_per_device_call\": {\"lane_wait\": %.1f, \"device\": %.1f, \"enqueue\": %.1f, \"wait\": %.1f}, "
           "\"sync_fallbacks\": %llu}",
           s ? ", " : "", T, calls / dt, (unsigned long long)wrong, nl ? all[nl / 2] * 1e3 : 0.0, nl ? all[(uint64_t)(nl * 0.99)] * 1e3 : 0.0,
           nl ? all[nl - 1] * 1e3 : 0.0, (unsigned long long)d_calls, (unsigned long long)d_batches, (unsigned long long)st[2], usr_s / dt, sys_s / dt,
           d_calls ? (double)csw / d_calls : 0.0, d_calls ? (double)icsw / d_calls : 0.0,
           d_calls ? 1e-3 * (tm1[0] - tm0[0]) / d_calls : 0.0, d_calls ? 1e-3 * (tm1[2] - tm0[2]) / d_calls : 0.0,
           d_batches ? 1e-3 * (tm1[1] - tm0[1]) / d_batches : 0.0, d_batches ? 1e-3 * (tm1[3] - tm0[3]) / d_batches : 0.0,
           d_batches ? 1e-3 * (tm1[4] - tm0[4]) / d_batches : 0.0, d_batches ? 1e-3 * (tm1[5] - tm0[5]) / d_batches : 0.0,
           (unsigned long long)(tm1[6] - tm0[6]));
    fflush(stdout);
    for (int t = 0; t < T; ++t) free(ws[t].lat);
    free(ws); free(th); free(all);
  }
  printf("]}\n");
  bftkv_gpu_batcher_destroy(b);
  bftkv_gpu_destroy(ctx);
  return 0;
}
