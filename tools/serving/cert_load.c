/* Serving-shaped load on bftkv_gpu_batcher_cert_verify: T caller threads, each issuing ONE Issuer(sig) + VerifyWithCertificate per
 * call for a client that is NOT in the server's keyring -- protocol.Server.sign (protocol/server.go:199-207), once per write of every
 * client, where the reference runs openpgp.ReadEntity (self-signature + certifications parsed, self-signature verified) and one
 * CheckDetachedSignature on the CPU per request.  First sight of a certificate registers its entity and verifies what ReadEntity
 * verifies (timed apart: "first_sight"); afterwards a request costs one signature verification against the registered entity.
 *   gcc -O2 -std=gnu99 -I include tools/serving/cert_load.c -L bftkv_amd -lbftkv_gpu -lpthread -Wl,-rpath,$PWD/bftkv_amd -o cert_load
 *   ./cert_load corpus.bin [max_items=256] [lanes=0] [threads,threads,...] [seconds per point=1.0]
 * corpus.bin: tools/serving/make_cert_corpus.py.  Every answer is checked (err 0, not fenced, the client's key id): "wrong" counts the rest. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "bftkv_gpu.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

typedef struct { uint32_t n; uint64_t* off; uint8_t* blob; } blobs;
typedef struct {
  bftkv_gpu_batcher* b; const blobs *certs, *tbs, *sigs; const uint64_t* want_id; int tid, n_threads; int* stop;
  uint64_t calls, wrong; double* lat; uint64_t lat_cap, lat_skip;
} worker;

static int one(const worker* w, uint32_t k) {
  uint8_t err = 0xEE, fenced = 0, fp[20];
  uint64_t id = 0;
  const int rc = bftkv_gpu_batcher_cert_verify(w->b, w->certs->blob + w->certs->off[k], w->certs->off[k + 1] - w->certs->off[k], w->tbs->blob + w->tbs->off[k],
                                               w->tbs->off[k + 1] - w->tbs->off[k], w->sigs->blob + w->sigs->off[k], w->sigs->off[k + 1] - w->sigs->off[k], &err,
                                               &fenced, &id, fp);
  return rc == 0 && err == BFTKV_ERR_NONE && !fenced && (!w->want_id[k] || id == w->want_id[k]);
}

static void* run(void* p) {
  worker* w = (worker*)p;
  uint32_t i = (uint32_t)w->tid;
  while (!__atomic_load_n(w->stop, __ATOMIC_RELAXED)) {
    const double t0 = now_s();
    const int ok = one(w, i % w->certs->n);
    const double dt = now_s() - t0;
    if (!ok) ++w->wrong;
    if (w->calls < w->lat_cap) w->lat[w->calls] = dt;
    __atomic_store_n(&w->calls, w->calls + 1, __ATOMIC_RELAXED);
    i += (uint32_t)w->n_threads;
  }
  return NULL;
}

static int cmp_d(const void* a, const void* b) { double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }
static void* slurp(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short input file\n"); exit(2); }
  return p;
}
static blobs read_blobs(FILE* f) {
  blobs b;
  if (fread(&b.n, 4, 1, f) != 1) exit(2);
  b.off = slurp(f, 8 * ((size_t)b.n + 1));
  b.blob = slurp(f, b.off[b.n]);
  return b;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s corpus.bin [max_items] [lanes] [threads,...] [seconds]\n", argv[0]); return 2; }
  const uint32_t max_items = argc > 2 ? (uint32_t)atoi(argv[2]) : 256, lanes = argc > 3 ? (uint32_t)atoi(argv[3]) : 0;
  int sweep[16] = {1, 64, 256};
  unsigned n_sweep = 3;
  if (argc > 4) {
    n_sweep = 0;
    for (char* tok = strtok(argv[4], ","); tok && n_sweep < 16; tok = strtok(NULL, ",")) sweep[n_sweep++] = atoi(tok);
  }
  const double seconds = argc > 5 ? atof(argv[5]) : 1.0;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("corpus"); return 2; }
  uint32_t n_keys = 0;
  if (fread(&n_keys, 4, 1, f) != 1) return 2;
  bftkv_gpu_pubkey* keys = calloc(n_keys, sizeof *keys);
  for (uint32_t i = 0; i < n_keys; ++i) {
    uint8_t* rec = slurp(f, 8 + 256 + 4);
    memcpy(&keys[i].key_id, rec, 8);
    keys[i].entity_id = keys[i].key_id; keys[i].pk_algo = 1; keys[i].usable_sign = 1;
    keys[i].n = rec + 8; keys[i].n_len = 256; keys[i].e = rec + 264; keys[i].e_len = 4;
  }
  blobs certs = read_blobs(f), tbs = read_blobs(f), sigs = read_blobs(f);
  fclose(f);
  bftkv_gpu_ctx* ctx = NULL;
  if (bftkv_gpu_init(0, &ctx)) { fprintf(stderr, "no GPU\n"); return 1; }
  if (bftkv_gpu_keyring_set(ctx, keys, n_keys)) { fprintf(stderr, "keyring: %s\n", bftkv_gpu_last_error(ctx)); return 1; }
  bftkv_gpu_batcher* b = bftkv_gpu_batcher_create_lanes(ctx, max_items, 0, lanes);
  if (!b) { fprintf(stderr, "batcher: %s\n", bftkv_gpu_last_error(ctx)); return 1; }
  /* first sight: every client once, one after another (entity registration + what ReadEntity verifies + the request's signature) */
  uint64_t* want_id = calloc(certs.n, 8);
  worker w0 = {b, &certs, &tbs, &sigs, want_id, 0, 1, NULL, 0, 0, NULL, 0, 0};
  uint32_t first_wrong = 0;
  const double tf = now_s();
  for (uint32_t k = 0; k < certs.n; ++k) first_wrong += !one(&w0, k);
  const double first_ms = 1e3 * (now_s() - tf) / certs.n;
  for (uint32_t k = 0; k < certs.n; ++k) {   /* the ids the steady-state answers must repeat */
    uint8_t err, fenced, fp[20];
    bftkv_gpu_batcher_cert_verify(b, certs.blob + certs.off[k], certs.off[k + 1] - certs.off[k], NULL, 0, NULL, 0, &err, &fenced, &want_id[k], fp);
  }
  printf("{\"clients\": %u, \"certificate_bytes\": %llu, \"max_items\": %u, \"first_sight\": {\"ms_per_certificate\": %.3f, \"wrong\": %u}, \"runs\": [", certs.n,
         (unsigned long long)(certs.off[1] - certs.off[0]), max_items, first_ms, first_wrong);
  for (unsigned s = 0; s < n_sweep; ++s) {
    const int T = sweep[s];
    int stop = 0;
    worker* ws = calloc((size_t)T, sizeof *ws);
    pthread_t* th = malloc(sizeof(pthread_t) * (size_t)T);
    for (int t = 0; t < T; ++t) {
      ws[t] = w0; ws[t].tid = t; ws[t].n_threads = T; ws[t].stop = &stop;
      ws[t].lat_cap = 400000 / (uint64_t)T + 64; ws[t].lat = malloc(8 * ws[t].lat_cap);
      pthread_create(&th[t], NULL, run, &ws[t]);
    }
    struct timespec warm = {0, 300000000};
    nanosleep(&warm, NULL);
    uint64_t base_calls = 0;
    for (int t = 0; t < T; ++t) { ws[t].lat_skip = __atomic_load_n(&ws[t].calls, __ATOMIC_RELAXED); base_calls += ws[t].lat_skip; }
    uint64_t st0[4] = {0, 0, 0, 0}, st1[4] = {0, 0, 0, 0};
    bftkv_gpu_batcher_stats(b, st0);
    const double t0 = now_s();
    struct timespec nap = {(time_t)seconds, (long)((seconds - (double)(time_t)seconds) * 1e9)};
    nanosleep(&nap, NULL);
    __atomic_store_n(&stop, 1, __ATOMIC_RELAXED);
    for (int t = 0; t < T; ++t) pthread_join(th[t], NULL);
    const double dt = now_s() - t0;
    bftkv_gpu_batcher_stats(b, st1);
    uint64_t calls = 0, wrong = 0, nl = 0;
    for (int t = 0; t < T; ++t) { calls += ws[t].calls; wrong += ws[t].wrong; nl += ws[t].calls < ws[t].lat_cap ? ws[t].calls : ws[t].lat_cap; }
    calls -= base_calls;
    double* all = malloc(8 * (nl + 1));
    uint64_t k = 0;
    for (int t = 0; t < T; ++t) {
      const uint64_t m = ws[t].calls < ws[t].lat_cap ? ws[t].calls : ws[t].lat_cap, sk = ws[t].lat_skip < m ? ws[t].lat_skip : m;
      memcpy(all + k, ws[t].lat + sk, 8 * (m - sk)); k += m - sk;
    }
    nl = k;
    qsort(all, nl, 8, cmp_d);
    printf("%s{\"threads\": %d, \"calls_per_s\": %.0f, \"wrong\": %llu, \"latency_ms\": {\"p50\": %.3f, \"p99\": %.3f, \"max\": %.3f}, \"device_calls\": %llu}",
           s ? ", " : "", T, calls / dt, (unsigned long long)wrong, nl ? all[nl / 2] * 1e3 : 0.0, nl ? all[(uint64_t)(nl * 0.99)] * 1e3 : 0.0,
           nl ? all[nl - 1] * 1e3 : 0.0, (unsigned long long)(st1[1] - st0[1]));
    fflush(stdout);
    for (int t = 0; t < T; ++t) free(ws[t].lat);
    free(ws); free(th); free(all);
  }
  printf("]}\n");
  bftkv_gpu_batcher_destroy(b);
  bftkv_gpu_destroy(ctx);
  return 0;
}
