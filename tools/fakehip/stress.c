/* Every kind of caller of libbftkv_gpu.so's host side at once, for ThreadSanitizer (tools/tsan_host.sh) over tools/fakehip:
 * one-operation callers of every micro-batcher entry (collective / signature / certificate / message verification, the four
 * threshold combines), big host-buffer calls through the pipelined path on forked contexts, and a writer that replaces the key
 * table and creates / destroys quorums meanwhile -- the mix protocol.Server's goroutines make (server.go:562-620) while
 * Protocol.Joining / Leaving change the keyring (protocol.go:21-60).  Kernels do not run there: answers are not checked, only
 * that every call returns, fails closed when it fails, and that TSan stays silent.
 *   stress corpus.bin extras.bin [seconds = 5] [callers per kind = 8] [lanes = 3] [writer pause ms = 200; 0 = no writer]
 * corpus.bin: tools/serving/make_load_corpus.py's format; extras.bin: tools/fakehip/make_extras.py. */
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "bftkv_gpu.h"

typedef struct { uint32_t n; uint64_t* off; uint8_t* blob; } blobs;
static atomic_int g_stop;
static atomic_ullong g_calls[16], g_failed[16], g_not_closed;
static atomic_int g_first_rc[16];

static bftkv_gpu_ctx* g_ctx;
static bftkv_gpu_batcher* g_b;
static int g_quorum;
static uint32_t g_items;
static uint64_t *g_tb_off, *g_ss_off;
static uint8_t *g_tb, *g_ss;
static blobs g_certs, g_msgs;
static bftkv_gpu_pubkey* g_keys;
static uint32_t g_nkeys;
static bftkv_gpu_qc g_qc;
static long g_writer_ms = 200;

static uint32_t rnd(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }
static void fill(uint8_t* p, size_t n, uint32_t* s) { for (size_t i = 0; i < n; ++i) p[i] = (uint8_t)rnd(s); }
static void note(int kind, int rc) { atomic_fetch_add(&g_calls[kind], 1); if (rc) { atomic_fetch_add(&g_failed[kind], 1); int z = 0; atomic_compare_exchange_strong(&g_first_rc[kind], &z, rc); } }

static void* collective(void* a) {
  uint32_t s = (uint32_t)(size_t)a * 7919u + 1;
  while (!atomic_load(&g_stop)) {
    const uint32_t k = rnd(&s) % g_items;
    uint8_t err = 0, fenced = 0;
    const int rc = bftkv_gpu_batcher_collective_verify(g_b, g_quorum, g_tb + g_tb_off[k], g_tb_off[k + 1] - g_tb_off[k], g_ss + g_ss_off[k],
                                                       g_ss_off[k + 1] - g_ss_off[k], &err, &fenced);
    if (rc && err == BFTKV_ERR_NONE) atomic_fetch_add(&g_not_closed, 1);
    note(0, rc);
  }
  return NULL;
}

static void* signature(void* a) {
  uint32_t s = (uint32_t)(size_t)a * 104729u + 3;
  while (!atomic_load(&g_stop)) {
    const uint32_t k = rnd(&s) % g_items;
    uint64_t len = g_ss_off[k + 1] - g_ss_off[k];
    if (len > 600) len = 287 * (1 + rnd(&s) % 2);      /* one or two packets of the collective stream */
    uint8_t err = 0, fenced = 0;
    const int rc = bftkv_gpu_batcher_signature_verify(g_b, g_tb + g_tb_off[k], g_tb_off[k + 1] - g_tb_off[k], g_ss + g_ss_off[k], len, NULL, &err, &fenced);
    if (rc && err == BFTKV_ERR_NONE) atomic_fetch_add(&g_not_closed, 1);
    note(1, rc);
  }
  return NULL;
}

static void* certificate(void* a) {
  uint32_t s = (uint32_t)(size_t)a * 1299709u + 5;
  uint32_t roles[512];
  while (!atomic_load(&g_stop) && g_certs.n) {
    const uint32_t c = rnd(&s) % g_certs.n, k = rnd(&s) % g_items;
    const uint8_t* cert = g_certs.blob + g_certs.off[c];
    const uint64_t clen = g_certs.off[c + 1] - g_certs.off[c];
    uint8_t err = 0, fenced = 0, fp[20];
    uint64_t issuer = 0, eo = 0, el = 0;
    uint32_t nr = 0;
    int rc;
    if (rnd(&s) & 1)
      rc = bftkv_gpu_batcher_cert_verify(g_b, cert, clen, g_tb + g_tb_off[k], g_tb_off[k + 1] - g_tb_off[k], g_ss + g_ss_off[k], 287, &err, &fenced, &issuer, fp);
    else
      rc = bftkv_gpu_batcher_cert_entity(g_b, cert, clen, &err, &fenced, &issuer, fp, &eo, &el, roles, 512, &nr);
    if (rc && err == BFTKV_ERR_NONE) atomic_fetch_add(&g_not_closed, 1);
    note(2, rc);
  }
  return NULL;
}

static void* message(void* a) {
  uint32_t s = (uint32_t)(size_t)a * 15485863u + 7;
  uint8_t* plain = malloc(1 << 20);
  while (!atomic_load(&g_stop) && g_msgs.n) {
    const uint32_t m = rnd(&s) % g_msgs.n;
    uint8_t st = 0, fname[256], fl = 0;
    uint64_t signer = 0, peer = 0, plen = 0;
    const int rc = bftkv_gpu_batcher_message_verify(g_b, g_msgs.blob + g_msgs.off[m], g_msgs.off[m + 1] - g_msgs.off[m], &st, &signer, &peer, plain, 1 << 20,
                                                    &plen, fname, &fl);
    if (rc && st == BFTKV_MSG_OK) atomic_fetch_add(&g_not_closed, 1);
    note(3, rc);
  }
  free(plain);
  return NULL;
}

static void* threshold(void* a) {
  uint32_t s = (uint32_t)(size_t)a * 32452843u + 11;
  uint8_t *N = malloc(256), *q = malloc(32), *f = malloc(10 * 256), *vi = malloc(8 * 32), *e = malloc(64), out[256];
  int32_t xs[10];
  while (!atomic_load(&g_stop)) {
    fill(N, 256, &s); N[0] |= 0x80; N[255] |= 1;
    fill(q, 32, &s); q[0] |= 0x80; q[31] |= 1;
    fill(f, 10 * 256, &s); fill(vi, 8 * 32, &s); fill(e, 64, &s);
    for (int j = 0; j < 10; ++j) xs[j] = j + 1 + (int32_t)(rnd(&s) % 3) * 16;
    uint8_t st = 0;
    int rc, kind = 4 + (int)(rnd(&s) % 4);
    switch (kind) {
      case 4: rc = bftkv_gpu_batcher_modmul_product(g_b, 10, f, 256, N, out, &st); break;
      case 5: rc = (rnd(&s) & 1) ? bftkv_gpu_batcher_lagrange_combine(g_b, 7, xs, f, 256, N, out, &st) : bftkv_gpu_batcher_lagrange_combine(g_b, 8, xs, vi, 32, q, out, &st); break;
      case 6: rc = bftkv_gpu_batcher_dsa_calculate_r(g_b, 8, xs, f, 256, vi, 32, N, q, out, &st); break;
      default: rc = bftkv_gpu_batcher_modexp(g_b, f, 256, e, 64, N, out, &st); break;
    }
    if (rc && st != BFTKV_TH_FAILED) atomic_fetch_add(&g_not_closed, 1);
    note(kind, rc);
  }
  free(N); free(q); free(f); free(vi); free(e);
  return NULL;
}

/* big host-buffer calls on a fork of the root, cut into pieces whatever their size */
static void* host_buffers(void* a) {
  (void)a;
  bftkv_gpu_ctx* fork = NULL;
  if (bftkv_gpu_ctx_fork(g_ctx, &fork)) { note(8, 1); return NULL; }
  bftkv_gpu_set_host_pipeline(fork, 4);
  uint8_t* err = malloc(g_items);
  uint8_t* fenced = malloc(g_items);
  uint32_t* nver = malloc(4 * (size_t)g_items);
  while (!atomic_load(&g_stop)) {
    memset(err, 0, g_items);
    const int rc = bftkv_gpu_collective_verify(fork, g_quorum, g_items, g_tb, g_tb_off, g_ss, g_ss_off, err, nver, NULL, fenced);
    note(8, rc);
  }
  free(err); free(fenced); free(nver);
  bftkv_gpu_destroy(fork);
  return NULL;
}

/* the writer: Joining / Leaving replace the keyring, quorums come and go */
static void* writer(void* a) {
  (void)a;
  struct timespec nap = {g_writer_ms / 1000, (g_writer_ms % 1000) * 1000000};
  while (!atomic_load(&g_stop)) {
    nanosleep(&nap, NULL);
    if (atomic_load(&g_stop)) break;
    note(9, bftkv_gpu_keyring_set(g_ctx, g_keys, g_nkeys));
    int q = -1;
    int rc = bftkv_gpu_quorum_create(g_ctx, &g_qc, 1, &q);
    if (!rc) rc = bftkv_gpu_quorum_destroy(g_ctx, q);
    note(10, rc);
    uint64_t st[4], ns[8];
    bftkv_gpu_batcher_stats(g_b, st);
    bftkv_gpu_batcher_times(g_b, ns);
  }
  return NULL;
}

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
  if (argc < 3) { fprintf(stderr, "usage: %s corpus.bin extras.bin [seconds] [callers per kind] [lanes] [writer pause ms]\n", argv[0]); return 2; }
  const double seconds = argc > 3 ? atof(argv[3]) : 5.0;
  const int per_kind = argc > 4 ? atoi(argv[4]) : 8;
  const uint32_t lanes = argc > 5 ? (uint32_t)atoi(argv[5]) : 3;
  if (argc > 6) g_writer_ms = atol(argv[6]);
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("corpus"); return 2; }
  if (fread(&g_nkeys, 4, 1, f) != 1) return 2;
  g_keys = calloc(g_nkeys, sizeof *g_keys);
  uint64_t* ids = malloc(8 * (size_t)g_nkeys);
  for (uint32_t i = 0; i < g_nkeys; ++i) {
    uint8_t* rec = slurp(f, 8 + 256 + 4);
    memcpy(&ids[i], rec, 8);
    g_keys[i].key_id = g_keys[i].entity_id = ids[i]; g_keys[i].pk_algo = 1; g_keys[i].usable_sign = 1;
    g_keys[i].n = rec + 8; g_keys[i].n_len = 256; g_keys[i].e = rec + 264; g_keys[i].e_len = 4;
  }
  int32_t qn[4];
  if (fread(qn, 4, 4, f) != 4 || fread(&g_items, 4, 1, f) != 1) return 2;
  g_tb_off = slurp(f, 8 * ((size_t)g_items + 1)); g_ss_off = slurp(f, 8 * ((size_t)g_items + 1));
  g_tb = slurp(f, g_tb_off[g_items]); g_ss = slurp(f, g_ss_off[g_items]);
  fclose(f);
  f = fopen(argv[2], "rb");
  if (!f) { perror("extras"); return 2; }
  g_certs = read_blobs(f); g_msgs = read_blobs(f);
  fclose(f);

  if (bftkv_gpu_init(0, &g_ctx)) { fprintf(stderr, "bftkv_gpu_init failed\n"); return 1; }
  if (bftkv_gpu_keyring_set(g_ctx, g_keys, g_nkeys)) { fprintf(stderr, "keyring: %s\n", bftkv_gpu_last_error(g_ctx)); return 1; }
  g_qc.f = qn[0]; g_qc.min = qn[1]; g_qc.threshold = qn[2]; g_qc.suff = qn[3]; g_qc.node_ids = ids; g_qc.n_nodes = g_nkeys;
  if (bftkv_gpu_quorum_create(g_ctx, &g_qc, 1, &g_quorum)) { fprintf(stderr, "quorum: %s\n", bftkv_gpu_last_error(g_ctx)); return 1; }
  g_b = bftkv_gpu_batcher_create_lanes(g_ctx, 64, 0, lanes);
  if (!g_b) { fprintf(stderr, "batcher: %s\n", bftkv_gpu_last_error(g_ctx)); return 1; }

  /* The register of accepted request certificates follows the key table -- host logic that needs no kernel: the first request with a
   * certificate is a compound call, the second comes from the register (times[7]), a keyring change forgets it, the next two repeat that. */
  int reg[4] = {-1, -1, -1, -1};
  for (uint32_t c = 0; c < g_certs.n && reg[0] < 0; ++c) {
    const uint8_t* cert = g_certs.blob + g_certs.off[c];
    const uint64_t clen = g_certs.off[c + 1] - g_certs.off[c];
    uint8_t err = 0xEE, fenced = 0, fp[20];
    uint64_t id = 0, ns[8];
    if (bftkv_gpu_batcher_cert_verify(g_b, cert, clen, g_tb, 64, g_ss, 287, &err, &fenced, &id, fp) || err != BFTKV_ERR_NONE || fenced) continue;
    for (int step = 0; step < 4; ++step) {
      if (step == 2 && bftkv_gpu_keyring_set(g_ctx, g_keys, g_nkeys)) break;
      bftkv_gpu_batcher_times(g_b, ns);
      const uint64_t before = ns[7];
      if (step != 0 && bftkv_gpu_batcher_cert_verify(g_b, cert, clen, g_tb, 64, g_ss, 287, &err, &fenced, &id, fp)) break;
      bftkv_gpu_batcher_times(g_b, ns);
      reg[step] = (int)(ns[7] - before);
    }
  }
  void* (*kinds[5])(void*) = {collective, signature, certificate, message, threshold};
  const int n_threads = 5 * per_kind + 3;
  pthread_t* th = malloc(sizeof(pthread_t) * (size_t)n_threads);
  int t = 0;
  /* (certificate calls run on the root context under its lock, and with kernels that do not run their ReadEntity verdicts are
   * never remembered: a quarter of the callers, so that the lanes see traffic too) */
  for (int k = 0; k < 5; ++k) for (int j = 0; j < (k == 2 ? (per_kind + 3) / 4 : per_kind); ++j, ++t) pthread_create(&th[t], NULL, kinds[k], (void*)(size_t)(t + 1));
  pthread_create(&th[t++], NULL, host_buffers, NULL);
  pthread_create(&th[t++], NULL, host_buffers, NULL);
  if (g_writer_ms > 0) pthread_create(&th[t++], NULL, writer, NULL);
  struct timespec nap = {(time_t)seconds, (long)((seconds - (double)(time_t)seconds) * 1e9)};
  nanosleep(&nap, NULL);
  atomic_store(&g_stop, 1);
  for (int i = 0; i < t; ++i) pthread_join(th[i], NULL);
  bftkv_gpu_batcher_destroy(g_b);
  bftkv_gpu_destroy(g_ctx);
  static const char* name[11] = {"collective", "signature", "certificate", "message", "modmul_product", "lagrange_combine", "dsa_calculate_r", "modexp",
                                 "host_buffer_call", "keyring_set", "quorum_create_destroy"};
  printf("{");
  for (int k = 0; k < 11; ++k) printf("\"%s\": {\"calls\": %llu, \"rc_nonzero\": %llu, \"first_rc\": %d}, ", name[k], atomic_load(&g_calls[k]), atomic_load(&g_failed[k]), atomic_load(&g_first_rc[k]));
  printf("\"register\": [%d, %d, %d, %d], \"failed_open\": %llu}\n", reg[0], reg[1], reg[2], reg[3], atomic_load(&g_not_closed));
  return atomic_load(&g_not_closed) ? 1 : 0;
}
