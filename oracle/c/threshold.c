/*
 * threshold.c -- plain-C restatement of the reference's threshold share-combine arithmetic (BASELINE config 5) on
 * OpenSSL bignums, one operation after the other per thread, threads over operations.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): bench.py's cpu_baseline leg for --config 5 and the tests that
 * cross-check it against oracle/threshold.py.  It follows the Go code step by step (same operation order, Euclidean
 * Mod, ModInverse of a possibly negative value), so it is also the reference's COST shape: every Lagrange coefficient is
 * recomputed per share with a fresh modular inverse, as sss.Lagrange does.
 *
 * Follows /root/reference:
 *   sss.Lagrange                         crypto/sss/sss.go:94-107
 *   SSSProcess.calculateSecret           crypto/sss/sss.go:81-92
 *   calculateS                           crypto/threshold/dsa/dsa_core.go:389-403
 *   dsaGroupOperations.CalculateR        crypto/threshold/dsa/dsa.go:33-52
 *   calculateSignature                   crypto/threshold/rsa/rsa.go:318-329
 * Pinned by tests/golden/threshold_kat.json through tests/test_oracle_c.py (same vectors as oracle/threshold.py).
 */
#include <openssl/bn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* sss.Lagrange(x, results, m): a = prod res, b = prod (res - x) over res != x; a * b^-1 mod m.
 * big.Int.ModInverse reduces a negative b first; returns 0 on "no inverse" (Go returns nil and the caller would panic). */
static int lagrange(int32_t x, const int32_t* xs, int k, const BIGNUM* m, BIGNUM* out, BN_CTX* ctx) {
  BN_CTX_start(ctx);
  BIGNUM* a = BN_CTX_get(ctx);
  BIGNUM* b = BN_CTX_get(ctx);
  BIGNUM* t = BN_CTX_get(ctx);
  int ok = 1;
  BN_one(a);
  BN_one(b);
  for (int j = 0; j < k; ++j) {
    if (xs[j] == x) continue;
    BN_set_word(t, (BN_ULONG)(xs[j] < 0 ? -(int64_t)xs[j] : xs[j]));
    BN_set_negative(t, xs[j] < 0);
    BN_mul(a, a, t, ctx);
    int64_t d = (int64_t)xs[j] - (int64_t)x;
    BN_set_word(t, (BN_ULONG)(d < 0 ? -d : d));
    BN_set_negative(t, d < 0);
    BN_mul(b, b, t, ctx);
  }
  BN_nnmod(b, b, m, ctx);
  if (!BN_mod_inverse(b, b, m, ctx)) ok = 0;
  else {
    BN_mul(a, a, b, ctx);
    BN_nnmod(out, a, m, ctx);
  }
  BN_CTX_end(ctx);
  return ok;
}

typedef struct {
  int kind;              /* 0 calculateSignature, 1 calculateSecret / calculateS (sum l_j y_j mod m), 2 CalculateR */
  uint32_t lo, hi, k, nbytes, vbytes;
  const int32_t* xs;     /* [n][k] */
  const uint8_t* ys;     /* [n][k][nbytes] big-endian */
  const uint8_t* vs;     /* kind 2: [n][k][vbytes] */
  const BIGNUM *m, *q;   /* kind 0/1: modulus m; kind 2: p = m, q */
  uint8_t* out;          /* [n][obytes] */
  uint32_t obytes;
  uint8_t* status;       /* [n] 0 ok, 1 no inverse */
} tjob;

static void* tworker(void* arg) {
  tjob* j = (tjob*)arg;
  BN_CTX* ctx = BN_CTX_new();
  BIGNUM *acc = BN_new(), *y = BN_new(), *l = BN_new(), *t = BN_new(), *v = BN_new();
  BN_MONT_CTX* mont = NULL;
  if (j->kind == 2) { mont = BN_MONT_CTX_new(); BN_MONT_CTX_set(mont, j->m, ctx); }
  for (uint32_t i = j->lo; i < j->hi; ++i) {
    const int32_t* xs = j->xs ? j->xs + (size_t)i * j->k : NULL;
    int bad = 0;
    if (j->kind == 0) {
      BN_one(acc);                                            /* s = 1; s = s * psig mod N per leaf */
      for (uint32_t s = 0; s < j->k; ++s) {
        BN_bin2bn(j->ys + ((size_t)i * j->k + s) * j->nbytes, (int)j->nbytes, y);
        BN_mod_mul(acc, acc, y, j->m, ctx);
      }
    } else if (j->kind == 1) {
      BN_zero(acc);
      for (uint32_t s = 0; s < j->k; ++s) {
        if (!lagrange(xs[s], xs, (int)j->k, j->m, l, ctx)) { bad = 1; break; }
        BN_bin2bn(j->ys + ((size_t)i * j->k + s) * j->nbytes, (int)j->nbytes, y);
        BN_mul(t, l, y, ctx);
        BN_nnmod(t, t, j->m, ctx);
        BN_mod_add(acc, acc, t, j->m, ctx);
      }
    } else {
      BN_one(acc);
      BN_zero(v);
      for (uint32_t s = 0; s < j->k; ++s) {
        if (!lagrange(xs[s], xs, (int)j->k, j->q, l, ctx)) { bad = 1; break; }
        BN_bin2bn(j->ys + ((size_t)i * j->k + s) * j->nbytes, (int)j->nbytes, y);
        BN_nnmod(y, y, j->m, ctx);
        BN_mod_exp_mont(t, y, l, j->m, ctx, mont);             /* t = Ri^l mod p */
        BN_mod_mul(acc, acc, t, j->m, ctx);
        BN_bin2bn(j->vs + ((size_t)i * j->k + s) * j->vbytes, (int)j->vbytes, y);
        BN_mod_mul(t, y, l, j->q, ctx);
        BN_mod_add(v, v, t, j->q, ctx);
      }
      if (!bad) {
        if (!BN_mod_inverse(v, v, j->q, ctx)) bad = 1;         /* v.ModInverse(v, q) */
        else {
          BN_mod_exp_mont(acc, acc, v, j->m, ctx, mont);       /* r = r^v mod p */
          BN_nnmod(acc, acc, j->q, ctx);                       /* mod q */
        }
      }
    }
    uint8_t* o = j->out + (size_t)i * j->obytes;
    memset(o, 0, j->obytes);
    if (j->status) j->status[i] = (uint8_t)bad;
    if (!bad) BN_bn2binpad(acc, o, (int)j->obytes);
  }
  if (mont) BN_MONT_CTX_free(mont);
  BN_free(acc); BN_free(y); BN_free(l); BN_free(t); BN_free(v);
  BN_CTX_free(ctx);
  return NULL;
}

static void run_jobs(tjob proto, uint32_t n, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if ((uint32_t)n_threads > n) n_threads = n ? (int)n : 1;
  pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof *th);
  tjob* jobs = (tjob*)calloc((size_t)n_threads, sizeof *jobs);
  for (int t = 0; t < n_threads; ++t) {
    jobs[t] = proto;
    jobs[t].lo = (uint32_t)((uint64_t)n * t / n_threads);
    jobs[t].hi = (uint32_t)((uint64_t)n * (t + 1) / n_threads);
    pthread_create(&th[t], NULL, tworker, &jobs[t]);
  }
  for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
}

/* calculateSignature: out[i] = prod_j factors[i][j] mod N */
void oracle_rsa_combine(uint32_t n, uint32_t k, const uint8_t* factors, uint32_t nbytes, const uint8_t* mod, uint8_t* out, int n_threads) {
  BIGNUM* m = BN_bin2bn(mod, (int)nbytes, NULL);
  tjob p;
  memset(&p, 0, sizeof p);
  p.kind = 0; p.k = k; p.nbytes = nbytes; p.ys = factors; p.m = m; p.out = out; p.obytes = nbytes;
  run_jobs(p, n, n_threads);
  BN_free(m);
}

/* calculateSecret / calculateS: out[i] = sum_j Lagrange(xs[i][j]) * ys[i][j] mod m */
void oracle_lagrange_combine(uint32_t n, uint32_t k, const int32_t* xs, const uint8_t* ys, uint32_t nbytes, const uint8_t* mod,
                             uint8_t* out, uint8_t* status, int n_threads) {
  BIGNUM* m = BN_bin2bn(mod, (int)nbytes, NULL);
  tjob p;
  memset(&p, 0, sizeof p);
  p.kind = 1; p.k = k; p.nbytes = nbytes; p.xs = xs; p.ys = ys; p.m = m; p.out = out; p.obytes = nbytes; p.status = status;
  run_jobs(p, n, n_threads);
  BN_free(m);
}

/* CalculateR: out[i] = (prod_j Ri^lj mod p)^((sum_j Vi*lj)^-1 mod q) mod p mod q */
void oracle_dsa_calculate_r(uint32_t n, uint32_t k, const int32_t* xs, const uint8_t* ri, uint32_t pbytes, const uint8_t* vi,
                            uint32_t qbytes, const uint8_t* p_be, const uint8_t* q_be, uint8_t* out, uint8_t* status, int n_threads) {
  BIGNUM* pm = BN_bin2bn(p_be, (int)pbytes, NULL);
  BIGNUM* qm = BN_bin2bn(q_be, (int)qbytes, NULL);
  tjob p;
  memset(&p, 0, sizeof p);
  p.kind = 2; p.k = k; p.nbytes = pbytes; p.vbytes = qbytes; p.xs = xs; p.ys = ri; p.vs = vi; p.m = pm; p.q = qm; p.out = out;
  p.obytes = qbytes; p.status = status;
  run_jobs(p, n, n_threads);
  BN_free(pm);
  BN_free(qm);
}
