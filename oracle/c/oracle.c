/*
 * oracle.c -- plain-C restatement of the reference's CPU path for the quorum-verification hot
 * path, in the reference's ALGORITHMIC SHAPE: per signature packet a fresh hash of the WHOLE signed
 * payload (crypto/pgp/crypto_pgp.go:490 -> openpgp.CheckDetachedSignature), one RSA/DSA public-key
 * operation, append the signer, then quorum.IsSufficient with the O(k*n) multiset intersection
 * (quorum/wotqs/wotqs.go:168-175, 195-206), early exit on sufficiency (crypto_pgp.go:493-496).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ to cross-check the Python oracle
 * and by bench.py's cpu_baseline leg ("kind": "port").  Big integers and hashes come from OpenSSL
 * (libcrypto) -- hand-tuned assembly, i.e. a FASTER baseline than the reference's Go math/big.
 * PARITY UNPINNED against the reference itself (no Go toolchain, x/crypto not vendored); pinned
 * against the Python oracle, which is pinned against GnuPG (tests/golden).
 *
 * Follows: SURVEY.md Appendix B.1-B.5 (x/crypto openpgp packet.Read, Signature.parse,
 * CheckDetachedSignature, PublicKey.VerifySignature; Go crypto/rsa.VerifyPKCS1v15, crypto/dsa.Verify).
 */
#include <openssl/bn.h>
#include <openssl/md5.h>
#include <openssl/ripemd.h>
#include <openssl/sha.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ST_OK = 0, ST_UNKNOWN_ISSUER, ST_PARSE_ERROR, ST_NOT_SIGNATURE, ST_NO_ISSUER, ST_HASH_UNSUPPORTED,
       ST_HASH_TAG, ST_ALGO_MISMATCH, ST_BAD_SIG, ST_KEY_CANNOT_SIGN, ST_UNSUPPORTED };

typedef struct {
  uint64_t key_id, entity_id;
  int pk_algo, usable_sign;
  BIGNUM *n, *e;          /* RSA */
  BN_MONT_CTX* mont;      /* precomputed once per key (OpenSSL would otherwise rebuild it per call) */
  BIGNUM *p, *q, *g, *y;  /* DSA */
} okey;

typedef struct {
  int f, min, threshold, suff;
  uint64_t* nodes;
  int n_nodes;
} oqc;

typedef struct {
  okey* keys;
  int n_keys;
  oqc* qcs;
  int n_qcs;
} oracle;

void* oracle_new(void) { return calloc(1, sizeof(oracle)); }

void oracle_free(void* h) {
  oracle* o = (oracle*)h;
  if (!o) return;
  for (int i = 0; i < o->n_keys; ++i) {
    BN_free(o->keys[i].n); BN_free(o->keys[i].e); BN_free(o->keys[i].p); BN_MONT_CTX_free(o->keys[i].mont);
    BN_free(o->keys[i].q); BN_free(o->keys[i].g); BN_free(o->keys[i].y);
  }
  free(o->keys);
  for (int i = 0; i < o->n_qcs; ++i) free(o->qcs[i].nodes);
  free(o->qcs);
  free(o);
}

/* keys in keyring order (secring first, crypto_pgp.go:195-197).  RSA: a=n, b=e.  DSA: a=p, b=q, g, y. */
void oracle_add_key(void* h, uint64_t key_id, uint64_t entity_id, int pk_algo, int usable_sign, const uint8_t* a, int alen,
                    const uint8_t* b, int blen, const uint8_t* g, int glen, const uint8_t* y, int ylen) {
  oracle* o = (oracle*)h;
  o->keys = (okey*)realloc(o->keys, sizeof(okey) * (o->n_keys + 1));
  okey* k = &o->keys[o->n_keys++];
  memset(k, 0, sizeof *k);
  k->key_id = key_id; k->entity_id = entity_id; k->pk_algo = pk_algo; k->usable_sign = usable_sign;
  if (pk_algo == 17) {
    k->p = BN_bin2bn(a, alen, NULL); k->q = BN_bin2bn(b, blen, NULL);
    k->g = BN_bin2bn(g, glen, NULL); k->y = BN_bin2bn(y, ylen, NULL);
  } else {
    k->n = BN_bin2bn(a, alen, NULL); k->e = BN_bin2bn(b, blen, NULL);
    if (BN_is_odd(k->n)) {
      BN_CTX* ctx = BN_CTX_new();
      k->mont = BN_MONT_CTX_new();
      BN_MONT_CTX_set(k->mont, k->n, ctx);
      BN_CTX_free(ctx);
    }
  }
}

void oracle_set_quorum(void* h, int n_qcs, const int32_t* f, const int32_t* mn, const int32_t* thr, const int32_t* suff,
                       const uint64_t* ids, const int32_t* counts) {
  oracle* o = (oracle*)h;
  for (int i = 0; i < o->n_qcs; ++i) free(o->qcs[i].nodes);
  free(o->qcs);
  o->qcs = (oqc*)calloc(n_qcs ? n_qcs : 1, sizeof(oqc));
  o->n_qcs = n_qcs;
  for (int i = 0; i < n_qcs; ++i) {
    o->qcs[i].f = f[i]; o->qcs[i].min = mn[i]; o->qcs[i].threshold = thr[i]; o->qcs[i].suff = suff[i];
    o->qcs[i].n_nodes = counts[i];
    o->qcs[i].nodes = (uint64_t*)malloc(sizeof(uint64_t) * (counts[i] ? counts[i] : 1));
    memcpy(o->qcs[i].nodes, ids, sizeof(uint64_t) * counts[i]);
    ids += counts[i];
  }
}

/* wotqs.go:195-206 + :168-175 -- deliberately the reference's nested loops */
static int is_sufficient(const oracle* o, const uint64_t* nodes, int n) {
  for (int c = 0; c < o->n_qcs; ++c) {
    const oqc* qc = &o->qcs[c];
    int cnt = 0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < qc->n_nodes; ++j)
        if (nodes[i] == qc->nodes[j]) { ++cnt; break; }
    if (qc->suff > 0 && cnt >= qc->suff) return 1;
  }
  return 0;
}

/* ---- OpenPGP parsing (Appendix B.1, B.2) ---------------------------------------------------- */
typedef struct {
  int sig_type, pk_algo, hash_id;
  const uint8_t* prefix; int prefix_len;      /* first 6+hl body bytes */
  uint8_t tag[2];
  int have_issuer; uint64_t issuer; int have_ctime; int have_embedded; int v3;
  const uint8_t* mpi[2]; int mpi_len[2];
} psig;

static int parse_body(const uint8_t* b, int n, psig* s, int depth);

static int parse_subpackets(const uint8_t* a, int len, int hashed, psig* s, int depth) {
  int p = 0;
  while (p < len) {
    int ln;
    int b = a[p];
    if (b < 192) { ln = b; p += 1; }
    else if (b < 255) { if (p + 2 > len) return 0; ln = ((b - 192) << 8) + a[p + 1] + 192; p += 2; }
    else { if (p + 5 > len) return 0; ln = (int)(((uint32_t)a[p + 1] << 24) | (a[p + 2] << 16) | (a[p + 3] << 8) | a[p + 4]); p += 5; }
    if (ln < 0 || ln > len - p) return 0;
    if (ln == 0) return 0;
    int typ = a[p] & 0x7F, critical = a[p] & 0x80;
    const uint8_t* body = a + p + 1;
    int bl = ln - 1;
    p += ln;
    switch (typ) {
      case 2: if (!hashed) return 0; /* "signature creation time in non-hashed area" */ if (bl != 4) return 0; s->have_ctime = 1; break;
      case 3: case 9: if (!hashed) break; if (bl != 4) return 0; break;
      case 11: case 21: case 22: case 30: break;
      case 16:
        if (bl != 8) return 0;
        s->issuer = 0;
        for (int i = 0; i < 8; ++i) s->issuer = (s->issuer << 8) | body[i];
        s->have_issuer = 1;
        break;
      case 25: if (!hashed) break; if (bl != 1) return 0; break;
      case 27: case 29: if (!hashed) break; if (bl == 0) return 0; break;
      case 32: {
        /* embedded signature: either area, at most one, must parse and be a primary-key binding (0x19) */
        if (s->have_embedded) return 0;
        s->have_embedded = 1;
        if (depth >= 2) return 0;   /* nesting bound shared with the kernels (deeper: fenced, tests skip such items) */
        psig tmp;
        if (!parse_body(body, bl, &tmp, depth + 1)) return 0;
        if (tmp.sig_type != 0x19) return 0;
        break;
      }
      default: if (critical) return 0;
    }
  }
  return 1;
}

static int parse_body(const uint8_t* b, int n, psig* s, int depth) {
  memset(s, 0, sizeof *s);
  if (n < 1 || b[0] != 4 || n < 6) return 0;
  s->sig_type = b[1]; s->pk_algo = b[2]; s->hash_id = b[3];
  if (!(s->pk_algo == 1 || s->pk_algo == 3 || s->pk_algo == 17 || s->pk_algo == 19)) return 0;
  int h = s->hash_id;
  if (!(h == 1 || h == 2 || h == 3 || (h >= 8 && h <= 11))) return 0;
  int hl = (b[4] << 8) | b[5];
  if (6 + hl > n) return 0;
  s->prefix = b; s->prefix_len = 6 + hl;
  if (!parse_subpackets(b + 6, hl, 1, s, depth)) return 0;
  if (!s->have_ctime) return 0;
  int p = 6 + hl;
  if (p + 2 > n) return 0;
  int ul = (b[p] << 8) | b[p + 1];
  p += 2;
  if (p + ul > n) return 0;
  if (!parse_subpackets(b + p, ul, 0, s, depth)) return 0;
  p += ul;
  if (p + 2 > n) return 0;
  s->tag[0] = b[p]; s->tag[1] = b[p + 1];
  p += 2;
  int nm = (s->pk_algo == 1 || s->pk_algo == 3) ? 1 : 2;
  for (int i = 0; i < nm; ++i) {
    if (p + 2 > n) return 0;
    int bits = (b[p] << 8) | b[p + 1];
    int nb = (bits + 7) / 8;
    p += 2;
    if (p + nb > n) return 0;
    s->mpi[i] = b + p; s->mpi_len[i] = nb;
    p += nb;
  }
  return 1;
}

/* SignatureV3.parse: version 2/3, "5", type, creation time, issuer, algorithms, tag, MPIs; hashed material = body[2..7) */
static int parse_body_v3(const uint8_t* b, int n, psig* s) {
  memset(s, 0, sizeof *s);
  if (n < 1 || b[0] < 2 || b[0] > 3 || n < 19 || b[1] != 5) return 0;
  s->sig_type = b[2]; s->pk_algo = b[15]; s->hash_id = b[16];
  s->have_ctime = 1; s->have_issuer = 1; s->issuer = 0;
  for (int i = 0; i < 8; ++i) s->issuer = (s->issuer << 8) | b[7 + i];
  if (!(s->pk_algo == 1 || s->pk_algo == 3 || s->pk_algo == 17)) return 0;
  int h = s->hash_id;
  if (!(h == 1 || h == 2 || h == 3 || (h >= 8 && h <= 11))) return 0;
  s->prefix = b + 2; s->prefix_len = 5; s->v3 = 1;
  s->tag[0] = b[17]; s->tag[1] = b[18];
  int p = 19, nm = (s->pk_algo == 17) ? 2 : 1;
  for (int i = 0; i < nm; ++i) {
    if (p + 2 > n) return 0;
    int bits = (b[p] << 8) | b[p + 1];
    int nb = (bits + 7) / 8;
    p += 2;
    if (p + nb > n) return 0;
    s->mpi[i] = b + p; s->mpi_len[i] = nb;
    p += nb;
  }
  return 1;
}

/* Low-level SHA contexts (no EVP): EVP_MD objects are reference-counted with atomics on ONE shared
 * cache line, which serialises hundreds of worker threads. */
typedef struct {
  int id;
  union { SHA_CTX s1; SHA256_CTX s256; SHA512_CTX s512; MD5_CTX m5; RIPEMD160_CTX r160; } u;
} hctx;
/* MD5 / RIPEMD-160 are "available" in the reference only when its binary links them (oracle/openpgp.py HASH_POLICY):
 * bit 0 MD5, bit 1 RIPEMD-160; default 0 = refused like an unsupported hash */
static int g_weak_hashes = 0;
void oracle_set_weak_hashes(int mask) { g_weak_hashes = mask; }
static int h_init(hctx* h, int hash_id) {
  h->id = hash_id;
  switch (hash_id) {
    case 1: if (!(g_weak_hashes & 1)) return 0; MD5_Init(&h->u.m5); return 1;
    case 3: if (!(g_weak_hashes & 2)) return 0; RIPEMD160_Init(&h->u.r160); return 1;
    case 2: SHA1_Init(&h->u.s1); return 1;
    case 8: SHA256_Init(&h->u.s256); return 1;
    case 9: SHA384_Init(&h->u.s512); return 1;
    case 10: SHA512_Init(&h->u.s512); return 1;
    case 11: SHA224_Init(&h->u.s256); return 1;
    default: return 0;
  }
}
static void h_update(hctx* h, const void* p, size_t n) {
  switch (h->id) {
    case 1: MD5_Update(&h->u.m5, p, n); break;
    case 3: RIPEMD160_Update(&h->u.r160, p, n); break;
    case 2: SHA1_Update(&h->u.s1, p, n); break;
    case 8: SHA256_Update(&h->u.s256, p, n); break;
    case 9: SHA384_Update(&h->u.s512, p, n); break;
    case 10: SHA512_Update(&h->u.s512, p, n); break;
    case 11: SHA224_Update(&h->u.s256, p, n); break;
  }
}
/* digest of a COPY of the running hash (the running hash keeps accumulating, B.3) */
static unsigned h_peek(const hctx* h, uint8_t* out) {
  hctx c = *h;
  switch (c.id) {
    case 1: MD5_Final(out, &c.u.m5); return 16;
    case 3: RIPEMD160_Final(out, &c.u.r160); return 20;
    case 2: SHA1_Final(out, &c.u.s1); return 20;
    case 8: SHA256_Final(out, &c.u.s256); return 32;
    case 9: SHA384_Final(out, &c.u.s512); return 48;
    case 10: SHA512_Final(out, &c.u.s512); return 64;
    case 11: SHA224_Final(out, &c.u.s256); return 28;
  }
  return 0;
}

static const uint8_t PFX_SHA1[] = {0x30, 0x21, 0x30, 0x09, 0x06, 0x05, 0x2b, 0x0e, 0x03, 0x02, 0x1a, 0x05, 0x00, 0x04, 0x14};
static const uint8_t PFX_SHA224[] = {0x30, 0x2d, 0x30, 0x0d, 0x06, 0x09, 0x60, 0x86, 0x48, 0x01, 0x65, 0x03, 0x04, 0x02, 0x04, 0x05, 0x00, 0x04, 0x1c};
static const uint8_t PFX_SHA256[] = {0x30, 0x31, 0x30, 0x0d, 0x06, 0x09, 0x60, 0x86, 0x48, 0x01, 0x65, 0x03, 0x04, 0x02, 0x01, 0x05, 0x00, 0x04, 0x20};
static const uint8_t PFX_SHA384[] = {0x30, 0x41, 0x30, 0x0d, 0x06, 0x09, 0x60, 0x86, 0x48, 0x01, 0x65, 0x03, 0x04, 0x02, 0x02, 0x05, 0x00, 0x04, 0x30};
static const uint8_t PFX_SHA512[] = {0x30, 0x51, 0x30, 0x0d, 0x06, 0x09, 0x60, 0x86, 0x48, 0x01, 0x65, 0x03, 0x04, 0x02, 0x03, 0x05, 0x00, 0x04, 0x40};

static const uint8_t PFX_MD5[] = {0x30, 0x20, 0x30, 0x0c, 0x06, 0x08, 0x2a, 0x86, 0x48, 0x86, 0xf7, 0x0d, 0x02, 0x05, 0x05, 0x00, 0x04, 0x10};
/* Go's entry (crypto/threshold/rsa/rsa.go:353): ISO/IEC 10118-3 identifier, not gpg's TeleTrusT one */
static const uint8_t PFX_RMD160[] = {0x30, 0x20, 0x30, 0x08, 0x06, 0x06, 0x28, 0xcf, 0x06, 0x03, 0x00, 0x31, 0x04, 0x14};
static const uint8_t* prefix_for(int hash_id, int* len) {
  switch (hash_id) {
    case 1: *len = sizeof PFX_MD5; return PFX_MD5;
    case 3: *len = sizeof PFX_RMD160; return PFX_RMD160;
    case 2: *len = sizeof PFX_SHA1; return PFX_SHA1;
    case 8: *len = sizeof PFX_SHA256; return PFX_SHA256;
    case 9: *len = sizeof PFX_SHA384; return PFX_SHA384;
    case 10: *len = sizeof PFX_SHA512; return PFX_SHA512;
    case 11: *len = sizeof PFX_SHA224; return PFX_SHA224;
    default: *len = 0; return NULL;
  }
}

/* B.4: Go 1.12/1.13 rsa.VerifyPKCS1v15 (no length check, no s<n check) */
static int rsa_verify(const okey* k, int hash_id, const uint8_t* digest, int dlen, const uint8_t* sig, int slen, BN_CTX* ctx) {
  int plen;
  const uint8_t* pfx = prefix_for(hash_id, &plen);
  int tlen = plen + dlen;
  int kb = (BN_num_bits(k->n) + 7) / 8;
  if (kb < tlen + 11) return 0;
  int ok = 0;
  BN_CTX_start(ctx);
  BIGNUM* c = BN_CTX_get(ctx);
  BIGNUM* m = BN_CTX_get(ctx);
  BN_bin2bn(sig, slen, c);
  if (BN_is_odd(k->n) && BN_cmp(c, k->n) < 0) {
    if (!BN_mod_exp_mont(m, c, k->e, k->n, ctx, k->mont)) goto done;
  } else {
    /* math/big.Exp reduces the base and accepts any modulus */
    BIGNUM* cr = BN_CTX_get(ctx);
    if (!BN_nnmod(cr, c, k->n, ctx)) goto done;
    if (!BN_mod_exp_simple(m, cr, k->e, k->n, ctx)) goto done;
  }
  {
    uint8_t em[1024], want[1024];
    if (kb > (int)sizeof em || BN_num_bytes(m) > kb) goto done;
    BN_bn2binpad(m, em, kb);
    want[0] = 0; want[1] = 1;
    memset(want + 2, 0xFF, kb - tlen - 3);
    want[kb - tlen - 1] = 0;
    memcpy(want + kb - tlen, pfx, plen);
    memcpy(want + kb - dlen, digest, dlen);
    ok = memcmp(em, want, kb) == 0;
  }
done:
  BN_CTX_end(ctx);
  return ok;
}

/* B.5: Go crypto/dsa.Verify */
static int dsa_verify(const okey* k, const uint8_t* digest, int dlen, const uint8_t* rb, int rlen, const uint8_t* sb, int slen,
                      BN_CTX* ctx) {
  int ok = 0;
  BN_CTX_start(ctx);
  BIGNUM *r = BN_CTX_get(ctx), *s = BN_CTX_get(ctx), *w = BN_CTX_get(ctx), *z = BN_CTX_get(ctx), *u1 = BN_CTX_get(ctx),
         *u2 = BN_CTX_get(ctx), *v1 = BN_CTX_get(ctx), *v2 = BN_CTX_get(ctx);
  BN_bin2bn(rb, rlen, r); BN_bin2bn(sb, slen, s);
  if (BN_is_zero(k->p)) goto done;
  if (BN_is_zero(r) || BN_cmp(r, k->q) >= 0) goto done;
  if (BN_is_zero(s) || BN_cmp(s, k->q) >= 0) goto done;
  int nq = BN_num_bits(k->q);
  if (nq & 7) goto done;
  if (!BN_mod_inverse(w, s, k->q, ctx)) goto done;
  if (dlen > nq / 8) dlen = nq / 8;
  BN_bin2bn(digest, dlen, z);
  BN_mod_mul(u1, z, w, k->q, ctx);
  BN_mod_mul(u2, r, w, k->q, ctx);
  BN_mod_exp(v1, k->g, u1, k->p, ctx);
  BN_mod_exp(v2, k->y, u2, k->p, ctx);
  BN_mod_mul(v1, v1, v2, k->p, ctx);
  BN_nnmod(v1, v1, k->q, ctx);
  ok = BN_cmp(v1, r) == 0;
done:
  BN_CTX_end(ctx);
  return ok;
}

static int known_tag(int tag) { return tag < 32 && ((0x00066BFEu >> tag) & 1u); }

/* ---- the signature stream as x/crypto READS it (oracle/openpgp.py B.1b is the commented twin of this) --------------------
 * Reader objects with Go's Read contract, err: 0 none, 1 io.EOF, 2 io.ErrUnexpectedEOF.  Only packets off the common shape
 * (definite length, body inside the stream, <= 4096 bytes) come through here. */
typedef struct { const uint8_t* d; uint64_t pos, end; } bstream;                       /* bytes.Reader */
static uint64_t bs_read(bstream* s, uint8_t* out, uint64_t k, int* err) {
  *err = 0;
  if (s->pos >= s->end) { *err = 1; return 0; }
  uint64_t n = s->end - s->pos < k ? s->end - s->pos : k;
  if (out) memcpy(out, s->d + s->pos, n);
  s->pos += n;
  return n;
}
static int bs_read_full(bstream* s, uint8_t* out, uint64_t k) {                        /* packet.readFull: any shortfall is an error */
  int err;
  uint64_t got = 0;
  while (got < k) { uint64_t n = bs_read(s, out + got, k - got, &err); got += n; if (err) return 2; }
  return 0;
}
/* packet.readLength */
static int read_length(bstream* s, uint64_t* len, int* partial) {
  uint8_t b[4];
  *partial = 0;
  if (bs_read_full(s, b, 1)) return 2;
  if (b[0] < 192) { *len = b[0]; return 0; }
  if (b[0] < 224) { const uint64_t hi = (uint64_t)(b[0] - 192) << 8; if (bs_read_full(s, b, 1)) return 2; *len = hi + b[0] + 192; return 0; }
  if (b[0] < 255) { *len = 1ull << (b[0] & 0x1F); *partial = 1; return 0; }
  if (bs_read_full(s, b, 4)) return 2;
  *len = ((uint64_t)b[0] << 24) | ((uint64_t)b[1] << 16) | ((uint64_t)b[2] << 8) | b[3];
  return 0;
}
/* spanReader (kind 0), partialLengthReader (1), the bare stream of an indeterminate-length packet (2) */
typedef struct { int kind; bstream* s; uint64_t rem; int partial; } body_rd;
static uint64_t body_read(body_rd* r, uint8_t* out, uint64_t k, int* err) {
  if (r->kind == 2) return bs_read(r->s, out, k, err);
  if (r->kind == 0) {
    if (r->rem == 0) { *err = 1; return 0; }
    const uint64_t n = bs_read(r->s, out, k < r->rem ? k : r->rem, err);
    r->rem -= n;
    if (r->rem > 0 && *err == 1) *err = 2;
    return n;
  }
  while (r->rem == 0) {
    if (!r->partial) { *err = 1; return 0; }
    if (read_length(r->s, &r->rem, &r->partial)) { *err = 2; return 0; }
  }
  const uint64_t want = k < r->rem ? k : r->rem;
  const uint64_t n = bs_read(r->s, out, want, err);
  r->rem -= n;
  if (n < want && *err == 1) *err = 2;
  return n;
}
static void consume_all(body_rd* r) {                                                  /* packet.consumeAll */
  int err;
  do { body_read(r, NULL, 1024, &err); } while (!err);
}
/* bufio.Reader, 4096 bytes: Peek(1) and Read */
typedef struct { body_rd* rd; uint8_t buf[4096]; uint32_t r, w; int err; } bufio_rd;
static int bufio_take_err(bufio_rd* b) { const int e = b->err; b->err = 0; return e; }
static int bufio_peek1(bufio_rd* b, uint8_t* v) {
  while (b->r == b->w && !b->err) {
    b->r = b->w = 0;
    b->w = (uint32_t)body_read(b->rd, b->buf, sizeof b->buf, &b->err);                 /* fill(): our readers never return (0, nil) */
  }
  if (b->r < b->w) { *v = b->buf[b->r]; return 0; }
  return bufio_take_err(b);
}
static uint64_t bufio_read(bufio_rd* b, uint8_t* out, uint64_t k, int* err) {
  *err = 0;
  if (b->r == b->w) {
    if (b->err) { *err = bufio_take_err(b); return 0; }
    if (k >= sizeof b->buf) {                                                          /* large read, empty buffer: no copy */
      const uint64_t n = body_read(b->rd, out, k, &b->err);
      *err = bufio_take_err(b);
      return n;
    }
    b->r = b->w = 0;
    const uint64_t n = body_read(b->rd, b->buf, sizeof b->buf, &b->err);
    if (n == 0) { *err = bufio_take_err(b); return 0; }
    b->w = (uint32_t)n;
  }
  uint64_t n = b->w - b->r < k ? b->w - b->r : k;
  if (out) memcpy(out, b->buf + b->r, n);
  b->r += (uint32_t)n;
  return n;
}
static int bufio_read_full(bufio_rd* b, uint64_t k) {                                  /* readFull(r, make([]byte, k)); data discarded */
  uint64_t got = 0;
  int err = 0;
  uint8_t* sink = k ? malloc(k) : NULL;       /* a real destination: the large-read path copies straight into it */
  while (got < k && !err) got += bufio_read(b, sink + got, k - got, &err);
  free(sink);
  return got < k;
}

/* One signature packet off the common shape.  `rd` is positioned on the first body byte.  The body is parsed from a linear copy
 * of what the readers can deliver; when it parses, the reads Signature.parse / SignatureV3.parse would issue are replayed through
 * a real bufio over the real readers, which leaves the shared stream where the reference leaves it; when it does not, the body
 * is drained.  Returns 1 when parsed (*lin is then the caller's to free: psig points into it). */
static int read_signature_general(body_rd* rd, psig* s, uint8_t** lin) {
  bstream probe_s = *rd->s;
  body_rd probe = *rd;
  probe.s = &probe_s;
  uint64_t cap = 4096, n = 0;
  uint8_t* buf = malloc(cap);
  for (;;) {
    int err;
    if (n == cap) { cap *= 2; buf = realloc(buf, cap); }
    n += body_read(&probe, buf + n, cap - n, &err);
    if (err) break;
  }
  *lin = buf;
  const int parsed = n > 0x7FFFFFFF ? 0 : ((n >= 1 && buf[0] < 4) ? parse_body_v3(buf, (int)n, s) : parse_body(buf, (int)n, s, 0));
  bufio_rd* b = calloc(1, sizeof *b);
  b->rd = rd;
  uint8_t ver;
  if (bufio_peek1(b, &ver)) { free(b); free(buf); *lin = NULL; return 0; }             /* empty body: Peek fails, nothing to drain */
  if (!parsed) {
    int err;
    do { bufio_read(b, NULL, 1024, &err); } while (!err);                              /* consumeAll(contents) -- contents is the bufio */
    free(b); free(buf); *lin = NULL;
    return 0;
  }
  int bad = 0;
  if (s->v3) {
    const uint64_t reads[6] = {1, 1, 5, 8, 2, 2};
    for (int i = 0; i < 6; ++i) bad |= bufio_read_full(b, reads[i]);
  } else {
    const uint64_t hl = (uint64_t)s->prefix_len - 6;
    const uint64_t ul = (uint64_t)(s->mpi[0] - buf) - 6 - hl - 2 - 2 - 2;
    const uint64_t reads[6] = {1, 5, hl, 2, ul, 2};
    for (int i = 0; i < 6; ++i) bad |= bufio_read_full(b, reads[i]);
  }
  for (int i = 0; i < 2 && s->mpi[i]; ++i) { bad |= bufio_read_full(b, 2); bad |= bufio_read_full(b, (uint64_t)s->mpi_len[i]); }
  free(b);
  if (bad) { fprintf(stderr, "oracle: replay of a parsed signature ran dry\n"); abort(); }
  return 1;
}

/* One openpgp.CheckDetachedSignature(keyring, signed, sigstream@pos) call (B.3).
 * Returns the call's status; *signer = entity id on ST_OK; *pos advanced; per-packet statuses
 * appended to trace (if non-NULL). */
static int check_detached(const oracle* o, const uint8_t* tbs, uint64_t tbs_len, const uint8_t* sd, uint64_t end, uint64_t* pos,
                          uint64_t* signer, uint8_t* trace, int* ntrace, int cap, BN_CTX* ctx, uint64_t* n_pk_ops) {
#define TRACE(st) do { if (trace && *ntrace < cap) trace[(*ntrace)++] = (uint8_t)(st); } while (0)
  uint8_t* lin = NULL;       /* linear copy of a signature body that did not lie in the stream in one piece */
#define RETURN(st) do { free(lin); return (st); } while (0)
  for (;;) {
    free(lin); lin = NULL;
    bstream bs = {sd, *pos, end};
    uint8_t b0;
    if (bs.pos >= end) RETURN(ST_UNKNOWN_ISSUER);   /* io.EOF => ErrUnknownIssuer */
    b0 = sd[bs.pos++];
    if (!(b0 & 0x80)) { *pos = bs.pos; TRACE(ST_PARSE_ERROR); RETURN(ST_PARSE_ERROR); }
    int tag;
    body_rd rd = {0, &bs, 0, 0};
    if (!(b0 & 0x40)) {
      tag = (b0 & 0x3F) >> 2;
      const int lt = b0 & 3;
      if (lt == 3) rd.kind = 2;
      else {
        uint8_t lb[4];
        if (bs_read_full(&bs, lb, 1u << lt)) { *pos = bs.pos; TRACE(ST_PARSE_ERROR); RETURN(ST_PARSE_ERROR); }
        for (int i = 0; i < (1 << lt); ++i) rd.rem = (rd.rem << 8) | lb[i];
      }
    } else {
      tag = b0 & 0x3F;
      if (read_length(&bs, &rd.rem, &rd.partial)) { *pos = bs.pos; TRACE(ST_PARSE_ERROR); RETURN(ST_PARSE_ERROR); }
      if (rd.partial) rd.kind = 1;
    }
    if (tag != 2) {
      /* unknown type: UnknownPacketTypeError, body drained, Next goes on.  Known type: parsed by code this restatement does not
       * follow (whole body taken; the verifier fences the types whose parser can stop early) */
      consume_all(&rd);
      *pos = bs.pos;
      if (known_tag(tag)) { TRACE(ST_NOT_SIGNATURE); RETURN(ST_NOT_SIGNATURE); }
      continue;
    }
    psig s;
    int parsed;
    if (rd.kind == 0 && rd.rem <= 4096 && rd.rem <= end - bs.pos) {
      /* the common shape: the body lies in the stream in one piece and bufio's first fetch takes all of it, so the reader
       * stands behind the packet whatever the parse says */
      const uint64_t start = bs.pos, ln = rd.rem;
      *pos = start + ln;
      /* packet.Read peeks the version: < 4 => SignatureV3, else Signature */
      parsed = (ln >= 1 && sd[start] < 4) ? parse_body_v3(sd + start, (int)ln, &s) : parse_body(sd + start, (int)ln, &s, 0);
    } else {
      parsed = read_signature_general(&rd, &s, &lin);
      *pos = bs.pos;
    }
    if (!parsed) { TRACE(ST_PARSE_ERROR); RETURN(ST_PARSE_ERROR); }
    if (!s.have_issuer) { TRACE(ST_NO_ISSUER); RETURN(ST_NO_ISSUER); }
    /* KeysByIdUsage(issuer, KeyFlagSign) */
    int found = 0;
    for (int i = 0; i < o->n_keys; ++i) if (o->keys[i].key_id == s.issuer && o->keys[i].usable_sign) { found = 1; break; }
    if (!found) { TRACE(ST_UNKNOWN_ISSUER); continue; }
    hctx hc;
    if ((s.sig_type != 0 && s.sig_type != 1) || !h_init(&hc, s.hash_id)) { TRACE(ST_HASH_UNSUPPORTED); RETURN(ST_HASH_UNSUPPORTED); }
    /* the reference hashes the WHOLE payload again for every signature packet; text-mode (0x01) signatures through
     * openpgp.NewCanonicalTextHash: a '\n' that does not follow a '\r' becomes "\r\n", the byte after a '\r' passes
     * unchanged (the hash suffix below goes into the raw hash) */
    if (s.sig_type == 0) h_update(&hc, tbs, tbs_len);
    else {
      int cs = 0;
      size_t start = 0;
      for (size_t i = 0; i < (size_t)tbs_len; ++i) {
        const uint8_t ch = tbs[i];
        if (cs == 0) {
          if (ch == '\r') cs = 1;
          else if (ch == '\n') { h_update(&hc, tbs + start, i - start); h_update(&hc, "\r\n", 2); start = i + 1; }
        } else cs = 0;
      }
      h_update(&hc, tbs + start, (size_t)tbs_len - start);
    }
    int st = ST_BAD_SIG;
    for (int i = 0; i < o->n_keys; ++i) {
      const okey* k = &o->keys[i];
      if (k->key_id != s.issuer || !k->usable_sign) continue;
      /* VerifySignature appends the suffix to the shared hash on every candidate */
      if (k->pk_algo == 2 || k->pk_algo == 16) { st = ST_KEY_CANNOT_SIGN; continue; }   /* checked before the hash is touched */
      uint8_t trailer[6] = {4, 0xFF, 0, 0, (uint8_t)(s.prefix_len >> 8), (uint8_t)s.prefix_len};
      h_update(&hc, s.prefix, s.prefix_len);
      if (!s.v3) h_update(&hc, trailer, 6);            /* VerifySignatureV3: type || creation time only */
      uint8_t dg[64];
      unsigned dl = h_peek(&hc, dg);
      if (dg[0] != s.tag[0] || dg[1] != s.tag[1]) { st = ST_HASH_TAG; continue; }
      if (k->pk_algo != s.pk_algo) { st = ST_ALGO_MISMATCH; continue; }
      if (n_pk_ops) ++*n_pk_ops;
      int ok;
      if (k->pk_algo == 1 || k->pk_algo == 3) ok = rsa_verify(k, s.hash_id, dg, (int)dl, s.mpi[0], s.mpi_len[0], ctx);
      else if (k->pk_algo == 17) {
        int sub = (BN_num_bits(k->q) + 7) / 8;
        ok = dsa_verify(k, dg, (int)dl > sub ? sub : (int)dl, s.mpi[0], s.mpi_len[0], s.mpi[1], s.mpi_len[1], ctx);
      } else { st = ST_UNSUPPORTED; continue; }
      if (ok) { *signer = k->entity_id; TRACE(ST_OK); RETURN(ST_OK); }
      st = ST_BAD_SIG;
    }
    TRACE(st);
    RETURN(st);
  }
#undef RETURN
#undef TRACE
}

/* PGPCollectiveSignature.Verify for one item (crypto_pgp.go:485-500) */
static int collective_one(const oracle* o, const uint8_t* tbs, uint64_t tl, const uint8_t* sd, uint64_t sl, uint32_t* nver,
                          uint8_t* trace, int* ntrace, int cap, BN_CTX* ctx, uint64_t* ops) {
  uint64_t pos = 0;
  uint64_t verified[4096];
  int nv = 0;
  while (pos < sl) {
    uint64_t signer = 0;
    int st = check_detached(o, tbs, tl, sd, sl, &pos, &signer, trace, ntrace, cap, ctx, ops);
    if (st == ST_OK) {
      if (nv < 4096) verified[nv++] = signer;
      if (is_sufficient(o, verified, nv)) { *nver = (uint32_t)nv; return 0; }
    }
  }
  *nver = (uint32_t)nv;
  return 2;   /* ErrInsufficientNumberOfSignatures */
}

int oracle_trace_item(void* h, const uint8_t* tbs, uint64_t tl, const uint8_t* sd, uint64_t sl, uint8_t* trace, int cap,
                      uint32_t* nver, int* err) {
  BN_CTX* ctx = BN_CTX_new();
  int nt = 0;
  uint64_t ops = 0;
  *err = collective_one((oracle*)h, tbs, tl, sd, sl, nver, trace, &nt, cap, ctx, &ops);
  BN_CTX_free(ctx);
  return nt;
}

typedef struct {
  const oracle* o;
  uint32_t lo, hi;
  const uint8_t* tbs; const uint64_t* tbs_off; const uint8_t* ss; const uint64_t* ss_off;
  uint8_t* err; uint32_t* nver;
  uint64_t ops;
} job;

static void* worker(void* arg) {
  job* j = (job*)arg;
  BN_CTX* ctx = BN_CTX_new();
  for (uint32_t i = j->lo; i < j->hi; ++i) {
    int nt = 0;
    uint32_t nv = 0;
    int e = collective_one(j->o, j->tbs + j->tbs_off[i], j->tbs_off[i + 1] - j->tbs_off[i], j->ss + j->ss_off[i],
                           j->ss_off[i + 1] - j->ss_off[i], &nv, NULL, &nt, 0, ctx, &j->ops);
    if (j->err) j->err[i] = (uint8_t)e;
    if (j->nver) j->nver[i] = nv;
  }
  BN_CTX_free(ctx);
  return NULL;
}

/* Batched collective verify over n_items with n_threads host threads (items split evenly).
 * Returns the number of public-key operations performed (the reference's early exit included). */
uint64_t oracle_collective_verify(void* h, uint32_t n_items, const uint8_t* tbs, const uint64_t* tbs_off, const uint8_t* ss,
                                  const uint64_t* ss_off, uint8_t* err, uint32_t* nver, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if ((uint32_t)n_threads > n_items) n_threads = n_items ? (int)n_items : 1;
  job* jobs = (job*)calloc(n_threads, sizeof(job));
  pthread_t* th = (pthread_t*)calloc(n_threads, sizeof(pthread_t));
  for (int t = 0; t < n_threads; ++t) {
    jobs[t].o = (oracle*)h;
    jobs[t].lo = (uint32_t)((uint64_t)n_items * t / n_threads);
    jobs[t].hi = (uint32_t)((uint64_t)n_items * (t + 1) / n_threads);
    jobs[t].tbs = tbs; jobs[t].tbs_off = tbs_off; jobs[t].ss = ss; jobs[t].ss_off = ss_off;
    jobs[t].err = err; jobs[t].nver = nver;
    if (n_threads == 1) worker(&jobs[t]);
    else pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  uint64_t ops = 0;
  for (int t = 0; t < n_threads; ++t) {
    if (n_threads > 1) pthread_join(th[t], NULL);
    ops += jobs[t].ops;
  }
  free(jobs);
  free(th);
  return ops;
}
