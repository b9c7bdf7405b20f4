// Share-combine kernels of the threshold-signature path (BASELINE config 5), gfx950.
//   k_modmul_product     S = prod psig_i mod N            calculateSignature, crypto/threshold/rsa/rsa.go:318-329
//   k_lagrange_inv       (prod_{m!=j}(x_m - x_j))^-1 mod m, exact for any modulus coprime to it
//   k_lagrange_terms     lambda_j = a_j * inv_j mod m  and  sum_j lambda_j * y_j mod m
//                                                         sss.Lagrange / calculateSecret (crypto/sss/sss.go:69-107),
//                                                         calculateS (crypto/threshold/dsa/dsa_core.go:389-403)
//   k_multiexp           prod_j base_j ^ e_j mod p          CalculateR (crypto/threshold/dsa/dsa.go:33-52)
//   k_u256_inv_modq, k_limbs_mod_q                          the mod-q tail of CalculateR
// All big-number work reuses the quad Montgomery multiplier of mont28.h (4 lanes per operation).
#pragma once
// (kernels.hip is included before this file by capi.hip)

namespace bftkv {

struct ModTab {               // per distinct modulus, built on the host (hostbn::mont_setup)
  const uint32_t* n_limbs;    // [n_mods][76]
  const uint32_t* r2_limbs;   // [n_mods][76]  (2^2128)^2 mod n: 4 lanes x 19 limbs
  const uint32_t* n0inv;      // [n_mods]
  const uint32_t* r2w_limbs;  // [n_mods][80]  (2^2240)^2 mod n: 8 lanes x 10 limbs (k_multiexp's form for calls that do not fill the chip)
};

#define QUAD_SETUP()                                                        \
  constexpr int L = MONT_L;                                                 \
  const uint32_t quad = threadIdx.x >> 2;                                   \
  const int qlane = threadIdx.x & 3;                                        \
  const uint32_t gq = blockIdx.x * QUADS_PER_BLOCK + quad;                  \
  const bool active = gq < n_ops;                                           \
  const uint32_t op = active ? gq : (n_ops - 1);                            \
  uint32_t* a_lds = a_sh + quad * MONT_N + qlane * L;                       \
  const uint32_t* a_rd = a_sh + quad * MONT_N;

#define MONT(out, bexpr)                                                    \
  do {                                                                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                  \
    mont_mul(out, a_rd, bexpr, n, n0inv, qlane);                            \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                  \
  } while (0)

__device__ __forceinline__ void store_mod_result(uint32_t* dst, uint32_t (&t)[MONT_L], const uint32_t (&n)[MONT_L], int qlane) {
  // t <= n after mont(., 1) + canonicalize; t == n only for 0
  uint32_t diff = 0;
#pragma unroll
  for (int k = 0; k < MONT_L; ++k) diff |= t[k] ^ n[k];
  diff = quad_or(diff);
#pragma unroll
  for (int k = 0; k < MONT_L; ++k) dst[k] = (diff == 0) ? 0u : t[k];
}

// ---- RSA combine: product of k factors mod N ---------------------------------------------------------
__global__ void __launch_bounds__(RSA_BLOCK) k_modmul_product(uint32_t n_ops, uint32_t k_factors, const uint32_t* __restrict__ f_limbs /*[n_ops][k][76]*/,
                                                              const uint32_t* __restrict__ mod_idx, ModTab mt, uint32_t* __restrict__ out_limbs) {
  __shared__ uint32_t a_sh[QUADS_PER_BLOCK * MONT_N];
  QUAD_SETUP();
  const uint32_t mi = mod_idx[op];
  uint32_t n[L], r2[L], acc[L], t[L];
#pragma unroll
  for (int k = 0; k < L; ++k) { n[k] = mt.n_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; r2[k] = mt.r2_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; }
  const uint32_t n0inv = mt.n0inv[mi];
  for (uint32_t j = 0; j < k_factors; ++j) {
    const uint32_t* fp = f_limbs + ((uint64_t)op * k_factors + j) * MONT_N + qlane * L;
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = fp[k];
    MONT(t, r2);                                   // f_j * R
    if (j == 0) {
#pragma unroll
      for (int k = 0; k < L; ++k) acc[k] = t[k];
    } else {
#pragma unroll
      for (int k = 0; k < L; ++k) a_lds[k] = t[k];
      MONT(t, acc);                                // acc * f_j (Montgomery form)
#pragma unroll
      for (int k = 0; k < L; ++k) acc[k] = t[k];
    }
  }
#pragma unroll
  for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
  MONT(t, acc);
  canonicalize(t, qlane);
  if (active) store_mod_result(out_limbs + (uint64_t)op * MONT_N + qlane * L, t, n, qlane);
}

// ---- Lagrange: small-integer side ------------------------------------------------------------------
// Per (op, share j): a = prod_{m != j} x_m, b = prod_{m != j} (x_m - x_j) (exact, |.| < 2^31 required),
// inv = (b mod m)^-1 mod m computed WITHOUT big-number inversion: with t = m^-1 mod |b|,
// |b|^-1 mod m = (1 + ((|b| - t) mod |b|) * m) / |b| exactly -- valid for any m coprime to b.
// status: 0 ok, 1 no inverse / duplicate x, 2 magnitude overflow (fenced: DESIGN.md)
__device__ __forceinline__ int64_t egcd_inv64(int64_t a, int64_t m) {   // a^-1 mod m, or -1
  int64_t r0 = m, r1 = a % m, s0 = 0, s1 = 1;
  while (r1 != 0) {
    int64_t qq = r0 / r1;
    int64_t t = r0 - qq * r1; r0 = r1; r1 = t;
    t = s0 - qq * s1; s0 = s1; s1 = t;
  }
  if (r0 != 1) return -1;
  return s0 < 0 ? s0 + m : s0;
}

__global__ void __launch_bounds__(64) k_lagrange_inv(uint32_t n_ops, uint32_t k_shares, const int32_t* __restrict__ xs /*[n_ops][k]*/,
                                                     const uint32_t* __restrict__ mod_idx, ModTab mt,
                                                     uint32_t* __restrict__ inv_limbs /*[n_ops][k][76]: a_j * (b_j^-1 mod m), an exact integer below 2^31 m*/,
                                                     uint8_t* __restrict__ status /*[n_ops]*/) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_ops * k_shares) return;
  const uint32_t op = t / k_shares, j = t % k_shares;
  const int32_t* x = xs + (uint64_t)op * k_shares;
  const uint32_t* m = mt.n_limbs + (uint64_t)mod_idx[op] * MONT_N;
  uint32_t* out = inv_limbs + (uint64_t)t * MONT_N;
  int64_t a = 1, b = 1;
  bool overflow = false;
  for (uint32_t i = 0; i < k_shares; ++i) {
    if (x[i] == x[j]) continue;                            // sss.Lagrange skips every res == x (sss.go:99-101)
    a *= x[i];
    b *= (int64_t)x[i] - x[j];
    if (a >= (1ll << 31) || a <= -(1ll << 31) || b >= (1ll << 31) || b <= -(1ll << 31)) overflow = true;
  }
  uint8_t st = 0;
  if (overflow || a < 0) st = 2;
  const int64_t ab = b < 0 ? -b : b;
  if (!st && ab == 0) st = 1;
  int64_t tinv = 0;
  if (!st && ab > 1) {
    uint64_t r = 0;
    for (int i = MONT_N - 1; i >= 0; --i) r = ((r << MONT_W) | m[i]) % (uint64_t)ab;   // m mod |b|
    tinv = egcd_inv64((int64_t)r, ab);
    if (tinv < 0) st = 1;
  }
  if (st) { atomicOr((unsigned int*)(status + (op & ~3u)), (unsigned int)st << (8 * (op & 3))); for (int i = 0; i < MONT_N; ++i) out[i] = 0; return; }
  if (ab == 1) {
    for (int i = 0; i < MONT_N; ++i) out[i] = (i == 0) ? 1u : 0u;
  } else {
    const uint64_t kk = (uint64_t)((ab - tinv) % ab);
    // prod = 1 + kk * m   (78 limbs), then exact division by |b| from the top
    uint64_t carry = 1;
    uint32_t prod[MONT_N + 2];
    for (int i = 0; i < MONT_N; ++i) { uint64_t v = (uint64_t)m[i] * kk + carry; prod[i] = (uint32_t)v & MONT_MASK; carry = v >> MONT_W; }
    prod[MONT_N] = (uint32_t)carry & MONT_MASK;
    prod[MONT_N + 1] = (uint32_t)(carry >> MONT_W);
    uint64_t rem = 0;
    for (int i = MONT_N + 1; i >= 0; --i) {
      uint64_t cur = (rem << MONT_W) | prod[i];
      uint64_t qd = cur / (uint64_t)ab;
      rem = cur % (uint64_t)ab;
      if (i < MONT_N) out[i] = (uint32_t)qd;
    }
  }
  if (b < 0) {   // (-|b|)^-1 = m - |b|^-1
    int64_t borrow = 0;
    for (int i = 0; i < MONT_N; ++i) {
      int64_t v = (int64_t)m[i] - out[i] - borrow;
      borrow = v < 0;
      out[i] = (uint32_t)(v + (borrow << MONT_W)) & MONT_MASK;
    }
  }
  // times a_j (0 <= a_j < 2^31) as an integer: below 2^31 m < R, so k_lagrange_terms takes it as the plain operand of a product and
  // the reduction happens there (one Montgomery product per term saved, two where the coefficients are written out)
  uint64_t carry = 0;
  for (int i = 0; i < MONT_N; ++i) { const uint64_t v = (uint64_t)out[i] * (uint64_t)a + carry; out[i] = (uint32_t)v & MONT_MASK; carry = v >> MONT_W; }
}

// Per op (quad): lambda_j = (a_j inv_j) mod m (optionally written out), S = sum_j lambda_j * y_j mod m.  Two products per term
// (the exact integer a_j inv_j into the domain, times y_j), three where the coefficients are written out.
__global__ void __launch_bounds__(RSA_BLOCK) k_lagrange_terms(uint32_t n_ops, uint32_t k_shares, const uint32_t* __restrict__ ainv_limbs,
                                                              const uint32_t* __restrict__ y_limbs /*[n_ops][k][76] or null*/,
                                                              const uint32_t* __restrict__ mod_idx, ModTab mt,
                                                              uint32_t* __restrict__ lambda_out /*[n_ops][k][76] or null*/,
                                                              uint32_t* __restrict__ sum_out /*[n_ops][76] or null*/,
                                                              const uint8_t* __restrict__ status /*k_lagrange_inv's: bit 1 = left to the big path*/) {
  __shared__ uint32_t a_sh[QUADS_PER_BLOCK * MONT_N];
  QUAD_SETUP();
  // Rows of flagged operations are zeroes from here on: k_lagrange_finish rewrites them when the big path follows, and when it does
  // not (a device-resident caller whose promised index bound did not hold) the caller reads a fenced status beside zeroes, never
  // stale pool memory (fail closed: ADVICE r05)
  const bool flagged = active && (status[op] & 2u);
  if (flagged) {
    if (sum_out) {
#pragma unroll
      for (int k = 0; k < L; ++k) sum_out[(uint64_t)op * MONT_N + qlane * L + k] = 0u;
    }
    if (lambda_out)
      for (uint32_t j = 0; j < k_shares; ++j) {
#pragma unroll
        for (int k = 0; k < L; ++k) lambda_out[((uint64_t)op * k_shares + j) * MONT_N + qlane * L + k] = 0u;
      }
  }
  if (!__any(active && !(status[op] & 2u))) return;      // every row of this wave is rewritten by k_lagrange_finish (64 nodes: all of them)
  const uint32_t mi = mod_idx[op];
  uint32_t n[L], r2[L], acc[L], t[L], u[L];
#pragma unroll
  for (int k = 0; k < L; ++k) { n[k] = mt.n_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; r2[k] = mt.r2_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; acc[k] = 0; }
  const uint32_t n0inv = mt.n0inv[mi];
  for (uint32_t j = 0; j < k_shares; ++j) {
    const uint64_t sj = (uint64_t)op * k_shares + j;
    const uint32_t* ip = ainv_limbs + sj * MONT_N + qlane * L;
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = ip[k];
    MONT(u, r2);                                                     // lambda_j R  (a_j inv_j < 2^31 m: the result is below m(1 + 2^-49))
    if (lambda_out) {
#pragma unroll
      for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
      MONT(t, u);                                                    // lambda_j, plain
      canonicalize(t, qlane);
      if (active && !flagged) store_mod_result(lambda_out + sj * MONT_N + qlane * L, t, n, qlane);
    }
    if (sum_out) {
      const uint32_t* yp = y_limbs + sj * MONT_N + qlane * L;
#pragma unroll
      for (int k = 0; k < L; ++k) a_lds[k] = yp[k];
      MONT(t, u);                                                    // lambda_j * y_j (plain, < 2m)
#pragma unroll
      for (int k = 0; k < L; ++k) acc[k] += t[k];
      canonicalize(acc, qlane);
    }
  }
  if (sum_out) {
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = acc[k];
    MONT(t, r2);
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
    MONT(u, t);
    canonicalize(u, qlane);
    if (active && !flagged) store_mod_result(sum_out + (uint64_t)op * MONT_N + qlane * L, u, n, qlane);
  }
}

// ---- Lagrange: the BIG-integer side (round 5) ----------------------------------------------------------------------
// k_lagrange_inv works on exact 31-bit products: enough for the reference's own parameters (n = 10 nodes: k = 7, 2t = 8), not
// for the clusters BASELINE names -- at n = 64, k = 22 the numerator alone is 64^21 -- and big.Int has no such bound
// (sss.Lagrange, crypto/sss/sss.go:94-107).  Operations k_lagrange_inv flagged (status bit 2: a product past 2^31, or a negative
// x) are redone here:
//   k_lagrange_big     thread / term: |a_j| = prod |x_i|, |b_j| = prod |x_i - x_j| (i != j, every x_i == x_j skipped as
//                      sss.Lagrange skips them) as EXACT integers of up to 2128 bits -- a limb row times a 32-bit factor is
//                      76 MACs, nothing next to a Montgomery product -- and the sign of a_j / b_j; a product beyond 2128 bits
//                      stays fenced (256 nodes with ids up to 255: k up to ~260 terms fit)
//   k_lagrange_prep    quad / op: B_j = b_j R mod m, the prefix products Pre_j = B_0 ... B_(j-1), and D = prod_j b_j mod m
//   k_modinv           D^-1 mod m -- ONE big inverse per operation (Montgomery's trick).  The reference inverts every b_j
//                      (ModInverse returns nil for one without inverse and the next Mul dereferences it): all b_j are
//                      invertible exactly when their product is, so "no inverse" is the same outcome for the operation
//   k_lagrange_finish  quad / op: lambda_j = a_j Pre_j Suf_j (+-D^-1) mod m (Suf_j = B_(j+1) ... B_(k-1), built backwards),
//                      written out and / or folded into S = sum_j lambda_j y_j mod m; sets the operation's final status
__device__ __forceinline__ bool limbs_mul_small(uint32_t* v, uint32_t& len, uint64_t f) {      // v *= f (f < 2^32); false: beyond 76 limbs
  uint64_t carry = 0;
  for (uint32_t i = 0; i < len; ++i) { const uint64_t w = (uint64_t)v[i] * f + carry; v[i] = (uint32_t)w & MONT_MASK; carry = w >> MONT_W; }
  while (carry) {
    if (len == (uint32_t)MONT_N) return false;
    v[len++] = (uint32_t)carry & MONT_MASK;
    carry >>= MONT_W;
  }
  return true;
}

__device__ __forceinline__ void status_byte_set(uint8_t* status, uint32_t op, uint8_t v) {      // bytes are packed four to an atomic word
  unsigned int* w = (unsigned int*)(status + (op & ~3u));
  const unsigned int sh = 8u * (op & 3u);
  atomicAnd(w, ~(0xFFu << sh));
  atomicOr(w, (unsigned int)v << sh);
}

__global__ void __launch_bounds__(64) k_lagrange_big(uint32_t n_ops, uint32_t k_shares, const int32_t* __restrict__ xs, const uint8_t* __restrict__ status,
                                                     uint32_t* __restrict__ a_big /*[n_ops][k][76]*/, uint32_t* __restrict__ b_big, uint8_t* __restrict__ sign /*[n_ops][k]*/,
                                                     uint8_t* __restrict__ big_st /*[n_ops]: 2 = a product beyond 2128 bits*/) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_ops * k_shares) return;
  const uint32_t op = t / k_shares, j = t % k_shares;
  if (!(status[op] & 2u)) return;
  const int32_t* x = xs + (uint64_t)op * k_shares;
  uint32_t A[MONT_N], B[MONT_N];
  for (int i = 0; i < MONT_N; ++i) { A[i] = 0; B[i] = 0; }
  A[0] = 1; B[0] = 1;
  uint32_t la = 1, lb = 1;
  bool neg = false, ok = true;
  for (uint32_t i = 0; i < k_shares && ok; ++i) {
    if (x[i] == x[j]) continue;
    const int64_t xi = x[i], d = (int64_t)x[i] - x[j];
    neg ^= (xi < 0) ^ (d < 0);
    ok = limbs_mul_small(A, la, (uint64_t)(xi < 0 ? -xi : xi)) && limbs_mul_small(B, lb, (uint64_t)(d < 0 ? -d : d));
  }
  if (!ok) { big_st[op] = 2; return; }          // (every thread that writes, writes the same value)
  uint32_t* ao = a_big + (uint64_t)t * MONT_N;
  uint32_t* bo = b_big + (uint64_t)t * MONT_N;
  for (int i = 0; i < MONT_N; ++i) { ao[i] = A[i]; bo[i] = B[i]; }
  sign[t] = neg ? 1 : 0;
}

__global__ void __launch_bounds__(RSA_BLOCK) k_lagrange_prep(uint32_t n_ops, uint32_t k_shares, uint32_t* __restrict__ b_big /*in: b_j; out: B_j*/, const uint8_t* __restrict__ status,
                                                             const uint8_t* __restrict__ big_st, const uint32_t* __restrict__ mod_idx, ModTab mt,
                                                             uint32_t* __restrict__ pre /*[n_ops][k][76], Montgomery form*/, uint32_t* __restrict__ d_plain /*[n_ops][76]*/) {
  __shared__ uint32_t a_sh[QUADS_PER_BLOCK * MONT_N];
  QUAD_SETUP();
  const bool mine = active && (status[op] & 2u) && !big_st[op];
  if (!__any(mine)) return;
  const uint32_t mi = mod_idx[op];
  uint32_t n[L], r2[L], acc[L], t[L];
#pragma unroll
  for (int k = 0; k < L; ++k) { n[k] = mt.n_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; r2[k] = mt.r2_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; }
  const uint32_t n0inv = mt.n0inv[mi];
#pragma unroll
  for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
  MONT(acc, r2);                                                     // R mod m: the empty product
  for (uint32_t j = 0; j < k_shares; ++j) {
    const uint64_t sj = (uint64_t)op * k_shares + j;
    if (mine) {
#pragma unroll
      for (int k = 0; k < L; ++k) pre[sj * MONT_N + qlane * L + k] = acc[k];
    }
    uint32_t* bp = b_big + sj * MONT_N + qlane * L;
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = mine ? bp[k] : 0u;
    MONT(t, r2);                                                     // B_j = b_j R, kept for k_lagrange_finish's suffix products
    if (mine) {
#pragma unroll
      for (int k = 0; k < L; ++k) bp[k] = t[k];
    }
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = t[k];
    MONT(t, acc);                                                    // Pre_(j+1)
#pragma unroll
    for (int k = 0; k < L; ++k) acc[k] = t[k];
  }
#pragma unroll
  for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
  MONT(t, acc);                                                      // D = prod b_j mod m, plain
  canonicalize(t, qlane);
  if (mine) store_mod_result(d_plain + (uint64_t)op * MONT_N + qlane * L, t, n, qlane);
}

// Products per term: a_j (plain, exact) * Pre_j R -> plain; * Suf_j R -> plain; * (+-D^-1) R^2 -> lambda_j R; then lambda_j R * y_j
// -> a term of the sum, lambda_j R * 1 -> lambda_j; Suf_(j-1) R = B_j * Suf_j R.  The sign rides on the inverse: the operation's two
// scratch rows hold D^-1 R^2 and -D^-1 R^2 (`d_inv` in: D^-1 plain; both rows are rewritten here and read back by the lanes that
// wrote them).  Five or six products per term where the first form took nine or ten.
__global__ void __launch_bounds__(RSA_BLOCK) k_lagrange_finish(uint32_t n_ops, uint32_t k_shares, const uint32_t* __restrict__ a_big, const uint32_t* __restrict__ bt_big /*B_j = b_j R*/,
                                                               const uint8_t* __restrict__ sign, const uint32_t* __restrict__ pre, uint32_t* d_inv /*[n_ops][76]*/,
                                                               uint32_t* d_neg /*[n_ops][76] scratch*/,
                                                               const uint8_t* __restrict__ inv_st /*[n_ops]: 1 = D has no inverse*/, const uint8_t* __restrict__ big_st,
                                                               const uint32_t* __restrict__ y_limbs, const uint32_t* __restrict__ mod_idx, ModTab mt,
                                                               uint32_t* __restrict__ lambda_out, uint32_t* __restrict__ sum_out, uint8_t* __restrict__ status) {
  __shared__ uint32_t a_sh[QUADS_PER_BLOCK * MONT_N];
  QUAD_SETUP();
  const bool flagged = active && (status[op] & 2u);
  if (!__any(flagged)) return;
  const uint8_t bad = flagged ? (big_st[op] ? 2 : (inv_st[op] ? 1 : 0)) : 0;
  const bool mine = flagged && !bad;
  const uint32_t mi = mod_idx[op];
  const uint64_t row = (uint64_t)op * MONT_N + qlane * L;
  uint32_t n[L], suf[L], acc[L], t[L], u[L];
#pragma unroll
  for (int k = 0; k < L; ++k) { n[k] = mt.n_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; u[k] = mt.r2_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; }
  const uint32_t n0inv = mt.n0inv[mi];
#pragma unroll
  for (int k = 0; k < L; ++k) a_lds[k] = mine ? d_inv[row + k] : 0u;
  MONT(t, u);                                                        // D^-1 R
#pragma unroll
  for (int k = 0; k < L; ++k) a_lds[k] = t[k];
  MONT(suf, u);                                                      // D^-1 R^2
  // m - 1 = -1 (m is odd: m - 1 is m with its lowest bit cleared)
#pragma unroll
  for (int k = 0; k < L; ++k) a_lds[k] = n[k] & ~((qlane == 0 && k == 0) ? 1u : 0u);
  MONT(t, u);                                                        // (m - 1) R
#pragma unroll
  for (int k = 0; k < L; ++k) a_lds[k] = suf[k];
  MONT(acc, t);                                                      // -D^-1 R^2
  if (mine) {
#pragma unroll
    for (int k = 0; k < L; ++k) { d_inv[row + k] = suf[k]; d_neg[row + k] = acc[k]; }
  }
#pragma unroll
  for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
  MONT(suf, u);                                                      // R mod m: the empty suffix product
#pragma unroll
  for (int k = 0; k < L; ++k) acc[k] = 0;
  for (uint32_t jj = 0; jj < k_shares; ++jj) {
    const uint32_t j = k_shares - 1 - jj;
    const uint64_t sj = (uint64_t)op * k_shares + j;
#pragma unroll
    for (int k = 0; k < L; ++k) { a_lds[k] = mine ? a_big[sj * MONT_N + qlane * L + k] : 0u; u[k] = mine ? pre[sj * MONT_N + qlane * L + k] : 0u; }
    MONT(t, u);                                                      // |a_j| Pre_j          (< 3m: |a_j| < R)
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = t[k];
    MONT(u, suf);                                                    // |a_j| Pre_j Suf_j
    const uint32_t* dsel = (mine && sign[sj]) ? d_neg : d_inv;
#pragma unroll
    for (int k = 0; k < L; ++k) { a_lds[k] = u[k]; t[k] = mine ? dsel[row + k] : 0u; }
    MONT(u, t);                                                      // lambda_j R, sign included
    if (lambda_out) {
#pragma unroll
      for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
      MONT(t, u);
      canonicalize(t, qlane);
      if (mine) store_mod_result(lambda_out + sj * MONT_N + qlane * L, t, n, qlane);
    }
    if (sum_out) {
#pragma unroll
      for (int k = 0; k < L; ++k) a_lds[k] = mine ? y_limbs[sj * MONT_N + qlane * L + k] : 0u;
      MONT(t, u);                                                    // lambda_j y_j (plain, < 2m)
#pragma unroll
      for (int k = 0; k < L; ++k) acc[k] += t[k];
      canonicalize(acc, qlane);
    }
    // Suf_(j-1) = B_j Suf_j
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = mine ? bt_big[sj * MONT_N + qlane * L + k] : 0u;
    MONT(t, suf);
#pragma unroll
    for (int k = 0; k < L; ++k) suf[k] = t[k];
  }
  if (sum_out) {
#pragma unroll
    for (int k = 0; k < L; ++k) { a_lds[k] = acc[k]; u[k] = mt.r2_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; }
    MONT(t, u);
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
    MONT(u, t);
    canonicalize(u, qlane);
    if (mine) store_mod_result(sum_out + (uint64_t)op * MONT_N + qlane * L, u, n, qlane);
    else if (flagged) {
#pragma unroll
      for (int k = 0; k < L; ++k) sum_out[(uint64_t)op * MONT_N + qlane * L + k] = 0;
    }
  }
  if (flagged && qlane == 0) status_byte_set(status, op, bad);
}

// r = prod_j base_j ^ e_j mod p, e_j up to 256 bits (radix-2^28 limbs; 10 limbs = 280 bits cover them).
// Straus' simultaneous exponentiation with 4-bit windows: per base the multiples b^1 .. b^15 (Montgomery form) go to a
// global table (4.5 KB per base, written and re-read by the same quad), then 70 windows of 4 shared squarings and at
// most one table multiplication per base.  28 = 7 * 4, so a window digit never straddles two limbs.
constexpr int MULTIEXP_WIN = 4, MULTIEXP_ENT = 15;
// `idx_div`: an operation's bases may be split over idx_div consecutive quads (each takes k_bases of them and yields a
// partial product that k_modmul_product multiplies up): 10,000 combines are only 625 waves on 1024 SIMDs, and one wave alone
// on a SIMD issues a VALU instruction only every ~9.5 cycles (tools/microbench), so a chain of ~900 dependent products per
// wave is bounded by single-wave issue; the modulus of quad `op` is that of operation op / idx_div.
//
// The whole schedule -- table build, squarings, table multiplications, leaving the Montgomery domain -- runs through ONE
// loop with two multiplier call sites (general and squaring): six inlined copies of mont_mul cost 239 VGPRs (two waves per
// SIMD); this form allows three.
//
// <L, TPI>: 19 limbs x 4 lanes per number (R = 2^2128) when the call fills the chip, 10 limbs x 8 lanes (R = 2^2240, its
// own R^2 column of the ModTab) when it does not: twice the waves, each half as long.  Rows in memory are MONT_N limbs
// either way; the 8-lane form reads zeros past them and stores nothing there (every value is below 2p < 2^2049).
constexpr int MULTI_L8 = 10, MULTI_TPI8 = 8;
constexpr int MONT_N_WIDE = MULTI_L8 * MULTI_TPI8;     // 80 limbs: row length of ModTab::r2w_limbs
// BLOCK: the kernel's waves never meet (wavefront fences only), so the block size only decides in what units the dispatcher hands
// waves to CUs: one-wave blocks go wherever a SIMD has room, four-wave blocks need room on all four SIMDs of one CU.
template <int L, int TPI, int BLOCK = RSA_BLOCK>
__global__ void __launch_bounds__(BLOCK) k_multiexp(uint32_t n_ops, uint32_t k_bases, const uint32_t* __restrict__ base_limbs /*[n_ops][k][76]*/,
                                                        const uint32_t* __restrict__ exp_limbs /*[n_ops][k][76] radix 2^28*/,
                                                        const uint32_t* __restrict__ mod_idx, ModTab mt, uint32_t* __restrict__ scratch /*[n_ops][k][15][76]*/,
                                                        uint32_t* __restrict__ out_limbs, uint32_t idx_div, uint32_t exp_windows) {
  constexpr int NL = L * TPI, GROUPS = BLOCK / TPI;
  static_assert(NL >= MONT_N, "a group holds a whole row");
  __shared__ uint32_t a_sh[GROUPS * NL];
  const uint32_t grp = threadIdx.x / TPI;
  const int qlane = threadIdx.x % TPI;
  const uint32_t gq = blockIdx.x * GROUPS + grp;
  const bool active = gq < n_ops;
  const uint32_t op = active ? gq : (n_ops - 1);
  uint32_t* a_lds = a_sh + grp * NL + qlane * L;
  const uint32_t* a_rd = a_sh + grp * NL;
  // limb k of this lane in a MONT_N-limb row of memory
  auto ld = [&](const uint32_t* row, int k) -> uint32_t { const int gi = qlane * L + k; return (NL == MONT_N || gi < MONT_N) ? row[gi] : 0u; };
  auto st = [&](uint32_t* row, int k, uint32_t v) { const int gi = qlane * L + k; if (NL == MONT_N || gi < MONT_N) row[gi] = v; };
  const uint32_t mi = mod_idx[op / idx_div];
  uint32_t n[L], y[L], t[L];          // the register operand of every product is y (R^2 is loaded into it where needed)
  const uint32_t* r2p = (TPI == MONT_TPI ? mt.r2_limbs + (uint64_t)mi * MONT_N : mt.r2w_limbs + (uint64_t)mi * MONT_N_WIDE) + qlane * L;
  const uint32_t* nrow = mt.n_limbs + (uint64_t)mi * MONT_N;
#pragma unroll
  for (int k = 0; k < L; ++k) n[k] = ld(nrow, k);
  const uint32_t n0inv = mt.n0inv[mi];
  // wave-uniform program counter
  enum : int { P_TO_MONT = 0, P_POWERS, P_ONE, P_SQR, P_TABMUL, P_LEAVE, P_DONE };
  int phase = P_TO_MONT, w = (int)exp_windows - 1, sq = 0;
  uint32_t j = 0, d = 0;           // base index / table digit (P_POWERS: entry being built)
  uint32_t dig = 0;                // this group's digit in P_TABMUL
  while (phase != P_DONE) {
    const uint64_t sj = (uint64_t)op * k_bases + j;
    uint32_t* tab = scratch + sj * MULTIEXP_ENT * MONT_N;
    // ---- operands: a through LDS, b in registers
    if (phase == P_TO_MONT) {                       // b_j * R
      const uint32_t* brow = base_limbs + sj * MONT_N;
#pragma unroll
      for (int k = 0; k < L; ++k) { a_lds[k] = ld(brow, k); y[k] = r2p[k]; }
    } else if (phase == P_POWERS) {                 // b^(d+1) = b^d * b; a = b_j stays in LDS, y = the previous power
    } else if (phase == P_ONE) {                    // Montgomery one = mont(1, R^2)
#pragma unroll
      for (int k = 0; k < L; ++k) { a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u; y[k] = r2p[k]; }
    } else if (phase == P_SQR) {
#pragma unroll
      for (int k = 0; k < L; ++k) a_lds[k] = y[k];
    } else if (phase == P_TABMUL) {
      const uint32_t* row = scratch + (sj * MULTIEXP_ENT + (dig ? dig - 1 : 0)) * MONT_N;
#pragma unroll
      for (int k = 0; k < L; ++k) a_lds[k] = ld(row, k);
    } else {                                        // P_LEAVE: mont(y, 1)
#pragma unroll
      for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (phase == P_SQR) mont_mul<L, TPI, true>(t, a_rd, y, n, n0inv, qlane);
    else mont_mul<L, TPI, false>(t, a_rd, y, n, n0inv, qlane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- results and next step (scalar control flow)
    if (phase == P_TO_MONT) {
#pragma unroll
      for (int k = 0; k < L; ++k) { y[k] = t[k]; a_lds[k] = t[k]; if (active) st(tab, k, t[k]); }      // a = b_j R for the powers
      d = 1; phase = P_POWERS;
    } else if (phase == P_POWERS) {
#pragma unroll
      for (int k = 0; k < L; ++k) { y[k] = t[k]; if (active) st(tab + (uint64_t)d * MONT_N, k, t[k]); }
      if (++d == MULTIEXP_ENT) {
        if (++j == k_bases) { j = 0; phase = P_ONE; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent"); }   // own table rows are read back below
        else phase = P_TO_MONT;
      }
    } else {
      if (phase == P_LEAVE) break;
      if (phase != P_TABMUL || dig) {
#pragma unroll
        for (int k = 0; k < L; ++k) y[k] = t[k];
      }
      // advance: 4 squarings per window, then one table multiplication per base with a non-zero digit somewhere in the wave
      if (phase == P_ONE) { phase = P_SQR; sq = 0; }
      else if (phase == P_SQR) { if (++sq == MULTIEXP_WIN) { phase = P_TABMUL; j = 0; } else continue; }
      else ++j;
      while (phase == P_TABMUL) {
        if (j == k_bases) { j = 0; if (--w < 0) phase = P_LEAVE; else { phase = P_SQR; sq = 0; } break; }
        const int bit = w * MULTIEXP_WIN;
        const uint64_t sj2 = (uint64_t)op * k_bases + j;
        dig = active ? ((exp_limbs[sj2 * MONT_N + bit / MONT_W] >> (bit % MONT_W)) & 15u) : 0u;
        if (__any(dig != 0)) break;
        ++j;
      }
    }
  }
  canonicalize<L, TPI>(t, qlane);
  // t <= p after mont(., 1); t == p only for 0
  uint32_t diff = 0;
#pragma unroll
  for (int k = 0; k < L; ++k) diff |= t[k] ^ n[k];
  diff = grp_or<TPI>(diff);
  if (active) {
    uint32_t* orow = out_limbs + (uint64_t)op * MONT_N;
#pragma unroll
    for (int k = 0; k < L; ++k) st(orow, k, (diff == 0) ? 0u : t[k]);
  }
}

// Diagnostic (bftkv_gpu_selftest_reduce): out = v - m when v >= m, else v, for v < 2m -- reduce_once of mont28.h on its own, so
// that the subtraction path (taken by about one value in 2^50 on the verify path) is exercised on chosen inputs.
template <int L, int TPI>
__global__ void __launch_bounds__(RSA_BLOCK) k_reduce_once(uint32_t n_ops, const uint32_t* __restrict__ v_limbs, const uint32_t* __restrict__ mod_idx,
                                                           ModTab mt, uint32_t* __restrict__ out_limbs) {
  constexpr int GROUPS = RSA_BLOCK / TPI;
  const uint32_t grp = threadIdx.x / TPI;
  const int qlane = threadIdx.x % TPI;
  const uint32_t gq = blockIdx.x * GROUPS + grp;
  const bool active = gq < n_ops;
  const uint32_t op = active ? gq : (n_ops - 1);
  const uint32_t* nrow = mt.n_limbs + (uint64_t)mod_idx[op] * MONT_N;
  const uint32_t* vrow = v_limbs + (uint64_t)op * MONT_N;
  uint32_t n[L], y[L];
#pragma unroll
  for (int k = 0; k < L; ++k) {
    const int gi = qlane * L + k;
    n[k] = gi < MONT_N ? nrow[gi] : 0u;
    y[k] = gi < MONT_N ? vrow[gi] : 0u;
  }
  reduce_once<L, TPI>(y, n, qlane);
  if (active) {
#pragma unroll
    for (int k = 0; k < L; ++k) {
      const int gi = qlane * L + k;
      if (gi < MONT_N) out_limbs[(uint64_t)op * MONT_N + gi] = y[k];
    }
  }
}

// thread per op: out = in^-1 mod q (numbers as 76-limb radix-2^28, q <= 256 bits, odd); status |= 1 when no inverse
__device__ __forceinline__ U256 u256_from_limbs(const uint32_t* l) {
  U256 r = u256_zero();
  for (int j = 0; j < 10; ++j) {
    const uint32_t bit = 28u * j;
    const uint64_t v = (uint64_t)l[j] << (bit & 31);
    if ((bit >> 5) < 8) r.w[bit >> 5] |= (uint32_t)v;
    if ((bit >> 5) + 1 < 8) r.w[(bit >> 5) + 1] |= (uint32_t)(v >> 32);
  }
  return r;
}
__device__ __forceinline__ void u256_to_limbs(const U256& a, uint32_t* l) {
  for (int j = 0; j < MONT_N; ++j) {
    const uint32_t bit = 28u * j, wi = bit >> 5, sh = bit & 31;
    uint64_t v = 0;
    if (wi < 8) v = a.w[wi];
    if (wi + 1 < 8) v |= (uint64_t)a.w[wi + 1] << 32;
    l[j] = (wi < 8) ? ((uint32_t)(v >> sh) & MONT_MASK) : 0u;
  }
}
// `gate` / `gate_not`: as k_modinv (the big Lagrange path under a modulus of at most 256 bits: the order q of calculateS and
// CalculateR); `status` is then a plain byte per operation.
__global__ void __launch_bounds__(64) k_u256_inv_modq(uint32_t n_ops, const uint32_t* __restrict__ in_limbs, const uint32_t* __restrict__ mod_idx,
                                                      ModTab mt, uint32_t* __restrict__ out_limbs, uint8_t* __restrict__ status,
                                                      const uint8_t* __restrict__ gate = nullptr, const uint8_t* __restrict__ gate_not = nullptr) {
  const uint32_t op = blockIdx.x * blockDim.x + threadIdx.x;
  if (op >= n_ops) return;
  // (u256_modinv_odd votes across the wave: gated-out lanes run it on a harmless value instead of leaving)
  const bool skip = gate && (!(gate[op] & 2u) || gate_not[op]);
  const U256 q = u256_from_limbs(mt.n_limbs + (uint64_t)mod_idx[op] * MONT_N);
  U256 v = u256_from_limbs(in_limbs + (uint64_t)op * MONT_N);
  if (skip) { v = u256_zero(); v.w[0] = 1; }
  U256 w = u256_zero();
  if (u256_is_zero(v) || !u256_modinv_odd(v, q, w)) {
    if (gate) { if (!skip) status[op] = 1; } else atomicOr((unsigned int*)(status + (op & ~3u)), 1u << (8 * (op & 3)));
    w = u256_zero();
  }
  if (!skip) u256_to_limbs(w, out_limbs + (uint64_t)op * MONT_N);
}
// quad per op: out = in mod q (in: 76 limbs below R, q: the modulus of `mt` -- the order of CalculateR's group).  Two products of the
// multiplier that is already here (in * R mod q, then out of the domain): ~450 instructions per operation and a chain of two
// products per wave, where a thread per operation shifting 2128 bits through 256-bit compares took 85,000 instructions in a row.
__global__ void __launch_bounds__(RSA_BLOCK) k_limbs_mod_q(uint32_t n_ops, const uint32_t* __restrict__ in_limbs, const uint32_t* __restrict__ mod_idx,
                                                           ModTab mt, uint32_t* __restrict__ out_limbs) {
  __shared__ uint32_t a_sh[QUADS_PER_BLOCK * MONT_N];
  QUAD_SETUP();
  const uint32_t mi = mod_idx[op];
  uint32_t n[L], r2[L], t[L], u[L];
#pragma unroll
  for (int k = 0; k < L; ++k) { n[k] = mt.n_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; r2[k] = mt.r2_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; }
  const uint32_t n0inv = mt.n0inv[mi];
#pragma unroll
  for (int k = 0; k < L; ++k) a_lds[k] = in_limbs[(uint64_t)op * MONT_N + qlane * L + k];
  MONT(t, r2);                                                       // in * R mod q (< 2q)
#pragma unroll
  for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
  MONT(u, t);
  canonicalize(u, qlane);
  if (active) store_mod_result(out_limbs + (uint64_t)op * MONT_N + qlane * L, u, n, qlane);
}

}  // namespace bftkv

namespace bftkv {

// ---- sss.Distribute (crypto/sss/sss.go:23-47): f(x) = sum_j poly_j x^j mod m for x = 1..n ------------------------
// quad per (op, x); Horner over the Montgomery multiplier, the small x as a one-limb operand.
__global__ void __launch_bounds__(RSA_BLOCK) k_sss_distribute(uint32_t n_ops_total /* = n_polys * n_shares */, uint32_t n_shares, uint32_t k_coeffs,
                                                              const uint32_t* __restrict__ poly_limbs /*[n_polys][k][76]*/,
                                                              const uint32_t* __restrict__ mod_idx /*[n_polys]*/, ModTab mt,
                                                              uint32_t* __restrict__ out_limbs /*[n_polys][n_shares][76]*/) {
  __shared__ uint32_t a_sh[QUADS_PER_BLOCK * MONT_N];
  const uint32_t n_ops = n_ops_total;
  QUAD_SETUP();
  const uint32_t poly = op / n_shares, x = op % n_shares + 1;
  const uint32_t mi = mod_idx[poly];
  uint32_t n[L], r2[L], acc[L], t[L], u[L];
#pragma unroll
  for (int k = 0; k < L; ++k) { n[k] = mt.n_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; r2[k] = mt.r2_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; }
  const uint32_t n0inv = mt.n0inv[mi];
  const uint32_t* pp = poly_limbs + (uint64_t)poly * k_coeffs * MONT_N + qlane * L;
#pragma unroll
  for (int k = 0; k < L; ++k) acc[k] = pp[(uint64_t)(k_coeffs - 1) * MONT_N + k];
  for (int j = (int)k_coeffs - 2; j >= 0; --j) {
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = acc[k];
    MONT(t, r2);                                                     // acc * R
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? x : 0u;
    MONT(u, t);                                                      // acc * x  (< 2m)
#pragma unroll
    for (int k = 0; k < L; ++k) acc[k] = u[k] + pp[(uint64_t)j * MONT_N + k];
    canonicalize(acc, qlane);                                        // < 3m < R
  }
#pragma unroll
  for (int k = 0; k < L; ++k) a_lds[k] = acc[k];
  MONT(t, r2);
#pragma unroll
  for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
  MONT(u, t);
  canonicalize(u, qlane);
  if (active) store_mod_result(out_limbs + (uint64_t)op * MONT_N + qlane * L, u, n, qlane);
}

// ---- modular inverse of up to 2048-bit numbers, odd modulus, by division steps (thread per op) ------------------
// big.Int.ModInverse inside rsaContext.Sign for negative key fragments (crypto/threshold/rsa/rsa.go:164-167), and the one inverse
// per operation of the big Lagrange path.  The arithmetic is safegcd.inc (its header has the algorithm and the step bound; the CPU
// suite runs the same text against Python's inverse): at most 208 rounds of 30 steps whatever the operands, no reduction loop.
#define SG_FN __device__ __forceinline__
#define SG_UNROLL _Pragma("unroll")
#include "safegcd.inc"

// `gate` / `gate_not` (the big Lagrange path): only operations whose gate byte has bit 2 set and whose gate_not byte is zero are
// inverted; `status` is then a plain byte per operation (1 = no inverse), not the packed array of the public entry point.
__global__ void __launch_bounds__(64) k_modinv(uint32_t n_ops, const uint32_t* __restrict__ in_limbs, const uint32_t* __restrict__ mod_idx, ModTab mt,
                                               uint32_t* __restrict__ out_limbs, uint8_t* __restrict__ status, const uint8_t* __restrict__ gate = nullptr,
                                               const uint8_t* __restrict__ gate_not = nullptr) {
  const uint32_t op = blockIdx.x * blockDim.x + threadIdx.x;
  if (op >= n_ops) return;
  if (gate && (!(gate[op] & 2u) || gate_not[op])) return;
  sg_num x, m, inv;
  sg_from28(in_limbs + (uint64_t)op * MONT_N, MONT_N, &x);
  sg_from28(mt.n_limbs + (uint64_t)mod_idx[op] * MONT_N, MONT_N, &m);
  const bool ok = sg_modinv(&x, &m, &inv) != 0;
  if (!ok) { if (gate) status[op] = 1; else atomicOr((unsigned int*)(status + (op & ~3u)), 1u << (8 * (op & 3))); }
  uint32_t* o = out_limbs + (uint64_t)op * MONT_N;
  if (ok) sg_to28(&inv, o, MONT_N);
  else for (int j = 0; j < MONT_N; ++j) o[j] = 0u;
}

// status bytes as the header documents them: 0, 1 (no inverse) or 2 (fenced; wins when both were recorded)
__global__ void k_status_normalise(uint8_t* __restrict__ status, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t v = status[i];
  if (v != 0xFFu) status[i] = (v & 2u) ? 2u : (v & 1u);
}

}  // namespace bftkv
