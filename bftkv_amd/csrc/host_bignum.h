// Minimal host-side big integers for key-table precomputation (R^2 mod n, -n^-1 mod 2^28,
// radix-2^28 limb conversion).  Runs once per uploaded key; not on the verification path.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

namespace bftkv {
namespace hostbn {

constexpr int W28 = 28;
constexpr uint32_t MASK28 = (1u << W28) - 1;

// big-endian bytes -> little-endian 32-bit words (nwords entries)
inline void from_be(const uint8_t* be, uint32_t len, uint32_t* w, int nwords) {
  memset(w, 0, sizeof(uint32_t) * nwords);
  for (uint32_t i = 0; i < len; ++i) {
    uint32_t k = len - 1 - i;  // byte index from the LSB
    if ((int)(k >> 2) < nwords) w[k >> 2] |= (uint32_t)be[i] << (8 * (k & 3));
  }
}

inline int bit_length(const uint8_t* be, uint32_t len) {
  uint32_t i = 0;
  while (i < len && be[i] == 0) ++i;
  if (i == len) return 0;
  int top = 8;
  while (!((be[i] >> (top - 1)) & 1)) --top;
  return (int)(len - i - 1) * 8 + top;
}

inline int cmp(const uint32_t* a, const uint32_t* b, int n) {
  for (int i = n - 1; i >= 0; --i) {
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  }
  return 0;
}
inline void sub(uint32_t* a, const uint32_t* b, int n) {
  uint64_t borrow = 0;
  for (int i = 0; i < n; ++i) {
    uint64_t d = (uint64_t)a[i] - b[i] - borrow;
    a[i] = (uint32_t)d;
    borrow = (d >> 63) & 1;
  }
}
// a = 2a mod m, a < m, top word of m has headroom
inline void dbl_mod(uint32_t* a, const uint32_t* m, int n) {
  uint32_t c = 0;
  for (int i = 0; i < n; ++i) {
    uint32_t nc = a[i] >> 31;
    a[i] = (a[i] << 1) | c;
    c = nc;
  }
  if (cmp(a, m, n) >= 0) sub(a, m, n);
}

// 32-bit words -> radix-2^28 limbs
inline void to_limbs28(const uint32_t* w, int nwords, uint32_t* limbs, int nlimbs) {
  for (int j = 0; j < nlimbs; ++j) {
    uint32_t bit = 28u * j, wi = bit >> 5, sh = bit & 31;
    uint64_t v = 0;
    if ((int)wi < nwords) v = w[wi];
    if ((int)wi + 1 < nwords) v |= (uint64_t)w[wi + 1] << 32;
    limbs[j] = (uint32_t)(v >> sh) & MASK28;
  }
}

// Montgomery constants for an odd modulus n < 2^(28*nlimbs - 2).
//   r2[] = (2^(28*nlimbs))^2 mod n as limbs, n0inv = -n^-1 mod 2^28.  false if n is even or zero.
// This form doubles 1 up to 2^(2*28*nlimbs) a bit at a time (4,256 passes for a 2048-bit modulus: 0.25 ms); mont_setup below gives
// the same numbers in a few dozen microseconds and is what the library calls.  Kept as the definition the CPU suite checks it against.
inline bool mont_setup_by_doubling(const uint8_t* n_be, uint32_t n_len, int nlimbs, uint32_t* n_limbs, uint32_t* r2_limbs,
                                   uint32_t* n0inv) {
  const int nwords = (28 * nlimbs + 31) / 32 + 1;
  std::vector<uint32_t> n(nwords), r(nwords);
  from_be(n_be, n_len, n.data(), nwords);
  if ((n[0] & 1) == 0) return false;
  r.assign(nwords, 0);
  r[0] = 1;
  if (cmp(r.data(), n.data(), nwords) >= 0) { r[0] = 0; }  // n == 1
  for (int i = 0; i < 2 * 28 * nlimbs; ++i) dbl_mod(r.data(), n.data(), nwords);
  to_limbs28(n.data(), nwords, n_limbs, nlimbs);
  to_limbs28(r.data(), nwords, r2_limbs, nlimbs);
  uint32_t n0 = n[0], inv = n0;            // inv = n0^-1 mod 2^32 by Newton iteration
  for (int i = 0; i < 5; ++i) inv *= 2u - n0 * inv;
  *n0inv = (0u - inv) & MASK28;
  return true;
}

// ---- the same constants by Montgomery exponentiation of 2 on 64-bit words ----
// a = 2a mod m on k 64-bit words (a < m < 2^(64k - 1))
inline void dbl_mod64(uint64_t* a, const uint64_t* m, int k) {
  uint64_t c = 0;
  for (int i = 0; i < k; ++i) { const uint64_t nc = a[i] >> 63; a[i] = (a[i] << 1) | c; c = nc; }
  bool ge = true;
  for (int i = k - 1; i >= 0; --i) if (a[i] != m[i]) { ge = a[i] > m[i]; break; }
  if (ge) {
    uint64_t borrow = 0;
    for (int i = 0; i < k; ++i) {
      const unsigned __int128 d = (unsigned __int128)a[i] - m[i] - borrow;
      a[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
  }
}
// out = a * b / 2^(64k) mod m (CIOS; a, b < m < 2^(64k - 1), m odd, m0inv = -m^-1 mod 2^64); out < m; out may alias a or b
inline void montmul64(const uint64_t* a, const uint64_t* b, const uint64_t* m, uint64_t m0inv, int k, uint64_t* out) {
  std::vector<uint64_t> t((size_t)k + 2, 0);
  for (int i = 0; i < k; ++i) {
    unsigned __int128 c = 0;
    for (int j = 0; j < k; ++j) { c += (unsigned __int128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[k]; t[k] = (uint64_t)c; t[k + 1] = (uint64_t)(c >> 64);
    const uint64_t u = t[0] * m0inv;
    c = (unsigned __int128)u * m[0] + t[0];
    c >>= 64;
    for (int j = 1; j < k; ++j) { c += (unsigned __int128)u * m[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[k]; t[k - 1] = (uint64_t)c; c >>= 64;
    t[k] = t[k + 1] + (uint64_t)c;
  }
  bool ge = t[k] != 0;
  if (!ge) { ge = true; for (int i = k - 1; i >= 0; --i) if (t[i] != m[i]) { ge = t[i] > m[i]; break; } }
  if (ge) {
    uint64_t borrow = 0;
    for (int i = 0; i < k; ++i) {
      const unsigned __int128 d = (unsigned __int128)t[i] - m[i] - borrow;
      t[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
  }
  memcpy(out, t.data(), sizeof(uint64_t) * (size_t)k);
}

// Montgomery constants for an odd modulus n < 2^(28*nlimbs - 2): the numbers of mont_setup_by_doubling (tests/test_host_arith.py
// compares the two word for word), computed as 2^(2*28*nlimbs) mod n by square-and-double in a host-side Montgomery domain of
// R' = 2^(64k) just above n: R' mod n by doubling up from the modulus' top bit (at most 65 times), then 13 squarings.
inline bool mont_setup(const uint8_t* n_be, uint32_t n_len, int nlimbs, uint32_t* n_limbs, uint32_t* r2_limbs, uint32_t* n0inv) {
  const int nwords = (28 * nlimbs + 31) / 32 + 1;
  std::vector<uint32_t> n(nwords);
  from_be(n_be, n_len, n.data(), nwords);
  if ((n[0] & 1) == 0) return false;
  to_limbs28(n.data(), nwords, n_limbs, nlimbs);
  uint32_t n0 = n[0], inv = n0;            // inv = n0^-1 mod 2^32 by Newton iteration
  for (int i = 0; i < 5; ++i) inv *= 2u - n0 * inv;
  *n0inv = (0u - inv) & MASK28;
  int top = nwords - 1;
  while (top > 0 && n[top] == 0) --top;
  int bits = 32 * top + 32;
  while (!((n[top] >> ((bits - 1) & 31)) & 1u)) --bits;              // bit length of n (n is odd: never 0)
  const int k = (bits + 64) / 64;                                    // host domain R' = 2^(64k) just above n: n < 2^(64k - 1)
  if (bits == 1) { for (int j = 0; j < nlimbs; ++j) r2_limbs[j] = 0; return true; }      // n == 1: everything is 0
  // a modulus outside the contract (no caller passes one: they check the width first) gets whatever the defining form gives it
  if (bits > 28 * nlimbs - 2) return mont_setup_by_doubling(n_be, n_len, nlimbs, n_limbs, r2_limbs, n0inv);
  std::vector<uint64_t> m(k, 0), acc(k, 0), one(k, 0);
  for (int i = 0; i < nwords && i < 2 * k; ++i) m[i >> 1] |= (uint64_t)n[i] << (32 * (i & 1));
  uint64_t m0inv = m[0];                                             // m0^-1 mod 2^64
  for (int i = 0; i < 6; ++i) m0inv *= 2ull - m[0] * m0inv;
  m0inv = 0ull - m0inv;
  acc[(bits - 1) >> 6] = 1ull << ((bits - 1) & 63);                  // 2^(bits-1) < n
  for (int i = bits - 1; i < 64 * k; ++i) dbl_mod64(acc.data(), m.data(), k);     // R' mod n
  dbl_mod64(acc.data(), m.data(), k);                                // 2 in the host domain
  const uint32_t e = 2u * 28u * (uint32_t)nlimbs;
  int hb = 31;
  while (!((e >> hb) & 1u)) --hb;
  for (int b = hb - 1; b >= 0; --b) {
    montmul64(acc.data(), acc.data(), m.data(), m0inv, k, acc.data());
    if ((e >> b) & 1u) dbl_mod64(acc.data(), m.data(), k);
  }
  one[0] = 1;
  montmul64(acc.data(), one.data(), m.data(), m0inv, k, acc.data());  // out of the host domain: 2^e mod n
  std::vector<uint32_t> r(nwords, 0);
  for (int i = 0; i < nwords && i < 2 * k; ++i) r[i] = (uint32_t)(acc[i >> 1] >> (32 * (i & 1)));
  to_limbs28(r.data(), nwords, r2_limbs, nlimbs);
  return true;
}

// a = (a + b) mod m for a, b < m
inline void add_mod(uint32_t* a, const uint32_t* b, const uint32_t* m, int n) {
  uint64_t c = 0;
  for (int i = 0; i < n; ++i) { c += (uint64_t)a[i] + b[i]; a[i] = (uint32_t)c; c >>= 32; }
  if (c || cmp(a, m, n) >= 0) sub(a, m, n);
}
// r = a * b mod m (double-and-add), a, b < m
inline void mul_mod(const uint32_t* a, const uint32_t* b, const uint32_t* m, uint32_t* r, int n) {
  std::vector<uint32_t> acc(n, 0);
  for (int i = 32 * n - 1; i >= 0; --i) {
    dbl_mod(acc.data(), m, n);
    if ((b[i >> 5] >> (i & 31)) & 1u) add_mod(acc.data(), a, m, n);
  }
  memcpy(r, acc.data(), sizeof(uint32_t) * n);
}
// a mod m by shift-subtract (a arbitrary, n words)
inline void reduce(uint32_t* a, const uint32_t* m, int n) {
  std::vector<uint32_t> r(n, 0);
  for (int i = 32 * n - 1; i >= 0; --i) {
    dbl_mod(r.data(), m, n);
    if ((a[i >> 5] >> (i & 31)) & 1u) {
      uint64_t c = 1;
      for (int k = 0; k < n && c; ++k) { c += r[k]; r[k] = (uint32_t)c; c >>= 32; }
      if (cmp(r.data(), m, n) >= 0) sub(r.data(), m, n);
    }
  }
  memcpy(a, r.data(), sizeof(uint32_t) * n);
}
// out_limbs = x * 2^(28*nlimbs) mod m (Montgomery form of x), x given big-endian
inline void to_mont_limbs(const uint32_t* x_words, const uint32_t* m, int nwords, int nlimbs, uint32_t* out_limbs) {
  std::vector<uint32_t> t(x_words, x_words + nwords);
  for (int i = 0; i < 28 * nlimbs; ++i) dbl_mod(t.data(), m, nwords);
  to_limbs28(t.data(), nwords, out_limbs, nlimbs);
}

}  // namespace hostbn
}  // namespace bftkv
