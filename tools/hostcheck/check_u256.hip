// Host-side unit check of the mod-q arithmetic used by the DSA kernels (bftkv_amd/csrc/u256.h): the functions are
// __host__ __device__, so the exact code the GPU runs is exercised here on the CPU against straightforward references.
//   hipcc -O2 -std=c++17 tools/hostcheck/check_u256.hip -o /tmp/check_u256 && /tmp/check_u256
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <random>
#include "../../bftkv_amd/csrc/u256.h"
using namespace bftkv;

static U256 rnd(std::mt19937_64& g, int bits) {
  U256 r = u256_zero();
  for (int i = 0; i < 8; ++i) r.w[i] = (uint32_t)g();
  for (int i = 0; i < 8; ++i) { int lo = 32 * i; if (lo >= bits) r.w[i] = 0; else if (lo + 32 > bits) r.w[i] &= (1u << (bits - lo)) - 1u; }
  return r;
}
// reference inverse: the textbook binary extended GCD (nested loops)
static bool ref_modinv(const U256& s, const U256& q, U256& out) {
  U256 u = s, v = q, x1 = u256_zero(), x2 = u256_zero();
  x1.w[0] = 1;
  for (int guard = 0; guard < 4096; ++guard) {
    if (u256_is_one(u)) { out = x1; return true; }
    if (u256_is_one(v)) { out = x2; return true; }
    if (u256_is_zero(u) || u256_is_zero(v)) return false;
    while (!(u.w[0] & 1u)) { u256_shr1(u, 0); uint32_t c = 0; if (x1.w[0] & 1u) c = u256_add(x1, q); u256_shr1(x1, c); }
    while (!(v.w[0] & 1u)) { u256_shr1(v, 0); uint32_t c = 0; if (x2.w[0] & 1u) c = u256_add(x2, q); u256_shr1(x2, c); }
    if (u256_cmp(u, v) >= 0) { u256_sub(u, v); if (u256_sub(x1, x2)) u256_add(x1, q); }
    else { u256_sub(v, u); if (u256_sub(x2, x1)) u256_add(x2, q); }
  }
  return false;
}

int main() {
  std::mt19937_64 g(12345);
  long n_inv = 0, n_noinv = 0, bad = 0;
  const int sizes[] = {32, 33, 64, 160, 224, 255, 256};
  for (int t = 0; t < 60000; ++t) {
    const int qb = sizes[t % 7];
    U256 q = rnd(g, qb);
    q.w[0] |= 1u;                                   // odd
    q.w[(qb - 1) >> 5] |= 1u << ((qb - 1) & 31);    // exact bit length
    if (t % 11 == 0) { q = u256_zero(); q.w[0] = 3u * 5u * 7u * 11u * 13u * 17u * 19u * 23u + 0; q.w[0] |= 1u; }   // composite: non-invertible inputs exist
    U256 s = rnd(g, qb);
    while (u256_cmp(s, q) >= 0) u256_shr1(s, 0);
    if (u256_is_zero(s)) s.w[0] = 1;
    if (t % 13 == 0) { s = u256_zero(); s.w[0] = 1; }
    if (t % 17 == 0) { s = q; U256 one = u256_zero(); one.w[0] = 1; u256_sub(s, one); }   // q - 1
    U256 w1, w2;
    const bool ok1 = u256_modinv_odd(s, q, w1), ok2 = ref_modinv(s, q, w2);
    if (ok1 != ok2) { ++bad; continue; }
    if (!ok1) { ++n_noinv; continue; }
    ++n_inv;
    if (u256_cmp(w1, q) >= 0) ++bad;
    U256 p = u256_mulmod(w1, s, q);                 // w * s mod q == 1
    if (!u256_is_one(p) && !(u256_is_one(q))) ++bad;
    if (u256_cmp(w1, w2) != 0) ++bad;
  }
  // Montgomery product against double-and-add
  long bad_m = 0;
  for (int t = 0; t < 20000; ++t) {
    const int qb = sizes[t % 7];
    U256 q = rnd(g, qb); q.w[0] |= 1u; q.w[(qb - 1) >> 5] |= 1u << ((qb - 1) & 31);
    U256 a = rnd(g, qb), b = rnd(g, qb);
    while (u256_cmp(a, q) >= 0) u256_shr1(a, 0);
    while (u256_cmp(b, q) >= 0) u256_shr1(b, 0);
    uint32_t inv = 1; for (int it = 0; it < 5; ++it) inv *= 2u - q.w[0] * inv;
    U256 r2 = u256_zero(); r2.w[0] = 1;
    while (u256_cmp(r2, q) >= 0) u256_sub(r2, q);
    for (int i = 0; i < 512; ++i) { uint32_t c = u256_shl1(r2); if (c || u256_cmp(r2, q) >= 0) u256_sub(r2, q); }
    U256 m = u256_mulmod_mont(a, b, q, 0u - inv, r2), ref = u256_mulmod(a, b, q);
    if (u256_cmp(m, ref) != 0) ++bad_m;
  }
  printf("modinv: %ld invertible, %ld not, %ld mismatches; montmul mismatches %ld\n", n_inv, n_noinv, bad, bad_m);
  return (bad || bad_m) ? 1 : 0;
}
