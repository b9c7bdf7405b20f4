// 256-bit unsigned arithmetic for the mod-q side of DSA verification (one thread per signature):
// modular inverse by binary extended GCD (exact for ANY odd modulus, like math/big.ModInverse --
// no primality assumption), modular multiplication by 8-word Montgomery products (double-and-add kept as the reference form), Horner reduction of a
// 2128-bit radix-2^28 number.  Off the critical path: ~2 % of a DSA verification's work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bftkv {

struct U256 { uint32_t w[8]; };

__host__ __device__ __forceinline__ U256 u256_zero() { U256 r; for (int i = 0; i < 8; ++i) r.w[i] = 0; return r; }
__host__ __device__ __forceinline__ bool u256_is_zero(const U256& a) { uint32_t o = 0; for (int i = 0; i < 8; ++i) o |= a.w[i]; return o == 0; }
__host__ __device__ __forceinline__ bool u256_is_one(const U256& a) { uint32_t o = a.w[0] ^ 1u; for (int i = 1; i < 8; ++i) o |= a.w[i]; return o == 0; }
__host__ __device__ __forceinline__ int u256_cmp(const U256& a, const U256& b) {
  int r = 0;
  for (int i = 0; i < 8; ++i) { if (a.w[i] != b.w[i]) r = a.w[i] < b.w[i] ? -1 : 1; }
  return r;   // the most significant differing word wins (loop runs LSW -> MSW)
}
__host__ __device__ __forceinline__ uint32_t u256_add(U256& a, const U256& b) {
  uint64_t c = 0;
  for (int i = 0; i < 8; ++i) { c += (uint64_t)a.w[i] + b.w[i]; a.w[i] = (uint32_t)c; c >>= 32; }
  return (uint32_t)c;
}
__host__ __device__ __forceinline__ uint32_t u256_sub(U256& a, const U256& b) {
  uint64_t br = 0;
  for (int i = 0; i < 8; ++i) { uint64_t d = (uint64_t)a.w[i] - b.w[i] - br; a.w[i] = (uint32_t)d; br = (d >> 63) & 1; }
  return (uint32_t)br;
}
__host__ __device__ __forceinline__ void u256_shr1(U256& a, uint32_t top) {
  for (int i = 0; i < 7; ++i) a.w[i] = (a.w[i] >> 1) | (a.w[i + 1] << 31);
  a.w[7] = (a.w[7] >> 1) | (top << 31);
}
__host__ __device__ __forceinline__ uint32_t u256_shl1(U256& a) {
  uint32_t c = a.w[7] >> 31;
  for (int i = 7; i > 0; --i) a.w[i] = (a.w[i] << 1) | (a.w[i - 1] >> 31);
  a.w[0] <<= 1;
  return c;
}
__host__ __device__ __forceinline__ int u256_bits(const U256& a) {
  int b = 0;
  for (int i = 0; i < 8; ++i) if (a.w[i]) b = 32 * i + (32 - __builtin_clz(a.w[i]));
  return b;
}
// big-endian bytes -> U256; false when the value does not fit in 256 bits
__host__ __device__ __forceinline__ bool u256_from_be(const uint8_t* p, uint32_t len, U256& out) {
  out = u256_zero();
  bool fits = true;
  for (uint32_t i = 0; i < len; ++i) {
    uint32_t k = len - 1 - i;   // byte index from the LSB
    uint32_t b = p[i];
    if (k < 32) out.w[k >> 2] |= b << (8 * (k & 3));
    else if (b) fits = false;
  }
  return fits;
}
// (a + b) mod q for a, b < q
__host__ __device__ __forceinline__ void u256_addmod(U256& a, const U256& b, const U256& q) {
  uint32_t c = u256_add(a, b);
  if (c || u256_cmp(a, q) >= 0) u256_sub(a, q);
}
// a * b mod q, a < q (b arbitrary 256-bit)
__host__ __device__ __forceinline__ U256 u256_mulmod(const U256& a, const U256& b, const U256& q) {
  U256 r = u256_zero();
  for (int i = 255; i >= 0; --i) {
    uint32_t c = u256_shl1(r);
    if (c || u256_cmp(r, q) >= 0) u256_sub(r, q);
    if ((b.w[i >> 5] >> (i & 31)) & 1u) u256_addmod(r, a, q);
  }
  return r;
}
// Montgomery product a*b*2^-256 mod q for odd q, a, b < q  (q0inv = -q^-1 mod 2^32); result < q
__host__ __device__ __forceinline__ U256 u256_montmul(const U256& a, const U256& b, const U256& q, uint32_t q0inv) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { c += (uint64_t)a.w[j] * b.w[i] + t[j]; t[j] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[8] = (uint32_t)c; t[9] = (uint32_t)(c >> 32);
    const uint32_t m = t[0] * q0inv;
    c = ((uint64_t)m * q.w[0] + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; ++j) { c += (uint64_t)m * q.w[j] + t[j]; t[j - 1] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[7] = (uint32_t)c; t[8] = t[9] + (uint32_t)(c >> 32);
  }
  U256 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.w[i] = t[i];
  if (t[8] || u256_cmp(r, q) >= 0) u256_sub(r, q);
  return r;
}
// a * b mod q for a, b < q through two Montgomery products (r2 = 2^512 mod q)
__host__ __device__ __forceinline__ U256 u256_mulmod_mont(const U256& a, const U256& b, const U256& q, uint32_t q0inv, const U256& r2) {
  return u256_montmul(u256_montmul(a, b, q, q0inv), r2, q, q0inv);
}
// s^-1 mod q for odd q > 1 and 0 < s < q; false when gcd(s, q) != 1 (math/big.ModInverse returns nil).
// Binary extended GCD with ONE uniform step per iteration and no data-dependent branches, so the 64 signatures of a
// wave stay in lockstep (the textbook form with its nested while-loops ran every lane through every path: 0.93 ms for a
// pass that is 0.2 ms of instructions).  Invariants: x1*s = u, x2*s = v (mod q).  Per step the pair to shrink is chosen
// by selects -- u if u is even; else v if v is even; else the larger of the two, minus the other -- and halved; every
// step removes a bit from bits(u)+bits(v), so 512 steps always suffice; the loop stops as soon as u (or, on the device,
// every u of the wave) has reached 0.  gcd = v at that point.
__host__ __device__ __forceinline__ U256 u256_select(bool c, const U256& a, const U256& b) {
  U256 r;
  for (int i = 0; i < 8; ++i) r.w[i] = c ? a.w[i] : b.w[i];
  return r;
}
// (x / 2) mod q for odd q: (x + (x odd ? q : 0)) >> 1
__host__ __device__ __forceinline__ void u256_halfmod(U256& x, const U256& q) {
  const uint32_t odd = x.w[0] & 1u;
  uint64_t c = 0;
  for (int i = 0; i < 8; ++i) { c += (uint64_t)x.w[i] + (odd ? q.w[i] : 0u); x.w[i] = (uint32_t)c; c >>= 32; }
  u256_shr1(x, (uint32_t)c);
}
__host__ __device__ __forceinline__ bool u256_modinv_odd(const U256& s, const U256& q, U256& out) {
  U256 u = s, v = q, x1 = u256_zero(), x2 = u256_zero();
  x1.w[0] = 1;
  for (int it = 0; it < 512; ++it) {
    const bool u_zero = u256_is_zero(u);
#if defined(__HIP_DEVICE_COMPILE__)
    if (!__any(!u_zero)) break;
#else
    if (u_zero) break;
#endif
    const bool ue = !(u.w[0] & 1u), ve = !(v.w[0] & 1u);
    // d = u - v and e = v - u with the borrow telling which is larger
    U256 d = u, e = v;
    const bool u_lt_v = u256_sub(d, v) != 0;
    u256_sub(e, u);
    U256 dx = x1, ex = x2;                               // x1 - x2 and x2 - x1 (mod q)
    if (u256_sub(dx, x2)) u256_add(dx, q);
    if (u256_sub(ex, x1)) u256_add(ex, q);
    const bool upd_u = !u_zero && (ue || (!ve && !u_lt_v));          // shrink u: even, or both odd and u >= v
    const bool upd_v = !u_zero && !upd_u;                            // otherwise shrink v (even, or both odd and v > u)
    const bool sub_u = upd_u && !ue, sub_v = upd_v && !ve;           // the subtracting variants
    U256 nu = u256_select(sub_u, d, u), nx1 = u256_select(sub_u, dx, x1);
    U256 nv = u256_select(sub_v, e, v), nx2 = u256_select(sub_v, ex, x2);
    U256 hu = nu, hx1 = nx1, hv = nv, hx2 = nx2;
    u256_shr1(hu, 0); u256_halfmod(hx1, q);
    u256_shr1(hv, 0); u256_halfmod(hx2, q);
    u = u256_select(upd_u, hu, u); x1 = u256_select(upd_u, hx1, x1);
    v = u256_select(upd_v, hv, v); x2 = u256_select(upd_v, hx2, x2);
  }
  out = x2;
  return u256_is_zero(u) && u256_is_one(v);
}

}  // namespace bftkv
