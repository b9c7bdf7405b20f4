// SHA-256 compression on the HOST, for one purpose: the micro-batcher's callers absorb the whole 64-byte blocks of their
// own signed payload on their own thread before they queue (batcher_capi.inc).  A payload's hash is a chain -- 134
// dependent compressions for the 8.6 KB write of a 64-replica quorum -- and one GPU lane walks it in ~0.43 ms whatever the
// batch size, which was the floor of every small device call; a host core with the SHA extensions walks it in ~5 us, and
// the callers do it in parallel.  The device receives the 32-byte midstate and the < 64 bytes behind it and finishes every
// signature's digest (k_digest_sha256) exactly as it does after k_sha256_mid.  Large resident batches keep hashing on
// the device, where 10,000 chains run side by side.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>
#endif

namespace hostsha {

static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

inline void init(uint32_t s[8]) {
  static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(s, iv, sizeof iv);
}

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

// FIPS 180-4 section 6.2.2, straight
inline void compress_portable(uint32_t s[8], const uint8_t* p, uint64_t nblk) {
  for (; nblk; --nblk, p += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
    for (int i = 0; i < 64; ++i) {
      const uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
      const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    s[0] += a; s[1] += b; s[2] += c; s[3] += d; s[4] += e; s[5] += f; s[6] += g; s[7] += h;
  }
}

#if defined(__x86_64__)
// SHA extensions (sha256rnds2 / sha256msg1 / sha256msg2): the state lives in two registers as ABEF / CDGH, four rounds per
// sha256rnds2 pair, the message schedule four words at a time.
__attribute__((target("sha,sse4.1,ssse3"))) inline void compress_shani(uint32_t s[8], const uint8_t* p, uint64_t nblk) {
  const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
  __m128i t = _mm_loadu_si128((const __m128i*)&s[0]);          // DCBA
  __m128i st1 = _mm_loadu_si128((const __m128i*)&s[4]);        // HGFE
  t = _mm_shuffle_epi32(t, 0xB1);                              // CDAB
  st1 = _mm_shuffle_epi32(st1, 0x1B);                          // EFGH
  __m128i st0 = _mm_alignr_epi8(t, st1, 8);                    // ABEF
  st1 = _mm_blend_epi16(st1, t, 0xF0);                         // CDGH
  for (; nblk; --nblk, p += 64) {
    const __m128i save0 = st0, save1 = st1;
    __m128i m[4];
    for (int i = 0; i < 4; ++i) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * i)), bswap);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      __m128i wk = _mm_add_epi32(m[r & 3], _mm_loadu_si128((const __m128i*)&K256[4 * r]));
      st1 = _mm_sha256rnds2_epu32(st1, st0, wk);
      if (r >= 3 && r <= 14) {
        // words 4(r+1) .. 4(r+1)+3: the sigma0 half was folded into m[r-3] two steps ago (msg1), now w[t-7] and sigma1
        const __m128i x = _mm_add_epi32(m[(r + 1) & 3], _mm_alignr_epi8(m[r & 3], m[(r + 3) & 3], 4));
        m[(r + 1) & 3] = _mm_sha256msg2_epu32(x, m[r & 3]);
      }
      wk = _mm_shuffle_epi32(wk, 0x0E);
      st0 = _mm_sha256rnds2_epu32(st0, st1, wk);
      if (r >= 1 && r <= 12) m[(r + 3) & 3] = _mm_sha256msg1_epu32(m[(r + 3) & 3], m[r & 3]);
    }
    st0 = _mm_add_epi32(st0, save0);
    st1 = _mm_add_epi32(st1, save1);
  }
  t = _mm_shuffle_epi32(st0, 0x1B);                            // FEBA
  st1 = _mm_shuffle_epi32(st1, 0xB1);                          // DCHG
  st0 = _mm_blend_epi16(t, st1, 0xF0);                         // DCBA
  st1 = _mm_alignr_epi8(st1, t, 8);                            // HGFE
  _mm_storeu_si128((__m128i*)&s[0], st0);
  _mm_storeu_si128((__m128i*)&s[4], st1);
}

inline bool have_shani() {
  static const int v = [] {
    unsigned a = 0, b = 0, c = 0, d = 0;
    if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return 0;
    const bool sha = (b >> 29) & 1u;
    if (!__get_cpuid(1, &a, &b, &c, &d)) return 0;
    const bool ssse3 = (c >> 9) & 1u, sse41 = (c >> 19) & 1u;
    return (sha && ssse3 && sse41) ? 1 : 0;
  }();
  return v != 0;
}
#else
inline bool have_shani() { return false; }
#endif

// mode: 0 = the fastest this CPU offers, 1 = the portable rounds (tests compare the two)
inline void compress(uint32_t s[8], const uint8_t* p, uint64_t nblk, int mode = 0) {
#if defined(__x86_64__)
  if (mode == 0 && have_shani()) { compress_shani(s, p, nblk); return; }
#endif
  compress_portable(s, p, nblk);
}

// state after the whole 64-byte blocks of [p, p+len): what k_sha256_mid leaves per item
inline void midstate(const uint8_t* p, uint64_t len, uint32_t s[8], int mode = 0) {
  init(s);
  compress(s, p, len >> 6, mode);
}

inline void digest(const uint8_t* p, uint64_t len, uint8_t out[32], int mode = 0) {
  uint32_t s[8];
  midstate(p, len, s, mode);
  uint8_t last[128];
  const uint64_t rem = len & 63;
  memset(last, 0, sizeof last);
  if (rem) memcpy(last, p + (len - rem), rem);
  last[rem] = 0x80;
  const uint64_t nb = rem + 9 > 64 ? 2 : 1, bits = len * 8;
  for (int i = 0; i < 8; ++i) last[nb * 64 - 1 - i] = (uint8_t)(bits >> (8 * i));
  compress(s, last, nb, mode);
  for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(s[i] >> 24); out[4 * i + 1] = (uint8_t)(s[i] >> 16); out[4 * i + 2] = (uint8_t)(s[i] >> 8); out[4 * i + 3] = (uint8_t)s[i]; }
}

}  // namespace hostsha
