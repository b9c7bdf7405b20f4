// SHA-1 / SHA-224 / SHA-256 / SHA-384 / SHA-512 / MD5 / RIPEMD-160 compression functions for gfx950 (device): the hashes
// golang.org/x/crypto/openpgp can be asked for on this path (s2k.HashIdToHash).  Whether MD5 and RIPEMD-160 are AVAILABLE
// in a bftkv binary depends on what it links (hashForSignature: "hash not available" otherwise), which this repository
// cannot establish: they are computed here but gated by a per-context policy (bftkv_gpu_set_hash_policy; default: fenced).  Plain VALU work: 32-bit rotations lower to v_alignbit_b32, message schedules
// live in 16 registers (fully unrolled rounds).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bftkv {

__device__ __constant__ const uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
// three-input XOR as ONE v_bitop3_b32 (truth table 0x96): the sigma functions are rot ^ rot ^ rot/shift, and the compiler
// only forms bitop3 for Ch / Maj by itself -- 4 of the ~27 VALU instructions of a SHA-256 round
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return (uint32_t)__builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint64_t xor3_64(uint64_t a, uint64_t b, uint64_t c) {
  return ((uint64_t)xor3((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32)) << 32) | xor3((uint32_t)a, (uint32_t)b, (uint32_t)c);
}

__device__ __forceinline__ void sha256_init(uint32_t (&s)[8]) {
  s[0] = 0x6a09e667; s[1] = 0xbb67ae85; s[2] = 0x3c6ef372; s[3] = 0xa54ff53a;
  s[4] = 0x510e527f; s[5] = 0x9b05688c; s[6] = 0x1f83d9ab; s[7] = 0x5be0cd19;
}

// One 64-byte block; w holds the 16 big-endian message words and is clobbered.
__device__ __forceinline__ void sha256_compress(uint32_t (&s)[8], uint32_t (&w)[16]) {
  uint32_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    if (i >= 16) {
      uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
      uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
    }
    uint32_t S1 = xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25));
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = h + S1 + ch + SHA256_K[i] + w[i & 15];
    uint32_t S0 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22));
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s[0] += a; s[1] += b; s[2] += c; s[3] += d; s[4] += e; s[5] += f; s[6] += g; s[7] += h;
}

__device__ __forceinline__ void sha224_init(uint32_t (&s)[8]) {
  s[0] = 0xc1059ed8; s[1] = 0x367cd507; s[2] = 0x3070dd17; s[3] = 0xf70e5939;
  s[4] = 0xffc00b31; s[5] = 0x68581511; s[6] = 0x64f98fa7; s[7] = 0xbefa4fa4;
}

// ---- SHA-1 ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotl(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }

__device__ __forceinline__ void sha1_init(uint32_t (&s)[8]) {
  s[0] = 0x67452301; s[1] = 0xefcdab89; s[2] = 0x98badcfe; s[3] = 0x10325476; s[4] = 0xc3d2e1f0;
  s[5] = s[6] = s[7] = 0;
}

__device__ __forceinline__ void sha1_compress(uint32_t (&s)[8], uint32_t (&w)[16]) {
  uint32_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4];
#pragma unroll
  for (int i = 0; i < 80; ++i) {
    if (i >= 16) w[i & 15] = rotl(xor3(w[(i + 13) & 15], w[(i + 8) & 15], w[(i + 2) & 15]) ^ w[i & 15], 1);
    uint32_t f, k;
    if (i < 20) { f = (b & c) | (~b & d); k = 0x5a827999; }
    else if (i < 40) { f = xor3(b, c, d); k = 0x6ed9eba1; }
    else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8f1bbcdc; }
    else { f = xor3(b, c, d); k = 0xca62c1d6; }
    uint32_t t = rotl(a, 5) + f + e + k + w[i & 15];
    e = d; d = c; c = rotl(b, 30); b = a; a = t;
  }
  s[0] += a; s[1] += b; s[2] += c; s[3] += d; s[4] += e;
}

// ---- SHA-512 / SHA-384 ---------------------------------------------------------------------------
__device__ __constant__ const uint64_t SHA512_K[80] = {
    0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull,
    0x59f111f1b605d019ull, 0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull,
    0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull, 0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull,
    0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull, 0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull,
    0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull, 0x983e5152ee66dfabull,
    0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
    0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull,
    0x53380d139d95b3dfull, 0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull,
    0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull, 0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull,
    0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull, 0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull,
    0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull, 0x5b9cca4f7763e373ull,
    0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
    0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull,
    0xd186b8c721c0c207ull, 0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull,
    0x113f9804bef90daeull, 0x1b710b35131c471bull, 0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull,
    0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull, 0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};

__device__ __forceinline__ uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

__device__ __forceinline__ void sha512_init(uint64_t (&s)[8]) {
  s[0] = 0x6a09e667f3bcc908ull; s[1] = 0xbb67ae8584caa73bull; s[2] = 0x3c6ef372fe94f82bull; s[3] = 0xa54ff53a5f1d36f1ull;
  s[4] = 0x510e527fade682d1ull; s[5] = 0x9b05688c2b3e6c1full; s[6] = 0x1f83d9abfb41bd6bull; s[7] = 0x5be0cd19137e2179ull;
}
__device__ __forceinline__ void sha384_init(uint64_t (&s)[8]) {
  s[0] = 0xcbbb9d5dc1059ed8ull; s[1] = 0x629a292a367cd507ull; s[2] = 0x9159015a3070dd17ull; s[3] = 0x152fecd8f70e5939ull;
  s[4] = 0x67332667ffc00b31ull; s[5] = 0x8eb44a8768581511ull; s[6] = 0xdb0c2e0d64f98fa7ull; s[7] = 0x47b5481dbefa4fa4ull;
}

// One 128-byte block; rounds rolled in groups of 16 to keep the code small (this hash is off the
// critical path: it only appears in certification / gpg-made signatures).
__device__ __forceinline__ void sha512_compress(uint64_t (&s)[8], uint64_t (&w)[16]) {
  uint64_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
#pragma unroll 1
  for (int r = 0; r < 80; r += 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int i = r + j;
      if (r > 0) {
        uint64_t w15 = w[(j + 1) & 15], w2 = w[(j + 14) & 15];
        uint64_t s0 = xor3_64(rotr64(w15, 1), rotr64(w15, 8), w15 >> 7);
        uint64_t s1 = xor3_64(rotr64(w2, 19), rotr64(w2, 61), w2 >> 6);
        w[j] = w[j] + s0 + w[(j + 9) & 15] + s1;
      }
      uint64_t S1 = xor3_64(rotr64(e, 14), rotr64(e, 18), rotr64(e, 41));
      uint64_t ch = (e & f) ^ (~e & g);
      uint64_t t1 = h + S1 + ch + SHA512_K[i] + w[j];
      uint64_t S0 = xor3_64(rotr64(a, 28), rotr64(a, 34), rotr64(a, 39));
      uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
      uint64_t t2 = S0 + mj;
      h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
  }
  s[0] += a; s[1] += b; s[2] += c; s[3] += d; s[4] += e; s[5] += f; s[6] += g; s[7] += h;
}

// ---- MD5 (RFC 1321) and RIPEMD-160: little-endian words, little-endian length ------------------------------------------
__device__ __constant__ const uint32_t MD5_T[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
    0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
    0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
    0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
    0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
    0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
__device__ __constant__ const uint8_t MD5_S[16] = {7, 12, 17, 22, 5, 9, 14, 20, 4, 11, 16, 23, 6, 10, 15, 21};

__device__ __forceinline__ void md5_init(uint32_t (&s)[8]) {
  s[0] = 0x67452301; s[1] = 0xefcdab89; s[2] = 0x98badcfe; s[3] = 0x10325476; s[4] = s[5] = s[6] = s[7] = 0;
}
// w: the 16 LITTLE-endian message words (rare path: rounds rolled, tables in constant memory)
__device__ __forceinline__ void md5_compress(uint32_t (&s)[8], const uint32_t (&w)[16]) {
  uint32_t a = s[0], b = s[1], c = s[2], d = s[3];
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
    uint32_t f, g;
    if (i < 16) { f = (b & c) | (~b & d); g = i; }
    else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
    else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
    else { f = c ^ (b | ~d); g = (7 * i) & 15; }
    uint32_t wv = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) wv = (g == (uint32_t)k) ? w[k] : wv;      // (register array: no dynamic indexing)
    const uint32_t t = a + f + MD5_T[i] + wv;
    const uint32_t sh = MD5_S[(i >> 4) * 4 + (i & 3)];
    a = d; d = c; c = b;
    b = b + ((t << sh) | (t >> (32 - sh)));
  }
  s[0] += a; s[1] += b; s[2] += c; s[3] += d;
}

__device__ __constant__ const uint8_t RMD_R1[80] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 7, 4, 13, 1, 10, 6, 15, 3, 12, 0, 9, 5, 2, 14, 11, 8,
                                                   3, 10, 14, 4, 9, 15, 8, 1, 2, 7, 0, 6, 13, 11, 5, 12, 1, 9, 11, 10, 0, 8, 12, 4, 13, 3, 7, 15, 14, 5, 6, 2,
                                                   4, 0, 5, 9, 7, 12, 2, 10, 14, 1, 3, 8, 11, 6, 15, 13};
__device__ __constant__ const uint8_t RMD_R2[80] = {5, 14, 7, 0, 9, 2, 11, 4, 13, 6, 15, 8, 1, 10, 3, 12, 6, 11, 3, 7, 0, 13, 5, 10, 14, 15, 8, 12, 4, 9, 1, 2,
                                                   15, 5, 1, 3, 7, 14, 6, 9, 11, 8, 12, 2, 10, 0, 4, 13, 8, 6, 4, 1, 3, 11, 15, 0, 5, 12, 2, 13, 9, 7, 10, 14,
                                                   12, 15, 10, 4, 1, 5, 8, 7, 6, 2, 13, 14, 0, 3, 9, 11};
__device__ __constant__ const uint8_t RMD_S1[80] = {11, 14, 15, 12, 5, 8, 7, 9, 11, 13, 14, 15, 6, 7, 9, 8, 7, 6, 8, 13, 11, 9, 7, 15, 7, 12, 15, 9, 11, 7, 13, 12,
                                                   11, 13, 6, 7, 14, 9, 13, 15, 14, 8, 13, 6, 5, 12, 7, 5, 11, 12, 14, 15, 14, 15, 9, 8, 9, 14, 5, 6, 8, 6, 5, 12,
                                                   9, 15, 5, 11, 6, 8, 13, 12, 5, 12, 13, 14, 11, 8, 5, 6};
__device__ __constant__ const uint8_t RMD_S2[80] = {8, 9, 9, 11, 13, 15, 15, 5, 7, 7, 8, 11, 14, 14, 12, 6, 9, 13, 15, 7, 12, 8, 9, 11, 7, 7, 12, 7, 6, 15, 13, 11,
                                                   9, 7, 15, 11, 8, 6, 6, 14, 12, 13, 5, 14, 13, 13, 7, 5, 15, 5, 8, 11, 14, 14, 6, 14, 6, 9, 12, 9, 12, 5, 15, 8,
                                                   8, 5, 12, 9, 12, 5, 14, 6, 8, 13, 6, 5, 15, 13, 11, 11};

__device__ __forceinline__ void ripemd160_init(uint32_t (&s)[8]) {
  s[0] = 0x67452301; s[1] = 0xefcdab89; s[2] = 0x98badcfe; s[3] = 0x10325476; s[4] = 0xc3d2e1f0; s[5] = s[6] = s[7] = 0;
}
__device__ __forceinline__ uint32_t rmd_f(int j, uint32_t x, uint32_t y, uint32_t z) {
  switch (j) {
    case 0: return x ^ y ^ z;
    case 1: return (x & y) | (~x & z);
    case 2: return (x | ~y) ^ z;
    case 3: return (x & z) | (y & ~z);
    default: return x ^ (y | ~z);
  }
}
__device__ __forceinline__ void ripemd160_compress(uint32_t (&s)[8], const uint32_t (&w)[16]) {
  const uint32_t K1[5] = {0x00000000, 0x5a827999, 0x6ed9eba1, 0x8f1bbcdc, 0xa953fd4e};
  const uint32_t K2[5] = {0x50a28be6, 0x5c4dd124, 0x6d703ef3, 0x7a6d76e9, 0x00000000};
  uint32_t a1 = s[0], b1 = s[1], c1 = s[2], d1 = s[3], e1 = s[4];
  uint32_t a2 = s[0], b2 = s[1], c2 = s[2], d2 = s[3], e2 = s[4];
  auto word = [&](uint32_t g) -> uint32_t {
    uint32_t wv = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) wv = (g == (uint32_t)k) ? w[k] : wv;
    return wv;
  };
#pragma unroll 1
  for (int i = 0; i < 80; ++i) {
    const int j = i >> 4;
    uint32_t t = a1 + rmd_f(j, b1, c1, d1) + word(RMD_R1[i]) + K1[j];
    uint32_t sh = RMD_S1[i];
    t = ((t << sh) | (t >> (32 - sh))) + e1;
    a1 = e1; e1 = d1; d1 = (c1 << 10) | (c1 >> 22); c1 = b1; b1 = t;
    t = a2 + rmd_f(4 - j, b2, c2, d2) + word(RMD_R2[i]) + K2[j];
    sh = RMD_S2[i];
    t = ((t << sh) | (t >> (32 - sh))) + e2;
    a2 = e2; e2 = d2; d2 = (c2 << 10) | (c2 >> 22); c2 = b2; b2 = t;
  }
  const uint32_t t = s[1] + c1 + d2;
  s[1] = s[2] + d1 + e2; s[2] = s[3] + e1 + a2; s[3] = s[4] + a1 + b2; s[4] = s[0] + b1 + c2; s[0] = t;
}

// ---- per-hash parameters (OpenPGP hash ids) -----------------------------------------------------------
// slot: index of the per-item midstate array.  32-bit family: 0 SHA-256, 1 SHA-224, 2 SHA-1, 3 MD5, 4 RIPEMD-160; 64-bit:
// 0 SHA-512, 1 SHA-384.  idx: the hash's bit in item_hash_mask and its row in the text-mode arrays (0 .. 6).  le: little-endian
// words and length (MD5, RIPEMD-160).
constexpr int N_MID32 = 5, N_MID64 = 2, N_HASHES = 7;
struct HashInfo { int family; int slot; uint32_t dlen; uint32_t plen; int idx; bool le; };   // family 0: unknown id, 32: 64-byte blocks, 64: 128-byte blocks
__device__ __forceinline__ HashInfo hash_info(uint32_t hash_id) {
  switch (hash_id) {
    case 8: return {32, 0, 32, 19, 0, false};    // SHA-256
    case 11: return {32, 1, 28, 19, 1, false};   // SHA-224
    case 2: return {32, 2, 20, 15, 2, false};    // SHA-1
    case 10: return {64, 0, 64, 19, 3, false};   // SHA-512
    case 9: return {64, 1, 48, 19, 4, false};    // SHA-384
    case 1: return {32, 3, 16, 18, 5, true};     // MD5          (gated: bftkv_gpu_set_hash_policy)
    case 3: return {32, 4, 20, 14, 6, true};     // RIPEMD-160   (gated)
    default: return {0, 0, 0, 0, 0, false};
  }
}

// DigestInfo prefixes (Go crypto/rsa hashPrefixes; the reference carries a copy at
// crypto/threshold/rsa/rsa.go:345-354), indexed by OpenPGP hash id.
__device__ __constant__ const uint8_t DI_SHA1[15] = {0x30, 0x21, 0x30, 0x09, 0x06, 0x05, 0x2b, 0x0e, 0x03, 0x02, 0x1a, 0x05, 0x00, 0x04, 0x14};
__device__ __constant__ const uint8_t DI_SHA2[19] = {0x30, 0x31, 0x30, 0x0d, 0x06, 0x09, 0x60, 0x86, 0x48, 0x01,
                                                     0x65, 0x03, 0x04, 0x02, 0x01, 0x05, 0x00, 0x04, 0x20};
__device__ __constant__ const uint8_t DI_MD5[18] = {0x30, 0x20, 0x30, 0x0c, 0x06, 0x08, 0x2a, 0x86, 0x48, 0x86, 0xf7, 0x0d, 0x02, 0x05, 0x05, 0x00, 0x04, 0x10};
// Go's RIPEMD-160 entry (the reference's copy: crypto/threshold/rsa/rsa.go:353) uses the ISO/IEC 10118-3 identifier -- not the
// TeleTrusT one gpg writes, so an RSA / RIPEMD-160 signature made by gpg does not verify under the reference
__device__ __constant__ const uint8_t DI_RMD160[14] = {0x30, 0x20, 0x30, 0x08, 0x06, 0x06, 0x28, 0xcf, 0x06, 0x03, 0x00, 0x31, 0x04, 0x14};
// byte i (0 = first) of the DigestInfo prefix of hash_id
__device__ __forceinline__ uint32_t digestinfo_byte(uint32_t hash_id, uint32_t i) {
  if (hash_id == 2) return DI_SHA1[i];
  if (hash_id == 1) return DI_MD5[i];
  if (hash_id == 3) return DI_RMD160[i];
  uint32_t b = DI_SHA2[i];
  // the SHA-2 prefixes differ in three bytes: total length, algorithm arc, digest length
  uint32_t dl = hash_id == 8 ? 0x20 : hash_id == 9 ? 0x30 : hash_id == 10 ? 0x40 : 0x1c;
  uint32_t arc = hash_id == 8 ? 1 : hash_id == 9 ? 2 : hash_id == 10 ? 3 : 4;
  if (i == 1) return 0x11 + dl;
  if (i == 14) return arc;
  if (i == 18) return dl;
  return b;
}

}  // namespace bftkv
