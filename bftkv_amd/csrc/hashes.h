// SHA-1 / SHA-224 / SHA-256 / SHA-384 / SHA-512 compression functions for gfx950 (device): the hashes
// golang.org/x/crypto/openpgp can be asked for on this path (s2k.HashIdToHash; MD5 and RIPEMD-160 are
// fenced).  Plain VALU work: 32-bit rotations lower to v_alignbit_b32, message schedules
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

// ---- per-hash parameters (OpenPGP hash ids) -----------------------------------------------------------
// slot: index of the per-item midstate array.  32-bit family: 0 SHA-256, 1 SHA-224, 2 SHA-1; 64-bit: 0 SHA-512, 1 SHA-384
struct HashInfo { int family; int slot; uint32_t dlen; uint32_t plen; };   // family 0: unsupported, 32: 64-byte blocks, 64: 128-byte blocks
__device__ __forceinline__ HashInfo hash_info(uint32_t hash_id) {
  switch (hash_id) {
    case 8: return {32, 0, 32, 19};    // SHA-256
    case 11: return {32, 1, 28, 19};   // SHA-224
    case 2: return {32, 2, 20, 15};    // SHA-1
    case 10: return {64, 0, 64, 19};   // SHA-512
    case 9: return {64, 1, 48, 19};    // SHA-384
    default: return {0, 0, 0, 0};      // MD5, RIPEMD-160: fenced as unavailable
  }
}

// DigestInfo prefixes (Go crypto/rsa hashPrefixes; the reference carries a copy at
// crypto/threshold/rsa/rsa.go:345-354), indexed by OpenPGP hash id.
__device__ __constant__ const uint8_t DI_SHA1[15] = {0x30, 0x21, 0x30, 0x09, 0x06, 0x05, 0x2b, 0x0e, 0x03, 0x02, 0x1a, 0x05, 0x00, 0x04, 0x14};
__device__ __constant__ const uint8_t DI_SHA2[19] = {0x30, 0x31, 0x30, 0x0d, 0x06, 0x09, 0x60, 0x86, 0x48, 0x01,
                                                     0x65, 0x03, 0x04, 0x02, 0x01, 0x05, 0x00, 0x04, 0x20};
// byte i (0 = first) of the DigestInfo prefix of hash_id
__device__ __forceinline__ uint32_t digestinfo_byte(uint32_t hash_id, uint32_t i) {
  if (hash_id == 2) return DI_SHA1[i];
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
