// Montgomery arithmetic for 2048/3072/4096-bit moduli on CDNA4 (gfx950) -- device code.
//
// Design (see DESIGN.md "k_rsa"):
//   * radix 2^28 limbs, N = 4L limbs with L = 19 / 28 / 37 limbs per lane for moduli up to 2048 / 3072 / 4096
//     bits (R = 2^(28N) > 4n, so values stay < 2n with NO conditional subtraction between multiplications);
//   * 4 lanes (one DPP quad) per big number, 16 numbers per wave64;
//   * lazy carries: every limb product is ONE v_mad_u64_u32 into a 64-bit column accumulator
//     (152 products of < 2^56.6 fit in 64 bits) -- measured on MI355X v_mad_u64_u32 issues at the
//     same rate as a plain 32-bit VALU op (tools/microbench), so carry handling, not the
//     multiplier, would otherwise dominate;
//   * row-wise (operand-scanning) Montgomery with a sliding window of 2L-1 columns per lane;
//     after L rows the window has slid by exactly one lane and is re-aligned with DPP row_shl:1;
//   * the broadcast operand a_i is read from LDS (same address across the quad = broadcast),
//     the per-row Montgomery factor m is computed by quad lane 0 and broadcast with DPP quad_perm.
// No MFMA: this is integer/modular work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bftkv {

constexpr int MONT_W = 28;
constexpr uint32_t MONT_MASK = (1u << MONT_W) - 1;
constexpr int MONT_TPI = 4;     // lanes per number
constexpr int MONT_L = 19;      // limbs per lane of the default (<= 2048-bit) instantiation
constexpr int MONT_N = MONT_TPI * MONT_L;  // 76 limbs = 2128 bits
constexpr int MONT_L3072 = 28, MONT_L4096 = 37;   // R = 2^3136 / 2^4144
constexpr int MONT_NMAX = MONT_TPI * MONT_L4096;  // 148 limbs: stride of per-key limb arrays

// DPP controls
constexpr int DPP_QUAD_BCAST0 = 0x00;          // quad_perm:[0,0,0,0]
constexpr int DPP_QUAD_SHR1 = 0x90;            // quad_perm:[0,0,1,2]  (lane l <- lane l-1, lane 0 keeps 0)
constexpr int DPP_QUAD_SWAP1 = 0xB1;           // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_SWAP2 = 0x4E;           // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_SHL1 = 0x101;            // lane i <- lane i+1 within a row of 16, 0 past the end

__device__ __forceinline__ uint32_t dpp_quad_bcast0(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_QUAD_BCAST0, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t dpp_quad_shr1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_QUAD_SHR1, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t dpp_row_shl1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHL1, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t quad_or(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_QUAD_SWAP1, 0xF, 0xF, false);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_QUAD_SWAP2, 0xF, 0xF, false);
  return v;
}

__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
  return (uint64_t)a * (uint64_t)b + c;   // -> v_mad_u64_u32
}

// out = a * b * R^-1 mod n, value < 2n, limbs "lazily normal": every limb <= 2^28.
//   a_lds : this number's N limbs of operand a in LDS (all 4 lanes of the quad pass the same pointer)
//   b, n  : this lane's L limbs (lane l of the quad holds limbs [l*L, l*L+L))
//   n0inv : -n^-1 mod 2^28
//   qlane : lane index within the quad (0..3)
template <int L>
__device__ __forceinline__ void mont_mul(uint32_t (&out)[L], const uint32_t* a_lds,
                                         const uint32_t (&b)[L], const uint32_t (&n)[L],
                                         uint32_t n0inv, int qlane) {
  // A column receives up to 2N = 8L products of < 2^56: 8L <= 255 fits 64 bits (L = 19, 28); beyond that
  // (L = 37) the live columns are carry-normalised at every block boundary.
  constexpr bool NORM = 8 * L > 255;
  uint64_t Q[2 * L - 1];
#pragma unroll
  for (int k = 0; k < L + (NORM ? 1 : 0); ++k) Q[k] = 0;

#pragma unroll 1
  for (int blk = 0; blk < MONT_TPI; ++blk) {
    const uint32_t* ap = a_lds + blk * L;
#pragma unroll
    for (int r = 0; r < L; ++r) {
      const uint32_t ai = ap[r];
      // Q[r+k] += a_i * b[k]
#pragma unroll
      for (int k = 0; k < L; ++k) {
        if (k == L - 1 && r > 0) Q[r + k] = mad64(ai, b[k], (NORM && r == 1) ? Q[r + k] : 0);   // first touch of a fresh column
        else Q[r + k] = mad64(ai, b[k], Q[r + k]);
      }
      // Montgomery factor from quad lane 0's column r
      uint32_t m = ((uint32_t)Q[r] * n0inv) & MONT_MASK;
      m = dpp_quad_bcast0(m);
#pragma unroll
      for (int k = 0; k < L; ++k) Q[r + k] = mad64(m, n[k], Q[r + k]);
      // retire column r: push its carry into column r+1 (value-preserving in every lane;
      // in quad lane 0 the low 28 bits are zero by construction, so the column dies)
      Q[r + 1] += Q[r] >> MONT_W;
      Q[r] &= MONT_MASK;
    }
    // window slid by L columns = one lane: re-align.  new Q[k] = own Q[L+k] + next lane's Q[k]
    // (quad lane 3 reads the next quad's lane 0, whose low columns are all zero; the last lane of
    //  a DPP row reads 0 through bound_ctrl).
#pragma unroll
    for (int k = 0; k < L; ++k) {
      uint32_t lo = dpp_row_shl1((uint32_t)Q[k]);
      uint32_t hi = dpp_row_shl1((uint32_t)(Q[k] >> 32));
      uint64_t nx = ((uint64_t)hi << 32) | lo;
      Q[k] = (k < L - 1) ? Q[L + k] + nx : nx;
    }
    if (NORM) {
      // carry-normalise the live columns; the top carry seeds column L (accumulated by the next first touch)
#pragma unroll
      for (int k = 0; k < L; ++k) {
        const uint64_t c = Q[k] >> MONT_W;
        Q[k] &= MONT_MASK;
        if (k < L - 1) Q[k + 1] += c; else Q[L] = c;
      }
    }
  }

  // lazy normalisation: local ripple, then one cross-lane carry hop with a 2-limb ripple.
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < L; ++k) {
    uint64_t v = Q[k] + c;
    out[k] = (uint32_t)v & MONT_MASK;
    c = v >> MONT_W;
  }
  if (NORM) c += Q[L];   // the top carry of the last block boundary belongs to the next lane's column 0
  uint32_t clo = dpp_quad_shr1((uint32_t)c), chi = dpp_quad_shr1((uint32_t)(c >> 32));
  uint64_t cin = (qlane == 0) ? 0 : (((uint64_t)chi << 32) | clo);
  uint64_t v0 = (uint64_t)out[0] + cin;
  out[0] = (uint32_t)v0 & MONT_MASK;
  uint32_t v1 = out[1] + (uint32_t)(v0 >> MONT_W);
  out[1] = v1 & MONT_MASK;
  out[2] += v1 >> MONT_W;
}

// Exact canonical form (every limb < 2^28) of a lazily-normal number (limbs <= 2^28 + small).
// Step s lets a carry hop from quad lane s-1 to lane s; lane l generates no new carry after
// step l, so TPI steps suffice.
template <int L>
__device__ __forceinline__ void canonicalize(uint32_t (&x)[L], int qlane) {
  uint32_t cout = 0;
#pragma unroll
  for (int step = 0; step < MONT_TPI; ++step) {
    uint32_t c = dpp_quad_shr1(cout);
    if (qlane == 0) c = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
      uint32_t v = x[k] + c;
      x[k] = v & MONT_MASK;
      c = v >> MONT_W;
    }
    cout = c;
  }
}

}  // namespace bftkv
