// Montgomery arithmetic for 2048/3072/4096-bit moduli on CDNA4 (gfx950) -- device code.
//
// Design:
//   * radix 2^28 limbs, L limbs per lane, TPI lanes per number: 19 x 4 (<= 2048-bit moduli, R = 2^2128), 14 x 8 (<= 3072,
//     R = 2^3136), 19 x 8 (<= 4096, R = 2^4256), 10 x 8 (k_multiexp's form for small calls, R = 2^2240); R > 4n, so values
//     stay < 2n with NO conditional subtraction between multiplications;
//   * lazy carries: every limb product is ONE v_mad_u64_u32 into a 64-bit column accumulator (152 products of < 2^56.6 fit
//     in 64 bits).  Measured on MI355X (tools/microbench): a wave64 v_mad_u64_u32 holds a SIMD's issue port for 4.64 cycles,
//     a 64-bit add or shift for ~4.3, a plain 32-bit op for 2.3 -- the multiplier is barely dearer than a carry, so the loop
//     is written to issue as few non-MAC instructions as possible (5 per row, 2 per shifted column);
//   * row-wise (operand-scanning) Montgomery with a sliding window of 2L-1 columns per lane; after L rows the window has
//     slid by exactly one lane and is re-aligned with DPP row_shl:1;
//   * the broadcast operand a_i is read from LDS (same address across the group = broadcast), the per-row Montgomery
//     factor m is computed by group lane 0 and broadcast with DPP quad_perm, the limb mask riding on that move;
//   * squarings form one triangle of every limb-product block (mont_mul<.., SQR = true>).
// No MFMA: this is integer/modular work (DESIGN.md section 9 for the arithmetic of that decision).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bftkv {

constexpr int MONT_W = 28;
constexpr uint32_t MONT_MASK = (1u << MONT_W) - 1;
constexpr int MONT_TPI = 4;     // lanes per number
constexpr int MONT_L = 19;      // limbs per lane of the default (<= 2048-bit) instantiation
constexpr int MONT_N = MONT_TPI * MONT_L;  // 76 limbs = 2128 bits
// Larger moduli keep 14 / 19 limbs per lane and spread a number over EIGHT lanes instead (112 / 152 limbs, R = 2^3136 /
// 2^4256): 28 or 37 limbs per lane would need > 256 VGPRs and spill (measured: 15x slower than the MAC count predicts).
constexpr int MONT_TPI_BIG = 8;
constexpr int MONT_L3072 = 14, MONT_L4096 = 19;
constexpr int MONT_NMAX = MONT_TPI_BIG * MONT_L4096;  // 152 limbs: stride of per-key limb arrays

// DPP controls
constexpr int DPP_QUAD_BCAST0 = 0x00;          // quad_perm:[0,0,0,0]
constexpr int DPP_QUAD_SHR1 = 0x90;            // quad_perm:[0,0,1,2]  (lane l <- lane l-1, lane 0 keeps 0)
constexpr int DPP_QUAD_SWAP1 = 0xB1;           // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_SWAP2 = 0x4E;           // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_SHL1 = 0x101;            // lane i <- lane i+1 within a row of 16, 0 past the end

__device__ __forceinline__ uint32_t dpp_quad_bcast0(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_QUAD_BCAST0, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t dpp_quad_shr1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_QUAD_SHR1, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t dpp_row_shl1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHL1, 0xF, 0xF, true);
}
constexpr int DPP_ROW_SHR1 = 0x111;            // lane i <- lane i-1 within a row of 16, 0 before the start
constexpr int DPP_ROW_SHR4 = 0x114;            // lane i <- lane i-4 within a row of 16
__device__ __forceinline__ uint32_t dpp_row_shr1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR1, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t dpp_row_shr4(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR4, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t quad_or(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_QUAD_SWAP1, 0xF, 0xF, false);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_QUAD_SWAP2, 0xF, 0xF, false);
  return v;
}

// Group = the TPI (4 or 8) adjacent lanes that hold one number; glane = lane index inside the group.
template <int TPI>
__device__ __forceinline__ uint32_t grp_bcast0(uint32_t v, int glane) {   // the value of group lane 0, in every lane
  const uint32_t q = dpp_quad_bcast0(v);
  if constexpr (TPI == 4) return q;
  else { const uint32_t u = dpp_row_shr4(q); return (glane & 4) ? u : q; }   // upper quad takes the lower quad's
}
template <int TPI>
__device__ __forceinline__ uint32_t grp_shr1(uint32_t v) {                // lane i <- lane i-1; group lane 0: caller masks
  if constexpr (TPI == 4) return dpp_quad_shr1(v);
  else return dpp_row_shr1(v);
}
template <int TPI>
__device__ __forceinline__ uint32_t grp_or(uint32_t v) {
  v = quad_or(v);
  if constexpr (TPI == 8) v |= (uint32_t)__shfl_xor((int)v, 4);
  return v;
}

__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
  return (uint64_t)a * (uint64_t)b + c;   // -> v_mad_u64_u32
}

// out = a * b * R^-1 mod n, value < 2n, limbs "lazily normal": every limb <= 2^28.
//   a_lds : this number's N limbs of operand a in LDS (all 4 lanes of the quad pass the same pointer)
//   b, n  : this lane's L limbs (lane l of the quad holds limbs [l*L, l*L+L))
//   n0inv : -n^-1 mod 2^28
//   qlane : lane index within the group (0..TPI-1)
//
// SQR = true: a and b are the SAME number (b = the lane's own limbs of it, a_lds = all of it) and only half of the a*b
// products are formed.  View the product as TPI x TPI blocks of L x L limb products; in block step s, group lane l forms
// block (s, l) = rows of block s times its own limbs.  Blocks (I, J) and (J, I) hold the same products, and both sit at
// the same column position I + J of the sliding windows, so each of the two lanes does one TRIANGLE of the block: row r
// times limbs k >= r, doubled for k > r.  Lane J at step I then covers the pairs (r, k), r <= k, of block I x J, lane I at
// step J the pairs with the roles swapped, i.e. the other triangle; the diagonal r == k is formed once by each of them
// (total weight 2, as a cross term needs), and in a diagonal block (I == J) once in all (weight 1, a square term).
// Every lane executes the same triangle in every step -- no idle lanes, no operand movement -- and the column order the
// Montgomery rows rely on is unchanged (row r only touches columns >= 2r of its block).  190 instead of 361 limb products
// per block step: 2204 instead of 2888 MACs per lane and product.
template <int L, int TPI = MONT_TPI, bool SQR = false>
__device__ __forceinline__ void mont_mul(uint32_t (&out)[L], const uint32_t* a_lds,
                                         const uint32_t (&b)[L], const uint32_t (&n)[L],
                                         uint32_t n0inv, int qlane) {
  // A column receives up to 2N = 2*TPI*L products of < 2^56: up to 255 of them fit 64 bits (76 and 112 limbs); beyond
  // that (152 limbs) the live columns are carry-normalised at every block boundary.
  constexpr bool NORM = 2 * TPI * L > 255;
  uint32_t mask_v;   // the limb mask in a VGPR: lets the Montgomery factor's mask carry the quad broadcast (v_and_b32_dpp)
  asm("v_mov_b32 %0, 0xfffffff" : "=v"(mask_v));
  uint64_t Q[2 * L - 1];
#pragma unroll
  for (int k = 0; k < L + (NORM ? 1 : 0); ++k) Q[k] = 0;

#pragma unroll 1
  for (int blk = 0; blk < TPI; ++blk) {
    const uint32_t* ap = a_lds + blk * L;
#pragma unroll
    for (int r = 0; r < L; ++r) {
      const uint32_t ai = ap[r];
      const uint32_t ai2 = ai << 1;      // SQR: off-diagonal products count twice (limbs <= 2^28: the product stays < 2^57)
      // Q[r+k] += a_i * b[k]
#pragma unroll
      for (int k = SQR ? r : 0; k < L; ++k) {
        const uint32_t av = (SQR && k > r) ? ai2 : ai;
        if (k == L - 1 && r > 0) Q[r + k] = mad64(av, b[k], (NORM && r == 1) ? Q[r + k] : 0);   // first touch of a fresh column
        else Q[r + k] = mad64(av, b[k], Q[r + k]);
      }
      // Montgomery factor from quad lane 0's column r
      const uint32_t m = grp_bcast0<TPI>((uint32_t)Q[r] * n0inv, qlane) & mask_v;
#pragma unroll
      for (int k = 0; k < L; ++k) Q[r + k] = mad64(m, n[k], Q[r + k]);
      // retire column r: push its carry into column r+1.  What stays behind is the column's low 28 bits (zero in quad
      // lane 0 by construction); the mask is applied where they are next read, the window shift below.
      Q[r + 1] += Q[r] >> MONT_W;
    }
    // window slid by L columns = one lane: re-align.  new Q[k] = own Q[L+k] + next lane's Q[k]
    // (quad lane 3 reads the next quad's lane 0, whose low columns are all zero; the last lane of
    //  a DPP row reads 0 through bound_ctrl).
#pragma unroll
    for (int k = 0; k < L; ++k) {
      const uint64_t nx = dpp_row_shl1((uint32_t)Q[k]) & mask_v;   // the retired column's low 28 bits: v_and_b32_dpp
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
  uint32_t clo = grp_shr1<TPI>((uint32_t)c), chi = grp_shr1<TPI>((uint32_t)(c >> 32));
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
template <int L, int TPI = MONT_TPI>
__device__ __forceinline__ void canonicalize(uint32_t (&x)[L], int qlane) {
  uint32_t cout = 0;
#pragma unroll
  for (int step = 0; step < TPI; ++step) {
    uint32_t c = grp_shr1<TPI>(cout);
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

// y -= n when y >= n (y, n canonical; y < 2n): the exact residue of a value mont_mul left below n(1 + 2^-79).
// The top lane of the group compares its limbs first; only a wave holding a candidate (the top limbs of y not below those
// of n: about 2^-50 per number for a 2048-bit modulus) runs the subtraction, whose borrow hops lane to lane like
// canonicalize's carry.
template <int L, int TPI = MONT_TPI>
__device__ __forceinline__ void reduce_once(uint32_t (&y)[L], const uint32_t (&n)[L], int qlane) {
  uint32_t ge = 1;
#pragma unroll
  for (int k = 0; k < L; ++k) ge = (y[k] > n[k]) ? 1u : ((y[k] < n[k]) ? 0u : ge);
  if (!__any(ge != 0 && qlane == TPI - 1)) return;
  uint32_t d[L];
#pragma unroll
  for (int k = 0; k < L; ++k) d[k] = y[k] - n[k];            // two's complement, |.| < 2^28
  int32_t cout = 0, top = 0;
#pragma unroll
  for (int step = 0; step < TPI; ++step) {
    int32_t c = (int32_t)grp_shr1<TPI>((uint32_t)cout);
    if (qlane == 0) c = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
      const int32_t v = (int32_t)d[k] + c;
      d[k] = (uint32_t)v & MONT_MASK;
      c = v >> MONT_W;                                        // arithmetic shift: -1 = borrow
    }
    cout = c;
    top += c;                                                 // borrows that left this lane, over all steps
  }
  // the group's verdict is its top lane's: no borrow out of it <=> y >= n
  uint32_t keep = (top == 0) ? 1u : 0u;
  if constexpr (TPI == 4) keep = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)keep, 0xFF, 0xF, 0xF, true);   // quad_perm:[3,3,3,3]
  else keep = (uint32_t)__shfl((int)keep, (int)((threadIdx.x & 63u) | (uint32_t)(TPI - 1)));
  if (keep) {
#pragma unroll
    for (int k = 0; k < L; ++k) y[k] = d[k];
  }
}

}  // namespace bftkv
