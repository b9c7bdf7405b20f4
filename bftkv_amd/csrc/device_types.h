// Shared host/device plain-data types of the bftkv GPU verifier.
#pragma once
#include <stdint.h>

namespace bftkv {

// Per-packet outcome of one CheckDetachedSignature-equivalent step.  Numeric values are part of the
// C ABI (include/bftkv_gpu.h BFTKV_ST_*) and mirror oracle/openpgp.py ST_*.
enum SigStatus : uint8_t {
  ST_OK = 0,
  ST_UNKNOWN_ISSUER = 1,
  ST_PARSE_ERROR = 2,
  ST_NOT_SIGNATURE = 3,
  ST_NO_ISSUER = 4,
  ST_HASH_UNSUPPORTED = 5,
  ST_HASH_TAG = 6,
  ST_ALGO_MISMATCH = 7,
  ST_BAD_SIG = 8,
  ST_KEY_CANNOT_SIGN = 9,
  ST_UNSUPPORTED = 10,
  ST_NOT_EXAMINED = 11,    // behind the early exit of CollectiveSignature.Verify: the reference never reads this packet
  // internal, never returned:
  ST_PENDING_CHUNKED = 98, // signature packet with partial body lengths located: its chunks are linearised before the parse
  ST_PENDING_PARSE = 99,   // signature packet located, body not parsed yet
  ST_PENDING_HASH = 100,   // parsed, key found; digest not computed yet
  ST_PENDING_RSA = 101,    // digest + tag OK; waiting for the modexp
  ST_PENDING_DSA = 102,
};

constexpr int PK_RSA = 1, PK_RSA_ENCRYPT_ONLY = 2, PK_RSA_SIGN_ONLY = 3, PK_ELGAMAL = 16, PK_DSA = 17, PK_ECDSA = 19;
constexpr int HASH_MD5 = 1, HASH_SHA1 = 2, HASH_RIPEMD160 = 3, HASH_SHA256 = 8, HASH_SHA384 = 9,
              HASH_SHA512 = 10, HASH_SHA224 = 11;

// One record per packet event of an item's signature stream (48 bytes).
struct SigRec {
  uint64_t body_off;      // offset of the packet body (version byte for signatures) in the signature blob
  uint32_t body_len;
  uint32_t item;          // index of the item (write / reply) the packet belongs to
  int32_t key_slot;       // slot in the device key table, -1 if none
  uint32_t mpi_off[2];    // offsets of the MPI *value* bytes relative to body_off
  uint16_t mpi_bits[2];   // bit counts as written in the packet
  uint16_t hashed_len;    // length of the hashed-subpacket area (hash suffix = 6+hl body bytes + 6 trailer)
  uint8_t hash_tag[2];
  uint8_t pk_algo, hash_id, sig_type, status;
  uint8_t after_tag;      // status once the hash tag has matched; AFTER_TAG_PUBKEY: the public-key operation decides
  uint8_t flags;          // bit0: signature value may be >= 2^(8k) (no x-shortcut in the exponent ladder)
  uint8_t q_kind1;        // 1 + public-key work list the record belongs on (0: none) -- set by the parse, never by the hash stream
  uint8_t queued;         // the record has been put on that list (k_parse_body directly, or k_plan in two-phase calls)
  uint32_t pk_idx;        // index in the public-key work list
};
constexpr uint8_t AFTER_TAG_PUBKEY = 0xFF;

// Device key table (structure of arrays, [limb][key] would also do; [key][limb] keeps a key's
// limbs contiguous for the quad's 4x19 loads).
struct KeyTableDev {
  uint32_t n_keys;
  const uint64_t* key_id;     // [n_keys] 64-bit OpenPGP key id (primary or subkey)
  const uint64_t* sorted_id;  // [n_keys] key ids ascending (ties in table order): issuer lookup by bisection
  const uint32_t* sorted_slot;// [n_keys] table row of sorted_id[i]
  const uint32_t* entity;     // [n_keys] index of the owning entity (node) in the node table
  const uint8_t* pk_algo;     // [n_keys]
  const uint8_t* flags;       // [n_keys] bit0: usable for signing per KeysByIdUsage; bit1: CanSign(); bit2: primary key of its entity
  const uint32_t* mod_bits;   // [n_keys] RSA: bits(n); DSA: bits(p)
  const uint32_t* rsa_e;      // [n_keys]
  const uint32_t* n_limbs;    // [n_keys][76] modulus (RSA n / DSA p), radix 2^28
  const uint32_t* r2_limbs;   // [n_keys][76] R^2 mod n
  const uint32_t* r2_limbs80; // [n_keys][80] (2^2240)^2 mod n: the 8-lane form of k_rsa_modexp for small calls (<= 2048-bit RSA keys)
  const uint32_t* n0inv;      // [n_keys] -n^-1 mod 2^28
  // DSA keys only (zero otherwise)
  const uint32_t* q_words;    // [n_keys][8] subgroup order q, little-endian 32-bit words
  const uint32_t* q_bits;     // [n_keys]
  const uint32_t* dsa_tab;    // [n_keys][2][DSA_N_BIG] g*R, y*R mod p (Montgomery form; R = 2^2128, or 2^3136 for p beyond 2048 bits): seeds of the fixed-base tables
  // Fixed-base window tables, resident in HBM for as long as the key is known (k_dsa_build_comb):
  //   dsa_comb[slot][base in {g, y}][window w][digit d-1][E] = base^(d * 2^(wbits*w)) * R mod p,  d = 1 .. 2^wbits - 1
  // so g^u1 * y^u2 is at most 2 * 256/wbits - 1 table multiplications and no squarings (the entries of the window multiplied
  // by last, dsa_plain_window, are stored without the factor R: that product leaves the Montgomery domain).  The slot ends with
  //   q_pow28[E][10] = 2^(28 j) mod q as radix-2^28 limbs, which folds v (mod p) down to v mod q.
  // E = dsa_entry_limbs, the limbs of an entry, ONE value for the whole arena: 76 (4 lanes x 19) while every DSA key has p <= 2048
  // bits, 112 (8 lanes x 14: k_dsa_modexp<14, 8>) from the moment a key with p up to 3072 bits is known -- the smaller keys then
  // leave the upper 36 limbs of their entries unused (the arena restarts when E changes).
  const uint32_t* dsa_slot;   // [n_keys] table slot of a DSA key (0xFFFFFFFF otherwise)
  const uint32_t* dsa_comb;
  uint32_t dsa_wbits;         // 18 (2.39 GB per key), 16 (637 MB), 8 (4.96 MB) or 4 (0.58 MB per key, very large DSA keyrings)
  uint32_t dsa_entry_limbs;   // E above: DSA_N_SMALL or DSA_N_BIG
  uint32_t hash_policy;       // bits 0-1 MD5, bits 2-3 RIPEMD-160: 0 unknown (fenced), 1 available, 2 not available (bftkv_gpu_set_hash_policy)
};
// windows per 256-bit exponent: the top one is narrower when the width does not divide 256 (18 bits: 14 full windows + 4 bits;
// its table is laid out like the others, only its first 15 entries are ever read)
constexpr uint32_t dsa_nwin(uint32_t wbits) { return (256u + wbits - 1u) / wbits; }
constexpr uint32_t DSA_N_SMALL = 76, DSA_N_BIG = 112;        // limbs of p: 4 x 19 (<= 2048 bits), 8 x 14 (<= 3072 bits)
constexpr uint64_t dsa_comb_limbs_per_key(uint32_t wbits, uint32_t entry_limbs) {
  return 2ull * dsa_nwin(wbits) * ((1u << wbits) - 1u) * entry_limbs;
}
constexpr uint32_t dsa_qpow_words(uint32_t entry_limbs) { return entry_limbs * 10u; }
// ... followed by the mod-q Montgomery constants: 2^512 mod q (8 words), -q^-1 mod 2^32 (1 word), 3 words padding
constexpr uint32_t dsa_qtail_words(uint32_t entry_limbs) { return dsa_qpow_words(entry_limbs) + 12u; }
// the (base, window) whose entries are stored in plain instead of Montgomery form: the one k_dsa_modexp multiplies by last
constexpr bool dsa_plain_window(uint32_t base, uint32_t w, uint32_t nwin) { return base == 1u && w == nwin - 1u; }
constexpr uint64_t dsa_slot_stride(uint32_t wbits, uint32_t entry_limbs) { return dsa_comb_limbs_per_key(wbits, entry_limbs) + dsa_qtail_words(entry_limbs); }

constexpr uint8_t KEYF_USABLE_SIGN = 1, KEYF_CAN_SIGN = 2, KEYF_PRIMARY = 4;
// key only exists inside a request's certificate: invisible to keyring lookups, reachable when the lookup is
// restricted to its entity (PGPSignature.VerifyWithCertificate, crypto_pgp.go:332-344)
constexpr uint8_t KEYF_CERT_ONLY = 8;
// certificate key whose self-signature forbids data signatures (key flags without Sign): it still verifies the
// certificate's own self-signatures / bindings (sig_class != 0) but never a detached signature
constexpr uint8_t KEYF_CERT_CHECK_ONLY = 16;
// another, different key of the keyring carries the same 64-bit key id: the reference tries every candidate (the second one
// against a hash that has absorbed the suffix twice); the device takes the first and raises the item's fence flag
constexpr uint8_t KEYF_AMBIGUOUS = 32;
// per-item word item_hash_mask: bits 1..6 = midstates wanted besides SHA-256 (hash_info().idx: SHA-224, SHA-1, SHA-512,
// SHA-384, MD5, RIPEMD-160), bits 8..14 = midstates wanted over the CANONICAL TEXT form of the payload (text-mode
// signatures; bit 8 + idx, SHA-256 included), bit 31 = the item met a fenced input shape
constexpr uint32_t ITEM_FENCED = 0x80000000u;
constexpr uint32_t ITEM_TEXT_SHIFT = 8, ITEM_NEEDS_PAYLOAD = 0x7F7Eu;   // NEEDS_PAYLOAD: any hashing a midstate-only call cannot do

// Text-mode (signature type 0x01) hashing state per (hash, item): openpgp.NewCanonicalTextHash rewrites the line endings of
// the signed data, so these signatures hash a different byte stream than the binary ones of the same item.
struct TextDev {
  uint32_t* mid32;      // [5][n_items][8]   SHA-256 | SHA-224 | SHA-1 | MD5 | RIPEMD-160 state after the whole blocks of the canonical stream
  uint64_t* mid64;      // [2][n_items][8]   SHA-512 | SHA-384
  uint8_t* tail;        // [7][n_items][128] the bytes behind the last whole block, by hash_info().idx
  uint64_t* len;        // [7][n_items]      length of the canonical stream
};

// Quorum (wotq) on the device: up to MAX_QC cliques, membership as a byte table over entities.
constexpr int MAX_QC = 8;
struct QuorumDev {
  int32_t n_qcs;
  int32_t f[MAX_QC], min[MAX_QC], threshold[MAX_QC], suff[MAX_QC];
  const uint8_t* member;   // [n_qcs][n_entities]
  uint32_t n_entities;
};

// verdict bits (one byte per item)
constexpr uint8_t V_IS_QUORUM = 1, V_IS_THRESHOLD = 2, V_IS_SUFFICIENT = 4, V_REJECT = 8;

}  // namespace bftkv
