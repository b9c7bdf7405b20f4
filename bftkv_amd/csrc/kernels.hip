// HIP kernels of the bftkv batched quorum verifier (gfx950 / MI355X only).
//
//   k_walk<COUNT|FILL>    wave per item: speculative walk of the OpenPGP packet HEADERS of the signature stream
//                         (x/crypto packet.Read framing) -> one event per packet (scratch row, or SigRec when the row is full)
//   k_scan_counts         exclusive scan of per-item event counts; the total also goes to a pinned host mailbox
//   k_parse_body[_items]  per packet: Signature.parse (subpackets, MPIs) over an LDS window of the packet head,
//                         KeysByIdUsage lookup (bisection), every check that does not need the digest; names the work list
//   k_plan<1|2>           two-phase queueing of the public-key work up to the reference's early exit
//   k_sha256_mid          SHA-256 midstate of every item's signed payload, computed ONCE per item
//                         (the reference re-hashes the whole payload per signature,
//                          crypto/pgp/crypto_pgp.go:490); k_hash_mid_other: SHA-1/224/384/512 on demand
//   k_digest_sha256/other per signature: finish the hash with the hash suffix, hash-tag check
//   k_rsa_modexp<L,TPI>   s^e mod n by Montgomery ladder, 4 (8) lanes per signature (mont28.h), EMSA padding above the
//                         low 84 bytes checked in place; runs CONCURRENTLY with the hash kernels (separate HIP stream)
//   k_rsa_compare         low 84 bytes of EMSA-PKCS1-v1_5 from the digest against those of s^e mod n
//   k_dsa_inv/_mul/_modexp  dsa.Verify: s^-1 mod q and u2 on a third stream, u1 after the digests, g^u1 y^u2 mod p
//                         from HBM-resident fixed-base window tables (k_dsa_build_comb), v mod q == r in place
//   k_modexp              generic b^x mod n (corpus signing, threshold-RSA partial signatures)
//   k_tally               per item: wavefront ballots over the verified signers -> per-clique
//                         counts -> IsSufficient / IsThreshold / IsQuorum / Reject bits
//                         (quorum/wotqs/wotqs.go:144-185) and the early-exit position of
//                         PGPCollectiveSignature.Verify (crypto_pgp.go:485-500)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"
#include "mont28.h"
#include "u256.h"
#include "hashes.h"

namespace bftkv {

// ------------------------------------------------------------------------------------------------
// OpenPGP packet walk (oracle/openpgp.py next_packet / parse_signature_body are the spec)
// ------------------------------------------------------------------------------------------------
struct ParsedPacket {
  uint64_t next;       // stream position after this event
  bool event;          // false: silently skipped (unknown packet type)
  SigRec rec;
};

__host__ __device__ __forceinline__ bool known_tag(uint32_t tag) {
  // packet types packet.Read constructs; anything else is an UnknownPacketTypeError that
  // Reader.Next skips
  return tag < 32 && ((0x00066BFEu >> tag) & 1u);  // {1..9,11,13,14,17,18}
}

// Signature.parse.  An embedded-signature subpacket (type 32) makes x/crypto parse recursively; here the nesting depth is
// a template parameter (bounded at 2: deeper nesting raises the item's `fenced` flag), so the whole parser inlines
// into its kernels -- device-side recursion would mean real function calls with a scratch stack in the packet-parse kernel.
struct SubpacketState {
  bool have_ctime = false, have_issuer = false, have_embedded = false;
  bool too_deep = false;      // an embedded signature nested deeper than this parser goes: outcome not claimed (fence)
  uint64_t issuer = 0;
};
// The parsers below read a packet body through a byte view B: a plain pointer, or WinBytes -- the head of the body
// staged in this lane's LDS window by one round of wide loads (k_parse_body*), the rest still in global memory.  A
// thread-per-packet parse straight from memory is ~40 dependent byte loads, each touching 64 different cache lines per
// wave instruction; the signature header they walk is the first 30-60 bytes of the packet.
constexpr uint32_t PARSE_WIN = 64;              // bytes staged per packet
constexpr uint32_t PARSE_WIN_STRIDE = 68;       // LDS bytes per lane: 17 dwords, so the lanes' windows start in different banks
typedef const __attribute__((address_space(3))) uint8_t* lds_bytes_t;   // typed so that reads are ds_read_u8, not flat loads
struct WinBytes {
  const uint8_t* g;     // the bytes in global memory
  lds_bytes_t l;        // the same bytes [0, n) in LDS
  uint32_t n;
  __device__ __forceinline__ uint8_t operator[](uint32_t i) const { return i < n ? l[i] : g[i]; }
  __device__ __forceinline__ WinBytes operator+(uint32_t d) const {
    return d < n ? WinBytes{g + d, l + d, n - d} : WinBytes{g + d, l, 0u};
  }
};

__host__ __device__ __forceinline__ const uint8_t* raw_bytes(const uint8_t* p) { return p; }
__device__ __forceinline__ const uint8_t* raw_bytes(const WinBytes& w) { return w.g; }
struct ChainBytes;
__host__ __device__ inline ChainBytes raw_bytes(const ChainBytes& c);

// subpacket area walk; returns false on structural/unsupported error
template <int DEPTH, class B = const uint8_t*>
__host__ __device__ __forceinline__ bool parse_subpackets_t(B p, uint32_t len, bool hashed, SubpacketState& st);

template <int DEPTH, class B = const uint8_t*>
__host__ __device__ __forceinline__ bool parse_sig_body_t(B body, uint32_t blen, SigRec& rec, SubpacketState& st) {
  if (blen < 1) return false;
  if (body[0] != 4) return false;  // v3 handled by the caller, others unsupported
  if (blen < 6) return false;
  rec.sig_type = body[1];
  rec.pk_algo = body[2];
  rec.hash_id = body[3];
  if (!(rec.pk_algo == PK_RSA || rec.pk_algo == PK_RSA_SIGN_ONLY || rec.pk_algo == PK_DSA || rec.pk_algo == PK_ECDSA))
    return false;
  uint32_t h = rec.hash_id;
  if (!(h == 1 || h == 2 || h == 3 || (h >= 8 && h <= 11))) return false;
  uint32_t hl = ((uint32_t)body[4] << 8) | body[5];
  if (6 + hl > blen) return false;
  rec.hashed_len = (uint16_t)hl;
  if (!parse_subpackets_t<DEPTH, B>(body + 6u, hl, true, st)) return false;
  if (!st.have_ctime) return false;
  uint32_t p = 6 + hl;
  if (p + 2 > blen) return false;
  uint32_t ul = ((uint32_t)body[p] << 8) | body[p + 1];
  p += 2;
  if (p + ul > blen) return false;
  if (!parse_subpackets_t<DEPTH, B>(body + p, ul, false, st)) return false;
  p += ul;
  if (p + 2 > blen) return false;
  rec.hash_tag[0] = body[p];
  rec.hash_tag[1] = body[p + 1];
  p += 2;
  int n_mpi = (rec.pk_algo == PK_RSA || rec.pk_algo == PK_RSA_SIGN_ONLY) ? 1 : 2;
  rec.mpi_off[1] = 0;
  rec.mpi_bits[1] = 0;
  for (int i = 0; i < n_mpi; ++i) {
    if (p + 2 > blen) return false;
    uint32_t bits = ((uint32_t)body[p] << 8) | body[p + 1];
    uint32_t nb = (bits + 7) >> 3;
    p += 2;
    if (p + nb > blen) return false;
    rec.mpi_off[i] = p;
    rec.mpi_bits[i] = (uint16_t)bits;
    p += nb;
  }
  return true;
}

template <int DEPTH, class B>
__host__ __device__ __forceinline__ bool parse_subpackets_t(B a, uint32_t len, bool hashed, SubpacketState& st) {
  uint32_t p = 0;
  while (p < len) {
    uint32_t b = a[p], ln;
    if (b < 192) { ln = b; p += 1; }
    else if (b < 255) {
      if (p + 2 > len) return false;
      ln = ((b - 192) << 8) + a[p + 1] + 192; p += 2;
    } else {
      if (p + 5 > len) return false;
      ln = ((uint32_t)a[p + 1] << 24) | ((uint32_t)a[p + 2] << 16) | ((uint32_t)a[p + 3] << 8) | a[p + 4]; p += 5;
    }
    if (ln > len - p) return false;
    if (ln == 0) return false;
    uint32_t typ = a[p] & 0x7F;
    bool critical = (a[p] & 0x80) != 0;
    const B body = a + (p + 1);
    uint32_t bl = ln - 1;
    p += ln;
    switch (typ) {
      case 2:
        if (!hashed) return false;       // "signature creation time in non-hashed area"
        if (bl != 4) return false;
        st.have_ctime = true;
        break;
      case 3: case 9:
        if (!hashed) break;
        if (bl != 4) return false;
        break;
      case 11: case 21: case 22: case 30:
        break;
      case 16:
        if (bl != 8) return false;
        st.issuer = 0;
        for (int i = 0; i < 8; ++i) st.issuer = (st.issuer << 8) | body[i];
        st.have_issuer = true;
        break;
      case 25:
        if (!hashed) break;
        if (bl != 1) return false;
        break;
      case 27: case 29:
        if (!hashed) break;
        if (bl == 0) return false;
        break;
      case 32: {
        // embedded signature (the 0x19 cross-certification of a signing subkey, normally in the UNHASHED area): the
        // reference parses it from either area, refuses a second one and any type other than primary-key binding
        if (st.have_embedded) return false;                 // "Cannot have multiple embedded signatures"
        st.have_embedded = true;
        if constexpr (DEPTH >= 2) { st.too_deep = true; return false; }   // bounded nesting
        else {
          SigRec tmp; SubpacketState inner;
          const bool ok = parse_sig_body_t<DEPTH + 1>(raw_bytes(body), bl, tmp, inner);   // rare: straight from memory
          if (inner.too_deep) st.too_deep = true;
          if (!ok) return false;
          if (tmp.sig_type != 0x19) return false;           // "cross-signature has unexpected type"
        }
        break;
      }
      default:
        if (critical) return false;
    }
  }
  return true;
}

template <class B = const uint8_t*>
__host__ __device__ __forceinline__ bool parse_sig_body(B body, uint32_t blen, SigRec& rec, bool& have_issuer,
                                                        uint64_t& issuer, bool* too_deep = nullptr) {
  SubpacketState st;
  const bool ok = parse_sig_body_t<0, B>(body, blen, rec, st);
  have_issuer = st.have_issuer;
  issuer = st.issuer;
  if (too_deep) *too_deep = st.too_deep;
  return ok;
}

// SignatureV3.parse (RFC 4880 5.2.2; x/crypto openpgp/packet/signature_v3.go): version 2 or 3, one octet "5", signature type,
// creation time, 8-octet issuer key id, public-key and hash algorithm, 16-bit hash tag, MPIs.  The hashed material is the
// 5 bytes type || creation time (body[2..7)) with NO trailer -- rec.hashed_len stays 0 and SIGF_V3 tells the digest kernels.
constexpr uint8_t SIGF_LONG_VALUE = 1, SIGF_V3 = 2, SIGF_MORE_CANDIDATES = 4, SIGF_TEXT = 8;   // MORE: other keys share the issuer id (k_candidates)
template <class B = const uint8_t*>
__host__ __device__ __forceinline__ bool parse_sig_body_v3(B body, uint32_t blen, SigRec& rec, uint64_t& issuer) {
  if (blen < 1 || body[0] < 2 || body[0] > 3) return false;      // "signature packet version"
  if (blen < 19) return false;
  if (body[1] != 5) return false;                                 // "invalid hashed material length"
  rec.sig_type = body[2];
  issuer = 0;
  for (int i = 0; i < 8; ++i) issuer = (issuer << 8) | body[7 + i];
  rec.pk_algo = body[15];
  rec.hash_id = body[16];
  if (!(rec.pk_algo == PK_RSA || rec.pk_algo == PK_RSA_SIGN_ONLY || rec.pk_algo == PK_DSA)) return false;
  const uint32_t h = rec.hash_id;
  if (!(h == 1 || h == 2 || h == 3 || (h >= 8 && h <= 11))) return false;
  rec.hashed_len = 0;
  rec.hash_tag[0] = body[17];
  rec.hash_tag[1] = body[18];
  uint32_t p = 19;
  const int n_mpi = (rec.pk_algo == PK_DSA) ? 2 : 1;
  rec.mpi_off[1] = 0;
  rec.mpi_bits[1] = 0;
  for (int i = 0; i < n_mpi; ++i) {
    if (p + 2 > blen) return false;
    const uint32_t bits = ((uint32_t)body[p] << 8) | body[p + 1];
    const uint32_t nb = (bits + 7) >> 3;
    p += 2;
    if (p + nb > blen) return false;
    rec.mpi_off[i] = p;
    rec.mpi_bits[i] = (uint16_t)bits;
    p += nb;
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// Packet bodies as x/crypto's readers deliver them (openpgp/packet/packet.go readHeader, readLength, spanReader,
// partialLengthReader; the 4096-byte bufio.Reader of peekVersion).  PGPCollectiveSignature.Verify keeps ONE bytes.Reader
// and calls CheckDetachedSignature on it again and again: what a packet consumes is what those readers pulled.
//   definite length       spanReader: at most `length` bytes; a stream that ends early only matters to a parser that asks for more
//   partial lengths       partialLengthReader: 2^k-byte chunks, each followed by the next length header, the last one definite
//   indeterminate length  (old format, length type 3) the rest of the stream
// A signature body that parses is NOT drained afterwards (the reader stays where bufio's last fetch ended); one that fails is.
// ------------------------------------------------------------------------------------------------
constexpr uint8_t BODY_DEFINITE = 0, BODY_PARTIAL = 1, BODY_INDETERMINATE = 2;
constexpr uint32_t BUFIO_SIZE = 4096;
// Bounds on what is followed natively for partial-length packets (beyond them: ST_UNSUPPORTED + fence, the reference decides):
// a chain is walked by ONE lane, one dependent load per length header, and a chunked signature body is copied by one lane byte
// by byte -- a stream of a million one-byte chunks must not hold a whole batch for seconds.  Real chunked signatures (nobody
// writes them) would have a handful of chunks and a body of a few hundred bytes.
constexpr uint32_t CHAIN_MAX_HOPS = 1024;            // length headers followed per packet
constexpr uint32_t CHUNKED_SIG_MAX_BODY = 16384;     // bytes of a partial-length signature body linearised / parsed

// position of the next body byte in the item's stream, bytes left in the current chunk, another length header behind it?
struct ChunkCursor { uint64_t pos; uint64_t rem; bool partial; };

// partialLengthReader.Read's `for r.remaining == 0` loop: readLength at c.pos.  Returns false when the stream ends inside a
// length header (c.pos = end); true with rem == 0 means the body is over (io.EOF).
__host__ __device__ inline bool cursor_next_chunk(const uint8_t* base, ChunkCursor& c, uint64_t end) {
  while (c.rem == 0) {
    if (!c.partial) return true;
    if (c.pos >= end) return false;
    const uint32_t b = base[c.pos];
    if (b < 192) { c.rem = b; c.pos += 1; c.partial = false; }
    else if (b < 224) {
      if (c.pos + 2 > end) { c.pos = end; return false; }
      c.rem = ((uint64_t)(b - 192) << 8) + base[c.pos + 1] + 192; c.pos += 2; c.partial = false;
    } else if (b == 255) {
      if (c.pos + 5 > end) { c.pos = end; return false; }
      c.rem = ((uint64_t)base[c.pos + 1] << 24) | ((uint64_t)base[c.pos + 2] << 16) | ((uint64_t)base[c.pos + 3] << 8) | base[c.pos + 4];
      c.pos += 5; c.partial = false;
    } else { c.rem = 1ull << (b & 31u); c.pos += 1; }
  }
  return true;
}

// Everything a chain of chunks can deliver: `avail` body bytes, the reader behind the last of them at `next`; truncated: the
// stream ends inside a chunk or a length header (next = end).  gave_up: more than max_hops headers (speculating lanes of k_walk).
struct BodyExtent { uint64_t next, avail; bool truncated, gave_up; };
__host__ __device__ inline BodyExtent chain_extent(const uint8_t* base, ChunkCursor c, uint64_t end, uint32_t max_hops) {
  BodyExtent e{end, 0, false, false};
  for (uint32_t hop = 0;; ++hop) {
    const uint64_t room = end - c.pos, take = c.rem < room ? c.rem : room;
    e.avail += take; c.pos += take;
    if (take < c.rem) { e.truncated = true; return e; }
    c.rem = 0;
    if (!c.partial) { e.next = c.pos; return e; }
    if (hop >= max_hops) { e.gave_up = true; return e; }
    if (!cursor_next_chunk(base, c, end)) { e.truncated = true; return e; }
  }
}

// the deliverable bytes of a chain, one after the other, into dst[0, n)
__host__ __device__ inline void chain_copy(const uint8_t* base, ChunkCursor c, uint64_t end, uint8_t* dst, uint64_t n) {
  uint64_t done = 0;
  while (done < n) {
    if (c.rem == 0 && !cursor_next_chunk(base, c, end)) return;
    if (c.rem == 0) return;
    uint64_t take = c.rem < end - c.pos ? c.rem : end - c.pos;
    if (take > n - done) take = n - done;
    if (take == 0) return;
    for (uint64_t i = 0; i < take; ++i) dst[done + i] = base[c.pos + i];
    done += take; c.pos += take; c.rem -= take;
  }
}

// Byte view of a chunked body for the parsers (k_signers, which has no arena to linearise into).  The parsers read forwards
// almost always: the view remembers the chunk it stood in and walks on from there (back to the start for a smaller index).
struct ChainBytes {
  const uint8_t* base; ChunkCursor c0; uint64_t end; uint32_t skip;
  mutable ChunkCursor cur; mutable uint64_t cur_at; mutable bool cur_ok;      // `cur` stands on logical byte cur_at (if cur_ok)
  __host__ __device__ inline ChainBytes(const uint8_t* b, ChunkCursor c, uint64_t e, uint32_t s)
      : base(b), c0(c), end(e), skip(s), cur(c), cur_at(0), cur_ok(true) {}
  __host__ __device__ inline uint8_t operator[](uint32_t i) const {
    const uint64_t want = (uint64_t)skip + i;
    if (!cur_ok || want < cur_at) { cur = c0; cur_at = 0; cur_ok = true; }
    for (;;) {
      if (cur.rem == 0 && (!cursor_next_chunk(base, cur, end) || cur.rem == 0)) { cur_ok = false; return 0; }
      const uint64_t off = want - cur_at;
      if (off < cur.rem) return cur.pos + off < end ? base[cur.pos + off] : 0;
      cur_at += cur.rem; cur.pos += cur.rem; cur.rem = 0;
      if (cur.pos >= end) { cur_ok = false; return 0; }
    }
  }
  __host__ __device__ inline ChainBytes operator+(uint32_t d) const { ChainBytes r(*this); r.skip = skip + d; return r; }
};
__host__ __device__ inline ChainBytes raw_bytes(const ChainBytes& c) { return c; }

// bufio.Reader over the body readers, reduced to positions.  fetch(req) is ONE Read(req) of the underlying reader: never across
// a chunk boundary, never more than the stream holds.
struct FetchSim {
  ChunkCursor c; uint64_t end; uint64_t buffered; bool dry;
  __host__ __device__ inline uint64_t fetch(const uint8_t* base, uint64_t req) {
    if (c.rem == 0 && !cursor_next_chunk(base, c, end)) { dry = true; return 0; }
    uint64_t take = c.rem < end - c.pos ? c.rem : end - c.pos;
    if (take > req) take = req;
    if (take == 0) { dry = true; return 0; }
    c.pos += take; c.rem -= take;
    return take;
  }
  // readFull(r, n bytes) through bufio.Read: from the buffer while it holds something; an empty buffer is refilled by one
  // Read(4096), or -- for a request of at least the buffer size -- bypassed by one Read straight into the caller's slice
  __host__ __device__ inline void read_full(const uint8_t* base, uint64_t n) {
    while (n > 0 && !dry) {
      if (buffered == 0) {
        if (n >= BUFIO_SIZE) { n -= fetch(base, n); continue; }
        buffered = fetch(base, BUFIO_SIZE);
        if (buffered == 0) return;
      }
      const uint64_t take = n < buffered ? n : buffered;
      n -= take; buffered -= take;
    }
  }
};

// Where does the shared reader stand after packet.Read has PARSED this signature body?  Replays peekVersion's Peek(1) and the
// reads of Signature.parse (1, 5, hashed area, 2, unhashed area, 2, then 2 + n per MPI) / SignatureV3.parse (1, 1, 5, 8, 2, 2,
// MPIs); zero-length reads issue no Read call (io.ReadFull).
__host__ __device__ inline uint64_t reader_position_after_parse(const uint8_t* base, ChunkCursor body, uint64_t end, const SigRec& rec, bool v3) {
  FetchSim f{body, end, 0, false};
  f.buffered = f.fetch(base, BUFIO_SIZE);
  const int n_mpi = rec.mpi_bits[1] || rec.mpi_off[1] ? 2 : 1;
  // (small reads that follow one another without an area between them cannot empty the buffer in mid-read differently when
  // merged: bufio refills by the same Read(4096) either way -- so 1+1+5+8+2+2 is one read of 19, 1+5 one of 6)
  if (v3) f.read_full(base, 19);
  else {
    const uint64_t hl = rec.hashed_len, ul = (uint64_t)rec.mpi_off[0] - hl - 12;
    f.read_full(base, 6); f.read_full(base, hl); f.read_full(base, 2); f.read_full(base, ul); f.read_full(base, 2);
  }
  f.read_full(base, 2); f.read_full(base, ((uint64_t)rec.mpi_bits[0] + 7) >> 3);
  if (n_mpi == 2) { f.read_full(base, 2); f.read_full(base, ((uint64_t)rec.mpi_bits[1] + 7) >> 3); }
  return f.dry ? ~0ull : f.c.pos;
}

// One packet.Read framing step at stream position pos of [.., end).
struct WalkStep {
  uint64_t next;       // stream position after the packet, ALL of its body taken (a drained or fully fetched body)
  uint64_t body_off;   // first body byte (of the first chunk)
  uint32_t body_len;   // body bytes the readers can deliver (a stream that ends early delivers fewer than announced)
  bool event;          // false: silently skipped (unknown packet type)
  bool reads_to_end;   // known non-signature packet whose parser consumes the whole body (user id, user attribute, private key)
  bool gave_up;        // a chain of chunks longer than this (speculating) caller follows: nothing decided
  uint8_t status;      // ST_PENDING_PARSE / ST_PENDING_CHUNKED for signature packets, final otherwise
  uint8_t kind;        // BODY_*
};

// `hdr(i)` yields stream byte pos+i (i < 6): straight from memory on the host / generic path, from registers on the
// device fast path below.  `base`: the stream itself, for the length headers between the chunks of a partial-length body.
template <typename HDR>
__host__ __device__ __forceinline__ WalkStep walk_step(HDR hdr, const uint8_t* base, uint64_t pos, uint64_t end, uint32_t max_hops) {
  WalkStep r;
  r.event = true;
  r.reads_to_end = false;
  r.gave_up = false;
  r.status = ST_PARSE_ERROR;
  r.kind = BODY_DEFINITE;
  r.body_off = pos;
  r.body_len = 0;
  uint32_t b0 = hdr(0);
  if ((b0 & 0x80) == 0) { r.next = pos + 1; return r; }  // "tag byte does not have MSB set"
  uint32_t tag;
  uint64_t start, ln;
  if ((b0 & 0x40) == 0) {
    tag = (b0 & 0x3F) >> 2;
    uint32_t lt = b0 & 3;
    if (lt == 3) { r.kind = BODY_INDETERMINATE; start = pos + 1; ln = end - start; }   // contents = the stream itself
    else {
      uint32_t nb = 1u << lt;
      if (pos + 1 + nb > end) { r.next = end; return r; }
      ln = 0;
      for (uint32_t i = 0; i < nb; ++i) ln = (ln << 8) | hdr(1 + i);
      start = pos + 1 + nb;
    }
  } else {
    tag = b0 & 0x3F;
    if (pos + 1 >= end) { r.next = end; return r; }
    uint32_t b1 = hdr(1);
    if (b1 < 192) { ln = b1; start = pos + 2; }
    else if (b1 < 224) {
      if (pos + 2 >= end) { r.next = end; return r; }
      ln = ((b1 - 192) << 8) + hdr(2) + 192; start = pos + 3;
    } else if (b1 == 255) {
      if (pos + 6 > end) { r.next = end; return r; }
      ln = ((uint64_t)hdr(2) << 24) | ((uint64_t)hdr(3) << 16) | ((uint64_t)hdr(4) << 8) | hdr(5);
      start = pos + 6;
    } else { r.kind = BODY_PARTIAL; ln = 1ull << (b1 & 31u); start = pos + 2; }
  }
  uint64_t avail;
  if (r.kind == BODY_PARTIAL) {
    const BodyExtent e = chain_extent(base, ChunkCursor{start, ln, true}, end, max_hops < CHAIN_MAX_HOPS ? max_hops : CHAIN_MAX_HOPS);
    if (e.gave_up) {
      if (max_hops < CHAIN_MAX_HOPS) { r.gave_up = true; r.event = false; r.next = ~0ull; return r; }   // a speculating lane: nothing decided
      r.next = end; r.status = ST_UNSUPPORTED; return r;      // too many chunks to follow: the stream ends here, fenced
    }
    avail = e.avail; r.next = e.next;
  } else {
    const uint64_t room = end - start;
    avail = ln < room ? ln : room;            // the stream may end inside the body: only a parser that asks for more notices
    r.next = start + avail;
  }
  r.body_off = start;
  r.body_len = avail < 0xFFFFFFFFull ? (uint32_t)avail : 0xFFFFFFFFu;
  if (tag != 2) {
    // unknown type: UnknownPacketTypeError, the body drained, Reader.Next goes on.  Known type: an event, "non signature packet"
    if (known_tag(tag)) { r.status = ST_NOT_SIGNATURE; r.reads_to_end = tag == 13 || tag == 17 || tag == 5 || tag == 7; }
    else r.event = false;
    return r;
  }
  if (avail > 0xFFFFFF00ull) { r.status = ST_UNSUPPORTED; return r; }   // beyond the record's 32-bit fields: fenced
  if (r.kind == BODY_PARTIAL && avail > CHUNKED_SIG_MAX_BODY) { r.status = ST_UNSUPPORTED; return r; }   // (see the bounds above)
  r.status = r.kind == BODY_PARTIAL ? ST_PENDING_CHUNKED : ST_PENDING_PARSE;
  return r;
}

__host__ __device__ __forceinline__ WalkStep walk_next(const uint8_t* base, uint64_t pos, uint64_t end, uint32_t max_hops = 0xFFFFFFFFu) {
  return walk_step([&](uint32_t i) -> uint32_t { return base[pos + i]; }, base, pos, end, max_hops);
}

// Device walk: the whole header (<= 6 bytes) arrives with ONE memory round trip -- three independent aligned dword
// loads and a funnel shift -- instead of up to three dependent byte loads; the per-item walk is a chain of ~53 such
// steps and nothing but load latency.  Needs 12 readable bytes from the aligned address, else the byte path.
__device__ __forceinline__ WalkStep walk_next_dev(const uint8_t* base, uint64_t pos, uint64_t end, uint32_t max_hops = 0xFFFFFFFFu) {
  const uint64_t a = pos & ~3ull;
  if (a + 12 <= end && ((uintptr_t)base & 3u) == 0) {
    const uint32_t* wp = (const uint32_t*)(base + a);
    const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
    const uint32_t sh = (uint32_t)(pos & 3u) * 8u;
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh);     // bytes pos .. pos+3
    const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sh);     // bytes pos+4 .. pos+7
    return walk_step([&](uint32_t i) -> uint32_t { return ((i < 4 ? lo >> (8 * i) : hi >> (8 * (i - 4)))) & 0xFFu; }, base, pos, end, max_hops);
  }
  return walk_next(base, pos, end, max_hops);
}

// Per-item scratch of the counting pass: the first WALK_CAP packet events of an item as
// (offset of the body relative to the item's stream, body length, status), so that the fill pass is a
// parallel thread-per-packet expansion instead of a second sequential walk.  Items with more events
// (n = 256 cliques carry 171) fall back to the sequential k_walk<true>.
// (the row length is chosen per call by the host from the average stream length: WALK_CAP_MIN .. WALK_CAP_MAX)
constexpr uint32_t WALK_CAP_MIN = 32, WALK_CAP_MAX = 320;
constexpr uint32_t DIGEST_OTHER_MAX_BLOCKS = 4096;
struct WalkEnt { uint32_t body_rel; uint32_t body_len_status; };
// linearisation arena for partial-length signature bodies: 16-byte units per body (one spare: the parse window reads whole
// 16-byte granules behind the last byte)
__host__ __device__ constexpr uint32_t chunk_arena_units(uint32_t body_len) { return (body_len + 15u) / 16u + 4u; }   // len in the low 24 bits, status in the high 8

// One WAVE per item.  A packet stream is a chain -- the position of packet j+1 is known only after the header of
// packet j -- and walking it one packet per memory round trip made this kernel pure latency (53 dependent misses per
// write at n = 64).  The wave speculates instead: lane j parses a header at pos + j * (length of the last packet).
// Lane 0 always stands on a true boundary; lane j is confirmed when every lane before it is and lane j-1's packet ends
// exactly where lane j started.  Signature packets of one quorum are all the same size (DetachSign: 287 B for RSA-2048),
// so a whole item is normally confirmed in two rounds; any other stream just confirms fewer lanes per round (>= 1).
template <bool FILL>
__global__ void __launch_bounds__(64) k_walk(const uint8_t* __restrict__ sig_blob, const uint64_t* __restrict__ sig_off,
                                             uint32_t n_items, uint32_t* __restrict__ counts,
                                             const uint32_t* __restrict__ rec_base, SigRec* __restrict__ recs,
                                             uint8_t* __restrict__ item_flags, WalkEnt* __restrict__ scratch, uint32_t WALK_CAP,
                                             uint32_t* __restrict__ chunk_units /* counting pass: 16-byte units the parse will need to
                                                                                   linearise partial-length signature bodies */) {
  const uint32_t item = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  if (item >= n_items) return;
  if (FILL && counts[item] <= WALK_CAP && !(item_flags[item] & 2)) return;     // expanded in parallel by k_parse_body
  uint64_t pos = sig_off[item];
  const uint64_t end = sig_off[item + 1], pos0 = pos;
  uint32_t n = 0;
  const uint32_t base = FILL ? rec_base[item] : 0;
  uint64_t stride = 0;
  bool trailing_skip = false;   // silently skipped packet(s) after the last event
  bool force = false;           // an event that does not fit the scratch encoding: the fill pass must write this item
  bool unsup = false;           // a packet after which this walk does not know where the reference's reader stands: the item is fenced
  uint32_t units = 0;           // this lane's share of the item's linearisation arena
  while (pos < end) {
    const uint64_t p = pos + (uint64_t)lane * stride;
    const bool act = lane == 0 || (stride != 0 && p < end);
    WalkStep w;
    w.next = ~0ull; w.body_off = 0; w.body_len = 0; w.event = false; w.reads_to_end = false; w.gave_up = false; w.status = ST_PARSE_ERROR;
    w.kind = BODY_DEFINITE;
    // (a speculating lane follows a chain of partial-length chunks for a few headers only: garbage positions can spell long
    // chains; lane 0 stands on a true boundary and follows its packet to the end)
    if (act) w = walk_next_dev(sig_blob, p, end, lane == 0 ? 0xFFFFFFFFu : 4u);
    const uint32_t pn_lo = __shfl_up((uint32_t)w.next, 1), pn_hi = __shfl_up((uint32_t)(w.next >> 32), 1);
    const bool link = act && !w.gave_up && (lane == 0 || (((uint64_t)pn_hi << 32) | pn_lo) == p);
    const uint64_t broken = __builtin_amdgcn_ballot_w64(!link);
    const uint32_t n_conf = broken ? (uint32_t)__builtin_ctzll(broken) : 64u;   // >= 1: lane 0 always links
    const bool conf = lane < n_conf;
    const uint64_t evm = __builtin_amdgcn_ballot_w64(conf && w.event);
    const uint32_t idx = n + (uint32_t)__builtin_popcountll(evm & ((1ull << lane) - 1ull));
    // Where the reference's reader stands AFTER a packet is part of the semantics (CollectiveSignature.Verify keeps calling
    // CheckDetachedSignature on the same reader).  packet.Read drains a body on every error.  On success nothing is drained: a
    // known non-signature packet whose parser does not read to the end (literal data, compressed, encrypted, one-pass, key
    // packets with trailing bytes ...: everything but user id / user attribute / private key, which end in ReadAll) leaves the
    // reader INSIDE the body, and what follows is parsed out of the middle of it -- not followed here: the item is fenced.
    // (Signature packets: the parse decides, from what bufio fetched -- parse_one.)
    if (conf && w.event && (w.status == ST_UNSUPPORTED || (w.status == ST_NOT_SIGNATURE && !w.reads_to_end)))
      unsup = true;
    if (!FILL && conf && w.event && w.status == ST_PENDING_CHUNKED) units += chunk_arena_units(w.body_len);
    if (conf && w.event) {
      if (!FILL) {
        if (idx < WALK_CAP) {
          if (w.body_len < (1u << 24) && w.body_off - pos0 < (1ull << 32)) {
            WalkEnt e;
            e.body_rel = (uint32_t)(w.body_off - pos0);
            e.body_len_status = w.body_len | ((uint32_t)w.status << 24);
            scratch[(uint64_t)item * WALK_CAP + idx] = e;
          } else force = true;
        }
      } else {
        SigRec rec;
        rec.body_off = w.body_off; rec.body_len = w.body_len; rec.item = item; rec.key_slot = -1;
        rec.mpi_off[0] = rec.mpi_off[1] = 0; rec.mpi_bits[0] = rec.mpi_bits[1] = 0;
        rec.hashed_len = 0; rec.hash_tag[0] = rec.hash_tag[1] = 0;
        rec.pk_algo = rec.hash_id = rec.sig_type = 0; rec.status = w.status;
        rec.after_tag = 0; rec.flags = 0; rec.q_kind1 = rec.queued = 0; rec.pk_idx = 0xFFFFFFFFu;
        recs[base + idx] = rec;
      }
    }
    n += (uint32_t)__builtin_popcountll(evm);
    const uint32_t last = n_conf - 1;
    const uint64_t next = ((uint64_t)__shfl((uint32_t)(w.next >> 32), last) << 32) | __shfl((uint32_t)w.next, last);
    trailing_skip = !__shfl((int)w.event, last);
    stride = next - (pos + (uint64_t)last * stride);
    pos = next;
  }
  if (!FILL) {
    const bool any_force = __builtin_amdgcn_ballot_w64(force) != 0;
    const bool any_unsup = __builtin_amdgcn_ballot_w64(unsup) != 0;
    if (lane == 0) { counts[item] = n; item_flags[item] = (trailing_skip ? 1 : 0) | (any_force ? 2 : 0) | (any_unsup ? 4 : 0); }
    if (__builtin_amdgcn_ballot_w64(units != 0)) {      // rare: a partial-length signature packet in this item
      for (int d = 32; d >= 1; d >>= 1) units += __shfl_down(units, d);
      if (lane == 0 && chunk_units) atomicAdd(chunk_units, units);
    }
  }
}

// Wave-aggregated slot allocation: ONE atomic per wave and counter instead of one per lane (534k single-address
// atomics cost 6 ms; the compiler only aggregates by itself when the address is provably wave-uniform).
__device__ __forceinline__ uint32_t wave_alloc(uint32_t* counter, bool want) {
  const uint64_t m = __builtin_amdgcn_ballot_w64(want);
  if (m == 0) return 0;
  const uint32_t lane = __builtin_amdgcn_mbcnt_hi((uint32_t)(~0ull >> 32), __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int leader = __builtin_ctzll(m);
  uint32_t base = 0;
  if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
  return base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1));
}

__device__ __forceinline__ bool sig_class_ok(uint32_t cls, uint32_t sig_type) {
  if (cls == 0) return sig_type == 0x00;
  if (cls == 1) return sig_type >= 0x10 && sig_type <= 0x13;
  if (cls == 2) return sig_type == 0x18 || sig_type == 0x28;   // VerifyKeySignature: subkey binding or revocation, over key || subkey
  if (cls == 3) return sig_type == 0x19;                       // ... the embedded cross-signature of a signing subkey, same bytes
  if (cls == 4) return sig_type == 0x20;                       // VerifyRevocationSignature: over the key alone
  return false;
}

// KeysByIdUsage(id, KeyFlagSign) row filter: usable for signing (or, for certificate checks, for certification), inside the
// keyring the call verifies against (the node keyring, or the single entity of VerifyWithCertificate).
__device__ __forceinline__ bool key_is_candidate(const KeyTableDev& kt, uint32_t k, uint32_t only_ent, uint32_t cls) {
  const bool usable = (kt.flags[k] & KEYF_USABLE_SIGN) || (cls != 0 && (kt.flags[k] & KEYF_CERT_CHECK_ONLY));
  return usable && (only_ent == 0xFFFFFFFFu ? !(kt.flags[k] & KEYF_CERT_ONLY) : kt.entity[k] == only_ent);
}

// Per packet: Signature.parse + KeysByIdUsage + every VerifySignature check that precedes the math.
struct ParseArgs {
  const uint8_t* sig_blob; const uint64_t* sig_off; const uint32_t* rec_base; const uint32_t* counts; uint32_t n_items;
  const WalkEnt* scratch; uint32_t walk_cap; SigRec* recs; uint32_t n_recs; const uint32_t* cert_ent;
  uint32_t *pk_list, *pk_list3072, *pk_list4096;
  uint32_t* pk_count;          // [0] RSA<=2048, [1] DSA, [2] RSA<=3072, [3] RSA<=4096, [4] some hash other than SHA-256
  uint32_t* dsa_list; uint32_t* item_hash_mask;
  const uint8_t* sig_class;    // per item or null
  // per item or null, certificate checks only (sig_class >= 1): the id of the key the CALLER holds for this check -- x/crypto's
  // VerifyUserIdSignature / VerifyKeySignature / VerifyRevocationSignature verify with the key ReadEntity has in hand (the primary
  // key; the subkey for a cross-signature) and never look at the issuer subpacket.  Non-zero: that id replaces the signature's
  // issuer in the key lookup, and a signature without issuer subpacket is not an error.
  const uint64_t* forced_issuer;
  const uint32_t* msg_slot;    // per item or null
  const uint8_t* msg_hash;     // per item, with msg_slot
  const uint8_t* item_flags;
  uint32_t defer_queue;        // two-phase calls: k_plan decides which records join the work lists
  const uint32_t* n_recs_dev;  // staged calls: the record count lives on the device (n_recs is then the grid's upper bound)
  // partial-length signature bodies are linearised here before they are parsed (the kernels downstream read a body as
  // sig_blob + body_off + i): 16-byte units handed out by an atomic bump; no room => the packet is not claimed (fence)
  uint8_t* chunk_arena; uint32_t chunk_cap_units; uint32_t* chunk_bump;
};

// `win`: this lane's PARSE_WIN_STRIDE bytes of LDS.
__device__ __forceinline__ void parse_one(const ParseArgs& a, const KeyTableDev& kt, uint32_t ri, uint32_t item, uint32_t* win) {
  const uint8_t* __restrict__ sig_blob = a.sig_blob; const uint64_t* __restrict__ sig_off = a.sig_off;
  const uint32_t* __restrict__ rec_base = a.rec_base; const uint32_t* __restrict__ counts = a.counts;
  const WalkEnt* __restrict__ scratch = a.scratch; SigRec* __restrict__ recs = a.recs;
  const uint32_t* __restrict__ cert_ent = a.cert_ent;
  uint32_t* __restrict__ pk_list = a.pk_list; uint32_t* __restrict__ pk_list3072 = a.pk_list3072; uint32_t* __restrict__ pk_list4096 = a.pk_list4096;
  uint32_t* __restrict__ pk_count = a.pk_count; uint32_t* __restrict__ dsa_list = a.dsa_list; uint32_t* __restrict__ item_hash_mask = a.item_hash_mask;
  const uint8_t* __restrict__ sig_class = a.sig_class; const uint32_t* __restrict__ msg_slot = a.msg_slot;
  const uint8_t* __restrict__ msg_hash = a.msg_hash; const uint8_t* __restrict__ item_flags = a.item_flags;
  SigRec rec;
  const uint32_t WALK_CAP = a.walk_cap;
  if (counts[item] <= WALK_CAP && !(item_flags[item] & 2)) {
    const WalkEnt e = scratch[(uint64_t)item * WALK_CAP + (ri - rec_base[item])];
    rec.body_off = sig_off[item] + e.body_rel; rec.body_len = e.body_len_status & 0xFFFFFFu; rec.item = item; rec.key_slot = -1;
    rec.mpi_off[0] = rec.mpi_off[1] = 0; rec.mpi_bits[0] = rec.mpi_bits[1] = 0;
    rec.hashed_len = 0; rec.hash_tag[0] = rec.hash_tag[1] = 0;
    rec.pk_algo = rec.hash_id = rec.sig_type = 0; rec.status = (uint8_t)(e.body_len_status >> 24);
    rec.after_tag = 0; rec.flags = 0; rec.q_kind1 = rec.queued = 0; rec.pk_idx = 0xFFFFFFFFu;
    if (rec.status != ST_PENDING_PARSE && rec.status != ST_PENDING_CHUNKED) { recs[ri] = rec; return; }
  } else {
    rec = recs[ri];                                  // written by the sequential k_walk<true>
    if (rec.status != ST_PENDING_PARSE && rec.status != ST_PENDING_CHUNKED) return;
  }
  // the body as the reference's readers see it in the item's stream: one span, or a chain of partial-length chunks
  const uint64_t item_end = sig_off[item + 1];
  const bool chunked = rec.status == ST_PENDING_CHUNKED;
  ChunkCursor body_cur{rec.body_off, rec.body_len, false};
  uint64_t body_next = rec.body_off + rec.body_len;
  if (chunked) {
    body_cur = ChunkCursor{rec.body_off, 1ull << (sig_blob[rec.body_off - 1] & 31u), true};      // the first chunk's length octet
    body_next = chain_extent(sig_blob, body_cur, item_end, 0xFFFFFFFFu).next;
    const uint32_t units = chunk_arena_units(rec.body_len);
    const uint32_t at = a.chunk_arena ? atomicAdd(a.chunk_bump, units) : 0xFFFFFFFFu;
    if (!a.chunk_arena || at > a.chunk_cap_units || units > a.chunk_cap_units - at) {
      rec.status = ST_UNSUPPORTED;                   // nowhere to linearise it: not claimed
      atomicOr(&item_hash_mask[item], ITEM_FENCED);
      recs[ri] = rec;
      return;
    }
    uint8_t* const lin = a.chunk_arena + 16ull * at;
    chain_copy(sig_blob, body_cur, item_end, lin, rec.body_len);
    rec.body_off = (uint64_t)((uintptr_t)lin - (uintptr_t)sig_blob);   // addressed like every other body: sig_blob + body_off
    rec.status = ST_PENDING_PARSE;
  }
  // the head of the body: four 16-byte loads per lane (never past the end of the blob), parked in LDS
  const uint8_t* gbody = sig_blob + rec.body_off;
  const uint64_t room = chunked ? PARSE_WIN : sig_off[a.n_items] - rec.body_off;   // (a linearised body has 64 spare bytes behind it)
  const uint32_t nwin = (room < PARSE_WIN ? (uint32_t)room : PARSE_WIN) & ~15u;
#pragma unroll
  for (uint32_t j = 0; j < PARSE_WIN / 16; ++j) {
    if (16 * j < nwin) {
      uint32_t v[4];
      __builtin_memcpy(v, gbody + 16 * j, 16);
      win[4 * j] = v[0]; win[4 * j + 1] = v[1]; win[4 * j + 2] = v[2]; win[4 * j + 3] = v[3];
    }
  }
  const WinBytes body{gbody, (lds_bytes_t)win, nwin};
  uint8_t st;
  int q_kind = -1;             // public-key work list this record joins (decided below, queued at the end)
  // fence: the packet has a shape on which this library does not claim the reference's outcome (DESIGN.md "fenced inputs");
  // the item's fenced_out flag tells the caller to take the reference path for it
  bool fence = false;
  const bool v3 = rec.body_len >= 1 && body[0] < 4;                // packet.Read: version < 4 => *packet.SignatureV3
  if (v3 && msg_slot) { st = ST_UNSUPPORTED; fence = true; }       // transport messages: a v3 trailing signature stays fenced
  else {
    bool have_issuer = false, too_deep = false;
    uint64_t issuer = 0;
    bool parsed;
    if (v3) { parsed = parse_sig_body_v3(body, rec.body_len, rec, issuer); have_issuer = parsed; }
    else parsed = parse_sig_body(body, rec.body_len, rec, have_issuer, issuer, &too_deep);
    // A body that parses is not drained: the shared reader stays where bufio's last fetch ended.  Short of the packet's end (a
    // body beyond the 4096-byte buffer, unread chunks, bytes behind an indeterminate-length signature) the next call parses
    // packets out of the middle of this one -- not followed: fenced.  (One span of <= 4096 bytes is taken by the first fetch.)
    if (parsed && (chunked || rec.body_len > BUFIO_SIZE) && reader_position_after_parse(sig_blob, body_cur, item_end, rec, v3) != body_next)
      fence = true;
    if (parsed && a.forced_issuer && sig_class && sig_class[rec.item] != 0 && a.forced_issuer[rec.item] != 0) {
      issuer = a.forced_issuer[rec.item];
      have_issuer = true;
    }
    if (!parsed) { st = ST_PARSE_ERROR; fence = too_deep; }
    else if (!have_issuer && !msg_slot) st = ST_NO_ISSUER;
    else {
      // VerifyWithCertificate: the keyring is the single entity of the certificate (crypto_pgp.go:333)
      const uint32_t only_ent = cert_ent ? cert_ent[rec.item] : 0xFFFFFFFFu;
      // KeysByIdUsage(issuer, KeyFlagSign): first usable key with that id (ids are unique in the
      // device table -- the host de-duplicates identical material, bftkv_gpu_keyring_set)
      int32_t slot = -1;
      const uint32_t cls = sig_class ? sig_class[rec.item] : 0;
      // Signed message (openpgp.ReadMessage): the key was chosen from the ONE-PASS packet (md.SignedBy = keys[0]) and the
      // body was hashed with the one-pass packet's algorithm; the trailing signature only supplies suffix, tag and MPIs.
      const uint8_t sig_hash_id = rec.hash_id;
      if (msg_slot) { slot = (int32_t)msg_slot[rec.item]; rec.hash_id = msg_hash[rec.item]; }
      if (!msg_slot) {
        // bisect the sorted id index, then take the first row IN TABLE ORDER (ties are kept in that order) that qualifies
        uint32_t lo_i = 0, hi_i = kt.n_keys;
        while (lo_i < hi_i) {
          const uint32_t mid = (lo_i + hi_i) >> 1;
          if (kt.sorted_id[mid] < issuer) lo_i = mid + 1; else hi_i = mid;
        }
        // CheckDetachedSignature tries every candidate in turn (VerifySignature: CanSign first, then the hash suffix is
        // written into the SHARED hash, then tag, algorithm, arithmetic) and returns the first success or the last error.
        // A candidate that cannot sign writes nothing, so the first one that CAN is the only one that ever sees the true
        // digest: it takes the record through the pipeline.  What the candidates behind it do to the status of a record it
        // did not verify is settled by k_candidates at the end (they see the suffix twice or more and cannot succeed).
        int32_t first_any = -1;
        for (uint32_t i = lo_i; i < kt.n_keys && kt.sorted_id[i] == issuer; ++i) {
          const uint32_t k = kt.sorted_slot[i];
          if (!key_is_candidate(kt, k, only_ent, cls)) continue;
          if (first_any < 0) first_any = (int32_t)k;
          if (kt.flags[k] & KEYF_CAN_SIGN) { slot = (int32_t)k; break; }
        }
        if (slot < 0) slot = first_any;          // nobody can sign: ST_KEY_CANNOT_SIGN below
      }
      rec.key_slot = slot;
      const HashInfo hi = hash_info(rec.hash_id);
      const uint32_t hlen = hi.dlen, plen = hi.plen;
      if (slot >= 0 && !msg_slot && (kt.flags[slot] & KEYF_AMBIGUOUS)) rec.flags |= SIGF_MORE_CANDIDATES;   // other keys under this 64-bit id
      if (slot < 0) st = ST_UNKNOWN_ISSUER;
      // hashForSignature: binary (0x00) only for detached signatures (text 0x01: fenced).  Certificate checks
      // (sig_class[item] != 0) hash caller-prepared key||uid / key||subkey bytes and accept exactly the classes
      // openpgp.ReadEntity verifies: 1 = certification 0x10..0x13, 2 = subkey binding 0x18 / revocation 0x28, 3 = the 0x19
      // cross-signature embedded in a signing subkey's binding, 4 = key revocation 0x20 (sig_class_ok).
      else if (!msg_slot && !sig_class_ok(cls, rec.sig_type) && !(cls == 0 && rec.sig_type == 0x01)) st = ST_HASH_UNSUPPORTED;
      else if (hi.family == 0) st = ST_HASH_UNSUPPORTED;                                          // no such hash id (parse refuses them earlier)
      // MD5 / RIPEMD-160: hashForSignature fails with "hash not available" unless the binary links them -- a property of
      // the deployment this library cannot see.  Policy per context (bftkv_gpu_set_hash_policy): 0 unknown => fenced,
      // 1 available => verified like any other hash, 2 not available => the reference's error, no fence.
      else if (hi.le && ((kt.hash_policy >> (hi.idx == 5 ? 0 : 2)) & 3u) != 1u) {
        st = ST_HASH_UNSUPPORTED;
        fence = ((kt.hash_policy >> (hi.idx == 5 ? 0 : 2)) & 3u) == 0u;
      }
      else if (!(kt.flags[slot] & KEYF_CAN_SIGN)) st = ST_KEY_CANNOT_SIGN;  // checked before the hash is finished
      else {
        // everything below is only reached when the hash tag matches (k_digest decides)
        st = ST_PENDING_HASH;
        if (!msg_slot && cls == 0 && rec.sig_type == 0x01) {
          // text mode: the signed data is hashed in canonical form (k_hash_mid_text), per item and hash on demand
          rec.flags |= SIGF_TEXT;
          atomicOr(&item_hash_mask[rec.item], 1u << (ITEM_TEXT_SHIFT + hi.idx));
          pk_count[4] = 1u;
        } else if (rec.hash_id != HASH_SHA256) { atomicOr(&item_hash_mask[rec.item], 1u << hi.idx); pk_count[4] = 1u; }
        if (kt.pk_algo[slot] != rec.pk_algo) rec.after_tag = ST_ALGO_MISMATCH;
        else if ((rec.pk_algo == PK_RSA || rec.pk_algo == PK_RSA_SIGN_ONLY) && sig_hash_id != rec.hash_id)
          rec.after_tag = ST_BAD_SIG;   // rsa.VerifyPKCS1v15(sig.Hash, digest of another algorithm): length mismatch
        else if (rec.pk_algo == PK_RSA || rec.pk_algo == PK_RSA_SIGN_ONLY) {
          const uint32_t mod_bits = kt.mod_bits[slot];
          const uint32_t kbytes = (mod_bits + 7) >> 3;
          const uint32_t nb = (rec.mpi_bits[0] + 7u) >> 3;
          const WinBytes mp = body + (uint32_t)rec.mpi_off[0];
          uint32_t lead = 0;
          while (lead < nb && mp[lead] == 0) ++lead;
          const uint32_t vbytes = nb - lead;
          // size class of the modulus: 0 <= 2048 bits (76 limbs), 1 <= 3072 (112), 2 <= 4096 (152)
          const uint32_t cls_sz = mod_bits <= 2048 ? 0u : (mod_bits <= 3072 ? 1u : 2u);
          const uint32_t cap_bytes = (cls_sz == 0 ? MONT_N : cls_sz == 1 ? MONT_TPI_BIG * MONT_L3072 : MONT_TPI_BIG * MONT_L4096) * MONT_W / 8;
          if (mod_bits == 0xFFFFFFFFu) { rec.after_tag = ST_UNSUPPORTED; fence = true; }   // > 4096 bits or no Montgomery form
          else if (kbytes < hlen + plen + 11) rec.after_tag = ST_BAD_SIG;       // rsa.VerifyPKCS1v15: k < tLen+11
          else if (vbytes > cap_bytes) { rec.after_tag = ST_BAD_SIG; fence = true; }   // value >= R: the reference reduces it mod n
          else {
            rec.after_tag = AFTER_TAG_PUBKEY;
            rec.flags |= (vbytes > kbytes) ? SIGF_LONG_VALUE : 0;
            q_kind = cls_sz == 0 ? 0 : (int)cls_sz + 1;                         // [0] <=2048, [1] DSA, [2] <=3072, [3] <=4096
          }
        } else if (rec.pk_algo == PK_DSA) {
          if (kt.mod_bits[slot] == 0xFFFFFFFFu) { rec.after_tag = ST_UNSUPPORTED; fence = true; }   // key shape outside the kernels
          else {
            rec.after_tag = AFTER_TAG_PUBKEY;
            q_kind = 1;
          }
        } else { rec.after_tag = ST_UNSUPPORTED; fence = true; }   // ECDSA: out of scope
      }
    }
  }
  if (fence) atomicOr(&item_hash_mask[rec.item], ITEM_FENCED);
  if (v3) rec.flags |= SIGF_V3;
  rec.q_kind1 = (uint8_t)(q_kind + 1);
  if (!a.defer_queue) {
    // queue the public-key work: one atomic per wave and list
    uint32_t* const lists[4] = {pk_list, dsa_list, pk_list3072, pk_list4096};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t idx = wave_alloc(pk_count + k, q_kind == k);
      if (q_kind == k) { rec.pk_idx = idx; rec.queued = 1; lists[k][idx] = ri; }
    }
  }
  rec.status = st;
  recs[ri] = rec;
}

// Record-major grid (thread per packet, the item found by bisection over the scan): batches of short items -- single
// signatures, certificate checks, transport messages.
__global__ void __launch_bounds__(256) k_parse_body(ParseArgs a, KeyTableDev kt) {
  const uint32_t ri = blockIdx.x * blockDim.x + threadIdx.x;
  if (ri >= (a.n_recs_dev ? *a.n_recs_dev : a.n_recs)) return;
  // which item does record ri belong to?  largest item with rec_base[item] <= ri (and a non-empty range)
  uint32_t lo = 0, hi = a.n_items;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.rec_base[mid] <= ri) lo = mid; else hi = mid;
  }
  __shared__ uint32_t win_sh[256 * (PARSE_WIN_STRIDE / 4)];
  parse_one(a, kt, ri, lo, win_sh + threadIdx.x * (PARSE_WIN_STRIDE / 4));
}

// Item-major grid (block per item): collective signatures carry tens of packets per item, the block's lanes read one
// contiguous ss.Data and one scratch row, and nobody bisects.
constexpr int PARSE_ITEM_BLOCK = 128;
__global__ void __launch_bounds__(PARSE_ITEM_BLOCK) k_parse_body_items(ParseArgs a, KeyTableDev kt) {
  const uint32_t item = blockIdx.x;
  const uint32_t cnt = a.counts[item], base = a.rec_base[item];
  __shared__ uint32_t win_sh[PARSE_ITEM_BLOCK * (PARSE_WIN_STRIDE / 4)];
  for (uint32_t j = threadIdx.x; j < cnt; j += PARSE_ITEM_BLOCK) parse_one(a, kt, base + j, item, win_sh + threadIdx.x * (PARSE_WIN_STRIDE / 4));
}

// PGPSignature.Signers (crypto_pgp.go:373-390): parse-only walk, issuers looked up among PRIMARY
// key ids (getCertById, :206-219); the walk ends at the first Reader.Next error.
template <bool FILL>
__global__ void __launch_bounds__(64) k_signers(const uint8_t* __restrict__ sig_blob, const uint64_t* __restrict__ sig_off,
                                                uint32_t n_items, uint32_t* __restrict__ counts,
                                                const uint32_t* __restrict__ out_base, uint64_t* __restrict__ ids_out,
                                                KeyTableDev kt, uint8_t* __restrict__ fenced /*[n_items] or null; written by the counting pass*/) {
  uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= n_items) return;
  uint64_t pos = sig_off[item], end = sig_off[item + 1];
  uint32_t n = 0;
  uint32_t base = FILL ? out_base[item] : 0;
  // fence: the walk met a shape on which this library does not follow the reference's reader (partial / indeterminate
  // lengths; a packet after which the reader stands inside the body, see k_walk; a v4 signature without issuer, on which the
  // reference dereferences nil) -- the caller takes the reference path for the item
  bool fence = false;
  while (pos < end) {
    WalkStep w = walk_next_dev(sig_blob, pos, end);
    pos = w.next;
    if (!w.event) continue;                               // unknown packet type: skipped by Next
    if (w.status == ST_UNSUPPORTED || (w.status == ST_NOT_SIGNATURE && !w.reads_to_end)) fence = true;
    if (w.status == ST_NOT_SIGNATURE) continue;           // other packet types fall through the type switch
    if (w.status != ST_PENDING_PARSE && w.status != ST_PENDING_CHUNKED) break;   // framing error => Next returns err => loop ends
    SigRec tmp;
    bool have_issuer = false, too_deep = false, parsed, v3;
    uint64_t issuer = 0;
    ChunkCursor cur{w.body_off, w.body_len, false};
    if (w.status == ST_PENDING_CHUNKED) {
      // partial body lengths: parsed through a view that walks the chunks (no arena in this parse-only kernel)
      cur = ChunkCursor{w.body_off, 1ull << (sig_blob[w.body_off - 1] & 31u), true};
      const ChainBytes body(sig_blob, cur, end, 0u);
      v3 = w.body_len >= 1 && body[0] < 4;
      parsed = v3 ? parse_sig_body_v3(body, w.body_len, tmp, issuer) : parse_sig_body(body, w.body_len, tmp, have_issuer, issuer, &too_deep);
    } else {
      const uint8_t* body = sig_blob + w.body_off;
      v3 = w.body_len >= 1 && body[0] < 4;
      parsed = v3 ? parse_sig_body_v3(body, w.body_len, tmp, issuer) : parse_sig_body(body, w.body_len, tmp, have_issuer, issuer, &too_deep);
    }
    if (!parsed) { fence |= too_deep; break; }            // parse error => Next returns err (the body drained)
    // parsed: the reader stays where bufio's last fetch ended (parse_one) -- short of the packet's end: not followed
    if ((w.status == ST_PENDING_CHUNKED || w.body_len > BUFIO_SIZE) && reader_position_after_parse(sig_blob, cur, end, tmp, v3) != w.next)
      fence = true;
    if (v3) continue;                                     // SignatureV3 is a different Go type: no case of the switch
    if (!have_issuer) { fence = true; break; }            // nil dereference in the reference: fenced
    for (uint32_t k = 0; k < kt.n_keys; ++k) {
      if (kt.key_id[k] == issuer && (kt.flags[k] & KEYF_PRIMARY) && !(kt.flags[k] & KEYF_CERT_ONLY)) {
        if (FILL) ids_out[base + n] = issuer;
        ++n;
        break;
      }
    }
  }
  if (!FILL) { counts[item] = n; if (fenced) fenced[item] = fence ? 1 : 0; }
}

// single-block exclusive scan (n up to a few million): each thread scans a contiguous chunk
// `host_total` (optional) is a mapped, pinned host word: the host spins on it instead of paying an interrupt-driven
// stream synchronisation for the one number it needs (arena and grid sizes) in the middle of the pipeline.
// The kernel also clears what the parse accumulates into (the 24 work-list words, the per-item hash masks): two memset
// launches fewer per call.
// `cap` (staged small calls, which size their arena from the stream length instead of asking the host in mid-pipeline): a
// call with more packet events than that becomes an EMPTY call -- every count zero, total[1] = the real total -- and the
// caller runs it again through the ordinary entry point.
__global__ void __launch_bounds__(1024) k_scan_counts(uint32_t* __restrict__ counts, uint32_t n,
                                                      uint32_t* __restrict__ base, uint32_t* __restrict__ total,
                                                      uint32_t* __restrict__ host_total, uint32_t* __restrict__ pk_count /*[24] or null*/,
                                                      uint32_t* __restrict__ item_hash_mask /*[n] or null*/, uint32_t cap,
                                                      uint32_t* __restrict__ chunk_ctr /*[4] or null: [0] arena units the walk asked
                                                        for (zeroed here for the next call), [1] the parse's bump, [2] copy of [0]*/) {
  __shared__ uint32_t part[1024];
  uint32_t t = threadIdx.x;
  uint32_t chunk = (n + 1023) / 1024;
  uint32_t lo = t * chunk, hi = min(n, lo + chunk);
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; ++i) s += counts[i];
  part[t] = s;
  if (pk_count && t < 24) pk_count[t] = 0;
  if (item_hash_mask) for (uint32_t i = lo; i < hi; ++i) item_hash_mask[i] = 0;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    uint32_t v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  const uint32_t all = part[1023];
  if (cap && all > cap) {
    for (uint32_t i = lo; i < hi; ++i) { base[i] = 0; counts[i] = 0; }
    if (t == 1023) { total[0] = 0; total[1] = all; if (chunk_ctr) { chunk_ctr[0] = 0; chunk_ctr[1] = 0; chunk_ctr[2] = 0; } }
    return;
  }
  uint32_t run = (t == 0) ? 0 : part[t - 1];
  for (uint32_t i = lo; i < hi; ++i) { base[i] = run; run += counts[i]; }
  if (t == 1023) {
    total[0] = all; total[1] = 0;
    uint32_t units = 0;
    if (chunk_ctr) { units = chunk_ctr[0]; chunk_ctr[0] = 0; chunk_ctr[1] = 0; chunk_ctr[2] = units; }
    if (host_total) {
      host_total[2] = units;      // (ahead of the release below: the host reads it after the packet count)
      __hip_atomic_store(host_total, all, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// hashing
// ------------------------------------------------------------------------------------------------
// 64-byte block at an arbitrarily aligned address as 16 big-endian words: 17 aligned dword loads
// funnel-shifted with v_alignbyte_b32 (the bytes before/after the block inside the same aligned
// dwords are readable: they belong to the same allocation).
__device__ __forceinline__ void load_block_raw(const uint8_t* p, uint32_t (&t)[17]) {
  const uint32_t mis = (uint32_t)((uintptr_t)p & 3);
  const uint32_t* ap = (const uint32_t*)((uintptr_t)p & ~(uintptr_t)3);
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = ap[i];
  t[16] = mis ? ap[16] : 0;
}
__device__ __forceinline__ void block_raw_to_be(const uint8_t* p, const uint32_t (&t)[17], uint32_t (&w)[16]) {
  const uint32_t mis = (uint32_t)((uintptr_t)p & 3);
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = __builtin_bswap32(__builtin_amdgcn_alignbyte(t[i + 1], t[i], mis));
}
__device__ __forceinline__ void load_block_be(const uint8_t* p, uint32_t (&w)[16]) {
  uint32_t t[17];
  load_block_raw(p, t);
  block_raw_to_be(p, t, w);
}

// Thread per item, one compression after the other: the loop is a chain of (uncoalesced) block loads and 64 rounds.
// The next block's 17 dwords are requested before the current block is compressed, so the load latency hides behind
// the rounds (two blocks per trip, no register shuffling).
__global__ void __launch_bounds__(64) k_sha256_mid(const uint8_t* __restrict__ tbs_blob, const uint64_t* __restrict__ tbs_off,
                                                   uint32_t n_items, uint32_t* __restrict__ mid /*[n][8]*/) {
  uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= n_items) return;
  const uint8_t* p = tbs_blob + tbs_off[item];
  const uint64_t nblk = (tbs_off[item + 1] - tbs_off[item]) >> 6;
  uint32_t s[8];
  sha256_init(s);
  uint32_t ta[17], tb[17], w[16];
  if (nblk) load_block_raw(p, ta);
  uint64_t blk = 0;
  for (; blk + 1 < nblk; blk += 2) {
    load_block_raw(p + (blk + 1) * 64, tb);
    block_raw_to_be(p + blk * 64, ta, w);
    sha256_compress(s, w);
    if (blk + 2 < nblk) load_block_raw(p + (blk + 2) * 64, ta);
    block_raw_to_be(p + (blk + 1) * 64, tb, w);
    sha256_compress(s, w);
  }
  if (blk < nblk) {
    block_raw_to_be(p + blk * 64, ta, w);
    sha256_compress(s, w);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) mid[(uint64_t)item * 8 + i] = s[i];
}

// Midstates of the other hashes, only for the items whose signatures ask for them
// (item_hash_mask bits: 1 SHA-224, 2 SHA-1, 3 SHA-512, 4 SHA-384, 5 MD5, 6 RIPEMD-160).  Thread per (algorithm, item).
__device__ __forceinline__ void load_block_le(const uint8_t* p, uint32_t (&w)[16]) {
  uint32_t t[17];
  load_block_raw(p, t);
  const uint32_t mis = (uint32_t)((uintptr_t)p & 3);
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = __builtin_amdgcn_alignbyte(t[i + 1], t[i], mis);
}
__global__ void __launch_bounds__(64) k_hash_mid_other(const uint8_t* __restrict__ tbs_blob, const uint64_t* __restrict__ tbs_off,
                                                       uint32_t n_items, const uint32_t* __restrict__ item_hash_mask,
                                                       uint32_t* __restrict__ mid32 /*[5][n][8]*/, uint64_t* __restrict__ mid64 /*[2][n][8]*/) {
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t which = blockIdx.y;   // 0 SHA-224, 1 SHA-1, 2 SHA-512, 3 SHA-384, 4 MD5, 5 RIPEMD-160  (= hash_info().idx - 1)
  if (item >= n_items) return;
  if (!((item_hash_mask[item] >> (which + 1)) & 1u)) return;
  const uint8_t* p = tbs_blob + tbs_off[item];
  const uint64_t len = tbs_off[item + 1] - tbs_off[item];
  if (which < 2 || which >= 4) {
    uint32_t s[8];
    if (which == 0) sha224_init(s); else if (which == 1) sha1_init(s); else if (which == 4) md5_init(s); else ripemd160_init(s);
    for (uint64_t blk = 0; blk < (len >> 6); ++blk) {
      uint32_t w[16];
      if (which >= 4) { load_block_le(p + blk * 64, w); if (which == 4) md5_compress(s, w); else ripemd160_compress(s, w); }
      else { load_block_be(p + blk * 64, w); if (which == 0) sha256_compress(s, w); else sha1_compress(s, w); }
    }
    const uint32_t slot32 = which < 2 ? which + 1 : which - 1;      // 1 SHA-224, 2 SHA-1, 3 MD5, 4 RIPEMD-160
    uint32_t* o = mid32 + ((uint64_t)slot32 * n_items + item) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = s[i];
  } else {
    uint64_t s[8];
    if (which == 2) sha512_init(s); else sha384_init(s);
    for (uint64_t blk = 0; blk < (len >> 7); ++blk) {
      uint32_t lo[16], hi[16];
      load_block_be(p + blk * 128, lo);
      load_block_be(p + blk * 128 + 64, hi);
      uint64_t w[16];
#pragma unroll
      for (int i = 0; i < 8; ++i) { w[i] = ((uint64_t)lo[2 * i] << 32) | lo[2 * i + 1]; w[8 + i] = ((uint64_t)hi[2 * i] << 32) | hi[2 * i + 1]; }
      sha512_compress(s, w);
    }
    uint64_t* o = mid64 + ((uint64_t)(which - 2) * n_items + item) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = s[i];
  }
}

// Canonical-text midstates (text-mode signatures, hashForSignature's NewCanonicalTextHash): thread per (hash, item), only for
// the items whose signatures ask.  The payload is streamed byte by byte through x/crypto's two-state rewriter -- a '\n' that
// does not follow a '\r' becomes "\r\n", the byte after a '\r' passes unchanged whatever it is -- into a block buffer.
// Rare path: nothing here is tuned.
__global__ void __launch_bounds__(64) k_hash_mid_text(const uint8_t* __restrict__ tbs_blob, const uint64_t* __restrict__ tbs_off,
                                                      uint32_t n_items, const uint32_t* __restrict__ item_hash_mask, TextDev txt) {
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t which = blockIdx.y;   // hash_info().idx: 0 SHA-256, 1 SHA-224, 2 SHA-1, 3 SHA-512, 4 SHA-384, 5 MD5, 6 RIPEMD-160
  if (item >= n_items) return;
  if (!((item_hash_mask[item] >> (ITEM_TEXT_SHIFT + which)) & 1u)) return;
  const uint8_t* p = tbs_blob + tbs_off[item];
  const uint64_t len = tbs_off[item + 1] - tbs_off[item];
  const bool wide = which == 3 || which == 4;
  const uint32_t B = wide ? 128u : 64u;
  uint32_t s32[8];
  uint64_t s64[8];
  if (which == 0) sha256_init(s32); else if (which == 1) sha224_init(s32); else if (which == 2) sha1_init(s32);
  else if (which == 3) sha512_init(s64); else if (which == 4) sha384_init(s64); else if (which == 5) md5_init(s32); else ripemd160_init(s32);
  uint8_t buf[128];
  uint32_t fill = 0;
  uint64_t total = 0;
  auto put = [&](uint8_t c) {
    buf[fill++] = c;
    ++total;
    if (fill == B) {
      if (which >= 5) {
        uint32_t w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) | ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
        if (which == 5) md5_compress(s32, w); else ripemd160_compress(s32, w);
      } else if (!wide) {
        uint32_t w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)buf[4 * i] << 24) | ((uint32_t)buf[4 * i + 1] << 16) | ((uint32_t)buf[4 * i + 2] << 8) | buf[4 * i + 3];
        if (which == 2) sha1_compress(s32, w); else sha256_compress(s32, w);
      } else {
        uint64_t w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          uint64_t v = 0;
#pragma unroll
          for (int t = 0; t < 8; ++t) v = (v << 8) | buf[8 * i + t];
          w[i] = v;
        }
        sha512_compress(s64, w);
      }
      fill = 0;
    }
  };
  int st = 0;
  for (uint64_t i = 0; i < len; ++i) {
    const uint8_t c = p[i];
    if (st == 0) {
      if (c == 0x0D) { st = 1; put(c); }
      else if (c == 0x0A) { put(0x0D); put(0x0A); }
      else put(c);
    } else { st = 0; put(c); }
  }
  const uint64_t slot = (uint64_t)which * n_items + item;
  if (!wide) {
    const uint32_t slot32 = which < 3 ? which : which - 2;       // 0 SHA-256, 1 SHA-224, 2 SHA-1, 3 MD5, 4 RIPEMD-160
    for (int i = 0; i < 8; ++i) txt.mid32[((uint64_t)slot32 * n_items + item) * 8 + i] = s32[i];
  } else { for (int i = 0; i < 8; ++i) txt.mid64[((uint64_t)(which - 3) * n_items + item) * 8 + i] = s64[i]; }
  for (uint32_t i = 0; i < fill; ++i) txt.tail[slot * 128 + i] = buf[i];
  txt.len[slot] = total;
}

struct TailSrc {
  const uint8_t* tail; uint32_t tail_len;     // last (len % 64) bytes of the signed payload
  const uint8_t* body; uint32_t pre_len;      // v4: first 6+hl bytes of the signature body; v3: type || creation time (5 bytes)
  uint32_t tr_len;                            // v4: the 6-byte trailer 04 FF len32; v3: none
  uint32_t rep;                               // how often the hash suffix (pre || trailer) is written: 1, or j+1 for the
                                              // j-th further candidate key of one issuer id (k_candidates)
};
__device__ __forceinline__ uint32_t tail_byte(const TailSrc& t, uint32_t j) {
  if (j < t.tail_len) return t.tail[j];
  j -= t.tail_len;
  const uint32_t unit = t.pre_len + t.tr_len;
  if (j >= unit * t.rep) return (j == unit * t.rep) ? 0x80 : 0;   // first byte after the message: the padding marker
  if (t.rep > 1) j %= unit;
  if (j < t.pre_len) return t.body[j];
  j -= t.pre_len;
  if (j == 0) return 0x04;
  if (j == 1) return 0xFF;
  return (t.pre_len >> (8 * (5 - j))) & 0xFF;
}

// value of a big-endian byte string as radix-2^28 limb j
template <typename F>
__device__ __forceinline__ uint32_t limb28(F byte_from_lsb, int j) {
  uint32_t bit = 28u * (uint32_t)j;
  uint32_t b0 = bit >> 3, sh = bit & 7;
  uint64_t v = 0;
#pragma unroll
  for (int t = 0; t < 5; ++t) v |= (uint64_t)byte_from_lsb(b0 + t) << (8 * t);
  return (uint32_t)(v >> sh) & MONT_MASK;
}

// Per signature: digest = H(signed || hash suffix) from the item's midstate; hash-tag check.
// digests: 64 bytes per record, the digest in its natural (big-endian) byte order.
template <bool OTHERS>   // false: SHA-256 only (the path's default, light on registers); true: every other hash
__device__ __forceinline__ void digest_body(const uint8_t* __restrict__ tbs_blob, const uint64_t* __restrict__ tbs_off,
                                            const uint8_t* __restrict__ sig_blob, const uint32_t* __restrict__ mid32,
                                            const uint64_t* __restrict__ mid64, uint32_t n_items,
                                            SigRec* __restrict__ recs, uint32_t n_recs, uint32_t* __restrict__ digests /*[n_recs][16]*/,
                                            uint32_t ri_in, const uint64_t* __restrict__ tbs_prefix = nullptr, uint32_t rep = 1,
                                            uint32_t* __restrict__ tag_only = nullptr, TextDev txt = TextDev{nullptr, nullptr, nullptr, nullptr}) {
  // tag_only (k_candidates): the top 16 bits of the digest with the hash suffix written `rep` times, nothing stored
  const uint32_t ri = ri_in;
  if (ri >= n_recs) return;
  const SigRec rec = recs[ri];
  if (!tag_only && rec.status != ST_PENDING_HASH) return;
  const bool text = OTHERS && (rec.flags & SIGF_TEXT) != 0;      // (text-mode records of any hash take the OTHERS kernels)
  if (!tag_only && ((rec.hash_id != HASH_SHA256) || (rec.flags & SIGF_TEXT) != 0) != OTHERS) return;
  if (OTHERS && text && !txt.len) return;                          // a call without payloads: the item is run again with them
  const HashInfo hi = OTHERS ? hash_info(rec.hash_id) : HashInfo{32, 0, 32, 19};
  // tbs_prefix (callers that absorbed the whole blocks of their payload themselves, host_sha256.h): the blob holds only the
  // < 64 bytes behind the midstate, tbs_prefix[item] bytes went before them
  const uint64_t seg = tbs_off[rec.item + 1] - tbs_off[rec.item];
  const uint32_t bmask = (OTHERS && hi.family == 64) ? 127u : 63u;
  const uint64_t tslot = (uint64_t)hi.idx * n_items + rec.item;     // text-mode state of (hash, item)
  const uint64_t tlen = text ? txt.len[tslot] : (tbs_prefix ? tbs_prefix[rec.item] + seg : seg);
  TailSrc ts;
  ts.tail_len = text ? (uint32_t)(tlen & bmask) : (tbs_prefix ? (uint32_t)seg : (uint32_t)(tlen & bmask));
  ts.tail = text ? txt.tail + tslot * 128 : tbs_blob + tbs_off[rec.item] + (seg - ts.tail_len);
  const bool v3 = (rec.flags & SIGF_V3) != 0;
  ts.body = sig_blob + rec.body_off + (v3 ? 2 : 0);
  ts.pre_len = v3 ? 5u : 6u + rec.hashed_len;
  ts.tr_len = v3 ? 0u : 6u;
  ts.rep = rep;
  const uint32_t rem = ts.tail_len + (ts.pre_len + ts.tr_len) * rep;     // message bytes still to hash
  const uint64_t bits = (tlen + (uint64_t)(ts.pre_len + ts.tr_len) * rep) * 8;
  uint32_t dg_dummy[16];
  uint32_t* dg = tag_only ? dg_dummy : digests + (uint64_t)ri * 16;
  uint32_t tag_hi;
  if (!OTHERS || hi.family == 32) {
    const uint32_t nblk = (rem + 9 + 63) >> 6;
    uint32_t s[8];
    const uint32_t* m = (text ? txt.mid32 : mid32) + ((uint64_t)hi.slot * n_items + rec.item) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = m[i];
    const bool le = OTHERS && hi.le;       // MD5 / RIPEMD-160: little-endian words and length
    for (uint32_t blk = 0; blk < nblk; ++blk) {
      uint32_t w[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        uint32_t j = blk * 64 + i * 4;
        const uint32_t be = (tail_byte(ts, j) << 24) | (tail_byte(ts, j + 1) << 16) | (tail_byte(ts, j + 2) << 8) | tail_byte(ts, j + 3);
        w[i] = le ? __builtin_bswap32(be) : be;
      }
      if (blk == nblk - 1) { w[14] = le ? (uint32_t)bits : (uint32_t)(bits >> 32); w[15] = le ? (uint32_t)(bits >> 32) : (uint32_t)bits; }
      if (OTHERS && hi.slot == 2) sha1_compress(s, w);
      else if (OTHERS && hi.slot == 3) md5_compress(s, w);
      else if (OTHERS && hi.slot == 4) ripemd160_compress(s, w);
      else sha256_compress(s, w);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) dg[i] = le ? s[i] : __builtin_bswap32(s[i]);
    tag_hi = le ? __builtin_bswap32(s[0]) : s[0];
  } else if (OTHERS) {
    const uint32_t nblk = (rem + 17 + 127) >> 7;
    uint64_t s[8];
    const uint64_t* m = (text ? txt.mid64 : mid64) + ((uint64_t)hi.slot * n_items + rec.item) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = m[i];
    for (uint32_t blk = 0; blk < nblk; ++blk) {
      uint64_t w[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        uint32_t j = blk * 128 + i * 8;
        uint64_t v = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) v = (v << 8) | tail_byte(ts, j + t);
        w[i] = v;
      }
      if (blk == nblk - 1) { w[14] = 0; w[15] = bits; }
      sha512_compress(s, w);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { dg[2 * i] = __builtin_bswap32((uint32_t)(s[i] >> 32)); dg[2 * i + 1] = __builtin_bswap32((uint32_t)s[i]); }
    tag_hi = (uint32_t)(s[0] >> 32);
  } else {
    tag_hi = 0;
  }
  if (tag_only) { *tag_only = tag_hi >> 16; return; }
  // PublicKey.VerifySignature: hash tag first, then whatever k_parse_body determined
  uint8_t st;
  if ((uint8_t)(tag_hi >> 24) != rec.hash_tag[0] || (uint8_t)(tag_hi >> 16) != rec.hash_tag[1]) st = ST_HASH_TAG;
  else st = (rec.after_tag == AFTER_TAG_PUBKEY) ? (uint8_t)ST_PENDING_RSA : rec.after_tag;
  recs[ri].status = st;
}

__global__ void __launch_bounds__(256) k_digest_sha256(const uint8_t* __restrict__ tbs_blob, const uint64_t* __restrict__ tbs_off,
                                                       const uint8_t* __restrict__ sig_blob, const uint32_t* __restrict__ mid32,
                                                       const uint64_t* __restrict__ mid64, uint32_t n_items, SigRec* __restrict__ recs,
                                                       uint32_t n_recs, uint32_t* __restrict__ digests,
                                                       const uint64_t* __restrict__ tbs_prefix, const uint32_t* __restrict__ n_recs_dev) {
  digest_body<false>(tbs_blob, tbs_off, sig_blob, mid32, mid64, n_items, recs, n_recs_dev ? *n_recs_dev : n_recs, digests,
                     blockIdx.x * blockDim.x + threadIdx.x, tbs_prefix);
}
// Every other hash (SHA-1 / 224 / 384 / 512).  Round 1 capped this kernel at 128 VGPRs so that its grid of immediate exits
// could co-schedule beside k_rsa_modexp, at the price of 510 spilled VGPRs in the SHA-512 path; with the bounded grid
// below an idle launch costs nothing, so the kernel keeps its natural register count and nothing spills.
__global__ void __launch_bounds__(256) k_digest_other(const uint8_t* __restrict__ tbs_blob, const uint64_t* __restrict__ tbs_off,
                                                         const uint8_t* __restrict__ sig_blob, const uint32_t* __restrict__ mid32,
                                                         const uint64_t* __restrict__ mid64, uint32_t n_items, SigRec* __restrict__ recs,
                                                         uint32_t n_recs, uint32_t* __restrict__ digests,
                                                         const uint32_t* __restrict__ any_other /*set by k_parse_body*/, TextDev txt,
                                                         const uint32_t* __restrict__ n_recs_dev /*calls sized by an upper bound: the real count*/) {
  if (*any_other == 0) return;        // every signature of the batch is binary SHA-256 (the path's default): nothing to read
  if (n_recs_dev) n_recs = *n_recs_dev;
  // bounded grid (the host launches at most DIGEST_OTHER_MAX_BLOCKS blocks): normally this kernel has nothing to do, and a
  // grid of one block per 256 records -- 104k blocks for a cfg-4 batch -- spent 40 ms just being dispatched beside the modexp
  for (uint64_t ri = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; ri < n_recs; ri += (uint64_t)gridDim.x * blockDim.x)
    digest_body<true>(tbs_blob, tbs_off, sig_blob, mid32, mid64, n_items, recs, n_recs, digests, (uint32_t)ri, nullptr, 1, nullptr, txt);
}

// Several different keys under one 64-bit key id (ids are 64 bits of a SHA-1: a collision costs ~2^32 work, or nothing for
// whoever generates both keys).  The reference asks every candidate in keyring order and returns the first success or the
// LAST error (openpgp.CheckDetachedSignature).  Only the first candidate that can sign ever sees the true digest -- it went
// through the pipeline as the record's key (parse_one).  Each later one finds the hash suffix written once more into the
// shared hash (candidate j hashes payload || suffix x (j+1)), so its tag check fails, or -- once in 2^16 -- passes and the
// arithmetic over a foreign digest fails unless the twin's owner signed that very digest (then the item is fenced, below);
// a candidate that cannot sign fails before it writes anything.  Otherwise none of them can turn a failure into a success,
// so verdicts and tallies stand as the pipeline left them; this kernel, launched only when the key table holds such ids,
// settles the STATUS of the records their first candidate did not verify.
__global__ void __launch_bounds__(64) k_candidates(const uint8_t* __restrict__ tbs_blob, const uint64_t* __restrict__ tbs_off,
                                                   const uint8_t* __restrict__ sig_blob, const uint32_t* __restrict__ mid32,
                                                   const uint64_t* __restrict__ mid64, uint32_t n_items, SigRec* __restrict__ recs,
                                                   uint32_t n_recs, const uint32_t* __restrict__ n_recs_dev, const uint64_t* __restrict__ tbs_prefix,
                                                   KeyTableDev kt, const uint32_t* __restrict__ cert_ent, const uint8_t* __restrict__ sig_class,
                                                   TextDev txt, uint32_t* __restrict__ item_hash_mask) {
  const uint32_t ri = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nr = n_recs_dev ? *n_recs_dev : n_recs;
  if (ri >= nr) return;
  const SigRec rec = recs[ri];
  if (!(rec.flags & SIGF_MORE_CANDIDATES)) return;
  uint8_t st = rec.status;
  // statuses the candidate loop produces; anything else was decided before the loop, is a success, or was never examined
  if (!(st == ST_KEY_CANNOT_SIGN || st == ST_HASH_TAG || st == ST_ALGO_MISMATCH || st == ST_BAD_SIG)) return;
  if (tbs_prefix && (rec.hash_id != HASH_SHA256 || (rec.flags & SIGF_TEXT))) return;      // staged call: the item is run again with its payload
  const uint32_t slot = (uint32_t)rec.key_slot;
  const uint64_t issuer = kt.key_id[slot];
  const uint32_t only_ent = cert_ent ? cert_ent[rec.item] : 0xFFFFFFFFu;
  const uint32_t cls = sig_class ? sig_class[rec.item] : 0;
  uint32_t lo_i = 0, hi_i = kt.n_keys;
  while (lo_i < hi_i) {
    const uint32_t mid = (lo_i + hi_i) >> 1;
    if (kt.sorted_id[mid] < issuer) lo_i = mid + 1; else hi_i = mid;
  }
  bool behind = false;
  uint32_t writes = (kt.flags[slot] & KEYF_CAN_SIGN) ? 1u : 0u;
  for (uint32_t i = lo_i; i < kt.n_keys && kt.sorted_id[i] == issuer; ++i) {
    const uint32_t k = kt.sorted_slot[i];
    if (!key_is_candidate(kt, k, only_ent, cls)) continue;
    if (k == slot) { behind = true; continue; }
    if (!behind) continue;                       // ahead of the record's key: could not sign, wrote nothing, error overwritten
    if (!(kt.flags[k] & KEYF_CAN_SIGN)) { st = ST_KEY_CANNOT_SIGN; continue; }
    ++writes;
    uint32_t tag = 0;
    if (rec.hash_id == HASH_SHA256 && !(rec.flags & SIGF_TEXT))
      digest_body<false>(tbs_blob, tbs_off, sig_blob, mid32, mid64, n_items, recs, nr, nullptr, ri, tbs_prefix, writes, &tag);
    else digest_body<true>(tbs_blob, tbs_off, sig_blob, mid32, mid64, n_items, recs, nr, nullptr, ri, tbs_prefix, writes, &tag, txt);
    if ((uint8_t)(tag >> 8) != rec.hash_tag[0] || (uint8_t)tag != rec.hash_tag[1]) st = ST_HASH_TAG;
    else if (kt.pk_algo[k] != rec.pk_algo) st = ST_ALGO_MISMATCH;
    else {
      // Tag and algorithm fit candidate j's digest H(payload || suffix x (j+1)): the reference now runs the public-key check
      // over THAT digest, and whoever owns this twin key can have signed exactly it -- the reference would then return the
      // twin as the signer.  The arithmetic is not repeated here (once in 2^16 by chance, otherwise a deliberate shape): the
      // item is fenced and the reference decides; the status stays what an honest signature would leave.
      st = ST_BAD_SIG;
      atomicOr(&item_hash_mask[rec.item], ITEM_FENCED);
    }
  }
  recs[ri].status = st;
}

// ------------------------------------------------------------------------------------------------
// RSA: 4 lanes per signature
// ------------------------------------------------------------------------------------------------
constexpr int RSA_BLOCK = 256;
constexpr int QUADS_PER_BLOCK = RSA_BLOCK / MONT_TPI;
// k_rsa_modexp's waves never meet (wavefront fences only), so its block size only decides how many waves must retire before
// their CU slots are handed out again; measured variants: tools/ab.sh with -DBFTKV_MODEXP_BLOCK=64|128|256.
#ifndef BFTKV_MODEXP_BLOCK
#define BFTKV_MODEXP_BLOCK 256
#endif
constexpr int MODEXP_BLOCK = BFTKV_MODEXP_BLOCK;

enum : int { OP_TO_MONT = 0, OP_SQR = 1, OP_MULX = 2, OP_MULP = 3, OP_MUL1 = 4 };

// EMSA-PKCS1-v1_5 splits at byte EM_LOW_BYTES: below it sit the digest, its DigestInfo prefix and the 00 separator (at most
// 64 + 19 + 1 bytes, SHA-512) plus FF padding, above it only padding and the 00 01 top, a pattern of the modulus' byte
// length alone.  k_rsa_modexp checks the upper part while the limbs of s^e mod n are still in registers and hands
// k_rsa_compare the low EM_LOW_LIMBS limbs (96 instead of 304 bytes per signature, written once and read once).
constexpr int EM_LOW_BYTES = 84;
constexpr int EM_LOW_LIMBS = EM_LOW_BYTES * 8 / MONT_W;     // 24: 84 bytes are exactly 24 limbs of 28 bits
static_assert(EM_LOW_LIMBS * MONT_W == EM_LOW_BYTES * 8, "the EM split must fall on a limb boundary");
constexpr uint32_t EM_HEAD_BAD = 0xFFFFFFFFu;               // no canonical limb: marks a residue whose upper part is not padding

// Limb gi (>= EM_LOW_LIMBS) of 00 01 FF .. FF for a modulus of kbytes bytes: ones from bit 8*EM_LOW_BYTES up to and
// including bit 8*(kbytes-2), the 01 byte.
__device__ __forceinline__ uint32_t em_head_limb(uint32_t gi, uint32_t kbytes) {
  const int32_t top = 8 * ((int32_t)kbytes - 2), lo = (int32_t)gi * MONT_W;
  if (top < lo) return 0u;
  if (top >= lo + MONT_W - 1) return MONT_MASK;
  return (1u << (top - lo + 1)) - 1u;
}

// r = s^e mod n for every queued signature: the upper limbs checked against the EMSA padding here, the low EM_LOW_LIMBS
// limbs (canonical radix 2^28) to r_low for k_rsa_compare.
template <int L, int TPI>   // limbs per lane x lanes per number: 19x4 (<= 2048-bit moduli), 14x8 (<= 3072), 19x8 (<= 4096)
__global__ void __launch_bounds__(MODEXP_BLOCK, 3) k_rsa_modexp(const uint8_t* __restrict__ sig_blob, const SigRec* __restrict__ recs,
                                                          const uint32_t* __restrict__ pk_list, const uint32_t* __restrict__ pk_count_ptr,
                                                          const uint32_t* __restrict__ pk_start_ptr,
                                                          KeyTableDev kt, uint32_t* __restrict__ r_low,
                                                          uint32_t* __restrict__ xr_scratch, uint64_t* __restrict__ clk) {
  constexpr int NL = TPI * L;
  constexpr int GROUPS = MODEXP_BLOCK / TPI;   // numbers per block
  // diagnostics: shader-clock ticks (s_memtime) and constant 100 MHz ticks (s_memrealtime) over the life of block 0's first
  // wave -- the clock the part actually ran this kernel at (bftkv_gpu_last_sclk_mhz); two scalar reads at each end
  const bool stamp = clk && blockIdx.x == 0 && threadIdx.x == 0;
  uint64_t t0 = 0, r0 = 0;
  if (stamp) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
  __shared__ uint32_t a_sh[GROUPS * NL];
  __shared__ uint32_t x_sh[GROUPS * NL];
  // work = list entries [start, count): phase 1 of a call starts at 0, phase 2 where phase 1 ended (k_plan)
  const uint32_t count = *pk_count_ptr, start = *pk_start_ptr;
  if (start + blockIdx.x * GROUPS >= count) return;   // whole block idle
  const uint32_t quad = threadIdx.x / TPI;
  const int qlane = threadIdx.x % TPI;
  const uint32_t gq = start + blockIdx.x * GROUPS + quad;
  const bool active = gq < count;
  const uint32_t pi = active ? gq : (count - 1);
  const uint32_t ri = pk_list[pi];
  const SigRec rec = recs[ri];
  const uint32_t key = (uint32_t)rec.key_slot;
  uint32_t* a_lds = a_sh + quad * NL + qlane * L;     // this lane's slice of the quad's operand
  const uint32_t* a_rd = a_sh + quad * NL;
  uint32_t* x_lds = x_sh + quad * NL + qlane * L;

  uint32_t n[L], b[L], y[L];
  const uint32_t* np = kt.n_limbs + (uint64_t)key * MONT_NMAX + qlane * L;
  // <10, 8>: a <= 2048-bit modulus over eight lanes (80 limbs, R = 2^2240) -- 0.68x the instructions of <19, 4> per wave, the
  // shorter chain for calls too small to fill the machine either way (run_pipeline picks it below ~8k signatures)
  const uint32_t* rp = (L == 10 && TPI == 8) ? kt.r2_limbs80 + (uint64_t)key * 80 + qlane * L
                                              : kt.r2_limbs + (uint64_t)key * MONT_NMAX + qlane * L;
  uint32_t* xrp = xr_scratch + (uint64_t)pi * NL + qlane * L;
#pragma unroll
  for (int k = 0; k < L; ++k) n[k] = np[k];
  // signature value: big-endian MPI bytes -> this lane's 19 limbs
  {
    const uint32_t nb = (rec.mpi_bits[0] + 7u) >> 3;
    const uint8_t* mp = sig_blob + rec.body_off + rec.mpi_off[0];
    auto sig_b = [&](uint32_t i) -> uint32_t { return i < nb ? mp[nb - 1 - i] : 0u; };
#pragma unroll
    for (int k = 0; k < L; ++k) x_lds[k] = limb28(sig_b, qlane * L + k);
  }
  const uint32_t n0inv = kt.n0inv[key];
  const uint32_t e = kt.rsa_e[key];
  const uint32_t kbytes = (kt.mod_bits[key] + 7) >> 3;
  // x-shortcut: the last multiplication of an odd exponent uses plain x instead of xR, which also
  // leaves the Montgomery domain.  Only when x < 2^(8k) so that the result stays below n(1+2^-79).
  const uint32_t cls = (e << 1) | (((e & 1u) && e > 1u && !(rec.flags & 1u)) ? 1u : 0u);

  // Exponent schedules are wave-uniform per (e, shortcut) class; a wave whose 16 signatures use
  // different public exponents runs the schedule once per class (all real keys use 65537).
  uint64_t todo = __builtin_amdgcn_ballot_w64(true);
  while (todo) {
    const int lead = __builtin_ctzll(todo);
    const uint32_t cls_u = (uint32_t)__builtin_amdgcn_readlane((int)cls, lead);
    const bool live = (cls == cls_u);
    const uint32_t e_u = cls_u >> 1;
    const bool sc_u = cls_u & 1u;
    const int top = 31 - __builtin_clz(e_u | 1u);
    int kind = OP_TO_MONT, bitpos = top;
    while (true) {
      // ---- operands of this step
      if (kind == OP_TO_MONT) {
#pragma unroll
        for (int k = 0; k < L; ++k) { b[k] = rp[k]; a_lds[k] = x_lds[k]; }
      } else {
#pragma unroll
        for (int k = 0; k < L; ++k) b[k] = y[k];
        if (kind == OP_SQR) {
#pragma unroll
          for (int k = 0; k < L; ++k) a_lds[k] = y[k];
        } else if (kind == OP_MULX) {
#pragma unroll
          for (int k = 0; k < L; ++k) a_lds[k] = xrp[k];
        } else if (kind == OP_MULP) {
#pragma unroll
          for (int k = 0; k < L; ++k) a_lds[k] = x_lds[k];
        } else {
#pragma unroll
          for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // 16 of the 18 products of e = 65537 are squarings: half the a*b limb products (mont28.h, SQR)
      if (kind == OP_SQR) mont_mul<L, TPI, true>(y, a_rd, b, n, n0inv, qlane);
      else mont_mul<L, TPI, false>(y, a_rd, b, n, n0inv, qlane);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // ---- next step (scalar control flow)
      if (kind == OP_TO_MONT && (e_u & (e_u - 1u)) != 0 && !(sc_u && __builtin_popcount(e_u) == 2)) {
#pragma unroll
        for (int k = 0; k < L; ++k) xrp[k] = y[k];   // xR is needed again by OP_MULX
      }
      if (kind == OP_MULP || kind == OP_MUL1) break;
      if (kind == OP_SQR && ((e_u >> bitpos) & 1u)) { kind = (bitpos == 0 && sc_u) ? OP_MULP : OP_MULX; continue; }
      --bitpos;
      kind = (bitpos >= 0) ? OP_SQR : OP_MUL1;
    }
    // y = t or t + n for the residue t = s^e mod n, and t + n only when t < n * 2^-63: the last product is by 1 (y < n + 1)
    // or by the plain value x < 2^(8k) (y < n(1 + 2^(8k+1)/R), R = 2^2128 / 2^3136 / 2^4256).  An encoded message is at
    // least 2^(8k-16) > n * 2^-16, so such a t is no valid signature and neither is y >= n: comparing y itself with the
    // encoding gives the verdict of comparing t, without a final subtraction.
    canonicalize<L, TPI>(y, qlane);
    if (e_u == 0) {                            // x^0 = 1
#pragma unroll
      for (int k = 0; k < L; ++k) y[k] = (qlane == 0 && k == 0) ? 1u : 0u;
    }
    uint32_t head = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
      const uint32_t gi = (uint32_t)qlane * L + k;
      if (gi >= (uint32_t)EM_LOW_LIMBS) head |= y[k] ^ em_head_limb(gi, kbytes);
    }
    head = grp_or<TPI>(head);
    // the low limbs leave through the group's LDS slice (free now): each of the group's first four lanes stores six
    // consecutive limbs, so a wave writes one contiguous run
    if (head != 0 && qlane == 0) y[0] = EM_HEAD_BAD;
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = y[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (live && active && qlane < 4) {
      uint32_t* out = r_low + (uint64_t)pi * EM_LOW_LIMBS + qlane * (EM_LOW_LIMBS / 4);
      const uint32_t* src = a_rd + qlane * (EM_LOW_LIMBS / 4);
#pragma unroll
      for (int k = 0; k < EM_LOW_LIMBS / 4; ++k) out[k] = src[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    todo &= ~__builtin_amdgcn_ballot_w64(live);
  }
  if (stamp && *pk_start_ptr == 0) { clk[0] = __builtin_readcyclecounter() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

// Low EM_LOW_LIMBS limbs of EMSA-PKCS1-v1_5(digest) == those of s^e mod n (k_rsa_modexp has checked everything above
// them); four lanes per signature, six limbs each, whatever the modulus size.
constexpr int CMP_LANES = 4;
constexpr int CMP_L = EM_LOW_LIMBS / CMP_LANES;
static_assert(CMP_L * CMP_LANES == EM_LOW_LIMBS, "the low limbs divide over the quad");
__global__ void __launch_bounds__(256) k_rsa_compare(SigRec* __restrict__ recs, const uint32_t* __restrict__ pk_list,
                                                     const uint32_t* __restrict__ pk_count_ptr, const uint32_t* __restrict__ pk_start_ptr,
                                                     KeyTableDev kt,
                                                     const uint32_t* __restrict__ r_low, const uint32_t* __restrict__ digests) {
  constexpr int GROUPS = 256 / CMP_LANES;
  const uint32_t count = *pk_count_ptr, start = *pk_start_ptr;
  const uint32_t q0 = start + blockIdx.x * GROUPS;
  if (q0 >= count) return;
  const uint32_t gq = q0 + threadIdx.x / CMP_LANES;
  const int qlane = threadIdx.x % CMP_LANES;
  const bool active = gq < count;
  const uint32_t pi = active ? gq : (count - 1);
  const uint32_t ri = pk_list[pi];
  const SigRec rec = recs[ri];
  const bool pending = rec.status == ST_PENDING_RSA;   // hash tag matched
  const uint32_t key = (uint32_t)rec.key_slot;
  const uint32_t kbytes = (kt.mod_bits[key] + 7) >> 3;
  const HashInfo hi = hash_info(rec.hash_id);
  const uint32_t hlen = hi.dlen, plen = hi.plen, tl = hlen + plen;
  // EM as little-endian 32-bit words, least significant first: [ digest | DigestInfo prefix | 00 ] (the variable tail,
  // tl+1 bytes, staged in LDS by the quad), then FF words, then -- only under a very short modulus -- the 00 01 top.
  // Each lane builds the 7-word window its six limbs live in, shifts it to a limb boundary once, and slices limbs at
  // compile-time offsets.
  constexpr int EM_TAIL_W = 24;   // >= (64 + 19 + 1) / 4
  __shared__ uint32_t tail_sh[GROUPS * EM_TAIL_W];
  uint32_t* tail = tail_sh + (threadIdx.x / CMP_LANES) * EM_TAIL_W;
  const uint32_t* dgw = digests + (uint64_t)ri * 16;
  const uint32_t hw = hlen >> 2;            // every supported digest length is a multiple of 4
#pragma unroll
  for (int t = 0; t < EM_TAIL_W / CMP_LANES; ++t) {
    const uint32_t w = (uint32_t)qlane * (EM_TAIL_W / CMP_LANES) + t;
    uint32_t v = 0;
    if (w < hw) v = __builtin_bswap32(dgw[hw - 1 - w]);
    else {
#pragma unroll
      for (uint32_t bq = 0; bq < 4; ++bq) {
        const uint32_t i = 4 * w + bq;
        if (i >= hlen && i < tl) v |= digestinfo_byte(rec.hash_id, plen - 1 - (i - hlen)) << (8 * bq);
      }
    }
    tail[w] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  auto em_word = [&](uint32_t w) -> uint32_t {
    if (4 * w + 3 <= tl) return tail[w];
    if (4 * w > tl && 4 * w + 3 + 2 < kbytes) return 0xFFFFFFFFu;
    uint32_t v = 0;                          // a word straddling a boundary
#pragma unroll
    for (uint32_t bq = 0; bq < 4; ++bq) {
      const uint32_t i = 4 * w + bq;
      uint32_t byte;
      if (i <= tl) byte = (tail[i >> 2] >> (8 * (i & 3))) & 0xFFu;
      else if (i + 2 < kbytes) byte = 0xFF;
      else byte = (i + 2 == kbytes) ? 1u : 0u;
      v |= byte << (8 * bq);
    }
    return v;
  };
  constexpr int NW = (CMP_L * MONT_W + 31) / 32 + 1;
  const uint32_t bit0 = (uint32_t)qlane * CMP_L * MONT_W, w0 = bit0 >> 5, s0 = bit0 & 31u;
  uint32_t Wd[NW + 1], Wn[NW];
#pragma unroll
  for (int i = 0; i <= NW; ++i) Wd[i] = em_word(w0 + i);
#pragma unroll
  for (int i = 0; i < NW; ++i) Wn[i] = __builtin_amdgcn_alignbit(Wd[i + 1], Wd[i], s0);
  // r: the block's 64 results are one contiguous run of the work list -- fetched with fully coalesced dword loads into LDS
  __shared__ uint32_t r_sh[GROUPS * EM_LOW_LIMBS];
  {
    const uint32_t n_valid = min((uint32_t)GROUPS, count - q0) * EM_LOW_LIMBS;
    const uint32_t* src = r_low + (uint64_t)q0 * EM_LOW_LIMBS;
    for (uint32_t t = threadIdx.x; t < GROUPS * EM_LOW_LIMBS; t += 256) r_sh[t] = t < n_valid ? src[t] : 0u;
    __syncthreads();
  }
  const uint32_t* rp = r_sh + (threadIdx.x / CMP_LANES) * EM_LOW_LIMBS + qlane * CMP_L;
  uint32_t diff = 0;
#pragma unroll
  for (int k = 0; k < CMP_L; ++k) {
    const int bit = MONT_W * k, wi = bit >> 5, sh = bit & 31;
    const uint32_t em = (sh == 0 ? Wn[wi] : __builtin_amdgcn_alignbit(Wn[wi + 1], Wn[wi], sh)) & MONT_MASK;
    diff |= em ^ rp[k];                     // EM_HEAD_BAD in limb 0 (upper part not padding) never matches a masked limb
  }
  diff = quad_or(diff);
  if (active && pending && qlane == 0) recs[ri].status = (diff == 0) ? ST_OK : ST_BAD_SIG;
}

// ------------------------------------------------------------------------------------------------
// DSA (Go crypto/dsa.Verify): mod-q side per thread, g^u1 * y^u2 mod p per quad
// ------------------------------------------------------------------------------------------------
// Per DSA signature, as soon as it is parsed (runs beside the RSA modexp and the hashing): range checks,
// w = s^-1 mod q, u2 = r*w.  dsa_u row = { w (all zero: signature already refused), u2, r }.
constexpr int DSA_U_WORDS = 24;
// a key's table slot, and the mod-q Montgomery constants behind its tables and its 2^(28 j) mod q rows (device_types.h)
__device__ __forceinline__ const uint32_t* dsa_slot_base(const KeyTableDev& kt, uint32_t key) {
  return kt.dsa_comb + (uint64_t)kt.dsa_slot[key] * dsa_slot_stride(kt.dsa_wbits, kt.dsa_entry_limbs);
}
__device__ __forceinline__ const uint32_t* dsa_qconst(const KeyTableDev& kt, uint32_t key) {
  return dsa_slot_base(kt, key) + dsa_comb_limbs_per_key(kt.dsa_wbits, kt.dsa_entry_limbs) + dsa_qpow_words(kt.dsa_entry_limbs);
}
__global__ void __launch_bounds__(64) k_dsa_inv(const uint8_t* __restrict__ sig_blob, const SigRec* __restrict__ recs,
                                                const uint32_t* __restrict__ dsa_list, const uint32_t* __restrict__ pk_count,
                                                const uint32_t* __restrict__ pk_start,
                                                KeyTableDev kt, uint32_t* __restrict__ dsa_u /*[n][24]*/, uint32_t batch_min) {
  if (pk_count[1] - pk_start[1] >= batch_min) return;     // enough DSA signatures for k_dsa_inv_batched: that one runs
  const uint32_t di = pk_start[1] + blockIdx.x * blockDim.x + threadIdx.x;
  if (di >= pk_count[1]) return;
  const uint32_t ri = dsa_list[di];
  const uint32_t key = (uint32_t)recs[ri].key_slot;      // parse-time fields only: the status byte is the hash stream's
  const uint64_t body_off = recs[ri].body_off;
  const uint32_t off0 = recs[ri].mpi_off[0], off1 = recs[ri].mpi_off[1], bits0 = recs[ri].mpi_bits[0], bits1 = recs[ri].mpi_bits[1];
  U256 q, r, s_, w;
  for (int i = 0; i < 8; ++i) q.w[i] = kt.q_words[(uint64_t)key * 8 + i];
  const uint32_t qbits = kt.q_bits[key];
  const uint8_t* body = sig_blob + body_off;
  bool ok = u256_from_be(body + off0, (bits0 + 7u) >> 3, r);
  ok = u256_from_be(body + off1, (bits1 + 7u) >> 3, s_) && ok;
  ok = ok && !u256_is_zero(r) && u256_cmp(r, q) < 0 && !u256_is_zero(s_) && u256_cmp(s_, q) < 0;   // 0 < r, s < q
  ok = ok && (qbits & 7u) == 0;
  ok = ok && u256_modinv_odd(s_, q, w);
  U256 u2 = u256_zero();
  if (ok) {
    const uint32_t* qc = dsa_qconst(kt, key);
    U256 r2;
    for (int i = 0; i < 8; ++i) r2.w[i] = qc[i];
    u2 = u256_mulmod_mont(w, r, q, qc[8], r2);
  } else w = u256_zero();
  uint32_t* o = dsa_u + (uint64_t)di * DSA_U_WORDS;
  for (int i = 0; i < 8; ++i) { o[i] = w.w[i]; o[8 + i] = u2.w[i]; o[16 + i] = r.w[i]; }
}

// The same rows by Montgomery's trick.  Signatures under one key share q, so a run of them needs ONE extended GCD (65,000
// instructions per signature in k_dsa_inv -- 5.3 G per cfg-3 step, three quarters of what the RSA modexp beside it issues)
// and six 256-bit Montgomery products each: prefix products forward, the inverse of the whole product, then backward
// w_i = inv * prefix_(i-1), inv *= s_i.  Grouping is tile-local: a block counting-sorts the table slots of its 4096 work-list
// entries in LDS (no global pass), then every thread takes 16 consecutive entries of the sorted order -- one run, or two
// where a key boundary falls inside.  The prefix products and the Montgomery forms of the s_i wait in the entries' own
// dsa_u rows.  A product that has no inverse (composite q and an s sharing a factor with it) sends its run through the
// per-signature routine, so the verdicts stay those of math/big.ModInverse.  Both kernels are launched and the DSA work-list
// length, known on the device only, decides which of them runs (batch_min, run_pipeline): with few signatures per key the
// runs shrink towards one entry and a thread would walk up to 16 GCDs in sequence.  Rows written are identical to k_dsa_inv's.
constexpr int INV_BLOCK = 256, INV_PER_THREAD = 16, INV_TILE = INV_BLOCK * INV_PER_THREAD, INV_MAX_SLOTS = 4096;
__global__ void __launch_bounds__(INV_BLOCK) k_dsa_inv_batched(const uint8_t* __restrict__ sig_blob, const SigRec* __restrict__ recs,
                                                               const uint32_t* __restrict__ dsa_list, const uint32_t* __restrict__ pk_count,
                                                               const uint32_t* __restrict__ pk_start,
                                                               KeyTableDev kt, uint32_t* __restrict__ dsa_u /*[n][24]*/, uint32_t batch_min) {
  if (pk_count[1] - pk_start[1] < batch_min) return;      // too few DSA signatures per key to form runs: k_dsa_inv runs
  __shared__ uint32_t bins[INV_MAX_SLOTS];
  __shared__ uint32_t tsum[INV_BLOCK];
  __shared__ uint16_t order[INV_TILE], slot_of[INV_TILE];
  const uint32_t count = pk_count[1], start = pk_start[1];
  const uint32_t tile0 = start + blockIdx.x * INV_TILE;
  if (tile0 >= count) return;
  const uint32_t n = min((uint32_t)INV_TILE, count - tile0);
  const uint32_t t = threadIdx.x;
  // ---- counting sort of the tile's entries by table slot
#pragma unroll
  for (int j = 0; j < INV_MAX_SLOTS / INV_BLOCK; ++j) bins[t + INV_BLOCK * j] = 0;
  __syncthreads();
  for (int j = 0; j < INV_PER_THREAD; ++j) {
    const uint32_t e = t + INV_BLOCK * j;
    if (e < n) {
      const uint32_t sl = kt.dsa_slot[(uint32_t)recs[dsa_list[tile0 + e]].key_slot] & (INV_MAX_SLOTS - 1);
      slot_of[e] = (uint16_t)sl;
      atomicAdd(&bins[sl], 1u);
    }
  }
  __syncthreads();
  constexpr int BPT = INV_MAX_SLOTS / INV_BLOCK;      // bins per thread
  uint32_t loc[BPT], sum = 0;
#pragma unroll
  for (int j = 0; j < BPT; ++j) { loc[j] = sum; sum += bins[BPT * t + j]; }
  tsum[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < INV_BLOCK; off <<= 1) {
    const uint32_t v = t >= off ? tsum[t - off] : 0u;
    __syncthreads();
    tsum[t] += v;
    __syncthreads();
  }
  const uint32_t base = tsum[t] - sum;
#pragma unroll
  for (int j = 0; j < BPT; ++j) bins[BPT * t + j] = base + loc[j];
  __syncthreads();
  for (int j = 0; j < INV_PER_THREAD; ++j) {
    const uint32_t e = t + INV_BLOCK * j;
    if (e < n) order[atomicAdd(&bins[slot_of[e]], 1u)] = (uint16_t)e;
  }
  __syncthreads();
  // ---- runs of one slot inside this thread's 16 sorted entries
  uint32_t pos = INV_PER_THREAD * t;
  const uint32_t end = min(pos + INV_PER_THREAD, n);
  U256 one = u256_zero();
  one.w[0] = 1;
  while (pos < end) {
    const uint32_t sl = slot_of[order[pos]];
    uint32_t re = pos + 1;
    while (re < end && slot_of[order[re]] == sl) ++re;
    const uint32_t key = (uint32_t)recs[dsa_list[tile0 + order[pos]]].key_slot;
    U256 q, r2;
#pragma unroll
    for (int i = 0; i < 8; ++i) q.w[i] = kt.q_words[(uint64_t)key * 8 + i];
    const uint32_t* qc = dsa_qconst(kt, key);
#pragma unroll
    for (int i = 0; i < 8; ++i) r2.w[i] = qc[i];
    const uint32_t q0inv = qc[8];
    const bool q_ok = (kt.q_bits[key] & 7u) == 0;
    uint32_t valid = 0;          // bit i: entry pos + i passed the range checks
    U256 P = one;                // product of the valid s so far, Montgomery form
    for (uint32_t i = pos; i < re; ++i) {
      const uint32_t di = tile0 + order[i], ri = dsa_list[di];
      const uint8_t* body = sig_blob + recs[ri].body_off;
      U256 r, s_;
      bool ok = u256_from_be(body + recs[ri].mpi_off[0], (recs[ri].mpi_bits[0] + 7u) >> 3, r);
      ok = u256_from_be(body + recs[ri].mpi_off[1], (recs[ri].mpi_bits[1] + 7u) >> 3, s_) && ok;
      ok = ok && q_ok && !u256_is_zero(r) && u256_cmp(r, q) < 0 && !u256_is_zero(s_) && u256_cmp(s_, q) < 0;   // 0 < r, s < q
      uint32_t* o = dsa_u + (uint64_t)di * DSA_U_WORDS;
      U256 st = u256_zero(), pw = u256_zero();
      if (ok) {
        st = u256_montmul(s_, r2, q, q0inv);
        P = valid ? u256_montmul(P, st, q, q0inv) : st;
        pw = P;
        valid |= 1u << (i - pos);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { o[k] = pw.w[k]; o[8 + k] = st.w[k]; o[16 + k] = r.w[k]; }
    }
    if (valid) {
      U256 inv;
      const bool inv_ok = u256_modinv_odd(u256_montmul(P, one, q, q0inv), q, inv);
      U256 I = u256_montmul(inv, r2, q, q0inv);      // (product of the s)^-1, Montgomery form
      for (uint32_t i = re; i-- > pos;) {
        if (!((valid >> (i - pos)) & 1u)) continue;
        uint32_t* o = dsa_u + (uint64_t)(tile0 + order[i]) * DSA_U_WORDS;
        U256 st, r, w, u2;
#pragma unroll
        for (int k = 0; k < 8; ++k) { st.w[k] = o[8 + k]; r.w[k] = o[16 + k]; }
        if (inv_ok) {
          const uint32_t below = valid & ((1u << (i - pos)) - 1u);      // valid entries before this one
          U256 wt = I;
          if (below) {
            const uint32_t pv = pos + (31u - (uint32_t)__builtin_clz(below));
            const uint32_t* op = dsa_u + (uint64_t)(tile0 + order[pv]) * DSA_U_WORDS;
            U256 Pp;
#pragma unroll
            for (int k = 0; k < 8; ++k) Pp.w[k] = op[k];
            wt = u256_montmul(I, Pp, q, q0inv);
          }
          I = u256_montmul(I, st, q, q0inv);
          w = u256_montmul(wt, one, q, q0inv);
          u2 = u256_montmul(wt, r, q, q0inv);
        } else {
          // no inverse for the product: this signature on its own
          const bool ok1 = u256_modinv_odd(u256_montmul(st, one, q, q0inv), q, w);
          if (ok1) u2 = u256_mulmod_mont(w, r, q, q0inv, r2);
          else { w = u256_zero(); u2 = u256_zero(); }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { o[k] = w.w[k]; o[8 + k] = u2.w[k]; }
      }
    }
    pos = re;
  }
}

// After the digests: u1 = z*w mod q replaces w in the row; refused signatures get their final status here.
__global__ void __launch_bounds__(64) k_dsa_mul(SigRec* __restrict__ recs, const uint32_t* __restrict__ dsa_list,
                                                const uint32_t* __restrict__ pk_count, const uint32_t* __restrict__ pk_start, KeyTableDev kt,
                                                const uint32_t* __restrict__ digests, uint32_t* __restrict__ dsa_u) {
  const uint32_t di = pk_start[1] + blockIdx.x * blockDim.x + threadIdx.x;
  if (di >= pk_count[1]) return;
  const uint32_t ri = dsa_list[di];
  const SigRec rec = recs[ri];
  if (rec.status != ST_PENDING_RSA) return;   // hash tag mismatch etc.: already final
  uint32_t* o = dsa_u + (uint64_t)di * DSA_U_WORDS;
  U256 q, w;
  for (int i = 0; i < 8; ++i) { q.w[i] = kt.q_words[(uint64_t)rec.key_slot * 8 + i]; w.w[i] = o[i]; }
  if (u256_is_zero(w)) { recs[ri].status = ST_BAD_SIG; return; }
  // z = leftmost min(len(digest), bytes(q)) digest bytes (openpgp truncates, then dsa.Verify again)
  const HashInfo hi = hash_info(rec.hash_id);
  const uint32_t zlen = min(hi.dlen, kt.q_bits[rec.key_slot] >> 3);
  U256 z;
  u256_from_be((const uint8_t*)(digests + (uint64_t)ri * 16), zlen, z);
  while (u256_cmp(z, q) >= 0) u256_sub(z, q);        // z < 2^bits(q) < 2q: at most one round
  const uint32_t* qc = dsa_qconst(kt, (uint32_t)rec.key_slot);
  U256 r2;
  for (int i = 0; i < 8; ++i) r2.w[i] = qc[i];
  const U256 u1 = u256_mulmod_mont(w, z, q, qc[8], r2);
  for (int i = 0; i < 8; ++i) o[i] = u1.w[i];
}

// Fixed-base window tables for one DSA key (see KeyTableDev::dsa_comb).  One group per (slot, base, window, part):
// B = seed^(2^(wbits*w)) by repeated squaring; the part's first entry B^(d0) by square-and-multiply over d0; then the
// part's entries by repeated multiplication with B.  `parts` splits the 2^wbits - 1 digits of a window so that the 16-bit
// layout (65,535 entries per window) is built by 16x more groups with 16x shorter chains.
// <19, 4>: keys with p <= 2048 bits; <14, 8>: p <= 3072 bits (the host hands each instantiation the new slots of its class).
// Runs once per new DSA key (bftkv_gpu_keyring_set / certificate upload), never on the verify path.
template <int L, int TPI>
__global__ void __launch_bounds__(RSA_BLOCK) k_dsa_build_comb(uint32_t n_new, const uint32_t* __restrict__ new_slots /*[n_new]*/,
                                                              const uint32_t* __restrict__ slot_key /*[n_new] key table row*/,
                                                              KeyTableDev kt, uint32_t* __restrict__ comb, uint32_t parts) {
  constexpr int NL = L * TPI, GROUPS = RSA_BLOCK / TPI;
  __shared__ uint32_t a_sh[GROUPS * NL];
  const uint32_t E = kt.dsa_entry_limbs;
  const uint32_t wbits = kt.dsa_wbits, nwin = dsa_nwin(wbits), nent = (1u << wbits) - 1u;
  const uint32_t n_quads = n_new * 2u * nwin * parts;
  const uint32_t quad = threadIdx.x / TPI;
  const int qlane = threadIdx.x % TPI;
  const uint32_t gq0 = blockIdx.x * GROUPS + quad;
  const bool active = gq0 < n_quads;
  const uint32_t gq = active ? gq0 : (n_quads - 1);
  const uint32_t part = gq % parts, gw = gq / parts;
  const uint32_t which = gw / (2u * nwin), base = (gw / nwin) & 1u, w = gw % nwin;
  const uint32_t key = slot_key[which];
  uint32_t* a_lds = a_sh + quad * NL + qlane * L;
  const uint32_t* a_rd = a_sh + quad * NL;
  uint32_t n[L], b[L], y[L], t[L], one[L];
  const uint32_t* np = kt.n_limbs + (uint64_t)key * MONT_NMAX + qlane * L;
  const uint32_t* sp = kt.dsa_tab + ((uint64_t)key * 2 + base) * DSA_N_BIG + qlane * L;
  const uint32_t* rp = kt.r2_limbs + (uint64_t)key * MONT_NMAX + qlane * L;
#pragma unroll
  for (int k = 0; k < L; ++k) { n[k] = np[k]; y[k] = sp[k]; }
  const uint32_t n0inv = kt.n0inv[key];
  const uint32_t nsq = wbits * w;
  for (uint32_t i = 0; i < wbits * (nwin - 1); ++i) {     // uniform trip count; groups past their own count keep y
#pragma unroll
    for (int k = 0; k < L; ++k) { a_lds[k] = y[k]; b[k] = y[k]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    mont_mul<L, TPI>(t, a_rd, b, n, n0inv, qlane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (i < nsq) {
#pragma unroll
      for (int k = 0; k < L; ++k) y[k] = t[k];
    }
  }
  // y = B.  Digits of this part: d0 .. d0 + per - 1 (digit 0 has no entry); b = B^d0 in Montgomery form
  const uint32_t per = (nent + 1u) / parts;               // 2^wbits / parts
  const uint32_t d0 = part * per;
  // one = R mod p = mont(1, R^2)
#pragma unroll
  for (int k = 0; k < L; ++k) { a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u; b[k] = rp[k]; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  mont_mul<L, TPI>(one, a_rd, b, n, n0inv, qlane);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int k = 0; k < L; ++k) b[k] = one[k];
  for (int bit = (int)wbits - 1; bit >= 0; --bit) {         // b = B^d0, left-to-right (uniform trip count, per-group select)
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = b[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    mont_mul<L, TPI>(t, a_rd, b, n, n0inv, qlane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int k = 0; k < L; ++k) { b[k] = t[k]; a_lds[k] = y[k]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    mont_mul<L, TPI>(t, a_rd, b, n, n0inv, qlane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if ((d0 >> bit) & 1u) {
#pragma unroll
      for (int k = 0; k < L; ++k) b[k] = t[k];
    }
  }
  uint32_t* out = comb + (uint64_t)new_slots[which] * dsa_slot_stride(wbits, E) + ((uint64_t)(base * nwin + w) * nent) * E + qlane * L;
  // The window k_dsa_modexp multiplies by LAST (base 1, top window) is stored in plain form, B^d instead of B^d R: that
  // product then leaves the Montgomery domain by itself.  One more product per entry here, for the waves that hold such a group.
  const bool plain = dsa_plain_window(base, w, nwin);
  const bool any_plain = __any(plain);
  for (uint32_t j = 0; j < per; ++j) {
    const uint32_t d = d0 + j;                        // entry index d - 1 holds B^d
    if (any_plain) {
#pragma unroll
      for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      mont_mul<L, TPI>(t, a_rd, b, n, n0inv, qlane);          // B^d R * 1 * R^-1
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      canonicalize<L, TPI>(t, qlane);
    }
    if (active && d >= 1) {
#pragma unroll
      for (int k = 0; k < L; ++k) out[(uint64_t)(d - 1) * E + k] = plain ? t[k] : b[k];
    }
#pragma unroll
    for (int k = 0; k < L; ++k) a_lds[k] = y[k];     // a = B for the chain
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    mont_mul<L, TPI>(t, a_rd, b, n, n0inv, qlane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int k = 0; k < L; ++k) b[k] = t[k];
  }
}

template <int TPI>
__device__ __forceinline__ uint64_t grp_sum64(uint64_t v) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  v += ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, DPP_QUAD_SWAP1, 0xF, 0xF, false) << 32) |
       (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, DPP_QUAD_SWAP1, 0xF, 0xF, false);
  lo = (uint32_t)v; hi = (uint32_t)(v >> 32);
  v += ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, DPP_QUAD_SWAP2, 0xF, 0xF, false) << 32) |
       (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, DPP_QUAD_SWAP2, 0xF, 0xF, false);
  if constexpr (TPI == 8) {     // the other quad of the group
    lo = (uint32_t)v; hi = (uint32_t)(v >> 32);
    v += ((uint64_t)(uint32_t)__shfl_xor((int)hi, 4) << 32) | (uint32_t)__shfl_xor((int)lo, 4);
  }
  return v;
}

// A key table that holds DSA keys of BOTH size classes: the positions of the DSA work list [start, count) sorted by class into two
// compact index lists, so that each instantiation of k_dsa_modexp runs waves full of its own rows (one list for both, rows of the
// other class riding along, cost a four-size ring 11.8 ms where the two pure rings take 5.6 and 9.4: profiles/r06_dsa_rates_…).
__global__ void __launch_bounds__(256) k_dsa_split(const SigRec* __restrict__ recs, const uint32_t* __restrict__ dsa_list,
                                                   const uint32_t* __restrict__ pk_count, const uint32_t* __restrict__ pk_start, KeyTableDev kt,
                                                   uint32_t* __restrict__ idx_small, uint32_t* __restrict__ idx_big, uint32_t* __restrict__ cls_count /*[2]*/) {
  const uint32_t di = pk_start[1] + blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = di < pk_count[1];
  const bool big = in && kt.mod_bits[(uint32_t)recs[dsa_list[di]].key_slot] > 2048u;
  const uint32_t a = wave_alloc(cls_count, in && !big);
  if (in && !big) idx_small[a] = di;
  const uint32_t b = wave_alloc(cls_count + 1, big);
  if (big) idx_big[b] = di;
}

// v = g^u1 * y^u2 mod p from the per-key fixed-base tables: one table multiplication per non-zero
// window digit of u1 and u2 but the first (<= 2 * 256/wbits - 1), no squarings.  A table row (304 B; 448 B for a 3072-bit p) is
// read straight from HBM/MALL into the group's LDS slot; a wave skips a (window, base) step when all its digits are zero.
// The tail finishes dsa.Verify in place: v mod q through the per-key table 2^(28 j) mod q (each group lane
// folds its L limbs, the group adds up, 35 shift-subtract steps finish), then (v mod q) == r.
// <19, 4> takes the rows of the DSA work list whose key has p <= 2048 bits, <14, 8> those with p <= 3072 bits (launched only
// when such a key is in the table); a table that holds both classes hands each its own compact index list (k_dsa_split).
template <int L, int TPI>
__global__ void __launch_bounds__(RSA_BLOCK) k_dsa_modexp(SigRec* __restrict__ recs, const uint32_t* __restrict__ dsa_list,
                                                          const uint32_t* __restrict__ pk_count, const uint32_t* __restrict__ pk_start, KeyTableDev kt,
                                                          const uint32_t* __restrict__ dsa_u, const uint32_t* __restrict__ cls_idx = nullptr,
                                                          const uint32_t* __restrict__ cls_count = nullptr) {
  constexpr int NL = L * TPI, GROUPS = RSA_BLOCK / TPI;
  constexpr bool BIG = NL > (int)DSA_N_SMALL;
  __shared__ uint32_t a_sh[GROUPS * NL];
  // cls_idx (a key table that MIXES the size classes, k_dsa_split): the work-list positions of this class, compacted -- a wave then
  // holds rows of its own class only
  const uint32_t count = cls_idx ? *cls_count : pk_count[1], start = cls_idx ? 0u : pk_start[1];
  if (start + blockIdx.x * GROUPS >= count) return;
  const uint32_t quad = threadIdx.x / TPI;
  const int qlane = threadIdx.x % TPI;
  const uint32_t gq = start + blockIdx.x * GROUPS + quad;
  const bool active = gq < count;
  const uint32_t di = cls_idx ? cls_idx[active ? gq : (count - 1)] : (active ? gq : (count - 1));
  const uint32_t ri = dsa_list[di];
  const SigRec rec = recs[ri];
  const uint32_t key = (uint32_t)rec.key_slot;
  const bool mine = active && (kt.mod_bits[key] > 2048u) == BIG;      // this instantiation's size class
  if (!__any(mine)) return;
  uint32_t* a_lds = a_sh + quad * NL + qlane * L;
  const uint32_t* a_rd = a_sh + quad * NL;
  uint32_t n[L], b[L], y[L], t[L];
  const uint32_t* np = kt.n_limbs + (uint64_t)key * MONT_NMAX + qlane * L;
  const uint32_t E = kt.dsa_entry_limbs;
  const uint32_t wbits = kt.dsa_wbits, nwin = dsa_nwin(wbits), nent = (1u << wbits) - 1u;
  const uint32_t* slot_base = dsa_slot_base(kt, key);
  const uint32_t* tab = slot_base + qlane * L;
#pragma unroll
  for (int k = 0; k < L; ++k) n[k] = mine ? np[k] : (qlane == 0 && k == 0 ? 1u : 0u);     // (a row of the other class: modulus 1, digits 0)
  const uint32_t n0inv = kt.n0inv[key];
  const uint32_t* up = dsa_u + (uint64_t)di * DSA_U_WORDS;
  const bool live = mine && rec.status == ST_PENDING_RSA;     // refused rows ride along with all-zero digits
  auto digit = [&](uint32_t step) -> uint32_t {
    const uint32_t bitpos = (step >> 1) * wbits, wi = bitpos >> 5, sh = bitpos & 31u;
    if (!live) return 0u;
    const uint32_t* e = up + (step & 1u) * 8;                       // u1 or u2: eight 32-bit words
    const uint32_t lo = e[wi], hi = (wi < 7u && sh + wbits > 32u) ? e[wi + 1] : 0u;     // (a window may straddle two words)
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & nent;
  };
  auto entry = [&](uint32_t step, uint32_t d) -> const uint32_t* {
    return tab + ((uint64_t)((step & 1u) * nwin + (step >> 1)) * nent + (d ? d - 1 : 0)) * E;
  };
  // The chain is 2*nwin - 1 products at most (31 for 16-bit windows): a signature's first non-zero digit just takes its
  // table entry as the starting value, and the last window's entries are stored in plain form (k_dsa_build_comb) so that
  // the product by them also leaves the Montgomery domain.  One multiplier call site; a wave skips a step no lane needs.
  const uint32_t last = 2u * nwin - 1u;
  bool started = false;
#pragma unroll
  for (int k = 0; k < L; ++k) y[k] = 0;
  for (uint32_t step = 0; step <= last; ++step) {
    const uint32_t d = digit(step);
    const bool need = mine && ((step == last) || d != 0);      // the last product always happens: a zero digit multiplies by the plain 1
    if (!__any(need)) continue;
    const uint32_t* tp = entry(step, d);
    if (mine) {
#pragma unroll
      for (int k = 0; k < L; ++k) a_lds[k] = tp[k];    // a zero digit reads entry 1 and (but for the last step) drops the product
    }
    if (step == last && __any(d == 0)) {
      if (d == 0) {
#pragma unroll
        for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (__any(need && started)) {
#pragma unroll
      for (int k = 0; k < L; ++k) b[k] = y[k];
      mont_mul<L, TPI>(t, a_rd, b, n, n0inv, qlane);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (__any(need && !started)) {                   // first non-zero digit of some signature: the entry is its starting value
      if (need) {
#pragma unroll
        for (int k = 0; k < L; ++k) y[k] = started ? t[k] : a_lds[k];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    } else if (need) {
#pragma unroll
      for (int k = 0; k < L; ++k) y[k] = t[k];
    }
    started = started || need;
  }
  // y = g^u1 y^u2 mod p, possibly + p (mont_mul leaves values below p(1 + 2^-79)): the mod-q fold below needs the residue
#pragma unroll
  for (int k = 0; k < L; ++k) t[k] = y[k];
  canonicalize<L, TPI>(t, qlane);
  reduce_once<L, TPI>(t, n, qlane);
  uint32_t diff = 0;
#pragma unroll
  for (int k = 0; k < L; ++k) diff |= t[k];
  diff = grp_or<TPI>(diff);                  // 0: v = 0, and r > 0 can never match
  // v mod q: sum_j v_j * (2^(28 j) mod q) over 10 radix-2^28 columns (NL terms of < 2^56 each)
  uint64_t col[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) col[j] = 0;
  const uint32_t* pw = slot_base + dsa_comb_limbs_per_key(wbits, E) + (qlane * L) * 10;
  if (mine) {
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
      for (int j = 0; j < 10; ++j) col[j] = mad64(t[k], pw[k * 10 + j], col[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 10; ++j) col[j] = grp_sum64<TPI>(col[j]);
  // columns -> 32-bit words W (value < q * 2^34.3; 112 limbs: q * 2^34.9)
  uint32_t W[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) W[i] = 0;
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const uint64_t sacc = (j < 10 ? col[j] : 0ull) + carry;
    const uint32_t limb = (uint32_t)sacc & MONT_MASK;
    carry = sacc >> MONT_W;
    const int bit = MONT_W * j, wi = bit >> 5, sh = bit & 31;
    W[wi] |= limb << sh;
    if (sh > 4 && wi + 1 < 12) W[wi + 1] |= limb >> (32 - sh);
  }
  U256 q, rr, acc;
#pragma unroll
  for (int i = 0; i < 8; ++i) { q.w[i] = kt.q_words[(uint64_t)key * 8 + i]; rr.w[i] = up[16 + i]; }
  // acc = top part (value >> 35 < q), then 35 double-and-reduce steps for the low bits
#pragma unroll
  for (int i = 0; i < 8; ++i) acc.w[i] = (W[i + 1] >> 3) | (W[i + 2] << 29);
  if (u256_cmp(acc, q) >= 0) u256_sub(acc, q);
  const uint64_t low = ((uint64_t)(W[1] & 7u) << 32) | W[0];
#pragma unroll 1
  for (int bit = 34; bit >= 0; --bit) {
    uint32_t cbit = u256_shl1(acc);
    acc.w[0] |= (uint32_t)(low >> bit) & 1u;
    if (cbit || u256_cmp(acc, q) >= 0) u256_sub(acc, q);
  }
  if (live && qlane == 0) recs[ri].status = (diff != 0 && u256_cmp(acc, rr) == 0) ? ST_OK : ST_BAD_SIG;
}


// ------------------------------------------------------------------------------------------------
// generic modular exponentiation  out = base^exp mod n   (corpus signing; threshold-RSA partials,
// crypto/threshold/rsa/rsa.go:161-171).  Square-and-always-multiply with a per-lane select on the
// exponent bit, so signatures with different exponents share a wave without divergence.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RSA_BLOCK) k_modexp(uint32_t n_ops, const uint32_t* __restrict__ base_limbs /*[n_ops][76]*/,
                                                      const uint32_t* __restrict__ mod_idx, const uint32_t* __restrict__ n_limbs,
                                                      const uint32_t* __restrict__ r2_limbs, const uint32_t* __restrict__ n0inv_tab,
                                                      const uint32_t* __restrict__ exp_words /*[n_mods | n_ops][exp_nwords] little-endian*/,
                                                      uint32_t exp_nwords, uint32_t exp_per_op, uint32_t* __restrict__ out_limbs) {
  __shared__ uint32_t a_sh[QUADS_PER_BLOCK * MONT_N];
  __shared__ uint32_t x_sh[QUADS_PER_BLOCK * MONT_N];
  constexpr int L = MONT_L;
  const uint32_t quad = threadIdx.x >> 2;
  const int qlane = threadIdx.x & 3;
  const uint32_t gq = blockIdx.x * QUADS_PER_BLOCK + quad;
  const bool active = gq < n_ops;
  const uint32_t op = active ? gq : (n_ops - 1);
  const uint32_t mi = mod_idx[op];
  uint32_t* a_lds = a_sh + quad * MONT_N + qlane * L;
  uint32_t* x_lds = x_sh + quad * MONT_N + qlane * L;
  const uint32_t* a_rd = a_sh + quad * MONT_N;
  const uint32_t* x_rd = x_sh + quad * MONT_N;
  uint32_t n[L], b[L], y[L], t[L];
#pragma unroll
  for (int k = 0; k < L; ++k) n[k] = n_limbs[(uint64_t)mi * MONT_N + qlane * L + k];
  const uint32_t n0inv = n0inv_tab[mi];
  const uint32_t* ew = exp_words + (uint64_t)(exp_per_op ? op : mi) * exp_nwords;   // per modulus (RSA private keys) or per operation
  // steps: -2: xR = mont(x, R^2); -1: y = mont(1, R^2) = R mod n; then per exponent bit a
  // squaring (even step) and a multiplication by xR (odd step); last: mont(y, 1).
  const int nbits = (int)exp_nwords * 32;
  for (int step = -2; step <= 2 * nbits; ++step) {
    const uint32_t* ard = a_rd;
    if (step == -2) {
#pragma unroll
      for (int k = 0; k < L; ++k) { b[k] = r2_limbs[(uint64_t)mi * MONT_N + qlane * L + k]; a_lds[k] = base_limbs[(uint64_t)op * MONT_N + qlane * L + k]; }
    } else if (step == -1 || step == 2 * nbits) {
#pragma unroll
      for (int k = 0; k < L; ++k) a_lds[k] = (qlane == 0 && k == 0) ? 1u : 0u;
      if (step >= 0) {
#pragma unroll
        for (int k = 0; k < L; ++k) b[k] = y[k];
      }
    } else if ((step & 1) == 0) {
#pragma unroll
      for (int k = 0; k < L; ++k) { a_lds[k] = y[k]; b[k] = y[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < L; ++k) b[k] = y[k];
      ard = x_rd;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (step >= 0 && step < 2 * nbits && (step & 1) == 0) mont_mul<L, MONT_TPI, true>(t, ard, b, n, n0inv, qlane);   // y * y
    else mont_mul(t, ard, b, n, n0inv, qlane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (step == -2) {
#pragma unroll
      for (int k = 0; k < L; ++k) x_lds[k] = t[k];
    } else if (step >= 0 && step < 2 * nbits && (step & 1)) {
      const int bi = nbits - 1 - (step >> 1);
      const bool bit = (ew[bi >> 5] >> (bi & 31)) & 1u;
      if (bit) {
#pragma unroll
        for (int k = 0; k < L; ++k) y[k] = t[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < L; ++k) y[k] = t[k];
    }
  }
  canonicalize(y, qlane);
  // y <= n, and y == n only when the value is 0 mod n
  uint32_t diff = 0;
#pragma unroll
  for (int k = 0; k < L; ++k) diff |= y[k] ^ n[k];
  diff = quad_or(diff);
  if (active) {
#pragma unroll
    for (int k = 0; k < L; ++k) out_limbs[(uint64_t)op * MONT_N + qlane * L + k] = (diff == 0) ? 0u : y[k];
  }
}

// big-endian bytes <-> radix-2^28 limbs (one thread per limb / per byte)
__global__ void k_bytes_to_limbs(const uint8_t* __restrict__ src, uint32_t nbytes, uint32_t n_nums, uint32_t* __restrict__ dst) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nums * MONT_N) return;
  uint32_t num = i / MONT_N, j = i % MONT_N;
  const uint8_t* p = src + (uint64_t)num * nbytes;
  auto bf = [&](uint32_t k) -> uint32_t { return k < nbytes ? p[nbytes - 1 - k] : 0u; };
  dst[i] = limb28(bf, (int)j);
}
__global__ void k_limbs_to_bytes(const uint32_t* __restrict__ src, uint32_t nbytes, uint32_t n_nums, uint8_t* __restrict__ dst) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nums * nbytes) return;
  uint32_t num = i / nbytes, k = i % nbytes;      // k: byte index from the LSB
  uint32_t bit = k * 8, j = bit / 28, sh = bit % 28;
  const uint32_t* l = src + (uint64_t)num * MONT_N;
  uint64_t v = (j < MONT_N ? l[j] : 0u);
  if (j + 1 < MONT_N) v |= (uint64_t)l[j + 1] << 28;
  dst[(uint64_t)num * nbytes + (nbytes - 1 - k)] = (uint8_t)(v >> sh);
}

// ------------------------------------------------------------------------------------------------
// quorum tally: one wave per item
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tally(const SigRec* __restrict__ recs, const uint32_t* __restrict__ rec_base,
                                               const uint32_t* __restrict__ counts, uint32_t n_items, KeyTableDev kt,
                                               QuorumDev q, uint8_t* __restrict__ verdict, uint32_t* __restrict__ n_verified,
                                               uint32_t* __restrict__ clique_counts /*[n_items][MAX_QC] or null*/) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  if (wave >= n_items) return;
  const uint32_t base = rec_base[wave], cnt = counts[wave];
  uint32_t cq[MAX_QC];
#pragma unroll
  for (int c = 0; c < MAX_QC; ++c) cq[c] = 0;
  uint32_t total_ok = 0;
  uint32_t first_suff = 0xFFFFFFFFu;   // number of verified signers consumed when IsSufficient first held
  for (uint32_t off = 0; off < cnt; off += 64) {
    const uint32_t i = off + lane;
    bool ok = false;
    uint32_t ent = 0;
    if (i < cnt) {
      const SigRec r = recs[base + i];
      ok = (r.status == ST_OK);
      if (ok) ent = kt.entity[r.key_slot];
    }
    const uint64_t okm = __builtin_amdgcn_ballot_w64(ok);
    const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    bool reached = false;
#pragma unroll
    for (int c = 0; c < MAX_QC; ++c) {
      if (c < q.n_qcs) {
        const bool mem = ok && q.member[(uint64_t)c * q.n_entities + ent];
        const uint64_t mm = __builtin_amdgcn_ballot_w64(mem);
        // inclusive running count of clique-c members among verified signers up to this lane
        const uint32_t run = cq[c] + (uint32_t)__builtin_popcountll(mm & below) + (mem ? 1u : 0u);
        if (ok && q.suff[c] > 0 && run >= (uint32_t)q.suff[c]) reached = true;
        cq[c] += (uint32_t)__builtin_popcountll(mm);
      }
    }
    const uint64_t rm = __builtin_amdgcn_ballot_w64(reached);
    if (rm != 0 && first_suff == 0xFFFFFFFFu) {
      const uint32_t fl = (uint32_t)__builtin_ctzll(rm);
      first_suff = total_ok + (uint32_t)__builtin_popcountll(okm & ((fl == 63) ? ~0ull : ((1ull << (fl + 1)) - 1)));
    }
    total_ok += (uint32_t)__builtin_popcountll(okm);
  }
  if (lane == 0) {
    // wotqs.go:144-185 over the full verified list (counts are monotone, so the early exit of
    // crypto_pgp.go:493 fires iff IsSufficient holds for the full list)
    bool is_quorum = q.n_qcs > 0, is_thr = q.n_qcs > 0, is_suff = false, reject = true;
    for (int c = 0; c < q.n_qcs; ++c) {
      if (q.f[c] > 0 && (int32_t)cq[c] < q.min[c]) is_quorum = false;
      if (q.threshold[c] > 0 && (int32_t)cq[c] < q.threshold[c]) is_thr = false;
      if (q.suff[c] > 0 && (int32_t)cq[c] >= q.suff[c]) is_suff = true;
      if (q.f[c] == 0 || (int32_t)cq[c] <= q.f[c]) reject = false;
    }
    verdict[wave] = (is_quorum ? V_IS_QUORUM : 0) | (is_thr ? V_IS_THRESHOLD : 0) | (is_suff ? V_IS_SUFFICIENT : 0) |
                    (reject ? V_REJECT : 0);
    n_verified[wave] = is_suff ? first_suff : total_ok;
    if (clique_counts)
      for (int c = 0; c < MAX_QC; ++c) clique_counts[(uint64_t)wave * MAX_QC + c] = cq[c];
  }
}

// ------------------------------------------------------------------------------------------------
// two-phase planning of the public-key work (CollectiveSignature.Verify only)
// ------------------------------------------------------------------------------------------------
// PGPCollectiveSignature.Verify returns at the first packet after which IsSufficient holds (crypto_pgp.go:491-496) and
// never reads the rest of ss.Data -- at n = 64 that is 43 of 53 packets on average.  Phase 1 therefore queues, per item,
// the candidate packets (parsed, key found, everything checked that precedes the hash) up to the position where the tally
// WOULD become sufficient if they all verified, plus `margin` more clique members so that an isolated bad signature does
// not cost a second pass.  After phase 1's tally, phase 2 queues what is left of the items that are still insufficient.
// The reference's exit position is never before phase 1's cut, so every packet the reference examines is examined here.
struct PlanArgs {
  SigRec* recs; const uint32_t* rec_base; const uint32_t* counts; uint32_t n_items;
  uint32_t *pk_list, *dsa_list, *pk_list3072, *pk_list4096;
  uint32_t* pk_count;          // [0..3] list lengths; [8..11] receive the phase-1 lengths (= phase 2's start)
  uint32_t* plan_cut;          // [n_items] records covered so far
  const uint8_t* verdict;      // phase 2: verdict bits of the phase-1 tally
  uint32_t margin;
  uint32_t* mail;              // phase 1: mapped host word that receives 0x80000000 | (some signature uses a hash other than SHA-256)
};

// One wave per item, 16 items per block.  Work-list slots come from ONE atomic per block and list: every wave first counts
// what it will queue, the block prefix-sums the counts in LDS, then the waves write.  (Allocating per 64-packet chunk
// with one atomic per wave made this kernel nothing but a queue of same-address atomics: ~11 ns each, 4.3 ms for the
// 500k chunks of a cfg-4 batch.)
constexpr int PLAN_BLOCK = 1024, PLAN_ITEMS = PLAN_BLOCK / 64;
template <int PHASE>
__global__ void __launch_bounds__(PLAN_BLOCK) k_plan(PlanArgs a, KeyTableDev kt, QuorumDev q) {
  __shared__ uint32_t cnt_sh[PLAN_ITEMS][4];
  const uint32_t wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t item = blockIdx.x * PLAN_ITEMS + wib;
  // the parse is complete: tell the host whether k_digest_other has anything to do, so that it need not launch it
  if (PHASE == 1 && a.mail && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(a.mail, 0x80000000u | (a.pk_count[4] ? 1u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const bool have = item < a.n_items;
  const uint32_t base = have ? a.rec_base[item] : 0, cnt = have ? a.counts[item] : 0;
  uint32_t* const lists[4] = {a.pk_list, a.dsa_list, a.pk_list3072, a.pk_list4096};
  const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // ---- pass A: which records does this item queue?  [from, covered) restricted to candidates (PHASE 2: still pending)
  uint32_t from = 0, covered = cnt;
  if (PHASE == 1) {
    uint32_t cq[MAX_QC];
#pragma unroll
    for (int c = 0; c < MAX_QC; ++c) cq[c] = 0;
    bool done = false;
    for (uint32_t off = 0; off < cnt && !done; off += 64) {
      const uint32_t i = off + lane;
      uint32_t kind = 0, ent = 0;
      if (i < cnt) {
        kind = a.recs[base + i].q_kind1;
        if (kind) ent = kt.entity[a.recs[base + i].key_slot];
      }
      bool reached = false;
#pragma unroll
      for (int c = 0; c < MAX_QC; ++c) {
        if (c < q.n_qcs) {
          const bool mem = kind != 0 && q.member[(uint64_t)c * q.n_entities + ent];
          const uint64_t mm = __builtin_amdgcn_ballot_w64(mem);
          const uint32_t run = cq[c] + (uint32_t)__builtin_popcountll(mm & below) + (mem ? 1u : 0u);
          if (mem && q.suff[c] > 0 && run >= (uint32_t)q.suff[c] + a.margin) reached = true;
          cq[c] += (uint32_t)__builtin_popcountll(mm);
        }
      }
      const uint64_t rm = __builtin_amdgcn_ballot_w64(reached);
      if (rm) { done = true; covered = off + (uint32_t)__builtin_ctzll(rm) + 1; }
    }
  } else {
    from = have ? a.plan_cut[item] : 0;
    if (!have || (a.verdict[item] & V_IS_SUFFICIENT)) from = covered;     // nothing left to do for this item
  }
  auto wants = [&](uint32_t i, uint32_t& kind) -> bool {
    kind = 0;
    if (i >= covered) return false;
    if (PHASE == 1) { kind = a.recs[base + i].q_kind1; return kind != 0; }
    const SigRec r = a.recs[base + i];
    kind = r.q_kind1;
    return kind != 0 && !r.queued && r.status == ST_PENDING_RSA;          // hash tag matched (the hash stream has been joined)
  };
  uint32_t n_k[4] = {0, 0, 0, 0};
  for (uint32_t off = from; off < covered; off += 64) {
    uint32_t kind;
    const bool w = wants(off + lane, kind);
#pragma unroll
    for (int k = 0; k < 4; ++k) n_k[k] += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(w && kind == (uint32_t)k + 1));
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) cnt_sh[wib][k] = n_k[k];
  }
  __syncthreads();
  // ---- block allocation: thread k < 4 owns list k
  if (threadIdx.x < 4) {
    uint32_t tot = 0;
    for (int w = 0; w < PLAN_ITEMS; ++w) { const uint32_t v = cnt_sh[w][threadIdx.x]; cnt_sh[w][threadIdx.x] = tot; tot += v; }
    const uint32_t b0 = tot ? atomicAdd(a.pk_count + threadIdx.x, tot) : 0u;
    for (int w = 0; w < PLAN_ITEMS; ++w) cnt_sh[w][threadIdx.x] += b0;
  }
  __syncthreads();
  // ---- pass B: write
  uint32_t run_k[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) run_k[k] = cnt_sh[wib][k];
  for (uint32_t off = from; off < covered; off += 64) {
    const uint32_t i = off + lane;
    uint32_t kind;
    const bool w = wants(i, kind);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool mine = w && kind == (uint32_t)k + 1;
      const uint64_t m = __builtin_amdgcn_ballot_w64(mine);
      if (mine) {
        const uint32_t idx = run_k[k] + (uint32_t)__builtin_popcountll(m & below);
        a.recs[base + i].pk_idx = idx; a.recs[base + i].queued = 1; lists[k][idx] = base + i;
      }
      run_k[k] += (uint32_t)__builtin_popcountll(m);
    }
  }
  if (have && lane == 0) a.plan_cut[item] = (PHASE == 1) ? covered : cnt;
}

// the work-list lengths at the end of phase 1 become phase 2's start offsets
__global__ void k_plan_snapshot(uint32_t* __restrict__ pk_count) {
  if (threadIdx.x < 4) pk_count[8 + threadIdx.x] = pk_count[threadIdx.x];
}

// per-item fence flag for the caller: the walk met a framing it does not follow, or the parse met a fenced packet shape
// rehash_bits (calls that handed over SHA-256 midstates instead of payloads): bit 1 of out[i] = a signature of the item asks
// for another hash, which needs the payload itself -- the caller submits the item again with its bytes
__global__ void k_fenced_out(const uint8_t* __restrict__ item_flags, const uint32_t* __restrict__ item_hash_mask, uint32_t n,
                             uint8_t* __restrict__ out, uint32_t rehash_bits = 0) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (((item_flags[i] & 4) || (item_hash_mask[i] & ITEM_FENCED)) ? 1 : 0) | ((item_hash_mask[i] & rehash_bits) ? 2 : 0);
}

// Segmented payloads (bftkv_gpu_collective_verify_segments): payload i = prefix_i || shared[seg_i].  The signed bytes of a bftkv
// write end in chunk(Cert) (packet/packet.go:192-212) -- the SAME client certificate behind every write of that client -- so the
// caller sends each distinct tail once and this kernel lays the payloads out in HBM as the unsegmented call would have received
// them: everything downstream (midstates, digests, the text-mode hashes) reads the same bytes at the same offsets.
// One wave per item; destination dwords are written whole (a lane per dword, the source funnelled through v_alignbyte), the up
// to three bytes before the first and after the last aligned dword one by one.  Source buffers carry 64 bytes of slack.
__device__ __forceinline__ void wave_copy_bytes(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint64_t len, uint32_t lane) {
  if (len == 0) return;
  const uint64_t head = ((4u - ((uintptr_t)dst & 3u)) & 3u) < len ? ((4u - ((uintptr_t)dst & 3u)) & 3u) : len;
  if (lane < head) dst[lane] = src[lane];
  const uint64_t body = (len - head) & ~(uint64_t)3;
  uint32_t* __restrict__ d4 = (uint32_t*)(dst + head);
  const uint8_t* sp = src + head;
  const uint32_t mis = (uint32_t)((uintptr_t)sp & 3u);
  const uint32_t* __restrict__ s4 = (const uint32_t*)(sp - mis);          // aligned down: stays inside the (256-byte aligned) buffer
  for (uint64_t w = lane; w < body / 4; w += 64) {
    const uint32_t lo = s4[w], hi = mis ? s4[w + 1] : 0u;                // (the dword past the end is the buffer's slack at worst)
    d4[w] = mis ? __builtin_amdgcn_alignbyte(hi, lo, mis) : lo;
  }
  const uint64_t done = head + body;
  if (lane < len - done) dst[done + lane] = src[done + lane];
}
__global__ void __launch_bounds__(256) k_expand_segments(const uint8_t* __restrict__ prefix, const uint64_t* __restrict__ prefix_off,
                                                         const uint8_t* __restrict__ shared, const uint64_t* __restrict__ shared_off,
                                                         const uint32_t* __restrict__ seg, const uint64_t* __restrict__ tbs_off,
                                                         uint32_t i0, uint32_t n, uint8_t* __restrict__ out) {
  const uint32_t item = i0 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63u;
  if (item >= i0 + n) return;
  const uint64_t p0 = prefix_off[item], pl = prefix_off[item + 1] - p0;
  uint8_t* dst = out + tbs_off[item];
  wave_copy_bytes(dst, prefix + p0, pl, lane);
  const uint32_t sg = seg[item];
  if (sg != 0xFFFFFFFFu) wave_copy_bytes(dst + pl, shared + shared_off[sg], shared_off[sg + 1] - shared_off[sg], lane);
}

__global__ void k_err_from_verdict(const uint8_t* __restrict__ v, uint32_t n, uint8_t* __restrict__ e) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) e[i] = (v[i] & V_IS_SUFFICIENT) ? 0 : 2;   // BFTKV_ERR_NONE : BFTKV_ERR_INSUFFICIENT_SIGNATURES
}

// Last kernel of a staged small call (the micro-batcher): per item the error byte (from the verdict bits, or as it stands) and
// the fenced / rehash flags, written straight into MAPPED HOST memory, then the call's sequence number into the mailbox
// word the host spins on -- no device-to-host copy, no interrupt-driven stream synchronisation at the end of a call that
// lasts ~0.1 ms.  One block: the flag is stored after every thread's result stores (block barrier, system-scope fence).
__global__ void __launch_bounds__(256) k_finish_staged(const uint8_t* __restrict__ verdict_or_err, uint32_t from_verdict,
                                                       const uint8_t* __restrict__ item_flags, const uint32_t* __restrict__ item_hash_mask,
                                                       uint32_t n, uint32_t rehash_bits, uint8_t* __restrict__ host_out /*[2][n] + 8*/,
                                                       uint32_t* __restrict__ host_flag, uint32_t seq,
                                                       const uint32_t* __restrict__ total /*[0] packets, [1] overflow (k_scan_counts cap)*/) {
  if (threadIdx.x < 8) host_out[2 * (size_t)n + threadIdx.x] = (uint8_t)(total[threadIdx.x >> 2] >> (8 * (threadIdx.x & 3)));
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint8_t v = verdict_or_err[i];
    host_out[i] = from_verdict ? ((v & V_IS_SUFFICIENT) ? 0 : 2) : v;
    host_out[n + i] = (((item_flags[i] & 4) || (item_hash_mask[i] & ITEM_FENCED)) ? 1 : 0) | ((item_hash_mask[i] & rehash_bits) ? 2 : 0);
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// per-record status export (diagnostics / parity tests)
__global__ void k_export_status(const SigRec* __restrict__ recs, uint32_t n, uint8_t* __restrict__ st, uint32_t* __restrict__ item) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t v = recs[i].status;
  st[i] = v >= ST_PENDING_PARSE ? (uint8_t)ST_NOT_EXAMINED : v;     // left pending: behind the reference's early exit
  if (item) item[i] = recs[i].item;
}

}  // namespace bftkv
