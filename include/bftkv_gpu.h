/*
 * bftkv_gpu.h -- C ABI of the MI355X batched quorum verifier for yahoo/bftkv.
 *
 * The reference has no FFI today; its plug-in seam for this path is the Go interface bundle
 * crypto.Crypto (crypto/crypto.go:103-111), specifically crypto.Signature (crypto/crypto.go:50-58),
 * crypto.CollectiveSignature (crypto/crypto.go:66-71) and quorum.Quorum (quorum/quorum.go:18-25),
 * implemented by crypto/pgp/crypto_pgp.go:319-344, 373-390, 485-519 and
 * quorum/wotqs/wotqs.go:144-193.  Every entry point below names the reference function(s) it
 * replaces; INTEGRATION.md shows the cgo package (crypto/pgpgpu) a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative BFTKV_E_* infrastructure error;
 *     cryptographic outcomes are DATA (status / verdict bytes), never return codes;
 *   - all pointers are host pointers unless the function name ends in _dev; buffers are only read
 *     for the duration of the call (cgo rule: no pointer is retained);
 *   - the library is re-entrant per context; calls on one context are serialised internally.
 *   - offsets arrays have n+1 entries: item i occupies blob[off[i] .. off[i+1]).
 */
#ifndef BFTKV_GPU_H
#define BFTKV_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bftkv_gpu_ctx bftkv_gpu_ctx;

/* infrastructure errors */
#define BFTKV_OK 0
#define BFTKV_E_INVALID (-1)       /* bad argument */
#define BFTKV_E_DEVICE (-2)        /* HIP runtime failure (message via bftkv_gpu_last_error) */
#define BFTKV_E_NOMEM (-3)
#define BFTKV_E_UNSUPPORTED (-4)   /* e.g. modulus > 2048 bits in the threshold entry points, more than 8 cliques in one quorum */
#define BFTKV_E_STATE (-5)         /* keyring / quorum not set */

/* per-packet status (one CheckDetachedSignature-equivalent step) */
#define BFTKV_ST_OK 0
#define BFTKV_ST_UNKNOWN_ISSUER 1
#define BFTKV_ST_PARSE_ERROR 2
#define BFTKV_ST_NOT_SIGNATURE 3
#define BFTKV_ST_NO_ISSUER 4
#define BFTKV_ST_HASH_UNSUPPORTED 5
#define BFTKV_ST_HASH_TAG 6
#define BFTKV_ST_ALGO_MISMATCH 7
#define BFTKV_ST_BAD_SIG 8
#define BFTKV_ST_KEY_CANNOT_SIGN 9
#define BFTKV_ST_UNSUPPORTED 10
#define BFTKV_ST_NOT_EXAMINED 11     /* behind the early exit of CollectiveSignature.Verify: the reference never reads it */

/* error identities of crypto/crypto.go:16-33 the shim maps verdicts to */
#define BFTKV_ERR_NONE 0
#define BFTKV_ERR_INVALID_SIGNATURE 1          /* crypto.ErrInvalidSignature */
#define BFTKV_ERR_INSUFFICIENT_SIGNATURES 2    /* crypto.ErrInsufficientNumberOfSignatures */
#define BFTKV_ERR_CERTIFICATE_NOT_FOUND 3      /* crypto.ErrCertificateNotFound (bftkv_gpu_batcher_cert_verify: Issuer(sig) == nil) */

/* quorum predicate bits (quorum/wotqs/wotqs.go:144-185) */
#define BFTKV_V_IS_QUORUM 1
#define BFTKV_V_IS_THRESHOLD 2
#define BFTKV_V_IS_SUFFICIENT 4
#define BFTKV_V_REJECT 8

/* ---- lifecycle ------------------------------------------------------------------------------ */
int bftkv_gpu_init(int device_ordinal, bftkv_gpu_ctx** out);
void bftkv_gpu_destroy(bftkv_gpu_ctx* ctx);
const char* bftkv_gpu_last_error(const bftkv_gpu_ctx* ctx);
/* "crypto: invalid signature" etc. -- the strings that travel in the X-error header
 * (transport/http/http.go:145, bftkv.go:34-48) */
const char* bftkv_gpu_error_string(int bftkv_err);

/* ---- keyring: replaces PGPKeyring.getKeyring()/KeysByIdUsage (crypto_pgp.go:195-219) ---------- */
typedef struct {
  uint64_t key_id;        /* 64-bit OpenPGP key id of this (sub)key */
  uint64_t entity_id;     /* primary key id of the owning entity = node.Node.Id() (crypto_pgp.go:43-45) */
  uint8_t pk_algo;        /* 1 RSA, 3 RSA-sign-only, 17 DSA, ... as in the public-key packet */
  uint8_t usable_sign;    /* KeysByIdUsage(id, KeyFlagSign) would return it: entity not revoked, self-signature
                             not a revocation, key flags absent or containing Sign */
  uint8_t reserved[6];
  const uint8_t* n; uint32_t n_len;   /* RSA modulus, big-endian       | DSA p */
  const uint8_t* e; uint32_t e_len;   /* RSA public exponent           | DSA q */
  const uint8_t* g; uint32_t g_len;   /* DSA g */
  const uint8_t* y; uint32_t y_len;   /* DSA y */
} bftkv_gpu_pubkey;

/* keys[] in keyring order (secring entities first, crypto_pgp.go:195-197).  Identical material
 * under one key id (the node's own key appears in both rings) is de-duplicated.  DIFFERENT keys
 * under one 64-bit id all stay in the table (ids can be made to collide with ~2^32 work) and are
 * asked in keyring order as openpgp.CheckDetachedSignature asks KeysByIdUsage's candidates: the first
 * that can sign sees the true digest, every later one the hash suffix once more (x/crypto writes it
 * into the shared hash per candidate), so the first success or the LAST candidate's error stands. */
int bftkv_gpu_keyring_set(bftkv_gpu_ctx* ctx, const bftkv_gpu_pubkey* keys, uint32_t n_keys);

/* MD5 (hash id 1) and RIPEMD-160 (3): openpgp's hashForSignature answers "hash not available" unless the BINARY links the
 * package (crypto/md5, golang.org/x/crypto/ripemd160) -- a property of the deployment, not of the input.  state 0 (default):
 * unknown, signatures naming the hash are fenced (the reference path decides); 1: available, they are hashed and verified
 * natively like any other; 2: not available, they fail with the reference's unsupported-hash outcome and are NOT fenced.
 * The Go shim sets it from crypto.MD5.Available() / crypto.RIPEMD160.Available() at start-up.  Transport
 * messages that name these hashes stay BFTKV_MSG_UNSUPPORTED. */
int bftkv_gpu_set_hash_policy(bftkv_gpu_ctx* ctx, int hash_id, int state);

/* DSA verification (Go crypto/dsa.Verify under packet.PublicKey.VerifySignature) multiplies from per-key
 * fixed-base window tables kept in HBM: 18-bit windows cost 2.39 GB per key and 29 multiplications per signature, 16-bit
 * windows 637 MB and 31, 8-bit windows 4.96 MB and <= 63, 4-bit windows 0.58 MB and <= 127.  The default is the widest
 * whose tables fit the free HBM with room to spare (18 bits: 45 % of it, e.g. 32 keys = 79 GB of the part's 288 GB;
 * 16 bits: a quarter), both only up to 64 DSA keys; 8 up to 4096 keys, 4 beyond; bits = 4, 8, 16 or 18 pins the width,
 * 0 returns to the default.  A caller that owns the device's memory may also pin 17, 19 (4.46 GB per key, 27
 * multiplications) or 20 (8.3 GB, 25): the policy never chooses them, and a table that does not fit fails the upload
 * with BFTKV_E_DEVICE.  Takes effect at the next bftkv_gpu_keyring_set. */
int bftkv_gpu_set_dsa_window_bits(bftkv_gpu_ctx* ctx, uint32_t bits);
/* The width the DSA tables of the current key table were built at (0: the table holds no DSA key): a verification is
 * 2 * ceil(256 / bits) - 1 table multiplications at most. */
int bftkv_gpu_dsa_window_bits(bftkv_gpu_ctx* ctx, uint32_t* bits_out);
/* A bound on the HBM the DSA tables may hold -- for a service that shares the GPU (the default policy takes up to 45 % of the
 * FREE memory: 76.5 GB for 32 keys on an idle MI355X).  With a budget the policy picks the widest window whose tables for every
 * DSA key of the node keyring fit `bytes` (32 keys: 8 GB -> 14 bits, 37 multiplications per signature; 24 GB -> 16 bits, 31;
 * 77 GB -> 18 bits, 29; keys with p beyond 2048 bits: entries are 448 instead of 304 bytes), the arena is never allocated beyond
 * it, cached tables of keys that left the keyring are dropped before it is exceeded, and certificate-only DSA keys (request
 * certificates, crypto/pgp/crypto_pgp.go:332-344) get only the slots the budget leaves (none: their signatures are fenced).
 * bytes = 0 returns to the free-memory policy.  A width pinned with bftkv_gpu_set_dsa_window_bits wins over the budget.
 * Replaces the environment variable BFTKV_DSA_WBITS (kept for experiments).  Takes effect at the next bftkv_gpu_keyring_set. */
int bftkv_gpu_set_dsa_table_budget(bftkv_gpu_ctx* ctx, uint64_t bytes);
/* Bytes of HBM the DSA tables hold right now (the arena's allocation) and the entry size in limbs (76: every key has p <= 2048
 * bits; 112: a key with p up to 3072 bits is in the node keyring).  Either pointer may be null. */
int bftkv_gpu_dsa_table_bytes(bftkv_gpu_ctx* ctx, uint64_t* bytes_out, uint32_t* entry_limbs_out);

/* ---- transport message signatures: the signature half of PGPMessage.Decrypt (crypto/pgp/crypto_pgp.go:453-471) ----
 * Every request and reply is an OpenPGP message encrypted to the peer and signed by the sender
 * (crypto_pgp.go:419-451; server.go:563; transport/transport.go:116-125).  Opening the container (session-key
 * decryption, AES-CFB, MDC) is private-key / symmetric work and stays with the caller; this call takes the plaintext
 * packet sequence found inside -- [one-pass signature] [literal data] [signature], MDC trailer removed -- for a batch of
 * messages and does what openpgp.ReadMessage does from there: md.SignedBy = KeysByIdUsage(one-pass key id, sign)[0],
 * hash the literal body with the ONE-PASS packet's algorithm, VerifySignature with the trailing signature packet.
 *   status_out[i]       BFTKV_MSG_* below
 *   signer_key_id_out   m.SignedByKeyId (0 when the message was unsigned or unreadable)
 *   peer_id_out         GetCertById(SignedByKeyId): the id when a keyring entity has it as PRIMARY key id, else 0
 *   plain_out/_off_out  literal bodies, concatenated (partial body lengths removed); plain_out may be NULL
 *   fname_out           [n][256] literal file names = base64(nonce) (crypto_pgp.go:464), fname_len_out their lengths
 * Fenced shapes are reported as BFTKV_MSG_UNSUPPORTED, never guessed. */
#define BFTKV_MSG_OK 0               /* verified: Decrypt returns (plain, nonce, peer, nil) */
#define BFTKV_MSG_SIGNATURE_ERROR 1  /* m.SignatureError != nil: Decrypt returns that error */
#define BFTKV_MSG_READ_ERROR 2       /* openpgp.ReadMessage failed: crypto.ErrDecryptionFailed */
#define BFTKV_MSG_NOT_SIGNED 3       /* crypto.ErrInvalidTransportSecurityData */
#define BFTKV_MSG_UNVERIFIED 4       /* signed, but no usable signing key with that id: Decrypt returns a NIL error */
#define BFTKV_MSG_UNSUPPORTED 5      /* compressed packet, text-mode / v3 / partial-length signature, MD5 / RIPEMD-160 */
int bftkv_gpu_message_verify(bftkv_gpu_ctx* ctx, uint32_t n_msgs, const uint8_t* msgs, const uint64_t* msg_off, uint8_t* status_out,
                             uint64_t* signer_key_id_out, uint64_t* peer_id_out, uint8_t* plain_out, uint64_t plain_cap,
                             uint64_t* plain_off_out, uint8_t* fname_out, uint8_t* fname_len_out);

/* ---- quorum: replaces wotq / qc (quorum/wotqs/wotqs.go:16-26) ------------------------------- */
typedef struct {
  int32_t f, min, threshold, suff;       /* as computed by wot.newQC (wotqs.go:36-70) */
  const uint64_t* node_ids; uint32_t n_nodes;
} bftkv_gpu_qc;
int bftkv_gpu_quorum_create(bftkv_gpu_ctx* ctx, const bftkv_gpu_qc* qcs, uint32_t n_qcs, int* quorum_out);
int bftkv_gpu_quorum_destroy(bftkv_gpu_ctx* ctx, int quorum);

/* ---- fenced inputs ------------------------------------------------------------------------------------
 * A few OpenPGP shapes that the reference accepts are not followed by the kernels (DESIGN.md "Fenced inputs":
 * a packet after which the reference's shared reader stands INSIDE that packet -- a signature that parses while bufio's
 * last fetch stopped short of its end (bodies over 4096 bytes with unread bytes, unread partial-length chunks), a known
 * non-signature packet whose parser may stop early --, MD5 / RIPEMD-160 while their availability in the reference binary is
 * unknown (bftkv_gpu_set_hash_policy), ECDSA, moduli beyond 4096 bits, signature values >= R, embedded signatures nested
 * deeper than 2).  Partial and indeterminate body lengths as such are read like the reference reads them.
 * The verify calls take an optional fenced_out[n_items]: fenced_out[i] = 1 when item i contains such a
 * shape -- its err_out is then NOT a statement about what the reference would decide, and the caller must run the
 * reference path for that item (the cgo shim calls the wrapped crypto/pgp implementation).  Items with
 * fenced_out[i] = 0 carry the reference's verdict.  None of the path's own writers produce a fenced shape. */

/* ---- CollectiveSignature.Verify, batched (crypto_pgp.go:485-500) ----------------------------- */
/* For item i: tbs = tbs_blob[tbs_off[i]..], ss.Data = ss_blob[ss_off[i]..], quorum q.
 *   err_out[i]        BFTKV_ERR_NONE (=> the shim sets ss.Completed = true, crypto_pgp.go:494) or
 *                     BFTKV_ERR_INSUFFICIENT_SIGNATURES
 *   n_verified_out[i] (optional) len(verified) when the reference returned
 *   verdict_out[i]    (optional) BFTKV_V_* bits of the full verified-signer list */
int bftkv_gpu_collective_verify(bftkv_gpu_ctx* ctx, int quorum, uint32_t n_items,
                                const uint8_t* tbs_blob, const uint64_t* tbs_off,
                                const uint8_t* ss_blob, const uint64_t* ss_off,
                                uint8_t* err_out, uint32_t* n_verified_out, uint8_t* verdict_out, uint8_t* fenced_out);
/* The same call for payloads that share their tails.  What a replica verifies is TBSS = Serialize(x, v, t, sig)
 * (packet/packet.go:170-190), and writeSignature ends it in chunk(Cert) (packet.go:192-212): the signer's whole certificate -- 8 KB
 * at 64 replicas, 28 KB at 256 -- the SAME bytes behind every write of one client.  A bulk caller holding host buffers (revoke /
 * audit sweeps, protocol/client.go:304-353; Server.write batches, server.go:286-300) hands them over once:
 *   payload i = prefix_blob[prefix_off[i] .. prefix_off[i+1])  ||  shared_blob[shared_off[g] .. shared_off[g+1]),  g = seg_of_item[i]
 * (g = 0xFFFFFFFF: no tail).  Only the prefixes, the n_shared distinct tails and the signature streams cross PCIe (cfg 2: 146 MB
 * instead of 226 MB per 10,000 writes); a kernel lays the payloads out in HBM as bftkv_gpu_collective_verify would have received
 * them, so every result byte -- error, exit count, verdict bits, fence flag, per-packet status -- is that call's on the
 * concatenated payloads (tests/test_gpu_host_pipeline.py).  Pipelined across PCIe like it; bftkv_gpu_set_host_pipeline applies. */
int bftkv_gpu_collective_verify_segments(bftkv_gpu_ctx* ctx, int quorum, uint32_t n_items,
                                         const uint8_t* prefix_blob, const uint64_t* prefix_off,
                                         const uint8_t* shared_blob, const uint64_t* shared_off, uint32_t n_shared,
                                         const uint32_t* seg_of_item,
                                         const uint8_t* ss_blob, const uint64_t* ss_off,
                                         uint8_t* err_out, uint32_t* n_verified_out, uint8_t* verdict_out, uint8_t* fenced_out);
/* same as bftkv_gpu_collective_verify, every pointer a DEVICE pointer (inputs already resident in HBM); asynchronous on the
 * context's stream until bftkv_gpu_sync */
int bftkv_gpu_collective_verify_dev(bftkv_gpu_ctx* ctx, int quorum, uint32_t n_items,
                                    const uint8_t* tbs_blob, const uint64_t* tbs_off,
                                    const uint8_t* ss_blob, const uint64_t* ss_off, uint64_t ss_blob_len,
                                    uint8_t* err_out, uint32_t* n_verified_out, uint8_t* verdict_out, uint8_t* fenced_out);
int bftkv_gpu_sync(bftkv_gpu_ctx* ctx);
/* Host-buffer calls of bftkv_gpu_collective_verify (the shape of crypto_pgp.go:485-500 seen from cgo: the caller owns the
 * slices).  A big batch is cut into pieces of about equal bytes at item boundaries; piece k is verified by a private worker
 * context while the pieces behind it are still crossing PCIe, and the results reach the caller's arrays after one
 * synchronisation.  Items are independent: results are those of the unsplit call.  pieces = 0 (default): by call size (one
 * piece below 24 MB), 1: never split, 2..8: that many pieces whatever the size (tests).  BFTKV_HB_PIECES in the environment
 * sets the default of contexts this call has not touched.  The caller's memory is pageable: by default its pointers go to
 * hipMemcpyAsync, piece by piece, from a helper thread (the runtime pins the pages in place and remembers them: a long-running
 * caller's heap is pinned once); BFTKV_HOST_PIPELINE_RING (or BFTKV_HB_COPY=ring) moves it through the library's own ring of
 * page-locked slots instead (helper threads copy, the DMA engine follows), which does not depend on that memory. */
int bftkv_gpu_set_host_pipeline(bftkv_gpu_ctx* ctx, uint32_t pieces);
#define BFTKV_HOST_PIPELINE_RING 0x100u    /* or-ed into `pieces`: copies staged through the library's page-locked ring (helper threads memcpy, DMA follows) */
#define BFTKV_HOST_PIPELINE_DIRECT 0x200u  /* ... or issued straight from the caller's memory (the default; the runtime pins the pages and remembers them) */
#define BFTKV_HOST_PIPELINE_TIGHT_BOUND 0x400u  /* tests: size every piece's arena by a bound real streams exceed, so that the second pass runs */
/* Diagnostics: host-side timeline of the last pipelined call in microseconds from its start: [0] pieces, [1] 1 = ring,
 * [2] copier threads joined, [3] copy stream drained, [4] results in the caller's arrays, [5] items of the largest piece, [6] pieces that outgrew their bound and ran a second pass,
 * then per piece k at [8 + 10k]: signature streams enqueued, payloads enqueued, piece picked up by the enqueuing thread, its
 * payload hook reached, piece fully enqueued, piece drained (host clock); then on the device, from the call's first copy:
 * piece started, its modexp's turn, modexp done, piece done. */
int bftkv_gpu_host_pipeline_trace(bftkv_gpu_ctx* ctx, float* out, uint32_t cap, uint32_t* n_out);
/* PGPCollectiveSignature.Verify returns at the first packet after which q.IsSufficient(verified) holds and never reads the
 * rest of ss.Data (crypto_pgp.go:491-496).  By default the batched call does the same amount of public-key work: per item
 * it verifies the packets up to the position where the quorum would be sufficient if they all verified (plus a small
 * margin), tallies, and verifies the remaining packets only of the items that are still insufficient.  err_out,
 * n_verified_out and verdict_out are those of the reference either way (verdict_out = predicate bits of the signers
 * verified until the exit); packets behind the exit report BFTKV_ST_NOT_EXAMINED.  on = 0 verifies every packet of
 * every item (diagnostics: verdict_out then covers the full signer list). */
int bftkv_gpu_set_early_exit(bftkv_gpu_ctx* ctx, int on);

/* ---- Signature.Verify / VerifyWithCertificate, batched (crypto_pgp.go:319-344) ---------------- */
/* err_out[i] = BFTKV_ERR_NONE iff sig.Data holds >= 1 packet and every CheckDetachedSignature call
 * succeeds, else BFTKV_ERR_INVALID_SIGNATURE.  cert_key_id == NULL: keyring = the node keyring;
 * otherwise the keyring of item i is the single entity whose primary key id is cert_key_id[i]
 * (subkeys of that entity match too).  The entity must be in the device table, i.e. have been uploaded with
 * bftkv_gpu_keyring_set: the reference verifies against the certificate it is HANDED, the device only against what it
 * holds, so an id that is not in the table comes back with fenced_out[i] = 1 (err_out[i] invalid-signature, fail-closed)
 * and the caller takes the reference path; bftkv_host_server_sign_verify registers request certificates instead. */
int bftkv_gpu_signature_verify(bftkv_gpu_ctx* ctx, uint32_t n_items,
                               const uint8_t* tbs_blob, const uint64_t* tbs_off,
                               const uint8_t* sig_blob, const uint64_t* sig_off,
                               const uint64_t* cert_key_id, uint8_t* err_out, uint8_t* fenced_out);

/* ---- forked verifier contexts ------------------------------------------------------------------------ */
/* A fork has its own streams, per-call arena and events -- several device calls in flight at once -- and reads the ROOT's
 * resident key table, DSA tables and quorum handles (no copy: 64 DSA keys hold 41 GB of window tables).  It accepts the
 * verify calls (collective / signature / message / signers); whatever changes the key table or the quorums
 * (bftkv_gpu_keyring_set, bftkv_gpu_quorum_create / _destroy, the certificate sites of bftkv_host.h,
 * bftkv_gpu_set_early_exit) is done on the root, waits there until the forks' calls in flight have drained, and is seen
 * by every fork at its next call.  Destroy the forks before their root.
 *
 * Resident batches in flight on several contexts of one device (forks or not): the machine-filling RSA exponentiation of such
 * a call (calls of >= 98,304 signature packets; not the staged small calls) takes turns with those of the other contexts -- it
 * waits, on its stream, for the one launched before it -- so that one call's walk / parse / compare / tally run under another's
 * exponentiation instead of two exponentiations sharing the machine.  Three batches in flight keep the device at 1.04 x the
 * kernel's duration per batch (bench.py --config 2).  BFTKV_NO_TURNSTILE=1 in the environment turns it off. */
int bftkv_gpu_ctx_fork(bftkv_gpu_ctx* root, bftkv_gpu_ctx** fork_out);

/* ---- small batches: the latency route ---------------------------------------------------------------- */
/* Same verdicts as bftkv_gpu_collective_verify / bftkv_gpu_signature_verify (err_out, fenced_out), for batches of a few
 * hundred items at most, where a call is bound by its chain of launches and by the hash chain of a payload (134 dependent
 * SHA-256 compressions for an 8.6 KB write: 0.43 ms on one GPU lane, 7 us on a host core with the SHA extensions): the
 * calling thread absorbs the whole blocks of every payload (SHA-256 midstates), everything crosses PCIe in one pinned
 * buffer, the kernels run on one stream with no host round trip in between, every packet is verified (no two-phase
 * planning), <= 2048-bit RSA numbers are spread over eight lanes, and the results come back through mapped host memory.
 * Items whose signatures ask for another hash than SHA-256 are run again through the ordinary entry point.  This is what a
 * batcher's leader does for its batch; 0.18 ms for one 53-signature write where the ordinary call takes 0.59 ms. */
int bftkv_gpu_collective_verify_small(bftkv_gpu_ctx* ctx, int quorum, uint32_t n_items, const uint8_t* tbs_blob, const uint64_t* tbs_off,
                                      const uint8_t* ss_blob, const uint64_t* ss_off, uint8_t* err_out, uint8_t* fenced_out);
int bftkv_gpu_signature_verify_small(bftkv_gpu_ctx* ctx, uint32_t n_items, const uint8_t* tbs_blob, const uint64_t* tbs_off,
                                     const uint8_t* sig_blob, const uint64_t* sig_off, const uint64_t* cert_key_id, uint8_t* err_out,
                                     uint8_t* fenced_out);

/* ---- micro-batching of concurrent single calls ---------------------------------------------------- */
/* The reference verifies ONE message per call, concurrently from one goroutine per HTTP request
 * (transport/http/http.go:85,143 -> protocol/server.go:562-620).  A batcher turns such calls into device batches:
 * each call blocks until its batch has been verified.  There is no worker thread: the first caller into an empty batch
 * leads it -- it takes a free LANE (a forked context; n_lanes of them, so that many batches are on the device at once),
 * runs the device call on its own thread and wakes the callers that joined meanwhile; a lone caller crosses no thread at
 * all.  While every lane is busy the callers pile up behind the waiting leader, at most max_items to a batch (the next
 * caller then leads a new one).  Callers hash the whole blocks of their own payload on their own thread (SHA-256
 * midstate, SHA extensions when the CPU has them) and the device finishes each signature's digest from there.
 * max_wait_us is accepted for compatibility and unused: nobody waits for company.  n_lanes = 0 (and
 * bftkv_gpu_batcher_create): $BFTKV_BATCHER_LANES or 3.  Thread-safe; buffers are only read for the duration of the call.
 * FAIL-CLOSED: the status byte is written on every path and is a failure (invalid signature / insufficient signatures /
 * read error) whenever the return code is not 0 -- a caller that only looks at the status can never read "verified" out
 * of an infrastructure error (allocation failure, stopped batcher, bad handle). */
typedef struct bftkv_gpu_batcher bftkv_gpu_batcher;
bftkv_gpu_batcher* bftkv_gpu_batcher_create(bftkv_gpu_ctx* ctx, uint32_t max_items, uint32_t max_wait_us);
bftkv_gpu_batcher* bftkv_gpu_batcher_create_lanes(bftkv_gpu_ctx* ctx, uint32_t max_items, uint32_t max_wait_us, uint32_t n_lanes);
void bftkv_gpu_batcher_destroy(bftkv_gpu_batcher* b);
/* CollectiveSignature.Verify(tbs, ss, q) (crypto_pgp.go:485-500): *err_out = BFTKV_ERR_NONE / _INSUFFICIENT_SIGNATURES */
int bftkv_gpu_batcher_collective_verify(bftkv_gpu_batcher* b, int quorum, const uint8_t* tbs, uint64_t tbs_len,
                                        const uint8_t* ss, uint64_t ss_len, uint8_t* err_out, uint8_t* fenced_out);
/* Signature.Verify / VerifyWithCertificate (crypto_pgp.go:319-344); cert_key_id NULL = node keyring */
int bftkv_gpu_batcher_signature_verify(bftkv_gpu_batcher* b, const uint8_t* tbs, uint64_t tbs_len, const uint8_t* sig,
                                       uint64_t sig_len, const uint64_t* cert_key_id, uint8_t* err_out, uint8_t* fenced_out);
/* Signature.Issuer(sig) + Signature.VerifyWithCertificate(tbs, sig, issuer) for a principal that is NOT in the node keyring --
 * the shape of protocol/server.go:199-207 (sign) and :460-468 (register), where the certificate travels inside the request
 * (sig.Cert) -- as ONE micro-batched call.  Replaces crypto_pgp.go:392-405 -> :236-249 (Certificate.Parse ->
 * openpgp.ReadEntity, which verifies every user-id self-signature, every subkey binding / revocation with the cross-signature
 * of a signing subkey, and every key revocation of the entity: bftkv_host.h bftkv_host_certs_verify) and :332-344:
 *   cert / cert_len   sig.Cert: serialised entities; the issuer is the FIRST one ("has to be the first one", :404);
 *   tbs, sig          packet.TBS(req) and sig.Data; sig == NULL asks for the issuer alone (Issuer(), server.go:330-331);
 *   *err_out          BFTKV_ERR_NONE, BFTKV_ERR_INVALID_SIGNATURE (VerifyWithCertificate's error), or
 *                     BFTKV_ERR_CERTIFICATE_NOT_FOUND (no entity parses, or ReadEntity would refuse the first one: the
 *                     reference's Issuer() returns nil and the server answers crypto.ErrCertificateNotFound);
 *   *fenced_out       1 = no verdict (a fenced shape in the certificate or the signature): take the reference path;
 *   *issuer_id_out    primary key id of the first entity; fingerprint_out[20] its v4 fingerprint (so that a caller holding
 *                     a parsed certificate can tell that it IS this one); either may be NULL.
 * The entity is registered in the root context's key table as a certificate-only entity (bounded and recycled: at most 1024
 * distinct certificates, certificate DSA keys share a bounded set of table slots) and its ReadEntity verdict is remembered
 * by certificate bytes, so a client's second request costs one signature verification.  The FIRST request with a certificate
 * runs on the ROOT context (registration changes the key table; the lanes' calls in flight drain first); once ReadEntity's
 * verdict on those bytes is "valid", later requests carrying the same bytes are one staged signature verification on a lane,
 * against the entity registered for them (resolved by the bytes, never by the 64-bit key id), several in flight at once. */
int bftkv_gpu_batcher_cert_verify(bftkv_gpu_batcher* b, const uint8_t* cert, uint64_t cert_len, const uint8_t* tbs, uint64_t tbs_len,
                                  const uint8_t* sig, uint64_t sig_len, uint8_t* err_out, uint8_t* fenced_out,
                                  uint64_t* issuer_id_out, uint8_t* fingerprint_out);
/* Signature.Issuer(sig) WITHOUT openpgp.ReadEntity on the CPU (crypto_pgp.go:392-405 -> 236-249): the ReadEntity verdict of the
 * first entity of sig.Cert as bftkv_gpu_batcher_cert_verify(sig = NULL) gives it -- its signature checks on the GPU, once per
 * distinct certificate -- plus what a caller needs to ASSEMBLE the entity ReadEntity would have returned from packets it
 * parses itself (x/crypto's packet.Read: no cryptography): for each packet packet.Reader.Next() yields inside the entity, in
 * order, what ReadEntity does with it:
 *   roles_out[i] = BFTKV_ROLE_* | index << 8 | chosen << 24   (index: ordinal of the identity / subkey the packet belongs to;
 *                  chosen, on a subkey signature: it is the one ReadEntity leaves in Subkey.Sig)
 *   *entity_off_out, *entity_len_out   the entity's bytes inside cert (packet types unknown to x/crypto may precede it)
 * err / fenced / issuer id / fingerprint as for bftkv_gpu_batcher_cert_verify.  Only *err_out == BFTKV_ERR_NONE with
 * *fenced_out == 0 says "ReadEntity returns this entity, built like this"; everything else: take the reference path.
 * BFTKV_E_NOMEM with *n_roles_out set when roles_cap is too small. */
#define BFTKV_ROLE_IGNORED 0u             /* read and dropped (a version-3 signature, a stray signature outside any run) */
#define BFTKV_ROLE_PRIMARY_KEY 1u
#define BFTKV_ROLE_USER_ID 2u             /* starts identity `index` */
#define BFTKV_ROLE_SELF_SIGNATURE 3u      /* identity.SelfSignature (the last one counts) and e.Identities[name] = identity */
#define BFTKV_ROLE_IDENTITY_SIGNATURE 4u  /* appended to identity.Signatures, unverified */
#define BFTKV_ROLE_SUBKEY 5u              /* starts subkey `index` */
#define BFTKV_ROLE_SUBKEY_SIGNATURE 6u    /* verified binding / revocation of subkey `index`; chosen = Subkey.Sig */
#define BFTKV_ROLE_REVOCATION 7u          /* appended to e.Revocations */
int bftkv_gpu_batcher_cert_entity(bftkv_gpu_batcher* b, const uint8_t* cert, uint64_t cert_len, uint8_t* err_out, uint8_t* fenced_out,
                                  uint64_t* issuer_id_out, uint8_t* fingerprint_out, uint64_t* entity_off_out, uint64_t* entity_len_out,
                                  uint32_t* roles_out, uint32_t roles_cap, uint32_t* n_roles_out);
/* stats[0] calls served, stats[1] device calls made, stats[2] largest batch, stats[3] lanes */
/* One transport message (bftkv_gpu_message_verify for a single caller): blocks until its batch has run.  plain_out
 * receives the literal body (BFTKV_E_NOMEM if plain_cap is too small; msg_len always suffices), fname_out[256] the
 * file name. */
int bftkv_gpu_batcher_message_verify(bftkv_gpu_batcher* b, const uint8_t* msg, uint64_t msg_len, uint8_t* status_out,
                                     uint64_t* signer_key_id_out, uint64_t* peer_id_out, uint8_t* plain_out, uint64_t plain_cap,
                                     uint64_t* plain_len_out, uint8_t* fname_out, uint8_t* fname_len_out);
/* ---- ONE threshold share-combine operation per call, micro-batched (BASELINE config 5 behind the reference's seam) ----
 * The reference combines one signature at a time: Client.DistSign (protocol/client.go:509-546) drives ONE
 * crypto.ThresholdProcess (crypto/crypto.go:98-101) whose ProcessResponse ends in exactly one of the operations below, and a
 * server answers one Server.distSign (protocol/server.go:528-541) per request.  Concurrent callers (one goroutine per
 * request) are gathered like the verify calls above: operations of one shape (kind, k, widths) share a device call on a
 * lane, whatever their moduli; each caller blocks until its batch has run.  Numbers are big-endian, fixed width.
 *   *status_out   BFTKV_TH_OK; BFTKV_TH_NO_INVERSE (math/big's ModInverse would return nil and the reference would
 *                 dereference it: take the reference path); BFTKV_TH_FENCED (Lagrange integers beyond 2128 bits: the reference
 *                 path decides); BFTKV_TH_FAILED whenever the return code is not 0 -- the byte starts out as a failure and
 *                 `out` as zeroes (fail closed: no caller reads a result out of an infrastructure error).
 * An even modulus or one wider than 2048 bits returns BFTKV_E_UNSUPPORTED for that caller alone. */
#define BFTKV_TH_OK 0
#define BFTKV_TH_NO_INVERSE 1
#define BFTKV_TH_FENCED 2
#define BFTKV_TH_FAILED 0xFF
/* s = prod_j factors[j] mod N: the fold of calculateSignature over the completed tree's partial signatures
 * (crypto/threshold/rsa/rsa.go:235-253, 318-329).  factors: [k][nbytes]; any value below 2^(8 nbytes) is reduced like
 * big.Int.Mod does. */
int bftkv_gpu_batcher_modmul_product(bftkv_gpu_batcher* b, uint32_t k, const uint8_t* factors, uint32_t nbytes, const uint8_t* mod,
                                     uint8_t* out, uint8_t* status_out);
/* S = sum_j Lagrange(x_j; xs) * y_j mod m: SSSProcess.calculateSecret (crypto/sss/sss.go:81-92) and calculateS
 * (crypto/threshold/dsa/dsa_core.go:389-403).  xs: [k] int32, ys: [k][nbytes]. */
int bftkv_gpu_batcher_lagrange_combine(bftkv_gpu_batcher* b, uint32_t k, const int32_t* xs, const uint8_t* ys, uint32_t nbytes,
                                       const uint8_t* mod, uint8_t* out, uint8_t* status_out);
/* dsaGroupOperations.CalculateR (crypto/threshold/dsa/dsa.go:33-52): ri [k][pbytes], vi [k][qbytes], r_out [qbytes]. */
int bftkv_gpu_batcher_dsa_calculate_r(bftkv_gpu_batcher* b, uint32_t k, const int32_t* xs, const uint8_t* ri, uint32_t pbytes,
                                      const uint8_t* vi, uint32_t qbytes, const uint8_t* p, const uint8_t* q, uint8_t* r_out,
                                      uint8_t* status_out);
/* out = base ^ exp mod m: dsaGroupOperations.CalculatePartialR (crypto/threshold/dsa/dsa.go:27-31) and the per-fragment
 * m^d_i mod N of rsaContext.Sign (crypto/threshold/rsa/rsa.go:161-171) -- exponents that are SECRET shares: a deployment
 * decides whether they may leave the host (the shim leaves these two on the CPU unless asked). */
int bftkv_gpu_batcher_modexp(bftkv_gpu_batcher* b, const uint8_t* base, uint32_t nbytes, const uint8_t* exp, uint32_t exp_len,
                             const uint8_t* mod, uint8_t* out, uint8_t* status_out);
int bftkv_gpu_batcher_stats(bftkv_gpu_batcher* b, uint64_t stats[4]);
/* where the callers' time went, nanoseconds summed over all calls so far: [0] hashing their payloads, [1] leaders waiting
 * for a lane, [2] leaders assembling batches, [3] leaders inside device calls, of which [4] enqueueing and [5] waiting
 * for the results; [6] = device calls whose wait fell back to a stream synchronisation (a count); [7] = certificate
 * requests (bftkv_gpu_batcher_cert_verify) answered for a certificate accepted before -- from the register, or by one staged
 * signature verification on a lane -- instead of a compound call on the root context (a count) */
int bftkv_gpu_batcher_times(bftkv_gpu_batcher* b, uint64_t ns[8]);

/* ---- diagnostics of the last verify call: one status per packet event, in stream order -------- */
int bftkv_gpu_last_statuses(bftkv_gpu_ctx* ctx, uint8_t* status_out, uint32_t* item_out, uint32_t cap, uint32_t* n_out);
/* counters of the last verify call: [0] packets parsed, [1] public-key operations performed */
int bftkv_gpu_last_counters(bftkv_gpu_ctx* ctx, uint64_t counters[4]);

/* ---- Signers (parse only) (crypto_pgp.go:373-390, 517-519) ------------------------------------ */
/* ids_out receives, per item, the entity ids of issuers present in the keyring, in packet order;
 * ids_off_out[n_items+1] delimits them.  cap = capacity of ids_out. */
int bftkv_gpu_signers(bftkv_gpu_ctx* ctx, uint32_t n_items, const uint8_t* ss_blob, const uint64_t* ss_off,
                      uint64_t* ids_out, uint64_t* ids_off_out, uint64_t cap);
/* same, with the per-item fence flag of the verify calls: fenced_out[i] = 1 when the stream holds a shape on which the
 * walk does not follow the reference's reader (partial / indeterminate lengths, a packet that leaves the reader inside
 * its body, a v4 signature without issuer -- on which the reference itself dereferences nil): take the reference path */
int bftkv_gpu_signers_fenced(bftkv_gpu_ctx* ctx, uint32_t n_items, const uint8_t* ss_blob, const uint64_t* ss_off,
                             uint64_t* ids_out, uint64_t* ids_off_out, uint64_t cap, uint8_t* fenced_out);

/* ---- quorum predicates over node lists, batched (wotqs.go:144-193) ---------------------------- */
/* verdict_out[i] = BFTKV_V_* bits for nodes = ids[list_off[i]..list_off[i+1]) (duplicates count
 * repeatedly, wotqs.go:195-206). */
int bftkv_gpu_quorum_tally(bftkv_gpu_ctx* ctx, int quorum, uint32_t n_lists,
                           const uint64_t* ids, const uint64_t* list_off, uint8_t* verdict_out);

/* ---- modular exponentiation, batched ---------------------------------------------------------- */
/* out[i] = base[i] ^ exp[mod_idx[i]] mod mod[mod_idx[i]]; numbers big-endian, nbytes each (<= 256),
 * exponents exp_len bytes each.  Replaces the per-fragment m^d_i mod N of threshold RSA
 * (crypto/threshold/rsa/rsa.go:161-171); also used to sign synthetic corpora. */
int bftkv_gpu_modexp(bftkv_gpu_ctx* ctx, uint32_t n_ops, const uint8_t* base, uint32_t nbytes,
                     const uint32_t* mod_idx, uint32_t n_mods, const uint8_t* mods,
                     const uint8_t* exps, uint32_t exp_len, uint8_t* out);

/* same with ONE EXPONENT PER OPERATION (exps: [n_ops][exp_len]): out[i] = base[i] ^ exp[i] mod mod[mod_idx[i]] --
 * the partial r_i = g^a_i mod p of threshold DSA, dsaGroupOperations.CalculatePartialR (crypto/threshold/dsa/dsa.go:27-31). */
int bftkv_gpu_modexp_ops(bftkv_gpu_ctx* ctx, uint32_t n_ops, const uint8_t* base, uint32_t nbytes,
                         const uint32_t* mod_idx, uint32_t n_mods, const uint8_t* mods,
                         const uint8_t* exps, uint32_t exp_len, uint8_t* out);

/* ---- multi-GPU: all-gather of verdict bitmaps over RCCL ----------------------------- */
/* One context per GPU/process.  uid: 128 opaque bytes from bftkv_gpu_comm_unique_id on rank 0, distributed by
 * the caller (the Go shim: over its own transport).  librccl.so.1 is dlopen'ed on first use. */
int bftkv_gpu_comm_unique_id(uint8_t uid_out[128]);
int bftkv_gpu_comm_init(bftkv_gpu_ctx* ctx, int n_ranks, int rank, const uint8_t uid[128]);
/* Which librccl the calls above go to, and whether the process already held it (ONE RCCL per process: an instance that
 * torch.distributed or the host program loaded is reused, dlopen(RTLD_NOLOAD); only otherwise the system's is loaded). */
int bftkv_gpu_comm_library(char* path_out, uint32_t cap, int* preloaded_out);
/* Before the first exchange: every rank stamps nbytes with a pattern of its rank, all-gathers on the verifier's stream and
 * checks every row on the device.  Collective: all ranks call it.  0 = every row is its rank's; otherwise
 * bftkv_gpu_last_error names the rank, the wrong rows and RCCL's own error string.  BFTKV_FORCE_RCCL=1 in the environment
 * makes a one-rank context build a real communicator, so the RCCL path can be exercised on a 1-GPU box. */
int bftkv_gpu_comm_selftest(bftkv_gpu_ctx* ctx, uint32_t nbytes, uint32_t* bad_rows_out);
/* local_bits: nbytes DEVICE bytes of this rank; all_bits_out: n_ranks*nbytes DEVICE bytes, rank-major. */
int bftkv_gpu_allgather_verdicts(bftkv_gpu_ctx* ctx, const uint8_t* local_bits, uint64_t nbytes, uint8_t* all_bits_out);
/* The exchange step of the path as ONE asynchronous call on the context's stream, to be issued right behind
 * bftkv_gpu_collective_verify_dev (no host synchronisation in between): packs err_dev[0..n_items) -- the err_out of that
 * call; bit = 1 where it is BFTKV_ERR_NONE -- into a bitmap of ceil(slots/8) bytes (slots >= n_items: the largest shard,
 * so that every rank contributes the same byte count; bit i of byte i/8 = item i, zero padded) and all-gathers the
 * bitmaps rank-major into all_bits_out[n_ranks * ceil(slots/8)] (DEVICE).  Every rank then holds the verdict of every
 * write, as every replica of the reference reaches every decision (protocol/server.go:300).  Wait with bftkv_gpu_sync. */
int bftkv_gpu_allgather_errs_dev(bftkv_gpu_ctx* ctx, const uint8_t* err_dev, uint32_t n_items, uint32_t slots, uint8_t* all_bits_out);

/* ---- threshold-signature share combine (BASELINE config 5) ------------------------------------ */
/* Numbers are big-endian, nbytes each (<= 256); moduli must be odd; mod_idx[op] selects the modulus.
 * status_out[op] (where present): 0 ok, 1 no modular inverse, 2 fenced -- a Lagrange numerator prod x_i or denominator
 * prod (x_i - x_j) beyond 2128 bits as an exact integer (256 nodes with ids below 256: about 260 terms fit; the reference's
 * big.Int has no bound).  Coefficients within 31 bits -- the reference's own n = 10 -- take a small-integer fast path, larger
 * ones (64 or 256 nodes) exact big integers and ONE modular inverse per operation. */

/* S = prod_j factors[op][j] mod N -- calculateSignature (crypto/threshold/rsa/rsa.go:318-329). */
int bftkv_gpu_modmul_product(bftkv_gpu_ctx* ctx, uint32_t n_ops, uint32_t k, const uint8_t* factors, uint32_t nbytes,
                             const uint32_t* mod_idx, uint32_t n_mods, const uint8_t* mods, uint8_t* out);
/* S = sum_j Lagrange(x_j; xs) * y_j mod m -- SSSProcess.calculateSecret (crypto/sss/sss.go:69-107) and
 * calculateS (crypto/threshold/dsa/dsa_core.go:389-403).  xs: [n_ops][k] int32, ys: [n_ops][k][nbytes]. */
int bftkv_gpu_lagrange_combine(bftkv_gpu_ctx* ctx, uint32_t n_ops, uint32_t k, const int32_t* xs, const uint8_t* ys,
                               uint32_t nbytes, const uint32_t* mod_idx, uint32_t n_mods, const uint8_t* mods,
                               uint8_t* out, uint8_t* status_out);
/* r = (prod_j Ri_j^l_j mod p)^((sum_j Vi_j*l_j)^-1 mod q) mod p mod q, l_j = Lagrange(x_j; xs) mod q --
 * dsaGroupOperations.CalculateR (crypto/threshold/dsa/dsa.go:33-52).  ri: [n_ops][k][pbytes],
 * vi: [n_ops][k][qbytes], groups: p[n_groups][pbytes], q[n_groups][qbytes] (q <= 256 bits); r_out: [n_ops][qbytes]. */
int bftkv_gpu_dsa_calculate_r(bftkv_gpu_ctx* ctx, uint32_t n_ops, uint32_t k, const int32_t* xs, const uint8_t* ri, uint32_t pbytes,
                              const uint8_t* vi, uint32_t qbytes, const uint32_t* group_idx, uint32_t n_groups,
                              const uint8_t* p, const uint8_t* q, uint8_t* r_out, uint8_t* status_out);

/* shares[poly][x-1] = sum_j coeffs[poly][j] * x^j mod m for x = 1..n_shares -- sss.Distribute (crypto/sss/sss.go:23-47)
 * with the random coefficients supplied by the caller (coeffs[poly][0] is the secret). */
int bftkv_gpu_sss_distribute(bftkv_gpu_ctx* ctx, uint32_t n_polys, uint32_t n_shares, uint32_t k, const uint8_t* coeffs, uint32_t nbytes,
                             const uint32_t* mod_idx, uint32_t n_mods, const uint8_t* mods, uint8_t* shares_out);
/* out[op] = values[op]^-1 mod m (any odd m, exact binary extended GCD; status 1: no inverse) -- the ModInverse of
 * rsaContext.Sign for negative key fragments (crypto/threshold/rsa/rsa.go:164-167). */
int bftkv_gpu_modinv(bftkv_gpu_ctx* ctx, uint32_t n_ops, const uint8_t* values, uint32_t nbytes, const uint32_t* mod_idx, uint32_t n_mods,
                     const uint8_t* mods, uint8_t* out, uint8_t* status_out);

/* Diagnostic: out[op] = values[op] - m when values[op] >= m, else values[op], for values < 2m (same width as m), through the
 * conditional subtraction the kernels use where a residue must be exact (the DSA tail; lanes = 4 | 8 lanes per number).  On the
 * verify path that subtraction is taken by about one value in 2^50, so the tests drive it here with chosen inputs. */
int bftkv_gpu_selftest_reduce(bftkv_gpu_ctx* ctx, uint32_t n_ops, const uint8_t* values, uint32_t nbytes, const uint32_t* mod_idx,
                              uint32_t n_mods, const uint8_t* mods, uint32_t lanes, uint8_t* out);

/* The *_dev forms below cannot look at their x's (they are in HBM), so they always enqueue the big-integer Lagrange kernels
 * behind the fast path (their grids return at once for operations that did not need them: a few launches per call).  A caller
 * that KNOWS its share indices -- sss.Distribute hands out x = 1..n (crypto/sss/sss.go:36-44) -- promises 0 <= x <= bound here
 * and the kernels are enqueued only when bound^(k-1) leaves the fast path's 31 bits.  0 = no promise (the default).  A promise
 * that does not hold costs nothing but speed: an operation outside the fast path then comes back fenced (status 2). */
int bftkv_gpu_set_lagrange_x_bound(bftkv_gpu_ctx* ctx, uint32_t bound);

/* The same five with the PER-OPERATION arrays (factors / xs / ys / ri / vi / coeffs / values / mod_idx / outputs / status)
 * already resident in HBM and the results left there: asynchronous on the context's stream until bftkv_gpu_sync.
 * mods / p / q stay HOST pointers (a few hundred bytes per distinct modulus, cached per context by value); mod_idx /
 * group_idx may be NULL (every operation uses modulus 0), out-of-range indices are clamped. */
int bftkv_gpu_modmul_product_dev(bftkv_gpu_ctx* ctx, uint32_t n_ops, uint32_t k, const uint8_t* factors, uint32_t nbytes,
                                 const uint32_t* mod_idx, uint32_t n_mods, const uint8_t* mods, uint8_t* out);
int bftkv_gpu_lagrange_combine_dev(bftkv_gpu_ctx* ctx, uint32_t n_ops, uint32_t k, const int32_t* xs, const uint8_t* ys,
                                   uint32_t nbytes, const uint32_t* mod_idx, uint32_t n_mods, const uint8_t* mods,
                                   uint8_t* out, uint8_t* status_out);
int bftkv_gpu_dsa_calculate_r_dev(bftkv_gpu_ctx* ctx, uint32_t n_ops, uint32_t k, const int32_t* xs, const uint8_t* ri, uint32_t pbytes,
                                  const uint8_t* vi, uint32_t qbytes, const uint32_t* group_idx, uint32_t n_groups,
                                  const uint8_t* p, const uint8_t* q, uint8_t* r_out, uint8_t* status_out);
int bftkv_gpu_sss_distribute_dev(bftkv_gpu_ctx* ctx, uint32_t n_polys, uint32_t n_shares, uint32_t k, const uint8_t* coeffs, uint32_t nbytes,
                                 const uint32_t* mod_idx, uint32_t n_mods, const uint8_t* mods, uint8_t* shares_out);
int bftkv_gpu_modinv_dev(bftkv_gpu_ctx* ctx, uint32_t n_ops, const uint8_t* values, uint32_t nbytes, const uint32_t* mod_idx, uint32_t n_mods,
                         const uint8_t* mods, uint8_t* out, uint8_t* status_out);

/* ---- timing of the last *_dev verify call (HIP events on the context's stream) ---------------- */
/* ms[0] whole call, ms[1] walk+parse, ms[2] hash stream (midstates+digests, overlaps the modexp),
 * ms[3] k_rsa_modexp, ms[4] tally, ms[5] compare (incl. joining the hash stream), ms[6] k_dsa_mul + k_dsa_modexp */
int bftkv_gpu_last_timing(bftkv_gpu_ctx* ctx, float ms[8]);
/* shader clock k_rsa_modexp ran at in the last verify call: s_memtime ticks per s_memrealtime (100 MHz) tick over the life
 * of its first wave; 0 when the call queued no RSA work.  The part clocks down under an all-MAC load, and the integer roof
 * of the path moves with it. */
int bftkv_gpu_last_sclk_mhz(bftkv_gpu_ctx* ctx, float* mhz);
void* bftkv_gpu_stream(bftkv_gpu_ctx* ctx);   /* hipStream_t of the context */

#ifdef __cplusplus
}
#endif
#endif /* BFTKV_GPU_H */
