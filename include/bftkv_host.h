/*
 * bftkv_host.h -- host-side mirror (C++ inside libbftkv_gpu.so, flat C ABI here) of the reference
 * code that sits directly on either side of the GPU path.  The reference is compiled Go and no Go
 * toolchain exists in the build image, so these are the pieces the cgo shim would otherwise keep in
 * Go; they exist so that the whole path -- request bytes in, reference error identity out -- can be
 * driven and parity-tested end to end through one library.
 *
 *   packet framing            packet/packet.go:35-115, 142-248
 *   trust graph + cliques     node/graph/graph.go:46-140, 279-393, 420-438
 *   quorum system             quorum/wotqs/wotqs.go:36-193
 *   vote collector            protocol/client.go:28-50, 125-205; protocol/server.go:286-302
 *
 * All functions return 0 on success, a negative BFTKV_E_* otherwise (bftkv_gpu.h).
 */
#ifndef BFTKV_HOST_H
#define BFTKV_HOST_H

#include <stdint.h>

#include "bftkv_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- packet (packet/packet.go) ------------------------------------------------------------- */
typedef struct {
  uint8_t type;          /* SignaturePacket.Type: 0 nil, 1 PGP (packet.go:13-19) */
  uint32_t version;
  uint8_t completed;
  const uint8_t* data; uint64_t data_len;
  const uint8_t* cert; uint64_t cert_len;
} bftkv_sigpkt;

/* Serialize(x, v, t, sig, ss, auth) with the first n_fields arguments present (1..6), packet.go:35-60.
 * A NULL sig/ss with the field present is written as the nil signature (22 zero bytes). */
int bftkv_host_packet_serialize(int n_fields, const uint8_t* x, uint64_t x_len, const uint8_t* v, uint64_t v_len, uint64_t t,
                                const bftkv_sigpkt* sig, const bftkv_sigpkt* ss, const uint8_t* auth, uint64_t auth_len,
                                uint8_t* out, uint64_t cap, uint64_t* out_len);

typedef struct {
  uint64_t x_off, x_len, v_off, v_len, t;
  int has_sig, has_ss;                 /* 0: absent or Type==0 (nil), packet.go:231-233 */
  bftkv_sigpkt sig, ss;                /* pointers into the parsed buffer */
  uint64_t auth_off, auth_len;
} bftkv_parsed;
/* Parse, packet.go:62-115: trailing fields may be absent (io.EOF => nil); a short read inside a
 * field is BFTKV_E_INVALID. */
int bftkv_host_packet_parse(const uint8_t* pkt, uint64_t len, bftkv_parsed* out);
/* TBS / TBSS return prefix LENGTHS of pkt (packet.go:156-190). */
int bftkv_host_packet_tbs(const uint8_t* pkt, uint64_t len, uint64_t* tbs_len);
int bftkv_host_packet_tbss(const uint8_t* pkt, uint64_t len, uint64_t* tbss_len);

/* ---- trust graph (node/graph/graph.go) + quorum system (quorum/wotqs/wotqs.go) ---------------- */
typedef struct bftkv_graph bftkv_graph;
typedef struct bftkv_quorum bftkv_quorum;

#define BFTKV_Q_READ 0x01   /* quorum/quorum.go:10-16 */
#define BFTKV_Q_WRITE 0x02
#define BFTKV_Q_AUTH 0x04
#define BFTKV_Q_CERT 0x08
#define BFTKV_Q_PEER 0x10

bftkv_graph* bftkv_host_graph_new(void);
void bftkv_host_graph_free(bftkv_graph* g);
/* AddNodes for one node: vertex id with the ids of the keys that certified it (graph.go:46-75). */
int bftkv_host_graph_add_node(bftkv_graph* g, uint64_t id, const uint64_t* signer_ids, uint32_t n_signers);
int bftkv_host_graph_set_self(bftkv_graph* g, uint64_t id);               /* SetSelfNodes, graph.go:77-88 */
int bftkv_host_graph_revoke(bftkv_graph* g, uint64_t id);                 /* Revoke, graph.go:131-140 */
/* GetReachableNodes (graph.go:279-295); ids_out may be NULL to query the count. */
int bftkv_host_graph_reachable(bftkv_graph* g, uint64_t sid, int distance, uint64_t* ids_out, uint32_t cap, uint32_t* n_out);
/* GetCliques (graph.go:297-320): concatenated member ids, per-clique sizes and weights. */
int bftkv_host_graph_cliques(bftkv_graph* g, uint64_t sid, int distance, uint64_t* ids_out, uint32_t ids_cap,
                             uint32_t* sizes_out, int32_t* weights_out, uint32_t cliques_cap, uint32_t* n_cliques_out);

/* Selector caching.  The reference redoes the clique search on every ChooseQuorum
 * (wotqs.go:95-127 -> graph.go:297-362); here every graph mutation bumps an epoch and both the GetCliques results and
 * the finished quorums are kept until the epoch moves.  On by default; off reproduces the reference's cost. */
int bftkv_host_graph_set_caching(bftkv_graph* g, int on);
int bftkv_host_graph_cache_stats(const bftkv_graph* g, uint64_t* epoch, uint64_t* hits, uint64_t* misses);
/* wot.ChooseQuorum(rw) (wotqs.go:117-127) from the graph's self vertex. */
bftkv_quorum* bftkv_host_choose_quorum(bftkv_graph* g, int rw);
/* A quorum from explicit cliques (tests, replay tools). */
bftkv_quorum* bftkv_host_quorum_from_qcs(const bftkv_gpu_qc* qcs, uint32_t n_qcs);
void bftkv_host_quorum_free(bftkv_quorum* q);
uint32_t bftkv_host_quorum_n_qcs(const bftkv_quorum* q);
int bftkv_host_quorum_qc(const bftkv_quorum* q, uint32_t i, bftkv_gpu_qc* out);   /* node_ids point into q */
/* Quorum interface (quorum/quorum.go:18-25, wotqs.go:132-193) on the host, one list at a time. */
int bftkv_host_quorum_is_quorum(const bftkv_quorum* q, const uint64_t* ids, uint32_t n);
int bftkv_host_quorum_is_threshold(const bftkv_quorum* q, const uint64_t* ids, uint32_t n);
int bftkv_host_quorum_is_sufficient(const bftkv_quorum* q, const uint64_t* ids, uint32_t n);
int bftkv_host_quorum_reject(const bftkv_quorum* q, const uint64_t* ids, uint32_t n);
int bftkv_host_quorum_get_threshold(const bftkv_quorum* q);
/* Registers the quorum with a GPU context (bftkv_gpu_quorum_create) once and returns the handle. */
int bftkv_host_quorum_gpu_handle(bftkv_quorum* q, bftkv_gpu_ctx* ctx, int* handle_out);

/* ---- vote collector (protocol/client.go, protocol/server.go) ---------------------------------- */
/* One Multicast reply as the callback sees it (transport/transport.go:129-136). */
typedef struct {
  uint64_t peer_id;
  int32_t err;                          /* 0: Data valid; != 0: res.Err (an index into err_strings of the caller) */
  const uint8_t* data; uint64_t data_len;
} bftkv_reply;

/* The verification sites below write one status byte per request / write.  Besides the error identities they mirror, an item
 * that contains a FENCED input shape (include/bftkv_gpu.h "fenced inputs") reports BFTKV_HOST_ERR_FENCED: no verdict is
 * claimed for it, the caller runs the reference path. */
#define BFTKV_HOST_ERR_FENCED 0xFC

/* Client.collectSignatures fold + final verification (client.go:139-169), for a batch of writes.
 * For write w the replies are replies[reply_off[w] .. reply_off[w+1]) in arrival order; each reply's
 * data is a serialized SignaturePacket (packet.ParseSignature).  The fold appends with Combine
 * (crypto_pgp.go:506-515) and stops at the first reply after which Combine returns true or Reject(failure)
 * holds; then CollectiveSignature.Verify(tbss, ss, qa) runs for ALL writes in one GPU batch.
 *   ss_out / ss_off_out   collected ss.Data per write (cap bytes)
 *   consumed_out[w]       replies folded before the callback stopped the multicast
 *   err_out[w]            BFTKV_ERR_NONE / BFTKV_ERR_INSUFFICIENT_SIGNATURES (majorityError is the caller's) */
int bftkv_host_collect_signatures(bftkv_gpu_ctx* ctx, bftkv_quorum* qa, uint32_t n_writes,
                                  const uint8_t* tbss_blob, const uint64_t* tbss_off,
                                  const bftkv_reply* replies, const uint64_t* reply_off,
                                  uint8_t* ss_out, uint64_t ss_cap, uint64_t* ss_off_out,
                                  uint32_t* consumed_out, uint8_t* err_out);

/* Server.write's verification site for a batch of requests <x,v,t,sig,ss> (server.go:286-302):
 * packet.Parse, TBSS, CollectiveSignature.Verify(tbss, ss, q).  err_out[i]: BFTKV_ERR_NONE,
 * BFTKV_ERR_INSUFFICIENT_SIGNATURES, or 0xFF for a malformed request / missing ss
 * (bftkv.ErrMalformedRequest, server.go:288-294). */
int bftkv_host_server_write_verify(bftkv_gpu_ctx* ctx, bftkv_quorum* q, uint32_t n_requests,
                                   const uint8_t* req_blob, const uint64_t* req_off, uint8_t* err_out);

/* Votes of a Multicast round folded the way Client.Write does it (client.go:67-86, 108-123): an accepted reply
 * appends the peer to `actives` and stops when IsThreshold(actives); a failed one appends to `failure` and stops
 * when Reject(failure).  ok[r] != 0: accepted.  consumed_out[i]: replies folded before the callback stopped the
 * multicast; threshold_out[i]: the final IsThreshold(actives) the caller tests. */
int bftkv_host_vote_fold(const bftkv_quorum* q, uint32_t n_rounds, const uint64_t* peer_ids, const uint8_t* ok,
                         const uint64_t* reply_off, uint32_t* consumed_out, uint8_t* threshold_out);

/* Certificates carried in SignaturePacket.Cert: PGPCertificate.Parse / PGPSignature.Issuer
 * (crypto_pgp.go:236-249, 392-405).  The certificate is walked the way openpgp.ReadEntity goes through its packets
 * (x/crypto openpgp/keys.go ReadEntity / addUserID / addSubkey; the rules are listed in oracle/openpgp.py
 * walk_certificate): per entity the key material (a key's identity and hash input are its re-serialization, not
 * its packet body), the key flags KeysByIdUsage reads (the chosen identity self-signature; Subkey.Sig), the
 * issuer ids of PGPCertificateInstance.Signers (crypto_pgp.go:80-88: the signatures on identities that have a
 * self-signature) and the list of signatures ReadEntity verifies.  No signature is verified here
 * (bftkv_host_certs_verify does that on the GPU).  Every entity the walk meets is listed, refused ones included
 * (Parse itself stops at the first one ReadEntity refuses): the issuer of a request is entity 0. */
typedef struct bftkv_certs bftkv_certs;
bftkv_certs* bftkv_host_certs_parse(const uint8_t* cert, uint64_t len);
void bftkv_host_certs_free(bftkv_certs* c);
uint32_t bftkv_host_certs_n_entities(const bftkv_certs* c);
/* entity e: primary key id, number of keys (primary + subkeys; 0: the primary key packet did not parse or is of a
 * shape left to the reference), certifier ids (Signers(), in packet order) */
int bftkv_host_certs_entity(const bftkv_certs* c, uint32_t e, uint64_t* id_out, uint32_t* n_keys_out,
                            const uint64_t** certifiers_out, uint32_t* n_certifiers_out);
int bftkv_host_certs_key(const bftkv_certs* c, uint32_t e, uint32_t k, bftkv_gpu_pubkey* out);   /* pointers into c */
/* What the walk alone says about entity e.  *refused_out: ReadEntity returns an error whatever the signatures say
 * (no identity with a self-signature, a subkey without or with a wrong-typed signature, a signing subkey without
 * cross-signature, a packet that does not parse, a primary key that cannot sign, ...).  *unknown_out: the entity has
 * a shape this library does not follow (elliptic-curve or version-3 keys, secret-key packets, user attributes and
 * other packet types whose parsers are not restated, bodies over 4096 bytes, partial lengths, embedded signatures
 * nested beyond the parser's bound, a certification without issuer subpacket -- on which the reference's Signers()
 * dereferences nil): no verdict, the reference decides.  Nothing is refused BEHIND such a shape: what it hides may be
 * exactly the self-signature or binding whose absence would be the refusal.  *why_out: the reference's
 * message or the shape (a string owned by c).  *n_checks_out: the signatures ReadEntity verifies. */
int bftkv_host_certs_structure(const bftkv_certs* c, uint32_t e, uint8_t* refused_out, uint8_t* unknown_out,
                               const char** why_out, uint32_t* n_checks_out);
/* What ReadEntity does with each packet that packet.Reader.Next() hands it inside entity e, in order (packet types
 * unknown to x/crypto are skipped by Next and have no entry): roles_out[i] = BFTKV_ROLE_* | index << 8 | chosen << 24
 * (bftkv_gpu.h).  *start_out / *len_out: the entity's bytes within the certificate.  Pointers into c. */
int bftkv_host_certs_roles(const bftkv_certs* c, uint32_t e, uint64_t* start_out, uint64_t* len_out,
                           const uint32_t** roles_out, uint32_t* n_out);
/* check i of entity e, in the order ReadEntity meets them (third-party certifications, which it does not verify,
 * come last but for key revocations).  kind: 0 user-id self-signature, 1 subkey binding / subkey revocation,
 * 2 third-party certification (CheckQuorumCert only), 3 cross-signature of a signing subkey, 4 key revocation;
 * key_index: the key of the entity it is verified with; the bytes it is computed over; the signature packet. */
int bftkv_host_certs_check(const bftkv_certs* c, uint32_t e, uint32_t i, int* kind_out, uint32_t* key_index_out,
                           const uint8_t** signed_out, uint64_t* signed_len_out, const uint8_t** sig_out, uint64_t* sig_len_out);

/* v4 fingerprint (SHA-1 over 0x99 || u16 length || the key as re-serialized) of the primary key of the first entity */
int bftkv_host_cert_fingerprint(const uint8_t* cert, uint64_t len, uint8_t out[20]);

/* What openpgp.ReadEntity verifies while reading, on the GPU, with the entity's own keys: every
 * user-id self-signature (0x10 / 0x13 issued by the primary key, over 0x99 len key || 0xB4 len uid), every subkey
 * binding and subkey revocation (0x18 / 0x28, over 0x99 len key || 0x99 len subkey), the embedded 0x19
 * cross-signature of every binding whose key flags say "sign" (same bytes, under the subkey), and every key
 * revocation outside a user-id / subkey run (0x20, over the key alone).  valid_out[e] = 1 iff ReadEntity returns the
 * entity; 0 iff it refuses it (bftkv_host_certs_structure's refusals, or a check that fails); 2: no verdict -- a
 * shape left to the reference, or a check that met a fenced shape (e.g. a DSA certificate key beyond the bounded
 * table slots of certificate keys); nothing is remembered about such a certificate. */
int bftkv_host_certs_verify(bftkv_gpu_ctx* ctx, const uint8_t* cert, uint64_t len, uint8_t* valid_out, uint32_t cap, uint32_t* n_out);

/* CheckQuorumCert as the paper states it (docs/tex/algo.tex:68-83; the code only counts certifier key ids,
 * server.go:211-214): the third-party certifications on the FIRST entity of `cert` that verify under a key of the node
 * keyring; ok_out = q.IsThreshold(verified certifiers). */
int bftkv_host_quorum_cert_verify(bftkv_gpu_ctx* ctx, const bftkv_quorum* q, const uint8_t* cert, uint64_t len, uint8_t* ok_out,
                                  uint64_t* verified_ids_out, uint32_t cap, uint32_t* n_out);

/* Server.sign's verification site for a batch of requests (server.go:189-214): packet.Parse; Issuer(sig) = first
 * entity of sig.Cert; VerifyWithCertificate(TBS(req), sig, issuer) on the GPU; then
 * ChooseQuorum(AUTH|CERT).IsThreshold(Certificate.Signers(issuer)) with the certifier ids looked up in the
 * node keyring (crypto_pgp.go:263-272).  An issuer entity that ReadEntity would refuse (bftkv_host_certs_verify) is
 * no issuer at all.  err_out: 0 ok, BFTKV_ERR_INVALID_SIGNATURE,
 * 0xFF malformed request / nil sig, 0xFE crypto.ErrCertificateNotFound, 0xFD bftkv.ErrInvalidQuorumCertificate. */
int bftkv_host_server_sign_verify(bftkv_gpu_ctx* ctx, const bftkv_quorum* q_cert, uint32_t n_requests,
                                  const uint8_t* req_blob, const uint64_t* req_off, uint8_t* err_out);

/* Server.read's proof check for a batch of read requests <x, nil, 0, nil, proof> (server.go:181-185, for variables written
 * with an authentication attribute): CollectiveSignature.Verify(variable, proof, ChooseQuorum(AUTH)) -- the signed bytes
 * are the VARIABLE NAME.  err_out: 0 ok, BFTKV_HOST_ERR_AUTH_FAILURE (proof missing or insufficient:
 * bftkv.ErrAuthenticationFailure), 0xFF malformed, BFTKV_HOST_ERR_FENCED. */
#define BFTKV_HOST_ERR_AUTH_FAILURE 0xFB
int bftkv_host_server_read_proof_verify(bftkv_gpu_ctx* ctx, bftkv_quorum* q_auth, uint32_t n_requests,
                                        const uint8_t* req_blob, const uint64_t* req_off, uint8_t* err_out);

/* Server.register's verification site (server.go:452-475): sig and ss both present; Issuer(sig) = first entity of sig.Cert
 * (ReadEntity-valid); VerifyWithCertificate(TBS(req), sig, issuer); then CollectiveSignature.Verify(variable, ss,
 * ChooseQuorum(AUTH)).  err_out: 0 ok, BFTKV_ERR_INVALID_SIGNATURE, BFTKV_ERR_INSUFFICIENT_SIGNATURES, 0xFF malformed,
 * 0xFE crypto.ErrCertificateNotFound, BFTKV_HOST_ERR_FENCED. */
int bftkv_host_server_register_verify(bftkv_gpu_ctx* ctx, bftkv_quorum* q_auth, uint32_t n_requests,
                                      const uint8_t* req_blob, const uint64_t* req_off, uint8_t* err_out);

/* Equivocation tally of Client.revoke (client.go:304-353) for one variable: values[v] of replies at the same
 * timestamp t != 0, value group g[v]; ids_out: the signer ids (sorted) that appear, by PARSE ONLY
 * (CollectiveSignature.Signers), under two different value groups. */
int bftkv_host_equivocation_signers(bftkv_gpu_ctx* ctx, uint32_t n_values, const uint32_t* group, const uint8_t* ss_blob,
                                    const uint64_t* ss_off, uint64_t* ids_out, uint32_t cap, uint32_t* n_out);

/* Diagnostic: Signature.parse / SignatureV3.parse as the verifier's kernels restate it (the same code, run on the host over
 * one signature packet BODY), so that a CPU-only test can fuzz the parser against the oracle.  parsed = 0: a parse error
 * (too_deep = 1: an embedded signature nested deeper than the parser goes -- a fenced shape, no claim). */
typedef struct {
  uint8_t parsed, too_deep, version, sig_type, pk_algo, hash_id, have_issuer, n_mpi;
  uint8_t hash_tag[2];
  uint16_t hashed_len;
  uint16_t mpi_bits[2];
  uint32_t mpi_off[2];
  uint64_t issuer;
} bftkv_sig_parse;
int bftkv_host_parse_signature(const uint8_t* body, uint32_t len, bftkv_sig_parse* out);

/* Diagnostic: packet.Read framing as the kernels restate it (walk_step), over one stream on the host: per packet event its
 * status (99 = a signature packet located, Signature.parse still to come; else a final BFTKV_ST_* status), body offset and
 * length; *n_out = number of events (may exceed cap).  Unknown packet types raise no event (Reader.Next skips them). */
int bftkv_host_walk_stream(const uint8_t* data, uint64_t len, uint32_t cap, uint8_t* status_out, uint64_t* body_off_out,
                           uint32_t* body_len_out, uint32_t* n_out);
/* PGPSignature.Signers' walk over ONE stream on the HOST (crypto/pgp/crypto_pgp.go:373-390), for callers that hold a single ss.Data:
 * CollectiveSignature.Combine asks for the signers after every signature it appends (protocol/client.go:153), and a parse-only walk
 * of a few dozen packets is microseconds on the caller's thread -- no device round trip, no context.  The code is the device
 * kernel's (k_signers: shared __host__ __device__ helpers).  issuers_out: the issuer key id of every version-4 signature packet
 * Reader.Next yields before its first error, in packet order, NOT filtered by a keyring (the reference looks each up with
 * getCertById: the caller does); *fenced_out as bftkv_gpu_signers_fenced (a shape on which the walk does not follow the reference's
 * reader, or a v4 signature without issuer: take the reference path).  BFTKV_E_NOMEM with *n_out set when cap is too small. */
int bftkv_host_signers_walk(const uint8_t* ss, uint64_t len, uint64_t* issuers_out, uint32_t cap, uint32_t* n_out, uint8_t* fenced_out);
/* Diagnostic: what the kernels decide about one signature stream before any key or hash is involved (framing, partial-length
 * bodies linearised, Signature.parse / SignatureV3.parse, where the reference's reader stands after each parsed signature), on
 * the host through the same code.  status_out[i]: BFTKV_ST_NOT_SIGNATURE, BFTKV_ST_PARSE_ERROR, BFTKV_ST_UNSUPPORTED, or 99 for a
 * signature body that parsed; *fenced_out: the stream holds a packet after which the reference's reader is not followed.
 * (crypto/pgp/crypto_pgp.go:486-498 -> x/crypto openpgp/packet.Read) */
int bftkv_host_scan_stream(const uint8_t* data, uint64_t len, uint32_t cap, uint8_t* status_out, uint32_t* n_out, uint8_t* fenced_out);

/* emsaEncode (crypto/threshold/rsa/rsa.go:356-378): 00 01 FF.. 00 prefix digest, emlen = ceil(bits(N)/8);
 * hash_id is the OpenPGP hash id (2, 8, 9, 10, 11).  BFTKV_E_INVALID when padlen < 3 (crypto.ErrInvalidInput). */
int bftkv_host_emsa_encode(int hash_id, const uint8_t* digest, uint32_t digest_len, uint32_t n_bits, uint8_t* em_out, uint32_t cap);

/* Client.Read tally (client.go:181-205) for a batch of variables: replies (peer, t, value) in arrival
 * order; value_idx_out[r] = index of the reply whose value wins at the maximum timestamp, or -1 for
 * errInProgress. */
int bftkv_host_max_timestamped_value(const bftkv_quorum* q, uint32_t n_reads, const uint64_t* peer_ids, const uint64_t* ts,
                                     const uint8_t* value_blob, const uint64_t* value_off, const uint64_t* reply_off,
                                     int64_t* value_idx_out);
/* The same fold over ALL replies received with the verifier's error byte per reply (bftkv_gpu_collective_verify's err_out):
 * replies with reply_err[i] != 0 are dropped as Client.Read drops failures, and value_idx_out counts the accepted replies of
 * the read only -- identical to filtering the arrays first, without the caller rebuilding them per batch. */
int bftkv_host_max_timestamped_value_masked(const bftkv_quorum* q, uint32_t n_reads, const uint64_t* peer_ids, const uint64_t* ts,
                                            const uint8_t* value_blob, const uint64_t* value_off, const uint64_t* reply_off,
                                            const uint8_t* reply_err, int64_t* value_idx_out);

/* Framing half of the transport message check (crypto_pgp.go:453-471 -> openpgp.ReadMessage / readSignedMessage), no
 * GPU and no keyring involved: walks [one-pass signature] [literal data] [signature] of ONE already-decrypted message.
 *   framing_out   BFTKV_MSG_READ_ERROR / _NOT_SIGNED / _UNSUPPORTED / _SIGNATURE_ERROR (literal not followed by a
 *                 signature packet), or 0xFF: a signature packet follows and the verdict is the device's
 *                 (bftkv_gpu_message_verify additionally reports _UNVERIFIED when the keyring has no signing key for it)
 *   signer_out    one-pass key id; hash_out: one-pass hash id; plain: literal body (partial lengths removed)
 *   sig_off/len   the trailing signature packet inside msg (when framing_out == 0xFF)
 * bftkv_gpu_message_verify performs exactly this walk before it batches the signatures. */
int bftkv_host_message_frame(const uint8_t* msg, uint64_t len, uint8_t* framing_out, uint64_t* signer_out, uint8_t* hash_out,
                             uint8_t* plain_out, uint64_t plain_cap, uint64_t* plain_len_out, uint8_t* fname_out /*[256]*/,
                             uint8_t* fname_len_out, uint64_t* sig_off_out, uint64_t* sig_len_out);

/* SHA-256 with the compression the micro-batcher's callers run over their own payload before they queue
 * (bftkv_gpu_batcher_*: the device receives midstates; reference: the hash inside openpgp.CheckDetachedSignature,
 * crypto/pgp/crypto_pgp.go:490).  mode 0: the fastest this CPU offers (SHA extensions when present), 1: the portable
 * rounds.  Returns 1 when the CPU has the SHA extensions, 0 when not, < 0 on bad arguments. */
int bftkv_host_sha256(const uint8_t* data, uint64_t len, int mode, uint8_t out[32]);

#ifdef __cplusplus
}
#endif
#endif /* BFTKV_HOST_H */
