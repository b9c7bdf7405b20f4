/* Plain-C caller of libbftkv_gpu.so through the headers under include/ only -- the position a cgo preamble is in.
 *   gcc -std=c99 -I include tests/c_harness/harness.c -L bftkv_amd -lbftkv_gpu -o harness
 * Without a GPU the verifier context cannot be created (and nothing falls back to the CPU); the host-side
 * entry points (packet framing, trust graph, quorum system) run anywhere.  With a GPU (argv[1] = "gpu") it uploads the
 * 4-key ring of fixture.h, creates the clique quorum and verifies the fixture's 8 signed writes twice: as one batch
 * (bftkv_gpu_collective_verify) and one by one through the micro-batcher, the way one goroutine per request would; it also
 * provokes an infrastructure error to show that the status byte fails closed.  Prints KEY=VALUE lines for
 * tests/test_c_harness.py, which checks them against the oracle. */
#include <stdio.h>
#include <string.h>
#include "bftkv_host.h"
#include "fixture.h"

static void print_u8(const char* key, const uint8_t* v, int n) {
  printf("%s=", key);
  for (int i = 0; i < n; ++i) printf("%s%u", i ? "," : "", (unsigned)v[i]);
  printf("\n");
}

int main(int argc, char** argv) {
  bftkv_gpu_ctx* ctx = NULL;
  int rc = bftkv_gpu_init(0, &ctx);
  printf("init_rc=%d\n", rc);
  printf("err_invalid=%s\n", bftkv_gpu_error_string(BFTKV_ERR_INVALID_SIGNATURE));
  printf("err_insufficient=%s\n", bftkv_gpu_error_string(BFTKV_ERR_INSUFFICIENT_SIGNATURES));

  /* packet.Serialize(x, v, t) -> Parse -> TBS */
  uint8_t pkt[256];
  uint64_t n = 0, tbs = 0;
  rc = bftkv_host_packet_serialize(3, (const uint8_t*)"key", 3, (const uint8_t*)"value", 5, 42, NULL, NULL, NULL, 0, pkt, sizeof pkt, &n);
  bftkv_parsed p;
  memset(&p, 0, sizeof p);
  int rc2 = bftkv_host_packet_parse(pkt, n, &p);
  int rc3 = bftkv_host_packet_tbs(pkt, n, &tbs);
  printf("packet=%d,%d,%d len=%llu x_len=%llu v_len=%llu t=%llu tbs=%llu has_sig=%d\n", rc, rc2, rc3, (unsigned long long)n,
         (unsigned long long)p.x_len, (unsigned long long)p.v_len, (unsigned long long)p.t, (unsigned long long)tbs, p.has_sig);

  /* a 4-clique that certified the client 9: ChooseQuorum(AUTH) from the client's view (wotqs.go:36-70, n=4: f=1) */
  bftkv_graph* g = bftkv_host_graph_new();
  uint64_t ids[4] = {1, 2, 3, 4};
  for (int i = 0; i < 4; ++i) {
    uint64_t signers[4];
    uint32_t k = 0;
    for (int j = 0; j < 4; ++j) if (j != i) signers[k++] = ids[j];
    if (i >= 2) signers[k++] = 9;              /* the client trusts the members that did not certify it */
    bftkv_host_graph_add_node(g, ids[i], signers, k);
  }
  uint64_t cert[2] = {1, 2};
  bftkv_host_graph_add_node(g, 9, cert, 2);
  bftkv_host_graph_set_self(g, 9);
  bftkv_quorum* q = bftkv_host_choose_quorum(g, BFTKV_Q_AUTH);
  bftkv_gpu_qc qc;
  memset(&qc, 0, sizeof qc);
  uint32_t nq = bftkv_host_quorum_n_qcs(q);
  if (nq) bftkv_host_quorum_qc(q, 0, &qc);
  uint64_t three[3] = {1, 2, 3}, two[2] = {1, 1};
  printf("quorum n_qcs=%u f=%d min=%d threshold=%d suff=%d n_nodes=%u suff3=%d suff_dup=%d thr3=%d reject2=%d\n", nq, qc.f, qc.min, qc.threshold,
         qc.suff, qc.n_nodes, bftkv_host_quorum_is_sufficient(q, three, 3), bftkv_host_quorum_is_sufficient(q, two, 2),
         bftkv_host_quorum_is_threshold(q, three, 3), bftkv_host_quorum_reject(q, two, 2));
  bftkv_host_quorum_free(q);
  bftkv_host_graph_free(g);

  /* PGPSignature.Signers over one ss.Data on the host (what Combine asks after every appended signature): every packet of the
   * fixture's first write names a replica of the ring, nothing fenced; cut in mid-packet the walk ends one signer short */
  {
    uint64_t iss[64];
    uint32_t ns = 0, in_ring = 0;
    uint8_t sfen = 0xEE;
    int rcs = bftkv_host_signers_walk(fx_ss, fx_ss_off[1], iss, 64, &ns, &sfen);
    for (uint32_t i = 0; i < ns; ++i) for (int k = 0; k < FX_N_KEYS; ++k) in_ring += iss[i] == fx_key_id[k];
    uint32_t ns_cut = 0;
    uint8_t sfen_cut = 0xEE;
    int rcc = bftkv_host_signers_walk(fx_ss, fx_ss_off[1] - 5, iss, 64, &ns_cut, &sfen_cut);
    printf("host_signers_walk=%d,%u,%u,%u cut=%d,%u,%u\n", rcs, ns, in_ring, (unsigned)sfen, rcc, ns_cut, (unsigned)sfen_cut);
  }

  if (argc > 1 && strcmp(argv[1], "gpu") == 0 && ctx) {
    /* keyring: the replicas in order (secring-first order is the caller's business, crypto_pgp.go:195-197) */
    bftkv_gpu_pubkey keys[FX_N_KEYS];
    memset(keys, 0, sizeof keys);
    for (int i = 0; i < FX_N_KEYS; ++i) {
      keys[i].key_id = fx_key_id[i]; keys[i].entity_id = fx_key_id[i]; keys[i].pk_algo = 1; keys[i].usable_sign = 1;
      keys[i].n = fx_key_n + 256 * i; keys[i].n_len = 256;
      keys[i].e = fx_key_e + 4 * i; keys[i].e_len = 4;
    }
    rc = bftkv_gpu_keyring_set(ctx, keys, FX_N_KEYS);
    printf("gpu_keyring_set_rc=%d\n", rc);
    bftkv_gpu_qc clique;
    clique.f = FX_F; clique.min = FX_MIN; clique.threshold = FX_THRESHOLD; clique.suff = FX_SUFF;
    clique.node_ids = fx_key_id; clique.n_nodes = FX_N_KEYS;
    int qh = -1;
    rc = bftkv_gpu_quorum_create(ctx, &clique, 1, &qh);
    printf("gpu_quorum_create_rc=%d\n", rc);
    /* CollectiveSignature.Verify over all writes in one batch (crypto_pgp.go:485-500) */
    uint8_t err[FX_N_WRITES], verdict[FX_N_WRITES], fenced[FX_N_WRITES];
    uint32_t nver[FX_N_WRITES];
    rc = bftkv_gpu_collective_verify(ctx, qh, FX_N_WRITES, fx_tbss, fx_tbss_off, fx_ss, fx_ss_off, err, nver, verdict, fenced);
    printf("gpu_verify_rc=%d\n", rc);
    print_u8("gpu_verify_err", err, FX_N_WRITES);
    print_u8("gpu_verify_fenced", fenced, FX_N_WRITES);
    printf("gpu_verify_nver=");
    for (int i = 0; i < FX_N_WRITES; ++i) printf("%s%u", i ? "," : "", nver[i]);
    printf("\n");
    for (int i = 0; i < FX_N_WRITES; ++i)
      if (err[i] != BFTKV_ERR_NONE) { printf("gpu_first_error_string=%s\n", bftkv_gpu_error_string(err[i])); break; }
    /* the same writes one call at a time through the micro-batcher */
    bftkv_gpu_batcher* b = bftkv_gpu_batcher_create(ctx, 4, 100);
    uint8_t berr[FX_N_WRITES], bfen[FX_N_WRITES];
    int brc = 0;
    for (int i = 0; i < FX_N_WRITES; ++i)
      brc |= bftkv_gpu_batcher_collective_verify(b, qh, fx_tbss + fx_tbss_off[i], fx_tbss_off[i + 1] - fx_tbss_off[i], fx_ss + fx_ss_off[i],
                                                 fx_ss_off[i + 1] - fx_ss_off[i], &berr[i], &bfen[i]);
    printf("gpu_batcher_rc=%d\n", brc);
    print_u8("gpu_batcher_err", berr, FX_N_WRITES);
    /* Signature.Verify of one packet of write 3 (all of its packets are good) against the node keyring */
    uint8_t serr = 0xEE, sfen = 0xEE;
    uint64_t one_off[2];
    one_off[0] = 0; one_off[1] = 287;
    uint64_t tb_off[2];
    tb_off[0] = 0; tb_off[1] = fx_tbss_off[4] - fx_tbss_off[3];
    rc = bftkv_gpu_signature_verify(ctx, 1, fx_tbss + fx_tbss_off[3], tb_off, fx_ss + fx_ss_off[3], one_off, NULL, &serr, &sfen);
    printf("gpu_signature_verify=%d,%u,%u\n", rc, (unsigned)serr, (unsigned)sfen);
    /* infrastructure error: no such quorum -- the return code says so AND the status byte is a failure */
    uint8_t e_bad = 0, f_bad = 0;
    int rc_bad = bftkv_gpu_batcher_collective_verify(b, qh + 77, fx_tbss, fx_tbss_off[1], fx_ss, fx_ss_off[1], &e_bad, &f_bad);
    printf("gpu_fail_closed=%d,%u\n", rc_bad < 0, (unsigned)e_bad);
    /* the same batch cut into pieces verified while the rest crosses PCIe (bftkv_gpu_set_host_pipeline): same answers */
    uint8_t perr[FX_N_WRITES], pverdict[FX_N_WRITES], pfenced[FX_N_WRITES];
    uint32_t pnver[FX_N_WRITES];
    rc = bftkv_gpu_set_host_pipeline(ctx, 3);
    rc |= bftkv_gpu_collective_verify(ctx, qh, FX_N_WRITES, fx_tbss, fx_tbss_off, fx_ss, fx_ss_off, perr, pnver, pverdict, pfenced);
    rc |= bftkv_gpu_set_host_pipeline(ctx, 0);
    printf("gpu_pipelined=%d,%d\n", rc, memcmp(perr, err, sizeof err) == 0 && memcmp(pnver, nver, sizeof nver) == 0 &&
                                          memcmp(pverdict, verdict, sizeof verdict) == 0 && memcmp(pfenced, fenced, sizeof fenced) == 0);
    /* Signature.Issuer + VerifyWithCertificate for a principal that is not in the keyring: its certificate travels with the
     * request (server.go:199-207); then the same certificate with a forged self-signature (ReadEntity refuses it: no issuer) */
    uint8_t cerr = 0xEE, cfen = 0xEE, fp[20];
    uint64_t iid = 0;
    rc = bftkv_gpu_batcher_cert_verify(b, fx_client_cert, sizeof fx_client_cert, fx_client_tbs, sizeof fx_client_tbs, fx_client_sig,
                                       sizeof fx_client_sig, &cerr, &cfen, &iid, fp);
    printf("gpu_cert_verify=%d,%u,%u,%d\n", rc, (unsigned)cerr, (unsigned)cfen, iid == FX_CLIENT_KEY_ID);
    rc = bftkv_gpu_batcher_cert_verify(b, fx_client_cert_forged, sizeof fx_client_cert_forged, fx_client_tbs, sizeof fx_client_tbs,
                                       fx_client_sig, sizeof fx_client_sig, &cerr, &cfen, &iid, fp);
    printf("gpu_cert_verify_forged=%d,%u,%u\n", rc, (unsigned)cerr, (unsigned)cfen);
    rc = bftkv_gpu_batcher_cert_verify(b, fx_client_cert, sizeof fx_client_cert, fx_client_tbs, sizeof fx_client_tbs - 1, fx_client_sig,
                                       sizeof fx_client_sig, &cerr, &cfen, &iid, fp);
    printf("gpu_cert_verify_other_bytes=%d,%u,%u\n", rc, (unsigned)cerr, (unsigned)cfen);
    /* config 5 behind the same handle: ONE share-combine operation per call (what one DistSign's ThresholdProcess ends in).
     * Under the first replica's modulus N: (N-1)(N-1)(N-1) = -1 mod N; the shares f(1), f(2), f(3) of f(x) = 7 + 3x + 2x^2
     * recover f(0) = 7 (sss.go:81-92); 3^5 mod N = 243; an even modulus is refused for this caller and fails closed. */
    {
      uint8_t nm1[3 * 256], ys[3 * 256], base[256], out[256], st = 0x55;
      const uint8_t* N = fx_key_n;
      for (int j = 0; j < 3; ++j) { memcpy(nm1 + 256 * j, N, 256); nm1[256 * j + 255] ^= 1; }      /* N is odd: N - 1 clears the low bit */
      rc = bftkv_gpu_batcher_modmul_product(b, 3, nm1, 256, N, out, &st);
      printf("gpu_th_product=%d,%u,%d\n", rc, (unsigned)st, memcmp(out, nm1, 256) == 0);
      int32_t xs[3] = {1, 2, 3};
      memset(ys, 0, sizeof ys);
      ys[255] = 12; ys[256 + 255] = 21; ys[512 + 255] = 34;
      rc = bftkv_gpu_batcher_lagrange_combine(b, 3, xs, ys, 256, N, out, &st);
      int zeros = 1;
      for (int i = 0; i < 255; ++i) zeros &= out[i] == 0;
      printf("gpu_th_lagrange=%d,%u,%d,%u\n", rc, (unsigned)st, zeros, (unsigned)out[255]);
      uint8_t e5 = 5;
      memset(base, 0, sizeof base); base[255] = 3;
      rc = bftkv_gpu_batcher_modexp(b, base, 256, &e5, 1, N, out, &st);
      zeros = 1;
      for (int i = 0; i < 255; ++i) zeros &= out[i] == 0;
      printf("gpu_th_modexp=%d,%u,%d,%u\n", rc, (unsigned)st, zeros, (unsigned)out[255]);
      memset(out, 0x77, sizeof out);
      rc = bftkv_gpu_batcher_modmul_product(b, 3, ys, 256, nm1, out, &st);       /* N - 1 is even */
      zeros = 1;
      for (int i = 0; i < 256; ++i) zeros &= out[i] == 0;
      printf("gpu_th_fail_closed=%d,%u,%d\n", rc, (unsigned)st, zeros);
    }
    bftkv_gpu_batcher_destroy(b);
    bftkv_gpu_quorum_destroy(ctx, qh);
  }
  if (ctx) bftkv_gpu_destroy(ctx);
  return 0;
}
