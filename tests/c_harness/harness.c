/* Plain-C caller of libbftkv_gpu.so through the headers under include/ only -- the position a cgo preamble is in.
 *   gcc -std=c99 -I include tests/c_harness/harness.c -L bftkv_amd -lbftkv_gpu -o harness
 * Without a GPU the verifier context cannot be created (and nothing falls back to the CPU); the host-side
 * entry points (packet framing, trust graph, quorum system) run anywhere.  With a GPU (argv[1] = "gpu") it also
 * uploads a keyring-less context and runs an empty batch.  Prints KEY=VALUE lines for tests/test_c_harness.py. */
#include <stdio.h>
#include <string.h>
#include "bftkv_host.h"

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

  if (argc > 1 && strcmp(argv[1], "gpu") == 0 && ctx) {
    uint64_t off0[1] = {0};
    rc = bftkv_gpu_keyring_set(ctx, NULL, 0);
    printf("gpu keyring_set_rc=%d\n", rc);
    (void)off0;
  }
  if (ctx) bftkv_gpu_destroy(ctx);
  return 0;
}
