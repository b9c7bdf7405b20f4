"""-m gpu: HIP path vs the CPU oracle through the C ABI, bit-exact."""
import numpy as np
import pytest

from corpus import build as cb
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_modexp_matches_pow(gpu_ctx):
    rng = np.random.default_rng(7)
    cl = cb.make_cluster(4)
    mods = np.stack([np.frombuffer(r.n.to_bytes(256, "big"), dtype=np.uint8) for r in cl.replicas])
    exps = np.stack([np.frombuffer(r.d.to_bytes(256, "big"), dtype=np.uint8) for r in cl.replicas])
    n = 70
    base = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
    base[:, 0] &= 0x3F
    base[0] = 0
    base[1] = 0; base[1, -1] = 1
    idx = rng.integers(0, 4, size=n).astype(np.uint32)
    out = gpu_ctx.modexp(base, idx, mods, exps)
    for i in range(n):
        kp = cl.replicas[int(idx[i])]
        want = pow(int.from_bytes(base[i].tobytes(), "big"), kp.d, kp.n)
        assert int.from_bytes(out[i].tobytes(), "big") == want, i
    # small public exponents too (different exponent per modulus)
    e2 = np.zeros((4, 4), dtype=np.uint8)
    es = [65537, 3, 17, 1]
    for j, e in enumerate(es):
        e2[j] = np.frombuffer(e.to_bytes(4, "big"), dtype=np.uint8)
    out = gpu_ctx.modexp(base, idx, mods, e2)
    for i in range(n):
        kp = cl.replicas[int(idx[i])]
        want = pow(int.from_bytes(base[i].tobytes(), "big"), es[int(idx[i])], kp.n)
        assert int.from_bytes(out[i].tobytes(), "big") == want, i


@pytest.mark.parametrize("n,items", [(4, 100), (10, 60), (64, 24)])
def test_collective_verify_matches_oracle(gpu_ctx, n, items):
    cl = cb.make_cluster(n)
    rates = {cb.MUT_BAD_MPI: 0.1, cb.MUT_UNKNOWN_ISSUER: 0.1, cb.MUT_DUP_SIGNER: 0.1, cb.MUT_ONE_SHORT: 0.15, cb.MUT_BAD_TAG: 0.1}
    c = cb.make_write_corpus(cl, items, mutation_rates=rates)
    kr = H.oracle_keyring(cl)
    q = H.clique_quorum(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    err, nver, verdict = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    st, st_item = gpu_ctx.last_statuses()
    want_st = []
    for i in range(items):
        r = H.oracle_collective(kr, q, c, i)
        assert (err[i] == 0) == (r.err is None), (i, c.mutation[i], r.statuses)
        assert nver[i] == len(r.verified), (i, nver[i], len(r.verified))
        # the oracle stops at the early exit; the GPU verifies every packet: compare the prefix
        got = st[st_item == i]
        assert list(got[:len(r.statuses)]) == r.statuses, (i, list(got), r.statuses)
    assert set(np.unique(err)) <= {0, 2}
    assert (err == 0).any() and (err == 2).any()
    gpu_ctx.quorum_destroy(qh)
