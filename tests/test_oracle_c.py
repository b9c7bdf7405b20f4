"""CPU: the C restatement (oracle/c/oracle.c, the cpu_baseline 'port') agrees with the Python oracle packet by
packet, and the collective-signature semantics hold on mutated corpora."""
import numpy as np
import pytest

from corpus import build as cb
from oracle import collective as col
from oracle.cbind import COracle
from oracle.packet import SignaturePacket
from tests import helpers as H

RATES = {cb.MUT_BAD_MPI: 0.1, cb.MUT_UNKNOWN_ISSUER: 0.1, cb.MUT_DUP_SIGNER: 0.1, cb.MUT_ONE_SHORT: 0.15, cb.MUT_BAD_TAG: 0.1}


@pytest.mark.parametrize("n,dsa,items,kind", [(4, 0.0, 40, "dsa2048"), (10, 0.3, 30, "dsa2048"),
                                              (12, 0.75, 24, ("dsa1024", "dsa3072", "dsa1536", "dsa2048"))])      # DSA of every group size in use
def test_c_oracle_matches_python_oracle(n, dsa, items, kind):
    cl = cb.make_cluster(n, dsa_fraction=dsa, dsa_kind=kind)
    c = cb.make_write_corpus(cl, items, mutation_rates=RATES)
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    co = COracle()
    co.set_keyring(kr)
    co.set_quorum(q)
    err, nver, ops = co.collective_verify(c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off, 2)
    for i in range(items):
        r = H.oracle_collective(kr, q, c, i)
        tr, nv, e = co.trace_item(c.tbss(i), c.ss_data(i))
        assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified) == nv and tr == r.statuses, i
    assert ops > 0 and (err == 0).any() and (err == 2).any()


def test_c_oracle_matches_python_oracle_on_random_framings():
    """Random packet framings (every header format, unknown / non-signature packets, stray bytes, truncations, > 96 events):
    the C port walks them exactly like the Python oracle -- the same streams the GPU walk is checked on."""
    cl = cb.make_cluster(7, dsa_fraction=0.3)
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    co = COracle()
    co.set_keyring(kr)
    co.set_quorum(q)
    tbs_l, ss_l, _, _ = H.random_framing_streams(cl, 90)
    n_err = 0
    for t, s in zip(tbs_l, ss_l):
        r = col.collective_verify(kr, t, SignaturePacket(1, 0, False, s or None, None), q)
        tr, nv, e = co.trace_item(t, s)
        assert tr == r.statuses and nv == len(r.verified) and (e == 0) == (r.err is None)
        n_err += r.err is not None
    assert 0 < n_err < 90


def test_c_oracle_matches_python_oracle_on_the_reader_model():
    """Both restatements of packet.Read's readers (Python: reader objects all the way; C: reader objects + a replay of the parser's
    reads) on 1,500 streams of plausible unsigned bodies in every framing, issuers known and unknown: the same statuses packet by
    packet (unknown issuers are skipped INSIDE a call, known ones end it at the hash tag), the same exit counts and errors."""
    cl = cb.make_cluster(5, dsa_fraction=0.4)
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    co = COracle()
    co.set_keyring(kr)
    co.set_quorum(q)
    issuers = [cl.replicas[0].key_id, cl.replicas[1].key_id, 0x1122334455667788]
    n_tag = n_skip = 0
    for i, s in enumerate(H.plausible_unsigned_streams(1500, seed=99, issuers=issuers)):
        r = col.collective_verify(kr, b"payload", SignaturePacket(1, 0, False, s or None, None), q)
        tr, nv, e = co.trace_item(b"payload", s)
        assert tr == r.statuses and nv == len(r.verified) == 0 and e != 0, (i, tr[:10], r.statuses[:10])
        n_tag += sum(1 for x in tr if x == 6)
        n_skip += sum(1 for x in tr if x == 1)
    assert n_tag > 300 and n_skip > 150


def test_collective_semantics_on_mutations():
    cl = cb.make_cluster(4)
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    c = cb.make_write_corpus(cl, 60, mutation_rates=RATES)
    for i in range(c.n_items):
        ss = SignaturePacket(1, 0, False, c.ss_data(i), None)
        r = col.collective_verify(kr, c.tbss(i), ss, q)
        m = int(c.mutation[i])
        if m == cb.MUT_ONE_SHORT:
            assert r.err == col.ErrInsufficientNumberOfSignatures and not ss.Completed
        if m in (cb.MUT_NONE, cb.MUT_UNKNOWN_ISSUER, cb.MUT_DUP_SIGNER):
            assert r.err is None and ss.Completed                 # Verify mutates ss.Completed (crypto_pgp.go:494)
        if r.err is None:
            assert len(r.verified) == cl.suff                     # early exit at the first sufficient prefix
        # Combine counts CLAIMED issuers only (crypto_pgp.go:506-515)
        acc = SignaturePacket()
        assert col.collective_combine(kr, acc, ss, q) == q.is_sufficient(col.signers(kr, ss))
        assert acc.Type == 1 and acc.Data == ss.Data
    # duplicate signer packets satisfy IsSufficient
    one = cb.make_write_corpus(cl, 1, mutation_rates={})
    first = one.ss_data(0)[:287]
    ss = SignaturePacket(1, 0, False, first * 3, None)
    assert col.collective_verify(kr, one.tbss(0), ss, q).err is None


def test_max_timestamped_value():
    q = H.clique_quorum(cb.make_cluster(4))   # threshold 3 under AUTH
    ids = q.qcs[0].nodes
    assert col.max_timestamped_value([(ids[0], 5, b"v"), (ids[1], 5, b"v"), (ids[2], 5, b"v")], q) == (b"v", 5)
    assert col.max_timestamped_value([(ids[0], 5, b"v"), (ids[1], 5, b"v"), (ids[2], 4, b"v")], q) is None
    # only the LARGEST t is considered
    assert col.max_timestamped_value([(ids[0], 4, b"a"), (ids[1], 4, b"a"), (ids[2], 4, b"a"), (ids[3], 9, b"b")], q) is None
    assert col.max_timestamped_value([], q) is None
