"""-m gpu: the vote collector and the server-side verification site (protocol/client.go:125-170,
protocol/server.go:286-302) through the host mirror + GPU, against the oracle.  Scenarios follow the reference's
own protocol tests (protocol/rw_test.go, mal_test.go: honest quorum, failing peers, colluding signers)."""
import numpy as np
import pytest

from corpus import build as cb
from oracle import collective as col
from oracle import packet as opk
from oracle import wotqs as W
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _world(n):
    """Cluster + the client's view of the trust graph (scripts/setup.sh shape), in oracle and host form."""
    from bftkv_amd import host
    cl = cb.make_cluster(n)
    members = [r.key_id for r in cl.replicas]
    # clique; the client trusts (certifies) the members that did NOT certify it, as scripts/setup.sh:33-43 does --
    # otherwise client + certifiers would form a second clique
    nodes = [(m, [x for x in members if x != m] + ([] if m in cl.client.certifiers else [cl.client.key_id])) for m in members]
    nodes.append((cl.client.key_id, cl.client.certifiers))                                   # quorum certificate
    og, hg = W.Graph(), host.Graph()
    og.add_nodes(nodes)
    hg.AddNodes(nodes)
    return cl, og, hg, host


def _oracle_collect(kr, qa, tbss, replies):
    ss = opk.SignaturePacket()
    failure, consumed = [], 0
    for r in replies:
        consumed += 1
        stop, failed = False, r.Err != 0
        if not failed and r.Data:
            try:
                s = opk.parse_signature(r.Data)
            except opk.PacketError:
                s = None
            if s is None:
                failed = True
            else:
                stop = col.collective_combine(kr, ss, s, qa)
        if failed:
            failure.append(r.Peer)
            stop = qa.reject(failure)
        if stop:
            break
    res = col.collective_verify(kr, tbss, ss, qa)
    return ss.Data or b"", consumed, res.err


@pytest.mark.parametrize("n", [4, 10])
def test_collect_signatures_scenarios(gpu_ctx, n):
    cl, og, hg, host = _world(n)
    og.set_self([cl.client.key_id])
    hg.SetSelfNodes([cl.client.key_id])
    oqa = W.Wot(og).choose_quorum(W.AUTH | W.PEER)            # collectSignatures (client.go:141)
    hqa = host.wotqs.New(hg).ChooseQuorum(host.AUTH | host.PEER)
    assert hqa.qcs() == [(q.f, q.min, q.threshold, q.suff, q.nodes) for q in oqa.qcs] and len(oqa.qcs) == 1 and oqa.qcs[0].suff > 0
    kr = H.oracle_keyring(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    rng = np.random.default_rng(n)
    f, suff = oqa.qcs[0].f, oqa.qcs[0].suff
    tbss_l, replies_l = [], []
    for w in range(40):
        tbs = cb.serialize_tbs(b"key%03d" % w, rng.bytes(32), w + 1)
        tbss = tbs + cb.sigpkt(cb.detach_sign(cl.client, tbs), cl.client.entity)
        other = tbs[:-1] + bytes([tbs[-1] ^ 1])                 # a conflicting <x,v,t'> colluders would sign
        order = [int(i) for i in rng.permutation(n)]
        reps = []
        scenario = w % 10
        for pos, i in enumerate(order):
            r = cl.replicas[i]
            good = opk.serialize_signature(opk.SignaturePacket(1, 0, False, cb.detach_sign(r, tbss), r.entity))
            rep = host.Reply(Peer=r.key_id, Data=good)
            if scenario == 1 and pos < f:                       # f peers fail: still enough
                rep = host.Reply(Peer=r.key_id, Err=1)
            elif scenario == 2 and pos <= f:                    # f+1 failures up front: Reject stops the multicast
                rep = host.Reply(Peer=r.key_id, Err=1)
            elif scenario == 3 and pos % 2 == 0:                # colluders sign a conflicting value (mal_test.go)
                bad = cb.detach_sign(r, other + tbss[len(tbs):])
                rep = host.Reply(Peer=r.key_id, Data=opk.serialize_signature(opk.SignaturePacket(1, 0, False, bad, r.entity)))
            elif scenario == 4 and pos == 1:                    # garbage instead of a signature packet
                rep = host.Reply(Peer=r.key_id, Data=rng.bytes(9))
            elif scenario == 5 and pos == 0:                    # empty reply body: no error, nothing combined
                rep = host.Reply(Peer=r.key_id, Data=None)
            elif scenario == 6 and pos == 2:                    # a different SignaturePacket.Type is refused by Combine
                rep = host.Reply(Peer=r.key_id, Data=opk.serialize_signature(opk.SignaturePacket(2, 0, False, cb.detach_sign(r, tbss), None)))
            elif scenario == 7:                                 # every reply is the same peer's signature
                r0 = cl.replicas[order[0]]
                rep = host.Reply(Peer=r0.key_id, Data=opk.serialize_signature(opk.SignaturePacket(1, 0, False, cb.detach_sign(r0, tbss), None)))
            elif scenario == 8 and pos >= 1:                    # nobody else answers
                break
            elif scenario == 9 and pos == 0:                    # a signature with partial body lengths: verified by the kernels,
                sig = cb.detach_sign(r, tbss)                   # not followed by the host's INCREMENTAL Signers walk => fenced
                rep = host.Reply(Peer=r.key_id, Data=opk.serialize_signature(opk.SignaturePacket(
                    1, 0, False, H.partial_frame(2, sig[3:] if sig[1] >= 192 else sig[2:], rng), r.entity)))
            reps.append(rep)
        tbss_l.append(tbss)
        replies_l.append(reps)
    data, consumed, err = host.Client(gpu_ctx).collect_signatures(hqa, tbss_l, replies_l)
    oks = 0
    for w in range(len(tbss_l)):
        if w % 10 == 9:
            assert err[w] == 0xFC, (w, err[w])                  # BFTKV_HOST_ERR_FENCED: the caller takes the reference path
            continue
        wd, wc, we = _oracle_collect(kr, oqa, tbss_l[w], replies_l[w])
        assert data[w] == wd and consumed[w] == wc, (w, w % 10, consumed[w], wc)
        assert (err[w] == 0) == (we is None), (w, w % 10, err[w], we)
        oks += we is None
    assert 0 < oks < len(tbss_l)
    # honest rounds stop after exactly `suff` replies; duplicates of one signer also reach sufficiency
    assert consumed[0] == suff and err[0] == 0 and err[7] == 0 and err[2] == 2 and err[8] == 2


def test_server_write_verify(gpu_ctx):
    cl, og, hg, host = _world(10)
    me = cl.replicas[3].key_id
    og.set_self([me])
    hg.SetSelfNodes([me])
    oq = W.Wot(og).choose_quorum(W.AUTH)                        # Server.write (server.go:300)
    hq = host.wotqs.New(hg).ChooseQuorum(host.AUTH)
    kr = H.oracle_keyring(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    c = cb.make_write_corpus(cl, 40, keep_requests=True,
                             mutation_rates={cb.MUT_BAD_MPI: 0.15, cb.MUT_ONE_SHORT: 0.2, cb.MUT_UNKNOWN_ISSUER: 0.1, cb.MUT_DUP_SIGNER: 0.1})
    reqs = list(c.requests)
    reqs[5] = reqs[5][:len(reqs[5]) - 7]                         # short read inside ss
    reqs[6] = opk.serialize(b"k", b"v", 3)                       # no sig, no ss  -> malformed request
    x, v, t, sig, ss, _ = opk.parse(reqs[7])
    reqs[7] = opk.serialize(x, v, t, sig)                        # ss absent
    reqs[8] = opk.serialize(x, v, t, None, ss)                   # nil sig
    reqs[9] = b""
    err = host.Server(gpu_ctx).write_verify(hq, reqs)
    for i, r in enumerate(reqs):
        try:
            x, v, t, sig, ss, _ = opk.parse(r)
            want = 0xFF if (sig is None or ss is None) else None
        except opk.PacketError:
            want = 0xFF
        if want is None:
            res = col.collective_verify(kr, opk.tbss(r), ss, oq)
            want = 0 if res.err is None else 2
        assert err[i] == want, (i, err[i], want)
    assert set(err) == {0, 2, 0xFF}
    assert host.packet.TBSS(reqs[0]) == c.tbss(0)


def test_server_sign_verify(gpu_ctx):
    """Server.sign's checks (server.go:189-214): VerifyWithCertificate with the issuer taken from the request's own
    certificate, then IsThreshold over the certificate's certifiers under ChooseQuorum(AUTH|CERT)."""
    from oracle import openpgp as pgp
    cl, og, hg, host = _world(10)
    me = cl.replicas[2].key_id
    og.set_self([me])
    hg.SetSelfNodes([me])
    oq = W.Wot(og).choose_quorum(W.AUTH | W.CERT)
    hq = host.wotqs.New(hg).ChooseQuorum(host.AUTH | host.CERT)
    assert oq.qcs[0].threshold == 4
    kr = H.oracle_keyring(cl)                                   # the server's keyring does NOT hold the client
    gpu_ctx.keyring_set(H.abi_keys(kr))
    rng = np.random.default_rng(12)
    # a second client certified by too few members, and an outsider nobody certified
    weak = cb.make_keypair(cb.PK_RSA, cb.load_keys("rsa2048", 80)[79], "u02 <u02@bftkv.example>")
    from corpus.keys import DRBG
    cb.build_entity(weak, [r for r in cl.replicas if r.algo == cb.PK_RSA][:3], DRBG("weak"))
    reqs, certs = [], []
    for i in range(24):
        signer = [cl.client, weak, cl.outsiders[0]][i % 3]
        x, v, t = b"key%03d" % i, rng.bytes(20), i + 1
        tbs = cb.serialize_tbs(x, v, t)
        sigdata = cb.detach_sign(signer, tbs)
        cert = signer.entity
        k = i % 8
        if k == 3: sigdata = cb.detach_sign(signer, tbs + b"x")                       # signature over other bytes
        if k == 4: cert = cl.replicas[0].entity                                        # certificate of somebody else
        if k == 5: cert = None                                                         # no certificate
        if k == 6: sigdata = sigdata + cb.detach_sign(cl.replicas[1], tbs)             # second packet by a key outside the certificate
        reqs.append(opk.serialize(x, v, t, opk.SignaturePacket(1, 0, False, sigdata, cert)))
        certs.append(cert)
    reqs.append(opk.serialize(b"k", b"v", 9))                                          # no signature
    reqs.append(reqs[0][:-3])
    err = host.Server(gpu_ctx).sign_verify(hq, reqs)
    seen = set()
    for i, r in enumerate(reqs):
        try:
            x, v, t, sig, ss, _ = opk.parse(r)
            want = 0xFF if sig is None else None
        except opk.PacketError:
            want = 0xFF
        if want is None:
            ents = pgp.read_entities(sig.Cert or b"")
            if not ents:
                want = 0xFE
            elif col.signature_verify_with_certificate(opk.tbs(r), sig, ents[0]) is not None:
                want = 1
            else:
                nodes = [c for c in ents[0].certifiers if kr.get_cert_by_id(c) is not None]
                want = 0 if oq.is_threshold(nodes) else 0xFD
        assert err[i] == want, (i, err[i], want)
        seen.add(int(want))
    assert seen == {0, 1, 0xFD, 0xFE, 0xFF}
    # the node keyring is unaffected by certificate entities: collective verification still sees 10 replicas only
    c = cb.make_write_corpus(cl, 6, mutation_rates={})
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    e2, _, _ = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    assert (e2 == 0).all()
    client_sig = cb.detach_sign(cl.client, b"abc")
    assert gpu_ctx.signature_verify(np.frombuffer(b"abc", dtype=np.uint8), np.array([0, 3], dtype=np.uint64),
                                    np.frombuffer(client_sig, dtype=np.uint8), np.array([0, len(client_sig)], dtype=np.uint64))[0] == 1
    gpu_ctx.quorum_destroy(qh)


def test_dsa_certificates_from_requests_take_bounded_table_slots(gpu_ctx):
    """Server.sign requests carry their principal's certificate (server.go:199-207).  A DSA certificate key needs fixed-base
    window tables -- 637 MB and a table build at the 16-bit width -- so certificate-only DSA keys share a bounded, recycled set
    of slots (8 at that width): 14 fresh DSA principals in one batch verify for the 8 that got slots, the others come back
    FENCED (the reference path decides), and verify when they are sent again; the node keys' tables are never rebuilt and
    the window width does not move."""
    from corpus.keys import DRBG
    from oracle import openpgp as pgp
    cl, og, hg, host = _world(10)
    me = cl.replicas[2].key_id
    og.set_self([me])
    hg.SetSelfNodes([me])
    oq = W.Wot(og).choose_quorum(W.AUTH | W.CERT)
    hq = host.wotqs.New(hg).ChooseQuorum(host.AUTH | host.CERT)
    kr = H.oracle_keyring(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    mats = cb.load_keys("dsa2048", 34)[20:34]
    certifiers = [r for r in cl.replicas if r.algo == cb.PK_RSA][:5]
    users = []
    for i, m in enumerate(mats):
        u = cb.make_keypair(cb.PK_DSA, m, "d%02d <d%02d@bftkv.example>" % (i, i))
        cb.build_entity(u, certifiers, DRBG("dsa-user-%d" % i))
        users.append(u)
    srng = DRBG("dsa-user-sigs")

    def request(u, i, good=True):
        x, v, t = b"key%03d" % i, b"value", i + 1
        tbs = cb.serialize_tbs(x, v, t)
        return opk.serialize(x, v, t, opk.SignaturePacket(1, 0, False, cb.detach_sign(u, tbs if good else tbs + b"!", srng), u.entity))
    reqs = [request(u, i, good=(i % 5 != 4)) for i, u in enumerate(users)]
    srv = host.Server(gpu_ctx)
    err = list(srv.sign_verify(hq, reqs))
    want = []
    for r in reqs:
        x, v, t, sig, ss, _ = opk.parse(r)
        ent = pgp.read_entities(sig.Cert)[0]
        if col.signature_verify_with_certificate(opk.tbs(r), sig, ent) is not None:
            want.append(1)
        else:
            nodes = [c for c in ent.certifiers if kr.get_cert_by_id(c) is not None]
            want.append(0 if oq.is_threshold(nodes) else 0xFD)
    fenced = [i for i, e in enumerate(err) if e == 0xFC]
    assert len(fenced) == len(users) - 8, err                     # 8 slots at the 16-bit width
    assert all(err[i] == want[i] for i in range(len(users)) if i not in fenced), (err, want)
    # the fenced ones again, alone: slots of certificate keys that are not in this batch are recycled
    again = list(srv.sign_verify(hq, [reqs[i] for i in fenced]))
    assert again == [want[i] for i in fenced], (again, [want[i] for i in fenced])
    assert 0 in want and 1 in want
    # and the whole batch once more: still 8 slots, whichever 8 hold them
    err3 = list(srv.sign_verify(hq, reqs))
    assert sum(e == 0xFC for e in err3) == len(users) - 8
    assert all(e == w for e, w in zip(err3, want) if e != 0xFC)


def test_equivocation_signers(gpu_ctx):
    """Client.revoke's tally (client.go:304-353): signers common to two different values at one timestamp."""
    cl, og, hg, host = _world(10)
    kr = H.oracle_keyring(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    tb_a, tb_b = b"value-a", b"value-b"
    sa = [cb.detach_sign(r, tb_a) for r in cl.replicas]
    sb = [cb.detach_sign(r, tb_b) for r in cl.replicas]
    values = [(0, b"".join(sa[:7])), (0, b"".join(sa[2:8])), (1, b"".join(sb[5:10])), (1, b"".join(sb[6:9]) + cb.detach_sign(cl.outsiders[0], tb_b)),
              (2, b"")]
    got = host.Client(gpu_ctx).equivocation_signers(values)
    by_group = {}
    for g, data in values:
        by_group.setdefault(g, set()).update(col.signers(kr, opk.SignaturePacket(1, 0, False, data or None, None)))
    want = sorted(i for i in set().union(*by_group.values()) if sum(i in s for s in by_group.values()) >= 2)
    assert got == want == sorted(r.key_id for r in cl.replicas[5:8])


def test_entity_verification_and_quorum_certificate(gpu_ctx):
    """what openpgp.ReadEntity verifies (uid self-signatures, subkey bindings) and the paper's
    CheckQuorumCert (certifications verified, not just counted), on the GPU, against the oracle and GnuPG's verdict."""
    import json
    import os
    from corpus.keys import DRBG
    from oracle import openpgp as pgp
    cl, og, hg, host = _world(10)
    me = cl.replicas[1].key_id
    og.set_self([me]); hg.SetSelfNodes([me])
    oq = W.Wot(og).choose_quorum(W.AUTH | W.CERT)
    hq = host.wotqs.New(hg).ChooseQuorum(host.AUTH | host.CERT)
    kr = H.oracle_keyring(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    # an entity with an (encryption) subkey
    sub_owner = cb.make_keypair(cb.PK_RSA, cb.load_keys("rsa2048", 82)[80], "s01 <s01@bftkv.example>")
    sub = cb.make_keypair(cb.PK_RSA, cb.load_keys("rsa2048", 82)[81], "")
    cb.build_entity(sub_owner, [], DRBG("sub"), subkey=sub)

    def flip(blob, marker_from_end):
        b = bytearray(blob); b[len(b) - marker_from_end] ^= 0x20; return bytes(b)
    client = cl.client.entity
    selfsig_end = client.index(b"\xc2", client.index(cl.client.name.encode()))    # first signature packet after the uid = self-signature
    bad_self = bytearray(client); bad_self[selfsig_end + 200] ^= 1; bad_self = bytes(bad_self)
    bad_cert = flip(client, 30)                                                     # last certification corrupted
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gpg_vectors.json")))
    gpg_ring = bytes.fromhex(vec["A_pubring"])                                      # entities gpg itself made (SHA-512 self-signatures)
    blobs = {"client": client, "bad_self": bad_self, "bad_cert": bad_cert, "replicas": b"".join(r.entity for r in cl.replicas[:3]),
             "subkey": sub_owner.entity, "bad_binding": flip(sub_owner.entity, 25), "gpg": gpg_ring, "empty": b""}
    for name, blob in blobs.items():
        want = [e["valid"] for e in pgp.entity_checks(blob)]
        got = host.certs_verify(gpu_ctx, blob)
        assert got == want, (name, got, want)
    assert host.certs_verify(gpu_ctx, client) == [True] and host.certs_verify(gpu_ctx, bad_self) == [False]
    assert host.certs_verify(gpu_ctx, sub_owner.entity) == [True] and host.certs_verify(gpu_ctx, blobs["bad_binding"]) == [False]
    assert host.certs_verify(gpu_ctx, gpg_ring) == [True, True]
    # CheckQuorumCert: the code path counts ids (4 of 4 => threshold), the paper's check verifies them
    for blob, n_ok in ((client, 4), (bad_cert, 3)):
        ent = pgp.entity_checks(blob)[0]
        want_ids = pgp.verified_certifiers(ent, kr.get_keyring())
        ok, ids = host.quorum_cert_verify(gpu_ctx, hq, blob)
        assert ids == want_ids and len(ids) == n_ok and ok == oq.is_threshold(want_ids)
    assert host.quorum_cert_verify(gpu_ctx, hq, client)[0] and not host.quorum_cert_verify(gpu_ctx, hq, bad_cert)[0]
    # Server.sign: a request whose certificate has a forged self-signature has no issuer
    tbs = cb.serialize_tbs(b"k", b"v", 1)
    sig = cb.detach_sign(cl.client, tbs)
    reqs = [opk.serialize(b"k", b"v", 1, opk.SignaturePacket(1, 0, False, sig, c)) for c in (client, bad_self, bad_cert)]
    err = host.Server(gpu_ctx).sign_verify(hq, reqs)
    assert list(err) == [0, 0xFE, 0]      # bad_cert still passes the CODE's check: its certifier ids are only counted


@pytest.mark.parametrize("n_lanes", [1, 4])
def test_micro_batcher_concurrent_single_calls(gpu_ctx, n_lanes):
    """One CollectiveSignature.Verify per call from many threads (the reference's goroutine-per-request shape,
    transport/http/http.go:85,143) -> device batches on n_lanes forked contexts at once; every caller gets the oracle's
    answer."""
    import threading
    from bftkv_amd import Batcher
    cl = cb.make_cluster(4)
    kr = H.oracle_keyring(cl, include_client=True)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    c = cb.make_write_corpus(cl, 96, mutation_rates={cb.MUT_ONE_SHORT: 0.3, cb.MUT_BAD_MPI: 0.2})
    want = [0 if H.oracle_collective(kr, q, c, i).err is None else 2 for i in range(c.n_items)]
    b = Batcher(gpu_ctx, max_items=32, max_wait_us=20000, n_lanes=n_lanes)
    got = [None] * c.n_items
    sig_got = [None] * 16
    msg_got = [None] * 24
    from corpus.keys import DRBG
    mrng = DRBG("batcher-msgs")
    msgs = [cb.signed_message(cl.replicas[i % 4] if i % 5 else cl.outsiders[0], b"body%d" % i, b"nonce-%010d" % i, mrng, "go" if i % 2 else "definite")
            for i in range(24)]
    msgs[3] = msgs[3][:-2] + bytes([msgs[3][-2] ^ 1]) + msgs[3][-1:]

    def msg_worker(i):
        msg_got[i] = b.message_verify(msgs[i])

    def worker(i):
        got[i] = b.collective_verify(qh, c.tbss(i), c.ss_data(i))

    def sig_worker(i):
        tbs = b"msg%d" % i
        s = cb.detach_sign(cl.client, tbs if i % 2 == 0 else tbs + b"!")
        sig_got[i] = b.signature_verify(tbs, s, cert_key_id=cl.client.key_id if i % 4 < 2 else None)
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(c.n_items)] + [threading.Thread(target=sig_worker, args=(i,)) for i in range(16)]
    ths += [threading.Thread(target=msg_worker, args=(i,)) for i in range(24)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    st = b.stats()
    # the framings no writer produces (partial / indeterminate lengths, bodies beyond bufio's buffer ...) through the batcher's
    # staged route: error byte and fence flag of every call are those of the batched entry point on the same inputs
    tbs_x, ss_x = H.exotic_framing_streams(cl, 28, seed=5)
    tbx, tox = H.cat(tbs_x)
    sbx, sox = H.cat(ss_x)
    ex_err, _, _ = gpu_ctx.collective_verify(qh, tbx, tox, sbx, sox)
    ex_fenced = gpu_ctx.last_fenced.copy()
    for i in range(len(ss_x)):
        rc, e1, f1 = b.collective_verify(qh, tbs_x[i], ss_x[i], raw=True)
        assert rc == 0 and e1 == ex_err[i] and f1 == ex_fenced[i], (i, rc, e1, ex_err[i], f1, ex_fenced[i])
    assert 3 < ex_fenced.sum() < 25 and (ex_err == 0).any()
    b.close()
    assert got == want and 0 < sum(g == 0 for g in got) < len(got)
    assert sig_got == [0 if i % 2 == 0 else 1 for i in range(16)]
    # (how the calls fall into batches depends on thread timing: a free lane is taken at once; only the bounds are asserted)
    assert st["calls"] == c.n_items + 16 + 24 and st["lanes"] == n_lanes and st["max_batch"] <= 32
    import base64
    for i, (mst, signer, peer, plain, fname) in enumerate(msg_got):
        who = cl.replicas[i % 4] if i % 5 else cl.outsiders[0]
        assert mst == (4 if i % 5 == 0 else (1 if i == 3 else 0)), (i, mst)          # outsider: unverified; tampered: signature error
        assert plain == b"body%d" % i and base64.standard_b64decode(fname) == b"nonce-%010d" % i
        assert signer == who.key_id and peer == (who.key_id if i % 5 else 0)
    gpu_ctx.quorum_destroy(qh)


def test_forked_contexts_verify_over_the_roots_key_table(gpu_ctx):
    """bftkv_gpu_ctx_fork: forks answer like their root, concurrently, follow the root's keyring and quorum changes, and
    refuse to change either themselves."""
    import threading
    from bftkv_amd._native import NativeError
    cl = cb.make_cluster(7, dsa_fraction=0.3)
    kr = H.oracle_keyring(cl, include_client=True)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    c = cb.make_write_corpus(cl, 48, mutation_rates={cb.MUT_ONE_SHORT: 0.3, cb.MUT_BAD_MPI: 0.2, cb.MUT_UNKNOWN_ISSUER: 0.1})
    want = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    for i in range(c.n_items):
        r = H.oracle_collective(kr, q, c, i)
        assert (want[0][i] == 0) == (r.err is None) and want[1][i] == len(r.verified)
    forks = [gpu_ctx.fork() for _ in range(3)]
    out = [None] * len(forks)

    def run(k):
        res = []
        for _ in range(4):
            res.append(forks[k].collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off))
        out[k] = res
    ths = [threading.Thread(target=run, args=(k,)) for k in range(len(forks))]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    for res in out:
        for e, nv, vd in res:
            assert (e == want[0]).all() and (nv == want[1]).all() and (vd == want[2]).all()
    with pytest.raises(NativeError):
        forks[0].keyring_set(H.abi_keys(kr))
    with pytest.raises(NativeError):
        forks[0].quorum_create(H.abi_qcs(q))
    # the root drops two replicas from its keyring: the forks see it at their next call
    drop = {cl.replicas[0].key_id, cl.replicas[1].key_id}
    kr2 = H.oracle_keyring(cl, include_client=True)
    keys2 = [k for k in H.abi_keys(kr2) if k["entity_id"] not in drop]
    gpu_ctx.keyring_set(keys2)
    want2 = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    assert (want2[1] <= want[1]).all() and (want2[1] < want[1]).any()
    got2 = forks[1].collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    assert (got2[0] == want2[0]).all() and (got2[1] == want2[1]).all()
    # a quorum created on the root afterwards is usable on a fork; a destroyed one is not
    qh2 = gpu_ctx.quorum_create(H.abi_qcs(q))
    got3 = forks[2].collective_verify(qh2, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    assert (got3[0] == want2[0]).all()
    gpu_ctx.quorum_destroy(qh2)
    with pytest.raises(NativeError):
        forks[2].collective_verify(qh2, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    for f in forks:
        f.close()
    gpu_ctx.quorum_destroy(qh)


def test_micro_batcher_midstates_and_other_hashes(gpu_ctx):
    """The batcher's callers hand over SHA-256 midstates of their payloads; a signature made with another hash sends its
    item through the payload path.  The gpg fixtures (SHA-1 / 224 / 256 / 384 / 512, RSA and DSA) and payloads of every
    length around the 64-byte block boundary, one call each, against the batched entry point."""
    import json
    import os
    import threading
    from bftkv_amd import Batcher
    from oracle import collective as col
    from oracle import openpgp as pgp
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gpg_vectors.json")))
    for ring_key, group in (("A_pubring", "A"), ("B_pubring", "B"), ("C_pubring", "C")):      # C: rsa3072 / rsa4096 beside rsa2048
        ring = pgp.read_entities(bytes.fromhex(vec[ring_key]))
        gpu_ctx.keyring_set(H.abi_keys(col.Keyring(keyring=ring)))
        tbs_l = [bytes.fromhex(v["payload"]) for v in vec[group]]
        sig_l = [bytes.fromhex(v["sig"]) for v in vec[group]]
        tbs_l += [t + b"?" for t in tbs_l]
        sig_l += sig_l
        tb, to = H.cat(tbs_l)
        sb, so = H.cat(sig_l)
        want = list(gpu_ctx.signature_verify(tb, to, sb, so))
        assert 0 in want and 1 in want
        b = Batcher(gpu_ctx, max_items=16, n_lanes=2)
        got = [None] * len(tbs_l)

        def one(i):
            got[i] = b.signature_verify(tbs_l[i], sig_l[i])
        ths = [threading.Thread(target=one, args=(i,)) for i in range(len(tbs_l))]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=120)
        b.close()
        assert got == want
    # payload lengths 0 .. 200 (every residue of the block size, zero whole blocks, several whole blocks)
    cl = cb.make_cluster(4)
    kr = H.oracle_keyring(cl, include_client=True)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    b = Batcher(gpu_ctx, max_items=64, n_lanes=2)
    try:
        for ln in list(range(0, 200, 7)) + [63, 64, 65, 127, 128, 129, 8599, 8640]:
            tbs = bytes((i * 31 + ln) & 0xFF for i in range(ln))
            sig = cb.detach_sign(cl.replicas[ln % 4], tbs)
            assert b.signature_verify(tbs, sig) == 0, ln
            assert b.signature_verify(tbs + b"x", sig) == 1, ln
        # more packet events than a staged call's arena is sized for (one event per stray byte): the call is run again through
        # the ordinary entry point and answers like it
        tbs = b"payload"
        good = cb.detach_sign(cl.replicas[0], tbs)
        for junk in (b"\x01" * 70000, good + b"\x01" * 70000, b"\xd4\x00" * 40000 + good):
            tb, to = H.cat([tbs])
            sb, so = H.cat([junk])
            want = int(gpu_ctx.signature_verify(tb, to, sb, so)[0])
            assert b.signature_verify(tbs, junk) == want
        q = H.clique_quorum(cl)
        qh = gpu_ctx.quorum_create(H.abi_qcs(q))
        ss = b"".join(cb.detach_sign(r, tbs) for r in cl.replicas)
        for stream in (ss, b"\xd4\x00" * 50000 + ss, ss[:300] + b"\x01" * 60000):
            tb, to = H.cat([tbs])
            sb, so = H.cat([stream])
            want = int(gpu_ctx.collective_verify(qh, tb, to, sb, so)[0][0])
            assert b.collective_verify(qh, tbs, stream) == want
        gpu_ctx.quorum_destroy(qh)
    finally:
        b.close()


def test_audit_plain_storage_db(gpu_ctx, tmp_path):
    """a storage/plain directory of accepted writes (files hex(x).t = request bytes) re-verified on the GPU."""
    import os
    from bftkv_amd import audit
    from corpus.keys import DRBG
    cl = cb.make_cluster(10)
    members = [r.key_id for r in cl.replicas]
    # pubring.gpg as the daemon loads it: clique members cross-certify (scripts/clique.sh), so the ring carries the graph
    rng = DRBG("ring")
    for r in cl.replicas:
        cb.build_entity(r, [o for o in cl.replicas if o is not r], rng)
    pubring = b"".join(r.entity for r in cl.replicas) + cl.client.entity
    c = cb.make_write_corpus(cl, 30, keep_requests=True, mutation_rates={cb.MUT_ONE_SHORT: 0.2, cb.MUT_BAD_MPI: 0.2})
    db = tmp_path / "db"
    db.mkdir()
    want = {}
    for i, req in enumerate(c.requests):
        x, v, t, sig, ss, _ = opk.parse(req)
        name = "%s.%d" % (x.hex(), t)
        blob = req
        status = "ok" if c.expected_valid[i] >= cl.suff else "insufficient"
        if i == 3:
            blob, status = req[:len(req) // 2], "malformed"
        if i == 4:
            name, status = "%s.%d" % (x.hex(), t + 1000), ("name-mismatch" if status == "ok" else status)
        (db / name).write_bytes(blob)
        want[name] = status
    (db / "README").write_text("not a record")
    recs = audit.audit_plain_db(gpu_ctx, str(db), pubring, members[2])
    got = {os.path.basename(r.path): r.status for r in recs}
    assert got == want and set(want.values()) >= {"ok", "insufficient", "malformed"}


def test_audit_leveldb_db(gpu_ctx, tmp_path):
    """a storage/leveldb database (key = x || t_BE, storage/leveldb/leveldb.go:30-53) -- tables, a live log,
    overwritten and deleted keys -- read without a LevelDB library and re-verified on the GPU."""
    import struct
    from bftkv_amd import audit
    from corpus.keys import DRBG
    from tests import leveldb_writer as LW
    cl = cb.make_cluster(10)
    rng = DRBG("ring")
    for r in cl.replicas:
        cb.build_entity(r, [o for o in cl.replicas if o is not r], rng)
    pubring = b"".join(r.entity for r in cl.replicas) + cl.client.entity
    c = cb.make_write_corpus(cl, 24, keep_requests=True, mutation_rates={cb.MUT_ONE_SHORT: 0.25, cb.MUT_BAD_MPI: 0.2})
    key = lambda x, t: x + struct.pack(">Q", t)
    table, log, want = [], [], {}
    for i, req in enumerate(c.requests):
        x, v, t, sig, ss, _ = opk.parse(req)
        status = "ok" if c.expected_valid[i] >= cl.suff else "insufficient"
        blob = req
        if i == 2: blob, status = req[:40], "malformed"
        if i == 5: t, status = t + 500, ("name-mismatch" if status == "ok" else status)      # stored under another timestamp
        name = "%s.%d" % (x.hex(), t)
        if i % 3 == 0:
            table.append((key(x, t), 10 + i, 1, b"superseded by the log"))                   # older version in a table ...
            log.append((100 + i, [(1, key(x, t), blob)]))                                    # ... the log holds the current one
        elif i % 3 == 1:
            table.append((key(x, t), 10 + i, 1, blob))
        else:
            log.append((100 + i, [(1, key(x, t), blob)]))
        want[name] = status
    table.append((key(b"deleted", 1), 9, 1, b"gone"))
    log.append((900, [(0, key(b"deleted", 1), b"")]))
    LW.write_db(str(tmp_path / "ldb"), [sorted(table)], log)
    recs = audit.audit_leveldb_db(gpu_ctx, str(tmp_path / "ldb"), pubring, cl.replicas[2].key_id)
    got = {r.path: r.status for r in recs}
    assert got == want and set(want.values()) >= {"ok", "insufficient", "malformed"}


def test_transport_message_signatures(gpu_ctx):
    """bftkv_gpu_message_verify == oracle.message.read_signed_message on gpg-made messages, generator-made
    ones judged by gpg, the outcome table, and a mutated batch from a mixed RSA/DSA cluster."""
    import json
    import os
    from oracle import message as om
    from oracle import openpgp as pgp
    from corpus.keys import DRBG
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gpg_messages.json")) as f:
        vec = json.load(f)

    def check(ring_blob, msgs, expect_ok=None):
        ents = pgp.read_entities(ring_blob)
        kr = col.Keyring(ents)
        gpu_ctx.keyring_set(H.abi_keys(kr))
        st, signer, peer, plains, names = gpu_ctx.message_verify(msgs)
        for i, m in enumerate(msgs):
            r = om.read_signed_message(ents, m)
            assert st[i] == r.status, (i, st[i], r.status, r.sig_status)
            if r.status not in (om.MSG_READ_ERROR, om.MSG_UNSUPPORTED):
                assert plains[i] == r.plain and names[i] == r.file_name
                assert int(signer[i]) == r.signed_by_key_id and int(peer[i]) == (r.peer or 0)
            if expect_ok is not None:
                assert (st[i] == om.MSG_OK) == expect_ok[i]
        return st

    d = vec["D"]
    st = check(bytes.fromhex(vec["D_pubring"]), [bytes.fromhex(x["msg"]) for x in d] + [bytes.fromhex(x["tampered"]) for x in d],
               [x["gpg_good"] for x in d] + [x["gpg_tampered_good"] for x in d])
    assert (st[:len(d)] == 0).all() and (st[len(d):] != 0).all()
    e = vec["E"]
    check(bytes.fromhex(vec["E_pubring"]), [bytes.fromhex(x["msg"]) for x in e], [x["gpg_good"] for x in e])

    # outcome table + mutated batch
    cl = cb.make_cluster(16, dsa_fraction=0.25, n_outsiders=2)
    rng = DRBG("msgbatch")
    ring_blob = b"".join(r.entity for r in cl.replicas)
    nrng = np.random.default_rng(8)
    msgs = []
    for i in range(600):
        kp = cl.replicas[i % 16] if i % 23 else cl.outsiders[i % 2]
        body = nrng.integers(0, 256, size=int(nrng.integers(0, 3000)), dtype=np.uint8).tobytes()
        m = bytearray(cb.signed_message(kp, body, bytes(nrng.integers(0, 256, size=16, dtype=np.uint8)), rng, "go" if i % 3 else "definite"))
        k = i % 11
        if k == 1:
            m[int(nrng.integers(0, len(m)))] ^= 1 << int(nrng.integers(0, 8))      # anywhere: header, lengths, body, signature
        elif k == 2:
            m = m[:int(nrng.integers(0, len(m)))]                                  # truncated
        elif k == 3:
            m[-5] ^= 0x40                                                          # signature value
        msgs.append(bytes(m))
    kp = cl.replicas[1]
    sig = cb.detach_sign(kp, b"request", rng)
    ops = om.one_pass_packet(0, 8, kp.algo, kp.key_id)
    lit = om.literal_packet(b"f", b"request")
    unk = pgp.new_format_header(60, 3) + b"abc"
    dsa_kp = next(r for r in cl.replicas if r.algo == cb.PK_DSA)
    dsa_sig = cb.detach_sign(dsa_kp, b"request", rng)
    msgs += [lit, om.one_pass_packet(0, 8, kp.algo, cl.replicas[2].key_id) + lit + sig, om.one_pass_packet(0, 10, kp.algo, kp.key_id) + lit + sig,
             ops + lit + lit, ops + lit, unk + ops + unk + lit + unk + sig, om.one_pass_packet(0, 8, kp.algo, kp.key_id, is_last=False) + lit + sig,
             om.one_pass_packet(0x10, 8, kp.algo, kp.key_id) + lit + sig, ops, b"", ops + lit[:5], b"\x00" + ops + lit + sig,
             om.one_pass_packet(1, 8, kp.algo, kp.key_id) + lit + sig, pgp.new_format_header(8, 2) + b"\x00\x00" + ops + lit + sig,
             ops + ops + lit + sig, ops + lit + sig + b"\x00garbage", ops + om.literal_packet(b"f", b"request", partial=[1, 0, 2]) + sig,
             om.one_pass_packet(0, 8, dsa_kp.algo, dsa_kp.key_id) + lit + dsa_sig,
             om.one_pass_packet(0, 10, dsa_kp.algo, dsa_kp.key_id) + lit + dsa_sig,      # body hashed with SHA-512: tag mismatch
             om.one_pass_packet(0, 8, kp.algo, dsa_kp.key_id) + lit + sig]               # key and signature algorithms differ
    # signature packets that NAME SHA-512 while the one-pass packet (and the signer) used SHA-256: dsa.Verify never looks
    # at the packet's hash id, rsa.VerifyPKCS1v15 refuses the 32-byte digest
    import hashlib
    for who in (dsa_kp, kp):
        pre = bytearray(cb.sig_prefix(0x00, who.algo, cb._hashed_area(who.key_id)))
        pre[3] = 10
        dg = hashlib.sha256(b"request" + cb.hash_suffix(bytes(pre))).digest()
        msgs.append(om.one_pass_packet(0, 8, who.algo, who.key_id) + lit + cb.make_sig_packet(who, bytes(pre), dg, rng))
    n_named = len(msgs)
    st = check(ring_blob, msgs)
    assert st[n_named - 2] == om.MSG_OK and st[n_named - 1] == om.MSG_SIGNATURE_ERROR
    counts = np.bincount(st, minlength=6)
    assert counts[om.MSG_OK] > 300 and counts[om.MSG_SIGNATURE_ERROR] > 30 and counts[om.MSG_READ_ERROR] > 10
    assert counts[om.MSG_UNVERIFIED] > 10 and counts[om.MSG_NOT_SIGNED] >= 1 and counts[om.MSG_UNSUPPORTED] >= 3


def test_write_vote_folds_over_gpu_tallies(gpu_ctx):
    """SURVEY a24, Client.Write / writeWithTimestamp (protocol/client.go:62-123): every Multicast round is a fold over the
    replies -- success -> actives, IsThreshold(actives) ends it; failure -> failure, Reject(failure) ends it.  The host
    mirror's fold (bftkv_host_vote_fold) must stop where the reference's callback returns true, and the predicate values it
    stops on must be the ones the GPU tally (bftkv_gpu_quorum_tally, k_tally_ids) and the oracle compute for those lists."""
    cl, og, hg, host = _world(10)
    og.set_self([cl.client.key_id])
    hg.SetSelfNodes([cl.client.key_id])
    kr = H.oracle_keyring(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    for rw in (W.AUTH | W.PEER, W.WRITE | W.AUTH, W.READ | W.AUTH):
        oq = W.Wot(og).choose_quorum(rw)
        hq = host.wotqs.New(hg).ChooseQuorum(rw)
        assert hq.qcs() == [(c.f, c.min, c.threshold, c.suff, c.nodes) for c in oq.qcs]
        qh = gpu_ctx.quorum_create(hq.qcs())
        peers = [r.key_id for r in cl.replicas] + [cl.client.key_id, 0xDEAD]
        rng = np.random.default_rng(rw)
        rounds = []
        for _ in range(60):
            order = rng.permutation(len(peers))[:int(rng.integers(1, len(peers) + 1))]
            p_fail = float(rng.choice([0.0, 0.1, 0.5, 0.9]))
            rounds.append([(peers[int(i)], bool(rng.random() >= p_fail)) for i in order])
        consumed, thr = host.Client.vote_fold(hq, rounds)
        ids, off = [], [0]
        want_consumed, want_thr = [], []
        for rd in rounds:
            actives, failure, n = [], [], 0
            for peer, ok in rd:                                  # the reference's callback, reply by reply
                n += 1
                if ok:
                    actives.append(peer)
                    if oq.is_threshold(actives):
                        break
                else:
                    failure.append(peer)
                    if oq.reject(failure):
                        break
            want_consumed.append(n); want_thr.append(oq.is_threshold(actives))
            ids += actives; off.append(len(ids)); ids += failure; off.append(len(ids))
        assert list(consumed) == want_consumed and [bool(t) for t in thr] == want_thr
        v = gpu_ctx.quorum_tally(qh, np.array(ids + [0], dtype=np.uint64)[:len(ids)], np.array(off, dtype=np.uint64))
        for k, rd in enumerate(rounds):
            assert bool(v[2 * k] & 2) == want_thr[k]             # IsThreshold(actives) on the device
            fails = ids[off[2 * k + 1]:off[2 * k + 2]]
            assert bool(v[2 * k + 1] & 8) == oq.reject(fails)    # Reject(failure) on the device
        assert any(want_thr) and not all(want_thr)
        gpu_ctx.quorum_destroy(qh)


def test_read_answers_after_gpu_reply_verification(gpu_ctx):
    """SURVEY a25, BASELINE configs[2] in small: read replies <x,v,t,sig,ss> of a mixed RSA / DSA clique verified on the GPU,
    the failures dropped, then maxTimestampedValue per variable through the host mirror (protocol/client.go:181-205) --
    against the oracle's verdict per reply and its restatement of the read tally."""
    from bftkv_amd import host
    cl = cb.make_cluster(8, dsa_fraction=0.5)
    kr = H.oracle_keyring(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    rc = cb.make_read_corpus(cl, 40, seed=3, p_stale=0.3, p_conflict_var=0.2, p_silent=0.1,
                             mutation_rates={cb.MUT_BAD_MPI: 0.15, cb.MUT_ONE_SHORT: 0.2, cb.MUT_UNKNOWN_ISSUER: 0.1})
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    err, nver, _ = gpu_ctx.collective_verify(qh, rc.tbss_blob, rc.tbss_off, rc.ss_blob, rc.ss_off)
    ns = len(rc.storage_ids)
    f = (ns - 1) // 3
    qc = (f, 3 * f + 1, f + 1, f + (ns - f) // 2 + 1, rc.storage_ids)          # the storage nodes under the READ rule (wotqs.go:36-70)
    hq = host.Quorum.from_qcs([qc])
    oq = W.WotQ([W.QC(nodes=list(rc.storage_ids), f=qc[0], min=qc[1], threshold=qc[2], suff=qc[3])])
    reply_t = rc.write_t[rc.reply_write]
    reads_gpu, reads_oracle = [], []
    for j in range(rc.n_vars):
        rows = np.nonzero(rc.reply_var == j)[0]
        g, o = [], []
        for r in rows:
            w = int(rc.reply_write[r])
            res = H.oracle_collective(kr, q, rc.writes, w)        # a reply is a byte-identical copy of stored packet w
            assert (err[r] == 0) == (res.err is None) and nver[r] == len(res.verified), (j, r)
            row = (int(rc.reply_peer[r]), int(reply_t[r]), rc.write_value[w])
            if err[r] == 0: g.append(row)
            if res.err is None: o.append(row)
        reads_gpu.append(g); reads_oracle.append(o)
    got = host.Client.max_timestamped_value(hq, reads_gpu)
    want = [col.max_timestamped_value(rs, oq) for rs in reads_oracle]
    assert [None if a is None else (a[0], a[1]) for a in got] == [None if b is None else (b[0], b[1]) for b in want]
    assert any(w is None for w in want) and any(w is not None and w[1] == 2 for w in want) and (err != 0).any()
    gpu_ctx.quorum_destroy(qh)


def test_read_proof_and_register_sites(gpu_ctx):
    """Server.read's proof check (protocol/server.go:181-185: CollectiveSignature.Verify(variable, proof, AUTH) --
    the signed bytes are the variable NAME) and Server.register (server.go:452-475: VerifyWithCertificate(TBS(req), sig,
    Issuer(sig)), then CollectiveSignature.Verify(variable, ss, AUTH))."""
    from oracle import openpgp as pgp
    cl, og, hg, host = _world(10)
    me = cl.replicas[0].key_id
    og.set_self([me])
    hg.SetSelfNodes([me])
    oq = W.Wot(og).choose_quorum(W.AUTH)
    hq = host.wotqs.New(hg).ChooseQuorum(host.AUTH)
    kr = H.oracle_keyring(cl)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    suff = oq.qcs[0].suff
    rng = np.random.default_rng(29)
    srv = host.Server(gpu_ctx)

    def proof_for(variable, k, spoil=None):
        members = [cl.replicas[int(i)] for i in rng.permutation(len(cl.replicas))[:k]]
        parts = [cb.detach_sign(m, variable) for m in members]
        if spoil == "value": parts[0] = parts[0][:-4] + bytes([parts[0][-4] ^ 1]) + parts[0][-3:]
        if spoil == "other-bytes": parts = [cb.detach_sign(m, variable + b"?") for m in members]
        return b"".join(parts)
    # ---- read proofs
    reqs, want = [], []
    for i in range(20):
        x = b"authvar%02d" % i
        kind = i % 5
        if kind == 0: ss = opk.SignaturePacket(1, 0, True, proof_for(x, suff + 1), None)
        elif kind == 1: ss = opk.SignaturePacket(1, 0, True, proof_for(x, suff - 1), None)
        elif kind == 2: ss = opk.SignaturePacket(1, 0, True, proof_for(x, suff, spoil="value"), None)
        elif kind == 3: ss = opk.SignaturePacket(1, 0, True, proof_for(x, suff + 2, spoil="other-bytes"), None)
        else: ss = None
        reqs.append(opk.serialize(x, None, 0, None, ss) if ss is not None else opk.serialize(x, None, 0))
        if ss is None:
            want.append(srv.ErrAuthenticationFailure)
        else:
            want.append(0 if col.collective_verify(kr, x, ss, oq).err is None else srv.ErrAuthenticationFailure)
    reqs.append(b"\x00\x00\x00"); want.append(0xFF)
    got = srv.read_proof_verify(hq, reqs)
    assert list(got) == want and 0 in want and srv.ErrAuthenticationFailure in want
    # ---- register
    reqs, want = [], []
    for i in range(18):
        x, v, t = b"regvar%02d" % i, cl.client.entity[:64], i + 1
        tbs = cb.serialize_tbs(x, v, t)
        kind = i % 6
        sigdata, cert = cb.detach_sign(cl.client, tbs), cl.client.entity
        proof = proof_for(x, suff)
        if kind == 1: sigdata = cb.detach_sign(cl.client, tbs + b"!")
        if kind == 2: proof = proof_for(x, suff - 1)
        if kind == 3: cert = None
        if kind == 4: proof = None
        sig = opk.SignaturePacket(1, 0, False, sigdata, cert)
        ss = opk.SignaturePacket(1, 0, False, proof, None) if proof is not None else None
        reqs.append(opk.serialize(x, v, t, sig, ss) if ss is not None else opk.serialize(x, v, t, sig))
        if ss is None:
            want.append(0xFF)
        else:
            ents = pgp.read_entities(cert or b"")
            if not ents: want.append(0xFE)
            elif col.signature_verify_with_certificate(tbs, sig, ents[0]) is not None: want.append(1)
            else: want.append(0 if col.collective_verify(kr, x, ss, oq).err is None else 2)
    got = srv.register_verify(hq, reqs)
    assert list(got) == want and set(want) == {0, 1, 2, 0xFE, 0xFF}


def test_batcher_cert_verify_for_principals_outside_the_keyring(gpu_ctx):
    """SURVEY.md 8(f)-1 behind the Go seam: Signature.Issuer(sig) (-> Certificate.Parse -> openpgp.ReadEntity,
    crypto_pgp.go:392-405, 236-249) and Signature.VerifyWithCertificate (crypto_pgp.go:332-344) for the principal whose certificate
    travels INSIDE the request (server.go:199-207, 460-468) as ONE micro-batched call, bftkv_gpu_batcher_cert_verify.  The server's
    keyring holds the replicas only; every verdict is the oracle's, nothing is fenced, and the node keyring is left as it was."""
    import hashlib
    import struct
    import threading
    from bftkv_amd import Batcher
    from corpus.keys import DRBG
    from oracle import openpgp as pgp
    cl, og, hg, host = _world(10)
    kr = H.oracle_keyring(cl)                                   # NOT the client, NOT the strangers below
    gpu_ctx.keyring_set(H.abi_keys(kr))
    weak = cb.make_keypair(cb.PK_RSA, cb.load_keys("rsa2048", 84)[79], "u02 <u02@bftkv.example>")
    cb.build_entity(weak, [r for r in cl.replicas if r.algo == cb.PK_RSA][:2], DRBG("weak"))
    sub_owner = cb.make_keypair(cb.PK_RSA, cb.load_keys("rsa2048", 84)[82], "s01 <s01@bftkv.example>")
    sub = cb.make_keypair(cb.PK_RSA, cb.load_keys("rsa2048", 84)[83], "")
    cb.build_entity(sub_owner, [], DRBG("sub"), subkey=sub)
    dsa_stranger = cb.make_keypair(cb.PK_DSA, cb.load_keys("dsa2048", 12)[11], "d01 <d01@bftkv.example>")
    cb.build_entity(dsa_stranger, cl.replicas[:4], DRBG("dsa-stranger"))
    client = cl.client.entity
    selfsig = client.index(b"\xc2", client.index(cl.client.name.encode()))
    bad_self = bytearray(client); bad_self[selfsig + 200] ^= 1; bad_self = bytes(bad_self)
    bad_binding = bytearray(sub_owner.entity); bad_binding[len(bad_binding) - 25] ^= 0x20; bad_binding = bytes(bad_binding)
    tbs = cb.serialize_tbs(b"variable", b"value", 7)
    S = lambda kp, data=tbs: cb.detach_sign(kp, data, DRBG("cert-verify"))
    cases = [
        ("client", client, tbs, S(cl.client)),
        ("client, signature over other bytes", client, tbs, S(cl.client, tbs + b"x")),
        ("somebody else's certificate", cl.replicas[0].entity, tbs, S(cl.client)),
        ("forged self-signature: no issuer", bad_self, tbs, S(cl.client)),
        ("no certificate", b"", tbs, S(cl.client)),
        ("garbage certificate", bytes(range(200)), tbs, S(cl.client)),
        ("weakly certified stranger (the quorum certificate is a later check)", weak.entity, tbs, S(weak)),
        ("outsider nobody certified", cl.outsiders[0].entity, tbs, S(cl.outsiders[0])),
        ("second packet by a key outside the certificate", client, tbs, S(cl.client) + S(cl.replicas[1])),
        ("two packets by the certificate's key", client, tbs, S(cl.client) + S(cl.client)),
        ("entity with a bound subkey", sub_owner.entity, tbs, S(sub_owner)),
        ("forged subkey binding: no issuer", bad_binding, tbs, S(sub_owner)),
        ("DSA stranger", dsa_stranger.entity, tbs, S(dsa_stranger)),
        ("empty signature data", client, tbs, b""),
        ("several entities: the first is the issuer", client + weak.entity, tbs, S(cl.client)),
        ("several entities, signed by the second", client + weak.entity, tbs, S(weak)),
        ("issuer alone", client, b"", None),
        ("issuer alone, forged self-signature", bad_self, b"", None),
    ]

    def want_of(cert, tb, sig):
        ents = pgp.entity_checks(cert)
        if not ents or not ents[0]["valid"]:
            return 3, 0
        iid = ents[0]["primary"].key_id
        if sig is None:
            return 0, iid
        e = col.signature_verify_with_certificate(tb, opk.SignaturePacket(1, 0, False, sig or None, cert), pgp.read_entities(cert)[0])
        return (0 if e is None else 1), iid
    wants = [want_of(c, t, s) for _, c, t, s in cases]
    assert {w[0] for w in wants} == {0, 1, 3}
    b = Batcher(gpu_ctx, max_items=64)
    try:
        for (name, cert, tb, sig), (w, iid) in zip(cases, wants):
            err, fenced, got_id, fp = b.cert_verify(cert, tb, sig)
            assert not fenced, name
            assert err == w, (name, err, w)
            if iid:
                body = pgp.next_packet(cert, 0).body
                assert got_id == iid and fp == hashlib.sha1(b"\x99" + struct.pack(">H", len(body)) + body).digest(), name
        # the goroutine-per-request shape: many callers at once, certificates old and new mixed
        got, errs = {}, []

        def caller(k):
            try:
                rng = np.random.default_rng(100 + k)
                for j in range(12):
                    i = int(rng.integers(len(cases)))
                    _, cert, tb, sig = cases[i]
                    got[(k, j)] = (i, b.cert_verify(cert, tb, sig)[:2])
            except Exception as e:      # noqa: BLE001
                errs.append(e)
        ths = [threading.Thread(target=caller, args=(k,)) for k in range(24)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=300)
        assert not errs, errs
        assert len(got) == 24 * 12
        for (k, j), (i, (err, fenced)) in got.items():
            assert not fenced and err == wants[i][0], (cases[i][0], err, wants[i])
        st = b.stats()
        assert st["calls"] >= len(cases) + 24 * 12 and st["batches"] < st["calls"]
        # certificates accepted once are not walked again: their later requests were single signature verifications on the lanes
        # (or, for the issuer alone, answers from the register) -- and gave the oracle's verdicts above all the same
        assert st["cert_fast"] >= 24 * 12 // 3, st
        # ... and the register follows the key table: after a keyring change the same requests take the compound route again
        gpu_ctx.keyring_set(H.abi_keys(kr))
        before = b.stats()["cert_fast"]
        for (name, cert, tb, sig), (w, iid) in zip(cases, wants):
            err, fenced, got_id, fp = b.cert_verify(cert, tb, sig)
            assert not fenced and err == w, (name, err, w)
        assert b.stats()["cert_fast"] - before <= 6, "only a certificate repeated within this round may come from the register"
        for (name, cert, tb, sig), (w, iid) in zip(cases, wants):
            err, fenced, got_id, fp = b.cert_verify(cert, tb, sig)
            assert not fenced and err == w and (not iid or got_id == iid), (name, err, w)
        assert b.stats()["cert_fast"] - before >= 8
        # fail closed
        rc, err, _ = Batcher.cert_verify(b, client, tbs, S(cl.client), raw=True)
        assert rc == 0 and err == 0
    finally:
        b.close()
    # the node keyring is what it was: the client is still unknown to Signature.Verify, the replicas still verify
    cs = S(cl.client, b"abc")
    assert gpu_ctx.signature_verify(np.frombuffer(b"abc", dtype=np.uint8), np.array([0, 3], dtype=np.uint64),
                                    np.frombuffer(cs, dtype=np.uint8), np.array([0, len(cs)], dtype=np.uint64))[0] == 1
    rs = S(cl.replicas[3], b"abc")
    assert gpu_ctx.signature_verify(np.frombuffer(b"abc", dtype=np.uint8), np.array([0, 3], dtype=np.uint64),
                                    np.frombuffer(rs, dtype=np.uint8), np.array([0, len(rs)], dtype=np.uint64))[0] == 0


def test_http_wire_replay(gpu_ctx):
    """SURVEY.md 8(f)-4, transport half: an opened-body HTTP exchange log (transport/http/http.go framing: POST /bftkv/v1/<cmd>,
    500 + X-error on failure) replayed through bftkv_amd.wire -- transport signatures, the command's verification site, and the
    verdict on the recorded answer -- against what the oracle says about every request."""
    from bftkv_amd import wire
    from corpus.keys import DRBG
    from oracle import message as om
    from oracle import openpgp as pgp
    cl = cb.make_cluster(10)
    rng = DRBG("wire")
    for r in cl.replicas:
        cb.build_entity(r, [o for o in cl.replicas if o is not r], rng)
    pubring = b"".join(r.entity for r in cl.replicas) + cl.client.entity      # the node knows the client: its messages verify
    me = cl.replicas[2].key_id
    ents = pgp.read_entities(pubring)
    kr = col.Keyring(ents)
    og = W.Graph()
    og.add_nodes([(e.id, list(e.certifiers)) for e in ents])
    og.set_self([me])
    oq_auth, oq_cert = W.Wot(og).choose_quorum(W.AUTH), W.Wot(og).choose_quorum(W.AUTH | W.CERT)
    c = cb.make_write_corpus(cl, 12, keep_requests=True, mutation_rates={cb.MUT_ONE_SHORT: 0.3, cb.MUT_BAD_MPI: 0.2})
    nrng = np.random.default_rng(3)

    def msg(kp, body, tamper=None):
        m = bytearray(cb.signed_message(kp, body, nrng.bytes(16), rng, "go"))
        if tamper == "sig":
            m[-5] ^= 0x40
        if tamper == "cut":
            m = m[:len(m) // 3]
        return bytes(m)

    def sign_req(i, bad=None):
        x, v, t = b"sk%02d" % i, nrng.bytes(24), i + 1
        tbs = cb.serialize_tbs(x, v, t)
        sigdata = cb.detach_sign(cl.client, tbs + (b"x" if bad == "sig" else b""))
        return opk.serialize(x, v, t, opk.SignaturePacket(1, 0, False, sigdata, None if bad == "nocert" else cl.client.entity))

    def oracle_expect(cmd, body):
        """(transport, site error string) as the reference's handler would reach them"""
        r = om.read_signed_message(ents, body)
        if r.status == om.MSG_READ_ERROR:
            return wire.ERR_DECRYPTION_FAILED, None
        if r.status == om.MSG_NOT_SIGNED:
            return wire.ERR_TRANSPORT_DATA, None
        if r.status == om.MSG_SIGNATURE_ERROR:
            return ("unverified" if cmd == "join" else wire.ERR_INVALID_SIGNATURE), None
        tr = "ok" if r.status == om.MSG_OK else "unverified"
        req = r.plain
        if cmd == "write":
            try:
                x, v, t, sig, ss, _ = opk.parse(req)
            except opk.PacketError:
                return tr, wire.ERR_MALFORMED
            if sig is None or ss is None:
                return tr, wire.ERR_MALFORMED
            return tr, (None if col.collective_verify(kr, opk.tbss(req), ss, oq_auth).err is None else wire.ERR_INSUFFICIENT_SIGNATURES)
        if cmd == "sign":
            try:
                x, v, t, sig, ss, _ = opk.parse(req)
            except opk.PacketError:
                return tr, wire.ERR_MALFORMED
            if sig is None:
                return tr, wire.ERR_MALFORMED
            es = pgp.read_entities(sig.Cert or b"")
            if not es:
                return tr, wire.ERR_CERT_NOT_FOUND
            if col.signature_verify_with_certificate(opk.tbs(req), sig, es[0]) is not None:
                return tr, wire.ERR_INVALID_SIGNATURE
            nodes = [cid for cid in es[0].certifiers if kr.get_cert_by_id(cid) is not None]
            return tr, (None if oq_cert.is_threshold(nodes) else wire.ERR_INVALID_QUORUM_CERT)
        return tr, None

    # (path, opened request body, how the LOG answers: "truth" = what the reference would answer, or a literal (status, X-error))
    plan = []
    for i, req in enumerate(c.requests):
        plan.append(("/bftkv/v1/write", msg(cl.client, req), "truth"))
    plan.append(("/bftkv/v1/write", msg(cl.client, c.requests[0][:40]), "truth"))                       # malformed request inside a good message
    plan.append(("/BFTKV/v1/Write", msg(cl.client, c.requests[1], tamper="sig"), "truth"))              # transport signature broken
    plan.append(("/bftkv/v1/write", msg(cl.client, c.requests[2], tamper="cut"), "truth"))              # not a readable message
    plan.append(("/bftkv/v1/write", msg(cl.outsiders[0], c.requests[3]), "truth"))                      # signer unknown to the keyring: nil error, goes on
    for i, bad in enumerate([None, None, "sig", "nocert"]):
        plan.append(("/bftkv/v1/sign", msg(cl.client, sign_req(i, bad)), "truth"))
    plan.append(("/bftkv/v1/time", msg(cl.client, b""), "truth"))
    plan.append(("/bftkv/v1/join", msg(cl.client, cl.client.entity, tamper="sig"), "truth"))           # join goes on despite the signature error
    plan.append(("/bftkv/v2/write", msg(cl.client, c.requests[0]), (404, None)))
    bad_ss = next(i for i in range(c.n_items) if c.expected_valid[i] < cl.suff)
    good = next(i for i in range(c.n_items) if c.expected_valid[i] >= cl.suff)
    plan.append(("/bftkv/v1/write", msg(cl.client, c.requests[bad_ss]), (200, None)))                   # a node that ACCEPTED an insufficient write
    plan.append(("/bftkv/v1/write", msg(cl.client, c.requests[good]), (500, "bad timestamp")))          # storage-level refusal: not judged
    plan.append(("/bftkv/v1/write", msg(cl.client, c.requests[bad_ss]), (500, wire.ERR_MALFORMED)))     # refused with the wrong identity
    plan.append(("/bftkv/v1/write", msg(cl.client, c.requests[good]), None))                            # no answer in the log
    cap = bytearray()
    want = []
    for k, (path, body, answer) in enumerate(plan):
        cmd = wire.command_of(path)
        tr, site = oracle_expect(cmd, body) if cmd else ("", None)
        failed = tr if tr not in ("ok", "unverified", "") else site
        chunked = k % 3 == 1
        if chunked:
            cap += b"POST %s HTTP/1.1\r\nHost: n\r\nTransfer-Encoding: chunked\r\n\r\n%x\r\n%s\r\n0\r\n\r\n" % (path.encode(), len(body), body)
        else:
            cap += b"POST %s HTTP/1.1\r\nHost: n\r\nContent-Type: application/octet-stream\r\nContent-Length: %d\r\n\r\n%s" % (path.encode(), len(body), body)
        if answer == "truth":
            answer = (200, None) if failed is None else (500, failed)
            verdict = "consistent"
        elif answer is None:
            verdict = "no-response"
        elif cmd is None:
            verdict = "consistent"
        elif failed is None:
            verdict = "consistent" if answer[0] == 200 else "not-judged"
        elif answer[0] == 200:
            verdict = "accepted-but-fails-verification"
        else:
            verdict = "consistent" if answer[1] == failed else "wrong-error"
        if answer is not None:
            st, xe = answer
            cap += b"HTTP/1.1 %d X\r\n%sContent-Length: 2\r\n\r\nok" % (st, (b"X-Error: %s\r\n" % xe.encode()) if xe else b"")
        want.append((cmd, tr, site, verdict))
    exs = wire.replay(gpu_ctx, bytes(cap), pubring, me)
    assert len(exs) == len(plan)
    for e, (cmd, tr, site, verdict) in zip(exs, want):
        assert e.command == cmd, e.index
        if cmd:
            assert e.transport == tr, (e.index, e.transport, tr)
            assert e.site_error == site, (e.index, cmd, e.site_error, site)
        assert e.verdict == verdict, (e.index, cmd, e.verdict, verdict, e.transport, e.site_error)
    seen = {w[3] for w in want}
    assert seen == {"consistent", "no-response", "not-judged", "accepted-but-fails-verification", "wrong-error"}
    assert {w[2] for w in want} >= {None, wire.ERR_MALFORMED, wire.ERR_INSUFFICIENT_SIGNATURES, wire.ERR_CERT_NOT_FOUND, wire.ERR_INVALID_SIGNATURE}
    assert {w[1] for w in want} >= {"ok", "unverified", wire.ERR_DECRYPTION_FAILED, wire.ERR_INVALID_SIGNATURE}


def test_read_entity_shape_by_shape_on_the_gpu(gpu_ctx):
    """openpgp.ReadEntity as Certificate.Parse / Signature.Issuer reach it (crypto_pgp.go:236-249, 392-405), signature checks on
    the GPU (bftkv_host_certs_verify): every hand-worked shape of tests/cert_shapes.py (self-signatures that must verify and the
    last of which counts, subkey bindings and revocations, the cross-signature a signing subkey's binding must carry -- verified
    under the SUBKEY --, key revocations verified at the end, refusals the walk alone decides, shapes left to the reference) and
    every certificate GnuPG made (tests/golden/gpg_cert_vectors.json) gets the verdict x/crypto's rules give; then Issuer +
    VerifyWithCertificate through the batcher for requests signed with a stranger's signing SUBKEY."""
    import json
    import os
    from bftkv_amd import Batcher, host
    from oracle import openpgp as pgp
    from tests import cert_shapes as CS
    cl = cb.make_cluster(4)
    gpu_ctx.keyring_set(H.abi_keys(H.oracle_keyring(cl)))
    seen, bad = set(), []
    for name, blob, valid, _, _ in CS.scenarios():
        got = host.certs_verify(gpu_ctx, blob)
        if got != valid:
            bad.append((name, got, valid))
        seen.update(got)
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gpg_cert_vectors.json")))
    for c in vec["certificates"] + vec["tampered"]:
        got, want = host.certs_verify(gpu_ctx, bytes.fromhex(c["blob"])), [c in vec["certificates"]]
        if got != want:
            bad.append((c["name"], got, want))
    assert not bad, bad
    assert seen == {True, False, None}
    # ---- a stranger whose request is signed with its signing subkey
    a, b_, s, s2, d = CS.keys()
    uid = a.name.encode()
    head = CS.pkt(6, a.pub_body) + CS.pkt(13, uid) + CS.self_sig(a, uid) + CS.pkt(14, s.pub_body)
    tbs = cb.serialize_tbs(b"variable", b"value", 9)
    by_sub, by_primary = cb.detach_sign(s, tbs), cb.detach_sign(a, tbs)
    certs = {
        "cross-signed signing subkey": head + CS.binding(a, s, flags=0x02),
        "the subkey is for encryption": head + CS.binding(a, s, flags=0x0c),
        "the subkey is revoked": head + CS.binding(a, s, flags=0x02) + CS.binding(a, s, sig_type=0x28, flags=None, reason=1),
        "no cross-signature: ReadEntity refuses the certificate": head + CS.binding(a, s, flags=0x02, cross=None),
        "the key is revoked": CS.pkt(6, a.pub_body) + CS.key_revocation(a) + CS.pkt(13, uid) + CS.self_sig(a, uid) + CS.pkt(14, s.pub_body) + CS.binding(a, s, flags=0x02),
        "a user attribute: left to the reference": head + CS.binding(a, s, flags=0x02) + CS.pkt(17, b"\x01\x01"),
    }
    # shapes the walk leaves to the reference must come out FENCED, never as "certificate not found" (what an unknown shape
    # hides -- the self-signature, the binding -- must not turn into a refusal), and the checks ReadEntity makes with the key it HOLDS
    # (binding / cross-signature / revocation that name another issuer or none) get their verdict on the device
    big_unhashed = b"".join(CS.sub(100, bytes(150)) for _ in range(29))
    certs.update({
        "self-signature body over 4096 bytes: left to the reference": CS.pkt(6, a.pub_body) + CS.pkt(13, uid) + CS.self_sig(a, uid, unhashed=big_unhashed),
        "user id under a partial length: left to the reference": CS.pkt(6, a.pub_body) + bytes([0xCD, 0xE4]) + uid[:16] + bytes([len(uid) - 16]) + uid[16:] + CS.self_sig(a, uid),
        "embedded signatures three deep: left to the reference": head + CS.nested_binding(a, s, 3),
        "binding and cross-signature without issuer subpackets": head + CS.binding(a, s, flags=0x02, issuer=None, cross="no-issuer"),
        "binding naming a stranger, forged": head + CS.binding(a, s, flags=0x02, issuer=b_.key_id, spoil=True),
    })
    # a certificate of more entities than the wrapper's first guess (bftkv_amd.host.certs_verify sizes its answer from the call)
    many = b"".join(CS.pkt(6, a.pub_body) + CS.pkt(13, uid) + (CS.self_sig(a, uid) if i % 3 else CS.self_sig(a, uid, spoil=True)) for i in range(70))
    assert host.certs_verify(gpu_ctx, many) == [bool(i % 3) for i in range(70)]
    bt = Batcher(gpu_ctx, max_items=16)
    try:
        outcomes, bad = set(), []
        for name, cert in certs.items():
            ent = pgp.entity_checks(cert)[0]
            if "left to the reference" in name:
                assert ent["valid"] is None, name
            if "without issuer subpackets" in name:
                assert ent["valid"] is True, name
            if "forged" in name:
                assert ent["valid"] is False, name
            for who, sig in (("subkey", by_sub), ("primary", by_primary)):
                err, fenced, got_id, _ = bt.cert_verify(cert, tbs, sig)
                if ent["valid"] is None:
                    want = "fenced"
                elif not ent["valid"]:
                    want = 3                                                         # no issuer: crypto.ErrCertificateNotFound
                else:
                    e = col.signature_verify_with_certificate(tbs, opk.SignaturePacket(1, 0, False, sig, cert), pgp.read_entities(cert)[0])
                    want = 0 if e is None else 1
                got = "fenced" if fenced else err
                if got != want or (want in (0, 1) and got_id != a.key_id):
                    bad.append((name, who, got, want, got_id))
                outcomes.add("fenced" if want == "fenced" else (who, want))
        assert not bad, bad
        assert outcomes == {"fenced", ("subkey", 0), ("subkey", 1), ("subkey", 3), ("primary", 0), ("primary", 1), ("primary", 3)}
    finally:
        bt.close()


def test_issuer_entity_structure_from_the_library(gpu_ctx):
    """Signature.Issuer without openpgp.ReadEntity on the CPU (bftkv_gpu_batcher_cert_entity): for the certificates GnuPG made and the
    76 certificate blobs of tests/golden/reference_inputs.json (the hand-worked ReadEntity shapes among them), the verdict on the first
    entity is the oracle's walk_valid -- its signature checks on the GPU --, and wherever ReadEntity returns the entity the role of
    every packet (what the shim assembles *openpgp.Entity from), the entity's byte range and its fingerprint are the oracle walk's."""
    import json
    import os
    from bftkv_amd import Batcher, host
    from oracle import openpgp as pgp
    cl = cb.make_cluster(4)
    gpu_ctx.keyring_set(H.abi_keys(H.oracle_keyring(cl)))
    gold = os.path.join(os.path.dirname(__file__), "golden")
    vec = json.load(open(os.path.join(gold, "gpg_cert_vectors.json")))
    ri = json.load(open(os.path.join(gold, "reference_inputs.json")))
    blobs = [(c["name"], bytes.fromhex(c["blob"])) for c in vec["certificates"] + vec["tampered"]]
    blobs += [("reference_inputs cert %d" % i, bytes.fromhex(c if isinstance(c, str) else c["blob"])) for i, c in enumerate(ri["certs"])]
    assert len(blobs) >= 80
    role_no = {n: i for i, n in enumerate(host.ROLE_NAMES)}
    bt = Batcher(gpu_ctx, max_items=16)
    seen = {"built": 0, "fenced": 0, "refused": 0}
    try:
        for name, blob in blobs:
            rc, err, fenced, iid, fp, off, ln, roles = bt.cert_entity(blob, cap=4)       # (a small first guess: the retry is part of the contract)
            assert rc == 0, (name, rc)
            ws = pgp.walk_certificate(blob)
            v = pgp.walk_valid(ws[0]) if ws else False
            if v is None:
                assert fenced == 1 and not roles, name
                seen["fenced"] += 1
            elif v is False:
                assert (err, fenced) == (3, 0) and not roles, (name, err, fenced)         # crypto.ErrCertificateNotFound: Issuer() == nil
                seen["refused"] += 1
            else:
                w = ws[0]
                assert (err, fenced) == (0, 0), (name, err, fenced)
                assert iid == w.primary.key_id and fp == w.primary.fingerprint, name
                assert (off, ln) == (w.start, w.end - w.start), name
                assert roles == [(role_no[r], i, ch) for r, i, ch in w.roles], name
                seen["built"] += 1
        assert seen["built"] >= 40 and seen["fenced"] >= 3 and seen["refused"] >= 10, seen
        # fail closed: a dead batcher handle answers with a failure byte and no roles
    finally:
        bt.close()


def test_read_entity_on_random_certificates_with_real_signatures(gpu_ctx):
    """openpgp.ReadEntity over RANDOM packet sequences whose signatures are real (tests/cert_shapes.py random_blobs(real=True): keys,
    user ids in every header format, self-signatures, certifications with and without issuer, bindings and cross-signatures that
    name the right key, another key or none, revocations, nested embedded signatures, secret-key and unmodelled packets, bodies over
    4096 bytes, partial lengths, then bit flips and truncations): the verdict of every entity -- its checks decided by the device's
    arithmetic, with the key ReadEntity holds -- is the oracle's walk_valid.  (One-off beside it: 2,400 more over six seeds, no
    difference.)"""
    from bftkv_amd import host
    from oracle import openpgp as pgp
    from tests import cert_shapes as CS
    cl = cb.make_cluster(4)
    gpu_ctx.keyring_set(H.abi_keys(H.oracle_keyring(cl)))
    hist, bad = {}, []
    for i, blob in enumerate(CS.random_blobs(300, seed=21, real=True)):
        want = [pgp.walk_valid(w) for w in pgp.walk_certificate(blob)]
        got = host.certs_verify(gpu_ctx, blob)
        if got != want:
            bad.append((i, got, want))
        for v in want:
            hist[v] = hist.get(v, 0) + 1
    assert not bad, bad[:3]
    assert min(hist.get(True, 0), hist.get(False, 0), hist.get(None, 0)) >= 40, hist


def test_request_certificates_with_dsa_keys_of_other_sizes(gpu_ctx):
    """A principal whose certificate travels in the request (server.go:199-207) holds a DSA key of 1024/160 or 3072/256 bits.  The
    smaller group is verified like any other key.  A 3072-bit p needs the arena's wide entries, which only the NODE keyring decides
    (an unauthenticated certificate must not make the node's tables be rebuilt): with no such key in the ring the request is FENCED --
    no verdict, the reference path decides --, with one in the ring it is verified, and every verdict given is the oracle's."""
    from bftkv_amd import Batcher
    from corpus.keys import DRBG
    from oracle import openpgp as pgp
    tbs = cb.serialize_tbs(b"variable", b"value", 9)
    strangers = {}
    for kind in ("dsa1024", "dsa3072"):
        kp = cb.make_keypair(cb.PK_DSA, cb.load_keys(kind, 4)[3], "d-%s <%s@bftkv.example>" % (kind, kind))
        cb.build_entity(kp, [], DRBG("stranger", kind))
        strangers[kind] = kp

    def want_of(kp, data):
        cert = kp.entity
        ents = pgp.entity_checks(cert)
        assert ents and ents[0]["valid"]
        sig = cb.detach_sign(kp, data, DRBG("cert-sizes"))
        e = col.signature_verify_with_certificate(tbs, opk.SignaturePacket(1, 0, False, sig, cert), pgp.read_entities(cert)[0])
        return sig, (0 if e is None else 1)

    for ring_kind, expect_3072_fenced in (("dsa2048", True), (("dsa2048", "dsa3072"), False)):
        cl = cb.make_cluster(6, dsa_fraction=0.5, dsa_kind=ring_kind)
        gpu_ctx.keyring_set(H.abi_keys(H.oracle_keyring(cl)))
        assert gpu_ctx.dsa_table_bytes()[1] == (76 if expect_3072_fenced else 112)
        b = Batcher(gpu_ctx, max_items=16)
        try:
            for kind, kp in strangers.items():
                for data in (tbs, tbs + b"!"):
                    sig, want = want_of(kp, data)
                    for _ in range(2):          # first sight (the compound route), then the register
                        err, fenced, got_id, _ = b.cert_verify(kp.entity, tbs, sig)
                        if kind == "dsa3072" and expect_3072_fenced:
                            assert fenced, (ring_kind, kind)
                        else:
                            assert not fenced and err == want and got_id == kp.key_id, (ring_kind, kind, err, want)
        finally:
            b.close()
