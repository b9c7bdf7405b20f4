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
def test_collective_verify_matches_oracle(gpu_ctx, exit_mode, n, items):
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


def _ring_and_ctx(gpu_ctx, cl, include_client=False):
    kr = H.oracle_keyring(cl, include_client=include_client)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    return kr


def _cat(parts):
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts], dtype=np.uint64)
    return np.frombuffer(b"".join(parts) + b"\0", dtype=np.uint8)[:int(off[-1])].copy(), off


@pytest.fixture(params=[True, False], ids=["early-exit", "every-packet"])
def exit_mode(gpu_ctx, request):
    """CollectiveSignature.Verify in both modes: public-key work stops where the reference stops reading (the default), or
    every packet of every item is verified (diagnostics).  err, n_verified and the statuses up to the exit are the same."""
    gpu_ctx.set_early_exit(request.param)
    yield request.param
    gpu_ctx.set_early_exit(True)


def _check_ok_counts(st, st_item, c, suff, early):
    ok = np.bincount(st_item[st == 0], minlength=c.n_items)
    if not early:
        assert (ok == c.expected_valid).all()
        assert not (st == 11).any()
        return
    want_ok = c.expected_valid >= suff
    assert (ok[~want_ok] == c.expected_valid[~want_ok]).all()          # insufficient items: every packet was examined
    assert (ok[want_ok] >= suff).all() and (ok[want_ok] <= c.expected_valid[want_ok]).all()


def test_signature_verify_and_with_certificate(gpu_ctx):
    """PGPSignature.Verify / VerifyWithCertificate (crypto_pgp.go:319-344)."""
    from oracle import collective as col
    from oracle import openpgp as pgp
    from oracle.packet import SignaturePacket
    cl = cb.make_cluster(4)
    kr = _ring_and_ctx(gpu_ctx, cl, include_client=True)
    client_ent = pgp.read_entities(cl.client.entity)[0]
    rng = np.random.default_rng(3)
    tbs_l, sig_l, cert_l = [], [], []
    for i in range(40):
        tbs = cb.serialize_tbs(b"key%04d" % i, rng.bytes(int(rng.integers(0, 200))), i)
        good = cb.detach_sign(cl.client, tbs)
        other = cb.detach_sign(cl.replicas[i % 4], tbs)
        outsider = cb.detach_sign(cl.outsiders[0], tbs)
        variants = [good, good + other, b"", other, outsider + good, good + outsider, good[:-1] + bytes([good[-1] ^ 1]),
                    good + b"\x00", b"\xd4\x01\x00" + good, good + b"\xd4\x01\x00", b"\xfe\x01\x00" + good, good + b"\xfe\x01\x00"]
        sig = variants[i % len(variants)]
        if i % 5 == 4:
            tbs = tbs + b"x"      # signed bytes differ
        tbs_l.append(tbs); sig_l.append(sig); cert_l.append(cl.client.key_id)
    tb, to = _cat(tbs_l)
    sb, so = _cat(sig_l)
    err = gpu_ctx.signature_verify(tb, to, sb, so)
    errc = gpu_ctx.signature_verify(tb, to, sb, so, cert_key_id=np.array(cert_l, dtype=np.uint64))
    n_ok = 0
    for i in range(40):
        sp = SignaturePacket(1, 0, False, sig_l[i] or None, None)
        want = col.signature_verify(kr, tbs_l[i], sp)
        wantc = col.signature_verify_with_certificate(tbs_l[i], sp, client_ent)
        assert (err[i] == 0) == (want is None), (i, err[i], want)
        assert (errc[i] == 0) == (wantc is None), (i, errc[i], wantc)
        assert err[i] in (0, 1) and errc[i] in (0, 1)
        n_ok += want is None
    assert 5 < n_ok < 35
    assert not gpu_ctx.last_fenced.any()
    # a certificate whose entity the device table does not hold (server.go:199-207 hands over sig.Cert, normally a principal
    # outside the node keyring): the library cannot speak for the reference there -- fenced, never a verdict
    stranger = np.array([cl.outsiders[0].key_id ^ 0x55] * 40, dtype=np.uint64)
    errs = gpu_ctx.signature_verify(tb, to, sb, so, cert_key_id=stranger)
    assert (errs == 1).all() and gpu_ctx.last_fenced.all()
    from bftkv_amd._native import Batcher
    b = Batcher(gpu_ctx, max_items=8, n_lanes=1)
    try:
        assert b.signature_verify(tbs_l[0], sig_l[0], cert_key_id=int(stranger[0]), raw=True) == (0, 1, 1)
        assert b.signature_verify(tbs_l[0], sig_l[0], cert_key_id=cl.client.key_id, raw=True) == (0, 0, 0)
    finally:
        b.close()


def test_text_mode_signatures(gpu_ctx, exit_mode):
    """Signature type 0x01: the signed data is hashed in canonical-text form (openpgp.NewCanonicalTextHash -- k_hash_mid_text),
    per item and hash only where a signature asks.  The gpg-judged vectors (every hash, RSA and DSA, the rewriter's corner
    cases), then quorum signatures that mix text-mode and binary packets over one payload, random payloads full of CR / LF,
    and payload lengths around the block boundaries -- verdicts, exit counts and statuses are the oracle's."""
    import hashlib
    import json
    import os
    import struct
    from oracle import collective as col
    from oracle import openpgp as pgp
    from oracle.packet import SignaturePacket
    tv = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gpg_text_vectors.json")))
    ring = pgp.read_entities(bytes.fromhex(tv["pubring"]))
    kr = col.Keyring(keyring=ring)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    tbs_l = [bytes.fromhex(v["payload"]) for v in tv["vectors"]]
    sig_l = [bytes.fromhex(v["sig"]) for v in tv["vectors"]]
    tb, to = _cat(tbs_l)
    sb, so = _cat(sig_l)
    err = gpu_ctx.signature_verify(tb, to, sb, so)
    assert not gpu_ctx.last_fenced.any()
    for v, e, t, s_ in zip(tv["vectors"], err, tbs_l, sig_l):
        want = col.signature_verify(kr, t, SignaturePacket(1, 0, False, s_, None)) is None
        assert (e == 0) == want, v["name"]
        if v["strict"]:
            assert (e == 0) == v["gpg_good"], v["name"]
    # quorum signatures mixing text-mode and binary packets; payloads with CR / LF everywhere and of every length class
    cl = cb.make_cluster(7, dsa_fraction=0.3)
    kr = _ring_and_ctx(gpu_ctx, cl)
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    rng = np.random.default_rng(17)
    from corpus.keys import DRBG
    srng = DRBG("text-mode-gpu")
    names = {8: "sha256", 10: "sha512", 2: "sha1", 9: "sha384", 11: "sha224"}

    def text_sig(kp, payload, hash_id):
        hashed = b"\x05\x02" + struct.pack(">I", cb.CREATION_TIME) + bytes([9, 16]) + struct.pack(">Q", kp.key_id)
        prefix = bytes([4, 1, kp.algo, hash_id]) + struct.pack(">H", len(hashed)) + hashed
        h = pgp.CanonicalTextHash(hashlib.new(names[hash_id]))
        h.update(payload)
        h.raw_update(cb.hash_suffix(prefix))
        digest = h.digest()
        if kp.algo == cb.PK_RSA:
            t = pgp.HASH_PREFIXES[names[hash_id]] + digest
            em = int.from_bytes(b"\x00\x01" + b"\xff" * (256 - len(t) - 3) + b"\x00" + t, "big")
            mp = cb.go_mpi_bytes(kp.rsa_private(em).to_bytes(256, "big"))
        else:
            r, s2 = cb._dsa_sign(kp, digest, srng)
            mp = b"".join(cb.go_mpi_bytes(v.to_bytes((v.bit_length() + 7) // 8, "big")) for v in (r, s2))
        body = prefix + b"\x00\x00" + digest[:2] + mp
        return cb._hdr(2, len(body)) + body
    tbs_l, ss_l = [], []
    alphabet = np.frombuffer(b"\r\n\r\nab \t", dtype=np.uint8)
    for i, ln in enumerate([0, 1, 2, 61, 62, 63, 64, 65, 126, 127, 128, 129, 191, 200, 1000, 5000] + [int(x) for x in rng.integers(0, 400, 16)]):
        payload = alphabet[rng.integers(0, len(alphabet), ln)].tobytes()
        pkts = []
        for k, kp in enumerate(cl.replicas):
            mode = (i + k) % 4
            if mode == 0:
                pkts.append(cb.detach_sign(kp, payload, srng))                       # binary
            else:
                pkts.append(text_sig(kp, payload, (8, 10, 2, 9, 11)[(i + k) % 5]))
        if i % 5 == 4:
            payload = payload + b"\n"                                               # signed bytes differ: nothing verifies
        tbs_l.append(payload); ss_l.append(b"".join(pkts))
    tb, to = _cat(tbs_l)
    sb, so = _cat(ss_l)
    err, nver, _ = gpu_ctx.collective_verify(qh, tb, to, sb, so)
    st, st_item = gpu_ctx.last_statuses()
    assert not gpu_ctx.last_fenced.any()
    n_ok = 0
    for i in range(len(tbs_l)):
        r = col.collective_verify(kr, tbs_l[i], SignaturePacket(Type=1, Data=ss_l[i]), q)
        assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified), (i, len(tbs_l[i]), err[i], nver[i], r.err, r.statuses)
        got = list(st[st_item == i])
        assert got[:len(r.statuses)] == r.statuses, (i, got, r.statuses)
        n_ok += r.err is None
    assert 10 < n_ok < len(tbs_l)
    gpu_ctx.quorum_destroy(qh)


def test_md5_and_ripemd160_by_availability_policy(gpu_ctx):
    """bftkv_gpu_set_hash_policy: unknown (default) => such signatures are FENCED; declared not available => the reference's
    unsupported-hash failure, not fenced; declared available => hashed and verified natively (k_hash_mid_other / k_hash_mid_text,
    little-endian compressions), RSA against Go's DigestInfo table.  Verdicts are the oracle's under the same policy; other
    signatures of the same items are unaffected."""
    import json
    import os
    from oracle import collective as col
    from oracle import openpgp as pgp
    from oracle.packet import SignaturePacket
    wv = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gpg_weak_hash_vectors.json")))
    ring = pgp.read_entities(bytes.fromhex(wv["pubring"]))
    kr = col.Keyring(keyring=ring)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    tbs_l = [bytes.fromhex(v["payload"]) for v in wv["vectors"]]
    sig_l = [bytes.fromhex(v["sig"]) for v in wv["vectors"]]
    # the same payloads signed with SHA-256 by the cluster's first key, appended as a second item set: untouched by the policy
    cl = cb.make_cluster(4, n_outsiders=1)
    extra = [cb.detach_sign(cl.replicas[0], t) for t in tbs_l[:3]]
    tb, to = _cat(tbs_l + tbs_l[:3])
    sb, so = _cat(sig_l + extra)
    n = len(sig_l)
    saved = dict(pgp.HASH_POLICY)
    try:
        for state, policy in ((0, None), (2, False), (1, True)):
            gpu_ctx.set_hash_policy(1, state)
            gpu_ctx.set_hash_policy(3, state)
            pgp.HASH_POLICY.update(md5=policy, ripemd160=policy)
            err = gpu_ctx.signature_verify(tb, to, sb, so)
            fenced = gpu_ctx.last_fenced.copy()
            assert (err[n:] == 0).all() and not fenced[n:].any()
            for i, v in enumerate(wv["vectors"]):
                want = col.signature_verify(kr, tbs_l[i], SignaturePacket(1, 0, False, sig_l[i], None)) is None
                assert fenced[i] == (1 if state == 0 else 0), (v["name"], state)
                if not fenced[i]:
                    assert (err[i] == 0) == want, (v["name"], state, err[i])
            if state == 1:
                assert (err[:n] == 0).sum() >= 7
        # one hash available, the other unknown
        gpu_ctx.set_hash_policy(1, 1)
        gpu_ctx.set_hash_policy(3, 0)
        err = gpu_ctx.signature_verify(tb, to, sb, so)
        for i, v in enumerate(wv["vectors"]):
            assert gpu_ctx.last_fenced[i] == (1 if "ripemd160" in v["name"] else 0), v["name"]
    finally:
        pgp.HASH_POLICY.update(saved)
        gpu_ctx.set_hash_policy(1, 0)
        gpu_ctx.set_hash_policy(3, 0)


def test_signers_parse_only(gpu_ctx):
    """PGPSignature.Signers / PGPCollectiveSignature.Signers (crypto_pgp.go:373-390, 517-519)."""
    from oracle import collective as col
    from oracle.packet import SignaturePacket
    cl = cb.make_cluster(10)
    kr = _ring_and_ctx(gpu_ctx, cl)
    c = cb.make_write_corpus(cl, 30, mutation_rates={cb.MUT_BAD_MPI: 0.2, cb.MUT_UNKNOWN_ISSUER: 0.3, cb.MUT_DUP_SIGNER: 0.2})
    parts = [c.ss_data(i) for i in range(30)]
    parts[3] = b""
    parts[4] = parts[4][:300] + b"\x00garbage" + parts[4][300:]      # bad tag byte ends the walk
    parts[5] = b"\xd4\x02\x01\x02" + parts[5]                          # unknown packet type is skipped
    parts[6] = parts[6][:287 + 100]                                    # truncated second packet
    sb, so = _cat(parts)
    ids, off = gpu_ctx.signers(sb, so)
    for i in range(30):
        want = col.signers(kr, SignaturePacket(1, 0, False, parts[i] or None, None))
        got = [int(x) for x in ids[int(off[i]):int(off[i + 1])]]
        assert got == want, i


def test_quorum_tally_over_id_lists(gpu_ctx):
    """wotq.IsQuorum / IsThreshold / IsSufficient / Reject over node lists (wotqs.go:144-185)."""
    from oracle import wotqs as W
    rng = np.random.default_rng(11)
    quorums = [
        W.WotQ([W.new_qc(list(range(1, 5)), 4, W.AUTH, 0)]),
        W.WotQ([W.new_qc(list(range(1, 11)), 10, W.AUTH, 0), W.new_qc(list(range(20, 27)), 0, W.READ, 0)]),
        W.WotQ([W.new_qc(list(range(1, 65)), 64, W.AUTH | W.CERT, 0)]),
        W.WotQ([W.QC(list(range(30, 36)), 0, 0, 0, 0), W.new_qc(list(range(1, 8)), 0, W.READ, 0)]),
        W.WotQ([]),
    ]
    for q in quorums:
        qh = gpu_ctx.quorum_create(H.abi_qcs(q))
        lists = [[]] + [list(rng.integers(0, 70, size=int(rng.integers(0, 90)))) for _ in range(60)]
        lists += [list(range(1, k)) for k in (3, 4, 5, 8, 11, 23, 44, 65)] + [[1] * 50, [1, 1, 2, 2, 3, 3]]
        flat = np.array([x for l in lists for x in l], dtype=np.uint64)
        off = np.zeros(len(lists) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(l) for l in lists])
        v = gpu_ctx.quorum_tally(qh, flat, off)
        for l, got in zip(lists, v):
            want = (1 if q.is_quorum(l) else 0) | (2 if q.is_threshold(l) else 0) | (4 if q.is_sufficient(l) else 0) | (8 if q.reject(l) else 0)
            assert got == want, (l, got, want)
        gpu_ctx.quorum_destroy(qh)


def test_malformed_and_edge_streams(gpu_ctx, exit_mode):
    """Ragged / empty / malformed inputs: statuses and verdicts follow the oracle packet by packet."""
    cl = cb.make_cluster(4)
    kr = _ring_and_ctx(gpu_ctx, cl)
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    rng = np.random.default_rng(5)
    tbs_l, ss_l = [], []
    pub = cl.replicas[0].entity[:280]                                   # a well-formed public-key packet (tag 6)
    for i in range(64):
        ln = [0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 200, 1000][i % 14]   # SHA-256 padding boundaries
        tbs = rng.bytes(ln)
        sigs = [cb.detach_sign(r, tbs) for r in cl.replicas]
        k = i % 16
        if k == 0: data = b""
        elif k == 1: data = b"".join(sigs)
        elif k == 2: data = sigs[0] + pub + sigs[1] + sigs[2]                         # non-signature packet in between
        elif k == 3: data = sigs[0] + b"\xd4\x03abc" + sigs[1] + sigs[2]              # unknown packet type
        elif k == 4: data = sigs[0] + sigs[1] + sigs[2][:100]                          # truncated
        elif k == 5: data = b"\x00\x01\x02" + b"".join(sigs[:3])                       # bytes without the tag MSB
        elif k == 6: data = rng.bytes(300)                                             # noise
        elif k == 7: data = sigs[0] + sigs[1] + b"\x89"                                # lone header byte
        elif k == 8:
            b = bytearray(sigs[0]); b[3] = 3; data = bytes(b) + sigs[1] + sigs[2] + sigs[3]      # version 3 body
        elif k == 9:
            b = bytearray(sigs[0]); b[6] = 99; data = bytes(b) + sigs[1] + sigs[2] + sigs[3]     # unknown hash id
        elif k == 10:
            b = bytearray(sigs[0]); b[5] = 22; data = bytes(b) + sigs[1] + sigs[2] + sigs[3]     # unknown pk algo
        elif k == 11:
            b = bytearray(sigs[0]); b[10] = 0x83; data = bytes(b) + sigs[1] + sigs[2] + sigs[3]  # creation time -> critical expiry: no creation time
        elif k == 12:
            b = bytearray(sigs[0]); b[4] = 1; data = bytes(b) + sigs[1] + sigs[2] + sigs[3]      # text signature type
        elif k == 13: data = sigs[0] * 3                                                # duplicates count
        elif k == 14: data = b"".join(cb.detach_sign(r, tbs + b"!") for r in cl.replicas)   # all over other bytes
        else: data = sigs[3] + sigs[2] + sigs[1]
        tbs_l.append(tbs); ss_l.append(data)
    tb, to = _cat(tbs_l)
    sb, so = _cat(ss_l)
    err, nver, verdict = gpu_ctx.collective_verify(qh, tb, to, sb, so)
    st, st_item = gpu_ctx.last_statuses()
    from oracle import collective as col
    from oracle.packet import SignaturePacket
    for i in range(64):
        r = col.collective_verify(kr, tbs_l[i], SignaturePacket(1, 0, False, ss_l[i] or None, None), q)
        assert (err[i] == 0) == (r.err is None), (i, i % 16, r.statuses, list(st[st_item == i]))
        assert nver[i] == len(r.verified), (i, i % 16)
        got = list(st[st_item == i])
        assert got[:len(r.statuses)] == r.statuses, (i, i % 16, got, r.statuses)
    gpu_ctx.quorum_destroy(qh)


def test_golden_gpg_vectors_on_gpu(gpu_ctx):
    """The committed GnuPG fixtures (tests/golden/gpg_vectors.json) through the HIP path."""
    import json
    import os
    from oracle import collective as col
    from oracle import openpgp as pgp
    from oracle.packet import SignaturePacket
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gpg_vectors.json")))
    for ring_key, group in (("A_pubring", "A"), ("B_pubring", "B"), ("C_pubring", "C")):   # C: rsa3072 / rsa4096
        ring = pgp.read_entities(bytes.fromhex(vec[ring_key]))
        kr = col.Keyring(keyring=ring)
        gpu_ctx.keyring_set(H.abi_keys(kr))
        tbs_l = [bytes.fromhex(v["payload"]) for v in vec[group]]
        sig_l = [bytes.fromhex(v["sig"]) for v in vec[group]]
        tb, to = _cat(tbs_l)
        sb, so = _cat(sig_l)
        err = gpu_ctx.signature_verify(tb, to, sb, so)
        if group == "C":   # tampered payloads and a flipped signature bit as well
            tbs_l = tbs_l + [t + b"!" for t in tbs_l] + tbs_l
            sig_l = sig_l + sig_l + [s[:-5] + bytes([s[-5] ^ 4]) + s[-4:] for s in sig_l]
            tb, to = _cat(tbs_l)
            sb, so = _cat(sig_l)
            err = gpu_ctx.signature_verify(tb, to, sb, so)
            for e, t, s in zip(err, tbs_l, sig_l):
                assert (e == 0) == (col.signature_verify(kr, t, SignaturePacket(1, 0, False, s, None)) is None)
            assert (err[:16] == 0).all() and (err[16:] == 1).all()
            continue
        checked = 0
        for v, e, t, s in zip(vec[group], err, tbs_l, sig_l):
            want = col.signature_verify(kr, t, SignaturePacket(1, 0, False, s, None))
            assert (e == 0) == (want is None), v
            if want is None or not v["gpg_good"]:
                assert (e == 0) == v["gpg_good"]
            checked += 1
        assert checked == len(vec[group])


def test_negative_gpg_vectors_on_gpu(gpu_ctx):
    """tests/golden/gpg_negative_vectors.json (hand-built edge cases judged by gpg): the HIP path gives the oracle's verdict
    on every vector -- including the SignatureV3 one, which gpg accepts -- and raises the fence flag exactly on the one fenced
    shape among them (text mode)."""
    import json
    import os
    from oracle import collective as col
    from oracle import openpgp as pgp
    from oracle.packet import SignaturePacket
    neg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gpg_negative_vectors.json")))
    kr = col.Keyring(keyring=pgp.read_entities(bytes.fromhex(neg["pubring"])))
    gpu_ctx.keyring_set(H.abi_keys(kr))
    payload = bytes.fromhex(neg["payload"])
    sigs = [bytes.fromhex(v["sig"]) for v in neg["vectors"]]
    tb, to = _cat([payload] * len(sigs))
    sb, so = _cat(sigs)
    err = gpu_ctx.signature_verify(tb, to, sb, so)
    for v, e, f, s in zip(neg["vectors"], err, gpu_ctx.last_fenced, sigs):
        want = col.signature_verify(kr, payload, SignaturePacket(1, 0, False, s, None)) is None
        assert (e == 0) == want, v["name"]
        assert f == 0, v["name"]          # (text mode is hashed natively since round 3: nothing among these is fenced)
        if v["strict"]:
            assert (e == 0) == v["gpg_good"], v["name"]


def test_full_size_properties_cfg2(gpu_ctx, exit_mode):
    """BASELINE configs[1] shape (64 replicas, suff 43) at 1,500 writes / ~80k signatures, signed on the GPU:
    verdicts follow from how the corpus was built -- no oracle in the loop."""
    cl = cb.make_cluster(64)
    mods, exps = cb.signer_tables(cl)
    signer = lambda em, ki: gpu_ctx.modexp(em, ki.astype(np.uint32), mods, exps)
    c = cb.make_write_corpus(cl, 1500, batch_signer=signer, seed=77,
                             mutation_rates={cb.MUT_BAD_MPI: 0.05, cb.MUT_UNKNOWN_ISSUER: 0.05, cb.MUT_DUP_SIGNER: 0.05,
                                             cb.MUT_ONE_SHORT: 0.05, cb.MUT_BAD_TAG: 0.05})
    kr = _ring_and_ctx(gpu_ctx, cl)
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    err, nver, verdict = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    suff = cl.suff
    want_ok = c.expected_valid >= suff
    assert ((err == 0) == want_ok).all()
    assert (nver == np.where(want_ok, suff, c.expected_valid)).all()
    assert ((verdict & 4) != 0).tolist() == want_ok.tolist()
    st, st_item = gpu_ctx.last_statuses()
    assert len(st) == c.n_sigs and (np.bincount(st_item, minlength=c.n_items) == c.sig_count).all()
    _check_ok_counts(st, st_item, c, suff, exit_mode)
    ops = gpu_ctx.last_counters()["pubkey_ops"]
    if exit_mode:
        # the reference examines (at least) suff packets of a sufficient write and all packets of an insufficient one
        floor = int(np.where(want_ok, suff, c.expected_valid).sum())
        assert floor <= ops <= floor + 0.06 * c.n_sigs, (floor, ops, c.n_sigs)
    else:
        assert ops >= 0.93 * c.n_sigs
    # idempotence: same inputs, same outputs
    err2, nver2, verdict2 = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    assert (err2 == err).all() and (nver2 == nver).all() and (verdict2 == verdict).all()
    # order independence of the verdict (counts are monotone): reverse every item's packet order
    rev = [b"".join(reversed([c.ss_data(i)[j:j + 287] for j in range(0, len(c.ss_data(i)), 287)])) for i in range(200)]
    sb, so = _cat(rev)
    err3, _, _ = gpu_ctx.collective_verify(qh, c.tbss_blob[:int(c.tbss_off[200])], c.tbss_off[:201], sb, so)
    assert (err3 == err[:200]).all()
    gpu_ctx.quorum_destroy(qh)


def test_cfg2_full_size_identity_against_the_c_oracle(gpu_ctx):
    """BASELINE configs[1] at FULL size -- 64 replicas, 10,000 RSA-2048 signed writes, the bench's seed and mutation mix --
    compared write by write with the reference-shaped CPU path (oracle/c/oracle.c): error byte, exit count, and the number of
    public-key operations the reference performs (bench.py's `value` counts exactly those)."""
    import os
    from oracle.cbind import COracle
    import bench
    cl = cb.make_cluster(64)
    mods, exps = cb.signer_tables(cl)
    signer = lambda em, ki: gpu_ctx.modexp(em, ki.astype(np.uint32), mods, exps)
    c = cb.make_write_corpus(cl, 10000, seed=cb.MASTER_SEED, batch_signer=signer, with_client_sig=True)
    kr = _ring_and_ctx(gpu_ctx, cl)
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    err, nver, verdict = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    st, st_item = gpu_ctx.last_statuses()
    co = COracle()
    co.set_keyring(kr)
    co.set_quorum(q)
    cerr, cnver, ops = co.collective_verify(c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off, n_threads=max(1, min(32, os.cpu_count() or 1)))
    assert (cerr == err).all() and (cnver == nver).all()
    assert len(st) == c.n_sigs and bench.reference_pubkey_ops(st, st_item, err, nver, c.n_items) == ops
    assert 400000 < ops <= gpu_ctx.last_counters()["pubkey_ops"] < c.n_sigs
    assert (err == 0).sum() > 9000 and (err == 2).sum() > 50
    # Several such batches in flight on forked contexts (what bench.py's default does): their machine-filling modexps take turns
    # at the device's turnstile (capi.hip), each waiting on its stream for the one another context launched before it.  Same
    # answers from every context, every time.
    import threading
    forks = [gpu_ctx.fork() for _ in range(2)]
    got = {}

    def storm(k, cx):
        got[k] = [cx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off) for _ in range(3)]
    ths = [threading.Thread(target=storm, args=(k, cx)) for k, cx in enumerate([gpu_ctx] + forks)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert len(got) == 3
    for res in got.values():
        for e2, nv2, vd2 in res:
            assert (e2 == err).all() and (nv2 == nver).all() and (vd2 == verdict).all()
    for f in forks:
        f.close()
    gpu_ctx.quorum_destroy(qh)


def test_rccl_allgather_entry_points_single_rank(gpu_ctx):
    """bftkv_gpu_comm_* / bftkv_gpu_allgather_verdicts: librccl loads, a 1-rank communicator gathers by copying.
    (The box has one GPU; the N>1 exchange is covered by tests/test_dist_gloo.py and bench.py --gpus N.)"""
    import torch
    from bftkv_amd import Context
    from bftkv_amd import dist as D
    uid = Context.comm_unique_id()
    assert uid.any()
    gpu_ctx.comm_init(1, 0, uid)
    ok = torch.tensor([1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 1], dtype=torch.uint8, device="cuda")
    bits = D.pack_verdicts(ok, 11)
    out = torch.zeros_like(bits)
    gpu_ctx.allgather_verdicts(bits.data_ptr(), bits.numel(), out.data_ptr())
    assert torch.equal(D.unpack_verdicts(out, 11), ok)
    assert torch.equal(D.allgather_verdicts(ok, 11), ok)


def test_rccl_path_on_one_gpu_with_a_real_communicator(monkeypatch):
    """The RCCL branch itself on the 1-GPU box: BFTKV_FORCE_RCCL makes a one-rank context build a REAL communicator
    (ncclCommInitRank through the dlsym'd pointer with the 128-byte id by value), the self-test and the exchange step then go
    through ncclAllGather on the verifier's stream.  torch is imported first, so the library must resolve the librccl the
    process already holds (one RCCL per process) -- the same order bench.py --gpus N produces."""
    import torch
    from bftkv_amd import Context
    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda:0")
    path, pre = Context.comm_library()
    assert "rccl" in path
    import ctypes
    already = any("librccl" in ln for ln in open("/proc/self/maps"))
    assert already
    monkeypatch.setenv("BFTKV_FORCE_RCCL", "1")
    ctx = Context(0)
    try:
        ctx.comm_init(1, 0, Context.comm_unique_id())
        ctx.comm_selftest(4096)
        ctx.comm_selftest(1)
        rng = np.random.default_rng(5)
        for n in (1, 9, 1000, 125000):
            err = rng.choice(np.array([0, 2], dtype=np.uint8), size=n)
            d_err = torch.from_numpy(err).to("cuda:0")
            out = torch.full(((n + 7) // 8,), 0xAA, dtype=torch.uint8, device="cuda:0")
            ctx.allgather_errs_dev(d_err.data_ptr(), n, n, out.data_ptr())
            ctx.sync()
            want = np.packbits(np.concatenate([err == 0, np.zeros(((n + 7) // 8) * 8 - n, dtype=bool)]), bitorder="little")
            assert out.cpu().numpy().tobytes() == want.tobytes()
        # exactly one librccl mapped into the process
        libs = {ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln}
        assert len(libs) == 1, libs
    finally:
        ctx.close()


def test_exchange_step_on_the_verifier_stream(gpu_ctx):
    """bftkv_gpu_allgather_errs_dev: the err bytes of a verify call packed into the verdict bitmap on the device and gathered
    rank-major, asynchronously on the context's stream (one rank here: its row is the output)."""
    import torch
    uid = __import__("bftkv_amd").Context.comm_unique_id()
    gpu_ctx.comm_init(1, 0, uid)
    rng = np.random.default_rng(8)
    for n, slots in ((1, 1), (7, 8), (8, 8), (9, 16), (1000, 1003), (100000, 100000)):
        err = rng.choice(np.array([0, 2], dtype=np.uint8), size=n)
        d_err = torch.from_numpy(err).to("cuda:0")
        out = torch.full(((slots + 7) // 8,), 0xAA, dtype=torch.uint8, device="cuda:0")
        gpu_ctx.allgather_errs_dev(d_err.data_ptr(), n, slots, out.data_ptr())
        gpu_ctx.sync()
        want = np.packbits(np.concatenate([err == 0, np.zeros(((slots + 7) // 8) * 8 - n, dtype=bool)]), bitorder="little")
        assert out.cpu().numpy().tobytes() == want.tobytes()


def test_two_phase_planning_reaches_the_reference_exit(gpu_ctx):
    """Phase 1 verifies up to the optimistic exit (+ margin); phase 2 must pick up every item that several bad signatures,
    signers from outside the clique, or a second clique push beyond it.  Statuses up to the reference's exit, n_verified and
    err follow the oracle; the public-key work stays below 'every packet'."""
    from oracle import collective as col
    from oracle import wotqs
    from oracle.packet import SignaturePacket
    cl = cb.make_cluster(13)                      # f = 4, suff = 9
    kr = _ring_and_ctx(gpu_ctx, cl)
    ids = [r.key_id for r in cl.replicas]
    # two cliques over the same keyring: the first 10 members (suff 7) and the last 7 (suff 5); IsSufficient is an OR
    qa, qb = wotqs.new_qc(ids[:10], 10, wotqs.AUTH, 0), wotqs.new_qc(ids[6:], 7, wotqs.AUTH, 0)
    for q in (H.clique_quorum(cl), wotqs.WotQ([qa, qb]), wotqs.WotQ([qa])):
        qh = gpu_ctx.quorum_create(H.abi_qcs(q))
        rng = np.random.default_rng(len(q.qcs) * 7 + q.qcs[0].suff)
        tbs_l, ss_l = [], []
        for i in range(120):
            tbs = rng.bytes(int(rng.integers(1, 120)))
            order = rng.permutation(13)[:int(rng.integers(5, 14))]
            parts = []
            for j in order:
                s = bytearray(cb.detach_sign(cl.replicas[int(j)], tbs))
                u = rng.random()
                if u < 0.25: s[-3] ^= 0x10                                # bad signature value
                elif u < 0.30: s[len(s) - 256 - 2 - 2] ^= 0x01               # hash tag
                elif u < 0.35: s = bytearray(cb.detach_sign(cl.outsiders[0], tbs))    # unknown issuer
                parts.append(bytes(s))
                if rng.random() < 0.1: parts.append(bytes(s))             # duplicate packet: counted again (wotqs.go:195-206)
            tbs_l.append(tbs); ss_l.append(b"".join(parts))
        tb, to = _cat(tbs_l)
        sb, so = _cat(ss_l)
        res = {}
        for early in (True, False):
            gpu_ctx.set_early_exit(early)
            err, nver, _ = gpu_ctx.collective_verify(qh, tb, to, sb, so)
            st, st_item = gpu_ctx.last_statuses()
            res[early] = (err.copy(), nver.copy(), st.copy(), gpu_ctx.last_counters()["pubkey_ops"])
            for i in range(120):
                r = col.collective_verify(kr, tbs_l[i], SignaturePacket(1, 0, False, ss_l[i], None), q)
                assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified), (i, early)
                assert list(st[st_item == i][:len(r.statuses)]) == r.statuses, (i, early)
        gpu_ctx.set_early_exit(True)
        assert (res[True][0] == res[False][0]).all() and (res[True][1] == res[False][1]).all()
        assert res[True][3] < res[False][3] and (res[True][2] == 11).any() and (res[True][0] == 0).any() and (res[True][0] == 2).any()
        gpu_ctx.quorum_destroy(qh)


def test_cfg4_shape_256_replicas(gpu_ctx, exit_mode):
    """BASELINE configs[3] shape: 256-replica clique (f=85, suff=171), 171..256 packets per write -- exercises the
    sequential-walk fallback (more than WALK_CAP packet events per item); verdicts vs construction and vs the oracle."""
    cl = cb.make_cluster(256)
    mods, exps = cb.signer_tables(cl)
    signer = lambda em, ki: gpu_ctx.modexp(em, ki.astype(np.uint32), mods, exps)
    c = cb.make_write_corpus(cl, 24, batch_signer=signer, seed=4,
                             mutation_rates={cb.MUT_BAD_MPI: 0.1, cb.MUT_UNKNOWN_ISSUER: 0.1, cb.MUT_DUP_SIGNER: 0.1, cb.MUT_ONE_SHORT: 0.2})
    assert cl.suff == 171 and c.sig_count.min() >= 171
    kr = _ring_and_ctx(gpu_ctx, cl)
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    err, nver, verdict = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    want_ok = c.expected_valid >= cl.suff
    assert ((err == 0) == want_ok).all() and (nver == np.where(want_ok, cl.suff, c.expected_valid)).all()
    assert want_ok.any() and (~want_ok).any()
    st, st_item = gpu_ctx.last_statuses()
    for i in (0, 5, 11, 23):
        r = H.oracle_collective(kr, q, c, i)
        assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified)
        assert list(st[st_item == i][:len(r.statuses)]) == r.statuses
    gpu_ctx.quorum_destroy(qh)


def test_cfg3_shape_mixed_rsa_dsa(gpu_ctx, exit_mode):
    """BASELINE configs[2] shape: 64 replicas, half RSA-2048 / half DSA-2048-256, collective signatures over reads."""
    cl = cb.make_cluster(64, dsa_fraction=0.5)
    assert sum(r.algo == cb.PK_DSA for r in cl.replicas) == 32
    mods, exps = cb.signer_tables(cl)
    signer = lambda em, ki: gpu_ctx.modexp(em, ki.astype(np.uint32), mods, exps)
    c = cb.make_write_corpus(cl, 40, batch_signer=signer, seed=9,
                             mutation_rates={cb.MUT_BAD_MPI: 0.15, cb.MUT_UNKNOWN_ISSUER: 0.1, cb.MUT_DUP_SIGNER: 0.1, cb.MUT_ONE_SHORT: 0.2})
    kr = _ring_and_ctx(gpu_ctx, cl)
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    err, nver, verdict = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    want_ok = c.expected_valid >= cl.suff
    assert ((err == 0) == want_ok).all() and (nver == np.where(want_ok, cl.suff, c.expected_valid)).all()
    st, st_item = gpu_ctx.last_statuses()
    _check_ok_counts(st, st_item, c, cl.suff, exit_mode)
    cnt = gpu_ctx.last_counters()
    assert cnt["pubkey_ops"] >= (0.7 if exit_mode else 0.9) * c.n_sigs and cnt["dsa_ops"] > 0.25 * cnt["pubkey_ops"]
    for i in (0, 7, 19, 39):
        r = H.oracle_collective(kr, q, c, i)
        assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified)
        assert list(st[st_item == i][:len(r.statuses)]) == r.statuses
    gpu_ctx.quorum_destroy(qh)


def test_dsa_fixed_base_tables_both_widths(gpu_ctx):
    """DSA verifies from per-key window tables (k_dsa_build_comb): the 18-, 16-, 8- and 4-bit layouts and the graded widths between
    (14, 13, 12, 10 bits: windows that straddle words, a narrow top window) give the oracle's statuses, and re-uploading the keyring
    (tables cached by key material) changes nothing."""
    cl = cb.make_cluster(12, dsa_fraction=1.0)
    c = cb.make_write_corpus(cl, 48, seed=21, mutation_rates={cb.MUT_BAD_MPI: 0.2, cb.MUT_DUP_SIGNER: 0.1, cb.MUT_ONE_SHORT: 0.2})
    q = H.clique_quorum(cl)
    want = None
    try:
        for bits in (8, 4, 4, 10, 12, 13, 14, 15, 16, 18, 0):
            gpu_ctx.set_dsa_window_bits(bits)
            kr = _ring_and_ctx(gpu_ctx, cl)
            qh = gpu_ctx.quorum_create(H.abi_qcs(q))
            err, nver, _ = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
            st, st_item = gpu_ctx.last_statuses()
            gpu_ctx.quorum_destroy(qh)
            if want is None:
                want = (err.copy(), nver.copy(), st.copy())
                _check_ok_counts(st, st_item, c, cl.suff, True)
                for i in range(0, c.n_items, 5):
                    r = H.oracle_collective(kr, q, c, i)
                    assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified)
                    assert list(st[st_item == i][:len(r.statuses)]) == r.statuses
            else:
                assert (err == want[0]).all() and (nver == want[1]).all() and (st == want[2]).all()
    finally:
        gpu_ctx.set_dsa_window_bits(0)


def test_signature_blob_beyond_4gib(gpu_ctx):
    """Offsets are 64-bit end to end: a 4.3 GiB signature blob (a small mixed RSA/DSA corpus tiled) verifies tile for tile
    like the first copy.  Sized for the 288 GB part: ~15 M packets, ~11 GB of arena."""
    cl = cb.make_cluster(16, dsa_fraction=0.25)
    mods, exps = cb.signer_tables(cl)
    signer = lambda em, ki: gpu_ctx.modexp(em, ki.astype(np.uint32), mods, exps)
    c = cb.make_write_corpus(cl, 256, batch_signer=signer, seed=31,
                             mutation_rates={cb.MUT_BAD_MPI: 0.1, cb.MUT_UNKNOWN_ISSUER: 0.05, cb.MUT_ONE_SHORT: 0.2})
    _ring_and_ctx(gpu_ctx, cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(H.clique_quorum(cl)))
    err0, nver0, vd0 = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    st0, _ = gpu_ctx.last_statuses()
    T = int((4.3 * 2**30) // len(c.ss_blob)) + 1
    sb = np.tile(c.ss_blob, T)
    assert len(sb) > 2**32
    tb = np.tile(c.tbss_blob, T)
    step_t, step_s = np.uint64(c.tbss_off[-1]), np.uint64(c.ss_off[-1])
    to = (np.arange(T, dtype=np.uint64)[:, None] * step_t + c.tbss_off[None, :-1].astype(np.uint64)).reshape(-1)
    so = (np.arange(T, dtype=np.uint64)[:, None] * step_s + c.ss_off[None, :-1].astype(np.uint64)).reshape(-1)
    to = np.concatenate([to, [np.uint64(T) * step_t]]).astype(np.uint64)
    so = np.concatenate([so, [np.uint64(T) * step_s]]).astype(np.uint64)
    err, nver, vd = gpu_ctx.collective_verify(qh, tb, to, sb, so)
    st, _ = gpu_ctx.last_statuses()
    assert (err.reshape(T, -1) == err0[None, :]).all() and (nver.reshape(T, -1) == nver0[None, :]).all()
    assert (vd.reshape(T, -1) == vd0[None, :]).all()
    assert (st.reshape(T, -1) == st0[None, :]).all()
    gpu_ctx.quorum_destroy(qh)


def test_bad_arguments_and_reentrancy(gpu_ctx):
    """Infrastructure errors are return codes (never verdicts); one context may be called from several threads."""
    import threading
    from bftkv_amd import NativeError
    cl = cb.make_cluster(4)
    kr = _ring_and_ctx(gpu_ctx, cl)
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    c = cb.make_write_corpus(cl, 40, mutation_rates={cb.MUT_ONE_SHORT: 0.3})
    bad = c.ss_off.copy(); bad[3], bad[4] = bad[4], bad[3]
    with pytest.raises(NativeError):
        gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, bad)
    with pytest.raises(NativeError):
        gpu_ctx.collective_verify(qh + 99, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    with pytest.raises(NativeError):
        gpu_ctx.quorum_create([(1, 4, 3, 3, [1, 2, 3, 4])] * 9)          # more cliques than MAX_QC
    want, _, _ = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
    res = [None] * 6

    def worker(k):
        res[k] = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)[0]
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert all((r == want).all() for r in res) and (want == 0).any() and (want == 2).any()
    # zero items is a no-op
    e, nv, vd = gpu_ctx.collective_verify(qh, np.zeros(0, np.uint8), np.zeros(1, np.uint64), np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert len(e) == 0
    gpu_ctx.quorum_destroy(qh)


def _rsa_key_with_e(e, idx):
    """A seeded RSA-2048 key pair with public exponent e (generic exponent ladder on the GPU)."""
    from corpus.keys import DRBG, gen_prime
    rng = DRBG("rsa-e", e, idx)
    while True:
        p, q = gen_prime(1024, rng), gen_prime(1024, rng)
        phi = (p - 1) * (q - 1)
        if p != q and (p * q).bit_length() == 2048 and np.gcd(e, phi % e if phi % e else e) == 1 and phi % e != 0:
            break
    kp = cb.make_keypair(cb.PK_RSA, {"p": p, "q": q, "e": e}, "e%d-%d <k@bftkv.example>" % (e, idx))
    cb.build_entity(kp, [], rng)
    return kp


def test_public_exponents_and_value_ranges(gpu_ctx):
    """Exponent ladder classes in one wave (e = 3, 17, 257, 65537), signature values >= n (Go <= 1.13 has no s < n
    check: s + n verifies when it fits the MPI), MPIs with leading zero bytes / non-canonical bit counts."""
    from oracle import collective as col
    from oracle import openpgp as pgp
    from oracle.packet import SignaturePacket
    keys = [_rsa_key_with_e(e, i) for i, e in enumerate((3, 17, 257, 65537, 3))]
    ents = [pgp.read_entities(k.entity)[0] for k in keys]
    kr = col.Keyring(keyring=ents)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    rng = np.random.default_rng(21)
    tbs_l, sig_l = [], []
    for i in range(60):
        kp = keys[i % len(keys)]
        tbs = rng.bytes(int(rng.integers(0, 300)))
        pkt = bytearray(cb.detach_sign(kp, tbs))
        hdr = 3
        mpi_at = hdr + 6 + 16 + 2 + 2            # body prefix (22) + unhashed len + tag
        s = int.from_bytes(pkt[mpi_at + 2:], "big")
        variant = i % 6
        if variant == 1:                          # s + n still fits 256 bytes for about half the keys
            if s + kp.n < 1 << 2048:
                pkt[mpi_at + 2:] = (s + kp.n).to_bytes(256, "big")
        elif variant == 2:                        # 257-byte MPI with a leading zero byte: same value
            body = bytes(pkt[hdr:mpi_at]) + (2056).to_bytes(2, "big") + b"\x00" + s.to_bytes(256, "big")
            pkt = bytearray(cb._hdr(2, len(body)) + body)
        elif variant == 3:                        # value >= 2^(8k): s + n*2^8-ish does not verify, but takes the no-shortcut path
            big = s + kp.n * 3
            body = bytes(pkt[hdr:mpi_at]) + (big.bit_length()).to_bytes(2, "big") + big.to_bytes((big.bit_length() + 7) // 8, "big")
            pkt = bytearray(cb._hdr(2, len(body)) + body)
        elif variant == 4:                        # canonical bit count instead of Go's 8*len
            pkt[mpi_at:mpi_at + 2] = s.bit_length().to_bytes(2, "big")
            pkt[mpi_at + 2:] = s.to_bytes((s.bit_length() + 7) // 8, "big")
            body = bytes(pkt[hdr:])
            pkt = bytearray(cb._hdr(2, len(body)) + body)
        elif variant == 5:
            pkt[-1] ^= 1
        tbs_l.append(tbs)
        sig_l.append(bytes(pkt))
    tb, to = _cat(tbs_l)
    sb, so = _cat(sig_l)
    err = gpu_ctx.signature_verify(tb, to, sb, so)
    n_ok = 0
    for i, (t, s_) in enumerate(zip(tbs_l, sig_l)):
        want = col.signature_verify(kr, t, SignaturePacket(1, 0, False, s_, None))
        assert (err[i] == 0) == (want is None), (i, i % 6, keys[i % 5].e, err[i], want)
        n_ok += want is None
    assert 30 < n_ok < 60


def test_hashed_area_variants(gpu_ctx):
    """Signature.parse corner cases on signatures we can sign ourselves: extra / unknown / critical
    subpackets, two-octet and five-octet subpacket lengths, issuer only in the unhashed area, two issuer subpackets,
    creation time missing from the hashed area, embedded signatures."""
    import hashlib
    import struct
    from oracle import collective as col
    from oracle import openpgp as pgp
    from oracle.packet import SignaturePacket
    cl = cb.make_cluster(4)
    kr = _ring_and_ctx(gpu_ctx, cl)
    kp, other = cl.replicas[0], cl.replicas[1]
    ct = b"\x05\x02" + struct.pack(">I", cb.CREATION_TIME)
    iss = lambda k: b"\x09\x10" + struct.pack(">Q", k.key_id)
    notation = lambda n: bytes([192 + ((n + 1 - 192) >> 8), (n + 1 - 192) & 0xFF, 20]) + b"n" * n if n + 1 >= 192 else bytes([n + 1, 20]) + b"n" * n
    inner = cb.detach_sign(other, b"inner")[3:]          # an embedded signature body (type 0x00: not a cross-certification)
    inner19 = cb.sig_prefix(0x19, other.algo, ct + iss(other)) + b"\x00\x00\xab\xcd" + cb.go_mpi_bytes(b"\x5a" * 256)

    def sub(typ, body):
        n = len(body) + 1
        return (bytes([n]) if n < 192 else bytes([((n - 192) >> 8) + 192, (n - 192) & 0xFF])) + bytes([typ]) + body
    cases = {
        "plain": (ct + iss(kp), b""),
        "notation": (ct + notation(30) + iss(kp), b""),
        "two-octet-len": (ct + notation(300) + iss(kp), b""),
        "five-octet-len": (ct + b"\xff" + struct.pack(">I", 41) + bytes([20]) + b"x" * 40 + iss(kp), b""),
        "unknown-critical": (ct + b"\x02\xe5\x01" + iss(kp), b""),
        "unknown-noncritical": (ct + b"\x02\x65\x01" + iss(kp), b""),
        "issuer-unhashed": (ct, iss(kp)),
        "issuer-twice": (ct + iss(other) + iss(kp), b""),
        "issuer-twice-rev": (ct + iss(kp) + iss(other), b""),
        "no-issuer": (ct, b""),
        "ctime-unhashed-only": (iss(kp), ct),
        "bad-ctime-len": (b"\x04\x02\x00\x00\x01" + iss(kp), b""),
        "zero-len-subpacket": (ct + b"\x00" + iss(kp), b""),
        "embedded-sig": (ct + sub(32, inner) + iss(kp), b""),
        "embedded-garbage": (ct + b"\x05\x20abcd" + iss(kp), b""),
        # the cross-certification shape gpg emits for signing subkeys: a 0x19 signature, usually in the UNHASHED area
        "embedded-0x19": (ct + sub(32, inner19) + iss(kp), b""),
        "embedded-0x19-unhashed": (ct + iss(kp), sub(32, inner19)),
        "embedded-twice": (ct + sub(32, inner19) + iss(kp), sub(32, inner19)),
        "embedded-wrong-type-unhashed": (ct + iss(kp), sub(32, inner)),
        "ctime-both-areas": (ct + iss(kp), ct),
        "key-flags+expiry": (ct + b"\x02\x1b\x03" + b"\x05\x03\x00\x00\x10\x00" + b"\x05\x09\x00\x00\x20\x00" + iss(kp), b""),
        "truncated-subpacket": (ct + b"\x30\x14abc", b""),
    }
    tbs = b"payload of the hashed-area test"
    tbs_l, sig_l, names = [], [], []
    for name, (hashed, unhashed) in cases.items():
        prefix = cb.sig_prefix(0x00, kp.algo, hashed)
        digest = hashlib.sha256(tbs + cb.hash_suffix(prefix)).digest()
        s = kp.rsa_private(cb.emsa(digest, 256))
        body = prefix + struct.pack(">H", len(unhashed)) + unhashed + digest[:2] + cb.go_mpi_bytes(s.to_bytes(256, "big"))
        names.append(name); tbs_l.append(tbs); sig_l.append(cb._hdr(2, len(body)) + body)
    tb, to = _cat(tbs_l)
    sb, so = _cat(sig_l)
    err = gpu_ctx.signature_verify(tb, to, sb, so)
    st, st_item = gpu_ctx.last_statuses()
    outcome = {}
    for i, name in enumerate(names):
        sp = SignaturePacket(1, 0, False, sig_l[i], None)
        tr = []
        want = col.signature_verify(kr, tbs, sp, trace=tr)
        assert (err[i] == 0) == (want is None), (name, err[i], want)
        assert list(st[st_item == i]) == tr, (name, list(st[st_item == i]), tr)
        outcome[name] = want is None
    assert outcome["plain"] and outcome["notation"] and outcome["two-octet-len"] and outcome["five-octet-len"] and outcome["issuer-unhashed"]
    assert outcome["issuer-twice"] and not outcome["issuer-twice-rev"] and outcome["unknown-noncritical"] and outcome["key-flags+expiry"]
    assert not outcome["unknown-critical"] and not outcome["no-issuer"] and not outcome["ctime-unhashed-only"] and not outcome["embedded-garbage"]
    assert outcome["embedded-0x19"] and outcome["embedded-0x19-unhashed"] and not outcome["embedded-twice"]
    assert not outcome["embedded-wrong-type-unhashed"] and not outcome["ctime-both-areas"] and not outcome["embedded-sig"]
    assert not gpu_ctx.last_fenced.any()


def test_fenced_shapes_are_flagged_and_nothing_else_is(gpu_ctx):
    """include/bftkv_gpu.h "fenced inputs": shapes the reference accepts and the kernels do not follow raise the item's
    fenced_out flag (the shim then takes the reference path); ordinary failures -- bad values, unknown issuers, hash-tag
    mismatches, parse errors -- never do.  The fenced packets here are VALID signatures by clique members, i.e. exactly the
    inputs on which a silent verdict would disagree with a Go replica."""
    import hashlib
    import struct
    cl = cb.make_cluster(4)
    kr = _ring_and_ctx(gpu_ctx, cl)
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    tbs = b"the signed payload of the fence test"
    good = [cb.detach_sign(r, tbs) for r in cl.replicas]
    kp = cl.replicas[0]
    ct = b"\x05\x02" + struct.pack(">I", cb.CREATION_TIME)
    iss = b"\x09\x10" + struct.pack(">Q", kp.key_id)

    def v4(sig_type=0, hash_id=8, value=None, hashed=None, h=hashlib.sha256, nest=None):
        hashed = (ct + iss) if hashed is None else hashed
        prefix = bytes([4, sig_type, kp.algo, hash_id]) + struct.pack(">H", len(hashed)) + hashed
        digest = h(tbs + cb.hash_suffix(prefix)).digest()
        sval = kp.rsa_private(cb.emsa(digest, 256)) if value is None else value
        sb_ = sval.to_bytes(max(256, (sval.bit_length() + 7) // 8), "big")
        return prefix + b"\x00\x00" + digest[:2] + cb.go_mpi_bytes(sb_)
    # SignatureV3 (RFC 4880 5.2.2): hashed material = type || creation time, no trailer
    d3 = hashlib.sha256(tbs + bytes([0]) + struct.pack(">I", cb.CREATION_TIME)).digest()
    v3 = (bytes([3, 5, 0]) + struct.pack(">I", cb.CREATION_TIME) + struct.pack(">Q", kp.key_id) + bytes([kp.algo, 8]) + d3[:2] +
          cb.go_mpi_bytes(kp.rsa_private(cb.emsa(d3, 256)).to_bytes(256, "big")))
    body = v4()
    partial = bytes([0xC2, 0xE0 + 7]) + body[:128] + cb._hdr(2, len(body) - 128)[1:] + body[128:]      # 2^7-byte first chunk
    indeterminate = bytes([0x80 | (2 << 2) | 3]) + body
    deep = v4()
    for _ in range(3):                                                                             # embedded nesting depth 3
        n = len(deep) + 1
        enc = bytes([((n - 192) >> 8) + 192, (n - 192) & 0xFF]) if n >= 192 else bytes([n])
        hashed = ct + iss + enc + bytes([32]) + deep
        deep = bytes([4, 0x19, kp.algo, 8]) + struct.pack(">H", len(hashed)) + hashed + b"\x00\x00\x00\x00" + cb.go_mpi_bytes(b"\x01" * 256)
    def pow2_chunks(b):                                   # the whole body in partial-length chunks, largest first
        out = bytes([0xC2])
        while b:
            k = len(b).bit_length() - 1
            out += bytes([0xE0 + k]) + b[:1 << k]
            b = b[1 << k:]
        return out
    fenced_cases = {
        # a signature that parses while 5000 bytes of its packet were never fetched by bufio: the next call starts inside it
        "bytes-behind-the-mpis": bytes([0xC2, 255]) + (len(body) + 5000).to_bytes(4, "big") + body + bytes(5000),
        # partial lengths whose zero-length last chunk is never read: the next call trips over its length octet
        "partial-length-zero-last-chunk": pow2_chunks(body) + b"\x00",
        # past the bounds of the native path for partial lengths (kernels.hip CHAIN_MAX_HOPS / CHUNKED_SIG_MAX_BODY): not claimed
        "partial-length-1100-chunks": bytes([0xC0 | 60]) + b"".join(bytes([0xE0, 7]) for _ in range(1100)) + b"\x00",
        "partial-length-17000-byte-body": pow2_chunks(v4(hashed=ct + iss + bytes([255]) + (17000).to_bytes(4, "big") + bytes([100]) + bytes(16999))),
        "md5": cb._hdr(2, len(v4(hash_id=1, h=hashlib.md5))) + v4(hash_id=1, h=hashlib.md5),
        "value-beyond-R": cb._hdr(2, len(v4(value=kp.rsa_private(5) + (1 << 2200)))) + v4(value=kp.rsa_private(5) + (1 << 2200)),
        "nesting-3": cb._hdr(2, len(deep)) + deep,
    }
    bad_mpi = bytearray(good[1]); bad_mpi[-7] ^= 2
    bad_tag = bytearray(good[2]); bad_tag[len(bad_tag) - 260] ^= 0x40
    v3_bad = bytearray(v3); v3_bad[-9] ^= 0x20
    plain_cases = {
        "all-good": b"".join(good),
        # SignatureV3 packets are verified natively (gpg accepts this construction: tests/golden/gpg_negative_vectors.json)
        "v3-good": cb._hdr(2, len(v3)) + v3 + good[1] + good[2],
        "v3-bad-value": cb._hdr(2, len(v3)) + bytes(v3_bad) + good[1] + good[2],
        "v3-version-1": cb._hdr(2, len(v3)) + b"\x01" + v3[1:] + b"".join(good),
        "v3-hashed-length-not-5": cb._hdr(2, len(v3)) + v3[:1] + b"\x06" + v3[2:] + b"".join(good),
        "bad-value": bytes(bad_mpi) + good[0] + good[2] + good[3],
        "bad-tag": bytes(bad_tag) + good[0],
        "unknown-issuer": cb.detach_sign(cl.outsiders[0], tbs) + b"".join(good),
        "garbage": b"\x01\x02\x03",
        "unknown-packet-type": bytes([0xC0 | 60, 3]) + b"abc" + b"".join(good),
        "empty": b"",
        "non-0x00-type": cb._hdr(2, len(v4(sig_type=2))) + v4(sig_type=2),          # hashForSignature refuses it in the reference too
        # text mode over a payload without line ends: the canonical form IS the payload, the signature verifies
        "text-mode": cb._hdr(2, len(v4(sig_type=1))) + v4(sig_type=1) + good[1] + good[2] + good[3],
        "truncated": good[0][:100],
        # partial body lengths, every chunk fetched whole: verified like any other packet (fenced until round 3)
        "partial-length": partial + good[1] + good[2] + good[3],
        # old-format indeterminate length: the rest of the stream IS the body -- the signature verifies, what follows is swallowed
        "indeterminate-length": indeterminate + good[1] + good[2] + good[3],
        "indeterminate-length-last": good[1] + good[2] + good[3] + indeterminate,
        # a declared length beyond the end of the stream: the signature is all there and verifies
        "length-past-the-end": good[1] + good[2] + bytes([0xC2, 255]) + (len(body) + 777).to_bytes(4, "big") + body,
    }
    names = list(fenced_cases) + list(plain_cases)
    ss_l = [fenced_cases[n] + good[1] + good[2] + good[3] for n in fenced_cases] + [plain_cases[n] for n in plain_cases]
    tb, to = _cat([tbs] * len(ss_l))
    sb, so = _cat(ss_l)
    for early in (True, False):
        gpu_ctx.set_early_exit(early)
        err, nver, _ = gpu_ctx.collective_verify(qh, tb, to, sb, so)
        fenced = gpu_ctx.last_fenced.copy()
        for i, n in enumerate(names):
            assert fenced[i] == (1 if n in fenced_cases else 0), (n, early, fenced[i])
            if not fenced[i]:                                  # un-fenced items carry the reference's verdict
                r = H.oracle_collective(kr, q, type("C", (), {"tbss": lambda self, i: tbs, "ss_data": lambda self, i: ss_l[i]})(), i)
                assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified), (n, early)
    gpu_ctx.set_early_exit(True)
    # Signature.Verify reports the same flags
    e1 = gpu_ctx.signature_verify(tb, to, sb, so)
    assert list(gpu_ctx.last_fenced) == [1 if n in fenced_cases else 0 for n in names] and len(e1) == len(names)
    gpu_ctx.quorum_destroy(qh)


def test_two_keys_under_one_key_id(gpu_ctx, exit_mode):
    """Key ids are 64 bits of a SHA-1: two different keys can be made to share one.  The reference asks every candidate in
    keyring order -- each through the SAME hash object, into which every VerifySignature that gets that far writes the hash
    suffix again -- and returns the first success or the last error.  So: the first candidate that can sign is the only one
    that can succeed; a genuine signer listed behind a twin that can sign is REFUSED (hash tag mismatch); a candidate that
    cannot sign is passed over without harm.  Verdicts, exit counts and per-packet statuses are the oracle's; nothing is
    fenced but the one shape in which a LATER candidate could succeed (its tag and algorithm fit the many-suffix digest)."""
    import copy
    from oracle import collective as col
    from oracle import openpgp as pgp
    from oracle.packet import SignaturePacket
    cl = cb.make_cluster(5, dsa_fraction=0.0)
    cld = cb.make_cluster(3, dsa_fraction=1.0, seed=cb.MASTER_SEED + 7)
    base = H.oracle_keyring(cl).get_keyring()
    id0 = cl.replicas[0].key_id

    def twin_of(ent, algo=None):
        t = copy.deepcopy(ent)
        t.primary = copy.deepcopy(ent.primary)
        t.primary.key_id = id0                       # another key's material under replica 0's id
        if algo is not None:
            t.primary.pk_algo = algo
        return t
    tbs = b"collision"
    s = [cb.detach_sign(r, tbs) for r in cl.replicas]
    streams = [s[0] + s[1] + s[2] + s[3], s[1] + s[2] + s[3] + s[4], s[3] + s[0] + s[2] + s[1], s[0], s[4] + s[0] + s[0]]
    # a packet whose 16-bit hash tag is that of the digest with the suffix written TWICE: the genuine key refuses it (tag), the
    # twin behind it gets past the tag and fails on the algorithm (DSA twin) or on the arithmetic (RSA twin)
    import hashlib
    pk = bytearray(s[0])
    hdr = 3 if pk[1] >= 192 else 2
    hl = (pk[hdr + 4] << 8) | pk[hdr + 5]
    suffix = cb.hash_suffix(bytes(pk[hdr:hdr + 6 + hl]))
    pk[hdr + 6 + hl + 2:hdr + 6 + hl + 4] = hashlib.sha256(tbs + suffix + suffix).digest()[:2]
    streams.append(bytes(pk) + s[1] + s[2] + s[3])
    # ... and the packet the owner of the twin key (replica 1's material under replica 0's id) can make: signed over the digest
    # with the suffix written twice.  Behind the genuine key the twin is candidate 1, sees exactly that digest, and the reference
    # returns it as the signer.  The device does not repeat the arithmetic for later candidates: it must FENCE.
    twin_prefix = cb.sig_prefix(0x00, cl.replicas[1].algo, cb._hashed_area(id0))
    twin_sfx = cb.hash_suffix(twin_prefix)
    streams.append(cb.make_sig_packet(cl.replicas[1], twin_prefix, hashlib.sha256(tbs + twin_sfx + twin_sfx).digest()) + s[2] + s[3] + s[4])
    FORGED_TAG, TWIN_SIGNED = len(streams) - 2, len(streams) - 1
    dsa_ent = pgp.read_entities(cld.replicas[0].entity)[0]
    scenarios = {
        "twin behind the genuine key": base + [twin_of(base[1])],
        "DSA twin behind the genuine key": base + [twin_of(dsa_ent)],
        "twin ahead of the genuine key": [twin_of(base[1])] + base,
        "encrypt-only twin ahead": [twin_of(base[2], algo=2)] + base,
        "DSA twin ahead": [twin_of(dsa_ent)] + base,
        "two twins around the genuine key": [twin_of(base[2], algo=2)] + base[:1] + [twin_of(base[3])] + base[1:],
        "only keys that cannot sign": [twin_of(base[2], algo=2), twin_of(base[3], algo=2)] + base[1:],
    }
    q = H.clique_quorum(cl)
    seen = set()
    for name, ring in scenarios.items():
        kr = col.Keyring(keyring=ring)
        gpu_ctx.keyring_set(H.abi_keys(kr))
        qh = gpu_ctx.quorum_create(H.abi_qcs(q))
        tb, to = _cat([tbs] * len(streams))
        sb, so = _cat(streams)
        err, nver, _ = gpu_ctx.collective_verify(qh, tb, to, sb, so)
        st, st_item = gpu_ctx.last_statuses()
        fenced = gpu_ctx.last_fenced.copy()
        # a fence only where a LATER candidate gets past the tag and the algorithm check with its many-suffix digest
        assert not fenced[:FORGED_TAG].any(), name
        twin_won = False
        for i, data in enumerate(streams):
            r = col.collective_verify(kr, tbs, SignaturePacket(Type=1, Data=data), q)
            seen.update(r.statuses)
            if i == TWIN_SIGNED and r.statuses[0] == 0:
                twin_won = True
                assert fenced[i], (name, "the twin's own signature verifies in the reference: the device must hand it over")
            if fenced[i]:
                continue
            assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified), (name, i, err[i], nver[i], r.err, r.verified)
            got = list(st[st_item == i])
            assert got[:len(r.statuses)] == r.statuses, (name, i, got, r.statuses)
        if name == "twin behind the genuine key":
            assert twin_won and fenced[FORGED_TAG] and fenced[TWIN_SIGNED]
        if name == "DSA twin behind the genuine key":
            assert not fenced.any()          # tag fits, the algorithm does not: ST_ALGO_MISMATCH in both, nothing to hand over
        # Signature.Verify over the same rings (every packet must verify: the last error decides)
        sig_err = gpu_ctx.signature_verify(tb, to, sb, so)
        sig_fenced = gpu_ctx.last_fenced.copy()
        for i, data in enumerate(streams):
            want = col.signature_verify(kr, tbs, SignaturePacket(1, 0, False, data, None))
            assert sig_fenced[i] or (sig_err[i] == 0) == (want is None), (name, i)
        assert (sig_fenced == fenced).all(), name
        gpu_ctx.quorum_destroy(qh)
    assert {0, 6, 7, 8, 9} <= seen      # ok, hash tag, algorithm mismatch, bad signature, key cannot sign: all were produced
    gpu_ctx.keyring_set(H.abi_keys(H.oracle_keyring(cl)))


def test_batcher_fails_closed(gpu_ctx):
    """An infrastructure error must never read as "verified": the status byte is a failure whenever the return code is
    not 0 (the shim looked at the status only, and 0 means nil error)."""
    from bftkv_amd._native import Batcher
    cl = cb.make_cluster(4)
    _ring_and_ctx(gpu_ctx, cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(H.clique_quorum(cl)))
    tbs = b"x"
    ss = b"".join(cb.detach_sign(r, tbs) for r in cl.replicas)
    b = Batcher(gpu_ctx, max_items=8, max_wait_us=100)
    try:
        assert b.collective_verify(qh, tbs, ss, raw=True) == (0, 0, 0)
        rc, err, _ = b.collective_verify(qh + 1000, tbs, ss, raw=True)          # no such quorum: the device call fails
        assert rc != 0 and err == 2
        gpu_ctx.quorum_destroy(qh)
        rc, err, _ = b.collective_verify(qh, tbs, ss, raw=True)                 # handle destroyed under the caller
        assert rc != 0 and err == 2
        rc, err, _ = b.signature_verify(tbs, cb.detach_sign(cl.replicas[0], tbs), raw=True)
        assert (rc, err) == (0, 0)
        rc, err, fenced = b.signature_verify(tbs, b"", raw=True)
        assert rc == 0 and err == 1 and fenced == 0
    finally:
        b.close()


def test_verdict_bitmap_pack_unpack_on_device(gpu_ctx):
    """bftkv_amd.dist: the verdict bitmap that is all-gathered over RCCL packs / unpacks identically on the GPU and on the CPU."""
    import torch
    from bftkv_amd import dist as D
    rng = np.random.default_rng(3)
    for n in (1, 7, 8, 9, 1000, 10000):
        ok = rng.integers(0, 2, size=n).astype(np.uint8)
        t = torch.from_numpy(ok).to("cuda:0")
        slots = D.max_shard(n, 1)
        bits = D.pack_verdicts(t, slots)
        assert bits.cpu().numpy().tobytes() == np.packbits(ok, bitorder="little").tobytes()
        assert (D.unpack_verdicts(bits, n).cpu().numpy() == ok).all()
        assert (D.allgather_verdicts(t == 1, n).cpu().numpy() == ok).all()


def test_random_packet_framings_follow_the_oracle(gpu_ctx, exit_mode):
    """The packet walk speculates on packet positions (k_walk): random streams mixing signature packets of different sizes,
    unknown and non-signature packets in every header format, stray bytes, truncations and more than WALK_CAP events per
    item must still yield the oracle's packet sequence, statuses, early-exit position and verdict."""
    from oracle import collective as col
    from oracle.packet import SignaturePacket
    cl = cb.make_cluster(7, dsa_fraction=0.3)
    kr = _ring_and_ctx(gpu_ctx, cl)
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    tbs_l, ss_l, hdr, srng = H.random_framing_streams(cl, 160)
    # a packet too long for the 24-bit length field of the walk's scratch row: the item takes the direct-record path
    tbs = b"big packet in the stream"
    sigs = [cb.detach_sign(r, tbs, srng) for r in cl.replicas]
    tbs_l.append(tbs)
    ss_l.append(sigs[0] + hdr(13, (1 << 24) + 5, 1) + bytes((1 << 24) + 5) + b"".join(sigs[1:]))
    n_all = len(tbs_l)
    tb, to = _cat(tbs_l)
    sb, so = _cat(ss_l)
    err, nver, verdict = gpu_ctx.collective_verify(qh, tb, to, sb, so)
    st, st_item = gpu_ctx.last_statuses()
    # fenced exactly where the reference's reader position after a packet depends on the packet type's parser (a literal-data
    # or key packet among the signatures: oracle.openpgp.position_is_type_dependent); statuses still follow the oracle's walk
    from oracle import openpgp as pgp
    want_fenced = [1 if pgp.position_is_type_dependent(s_) else 0 for s_ in ss_l]
    assert list(gpu_ctx.last_fenced) == want_fenced and 10 < sum(want_fenced) < len(ss_l) - 40
    n_long = 0
    r_big = col.collective_verify(kr, tbs_l[-1], SignaturePacket(1, 0, False, ss_l[-1], None), q)
    assert list(st[st_item == n_all - 1])[:len(r_big.statuses)] == r_big.statuses and (err[-1] == 0) == (r_big.err is None)
    assert 3 in list(st[st_item == n_all - 1])         # the oversized user-id packet is an event (not a signature)
    for i in range(160):
        r = col.collective_verify(kr, tbs_l[i], SignaturePacket(1, 0, False, ss_l[i] or None, None), q)
        got = list(st[st_item == i])
        assert _events_until_unread_signature(pgp, ss_l[i]) is None        # (these streams are fenced for lazy parsers only: the
        assert got[:len(r.statuses)] == r.statuses, (i, got[:12], r.statuses[:12])   # oracle takes those bodies whole, like the walk)
        assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified), i
        n_long += len(got) > 96
    assert n_long >= 8 and (err == 0).any() and (err != 0).any()
    # Signers (parse-only walk) over the same streams
    ids, off = gpu_ctx.signers(sb, so)
    sf = gpu_ctx.last_fenced.copy()
    # the same shapes raise the fence of the parse-only walk, as far as that walk gets (its first error ends it)
    assert list(sf) == [1 if pgp.position_is_type_dependent(s_, stop_at_error=True) else 0 for s_ in ss_l]
    for i in range(0, 160, 9):
        want = col.signers(kr, SignaturePacket(1, 0, False, ss_l[i] or None, None))
        assert [int(x) for x in ids[int(off[i]):int(off[i + 1])]] == want, i
    gpu_ctx.quorum_destroy(qh)


def _events_until_unread_signature(pgp, stream):
    """Number of packet events up to and including the first signature that parses while part of its packet stays unread (the
    oracle follows the reference's reader into that packet, the verifier does not): None when the stream has none."""
    pos = n = 0
    while True:
        pk = pgp.packet_read_stream(stream, pos)
        pos = pk.pos
        if pk.kind == "eof":
            return None
        if pk.kind == "unknown":
            continue
        n += 1
        if pk.kind == "sig" and pk.body_unread:
            return n


def test_exotic_framings_follow_the_reference_readers(gpu_ctx, exit_mode):
    """The framings x/crypto reads and no writer of the path produces -- partial body lengths (signature, unknown and user-id
    packets; a zero-length last chunk), indeterminate lengths, lengths past the end of the stream, bodies beyond bufio's 4096
    bytes, cut at random places -- around VALID signatures: verdicts, exit counts and per-packet statuses are the oracle's
    (whose reader objects restate packet.Read literally), the fence goes up exactly where a parsed signature leaves the
    reference's reader inside its packet, and up to that packet the statuses still agree.  Chunked bodies are linearised by the
    parse (k_parse_body) and walked in place by the parse-only kernel (k_signers)."""
    from oracle import collective as col
    from oracle import openpgp as pgp
    from oracle.packet import SignaturePacket
    cl = cb.make_cluster(5, dsa_fraction=0.4)
    kr = _ring_and_ctx(gpu_ctx, cl)
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    tbs_l, ss_l = H.exotic_framing_streams(cl, 320)
    tb, to = _cat(tbs_l)
    sb, so = _cat(ss_l)
    err, nver, verdict = gpu_ctx.collective_verify(qh, tb, to, sb, so)
    st, st_item = gpu_ctx.last_statuses()
    fenced = gpu_ctx.last_fenced.copy()
    want_fenced = [1 if pgp.position_is_type_dependent(s_) else 0 for s_ in ss_l]
    assert list(fenced) == want_fenced and 60 < sum(want_fenced) < 200
    n_chunked_ok = 0
    for i in range(len(ss_l)):
        r = col.collective_verify(kr, tbs_l[i], SignaturePacket(1, 0, False, ss_l[i] or None, None), q)
        got = list(st[st_item == i])
        assert pgp.fence_reason(ss_l[i]) != "bounds"
        k = _events_until_unread_signature(pgp, ss_l[i])
        want = r.statuses if k is None else r.statuses[:k]
        assert got[:len(want)] == want, (i, got[:10], want[:10])
        if not fenced[i]:
            assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified), i
            n_chunked_ok += sum(1 for x in r.statuses if x == 0)
    assert n_chunked_ok > 200 and (err == 0).any()
    # PGPSignature.Verify (every call must succeed) over the same streams
    e1 = gpu_ctx.signature_verify(tb, to, sb, so)
    assert list(gpu_ctx.last_fenced) == want_fenced
    for i in range(0, len(ss_l), 3):
        if not want_fenced[i]:
            assert (e1[i] == 0) == (col.signature_verify(kr, tbs_l[i], SignaturePacket(1, 0, False, ss_l[i] or None, None)) is None), i
    # Signers: the parse-only walk, chunked bodies read through the chunk-walking view
    ids, off = gpu_ctx.signers(sb, so)
    sf = gpu_ctx.last_fenced.copy()
    assert list(sf) == [1 if pgp.position_is_type_dependent(s_, stop_at_error=True) else 0 for s_ in ss_l]
    n_ids = 0
    for i in range(len(ss_l)):
        if sf[i]:
            continue
        want = col.signers(kr, SignaturePacket(1, 0, False, ss_l[i] or None, None))
        assert [int(x) for x in ids[int(off[i]):int(off[i + 1])]] == want, i
        n_ids += len(want)
    assert n_ids > 150
    gpu_ctx.quorum_destroy(qh)


def test_reader_model_at_volume_on_the_device(gpu_ctx, exit_mode):
    """1,200 streams of plausible UNSIGNED signature bodies in every framing (tests/helpers.py plausible_unsigned_streams), issuers
    known and unknown: nothing verifies, so this is the device's walk (speculating lanes, chunk chains, the sequential fill pass),
    its parse (linearised chunks, LDS windows) and its reader-position rule against the oracle's reader objects -- per-packet
    statuses up to the first fenced packet, exit counts, error bytes and the fence flags."""
    from oracle import collective as col
    from oracle import openpgp as pgp
    from oracle.packet import SignaturePacket
    cl = cb.make_cluster(5, dsa_fraction=0.4)
    kr = _ring_and_ctx(gpu_ctx, cl)
    q = H.clique_quorum(cl)
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    issuers = [cl.replicas[0].key_id, cl.replicas[1].key_id, 0x1122334455667788]
    ss_l = list(H.plausible_unsigned_streams(1200, seed=7, issuers=issuers))
    tbs = b"payload"
    tb, to = _cat([tbs] * len(ss_l))
    sb, so = _cat(ss_l)
    err, nver, _ = gpu_ctx.collective_verify(qh, tb, to, sb, so)
    st, st_item = gpu_ctx.last_statuses()
    fenced = gpu_ctx.last_fenced.copy()
    order = np.argsort(st_item, kind="stable")
    first = np.searchsorted(st_item[order], np.arange(len(ss_l) + 1))
    n_cmp = 0
    rsa_ids = {r_.key_id for r_ in cl.replicas if r_.algo == cb.PK_RSA}

    def semantic_fence(stream):
        """Fences that are about a signature's content, not its framing: an RSA value beyond R under a known RSA
        key (512- and 8000-byte MPIs here), MD5 / RIPEMD-160 under the default 'unknown' policy (a mutated hash byte)."""
        pos = 0
        while True:
            pk = pgp.packet_read_stream(stream, pos)
            pos = pk.pos
            if pk.kind == "eof" or pk.beyond_native_bounds or (pk.kind == "not_signature" and pk.lazy_parser):
                return False
            if pk.kind == "sig":
                g = pk.sig
                if g.issuer in rsa_ids and g.pk_algo in (1, 3) and len(g.mpis[0][1].lstrip(b"\0")) > 266:
                    return True
                if g.issuer is not None and (g.issuer in rsa_ids or any(g.issuer == r_.key_id for r_ in cl.replicas)) and g.hash_id in (1, 3) and g.sig_type in (0, 1):
                    return True
                if pk.body_unread:
                    return False

    for i, s_ in enumerate(ss_l):
        reason = pgp.fence_reason(s_)
        assert fenced[i] == (1 if (reason or semantic_fence(s_)) else 0), (i, reason, fenced[i], s_[:48].hex())
        r = col.collective_verify(kr, tbs, SignaturePacket(1, 0, False, s_ or None, None), q)
        got = list(st[order[first[i]:first[i + 1]]])
        if reason == "bounds":
            continue
        k = _events_until_unread_signature(pgp, s_)
        want = r.statuses if k is None else r.statuses[:k]
        assert got[:len(want)] == want, (i, got[:10], want[:10], s_[:48].hex())
        if not fenced[i]:
            assert err[i] != 0 and nver[i] == 0 and len(got) == len(r.statuses), i
        n_cmp += len(want)
    assert n_cmp > 5000 and 200 < fenced.sum() < 1000
    gpu_ctx.quorum_destroy(qh)


def test_mixed_modulus_sizes_2048_3072_4096(gpu_ctx):
    """One batch whose signers hold RSA-2048, -3072 and -4096 keys (own work lists, 4 / 8 / 8 lanes per number): verdicts and
    per-packet statuses follow the oracle, including values >= n, long MPIs, e = 3 on a 3072-bit key and corrupted values."""
    from corpus.keys import DRBG, load_keys
    from oracle import collective as col
    from oracle import openpgp as pgp
    from oracle import wotqs as W
    from oracle.packet import SignaturePacket
    rng = DRBG("mixed-sizes")
    keys = []
    for kind, cnt in (("rsa2048", 2), ("rsa3072", 2), ("rsa4096", 2)):
        for i, mat in enumerate(load_keys(kind, cnt)):
            kp = cb.make_keypair(cb.PK_RSA, mat, "%s-%d <k@bftkv.example>" % (kind, i))
            cb.build_entity(kp, [], rng)
            keys.append(kp)
    # a 3072-bit key with e = 3 (generic exponent ladder on the 8-lane kernel)
    m3 = load_keys("rsa3072", 2)[1]
    if ((m3["p"] - 1) * (m3["q"] - 1)) % 3 != 0:
        kp = cb.make_keypair(cb.PK_RSA, {"p": m3["p"], "q": m3["q"], "e": 3}, "rsa3072-e3 <k@bftkv.example>")
        cb.build_entity(kp, [], rng)
        keys[3] = kp
    ents = [pgp.read_entities(k.entity)[0] for k in keys]
    kr = col.Keyring(keyring=ents)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    q = W.WotQ([W.new_qc([k.key_id for k in keys], len(keys), W.AUTH, 0)])
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    nrng = np.random.default_rng(33)
    tbs_l, ss_l = [], []
    for i in range(40):
        tbs = nrng.bytes(int(nrng.integers(0, 200)))
        parts = []
        for j, kp in enumerate(keys):
            pkt = bytearray(cb.detach_sign(kp, tbs))
            k = (kp.n.bit_length() + 7) // 8
            hdr = len(pkt) - (k + 2 + 26)              # header length: body = 22 prefix + 2 + 2 + (2 + k)
            mpi_at = hdr + 26
            s = int.from_bytes(pkt[mpi_at + 2:], "big")
            variant = (i + j) % 7
            if variant == 1 and s + kp.n < 1 << (8 * k):               # s + n verifies (no s < n check in Go <= 1.13)
                pkt[mpi_at + 2:] = (s + kp.n).to_bytes(k, "big")
            elif variant == 2:                                          # value longer than the modulus: not the x-shortcut path
                big = s + kp.n * 5
                body = bytes(pkt[hdr:mpi_at]) + big.bit_length().to_bytes(2, "big") + big.to_bytes((big.bit_length() + 7) // 8, "big")
                pkt = bytearray(cb._hdr(2, len(body)) + body)
            elif variant == 3:
                pkt[-2] ^= 0x04                                         # corrupted value
            elif variant == 4:                                          # leading zero byte in the MPI
                body = bytes(pkt[hdr:mpi_at]) + (8 * (k + 1)).to_bytes(2, "big") + b"\x00" + s.to_bytes(k, "big")
                pkt = bytearray(cb._hdr(2, len(body)) + body)
            parts.append(bytes(pkt))
        order = nrng.permutation(len(parts))
        tbs_l.append(tbs)
        ss_l.append(b"".join(parts[int(o)] for o in order))
    tb, to = _cat(tbs_l)
    sb, so = _cat(ss_l)
    err, nver, verdict = gpu_ctx.collective_verify(qh, tb, to, sb, so)
    st, st_item = gpu_ctx.last_statuses()
    seen = set()
    for i in range(40):
        r = col.collective_verify(kr, tbs_l[i], SignaturePacket(1, 0, False, ss_l[i], None), q)
        got = list(st[st_item == i])
        assert got[:len(r.statuses)] == r.statuses, (i, got, r.statuses)
        assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified), i
        seen.update(got)
    assert 0 in seen and 8 in seen
    cnt = gpu_ctx.last_counters()
    assert cnt["pubkey_ops"] <= 40 * len(keys)
    gpu_ctx.set_early_exit(False)
    try:
        err_all, nver_all, _ = gpu_ctx.collective_verify(qh, tb, to, sb, so)
        assert (err_all == err).all() and (nver_all == nver).all()
        assert gpu_ctx.last_counters()["pubkey_ops"] == 40 * len(keys)
    finally:
        gpu_ctx.set_early_exit(True)
    gpu_ctx.quorum_destroy(qh)


def test_dsa_batched_inversion_gives_the_per_signature_rows():
    """k_dsa_inv_batched (one extended GCD per run of signatures under a key, Montgomery's trick) against k_dsa_inv (one per
    signature) and the oracle: runs of every length up to a tile's, keys whose signatures are few (runs of one), range-check
    failures inside a run (r = 0, s = 0, r >= q, s >= q: dsa.Verify refuses before inverting) and corrupted values."""
    import os
    import hashlib
    from bftkv_amd import Context
    from corpus.keys import DRBG
    from oracle import collective as col
    from oracle.packet import SignaturePacket
    cl = cb.make_cluster(9, dsa_fraction=1.0)
    kr = H.oracle_keyring(cl)
    q = H.clique_quorum(cl)
    rng = np.random.default_rng(31)
    srng = DRBG("dsa-batched")

    def dsa_packet(kp, tbs, r=None, s=None):
        prefix = cb.sig_prefix(0x00, kp.algo, cb._hashed_area(kp.key_id))
        digest = hashlib.sha256(tbs + cb.hash_suffix(prefix)).digest()
        pkt = cb.make_sig_packet(kp, prefix, digest, srng)
        if r is None and s is None:
            return pkt
        body = bytearray(pkt[3:] if pkt[1] >= 192 else pkt[2:])
        p = len(prefix) + 4                          # unhashed length (2) + hash tag (2)
        mp = []
        for _ in range(2):
            nb = (int.from_bytes(body[p:p + 2], "big") + 7) // 8
            mp.append(int.from_bytes(body[p + 2:p + 2 + nb], "big")); p += 2 + nb
        r_, s_ = (mp[0] if r is None else r), (mp[1] if s is None else s)
        enc = b"".join(cb.go_mpi_bytes(v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")) for v in (r_, s_))
        nbody = bytes(body[:len(prefix) + 4]) + enc
        return cb._hdr(2, len(nbody)) + nbody

    tbs_l, ss_l = [], []
    for i in range(700):
        tbs = rng.bytes(int(rng.integers(1, 80)))
        # replicas 0..2 sign nearly every item (long runs), 3..5 about one in eight, 6..8 a handful of items (runs of one)
        who = [j for j in range(9) if (j < 3 and rng.random() < 0.95) or (3 <= j < 6 and rng.random() < 0.125) or (j >= 6 and i % 97 == j)]
        parts = []
        for j in rng.permutation(who):
            kp = cl.replicas[int(j)]
            m = int(rng.integers(0, 40))
            if m == 0: parts.append(dsa_packet(kp, tbs, r=0))
            elif m == 1: parts.append(dsa_packet(kp, tbs, s=0))
            elif m == 2: parts.append(dsa_packet(kp, tbs, r=kp.q))
            elif m == 3: parts.append(dsa_packet(kp, tbs, s=kp.q + 5))
            elif m == 4: parts.append(dsa_packet(kp, tbs, s=int(rng.integers(1, 1 << 62))))
            else: parts.append(dsa_packet(kp, tbs))
        tbs_l.append(tbs); ss_l.append(b"".join(parts))
    tb, to = _cat(tbs_l)
    sb, so = _cat(ss_l)
    got = {}
    for mode in ("single", "batched"):
        os.environ["BFTKV_DSA_INV"] = mode
        try:
            ctx = Context(0)
        finally:
            del os.environ["BFTKV_DSA_INV"]
        ctx.set_early_exit(False)
        ctx.keyring_set(H.abi_keys(kr))
        qh = ctx.quorum_create(H.abi_qcs(q))
        err, nver, _ = ctx.collective_verify(qh, tb, to, sb, so)
        st, st_item = ctx.last_statuses()
        got[mode] = (err.copy(), nver.copy(), st.copy(), st_item.copy())
        ctx.close()
    for a, b in zip(got["single"], got["batched"]):
        assert (a == b).all()
    err, nver, st, st_item = got["batched"]
    assert (st == 0).sum() > 1000 and (st != 0).sum() > 100
    for i in range(0, 700, 23):
        r = col.collective_verify(kr, tbs_l[i], SignaturePacket(1, 0, False, ss_l[i] or None, None), q)
        assert list(st[st_item == i])[:len(r.statuses)] == r.statuses and (err[i] == 0) == (r.err is None), i


def test_big_batch_with_other_hashes_still_runs_their_digests(gpu_ctx):
    """The host skips the k_digest_other launch of a big batch when k_plan reports (mailbox) that no signature names another hash
    than SHA-256.  Here a batch above that size threshold DOES carry SHA-1 / 224 / 384 / 512 signatures (the gpg fixtures of group
    A, RSA and DSA signers, tiled to 100,020 one-packet items): every item must get the verdict its vector gets in a small call."""
    import json
    import os
    from oracle import collective as col
    from oracle import openpgp as pgp
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gpg_vectors.json")))
    ring = pgp.read_entities(bytes.fromhex(vec["A_pubring"]))
    kr = col.Keyring(keyring=ring)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    qh = gpu_ctx.quorum_create([(0, 1, 1, 1, [e.id for e in ring])])          # one verified signer suffices
    tbs_l = [bytes.fromhex(v["payload"]) for v in vec["A"]]
    sig_l = [bytes.fromhex(v["sig"]) for v in vec["A"]]
    assert {v["digest"] for v in vec["A"]} >= {"SHA1", "SHA256", "SHA512"}
    tb, to = _cat(tbs_l)
    sb, so = _cat(sig_l)
    err0, nver0, _ = gpu_ctx.collective_verify(qh, tb, to, sb, so)
    from oracle.packet import SignaturePacket
    want = [col.signature_verify(kr, t, SignaturePacket(1, 0, False, s_, None)) is None for t, s_ in zip(tbs_l, sig_l)]
    assert [bool(e == 0) for e in err0] == want and sum(want) >= 25
    T = 3334
    tbT, toT = _cat(tbs_l * T)
    sbT, soT = _cat(sig_l * T)
    gpu_ctx.set_host_pipeline(1)          # one resident-sized call: 100,020 packets, above the turnstile / mailbox threshold
    try:
        err, nver, _ = gpu_ctx.collective_verify(qh, tbT, toT, sbT, soT)
    finally:
        gpu_ctx.set_host_pipeline(0)
    assert len(err) == 30 * T and (err.reshape(T, 30) == err0[None, :]).all() and (nver.reshape(T, 30) == nver0[None, :]).all()
    err2, _, _ = gpu_ctx.collective_verify(qh, tbT, toT, sbT, soT)            # and cut into pieces by the size rule
    assert (err2 == err).all()
    gpu_ctx.quorum_destroy(qh)


# ---- DSA at the group sizes gpg and the reference use (a17; tests/golden/make_gpg_dsa_sizes_vectors.py) ---------------------------
def _dsa_sizes():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gpg_dsa_sizes_vectors.json")))


def _verify_singles(ctx, kr, payloads, sigs):
    """one detached signature per item through bftkv_gpu_signature_verify; returns (err, fenced, per-item statuses)"""
    tb, to = _cat(payloads)
    sb, so = _cat(sigs)
    err = ctx.signature_verify(tb, to, sb, so)
    fenced = np.array(ctx.last_fenced).copy()
    st, st_item = ctx.last_statuses()
    return err, fenced, [list(st[st_item == i]) for i in range(len(payloads))]


def test_dsa_1024_160_and_3072_256_gpg_vectors_on_gpu(gpu_ctx):
    """gpg-made dsa1024 (q 160 bits) and dsa3072 (q 256 bits) keys and signatures under SHA-1 / 224 / 256 / 384 / 512 (digest as wide
    as q, or cut to its leftmost bits(q) / 8 bytes), intact and with a tampered payload: the device's verdict is gpg's and the
    oracle's, the per-packet status is the oracle's, and NOTHING is fenced -- neither size is handed back to the reference path."""
    from oracle import collective as col
    from oracle import openpgp as pgp
    vec = _dsa_sizes()
    ring = pgp.read_entities(bytes.fromhex(vec["A_pubring"]))
    kr = col.Keyring(keyring=ring)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    pl = [bytes.fromhex(v["payload"]) for v in vec["A"]]
    sg = [bytes.fromhex(v["sig"]) for v in vec["A"]]
    n = len(pl)
    err, fenced, sts = _verify_singles(gpu_ctx, kr, pl + [p + b"!" for p in pl], sg + sg)
    assert not fenced.any()
    for i, v in enumerate(vec["A"]):
        assert (err[i] == 0) == v["gpg_good"] and (err[n + i] == 0) == v["gpg_tampered_good"], v
        for k, payload in ((i, pl[i]), (n + i, pl[i] + b"!")):
            r = pgp.check_detached_signature(ring, payload, sg[i], 0)
            assert sts[k] == r.statuses, (v["key"], v["digest"], sts[k], r.statuses)
    assert (err[:n] == 0).all() and (err[n:] != 0).all()
    by_key = {k: sum(1 for v in vec["A"] if v["key"] == k) for k in ("dsa1024", "dsa3072")}
    assert by_key == {"dsa1024": 16, "dsa3072": 12}


def test_generator_dsa_sizes_judged_by_gpg_on_gpu(gpu_ctx):
    """Generator keys of 1024/160, 1536/224 and 3072/256 bits with Go-shaped signatures under SHA-1 / SHA-256 / SHA-512, intact and
    tampered (payload; a bit of s): verdicts are gpg's, statuses the oracle's, nothing fenced."""
    from oracle import collective as col
    from oracle import openpgp as pgp
    vec = _dsa_sizes()
    ring = pgp.read_entities(bytes.fromhex(vec["B_pubring"]))
    kr = col.Keyring(keyring=ring)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    pl = [bytes.fromhex(v["payload"]) for v in vec["B"]]
    sg = [bytes.fromhex(v["sig"]) for v in vec["B"]]
    err, fenced, sts = _verify_singles(gpu_ctx, kr, pl, sg)
    assert not fenced.any()
    good = {}
    for i, v in enumerate(vec["B"]):
        assert (err[i] == 0) == v["gpg_good"], v
        r = pgp.check_detached_signature(ring, pl[i], sg[i], 0)
        assert sts[i] == r.statuses, (v["p_bits"], v["q_bits"], v["hash_id"], v["tamper"], sts[i], r.statuses)
        if err[i] == 0:
            good[v["p_bits"]] = good.get(v["p_bits"], 0) + 1
    assert good == {1024: 18, 1536: 12, 3072: 12}


@pytest.mark.parametrize("wbits", (0, 8, 13))
def test_mixed_dsa_group_sizes_in_one_quorum(gpu_ctx, wbits):
    """One 16-replica clique whose members hold RSA-2048 keys and DSA keys of FOUR group sizes (1024/160, 1536/224, 2048/256,
    3072/256), writes with the usual mutations: error bytes, exit counts and per-packet statuses against the oracle, nothing fenced.
    Table widths: the library's choice, 8 and 13 bits (every size class builds and reads its own fixed-base tables)."""
    cl = cb.make_cluster(16, dsa_fraction=0.75, dsa_kind=("dsa1024", "dsa3072", "dsa1536", "dsa2048"))
    assert sorted({r.p.bit_length() for r in cl.replicas if r.algo == cb.PK_DSA}) == [1024, 1536, 2048, 3072]
    rates = {cb.MUT_BAD_MPI: 0.15, cb.MUT_ONE_SHORT: 0.15, cb.MUT_UNKNOWN_ISSUER: 0.1, cb.MUT_DUP_SIGNER: 0.1}
    c = cb.make_write_corpus(cl, 96, mutation_rates=rates, seed=606 + wbits)
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    gpu_ctx.set_dsa_window_bits(wbits)
    try:
        gpu_ctx.keyring_set(H.abi_keys(kr))
        qh = gpu_ctx.quorum_create(H.abi_qcs(q))
        err, nver, _ = gpu_ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
        assert not np.array(gpu_ctx.last_fenced).any()
        st, st_item = gpu_ctx.last_statuses()
        n_dsa_ok = 0
        dsa_ids = {r.key_id for r in cl.replicas if r.algo == cb.PK_DSA}
        for i in range(c.n_items):
            r = H.oracle_collective(kr, q, c, i)
            assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified), (i, err[i], r.err)
            got = list(st[st_item == i])
            assert got[:len(r.statuses)] == r.statuses, (i, got, r.statuses)
            n_dsa_ok += sum(1 for kid in r.verified if kid in dsa_ids)
        assert n_dsa_ok > 300 and 0 < int((err == 0).sum()) < c.n_items
        gpu_ctx.quorum_destroy(qh)
    finally:
        gpu_ctx.set_dsa_window_bits(0)


def test_dsa_table_budget_bounds_the_hbm_the_tables_hold():
    """bftkv_gpu_set_dsa_table_budget (a service that shares the GPU): 32 DSA-2048 keys (BASELINE configs[2]'s keyring) under 8 GB /
    24 GB / no budget get 14- / 16- / 18-bit tables (37 / 31 / 29 multiplications per signature), the arena's allocation stays within
    the budget, and the verdicts stay the oracle's at every width.  Its own context: the arena is tens of GB."""
    import torch  # noqa: F401
    from bftkv_amd import Context
    cl = cb.make_cluster(64, dsa_fraction=0.5)
    assert sum(1 for r in cl.replicas if r.algo == cb.PK_DSA) == 32
    c = cb.make_write_corpus(cl, 24, seed=77, mutation_rates={cb.MUT_BAD_MPI: 0.2, cb.MUT_ONE_SHORT: 0.2})
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    want = [H.oracle_collective(kr, q, c, i) for i in range(c.n_items)]
    ctx = Context(0)
    try:
        picked = {}
        for budget in (8 << 30, 24 << 30, 0):
            ctx.set_dsa_table_budget(budget)
            ctx.keyring_set(H.abi_keys(kr))
            qh = ctx.quorum_create(H.abi_qcs(q))
            bits = ctx.dsa_window_bits()
            held, entry = ctx.dsa_table_bytes()
            picked[budget] = bits
            assert entry == 76
            if budget:
                assert held <= budget, (budget, held)
            err, nver, _ = ctx.collective_verify(qh, c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off)
            st, st_item = ctx.last_statuses()
            for i, r in enumerate(want):
                assert (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified), (budget, i)
                assert list(st[st_item == i][:len(r.statuses)]) == r.statuses, (budget, i)
            assert not np.array(ctx.last_fenced).any()
            ctx.quorum_destroy(qh)
        assert picked[8 << 30] == 14 and picked[24 << 30] == 16 and picked[0] in (16, 18), picked
        # a budget too small for anything but the narrowest tables still verifies (4.96 MB per key at 8 bits)
        ctx.set_dsa_table_budget(200 << 20)
        ctx.keyring_set(H.abi_keys(kr))
        assert ctx.dsa_window_bits() == 8 and ctx.dsa_table_bytes()[0] <= 200 << 20
    finally:
        ctx.close()


def test_dsa_group_sizes_at_volume(gpu_ctx):
    """The four DSA group sizes at volume: a 16-replica all-DSA clique (1024/160, 3072/256, 1536/224, 2048/256 dealt round-robin), 300
    signed writes with the usual mutations tiled 64 x in the call (~300 k signature packets, both k_dsa_modexp instantiations over
    the class lists of k_dsa_split).  Tile 0 is the C restatement's answer on every write (error byte, exit count) and the Python
    oracle's per-packet statuses on a sample; every other tile equals tile 0 -- items are independent."""
    from oracle.cbind import COracle
    cl = cb.make_cluster(16, dsa_fraction=1.0, dsa_kind=("dsa1024", "dsa3072", "dsa1536", "dsa2048"))
    rates = {cb.MUT_BAD_MPI: 0.1, cb.MUT_ONE_SHORT: 0.15, cb.MUT_UNKNOWN_ISSUER: 0.05, cb.MUT_DUP_SIGNER: 0.05, cb.MUT_BAD_TAG: 0.05}
    c = cb.make_write_corpus(cl, 300, mutation_rates=rates, seed=6066)
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    co = COracle()
    co.set_keyring(kr)
    co.set_quorum(q)
    cerr, cnver, _ = co.collective_verify(c.tbss_blob, c.tbss_off, c.ss_blob, c.ss_off, 4)
    gpu_ctx.keyring_set(H.abi_keys(kr))
    assert gpu_ctx.dsa_table_bytes()[1] == 112
    qh = gpu_ctx.quorum_create(H.abi_qcs(q))
    T = 64
    tb, sb = np.tile(c.tbss_blob, T), np.tile(c.ss_blob, T)
    step_t, step_s = np.uint64(c.tbss_off[-1]), np.uint64(c.ss_off[-1])
    to = np.concatenate([(np.arange(T, dtype=np.uint64)[:, None] * step_t + c.tbss_off[None, :-1].astype(np.uint64)).reshape(-1), [np.uint64(T) * step_t]]).astype(np.uint64)
    so = np.concatenate([(np.arange(T, dtype=np.uint64)[:, None] * step_s + c.ss_off[None, :-1].astype(np.uint64)).reshape(-1), [np.uint64(T) * step_s]]).astype(np.uint64)
    err, nver, vd = gpu_ctx.collective_verify(qh, tb, to, sb, so)
    st, st_item = gpu_ctx.last_statuses()
    assert not np.array(gpu_ctx.last_fenced).any()
    err, nver, vd = err.reshape(T, -1), nver.reshape(T, -1), vd.reshape(T, -1)
    assert ((err[0] == 0) == (cerr == 0)).all() and (nver[0] == cnver).all()
    assert (err == err[0][None, :]).all() and (nver == nver[0][None, :]).all() and (vd == vd[0][None, :]).all()
    per_tile = len(st) // T
    assert len(st) == per_tile * T and (st.reshape(T, per_tile) == st[:per_tile][None, :]).all()
    for i in range(0, c.n_items, 15):
        r = H.oracle_collective(kr, q, c, i)
        assert list(st[:per_tile][st_item[:per_tile] == i][:len(r.statuses)]) == r.statuses, i
    assert 0 < int((err[0] == 0).sum()) < c.n_items and int(gpu_ctx.last_counters()["dsa_ops"]) > 150000
    gpu_ctx.quorum_destroy(qh)
