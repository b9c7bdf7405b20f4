"""CPU: the oracle against the reference ITSELF -- strict once someone with a Go toolchain has run
shim/tools/genvectors/main.go (the reference's crypto/pgp + quorum/wotqs with the x/crypto go.mod:8 pins) over
tests/golden/reference_inputs.json and committed tests/golden/reference_vectors.json; skipped while that file is absent
(this image has no Go, DESIGN.md section 5 "parity unpinned").  The replay half always runs, so the comparison code cannot rot."""
import json
import os

import pytest

from oracle import collective as col
from oracle import openpgp as pgp
from oracle import wotqs as W
from oracle.packet import SignaturePacket

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INPUTS = os.path.join(GOLD, "reference_inputs.json")
VECTORS = os.path.join(GOLD, "reference_vectors.json")

# oracle status of a failed CheckDetachedSignature call -> what the reference's error (class:text, genvectors.class) must look
# like.  The TYPE is asserted where x/crypto's type is certain, the text only where the oracle's status names one message.
CLASS_OF = {
    pgp.ST_UNKNOWN_ISSUER: lambda c: c == "unknown-issuer",
    pgp.ST_PARSE_ERROR: lambda c: c.startswith(("structural:", "unsupported:", "error:")),
    pgp.ST_NOT_SIGNATURE: lambda c: c.startswith("structural:"),
    pgp.ST_NO_ISSUER: lambda c: c.startswith("structural:"),
    pgp.ST_HASH_UNSUPPORTED: lambda c: c.startswith("unsupported:"),
    pgp.ST_HASH_TAG: lambda c: c.startswith("signature-error:") and "hash tag" in c,
    pgp.ST_ALGO_MISMATCH: lambda c: "different algorithms" in c,
    pgp.ST_BAD_SIG: lambda c: c.startswith("signature-error:"),
    pgp.ST_KEY_CANNOT_SIGN: lambda c: "cannot generate signatures" in c,
    pgp.ST_UNSUPPORTED: lambda c: c.startswith(("unsupported:", "structural:", "error:")),
}


def replay(keyring, tbs, data):
    """The loop of crypto_pgp.go:485-500 over the oracle: [(status, signer id or None)] per CheckDetachedSignature call."""
    out, pos = [], 0
    while len(data) - pos > 0:
        r = pgp.check_detached_signature(keyring, tbs, data, pos)
        pos = r.pos
        out.append((r.status, r.signer.id if r.status == pgp.ST_OK else None))
    return out


def cluster_quorum(c):
    """ChooseQuorum(AUTH) from the certified ring, as the reference's daemon builds it (graph.AddNodes over the pubring,
    SetSelfNodes, wotqs.New(g).ChooseQuorum)."""
    ents = pgp.read_entities(bytes.fromhex(c["pubring"]))
    g = W.Graph()
    g.add_nodes([(e.id, list(e.certifiers)) for e in ents])
    g.set_self([int(c["self"], 16)])
    return ents, W.Wot(g).choose_quorum(W.AUTH)


def oracle_item(ents, q, tbs, data):
    kr = col.Keyring(keyring=ents)
    calls = replay(kr.get_keyring(), tbs, data)
    sp = SignaturePacket(1, 0, False, data or None, None)
    o = {"calls": calls, "signature": col.signature_verify(kr, tbs, sp)}
    if q is not None:
        r = col.collective_verify(kr, tbs, SignaturePacket(Type=1, Data=data or None), q)
        allv = [s for st, s in calls if st == pgp.ST_OK]
        o.update(collective=r.err, completed=r.completed, n_verified=len(r.verified), is_quorum=q.is_quorum(allv),
                 is_threshold=q.is_threshold(allv), is_sufficient=q.is_sufficient(allv), reject=q.reject(allv))
    return o


@pytest.fixture(scope="module")
def inputs():
    return json.load(open(INPUTS))


def oracle_packet(b):
    """packet.TBS / packet.TBSS of one byte string through the oracle: (prefix or None, prefix or None)"""
    from oracle import packet as opk
    res = []
    for fn in (opk.tbs, opk.tbss):
        try:
            res.append(fn(b))
        except opk.PacketError:
            res.append(None)
    return tuple(res)


def oracle_structure(w):
    """The entity ReadEntity returns, as genvectors writes it (`structure`): identities by name -- a later one of the same name in the
    place of the earlier -- with the self-signature that counts and the issuers of the signatures collected on them, subkeys with
    their Subkey.Sig, the number of revocations.  It is also what the library's packet roles assemble (tests/test_cert_walk.py)."""
    by_name = {}
    for ident in w.identities:
        if ident["self_sig"] is not None:
            by_name[bytes(ident["name"])] = ident
    # Go sorts the names as strings (bytes): the same order
    idents = [{"name": nm.hex(), "self_type": i["self_sig"].sig_type, "self_creation": i["self_sig"].creation_time,
               "signatures": ["nil" if s_.issuer is None else "%016x" % s_.issuer for s_, _, _ in i["sigs"]]} for nm, i in sorted(by_name.items())]
    subs = [{"key_id": "%016x" % sk["key"].key_id, "sig_type": sk["sig"].sig_type, "sig_creation": sk["sig"].creation_time} for sk in w.subkeys]
    return {"identities": idents, "subkeys": subs, "revocations": len(w.revocations)}


def oracle_cert(blob):
    """crypto.Certificate.Parse: the entities ReadEntity accepts, in order, up to the first one it refuses (crypto_pgp.go:236-249);
    for the first one Signers() and the key ids KeysByIdUsage(id, KeyFlagSign) returns.  None in place of an entity: a shape the
    restatement leaves to the reference (the comparison skips that certificate)."""
    ws = pgp.parse_certificate(blob)
    out = {"ids": [None if w is None else w.primary.key_id for w in ws], "signers": [], "usable": []}
    if ws and ws[0] is not None:
        e = [x for x in pgp.read_entities(blob) if x.serialized == blob[ws[0].start:ws[0].end]][0]
        out["signers"] = e.certifiers
        out["structure"] = oracle_structure(ws[0])
        out["usable"] = [k for k in [e.primary.key_id] + [sk.key_id for sk, _, _, _ in e.subkeys] if pgp.keys_by_id_usage_sign([e], k)]
    return out


@pytest.fixture(scope="module")
def replayed(inputs):
    out = {"clusters": [], "streams": [], "gpg": [], "packets": [oracle_packet(bytes.fromhex(p)) for p in inputs.get("packets", [])],
           "certs": [oracle_cert(bytes.fromhex(c)) for c in inputs.get("certs", [])]}
    by_name = {}
    for c in inputs["clusters"]:
        ents, q = cluster_quorum(c)
        by_name[c["name"]] = (ents, q)
        out["clusters"].append({"name": c["name"], "q": q,
                                "items": [oracle_item(ents, q, bytes.fromhex(i["tbs"]), bytes.fromhex(i["ss"])) for i in c["items"]]})
    for s in inputs["streams"]:
        ents, q = by_name[s["cluster"]]
        out["streams"].append(oracle_item(ents, q, bytes.fromhex(s["tbs"]), bytes.fromhex(s["ss"])))
    rings = {k: pgp.read_entities(bytes.fromhex(v)) for k, v in inputs["rings"].items()}
    for v in inputs["gpg"]:
        out["gpg"].append(oracle_item(rings[v["ring"]], None, bytes.fromhex(v["tbs"]), bytes.fromhex(v["sig"])))
    return out


def test_inputs_are_replayable_and_cover_both_outcomes(inputs, replayed):
    """Always runs: every committed input goes through the oracle's replay, the clusters form the cliques their members name,
    and the set holds accepted and refused items, every mutation class and several failure statuses."""
    assert inputs["format"] == 1 and len(inputs["clusters"]) == 3 and len(inputs["gpg"]) > 100
    statuses = set()
    for c, cin in zip(replayed["clusters"], inputs["clusters"]):
        q = c["q"]
        assert len(q.qcs) == 1 and sorted(q.qcs[0].nodes) == sorted(int(m, 16) for m in cin["members"])
        ok = [i["collective"] is None for i in c["items"]]
        assert any(ok) and not all(ok)
        for i in c["items"]:
            statuses.update(st for st, _ in i["calls"])
            assert i["completed"] == (i["collective"] is None)
    assert {pgp.ST_OK, pgp.ST_UNKNOWN_ISSUER, pgp.ST_BAD_SIG, pgp.ST_HASH_TAG} <= statuses
    stream_statuses = {st for s in replayed["streams"] for st, _ in s["calls"]}
    assert pgp.ST_PARSE_ERROR in stream_statuses and pgp.ST_NOT_SIGNATURE in stream_statuses
    good = [g["signature"] is None for g in replayed["gpg"]]
    assert sum(good) > 40 and sum(not g for g in good) > 20
    # packets: TBS / TBSS answers of both kinds; certificates: accepted, refused at the first entity, refused in the middle
    assert len(replayed["packets"]) > 150 and any(t is None for t, _ in replayed["packets"]) and any(t is not None and u is None for t, u in replayed["packets"])
    assert any(u is not None for _, u in replayed["packets"])
    n_ents = [len(c["ids"]) for c in replayed["certs"]]
    assert 0 in n_ents and 1 in n_ents and 3 in n_ents and len(n_ents) >= 70
    assert any(c["signers"] for c in replayed["certs"]) and any(len(c["usable"]) > 1 for c in replayed["certs"])
    assert any(None in c["ids"] for c in replayed["certs"])


def _same_item(tag, ours, ref):
    assert len(ours["calls"]) == len(ref["calls"]), (tag, "number of CheckDetachedSignature calls", ours["calls"], ref["calls"])
    for k, ((st, signer), rc) in enumerate(zip(ours["calls"], ref["calls"])):
        if st == pgp.ST_OK:
            assert rc == "ok:%016x" % signer, (tag, k, rc)
        else:
            assert not rc.startswith("ok:") and CLASS_OF[st](rc), (tag, k, st, rc)
    assert (ours["signature"] is None) == (ref["signature"] == ""), (tag, "Signature.Verify", ref["signature"])
    if ref["signature"]:
        assert ref["signature"] == "crypto: invalid signature"
    if ref.get("has_quorum"):
        assert (ours["collective"] is None) == (ref.get("collective", "") == ""), (tag, "CollectiveSignature.Verify")
        if ref.get("collective"):
            assert ref["collective"] == "crypto: insufficient number of signatures"
        assert ours["completed"] == ref.get("completed", False) and ours["n_verified"] == ref["n_verified"], tag
        for k in ("is_quorum", "is_threshold", "is_sufficient", "reject"):
            assert ours[k] == ref.get(k, False), (tag, k)


def test_oracle_matches_the_reference_vectors(inputs, replayed):
    if not os.path.exists(VECTORS):
        pytest.skip("tests/golden/reference_vectors.json absent: run shim/tools/genvectors/main.go with a Go toolchain "
                    "(x/crypto pinned by the reference's go.mod:8) to pin the oracle to the reference itself")
    ref = json.load(open(VECTORS))
    assert ref["format"] == 1 and "53104e6ec876" in ref["x_crypto"]
    # what the reference binary links decides the MD5 / RIPEMD-160 outcomes: replay under the same policy
    pgp.HASH_POLICY.update(md5=bool(ref.get("md5_available")), ripemd160=bool(ref.get("ripemd160_available")))
    assert len(ref["clusters"]) == len(replayed["clusters"])
    for ours, theirs, cin in zip(replayed["clusters"], ref["clusters"], inputs["clusters"]):
        assert ours["name"] == theirs["name"] and len(theirs["items"]) == len(cin["items"])
        # the cliques ChooseQuorum(AUTH) built: newQC's numbers and members (wotqs.go:36-70, graph.go:297-362)
        assert len(theirs["cliques"]) == len(ours["q"].qcs)
        for qc, k in zip(ours["q"].qcs, theirs["cliques"]):
            assert (qc.f, qc.min, qc.threshold, qc.suff) == (k["F"], k["Min"], k["Threshold"], k["Suff"])
            assert sorted(qc.nodes) == sorted(int(x, 16) for x in k["Nodes"])
        for n, (a, b) in enumerate(zip(ours["items"], theirs["items"])):
            _same_item("%s/%d" % (ours["name"], n), a, b)
    assert len(ref["streams"]) == len(replayed["streams"]) and len(ref["gpg"]) == len(replayed["gpg"])
    skipped = 0
    for n, (a, b) in enumerate(zip(replayed["streams"], ref["streams"])):
        # streams after whose packets the reference's reader position depends on the parser of a non-signature packet type are
        # not modelled by the oracle (oracle.openpgp.fence_reason "lazy") and are FENCED by the verifier: nothing to compare.
        # (Signatures that leave the reader inside their own packet ARE followed by the oracle: compared.)
        if pgp.fence_reason(bytes.fromhex(inputs["streams"][n]["ss"])) == "lazy":
            skipped += 1
            continue
        _same_item("stream/%d" % n, a, b)
    assert skipped < len(ref["streams"])
    for n, (a, b) in enumerate(zip(replayed["gpg"], ref["gpg"])):
        _same_item("gpg/%s" % inputs["gpg"][n]["name"], a, b)
    # packet.TBS / TBSS (seek2tbs ignores its errors) and Certificate.Parse (ReadEntity), when the vectors carry them
    if "packets" in ref:
        assert len(ref["packets"]) == len(replayed["packets"])
        for n, (a, b) in enumerate(zip(replayed["packets"], ref["packets"])):
            _same_packet(n, a, b)
    if "certs" in ref:
        assert len(ref["certs"]) == len(replayed["certs"])
        for n, (a, b) in enumerate(zip(replayed["certs"], ref["certs"])):
            _same_cert(n, a, b)


def _same_cert(n, a, b):
    if None in a["ids"]:
        return            # a shape left to the reference: nothing to compare
    assert ["%016x" % i for i in a["ids"]] == b["ids"], ("cert", n, a, b)
    if "usable" in b:
        assert ["%016x" % i for i in a["usable"]] == b["usable"], ("cert", n, "KeysByIdUsage", a, b)
    # Signers() walks a Go map: the order across identities is not defined
    if "signers" in b and not b.get("signers_panic"):
        assert sorted("%016x" % i for i in a["signers"]) == sorted(b["signers"]), ("cert", n, "Signers()", a, b)
    if b.get("structure") is not None and a.get("structure") is not None:
        assert a["structure"] == b["structure"], ("cert", n, "the entity ReadEntity built", a["structure"], b["structure"])


def _same_packet(tag, ours, ref):
    for (mine, key) in ((ours[0], "tbs"), (ours[1], "tbss")):
        if mine is None:
            assert ref.get(key + "_err"), (tag, key, "the reference returned a prefix, the oracle an error", ref)
        else:
            assert not ref.get(key + "_err") and ref[key] == mine.hex(), (tag, key, ref)


REPRESENTATIVE = {
    pgp.ST_UNKNOWN_ISSUER: "unknown-issuer", pgp.ST_PARSE_ERROR: "structural:parse", pgp.ST_NOT_SIGNATURE: "structural:non-signature packet found",
    pgp.ST_NO_ISSUER: "structural:signature doesn't have an issuer", pgp.ST_HASH_UNSUPPORTED: "unsupported:hash",
    pgp.ST_HASH_TAG: "signature-error:hash tag doesn't match", pgp.ST_ALGO_MISMATCH: "error:public key and signature use different algorithms",
    pgp.ST_BAD_SIG: "signature-error:RSA verification failure", pgp.ST_KEY_CANNOT_SIGN: "error:public key cannot generate signatures",
    pgp.ST_UNSUPPORTED: "unsupported:fenced",
}


def _as_reference_would_write(o):
    ref = {"calls": ["ok:%016x" % s if st == pgp.ST_OK else REPRESENTATIVE[st] for st, s in o["calls"]],
           "signature": "" if o["signature"] is None else "crypto: invalid signature", "n_verified": o.get("n_verified", 0)}
    if "collective" in o:
        ref.update(has_quorum=True, collective="" if o["collective"] is None else "crypto: insufficient number of signatures",
                   completed=o["completed"], is_quorum=o["is_quorum"], is_threshold=o["is_threshold"], is_sufficient=o["is_sufficient"],
                   reject=o["reject"])
    return ref


def test_comparison_accepts_the_oracles_own_answers_and_refuses_a_flipped_one(replayed):
    """The strict comparison is exercised even while the reference's file is absent: fed the oracle's own answers in the
    reference's format it passes, and any single flipped verdict, signer or count makes it fail."""
    items = [i for c in replayed["clusters"] for i in c["items"]] + replayed["streams"] + replayed["gpg"]
    for n, o in enumerate(items):
        _same_item(n, o, _as_reference_would_write(o))
    for n, pk_ in enumerate(replayed["packets"]):
        as_ref = {"tbs": "" if pk_[0] is None else pk_[0].hex(), "tbss": "" if pk_[1] is None else pk_[1].hex()}
        if pk_[0] is None:
            as_ref["tbs_err"] = "unexpected EOF"
        if pk_[1] is None:
            as_ref["tbss_err"] = "EOF"
        _same_packet(n, pk_, as_ref)
        if pk_[0] is not None:
            with pytest.raises(AssertionError):
                _same_packet(n, pk_, dict(as_ref, tbs=(pk_[0] + b"x").hex()))
    n_cmp = 0
    for n, c in enumerate(replayed["certs"]):
        if None in c["ids"]:
            continue
        import copy
        as_ref = {"ids": ["%016x" % i for i in c["ids"]], "signers": ["%016x" % i for i in reversed(c["signers"])], "usable": ["%016x" % i for i in c["usable"]],
                  "structure": copy.deepcopy(c.get("structure"))}
        _same_cert(n, c, as_ref)
        n_cmp += 1
        other_entity = copy.deepcopy(as_ref["structure"]) if as_ref["structure"] else None
        if other_entity and other_entity["identities"]:
            other_entity["identities"][0]["self_creation"] += 1
        for tampered in (dict(as_ref, ids=as_ref["ids"] + ["00" * 8]), dict(as_ref, usable=as_ref["usable"][1:] if as_ref["usable"] else ["00" * 8]),
                         dict(as_ref, signers=as_ref["signers"] + ["00" * 8])) + ((dict(as_ref, structure=other_entity),) if other_entity else ()):
            with pytest.raises(AssertionError):
                _same_cert(n, c, tampered)
    assert n_cmp > 50
    o = next(i for i in items if i.get("collective", 1) is None and len(i["calls"]) > 2)
    for tamper in ("calls", "collective", "n_verified", "signature"):
        ref = _as_reference_would_write(o)
        if tamper == "calls":
            k = next(k for k, c in enumerate(ref["calls"]) if c.startswith("ok:"))
            ref["calls"][k] = "signature-error:RSA verification failure"
        elif tamper == "collective":
            ref["collective"] = "crypto: insufficient number of signatures"
        elif tamper == "n_verified":
            ref["n_verified"] += 1
        else:
            ref["signature"] = "" if ref["signature"] else "crypto: invalid signature"
        with pytest.raises(AssertionError):
            _same_item("tampered", o, ref)
