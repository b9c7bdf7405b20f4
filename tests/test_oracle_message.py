"""Signed-message reader of the transport row: oracle/message.py pinned against GnuPG's verdicts
(tests/golden/gpg_messages.json, made by tests/golden/make_gpg_messages.py) and its outcome table checked case by case."""
import json
import os

import pytest

from corpus import build as cb
from corpus.keys import DRBG
from oracle import message as om
from oracle import openpgp as pgp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gpg_messages.json")


@pytest.fixture(scope="module")
def vec():
    with open(GOLD) as f:
        return json.load(f)


def test_gpg_made_messages(vec):
    ring = pgp.read_entities(bytes.fromhex(vec["D_pubring"]))
    assert len(vec["D"]) == 32
    partial = 0
    for d in vec["D"]:
        msg = bytes.fromhex(d["msg"])
        r = om.read_signed_message(ring, msg)
        assert d["gpg_good"] and r.status == om.MSG_OK and r.plain == bytes.fromhex(d["payload"])
        assert r.peer == r.signed_by_key_id and r.peer in [e.id for e in ring]
        t = om.read_signed_message(ring, bytes.fromhex(d["tampered"]))
        assert (t.status == om.MSG_OK) == d["gpg_tampered_good"] and not d["gpg_tampered_good"]
        partial += any(224 <= b < 255 for b in msg[15:17])
    assert partial >= 4      # piped gpg input uses partial body lengths for the literal packet


def test_generator_messages_judged_by_gpg(vec):
    ring = pgp.read_entities(bytes.fromhex(vec["E_pubring"]))
    good = 0
    for d in vec["E"]:
        r = om.read_signed_message(ring, bytes.fromhex(d["msg"]))
        assert (r.status == om.MSG_OK) == d["gpg_good"], d
        if d["tamper"] is None:
            assert d["gpg_good"] and r.plain == bytes.fromhex(d["payload"]) and r.file_name == b"MDEyMzQ1Njc4OWFiY2RlZg=="
        good += d["gpg_good"]
    assert good >= 30


def test_outcome_table():
    """One case per Decrypt outcome (crypto_pgp.go:453-471)."""
    cl = cb.make_cluster(4, n_outsiders=1)
    rng = DRBG("msgcases")
    ring = pgp.read_entities(b"".join(r.entity for r in cl.replicas))
    kp, out = cl.replicas[1], cl.outsiders[0]
    msg = cb.signed_message(kp, b"request", b"n" * 16, rng)
    r = om.read_signed_message(ring, msg)
    assert (r.status, r.plain, r.peer, r.signed_by_key_id) == (om.MSG_OK, b"request", kp.key_id, kp.key_id)
    sig = cb.detach_sign(kp, b"request", rng)
    ops = om.one_pass_packet(0, 8, kp.algo, kp.key_id)
    lit = om.literal_packet(b"f", b"request")
    # unsigned: ErrInvalidTransportSecurityData
    assert om.read_signed_message(ring, lit).status == om.MSG_NOT_SIGNED
    # signer not in the keyring: NIL error, peer nil (the join case, server.go:565-569)
    r = om.read_signed_message(ring, cb.signed_message(out, b"join me", b"n" * 16, rng))
    assert (r.status, r.peer, r.plain) == (om.MSG_UNVERIFIED, None, b"join me")
    # one-pass packet names a key id the signature was not made with
    r = om.read_signed_message(ring, om.one_pass_packet(0, 8, kp.algo, cl.replicas[2].key_id) + lit + sig)
    assert r.status == om.MSG_SIGNATURE_ERROR and r.peer == cl.replicas[2].key_id
    # one-pass hash differs from the signature's hash: the body was hashed with the one-pass algorithm
    r = om.read_signed_message(ring, om.one_pass_packet(0, 10, kp.algo, kp.key_id) + lit + sig)
    assert r.status == om.MSG_SIGNATURE_ERROR
    # literal not followed by a signature / by nothing
    assert om.read_signed_message(ring, ops + lit + lit).status == om.MSG_SIGNATURE_ERROR
    assert om.read_signed_message(ring, ops + lit).status == om.MSG_SIGNATURE_ERROR
    # unknown packet types are skipped by Reader.Next on both sides of the literal
    unk = pgp.new_format_header(60, 3) + b"abc"
    assert om.read_signed_message(ring, unk + ops + unk + lit + unk + sig).status == om.MSG_OK
    # ReadMessage errors
    assert om.read_signed_message(ring, om.one_pass_packet(0, 8, kp.algo, kp.key_id, is_last=False) + lit + sig).status == om.MSG_READ_ERROR
    assert om.read_signed_message(ring, om.one_pass_packet(0x10, 8, kp.algo, kp.key_id) + lit + sig).status == om.MSG_READ_ERROR
    assert om.read_signed_message(ring, ops).status == om.MSG_READ_ERROR
    assert om.read_signed_message(ring, b"").status == om.MSG_READ_ERROR
    assert om.read_signed_message(ring, ops + lit[:5]).status == om.MSG_READ_ERROR
    assert om.read_signed_message(ring, b"\x00" + ops + lit + sig).status == om.MSG_READ_ERROR
    # fenced shapes
    assert om.read_signed_message(ring, om.one_pass_packet(1, 8, kp.algo, kp.key_id) + lit + sig).status == om.MSG_UNSUPPORTED
    assert om.read_signed_message(ring, pgp.new_format_header(8, 2) + b"\x00\x00" + ops + lit + sig).status == om.MSG_UNSUPPORTED
    assert om.read_signed_message(ring, ops + ops + lit + sig).status == om.MSG_UNSUPPORTED
    # bytes after the signature are never read
    assert om.read_signed_message(ring, ops + lit + sig + b"\x00garbage").status == om.MSG_OK
    # partial-length literal, every chunking of the same content gives the same answer
    for parts in ([0], [1, 0, 2], om.go_partial_chunks(len(b"b\x01f\x00\x00\x00\x00request"))):
        assert om.read_signed_message(ring, ops + om.literal_packet(b"f", b"request", partial=parts) + sig).status == om.MSG_OK
