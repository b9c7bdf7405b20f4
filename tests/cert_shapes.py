"""Certificate shapes for the ReadEntity walk (oracle/openpgp.py walk_certificate, bftkv_host_certs_parse): hand-built entities
with REAL signatures whose verdict is worked out from x/crypto's rules, and a generator of random packet sequences (fake signature
values: structure only).  Shared by tests/test_cert_walk.py (CPU: oracle against the expectations, mirror against the oracle) and
the GPU test of bftkv_host_certs_verify."""
import hashlib
import struct

from corpus import build as cb
from corpus.keys import DRBG

T0 = cb.CREATION_TIME


def sub(typ: int, body: bytes, critical: bool = False) -> bytes:
    """One signature subpacket."""
    ln = 1 + len(body)
    assert ln < 192
    return bytes([ln, typ | (0x80 if critical else 0)]) + body


def hashed_area(issuer=None, ctime=T0, extra=b""):
    out = sub(2, struct.pack(">I", ctime)) + extra
    if issuer is not None:
        out += sub(16, struct.pack(">Q", issuer))
    return out


def key_framed(kp) -> bytes:
    return b"\x99" + struct.pack(">H", len(kp.pub_body)) + kp.pub_body


def uid_framed(uid: bytes) -> bytes:
    return b"\xb4" + struct.pack(">I", len(uid)) + uid


def sig_body(signer, sig_type, signed, hashed, unhashed=b"", rng=None, fake=False, spoil=False):
    """A v4 signature BODY by `signer` over `signed` (fake: a constant in place of the signature value)."""
    prefix = cb.sig_prefix(sig_type, signer.algo, hashed)
    digest = hashlib.sha256(signed + cb.hash_suffix(prefix)).digest()
    if spoil:
        digest = hashlib.sha256(digest).digest()
    if fake:
        mp = cb.go_mpi_bytes(b"\x01" * 8) * (1 if signer.algo == cb.PK_RSA else 2)
        return prefix + struct.pack(">H", len(unhashed)) + unhashed + digest[:2] + mp
    whole = cb.make_sig_packet(signer, prefix, digest, rng or DRBG("cert-shapes"))
    # make_sig_packet writes an empty unhashed area: splice ours in
    body = whole[2 if whole[1] < 192 else 3 if whole[1] < 224 else 6:]
    assert body.startswith(prefix + b"\x00\x00")
    return prefix + struct.pack(">H", len(unhashed)) + unhashed + body[len(prefix) + 2:]


def pkt(tag: int, body: bytes) -> bytes:
    return cb._hdr(tag, len(body)) + body


def sig_pkt(*a, **k) -> bytes:
    return pkt(2, sig_body(*a, **k))


def self_sig(kp, uid, sig_type=0x13, flags=0x03, ctime=T0, primary=None, **k):
    extra = (sub(27, bytes([flags])) if flags is not None else b"") + (sub(25, bytes([1 if primary else 0])) if primary is not None else b"")
    return sig_pkt(kp, sig_type, key_framed(kp) + uid_framed(uid), hashed_area(kp.key_id, ctime, extra), **k)


def certification(signer, kp, uid, sig_type=0x10, issuer="signer", **k):
    iss = signer.key_id if issuer == "signer" else issuer
    return sig_pkt(signer, sig_type, key_framed(kp) + uid_framed(uid), hashed_area(iss), **k)


def binding(kp, sk, sig_type=0x18, flags=0x0c, ctime=T0, cross="auto", issuer="primary", reason=None, **k):
    """Subkey binding (or 0x28 revocation) by the primary key; flags with bit 1 ("sign") get the subkey's 0x19 cross-signature
    embedded in the unhashed area (cross: "auto" / None / "bad" / "wrong-issuer")."""
    signed = key_framed(kp) + key_framed(sk)
    extra = (sub(27, bytes([flags])) if flags is not None else b"") + (sub(29, bytes([reason]) + b"gone") if reason is not None else b"")
    unhashed = b""
    fake = k.get("fake", False)
    if cross is not None and flags is not None and (flags & 2):
        eb = sig_body(sk, 0x19, signed, hashed_area(None if cross == "no-issuer" else (kp.key_id if cross in ("wrong-issuer", "bad-wrong-issuer") else sk.key_id), ctime),
                      spoil=cross in ("bad", "bad-wrong-issuer"), fake=fake)
        unhashed = bytes([255]) + struct.pack(">I", 1 + len(eb)) + bytes([32]) + eb
    iss = kp.key_id if issuer == "primary" else issuer
    return sig_pkt(kp, sig_type, signed, hashed_area(iss, ctime, extra), unhashed, **k)


def key_revocation(kp, issuer="primary", **k):
    return sig_pkt(kp, 0x20, key_framed(kp), hashed_area(kp.key_id if issuer == "primary" else issuer, T0, sub(29, b"\x00")), **k)


def embedded(eb: bytes) -> bytes:
    """An embedded-signature subpacket (type 32, five-octet length)."""
    return bytes([255]) + struct.pack(">I", 1 + len(eb)) + bytes([32]) + eb


def nested_binding(kp, sk, depth: int) -> bytes:
    """A signing subkey's binding whose cross-signature (real) carries `depth - 1` further 0x19 signatures nested in its unhashed
    area (fake values: x/crypto parses embedded signatures recursively but verifies only the outermost one)."""
    signed = key_framed(kp) + key_framed(sk)
    inner = b""
    for _ in range(depth - 1):
        inner = embedded(sig_body(sk, 0x19, signed, hashed_area(sk.key_id), inner, fake=True))
    eb = sig_body(sk, 0x19, signed, hashed_area(sk.key_id), inner)
    return sig_pkt(kp, 0x18, signed, hashed_area(kp.key_id, T0, sub(27, bytes([0x02]))), embedded(eb))


def body_over_4096(kp, sk) -> bytes:
    """A binding-typed signature body of more than 4096 bytes (fake value, bulky unhashed area)."""
    return sig_body(kp, 0x18, key_framed(kp) + key_framed(sk), hashed_area(kp.key_id, T0, sub(27, bytes([0x0c]))),
                    b"".join(sub(100, bytes(150)) for _ in range(29)), fake=True)


def keys():
    rsa = cb.load_keys("rsa2048", 90)
    dsa = cb.load_keys("dsa2048", 12)
    a = cb.make_keypair(cb.PK_RSA, rsa[85], "c01 <c01@bftkv.example>")
    b = cb.make_keypair(cb.PK_RSA, rsa[86], "c02 <c02@bftkv.example>")
    s = cb.make_keypair(cb.PK_RSA, rsa[87], "")
    s2 = cb.make_keypair(cb.PK_RSA, rsa[88], "")
    d = cb.make_keypair(cb.PK_DSA, dsa[10], "d02 <d02@bftkv.example>")
    return a, b, s, s2, d


def scenarios():
    """-> [(name, blob, valid of each entity (True / False / None), Signers() of entity 0 or None, usable keys of entity 0 or None)].
    The expectations are worked from x/crypto's ReadEntity by hand -- not computed with the oracle."""
    a, b, s, s2, d = keys()
    uid, uid2 = a.name.encode(), b"c01 second <c01b@bftkv.example>"
    K = pkt(6, a.pub_body)
    U, U2 = pkt(13, uid), pkt(13, uid2)
    S = pkt(14, s.pub_body)
    S2 = pkt(14, s2.pub_body)
    ss = self_sig(a, uid)
    cert_b = certification(b, a, uid)
    out = []
    add = lambda name, blob, valid, signers=None, usable=None: out.append((name, blob, valid, signers, usable))
    add("plain: key, uid, self-signature, one certification", K + U + ss + cert_b, [True], [b.key_id], [a.key_id])
    add("gen.sh's shape: [SC] primary and an [E] subkey", K + U + ss + S + binding(a, s), [True], [], [a.key_id])
    add("a uid without self-signature is dropped with the certifications on it",
        K + U + ss + U2 + certification(b, a, uid2) + certification(d, a, uid2), [True], [], [a.key_id])
    add("... and an entity with no other uid is refused", K + U + cert_b, [False])
    add("0x11 / 0x12 by the primary key are not self-signatures: they count as signers",
        K + U + ss + self_sig(a, uid, sig_type=0x11) + cert_b, [True], [a.key_id, b.key_id], [a.key_id])
    add("a 0x12 by the primary key alone leaves the uid without self-signature", K + U + self_sig(a, uid, sig_type=0x12), [False])
    add("a generic (0x10) self-certification is a self-signature", K + U + self_sig(a, uid, sig_type=0x10), [True], [], [a.key_id])
    add("two self-signatures: the LAST one's key flags (encrypt only) apply", K + U + ss + self_sig(a, uid, flags=0x0c, ctime=T0 + 5), [True], [], [])
    add("... in the other order the key can sign", K + U + self_sig(a, uid, flags=0x0c, ctime=T0 + 5) + ss, [True], [], [a.key_id])
    add("a self-signature that does not verify refuses the entity", K + U + ss + self_sig(a, uid, ctime=T0 + 1, spoil=True), [False])
    add("two uids: the first one's self-signature gives the flags", K + U + ss + U2 + self_sig(a, uid2, flags=0x0c), [True], [], [a.key_id])
    add("... unless a later one is the primary user id", K + U + ss + U2 + self_sig(a, uid2, flags=0x0c, primary=True), [True], [], [])
    add("the same uid twice: the later identity replaces the earlier with its signatures",
        K + U + ss + cert_b + U + self_sig(a, uid, ctime=T0 + 2) + certification(d, a, uid), [True], [d.key_id], [a.key_id])
    add("a certification without issuer subpacket: ReadEntity takes it, PGPCertificateInstance.Signers dereferences nil on it -- no verdict",
        K + U + ss + certification(b, a, uid, issuer=None), [None])
    add("signing subkey with its cross-signature", K + U + ss + S + binding(a, s, flags=0x02), [True], [], [a.key_id, s.key_id])
    add("signing subkey WITHOUT cross-signature", K + U + ss + S + binding(a, s, flags=0x02, cross=None), [False])
    add("signing subkey, cross-signature does not verify", K + U + ss + S + binding(a, s, flags=0x02, cross="bad"), [False])
    add("cross-signature naming another issuer: verified under the subkey all the same (VerifyKeySignature never reads the issuer)",
        K + U + ss + S + binding(a, s, flags=0x02, cross="wrong-issuer"), [True], [], [a.key_id, s.key_id])
    add("cross-signature without issuer subpacket: the same", K + U + ss + S + binding(a, s, flags=0x02, cross="no-issuer"), [True], [], [a.key_id, s.key_id])
    add("... and one that does not verify under the subkey is refused whatever it names",
        K + U + ss + S + binding(a, s, flags=0x02, cross="bad-wrong-issuer"), [False])
    add("binding that does not verify", K + U + ss + S + binding(a, s, spoil=True), [False])
    add("binding without issuer subpacket: verified under the primary key ReadEntity holds", K + U + ss + S + binding(a, s, issuer=None), [True], [], [a.key_id])
    add("binding naming another issuer: the same", K + U + ss + S + binding(a, s, issuer=b.key_id), [True], [], [a.key_id])
    add("binding without issuer subpacket that does not verify", K + U + ss + S + binding(a, s, issuer=None, spoil=True), [False])
    add("subkey without any signature", K + U + ss + S, [False])
    add("subkey followed by a certification: wrong type", K + U + ss + S + cert_b, [False])
    add("subkey with two bindings: the newer one's flags", K + U + ss + S + binding(a, s, flags=0x02) + binding(a, s, flags=0x0c, ctime=T0 + 9),
        [True], [], [a.key_id])
    add("... the older one coming second changes nothing", K + U + ss + S + binding(a, s, flags=0x0c, ctime=T0 + 9) + binding(a, s, flags=0x02),
        [True], [], [a.key_id])
    add("subkey revocation with a reason: the subkey is out", K + U + ss + S + binding(a, s, flags=0x02) + binding(a, s, sig_type=0x28, flags=None, reason=1),
        [True], [], [a.key_id])
    add("subkey revocation WITHOUT reason subpacket: Subkey.Sig has no flags and no reason -- the key stays usable",
        K + U + ss + S + binding(a, s, flags=0x0c) + binding(a, s, sig_type=0x28, flags=None), [True], [], [a.key_id, s.key_id])
    add("a binding after the revocation does not replace it",
        K + U + ss + S + binding(a, s, sig_type=0x28, flags=None, reason=1) + binding(a, s, flags=0x02, ctime=T0 + 20), [True], [], [a.key_id])
    add("key revocation before the uid: verified, the entity's keys are out", K + key_revocation(a) + U + ss, [True], [], [])
    add("key revocation that does not verify refuses the entity", K + key_revocation(a, spoil=True) + U + ss, [False])
    add("key revocation without issuer subpacket: verified under the primary key all the same", K + key_revocation(a, issuer=None) + U + ss, [True], [], [])
    add("... one naming another key that does not verify", K + key_revocation(a, issuer=b.key_id, spoil=True) + U + ss, [False])
    add("a 0x20 inside a uid's run is just one of its signatures", K + U + ss + key_revocation(a), [True], [a.key_id], [a.key_id])
    add("a version-3 signature ends the uid's run: what follows is ignored",
        K + U + ss + pkt(2, bytes([3, 5, 0x10]) + struct.pack(">I", T0) + struct.pack(">Q", b.key_id) + bytes([1, 8, 0, 0]) + cb.go_mpi_bytes(b"\x01" * 8)) + cert_b,
        [True], [], [a.key_id])
    add("an unknown packet type (trust, tag 12) is skipped: the run goes on", K + U + ss + pkt(12, b"\x00\x00") + cert_b, [True], [b.key_id], [a.key_id])
    add("a user attribute: a packet type left to the reference", K + U + ss + pkt(17, b"\x01\x01") + cert_b, [None])
    add("a signature that does not parse (critical unknown subpacket) refuses the entity",
        K + U + ss + sig_pkt(b, 0x10, key_framed(a) + uid_framed(uid), hashed_area(b.key_id, T0, sub(99, b"x", critical=True)), fake=True), [False])
    add("a packet before the key: Parse returns nothing", U + K + U + ss, [False])
    add("an unknown packet before the key is skipped", pkt(12, b"\x00") + K + U + ss, [True], [], [a.key_id])
    add("two entities", K + U + ss + cert_b + pkt(6, b.pub_body) + pkt(13, b.name.encode()) + self_sig(b, b.name.encode()), [True, True], [b.key_id], [a.key_id])
    add("the second entity refused: the first stands", K + U + ss + pkt(6, b.pub_body) + pkt(13, b.name.encode()), [True, False], [], [a.key_id])
    add("a subkey packet first: taken for the primary key", S + U + self_sig(s, uid), [True], [], [s.key_id])
    add("DSA primary", pkt(6, d.pub_body) + pkt(13, d.name.encode()) + self_sig(d, d.name.encode()) + certification(a, d, d.name.encode()),
        [True], [a.key_id], [d.key_id])
    add("an encrypt-only RSA primary cannot sign", pkt(6, a.pub_body[:5] + b"\x02" + a.pub_body[6:]) + U + ss, [False])
    add("an EdDSA primary: unsupported, the entity is refused", pkt(6, a.pub_body[:5] + b"\x16" + a.pub_body[6:]) + U + ss, [False])
    add("an ECDSA primary: left to the reference", pkt(6, a.pub_body[:5] + b"\x13" + a.pub_body[6:]) + U + ss, [None])
    add("an ElGamal subkey parses and needs its binding", K + U + ss + pkt(14, bytes([4]) + struct.pack(">I", T0) + bytes([16]) + cb._mpi(d.p) + cb._mpi(d.g) + cb._mpi(d.y)),
        [False])
    add("the uid cut off by the end of the certificate", (K + U)[:-3], [False])
    add("the last certification cut off inside its signature value", (K + U + ss + cert_b)[:-40], [False])
    add("an empty-bodied signature packet is the END of the stream for Reader.Next: the certification behind it is never read",
        K + U + ss + pkt(2, b"") + cert_b, [True], [], [a.key_id])
    add("a stray byte behind the last packet is an error of Next: the entity it ends is refused", K + U + ss + b"\x00", [False])
    add("an unknown packet type cut off by the end of the certificate is skipped like a whole one", K + U + ss + pkt(12, b"abcdef")[:-2], [True], [], [a.key_id])
    add("a second primary key packet that does not parse refuses only its own entity", K + U + ss + pkt(6, a.pub_body[:20]), [True, False], [], [a.key_id])
    # shapes left to the reference must never turn into a refusal by what they HIDE: the self-signature below is
    # real and x/crypto accepts the entity -- its 4.4 KB unhashed area is read through bufio like any other
    big_unhashed = b"".join(sub(100, bytes(150)) for _ in range(29))
    add("a valid self-signature with a 4.4 KB unhashed area: its body is over 4096 bytes -- no verdict, and NOT 'entity without any identities'",
        K + U + self_sig(a, uid, unhashed=big_unhashed), [None])
    add("... the same behind a first, ordinary self-signature", K + U + ss + self_sig(a, uid, ctime=T0 + 3, unhashed=big_unhashed) + cert_b, [None])
    add("a user id under a partial length, then its self-signature: no verdict (the walk ends at the partial length)",
        K + bytes([0xCD, 0xE4]) + uid[:16] + bytes([len(uid) - 16]) + uid[16:] + ss, [None])
    add("a subkey whose binding sits behind a signature body over 4096 bytes: no 'subkey packet not followed by signature'",
        K + U + ss + S + pkt(2, body_over_4096(a, s)) + binding(a, s), [None])
    add("a secret-key packet first: ReadEntity takes a private key for the primary key -- no verdict", pkt(5, a.pub_body + b"\x00" + bytes(16)) + U + ss, [None])
    add("... and the public entity behind it is walked on its own", pkt(5, a.pub_body + b"\x00" + bytes(16)) + U + ss + pkt(6, b.pub_body) + pkt(13, b.name.encode()) + self_sig(b, b.name.encode()),
        [None, True])
    add("a secret subkey in mid-entity", K + U + ss + pkt(7, s.pub_body + b"\x00" + bytes(16)) + binding(a, s), [None])
    add("embedded signatures nested three deep in a binding: x/crypto parses them recursively -- no verdict (the parser here is bounded)",
        K + U + ss + S + nested_binding(a, s, 3), [None])
    add("... two deep is followed: the inner ones must be 0x19, the outer cross-signature verifies", K + U + ss + S + nested_binding(a, s, 2), [True], [], [a.key_id, s.key_id])
    add("empty", b"", [])
    add("not a packet", bytes(range(200)), [])
    # a key packet with bytes behind its last MPI: key id, fingerprint and hashes are those of the key without them
    add("bytes behind the key material take no part", pkt(6, a.pub_body + b"junk") + U + ss + cert_b, [True], [b.key_id], [a.key_id])
    add("RSA exponent of four bytes: 'large public exponent'",
        pkt(6, bytes([4]) + struct.pack(">I", T0) + bytes([1]) + cb._mpi(a.n) + struct.pack(">H", 32) + b"\x00\x01\x00\x01") + U + ss, [False])
    return out


def random_blobs(n: int, seed: int = 7, real: bool = False):
    """Random packet sequences over the building blocks above (fake signature values: the walk's structure only -- or, `real`, true
    signatures, so that entities can come out VALID and the checks ReadEntity makes are decided by arithmetic), some with a byte
    changed afterwards."""
    import random
    rnd = random.Random(seed)
    a, b, s, s2, d = keys()
    uid, uid2 = a.name.encode(), b"other"
    fake = dict(fake=not real)
    body_of = lambda p: p[2 if p[1] < 192 else 3 if p[1] < 224 else 6:]      # a new-format packet without its header
    blocks = [
        lambda: pkt(6, a.pub_body), lambda: pkt(6, b.pub_body), lambda: pkt(6, d.pub_body),
        lambda: pkt(13, uid), lambda: pkt(13, uid), lambda: pkt(13, uid2), lambda: pkt(14, s.pub_body), lambda: pkt(14, s2.pub_body),
        lambda: self_sig(a, uid, **fake), lambda: self_sig(a, uid, **fake), lambda: self_sig(a, uid2, flags=rnd.choice([None, 1, 2, 0x0c]), ctime=T0 + rnd.randrange(4), primary=rnd.choice([None, True, False]), **fake),
        lambda: self_sig(a, uid, sig_type=rnd.choice([0x10, 0x11, 0x12, 0x13, 0x30, 0x18, 0x20]), **fake),
        lambda: certification(b, a, uid, **fake), lambda: certification(d, a, uid, issuer=rnd.choice(["signer", None, a.key_id]), **fake),
        lambda: binding(a, s, flags=rnd.choice([None, 0x0c, 0x02, 0x0e]), ctime=T0 + rnd.randrange(4), cross=rnd.choice(["auto", "auto", None, "wrong-issuer", "no-issuer"]),
                        issuer=rnd.choice(["primary", "primary", None, b.key_id]), **fake),
        lambda: binding(a, s2, sig_type=rnd.choice([0x18, 0x28, 0x28, 0x10]), flags=rnd.choice([None, 0x0c]), reason=rnd.choice([None, 1]), ctime=T0 + rnd.randrange(4), **fake),
        lambda: key_revocation(a, issuer=rnd.choice(["primary", "primary", None, b.key_id]), **fake),
        lambda: pkt(12, b"\x00\x00"), lambda: pkt(17, b"\x01\x01"), lambda: pkt(11, b"b\x00\x00\x00\x00\x00data"), lambda: pkt(61, b"?"),
        lambda: pkt(2, b""), lambda: pkt(2, bytes([3, 5, 0x10]) + struct.pack(">I", T0) + struct.pack(">Q", b.key_id) + bytes([1, 8, 0, 0]) + cb.go_mpi_bytes(b"\x01" * 8)),
        lambda: pkt(6, a.pub_body[:5] + bytes([rnd.choice([2, 16, 19, 22, 18])]) + a.pub_body[6:]), lambda: pkt(14, bytes([3]) + a.pub_body[1:]),
        lambda: pkt(6, a.pub_body + b"tail"), lambda: pkt(14, s.pub_body[:rnd.randrange(1, len(s.pub_body))]),
        lambda: bytes([0x88, 3]) + b"\x04\x00\x01",                      # old-format header, one length byte
        lambda: b"\x99" + struct.pack(">H", len(b.pub_body)) + b.pub_body,                       # old-format key packet (what gpg exports)
        lambda: b"\xb4" + bytes([len(uid)]) + uid,                                                # old-format user id
        lambda: b"\x89" + struct.pack(">H", len(body_of(self_sig(a, uid, **fake)))) + body_of(self_sig(a, uid, **fake)),   # old-format signature
        lambda: pkt(5, a.pub_body + b"\x00" + bytes(16)), lambda: pkt(7, s.pub_body + b"\x00" + bytes(16)),     # secret key / subkey packets
        lambda: pkt(6, b""), lambda: pkt(14, b""), lambda: pkt(13, b""),                             # empty bodies
        lambda: bytes([0xC6, 0xFF]) + struct.pack(">I", len(a.pub_body)) + a.pub_body,              # five-octet length
        lambda: bytes([0xCD, 0xC0, 0x10]) + bytes(208),                                              # two-octet length, a 208-byte user id
        lambda: bytes([0xC2, 0xE0]) + b"\x04",                          # partial length
        lambda: pkt(2, self_sig(a, uid, **fake)[3:] + bytes(4200)),      # a signature body over 4096 bytes
        lambda: self_sig(a, uid, unhashed=b"".join(sub(100, bytes(150)) for _ in range(29)), **fake),      # ... a well-formed one
        lambda: bytes([0xCD, 0xE4]) + uid[:16] + bytes([len(uid) - 16]) + uid[16:],                        # a user id under a partial length
        lambda: nested_binding(a, s, rnd.choice([2, 3, 3, 4])),                                             # embedded signatures 2 / 3 / 4 deep
        lambda: sig_pkt(b, 0x10, key_framed(a) + uid_framed(uid), hashed_area(b.key_id), embedded(sig_body(s, 0x19, b"", hashed_area(s.key_id), embedded(sig_body(s, 0x19, b"", hashed_area(s.key_id), embedded(sig_body(s, 0x19, b"", hashed_area(s.key_id), fake=True)), fake=True)), fake=True)), fake=True),
    ]
    out = []
    for i in range(n):
        seq = [rnd.choice(blocks)() for _ in range(rnd.randrange(1, 12))]
        if rnd.random() < 0.7:
            seq = [pkt(6, a.pub_body), pkt(13, uid), self_sig(a, uid, **fake)] + seq
        if real and rnd.random() < 0.5:          # (true signatures: keep half of the sequences short enough to come out valid)
            seq = seq[:3 + rnd.randrange(3)]
        blob = b"".join(seq)
        r = rnd.random()
        if r < 0.25 and blob:
            bb = bytearray(blob)
            for _ in range(rnd.randrange(1, 3)):
                bb[rnd.randrange(len(bb))] ^= 1 << rnd.randrange(8)
            blob = bytes(bb)
        elif r < 0.35 and blob:
            blob = blob[:rnd.randrange(len(blob))]
        out.append(blob)
    return out
