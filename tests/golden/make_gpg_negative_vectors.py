#!/usr/bin/env python3
"""Generates tests/golden/gpg_negative_vectors.json: single-signature EDGE CASES built by hand (corpus-generator RSA-2048 and
DSA-2048 keys) and judged by GnuPG 2.2.27 -- wrong hash id in the packet, MPI encodings, critical / unknown subpackets, where
the issuer and the creation time live, embedded signatures, SignatureV3, text mode, a value beyond the modulus.

Each vector records gpg's verdict and whether RFC 4880 leaves NO latitude for it (`strict`): the oracle must agree with gpg
on every strict vector (tests/test_oracle_golden.py); on the others -- where x/crypto and gpg are both entitled to their own
behaviour -- gpg's verdict is recorded beside the oracle's as documentation (DESIGN.md section 5 lists which x/crypto rules
the restatement rests on).

    python tests/golden/make_gpg_negative_vectors.py      (build container: gpg present)
"""
import hashlib
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corpus import build as cb  # noqa: E402
from corpus.keys import DRBG  # noqa: E402

CT = b"\x05\x02" + struct.pack(">I", cb.CREATION_TIME)
HASHES = {8: hashlib.sha256, 10: hashlib.sha512, 2: hashlib.sha1}
PREFIX = {8: cb.SHA256_PREFIX, 10: bytes.fromhex("3051300d060960864801650304020305000440"), 2: bytes.fromhex("3021300906052b0e03021a05000414")}


def sub(typ, body, critical=False):
    n = len(body) + 1
    ln = bytes([n]) if n < 192 else bytes([((n - 192) >> 8) + 192, (n - 192) & 0xFF])
    return ln + bytes([typ | (0x80 if critical else 0)]) + body


def iss(kp):
    return sub(16, struct.pack(">Q", kp.key_id))


def rsa_value(kp, digest, hash_id):
    t = PREFIX[hash_id] + digest
    em = int.from_bytes(b"\x00\x01" + b"\xff" * (256 - len(t) - 3) + b"\x00" + t, "big")
    return kp.rsa_private(em)


def v4(kp, payload, *, sig_type=0, hash_id=8, digest_hash=None, hashed=None, unhashed=b"", tag=None, mpi=None, value_add=0):
    hashed = (CT + iss(kp)) if hashed is None else hashed
    prefix = bytes([4, sig_type, kp.algo, hash_id]) + struct.pack(">H", len(hashed)) + hashed
    h = HASHES[digest_hash or hash_id]
    digest = h(payload + cb.hash_suffix(prefix)).digest()
    s = rsa_value(kp, digest, digest_hash or hash_id) + value_add
    sb = s.to_bytes(max(256, (s.bit_length() + 7) // 8), "big")
    m = mpi(sb) if mpi else cb.go_mpi_bytes(sb)
    body = prefix + struct.pack(">H", len(unhashed)) + unhashed + (tag or digest[:2]) + m
    return cb._hdr(2, len(body)) + body


def v3(kp, payload, sig_type=0):
    d = hashlib.sha256(payload + bytes([sig_type]) + struct.pack(">I", cb.CREATION_TIME)).digest()
    body = (bytes([3, 5, sig_type]) + struct.pack(">I", cb.CREATION_TIME) + struct.pack(">Q", kp.key_id) + bytes([kp.algo, 8]) + d[:2] +
            cb.go_mpi_bytes(rsa_value(kp, d, 8).to_bytes(256, "big")))
    return cb._hdr(2, len(body)) + body


def dsa_v4(kp, payload, rng, *, hash_id=8, r_add=0, s_add=0, r_set=None, s_set=None, lead_zero=False):
    hashed = CT + iss(kp)
    prefix = bytes([4, 0, kp.algo, hash_id]) + struct.pack(">H", len(hashed)) + hashed
    digest = HASHES[hash_id](payload + cb.hash_suffix(prefix)).digest()
    r, s = cb._dsa_sign(kp, digest, rng)            # z = leftmost min(len(digest), bytes(q)) bytes, as dsa.Verify takes it
    r = (r + r_add) if r_set is None else r_set
    s = (s + s_add) if s_set is None else s_set
    enc = b""
    for v in (r, s):
        b = v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")
        enc += (struct.pack(">H", 8 * len(b) + 8) + b"\x00" + b) if lead_zero else cb.go_mpi_bytes(b)
    body = prefix + b"\x00\x00" + digest[:2] + enc
    return cb._hdr(2, len(body)) + body


def dsa_cases(kd, rng):
    pl = b"negative-vector payload\n with a line ending"
    return [
        ("dsa-control-good", dsa_v4(kd, pl, rng), True, "DSA-2048/256 over SHA-256 as DetachSign emits it"),
        ("dsa-r-zero", dsa_v4(kd, pl, rng, r_set=0), True, "FIPS 186: 0 < r < q"),
        ("dsa-s-zero", dsa_v4(kd, pl, rng, s_set=0), True, "FIPS 186: 0 < s < q"),
        ("dsa-r-plus-q", dsa_v4(kd, pl, rng, r_add=kd.q), True, "r >= q: equal mod q, must be refused before any arithmetic"),
        ("dsa-s-plus-q", dsa_v4(kd, pl, rng, s_add=kd.q), True, "s >= q"),
        ("dsa-leading-zero-mpis", dsa_v4(kd, pl, rng, lead_zero=True), False, "same values, non-canonical encoding"),
        ("dsa-sha512-truncated", dsa_v4(kd, pl, rng, hash_id=10), True, "hash longer than q: leftmost 256 bits (FIPS 186-3 4.6)"),
        ("dsa-sha1-shorter-than-q", dsa_v4(kd, pl, rng, hash_id=2), False, "160-bit hash under a 256-bit q: gpg demands >= 256 bits, dsa.Verify takes what it gets"),
        ("dsa-wrong-payload", dsa_v4(kd, pl + b"!", rng), True, "signature over other bytes"),
    ]


def cases(kp, other):
    pl = b"negative-vector payload\n with a line ending"
    inner19 = cb.sig_prefix(0x19, other.algo, CT + iss(other)) + b"\x00\x00\xab\xcd" + cb.go_mpi_bytes(b"\x5a" * 256)
    canon = lambda sb: struct.pack(">H", int.from_bytes(sb, "big").bit_length()) + sb.lstrip(b"\x00")
    c = [
        # name, signature bytes, strict (RFC 4880 leaves no latitude), note
        ("control-good", v4(kp, pl), True, "Go-shaped signature as DetachSign emits it"),
        ("control-canonical-mpi", v4(kp, pl, mpi=canon), True, "MPI with the true bit count"),
        ("wrong-hash-id-in-packet", v4(kp, pl, hash_id=10, digest_hash=8), True, "packet says SHA-512, value and tag are SHA-256's"),
        ("wrong-hash-tag", v4(kp, pl, tag=b"\x00\x00"), False, "left 16 bits of the hash do not match: x/crypto refuses, gpg 2.2 only uses them as a hint"),
        ("mpi-bitcount-short-by-8", v4(kp, pl, mpi=lambda sb: struct.pack(">H", 8 * len(sb) - 8) + sb), True, "the reader takes one byte less: another value"),
        ("mpi-leading-zero-byte", v4(kp, pl, mpi=lambda sb: struct.pack(">H", 8 * len(sb) + 8) + b"\x00" + sb), False, "same value, non-canonical encoding"),
        ("value-plus-modulus", v4(kp, pl, value_add=kp.n), False, "s + n: equal mod n; Go <= 1.13 has no s < n check"),
        ("critical-unknown-subpacket", v4(kp, pl, hashed=CT + sub(101, b"\x01", critical=True) + iss(kp)), True, "5.2.3.1: unknown critical => invalid"),
        ("noncritical-unknown-subpacket", v4(kp, pl, hashed=CT + sub(101, b"\x01") + iss(kp)), True, "ignored"),
        ("issuer-only-unhashed", v4(kp, pl, hashed=CT, unhashed=iss(kp)), True, "what gpg itself emits"),
        ("no-issuer-at-all", v4(kp, pl, hashed=CT), False, "gpg tries its keys; x/crypto cannot look one up"),
        ("no-creation-time", v4(kp, pl, hashed=iss(kp)), False, "5.2.3.4 MUST be present; enforcement differs"),
        ("creation-time-unhashed-only", v4(kp, pl, hashed=iss(kp), unhashed=CT), False, ""),
        ("creation-time-in-both-areas", v4(kp, pl, unhashed=CT), False, "x/crypto: 'signature creation time in non-hashed area'"),
        ("embedded-0x19-unhashed", v4(kp, pl, unhashed=sub(32, inner19)), False, "cross-certification shape inside a data signature"),
        ("embedded-twice", v4(kp, pl, hashed=CT + iss(kp) + sub(32, inner19), unhashed=sub(32, inner19)), False, "x/crypto: 'Cannot have multiple embedded signatures'"),
        ("embedded-wrong-type", v4(kp, pl, unhashed=sub(32, cb.detach_sign(other, b"i")[3:])), False, "x/crypto: 'cross-signature has unexpected type'"),
        ("signature-v3", v3(kp, pl), False, "fenced here (fenced_out): gpg's verdict documents the construction"),
        ("text-mode", v4(kp, pl, sig_type=1), False, "canonical-text hashing; fenced here"),
        ("standalone-type-2", v4(kp, pl, sig_type=2), False, "hashForSignature refuses it in x/crypto"),
        ("truncated-packet", v4(kp, pl)[:200], True, ""),
        ("wrong-payload", v4(kp, pl + b"!"), True, "signature over other bytes"),
    ]
    return pl, c


def gpg(home, *args, inp=None):
    return subprocess.run(["gpg", "--homedir", home, "--batch", "--no-tty", "--quiet", *args], input=inp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def main():
    cl = cb.make_cluster(4, n_outsiders=1)
    kp, other = cl.replicas[0], cl.replicas[1]
    kd = [r for r in cb.make_cluster(4, dsa_fraction=0.5, n_outsiders=1).replicas if r.algo == cb.PK_DSA][0]
    home = tempfile.mkdtemp(prefix="gnupg")
    os.chmod(home, 0o700)
    out = {"gpg_version": subprocess.run(["gpg", "--version"], stdout=subprocess.PIPE).stdout.decode().splitlines()[0], "vectors": []}
    try:
        ring = b"".join(r.entity for r in cl.replicas) + kd.entity
        imp = gpg(home, "--import", inp=ring)
        out["pubring"] = ring.hex()
        out["import_rc"] = imp.returncode
        pl, cs = cases(kp, other)
        cs = cs + dsa_cases(kd, DRBG("negative-dsa"))
        out["payload"] = pl.hex()
        for name, sig, strict, note in cs:
            with open(os.path.join(home, "pl"), "wb") as f:
                f.write(pl)
            with open(os.path.join(home, "sg"), "wb") as f:
                f.write(sig)
            r = gpg(home, "--verify", os.path.join(home, "sg"), os.path.join(home, "pl"))
            msg = r.stderr.decode(errors="replace").strip().splitlines()
            out["vectors"].append({"name": name, "sig": sig.hex(), "strict": strict, "note": note, "gpg_good": r.returncode == 0,
                                   "gpg_says": msg[-1][:120] if msg else ""})
    finally:
        subprocess.run(["gpgconf", "--homedir", home, "--kill", "gpg-agent"], stderr=subprocess.DEVNULL)
        shutil.rmtree(home, ignore_errors=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpg_negative_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    for v in out["vectors"]:
        print("%-32s strict=%-5s gpg_good=%-5s %s" % (v["name"], v["strict"], v["gpg_good"], v["gpg_says"]))


if __name__ == "__main__":
    main()
