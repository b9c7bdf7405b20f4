#!/usr/bin/env python3
"""Generates tests/golden/gpg_text_vectors.json: TEXT-MODE (signature type 0x01) detached signatures, made here over the
canonical form of the payload as x/crypto's NewCanonicalTextHash produces it (oracle/openpgp.py CanonicalTextHash: a '\\n'
that does not follow a '\\r' becomes "\\r\\n", the byte after a '\\r' passes unchanged) and judged by GnuPG 2.2.27.

`strict` vectors use payloads on which RFC 4880 5.2.1 leaves no latitude (lines ending in LF or CRLF, no trailing blanks, no
lone CR): gpg must accept them -- that pins the rewriting rule for everything the wild produces.  On the others (lone CR,
CR CR LF, trailing blanks) gpg's verdict is recorded as documentation; x/crypto's state machine decides for the oracle.

    python tests/golden/make_gpg_text_vectors.py      (build container: gpg present)
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
from corpus import build as cb            # noqa: E402
from corpus.keys import DRBG              # noqa: E402
from oracle import openpgp as pgp         # noqa: E402

CT = b"\x05\x02" + struct.pack(">I", cb.CREATION_TIME)
HASHES = {8: "sha256", 10: "sha512", 2: "sha1", 9: "sha384", 11: "sha224"}
PREFIX = {8: cb.SHA256_PREFIX, 10: bytes.fromhex("3051300d060960864801650304020305000440"), 2: bytes.fromhex("3021300906052b0e03021a05000414"),
          9: bytes.fromhex("3041300d060960864801650304020205000430"), 11: bytes.fromhex("302d300d06096086480165030402040500041c")}


def issuer(kp):
    return bytes([9, 16]) + struct.pack(">Q", kp.key_id)


def text_digest(payload, suffix, hash_id):
    h = pgp.CanonicalTextHash(hashlib.new(HASHES[hash_id]))
    h.update(payload)
    h.raw_update(suffix)
    return h.digest()


def text_sig(kp, payload, hash_id=8, rng=None, canonical=True):
    hashed = CT + issuer(kp)
    prefix = bytes([4, 1, kp.algo, hash_id]) + struct.pack(">H", len(hashed)) + hashed
    suffix = cb.hash_suffix(prefix)
    digest = text_digest(payload, suffix, hash_id) if canonical else hashlib.new(HASHES[hash_id], payload + suffix).digest()
    if kp.algo == cb.PK_RSA:
        t = PREFIX[hash_id] + digest
        em = int.from_bytes(b"\x00\x01" + b"\xff" * (256 - len(t) - 3) + b"\x00" + t, "big")
        mp = cb.go_mpi_bytes(kp.rsa_private(em).to_bytes(256, "big"))
    else:
        r, s = cb._dsa_sign(kp, digest, rng)
        mp = b"".join(cb.go_mpi_bytes(v.to_bytes((v.bit_length() + 7) // 8, "big")) for v in (r, s))
    body = prefix + b"\x00\x00" + digest[:2] + mp
    return cb._hdr(2, len(body)) + body


PAYLOADS = [
    # name, bytes, strict
    ("lf-lines", b"first line\nsecond line\nthird\n", True),
    ("crlf-lines", b"first line\r\nsecond line\r\n", True),
    ("mixed-lf-crlf", b"a\nb\r\nc\nd", True),
    ("no-line-end", b"plain text without a line end", True),
    ("empty", b"", True),
    ("only-lf", b"\n\n\n", True),
    ("long-lines", (b"x" * 200 + b"\n") * 40, True),
    ("lone-cr", b"a\rb\nc", False),
    ("cr-cr-lf", b"a\r\r\nb", False),
    ("cr-at-end", b"tail\r", False),
    ("trailing-blanks", b"a \nb\t\nc  ", False),
]


def gpg(home, *args, inp=None):
    return subprocess.run(["gpg", "--homedir", home, "--batch", "--no-tty", "--quiet", *args], input=inp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def main():
    cl = cb.make_cluster(4, n_outsiders=1)
    kp = cl.replicas[0]
    kd = [r for r in cb.make_cluster(4, dsa_fraction=0.5, n_outsiders=1).replicas if r.algo == cb.PK_DSA][0]
    rng = DRBG("text-vectors")
    home = tempfile.mkdtemp(prefix="gnupg")
    os.chmod(home, 0o700)
    out = {"gpg_version": subprocess.run(["gpg", "--version"], stdout=subprocess.PIPE).stdout.decode().splitlines()[0], "vectors": []}
    try:
        ring = b"".join(r.entity for r in cl.replicas) + kd.entity
        out["pubring"] = ring.hex()
        out["import_rc"] = gpg(home, "--import", inp=ring).returncode
        cases = []
        for name, pl, strict in PAYLOADS:
            cases.append((name, pl, text_sig(kp, pl), strict))
        cases.append(("lf-lines-sha512", PAYLOADS[0][1], text_sig(kp, PAYLOADS[0][1], hash_id=10), True))
        cases.append(("mixed-sha1", PAYLOADS[2][1], text_sig(kp, PAYLOADS[2][1], hash_id=2), True))
        cases.append(("mixed-sha384", PAYLOADS[2][1], text_sig(kp, PAYLOADS[2][1], hash_id=9), True))
        cases.append(("lf-lines-sha224", PAYLOADS[0][1], text_sig(kp, PAYLOADS[0][1], hash_id=11), True))
        cases.append(("lf-lines-dsa", PAYLOADS[0][1], text_sig(kd, PAYLOADS[0][1], rng=rng), True))
        cases.append(("long-lines-dsa-sha512", PAYLOADS[6][1], text_sig(kd, PAYLOADS[6][1], hash_id=10, rng=rng), True))
        cases.append(("lf-lines-hashed-raw", PAYLOADS[0][1], text_sig(kp, PAYLOADS[0][1], canonical=False), True))      # must be refused
        cases.append(("crlf-lines-tampered", PAYLOADS[1][1] + b"!", text_sig(kp, PAYLOADS[1][1]), True))                   # must be refused
        for name, pl, sig, strict in cases:
            with open(os.path.join(home, "pl"), "wb") as f:
                f.write(pl)
            with open(os.path.join(home, "sg"), "wb") as f:
                f.write(sig)
            r = gpg(home, "--verify", os.path.join(home, "sg"), os.path.join(home, "pl"))
            msg = r.stderr.decode(errors="replace").strip().splitlines()
            out["vectors"].append({"name": name, "payload": pl.hex(), "sig": sig.hex(), "strict": strict, "gpg_good": r.returncode == 0,
                                   "gpg_says": msg[-1][:120] if msg else ""})
    finally:
        subprocess.run(["gpgconf", "--homedir", home, "--kill", "gpg-agent"], stderr=subprocess.DEVNULL)
        shutil.rmtree(home, ignore_errors=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpg_text_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    for v in out["vectors"]:
        print("%-28s strict=%-5s gpg_good=%-5s %s" % (v["name"], v["strict"], v["gpg_good"], v["gpg_says"]))


if __name__ == "__main__":
    main()
