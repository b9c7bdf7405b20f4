#!/usr/bin/env python3
"""Generates tests/golden/gpg_weak_hash_vectors.json: MD5 and RIPEMD-160 signatures (RSA-2048, DSA-2048; binary and text mode)
built by hand and judged by GnuPG 2.2.27 where gpg has an opinion.

Two things are recorded here that only matter once a deployment declares these hashes available (bftkv_gpu_set_hash_policy,
oracle.openpgp.HASH_POLICY):
  * Go's RSA DigestInfo for RIPEMD-160 (crypto/rsa/pkcs1v15.go; the reference's copy: crypto/threshold/rsa/rsa.go:353) uses the
    ISO/IEC 10118-3 identifier, gpg the TeleTrusT one: "rsa-ripemd160-go-prefix" is what Go accepts and gpg refuses,
    "rsa-ripemd160-gpg-prefix" the other way round -- and gpg accepting the latter pins the RIPEMD-160 code itself (hashlib of
    this image has none).  DSA has no DigestInfo, but gpg 2.2 demands a hash at least as long as q (256 bits here) and
    refuses the 160-bit one with "General error" where Go's dsa.Verify takes what it gets (as with SHA-1).
  * gpg 2.2 refuses MD5 outright; those vectors are judged by the oracle alone (digest checked against hashlib's MD5).

    python tests/golden/make_gpg_weak_hash_vectors.py      (build container: gpg present)
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
NAMES = {1: "md5", 3: "ripemd160"}
GO_PREFIX = {1: pgp.HASH_PREFIXES["md5"], 3: pgp.HASH_PREFIXES["ripemd160"]}
GPG_RMD_PREFIX = bytes.fromhex("3021300906052b2403020105000414")


def sig(kp, payload, hash_id, *, sig_type=0, prefix_bytes=None, rng=None):
    hashed = CT + bytes([9, 16]) + struct.pack(">Q", kp.key_id)
    prefix = bytes([4, sig_type, kp.algo, hash_id]) + struct.pack(">H", len(hashed)) + hashed
    h = pgp.CanonicalTextHash(pgp.new_hash(NAMES[hash_id])) if sig_type == 1 else pgp._BinaryHash(pgp.new_hash(NAMES[hash_id]))
    h.update(payload)
    h.raw_update(cb.hash_suffix(prefix))
    digest = h.digest()
    if kp.algo == cb.PK_RSA:
        t = (prefix_bytes or GO_PREFIX[hash_id]) + digest
        em = int.from_bytes(b"\x00\x01" + b"\xff" * (256 - len(t) - 3) + b"\x00" + t, "big")
        mp = cb.go_mpi_bytes(kp.rsa_private(em).to_bytes(256, "big"))
    else:
        r, s = cb._dsa_sign(kp, digest, rng)
        mp = b"".join(cb.go_mpi_bytes(v.to_bytes((v.bit_length() + 7) // 8, "big")) for v in (r, s))
    body = prefix + b"\x00\x00" + digest[:2] + mp
    return cb._hdr(2, len(body)) + body


def gpg(home, *args, inp=None):
    return subprocess.run(["gpg", "--homedir", home, "--batch", "--no-tty", "--quiet", *args], input=inp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def main():
    assert hashlib.md5(b"abc").hexdigest() == "900150983cd24fb0d6963f7d28e17f72"
    assert pgp.new_hash("ripemd160").__class__ and pgp._Ripemd160(b"abc").digest().hex() == "8eb208f7e05d987a9b044a8e98c6b087f15a0bfc"
    cl = cb.make_cluster(4, n_outsiders=1)
    kp = cl.replicas[0]
    kd = [r for r in cb.make_cluster(4, dsa_fraction=0.5, n_outsiders=1).replicas if r.algo == cb.PK_DSA][0]
    rng = DRBG("weak-hash-vectors")
    pl = b"weak-hash vector payload\nsecond line\n" + b"z" * 300
    cases = [
        # name, payload, signature, strict (gpg's verdict must be the oracle's when the hash is declared available)
        ("dsa-ripemd160", pl, sig(kd, pl, 3, rng=rng), False),
        ("dsa-ripemd160-text", pl, sig(kd, pl, 3, sig_type=1, rng=rng), False),
        ("dsa-ripemd160-tampered", pl + b"!", sig(kd, pl, 3, rng=rng), False),
        ("rsa-ripemd160-go-prefix", pl, sig(kp, pl, 3), False),
        ("rsa-ripemd160-gpg-prefix", pl, sig(kp, pl, 3, prefix_bytes=GPG_RMD_PREFIX), False),
        ("rsa-md5", pl, sig(kp, pl, 1), False),
        ("rsa-md5-text", pl, sig(kp, pl, 1, sig_type=1), False),
        ("dsa-md5", pl, sig(kd, pl, 1, rng=rng), False),
        ("rsa-md5-tampered", pl + b"!", sig(kp, pl, 1), False),
        ("rsa-md5-short-payload", b"abc", sig(kp, b"abc", 1), False),
        ("rsa-ripemd160-empty-payload", b"", sig(kp, b"", 3), False),
    ]
    home = tempfile.mkdtemp(prefix="gnupg")
    os.chmod(home, 0o700)
    out = {"gpg_version": subprocess.run(["gpg", "--version"], stdout=subprocess.PIPE).stdout.decode().splitlines()[0], "vectors": []}
    try:
        ring = b"".join(r.entity for r in cl.replicas) + kd.entity
        out["pubring"] = ring.hex()
        out["import_rc"] = gpg(home, "--import", inp=ring).returncode
        for name, payload, s, strict in cases:
            with open(os.path.join(home, "pl"), "wb") as f:
                f.write(payload)
            with open(os.path.join(home, "sg"), "wb") as f:
                f.write(s)
            r = gpg(home, "--verify", os.path.join(home, "sg"), os.path.join(home, "pl"))
            msg = r.stderr.decode(errors="replace").strip().splitlines()
            out["vectors"].append({"name": name, "payload": payload.hex(), "sig": s.hex(), "strict": strict, "gpg_good": r.returncode == 0,
                                   "gpg_says": msg[-1][:120] if msg else ""})
    finally:
        subprocess.run(["gpgconf", "--homedir", home, "--kill", "gpg-agent"], stderr=subprocess.DEVNULL)
        shutil.rmtree(home, ignore_errors=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpg_weak_hash_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    for v in out["vectors"]:
        print("%-30s strict=%-5s gpg_good=%-5s %s" % (v["name"], v["strict"], v["gpg_good"], v["gpg_says"]))


if __name__ == "__main__":
    main()
