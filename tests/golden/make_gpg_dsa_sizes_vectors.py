#!/usr/bin/env python3
"""Generates tests/golden/gpg_dsa_sizes_vectors.json -- DSA at the group sizes gpg and the reference actually use, pinned on
GnuPG 2.2.27 / libgcrypt in both directions (round 5's vectors knew 2048/256 only).

dsa.Verify behind crypto/pgp/crypto_pgp.go:490 takes any (L, N); the reference's era and its threshold tests use 1024/160
(crypto/threshold/dsa/dsa_test.go:26-28).  What changes with the size is the digest truncation (the leftmost
ceil(bits(q) / 8) digest bytes) and, on the device, the multiplier width for p.

  A. gpg-made keys dsa1024 (q 160 bits) and dsa3072 (q 256) (gpg 2.2 rounds a 1536-bit request up to 2048/256, so a 224-bit q
     only exists in direction B) + gpg-made binary detached signatures under every
     digest gpg accepts for the key (not shorter than q): SHA-1 / SHA-224 / SHA-256 / SHA-384 / SHA-512, so that the truncation is
     hit from both sides (digest == q, digest > q); gpg's verdict on the intact and on a tampered payload.
  B. corpus-generator keys of 1024/160, 1536/224 and 3072/256 bits (corpus/keys.py) imported into gpg + Go-shaped detached signatures made by the
     generator under SHA-1 (q 160 only) / SHA-256 / SHA-512, intact and tampered (payload, r/s value) -> gpg's verdict.

Run in the build container (gpg is present there):  python tests/golden/make_gpg_dsa_sizes_vectors.py
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corpus import build as cb  # noqa: E402
from corpus.keys import DRBG  # noqa: E402


def gpg(home, *args, inp=None, ok=(0,)):
    r = subprocess.run(["gpg", "--homedir", home, "--batch", "--no-tty", "--quiet", *args], input=inp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if ok is not None and r.returncode not in ok:
        raise RuntimeError("gpg %s failed: %s" % (args, r.stderr.decode()))
    return r


def verdict(home, sig, payload):
    with open(os.path.join(home, "pl"), "wb") as f:
        f.write(payload)
    with open(os.path.join(home, "sg"), "wb") as f:
        f.write(sig)
    return gpg(home, "--verify", os.path.join(home, "sg"), os.path.join(home, "pl"), ok=None).returncode == 0


def main():
    vectors = {"gpg_version": subprocess.run(["gpg", "--version"], stdout=subprocess.PIPE).stdout.decode().splitlines()[0], "A": [], "B": []}
    payloads = [b"", b"tbs", bytes(range(256)) * 5, b"y" * 119]
    # ---- A: gpg keys, gpg signatures
    home = tempfile.mkdtemp(prefix="gnupg")
    os.chmod(home, 0o700)
    try:
        who = (("dsa1024", "d01@gpg.example", ("SHA1", "SHA224", "SHA256", "SHA512")),
               ("dsa3072", "d03@gpg.example", ("SHA256", "SHA384", "SHA512")))
        for algo, uid, _ in who:
            gpg(home, "--passphrase", "", "--faked-system-time", "20200101T000000", "--quick-gen-key",
                "%s (http://localhost:58%s) <%s>" % (uid[:3], uid[1:3], uid), algo, "sign,cert", "never")
        vectors["A_pubring"] = gpg(home, "--export").stdout.hex()
        for algo, uid, digests in who:
            for digest in digests:
                for pl in payloads:
                    sig = gpg(home, "--faked-system-time", "20200102T000000", "--digest-algo", digest, "-u", uid, "--detach-sign", "-o", "-", inp=pl).stdout
                    vectors["A"].append({"signer": uid, "key": algo, "digest": digest, "payload": pl.hex(), "sig": sig.hex(),
                                         "gpg_good": verdict(home, sig, pl), "gpg_tampered_good": verdict(home, sig, pl + b"!")})
    finally:
        subprocess.run(["gpgconf", "--homedir", home, "--kill", "gpg-agent"], stderr=subprocess.DEVNULL)
        shutil.rmtree(home, ignore_errors=True)
    # ---- B: generator keys + Go-shaped signatures, judged by gpg
    cl = cb.make_cluster(6, dsa_fraction=1.0, n_outsiders=0, dsa_kind=("dsa1024", "dsa3072", "dsa1536"))
    home2 = tempfile.mkdtemp(prefix="gnupg")
    os.chmod(home2, 0o700)
    try:
        ring = b"".join(r.entity for r in cl.replicas)
        vectors["B_pubring"] = ring.hex()
        vectors["B_import_rc"] = gpg(home2, "--import", inp=ring, ok=None).returncode
        rng = DRBG("gpgvec-dsa-sizes")
        for kp in cl.replicas:
            qbits = kp.q.bit_length()
            for hash_id in (2, 8, 10):
                if hash_id == 2 and qbits > 160:
                    continue          # gpg refuses a digest shorter than q
                for pl in (b"tbs", bytes(range(200)), b""):
                    sig = cb.detach_sign(kp, pl, rng, hash_id=hash_id)
                    for tamper in (None, "payload", "mpi"):
                        s, p = sig, pl
                        if tamper == "payload":
                            p = pl + b"?"
                        elif tamper == "mpi":
                            b = bytearray(sig); b[-3] ^= 0x10; s = bytes(b)
                        vectors["B"].append({"key_id": "%016x" % kp.key_id, "p_bits": kp.p.bit_length(), "q_bits": qbits, "hash_id": hash_id,
                                             "payload": p.hex(), "sig": s.hex(), "tamper": tamper, "gpg_good": verdict(home2, s, p)})
    finally:
        subprocess.run(["gpgconf", "--homedir", home2, "--kill", "gpg-agent"], stderr=subprocess.DEVNULL)
        shutil.rmtree(home2, ignore_errors=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpg_dsa_sizes_vectors.json")
    with open(out, "w") as f:
        json.dump(vectors, f)
    print("wrote %s: A %d vectors (%d good, %d tampered good), B %d vectors (%d good), import rc %s" %
          (out, len(vectors["A"]), sum(v["gpg_good"] for v in vectors["A"]), sum(v["gpg_tampered_good"] for v in vectors["A"]),
           len(vectors["B"]), sum(v["gpg_good"] for v in vectors["B"]), vectors["B_import_rc"]))


if __name__ == "__main__":
    main()
