#!/usr/bin/env python3
"""Generates tests/golden/gpg_vectors.json -- the fixtures that PIN the oracle (and the corpus
generator) against an independent OpenPGP implementation, GnuPG 2.2.27 / libgcrypt 1.9.4.

Run in the build container (gpg is present there):  python tests/golden/make_gpg_vectors.py
Two directions:
  A. gpg-made keys + gpg-made binary detached signatures (RSA-2048 and DSA-2048/256; SHA-256,
     SHA-512, SHA-1; old-format headers, issuer in the unhashed area) with gpg's own verdict on the
     intact and on a tampered payload -> the oracle must agree.
  B. corpus-generator keys (corpus/build.py) imported into gpg + Go-shaped detached signatures made
     by the generator (and tampered variants) -> gpg's verdict is recorded; the oracle must agree.
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
    r = subprocess.run(["gpg", "--homedir", home, "--batch", "--no-tty", "--quiet", *args], input=inp,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if ok is not None and r.returncode not in ok:
        raise RuntimeError("gpg %s failed: %s" % (args, r.stderr.decode()))
    return r


def main():
    home = tempfile.mkdtemp(prefix="gnupg")
    os.chmod(home, 0o700)
    vectors = {"gpg_version": subprocess.run(["gpg", "--version"], stdout=subprocess.PIPE).stdout.decode().splitlines()[0],
               "A": [], "B": []}
    try:
        # ---- A: gpg keys, gpg signatures
        for algo, uid in (("rsa2048", "a01 (http://localhost:5701) <a01@gpg.example>"),
                          ("dsa2048", "a02 (http://localhost:5702) <a02@gpg.example>")):
            gpg(home, "--passphrase", "", "--faked-system-time", "20200101T000000", "--quick-gen-key", uid, algo,
                "sign,cert", "never")
        pub = gpg(home, "--export").stdout
        vectors["A_pubring"] = pub.hex()
        payloads = [b"", b"tbs", bytes(range(256)) * 5, b"x" * 64, b"y" * 119, b"z" * 120]
        for uid in ("a01@gpg.example", "a02@gpg.example"):
            for digest in ("SHA256", "SHA512", "SHA1", "SHA384", "SHA224"):
                if uid.startswith("a02") and digest in ("SHA1", "SHA224"):
                    continue  # gpg refuses digests shorter than q (256 bits) for DSA-2048
                for pl in payloads[:3] if digest != "SHA256" else payloads:
                    sig = gpg(home, "--faked-system-time", "20200102T000000", "--digest-algo", digest, "-u", uid,
                              "--detach-sign", "-o", "-", inp=pl).stdout
                    with open(os.path.join(home, "pl"), "wb") as f:
                        f.write(pl)
                    with open(os.path.join(home, "sg"), "wb") as f:
                        f.write(sig)
                    good = gpg(home, "--verify", os.path.join(home, "sg"), os.path.join(home, "pl"), ok=None).returncode == 0
                    with open(os.path.join(home, "pl"), "wb") as f:
                        f.write(pl + b"!")
                    bad = gpg(home, "--verify", os.path.join(home, "sg"), os.path.join(home, "pl"), ok=None).returncode == 0
                    vectors["A"].append({"signer": uid, "digest": digest, "payload": pl.hex(), "sig": sig.hex(),
                                         "gpg_good": good, "gpg_tampered_good": bad})
        # ---- C: larger RSA moduli (gpg 2.2's default is rsa3072), same checks as A
        home3 = tempfile.mkdtemp(prefix="gnupg")
        os.chmod(home3, 0o700)
        try:
            vectors["C"] = []
            for algo, uid in (("rsa3072", "a03 (http://localhost:5703) <a03@gpg.example>"), ("rsa4096", "a04 (http://localhost:5704) <a04@gpg.example>")):
                gpg(home3, "--passphrase", "", "--faked-system-time", "20200101T000000", "--quick-gen-key", uid, algo, "sign,cert", "never")
            vectors["C_pubring"] = gpg(home3, "--export").stdout.hex()
            for uid in ("a03@gpg.example", "a04@gpg.example"):
                for digest in ("SHA256", "SHA512"):
                    for pl in payloads[:4]:
                        sig = gpg(home3, "--faked-system-time", "20200102T000000", "--digest-algo", digest, "-u", uid, "--detach-sign", "-o", "-", inp=pl).stdout
                        with open(os.path.join(home3, "pl"), "wb") as f:
                            f.write(pl)
                        with open(os.path.join(home3, "sg"), "wb") as f:
                            f.write(sig)
                        good = gpg(home3, "--verify", os.path.join(home3, "sg"), os.path.join(home3, "pl"), ok=None).returncode == 0
                        with open(os.path.join(home3, "pl"), "wb") as f:
                            f.write(pl + b"!")
                        bad = gpg(home3, "--verify", os.path.join(home3, "sg"), os.path.join(home3, "pl"), ok=None).returncode == 0
                        vectors["C"].append({"signer": uid, "digest": digest, "payload": pl.hex(), "sig": sig.hex(), "gpg_good": good,
                                             "gpg_tampered_good": bad})
        finally:
            subprocess.run(["gpgconf", "--homedir", home3, "--kill", "gpg-agent"], stderr=subprocess.DEVNULL)
            shutil.rmtree(home3, ignore_errors=True)
        # ---- B: generator keys + Go-shaped signatures, judged by gpg
        cl = cb.make_cluster(4, dsa_fraction=0.5, n_outsiders=1)
        home2 = tempfile.mkdtemp(prefix="gnupg")
        os.chmod(home2, 0o700)
        try:
            ring = b"".join(r.entity for r in cl.replicas) + cl.client.entity
            imp = gpg(home2, "--import", inp=ring, ok=None)
            vectors["B_pubring"] = ring.hex()
            vectors["B_import_rc"] = imp.returncode
            rng = DRBG("gpgvec")
            for kp in cl.replicas + [cl.client]:
                for pl in (b"tbs", bytes(range(200)), b""):
                    sig = cb.detach_sign(kp, pl, rng)
                    for tamper in (None, "payload", "mpi"):
                        s, p = sig, pl
                        if tamper == "payload":
                            p = pl + b"?"
                        elif tamper == "mpi":
                            b = bytearray(sig); b[-3] ^= 0x10; s = bytes(b)
                        with open(os.path.join(home2, "pl"), "wb") as f:
                            f.write(p)
                        with open(os.path.join(home2, "sg"), "wb") as f:
                            f.write(s)
                        rc = gpg(home2, "--verify", os.path.join(home2, "sg"), os.path.join(home2, "pl"), ok=None).returncode
                        vectors["B"].append({"key_id": "%016x" % kp.key_id, "algo": kp.algo, "payload": p.hex(),
                                             "sig": s.hex(), "tamper": tamper, "gpg_good": rc == 0})
        finally:
            subprocess.run(["gpgconf", "--homedir", home2, "--kill", "gpg-agent"], stderr=subprocess.DEVNULL)
            shutil.rmtree(home2, ignore_errors=True)
    finally:
        subprocess.run(["gpgconf", "--homedir", home, "--kill", "gpg-agent"], stderr=subprocess.DEVNULL)
        shutil.rmtree(home, ignore_errors=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpg_vectors.json")
    with open(out, "w") as f:
        json.dump(vectors, f)
    na = sum(v["gpg_good"] for v in vectors["A"]); nb = sum(v["gpg_good"] for v in vectors["B"])
    print("wrote %s: A %d vectors (%d good), B %d vectors (%d good), C %d vectors, import rc %s" %
          (out, len(vectors["A"]), na, len(vectors["B"]), nb, len(vectors["C"]), vectors["B_import_rc"]))


if __name__ == "__main__":
    main()
