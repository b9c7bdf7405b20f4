#!/usr/bin/env python3
"""Generates tests/golden/gpg_messages.json -- pins the signed-message reader (oracle/message.py, the transport
message-signature row SURVEY.md 8(f)-2) against GnuPG 2.2.27.

Run in the build container:  python tests/golden/make_gpg_messages.py
  D. gpg keys + gpg-made signed messages (`gpg --sign -z 0`: one-pass signature, literal data, signature; definite and,
     for piped input, partial body lengths) with gpg's verdict on the intact and on a tampered message.
  E. corpus-generator keys + generator-made messages in both literal framings, intact and tampered, judged by gpg.
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
from tests.golden.make_gpg_vectors import gpg  # noqa: E402


def judge(home, msg):
    path = os.path.join(home, "m.gpg")
    with open(path, "wb") as f:
        f.write(msg)
    r = gpg(home, "--verify", path, ok=None)
    return r.returncode == 0


def main():
    out = {"gpg_version": subprocess.run(["gpg", "--version"], stdout=subprocess.PIPE).stdout.decode().splitlines()[0], "D": [], "E": []}
    home = tempfile.mkdtemp(prefix="gnupg")
    os.chmod(home, 0o700)
    try:
        for algo, uid in (("rsa2048", "m01 (http://localhost:5801) <m01@gpg.example>"), ("dsa2048", "m02 (http://localhost:5802) <m02@gpg.example>")):
            gpg(home, "--passphrase", "", "--faked-system-time", "20200101T000000", "--quick-gen-key", uid, algo, "sign,cert", "never")
        out["D_pubring"] = gpg(home, "--export").stdout.hex()
        payloads = [b"", b"tbs", bytes(range(256)) * 5, b"q" * 20000]
        for uid in ("m01@gpg.example", "m02@gpg.example"):
            for digest in ("SHA256", "SHA512"):
                for pl in payloads:
                    for piped in (False, True):
                        if piped:
                            msg = gpg(home, "--faked-system-time", "20200102T000000", "--digest-algo", digest, "-u", uid, "-z", "0", "--sign", "-o", "-", inp=pl).stdout
                        else:
                            src = os.path.join(home, "src")
                            with open(src, "wb") as f:
                                f.write(pl)
                            msg = gpg(home, "--faked-system-time", "20200102T000000", "--digest-algo", digest, "-u", uid, "-z", "0", "--sign", "-o", "-", src).stdout
                        good = judge(home, msg)
                        t = bytearray(msg)
                        t[len(t) // 2] ^= 0x01
                        out["D"].append({"signer": uid, "digest": digest, "payload": pl.hex(), "piped": piped, "msg": msg.hex(), "gpg_good": good,
                                         "tampered": bytes(t).hex(), "gpg_tampered_good": judge(home, bytes(t))})
    finally:
        subprocess.run(["gpgconf", "--homedir", home, "--kill", "gpg-agent"], stderr=subprocess.DEVNULL)
        shutil.rmtree(home, ignore_errors=True)
    cl = cb.make_cluster(4, dsa_fraction=0.5, n_outsiders=1)
    home2 = tempfile.mkdtemp(prefix="gnupg")
    os.chmod(home2, 0o700)
    try:
        ring = b"".join(r.entity for r in cl.replicas) + cl.client.entity
        gpg(home2, "--import", inp=ring, ok=None)
        out["E_pubring"] = ring.hex()
        rng = DRBG("gpgmsg")
        for kp in cl.replicas + [cl.client]:
            for pl in (b"", b"req", bytes(range(200)) * 3):
                for shape in ("go", "definite"):
                    msg = cb.signed_message(kp, pl, b"0123456789abcdef", rng, shape)
                    for tamper in (None, "body", "mpi"):
                        m = bytearray(msg)
                        if tamper == "body":
                            if not pl:
                                continue
                            m[len(m) - 287 - 3 if kp.algo == cb.PK_RSA else len(m) - 100] ^= 0x20
                        elif tamper == "mpi":
                            m[-3] ^= 0x10
                        out["E"].append({"key_id": "%016x" % kp.key_id, "algo": kp.algo, "shape": shape, "payload": pl.hex(), "tamper": tamper,
                                         "msg": bytes(m).hex(), "gpg_good": judge(home2, bytes(m))})
    finally:
        subprocess.run(["gpgconf", "--homedir", home2, "--kill", "gpg-agent"], stderr=subprocess.DEVNULL)
        shutil.rmtree(home2, ignore_errors=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpg_messages.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote %s: D %d (%d good, %d tampered-good), E %d (%d good)" % (path, len(out["D"]), sum(v["gpg_good"] for v in out["D"]),
          sum(v["gpg_tampered_good"] for v in out["D"]), len(out["E"]), sum(v["gpg_good"] for v in out["E"])))


if __name__ == "__main__":
    main()
