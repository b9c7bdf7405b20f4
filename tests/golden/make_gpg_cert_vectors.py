#!/usr/bin/env python3
"""Generates tests/golden/gpg_cert_vectors.json: certificates GnuPG 2.2 made, for the ReadEntity walk
(oracle/openpgp.py walk_certificate; bftkv_host_certs_parse / bftkv_host_certs_verify).

Run in the build container (gpg is present there):  python tests/golden/make_gpg_cert_vectors.py

Every shape a deployment's keys take when their owner goes beyond scripts/gen.sh's `--quick-gen-key ... default default never`:
the default key itself ([SC] primary, [E] subkey), an added SIGNING subkey (its binding signature carries the subkey's 0x19
cross-signature in the unhashed area -- the one place x/crypto's ReadEntity verifies a signature with a key other than the
primary), several user ids with a primary-uid flag, a revoked user id, a changed expiry (a second self-signature), a revoked
subkey (0x28 with a reason), a revoked key (0x20), a certified key, a DSA key with an ElGamal subkey.  Recorded beside each
blob: what `gpg --with-colons --list-sigs` says -- the primary key id, the keys gpg would sign with (capability "s", not
revoked), and the issuers of every signature on a user id other than the 0x10..0x13 self-signatures (what x/crypto collects in
identity.Signatures and bftkv's Signers() returns).  `tampered`: the signing-subkey certificate with its cross-signature
spoiled / removed (gpg: "signing subkey ... is not cross-certified" -- x/crypto refuses the entity)."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
T0, T1, T2 = "20200101T000000", "20200102T000000", "20200103T000000"


def gpg(home, *args, inp=None, ok=(0,), when=T0):
    r = subprocess.run(["gpg", "--homedir", home, "--batch", "--no-tty", "--quiet", "--pinentry-mode", "loopback", "--passphrase", "",
                        "--faked-system-time", when, *args], input=inp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if ok is not None and r.returncode not in ok:
        raise RuntimeError("gpg %s failed: %s" % (args, r.stderr.decode()))
    return r


def new_home():
    h = tempfile.mkdtemp(prefix="gnupg")
    os.chmod(h, 0o700)
    return h


def fpr_of(home, uid):
    for ln in gpg(home, "--with-colons", "--list-keys", uid).stdout.decode().splitlines():
        if ln.startswith("fpr:"):
            return ln.split(":")[9]
    raise RuntimeError("no key " + uid)


def describe(home, fpr):
    """(primary key id, signing-capable key ids, signers per x/crypto's identity.Signatures)."""
    primary, signing, signers = None, [], []
    under_uid = False
    for ln in gpg(home, "--with-colons", "--list-sigs", fpr).stdout.decode().splitlines():
        f = ln.split(":")
        if f[0] in ("pub", "sub"):
            under_uid = False
            if f[0] == "pub":
                primary = f[4]
            if "s" in f[11] and f[1] not in ("r", "e", "i", "d"):
                signing.append(f[4])
        elif f[0] == "uid":
            under_uid = True
        elif f[0] in ("sig", "rev") and under_uid:
            cls = f[10][:2]
            if not (f[4] == primary and cls in ("10", "13")):
                signers.append(f[4])
    # a revoked primary key takes every key of the entity out (Entity.Revocations)
    for ln in gpg(home, "--with-colons", "--list-keys", fpr).stdout.decode().splitlines():
        f = ln.split(":")
        if f[0] == "pub" and f[1] == "r":
            signing = []
    return primary, signing, signers


def edit(home, fpr, script, when=T1):
    r = subprocess.run(["gpg", "--homedir", home, "--no-tty", "--pinentry-mode", "loopback", "--passphrase", "", "--faked-system-time", when,
                        "--command-fd", "0", "--status-fd", "2", "--edit-key", fpr], input=script.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise RuntimeError("gpg --edit-key failed: %s" % r.stderr.decode())


def main():
    out = {"gpg_version": subprocess.run(["gpg", "--version"], stdout=subprocess.PIPE).stdout.decode().splitlines()[0], "certificates": [], "tampered": []}
    homes = []

    def record(name, home, fpr, sign_with=()):
        blob = gpg(home, "--export", fpr).stdout
        primary, signing, signers = describe(home, fpr)
        rec = {"name": name, "blob": blob.hex(), "primary_key_id": primary, "signing_key_ids": signing, "signers": signers, "detached": []}
        # binary detached signatures made with a NAMED key of the certificate ("keyid!": no automatic choice of the newest signing
        # subkey), judged by gpg on the intact and on a changed payload
        for kid in sign_with:
            pl = b"signed with " + kid.encode()
            sig = gpg(home, "--digest-algo", "SHA256", "-u", kid + "!", "--detach-sign", "-o", "-", inp=pl, when=T2).stdout
            verdicts = []
            for data in (pl, pl + b"!"):
                with open(os.path.join(home, "pl"), "wb") as f:
                    f.write(data)
                with open(os.path.join(home, "sg"), "wb") as f:
                    f.write(sig)
                verdicts.append(gpg(home, "--verify", os.path.join(home, "sg"), os.path.join(home, "pl"), ok=None, when=T2).returncode == 0)
            rec["detached"].append({"key_id": kid, "payload": pl.hex(), "sig": sig.hex(), "gpg_good": verdicts[0], "gpg_tampered_good": verdicts[1]})
        out["certificates"].append(rec)
        return blob

    def keys_of(home, fpr):
        return [ln.split(":")[4] for ln in gpg(home, "--with-colons", "--list-keys", fpr).stdout.decode().splitlines() if ln[:4] in ("pub:", "sub:")]

    try:
        # 1. scripts/gen.sh's command, word for word (gpg 2.2's "default": rsa3072 [SC] + rsa3072 [E])
        h = new_home(); homes.append(h)
        gpg(h, "--quick-gen-key", "a01 (http://localhost:5701) <a01@gpg.example>", "default", "default", "never")
        record("gen.sh default key", h, fpr_of(h, "a01@gpg.example"))
        # 2. an added signing subkey: binding signature with the embedded cross-signature
        h = new_home(); homes.append(h)
        gpg(h, "--quick-gen-key", "s01 <s01@gpg.example>", "rsa2048", "sign,cert", "never")
        f = fpr_of(h, "s01@gpg.example")
        gpg(h, "--quick-add-key", f, "rsa2048", "sign", "never", when=T1)
        gpg(h, "--quick-add-key", f, "rsa2048", "encr", "never", when=T1)
        ks = keys_of(h, f)
        signing_blob = record("signing subkey (cross-signature) and an encryption subkey", h, f, sign_with=ks[:2])
        # 3. DSA signing subkey under an RSA primary
        gpg(h, "--quick-add-key", f, "dsa2048", "sign", "never", when=T2)
        record("... plus a DSA signing subkey", h, f, sign_with=[keys_of(h, f)[-1]])
        # 4. two user ids, the second flagged primary; then the first revoked
        h = new_home(); homes.append(h)
        gpg(h, "--quick-gen-key", "u01 <u01@gpg.example>", "rsa2048", "sign,cert", "never")
        f = fpr_of(h, "u01@gpg.example")
        gpg(h, "--quick-add-uid", f, "u01 second <u01b@gpg.example>", when=T1)
        gpg(h, "--quick-set-primary-uid", f, "u01 second <u01b@gpg.example>", when=T2)
        record("two user ids, primary-uid flag", h, f)
        gpg(h, "--quick-revoke-uid", f, "u01 <u01@gpg.example>", when=T2)
        record("... the first user id revoked (a 0x30 by the primary key among its signatures)", h, f)
        # 5. expiry changed: another self-signature
        h = new_home(); homes.append(h)
        gpg(h, "--quick-gen-key", "e01 <e01@gpg.example>", "default", "default", "never")
        f = fpr_of(h, "e01@gpg.example")
        gpg(h, "--quick-set-expire", f, "2y", when=T1)
        gpg(h, "--quick-set-expire", f, "3y", "*", when=T2)
        record("expiry set on key and subkey (new self-signature, new binding)", h, f)
        # 6. revoked subkey
        h = new_home(); homes.append(h)
        gpg(h, "--quick-gen-key", "r01 <r01@gpg.example>", "rsa2048", "sign,cert", "never")
        f = fpr_of(h, "r01@gpg.example")
        gpg(h, "--quick-add-key", f, "rsa2048", "sign", "never", when=T1)
        sub_id = keys_of(h, f)[1]
        pl = b"signed before the revocation"
        early = gpg(h, "--digest-algo", "SHA256", "-u", sub_id + "!", "--detach-sign", "-o", "-", inp=pl, when=T1).stdout
        edit(h, f, "key 1\nrevkey\ny\n1\n\ny\nsave\n", when=T2)
        record("signing subkey revoked (0x28 with a reason)", h, f)
        # a signature the subkey made while it was good: gpg still calls it good (with a warning); x/crypto's KeysByIdUsage drops a
        # subkey whose Sig carries a revocation reason, so CheckDetachedSignature ends in ErrUnknownIssuer -- recorded, not compared
        with open(os.path.join(h, "pl"), "wb") as fh:
            fh.write(pl)
        with open(os.path.join(h, "sg"), "wb") as fh:
            fh.write(early)
        v = gpg(h, "--verify", os.path.join(h, "sg"), os.path.join(h, "pl"), ok=None, when=T2)
        out["certificates"][-1]["detached_by_revoked_subkey"] = {"key_id": sub_id, "payload": pl.hex(), "sig": early.hex(), "gpg_rc": v.returncode,
                                                                 "gpg_says": v.stderr.decode()[-300:]}
        # 7. revoked key (the revocation certificate gpg prepares at key generation)
        h = new_home(); homes.append(h)
        gpg(h, "--quick-gen-key", "k01 <k01@gpg.example>", "default", "default", "never")
        f = fpr_of(h, "k01@gpg.example")
        rev = open(os.path.join(h, "openpgp-revocs.d", f + ".rev")).read().replace(":-----BEGIN", "-----BEGIN")
        gpg(h, "--import", inp=rev.encode(), when=T1)
        record("revoked key (0x20 before the user id)", h, f)
        # 8. certified by two other keys
        h = new_home(); homes.append(h)
        for u in ("c01", "c02", "c03"):
            gpg(h, "--quick-gen-key", "%s <%s@gpg.example>" % (u, u), "rsa2048", "sign,cert", "never")
        f = fpr_of(h, "c01@gpg.example")
        gpg(h, "-u", "c02@gpg.example", "--quick-sign-key", f, when=T1)
        gpg(h, "-u", "c03@gpg.example", "--quick-sign-key", f, when=T2)
        record("certified by two other keys", h, f)
        # 9. DSA primary with an ElGamal subkey
        h = new_home(); homes.append(h)
        gpg(h, "--quick-gen-key", "d01 <d01@gpg.example>", "dsa2048", "sign,cert", "never")
        f = fpr_of(h, "d01@gpg.example")
        gpg(h, "--quick-add-key", f, "elg2048", "encr", "never", when=T1)
        record("DSA primary, ElGamal subkey", h, f)

        # ---- tampered: the signing-subkey certificate without / with a spoiled cross-signature
        from oracle import openpgp as pgp
        ws = pgp.walk_certificate(signing_blob)
        cross = [c for c in ws[0].checks if c.kind == "cross"][0]
        emb = cross.raw[2 if cross.raw[1] < 192 else 3 if cross.raw[1] < 224 else 6:]
        at = signing_blob.index(emb)
        spoiled = bytearray(signing_blob)
        spoiled[at + len(emb) - 20] ^= 1
        for name, blob in (("cross-signature spoiled (unhashed area: the binding itself still verifies)", bytes(spoiled)),):
            hh = new_home(); homes.append(hh)
            r = gpg(hh, "--import", inp=blob, ok=None)
            lst = gpg(hh, "--with-colons", "--list-keys", ok=None).stdout.decode()
            subs = [ln.split(":") for ln in lst.splitlines() if ln.startswith("sub:")]
            out["tampered"].append({"name": name, "blob": blob.hex(), "gpg_import_stderr": r.stderr.decode()[-400:],
                                    "gpg_signing_subkeys_left": [s[4] for s in subs if "s" in s[11]]})
    finally:
        for h in homes:
            shutil.rmtree(h, ignore_errors=True)
    path = os.path.join(ROOT, "tests", "golden", "gpg_cert_vectors.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote %s: %d certificates, %d tampered" % (path, len(out["certificates"]), len(out["tampered"])))


if __name__ == "__main__":
    main()
