#!/usr/bin/env python3
"""Generates tests/golden/threshold_kat.json from the reference's own test fixtures and constants:

  crypto/threshold/rsa/test.pkcs8     RSA-2048 key of TestCombine / TestEMSA (rsa_test.go:131-206)
  crypto/threshold/dsa/dsa_test.go:26-28   fixed DSA group (P 1024 bit, Q 160 bit, G)
  crypto/sss/sss_test.go:15-47        2048-bit prime pb, secret "secret"
  crypto/auth/auth_test.go:121-155    Shamir example over q=1237 (poly 1234,166,94,666; shares 2,4,5,6 => 1234)

plus the deterministic expectation the reference's TestCombine asserts: PKCS#1 v1.5 / SHA-256 signature of
"tbs..." (rsa_test.go testTBS) under that key, computed here by OpenSSL.  Run in the build container only.
"""
import json
import os
import re
import subprocess

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "threshold_kat.json")


def rsa_numbers(path):
    txt = subprocess.run(["openssl", "pkey", "-in", path, "-inform", "DER", "-text", "-noout"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE)
    if txt.returncode != 0:
        txt = subprocess.run(["openssl", "pkey", "-in", path, "-text", "-noout"], stdout=subprocess.PIPE, check=True)
    t = txt.stdout.decode()

    def grab(name):
        m = re.search(name + r":\s*\n((?:\s+[0-9a-f:]+\n)+)", t)
        return int(re.sub(r"[\s:]", "", m.group(1)), 16)
    e = int(re.search(r"publicExponent:\s*(\d+)", t).group(1))
    return {"n": "%x" % grab("modulus"), "e": e, "d": "%x" % grab("privateExponent"),
            "p": "%x" % grab("prime1"), "q": "%x" % grab("prime2")}


def main():
    kat = {}
    key = os.path.join(REF, "crypto/threshold/rsa/test.pkcs8")
    kat["rsa"] = rsa_numbers(key)
    src = open(os.path.join(REF, "crypto/threshold/rsa/rsa_test.go")).read()
    tbs = re.search(r'testTBS\s*=\s*"([^"]*)"', src).group(1)
    kat["rsa"]["tbs"] = tbs
    der = subprocess.run(["openssl", "pkey", "-in", key, "-inform", "DER"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    pem = der.stdout if der.returncode == 0 else open(key, "rb").read()
    with open("/tmp/_kat_key.pem", "wb") as f:
        f.write(pem)
    sig = subprocess.run(["openssl", "dgst", "-sha256", "-sign", "/tmp/_kat_key.pem"], input=tbs.encode(), stdout=subprocess.PIPE,
                         check=True).stdout
    os.unlink("/tmp/_kat_key.pem")
    kat["rsa"]["sha256_pkcs1v15_sig"] = sig.hex()
    dsrc = open(os.path.join(REF, "crypto/threshold/dsa/dsa_test.go")).read()
    kat["dsa_group"] = {k.lower()[0]: re.search(k + r'\s*=\s*"([0-9A-Fa-f]+)"', dsrc).group(1).lower() for k in ("Pstr", "Qstr", "Gstr")}
    ssrc = open(os.path.join(REF, "crypto/sss/sss_test.go")).read()
    blob = ssrc[ssrc.index("pb = []byte{"):ssrc.index("secret = []byte")]
    kat["sss"] = {"pb": "".join(re.findall(r"0x([0-9a-fA-F]{2})", blob)).lower(), "secret": "secret", "n": 10, "k": 7}
    kat["auth_sss_example"] = {"q": 1237, "poly": [1234, 166, 94, 666], "sample_x": [2, 4, 5, 6], "secret": 1234}
    with open(OUT, "w") as f:
        json.dump(kat, f, indent=1)
    print("wrote", OUT, "sig begins", sig[:8].hex())


if __name__ == "__main__":
    main()
