"""Oracle restatement of the threshold-signature share-combine arithmetic (BASELINE config 5).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference:
  sss.Distribute                      crypto/sss/sss.go:23-47
  sss.Lagrange                        crypto/sss/sss.go:94-107
  SSSProcess.calculateSecret          crypto/sss/sss.go:69-92
  calculateS                          crypto/threshold/dsa/dsa_core.go:389-403
  formatDSA                           crypto/threshold/dsa/dsa_core.go:375-387
  dsaGroupOperations.CalculatePartialR / CalculateR / OS2I   crypto/threshold/dsa/dsa.go:27-52, 72-75
  splitKey                            crypto/threshold/rsa/rsa.go:98-117
  rsaContext.Sign (per fragment)      crypto/threshold/rsa/rsa.go:161-171
  calculateSignature / I2OS           crypto/threshold/rsa/rsa.go:318-329, 380-393
  emsaEncode / hashPrefixes           crypto/threshold/rsa/rsa.go:345-378
Pinned by tests/golden/threshold_kat.json (the reference's own test key, constants and the
TestCombine / TestSSS relations, rsa_test.go:165-206, sss_test.go:49-75, auth_test.go:121-155).
Go's math/big semantics that matter: Mod is Euclidean (result >= 0); ModInverse accepts negative g.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

HASH_PREFIXES = {   # rsa.go:345-354
    "sha1": bytes.fromhex("3021300906052b0e03021a05000414"),
    "sha224": bytes.fromhex("302d300d06096086480165030402040500041c"),
    "sha256": bytes.fromhex("3031300d060960864801650304020105000420"),
    "sha384": bytes.fromhex("3041300d060960864801650304020205000430"),
    "sha512": bytes.fromhex("3051300d060960864801650304020305000440"),
}


# ---- crypto/sss ------------------------------------------------------------------------------------
def distribute(secret: int, n: int, k: int, m: int, coeffs: Sequence[int]) -> List[Tuple[int, int]]:
    """sss.Distribute with the random coefficients supplied (rand.Int(rand.Reader, m) in the reference)."""
    poly = [secret] + list(coeffs)[:k - 1]
    res = []
    for i in range(n):
        x0 = i + 1
        x = x0
        f = poly[0]
        for j in range(1, k):
            f = (f + poly[j] * x) % m
            x *= x0
        res.append((i + 1, f))
    return res


def lagrange(x: int, xs: Sequence[int], m: int) -> int:
    a, b = 1, 1
    for r in xs:
        if r == x:
            continue
        a *= r
        b *= r - x
    inv = pow(b % m, -1, m)     # big.Int.ModInverse reduces a negative argument first
    return (a * inv) % m


def calculate_secret(shares: Sequence[Tuple[int, int]], m: int) -> int:
    xs = [x for x, _ in shares]
    s = 0
    for x, y in shares:
        s = (s + lagrange(x, xs, m) * y) % m
    return s


# ---- crypto/threshold/dsa ----------------------------------------------------------------------------
def calculate_s(shares: Sequence[Tuple[int, int]], q: int) -> int:
    xs = [x for x, _ in shares]
    s = 0
    for x, y in shares:
        s = (s + (y * lagrange(x, xs, q)) % q) % q
    return s


def calculate_partial_r(g: int, ai: int, p: int) -> bytes:
    r = pow(g, ai, p)
    return r.to_bytes((r.bit_length() + 7) // 8, "big")


def calculate_r(rs: Sequence[Tuple[int, bytes, int]], p: int, q: int) -> int:
    """rs: (X, Ri bytes, Vi).  r = (prod Ri^li mod p)^((sum Vi*li)^-1 mod q) mod p mod q."""
    xs = [x for x, _, _ in rs]
    r, v = 1, 0
    for x, ri, vi in rs:
        l = lagrange(x, xs, q)
        r = (r * pow(int.from_bytes(ri, "big"), l, p)) % p
        v = (v + (vi * l) % q) % q
    vinv = pow(v, -1, q)
    return pow(r, vinv, p) % q


def os2i(os: bytes, q: int) -> int:
    return int.from_bytes(os[:(q.bit_length() + 7) // 8], "big")


def format_dsa(r: int, s: int, q: int) -> bytes:
    n = (q.bit_length() + 7) // 8
    return r.to_bytes(n, "big") + s.to_bytes(n, "big")


# ---- crypto/threshold/rsa ----------------------------------------------------------------------------
def split_key(d: int, n: int, randoms: Sequence[int]) -> List[int]:
    """splitKey with the n-1 random draws x < 2^(2*bits(d)) supplied: sign = bit 0, magnitude = x >> 1."""
    di, total = [], 0
    for i in range(n - 1):
        x = randoms[i]
        sign = x & 1
        x >>= 1
        if sign:
            x = -x
        di.append(x)
        total += x
    di.append(d - total)
    return di


def emsa_encode(hash_name: str, digest: bytes, n: int) -> int:
    emlen = (n.bit_length() + 7) // 8
    prefix = HASH_PREFIXES[hash_name]
    mlen = len(prefix) + len(digest)
    padlen = emlen - mlen
    if padlen < 3:
        raise ValueError("crypto: invalid input")
    em = b"\x00\x01" + b"\xff" * (padlen - 3) + b"\x00" + prefix + digest
    return int.from_bytes(em, "big")


def partial_sign(m: int, di: int, n: int) -> int:
    """rsa.go:161-171: m^|di| mod N, inverted when the fragment is negative."""
    if di < 0:
        return pow(pow(m, -di, n), -1, n)
    return pow(m, di, n)


def calculate_signature(psigs: Sequence[int], n: int) -> int:
    s = 1
    for p in psigs:
        s = (s * p) % n
    return s


def i2os(b: int, sz: int) -> bytes:
    c = b.to_bytes((b.bit_length() + 7) // 8, "big")
    return c if len(c) >= sz else bytes(sz - len(c)) + c
