"""Deterministic RSA-2048 / DSA-2048-256 key material from a master seed.

The reference's fixtures come from gpg2 (`scripts/gen.sh:26`, `scripts/setup.sh`) and are not
checked in; here every key derives from one seed so corpora are reproducible.  Generated keys are
cached as JSON under tests/golden/ (generation of 2048-bit primes in pure Python takes ~0.3 s per
RSA key and a few seconds per DSA group).
"""
from __future__ import annotations

import hashlib
import json
import os
from typing import Dict, List

MASTER_SEED = 0xBF7C0DE

_SMALL_PRIMES: List[int] = []


def _small_primes(limit=20000):
    global _SMALL_PRIMES
    if not _SMALL_PRIMES:
        sieve = bytearray([1]) * (limit + 1)
        sieve[0:2] = b"\0\0"
        for i in range(2, int(limit ** 0.5) + 1):
            if sieve[i]:
                sieve[i * i::i] = bytearray(len(sieve[i * i::i]))
        _SMALL_PRIMES = [i for i in range(limit + 1) if sieve[i]]
    return _SMALL_PRIMES


class DRBG:
    """SHA-256 counter-mode byte stream (deterministic, platform independent)."""

    def __init__(self, *label):
        self.key = hashlib.sha256(repr(label).encode()).digest()
        self.ctr = 0
        self.buf = b""

    def bytes(self, n: int) -> bytes:
        while len(self.buf) < n:
            self.buf += hashlib.sha256(self.key + self.ctr.to_bytes(8, "big")).digest()
            self.ctr += 1
        out, self.buf = self.buf[:n], self.buf[n:]
        return out

    def bits(self, nbits: int) -> int:
        nb = (nbits + 7) // 8
        return int.from_bytes(self.bytes(nb), "big") >> (nb * 8 - nbits)

    def below(self, n: int) -> int:
        nb = n.bit_length()
        while True:
            x = self.bits(nb)
            if x < n:
                return x


def is_probable_prime(n: int, rng: DRBG, rounds: int = 12) -> bool:
    if n < 2:
        return False
    for p in _small_primes():
        if n == p:
            return True
        if n % p == 0:
            return False
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for i in range(rounds):
        a = 2 if i == 0 else 2 + rng.below(n - 3)
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def gen_prime(bits: int, rng: DRBG) -> int:
    while True:
        c = rng.bits(bits) | (3 << (bits - 2)) | 1
        # incremental search with a small-prime sieve
        rems = [c % p for p in _small_primes()[1:400]]
        for delta in range(0, 4000, 2):
            if all((r + delta) % p for r, p in zip(rems, _small_primes()[1:400])):
                n = c + delta
                if n.bit_length() == bits and is_probable_prime(n, rng):
                    return n


def gen_rsa(index: int, bits: int = 2048, seed: int = MASTER_SEED) -> Dict[str, int]:
    rng = DRBG("rsa", seed, bits, index)
    e = 65537
    while True:
        p = gen_prime(bits // 2, rng)
        q = gen_prime(bits // 2, rng)
        if p == q:
            continue
        n = p * q
        phi = (p - 1) * (q - 1)
        if n.bit_length() != bits or phi % e == 0:
            continue
        return {"p": p, "q": q, "e": e}


def gen_dsa(index: int, L: int = 2048, N: int = 256, seed: int = MASTER_SEED) -> Dict[str, int]:
    """One (p, q, g) group per key, as gpg's dsa2048 does, plus private x."""
    rng = DRBG("dsa", seed, L, N, index)
    q = gen_prime(N, rng)
    while True:
        x = rng.bits(L) | (1 << (L - 1))
        c = x - (x % (2 * q)) + 1
        if c.bit_length() != L:
            continue
        if is_probable_prime(c, rng, rounds=6):
            p = c
            break
    while True:
        h = 2 + rng.below(p - 3)
        g = pow(h, (p - 1) // q, p)
        if g > 1:
            break
    priv = 1 + rng.below(q - 1)
    return {"p": p, "q": q, "g": g, "x": priv}


DSA_Q_BITS = {1024: 160, 1536: 224, 2048: 256, 3072: 256}      # FIPS 186-3 pairs as GnuPG picks them (g10/keygen.c: > 2047 bits 256, > 1024 bits 224, else 160)


_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _cache_path(kind: str) -> str:
    return os.path.join(_GOLDEN, "keys_%s.json" % kind)


def load_keys(kind: str, count: int) -> List[Dict[str, int]]:
    """kind: 'rsa2048', 'rsa3072', 'rsa4096', 'dsa2048' (q 256 bits), or the other DSA sizes in use -- 'dsa1024' (q 160 bits: the
    group size of the reference's era and of its threshold tests, crypto/threshold/dsa/dsa_test.go:26-28), 'dsa1536' (q 224 bits,
    what gpg 2.2 makes of a 1536-bit request) and 'dsa3072' (q 256 bits).  Returns the first ``count`` keys, generating and
    extending the on-disk cache when it is short."""
    path = _cache_path(kind)
    keys: List[Dict[str, str]] = []
    if os.path.exists(path):
        with open(path) as f:
            keys = json.load(f)["keys"]
    if len(keys) < count:
        if kind.startswith("rsa"):
            gen = lambda i: gen_rsa(i, bits=int(kind[3:]))                     # noqa: E731
        else:
            L = int(kind[3:])
            gen = lambda i: gen_dsa(i, L=L, N=DSA_Q_BITS[L])                   # noqa: E731
        for i in range(len(keys), count):
            k = gen(i)
            keys.append({name: "%x" % v for name, v in k.items()})
        os.makedirs(_GOLDEN, exist_ok=True)
        with open(path, "w") as f:
            json.dump({"kind": kind, "seed": "%x" % MASTER_SEED, "generator": "corpus/keys.py", "keys": keys}, f)
    return [{name: int(v, 16) for name, v in k.items()} for k in keys[:count]]


if __name__ == "__main__":
    import sys
    import time
    kind, count = sys.argv[1], int(sys.argv[2])
    t = time.time()
    load_keys(kind, count)
    print("have %d %s keys (%.1f s)" % (count, kind, time.time() - t))
