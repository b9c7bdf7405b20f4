"""-m gpu: threshold share-combine kernels (config 5) vs the oracle and the reference's known answers."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import threshold as T

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "threshold_kat.json")))


def test_rsa_combine_kat_and_random(gpu_ctx):
    r = KAT["rsa"]
    n, d = int(r["n"], 16), int(r["d"], 16)
    rng = np.random.default_rng(5)
    digest = hashlib.sha256(r["tbs"].encode()).digest()
    m = T.emsa_encode("sha256", digest, n)
    rnd = [int.from_bytes(rng.bytes(513), "big") % (1 << (2 * d.bit_length())) for _ in range(9)]
    di = T.split_key(d, 10, rnd)
    psigs = [T.partial_sign(m, x, n) for x in di]
    # partial signatures of the positive fragments straight from the GPU modexp (rsa.go:161-171), 4096-bit exponents
    pos = [x for x in di if x >= 0]
    nb = 520
    base = np.frombuffer(b"".join(m.to_bytes(256, "big") for _ in pos), dtype=np.uint8).reshape(len(pos), 256).copy()
    exps = np.frombuffer(b"".join(x.to_bytes(nb, "big") for x in pos), dtype=np.uint8).reshape(len(pos), nb).copy()
    mods = np.frombuffer(b"".join(n.to_bytes(256, "big") for _ in pos), dtype=np.uint8).reshape(len(pos), 256).copy()
    out = gpu_ctx.modexp(base, np.arange(len(pos), dtype=np.uint32), mods, exps)
    assert [int.from_bytes(out[i].tobytes(), "big") for i in range(len(pos))] == [pow(m, x, n) for x in pos]
    # the combine reproduces the deterministic PKCS#1 v1.5 signature of the reference's TestCombine
    other = int(KAT["sss"]["pb"], 16)
    ops = [psigs, psigs[::-1], [int(rng.integers(1, 1 << 62)) for _ in range(10)], [0] + psigs[1:], [n - 1] * 10]
    mods_l, idx = [n, other], [0, 0, 1, 0, 0]
    got = gpu_ctx.modmul_product(ops, mods_l, idx)
    assert T.i2os(got[0], 256).hex() == r["sha256_pkcs1v15_sig"] and got[1] == got[0]
    for o, i, g in zip(ops, idx, got):
        assert g == T.calculate_signature(o, mods_l[i])


def test_lagrange_combine_sss_and_calculate_s(gpu_ctx):
    s = KAT["sss"]
    m, secret = int(s["pb"], 16), int.from_bytes(s["secret"].encode(), "big")
    g_ = KAT["dsa_group"]
    q = int(g_["q"], 16)
    q256 = (1 << 255) + 95                       # an odd 256-bit modulus (not assumed prime by the kernels)
    rng = np.random.default_rng(6)
    xs, ys, mi, want = [], [], [], []
    moduli = [m, q, q256, 1237]
    coeffs = [int.from_bytes(rng.bytes(256), "big") % m for _ in range(6)]
    shares = T.distribute(secret, 10, 7, m, coeffs)
    for trial in range(12):                      # SSS mod the 2048-bit prime: recovers "secret"
        pick = [shares[i] for i in rng.choice(10, size=7, replace=False)]
        xs.append([p[0] for p in pick]); ys.append([p[1] for p in pick]); mi.append(0); want.append(secret)
    for trial in range(40):                      # calculateS-style sums with 7 shares over small and 160/256-bit moduli
        mod_i = int(rng.integers(1, 4))
        mod = moduli[mod_i]
        x = [int(v) for v in rng.choice(np.arange(1, 16), size=7, replace=False)]
        y = [int.from_bytes(rng.bytes(40), "big") % mod for _ in range(7)]
        try:
            w = T.calculate_s(list(zip(x, y)), mod)
        except ValueError:                       # no inverse (1237 is prime, q256 may share a factor): status 1
            w = None
        xs.append(x); ys.append(y); mi.append(mod_i); want.append(w)
    got, st = gpu_ctx.lagrange_combine(xs, ys, moduli, mi)
    for g, w, s_ in zip(got, want, st):
        if w is None:
            assert s_ == 1
        else:
            assert s_ == 0 and g == w
    assert got[0] == secret


def test_dsa_calculate_r(gpu_ctx):
    g_ = KAT["dsa_group"]
    p, q, g = int(g_["p"], 16), int(g_["q"], 16), int(g_["g"], 16)
    rng = np.random.default_rng(8)
    n, t = 10, 4
    rnd = lambda: int.from_bytes(rng.bytes(40), "big") % q
    xs, ri, vi, want = [], [], [], []
    for trial in range(20):
        kk, aa = rnd(), rnd()
        ks = T.distribute(kk, n, t, q, [rnd() for _ in range(t - 1)])
        as_ = T.distribute(aa, n, t, q, [rnd() for _ in range(t - 1)])
        zs = T.distribute(0, n, 2 * t, q, [rnd() for _ in range(2 * t - 1)])
        pick = [int(i) for i in rng.choice(n, size=2 * t, replace=False)]
        rs = [(as_[i][0], T.calculate_partial_r(g, as_[i][1], p), (ks[i][1] * as_[i][1] + zs[i][1]) % q) for i in pick]
        xs.append([r[0] for r in rs]); ri.append([int.from_bytes(r[1], "big") for r in rs]); vi.append([r[2] for r in rs])
        want.append(T.calculate_r(rs, p, q))
        assert want[-1] == pow(g, pow(kk, -1, q), p) % q
    got, st = gpu_ctx.dsa_calculate_r(xs, ri, vi, [(p, q)], [0] * len(xs))
    assert list(st) == [0] * len(xs) and got == want


def test_sss_distribute_and_round_trip(gpu_ctx):
    """sss.Distribute on the GPU == the restatement; Distribute -> (drop shares) -> calculateSecret recovers the secret."""
    s = KAT["sss"]
    m = int(s["pb"], 16)
    q = int(KAT["dsa_group"]["q"], 16)
    rng = np.random.default_rng(10)
    polys, mi, moduli = [], [], [m, q, 1237]
    for trial in range(9):
        mod_i = trial % 3
        mod = moduli[mod_i]
        polys.append([int.from_bytes(rng.bytes(260), "big") % mod for _ in range(7)])
        mi.append(mod_i)
    polys[0][0] = int.from_bytes(s["secret"].encode(), "big")
    polys.append(KAT["auth_sss_example"]["poly"] + [0, 0, 0]); mi.append(2)
    shares = gpu_ctx.sss_distribute(polys, 10, moduli, mi)
    for p, sh, i in zip(polys, shares, mi):
        assert sh == [y for _, y in T.distribute(p[0], 10, 7, moduli[i], p[1:])]
    xs = [[int(x) + 1 for x in rng.choice(10, size=7, replace=False)] for _ in polys]
    ys = [[sh[x - 1] for x in xr] for sh, xr in zip(shares, xs)]
    got, st = gpu_ctx.lagrange_combine(xs, ys, moduli, mi)
    assert list(st) == [0] * len(polys) and got == [p[0] for p in polys]
    assert got[0].to_bytes(6, "big") == b"secret"


def test_partial_sign_with_negative_fragments(gpu_ctx):
    """rsaContext.Sign per fragment (rsa.go:161-171): m^|d_i| mod N by the modexp kernel, inverted on the GPU when the
    fragment is negative; the product of all fragments' partial signatures is the PKCS#1 signature (TestCombine)."""
    r = KAT["rsa"]
    n, d = int(r["n"], 16), int(r["d"], 16)
    rng = np.random.default_rng(11)
    m = T.emsa_encode("sha256", hashlib.sha256(r["tbs"].encode()).digest(), n)
    di = T.split_key(d, 10, [int.from_bytes(rng.bytes(513), "big") % (1 << (2 * d.bit_length())) for _ in range(9)])
    nb = 520
    base = np.frombuffer(b"".join(m.to_bytes(256, "big") for _ in di), dtype=np.uint8).reshape(len(di), 256).copy()
    exps = np.frombuffer(b"".join(abs(x).to_bytes(nb, "big") for x in di), dtype=np.uint8).reshape(len(di), nb).copy()
    mods = np.frombuffer(b"".join(n.to_bytes(256, "big") for _ in di), dtype=np.uint8).reshape(len(di), 256).copy()
    out = gpu_ctx.modexp(base, np.arange(len(di), dtype=np.uint32), mods, exps)
    ci = [int.from_bytes(out[i].tobytes(), "big") for i in range(len(di))]
    neg = [i for i, x in enumerate(di) if x < 0]
    assert neg
    inv, st = gpu_ctx.modinv([ci[i] for i in neg], [n], [0] * len(neg))
    assert list(st) == [0] * len(neg)
    for i, v in zip(neg, inv):
        ci[i] = v
    assert ci == [T.partial_sign(m, x, n) for x in di]
    assert T.i2os(gpu_ctx.modmul_product([ci], [n], [0])[0], 256).hex() == r["sha256_pkcs1v15_sig"]
    # no inverse: a multiple of a factor of N; zero
    p = int(r["p"], 16)
    vals, st = gpu_ctx.modinv([p, 0, 3 * p, 5, n + 7], [n], [0] * 5)
    assert list(st) == [1, 1, 1, 0, 0] and vals[3] == pow(5, -1, n) and vals[4] == pow(7, -1, n)
