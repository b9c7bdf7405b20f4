"""CPU: threshold share-combine restatement (oracle/threshold.py) against the reference's own known answers
(tests/golden/threshold_kat.json, made by tests/golden/make_threshold_kat.py from the reference's fixtures)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import threshold as T

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "threshold_kat.json")))


def test_rsa_combine_reproduces_the_pkcs1_signature():
    """TestCombine (rsa_test.go:165-206): prod m^{d_i} mod N == rsa.SignPKCS1v15(SHA-256, testTBS) -- deterministic."""
    r = KAT["rsa"]
    n, d = int(r["n"], 16), int(r["d"], 16)
    rng = np.random.default_rng(1)
    digest = hashlib.sha256(r["tbs"].encode()).digest()
    m = T.emsa_encode("sha256", digest, n)
    assert m == int.from_bytes(b"\x00\x01" + b"\xff" * (256 - 51 - 3) + b"\x00" + T.HASH_PREFIXES["sha256"] + digest, "big")   # TestEMSA
    for trial in range(3):
        rnd = [int.from_bytes(rng.bytes(2 * 256 + 1), "big") % (1 << (2 * d.bit_length())) for _ in range(9)]
        di = T.split_key(d, 10, rnd)
        assert sum(di) == d and any(x < 0 for x in di)
        psigs = [T.partial_sign(m, x, n) for x in di]
        sig = T.i2os(T.calculate_signature(psigs, n), 256)
        assert sig.hex() == r["sha256_pkcs1v15_sig"]
    assert hashlib.sha256(bytes.fromhex(r["sha256_pkcs1v15_sig"])).hexdigest() == "1d2cef7b44c674e771fdac4fb0f278c75e7b68fe40a835cd39ba5c23cd127998"


def test_sss_recovers_the_reference_secret():
    """TestSSS (sss_test.go:49-75): any k of n shares of "secret" mod the fixed 2048-bit prime recover it."""
    s = KAT["sss"]
    m, secret = int(s["pb"], 16), int.from_bytes(s["secret"].encode(), "big")
    rng = np.random.default_rng(2)
    coeffs = [int.from_bytes(rng.bytes(256), "big") % m for _ in range(s["k"] - 1)]
    shares = T.distribute(secret, s["n"], s["k"], m, coeffs)
    for trial in range(5):
        pick = [shares[i] for i in rng.choice(s["n"], size=s["k"], replace=False)]
        assert T.calculate_secret(pick, m) == secret
    assert T.calculate_secret(shares[:s["k"] - 1], m) != secret


def test_auth_sss_example():
    """crypto/auth/auth_test.go:121-155: poly (1234,166,94,666) over 1237, shares x=2,4,5,6 => 1234."""
    e = KAT["auth_sss_example"]
    shares = T.distribute(e["poly"][0], 6, 4, e["q"], e["poly"][1:])
    pick = [sh for sh in shares if sh[0] in e["sample_x"]]
    assert T.calculate_secret(pick, e["q"]) == e["secret"]


def test_threshold_dsa_relations_on_the_reference_group():
    """dsa_test.go:47-216 identities on the fixed group: sums and products of shared secrets, and CalculateR ==
    g^(k^-1) mod p mod q for jointly shared k, a (the R of the 3-phase protocol, dsa_core.go:126-141)."""
    g_ = KAT["dsa_group"]
    p, q, g = int(g_["p"], 16), int(g_["q"], 16), int(g_["g"], 16)
    rng = np.random.default_rng(3)
    n, t = 10, 4
    rnd = lambda: int.from_bytes(rng.bytes(40), "big") % q
    kk, aa = rnd(), rnd()
    ks = T.distribute(kk, n, t, q, [rnd() for _ in range(t - 1)])
    as_ = T.distribute(aa, n, t, q, [rnd() for _ in range(t - 1)])
    zs = T.distribute(0, n, 2 * t, q, [rnd() for _ in range(2 * t - 1)])       # degree-(2t-1) zero sharing masks the product
    pick = [int(i) for i in rng.choice(n, size=2 * t, replace=False)]
    # TestSum / TestMul: f(0)+g(0) and f(0)*g(0) from 2t shares
    assert T.calculate_s([(ks[i][0], (ks[i][1] + as_[i][1]) % q) for i in pick], q) == (kk + aa) % q
    assert T.calculate_s([(ks[i][0], (ks[i][1] * as_[i][1] + zs[i][1]) % q) for i in pick], q) == (kk * aa) % q
    # CalculateR: ri = g^ai, vi = ki*ai + zi  =>  r = (g^a)^((k a)^-1) = g^(k^-1)
    rs = [(as_[i][0], T.calculate_partial_r(g, as_[i][1], p), (ks[i][1] * as_[i][1] + zs[i][1]) % q) for i in pick]
    assert T.calculate_r(rs, p, q) == pow(g, pow(kk, -1, q), p) % q
    assert T.format_dsa(5, 7, q) == (5).to_bytes(20, "big") + (7).to_bytes(20, "big")
    assert T.os2i(bytes(range(1, 33)), q) == int.from_bytes(bytes(range(1, 21)), "big")


def test_c_port_reproduces_the_reference_answers_and_the_python_oracle():
    """oracle/c/threshold.c (bench.py's cfg-5 cpu_baseline) on the reference's own known answers -- TestCombine's
    signature, TestSSS's secret, CalculateR == g^(k^-1) on the fixed group -- and on random inputs against oracle/threshold.py."""
    from oracle.cbind import CThreshold
    ct = CThreshold()
    be = CThreshold._be
    toi = lambda row: int.from_bytes(row.tobytes(), "big")
    rng = np.random.default_rng(5)
    # RSA: TestCombine
    r = KAT["rsa"]
    n, d = int(r["n"], 16), int(r["d"], 16)
    m = T.emsa_encode("sha256", hashlib.sha256(r["tbs"].encode()).digest(), n)
    rnd = [int.from_bytes(rng.bytes(2 * 256 + 1), "big") % (1 << (2 * d.bit_length())) for _ in range(9)]
    psigs = [T.partial_sign(m, x, n) for x in T.split_key(d, 10, rnd)]
    out = ct.rsa_combine(be(psigs, 256), 10, 256, n, n_threads=1)
    assert out[0].tobytes().hex() == r["sha256_pkcs1v15_sig"]
    # SSS: TestSSS
    s = KAT["sss"]
    pb, secret = int(s["pb"], 16), int.from_bytes(s["secret"].encode(), "big")
    shares = T.distribute(secret, s["n"], s["k"], pb, [int.from_bytes(rng.bytes(256), "big") % pb for _ in range(s["k"] - 1)])
    picks = [[shares[i] for i in rng.choice(s["n"], size=s["k"], replace=False)] for _ in range(6)]
    xs = np.array([[x for x, _ in p] for p in picks], dtype=np.int32)
    out, st = ct.lagrange_combine(xs, be([y for p in picks for _, y in p], 256), 256, pb, n_threads=3)
    assert not st.any() and all(toi(out[i]) == secret for i in range(len(picks)))
    # threshold DSA on the reference's group: calculateS and CalculateR
    g_ = KAT["dsa_group"]
    p, q, g = int(g_["p"], 16), int(g_["q"], 16), int(g_["g"], 16)
    nn, t = 10, 4
    rq = lambda: int.from_bytes(rng.bytes(40), "big") % q
    kk, aa = rq(), rq()
    ks = T.distribute(kk, nn, t, q, [rq() for _ in range(t - 1)])
    as_ = T.distribute(aa, nn, t, q, [rq() for _ in range(t - 1)])
    zs = T.distribute(0, nn, 2 * t, q, [rq() for _ in range(2 * t - 1)])
    pick = [int(i) for i in rng.choice(nn, size=2 * t, replace=False)]
    qb = (q.bit_length() + 7) // 8
    xs = np.array([[ks[i][0] for i in pick]], dtype=np.int32)
    out, st = ct.lagrange_combine(xs, be([(ks[i][1] * as_[i][1] + zs[i][1]) % q for i in pick], qb), qb, q)
    assert not st.any() and toi(out[0]) == (kk * aa) % q
    ri = [pow(g, as_[i][1], p) for i in pick]
    vi = [(ks[i][1] * as_[i][1] + zs[i][1]) % q for i in pick]
    out, st = ct.dsa_calculate_r(xs, be(ri, 256), 256, be(vi, qb), qb, p, q)
    assert not st.any() and toi(out[0]) == pow(g, pow(kk, -1, q), p) % q
    # random residues (the bench's corpus shape) against the Python oracle, several threads
    from corpus import build as cb
    k0 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "keys_dsa2048.json")))["keys"][0]
    as_int = lambda v: int(v, 16) if isinstance(v, str) else int(v)
    tc = cb.make_threshold_corpus(24, n, pb, as_int(k0["p"]), as_int(k0["q"]), seed=99)
    flat = lambda rows: [v for r_ in rows for v in r_]
    got = ct.rsa_combine(be(flat(tc.rsa_factors), 256), 10, 256, tc.rsa_n, n_threads=4)
    assert all(toi(got[i]) == T.calculate_signature(tc.rsa_factors[i], tc.rsa_n) for i in range(24))
    got, st = ct.lagrange_combine(tc.s_xs, be(flat(tc.s_ys), 32), 32, tc.dsa_q, n_threads=4)
    assert all(toi(got[i]) == T.calculate_s(list(zip([int(v) for v in tc.s_xs[i]], tc.s_ys[i])), tc.dsa_q) for i in range(24))
    got, st = ct.dsa_calculate_r(tc.r_xs, be(flat(tc.r_ri), 256), 256, be(flat(tc.r_vi), 32), 32, tc.dsa_p, tc.dsa_q, n_threads=4)
    for i in range(24):
        try:
            want = T.calculate_r([(int(tc.r_xs[i][j]), tc.r_ri[i][j].to_bytes(256, "big"), tc.r_vi[i][j]) for j in range(8)], tc.dsa_p, tc.dsa_q)
            assert st[i] == 0 and toi(got[i]) == want
        except ValueError:
            assert st[i] == 1
