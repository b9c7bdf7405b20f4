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


@pytest.mark.parametrize("lanes,parts", ((4, 1), (8, 1), (8, 0), (4, 2), (8, 4), (4, 8)))
def test_dsa_calculate_r(lanes, parts):
    """CalculateR through both forms of k_multiexp: 4 lanes x 19 limbs (R = 2^2128) and 8 lanes x 10 limbs (R = 2^2240, picked by
    default for calls of at most one wave per SIMD); with the bases of an operation in one chain (parts = 1), spread over 2 / 4 / 8
    quad groups whose partial products k_modmul_product multiplies up, and by the default policy (parts = 0: a small call spreads
    its 8 bases over 8 groups)."""
    from bftkv_amd import Context
    os.environ["BFTKV_MULTIEXP_LANES"] = str(lanes)        # read when a context is created
    if parts:
        os.environ["BFTKV_MULTIEXP_PARTS"] = str(parts)
    try:
        gpu_ctx = Context(0)
    finally:
        del os.environ["BFTKV_MULTIEXP_LANES"]
        os.environ.pop("BFTKV_MULTIEXP_PARTS", None)
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
    gpu_ctx.close()


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


def test_modular_inverse_on_unreduced_arguments_and_moduli_of_every_width(gpu_ctx):
    """bftkv_gpu_modinv against Python's inverse: arguments far above a small modulus (big.Int.ModInverse reduces them; a
    subtract-until-smaller loop would never return), every modulus width from 2 bits to 2048, shared factors, modulus 1."""
    import math
    import random
    rng = random.Random(12)
    mods, vals, idx = [], [], []
    for bits in (1, 2, 9, 30, 31, 60, 61, 160, 255, 256, 257, 1024, 2047, 2048):
        for _ in range(6):
            m = rng.getrandbits(bits) | 1 | (1 << (bits - 1))
            mods.append(m)
            for x in (rng.randrange(m), rng.getrandbits(2048), (1 << 2048) - 1, 0, 1, m, m - 1, m + 1, 3 * rng.getrandbits(700)):
                vals.append(x); idx.append(len(mods) - 1)
    f = rng.getrandbits(300) | 1
    mods.append(f * (rng.getrandbits(1700) | 1))
    for x in (f, f * rng.getrandbits(1000), rng.getrandbits(2048)):
        vals.append(x); idx.append(len(mods) - 1)
    got, st = gpu_ctx.modinv(vals, mods, idx)
    for x, i, g, s in zip(vals, idx, got, st):
        m = mods[i]
        if math.gcd(x, m) != 1:
            assert (int(s), g) == (1, 0), (x, m)
        else:
            assert (int(s), g) == (0, pow(x, -1, m)), (x, m)
    assert 0 in st and 1 in st


def bench_needs_big(x):
    """k_lagrange_inv's rule for one operation: does any term's numerator or denominator leave 31 bits?"""
    x = [int(v) for v in x]
    for j in range(len(x)):
        a = b = 1
        for i in range(len(x)):
            if x[i] == x[j]:
                continue
            a *= x[i]; b *= x[i] - x[j]
            if abs(a) >= 1 << 31 or abs(b) >= 1 << 31:
                return True
    return False


def test_device_resident_entry_points_match_the_host_ones(gpu_ctx):
    """The *_dev forms (inputs already in HBM, results left there, asynchronous until bftkv_gpu_sync) compute what the
    host-pointer forms compute: they share one body, the difference is only who moves the bytes."""
    import torch
    from bftkv_amd._native import _ints_to_be, _ptr
    from corpus import build as cb
    g_ = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "keys_dsa2048.json")))["keys"][0]
    as_int = lambda v: int(v, 16) if isinstance(v, str) else int(v)
    N = 300
    tc = cb.make_threshold_corpus(N, int(KAT["rsa"]["n"], 16), int(KAT["sss"]["pb"], 16), as_int(g_["p"]), as_int(g_["q"]), seed=77)
    rng_x = np.random.default_rng(771)
    lib, h = gpu_ctx.lib, gpu_ctx.h
    flat = lambda rows: [v for r in rows for v in r]
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    toi = lambda row: int.from_bytes(row.tobytes(), "big")
    P = lambda t: t.data_ptr()
    # RSA combine
    o = torch.zeros((N, 256), dtype=torch.uint8, device="cuda:0")
    f = up(_ints_to_be(flat(tc.rsa_factors), 256))
    m = _ints_to_be([tc.rsa_n], 256)
    gpu_ctx._check(lib.bftkv_gpu_modmul_product_dev(h, N, 10, P(f), 256, None, 1, _ptr(m), P(o)), "modmul_product_dev")
    gpu_ctx.sync()
    assert [toi(r) for r in o.cpu().numpy()] == [T.calculate_signature(tc.rsa_factors[i], tc.rsa_n) for i in range(N)]
    assert [toi(r) for r in o.cpu().numpy()] == gpu_ctx.modmul_product(tc.rsa_factors, [tc.rsa_n], [0] * N)
    # Lagrange combine (SSS and calculateS), with an explicit device mod_idx that needs clamping on one row
    for xs, ys, mod, nb, k in ((tc.sss_xs, tc.sss_ys, tc.sss_mod, 256, 7), (tc.s_xs, tc.s_ys, tc.dsa_q, 32, 8)):
        o = torch.zeros((N, nb), dtype=torch.uint8, device="cuda:0")
        st = torch.zeros(N + 8, dtype=torch.uint8, device="cuda:0")
        mi = torch.zeros(N, dtype=torch.int32, device="cuda:0")
        mi[5] = 9                                                     # out of range: clamped to the last modulus (= 0)
        mb = _ints_to_be([mod], nb)
        d_x, d_y = up(xs), up(_ints_to_be(flat(ys), nb))            # keep the tensors alive: the call is asynchronous
        gpu_ctx._check(lib.bftkv_gpu_lagrange_combine_dev(h, N, k, P(d_x), P(d_y), nb, P(mi), 1, _ptr(mb), P(o), P(st)), "lagrange_dev")
        gpu_ctx.sync()
        want = [T.calculate_s(list(zip([int(v) for v in xs[i]], ys[i])), mod) for i in range(N)]
        assert [toi(r) for r in o.cpu().numpy()] == want and not st.cpu().numpy()[:N].any()
    # The device forms cannot see their x's: without a promise they enqueue the big-integer Lagrange kernels behind the fast path
    # (above: no-ops); with bftkv_gpu_set_lagrange_x_bound they skip them -- and a promise that does not hold costs a fence, never
    # a wrong number
    xs_big = np.ascontiguousarray(np.stack([rng_x.permutation(60)[:8] + 1 for _ in range(N)]).astype(np.int32))
    ys_big = [[int.from_bytes(rng_x.bytes(40), "big") % tc.dsa_q for _ in range(8)] for _ in range(N)]
    want_big = [T.calculate_s(list(zip([int(v) for v in xs_big[i]], ys_big[i])), tc.dsa_q) for i in range(N)]
    mb = _ints_to_be([tc.dsa_q], 32)
    d_x, d_y = up(xs_big), up(_ints_to_be(flat(ys_big), 32))
    for bound, ok in ((0, True), (10, False), (60, True), (0, True)):
        gpu_ctx._check(lib.bftkv_gpu_set_lagrange_x_bound(h, bound), "set_lagrange_x_bound")
        o = torch.zeros((N, 32), dtype=torch.uint8, device="cuda:0")
        st = torch.zeros(N + 8, dtype=torch.uint8, device="cuda:0")
        gpu_ctx._check(lib.bftkv_gpu_lagrange_combine_dev(h, N, 8, P(d_x), P(d_y), 32, None, 1, _ptr(mb), P(o), P(st)), "lagrange_dev")
        gpu_ctx.sync()
        stn = st.cpu().numpy()[:N]
        if ok:
            assert not stn.any() and [toi(r) for r in o.cpu().numpy()] == want_big, bound
        else:
            big = np.array([bench_needs_big(xs_big[i]) for i in range(N)])
            assert big.sum() > N // 2 and (stn[big] == 2).all() and not stn[~big].any()
            assert [toi(r) for r, b_ in zip(o.cpu().numpy(), big) if not b_] == [w for w, b_ in zip(want_big, big) if not b_]
    # CalculateR
    o = torch.zeros((N, 32), dtype=torch.uint8, device="cuda:0")
    st = torch.zeros(N + 8, dtype=torch.uint8, device="cuda:0")
    pb, qb = _ints_to_be([tc.dsa_p], 256), _ints_to_be([tc.dsa_q], 32)
    d_x, d_ri, d_vi = up(tc.r_xs), up(_ints_to_be(flat(tc.r_ri), 256)), up(_ints_to_be(flat(tc.r_vi), 32))
    gpu_ctx._check(lib.bftkv_gpu_dsa_calculate_r_dev(h, N, 8, P(d_x), P(d_ri), 256, P(d_vi), 32,
                                                     None, 1, _ptr(pb), _ptr(qb), P(o), P(st)), "calculate_r_dev")
    gpu_ctx.sync()
    got, stn = o.cpu().numpy(), st.cpu().numpy()
    for i in range(0, N, 13):
        rs = [(int(tc.r_xs[i][j]), tc.r_ri[i][j].to_bytes(256, "big"), tc.r_vi[i][j]) for j in range(8)]
        assert stn[i] == 0 and toi(got[i]) == T.calculate_r(rs, tc.dsa_p, tc.dsa_q)


def test_partial_r_is_a_per_operation_exponent(gpu_ctx):
    """CalculatePartialR (crypto/threshold/dsa/dsa.go:27-31): r_i = g^a_i mod p, one exponent per operation."""
    g_ = KAT["dsa_group"]
    p, q, g = int(g_["p"], 16), int(g_["q"], 16), int(g_["g"], 16)
    rng = np.random.default_rng(12)
    n = 70
    a = [int.from_bytes(rng.bytes(40), "big") % q for _ in range(n)]
    a[0], a[1] = 0, 1
    nb = (p.bit_length() + 7) // 8
    base = np.frombuffer(b"".join(g.to_bytes(nb, "big") for _ in range(n)), dtype=np.uint8).reshape(n, nb).copy()
    mods = np.frombuffer(p.to_bytes(nb, "big"), dtype=np.uint8).reshape(1, nb).copy()
    exps = np.frombuffer(b"".join(x.to_bytes(20, "big") for x in a), dtype=np.uint8).reshape(n, 20).copy()
    out = gpu_ctx.modexp_ops(base, np.zeros(n, dtype=np.uint32), mods, exps)
    assert [int.from_bytes(out[i].tobytes(), "big") for i in range(n)] == [int.from_bytes(T.calculate_partial_r(g, x, p), "big") for x in a]


@pytest.mark.parametrize("lanes", (4, 8))
def test_conditional_subtraction_on_chosen_inputs(gpu_ctx, lanes):
    """reduce_once (mont28.h): v - m for v >= m, v otherwise, v < 2m.  The verify path takes the subtraction for about one
    value in 2^50 (only when the top limbs of v reach those of m), so it is driven here: values around m and 2m, borrows that
    cross every lane boundary (limb runs of zeros), moduli whose top lane is empty (1024 bits: every value takes the path)."""
    rng = np.random.default_rng(77)
    mods, vals, idx = [], [], []
    for bits in (2047, 2047, 1024, 1536, 600):
        m = int.from_bytes(rng.bytes(256), "big") >> (2048 - bits) | (1 << (bits - 1)) | 1
        mi = len(mods)
        mods.append(m)
        cases = [0, 1, m - 1, m, m + 1, 2 * m - 1, m >> 1, m + (m >> 1)]
        for sh in (28 * 19, 28 * 38, 28 * 57, 28 * 10, 28 * 40, 28 * 70, 1000, 531):   # lane boundaries of both forms
            if (1 << sh) < m:
                cases += [m + (1 << sh) - 1, m + (1 << sh), m - (1 << sh), (m >> sh << sh), (m >> sh << sh) + m - 1]
        cases += [int.from_bytes(rng.bytes(256), "big") % (2 * m) for _ in range(40)]
        for v in cases:
            if 0 <= v < 2 * m:
                vals.append(v); idx.append(mi)
    got = gpu_ctx.selftest_reduce(vals, mods, idx, lanes)
    for v, i, g in zip(vals, idx, got):
        assert g == (v - mods[i] if v >= mods[i] else v), (lanes, mods[i].bit_length(), hex(v)[:20])


def test_one_combine_per_call_through_the_micro_batcher(gpu_ctx):
    """BASELINE config 5 behind the reference's seam: Client.DistSign drives ONE ThresholdProcess whose ProcessResponse ends in ONE
    combine (rsa.go:235-253, dsa_core.go:318-362, sss.go:69-79).  256 threads, each issuing one operation per call through
    bftkv_gpu_batcher_{modmul_product,lagrange_combine,dsa_calculate_r,modexp}; every byte against oracle/c/threshold.c (the
    reference's arithmetic on OpenSSL bignums) and the reference's TestCombine known answer; moduli differ between callers of one
    device call; what the batched entry points refuse for a whole call is refused for that caller alone."""
    import threading
    from bftkv_amd import Batcher
    from corpus import build as cb
    from oracle.cbind import CThreshold
    g_ = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "keys_dsa2048.json")))["keys"][0]
    as_int = lambda v: int(v, 16) if isinstance(v, str) else int(v)
    N = 768
    r = KAT["rsa"]
    n_kat = int(r["n"], 16)
    sss_m = int(KAT["sss"]["pb"], 16)
    p1, q1 = as_int(g_["p"]), as_int(g_["q"])
    kg = KAT["dsa_group"]
    p2, q2, g2 = int(kg["p"], 16), int(kg["q"], 16), int(kg["g"], 16)      # the reference's 1024 / 160-bit test group
    tc = cb.make_threshold_corpus(N, n_kat, sss_m, p1, q1, seed=501)
    rng = np.random.default_rng(502)
    n_other = (int.from_bytes(rng.bytes(256), "big") | (1 << 2047) | 1)       # a second odd 2048-bit modulus for half the callers
    CT = CThreshold()
    be = lambda vals, nb: np.frombuffer(b"".join(int(v).to_bytes(nb, "big") for v in vals), dtype=np.uint8).copy()
    flat = lambda rows: [v for row in rows for v in row]
    toi = lambda row: int.from_bytes(row.tobytes(), "big")
    # expected answers from the C restatement, per modulus
    rsa_mod = [n_kat if i % 2 == 0 else n_other for i in range(N)]
    # (factors of the odd rows are residues of n_kat, i.e. possibly >= n_other's... both are 2048 bits: reduced like big.Int.Mod)
    want_rsa = {}
    for m in (n_kat, n_other):
        rows = [i for i in range(N) if rsa_mod[i] == m]
        out = CT.rsa_combine(be(flat([tc.rsa_factors[i] for i in rows]), 256), 10, 256, m)
        want_rsa.update({i: toi(out[j]) for j, i in enumerate(rows)})
    o, st = CT.lagrange_combine(tc.sss_xs, be(flat(tc.sss_ys), 256), 256, sss_m)
    want_sss = [toi(x) for x in o]; assert not st.any()
    o, st = CT.lagrange_combine(tc.s_xs, be(flat(tc.s_ys), 32), 32, q1)
    want_s = [toi(x) for x in o]; assert not st.any()
    NR = 160                                                                    # CalculateR: a chain of ~1,200 products per operation
    o, st = CT.dsa_calculate_r(tc.r_xs[:NR], be(flat(tc.r_ri[:NR]), 256), 256, be(flat(tc.r_vi[:NR]), 32), 32, p1, q1)
    want_r = [toi(x) for x in o]; assert not st.any()
    # the same operation in the reference's own small group (dsa_test.go:26-28), 20-byte order: its own shape, its own device call
    xs2 = [[int(v) for v in tc.r_xs[i][:4]] for i in range(24)]
    ri2 = [[pow(g2, int(tc.r_vi[i][j]) % q2 + 1, p2) for j in range(4)] for i in range(24)]
    vi2 = [[int(tc.s_ys[i][j]) % q2 for j in range(4)] for i in range(24)]
    want_r2 = [T.calculate_r([(xs2[i][j], ri2[i][j].to_bytes(128, "big"), vi2[i][j]) for j in range(4)], p2, q2) for i in range(24)]
    exps = [int.from_bytes(rng.bytes(32), "big") % q1 for _ in range(64)]
    g1 = as_int(g_["g"])

    b = Batcher(gpu_ctx, max_items=256, n_lanes=3)
    jobs = []
    for i in range(N):
        jobs.append(("rsa", i)); jobs.append(("sss", i)); jobs.append(("s", i))
    jobs += [("r", i) for i in range(NR)] + [("r2", i) for i in range(24)] + [("exp", i) for i in range(64)]
    order = rng.permutation(len(jobs))
    results, errors = {}, []
    T_ = 256

    def worker(t):
        try:
            for j in order[t::T_]:
                kind, i = jobs[j]
                if kind == "rsa":
                    res = b.modmul_product(tc.rsa_factors[i], rsa_mod[i])
                elif kind == "sss":
                    res = b.lagrange_combine([int(v) for v in tc.sss_xs[i]], tc.sss_ys[i], sss_m)
                elif kind == "s":
                    res = b.lagrange_combine([int(v) for v in tc.s_xs[i]], tc.s_ys[i], q1, nbytes=32)
                elif kind == "r":
                    res = b.dsa_calculate_r([int(v) for v in tc.r_xs[i]], tc.r_ri[i], tc.r_vi[i], p1, q1)
                elif kind == "r2":
                    res = b.dsa_calculate_r(xs2[i], ri2[i], vi2[i], p2, q2, pbytes=128, qbytes=20)
                else:
                    res = b.modexp(g1, exps[i], p1)
                results[(kind, i)] = res
        except Exception as e:      # pragma: no cover
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T_)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errors, errors[:3]
    assert len(results) == len(jobs)
    for (kind, i), (rc, st_, val) in results.items():
        want = {"rsa": lambda: want_rsa[i], "sss": lambda: want_sss[i], "s": lambda: want_s[i], "r": lambda: want_r[i],
                "r2": lambda: want_r2[i], "exp": lambda: pow(g1, exps[i], p1)}[kind]()
        assert (rc, st_) == (0, 0) and val == want, (kind, i, rc, st_)
    stats = b.stats()
    assert stats["calls"] == len(jobs) and stats["batches"] < len(jobs) // 4, stats      # callers did share device calls

    # the reference's own known answer (rsa_test.go:165-206 TestCombine): the product of the ten fragments' partial signatures
    digest = hashlib.sha256(r["tbs"].encode()).digest()
    m = T.emsa_encode("sha256", digest, n_kat)
    d = int(r["d"], 16)
    rnd = [int.from_bytes(rng.bytes(513), "big") % (1 << (2 * d.bit_length())) for _ in range(9)]
    psigs = [T.partial_sign(m, x, n_kat) for x in T.split_key(d, 10, rnd)]
    rc, st_, sig = b.modmul_product(psigs, n_kat)
    assert (rc, st_) == (0, 0) and T.i2os(sig, 256).hex() == r["sha256_pkcs1v15_sig"]
    # ... and TestSSS's secret out of a real dealing
    secret = int.from_bytes(KAT["sss"]["secret"].encode(), "big")
    shares = T.distribute(secret, 10, 7, sss_m, [int.from_bytes(rng.bytes(256), "big") % sss_m for _ in range(6)])
    rc, st_, got = b.lagrange_combine([s_[0] for s_ in shares[2:9]], [s_[1] for s_ in shares[2:9]], sss_m)
    assert (rc, st_, got) == (0, 0, secret)
    # factors at and above the modulus are reduced like big.Int.Mod reduces them
    big_f = [n_kat, n_kat + 5, (1 << 2048) - 1, 7]
    rc, st_, got = b.modmul_product(big_f, n_kat)
    assert (rc, st_) == (0, 0) and got == 0
    rc, st_, got = b.modmul_product(big_f[1:], n_kat)
    assert (rc, st_) == (0, 0) and got == (5 * ((1 << 2048) - 1) * 7) % n_kat
    # fail closed, caller by caller: an even modulus; a denominator that shares a factor with the modulus (math/big's ModInverse
    # returns nil there and the reference dereferences it); a repeated x is NOT an error (sss.Lagrange skips every res == x)
    rc, st_, got = b.modmul_product([3, 5], n_kat + 1)
    assert rc == -4 and st_ == 0xFF and got == 0
    rc, st_, got = b.lagrange_combine([1, 4], [5, 6], 9, nbytes=32)
    assert rc == 0 and st_ == 1 and got == 0
    rc, st_, got = b.lagrange_combine([1, 2, 2], [5, 6, 7], q1, nbytes=32)
    assert (rc, st_) == (0, 0) and got == T.calculate_s([(1, 5), (2, 6), (2, 7)], q1)
    rc, st_, got = b.dsa_calculate_r([1, 2], [3, 4], [0, 0], p1, q1)             # v = 0 has no inverse mod q
    assert rc == 0 and st_ == 1 and got == 0
    b.close()
    rc, st_, got = Batcher.modmul_product(type("Dead", (), {"lib": gpu_ctx.lib, "h": None})(), [3, 5], n_kat)
    assert rc != 0 and st_ == 0xFF and got == 0


def test_one_op_entries_on_random_shapes_and_unreduced_operands(gpu_ctx):
    """The one-operation entries over shapes nobody tuned for -- k from 1 to 12, numbers of 20 / 32 / 128 / 256 bytes, odd moduli that
    are prime or composite and narrower than their byte width, operands ANYWHERE below 2^(8 nbytes) (the reference reduces every
    product and sum mod m, so an unreduced operand must give the residue's result) -- against Python integers through the
    oracle's own functions.  A combine whose Lagrange denominator shares a factor with the modulus must come back as
    BFTKV_TH_NO_INVERSE with zeroes, never as a number."""
    from bftkv_amd import Batcher
    rng = np.random.default_rng(2026)
    rnd = lambda nb: int.from_bytes(rng.bytes(nb), "big")
    b = Batcher(gpu_ctx, max_items=64, n_lanes=2)
    kg = KAT["dsa_group"]
    p2, q2 = int(kg["p"], 16), int(kg["q"], 16)
    g_ = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "keys_dsa2048.json")))["keys"][1]
    as_int = lambda v: int(v, 16) if isinstance(v, str) else int(v)
    p1, q1 = as_int(g_["p"]), as_int(g_["q"])
    n_no_inverse = 0
    try:
        for trial in range(160):
            nb = int(rng.choice([20, 32, 128, 256]))
            k = int(rng.integers(1, 13))
            m = (rnd(nb) >> int(rng.integers(0, 8 * nb - 9))) | 1          # odd, anything from 9 bits to the full width
            if m < 3:
                m = 3
            kind = trial % 4
            if kind == 0:
                f = [rnd(nb) for _ in range(k)]
                rc, st, got = b.modmul_product(f, m, nbytes=nb)
                assert (rc, st) == (0, 0) and got == T.calculate_signature(f, m), (trial, nb, k, hex(m)[:20])
            elif kind == 1:
                xs = [int(v) for v in rng.choice(np.arange(1, 40), size=k, replace=False)]
                ys = [rnd(nb) for _ in range(k)]
                rc, st, got = b.lagrange_combine(xs, ys, m, nbytes=nb)
                try:
                    want = T.calculate_secret(list(zip(xs, ys)), m)
                except ValueError:
                    want = None
                if want is None:
                    assert (rc, st, got) == (0, 1, 0), (trial, xs, hex(m)[:20])
                    n_no_inverse += 1
                else:
                    assert (rc, st) == (0, 0) and got == want, (trial, nb, k, xs, hex(m)[:20])      # (x up to 39: the big-integer path too)
            elif kind == 2:
                base, e = rnd(nb), rnd(int(rng.integers(1, 40)))
                el = int(rng.integers((e.bit_length() + 7) // 8 or 1, 48))
                rc, st, got = b.modexp(base, e, m, nbytes=nb, exp_len=el)
                assert (rc, st) == (0, 0) and got == pow(base, e, m), (trial, nb, el, hex(m)[:20])
            else:
                # (an order of one word as well: the reduction of r mod q must divide, not subtract its way down)
                p, q, pb, qb = ((p1, q1, 256, 32) if trial % 8 == 3 else (p2, 1237, 128, 2) if trial % 8 == 7 else (p2, q2, 128, 20))
                kk = int(rng.integers(1, 9))
                xs = [int(v) for v in rng.choice(np.arange(1, 30), size=kk, replace=False)]
                ri = [rnd(pb) for _ in range(kk)]                     # anywhere below 2^(8 pbytes): Exp reduces its base mod p
                vi = [rnd(qb) for _ in range(kk)]                     # ... and Vi * l is reduced mod q
                rc, st, got = b.dsa_calculate_r(xs, ri, vi, p, q, pbytes=pb, qbytes=qb)
                try:
                    want = T.calculate_r([(x, r.to_bytes(pb, "big"), v) for x, r, v in zip(xs, ri, vi)], p, q)
                except ValueError:
                    want = None
                if want is None:
                    assert (rc, st, got) == (0, 1, 0), trial
                else:
                    assert (rc, st) == (0, 0) and got == want, (trial, kk, xs)
        assert n_no_inverse >= 3          # the composite moduli did meet denominators they share a factor with
    finally:
        b.close()


def test_degenerate_moduli_and_operands_return_at_once(gpu_ctx):
    """Moduli of two, three and nine bits under operands of the full byte width (every reduction divides or runs a fixed number of
    rounds: none subtracts its way down), zero operands, a single share, an exponent of zero; even and zero moduli are refused per
    caller.  Answers are Python's."""
    import time
    from bftkv_amd import Batcher
    big = (1 << 2048) - 1
    b = Batcher(gpu_ctx, max_items=8, max_wait_us=50, n_lanes=1)
    t0 = time.time()
    try:
        for m in (3, 5, 511, 1237):
            rc, st, got = b.modmul_product([big, big - 2, 1 << 2047], m)
            assert (rc, st, got) == (0, 0, (big * (big - 2) * (1 << 2047)) % m), m
            rc, st, got = b.modexp(big, big, m, exp_len=256)
            assert (rc, st, got) == (0, 0, pow(big, big, m)), m
            rc, st, got = b.modexp(big, 0, m, exp_len=4)
            assert (rc, st, got) == (0, 0, 1 % m), m
            xs, ys = [1, 2, 3], [big, 0, big - 1]
            rc, st, got = b.lagrange_combine(xs, ys, m)
            try:
                want = T.calculate_secret(list(zip(xs, ys)), m)
            except ValueError:
                want = None
            assert (rc, st, got) == ((0, 0, want) if want is not None else (0, 1, 0)), m
            rc, st, got = b.lagrange_combine([7], [big], m)                      # one share: the secret is that share
            assert (rc, st, got) == (0, 0, big % m), m
        p = int(KAT["sss"]["pb"], 16)
        for q in (3, 5, 1237):
            xs, ri, vi = [1, 2], [big, p - 1], [q - 1, 1]
            rc, st, got = b.dsa_calculate_r(xs, ri, vi, p, q, qbytes=2)
            try:
                want = T.calculate_r([(x, r.to_bytes(256, "big"), v) for x, r, v in zip(xs, ri, vi)], p, q)
            except ValueError:
                want = None
            assert (rc, st, got) == ((0, 0, want) if want is not None else (0, 1, 0)), q
        # share indices at the ends of int32 (differences of 2^32 - 1), and an index of zero (its product is 0 for every other term)
        for xs in ([-(1 << 31), (1 << 31) - 1, 5], [0, 3, -7, (1 << 31) - 1], [-(1 << 31), -(1 << 31) + 1]):
            for m in (p, 1237, (1 << 255) - 19):
                ys = [big % m, 1, m - 1, 12345][:len(xs)]
                rc, st, got = b.lagrange_combine(xs, ys, m)
                try:
                    want = T.calculate_secret(list(zip(xs, ys)), m)
                except ValueError:
                    want = None
                assert (rc, st, got) == ((0, 0, want) if want is not None else (0, 1, 0)), (xs, m)
        for m in (0, 1, 2, 1 << 2047):                                          # zero, one and even moduli: refused for this caller alone
            assert b.modmul_product([3, 5], m)[0] != 0 and b.lagrange_combine([1, 2], [3, 4], m)[0] != 0 and b.modexp(3, 5, m)[0] != 0
    finally:
        b.close()
    assert time.time() - t0 < 30


def test_lagrange_coefficients_beyond_31_bits(gpu_ctx):
    """sss.Lagrange has no bound on its integers (big.Int); the kernels' fast path holds them in 31 bits -- enough for the
    reference's own n = 10, not for the clusters BASELINE names.  64 and 256 nodes: real dealings recovered through the
    big-integer path (exact products of up to 2128 bits, ONE modular inverse per operation), mixed in one call with operations that
    stay on the fast path; negative x; a denominator that shares a factor with a composite modulus (status 1, as the reference's nil
    dereference); products beyond 2128 bits stay fenced (status 2); CalculateR with 2t = 22 of 64 nodes; one operation per call."""
    from bftkv_amd import Batcher
    m = int(KAT["sss"]["pb"], 16)
    kg = KAT["dsa_group"]
    p, q, g = int(kg["p"], 16), int(kg["q"], 16), int(kg["g"], 16)
    rng = np.random.default_rng(64)
    rnd = lambda mod: int.from_bytes(rng.bytes(264), "big") % mod
    secret = int.from_bytes(b"a secret shared among 64 and 256 nodes", "big")
    # ---- SSS: n = 64, k = 22 and n = 256, k = 171 (real dealings); a fast-path operation and a fenced one in the same call
    cases, want, want_st = [], [], []
    for n, k in ((64, 22), (256, 171), (64, 43), (10, 7)):
        shares = T.distribute(secret, n, k, m, [rnd(m) for _ in range(k - 1)])
        pick = [shares[i] for i in rng.choice(n, size=k, replace=False)]
        cases.append(pick); want.append(secret); want_st.append(0)
    assert T.calculate_secret(cases[0], m) == secret
    for pick, w, ws in zip(cases, want, want_st):
        xs, ys = [[s_[0] for s_ in pick]], [[s_[1] for s_ in pick]]
        got, st = gpu_ctx.lagrange_combine(xs, ys, [m], [0])
        assert (int(st[0]), got[0]) == (ws, w), (len(pick), int(st[0]))
    # one call, k = 22: big path / fast path (x in 1..8 padded with repeats is not possible: distinct small x instead) / negative x /
    # no inverse under a composite modulus / fenced
    k = 22
    comp = 3 * 5 * 7 * 11 * 13 * (2**89 - 1)
    rows = [
        ([int(v) for v in rng.choice(np.arange(1, 65), size=k, replace=False)], m),
        ([int(v) for v in rng.choice(np.arange(-40, 0), size=k, replace=False)], q),
        ([int(v) for v in rng.choice(np.arange(1, 65), size=k, replace=False)], comp),          # differences of 3, 5, 7 ... share factors with it
        ([int(v) for v in rng.choice(np.arange(2**30, 2**31 - 1), size=k, replace=False)], m),  # 21 x 31 bits: fits
    ]
    xs, ys, mods = [r_[0] for r_ in rows], [[rnd(r_[1]) for _ in range(k)] for r_ in rows], [r_[1] for r_ in rows]
    got, st = gpu_ctx.lagrange_combine(xs, ys, mods, list(range(len(rows))))
    for i, (x, mod) in enumerate(rows):
        try:
            w = T.calculate_s(list(zip(x, ys[i])), mod)
        except ValueError:
            w = None
        if w is None:
            assert int(st[i]) == 1, i
        else:
            assert (int(st[i]), got[i]) == (0, w), (i, int(st[i]))
    assert int(st[2]) == 1
    # 120 factors of 31 bits: 3,720 bits, beyond the exact integers the big path holds -- fenced, never a number
    x_huge = [int(v) for v in rng.choice(np.arange(2**30, 2**31 - 1), size=120, replace=False)]
    got, st = gpu_ctx.lagrange_combine([x_huge], [[rnd(m) for _ in range(120)]], [m], [0])
    assert int(st[0]) == 2
    # ---- CalculateR: 2t = 22 of n = 64 nodes (the exponents are the big-path coefficients mod q)
    n, t = 64, 11
    rq = lambda: int.from_bytes(rng.bytes(40), "big") % q
    xs_r, ri, vi, want_r = [], [], [], []
    for trial in range(6):
        kk, aa = rq(), rq()
        ks = T.distribute(kk, n, t, q, [rq() for _ in range(t - 1)])
        as_ = T.distribute(aa, n, t, q, [rq() for _ in range(t - 1)])
        zs = T.distribute(0, n, 2 * t, q, [rq() for _ in range(2 * t - 1)])
        pick = [int(i) for i in rng.choice(n, size=2 * t, replace=False)]
        rs = [(as_[i][0], T.calculate_partial_r(g, as_[i][1], p), (ks[i][1] * as_[i][1] + zs[i][1]) % q) for i in pick]
        xs_r.append([r_[0] for r_ in rs]); ri.append([int.from_bytes(r_[1], "big") for r_ in rs]); vi.append([r_[2] for r_ in rs])
        want_r.append(T.calculate_r(rs, p, q))
        assert want_r[-1] == pow(g, pow(kk, -1, q), p) % q
    got, st = gpu_ctx.dsa_calculate_r(xs_r, ri, vi, [(p, q)], [0] * len(xs_r), pbytes=128, qbytes=20)
    assert list(st) == [0] * len(xs_r) and got == want_r
    # ---- the same shapes, one operation per call
    b = Batcher(gpu_ctx, max_items=16, n_lanes=2)
    try:
        pick = cases[0]
        assert b.lagrange_combine([s_[0] for s_ in pick], [s_[1] for s_ in pick], m) == (0, 0, secret)
        assert b.dsa_calculate_r(xs_r[0], ri[0], vi[0], p, q, pbytes=128, qbytes=20) == (0, 0, want_r[0])
        rc, st_, got1 = b.lagrange_combine(rows[2][0], ys[2], comp)
        assert (rc, st_, got1) == (0, 1, 0)
        rc, st_, got1 = b.lagrange_combine(x_huge, [1] * 120, m)
        assert (rc, st_, got1) == (0, 2, 0)
    finally:
        b.close()
