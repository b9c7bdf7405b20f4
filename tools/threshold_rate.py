"""Ad-hoc measurement for BASELINE configs[4] (threshold share-combine): 10,000 combine operations per scheme through the
C ABI (host pointers in, host pointers out -- PCIe included), with the Python oracle timed on a sample beside it.

    python tools/threshold_rate.py [--ops 10000] > profiles/<round>_cfg5_threshold_rates.json
Not the bench line (bench.py measures configs[1]); numbers feed DESIGN.md.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bftkv_amd import Context                      # noqa: E402
from bftkv_amd._native import _ints_to_be, _ptr    # noqa: E402
from oracle import threshold as T                  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def rand_ints(rng, n, mod):
    nb = (mod.bit_length() + 7) // 8 + 8
    return [int.from_bytes(rng.bytes(nb), "big") % mod for _ in range(n)]


def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", type=int, default=10000)
    ap.add_argument("--nodes", type=int, default=10, help="share indices are drawn from 1..nodes (10: the reference's own parameters)")
    ap.add_argument("--k", type=int, default=0, help="shares per combine (default: 7 for SSS, 8 for threshold DSA, the reference's); with "
                    "--nodes 64 --k 22 the Lagrange coefficients leave the 31-bit fast path: the big-integer kernels")
    a = ap.parse_args()
    kat = json.load(open(os.path.join(GOLD, "threshold_kat.json")))
    dsa = json.load(open(os.path.join(GOLD, "keys_dsa2048.json")))
    k0 = dsa["keys"][0]
    as_int = lambda v: int(v, 16) if isinstance(v, str) else int(v)
    p2048, q256 = as_int(k0["p"]), as_int(k0["q"])
    n_rsa, pb = int(kat["rsa"]["n"], 16), int(kat["sss"]["pb"], 16)
    ctx = Context(0)
    lib, h = ctx.lib, ctx.h
    rng = np.random.default_rng(1)
    N = a.ops
    res = {"ops": N, "note": "wall time of one C-ABI call on host buffers (H2D + kernels + D2H); oracle = CPython big ints, 1 thread, on a sample"}

    # --- RSA: S = prod of 10 partial signatures mod N (rsa.go:318-329)
    k = 10
    fac = rand_ints(rng, N * k, n_rsa)
    f = _ints_to_be(fac, 256); m = _ints_to_be([n_rsa], 256); mi = np.zeros(N, dtype=np.uint32); out = np.zeros((N, 256), dtype=np.uint8)
    call = lambda: ctx._check(lib.bftkv_gpu_modmul_product(h, N, k, _ptr(f), 256, _ptr(mi), 1, _ptr(m), _ptr(out)), "modmul_product")
    call()
    dt = timed(call)
    S = 200
    t0 = time.perf_counter(); want = [T.calculate_signature(fac[i * k:(i + 1) * k], n_rsa) for i in range(S)]; cpu = (time.perf_counter() - t0) / S
    assert [int.from_bytes(out[i].tobytes(), "big") for i in range(S)] == want
    res["rsa_combine_n10"] = {"gpu_ops_per_s": N / dt, "ms": dt * 1e3, "oracle_ops_per_s": 1 / cpu}

    # --- SSS calculateSecret k=7 mod the 2048-bit prime (sss.go:69-92) and calculateS 2t=8 mod q (dsa_core.go:389-403)
    for name, kk, mod, nbytes in (("sss_calculate_secret_k%d_2048" % (a.k or 7), a.k or 7, pb, 256), ("dsa_calculate_s_2t%d_q256" % (a.k or 8), a.k or 8, q256, 256)):
        xs = np.stack([rng.choice(np.arange(1, a.nodes + 1), size=kk, replace=False) for _ in range(N)]).astype(np.int32)
        ys = rand_ints(rng, N * kk, mod)
        y = _ints_to_be(ys, nbytes); m = _ints_to_be([mod], nbytes); out = np.zeros((N, nbytes), dtype=np.uint8); st = np.zeros(N + 8, dtype=np.uint8)
        xs = np.ascontiguousarray(xs)
        call = lambda: ctx._check(lib.bftkv_gpu_lagrange_combine(h, N, kk, _ptr(xs), _ptr(y), nbytes, _ptr(mi), 1, _ptr(m), _ptr(out), _ptr(st)), "lagrange")
        call()
        dt = timed(call)
        t0 = time.perf_counter()
        want = [T.calculate_s(list(zip([int(v) for v in xs[i]], ys[i * kk:(i + 1) * kk])), mod) for i in range(S)]
        cpu = (time.perf_counter() - t0) / S
        assert [int.from_bytes(out[i].tobytes(), "big") for i in range(S)] == want and not st[:N].any()
        res[name] = {"gpu_ops_per_s": N / dt, "ms": dt * 1e3, "oracle_ops_per_s": 1 / cpu}

    # --- threshold DSA CalculateR over 2t=8 partial r's, 2048/256-bit group (dsa.go:33-52)
    kk = a.k or 8
    xs = np.ascontiguousarray(np.stack([rng.choice(np.arange(1, a.nodes + 1), size=kk, replace=False) for _ in range(N)]).astype(np.int32))
    ri = rand_ints(rng, N * kk, p2048)
    vi = rand_ints(rng, N * kk, q256)
    r = _ints_to_be(ri, 256); v = _ints_to_be(vi, 32); p = _ints_to_be([p2048], 256); q = _ints_to_be([q256], 32)
    out = np.zeros((N, 32), dtype=np.uint8); st = np.zeros(N + 8, dtype=np.uint8)
    call = lambda: ctx._check(lib.bftkv_gpu_dsa_calculate_r(h, N, kk, _ptr(xs), _ptr(r), 256, _ptr(v), 32, _ptr(mi), 1, _ptr(p), _ptr(q), _ptr(out), _ptr(st)), "calcr")
    call()
    dt = timed(call)
    S2 = 20
    t0 = time.perf_counter()
    want = []
    for i in range(S2):
        rs = [(int(xs[i][j]), ri[i * kk + j].to_bytes(256, "big"), vi[i * kk + j]) for j in range(kk)]
        try:
            want.append(T.calculate_r(rs, p2048, q256))
        except ValueError:
            want.append(None)
    cpu = (time.perf_counter() - t0) / S2
    for i in range(S2):
        if want[i] is not None:
            assert st[i] == 0 and int.from_bytes(out[i].tobytes(), "big") == want[i]
    res["dsa_calculate_r_2t%d_2048_256" % kk] = {"gpu_ops_per_s": N / dt, "ms": dt * 1e3, "oracle_ops_per_s": 1 / cpu}
    res["nodes"], res["k"] = a.nodes, a.k
    print(json.dumps(res))


if __name__ == "__main__":
    main()
