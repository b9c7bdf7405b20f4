"""Hostile ARGUMENTS into the C ABI (include/bftkv_gpu.h), for the sanitizer build over tools/fakehip (tools/sanitize_host.sh):
every buffer is as large as the call's own arguments say -- the caller keeps its side of the contract -- but counts, widths,
offsets, indices and values take every awkward shape: zero / one / over-wide widths, k = 0 and k past the limits, even / zero /
one moduli, index arrays pointing past their tables, offsets that are not monotone or do not start at zero, NULL where a length
is zero, quorums with no cliques or too many, key material of impossible sizes.  Kernels do not run there; the point is that
the host side answers with an error code (or succeeds) and AddressSanitizer / UBSan stay silent.
  python tools/fuzz_abi_args.py [seed = 1] [seconds = 60]"""
import ctypes as C
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bftkv_amd import _native as N      # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = random.Random(seed)
lib = N.load_library()
vp = C.c_void_p
ctx = vp()
assert lib.bftkv_gpu_init(0, C.byref(ctx)) == 0, "bftkv_gpu_init (run under tools/fakehip on a machine without a GPU)"
lib.bftkv_gpu_batcher_create_lanes.restype = vp
batcher = vp(lib.bftkv_gpu_batcher_create_lanes(ctx, 16, 0, 2))
assert batcher.value

WIDTHS = [0, 1, 2, 3, 4, 7, 20, 31, 32, 33, 64, 128, 255, 256, 257, 260, 512, 1024]
KS = [0, 1, 2, 3, 7, 8, 12, 13, 22, 64, 171, 256, 1024, 1025, 4096]


def buf(n, kind="rand"):
    """numpy uint8 buffer of exactly n bytes (at least one allocated, so that the pointer is never NULL by accident)"""
    a = np.zeros(max(n, 1), dtype=np.uint8)
    if kind == "rand" and n:
        a[:n] = np.frombuffer(rng.randbytes(n), dtype=np.uint8)
    elif kind == "ff":
        a[:] = 0xFF
    return a


def p8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def moduli(n_mods, nbytes):
    m = buf(n_mods * nbytes)
    for j in range(n_mods):
        if nbytes == 0:
            break
        style = rng.random()
        row = m[j * nbytes:(j + 1) * nbytes]
        if style < 0.55:
            row[-1] |= 1
            row[0] |= 0x80 if rng.random() < 0.7 else 0
        elif style < 0.65:
            row[:] = 0                      # zero
        elif style < 0.75:
            row[:] = 0; row[-1] = 1         # one
        elif style < 0.85:
            row[-1] &= 0xFE                 # even
        elif style < 0.92:
            row[:] = 0; row[-1] = rng.choice([3, 5, 7, 251])    # tiny
        else:
            row[:] = 0xFF
    return m


def idx(n_ops, n_mods):
    a = np.zeros(max(n_ops, 1), dtype=np.uint32)
    for i in range(n_ops):
        a[i] = rng.randrange(max(n_mods, 1)) if rng.random() < 0.9 else rng.choice([n_mods, n_mods + 1, 0xFFFFFFFF])
    return a


def xs_of(n):
    a = np.zeros(max(n, 1), dtype=np.int32)
    for i in range(n):
        r = rng.random()
        a[i] = i + 1 if r < 0.6 else rng.choice([0, -1, 1, 2 ** 31 - 1, -2 ** 31, rng.randrange(-1000, 1000)])
    return a


def small(cands, cap_bytes, per):
    """a count from cands whose buffers (count * per bytes) stay under cap_bytes"""
    c = rng.choice(cands)
    while per and c * per > cap_bytes:
        c //= 2
    return c


calls = {}


def note(name, rc):
    calls.setdefault(name, {}).setdefault(int(rc), 0)
    calls[name][int(rc)] += 1


def threshold_batched():
    n_ops = rng.choice([0, 1, 2, 5, 64, 300])
    nbytes = rng.choice(WIDTHS)
    k = small(KS, 1 << 22, max(n_ops, 1) * max(nbytes, 1))
    n_mods = rng.choice([0, 1, 2, 5])
    mods, mi = moduli(n_mods, nbytes), idx(n_ops, n_mods)
    out, st = buf(n_ops * max(nbytes, 1), "zero"), buf(n_ops, "zero")
    which = rng.randrange(6)
    u32p = mi.ctypes.data_as(vp)
    if which == 0:
        f = buf(n_ops * k * nbytes)
        note("modmul_product", lib.bftkv_gpu_modmul_product(ctx, n_ops, k, p8(f), nbytes, u32p, n_mods, p8(mods), p8(out)))
    elif which == 1:
        ys, xs = buf(n_ops * k * nbytes), xs_of(n_ops * k)
        note("lagrange_combine", lib.bftkv_gpu_lagrange_combine(ctx, n_ops, k, xs.ctypes.data_as(vp), p8(ys), nbytes, u32p, n_mods, p8(mods), p8(out), p8(st)))
    elif which == 2:
        qb = rng.choice([0, 1, 2, 20, 28, 32, 33, 64])
        ri, vi, xs = buf(n_ops * k * nbytes), buf(n_ops * k * qb), xs_of(n_ops * k)
        q = moduli(n_mods, qb)
        r_out = buf(n_ops * max(qb, 1), "zero")
        note("dsa_calculate_r", lib.bftkv_gpu_dsa_calculate_r(ctx, n_ops, k, xs.ctypes.data_as(vp), p8(ri), nbytes, p8(vi), qb, u32p, n_mods, p8(mods), p8(q),
                                                               p8(r_out), p8(st)))
    elif which == 3:
        el = rng.choice([0, 1, 3, 4, 32, 33, 256, 520, 1024, 1025, 2048])
        per_op = rng.random() < 0.5
        ex = buf((n_ops if per_op else n_mods) * el)
        b = buf(n_ops * nbytes)
        fn = lib.bftkv_gpu_modexp_ops if per_op else lib.bftkv_gpu_modexp
        note("modexp_ops" if per_op else "modexp", fn(ctx, n_ops, p8(b), nbytes, u32p, n_mods, p8(mods), p8(ex), el, p8(out)))
    elif which == 4:
        n_sh = rng.choice([0, 1, 4, 10, 64, 256, 1000])
        kk = min(k, 64)
        co = buf(n_ops * kk * nbytes)
        sh = buf(n_ops * n_sh * max(nbytes, 1), "zero")
        note("sss_distribute", lib.bftkv_gpu_sss_distribute(ctx, n_ops, n_sh, kk, p8(co), nbytes, u32p, n_mods, p8(mods), p8(sh)))
    else:
        v = buf(n_ops * nbytes)
        note("modinv", lib.bftkv_gpu_modinv(ctx, n_ops, p8(v), nbytes, u32p, n_mods, p8(mods), p8(out), p8(st)))


def threshold_one():
    nbytes = rng.choice(WIDTHS)
    k = small(KS, 1 << 20, max(nbytes, 1))
    mod = moduli(1, nbytes)
    out, st = buf(max(nbytes, 64), "zero"), C.c_uint8(0x55)
    which = rng.randrange(4)
    if which == 0:
        rc = lib.bftkv_gpu_batcher_modmul_product(batcher, k, p8(buf(k * nbytes)), nbytes, p8(mod), p8(out), C.byref(st))
        name = "batcher_modmul_product"
    elif which == 1:
        rc = lib.bftkv_gpu_batcher_lagrange_combine(batcher, k, xs_of(k).ctypes.data_as(vp), p8(buf(k * nbytes)), nbytes, p8(mod), p8(out), C.byref(st))
        name = "batcher_lagrange_combine"
    elif which == 2:
        qb = rng.choice([0, 1, 2, 20, 32, 33])
        rc = lib.bftkv_gpu_batcher_dsa_calculate_r(batcher, k, xs_of(k).ctypes.data_as(vp), p8(buf(k * nbytes)), nbytes, p8(buf(k * qb)), qb, p8(mod),
                                                   p8(moduli(1, qb)), p8(out), C.byref(st))
        name = "batcher_dsa_calculate_r"
    else:
        el = rng.choice([0, 1, 32, 33, 64, 1024, 1025])
        rc = lib.bftkv_gpu_batcher_modexp(batcher, p8(buf(nbytes)), nbytes, p8(buf(el)), el, p8(mod), p8(out), C.byref(st))
        name = "batcher_modexp"
    assert rc == 0 or st.value == 0xFF, (name, rc, st.value)      # fail closed
    note(name, rc)


def offsets(n, total):
    o = np.zeros(n + 1, dtype=np.uint64)
    style = rng.random()
    if style < 0.6:
        cuts = sorted(rng.randrange(total + 1) for _ in range(max(n - 1, 0)))
        o[1:n] = cuts[:max(n - 1, 0)]
        o[n] = total
        if n == 0:
            o[0] = 0
    elif style < 0.7:
        o[:] = total                      # does not start at zero
    elif style < 0.8:
        for i in range(n + 1):
            o[i] = rng.randrange(total + 1)    # not monotone (never past the blob)
    elif style < 0.9:
        o[:] = 0                          # all empty
    else:
        o[1:] = total                     # one item holds everything
    return o


def verify_calls():
    n = rng.choice([0, 1, 2, 7, 100])
    tb_len, ss_len = rng.choice([0, 1, 64, 5000]), rng.choice([0, 1, 287, 3000, 70000])
    tb, ss = buf(tb_len), buf(ss_len)
    if ss_len >= 2 and rng.random() < 0.7:      # packet headers here and there
        for _ in range(rng.randrange(1, 30)):
            ss[rng.randrange(ss_len)] = rng.choice([0xC2, 0x88, 0x89, 0x8A, 0xFF, 0xE0, 0xC6])
    to, so = offsets(n, tb_len), offsets(n, ss_len)
    err, fen, nver, ver = buf(n, "zero"), buf(n, "zero"), np.zeros(max(n, 1), dtype=np.uint32), buf(n, "zero")
    u64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))
    q = rng.choice([quorum.value, quorum.value, -1, 99, 0x7FFFFFFF])
    which = rng.randrange(6)
    if which == 5:
        # segmented payloads: hostile prefix / shared offsets, segment indices past n_shared, no segments at all
        n_sh = rng.choice([0, 1, 3])
        sh_len = rng.choice([0, 1, 300, 9000])
        sh, sho = buf(sh_len), offsets(n_sh, sh_len)
        seg = np.array([rng.choice([0, 1, 2, 3, 0xFFFFFFFF, 0x7FFFFFFF]) for _ in range(max(n, 1))], dtype=np.uint32)
        rc = lib.bftkv_gpu_collective_verify_segments(ctx, q, n, p8(tb), u64(to), p8(sh), u64(sho), n_sh, seg.ctypes.data_as(vp), p8(ss), u64(so),
                                                      p8(err), nver.ctypes.data_as(vp), p8(ver), p8(fen))
        assert rc == 0 or n == 0 or all(int(e) != 0 for e in err[:n]), ("collective_verify_segments", rc)      # fail closed
        note("collective_verify_segments", rc)
    elif which == 0:
        note("collective_verify", lib.bftkv_gpu_collective_verify(ctx, q, n, p8(tb), u64(to), p8(ss), u64(so), p8(err), nver.ctypes.data_as(vp), p8(ver), p8(fen)))
    elif which == 1:
        note("signature_verify", lib.bftkv_gpu_signature_verify(ctx, n, p8(tb), u64(to), p8(ss), u64(so), None, p8(err), p8(fen)))
    elif which == 2:
        note("collective_verify_small", lib.bftkv_gpu_collective_verify_small(ctx, q, n, p8(tb), u64(to), p8(ss), u64(so), p8(err), p8(fen)))
    elif which == 3:
        cap = rng.choice([0, 1, 16, 4096])
        ids, ioff = np.zeros(max(cap, 1), dtype=np.uint64), np.zeros(n + 1, dtype=np.uint64)
        note("signers_fenced", lib.bftkv_gpu_signers_fenced(ctx, n, p8(ss), u64(so), u64(ids), u64(ioff), cap, p8(fen)))
    else:
        e, f = C.c_uint8(0), C.c_uint8(0)
        rc = lib.bftkv_gpu_batcher_collective_verify(batcher, q, tb.tobytes()[:tb_len], tb_len, ss.tobytes()[:ss_len], ss_len, C.byref(e), C.byref(f))
        assert rc == 0 or e.value != 0, ("batcher_collective_verify", rc, e.value)
        note("batcher_collective_verify", rc)


def keyring_and_quorum():
    if rng.random() < 0.5:
        n = rng.choice([0, 1, 3, 9])
        keys = (N.PubKey * max(n, 1))()
        keep = []
        for i in range(n):
            algo = rng.choice([1, 1, 3, 17, 17, 2, 16, 19, 0, 255])
            ln = [rng.choice([0, 1, 2, 32, 128, 256, 257, 384, 512, 513, 1024]) for _ in range(4)]
            if algo == 17 and rng.random() < 0.6:
                ln = [256, rng.choice([20, 28, 32, 33]), 256, 256]
            bs = [buf(x) for x in ln]
            if rng.random() < 0.7 and ln[0]:
                bs[0][ln[0] - 1] |= 1
            keep.append(bs)
            k = keys[i]
            k.key_id, k.entity_id, k.pk_algo, k.usable_sign = rng.getrandbits(64), rng.getrandbits(64), algo, rng.randrange(2)
            for name, b, x in zip("negy", bs, ln):
                setattr(k, name, b.ctypes.data if (x or rng.random() < 0.5) else None)
                setattr(k, name + "_len", x)
        note("keyring_set", lib.bftkv_gpu_keyring_set(ctx, keys, n))
    else:
        n_qcs = rng.choice([0, 1, 2, 8, 9, 20])
        qcs = (N.QC * max(n_qcs, 1))()
        keep = []
        for i in range(n_qcs):
            nn = rng.choice([0, 1, 4, 64, 300, 5000])
            ids = np.array([rng.getrandbits(64) for _ in range(nn)] or [0], dtype=np.uint64)
            keep.append(ids)
            qc = qcs[i]
            qc.f, qc.min, qc.threshold, qc.suff = (rng.choice([0, 1, 21, -1, 2 ** 31 - 1, -2 ** 31]) for _ in range(4))
            qc.node_ids = ids.ctypes.data if (nn or rng.random() < 0.5) else None
            qc.n_nodes = nn
        h = C.c_int(-1)
        rc = lib.bftkv_gpu_quorum_create(ctx, qcs, n_qcs, C.byref(h))
        note("quorum_create", rc)
        if rc == 0:
            note("quorum_destroy", lib.bftkv_gpu_quorum_destroy(ctx, h.value))
        note("quorum_destroy(bad)", lib.bftkv_gpu_quorum_destroy(ctx, rng.choice([-1, 12345, h.value])))


def _cert_seeds():
    import json
    g = os.path.join(ROOT, "tests", "golden")
    ref = json.load(open(os.path.join(g, "reference_inputs.json")))
    return [bytes.fromhex(c) for c in ref["certs"]], [bytes.fromhex(x["sig"]) for x in ref["gpg"][:40]]


CERTS, SIGS = _cert_seeds()


def cert_calls():
    """request certificates (fixtures, mutated or not -- the register keeps what it has accepted, so repeats take the lane route) with
    signatures that are the fixtures', mutated, empty or absent"""
    cert = bytearray(rng.choice(CERTS))
    r = rng.random()
    if r < 0.3 and cert:
        for _ in range(rng.randrange(1, 4)):
            cert[rng.randrange(len(cert))] = rng.randrange(256)
    elif r < 0.4:
        cert = cert[:rng.randrange(len(cert) + 1)]
    elif r < 0.5:
        cert += rng.choice(CERTS)
    cert = bytes(cert)
    tb = rng.randbytes(rng.choice([0, 1, 63, 64, 65, 300]))
    sig = rng.choice([None, b"", rng.choice(SIGS), rng.choice(SIGS) + rng.choice(SIGS), rng.randbytes(rng.choice([1, 20, 287, 600]))])
    e, f, iid = C.c_uint8(0), C.c_uint8(0), C.c_uint64(0)
    fp = (C.c_uint8 * 20)()
    if rng.random() < 0.75:
        rc = lib.bftkv_gpu_batcher_cert_verify(batcher, cert, len(cert), tb, len(tb), sig, len(sig) if sig is not None else 0, C.byref(e), C.byref(f),
                                               C.byref(iid), fp)
        name = "batcher_cert_verify"
    else:
        cap = rng.choice([0, 1, 8, 256])
        roles = (C.c_uint32 * max(cap, 1))()
        off, ln, nr = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        rc = lib.bftkv_gpu_batcher_cert_entity(batcher, cert, len(cert), C.byref(e), C.byref(f), C.byref(iid), fp, C.byref(off), C.byref(ln), roles, cap,
                                               C.byref(nr))
        name = "batcher_cert_entity"
        assert off.value + ln.value <= len(cert), (name, off.value, ln.value, len(cert))
    assert rc == 0 or e.value != 0, (name, rc, e.value)      # fail closed
    note(name, rc)


# a usable quorum for the verify calls
ids0 = np.arange(1, 5, dtype=np.uint64)
qc0 = (N.QC * 1)()
qc0[0].f, qc0[0].min, qc0[0].threshold, qc0[0].suff = 1, 4, 3, 3
qc0[0].node_ids, qc0[0].n_nodes = ids0.ctypes.data, 4
quorum = C.c_int(-1)
assert lib.bftkv_gpu_quorum_create(ctx, qc0, 1, C.byref(quorum)) == 0

t0 = time.time()
n = 0
while time.time() - t0 < budget:
    rng.choice([threshold_batched, threshold_batched, threshold_one, verify_calls, verify_calls, keyring_and_quorum, cert_calls, cert_calls])()
    n += 1
lib.bftkv_gpu_batcher_destroy(batcher)
lib.bftkv_gpu_destroy(ctx)
print("seed %d: %d calls; return codes per entry: %s" % (seed, n, {k: dict(sorted(v.items())) for k, v in sorted(calls.items())}))
