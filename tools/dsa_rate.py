"""Ad-hoc measurement: verification rate on a BASELINE configs[2]-shaped corpus (half DSA-2048/256 signers).

    python tools/dsa_rate.py [--writes 400] [--tile 16] [--bits 8]
Not a bench line (bench.py measures configs[1]); prints per-call wall time and the DSA signature rate.
"""
import argparse
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bftkv_amd import Context           # noqa: E402
from corpus import build as cb          # noqa: E402
from tests import helpers as H          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--writes", type=int, default=400)
    ap.add_argument("--tile", type=int, default=16)
    ap.add_argument("--bits", type=int, default=0)
    ap.add_argument("--dsa-fraction", type=float, default=0.5)
    ap.add_argument("--dsa-kind", default="dsa2048", help="dsa1024 (q 160 bits), dsa1536 (224), dsa2048 (256), dsa3072 (256: the 8-lane kernel), "
                    "or several separated by commas (dealt round-robin)")
    ap.add_argument("--replicas", type=int, default=64)
    ap.add_argument("--json", action="store_true", help="one JSON line instead of the per-call prints")
    a = ap.parse_args()
    ctx = Context(0)
    ctx.set_dsa_window_bits(a.bits)
    kinds = a.dsa_kind.split(",")
    cl = cb.make_cluster(a.replicas, dsa_fraction=a.dsa_fraction, dsa_kind=kinds[0] if len(kinds) == 1 else kinds)
    mods, exps = cb.signer_tables(cl)
    signer = lambda em, ki: ctx.modexp(em, ki.astype(np.uint32), mods, exps)
    c = cb.make_write_corpus(cl, a.writes, batch_signer=signer, seed=5)
    kr = H.oracle_keyring(cl)
    t0 = time.perf_counter()
    ctx.keyring_set(H.abi_keys(kr))
    ctx.sync()
    t_ring = time.perf_counter() - t0
    if not a.json:
        print("keyring_set (tables for %d DSA keys): %.1f ms" % (sum(r.algo == cb.PK_DSA for r in cl.replicas), 1e3 * t_ring))
    qh = ctx.quorum_create(H.abi_qcs(H.clique_quorum(cl)))
    # tile the corpus: offsets are absolute, so repeat blobs and shift
    T = a.tile
    tb = np.tile(c.tbss_blob, T); sb = np.tile(c.ss_blob, T)
    to = np.concatenate([c.tbss_off[:-1] + i * c.tbss_off[-1] for i in range(T)] + [np.array([T * c.tbss_off[-1]], dtype=np.uint64)]).astype(np.uint64)
    so = np.concatenate([c.ss_off[:-1] + i * c.ss_off[-1] for i in range(T)] + [np.array([T * c.ss_off[-1]], dtype=np.uint64)]).astype(np.uint64)
    ctx.set_host_pipeline(1)          # one call on one context: last_timing describes it
    dsa_ms = []
    for it in range(4):
        t0 = time.perf_counter()
        err, nver, _ = ctx.collective_verify(qh, tb, to, sb, so)
        dt = time.perf_counter() - t0
        st, _ = ctx.last_statuses()
        tm = ctx.last_timing()
        dsa_ms.append(tm["dsa"])
        if not a.json:
            print("call %d: %.2f ms host wall, device %s, %d sigs, ok=%d, writes ok=%d/%d" % (it, dt * 1e3, tm, len(st), int((st == 0).sum()), int((err == 0).sum()), len(err)))
    counters = ctx.last_counters()
    n_dsa = int(counters["dsa_ops"])
    if a.json:
        import json
        bits = ctx.dsa_window_bits()
        held, entry = ctx.dsa_table_bytes()
        products = 2 * ((256 + bits - 1) // bits) - 1
        # limb MACs of a general Montgomery product: 2 N^2 (a*b and m*n), N = 76 (4 x 19) or 112 (8 x 14) limbs by the key's size class
        big = sum(1 for r in cl.replicas if r.algo == cb.PK_DSA and r.p.bit_length() > 2048)
        small = sum(1 for r in cl.replicas if r.algo == cb.PK_DSA) - big
        macs = products * (2 * 76 * 76 * small + 2 * 112 * 112 * big) / max(1, small + big)
        ms = min(dsa_ms[1:])
        print(json.dumps({"dsa_kind": a.dsa_kind, "dsa_keys": small + big, "window_bits": bits, "entry_limbs": entry, "table_gb": held / 1e9,
                          "keyring_set_ms": t_ring * 1e3, "dsa_verifies_per_call": n_dsa, "k_dsa_mul_modexp_ms": ms,
                          "dsa_verifies_per_sec": n_dsa / (ms * 1e-3), "products_per_verify": products,
                          "int_mac_frac_of_measured_roof": n_dsa * macs / (ms * 1e-3) / 29.66e12,
                          "accepted_writes": int((err == 0).sum()), "writes": len(err)}))
    else:
        print("DSA public-key operations per call: %d" % n_dsa)


if __name__ == "__main__":
    main()
