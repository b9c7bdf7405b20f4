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
    a = ap.parse_args()
    ctx = Context(0)
    ctx.set_dsa_window_bits(a.bits)
    cl = cb.make_cluster(64, dsa_fraction=a.dsa_fraction)
    mods, exps = cb.signer_tables(cl)
    signer = lambda em, ki: ctx.modexp(em, ki.astype(np.uint32), mods, exps)
    c = cb.make_write_corpus(cl, a.writes, batch_signer=signer, seed=5)
    kr = H.oracle_keyring(cl)
    t0 = time.perf_counter()
    ctx.keyring_set(H.abi_keys(kr))
    ctx.sync()
    print("keyring_set (tables for %d DSA keys): %.1f ms" % (sum(r.algo == cb.PK_DSA for r in cl.replicas), 1e3 * (time.perf_counter() - t0)))
    qh = ctx.quorum_create(H.abi_qcs(H.clique_quorum(cl)))
    # tile the corpus: offsets are absolute, so repeat blobs and shift
    T = a.tile
    tb = np.tile(c.tbss_blob, T); sb = np.tile(c.ss_blob, T)
    to = np.concatenate([c.tbss_off[:-1] + i * c.tbss_off[-1] for i in range(T)] + [np.array([T * c.tbss_off[-1]], dtype=np.uint64)]).astype(np.uint64)
    so = np.concatenate([c.ss_off[:-1] + i * c.ss_off[-1] for i in range(T)] + [np.array([T * c.ss_off[-1]], dtype=np.uint64)]).astype(np.uint64)
    for it in range(4):
        t0 = time.perf_counter()
        err, nver, _ = ctx.collective_verify(qh, tb, to, sb, so)
        dt = time.perf_counter() - t0
        st, _ = ctx.last_statuses()
        tm = ctx.last_timing()
        print("call %d: %.2f ms host wall, device %s, %d sigs, ok=%d, writes ok=%d/%d" % (it, dt * 1e3, tm, len(st), int((st == 0).sum()), int((err == 0).sum()), len(err)))
    n_dsa = c.n_sigs * T * a.dsa_fraction
    print("approx DSA sigs per call: %d" % n_dsa)


if __name__ == "__main__":
    main()
