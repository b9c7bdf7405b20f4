"""Differential fuzz of Signature.parse on the GPU box: valid quorum signatures with 1-3 bytes of the signature header /
subpacket areas (the first 64 bytes of the packet body -- what the parse kernel reads through its LDS window -- and the MPI
length fields) overwritten at random, several packets per item, GPU per-packet statuses and verdicts against the Python oracle.
Items the library fences are skipped (it makes no claim for them).  usage: python tools/fuzz_bodies.py [rounds=20] [seed=1]
BFTKV_FUZZ_PIECES=N runs the same through the pipelined host-buffer path (N pieces on worker contexts)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (HIP runtime first, README)
import helpers as H
from bftkv_amd import Context
from corpus import build as cb
from oracle import collective as col
from oracle.packet import SignaturePacket


def main(rounds, seed):
    ctx = Context(0)
    if os.environ.get("BFTKV_FUZZ_PIECES"):          # the pipelined host-buffer path: that many pieces, bound-sized arenas
        ctx.set_host_pipeline(int(os.environ["BFTKV_FUZZ_PIECES"]))
    cl = cb.make_cluster(7, dsa_fraction=0.3)
    kr = H.oracle_keyring(cl)
    ctx.keyring_set(H.abi_keys(kr))
    q = H.clique_quorum(cl)
    qh = ctx.quorum_create(H.abi_qcs(q))
    rng = np.random.default_rng(seed)
    from corpus.keys import DRBG
    srng = DRBG("fuzz-bodies-%d" % seed)
    n_pk = n_diff = n_fenced = 0
    hist = {}
    for rd in range(rounds):
        tbs_l, ss_l = [], []
        for i in range(200):
            tbs = rng.bytes(int(rng.integers(0, 200)))
            parts = []
            for r in rng.permutation(len(cl.replicas))[:int(rng.integers(1, 7))]:
                s = bytearray(cb.detach_sign(cl.replicas[int(r)], tbs, srng))
                hl = 3 if s[1] >= 192 else 2
                if rng.random() < 0.85:
                    for _ in range(int(rng.integers(1, 4))):
                        pos = hl + int(rng.integers(0, min(64, len(s) - hl)))
                        mode = int(rng.integers(0, 4))
                        if mode == 0: s[pos] ^= 1 << int(rng.integers(0, 8))
                        elif mode == 1: s[pos] = int(rng.integers(0, 256))
                        elif mode == 2: s[pos] = int(rng.choice([0, 1, 2, 3, 4, 5, 16, 32, 191, 192, 254, 255]))
                        else: s[pos] = (s[pos] + int(rng.choice([-1, 1]))) & 0xFF
                parts.append(bytes(s))
            tbs_l.append(tbs); ss_l.append(b"".join(parts))
        off_t = np.zeros(len(tbs_l) + 1, dtype=np.uint64); off_t[1:] = np.cumsum([len(p) for p in tbs_l], dtype=np.uint64)
        off_s = np.zeros(len(ss_l) + 1, dtype=np.uint64); off_s[1:] = np.cumsum([len(p) for p in ss_l], dtype=np.uint64)
        tb = np.frombuffer(b"".join(tbs_l) + b"\0", dtype=np.uint8)[:int(off_t[-1])].copy()
        sb = np.frombuffer(b"".join(ss_l) + b"\0", dtype=np.uint8)[:int(off_s[-1])].copy()
        for mode in (0, 1):
            ctx.set_early_exit(bool(mode))
            err, nver, _ = ctx.collective_verify(qh, tb, off_t, sb, off_s)
            fenced = ctx.last_fenced
            st, st_item = ctx.last_statuses()
            for i in range(len(tbs_l)):
                if fenced[i]:
                    n_fenced += mode == 0
                    continue
                r = col.collective_verify(kr, tbs_l[i], SignaturePacket(1, 0, False, ss_l[i] or None, None), q)
                got = list(st[st_item == i])
                want = r.statuses
                if mode:            # early exit: statuses up to the exit are the reference's
                    ok = got[:len(want)] == want
                else:               # verify-everything: the oracle stops at the exit, the library goes on
                    ok = got[:len(want)] == want
                ok = ok and (err[i] == 0) == (r.err is None) and nver[i] == len(r.verified)
                n_pk += len(want) if mode == 0 else 0
                for s_ in want:
                    hist[s_] = hist.get(s_, 0) + (mode == 0)
                if not ok:
                    n_diff += 1
                    print("DIFF round %d item %d mode %d: gpu %s err %d nver %d | oracle %s err %s nver %d\n  ss=%s" % (
                        rd, i, mode, got[:10], err[i], nver[i], want[:10], r.err, len(r.verified), ss_l[i][:120].hex()), flush=True)
    print("fuzz_bodies: %d rounds, %d oracle-examined packets, %d fenced items skipped, %d differing items; status histogram %s" % (
        rounds, n_pk, n_fenced, n_diff, dict(sorted(hist.items()))))
    return 1 if n_diff else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
