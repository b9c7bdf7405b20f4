#!/usr/bin/env python3
"""Writes a binary corpus for tools/serving/batcher_load.c: cfg-2 shaped signed writes (64 replicas, RSA-2048), signed on this GPU.
    python tools/serving/make_load_corpus.py OUT.bin [n_writes=4096] [replicas=64]"""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402  (HIP runtime first)
from bftkv_amd import Context  # noqa: E402
from corpus import build as cb  # noqa: E402


def main():
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    cl = cb.make_cluster(reps)
    ctx = Context(0)
    mods, exps = cb.signer_tables(cl)
    c = cb.make_write_corpus(cl, n, batch_signer=lambda em, ki: ctx.modexp(em, ki.astype(np.uint32), mods, exps))
    ctx.close()
    f, mn, thr, suff = cb.quorum_numbers(reps)
    with open(out, "wb") as fh:
        fh.write(struct.pack("<I", reps))
        for r in cl.replicas:
            fh.write(struct.pack("<Q", r.key_id) + r.n.to_bytes(256, "big") + r.e.to_bytes(4, "big"))
        fh.write(struct.pack("<iiiiI", f, mn, thr, suff, n))
        fh.write(c.tbss_off.astype("<u8").tobytes() + c.ss_off.astype("<u8").tobytes())
        fh.write(c.tbss_blob.tobytes() + c.ss_blob.tobytes())
        fh.write(((c.expected_valid >= cl.suff).astype(np.uint8)).tobytes())
    print("wrote %s: %d writes, %d signature packets" % (out, n, c.n_sigs))


if __name__ == "__main__":
    main()
