#!/usr/bin/env python3
"""Where does a SMALL verify call spend its time?  Wall clock of bftkv_gpu_collective_verify (host buffers) for batches of
1 / 64 / 256 cfg-2 shaped writes against the device span of the same call (HIP events), plus a hip-trace summary hint."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402
from bftkv_amd import Context  # noqa: E402
from corpus import build as cb  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

cl = cb.make_cluster(64)
ctx = Context(0)
mods, exps = cb.signer_tables(cl)
c = cb.make_write_corpus(cl, 256, batch_signer=lambda em, ki: ctx.modexp(em, ki.astype(np.uint32), mods, exps))
ctx.keyring_set(bench.abi_keys_of(cl))
f, mn, thr, suff = cb.quorum_numbers(64)
qh = ctx.quorum_create([(f, mn, thr, suff, [r.key_id for r in cl.replicas])])
for n in (1, 64, 256):
    tb, to = c.tbss_blob[:int(c.tbss_off[n])], c.tbss_off[:n + 1]
    sb, so = c.ss_blob[:int(c.ss_off[n])], c.ss_off[:n + 1]
    for _ in range(5):
        ctx.collective_verify(qh, tb, to, sb, so)
    walls, spans, parts = [], [], []
    for _ in range(100):
        t0 = time.perf_counter()
        ctx.collective_verify(qh, tb, to, sb, so)
        walls.append(time.perf_counter() - t0)
        tm = ctx.last_timing()
        spans.append(tm["total"]); parts.append(tm)
    p = {k: round(float(np.median([x[k] for x in parts])), 3) for k in parts[0]}
    print("items %4d  wall p50 %.3f ms  device span p50 %.3f ms  phases %s" % (n, np.median(walls) * 1e3, np.median(spans), p), flush=True)

# the same writes, ONE per call through the micro-batcher from this one thread (no company: the lone-call latency)
from bftkv_amd import Batcher  # noqa: E402
b = Batcher(ctx, max_items=256, n_lanes=2)
items = [(c.tbss(i), c.ss_data(i)) for i in range(64)]
for t, s in items[:8]:
    b.collective_verify(qh, t, s)
walls = []
for k in range(400):
    t, s = items[k % 64]
    t0 = time.perf_counter()
    b.collective_verify(qh, t, s)
    walls.append(time.perf_counter() - t0)
b.close()
w = np.array(walls) * 1e3
print("batcher, one caller thread: wall p50 %.3f ms  p90 %.3f  min %.3f (includes ~5 us of ctypes and the payload's host-side SHA-256 chain)"
      % (np.median(w), np.percentile(w, 90), w.min()), flush=True)
