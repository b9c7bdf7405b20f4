#!/usr/bin/env python3
"""Host-buffer rates of the cfg-2 batch: bftkv_gpu_collective_verify and bftkv_gpu_collective_verify_segments, one caller and three
callers at once (what bench.py's `end_to_end` leg measures, alone -- for A/B runs under environment knobs: BFTKV_HB_PIECES,
BFTKV_NO_TURNSTILE, BFTKV_HB_COPY ...).  Every answer is checked against the resident call.  Prints one JSON line."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


CALLS = int(os.environ.get("HOSTBUF_CALLS", "6"))      # calls per caller thread in the three-caller leg


def main():
    import torch
    import bench
    from bftkv_amd import Context, host as HM
    from corpus import build as cb
    items = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    cl = cb.make_cluster(64)
    ctx0 = Context(0)
    signer, _ = bench.gpu_signers(ctx0, cl)
    z = bench.write_corpus_arrays(cb.make_write_corpus(cl, items, seed=cb.MASTER_SEED, batch_signer=signer, with_client_sig=True))
    f, mn, thr, suff = cb.quorum_numbers(cl.n)
    ctx0.keyring_set(bench.abi_keys_of(cl))
    qh = ctx0.quorum_create([(f, mn, thr, suff, [r.key_id for r in cl.replicas])])
    ctxs = [ctx0, ctx0.fork(), ctx0.fork()]
    ctx0.set_host_pipeline(1)
    err, nver, _ = ctx0.collective_verify(qh, z["tb"], z["to"], z["sb"], z["so"])
    ctx0.set_host_pipeline(0)
    pb, po, shb, sho, seg = HM.split_tails(z["tb"], z["to"], [cl.client.entity])
    if os.environ.get("HOSTBUF_PINNED") == "1":      # the caller's buffers page-locked (what bftkv_gpu_host_alloc would hand a caller)
        pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()
        z = {k: (pin(v) if isinstance(v, np.ndarray) and v.ndim == 1 and v.size > 1 else v) for k, v in dict(z).items()}
        pb, po, shb, sho, seg = pin(pb), pin(po), pin(shb), pin(sho), pin(seg)

    def plain(cx):
        e, nv, _ = cx.collective_verify(qh, z["tb"], z["to"], z["sb"], z["so"])
        assert (e == err).all() and (nv == nver).all()

    def segd(cx):
        e, nv, _ = cx.collective_verify_segments(qh, pb, po, shb, sho, seg, z["sb"], z["so"])
        assert (e == err).all() and (nv == nver).all()

    out = {"items": items, "env": {k: v for k, v in os.environ.items() if k.startswith("BFTKV_")}}
    for name, fn in (("plain", plain), ("segments", segd)):
        for _ in range(2):
            fn(ctx0)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); fn(ctx0); ts.append(time.perf_counter() - t0)
        tr = ctx0.host_pipeline_trace()
        spans = {}
        gate = threading.Barrier(3)

        def caller(k):
            fn(ctxs[k]); fn(ctxs[k])
            gate.wait()
            t0 = time.perf_counter()
            for _ in range(CALLS):
                fn(ctxs[k])
            spans[k] = (t0, time.perf_counter())
        th = [threading.Thread(target=caller, args=(k,)) for k in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        span = max(b for _, b in spans.values()) - min(a for a, _ in spans.values())
        tr3 = ctx0.host_pipeline_trace()
        out[name] = {"alone_ms": round(min(ts) * 1e3, 3), "alone_median_ms": round(float(np.median(ts)) * 1e3, 3), "three_callers_ms_per_call": round(span / (3 * CALLS) * 1e3, 3),
                     "pieces": tr["pieces"] if tr else None, "pieces_last_call_of_three_callers": tr3["pieces"] if tr3 else None,
                     "piece_modexp_ms": [round((p["gpu_modexp_end"] - p["gpu_modexp_start"]) / 1e3, 3) for p in tr["per_piece_us"]] if tr else None,
                     "copies_done_us": round(tr["copy_stream_drained_us"]) if tr else None, "done_us": round(tr["done_us"]) if tr else None}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
