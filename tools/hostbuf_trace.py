#!/usr/bin/env python3
"""A few segmented host-buffer calls of the cfg-2 batch for a kernel trace (rocprofv3 --kernel-trace): what runs between a piece's
modexp and the piece's end.  Prints the library's own timeline of the last call."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401
    import bench
    from bftkv_amd import Context, host as HM
    from corpus import build as cb
    cl = cb.make_cluster(64)
    ctx0 = Context(0)
    signer, _ = bench.gpu_signers(ctx0, cl)
    z = bench.write_corpus_arrays(cb.make_write_corpus(cl, 10000, seed=cb.MASTER_SEED, batch_signer=signer, with_client_sig=True))
    f, mn, thr, suff = cb.quorum_numbers(cl.n)
    ctx0.keyring_set(bench.abi_keys_of(cl))
    qh = ctx0.quorum_create([(f, mn, thr, suff, [r.key_id for r in cl.replicas])])
    pb, po, shb, sho, seg = HM.split_tails(z["tb"], z["to"], [cl.client.entity])
    for _ in range(4):
        ctx0.collective_verify_segments(qh, pb, po, shb, sho, seg, z["sb"], z["so"])
    print(json.dumps(ctx0.host_pipeline_trace()))


if __name__ == "__main__":
    main()
