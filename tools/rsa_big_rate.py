"""Ad-hoc measurement: verification rate for RSA-3072 / RSA-4096 signatures (gpg 2.2's default key size is 3072).

Replicates the gpg-made signatures of tests/golden/gpg_vectors.json (group C) into a large batch -- the work per
verification does not depend on the signature being distinct -- and reports the device time of the modexp phase.
    python tools/rsa_big_rate.py [--sigs 200000]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bftkv_amd import Context              # noqa: E402
from oracle import collective as col       # noqa: E402  (tool only: key parsing for the ABI upload)
from oracle import openpgp as pgp          # noqa: E402
from tests import helpers as H             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sigs", type=int, default=200000)
    a = ap.parse_args()
    vec = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gpg_vectors.json")))
    ents = pgp.read_entities(bytes.fromhex(vec["C_pubring"]))
    ctx = Context(0)
    ctx.keyring_set(H.abi_keys(col.Keyring(ents)))
    qh = ctx.quorum_create([(1, 4, 3, 3, [e.id for e in ents] + [1, 2])])
    for bits, signer in ((3072, "a03@gpg.example"), (4096, "a04@gpg.example")):
        vs = [v for v in vec["C"] if v["signer"] == signer and v["gpg_good"]]
        per_item = 50
        n_items = a.sigs // per_item
        tbs_l, ss_l = [], []
        for i in range(n_items):
            v = vs[i % len(vs)]
            tbs_l.append(bytes.fromhex(v["payload"]))
            ss_l.append(bytes.fromhex(v["sig"]) * per_item)
        tb = np.frombuffer(b"".join(tbs_l) or b"\0", dtype=np.uint8)
        to = np.concatenate([[0], np.cumsum([len(t) for t in tbs_l])]).astype(np.uint64)
        sb = np.frombuffer(b"".join(ss_l), dtype=np.uint8)
        so = np.concatenate([[0], np.cumsum([len(s) for s in ss_l])]).astype(np.uint64)
        for it in range(3):
            err, nver, _ = ctx.collective_verify(qh, tb, to, sb, so)
            tm = ctx.last_timing()
        st, _ = ctx.last_statuses()
        n = len(st)
        print("RSA-%d: %d signatures, ok=%d, modexp phase %.2f ms -> %.1f M verifies/s (device), phases %s" %
              (bits, n, int((st == 0).sum()), tm["rsa"], n / tm["rsa"] / 1e3, {k: round(v, 2) for k, v in tm.items()}))


if __name__ == "__main__":
    main()
