#!/usr/bin/env python3
"""Writes tests/golden/reference_inputs.json: the inputs on which SOMEONE WITH A GO TOOLCHAIN pins this repository's
oracle to the reference itself (shim/tools/genvectors/main.go reads this file, runs the reference's crypto/pgp and
quorum/wotqs with the x/crypto version go.mod:8 pins, and writes tests/golden/reference_vectors.json;
tests/test_reference_vectors.py is skipped while that file is absent and strict once it exists).

Contents, all seeded (corpus/keys.py) -- nothing here depends on the GPU:
  clusters   clique-certified rings (scripts/clique.sh shape: every member certifies every other), the node that plays
             "self", and quorum signatures over payloads: valid ones and every mutation class of corpus/build.py
  streams    random packet framings around valid signatures (tests/helpers.py random_framing_streams), and the framings
             no writer of the path produces (exotic_framing_streams: partial / indeterminate lengths, oversized bodies)
  gpg        the GnuPG fixtures and gpg-judged edge cases (tests/golden/gpg_vectors.json, gpg_negative_vectors.json)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from corpus import build as cb          # noqa: E402
from corpus.keys import DRBG            # noqa: E402
from tests import helpers as H          # noqa: E402


def cluster_block(n, n_items, dsa_fraction, seed_tag):
    cl = cb.make_cluster(n, dsa_fraction=dsa_fraction)
    rng = DRBG("reference-inputs-" + seed_tag)
    for r in cl.replicas:                                   # clique: everyone certifies everyone (scripts/clique.sh)
        cb.build_entity(r, [o for o in cl.replicas if o is not r], rng)
    rates = {cb.MUT_BAD_MPI: 0.15, cb.MUT_ONE_SHORT: 0.2, cb.MUT_UNKNOWN_ISSUER: 0.1, cb.MUT_BAD_TAG: 0.1, cb.MUT_DUP_SIGNER: 0.1}
    c = cb.make_write_corpus(cl, n_items, mutation_rates=rates)
    items = [{"tbs": c.tbss(i).hex(), "ss": c.ss_data(i).hex()} for i in range(c.n_items)]
    return {"name": "n%d%s" % (n, "-dsa" if dsa_fraction else ""), "pubring": b"".join(r.entity for r in cl.replicas).hex(),
            "outsiders": b"".join(o.entity for o in cl.outsiders).hex(), "self": "%016x" % cl.replicas[0].key_id,
            "members": ["%016x" % r.key_id for r in cl.replicas], "items": items}, cl


def main():
    out = {"format": 1, "clusters": [], "streams": [], "gpg": [], "rings": {}}
    for n, k, dsa, tag in ((4, 24, 0.0, "a"), (10, 16, 0.0, "b"), (7, 16, 0.4, "c")):
        blk, cl = cluster_block(n, k, dsa, tag)
        out["clusters"].append(blk)
        if n == 7:
            tbs_l, stream_l, _, _ = H.random_framing_streams(cl, 24, seed=91)
            out["streams"] = [{"cluster": blk["name"], "tbs": t.hex(), "ss": s.hex()} for t, s in zip(tbs_l, stream_l)]
            # partial / indeterminate lengths, lengths past the end of the stream, bodies beyond bufio's buffer: the reader
            # model of oracle/openpgp.py B.1b (restated from memory of x/crypto and of Go's bufio) is pinned by these
            tbs_l, stream_l = H.exotic_framing_streams(cl, 36, seed=92)
            out["streams"] += [{"cluster": blk["name"], "tbs": t.hex(), "ss": s.hex()} for t, s in zip(tbs_l, stream_l)]
    vec = json.load(open(os.path.join(HERE, "gpg_vectors.json")))
    for ring_key, group in (("A_pubring", "A"), ("B_pubring", "B"), ("C_pubring", "C")):
        for k, v in enumerate(vec[group]):
            out["gpg"].append({"name": "%s/%d" % (group, k), "ring": ring_key, "tbs": v["payload"], "sig": v["sig"]})
        out["rings"][ring_key] = vec[ring_key]
    neg = json.load(open(os.path.join(HERE, "gpg_negative_vectors.json")))
    out["rings"]["neg"] = neg["pubring"]
    for v in neg["vectors"]:
        out["gpg"].append({"name": "neg/" + v["name"], "ring": "neg", "tbs": neg["payload"], "sig": v["sig"]})
    path = os.path.join(HERE, "reference_inputs.json")
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print("wrote %s: %d clusters, %d streams, %d gpg vectors, %d bytes" % (path, len(out["clusters"]), len(out["streams"]), len(out["gpg"]), os.path.getsize(path)))


if __name__ == "__main__":
    main()
