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


def packet_cases():
    """Byte strings for packet.TBS / packet.TBSS: well-formed requests, every truncation class, stale and negative lengths
    (seek2tbs, packet/packet.go:142-154, ignores the errors of binary.Read and Seek -- round 4 restated that literally)."""
    import struct
    import numpy as np
    from oracle import packet as opk
    u64 = lambda v: struct.pack(">q", int(v))
    sig = opk.SignaturePacket(1, 0, False, b"s" * 40, b"c" * 30)
    ss = opk.SignaturePacket(1, 0, True, b"t" * 50, None)
    full = opk.serialize(b"variable", b"value-bytes", 9, sig, ss, b"auth")
    cases = [full, opk.serialize(b"v"), opk.serialize(b"v", b"w", 1), opk.serialize(b"v", b"w", 1, sig), b"", b"abc",
             u64(0) + b"abcd", u64(2) + b"ab" + b"xyz", u64(5) + b"ab", u64(-9) + u64(1) + b"v" + u64(9) + b"tail", u64(-1) + u64(1) + b"v" + u64(9),
             u64((1 << 63) - 1) + u64(0) + u64(7) + b"!", u64(0) + u64(0) + u64(3), u64(0) + u64(0) + u64(3) + bytes(22)]
    cases += [full[:k] for k in range(0, len(full), 7)]
    rng = np.random.default_rng(41)
    lens = [0, 1, 2, 5, 8, 9, 17, -1, -9, -17, 1 << 20]          # (no absurd positive lengths: TBS would allocate them)
    for _ in range(120):
        l1, l2 = (lens[int(rng.integers(len(lens)))] for _ in range(2))
        body = u64(l1) + rng.bytes(int(rng.integers(0, 12))) + u64(l2) + rng.bytes(int(rng.integers(0, 12))) + u64(rng.integers(0, 1 << 62))
        if rng.random() < 0.5:
            body += opk.write_signature(opk.SignaturePacket(1, 0, False, rng.bytes(int(rng.integers(0, 9))), None))
        cases.append(body[:int(rng.integers(0, len(body) + 1))])
    return [c.hex() for c in cases]


def cert_cases():
    """Certificate blobs for crypto.Certificate.Parse (openpgp.ReadEntity until it fails): what bftkv_gpu_batcher_cert_verify and
    bftkv_host_certs_verify answer for on the GPU -- valid entities, forged self-signature, forged subkey binding, several
    entities with a bad one in the middle, a DSA principal, garbage."""
    cl = cb.make_cluster(6, dsa_fraction=0.34)
    rng = DRBG("reference-inputs-certs")
    client = cl.client.entity
    selfsig = client.index(b"\xc2", client.index(cl.client.name.encode()))
    bad_self = bytearray(client); bad_self[selfsig + 200] ^= 1
    sub_owner = cb.make_keypair(cb.PK_RSA, cb.load_keys("rsa2048", 84)[82], "s01 <s01@bftkv.example>")
    sub = cb.make_keypair(cb.PK_RSA, cb.load_keys("rsa2048", 84)[83], "")
    cb.build_entity(sub_owner, [], rng, subkey=sub)
    bad_binding = bytearray(sub_owner.entity); bad_binding[len(bad_binding) - 25] ^= 0x20
    dsa = next(r for r in cl.replicas if r.algo == cb.PK_DSA)
    blobs = [client, bytes(bad_self), sub_owner.entity, bytes(bad_binding), dsa.entity, b"".join(r.entity for r in cl.replicas[:3]),
             cl.replicas[0].entity + bytes(bad_self) + cl.replicas[1].entity, bytes(bad_self) + cl.replicas[1].entity, b"", bytes(range(200)),
             client[:len(client) // 2]]
    # the ReadEntity walk shape by shape (tests/cert_shapes.py: hand-built entities, verdicts worked from x/crypto's rules) and
    # the certificates GnuPG made (gpg_cert_vectors.json: cross-signed signing subkeys, revocations, several user ids ...)
    from tests import cert_shapes as CS
    blobs += [blob for _, blob, _, _, _ in CS.scenarios()]
    gv = json.load(open(os.path.join(HERE, "gpg_cert_vectors.json")))
    blobs += [bytes.fromhex(c["blob"]) for c in gv["certificates"] + gv["tampered"]]
    return [b.hex() for b in blobs]


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
    # detached signatures gpg made with named keys of certificates that carry signing subkeys (KeysByIdUsage over Subkey.Sig)
    gv = json.load(open(os.path.join(HERE, "gpg_cert_vectors.json")))
    for k, c in enumerate(gv["certificates"]):
        dets = list(c.get("detached", [])) + ([c["detached_by_revoked_subkey"]] if "detached_by_revoked_subkey" in c else [])
        if dets:
            out["rings"]["cert/%d" % k] = c["blob"]
        for j, d in enumerate(dets):
            out["gpg"].append({"name": "cert/%d/%d by %s" % (k, j, d["key_id"]), "ring": "cert/%d" % k, "tbs": d["payload"], "sig": d["sig"]})
    out["packets"] = packet_cases()
    out["certs"] = cert_cases()
    path = os.path.join(HERE, "reference_inputs.json")
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print("wrote %s: %d clusters, %d streams, %d gpg vectors, %d bytes" % (path, len(out["clusters"]), len(out["streams"]), len(out["gpg"]), os.path.getsize(path)))


if __name__ == "__main__":
    main()
