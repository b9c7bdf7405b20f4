#!/usr/bin/env python3
"""Writes a binary corpus for tools/serving/cert_load.c: sign requests of CLIENTS that are not in the server's keyring -- the shape of
protocol/server.go:199-207, where the issuer's certificate travels in sig.Cert -- signed on the CPU (no GPU needed).
    python tools/serving/make_cert_corpus.py OUT.bin [clients=64] [replicas=10]
Layout: u32 replicas, per replica (u64 key id, 256 B modulus, 4 B exponent); then three blob lists (u32 count, u64 offsets[count+1],
bytes): the clients' certificates (own self-signature + f+1 certifications by replicas), one tbs per client, one detached signature."""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from corpus import build as cb  # noqa: E402
from corpus.keys import DRBG  # noqa: E402


def blobs(fh, items):
    off = [0]
    for b in items:
        off.append(off[-1] + len(b))
    fh.write(struct.pack("<I", len(items)) + struct.pack("<%dQ" % len(off), *off) + b"".join(items))


def main():
    out = sys.argv[1]
    n_clients = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    cl = cb.make_cluster(reps)
    mats = cb.load_keys("rsa2048", 84)
    first = reps + 3                                     # (the cluster took reps + client + two outsiders)
    assert first + n_clients <= len(mats), "the key fixture holds %d RSA keys" % len(mats)
    certifiers = cl.replicas[:cl.f + 1]
    certs, tbss, sigs = [], [], []
    for i in range(n_clients):
        kp = cb.make_keypair(cb.PK_RSA, mats[first + i], "c%03d <c%03d@bftkv.example>" % (i, i))
        cb.build_entity(kp, certifiers, DRBG("cert-load", i))
        tbs = cb.serialize_tbs(b"variable-%04d" % i, b"value" * 12, 1 + i)
        certs.append(kp.entity)
        tbss.append(tbs)
        sigs.append(cb.detach_sign(kp, tbs, DRBG("cert-load-sig", i)))
    with open(out, "wb") as fh:
        fh.write(struct.pack("<I", reps))
        for r in cl.replicas:
            fh.write(struct.pack("<Q", r.key_id) + r.n.to_bytes(256, "big") + r.e.to_bytes(4, "big"))
        blobs(fh, certs); blobs(fh, tbss); blobs(fh, sigs)
    print("wrote %s: %d clients, certificates of %d bytes" % (out, n_clients, len(certs[0])))


if __name__ == "__main__":
    main()
