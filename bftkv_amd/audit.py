"""Re-verify a stored bftkv database on the GPU: storage/plain directories and storage/leveldb
databases.

The reference's plain storage keeps every accepted write as a file ``hex(x).t`` holding the request bytes
``<x,v,t,sig,ss>`` exactly as Server.write received them (storage/plain/plain.go:48-90, protocol/server.go:348).
``audit_plain_db`` walks such a directory, rebuilds the verifier's inputs from a public key ring
(``pubring.gpg`` as cmd/bftkv/main.go:70-71 loads it) and runs the verification site of Server.write
(server.go:286-302) over all stored writes in device batches.  Product path only: host mirror + C ABI + HIP kernels.

``audit_leveldb_db`` does the same for a storage/leveldb database (storage/leveldb/leveldb.go:30-53: key = variable || t as
8 big-endian bytes, value = the stored packet), read with bftkv_amd/leveldb_reader.py -- no LevelDB library needed.

``--kind http`` replays an opened-body HTTP exchange log instead (transport/http framing, bftkv_amd/wire.py).

CLI:  python -m bftkv_amd.audit --db DIR|CAPTURE --pubring FILE --self KEYID_HEX [--kind plain|leveldb|http]
"""
from __future__ import annotations

import argparse
import os
from dataclasses import dataclass
from typing import Dict, List, Tuple

from . import host
from ._native import Context


@dataclass
class AuditRecord:
    path: str
    variable: bytes
    t: int
    status: str        # "ok" | "insufficient" | "malformed" | "name-mismatch"


def load_ring(ctx: Context, pubring: bytes):
    """Certificate.ParseStream + Keyring.Register + graph.AddNodes (cmd/bftkv/main.go:70-98) for a key ring blob.
    Returns the trust graph (vertices = entities, edges = certifications) after uploading the key table."""
    # (ReadKeyRing skips the entities ReadEntity refuses; the refusals the walk alone decides are honoured here)
    ents = [e for e in host.Certificate.Parse(pubring) if e["keys"] and not e["refused"]]
    keys = [k for e in ents for k in e["keys"]]
    ctx.keyring_set(keys)
    g = host.Graph()
    g.AddNodes([(e["id"], e["certifiers"]) for e in ents])
    return g, ents


def read_plain_db(db_dir: str) -> List[Tuple[str, bytes, int, bytes]]:
    """(path, variable, t, stored bytes) of every ``hex(x).t`` file (storage/plain/plain.go:48-61)."""
    out = []
    for name in sorted(os.listdir(db_dir)):
        stem, dot, tstr = name.rpartition(".")
        if not dot or not tstr.isdigit():
            continue
        try:
            variable = bytes.fromhex(stem)
        except ValueError:
            continue
        path = os.path.join(db_dir, name)
        with open(path, "rb") as f:
            out.append((path, variable, int(tstr), f.read()))
    return out


def audit_plain_db(ctx: Context, db_dir: str, pubring: bytes, self_id: int, batch: int = 4096) -> List[AuditRecord]:
    return _audit(ctx, read_plain_db(db_dir), pubring, self_id, batch)


def read_leveldb_db(db_dir: str) -> List[Tuple[str, bytes, int, bytes]]:
    """(label, variable, t, stored bytes) of every entry of a storage/leveldb database."""
    from . import leveldb_reader
    return [("%s.%d" % (x.hex(), t), x, t, v) for x, t, v in leveldb_reader.bftkv_records(db_dir)]


def audit_leveldb_db(ctx: Context, db_dir: str, pubring: bytes, self_id: int, batch: int = 4096) -> List[AuditRecord]:
    return _audit(ctx, read_leveldb_db(db_dir), pubring, self_id, batch)


def _audit(ctx: Context, files, pubring: bytes, self_id: int, batch: int) -> List[AuditRecord]:
    g, _ = load_ring(ctx, pubring)
    g.SetSelfNodes([self_id])
    q = host.wotqs.New(g).ChooseQuorum(host.AUTH)          # Server.write (server.go:300)
    server = host.Server(ctx)
    records: List[AuditRecord] = []
    for lo in range(0, len(files), batch):
        chunk = files[lo:lo + batch]
        err = server.write_verify(q, [c[3] for c in chunk])
        for (path, variable, t, blob), e in zip(chunk, err):
            status = {0: "ok", 2: "insufficient", 0xFC: "fenced"}.get(int(e), "malformed")
            if status != "malformed":
                x, _, pt, _, _, _ = host.packet.Parse(blob)
                if (x or b"") != variable or pt != t:
                    status = "name-mismatch"
            records.append(AuditRecord(path, variable, t, status))
    return records


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--db", required=True)
    ap.add_argument("--pubring", required=True)
    ap.add_argument("--self", dest="self_id", required=True, help="key id (hex) of the auditing node's own certificate")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--kind", choices=("plain", "leveldb", "http"), default="plain",
                    help="plain / leveldb: a storage database; http: an opened-body HTTP exchange log (bftkv_amd/wire.py), --db names the file")
    args = ap.parse_args()
    ctx = Context(args.device)
    with open(args.pubring, "rb") as f:
        ring = f.read()
    if args.kind == "http":
        from . import wire
        with open(args.db, "rb") as f:
            exs = wire.replay(ctx, f.read(), ring, int(args.self_id, 16))
        counts: Dict[str, int] = {}
        for e in exs:
            counts[e.verdict] = counts.get(e.verdict, 0) + 1
            if e.verdict not in ("consistent", "not-judged"):
                print("%-32s #%d %s %s: transport %s, %s %s, recorded %s %s" % (
                    e.verdict, e.index, e.request.method, e.request.target, e.transport, e.site or "-", e.site_error or "ok",
                    e.response.status if e.response else "-", (e.response.header("X-error") if e.response else "")))
        print("replayed %d exchanges: %s" % (len(exs), ", ".join("%s=%d" % kv for kv in sorted(counts.items()))))
        ctx.close()
        return
    recs = (audit_plain_db if args.kind == "plain" else audit_leveldb_db)(ctx, args.db, ring, int(args.self_id, 16))
    counts: Dict[str, int] = {}
    for r in recs:
        counts[r.status] = counts.get(r.status, 0) + 1
        if r.status != "ok":
            print("%-14s %s" % (r.status, r.path))
    print("audited %d stored writes: %s" % (len(recs), ", ".join("%s=%d" % kv for kv in sorted(counts.items()))))
    ctx.close()


if __name__ == "__main__":
    main()
