"""extras.bin for tools/fakehip/stress.c: certificates and signed transport messages from the committed fixtures, as two blob lists
(u32 count, u64 offsets[count + 1], bytes).   python tools/fakehip/make_extras.py OUT.bin"""
import json
import os
import struct
import sys

G = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")


def blobs(fh, items):
    off = [0]
    for b in items:
        off.append(off[-1] + len(b))
    fh.write(struct.pack("<I", len(items)) + struct.pack("<%dQ" % len(off), *off) + b"".join(items))


ref = json.load(open(os.path.join(G, "reference_inputs.json")))
certs = [bytes.fromhex(c) for c in ref["certs"]]
msgs = [bytes.fromhex(m["msg"]) for k in ("D", "E") for m in json.load(open(os.path.join(G, "gpg_messages.json")))[k]]
with open(sys.argv[1], "wb") as fh:
    blobs(fh, certs)
    blobs(fh, msgs)
print("%d certificates, %d messages" % (len(certs), len(msgs)))
