"""Hostile bytes into every host-side parser of libbftkv_gpu.so (include/bftkv_host.h): mutations of the certificate shapes of
tests/cert_shapes.py and random bytes.  No oracle here -- tests/ compare the parsers with the oracle; this is for
tools/sanitize_host.sh, where the point is that AddressSanitizer / UBSan stay silent.
  python tools/fuzz_host_parsers.py [seed = 1] [seconds = 60]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bftkv_amd import host as H      # noqa: E402
import cert_shapes as CS             # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = random.Random(seed)
seeds = []
for sc in CS.scenarios():
    for v in (sc if isinstance(sc, (tuple, list)) else [sc]):
        if isinstance(v, (bytes, bytearray)) and len(v) > 8:
            seeds.append(bytes(v))
for b in CS.random_blobs(200, seed):
    seeds.append(bytes(b if isinstance(b, (bytes, bytearray)) else b[0]))
# well-formed inputs of the other parsers: bftkv packets (packet.Parse / TBS / TBSS), collective-signature streams and detached
# signatures (signers_walk / scan_stream / walk_stream / parse_signature), gpg-made transport messages (message_frame)
import json                          # noqa: E402
G = os.path.join(ROOT, "tests", "golden")
ref = json.load(open(os.path.join(G, "reference_inputs.json")))
seeds += [bytes.fromhex(p) for p in ref["packets"]]
seeds += [bytes.fromhex(s["ss"]) for s in ref["streams"]] + [bytes.fromhex(g["sig"]) for g in ref["gpg"]]
seeds += [bytes.fromhex(m["msg"]) for k in ("D", "E") for m in json.load(open(os.path.join(G, "gpg_messages.json")))[k][:40]]


def mutate(b):
    b = bytearray(b)
    for _ in range(rng.randrange(1, 6)):
        r = rng.random()
        if not b:
            b += bytes([rng.randrange(256)])
        elif r < 0.4:
            b[rng.randrange(len(b))] = rng.randrange(256)
        elif r < 0.55:
            i = rng.randrange(len(b))
            del b[i:i + rng.randrange(1, 40)]
        elif r < 0.7:
            i = rng.randrange(len(b))
            b[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 40)))
        elif r < 0.8:
            b = b[:rng.randrange(len(b))]
        elif r < 0.9:                    # packet headers and length octets of every form
            b[rng.randrange(len(b))] = rng.choice([0xFF, 0xFE, 0xE0, 0xC2, 0x88, 0x00, 0x99, 0xC6, 0xCD, 0xD1])
        else:
            i, j = rng.randrange(len(b)), rng.randrange(len(b))
            b[i:i] = b[j:j + rng.randrange(1, 200)]
    return bytes(b)


def quietly(fn, *a):
    try:
        fn(*a)
    except (ValueError, RuntimeError, OSError):      # the wrappers' own "malformed input" answers
        pass


n, t0 = 0, time.time()
while time.time() - t0 < budget:
    base = rng.choice(seeds) if rng.random() < 0.85 else bytes(rng.randrange(256) for _ in range(rng.randrange(0, 600)))
    blob = mutate(base)
    quietly(H.Certificate.Parse, blob)
    H.cert_fingerprint(blob)
    H.signers_walk(blob)
    H.scan_stream(blob)
    H.walk_stream(blob)
    H.parse_signature(blob[:rng.randrange(0, len(blob) + 1)])
    quietly(H.packet.Parse, blob)
    quietly(H.packet.TBS, blob)
    quietly(H.packet.TBSS, blob)
    quietly(H.message_frame, blob)
    n += 1
print("seed %d: %d inputs through 10 parsers each, %d seed shapes" % (seed, n, len(seeds)))
