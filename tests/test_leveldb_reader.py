"""CPU: bftkv_amd/leveldb_reader.py against a database written by tests/leveldb_writer.py in the published formats, and the
Snappy decoder against hand-assembled streams (format_description.txt)."""
import os
import struct

import numpy as np
import pytest

from bftkv_amd import leveldb_reader as R
from tests import leveldb_writer as W


def test_snappy_elements():
    # "abcdabcdabcdabcd!": literal "abcd", copy(1-byte offset) len 4+? ...
    lit = bytes([(4 - 1) << 2]) + b"abcd"
    copy1 = bytes([((8 - 4) << 2) | 1 | (0 << 5), 4])              # length 8, offset 4: overlapping run
    copy2 = bytes([((4 - 1) << 2) | 2]) + struct.pack("<H", 12)     # length 4, offset 12
    tail = bytes([0 << 2]) + b"!"
    s = W.varint(17) + lit + copy1 + copy2 + tail
    assert R.snappy_decompress(s) == b"abcdabcdabcdabcd!"
    big = bytes(range(256)) * 3                                       # literal with a 2-byte length field
    s = W.varint(len(big)) + bytes([61 << 2]) + struct.pack("<H", len(big) - 1) + big
    assert R.snappy_decompress(s) == big
    copy4 = bytes([((6 - 1) << 2) | 3]) + struct.pack("<I", 6)
    assert R.snappy_decompress(W.varint(12) + bytes([(6 - 1) << 2]) + b"xyzxyz" + copy4) == b"xyzxyzxyzxyz"
    with pytest.raises(R.LevelDBFormatError):
        R.snappy_decompress(W.varint(9) + lit + bytes([((8 - 4) << 2) | 1, 9]))      # offset beyond the output
    rng = np.random.default_rng(1)
    for _ in range(20):
        d = bytes(rng.integers(0, 4, size=int(rng.integers(0, 400))).astype(np.uint8)) + b"\x07" * 24
        assert R.snappy_decompress(W.snappy_compress(d)) == d


def test_database_merge(tmp_path):
    rng = np.random.default_rng(7)
    model = {}
    seq = 1
    tables = []
    for _ in range(2):                                   # two tables (one Snappy, one raw), ascending sequence numbers
        items = {}
        for _ in range(40):
            k = b"var%02d" % int(rng.integers(0, 30)) + struct.pack(">Q", int(rng.integers(1, 4)))
            typ = 1 if rng.random() > 0.15 else 0
            v = bytes(rng.integers(0, 256, size=int(rng.integers(0, 300))).astype(np.uint8)) if typ else b""
            items[k] = (k, seq, typ, v)
            seq += 1
        tables.append(sorted(items.values()))
        for k, s, t, v in items.values():
            if k not in model or model[k][0] < s:
                model[k] = (s, t, v)
    batches = []
    for _ in range(12):                                  # the live log; values up to 50 KB span 32 KiB blocks
        ents = []
        for _ in range(int(rng.integers(1, 5))):
            k = b"var%02d" % int(rng.integers(0, 30)) + struct.pack(">Q", int(rng.integers(1, 4)))
            typ = 1 if rng.random() > 0.2 else 0
            v = bytes(rng.integers(0, 256, size=int(rng.choice([10, 2000, 50000]))).astype(np.uint8)) if typ else b""
            ents.append((typ, k, v))
        batches.append((seq, ents))
        for i, (typ, k, v) in enumerate(ents):
            model[k] = (seq + i, typ, v)                 # later entries of a batch carry later sequence numbers
        seq += len(ents)
    stale = [(0, [(1, b"var00" + struct.pack(">Q", 1), b"stale value from a retired log")])]
    W.write_db(str(tmp_path), tables, batches, stale_log_batches=stale)
    got = R.read_db(str(tmp_path))
    want = {k: v for k, (s, t, v) in model.items() if t == 1}
    assert got == want and len(want) > 20
    recs = R.bftkv_records(str(tmp_path))
    assert [(x + struct.pack(">Q", t)) for x, t, _ in recs] == sorted(want)
    # torn tail: the last log record cut in the middle is dropped, nothing else changes
    logp = os.path.join(str(tmp_path), "%06d.log" % 7)
    raw = open(logp, "rb").read()
    open(logp, "wb").write(raw[:-5])
    assert set(R.read_db(str(tmp_path))) <= set(want) | set(k for _, es in batches for _, k, _ in es)


def test_not_a_database(tmp_path):
    with pytest.raises(R.LevelDBFormatError):
        R.read_db(str(tmp_path))
    os.makedirs(tmp_path / "d")
    (tmp_path / "d" / "CURRENT").write_text("MANIFEST-000002\n")
    (tmp_path / "d" / "MANIFEST-000002").write_bytes(W.log_bytes([W.varint(7) + W.varint(0) + W.varint(9) + W.varint(10) + W.varint(0) + W.varint(0)]))
    with pytest.raises(R.LevelDBFormatError):
        R.read_db(str(tmp_path / "d"))                   # the MANIFEST names a table that is not there
