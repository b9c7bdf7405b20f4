"""TEST INFRASTRUCTURE: writes a small LevelDB directory in the published on-disk formats (doc/log_format.md,
doc/table_format.md) so that bftkv_amd/leveldb_reader.py can be exercised without a LevelDB library: one or more sorted tables
(prefix-compressed keys, restart interval 16, blocks stored raw or as Snappy streams made of literal and copy elements), a
MANIFEST naming them, and a write-ahead log whose batches span 32 KiB block boundaries."""
import os
import struct

BLOCK = 32768
MAGIC = 0xDB4775248B80FB57


def varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def snappy_compress(data: bytes) -> bytes:
    """A valid Snappy stream: literals, plus a copy element wherever the next 8 bytes repeat the 8 bytes before them
    (enough to exercise the reader's copy path; not meant to compress well)."""
    out = bytearray(varint(len(data)))
    lit = bytearray()

    def flush():
        nonlocal lit
        while lit:
            chunk, lit = lit[:60], lit[60:]
            out.append((len(chunk) - 1) << 2)
            out.extend(chunk)
    i = 0
    while i < len(data):
        if i >= 8 and data[i:i + 8] == data[i - 8:i] and len(data) - i >= 8:
            flush()
            out.append(((8 - 1) << 2) | 2)              # copy, 2-byte offset, length 8
            out += struct.pack("<H", 8)
            i += 8
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def build_block(entries, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def ikey(user_key, seq, typ=1):
    return user_key + struct.pack("<Q", (seq << 8) | typ)


def write_table(path, items, compress=True, per_block=7):
    """items: sorted [(user_key, seq, type, value)]"""
    data = bytearray()
    index = []

    def put(block):
        raw = snappy_compress(block) if compress else block
        off = len(data)
        data.extend(raw + bytes([1 if compress else 0]) + b"\0\0\0\0")
        return off, len(raw)
    for i in range(0, len(items), per_block):
        chunk = items[i:i + per_block]
        off, size = put(build_block([(ikey(k, s, t), v if t == 1 else b"") for k, s, t, v in chunk]))
        last = chunk[-1]
        index.append((ikey(last[0], last[1], last[2]), varint(off) + varint(size)))
    moff, msize = put(build_block([]))
    ioff, isize = put(build_block(index, restart_interval=1))
    footer = varint(moff) + varint(msize) + varint(ioff) + varint(isize)
    footer += b"\0" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    with open(path, "wb") as f:
        f.write(bytes(data) + footer)
    return len(data) + 48


def log_bytes(records):
    out = bytearray()
    for rec in records:
        first = True
        while True:
            left = BLOCK - (len(out) % BLOCK)
            if left < 7:
                out += b"\0" * left
                continue
            take = min(len(rec), left - 7)
            last = take == len(rec)
            typ = 1 if first and last else 2 if first else 4 if last else 3
            out += b"\0\0\0\0" + struct.pack("<HB", take, typ) + rec[:take]
            rec = rec[take:]
            first = False
            if last:
                break
    return bytes(out)


def batch(seq, entries):
    out = struct.pack("<QI", seq, len(entries))
    for typ, k, v in entries:
        out += bytes([typ]) + varint(len(k)) + k
        if typ == 1:
            out += varint(len(v)) + v
    return out


def write_db(path, table_items, log_batches, stale_log_batches=()):
    """table_items: list of item lists (one table each); log_batches: [(seq, [(type, key, value)])] for the live log."""
    os.makedirs(path, exist_ok=True)
    edits = varint(1) + varint(len(b"leveldb.BytewiseComparator")) + b"leveldb.BytewiseComparator"
    num = 5
    for items in table_items:
        size = write_table(os.path.join(path, "%06d.ldb" % num), items, compress=(num % 2 == 1))
        edits += varint(7) + varint(0) + varint(num) + varint(size)
        for it in (items[0], items[-1]):
            k = ikey(it[0], it[1], it[2])
            edits += varint(len(k)) + k
        num += 1
    if stale_log_batches:                                   # a log the MANIFEST has already retired (number < log number)
        with open(os.path.join(path, "%06d.log" % 3), "wb") as f:
            f.write(log_bytes([batch(s, e) for s, e in stale_log_batches]))
    log_no = num
    edits += varint(2) + varint(log_no) + varint(3) + varint(log_no + 1) + varint(4) + varint(10 ** 6)
    with open(os.path.join(path, "MANIFEST-000002"), "wb") as f:
        f.write(log_bytes([edits]))
    with open(os.path.join(path, "CURRENT"), "w") as f:
        f.write("MANIFEST-000002\n")
    with open(os.path.join(path, "%06d.log" % log_no), "wb") as f:
        f.write(log_bytes([batch(s, e) for s, e in log_batches]))
