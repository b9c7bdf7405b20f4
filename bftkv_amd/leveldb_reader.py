"""Read-only access to a LevelDB directory as goleveldb / LevelDB write it -- for auditing a bftkv `storage/leveldb`
database (storage/leveldb/leveldb.go:30-53: key = variable || t as 8 big-endian bytes, value = the stored packet
<x,v,t,sig,ss>; `Read(variable, 0)` takes the last key with that prefix).

Implements the published on-disk formats (LevelDB doc/log_format.md, doc/table_format.md, doc/impl.md) without any LevelDB
library (none is installed, and the audit tool must not need one):
  CURRENT -> MANIFEST-n        log-format file of VersionEdits: the live table files per level and the current log number
  nnnnnn.log                   write-ahead log: 32 KiB blocks of records (crc32c, length, type FULL/FIRST/MIDDLE/LAST) that
                               concatenate to WriteBatches (sequence, count, then [tag, key, value] entries)
  nnnnnn.ldb / .sst            sorted tables: data blocks of prefix-compressed entries with restart points, optional Snappy
                               compression per block (goleveldb's default), an index block, a 48-byte footer with the magic number
Internal keys are user_key || uint64_le(sequence << 8 | type); the newest sequence of a user key wins, type 0 is a deletion.
Checksums are not verified (the audit re-verifies every stored packet cryptographically anyway).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterator, List, Optional, Tuple

BLOCK = 32768
TABLE_MAGIC = 0xDB4775248B80FB57
T_DELETE, T_VALUE = 0, 1


class LevelDBFormatError(Exception):
    pass


# ------------------------------------------------------------------------------------------------------------------
# varints and Snappy
# ------------------------------------------------------------------------------------------------------------------
def get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        if pos >= len(buf):
            raise LevelDBFormatError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if b < 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise LevelDBFormatError("varint too long")


def snappy_decompress(src: bytes) -> bytes:
    """Snappy raw format (format_description.txt): varint uncompressed length, then literal / copy elements."""
    n, pos = get_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                    # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], "little")
                pos += nb
            ln += 1
            if pos + ln > len(src):
                raise LevelDBFormatError("snappy literal beyond input")
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:                                    # copy, 1-byte offset
            ln = 4 + ((tag >> 2) & 7)
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:                                  # copy, 2-byte offset
            ln = 1 + (tag >> 2)
            off = src[pos] | (src[pos + 1] << 8)
            pos += 2
        else:                                            # copy, 4-byte offset
            ln = 1 + (tag >> 2)
            off = int.from_bytes(src[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise LevelDBFormatError("snappy copy offset out of range")
        for _ in range(ln):                              # byte-wise: copies may overlap their own output (run-length)
            out.append(out[-off])
    if len(out) != n:
        raise LevelDBFormatError("snappy length mismatch: %d != %d" % (len(out), n))
    return bytes(out)


# ------------------------------------------------------------------------------------------------------------------
# log format (write-ahead log and MANIFEST)
# ------------------------------------------------------------------------------------------------------------------
def log_records(data: bytes) -> Iterator[bytes]:
    """Logical records of a log-format file; a torn tail (crash during the last write) ends the iteration silently."""
    pos, cur = 0, None
    while pos + 7 <= len(data):
        left = BLOCK - (pos % BLOCK)
        if left < 7:                                     # block trailer: zero padding
            pos += left
            continue
        ln, typ = struct.unpack_from("<HB", data, pos + 4)
        if typ == 0 and ln == 0:                         # pre-allocated / zeroed region
            pos += left
            continue
        body = data[pos + 7:pos + 7 + ln]
        if len(body) < ln or 7 + ln > left:
            return
        pos += 7 + ln
        if typ == 1:
            cur = None
            yield body
        elif typ == 2:
            cur = bytearray(body)
        elif typ == 3 and cur is not None:
            cur += body
        elif typ == 4 and cur is not None:
            cur += body
            yield bytes(cur)
            cur = None
        else:
            cur = None                                   # fragment without a beginning: skip


def batch_entries(rec: bytes) -> Iterator[Tuple[int, int, bytes, Optional[bytes]]]:
    """(sequence, type, key, value) of one WriteBatch record."""
    if len(rec) < 12:
        raise LevelDBFormatError("short write batch")
    seq, count = struct.unpack_from("<QI", rec, 0)
    pos = 12
    for i in range(count):
        typ = rec[pos]
        pos += 1
        kl, pos = get_varint(rec, pos)
        key = rec[pos:pos + kl]
        pos += kl
        val = None
        if typ == T_VALUE:
            vl, pos = get_varint(rec, pos)
            val = rec[pos:pos + vl]
            pos += vl
        elif typ != T_DELETE:
            raise LevelDBFormatError("unknown batch entry type %d" % typ)
        yield seq + i, typ, key, val


def manifest_state(data: bytes) -> Tuple[List[int], int]:
    """(live table file numbers, log number) after applying every VersionEdit of a MANIFEST."""
    live: Dict[int, int] = {}
    log_number = 0
    for rec in log_records(data):
        pos = 0
        while pos < len(rec):
            tag, pos = get_varint(rec, pos)
            if tag == 1:                                 # comparator name
                ln, pos = get_varint(rec, pos)
                pos += ln
            elif tag in (2, 3, 4, 9):                    # log number, next file number, last sequence, prev log number
                v, pos = get_varint(rec, pos)
                if tag == 2:
                    log_number = v
            elif tag == 5:                               # compact pointer: level, internal key
                _, pos = get_varint(rec, pos)
                ln, pos = get_varint(rec, pos)
                pos += ln
            elif tag == 6:                               # deleted file: level, number
                _, pos = get_varint(rec, pos)
                num, pos = get_varint(rec, pos)
                live.pop(num, None)
            elif tag == 7:                               # new file: level, number, size, smallest, largest
                level, pos = get_varint(rec, pos)
                num, pos = get_varint(rec, pos)
                _, pos = get_varint(rec, pos)
                for _ in range(2):
                    ln, pos = get_varint(rec, pos)
                    pos += ln
                live[num] = level
            else:
                raise LevelDBFormatError("unknown VersionEdit tag %d" % tag)
    return sorted(live), log_number


# ------------------------------------------------------------------------------------------------------------------
# table format
# ------------------------------------------------------------------------------------------------------------------
def _read_block(data: bytes, off: int, size: int) -> bytes:
    if off + size + 5 > len(data):
        raise LevelDBFormatError("block beyond end of table")
    raw, ctype = data[off:off + size], data[off + size]
    if ctype == 0:
        return raw
    if ctype == 1:
        return snappy_decompress(raw)
    raise LevelDBFormatError("unknown block compression %d" % ctype)


def block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise LevelDBFormatError("short block")
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    if end < 0:
        raise LevelDBFormatError("bad restart array")
    pos, key = 0, b""
    while pos < end:
        shared, pos = get_varint(block, pos)
        non_shared, pos = get_varint(block, pos)
        vlen, pos = get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise LevelDBFormatError("bad block entry")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def table_entries(data: bytes) -> Iterator[Tuple[int, int, bytes, Optional[bytes]]]:
    """(sequence, type, user_key, value) of every entry of a sorted table."""
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise LevelDBFormatError("not a LevelDB table (bad magic)")
    footer = data[len(data) - 48:]
    _, p = get_varint(footer, 0)          # metaindex handle
    _, p = get_varint(footer, p)
    ioff, p = get_varint(footer, p)       # index handle
    isize, p = get_varint(footer, p)
    for _, handle in block_entries(_read_block(data, ioff, isize)):
        boff, q = get_varint(handle, 0)
        bsize, q = get_varint(handle, q)
        for ikey, val in block_entries(_read_block(data, boff, bsize)):
            if len(ikey) < 8:
                raise LevelDBFormatError("internal key shorter than its trailer")
            trailer = struct.unpack_from("<Q", ikey, len(ikey) - 8)[0]
            typ = trailer & 0xFF
            yield trailer >> 8, typ, ikey[:-8], (val if typ == T_VALUE else None)


# ------------------------------------------------------------------------------------------------------------------
# database
# ------------------------------------------------------------------------------------------------------------------
def read_db(path: str) -> Dict[bytes, bytes]:
    """Current contents of the database at `path`: user key -> value (newest sequence wins, deletions dropped)."""
    try:
        with open(os.path.join(path, "CURRENT")) as f:
            manifest = f.read().strip()
        with open(os.path.join(path, manifest), "rb") as f:
            tables, log_number = manifest_state(f.read())
    except OSError as e:
        raise LevelDBFormatError("no readable CURRENT / MANIFEST in %s: %s" % (path, e))
    best: Dict[bytes, Tuple[int, int, Optional[bytes]]] = {}

    def offer(seq, typ, key, val):
        cur = best.get(key)
        if cur is None or seq > cur[0]:
            best[key] = (seq, typ, val)
    for num in tables:
        for ext in (".ldb", ".sst"):
            p = os.path.join(path, "%06d%s" % (num, ext))
            if os.path.exists(p):
                with open(p, "rb") as f:
                    for e in table_entries(f.read()):
                        offer(*e)
                break
        else:
            raise LevelDBFormatError("table %06d named by the MANIFEST is missing" % num)
    logs = sorted(int(n[:-4]) for n in os.listdir(path) if n.endswith(".log") and n[:-4].isdigit())
    for num in logs:
        if num < log_number:
            continue                                     # already compacted into tables
        with open(os.path.join(path, "%06d.log" % num), "rb") as f:
            for rec in log_records(f.read()):
                for e in batch_entries(rec):
                    offer(*e)
    return {k: v for k, (_, typ, v) in best.items() if typ == T_VALUE and v is not None}


def bftkv_records(path: str) -> List[Tuple[bytes, int, bytes]]:
    """(variable, t, stored packet) of every entry of a bftkv storage/leveldb database, in key order
    (storage/leveldb/leveldb.go:47-53: key = variable || uint64_be(t))."""
    out = []
    for k, v in sorted(read_db(path).items()):
        if len(k) < 8:
            continue
        out.append((k[:-8], struct.unpack(">Q", k[-8:])[0], v))
    return out
