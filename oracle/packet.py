"""Oracle restatement of bftkv's wire packet ``<x, v, t, sig, ss, auth>``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference packet/packet.go:
  SignaturePacket            packet.go:25-31
  Serialize                  packet.go:35-60
  Parse                      packet.go:62-115
  WriteChunk / ReadChunk     packet.go:117-140
  seek2tbs / TBS / TBSS      packet.go:142-190
  writeSignature / readSignature  packet.go:192-235
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Optional, Tuple

SignatureTypeNil = 0  # packet.go:14
SignatureTypePGP = 1  # packet.go:15


class PacketError(Exception):
    """Any non-EOF read failure (io.ErrUnexpectedEOF in the reference)."""


class _EOF(Exception):
    """io.EOF: nothing at all could be read."""


@dataclass
class SignaturePacket:  # packet.go:25-31
    Type: int = 0
    Version: int = 0
    Completed: bool = False
    Data: Optional[bytes] = None
    Cert: Optional[bytes] = None


class _Reader:
    def __init__(self, b: bytes):
        self.b = b
        self.pos = 0

    def read_exact(self, n: int) -> bytes:
        """binary.Read / io.ReadFull semantics: io.EOF if zero bytes are available, else
        io.ErrUnexpectedEOF when short."""
        avail = len(self.b) - self.pos
        if n == 0:
            return b""
        if avail <= 0:
            raise _EOF()
        if avail < n:
            self.pos = len(self.b)
            raise PacketError("unexpected EOF")
        out = self.b[self.pos:self.pos + n]
        self.pos += n
        return out


def write_chunk(chunk: Optional[bytes]) -> bytes:  # packet.go:117-124
    chunk = chunk or b""
    return struct.pack(">Q", len(chunk)) + chunk


def _read_chunk(r: _Reader) -> Optional[bytes]:  # packet.go:126-140
    (l,) = struct.unpack(">Q", r.read_exact(8))
    if l == 0:
        return None
    # make([]byte, l) with an absurd l would panic in Go; corpora never do that (fenced).
    # io.ReadFull: io.EOF if nothing could be read, io.ErrUnexpectedEOF if short -- the same
    # distinction read_exact makes, and Parse maps io.EOF to "field absent".
    return r.read_exact(l)


def write_signature(sig: Optional[SignaturePacket]) -> bytes:  # packet.go:192-212
    if sig is None:
        sig = SignaturePacket()
    return (bytes([sig.Type & 0xFF]) + struct.pack(">I", sig.Version) +
            (b"\x01" if sig.Completed else b"\x00") + write_chunk(sig.Data) + write_chunk(sig.Cert))


def _read_signature(r: _Reader) -> Optional[SignaturePacket]:  # packet.go:214-235
    sig = SignaturePacket()
    sig.Type = r.read_exact(1)[0]
    try:
        (sig.Version,) = struct.unpack(">I", r.read_exact(4))
        sig.Completed = r.read_exact(1)[0] != 0
        sig.Data = _read_chunk(r)
        sig.Cert = _read_chunk(r)
    except _EOF:
        # a partially present signature: binary.Read returns io.EOF only when zero bytes were
        # read for that field; the reference then propagates io.EOF which Parse treats as "absent".
        raise
    if sig.Type == SignatureTypeNil:
        return None
    return sig


def serialize(*args) -> bytes:  # packet.go:35-60
    out = bytearray()
    for i, arg in enumerate(args):
        if i in (0, 1, 5):
            out += write_chunk(arg)
        elif i == 2:
            out += struct.pack(">Q", arg)
        elif i in (3, 4):
            out += write_signature(arg)
    return bytes(out)


def parse(pkt: bytes):  # packet.go:62-115
    """Returns (variable, value, t, sig, ss, auth); trailing fields that are absent come back
    None/0 exactly as the reference maps io.EOF to nil."""
    r = _Reader(pkt)
    variable = value = sig = ss = auth = None
    t = 0
    try:
        variable = _read_chunk(r)
    except _EOF:
        raise PacketError("EOF")  # first field: the error is returned as-is (packet.go:66-68)
    try:
        value = _read_chunk(r)
        (t,) = struct.unpack(">Q", r.read_exact(8))
        sig = _read_signature(r)
        ss = _read_signature(r)
        auth = _read_chunk(r)
    except _EOF:
        pass
    return variable, value, t, sig, ss, auth


def _seek2tbs(pkt: bytes) -> int:
    """seek2tbs, packet.go:142-154, with what ignoring its errors means: a binary.Read that hits the end stores nothing (``l``
    keeps its previous value) but consumes the bytes that were there (io.ReadFull); Seek moves by that stale ``l``, may park the
    reader past the end, and refuses only a negative target (position unchanged).  Returns the final position."""
    n = len(pkt)
    pos = 0
    l = 0

    def read8(pos, l, keep=True):
        if pos >= n:
            return pos, l                   # io.EOF
        if pos + 8 > n:
            return n, l                     # io.ErrUnexpectedEOF: the tail is consumed
        v = struct.unpack(">q", pkt[pos:pos + 8])[0]
        return pos + 8, (v if keep else l)

    def seek(pos, l):
        a = (pos + l + (1 << 63)) % (1 << 64) - (1 << 63)      # int64 wrap-around
        return a if a >= 0 else pos

    pos, l = read8(pos, l)
    pos = seek(pos, l)
    pos, l = read8(pos, l)
    pos = seek(pos, l)
    pos, _ = read8(pos, l, keep=False)
    return pos


def tbs(pkt: bytes) -> bytes:  # packet.go:156-168
    off = _seek2tbs(pkt)
    if off > len(pkt) or off < 0:
        raise PacketError("unexpected EOF")
    return pkt[:off]


def tbss(pkt: bytes) -> bytes:  # packet.go:170-190
    off = _seek2tbs(pkt)
    if off > len(pkt) or off < 0:
        raise PacketError("EOF")
    r = _Reader(pkt)
    r.pos = off
    try:
        _read_signature(r)
    except _EOF:
        raise PacketError("EOF")
    return pkt[:r.pos]


def parse_signature(pkt: bytes) -> Optional[SignaturePacket]:  # packet.go:237-240
    try:
        return _read_signature(_Reader(pkt))
    except _EOF:
        raise PacketError("EOF")


def serialize_signature(sig: Optional[SignaturePacket]) -> bytes:  # packet.go:242-248
    return write_signature(sig)
