"""Oracle restatement of the OpenPGP arithmetic the bftkv hot path reaches through
golang.org/x/crypto/openpgp (pinned v0.0.0-20191227163750-53104e6ec876, go.mod:8; source NOT in
/root/reference -- PARITY UNPINNED, see oracle/__init__.py) and the Go 1.12/1.13 standard library
(crypto/rsa, crypto/dsa, math/big, crypto/sha*).

TEST INFRASTRUCTURE ONLY.  Reference call sites this file answers for:
  openpgp.CheckDetachedSignature      crypto/pgp/crypto_pgp.go:324, :338, :490
  packet.NewReader / Reader.Next      crypto/pgp/crypto_pgp.go:375-377
  openpgp.ReadEntity / ReadKeyRing    crypto/pgp/crypto_pgp.go:242, :252   (key material + flags only)
Published behaviour restated (RFC 4880):
  B.1 packet header  B.2 Signature.parse  B.3 CheckDetachedSignature
  B.4 RSA PKCS#1 v1.5 verify  B.5 DSA verify
Pinned against GnuPG 2.2.27 / OpenSSL 3 through tests/golden/gpg_vectors.json.
"""
from __future__ import annotations

import hashlib
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

# ---- status codes of one CheckDetachedSignature-equivalent step (shared with include/bftkv_gpu.h)
ST_OK = 0                # verified; signer entity returned
ST_UNKNOWN_ISSUER = 1    # no sign-capable key with that id in the keyring (packet skipped in-call)
ST_PARSE_ERROR = 2       # structural / unsupported error while parsing the packet (packet consumed)
ST_NOT_SIGNATURE = 3     # a well-formed packet that is not a signature
ST_NO_ISSUER = 4         # v4 signature without issuer subpacket
ST_HASH_UNSUPPORTED = 5  # hashForSignature failed (hash not linked in / sig type unsupported)
ST_HASH_TAG = 6          # "hash tag doesn't match"
ST_ALGO_MISMATCH = 7     # "public key and signature use different algorithms"
ST_BAD_SIG = 8           # RSA/DSA verification failure
ST_KEY_CANNOT_SIGN = 9   # "public key cannot generate signatures"
ST_UNSUPPORTED = 10      # fenced-off input: oversize MPI, partial lengths, v3 ...

PK_RSA = 1
PK_RSA_ENCRYPT_ONLY = 2
PK_RSA_SIGN_ONLY = 3
PK_ELGAMAL = 16
PK_DSA = 17
PK_ECDH = 18
PK_ECDSA = 19

# s2k.HashIdToHash table; "available" = linked into a bftkv binary through x/crypto/openpgp's
# imports (crypto/md5, sha1, sha256, sha512, x/crypto/ripemd160).
HASH_BY_ID = {1: "md5", 2: "sha1", 3: "ripemd160", 8: "sha256", 9: "sha384", 10: "sha512", 11: "sha224"}

# DigestInfo prefixes, Go crypto/rsa hashPrefixes (the reference carries a copy at
# crypto/threshold/rsa/rsa.go:345-354).
HASH_PREFIXES = {
    "md5": bytes.fromhex("3020300c06082a864886f70d020505000410"),
    "sha1": bytes.fromhex("3021300906052b0e03021a05000414"),
    "sha224": bytes.fromhex("302d300d06096086480165030402040500041c"),
    "sha256": bytes.fromhex("3031300d060960864801650304020105000420"),
    "sha384": bytes.fromhex("3041300d060960864801650304020205000430"),
    "sha512": bytes.fromhex("3051300d060960864801650304020305000440"),
    # Go's table (crypto/rsa/pkcs1v15.go; the reference carries a copy at crypto/threshold/rsa/rsa.go:345-354) names RIPEMD-160
    # by its ISO/IEC 10118-3 identifier, not by the TeleTrusT one gpg writes: an RSA / RIPEMD-160 signature made by gpg does
    # not verify under Go and vice versa (tests/golden/gpg_weak_hash_vectors.json)
    "ripemd160": bytes.fromhex("30203008060628cf060300310414"),
}


class StructuralError(Exception):
    pass


class UnsupportedError(Exception):
    pass


class UnknownPacketType(Exception):
    pass


class _Truncated(Exception):
    """io.ErrUnexpectedEOF / io.EOF in the middle of a packet."""


class _TooDeep(Exception):
    """An embedded-signature subpacket met at the nesting depth where the verifier's bounded parser stops (kernels.hip
    parse_subpackets_t: DEPTH >= 2).  Only raised when the caller asks for that bound (walk_certificate, which must say "no verdict"
    exactly where the mirror does); x/crypto itself parses recursively without one."""


# ------------------------------------------------------------------------------------------------
# B.1 packet framing
# ------------------------------------------------------------------------------------------------
def read_header(buf: bytes, pos: int) -> Tuple[int, int, int]:
    """openpgp/packet.readHeader.  Returns (tag, body_start, body_len).
    body_len == -1: indeterminate length (old format type 3); -2: partial body length (new format)."""
    if pos >= len(buf):
        raise EOFError()
    b0 = buf[pos]
    if b0 & 0x80 == 0:
        raise StructuralError("tag byte does not have MSB set")
    if b0 & 0x40 == 0:  # old format
        tag = (b0 & 0x3F) >> 2
        lt = b0 & 3
        if lt == 3:
            return tag, pos + 1, -1
        nb = 1 << lt
        if pos + 1 + nb > len(buf):
            raise _Truncated()
        ln = int.from_bytes(buf[pos + 1:pos + 1 + nb], "big")
        return tag, pos + 1 + nb, ln
    tag = b0 & 0x3F
    if pos + 1 >= len(buf):
        raise _Truncated()
    b1 = buf[pos + 1]
    if b1 < 192:
        return tag, pos + 2, b1
    if b1 < 224:
        if pos + 2 >= len(buf):
            raise _Truncated()
        return tag, pos + 3, ((b1 - 192) << 8) + buf[pos + 2] + 192
    if b1 == 255:
        if pos + 6 > len(buf):
            raise _Truncated()
        return tag, pos + 6, int.from_bytes(buf[pos + 2:pos + 6], "big")
    return tag, pos + 2, -2


@dataclass
class Signature:
    """What Signature.parse keeps (openpgp/packet/signature.go) -- only the fields the path uses."""
    version: int = 4
    sig_type: int = 0
    pk_algo: int = 0
    hash_id: int = 0
    hash_suffix: bytes = b""
    hash_tag: bytes = b"\0\0"
    issuer: Optional[int] = None
    creation_time: Optional[int] = None
    mpis: List[Tuple[int, bytes]] = field(default_factory=list)  # (bit length as written, bytes)
    flags_valid: bool = False
    flag_certify: bool = False
    flag_sign: bool = False
    is_primary_id: Optional[bool] = None
    revocation_reason: Optional[int] = None
    embedded: Optional["Signature"] = None
    embedded_body: bytes = b""


def _parse_subpackets(sig: Signature, area: bytes, hashed: bool, depth: int = 0, max_depth: Optional[int] = None) -> None:
    """parseSignatureSubpackets / parseSignatureSubpacket (B.2)."""
    p = 0
    while p < len(area):
        b = area[p]
        if b < 192:
            ln, p = b, p + 1
        elif b < 255:
            if p + 2 > len(area):
                raise StructuralError("subpacket truncated")
            ln, p = ((b - 192) << 8) + area[p + 1] + 192, p + 2
        else:
            if p + 5 > len(area):
                raise StructuralError("subpacket truncated")
            ln, p = int.from_bytes(area[p + 1:p + 5], "big"), p + 5
        if ln > len(area) - p:
            raise StructuralError("subpacket truncated")
        if ln == 0:
            raise StructuralError("zero length signature subpacket")
        sub = area[p:p + ln]
        p += ln
        typ = sub[0] & 0x7F
        critical = sub[0] & 0x80 != 0
        body = sub[1:]
        if typ == 2:  # creation time
            if not hashed:
                raise StructuralError("signature creation time in non-hashed area")
            if len(body) != 4:
                raise StructuralError("signature creation time not four bytes")
            sig.creation_time = int.from_bytes(body, "big")
        elif typ == 3:  # signature expiry
            if not hashed:
                continue
            if len(body) != 4:
                raise StructuralError("expiration subpacket with bad length")
        elif typ == 9:  # key lifetime
            if not hashed:
                continue
            if len(body) != 4:
                raise StructuralError("key expiration subpacket with bad length")
        elif typ in (11, 21, 22):  # preferences
            if not hashed:
                continue
        elif typ == 16:  # issuer: accepted from either area
            if len(body) != 8:
                raise StructuralError("issuer subpacket with bad length")
            sig.issuer = int.from_bytes(body, "big")
        elif typ == 25:  # primary user id
            if not hashed:
                continue
            if len(body) != 1:
                raise StructuralError("primary user id subpacket with bad length")
            sig.is_primary_id = body[0] != 0
        elif typ == 27:  # key flags
            if not hashed:
                continue
            if len(body) == 0:
                raise StructuralError("empty key flags subpacket")
            sig.flags_valid = True
            sig.flag_certify = bool(body[0] & 0x01)
            sig.flag_sign = bool(body[0] & 0x02)
        elif typ == 29:  # reason for revocation
            if not hashed:
                continue
            if len(body) == 0:
                raise StructuralError("empty revocation reason subpacket")
            sig.revocation_reason = body[0]
        elif typ == 30:  # features
            if not hashed:
                continue
        elif typ == 32:
            # embedded signature (cross-certification of a signing subkey; gpg puts it in the UNHASHED area): parsed
            # recursively from either area; a second one and any type other than primary-key binding are refused
            if sig.embedded is not None:
                raise StructuralError("Cannot have multiple embedded signatures")
            if max_depth is not None and depth >= max_depth:
                raise _TooDeep()
            sig.embedded = parse_signature_body(body, depth + 1, max_depth)
            sig.embedded_body = bytes(body)
            if sig.embedded.sig_type != 0x19:
                raise StructuralError("cross-signature has unexpected type %d" % sig.embedded.sig_type)
        else:
            if critical:
                raise UnsupportedError("unknown critical signature subpacket type %d" % typ)


def parse_signature_body(body: bytes, depth: int = 0, max_depth: Optional[int] = None) -> Signature:
    """Signature.parse for a version-4 body (B.2).  Raises StructuralError/UnsupportedError (and _TooDeep under max_depth)."""
    if len(body) < 1:
        raise _Truncated()
    if body[0] != 4:
        raise UnsupportedError("signature packet version %d" % body[0])
    if len(body) < 6:
        raise _Truncated()
    sig = Signature(version=4, sig_type=body[1], pk_algo=body[2], hash_id=body[3])
    if sig.pk_algo not in (PK_RSA, PK_RSA_SIGN_ONLY, PK_DSA, PK_ECDSA):
        raise UnsupportedError("public key algorithm %d" % sig.pk_algo)
    if sig.hash_id not in HASH_BY_ID:
        raise UnsupportedError("hash function %d" % sig.hash_id)
    hl = (body[4] << 8) | body[5]
    if 6 + hl > len(body):
        raise _Truncated()
    hashed = body[6:6 + hl]
    l = 6 + hl
    sig.hash_suffix = body[0:l] + bytes([4, 0xFF]) + struct.pack(">I", l)
    _parse_subpackets(sig, hashed, True, depth, max_depth)
    if sig.creation_time is None:
        raise StructuralError("no creation time in signature")
    p = l
    if p + 2 > len(body):
        raise _Truncated()
    ul = (body[p] << 8) | body[p + 1]
    p += 2
    if p + ul > len(body):
        raise _Truncated()
    _parse_subpackets(sig, body[p:p + ul], False, depth, max_depth)
    p += ul
    if p + 2 > len(body):
        raise _Truncated()
    sig.hash_tag = body[p:p + 2]
    p += 2
    n_mpi = 1 if sig.pk_algo in (PK_RSA, PK_RSA_SIGN_ONLY) else 2
    for _ in range(n_mpi):
        if p + 2 > len(body):
            raise _Truncated()
        bits = (body[p] << 8) | body[p + 1]
        nb = (bits + 7) // 8
        p += 2
        if p + nb > len(body):
            raise _Truncated()
        sig.mpis.append((bits, body[p:p + nb]))
        p += nb
    return sig


def parse_signature_v3_body(body: bytes) -> Signature:
    """SignatureV3.parse (RFC 4880 5.2.2; x/crypto openpgp/packet/signature_v3.go).  hash_suffix = type || creation time,
    no trailer (VerifySignatureV3).  gpg 2.2.27 verifies signatures of this construction (tests/golden/gpg_negative_vectors.json)."""
    if len(body) < 1:
        raise _Truncated()
    if body[0] < 2 or body[0] > 3:
        raise UnsupportedError("signature packet version %d" % body[0])
    if len(body) < 19:
        raise _Truncated()
    if body[1] != 5:
        raise UnsupportedError("invalid hashed material length %d" % body[1])
    sig = Signature(version=body[0], sig_type=body[2], pk_algo=body[15], hash_id=body[16])
    sig.creation_time = int.from_bytes(body[3:7], "big")
    sig.issuer = int.from_bytes(body[7:15], "big")
    if sig.pk_algo not in (PK_RSA, PK_RSA_SIGN_ONLY, PK_DSA):
        raise UnsupportedError("public key algorithm %d" % sig.pk_algo)
    if sig.hash_id not in HASH_BY_ID:
        raise UnsupportedError("hash function %d" % sig.hash_id)
    sig.hash_suffix = body[2:7]
    sig.hash_tag = body[17:19]
    p = 19
    for _ in range(2 if sig.pk_algo == PK_DSA else 1):
        if p + 2 > len(body):
            raise _Truncated()
        bits = (body[p] << 8) | body[p + 1]
        nb = (bits + 7) // 8
        p += 2
        if p + nb > len(body):
            raise _Truncated()
        sig.mpis.append((bits, body[p:p + nb]))
        p += nb
    return sig


# ------------------------------------------------------------------------------------------------
# B.1b  The signature stream as x/crypto READS it: reader objects, not byte ranges
#
# PGPCollectiveSignature.Verify hands ONE bytes.Reader to CheckDetachedSignature again and again (crypto_pgp.go:486-498); what a
# call consumes is whatever the readers stacked on top of it pulled: packet.readHeader's spanReader (definite lengths) or
# partialLengthReader (new-format partial lengths) or the bare stream (old-format indeterminate length), and -- for signature
# packets -- the 4096-byte bufio.Reader that peekVersion wraps around them.  A body that parses is NOT drained afterwards; a body
# that fails is (consumeAll).  Consequences the byte-range view above cannot express, all restated literally here:
#   * a signature whose declared length runs past the end of the stream still verifies when the signature itself is complete;
#   * partial-length and indeterminate-length signature packets parse like any other;
#   * after a SUCCESSFUL parse the shared reader stands where bufio's last fetch ended, which can be short of the packet's end
#     (bodies over 4096 bytes, unread chunks): the next call then parses packets out of the middle of this one.
# Go's Read contract is kept as (data, err) pairs: err is None, "EOF" (io.EOF) or "UEOF" (io.ErrUnexpectedEOF).
# (golang.org/x/crypto v0.0.0-20191227163750-53104e6ec876 openpgp/packet/packet.go, reader.go; Go's bufio and io -- sources absent
# here: restated from their documented behaviour, pinned by shim/tools/genvectors when someone runs it.)
# ------------------------------------------------------------------------------------------------
class _ByteStream:
    """bytes.Reader: Read returns what is there, (0, io.EOF) once nothing is."""
    def __init__(self, buf: bytes, pos: int = 0):
        self.buf, self.pos = buf, pos

    def read(self, n: int):
        if self.pos >= len(self.buf):
            return b"", "EOF"
        d = self.buf[self.pos:self.pos + n]
        self.pos += len(d)
        return d, None


def _io_read_full(r, n: int):
    """io.ReadFull: no Read call at all for an empty buffer; io.EOF only when nothing was read."""
    out = b""
    err = None
    while len(out) < n and err is None:
        d, err = r.read(n - len(out))
        out += d
    if len(out) >= n:
        return out, None
    if out and err == "EOF":
        err = "UEOF"
    return out, err


def _read_full(r, n: int):
    """packet.readFull: io.ReadFull with io.EOF turned into io.ErrUnexpectedEOF."""
    d, err = _io_read_full(r, n)
    return d, ("UEOF" if err == "EOF" else err)


class _SpanReader:
    """packet.spanReader: at most n bytes of r; the stream ending early is io.ErrUnexpectedEOF."""
    def __init__(self, r, n: int):
        self.r, self.n = r, n

    def read(self, k: int):
        if self.n <= 0:
            return b"", "EOF"
        d, err = self.r.read(min(k, self.n))
        self.n -= len(d)
        if self.n > 0 and err == "EOF":
            err = "UEOF"
        return d, err


def _read_length(r):
    """packet.readLength (RFC 4880 4.2.2): (length, is_partial, err)."""
    b, err = _read_full(r, 1)
    if err:
        return 0, False, err
    b0 = b[0]
    if b0 < 192:
        return b0, False, None
    if b0 < 224:
        b, err = _read_full(r, 1)
        if err:
            return 0, False, err
        return ((b0 - 192) << 8) + b[0] + 192, False, None
    if b0 < 255:
        return 1 << (b0 & 0x1F), True, None
    b, err = _read_full(r, 4)
    if err:
        return 0, False, err
    return int.from_bytes(b, "big"), False, None


class _PartialLengthReader:
    """packet.partialLengthReader: chunk after chunk, each announced by its own length header."""
    def __init__(self, r, remaining: int):
        self.r, self.remaining, self.is_partial = r, remaining, True
        self.headers = 0          # length headers read after the first one (bookkeeping for the tests, not in the reference)
        self.delivered = 0        # body bytes handed out

    def read(self, k: int):
        while self.remaining == 0:
            if not self.is_partial:
                return b"", "EOF"
            self.remaining, self.is_partial, err = _read_length(self.r)
            if err:
                return b"", err
            self.headers += 1
        want = min(k, self.remaining)
        d, err = self.r.read(want)
        self.remaining -= len(d)
        self.delivered += len(d)
        if len(d) < want and err == "EOF":
            err = "UEOF"
        return d, err


class _Bufio:
    """bufio.Reader (default size 4096) as far as Peek(1) and Read go."""
    SIZE = 4096

    def __init__(self, rd):
        self.rd, self.buf, self.err = rd, b"", None

    def _read_err(self):
        e, self.err = self.err, None
        return e

    def peek1(self):
        while not self.buf and self.err is None:
            for _ in range(100):                      # fill(): one Read, repeated only while it returns (0, nil)
                d, err = self.rd.read(self.SIZE)
                self.buf += d
                if err is not None:
                    self.err = err
                    break
                if d:
                    break
            else:
                self.err = "NOPROGRESS"
        if self.buf:
            return self.buf[:1], None
        return b"", self._read_err()

    def read(self, n: int):
        if n == 0:
            return b"", (None if self.buf else self._read_err())
        if not self.buf:
            if self.err is not None:
                return b"", self._read_err()
            if n >= self.SIZE:                        # large read, empty buffer: straight into the caller's slice
                d, self.err = self.rd.read(n)
                return d, self._read_err()
            d, self.err = self.rd.read(self.SIZE)     # one read, not fill()
            if not d:
                return b"", self._read_err()
            self.buf = d
        out, self.buf = self.buf[:n], self.buf[n:]
        return out, None


def _consume_all(r) -> None:
    """packet.consumeAll: 1024 bytes at a time until any error."""
    while True:
        _, err = r.read(1024)
        if err is not None:
            return


def _read_header_stream(r):
    """packet.readHeader: (tag, contents reader, err)."""
    b, err = _io_read_full(r, 1)
    if err:
        return 0, None, err                           # io.EOF here is the clean end of the stream
    b0 = b[0]
    if b0 & 0x80 == 0:
        return 0, None, "STRUCT"                      # "tag byte does not have MSB set"
    if b0 & 0x40 == 0:
        tag, lt = (b0 & 0x3F) >> 2, b0 & 3
        if lt == 3:
            return tag, r, None                       # indeterminate length: the stream itself
        nb = 1 << lt
        b, err = _read_full(r, nb)
        if err:
            return 0, None, err
        return tag, _SpanReader(r, int.from_bytes(b, "big")), None
    tag = b0 & 0x3F
    ln, partial, err = _read_length(r)
    if err:
        return 0, None, err
    return tag, (_PartialLengthReader(r, ln) if partial else _SpanReader(r, ln)), None


class _ReadErr(Exception):
    """A reader ran dry in the middle of Signature.parse."""


def _need(r, n: int) -> bytes:
    d, err = _read_full(r, n)
    if err:
        raise _ReadErr(err)
    return d


def parse_signature_stream(r) -> "Signature":
    """Signature.parse reading from r in x/crypto's order: 1, 5, hashed area, 2, unhashed area, 2, then 2 + n per MPI."""
    v = _need(r, 1)
    if v[0] != 4:
        raise UnsupportedError("signature packet version %d" % v[0])
    h = _need(r, 5)
    sig = Signature(version=4, sig_type=h[0], pk_algo=h[1], hash_id=h[2])
    if sig.pk_algo not in (PK_RSA, PK_RSA_SIGN_ONLY, PK_DSA, PK_ECDSA):
        raise UnsupportedError("public key algorithm %d" % sig.pk_algo)
    if sig.hash_id not in HASH_BY_ID:
        raise UnsupportedError("hash function %d" % sig.hash_id)
    hl = (h[3] << 8) | h[4]
    hashed = _need(r, hl)
    l = 6 + hl
    sig.hash_suffix = v + h + hashed + bytes([4, 0xFF]) + struct.pack(">I", l)
    _parse_subpackets(sig, hashed, True)
    if sig.creation_time is None:
        raise StructuralError("no creation time in signature")
    u = _need(r, 2)
    _parse_subpackets(sig, _need(r, (u[0] << 8) | u[1]), False)
    sig.hash_tag = _need(r, 2)
    for _ in range(1 if sig.pk_algo in (PK_RSA, PK_RSA_SIGN_ONLY) else 2):
        b = _need(r, 2)
        bits = (b[0] << 8) | b[1]
        sig.mpis.append((bits, _need(r, (bits + 7) // 8)))
    return sig


def parse_signature_v3_stream(r) -> "Signature":
    """SignatureV3.parse in its own order of reads: 1, 1, 5, 8, 2, 2, MPIs."""
    v = _need(r, 1)
    if v[0] < 2 or v[0] > 3:
        raise UnsupportedError("signature packet version %d" % v[0])
    ln = _need(r, 1)
    if ln[0] != 5:
        raise UnsupportedError("invalid hashed material length %d" % ln[0])
    hm = _need(r, 5)
    iss = _need(r, 8)
    a = _need(r, 2)
    sig = Signature(version=v[0], sig_type=hm[0], pk_algo=a[0], hash_id=a[1])
    sig.creation_time = int.from_bytes(hm[1:5], "big")
    sig.issuer = int.from_bytes(iss, "big")
    if sig.pk_algo not in (PK_RSA, PK_RSA_SIGN_ONLY, PK_DSA):
        raise UnsupportedError("public key algorithm %d" % sig.pk_algo)
    if sig.hash_id not in HASH_BY_ID:
        raise UnsupportedError("hash function %d" % sig.hash_id)
    sig.hash_suffix = hm
    sig.hash_tag = _need(r, 2)
    for _ in range(2 if sig.pk_algo == PK_DSA else 1):
        b = _need(r, 2)
        bits = (b[0] << 8) | b[1]
        sig.mpis.append((bits, _need(r, (bits + 7) // 8)))
    return sig


@dataclass
class StreamPacket:
    """One packet.Read call on the shared stream."""
    kind: str                    # "sig" | "sig_error" | "not_signature" | "unknown" | "error" | "eof"
    tag: int = 0
    sig: Optional["Signature"] = None
    pos: int = 0                 # where the shared reader stands after the call
    body_unread: bool = False    # a parsed signature left bytes of its own packet in the stream (bufio stopped short of its end)
    lazy_parser: bool = False    # a known non-signature type whose x/crypto parser may stop before the end of the body
    beyond_native_bounds: bool = False   # partial lengths past what the VERIFIER follows (kernels.hip CHAIN_MAX_HOPS /
                                         # CHUNKED_SIG_MAX_BODY): it fences the stream from this packet on; the reference does not care


VERIFIER_CHAIN_MAX_HOPS = 1024          # kernels.hip CHAIN_MAX_HOPS
VERIFIER_CHUNKED_SIG_MAX_BODY = 16384   # kernels.hip CHUNKED_SIG_MAX_BODY


def packet_read_stream(buf: bytes, pos: int) -> StreamPacket:
    """packet.Read on the shared reader at pos."""
    pk = _packet_read_stream(buf, pos)
    return pk


def _beyond_bounds(contents, is_sig: bool) -> bool:
    if not isinstance(contents, _PartialLengthReader):
        return False
    _consume_all(contents)             # (the whole chain: the verifier measures it before it parses anything)
    return contents.headers > VERIFIER_CHAIN_MAX_HOPS or (is_sig and contents.delivered > VERIFIER_CHUNKED_SIG_MAX_BODY)


def _packet_read_stream(buf: bytes, pos: int) -> StreamPacket:
    s = _ByteStream(buf, pos)
    tag, contents, err = _read_header_stream(s)
    if err == "EOF":
        return StreamPacket("eof", pos=s.pos)
    if err:
        return StreamPacket("error", pos=s.pos)
    if tag != 2:
        # every other type: the known ones are parsed by code this restatement does not follow (where their parser stops is the
        # fence of position_is_type_dependent); here their whole body is taken, as consumeAll does for unknown types and errors
        _consume_all(contents)
        bb = _beyond_bounds(contents, False)
        if tag in _KNOWN_TAGS:
            return StreamPacket("not_signature", tag=tag, pos=s.pos, lazy_parser=tag not in _READS_TO_END, beyond_native_bounds=bb)
        return StreamPacket("unknown", tag=tag, pos=s.pos, beyond_native_bounds=bb)
    bufr = _Bufio(contents)
    ver, err = bufr.peek1()
    if err:
        return StreamPacket("sig_error", tag=2, pos=s.pos)        # io.EOF (empty body) ends Reader.Next the same way an error does
    try:
        sig = parse_signature_v3_stream(bufr) if ver[0] < 4 else parse_signature_stream(bufr)
    except (StructuralError, UnsupportedError, _ReadErr):
        _consume_all(bufr)
        return StreamPacket("sig_error", tag=2, pos=s.pos, beyond_native_bounds=_beyond_bounds(contents, True))
    after = s.pos
    _consume_all(contents)                                        # (not done by the reference: only to see whether anything was left)
    return StreamPacket("sig", tag=2, sig=sig, pos=after, body_unread=s.pos != after, beyond_native_bounds=_beyond_bounds(contents, True))


@dataclass
class RawPacket:
    tag: int
    body: bytes
    end: int  # stream position after the packet


def next_packet(buf: bytes, pos: int) -> RawPacket:
    """One packet.Read step on an unbuffered stream.  Raises EOFError at a clean end,
    StructuralError for a bad tag byte (1 byte consumed: ``.consumed``), _Truncated mid-packet."""
    try:
        tag, start, ln = read_header(buf, pos)
    except StructuralError as e:
        e.consumed = pos + 1
        raise
    if ln < 0:
        # indeterminate / partial body lengths: legal OpenPGP, never produced by the path's
        # writers (DetachSign emits definite lengths).  FENCED: the whole rest of the
        # stream is treated as unsupported.
        u = UnsupportedError("indeterminate/partial packet length")
        u.consumed = len(buf)
        raise u
    if start + ln > len(buf):
        raise _Truncated()
    return RawPacket(tag, buf[start:start + ln], start + ln)


# packet types whose x/crypto parser reads its body to the end (ioutil.ReadAll as its last step): user id, user attribute,
# secret key / secret subkey.  Every other known non-signature type can return with part of its body unread.
_READS_TO_END = {5, 7, 13, 17}


def position_is_type_dependent(buf: bytes, stop_at_error: bool = False) -> bool:
    """Does this signature stream contain a packet after which the verifier does not follow the reference's reader?

    Two shapes (DESIGN.md "fenced inputs"; kernels.hip k_walk / parse_one raise the item's fence flag on them):
      * a known non-signature packet whose x/crypto parser may return before the end of its body (literal data, compressed,
        encrypted, one-pass, key packets ...: everything but user id / user attribute / private key, which end in ReadAll) --
        this restatement does not model those parsers, it takes the whole body;
      * a signature packet that PARSES while bufio's last fetch stopped short of the end of its packet (packet_read_stream's
        body_unread: bodies beyond 4096 bytes, unread partial-length chunks, an indeterminate-length packet with more than the
        signature behind it): the next CheckDetachedSignature call parses packets out of the middle of this one.  This
        restatement DOES follow that exactly; the kernels do not.
    stop_at_error: PGPSignature.Signers' walk, which the first error of Reader.Next ends."""
    return fence_reason(buf, stop_at_error) is not None


def fence_reason(buf: bytes, stop_at_error: bool = False) -> Optional[str]:
    """The first fenced shape in the stream: "lazy" (a parser this restatement does not model: nothing to compare against),
    "unread" (followed exactly here, fenced by the verifier), "bounds" (a partial-length chain or chunked signature body past the
    verifier's native bounds: it reports ST_UNSUPPORTED for that packet and fences), or None."""
    pos = 0
    while True:
        pkt = packet_read_stream(buf, pos)
        pos = pkt.pos
        if pkt.kind == "eof":
            return None
        if pkt.beyond_native_bounds:
            return "bounds"
        if pkt.kind in ("error", "sig_error"):
            if stop_at_error:
                return None
            continue
        if pkt.kind == "not_signature" and pkt.lazy_parser:
            return "lazy"
        if pkt.kind == "sig" and pkt.body_unread:
            return "unread"


# ------------------------------------------------------------------------------------------------
# Keys and keyrings
# ------------------------------------------------------------------------------------------------
@dataclass
class PublicKey:
    key_id: int
    pk_algo: int
    # RSA
    n: int = 0
    e: int = 0
    # DSA
    p: int = 0
    q: int = 0
    g: int = 0
    y: int = 0
    fingerprint: bytes = b""
    is_subkey: bool = False

    def can_sign(self) -> bool:  # PublicKey.CanSign
        return self.pk_algo not in (PK_RSA_ENCRYPT_ONLY, PK_ELGAMAL)


@dataclass
class Entity:
    """openpgp.Entity reduced to what KeysByIdUsage and bftkv's node wrapper read."""
    primary: PublicKey
    name: str = ""
    # self-signature facts of the (first / primary) identity
    flags_valid: bool = True
    flag_sign: bool = True
    flag_certify: bool = True
    self_sig_revoked: bool = False
    revoked: bool = False                      # len(Entity.Revocations) > 0
    subkeys: List[Tuple[PublicKey, bool, bool, bool]] = field(default_factory=list)  # (key, flags_valid, flag_sign, revocation reason) of Subkey.Sig
    certifiers: List[int] = field(default_factory=list)   # issuer key ids of 3rd-party certifications
    serialized: bytes = b""

    @property
    def id(self) -> int:  # PGPCertificateInstance.Id, crypto_pgp.go:43-45
        return self.primary.key_id


def keys_by_id_usage_sign(keyring: List[Entity], key_id: int) -> List[Tuple[Entity, PublicKey]]:
    """EntityList.KeysByIdUsage(id, KeyFlagSign) (B.3)."""
    out = []
    for e in keyring:
        cands = []
        if e.primary.key_id == key_id:
            cands.append((e.primary, e.flags_valid, e.flag_sign, e.self_sig_revoked))
        for sk, fv, fs, rr in e.subkeys:
            if sk.key_id == key_id:
                cands.append((sk, fv, fs, rr))            # Subkey.Sig.RevocationReason != nil
        for k, fv, fs, rr in cands:
            if e.revoked or rr:
                continue
            if fv and not fs:
                continue
            out.append((e, k))
    return out


# ------------------------------------------------------------------------------------------------
# B.4 / B.5 public-key verification
# ------------------------------------------------------------------------------------------------
def rsa_verify_pkcs1v15(n: int, e: int, hash_name: str, digest: bytes, sig: bytes) -> bool:
    """Go 1.12/1.13 rsa.VerifyPKCS1v15 after openpgp's padToKeySize: no len(sig)==k check and no
    s<n check (math/big.Exp reduces) in those releases (B.4)."""
    prefix = HASH_PREFIXES[hash_name]
    t_len = len(prefix) + len(digest)
    k = (n.bit_length() + 7) // 8
    if k < t_len + 11:
        return False
    c = int.from_bytes(sig, "big")
    m = pow(c, e, n)
    em = m.to_bytes(k, "big")
    expect = b"\x00\x01" + b"\xff" * (k - t_len - 3) + b"\x00" + prefix + digest
    return em == expect


def dsa_verify(p: int, q: int, g: int, y: int, digest: bytes, r: int, s: int) -> bool:
    """Go crypto/dsa.Verify (B.5); ``digest`` already truncated by openpgp to ceil(bits(q)/8)."""
    if p == 0:  # pub.P.Sign() == 0
        return False
    if r < 1 or r >= q:
        return False
    if s < 1 or s >= q:
        return False
    n = q.bit_length()
    if n & 7 != 0:
        return False
    w = pow(s, -1, q) if _gcd(s, q) == 1 else 0   # ModInverse returns nil on failure -> treated as 0
    if w == 0:
        return False
    z = int.from_bytes(digest[:n // 8] if len(digest) > n // 8 else digest, "big")
    u1 = (z * w) % q
    u2 = (r * w) % q
    v = (pow(g, u1, p) * pow(y, u2, p)) % p % q
    return v == r


def _gcd(a: int, b: int) -> int:
    while b:
        a, b = b, a % b
    return a


# MD5 / RIPEMD-160: hashForSignature returns "hash not available" unless the reference BINARY links the package.  That cannot
# be read off the reference's sources here (x/crypto is not vendored), so it is a setting: None = unknown (the verifier fences
# such signatures and this oracle refuses them), True = linked, False = not linked.  tests set it per case.
HASH_POLICY = {"md5": None, "ripemd160": None}


class _Ripemd160:
    """RIPEMD-160 (hashlib of this image has none): the published algorithm, for the oracle only."""
    _R1 = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 7, 4, 13, 1, 10, 6, 15, 3, 12, 0, 9, 5, 2, 14, 11, 8,
           3, 10, 14, 4, 9, 15, 8, 1, 2, 7, 0, 6, 13, 11, 5, 12, 1, 9, 11, 10, 0, 8, 12, 4, 13, 3, 7, 15, 14, 5, 6, 2,
           4, 0, 5, 9, 7, 12, 2, 10, 14, 1, 3, 8, 11, 6, 15, 13]
    _R2 = [5, 14, 7, 0, 9, 2, 11, 4, 13, 6, 15, 8, 1, 10, 3, 12, 6, 11, 3, 7, 0, 13, 5, 10, 14, 15, 8, 12, 4, 9, 1, 2,
           15, 5, 1, 3, 7, 14, 6, 9, 11, 8, 12, 2, 10, 0, 4, 13, 8, 6, 4, 1, 3, 11, 15, 0, 5, 12, 2, 13, 9, 7, 10, 14,
           12, 15, 10, 4, 1, 5, 8, 7, 6, 2, 13, 14, 0, 3, 9, 11]
    _S1 = [11, 14, 15, 12, 5, 8, 7, 9, 11, 13, 14, 15, 6, 7, 9, 8, 7, 6, 8, 13, 11, 9, 7, 15, 7, 12, 15, 9, 11, 7, 13, 12,
           11, 13, 6, 7, 14, 9, 13, 15, 14, 8, 13, 6, 5, 12, 7, 5, 11, 12, 14, 15, 14, 15, 9, 8, 9, 14, 5, 6, 8, 6, 5, 12,
           9, 15, 5, 11, 6, 8, 13, 12, 5, 12, 13, 14, 11, 8, 5, 6]
    _S2 = [8, 9, 9, 11, 13, 15, 15, 5, 7, 7, 8, 11, 14, 14, 12, 6, 9, 13, 15, 7, 12, 8, 9, 11, 7, 7, 12, 7, 6, 15, 13, 11,
           9, 7, 15, 11, 8, 6, 6, 14, 12, 13, 5, 14, 13, 13, 7, 5, 15, 5, 8, 11, 14, 14, 6, 14, 6, 9, 12, 9, 12, 5, 15, 8,
           8, 5, 12, 9, 12, 5, 14, 6, 8, 13, 6, 5, 15, 13, 11, 11]
    _K1 = [0x00000000, 0x5A827999, 0x6ED9EBA1, 0x8F1BBCDC, 0xA953FD4E]
    _K2 = [0x50A28BE6, 0x5C4DD124, 0x6D703EF3, 0x7A6D76E9, 0x00000000]
    digest_size = 20

    def __init__(self, data: bytes = b""):
        self.h = [0x67452301, 0xEFCDAB89, 0x98BADCFE, 0x10325476, 0xC3D2E1F0]
        self.buf, self.n = b"", 0
        if data:
            self.update(data)

    @staticmethod
    def _f(j, x, y, z):
        if j == 0: return x ^ y ^ z
        if j == 1: return (x & y) | (~x & z)
        if j == 2: return (x | ~y) ^ z
        if j == 3: return (x & z) | (y & ~z)
        return x ^ (y | ~z)

    def _block(self, blk: bytes):
        M = 0xFFFFFFFF
        rol = lambda v, s: ((v << s) | (v >> (32 - s))) & M
        w = struct.unpack("<16I", blk)
        a1, b1, c1, d1, e1 = self.h
        a2, b2, c2, d2, e2 = self.h
        for i in range(80):
            j = i >> 4
            t = (rol((a1 + (self._f(j, b1, c1, d1) & M) + w[self._R1[i]] + self._K1[j]) & M, self._S1[i]) + e1) & M
            a1, e1, d1, c1, b1 = e1, d1, rol(c1, 10), b1, t
            t = (rol((a2 + (self._f(4 - j, b2, c2, d2) & M) + w[self._R2[i]] + self._K2[j]) & M, self._S2[i]) + e2) & M
            a2, e2, d2, c2, b2 = e2, d2, rol(c2, 10), b2, t
        h = self.h
        t = (h[1] + c1 + d2) & M
        self.h = [t, (h[2] + d1 + e2) & M, (h[3] + e1 + a2) & M, (h[4] + a1 + b2) & M, (h[0] + b1 + c2) & M]
        self.h = [self.h[0], self.h[1], self.h[2], self.h[3], self.h[4]]

    def update(self, data: bytes):
        self.buf += data
        self.n += len(data)
        while len(self.buf) >= 64:
            self._block(self.buf[:64])
            self.buf = self.buf[64:]

    def copy(self):
        c = _Ripemd160()
        c.h, c.buf, c.n = list(self.h), self.buf, self.n
        return c

    def digest(self) -> bytes:
        c = self.copy()
        pad = b"\x80" + b"\x00" * ((55 - c.n) % 64) + struct.pack("<Q", c.n * 8)
        c.update(pad)
        return struct.pack("<5I", *c.h)


def new_hash(name: str):
    """hashlib.new, with RIPEMD-160 supplied where OpenSSL 3's default provider has dropped it."""
    if name == "ripemd160":
        try:
            return hashlib.new(name)
        except ValueError:
            return _Ripemd160()
    return hashlib.new(name)


class CanonicalTextHash:
    """openpgp.NewCanonicalTextHash (x/crypto openpgp/canonical_text.go): what the SIGNED DATA of a text-mode (0x01)
    signature passes through on its way into the hash -- a '\n' that does not follow a '\r' becomes "\r\n"; the byte after
    a '\r' passes unchanged whatever it is (state 1 only resets), so "\r\r\n" comes out as "\r\r\r\n".  The state
    carries across writes.  VerifySignature writes the hash suffix into the RAW hash behind it: ``raw_update``."""

    def __init__(self, h, state=0):
        self.h, self.s = h, state

    def update(self, data: bytes):
        out = bytearray()
        for c in data:
            if self.s == 0:
                if c == 0x0D:
                    self.s = 1
                    out.append(c)
                elif c == 0x0A:
                    out += b"\r\n"
                else:
                    out.append(c)
            else:
                self.s = 0
                out.append(c)
        self.h.update(bytes(out))

    def raw_update(self, data: bytes):
        self.h.update(data)

    def copy(self):
        return CanonicalTextHash(self.h.copy(), self.s)

    def digest(self):
        return self.h.digest()


class _BinaryHash:
    """hashForSignature for binary (0x00) signatures: signed data and hash suffix go into the same hash."""

    def __init__(self, h):
        self.h = h

    def update(self, data: bytes):
        self.h.update(data)

    raw_update = update

    def copy(self):
        return _BinaryHash(self.h.copy())

    def digest(self):
        return self.h.digest()


def hash_for_signature(hash_id: int, sig_type: int):
    """hashForSignature: binary (0x00) hashes raw bytes; text (0x01) canonicalises the line endings of the signed data
    (CanonicalTextHash); other signature types are unsupported."""
    name = HASH_BY_ID.get(hash_id)
    if name is None:
        return None
    if name in HASH_POLICY and HASH_POLICY[name] is not True:
        # whether MD5 / RIPEMD-160 are linked into a bftkv binary cannot be established without the x/crypto source
        # (HASH_POLICY above): unknown and "not linked" both refuse here; the verifier FENCES the unknown case
        return None
    if sig_type not in (0x00, 0x01):
        return None
    try:
        h = new_hash(name)
    except ValueError:
        return None
    return CanonicalTextHash(h) if sig_type == 0x01 else _BinaryHash(h)


def verify_signature(key: PublicKey, hash_id: int, digest: bytes, sig: Signature) -> int:
    """PublicKey.VerifySignature after the hash has been finalised (B.4)."""
    if not key.can_sign():
        return ST_KEY_CANNOT_SIGN
    if digest[0] != sig.hash_tag[0] or digest[1] != sig.hash_tag[1]:
        return ST_HASH_TAG
    if key.pk_algo != sig.pk_algo:
        return ST_ALGO_MISMATCH
    if key.pk_algo in (PK_RSA, PK_RSA_SIGN_ONLY):
        ok = rsa_verify_pkcs1v15(key.n, key.e, HASH_BY_ID[hash_id], digest, sig.mpis[0][1])
        return ST_OK if ok else ST_BAD_SIG
    if key.pk_algo == PK_DSA:
        sub = (key.q.bit_length() + 7) // 8
        hb = digest[:sub] if len(digest) > sub else digest
        ok = dsa_verify(key.p, key.q, key.g, key.y, hb,
                        int.from_bytes(sig.mpis[0][1], "big"), int.from_bytes(sig.mpis[1][1], "big"))
        return ST_OK if ok else ST_BAD_SIG
    return ST_UNSUPPORTED  # ECDSA: out of scope


@dataclass
class StepResult:
    status: int
    signer: Optional[Entity]
    pos: int                # stream position after the call
    statuses: List[int]     # one status per packet consumed by this call (diagnostics / GPU parity)


def check_detached_signature(keyring: List[Entity], signed: bytes, sigdata: bytes, pos: int) -> StepResult:
    """openpgp.CheckDetachedSignature(keyring, signed, signature) where ``signature`` is the shared
    bytes.Reader positioned at ``pos`` (crypto_pgp.go:321-329, 486-498).  B.3."""
    per_packet: List[int] = []
    while True:
        pkt = packet_read_stream(sigdata, pos)
        pos = pkt.pos                                     # wherever the readers of that packet.Read call stopped pulling
        if pkt.kind == "eof":
            return StepResult(ST_UNKNOWN_ISSUER, None, pos, per_packet)  # io.EOF => ErrUnknownIssuer
        if pkt.kind == "unknown":
            continue  # Reader.Next silently skips unknown packet types (their body drained by consumeAll)
        if pkt.kind == "not_signature":
            per_packet.append(ST_NOT_SIGNATURE)
            return StepResult(ST_NOT_SIGNATURE, None, pos, per_packet)
        if pkt.kind != "sig":                             # framing error, or a signature body that does not parse (drained)
            per_packet.append(ST_PARSE_ERROR)
            return StepResult(ST_PARSE_ERROR, None, pos, per_packet)
        sig = pkt.sig
        if sig.issuer is None:
            per_packet.append(ST_NO_ISSUER)
            return StepResult(ST_NO_ISSUER, None, pos, per_packet)
        keys = keys_by_id_usage_sign(keyring, sig.issuer)
        if not keys:
            per_packet.append(ST_UNKNOWN_ISSUER)
            continue
        h = hash_for_signature(sig.hash_id, sig.sig_type)
        if h is None:
            per_packet.append(ST_HASH_UNSUPPORTED)
            return StepResult(ST_HASH_UNSUPPORTED, None, pos, per_packet)
        h.update(signed)
        st = ST_BAD_SIG
        for ent, key in keys:
            # VerifySignature writes HashSuffix into the *shared* hash object each time it is
            # called, so candidate j sees the suffix j+1 times.
            if not key.can_sign():          # checked before the suffix is written
                st = ST_KEY_CANNOT_SIGN
                continue
            h.raw_update(sig.hash_suffix)
            st = verify_signature(key, sig.hash_id, h.copy().digest(), sig)
            if st == ST_OK:
                per_packet.append(ST_OK)
                return StepResult(ST_OK, ent, pos, per_packet)
        per_packet.append(st)
        return StepResult(st, None, pos, per_packet)


# packet tags packet.Read knows how to construct (anything else => UnknownPacketTypeError)
_KNOWN_TAGS = {1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 13, 14, 17, 18}


# ------------------------------------------------------------------------------------------------
# Building blocks used by the corpus generator and the gpg pinning tests
# ------------------------------------------------------------------------------------------------
def mpi(x: int) -> bytes:
    """Canonical MPI (true bit count)."""
    nb = (x.bit_length() + 7) // 8
    return struct.pack(">H", x.bit_length()) + x.to_bytes(nb, "big")


def new_format_header(tag: int, ln: int) -> bytes:
    """serializeHeader: new-format packet header."""
    if ln < 192:
        return bytes([0xC0 | tag, ln])
    if ln < 8384:
        ln -= 192
        return bytes([0xC0 | tag, 192 + (ln >> 8), ln & 0xFF])
    return bytes([0xC0 | tag, 255]) + struct.pack(">I", ln)


def public_key_body(key: PublicKey, creation_time: int) -> bytes:
    body = bytes([4]) + struct.pack(">I", creation_time) + bytes([key.pk_algo])
    if key.pk_algo in (PK_RSA, PK_RSA_SIGN_ONLY, PK_RSA_ENCRYPT_ONLY):
        body += mpi(key.n) + mpi(key.e)
    elif key.pk_algo == PK_DSA:
        body += mpi(key.p) + mpi(key.q) + mpi(key.g) + mpi(key.y)
    else:
        raise ValueError("algo")
    return body


def fingerprint_v4(body: bytes) -> bytes:
    """SHA-1(0x99 || u16(len) || body)."""
    return hashlib.sha1(b"\x99" + struct.pack(">H", len(body)) + body).digest()


def parse_public_key_body(body: bytes, is_subkey: bool = False) -> PublicKey:
    if body[0] != 4:
        raise UnsupportedError("public key version")
    algo = body[5]
    p = 6

    def rd():
        nonlocal p
        bits = (body[p] << 8) | body[p + 1]
        nb = (bits + 7) // 8
        v = int.from_bytes(body[p + 2:p + 2 + nb], "big")
        p += 2 + nb
        return v

    fp = fingerprint_v4(body)
    kid = int.from_bytes(fp[12:20], "big")
    if algo in (PK_RSA, PK_RSA_SIGN_ONLY, PK_RSA_ENCRYPT_ONLY):
        n = rd()
        e = rd()
        return PublicKey(kid, algo, n=n, e=e, fingerprint=fp, is_subkey=is_subkey)
    if algo == PK_DSA:
        pp, q, g, y = rd(), rd(), rd(), rd()
        return PublicKey(kid, algo, p=pp, q=q, g=g, y=y, fingerprint=fp, is_subkey=is_subkey)
    raise UnsupportedError("public key algorithm %d" % algo)


# ------------------------------------------------------------------------------------------------
# B.6  openpgp.ReadEntity, packet by packet (x/crypto openpgp/keys.go @ go.mod:8: ReadEntity, addUserID, addSubkey,
# shouldReplaceSubkeySig; packet/public_key.go: parse, VerifyKeySignature, VerifyUserIdSignature, VerifyRevocationSignature).
#
# PGPCertificate.Parse (crypto/pgp/crypto_pgp.go:236-249) calls ReadEntity on one packet.Reader until the first error; the
# entities before it are the certificate.  What ReadEntity does with each packet that Reader.Next hands it:
#   first packet        must be a public-key packet (tag 6 -- or 14: the type assertion does not look at IsSubkey), else "first
#                       packet was not a public/private key"; a primary key that cannot sign (RSA encrypt-only, ElGamal) is refused
#   user id (13)        addUserID: the v4 signature packets that follow belong to it.  A 0x10 / 0x13 signature whose issuer is the
#                       primary key MUST verify (else the entity is refused); each one that does replaces identity.SelfSignature
#                       and puts the identity into Entity.Identities (keyed by the uid string).  Every other signature in the run
#                       -- 0x11 / 0x12 by the primary key, 0x30, a 0x20 -- goes to identity.Signatures unverified.  The first
#                       packet that is not a v4 signature ends the run.  An identity that never got a self-signature is dropped
#                       with the signatures collected on it (PGPCertificateInstance.Signers walks Entity.Identities only)
#   subkey (14)         addSubkey: the v4 signatures that follow must be 0x18 or 0x28 ("subkey signature with wrong type"), and
#                       each must verify under the primary key over key || subkey (VerifyKeySignature; when the signature's key
#                       flags say "sign", its embedded 0x19 cross-signature must be there and verify under the SUBKEY over the same
#                       bytes).  A 0x28 becomes Subkey.Sig; a 0x18 does when none is there yet or it is newer than a 0x18 that is
#                       (a revocation is never replaced).  No signature at all: "subkey packet not followed by signature"
#   signature (2)       outside such a run: a 0x20 is kept and verified at the end under the primary key over the key alone (a
#                       failure refuses the entity, a success fills Entity.Revocations); anything else is ignored
#   public key (6)      ends the entity (unread: the next ReadEntity starts there)
#   other packets       ignored -- but they do end a signature run
#   errors of Next      refuse the entity (Reader.Next itself skips packet types packet.Read does not know).  A key or signature
#                       packet cut off by the end of the certificate parses when everything its parser asks for is there
#                       (peekVersion's bufio has fetched what there was); a user id (ReadAll) does not
#   at the end          no identity: "entity without any identities"
# The bytes a key contributes to a hash or a fingerprint are its RE-SERIALIZATION (SerializeSignaturePrefix +
# serializeWithoutHeaders: version, creation time, algorithm, the MPIs as read), not the packet body: bytes behind the last MPI
# take no part (key packets are read through peekVersion's bufio, which swallows them with the rest of a body of <= 4096 bytes).
# Shapes outside this restatement are reported as `unknown` (the verifier fences them, the reference decides): version-2/3 keys,
# elliptic-curve keys, secret-key packets, user attributes and the other packet types whose parsers are not modelled, key or
# signature bodies over 4096 bytes, partial / indeterminate lengths, signatures the device cannot look up by issuer (binding / revocation signatures whose issuer
# subpacket does not name the primary key, cross-signatures that do not name the subkey).  (The mirror also gives no verdict on
# a certificate whose signatures are computed over more than 16 MB in all: it hands the device a copy per signature.)
# ------------------------------------------------------------------------------------------------
_KNOWN_TAGS = {1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 13, 14, 17, 18}     # packet.Read's switch; everything else: UnknownPacketTypeError


@dataclass
class CertCheck:
    """One signature ReadEntity verifies: `kind` uid / binding / cross / revocation, the key it is verified with, the bytes it
    is computed over and the signature (raw = its packet, or for a cross-signature the embedded body framed as a packet)."""
    kind: str
    key: "PublicKey"
    signed: bytes
    sig: Signature
    raw: bytes


@dataclass
class EntityWalk:
    start: int
    end: int = 0
    primary: Optional[PublicKey] = None
    error: Optional[str] = None          # ReadEntity refuses the entity whatever its signatures say
    unknown: Optional[str] = None        # a shape this restatement does not follow (see above)
    identities: list = field(default_factory=list)   # dicts: name, self_sig (last one), sigs [(Signature, raw)], uid_framed
    subkeys: list = field(default_factory=list)      # dicts: key, sig (Subkey.Sig)
    revocations: list = field(default_factory=list)
    checks: List[CertCheck] = field(default_factory=list)
    key_hash: bytes = b""
    # what ReadEntity does with each packet Reader.Next() hands it inside the entity, in order: (role, index, chosen) with role in
    # "primary" | "uid" | "self" | "ident_sig" | "subkey" | "subkey_sig" | "revocation" | "ignored" (include/bftkv_gpu.h BFTKV_ROLE_*)
    roles: list = field(default_factory=list)


def _read_mpi_strict(body: bytes, p: int) -> Tuple[int, bytes, int]:
    if p + 2 > len(body):
        raise _Truncated()
    bits = (body[p] << 8) | body[p + 1]
    nb = (bits + 7) // 8
    if p + 2 + nb > len(body):
        raise _Truncated()
    return bits, body[p + 2:p + 2 + nb], p + 2 + nb


def parse_public_key_strict(body: bytes, is_subkey: bool) -> Tuple[Optional[PublicKey], bytes, Optional[str]]:
    """PublicKey.parse of a version-4 body: (key, re-serialization, None), or (None, b"", "unknown: ...") for the shapes left to
    the reference.  Raises UnsupportedError / _Truncated as the reference's parser would fail."""
    if len(body) < 6:
        raise _Truncated()
    if body[0] != 4:
        raise UnsupportedError("public key version")
    algo = body[5]
    if algo in (PK_ECDH, PK_ECDSA):
        return None, b"", "elliptic-curve key"
    n_mpi = {PK_RSA: 2, PK_RSA_ENCRYPT_ONLY: 2, PK_RSA_SIGN_ONLY: 2, PK_DSA: 4, PK_ELGAMAL: 3}.get(algo)
    if n_mpi is None:
        raise UnsupportedError("public key type: %d" % algo)
    p = 6
    vals = []
    for _ in range(n_mpi):
        bits, raw, p = _read_mpi_strict(body, p)
        vals.append(raw)
    if algo in (PK_RSA, PK_RSA_ENCRYPT_ONLY, PK_RSA_SIGN_ONLY) and len(vals[1]) > 3:
        raise UnsupportedError("large public exponent")
    ser = body[:p]
    fp = fingerprint_v4(ser)
    kid = int.from_bytes(fp[12:20], "big")
    iv = [int.from_bytes(v, "big") for v in vals]
    if algo == PK_DSA:
        key = PublicKey(kid, algo, p=iv[0], q=iv[1], g=iv[2], y=iv[3], fingerprint=fp, is_subkey=is_subkey)
    elif algo == PK_ELGAMAL:
        key = PublicKey(kid, algo, p=iv[0], g=iv[1], y=iv[2], fingerprint=fp, is_subkey=is_subkey)
    else:
        key = PublicKey(kid, algo, n=iv[0], e=iv[1], fingerprint=fp, is_subkey=is_subkey)
    return key, b"\x99" + struct.pack(">H", len(ser)) + ser, None


def _frame_sig(body: bytes) -> bytes:
    return new_format_header(2, len(body)) + body


def walk_certificate(blob: bytes) -> List[EntityWalk]:
    """Every entity of a certificate as ReadEntity would go through it (see the table above).  A refused entity does not end the
    walk here (the next one starts at the next primary-key packet, as ReadKeyRing's readToNextPublicKey would): `parse_certificate`
    applies Parse's stop-at-the-first-error."""
    out: List[EntityWalk] = []
    pos = 0
    cur: Optional[EntityWalk] = None
    run = None                      # None | ("uid", identity) | ("sub", subkey)
    leading_junk = False

    def close(at: int):
        nonlocal cur, run
        if cur is None:
            return
        cur.end = at
        # What is MISSING at the end of an entity is only known when every packet of it was followed: a shape left to the
        # reference (a signature body over 4096 bytes, a partial length, an unmodelled packet type ...) may have hidden exactly the
        # self-signature or binding that would be missed here.  No structural refusal after an unknown shape.
        if cur.unknown is None:
            for sk in cur.subkeys:
                if sk["sig"] is None and cur.error is None:
                    cur.error = "subkey packet not followed by signature"
            if cur.error is None and not any(i["self_sig"] is not None for i in cur.identities):
                cur.error = "entity without any identities"
        run = None
        for sig, raw in cur.revocations:
            # VerifyRevocationSignature verifies with the primary key whatever issuer the signature names
            cur.checks.append(CertCheck("revocation", cur.primary, cur.key_hash, sig, raw))
        cur = None

    def fail(msg: str):
        # a refusal is only established while the walk has followed every packet of the entity (see close)
        if cur is not None and cur.error is None and cur.unknown is None:
            cur.error = msg

    def skip_to_next_primary(p: int) -> int:
        """readToNextPublicKey: the position of the next complete tag-6 packet, or the end."""
        while p < len(blob):
            try:
                tag, start, ln = read_header(blob, p)
            except (EOFError, _Truncated, StructuralError):
                return len(blob)
            if ln < 0 or start + ln > len(blob):
                return len(blob)
            if tag == 6:
                return p
            p = start + ln
        return len(blob)

    while pos < len(blob):
        at = pos
        try:
            tag, start, ln = read_header(blob, pos)
        except (EOFError, _Truncated, StructuralError):
            fail("packet header")                 # Reader.Next returns the error; outside an entity Parse just ends
            break
        if ln < 0:
            if cur is not None:
                cur.unknown = cur.unknown or "partial / indeterminate length"
            break
        truncated = start + ln > len(blob)
        body = blob[start:start + ln]
        pos = min(start + ln, len(blob))
        if tag not in _KNOWN_TAGS:
            continue                              # UnknownPacketTypeError: Reader.Next goes on (a body cut short too: consumeAll's own error is dropped)
        if tag in (2, 6, 14) and len(body) == 0 and not truncated:
            break                                 # peekVersion's Peek(1) returns io.EOF: Reader.Next takes it for the end of the stream
        # ---- a secret-key packet: ReadEntity takes a *packet.PrivateKey for the primary key (and a secret subkey for a subkey); their
        # parsers (S2K, checksum) are not restated: the entity it starts -- or the rest of the entity it is a subkey of -- gets no verdict
        if tag in (5, 7) and (cur is None or tag == 5):
            close(at)
            cur = EntityWalk(start=at)
            out.append(cur)
            if leading_junk:
                cur.error = "first packet was not a public/private key"
            else:
                cur.unknown = "secret-key packet"
            leading_junk = False
            pos = skip_to_next_primary(pos)
            cur.end = pos
            cur = None
            run = None
            continue
        # ---- a key packet
        if tag in (6, 14) and (cur is None or tag == 6):
            close(at)
            cur = EntityWalk(start=at)
            out.append(cur)
            cur.roles.append(["primary", 0, False])
            if leading_junk:
                cur.error = "first packet was not a public/private key"
            leading_junk = False
            key = None
            try:
                if ln > 4096:
                    cur.unknown = "body over 4096 bytes"
                elif len(body) == 0:
                    raise _Truncated()
                elif body[0] < 4:
                    cur.unknown = "version-3 key"
                else:
                    key, cur.key_hash, unk = parse_public_key_strict(body, False)
                    if unk:
                        cur.unknown = unk
            except (UnsupportedError, _Truncated) as e:
                cur.error = cur.error or ("public key: %s" % (e or "truncated"))
            cur.primary = key
            if key is not None and not key.can_sign() and cur.error is None:
                cur.error = "primary key cannot be used for signatures"
            if key is None:
                # nothing more can be said about this entity: go to where the next one starts (readToNextPublicKey)
                pos = skip_to_next_primary(pos)
                cur.end = pos
                cur = None
            run = None
            continue
        if tag in (2, 14) and ln > 4096:
            if cur is not None:
                cur.unknown = cur.unknown or "body over 4096 bytes"
                run = ("dead", None)
            else:
                leading_junk = True
            continue
        if cur is None:
            # ReadEntity's first packet is not a key: Parse ends with no (further) entity.  Only possible at the very start.
            leading_junk = True
            continue
        if cur.error is not None and tag != 6:
            continue                              # already refused: its remaining packets change nothing
        if tag == 7:
            cur.unknown = cur.unknown or "secret-subkey packet"
            run = ("dead", None)
            continue
        if truncated and tag == 13:
            fail("truncated packet")              # UserId.parse is ioutil.ReadAll: io.ErrUnexpectedEOF
            break
        if tag == 13:
            ident = {"name": body, "self_sig": None, "sigs": [], "framed": b"\xb4" + struct.pack(">I", len(body)) + body, "at": at}
            cur.identities.append(ident)
            cur.roles.append(["uid", len(cur.identities) - 1, False])
            run = ("uid", ident)
            continue
        if tag == 14:
            sk = {"key": None, "sig": None, "framed": b""}
            try:
                if len(body) == 0:
                    raise _Truncated()
                if body[0] < 4:
                    cur.unknown = cur.unknown or "version-3 key"
                else:
                    sk["key"], sk["framed"], unk = parse_public_key_strict(body, True)
                    if unk:
                        cur.unknown = cur.unknown or unk
            except (UnsupportedError, _Truncated) as e:
                fail("subkey: %s" % (e or "truncated"))
            if sk["key"] is None:
                run = ("dead", None)               # no verdict about what follows this subkey (or the entity is refused already)
                if cur.error is None and cur.unknown is None:
                    fail("subkey")
                continue
            cur.subkeys.append(sk)
            sk["chosen_role"] = None
            cur.roles.append(["subkey", len(cur.subkeys) - 1, False])
            run = ("sub", sk)
            continue
        if tag == 2:
            v3 = len(body) > 0 and body[0] < 4
            try:
                sig = parse_signature_v3_body(body) if v3 else parse_signature_body(body, 0, 2)
            except _TooDeep:
                cur.unknown = cur.unknown or "embedded signature nesting"      # x/crypto parses it recursively and may accept
                continue
            except (StructuralError, UnsupportedError, _Truncated) as e:
                fail("signature: %s" % (e or "truncated"))
                continue
            raw = blob[at:pos]
            if v3:
                cur.roles.append(["ignored", 0, False])
                run = None                         # a SignatureV3 is not a *packet.Signature: it ends the run and is ignored
                continue
            if run is not None and run[0] == "dead":
                continue
            if run is not None and run[0] == "uid":
                ident = run[1]
                if sig.sig_type in (0x10, 0x13) and sig.issuer is not None and sig.issuer == cur.primary.key_id:
                    cur.checks.append(CertCheck("uid", cur.primary, cur.key_hash + ident["framed"], sig, raw))
                    ident["self_sig"] = sig
                    cur.roles.append(["self", len(cur.identities) - 1, False])
                else:
                    if sig.issuer is not None:
                        cur.roles.append(["ident_sig", len(cur.identities) - 1, False])
                    if sig.issuer is None:
                        # one of identity.Signatures without issuer subpacket: PGPCertificateInstance.Signers dereferences nil
                        # (crypto_pgp.go:80-88) -- the reference panics where this walk would silently drop a signer
                        cur.unknown = cur.unknown or "certification without issuer subpacket"
                    ident["sigs"].append((sig, raw, cur.key_hash + ident["framed"]))
                continue
            if run is not None and run[0] == "sub":
                sk = run[1]
                if sig.sig_type not in (0x18, 0x28):
                    fail("subkey signature with wrong type")
                    continue
                signed = cur.key_hash + sk["framed"]
                # VerifyKeySignature verifies with the primary key it holds, whatever issuer the signature names (or none)
                cur.checks.append(CertCheck("binding", cur.primary, signed, sig, raw))
                if sig.flag_sign:
                    if sig.embedded is None:
                        fail("signing subkey is missing cross-signature")
                        continue
                    cur.checks.append(CertCheck("cross", sk["key"], signed, sig.embedded, _frame_sig(sig.embedded_body)))
                take = sig.sig_type == 0x28 or sk["sig"] is None or (sk["sig"].sig_type != 0x28 and sig.creation_time > sk["sig"].creation_time)
                cur.roles.append(["subkey_sig", len(cur.subkeys) - 1, False])
                if take:
                    if sk["chosen_role"] is not None:
                        cur.roles[sk["chosen_role"]][2] = False
                    sk["chosen_role"] = len(cur.roles) - 1
                    cur.roles[-1][2] = True
                    sk["sig"] = sig
                continue
            cur.roles.append(["revocation" if sig.sig_type == 0x20 else "ignored", 0, False])
            if sig.sig_type == 0x20:
                cur.revocations.append((sig, raw))
            continue
        # every other packet type packet.Read knows: ignored by ReadEntity, but its parser is not modelled here
        cur.unknown = cur.unknown or ("packet type %d" % tag)
        run = None
    close(pos)
    return out


def _verify_with_key(key: PublicKey, signed: bytes, sig: Signature) -> Optional[bool]:
    """VerifyUserIdSignature / VerifyKeySignature / VerifyRevocationSignature: hash (the reference's `hashFunc.Available()` first:
    MD5 / RIPEMD-160 by HASH_POLICY -- None when the deployment has not said whether its binary links them), then VerifySignature."""
    name = HASH_BY_ID.get(sig.hash_id)
    if name is None:
        return False
    if name in HASH_POLICY and HASH_POLICY[name] is not True:
        return None if HASH_POLICY[name] is None else False
    h = new_hash(name)
    h.update(signed)
    h.update(sig.hash_suffix)
    return verify_signature(key, sig.hash_id, h.digest(), sig) == ST_OK


def _check_ok(c: CertCheck) -> Optional[bool]:
    """VerifySignature of one certificate check: CanSign, hash tag, algorithm match, the public-key operation."""
    return _verify_with_key(c.key, c.signed, c.sig)


def walk_valid(w: EntityWalk) -> Optional[bool]:
    """Would ReadEntity return this entity?  True / False, or None when the walk met a shape left to the reference and nothing it
    did follow refuses the entity."""
    if w.error is not None:
        return False
    verdicts = [_check_ok(c) for c in w.checks]
    if any(v is False for v in verdicts):
        return False
    return None if (w.unknown is not None or any(v is None for v in verdicts)) else True


def walk_signers(w: EntityWalk) -> List[int]:
    """PGPCertificateInstance.Signers (crypto_pgp.go:80-88): the issuers of identity.Signatures over Entity.Identities -- the
    identities that have a self-signature, a later one of the same name replacing an earlier one.  (Go walks the map in random
    order; here: order of appearance.  A signature without issuer subpacket makes the reference dereference nil: skipped.)"""
    by_name = {}
    for ident in w.identities:
        if ident["self_sig"] is not None:
            by_name[ident["name"]] = ident
    return [s.issuer for ident in by_name.values() for s, _, _ in ident["sigs"] if s.issuer is not None]


def _primary_self_sig(w: EntityWalk) -> Optional[Signature]:
    """EntityList.KeysById's choice for the primary key: the first identity's self-signature unless a later identity's says
    "primary user id" (map order in Go; order of appearance here)."""
    by_name = {}
    for ident in w.identities:
        if ident["self_sig"] is not None:
            by_name[ident["name"]] = ident
    chosen = None
    for ident in by_name.values():
        if chosen is None:
            chosen = ident["self_sig"]
        elif ident["self_sig"].is_primary_id:
            chosen = ident["self_sig"]
            break
    return chosen


def read_entities(blob: bytes) -> List[Entity]:
    """Key material, KeysByIdUsage facts and Signers() of every entity of a certificate / keyring blob whose key packets parse
    (no signature is verified here: `entity_checks` / `parse_certificate` do that)."""
    ents: List[Entity] = []
    for w in walk_certificate(blob):
        if w.primary is None:
            continue
        ss = _primary_self_sig(w)
        e = Entity(primary=w.primary,
                   flags_valid=bool(ss and ss.flags_valid), flag_sign=bool(ss and ss.flag_sign), flag_certify=bool(ss and ss.flag_certify))
        e.self_sig_revoked = bool(ss and ss.revocation_reason is not None)
        e.revoked = bool(w.revocations)
        named = [i for i in w.identities if i["self_sig"] is not None] or w.identities
        e.name = named[0]["name"].decode("utf-8", "replace") if named else ""
        for sk in w.subkeys:
            s = sk["sig"]
            e.subkeys.append((sk["key"], bool(s and s.flags_valid), bool(s and s.flag_sign), bool(s and s.revocation_reason is not None)))
        e.certifiers = walk_signers(w)
        e.serialized = blob[w.start:w.end]
        ents.append(e)
    return ents


def entity_checks(blob: bytes):
    """Per entity the walk meets (as bftkv_host_certs_verify lists them): primary (None: its key packet does not parse or is of a
    shape left to the reference), valid (ReadEntity returns it: True / False, None = left to the reference), third_party =
    [(issuer, signed bytes, Signature)] of the identities Signers() walks."""
    out = []
    for w in walk_certificate(blob):
        by_name = {}
        for ident in w.identities:
            if ident["self_sig"] is not None:
                by_name[ident["name"]] = ident
        tp = [(s.issuer, signed, s) for ident in by_name.values() for s, _, signed in ident["sigs"] if s.issuer is not None]
        out.append({"primary": w.primary, "valid": walk_valid(w), "third_party": tp, "walk": w})
    return out


def parse_certificate(blob: bytes) -> List[EntityWalk]:
    """PGPCertificate.Parse (crypto_pgp.go:236-249): the entities ReadEntity returns before its first error.  An entity whose
    validity is left to the reference (`unknown`) ends the list with a None."""
    out = []
    for w in walk_certificate(blob):
        v = walk_valid(w)
        if v is None:
            out.append(None)
            break
        if not v:
            break
        out.append(w)
    return out


def verified_certifiers(blob_entity: dict, keyring: List[Entity]) -> List[int]:
    """CheckQuorumCert: ids of keyring members whose certification over (key, uid) verifies."""
    ids = []
    for issuer, signed, sig in blob_entity["third_party"]:
        for ent, key in keys_by_id_usage_sign(keyring, issuer):
            if _verify_with_key(key, signed, sig):
                ids.append(ent.id)
                break
    return ids
