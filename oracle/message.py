"""Oracle restatement of the transport message-signature check.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference:
  PGPMessage.Decrypt                  crypto/pgp/crypto_pgp.go:453-471   (what is returned for which outcome)
  PGPMessage.Encrypt / EncryptStream  crypto/pgp/crypto_pgp.go:419-451   (what a sender emits: signed, binary, file name =
                                                                          base64(nonce), time 0)
  callers                             protocol/server.go:563-569, transport/transport.go:116-125
and, for the un-vendored half (golang.org/x/crypto v0.0.0-20191227163750-53104e6ec876, openpgp.ReadMessage ->
readSignedMessage -> signatureCheckReader.Read; packet.OnePassSignature.parse; packet.LiteralData.parse;
packet.partialLengthReader), a restatement FROM MEMORY -- parity unpinned against Go, pinned against GnuPG by
tests/golden/gpg_messages.json (gpg-made signed messages + generator-made ones judged by gpg).

Scope: the part of Decrypt AFTER the session key has opened the message.  The public-key decryption of the session key and
the AES-CFB stream are private-key / symmetric work that stays on the CPU; the input here is the
plaintext packet sequence inside the encrypted container with the MDC trailer already stripped:
    [one-pass signature] [literal data] [signature]
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import List, Optional, Tuple

from . import openpgp as pgp

# per-message outcome (bftkv_gpu.h BFTKV_MSG_*)
MSG_OK = 0               # signed by a keyring signing key and verified: Decrypt returns (plain, nonce, peer, nil)
MSG_SIGNATURE_ERROR = 1  # m.SignatureError != nil: Decrypt returns it (server.go:564 refuses the request)
MSG_READ_ERROR = 2       # openpgp.ReadMessage failed: crypto.ErrDecryptionFailed
MSG_NOT_SIGNED = 3       # no one-pass signature before the literal: crypto.ErrInvalidTransportSecurityData
MSG_UNVERIFIED = 4       # signed by a key the keyring cannot verify with: Decrypt returns a NIL error (crypto_pgp.go:458 comment)
MSG_UNSUPPORTED = 5      # fenced input (compressed packet, text-mode signature, v3 signature, partial-length signature ...)


@dataclass
class MessageResult:
    status: int
    plain: bytes = b""
    file_name: bytes = b""
    signed_by_key_id: int = 0
    peer: Optional[int] = None      # GetCertById(SignedByKeyId): entity id, by PRIMARY key id only (crypto_pgp.go:206-219)
    sig_status: Optional[int] = None  # oracle.openpgp ST_* of the trailing signature, when one was evaluated


def read_packet(buf: bytes, pos: int) -> Tuple[int, bytes, int]:
    """packet.Read framing including partial body lengths (partialLengthReader): returns (tag, body, next position).
    Raises EOFError at a clean end, pgp.StructuralError / pgp._Truncated / pgp.UnsupportedError otherwise."""
    tag, start, ln = pgp.read_header(buf, pos)
    if ln == -1:
        raise pgp.UnsupportedError("indeterminate length")     # fenced, as in oracle.openpgp.next_packet
    if ln >= 0:
        if start + ln > len(buf):
            raise pgp._Truncated()
        return tag, buf[start:start + ln], start + ln
    # partial body lengths: the length octet at start-1 is 224..254; chunks of 2^(b & 0x1F) until a definite length closes
    body = bytearray()
    p = start - 1
    while True:
        if p >= len(buf):
            raise pgp._Truncated()
        b1 = buf[p]
        if b1 < 192:
            ln, p = b1, p + 1
            last = True
        elif b1 < 224:
            if p + 1 >= len(buf):
                raise pgp._Truncated()
            ln, p = ((b1 - 192) << 8) + buf[p + 1] + 192, p + 2
            last = True
        elif b1 == 255:
            if p + 5 > len(buf):
                raise pgp._Truncated()
            ln, p = int.from_bytes(buf[p + 1:p + 5], "big"), p + 5
            last = True
        else:
            ln, p = 1 << (b1 & 0x1F), p + 1
            last = False
        if p + ln > len(buf):
            raise pgp._Truncated()
        body += buf[p:p + ln]
        p += ln
        if last:
            return tag, bytes(body), p


def _peer(keyring: List[pgp.Entity], key_id: int) -> Optional[int]:
    for e in keyring:
        if e.primary.key_id == key_id:
            return e.primary.key_id
    return None


def read_signed_message(keyring: List[pgp.Entity], stream: bytes) -> MessageResult:
    """readSignedMessage + signatureCheckReader over an already-decrypted packet sequence, folded into Decrypt's outcome."""
    pos = 0
    signed_by = None          # (entity, key) = keys[0]
    ops = None                # (sig_type, hash_id, key_id)
    res = MessageResult(MSG_READ_ERROR)
    literal = None
    # FindLiteralData
    while True:
        try:
            tag, body, pos = read_packet(stream, pos)
        except EOFError:
            return MessageResult(MSG_READ_ERROR)            # io.EOF before any literal data
        except pgp.UnsupportedError:
            return MessageResult(MSG_UNSUPPORTED)
        except (pgp.StructuralError, pgp._Truncated):
            return MessageResult(MSG_READ_ERROR)
        if tag == 8:                                        # Compressed: Go pushes the inflated body; fenced here
            return MessageResult(MSG_UNSUPPORTED)
        if tag == 4:                                        # OnePassSignature.parse: exactly 13 bytes are read
            if len(body) < 13:
                return MessageResult(MSG_READ_ERROR)
            if body[0] != 3:
                return MessageResult(MSG_READ_ERROR)        # unsupported one-pass-signature version
            if body[2] not in pgp.HASH_BY_ID:
                return MessageResult(MSG_READ_ERROR)        # s2k.HashIdToHash miss
            if ops is not None:
                return MessageResult(MSG_UNSUPPORTED)       # several one-pass packets: legal (each IsLast) but fenced
            if body[12] == 0:
                return MessageResult(MSG_READ_ERROR)        # !IsLast: "nested signatures"
            sig_type, hash_id, key_id = body[1], body[2], int.from_bytes(body[4:12], "big")
            if pgp.HASH_BY_ID[hash_id] in ("md5", "ripemd160"):
                return MessageResult(MSG_UNSUPPORTED)       # availability unknown without the x/crypto build: fenced
            if sig_type == 0x01:
                return MessageResult(MSG_UNSUPPORTED)       # canonical-text hashing: fenced
            if sig_type != 0x00:
                return MessageResult(MSG_READ_ERROR)        # hashForSignature: unsupported signature type
            ops = (sig_type, hash_id, key_id)
            keys = pgp.keys_by_id_usage_sign(keyring, key_id)
            signed_by = keys[0] if keys else None
            continue
        if tag == 11:                                       # LiteralData.parse
            if len(body) < 2:
                return MessageResult(MSG_READ_ERROR)
            fl = body[1]
            if len(body) < 2 + fl + 4:
                return MessageResult(MSG_READ_ERROR)
            literal = (bytes(body[2:2 + fl]), bytes(body[2 + fl + 4:]))
            break
        if tag in pgp._KNOWN_TAGS:
            continue                                        # parsed and ignored by the type switch (bodies not validated: fenced)
        continue                                            # unknown packet type: skipped by Reader.Next
    file_name, plain = literal
    if ops is None:
        return MessageResult(MSG_NOT_SIGNED, plain, file_name)
    res = MessageResult(MSG_UNVERIFIED, plain, file_name, ops[2], _peer(keyring, ops[2]))
    if signed_by is None:
        return res                                          # UnverifiedBody is not wrapped: SignatureError stays nil
    # signatureCheckReader at EOF of the literal body
    while True:
        try:
            tag, body, pos = read_packet(stream, pos)
        except pgp.UnsupportedError:
            res.status = MSG_UNSUPPORTED
            return res
        except (EOFError, pgp.StructuralError, pgp._Truncated):
            res.status = MSG_SIGNATURE_ERROR                # SignatureError = the Next() error (io.EOF included)
            return res
        if tag == 2 or tag in pgp._KNOWN_TAGS:
            break
    if tag != 2:
        res.status = MSG_SIGNATURE_ERROR                    # "LiteralData not followed by signature"
        return res
    if len(body) >= 1 and body[0] < 4:
        res.status = MSG_UNSUPPORTED                        # SignatureV3: fenced as everywhere else
        return res
    try:
        sig = pgp.parse_signature_body(body)
    except (pgp.StructuralError, pgp.UnsupportedError, pgp._Truncated):
        res.status, res.sig_status = MSG_SIGNATURE_ERROR, pgp.ST_PARSE_ERROR
        return res
    ent, key = signed_by
    # VerifySignature(h, sig): h was created from the ONE-PASS packet's hash id and has absorbed the body
    h = hashlib.new(pgp.HASH_BY_ID[ops[1]])
    h.update(plain)
    if not key.can_sign():
        st = pgp.ST_KEY_CANNOT_SIGN
    else:
        h.update(sig.hash_suffix)
        digest = h.digest()
        if sig.hash_id != ops[1] and key.pk_algo in (pgp.PK_RSA, pgp.PK_RSA_SIGN_ONLY) and key.pk_algo == sig.pk_algo and \
                digest[:2] == bytes(sig.hash_tag):
            # rsa.VerifyPKCS1v15(hash = sig.Hash, hashed = digest of another algorithm): length mismatch -> error.
            # (SHA-1 vs RIPEMD-160 share a length, but RIPEMD-160 is fenced above / in parse.)
            st = pgp.ST_BAD_SIG
        else:
            st = pgp.verify_signature(key, sig.hash_id, digest, sig)
    res.sig_status = st
    if st == pgp.ST_UNSUPPORTED:
        res.status = MSG_UNSUPPORTED
    else:
        res.status = MSG_OK if st == pgp.ST_OK else MSG_SIGNATURE_ERROR
    return res


# ---- writer side (test corpus): what openpgp.Encrypt hands to the encrypted container -----------------------------------
def one_pass_packet(sig_type: int, hash_id: int, pk_algo: int, key_id: int, is_last: bool = True) -> bytes:
    body = bytes([3, sig_type, hash_id, pk_algo]) + key_id.to_bytes(8, "big") + bytes([1 if is_last else 0])
    return pgp.new_format_header(4, len(body)) + body


def literal_packet(file_name: bytes, body: bytes, partial: Optional[List[int]] = None, is_binary: bool = True, time: int = 0) -> bytes:
    """Literal data packet; ``partial`` lists the powers of two of the leading partial-length chunks (Go's
    partialLengthWriter emits one chunk per Write of the largest power of two that fits), the rest closes the packet."""
    content = (b"b" if is_binary else b"t") + bytes([len(file_name)]) + file_name + time.to_bytes(4, "big") + body
    if not partial:
        return pgp.new_format_header(11, len(content)) + content
    out = bytearray([0xC0 | 11])
    p = 0
    for power in partial:
        ln = 1 << power
        assert p + ln <= len(content)
        out += bytes([224 + power]) + content[p:p + ln]
        p += ln
    rest = len(content) - p
    out += pgp.new_format_header(11, rest)[1:] + content[p:]
    return bytes(out)


def go_partial_chunks(n: int) -> List[int]:
    """Chunking of one Write of n bytes by x/crypto's partialLengthWriter (largest power of two <= remaining, capped at
    2^14... the writer tries 14 down to 0).  The closing chunk is the zero-length definite one written by Close()."""
    out = []
    while n > 0:
        for power in range(14, -1, -1):
            if n >= (1 << power):
                out.append(power)
                n -= 1 << power
                break
    return out
