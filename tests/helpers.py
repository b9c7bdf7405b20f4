"""Shared helpers of the parity tests: build oracle keyrings / quorums from a corpus cluster and the
matching C-ABI inputs."""
from __future__ import annotations

import numpy as np

from corpus import build as cb
from oracle import collective as col
from oracle import openpgp as pgp
from oracle import wotqs


def nbytes(x: int) -> bytes:
    return x.to_bytes(max(1, (x.bit_length() + 7) // 8), "big")


def oracle_keyring(cluster: cb.Cluster, include_client=False) -> col.Keyring:
    ents = []
    for kp in cluster.replicas + ([cluster.client] if include_client else []):
        e = pgp.read_entities(kp.entity)
        assert len(e) == 1 and e[0].id == kp.key_id
        ents.append(e[0])
    return col.Keyring(keyring=ents)


def abi_keys(kr: col.Keyring):
    """bftkv_gpu_pubkey records from an oracle keyring, in getKeyring() order (secring first)."""
    out = []
    for e in kr.get_keyring():
        cands = [(e.primary, e.flags_valid, e.flag_sign, e.self_sig_revoked)] + [(k, fv, fs, rr) for k, fv, fs, rr in e.subkeys]
        for k, fv, fs, rr in cands:
            usable = not (e.revoked or rr) and not (fv and not fs)
            d = {"key_id": k.key_id, "entity_id": e.id, "pk_algo": k.pk_algo, "usable_sign": usable}
            if k.pk_algo in (1, 2, 3):
                d.update(n=nbytes(k.n), e=nbytes(k.e))
            elif k.pk_algo == 17:
                d.update(n=nbytes(k.p), e=nbytes(k.q), g=nbytes(k.g), y=nbytes(k.y))
            out.append(d)
    return out


def clique_quorum(cluster: cb.Cluster) -> wotqs.WotQ:
    ids = [r.key_id for r in cluster.replicas]
    qc = wotqs.new_qc(ids, len(ids), wotqs.AUTH, 0)
    return wotqs.WotQ([qc] if qc else [])


def abi_qcs(q: wotqs.WotQ):
    return [(qc.f, qc.min, qc.threshold, qc.suff, qc.nodes) for qc in q.qcs]


def oracle_collective(kr, q, corpus, i):
    from oracle.packet import SignaturePacket
    ss = SignaturePacket(Type=1, Data=corpus.ss_data(i) or None)
    return col.collective_verify(kr, corpus.tbss(i), ss, q)


def random_framing_streams(cl, n_items, seed=77):
    """Random OpenPGP packet streams around the signatures of cluster ``cl``: signature packets re-framed in every header
    format, unknown and non-signature packets, stray bytes, truncations, and items with more than 96 packet events.
    Returns (tbs_list, stream_list, hdr, signing_rng)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    from corpus.keys import DRBG
    srng = DRBG("framing-fuzz")

    def hdr(tag, ln, fmt):
        if fmt == 0:                                   # new format, shortest legal length encoding
            if ln < 192: return bytes([0xC0 | tag, ln])
            if ln < 8384: return bytes([0xC0 | tag, ((ln - 192) >> 8) + 192, (ln - 192) & 0xFF])
            return bytes([0xC0 | tag, 255]) + ln.to_bytes(4, "big")
        if fmt == 1: return bytes([0xC0 | tag, 255]) + ln.to_bytes(4, "big")          # new format, 5-octet length
        if fmt == 2 and tag < 16 and ln < 256: return bytes([0x80 | (tag << 2)]) + bytes([ln])
        if fmt == 3 and tag < 16 and ln < 65536: return bytes([0x80 | (tag << 2) | 1]) + ln.to_bytes(2, "big")
        if tag < 16: return bytes([0x80 | (tag << 2) | 2]) + ln.to_bytes(4, "big")
        return bytes([0xC0 | tag, 255]) + ln.to_bytes(4, "big")

    tbs_l, ss_l = [], []
    for i in range(n_items):
        tbs = rng.bytes(int(rng.integers(0, 300)))
        sigs = [cb.detach_sign(r, tbs, srng) for r in cl.replicas]
        parts = []
        n_parts = int(rng.integers(1, 12)) if i % 10 else int(rng.integers(100, 140))      # every 10th: > WALK_CAP events
        for _ in range(n_parts):
            k = int(rng.integers(0, 12))
            if k < 5:                                   # a signature, re-framed in a random header format
                s = sigs[int(rng.integers(0, len(sigs)))]
                body = s[3:] if s[1] >= 192 else s[2:]
                parts.append(hdr(2, len(body), int(rng.integers(0, 5))) + body)
            elif k < 7:                                 # unknown packet type, silently skipped
                ln = int(rng.integers(0, 400))
                parts.append(hdr(int(rng.choice([20, 40, 60, 63])), ln, int(rng.integers(0, 2))) + rng.bytes(ln))
            elif k < 9:                                 # known non-signature packet (user id / literal / marker-like)
                ln = int(rng.integers(0, 60))
                parts.append(hdr(int(rng.choice([13, 11, 6, 14])), ln, int(rng.integers(0, 5))) + rng.bytes(ln))
            elif k == 9:                                # small packets back to back
                parts.append(b"".join(hdr(13, 1, 2) + b"x" for _ in range(int(rng.integers(1, 9)))))
            elif k == 10 and i % 3 == 0:                # stray byte without the tag MSB: ends the stream
                parts.append(bytes([int(rng.integers(0, 128))]))
            else:
                parts.append(sigs[int(rng.integers(0, len(sigs)))])
        data = b"".join(parts)
        if i % 7 == 3 and len(data) > 4:
            data = data[:int(rng.integers(1, len(data)))]                               # truncated somewhere
        tbs_l.append(tbs); ss_l.append(data)
    return tbs_l, ss_l, hdr, srng


def sign_body(kp, signed: bytes, srng, hashed_extra: bytes = b"", unhashed: bytes = b"", trailing: bytes = b"") -> bytes:
    """One valid v4 signature BODY (no packet header) over ``signed`` with extra hashed subpackets, an unhashed area and bytes
    behind the MPIs."""
    import hashlib, struct
    prefix = cb.sig_prefix(0x00, kp.algo, cb._hashed_area(kp.key_id, hashed_extra))
    digest = hashlib.sha256(signed + cb.hash_suffix(prefix)).digest()
    pkt = cb.make_sig_packet(kp, prefix, digest, srng)
    body = pkt[6:] if pkt[1] == 255 else (pkt[3:] if pkt[1] >= 192 else pkt[2:])
    assert body[:len(prefix)] == prefix and body[len(prefix):len(prefix) + 2] == b"\0\0"
    return prefix + struct.pack(">H", len(unhashed)) + unhashed + body[len(prefix) + 2:] + trailing


def partial_frame(tag: int, body: bytes, rng, zero_final=False, max_pow=9) -> bytes:
    """New-format packet with partial body lengths: 2^k-byte chunks (k random), the last chunk with a definite length in a
    random encoding (zero-length when asked)."""
    assert len(body) >= 1
    out = bytearray([0xC0 | tag])
    pos = 0
    first = True
    while True:
        left = len(body) - pos
        kmax = min(max_pow, left.bit_length() - 1) if left else -1
        if kmax >= 0 and (first or (rng.random() < 0.7 and not (zero_final and left == 0))):
            k = int(rng.integers(0, kmax + 1))
            out.append(224 + k)
            out += body[pos:pos + (1 << k)]
            pos += 1 << k
            first = False
            continue
        last = b"" if zero_final else body[pos:]
        if zero_final and left:
            # spend what is left in partial chunks first
            k = left.bit_length() - 1
            out.append(224 + k); out += body[pos:pos + (1 << k)]; pos += 1 << k
            continue
        ln = len(last)
        enc = int(rng.integers(0, 3))
        if enc == 0 and ln < 192: out.append(ln)
        elif enc <= 1 and 192 <= ln < 8384: out += bytes([((ln - 192) >> 8) + 192, (ln - 192) & 0xFF])
        else: out += bytes([255]) + ln.to_bytes(4, "big")
        out += last
        return bytes(out)


def exotic_framing_streams(cl, n_items, seed=1234):
    """Signature streams in the shapes x/crypto reads but no writer of the path produces: partial body lengths (also on unknown
    and user-id packets, also with a zero-length last chunk), old-format indeterminate lengths, lengths that run past the end of
    the stream, bodies beyond bufio's 4096 bytes (a big hashed or unhashed area; bytes behind the MPIs), cut at random places.
    Returns (tbs_list, stream_list)."""
    import numpy as np, struct
    from corpus.keys import DRBG
    rng = np.random.default_rng(seed)
    srng = DRBG("exotic-framing")

    def notation(n):       # an unknown, non-critical subpacket (type 100) of n bytes in all
        body = bytes(rng.integers(0, 256, size=n - 6, dtype=np.uint8))
        return bytes([255]) + struct.pack(">I", len(body) + 1) + bytes([100]) + body

    def definite(tag, body, declared=None):
        ln = len(body) if declared is None else declared
        return bytes([0xC0 | tag, 255]) + ln.to_bytes(4, "big") + body

    tbs_l, ss_l = [], []
    for i in range(n_items):
        tbs = bytes(rng.integers(0, 256, size=int(rng.integers(0, 200)), dtype=np.uint8))
        parts = []
        long_item = i % 16 == 15                      # more packet events than the walk's scratch row holds (sequential fill pass)
        pool = []
        for _ in range(int(rng.integers(100, 130)) if long_item else int(rng.integers(1, 8))):
            kp = cl.replicas[int(rng.integers(0, len(cl.replicas)))]
            shape = int(rng.integers(0, 14))
            if long_item:
                shape = int(rng.choice([0, 3, 4, 9, 10, 13]))                 # (nothing that ends or swallows the stream)
            big = 5 if long_item else int(rng.integers(0, 6))
            extra = notation(int(rng.choice([4090, 4096, 4200, 9000]))) if big == 0 else b""
            unh = notation(int(rng.choice([4000, 4096, 5000]))) if big == 1 else b""
            trail = bytes(rng.integers(0, 256, size=int(rng.choice([1, 100, 3800, 5000])), dtype=np.uint8)) if big == 2 else b""
            if long_item and len(pool) >= 6:
                body = pool[int(rng.integers(0, len(pool)))]
            else:
                body = sign_body(kp, tbs, srng, extra, unh, trail)
                pool.append(body)
            if shape <= 2: parts.append(definite(2, body))
            elif shape <= 5: parts.append(partial_frame(2, body, rng, max_pow=int(rng.choice([3, 6, 9, 13]))))
            elif shape == 6: parts.append(partial_frame(2, body, rng, zero_final=True))
            elif shape == 7: parts.append(bytes([0x80 | (2 << 2) | 3]) + body)                       # indeterminate length
            elif shape == 8: parts.append(definite(2, body, declared=len(body) + int(rng.integers(1, 6000))))
            elif shape == 9: parts.append(partial_frame(int(rng.choice([20, 60, 63])), bytes(rng.integers(0, 256, size=int(rng.integers(1, 900)), dtype=np.uint8)), rng))
            elif shape == 10: parts.append(partial_frame(13, b"user id " * int(rng.integers(1, 40)), rng))
            elif shape == 11: parts.append(bytes([0x80 | (15 << 2) | 3]) + b"unknown old-format type, indeterminate")
            elif shape == 12: parts.append(partial_frame(2, body[:int(rng.integers(1, len(body)))], rng))   # a chain that ends early
            else: parts.append(cb.detach_sign(kp, tbs, srng))
        data = b"".join(parts)
        if i % 5 == 4 and len(data) > 8:
            data = data[:int(rng.integers(1, len(data)))]
        tbs_l.append(tbs); ss_l.append(data)
    return tbs_l, ss_l


def plausible_unsigned_streams(n_streams, seed=4242, issuers=(0x1122334455667788,)):
    """Streams of well-formed but UNSIGNED signature bodies (v4 RSA / DSA with issuers drawn from ``issuers``, 4096-bit and 8000-byte
    MPIs, hashed / unhashed areas beyond bufio's buffer, bytes behind the MPIs, one v3 body) in every framing -- definite in all
    header formats, partial chunks of random sizes, indeterminate, lengths past the end -- mixed with junk packets of known and
    unknown types, every third stream cut, every fourth mutated."""
    import struct
    rng = np.random.default_rng(seed)
    ct = b"\x05\x02" + struct.pack(">I", 1500000000)

    def v4(issuer, extra=b"", unh=b"", nb=256, algo=1):
        hashed = ct + b"\x09\x10" + struct.pack(">Q", issuer) + extra
        b = bytes([4, 0, algo, 8]) + struct.pack(">H", len(hashed)) + hashed + struct.pack(">H", len(unh)) + unh + b"\xab\xcd"
        b += struct.pack(">H", nb * 8) + bytes([0xD5]) * nb
        return b + (struct.pack(">H", 160) + bytes([0xE6]) * 20 if algo == 17 else b"")

    def notation(n):
        return bytes([255]) + struct.pack(">I", n - 5) + bytes([100]) + bytes(n - 6)
    bodies = []
    for iss in issuers:
        bodies += [v4(iss), v4(iss, algo=17, nb=20), v4(iss, extra=notation(4300)), v4(iss, unh=notation(5000)), v4(iss, nb=512), v4(iss, nb=8000),
                   bytes([3, 5, 0]) + struct.pack(">I", 1) + struct.pack(">Q", iss) + bytes([1, 8]) + b"\x12\x34" + struct.pack(">H", 2048) + bytes([0xC1]) * 256]

    def hdr(tag, ln, fmt):
        if fmt == 0:
            if ln < 192: return bytes([0xC0 | tag, ln])
            if ln < 8384: return bytes([0xC0 | tag, ((ln - 192) >> 8) + 192, (ln - 192) & 0xFF])
        if fmt == 2 and tag < 16 and ln < 256: return bytes([0x80 | (tag << 2), ln])
        if fmt == 3 and tag < 16 and ln < 65536: return bytes([0x80 | (tag << 2) | 1]) + ln.to_bytes(2, "big")
        if fmt == 4 and tag < 16: return bytes([0x80 | (tag << 2) | 3])
        if fmt == 5: return bytes([0xC0 | tag, 224 + int(rng.integers(0, 14))])
        return bytes([0xC0 | tag, 255]) + ln.to_bytes(4, "big")

    def chunked(tag, body):
        out, pos, first = bytearray([0xC0 | tag]), 0, True
        while True:
            left = len(body) - pos
            if left > 0 and (first or rng.random() < 0.75):
                k = int(rng.integers(0, min(13, left.bit_length() - 1) + 1))
                out.append(224 + k); out += body[pos:pos + (1 << k)]; pos += 1 << k; first = False
                continue
            last = body[pos:]
            ln, enc = len(last), int(rng.integers(0, 3))
            if enc == 0 and ln < 192: out.append(ln)
            elif enc <= 1 and 192 <= ln < 8384: out += bytes([((ln - 192) >> 8) + 192, (ln - 192) & 0xFF])
            else: out += bytes([255]) + ln.to_bytes(4, "big")
            return bytes(out + last)

    for it in range(n_streams):
        parts = []
        for _ in range(int(rng.integers(0, 8))):
            if rng.random() < 0.45:
                b = bodies[int(rng.integers(0, len(bodies)))]
                if rng.random() < 0.3:
                    b = b + rng.bytes(int(rng.choice([1, 50, 3900, 4200])))
                m = int(rng.integers(0, 4))
                parts.append(hdr(2, len(b), int(rng.integers(0, 4))) + b if m == 0 else chunked(2, b) if m == 1 else
                             bytes([0x8B]) + b if m == 2 else hdr(2, len(b) + int(rng.integers(1, 5000)), 1) + b)
            else:
                ln = int(rng.integers(0, 300))
                parts.append(hdr(int(rng.choice([2, 13, 11, 6, 14, 20, 40, 63, 1, 9, 17, 10, 12, 15])), ln, int(rng.integers(0, 7))) + rng.bytes(ln))
        data = bytearray(b"".join(parts))
        if it % 3 == 1 and len(data) > 1:
            data = data[:int(rng.integers(1, len(data)))]
        if it % 4 == 2 and len(data):
            for _ in range(int(rng.integers(1, 4))):
                data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
        yield bytes(data)


def cat(parts):
    """byte strings -> (blob, n+1 offsets) as the batched entry points take them"""
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts], dtype=np.uint64)
    return np.frombuffer(b"".join(parts) + b"\0", dtype=np.uint8)[:int(off[-1])].copy(), off
