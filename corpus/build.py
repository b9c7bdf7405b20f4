"""Synthetic corpora for BASELINE.json's configs.

Everything is derived from one master seed.  Output formats are the reference's own wire formats:
OpenPGP v4 packets as golang.org/x/crypto/openpgp's DetachSign emits them (new-format
header, SHA-256, creation-time + issuer in the hashed area, MPI bit count = 8*len(bytes)) inside
bftkv ``<x,v,t,sig,ss>`` packets (packet/packet.go:35-60, 192-212).

No import of ``oracle/`` and none of the product package: signing uses Python ``pow`` (with CRT) or
a pluggable batch signer (bench.py plugs in the GPU modexp for the 10^5..10^6-signature corpora).
"""
from __future__ import annotations

import hashlib
import struct
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .keys import DRBG, MASTER_SEED, load_keys

PK_RSA, PK_DSA = 1, 17
HASH_SHA256 = 8
SHA256_PREFIX = bytes.fromhex("3031300d060960864801650304020105000420")
CREATION_TIME = 0x5E000000  # fixed (2019-12-22), reproducible like gpg --faked-system-time


# ------------------------------------------------------------------------------------------------
# keys / entities
# ------------------------------------------------------------------------------------------------
@dataclass
class KeyPair:
    algo: int
    # RSA
    n: int = 0
    e: int = 0
    d: int = 0
    p: int = 0
    q: int = 0
    # DSA
    g: int = 0
    y: int = 0
    x: int = 0
    key_id: int = 0
    pub_body: bytes = b""
    name: str = ""
    entity: bytes = b""          # serialized OpenPGP entity (pubkey, uid, self-sig, certifications)
    certifiers: List[int] = field(default_factory=list)

    _dp: int = 0
    _dq: int = 0
    _qinv: int = 0

    def rsa_private(self, m: int) -> int:
        if not self._dp:
            self._dp, self._dq = self.d % (self.p - 1), self.d % (self.q - 1)
            self._qinv = pow(self.q, -1, self.p)
        m1, m2 = pow(m % self.p, self._dp, self.p), pow(m % self.q, self._dq, self.q)
        h = (self._qinv * (m1 - m2)) % self.p
        return m2 + h * self.q


def _mpi(x: int) -> bytes:
    return struct.pack(">H", x.bit_length()) + x.to_bytes((x.bit_length() + 7) // 8, "big")


def _hdr(tag: int, ln: int) -> bytes:
    if ln < 192:
        return bytes([0xC0 | tag, ln])
    if ln < 8384:
        ln -= 192
        return bytes([0xC0 | tag, 192 + (ln >> 8), ln & 0xFF])
    return bytes([0xC0 | tag, 255]) + struct.pack(">I", ln)


def make_keypair(algo: int, mat: Dict[str, int], name: str) -> KeyPair:
    if algo == PK_RSA:
        p, q, e = mat["p"], mat["q"], mat["e"]
        n = p * q
        d = pow(e, -1, (p - 1) * (q - 1))
        kp = KeyPair(PK_RSA, n=n, e=e, d=d, p=p, q=q, name=name)
        body = bytes([4]) + struct.pack(">I", CREATION_TIME) + bytes([PK_RSA]) + _mpi(n) + _mpi(e)
    else:
        p, q, g, x = mat["p"], mat["q"], mat["g"], mat["x"]
        y = pow(g, x, p)
        kp = KeyPair(PK_DSA, p=p, q=q, g=g, y=y, x=x, name=name)
        body = bytes([4]) + struct.pack(">I", CREATION_TIME) + bytes([PK_DSA]) + _mpi(p) + _mpi(q) + _mpi(g) + _mpi(y)
    kp.pub_body = body
    fp = hashlib.sha1(b"\x99" + struct.pack(">H", len(body)) + body).digest()
    kp.key_id = int.from_bytes(fp[12:], "big")
    return kp


def _hashed_area(issuer: int, extra: bytes = b"") -> bytes:
    return (b"\x05\x02" + struct.pack(">I", CREATION_TIME) + extra + b"\x09\x10" + struct.pack(">Q", issuer))


def sig_prefix(sig_type: int, pk_algo: int, hashed: bytes, hash_id: int = HASH_SHA256) -> bytes:
    """The first 6+hl bytes of a v4 signature body (also the start of the hash suffix)."""
    return bytes([4, sig_type, pk_algo, hash_id]) + struct.pack(">H", len(hashed)) + hashed


def hash_suffix(prefix: bytes) -> bytes:
    return prefix + b"\x04\xff" + struct.pack(">I", len(prefix))


def emsa(digest: bytes, k: int) -> int:
    t = SHA256_PREFIX + digest
    return int.from_bytes(b"\x00\x01" + b"\xff" * (k - len(t) - 3) + b"\x00" + t, "big")


def _dsa_sign(kp: KeyPair, digest: bytes, rng: DRBG) -> Tuple[int, int]:
    z = int.from_bytes(digest[:(kp.q.bit_length() + 7) // 8], "big")
    while True:
        k = 1 + rng.below(kp.q - 1)
        r = pow(kp.g, k, kp.p) % kp.q
        if r == 0:
            continue
        s = pow(k, -1, kp.q) * (z + kp.x * r) % kp.q
        if s:
            return r, s


def go_mpi_bytes(b: bytes) -> bytes:
    """x/crypto's signing-side MPI: bit count written as 8*len(bytes)."""
    return struct.pack(">H", 8 * len(b)) + b


def make_sig_packet(kp: KeyPair, prefix: bytes, digest: bytes, rng: Optional[DRBG] = None,
                    sig_value: Optional[int] = None) -> bytes:
    """Assemble one Go-shaped signature packet for ``digest`` = H(signed || hash_suffix(prefix))."""
    if kp.algo == PK_RSA:
        k = (kp.n.bit_length() + 7) // 8
        s = kp.rsa_private(emsa(digest, k)) if sig_value is None else sig_value
        mp = go_mpi_bytes(s.to_bytes(k, "big"))
    else:
        r, s = _dsa_sign(kp, digest, rng)
        rb = r.to_bytes((r.bit_length() + 7) // 8, "big")
        sb = s.to_bytes((s.bit_length() + 7) // 8, "big")
        mp = go_mpi_bytes(rb) + go_mpi_bytes(sb)
    body = prefix + b"\x00\x00" + digest[:2] + mp
    return _hdr(2, len(body)) + body


HASH_NAMES = {2: "sha1", 8: "sha256", 9: "sha384", 10: "sha512", 11: "sha224"}      # RFC 4880 9.4 ids -> hashlib


def detach_sign(kp: KeyPair, signed: bytes, rng: Optional[DRBG] = None, hash_id: int = HASH_SHA256) -> bytes:
    """openpgp.DetachSign(w, priv, r, nil) shape (crypto_pgp.go:353; nil config = SHA-256).  ``hash_id`` other than SHA-256 is what
    a packet.Config{DefaultHash: ...} signer emits -- DSA only (the RSA EMSA prefix of make_sig_packet is SHA-256's): the digest is
    cut to the leftmost ceil(bits(q) / 8) bytes by the signer and by the verifier alike."""
    assert hash_id == HASH_SHA256 or kp.algo == PK_DSA
    prefix = sig_prefix(0x00, kp.algo, _hashed_area(kp.key_id), hash_id)
    digest = hashlib.new(HASH_NAMES[hash_id], signed + hash_suffix(prefix)).digest()
    return make_sig_packet(kp, prefix, digest, rng)


def certify(signer: KeyPair, signee: KeyPair, uid: bytes, sig_type: int, rng: Optional[DRBG] = None) -> bytes:
    """Certification (0x10) or self-signature (0x13) over key + user id."""
    extra = b"\x02\x1b\x03" if sig_type == 0x13 else b""   # key flags: certify|sign
    prefix = sig_prefix(sig_type, signer.algo, _hashed_area(signer.key_id, extra))
    signed = (b"\x99" + struct.pack(">H", len(signee.pub_body)) + signee.pub_body +
              b"\xb4" + struct.pack(">I", len(uid)) + uid)
    digest = hashlib.sha256(signed + hash_suffix(prefix)).digest()
    return make_sig_packet(signer, prefix, digest, rng)


def bind_subkey(primary: KeyPair, sub: KeyPair, rng: Optional[DRBG] = None) -> bytes:
    """Public-subkey packet (tag 14) + subkey binding signature (0x18, key flags = encrypt) by the primary key."""
    prefix = sig_prefix(0x18, primary.algo, _hashed_area(primary.key_id, b"\x02\x1b\x0c"))
    signed = (b"\x99" + struct.pack(">H", len(primary.pub_body)) + primary.pub_body +
              b"\x99" + struct.pack(">H", len(sub.pub_body)) + sub.pub_body)
    digest = hashlib.sha256(signed + hash_suffix(prefix)).digest()
    return _hdr(14, len(sub.pub_body)) + sub.pub_body + make_sig_packet(primary, prefix, digest, rng)


def build_entity(kp: KeyPair, certifiers: Sequence[KeyPair], rng: DRBG, subkey: Optional[KeyPair] = None) -> None:
    uid = kp.name.encode()
    out = _hdr(6, len(kp.pub_body)) + kp.pub_body + _hdr(13, len(uid)) + uid
    out += certify(kp, kp, uid, 0x13, rng)
    for c in certifiers:
        out += certify(c, kp, uid, 0x10, rng)
    if subkey is not None:
        out += bind_subkey(kp, subkey, rng)
    kp.entity = out
    kp.certifiers = [c.key_id for c in certifiers]


# ------------------------------------------------------------------------------------------------
# bftkv packet framing (packet/packet.go:35-60, 117-124, 192-212)
# ------------------------------------------------------------------------------------------------
def chunk(b: Optional[bytes]) -> bytes:
    b = b or b""
    return struct.pack(">Q", len(b)) + b


def sigpkt(data: Optional[bytes], cert: Optional[bytes], completed: bool = False, typ: int = 1) -> bytes:
    if data is None and cert is None:
        typ = 0
    return bytes([typ]) + b"\x00\x00\x00\x00" + (b"\x01" if completed else b"\x00") + chunk(data) + chunk(cert)


def serialize_tbs(x: bytes, v: bytes, t: int) -> bytes:
    return chunk(x) + chunk(v) + struct.pack(">Q", t)


# ------------------------------------------------------------------------------------------------
# cluster = replica key set + client + quorum description
# ------------------------------------------------------------------------------------------------
@dataclass
class Cluster:
    replicas: List[KeyPair]
    client: KeyPair
    outsiders: List[KeyPair]      # keys NOT in the verifier's keyring (unknown issuers)
    n: int
    f: int
    threshold: int
    suff: int


def quorum_numbers(n: int) -> Tuple[int, int, int, int]:
    """wotqs.newQC arithmetic (quorum/wotqs/wotqs.go:55-62), restated for sizing corpora only."""
    f = (n - 1) // 3
    return f, 3 * f + 1, 2 * f + 1, f + (n - f) // 2 + 1


def make_cluster(n: int, dsa_fraction: float = 0.0, seed: int = MASTER_SEED, n_outsiders: int = 2,
                 key_offset: int = 0, dsa_kind: str = "dsa2048") -> Cluster:
    """``dsa_kind``: the DSA replicas' group size -- 'dsa2048' (q 256 bits, BASELINE's configs), 'dsa1024' (q 160), 'dsa1536'
    (q 224), 'dsa3072' (q 256), or a sequence of kinds dealt round-robin to the DSA replicas."""
    rng = DRBG("cluster", seed, n, dsa_fraction)
    n_dsa = int(round(n * dsa_fraction))
    rsa = load_keys("rsa2048", key_offset + (n - n_dsa) + 1 + n_outsiders)[key_offset:]
    if isinstance(dsa_kind, str):
        dsa = load_keys(dsa_kind, n_dsa) if n_dsa else []
    else:
        kinds = list(dsa_kind)
        pools = {k: load_keys(k, (n_dsa + len(kinds) - 1) // len(kinds)) for k in kinds}
        dsa = [pools[kinds[i % len(kinds)]][i // len(kinds)] for i in range(n_dsa)]
    # seeded assignment of which replica indices are DSA
    idx = list(range(n))
    for i in range(n - 1, 0, -1):
        j = rng.below(i + 1)
        idx[i], idx[j] = idx[j], idx[i]
    dsa_slots = set(idx[:n_dsa])
    replicas: List[KeyPair] = []
    ri = di = 0
    for i in range(n):
        name = "a%03d (http://localhost:%d) <a%03d@bftkv.example>" % (i + 1, 5700 + i + 1, i + 1)
        if i in dsa_slots:
            replicas.append(make_keypair(PK_DSA, dsa[di], name)); di += 1
        else:
            replicas.append(make_keypair(PK_RSA, rsa[ri], name)); ri += 1
    client = make_keypair(PK_RSA, rsa[ri], "u01 <u01@bftkv.example>"); ri += 1
    outsiders = []
    for k in range(n_outsiders):
        outsiders.append(make_keypair(PK_RSA, rsa[ri], "x%02d <x%02d@elsewhere.example>" % (k, k))); ri += 1
    f, mn, thr, suff = quorum_numbers(n)
    # replicas: self-signed only (clique cross-certifications are graph input, carried as ids);
    # client: certified by f+1 RSA clique members (scripts/setup.sh:40-43 scaled)
    for r in replicas:
        build_entity(r, [], rng)
    certs = [r for r in replicas if r.algo == PK_RSA][:f + 1]
    build_entity(client, certs, rng)
    for o in outsiders:
        build_entity(o, [], rng)
    return Cluster(replicas, client, outsiders, n, f, thr, suff)


# ------------------------------------------------------------------------------------------------
# signed-write corpus
# ------------------------------------------------------------------------------------------------
MUT_NONE, MUT_BAD_MPI, MUT_UNKNOWN_ISSUER, MUT_DUP_SIGNER, MUT_ONE_SHORT, MUT_BAD_TAG = 0, 1, 2, 3, 4, 5


@dataclass
class WriteCorpus:
    cluster: Cluster
    n_items: int
    tbss_blob: np.ndarray         # uint8, concatenated TBSS payloads
    tbss_off: np.ndarray          # uint64 [n_items+1]
    ss_blob: np.ndarray           # uint8, concatenated ss.Data (OpenPGP signature packets)
    ss_off: np.ndarray            # uint64 [n_items+1]
    n_sigs: int                   # total signature packets
    mutation: np.ndarray          # uint8 [n_items]
    expected_valid: Optional[np.ndarray] = None   # int32 [n_items]: packets that must verify (duplicates counted)
    sig_count: Optional[np.ndarray] = None        # int32 [n_items]: packets in ss.Data
    requests: Optional[List[bytes]] = None   # full <x,v,t,sig,ss> packets (small corpora only)

    def tbss(self, i: int) -> bytes:
        return self.tbss_blob[int(self.tbss_off[i]):int(self.tbss_off[i + 1])].tobytes()

    def ss_data(self, i: int) -> bytes:
        return self.ss_blob[int(self.ss_off[i]):int(self.ss_off[i + 1])].tobytes()


BatchSigner = Callable[[np.ndarray, np.ndarray], np.ndarray]
"""(em[n,256] uint8 big-endian, key_index[n] int32 into cluster.replicas; index len(replicas) = the client)
-> sig[n,256] uint8."""


DsaBatchPow = Callable[[np.ndarray, List[int]], List[int]]
"""(key_index[n] int32 into cluster.replicas (DSA keys), nonces k[n]) -> [g_key^k mod p_key] as Python ints."""


def dsa_pow_tables(cluster: Cluster):
    """(moduli p [n,256], bases g [n,256]) big-endian uint8 per replica for a modexp-based DsaBatchPow (RSA replicas get
    the dummy modulus 3 / base 1 and are never indexed)."""
    ps = np.stack([np.frombuffer((k.p if k.algo == PK_DSA else 3).to_bytes(256, "big"), dtype=np.uint8) for k in cluster.replicas])
    gs = np.stack([np.frombuffer((k.g if k.algo == PK_DSA else 1).to_bytes(256, "big"), dtype=np.uint8) for k in cluster.replicas])
    return ps, gs


def signer_tables(cluster: Cluster):
    """(moduli[n+1,256], private exponents[n+1,256]) big-endian uint8 for a modexp-based BatchSigner: rows are the
    replicas in order, then the client (RSA keys only; DSA replicas get a dummy odd modulus and are never indexed)."""
    ks = cluster.replicas + [cluster.client]
    mods = np.stack([np.frombuffer((k.n if k.algo == PK_RSA else 3).to_bytes(256, "big"), dtype=np.uint8) for k in ks])
    exps = np.stack([np.frombuffer((k.d if k.algo == PK_RSA else 1).to_bytes(256, "big"), dtype=np.uint8) for k in ks])
    return mods, exps


def python_batch_signer(cluster: Cluster) -> BatchSigner:
    def sign(em: np.ndarray, key_index: np.ndarray) -> np.ndarray:
        out = np.empty_like(em)
        for i in range(em.shape[0]):
            ki = int(key_index[i])
            kp = cluster.replicas[ki] if ki < len(cluster.replicas) else cluster.client
            s = kp.rsa_private(int.from_bytes(em[i].tobytes(), "big"))
            out[i] = np.frombuffer(s.to_bytes(em.shape[1], "big"), dtype=np.uint8)
        return out
    return sign


def make_write_corpus(cluster: Cluster, n_items: int, *, seed: int = MASTER_SEED, value_len: int = 64,
                      min_sigs: Optional[int] = None, max_sigs: Optional[int] = None,
                      mutation_rates: Optional[Dict[int, float]] = None,
                      batch_signer: Optional[BatchSigner] = None, keep_requests: bool = False,
                      with_client_sig: bool = True, items: Optional[Sequence[Tuple[bytes, bytes, int]]] = None,
                      dsa_batch_pow: Optional["DsaBatchPow"] = None) -> WriteCorpus:
    """Signed writes as they reach Server.write (protocol/server.go:286-300): per item the TBSS
    payload ``Serialize(x,v,t,sig)`` and ``ss.Data`` = concatenated detached signatures of a
    shuffled subset of clique members over that payload (collectSignatures, client.go:125-170).
    ``items``: explicit (x, v, t) per write instead of key%08d / random value / t = 1 + i (read corpora store several
    versions of one variable).  ``dsa_batch_pow``: batched g^k mod p for the DSA signers (else Python ``pow`` per signature)."""
    if items is not None:
        n_items = len(items)
    rng = DRBG("writes", seed, cluster.n, n_items)
    nprng = np.random.default_rng(seed ^ (n_items * 2654435761 & 0xFFFFFFFF))
    n = cluster.n
    suff = cluster.suff
    min_sigs = suff if min_sigs is None else min_sigs
    max_sigs = n if max_sigs is None else max_sigs
    rates = {MUT_BAD_MPI: 0.01, MUT_UNKNOWN_ISSUER: 0.005, MUT_DUP_SIGNER: 0.005, MUT_ONE_SHORT: 0.01,
             MUT_BAD_TAG: 0.0} if mutation_rates is None else mutation_rates
    # mutation class per item
    u = nprng.random(n_items)
    mutation = np.zeros(n_items, dtype=np.uint8)
    acc = 0.0
    for m, r in sorted(rates.items()):
        mutation[(u >= acc) & (u < acc + r)] = m
        acc += r

    client_cert = cluster.client.entity
    tbss_parts: List[bytes] = []
    # per-signature work lists
    sig_item: List[int] = []
    sig_key: List[int] = []       # index into replicas, or -1-k for outsider k
    sig_prefix_l: List[bytes] = []
    digests: List[bytes] = []
    flip: List[int] = []          # 1 => corrupt one MPI byte after signing; 2 => corrupt hash tag
    per_item_counts = np.zeros(n_items, dtype=np.int64)
    expected_valid = np.zeros(n_items, dtype=np.int32)

    signer = batch_signer or python_batch_signer(cluster)
    # client signatures over tbs = Serialize(x,v,t) (client.go:127-131), signed in one batch
    tbs_list = []
    for i in range(n_items):
        if items is not None:
            x, v, t = items[i]
        else:
            x, v, t = b"key%08d" % i, nprng.bytes(value_len), 1 + i
        tbs_list.append(serialize_tbs(x, v, t))
    csigs: List[Optional[bytes]] = [None] * n_items
    if with_client_sig:
        cprefix = sig_prefix(0x00, cluster.client.algo, _hashed_area(cluster.client.key_id))
        csuffix = hash_suffix(cprefix)
        cdig = [hashlib.sha256(t + csuffix).digest() for t in tbs_list]
        cem = np.zeros((n_items, 256), dtype=np.uint8)
        for i in range(n_items):
            cem[i] = np.frombuffer(emsa(cdig[i], 256).to_bytes(256, "big"), dtype=np.uint8)
        csv = signer(cem, np.full(n_items, len(cluster.replicas), dtype=np.int32))
        for i in range(n_items):
            csigs[i] = make_sig_packet(cluster.client, cprefix, cdig[i], sig_value=int.from_bytes(csv[i].tobytes(), "big"))

    for i in range(n_items):
        tbs = tbs_list[i]
        if with_client_sig:
            tbss = tbs + sigpkt(csigs[i], client_cert)
        else:
            tbss = tbs + sigpkt(None, None)
        tbss_parts.append(tbss)
        k = int(nprng.integers(min_sigs, max_sigs + 1))
        order = nprng.permutation(n)[:k].tolist()
        mut = int(mutation[i])
        flips = [0] * len(order)
        if mut == MUT_ONE_SHORT:
            order = order[:suff]
            flips = [0] * len(order)
            flips[int(nprng.integers(0, len(order)))] = 1       # exactly suff-1 valid
        elif mut == MUT_BAD_MPI:
            flips[int(nprng.integers(0, len(order)))] = 1
        elif mut == MUT_BAD_TAG:
            flips[int(nprng.integers(0, len(order)))] = 2
        elif mut == MUT_UNKNOWN_ISSUER:
            pos = int(nprng.integers(0, len(order) + 1))
            order.insert(pos, -1 - int(nprng.integers(0, len(cluster.outsiders))))
            flips.insert(pos, 0)
        elif mut == MUT_DUP_SIGNER:
            pos = int(nprng.integers(0, len(order)))
            order.insert(int(nprng.integers(0, len(order) + 1)), order[pos])
            flips.append(0)
        mid = hashlib.sha256(tbss)
        for kidx, fl in zip(order, flips):
            kp = cluster.replicas[kidx] if kidx >= 0 else cluster.outsiders[-1 - kidx]
            prefix = sig_prefix(0x00, kp.algo, _hashed_area(kp.key_id))
            h = mid.copy()
            h.update(hash_suffix(prefix))
            sig_item.append(i)
            sig_key.append(kidx)
            sig_prefix_l.append(prefix)
            digests.append(h.digest())
            flip.append(fl)
        per_item_counts[i] = len(order)
        expected_valid[i] = sum(1 for kidx, fl in zip(order, flips) if kidx >= 0 and fl == 0)

    total = len(sig_item)
    # ---- sign: RSA in one batch, DSA and outsiders one by one
    packets: List[Optional[bytes]] = [None] * total
    rsa_rows = [j for j in range(total) if sig_key[j] >= 0 and cluster.replicas[sig_key[j]].algo == PK_RSA]
    if rsa_rows:
        em = np.zeros((len(rsa_rows), 256), dtype=np.uint8)
        for r, j in enumerate(rsa_rows):
            em[r] = np.frombuffer(emsa(digests[j], 256).to_bytes(256, "big"), dtype=np.uint8)
        kidx = np.array([sig_key[j] for j in rsa_rows], dtype=np.int32)
        sv = signer(em, kidx)
        for r, j in enumerate(rsa_rows):
            kp = cluster.replicas[sig_key[j]]
            packets[j] = make_sig_packet(kp, sig_prefix_l[j], digests[j],
                                         sig_value=int.from_bytes(sv[r].tobytes(), "big"))
    dsa_rows = [j for j in range(total) if sig_key[j] >= 0 and cluster.replicas[sig_key[j]].algo == PK_DSA]
    if dsa_rows and dsa_batch_pow is not None:
        # r = (g^k mod p) mod q for every DSA signature in one batch; s = k^-1 (z + x r) mod q on the host
        ks = []
        for j in dsa_rows:
            kp = cluster.replicas[sig_key[j]]
            ks.append(1 + rng.below(kp.q - 1))
        gk = dsa_batch_pow(np.array([sig_key[j] for j in dsa_rows], dtype=np.int32), ks)
        for j, k, r_full in zip(dsa_rows, ks, gk):
            kp = cluster.replicas[sig_key[j]]
            r = r_full % kp.q
            z = int.from_bytes(digests[j][:(kp.q.bit_length() + 7) // 8], "big")
            sv = pow(k, -1, kp.q) * (z + kp.x * r) % kp.q
            if r == 0 or sv == 0:
                continue                                   # (probability 2^-256) falls through to the per-signature path
            rb = r.to_bytes((r.bit_length() + 7) // 8, "big")
            sb_ = sv.to_bytes((sv.bit_length() + 7) // 8, "big")
            body = sig_prefix_l[j] + b"\x00\x00" + digests[j][:2] + go_mpi_bytes(rb) + go_mpi_bytes(sb_)
            packets[j] = _hdr(2, len(body)) + body
    for j in range(total):
        if packets[j] is None:
            kidx = sig_key[j]
            kp = cluster.replicas[kidx] if kidx >= 0 else cluster.outsiders[-1 - kidx]
            packets[j] = make_sig_packet(kp, sig_prefix_l[j], digests[j], rng)
        if flip[j] == 1:
            b = bytearray(packets[j])
            b[len(b) - 1 - int(nprng.integers(0, 20))] ^= 1 << int(nprng.integers(0, 8))
            packets[j] = bytes(b)
        elif flip[j] == 2:
            b = bytearray(packets[j])
            hdr = 3 if b[1] >= 192 else 2
            hl = (b[hdr + 4] << 8) | b[hdr + 5]
            b[hdr + 6 + hl + 2] ^= 0x40   # first hash-tag byte
            packets[j] = bytes(b)

    ss_parts: List[bytes] = []
    j = 0
    for i in range(n_items):
        c = int(per_item_counts[i])
        ss_parts.append(b"".join(packets[j:j + c]))
        j += c

    def cat(parts):
        off = np.zeros(len(parts) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(p) for p in parts], dtype=np.uint64)
        blob = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
        return blob, off
    tb, to = cat(tbss_parts)
    sb, so = cat(ss_parts)
    reqs = None
    if keep_requests:
        reqs = [tbss_parts[i] + sigpkt(ss_parts[i], None) for i in range(n_items)]
    return WriteCorpus(cluster, n_items, tb, to, sb, so, total, mutation, expected_valid=expected_valid,
                       sig_count=per_item_counts.astype(np.int32), requests=reqs)


# ------------------------------------------------------------------------------------------------
# signed-read corpus (BASELINE configs[2]): replies <x,v,t,sig,ss> of storage nodes to Client.Read
# ------------------------------------------------------------------------------------------------
@dataclass
class ReadCorpus:
    writes: WriteCorpus           # the DISTINCT stored packets (several versions per variable)
    n_vars: int
    storage_ids: List[int]        # node ids of the read quorum's members (storage nodes: they answer, they do not sign)
    reply_var: np.ndarray         # int32 [n_replies] variable of the reply
    reply_peer: np.ndarray        # uint64 [n_replies] responder id
    reply_write: np.ndarray       # int32 [n_replies] index into ``writes`` of the packet the responder returned
    write_t: np.ndarray           # uint64 [n_writes]
    write_value: List[bytes]      # value v of each stored packet
    tbss_blob: np.ndarray         # the reply batch as the verifier sees it: one copy of the stored packet per reply
    tbss_off: np.ndarray
    ss_blob: np.ndarray
    ss_off: np.ndarray
    n_sigs: int                   # signature packets in the reply batch


def make_read_corpus(cluster: Cluster, n_vars: int, *, n_storage: int = 10, seed: int = MASTER_SEED, value_len: int = 64,
                     p_stale: float = 0.15, p_conflict_var: float = 0.01, p_silent: float = 0.02,
                     mutation_rates: Optional[Dict[int, float]] = None, batch_signer: Optional[BatchSigner] = None,
                     dsa_batch_pow: Optional[DsaBatchPow] = None) -> ReadCorpus:
    """Read replies as Client.Read folds them (protocol/client.go:206-279): every storage node of the read quorum returns
    the packet <x,v,t,sig,ss> it stores for the variable.  Per variable two versions exist (t = 1, 2); a responder returns
    the latest one, a stale one with probability ``p_stale``, nothing with ``p_silent``; ``p_conflict_var`` of the variables
    carry a second, equally signed value at t = 2 that some responders return (the equivocation case of client.go:304-353).
    Each stored packet is collectively signed by a shuffled >= suff subset of the clique (as make_write_corpus)."""
    nprng = np.random.default_rng(seed ^ 0x5EAD)
    items: List[Tuple[bytes, bytes, int]] = []
    var_writes: List[List[int]] = []
    for j in range(n_vars):
        x = b"key%08d" % j
        ws = []
        for t in (1, 2):
            ws.append(len(items)); items.append((x, nprng.bytes(value_len), t))
        if nprng.random() < p_conflict_var:
            ws.append(len(items)); items.append((x, nprng.bytes(value_len), 2))
        var_writes.append(ws)
    w = make_write_corpus(cluster, len(items), seed=seed, mutation_rates=mutation_rates, batch_signer=batch_signer,
                          items=items, dsa_batch_pow=dsa_batch_pow)
    storage_ids = [0x5700000000000000 + k for k in range(n_storage)]
    rv, rp, rw = [], [], []
    for j in range(n_vars):
        ws = var_writes[j]
        for k in range(n_storage):
            u = nprng.random()
            if u < p_silent:
                continue
            if u < p_silent + p_stale:
                wi = ws[0]
            elif len(ws) == 3 and nprng.random() < 0.3:
                wi = ws[2]
            else:
                wi = ws[1]
            rv.append(j); rp.append(storage_ids[k]); rw.append(wi)
    rw_a = np.array(rw, dtype=np.int64)

    def gather(blob, off):
        lens = (off[1:] - off[:-1]).astype(np.int64)[rw_a]
        out_off = np.zeros(len(rw_a) + 1, dtype=np.uint64)
        out_off[1:] = np.cumsum(lens, dtype=np.uint64)
        out = np.empty(int(out_off[-1]), dtype=np.uint8)
        for r in range(len(rw_a)):
            a = int(off[rw_a[r]])
            out[int(out_off[r]):int(out_off[r + 1])] = blob[a:a + int(lens[r])]
        return out, out_off
    tb, to = gather(w.tbss_blob, w.tbss_off)
    sb, so = gather(w.ss_blob, w.ss_off)
    return ReadCorpus(w, n_vars, storage_ids, np.array(rv, dtype=np.int32), np.array(rp, dtype=np.uint64), rw_a.astype(np.int32),
                      np.array([it[2] for it in items], dtype=np.uint64), [it[1] for it in items], tb, to, sb, so,
                      int(w.sig_count[rw_a].sum()))


# ------------------------------------------------------------------------------------------------
# threshold share-combine corpus (BASELINE configs[4]): inputs of the combine step of each scheme
# ------------------------------------------------------------------------------------------------
@dataclass
class ThresholdCorpus:
    n_ops: int
    rsa_n: int                    # modulus of crypto/threshold/rsa/test.pkcs8
    rsa_factors: List[List[int]]  # [n_ops][10] partial signatures m^d_i mod N (rsa.go:318-329 multiplies them)
    sss_mod: int                  # the 2048-bit prime pb of crypto/sss/sss_test.go:15-47
    sss_xs: np.ndarray            # int32 [n_ops][7]
    sss_ys: List[List[int]]
    dsa_p: int
    dsa_q: int
    s_xs: np.ndarray              # int32 [n_ops][8]  calculateS shares (dsa_core.go:389-403)
    s_ys: List[List[int]]
    r_xs: np.ndarray              # int32 [n_ops][8]  CalculateR partial r's (dsa.go:33-52)
    r_ri: List[List[int]]
    r_vi: List[List[int]]


def make_threshold_corpus(n_ops: int, rsa_n: int, sss_mod: int, dsa_p: int, dsa_q: int, seed: int = MASTER_SEED,
                          n_shares: int = 10, rsa_k: int = 10, sss_k: int = 7, dsa_2t: int = 8) -> ThresholdCorpus:
    """Seeded inputs of the four combine operations at the reference's parameters (n = 10 shares; RSA: all 10 fragments'
    partial signatures; SSS: k = 7; threshold DSA: 2t = 8).  Values are uniform residues: the combine arithmetic does not
    care whether they came from a real dealing (tests/test_gpu_threshold.py covers real dealings against the reference's
    known answers)."""
    nprng = np.random.default_rng(seed ^ 0x7455)

    def residues(count, mod):
        nb = (mod.bit_length() + 7) // 8 + 8
        raw = nprng.bytes(count * nb)
        return [int.from_bytes(raw[i * nb:(i + 1) * nb], "big") % mod for i in range(count)]

    def xs(k):
        return np.ascontiguousarray(np.stack([nprng.permutation(n_shares)[:k] + 1 for _ in range(n_ops)]).astype(np.int32))

    def rows(vals, k):
        return [vals[i * k:(i + 1) * k] for i in range(n_ops)]
    return ThresholdCorpus(n_ops, rsa_n, rows(residues(n_ops * rsa_k, rsa_n), rsa_k), sss_mod, xs(sss_k),
                           rows(residues(n_ops * sss_k, sss_mod), sss_k), dsa_p, dsa_q, xs(dsa_2t),
                           rows(residues(n_ops * dsa_2t, dsa_q), dsa_2t), xs(dsa_2t),
                           rows([1 + v for v in residues(n_ops * dsa_2t, dsa_p - 1)], dsa_2t),
                           rows(residues(n_ops * dsa_2t, dsa_q), dsa_2t))


# ------------------------------------------------------------------------------------------------
# Transport messages (crypto_pgp.go:419-451): the signed packet sequence openpgp.Encrypt wraps in the
# encrypted container -- one-pass signature, literal data (binary, file name = base64(nonce), time 0),
# signature over the literal body.
# ------------------------------------------------------------------------------------------------
def one_pass_packet(sig_type: int, hash_id: int, pk_algo: int, key_id: int, is_last: bool = True) -> bytes:
    body = bytes([3, sig_type, hash_id, pk_algo]) + struct.pack(">Q", key_id) + bytes([1 if is_last else 0])
    return _hdr(4, len(body)) + body


def go_partial_chunks(n: int) -> List[int]:
    """Powers of two x/crypto's partialLengthWriter cuts one Write of n bytes into (largest power <= 2^14 that fits)."""
    out = []
    while n > 0:
        for power in range(14, -1, -1):
            if n >= (1 << power):
                out.append(power)
                n -= 1 << power
                break
    return out


def literal_packet(file_name: bytes, body: bytes, partial: Optional[List[int]] = None, is_binary: bool = True, time: int = 0) -> bytes:
    """Literal data packet; ``partial`` = powers of two of the leading partial-length chunks, the rest closes the packet."""
    content = (b"b" if is_binary else b"t") + bytes([len(file_name)]) + file_name + struct.pack(">I", time) + body
    if not partial:
        return _hdr(11, len(content)) + content
    out = bytearray([0xC0 | 11])
    p = 0
    for power in partial:
        ln = 1 << power
        assert p + ln <= len(content)
        out += bytes([224 + power]) + content[p:p + ln]
        p += ln
    out += _hdr(11, len(content) - p)[1:] + content[p:]
    return bytes(out)


def signed_message(kp: KeyPair, plain: bytes, nonce: bytes, rng: Optional[DRBG] = None, shape: str = "go") -> bytes:
    """shape "go": literal data as x/crypto's partialLengthWriter frames it (one run of power-of-two partial chunks per
    Write: 2 header bytes, the file name, 4 time bytes, the body; Close() ends with a zero-length chunk) -- restated from
    memory; shape "definite": one definite-length literal packet."""
    import base64
    fname = base64.standard_b64encode(nonce)
    ops = one_pass_packet(0x00, HASH_SHA256, kp.algo, kp.key_id)
    if shape == "definite":
        lit = literal_packet(fname, plain)
    else:
        chunks = []
        for part in (2, len(fname), 4, len(plain)):
            chunks += go_partial_chunks(part)
        lit = literal_packet(fname, plain, partial=chunks)
    return ops + lit + detach_sign(kp, plain, rng)
