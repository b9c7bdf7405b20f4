"""Oracle restatement of bftkv's signature / collective-signature semantics and the read tally.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference:
  PGPKeyring.getKeyring / getCertById        crypto/pgp/crypto_pgp.go:195-219
  PGPSignature.Verify                        crypto/pgp/crypto_pgp.go:319-330
  PGPSignature.VerifyWithCertificate         crypto/pgp/crypto_pgp.go:332-344
  PGPSignature.Signers                       crypto/pgp/crypto_pgp.go:373-390
  PGPCollectiveSignature.Verify              crypto/pgp/crypto_pgp.go:485-500
  PGPCollectiveSignature.Combine / Signers   crypto/pgp/crypto_pgp.go:506-519
  isThreshold / maxTimestampedValue          protocol/client.go:181-205
  error identities                           crypto/crypto.go:16-33
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

from . import openpgp as pgp
from .packet import SignaturePacket, SignatureTypeNil
from .wotqs import WotQ

ErrInvalidSignature = "crypto: invalid signature"                                   # crypto.go:20
ErrInsufficientNumberOfSignatures = "crypto: insufficient number of signatures"     # crypto.go:19


@dataclass
class Keyring:  # crypto_pgp.go:115-119
    keyring: List[pgp.Entity] = field(default_factory=list)
    secring: List[pgp.Entity] = field(default_factory=list)

    def get_keyring(self) -> List[pgp.Entity]:  # :195-197  secring first
        return list(self.secring) + list(self.keyring)

    def get_cert_by_id(self, key_id: int) -> Optional[pgp.Entity]:  # :206-219 primary ids only
        for e in self.keyring:
            if e.primary.key_id == key_id:
                return e
        for e in self.secring:
            if e.primary.key_id == key_id:
                return e
        return None


def signature_verify(kr: Keyring, tbs: bytes, sig: SignaturePacket, trace: Optional[list] = None) -> Optional[str]:
    """PGPSignature.Verify: every CheckDetachedSignature call must succeed; zero calls => invalid."""
    return _verify_all(kr.get_keyring(), tbs, sig, trace)


def signature_verify_with_certificate(tbs: bytes, sig: SignaturePacket, cert: pgp.Entity,
                                      trace: Optional[list] = None) -> Optional[str]:
    """PGPSignature.VerifyWithCertificate: keyring = the single entity of the certificate."""
    return _verify_all([cert], tbs, sig, trace)


def _verify_all(keyring, tbs, sig, trace):
    data = sig.Data or b""
    pos = 0
    err: Optional[str] = ErrInvalidSignature
    while len(data) - pos > 0:
        r = pgp.check_detached_signature(keyring, tbs, data, pos)
        pos = r.pos
        if trace is not None:
            trace.extend(r.statuses)
        if r.status != pgp.ST_OK:
            return ErrInvalidSignature
        err = None
    return err


def signers(kr: Keyring, sig: SignaturePacket) -> List[int]:
    """PGPSignature.Signers: parse-only walk; ids of issuers present in the keyring.  Any error
    from Reader.Next ends the walk.  (A v4 signature without issuer subpacket nil-panics in the
    reference, SURVEY.md D.8 -- fenced: it ends the walk here.)"""
    out: List[int] = []
    data = sig.Data or b""
    pos = 0
    while True:
        pkt = pgp.packet_read_stream(data, pos)
        pos = pkt.pos
        if pkt.kind in ("eof", "error", "sig_error"):
            break     # any error of Reader.Next (io.EOF included) ends the walk
        if pkt.kind != "sig":
            continue  # unknown types are skipped by Next; other known types fall through the type switch
        s = pkt.sig
        if s.version < 4:
            continue  # SignatureV3 is a different Go type: not matched by the switch
        if s.issuer is None:
            break
        e = kr.get_cert_by_id(s.issuer)
        if e is not None:
            out.append(e.id)
    return out


@dataclass
class CollectiveResult:
    err: Optional[str]
    completed: bool
    verified: List[int]          # signer ids in order of verification, up to the early exit
    n_calls: int                 # CheckDetachedSignature calls made
    statuses: List[int]          # per-packet statuses of everything consumed


def collective_verify(kr: Keyring, tbs: bytes, ss: SignaturePacket, q: WotQ) -> CollectiveResult:
    """PGPCollectiveSignature.Verify (crypto_pgp.go:485-500).  Mutates ss.Completed on success."""
    data = ss.Data or b""
    keyring = kr.get_keyring()
    verified: List[int] = []
    statuses: List[int] = []
    pos = 0
    calls = 0
    while len(data) - pos > 0:
        r = pgp.check_detached_signature(keyring, tbs, data, pos)
        pos = r.pos
        calls += 1
        statuses.extend(r.statuses)
        if r.status == pgp.ST_OK:
            verified.append(r.signer.id)
            if q.is_sufficient(verified):
                ss.Completed = True
                return CollectiveResult(None, True, verified, calls, statuses)
    return CollectiveResult(ErrInsufficientNumberOfSignatures, False, verified, calls, statuses)


def collective_combine(kr: Keyring, ss: SignaturePacket, s: SignaturePacket, q: WotQ) -> bool:
    """PGPCollectiveSignature.Combine (crypto_pgp.go:506-515)."""
    if ss.Type == SignatureTypeNil:
        ss.Type = s.Type
    elif ss.Type != s.Type:
        return False
    ss.Data = (ss.Data or b"") + (s.Data or b"")
    return q.is_sufficient(signers(kr, ss))


def max_timestamped_value(replies: Sequence[Tuple[int, int, bytes]], q: WotQ):
    """protocol/client.go:181-205 over replies (peer_id, t, value) in arrival order.
    Returns (value, t) or None for errInProgress.  Go iterates maps in unspecified order; when more
    than one value at max t reaches the threshold the reference's answer is order-dependent -- the
    corpora never produce that, and this restatement returns the first in arrival order."""
    m: Dict[int, Dict[bytes, List[int]]] = {}
    for peer, t, val in replies:
        m.setdefault(t, {}).setdefault(val or b"", []).append(peer)
    if not m:
        return None
    maxt = max(m.keys())
    for v, peers in m[maxt].items():
        if q.is_threshold(peers):
            return v, maxt
    return None
