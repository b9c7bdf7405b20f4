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
        cands = [(e.primary, e.flags_valid, e.flag_sign, e.self_sig_revoked)] + [(k, fv, fs, False) for k, fv, fs in e.subkeys]
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
