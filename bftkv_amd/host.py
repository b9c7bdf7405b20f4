"""Python face of the host-side mirror (include/bftkv_host.h): the reference's names over the C++
implementation inside libbftkv_gpu.so.  No oracle import; the packet / graph / quorum parts need no GPU.

  packet.Serialize / Parse / TBS / TBSS          packet/packet.go
  Graph.AddNodes / SetSelfNodes / Revoke / GetCliques / GetReachableNodes    node/graph/graph.go
  wotqs.New(g).ChooseQuorum(rw) -> Quorum.IsQuorum / IsThreshold / IsSufficient / Reject / GetThreshold
  Client.collect_signatures, Client.max_timestamped_value, Server.write_verify   protocol/{client,server}.go
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from ._native import QC as _QC

READ, WRITE, AUTH, CERT, PEER = 0x01, 0x02, 0x04, 0x08, 0x10   # quorum/quorum.go:10-16


class _SigPkt(C.Structure):
    _fields_ = [("type", C.c_uint8), ("version", C.c_uint32), ("completed", C.c_uint8),
                ("data", C.c_void_p), ("data_len", C.c_uint64), ("cert", C.c_void_p), ("cert_len", C.c_uint64)]


class _Parsed(C.Structure):
    _fields_ = [("x_off", C.c_uint64), ("x_len", C.c_uint64), ("v_off", C.c_uint64), ("v_len", C.c_uint64), ("t", C.c_uint64),
                ("has_sig", C.c_int), ("has_ss", C.c_int), ("sig", _SigPkt), ("ss", _SigPkt),
                ("auth_off", C.c_uint64), ("auth_len", C.c_uint64)]


class _Reply(C.Structure):
    _fields_ = [("peer_id", C.c_uint64), ("err", C.c_int32), ("data", C.c_void_p), ("data_len", C.c_uint64)]


class SigParse(C.Structure):
    """bftkv_sig_parse (include/bftkv_host.h): what the kernels' signature parser keeps of one packet body."""
    _fields_ = [("parsed", C.c_uint8), ("too_deep", C.c_uint8), ("version", C.c_uint8), ("sig_type", C.c_uint8), ("pk_algo", C.c_uint8),
                ("hash_id", C.c_uint8), ("have_issuer", C.c_uint8), ("n_mpi", C.c_uint8), ("hash_tag", C.c_uint8 * 2),
                ("hashed_len", C.c_uint16), ("mpi_bits", C.c_uint16 * 2), ("mpi_off", C.c_uint32 * 2), ("issuer", C.c_uint64)]


def split_tails(blob, off, tails: Sequence[bytes]):
    """Payloads that end in one of the byte strings `tails` (TBSS ends in chunk(sig.Cert), packet/packet.go:192-212: the signer's
    certificate) -> the arguments of Context.collective_verify_segments: (prefix_blob, prefix_off, shared_blob, shared_off,
    seg_of_item).  A payload that ends in none of them stays whole (segment 0xFFFFFFFF); the longest matching tail wins."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64).astype(np.int64)
    n = len(off) - 1
    lens = off[1:] - off[:-1]
    seg = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    cut = np.zeros(n, dtype=np.int64)
    order = sorted(range(len(tails)), key=lambda k: -len(tails[k]))
    for k in order:
        t = np.frombuffer(tails[k], dtype=np.uint8)
        L = len(t)
        if L == 0:
            continue
        cand = np.nonzero((lens >= L) & (seg == 0xFFFFFFFF))[0]
        if cand.size == 0:
            continue
        # compare the last L bytes of every candidate with the tail, a slab at a time
        for lo in range(0, cand.size, 1024):
            c = cand[lo:lo + 1024]
            idx = (off[c + 1] - L)[:, None] + np.arange(L, dtype=np.int64)[None, :]
            hit = (blob[idx] == t[None, :]).all(axis=1)
            seg[c[hit]] = k
            cut[c[hit]] = L
    keep = lens - cut
    prefix_off = np.zeros(n + 1, dtype=np.uint64)
    prefix_off[1:] = np.cumsum(keep, dtype=np.uint64)
    mask = np.ones(blob.size, dtype=bool)
    hit = np.nonzero(cut)[0]
    if hit.size:
        # drop [end - cut, end) of every matched payload
        starts = off[hit + 1] - cut[hit]
        delta = np.zeros(blob.size + 1, dtype=np.int32)
        np.add.at(delta, starts, 1)
        np.add.at(delta, off[hit + 1], -1)
        mask = np.cumsum(delta[:-1]) == 0
    prefix_blob = blob[mask]
    shared_off = np.zeros(len(tails) + 1, dtype=np.uint64)
    shared_off[1:] = np.cumsum([len(t) for t in tails], dtype=np.uint64)
    shared_blob = np.frombuffer(b"".join(tails) or b"\0", dtype=np.uint8)[:int(shared_off[-1])].copy()
    return prefix_blob, prefix_off, shared_blob, shared_off, seg


def parse_signature(body: bytes) -> SigParse:
    """bftkv_host_parse_signature: the verifier's Signature.parse / SignatureV3.parse on one packet body, on the host."""
    out = SigParse()
    rc = _lib().bftkv_host_parse_signature(body, len(body), C.byref(out))
    if rc:
        raise RuntimeError("parse_signature: %d" % rc)
    return out


def scan_stream(data: bytes, cap: int = 4096):
    """bftkv_host_scan_stream: (statuses of the packet events, fence flag) of one signature stream, by the kernels' walk + parse."""
    st = np.zeros(cap, dtype=np.uint8)
    n, f = C.c_uint32(0), C.c_uint8(0)
    rc = _lib().bftkv_host_scan_stream(data, len(data), cap, st.ctypes.data, C.byref(n), C.byref(f))
    if rc:
        raise RuntimeError("scan_stream: %d" % rc)
    return [int(x) for x in st[:min(n.value, cap)]], n.value, bool(f.value)


def walk_stream(data: bytes, cap: int = 4096):
    """bftkv_host_walk_stream: (status, body offset, body length) of every packet event of one stream, by the kernels' walk."""
    st = np.zeros(cap, dtype=np.uint8); bo = np.zeros(cap, dtype=np.uint64); bl = np.zeros(cap, dtype=np.uint32)
    n = C.c_uint32(0)
    rc = _lib().bftkv_host_walk_stream(data, len(data), cap, st.ctypes.data, bo.ctypes.data, bl.ctypes.data, C.byref(n))
    if rc:
        raise RuntimeError("walk_stream: %d" % rc)
    k = min(int(n.value), cap)
    return [(int(st[i]), int(bo[i]), int(bl[i])) for i in range(k)], int(n.value)


HOST_EXPORTS = [
    "bftkv_host_packet_serialize", "bftkv_host_packet_parse", "bftkv_host_packet_tbs", "bftkv_host_packet_tbss",
    "bftkv_host_graph_new", "bftkv_host_graph_free", "bftkv_host_graph_add_node", "bftkv_host_graph_set_self",
    "bftkv_host_graph_revoke", "bftkv_host_graph_reachable", "bftkv_host_graph_cliques", "bftkv_host_choose_quorum",
    "bftkv_host_quorum_from_qcs", "bftkv_host_quorum_free", "bftkv_host_quorum_n_qcs", "bftkv_host_quorum_qc",
    "bftkv_host_quorum_is_quorum", "bftkv_host_quorum_is_threshold", "bftkv_host_quorum_is_sufficient", "bftkv_host_quorum_reject",
    "bftkv_host_quorum_get_threshold", "bftkv_host_quorum_gpu_handle", "bftkv_host_collect_signatures",
    "bftkv_host_server_write_verify", "bftkv_host_max_timestamped_value", "bftkv_host_max_timestamped_value_masked", "bftkv_host_vote_fold", "bftkv_host_certs_parse",
    "bftkv_host_certs_free", "bftkv_host_certs_n_entities", "bftkv_host_certs_entity", "bftkv_host_certs_key", "bftkv_host_certs_structure", "bftkv_host_certs_check", "bftkv_host_certs_roles", "bftkv_host_signers_walk",
    "bftkv_host_server_sign_verify", "bftkv_host_server_read_proof_verify", "bftkv_host_server_register_verify", "bftkv_host_equivocation_signers", "bftkv_host_emsa_encode", "bftkv_host_certs_verify",
    "bftkv_host_quorum_cert_verify", "bftkv_host_graph_set_caching", "bftkv_host_graph_cache_stats", "bftkv_host_message_frame",
    "bftkv_host_parse_signature", "bftkv_host_walk_stream", "bftkv_host_scan_stream", "bftkv_host_sha256", "bftkv_host_cert_fingerprint",
]

_ready = False


def _lib():
    global _ready
    lib = _native.load_library()
    if not _ready:
        vp = C.c_void_p
        lib.bftkv_host_graph_new.restype = vp
        lib.bftkv_host_message_frame.argtypes = [C.c_char_p, C.c_uint64, vp, vp, vp, vp, C.c_uint64, vp, vp, vp, vp, vp]
        lib.bftkv_host_choose_quorum.restype = vp
        lib.bftkv_host_choose_quorum.argtypes = [vp, C.c_int]
        lib.bftkv_host_quorum_from_qcs.restype = vp
        lib.bftkv_host_quorum_from_qcs.argtypes = [C.POINTER(_QC), C.c_uint32]
        lib.bftkv_host_graph_free.argtypes = [vp]
        lib.bftkv_host_graph_free.restype = None
        lib.bftkv_host_quorum_free.argtypes = [vp]
        lib.bftkv_host_quorum_free.restype = None
        lib.bftkv_host_graph_add_node.argtypes = [vp, C.c_uint64, vp, C.c_uint32]
        lib.bftkv_host_graph_set_self.argtypes = [vp, C.c_uint64]
        lib.bftkv_host_graph_revoke.argtypes = [vp, C.c_uint64]
        lib.bftkv_host_graph_reachable.argtypes = [vp, C.c_uint64, C.c_int, vp, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.bftkv_host_graph_cliques.argtypes = [vp, C.c_uint64, C.c_int, vp, C.c_uint32, vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.bftkv_host_quorum_n_qcs.argtypes = [vp]
        lib.bftkv_host_quorum_n_qcs.restype = C.c_uint32
        lib.bftkv_host_quorum_qc.argtypes = [vp, C.c_uint32, C.POINTER(_QC)]
        for n in ("is_quorum", "is_threshold", "is_sufficient", "reject"):
            getattr(lib, "bftkv_host_quorum_" + n).argtypes = [vp, vp, C.c_uint32]
        lib.bftkv_host_quorum_get_threshold.argtypes = [vp]
        lib.bftkv_host_quorum_gpu_handle.argtypes = [vp, vp, C.POINTER(C.c_int)]
        lib.bftkv_host_packet_serialize.argtypes = [C.c_int, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint64,
                                                    C.POINTER(_SigPkt), C.POINTER(_SigPkt), C.c_char_p, C.c_uint64,
                                                    vp, C.c_uint64, C.POINTER(C.c_uint64)]
        lib.bftkv_host_packet_parse.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(_Parsed)]
        lib.bftkv_host_packet_tbs.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
        lib.bftkv_host_packet_tbss.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
        lib.bftkv_host_collect_signatures.argtypes = [vp, vp, C.c_uint32, vp, vp, C.POINTER(_Reply), vp, vp, C.c_uint64, vp, vp, vp]
        lib.bftkv_host_server_write_verify.argtypes = [vp, vp, C.c_uint32, vp, vp, vp]
        lib.bftkv_host_max_timestamped_value.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, vp]
        lib.bftkv_host_max_timestamped_value_masked.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp]
        lib.bftkv_host_parse_signature.argtypes = [C.c_char_p, C.c_uint32, vp]
        lib.bftkv_host_walk_stream.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, vp, vp, vp, vp]
        lib.bftkv_host_scan_stream.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, vp, vp, vp]
        lib.bftkv_host_vote_fold.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp]
        lib.bftkv_host_certs_parse.restype = vp
        lib.bftkv_host_certs_parse.argtypes = [C.c_char_p, C.c_uint64]
        lib.bftkv_host_certs_free.argtypes = [vp]
        lib.bftkv_host_certs_free.restype = None
        lib.bftkv_host_certs_n_entities.argtypes = [vp]
        lib.bftkv_host_certs_n_entities.restype = C.c_uint32
        lib.bftkv_host_certs_entity.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(vp), C.POINTER(C.c_uint32)]
        lib.bftkv_host_certs_key.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(_native.PubKey)]
        lib.bftkv_host_certs_roles.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(vp), C.POINTER(C.c_uint32)]
        lib.bftkv_host_certs_structure.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32)]
        lib.bftkv_host_certs_check.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(vp), C.POINTER(C.c_uint64),
                                               C.POINTER(vp), C.POINTER(C.c_uint64)]
        lib.bftkv_host_server_sign_verify.argtypes = [vp, vp, C.c_uint32, vp, vp, vp]
        lib.bftkv_host_server_read_proof_verify.argtypes = [vp, vp, C.c_uint32, vp, vp, vp]
        lib.bftkv_host_server_register_verify.argtypes = [vp, vp, C.c_uint32, vp, vp, vp]
        lib.bftkv_host_equivocation_signers.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.bftkv_host_certs_verify.argtypes = [vp, C.c_char_p, C.c_uint64, vp, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.bftkv_host_quorum_cert_verify.argtypes = [vp, vp, C.c_char_p, C.c_uint64, vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.bftkv_host_emsa_encode.argtypes = [C.c_int, C.c_char_p, C.c_uint32, C.c_uint32, vp, C.c_uint32]
        _ready = True
    return lib


class MalformedPacket(ValueError):
    pass


@dataclass
class SignaturePacket:   # packet/packet.go:25-31
    Type: int = 0
    Version: int = 0
    Completed: bool = False
    Data: Optional[bytes] = None
    Cert: Optional[bytes] = None


def _c_sig(s: Optional[SignaturePacket], keep: list):
    if s is None:
        return None
    c = _SigPkt()
    c.type, c.version, c.completed = s.Type, s.Version, 1 if s.Completed else 0
    for name in ("data", "cert"):
        b = getattr(s, name.capitalize()) or b""
        buf = C.create_string_buffer(b, len(b)) if b else None
        keep.append(buf)
        setattr(c, name, C.cast(buf, C.c_void_p) if buf is not None else None)
        setattr(c, name + "_len", len(b))
    return C.pointer(c)


class packet:
    """packet/packet.go"""

    @staticmethod
    def Serialize(*args) -> bytes:
        lib = _lib()
        n = len(args)
        a = list(args) + [None] * (6 - n)
        keep: list = []
        x, v, t = a[0] or b"", a[1] or b"", a[2] or 0
        auth = a[5] or b""
        out_len = C.c_uint64(0)
        cap = 64 + len(x) + len(v) + len(auth) + sum(len(getattr(s, f) or b"") for s in (a[3], a[4]) if s for f in ("Data", "Cert")) + 64
        buf = C.create_string_buffer(cap)
        rc = lib.bftkv_host_packet_serialize(n, x, len(x), v, len(v), t, _c_sig(a[3], keep), _c_sig(a[4], keep), auth, len(auth),
                                             C.cast(buf, C.c_void_p), cap, C.byref(out_len))
        if rc:
            raise ValueError("Serialize failed: %d" % rc)
        return buf.raw[:out_len.value]

    @staticmethod
    def Parse(pkt: bytes):
        p = _Parsed()
        if _lib().bftkv_host_packet_parse(pkt, len(pkt), C.byref(p)):
            raise MalformedPacket("Parse")

        def sl(off, ln):
            return pkt[off:off + ln] if ln else None

        def sg(has, s):
            if not has:
                return None
            data = C.string_at(s.data, s.data_len) if s.data_len else None
            cert = C.string_at(s.cert, s.cert_len) if s.cert_len else None
            return SignaturePacket(s.type, s.version, bool(s.completed), data, cert)
        return sl(p.x_off, p.x_len), sl(p.v_off, p.v_len), p.t, sg(p.has_sig, p.sig), sg(p.has_ss, p.ss), sl(p.auth_off, p.auth_len)

    @staticmethod
    def TBS(pkt: bytes) -> bytes:
        n = C.c_uint64(0)
        if _lib().bftkv_host_packet_tbs(pkt, len(pkt), C.byref(n)):
            raise MalformedPacket("TBS")
        return pkt[:n.value]

    @staticmethod
    def TBSS(pkt: bytes) -> bytes:
        n = C.c_uint64(0)
        if _lib().bftkv_host_packet_tbss(pkt, len(pkt), C.byref(n)):
            raise MalformedPacket("TBSS")
        return pkt[:n.value]


class Quorum:
    """quorum.Quorum (quorum/quorum.go:18-25) backed by wotq (quorum/wotqs/wotqs.go:24-26)."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("no quorum")
        self.h = C.c_void_p(handle)

    def __del__(self):
        try:
            _lib().bftkv_host_quorum_free(self.h)
        except Exception:
            pass

    @classmethod
    def from_qcs(cls, qcs):
        arr = (_QC * max(1, len(qcs)))()
        keep = []
        for i, (f, mn, thr, suff, nodes) in enumerate(qcs):
            ids = np.ascontiguousarray(np.array(list(nodes), dtype=np.uint64))
            keep.append(ids)
            arr[i].f, arr[i].min, arr[i].threshold, arr[i].suff = f, mn, thr, suff
            arr[i].node_ids = ids.ctypes.data if len(ids) else None
            arr[i].n_nodes = len(ids)
        return cls(_lib().bftkv_host_quorum_from_qcs(arr, len(qcs)))

    def qcs(self) -> List[Tuple[int, int, int, int, List[int]]]:
        out = []
        for i in range(_lib().bftkv_host_quorum_n_qcs(self.h)):
            q = _QC()
            _lib().bftkv_host_quorum_qc(self.h, i, C.byref(q))
            ids = list(np.ctypeslib.as_array((C.c_uint64 * q.n_nodes).from_address(q.node_ids))) if q.n_nodes else []
            out.append((q.f, q.min, q.threshold, q.suff, [int(x) for x in ids]))
        return out

    def _pred(self, name, nodes) -> bool:
        ids = np.ascontiguousarray(np.array(list(nodes), dtype=np.uint64))
        return bool(getattr(_lib(), "bftkv_host_quorum_" + name)(self.h, ids.ctypes.data if len(ids) else None, len(ids)))

    def Nodes(self) -> List[int]:
        return [n for q in self.qcs() for n in q[4]]

    def IsQuorum(self, nodes) -> bool:
        return self._pred("is_quorum", nodes)

    def IsThreshold(self, nodes) -> bool:
        return self._pred("is_threshold", nodes)

    def IsSufficient(self, nodes) -> bool:
        return self._pred("is_sufficient", nodes)

    def Reject(self, nodes) -> bool:
        return self._pred("reject", nodes)

    def GetThreshold(self) -> int:
        return _lib().bftkv_host_quorum_get_threshold(self.h)


class Graph:
    """node/graph/graph.go"""

    def __init__(self):
        self.h = C.c_void_p(_lib().bftkv_host_graph_new())

    def __del__(self):
        try:
            _lib().bftkv_host_graph_free(self.h)
        except Exception:
            pass

    def AddNodes(self, nodes):
        """nodes: iterable of (id, [ids of the keys that certified it])."""
        for i, signers in nodes:
            s = np.ascontiguousarray(np.array(list(signers), dtype=np.uint64))
            _lib().bftkv_host_graph_add_node(self.h, i, s.ctypes.data if len(s) else None, len(s))

    def SetSelfNodes(self, ids):
        for i in ids:
            _lib().bftkv_host_graph_set_self(self.h, i)

    def Revoke(self, i: int):
        _lib().bftkv_host_graph_revoke(self.h, i)

    def set_caching(self, on: bool) -> None:
        """Selector cache per graph epoch (on by default); off = the reference's recompute-per-call cost."""
        _lib().bftkv_host_graph_set_caching(self.h, 1 if on else 0)

    def cache_stats(self):
        e, h, m = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _lib().bftkv_host_graph_cache_stats(self.h, C.byref(e), C.byref(h), C.byref(m))
        return {"epoch": e.value, "hits": h.value, "misses": m.value}

    def GetReachableNodes(self, sid: int, distance: int) -> List[int]:
        n = C.c_uint32(0)
        _lib().bftkv_host_graph_reachable(self.h, sid, distance, None, 0, C.byref(n))
        out = np.zeros(max(1, n.value), dtype=np.uint64)
        _lib().bftkv_host_graph_reachable(self.h, sid, distance, out.ctypes.data, len(out), C.byref(n))
        return [int(x) for x in out[:n.value]]

    def GetCliques(self, sid: int, distance: int):
        ids = np.zeros(1 << 16, dtype=np.uint64)
        sizes = np.zeros(256, dtype=np.uint32)
        weights = np.zeros(256, dtype=np.int32)
        n = C.c_uint32(0)
        rc = _lib().bftkv_host_graph_cliques(self.h, sid, distance, ids.ctypes.data, len(ids), sizes.ctypes.data, weights.ctypes.data,
                                             len(sizes), C.byref(n))
        if rc:
            raise RuntimeError("GetCliques: %d" % rc)
        out, k = [], 0
        for i in range(n.value):
            out.append(([int(x) for x in ids[k:k + sizes[i]]], int(weights[i])))
            k += int(sizes[i])
        return out


class wotqs:
    """quorum/wotqs/wotqs.go"""

    def __init__(self, g: Graph):
        self.g = g

    @classmethod
    def New(cls, g: Graph):
        return cls(g)

    def ChooseQuorum(self, rw: int) -> Quorum:
        return Quorum(_lib().bftkv_host_choose_quorum(self.g.h, rw))


@dataclass
class Reply:   # transport.MulticastResponse (transport/transport.go:30-34)
    Peer: int
    Data: Optional[bytes] = None
    Err: int = 0


def _cat(parts: Sequence[bytes]):
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts], dtype=np.uint64)
    blob = np.frombuffer(b"".join(parts) + b"\0", dtype=np.uint8)[:int(off[-1])].copy()
    if blob.size == 0:
        blob = np.zeros(1, dtype=np.uint8)
    return blob, off


ROLE_NAMES = ["ignored", "primary", "uid", "self", "ident_sig", "subkey", "subkey_sig", "revocation"]      # BFTKV_ROLE_* (bftkv_gpu.h)


class Certificate:
    """PGPCertificate.Parse / Signers reduced to what the path reads (crypto_pgp.go:236-272, 80-88)."""

    @staticmethod
    def Parse(cert: bytes):
        """The certificate as openpgp.ReadEntity walks it -> list of entities (refused ones included; entity 0 is a request's
        issuer): dict(id, keys=[dict(key_id, pk_algo, usable_sign, n, e, g, y)], certifiers=[ids of Signers()], refused, unknown,
        why, checks=[dict(kind, key_index, signed, sig)] -- the signatures ReadEntity verifies (kind 2: third-party certifications,
        which it does not))."""
        lib = _lib()
        h = C.c_void_p(lib.bftkv_host_certs_parse(cert, len(cert)))
        out = []
        try:
            for e in range(lib.bftkv_host_certs_n_entities(h)):
                eid, nk, cp, nc = C.c_uint64(0), C.c_uint32(0), C.c_void_p(), C.c_uint32(0)
                lib.bftkv_host_certs_entity(h, e, C.byref(eid), C.byref(nk), C.byref(cp), C.byref(nc))
                certifiers = [int(x) for x in (C.c_uint64 * nc.value).from_address(cp.value)] if nc.value else []
                keys = []
                for k in range(nk.value):
                    pk = _native.PubKey()
                    lib.bftkv_host_certs_key(h, e, k, C.byref(pk))
                    g = lambda p, l: C.string_at(p, l) if l else b""
                    keys.append({"key_id": pk.key_id, "entity_id": pk.entity_id, "pk_algo": pk.pk_algo, "usable_sign": bool(pk.usable_sign),
                                 "n": g(pk.n, pk.n_len), "e": g(pk.e, pk.e_len), "g": g(pk.g, pk.g_len), "y": g(pk.y, pk.y_len)})
                refused, unknown, why, nchk = C.c_uint8(0), C.c_uint8(0), C.c_char_p(), C.c_uint32(0)
                lib.bftkv_host_certs_structure(h, e, C.byref(refused), C.byref(unknown), C.byref(why), C.byref(nchk))
                checks = []
                for i in range(nchk.value):
                    kind, ki, sp, sl, gp, gl = C.c_int(0), C.c_uint32(0), C.c_void_p(), C.c_uint64(0), C.c_void_p(), C.c_uint64(0)
                    lib.bftkv_host_certs_check(h, e, i, C.byref(kind), C.byref(ki), C.byref(sp), C.byref(sl), C.byref(gp), C.byref(gl))
                    checks.append({"kind": kind.value, "key_index": ki.value, "signed": C.string_at(sp.value, sl.value) if sl.value else b"",
                                   "sig": C.string_at(gp.value, gl.value) if gl.value else b""})
                st, ln, rp, nr = C.c_uint64(0), C.c_uint64(0), C.c_void_p(), C.c_uint32(0)
                lib.bftkv_host_certs_roles(h, e, C.byref(st), C.byref(ln), C.byref(rp), C.byref(nr))
                raw = [int(x) for x in (C.c_uint32 * nr.value).from_address(rp.value)] if nr.value else []
                out.append({"id": eid.value, "keys": keys, "certifiers": certifiers, "refused": bool(refused.value), "unknown": bool(unknown.value),
                            "why": (why.value or b"").decode(), "checks": checks, "start": st.value, "len": ln.value,
                            "roles": [(ROLE_NAMES[r & 0xFF], (r >> 8) & 0xFFFF, bool(r >> 24)) for r in raw]})
        finally:
            lib.bftkv_host_certs_free(h)
        return out


def signers_walk(ss: bytes):
    """bftkv_host_signers_walk: (issuer ids of the v4 signatures Signers() walks, in packet order, unfiltered; fenced)."""
    lib = _lib()
    lib.bftkv_host_signers_walk.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
    cap = len(ss) // 12 + 1
    ids = np.zeros(cap, dtype=np.uint64)
    n, fenced = C.c_uint32(0), C.c_uint8(0)
    rc = lib.bftkv_host_signers_walk(ss, len(ss), ids.ctypes.data, cap, C.byref(n), C.byref(fenced))
    if rc:
        raise _native.NativeError("signers_walk failed: %d" % rc)
    return [int(x) for x in ids[:n.value]], bool(fenced.value)


def cert_fingerprint(cert: bytes) -> Optional[bytes]:
    """v4 fingerprint of the primary key of the first entity (bftkv_host_cert_fingerprint), None when no key packet leads the
    certificate (packet types x/crypto skips may precede it)."""
    out = np.zeros(20, dtype=np.uint8)
    lib = _lib()
    lib.bftkv_host_cert_fingerprint.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p]
    return out.tobytes() if lib.bftkv_host_cert_fingerprint(cert, len(cert), out.ctypes.data) == 0 else None


def certs_verify(ctx: _native.Context, cert: bytes) -> List[Optional[bool]]:
    """Which entities of a certificate blob openpgp.ReadEntity returns: True, False (refused), None (no verdict: a shape left to
    the reference, or a check that met a fenced shape)."""
    # sized from the call: the entry point reports how many entities it found (BFTKV_E_NOMEM with *n_out set when the array is
    # too short), so a certificate of any number of entities is answered in at most two calls
    cap = 64
    while True:
        valid = np.zeros(cap, dtype=np.uint8)
        n = C.c_uint32(0)
        rc = _lib().bftkv_host_certs_verify(ctx.h, cert, len(cert), valid.ctypes.data, len(valid), C.byref(n))
        if rc == -3 and n.value > cap:          # BFTKV_E_NOMEM
            cap = n.value
            continue
        if rc:
            raise _native.NativeError("certs_verify failed: %d" % rc)
        return [None if v == 2 else bool(v) for v in valid[:n.value]]


def quorum_cert_verify(ctx: _native.Context, q: "Quorum", cert: bytes):
    """CheckQuorumCert of the paper: (IsThreshold over VERIFIED certifiers, their ids)."""
    ok = np.zeros(4, dtype=np.uint8)
    cap = 1024
    while True:
        ids = np.zeros(cap, dtype=np.uint64)
        n = C.c_uint32(0)
        rc = _lib().bftkv_host_quorum_cert_verify(ctx.h, q.h, cert, len(cert), ok.ctypes.data, ids.ctypes.data, len(ids), C.byref(n))
        if rc == -3 and n.value > cap:          # BFTKV_E_NOMEM: *n_out says how many verified certifiers there are
            cap = n.value
            continue
        if rc:
            raise _native.NativeError("quorum_cert_verify failed: %d" % rc)
        return bool(ok[0]), [int(x) for x in ids[:n.value]]


def sha256(data: bytes, mode: int = 0):
    """bftkv_host_sha256: the compression the micro-batcher's callers run over their payloads (mode 0: SHA extensions when the
    CPU has them, 1: portable).  Returns (digest, cpu_has_sha_extensions)."""
    lib = _lib()
    lib.bftkv_host_sha256.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_void_p]
    buf = C.create_string_buffer(32)
    rc = lib.bftkv_host_sha256(data, len(data), mode, C.cast(buf, C.c_void_p))
    if rc < 0:
        raise ValueError("bftkv_host_sha256: %d" % rc)
    return buf.raw, bool(rc)


def emsa_encode(hash_id: int, digest: bytes, n_bits: int) -> bytes:
    """emsaEncode (crypto/threshold/rsa/rsa.go:356-378)."""
    emlen = (n_bits + 7) // 8
    buf = C.create_string_buffer(emlen)
    if _lib().bftkv_host_emsa_encode(hash_id, digest, len(digest), n_bits, C.cast(buf, C.c_void_p), emlen):
        raise ValueError("crypto: invalid input")
    return buf.raw


class Client:
    """The vote-collecting half of protocol.Client (protocol/client.go) over a GPU context."""

    def __init__(self, ctx: _native.Context):
        self.ctx = ctx

    def collect_signatures(self, qa: Quorum, tbss_list: Sequence[bytes], replies: Sequence[Sequence[Reply]]):
        """client.go:139-169 for a batch of writes.  Returns (ss_data[w], consumed[w], err[w])."""
        n = len(tbss_list)
        tb, to = _cat(tbss_list)
        flat = [r for rs in replies for r in rs]
        arr = (_Reply * max(1, len(flat)))()
        keep = []
        for i, r in enumerate(flat):
            arr[i].peer_id, arr[i].err = r.Peer, r.Err
            b = r.Data or b""
            buf = C.create_string_buffer(b, len(b)) if b else None
            keep.append(buf)
            arr[i].data = C.cast(buf, C.c_void_p) if buf is not None else None
            arr[i].data_len = len(b)
        roff = np.zeros(n + 1, dtype=np.uint64)
        roff[1:] = np.cumsum([len(rs) for rs in replies], dtype=np.uint64)
        cap = sum(len(r.Data or b"") for r in flat) + 16
        ss = np.zeros(cap, dtype=np.uint8)
        ss_off = np.zeros(n + 1, dtype=np.uint64)
        consumed = np.zeros(n, dtype=np.uint32)
        err = np.zeros(n, dtype=np.uint8)
        rc = _lib().bftkv_host_collect_signatures(self.ctx.h, qa.h, n, tb.ctypes.data, to.ctypes.data, arr, roff.ctypes.data,
                                                  ss.ctypes.data, cap, ss_off.ctypes.data, consumed.ctypes.data, err.ctypes.data)
        if rc:
            raise _native.NativeError("collect_signatures failed: %d" % rc)
        data = [ss[int(ss_off[i]):int(ss_off[i + 1])].tobytes() for i in range(n)]
        return data, consumed, err

    @staticmethod
    def vote_fold(q: Quorum, rounds: Sequence[Sequence[Tuple[int, bool]]]):
        """client.go:67-86 / 108-123 for a batch of Multicast rounds of (peer, accepted) replies.
        Returns (consumed[i], is_threshold[i])."""
        flat = [r for rs in rounds for r in rs]
        peers = np.ascontiguousarray(np.array([r[0] for r in flat], dtype=np.uint64))
        ok = np.ascontiguousarray(np.array([1 if r[1] else 0 for r in flat], dtype=np.uint8))
        roff = np.zeros(len(rounds) + 1, dtype=np.uint64)
        roff[1:] = np.cumsum([len(rs) for rs in rounds], dtype=np.uint64)
        consumed = np.zeros(max(1, len(rounds)), dtype=np.uint32)
        thr = np.zeros(max(1, len(rounds)), dtype=np.uint8)
        p = lambda a: a.ctypes.data if a.size else None
        rc = _lib().bftkv_host_vote_fold(q.h, len(rounds), p(peers), p(ok), roff.ctypes.data, consumed.ctypes.data, thr.ctypes.data)
        if rc:
            raise RuntimeError("vote_fold: %d" % rc)
        return consumed[:len(rounds)], thr[:len(rounds)].astype(bool)

    def equivocation_signers(self, values: Sequence[Tuple[int, bytes]]) -> List[int]:
        """client.go:304-353 for one (variable, t): values = (value group, ss.Data) per stored reply."""
        grp = np.ascontiguousarray(np.array([v[0] for v in values], dtype=np.uint32))
        sb, so = _cat([v[1] or b"" for v in values])
        out = np.zeros(4096, dtype=np.uint64)
        n = C.c_uint32(0)
        rc = _lib().bftkv_host_equivocation_signers(self.ctx.h, len(values), grp.ctypes.data, sb.ctypes.data, so.ctypes.data, out.ctypes.data,
                                                    len(out), C.byref(n))
        if rc:
            raise _native.NativeError("equivocation_signers failed: %d" % rc)
        return [int(x) for x in out[:n.value]]

    @staticmethod
    def max_timestamped_value(q: Quorum, reads: Sequence[Sequence[Tuple[int, int, bytes]]]):
        """client.go:181-205 for a batch of variables; per read the winning (value, t) or None."""
        flat = [r for rs in reads for r in rs]
        peers = np.ascontiguousarray(np.array([r[0] for r in flat], dtype=np.uint64))
        ts = np.ascontiguousarray(np.array([r[1] for r in flat], dtype=np.uint64))
        vb, vo = _cat([r[2] or b"" for r in flat])
        roff = np.zeros(len(reads) + 1, dtype=np.uint64)
        roff[1:] = np.cumsum([len(rs) for rs in reads], dtype=np.uint64)
        out = np.zeros(max(1, len(reads)), dtype=np.int64)
        p = lambda a: a.ctypes.data if a.size else None
        rc = _lib().bftkv_host_max_timestamped_value(q.h, len(reads), p(peers), p(ts), vb.ctypes.data, vo.ctypes.data, roff.ctypes.data,
                                                     out.ctypes.data)
        if rc:
            raise RuntimeError("max_timestamped_value: %d" % rc)
        res = []
        for i, rs in enumerate(reads):
            res.append(None if out[i] < 0 else (rs[int(out[i])][2] or b"", rs[int(out[i])][1]))
        return res


def max_timestamped_value_raw(q: Quorum, n_reads: int, peers: np.ndarray, ts: np.ndarray, value_blob: np.ndarray, value_off: np.ndarray,
                              read_off: np.ndarray) -> np.ndarray:
    """bftkv_host_max_timestamped_value on flat arrays (batch callers): replies of read i are rows read_off[i]..read_off[i+1];
    returns per read the row index (relative to the read's first row... see below) of a reply carrying the winning <t, v>, or -1
    for errInProgress.  The index is relative to read_off[i]."""
    out = np.zeros(max(1, n_reads), dtype=np.int64)
    peers = np.ascontiguousarray(peers, dtype=np.uint64); ts = np.ascontiguousarray(ts, dtype=np.uint64)
    value_blob = np.ascontiguousarray(value_blob, dtype=np.uint8) if value_blob.size else np.zeros(1, dtype=np.uint8)
    value_off = np.ascontiguousarray(value_off, dtype=np.uint64); read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
    p = lambda a: a.ctypes.data if a.size else None
    rc = _lib().bftkv_host_max_timestamped_value(q.h, n_reads, p(peers), p(ts), value_blob.ctypes.data, value_off.ctypes.data,
                                                 read_off.ctypes.data, out.ctypes.data)
    if rc:
        raise RuntimeError("max_timestamped_value: %d" % rc)
    return out[:n_reads]


def max_timestamped_value_masked(q: Quorum, n_reads: int, peers: np.ndarray, ts: np.ndarray, value_blob: np.ndarray, value_off: np.ndarray,
                                 read_off: np.ndarray, reply_err: np.ndarray) -> np.ndarray:
    """bftkv_host_max_timestamped_value_masked: the fold over ALL replies with the verifier's error byte per reply (arrays
    prepared once, reply_err from each batch); same answers as max_timestamped_value_raw over the accepted replies."""
    out = np.zeros(max(1, n_reads), dtype=np.int64)
    for a, dt in ((peers, np.uint64), (ts, np.uint64), (value_blob, np.uint8), (value_off, np.uint64), (read_off, np.uint64), (reply_err, np.uint8)):
        assert a.dtype == dt and a.flags["C_CONTIGUOUS"]
    rc = _lib().bftkv_host_max_timestamped_value_masked(q.h, n_reads, peers.ctypes.data, ts.ctypes.data, value_blob.ctypes.data,
                                                        value_off.ctypes.data, read_off.ctypes.data, reply_err.ctypes.data, out.ctypes.data)
    if rc:
        raise RuntimeError("max_timestamped_value_masked: %d" % rc)
    return out[:n_reads]


class Server:
    """The verification site of protocol.Server.write (protocol/server.go:286-302) over a GPU context."""

    ErrMalformedRequest = 0xFF
    ErrCertificateNotFound = 0xFE
    ErrInvalidQuorumCertificate = 0xFD

    def sign_verify(self, q_cert: Quorum, requests: Sequence[bytes]) -> np.ndarray:
        """server.go:189-214 for a batch of Sign requests."""
        rb, ro = _cat(requests)
        err = np.zeros(len(requests), dtype=np.uint8)
        rc = _lib().bftkv_host_server_sign_verify(self.ctx.h, q_cert.h, len(requests), rb.ctypes.data, ro.ctypes.data, err.ctypes.data)
        if rc:
            raise _native.NativeError("server_sign_verify failed: %d" % rc)
        return err

    def __init__(self, ctx: _native.Context):
        self.ctx = ctx

    ErrAuthenticationFailure = 0xFB
    ErrFenced = 0xFC

    def _site(self, fn, name, q: Quorum, requests: Sequence[bytes]) -> np.ndarray:
        rb, ro = _cat(requests)
        err = np.zeros(len(requests), dtype=np.uint8)
        rc = fn(self.ctx.h, q.h, len(requests), rb.ctypes.data, ro.ctypes.data, err.ctypes.data)
        if rc:
            raise _native.NativeError("%s failed: %d" % (name, rc))
        return err

    def read_proof_verify(self, q_auth: Quorum, requests: Sequence[bytes]) -> np.ndarray:
        """server.go:181-185 for a batch of read requests carrying a proof."""
        return self._site(_lib().bftkv_host_server_read_proof_verify, "server_read_proof_verify", q_auth, requests)

    def register_verify(self, q_auth: Quorum, requests: Sequence[bytes]) -> np.ndarray:
        """server.go:452-475 for a batch of register requests."""
        return self._site(_lib().bftkv_host_server_register_verify, "server_register_verify", q_auth, requests)

    def write_verify(self, q: Quorum, requests: Sequence[bytes]) -> np.ndarray:
        rb, ro = _cat(requests)
        err = np.zeros(len(requests), dtype=np.uint8)
        rc = _lib().bftkv_host_server_write_verify(self.ctx.h, q.h, len(requests), rb.ctypes.data, ro.ctypes.data, err.ctypes.data)
        if rc:
            raise _native.NativeError("server_write_verify failed: %d" % rc)
        return err


def message_frame(msg: bytes):
    """Framing half of the transport message check (bftkv_host_message_frame, no GPU): returns a dict with
    framing (BFTKV_MSG_* or 0xFF = trailing signature found), signer, hash_id, plain, file_name, sig (the trailing packet)."""
    lib = _lib()
    st = C.c_uint8(0)
    signer = C.c_uint64(0)
    hid = C.c_uint8(0)
    plain = np.zeros(max(1, len(msg)), dtype=np.uint8)
    plen = C.c_uint64(0)
    fn = np.zeros(256, dtype=np.uint8)
    fl = C.c_uint8(0)
    so, sl = C.c_uint64(0), C.c_uint64(0)
    rc = lib.bftkv_host_message_frame(msg, len(msg), C.byref(st), C.byref(signer), C.byref(hid), plain.ctypes.data, len(plain), C.byref(plen),
                                      fn.ctypes.data, C.byref(fl), C.byref(so), C.byref(sl))
    if rc:
        raise RuntimeError("bftkv_host_message_frame: %d" % rc)
    return {"framing": st.value, "signer": signer.value, "hash_id": hid.value, "plain": plain[:plen.value].tobytes(),
            "file_name": fn[:fl.value].tobytes(), "sig": msg[so.value:so.value + sl.value]}
