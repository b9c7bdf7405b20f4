"""Oracle restatement of the web-of-trust quorum system.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference:
  quorum flag bits                      quorum/quorum.go:10-16
  graph.AddNodes (edge direction)       node/graph/graph.go:46-75
  graph.GetPeers                        node/graph/graph.go:117-125
  graph.GetReachableNodes / bfs         node/graph/graph.go:279-295, 420-438
  graph.GetCliques / findMaximalClique / bidirect / inClique / putWeight
                                        node/graph/graph.go:297-393
  wot.newQC                             quorum/wotqs/wotqs.go:36-70
  wot.complement                        quorum/wotqs/wotqs.go:72-93
  wot.getQuorumFrom / ChooseQuorum      quorum/wotqs/wotqs.go:95-127
  wotq.IsQuorum/IsThreshold/IsSufficient/Reject/GetThreshold
                                        quorum/wotqs/wotqs.go:144-193
  intersection (multiset-preserving)    quorum/wotqs/wotqs.go:195-206
Nodes are represented by their 64-bit ids (node.Node.Id()); a node is "present" in the graph when
its vertex has an instance.  Go map iteration order is unspecified; where the reference's result
depends on it (findMaximalClique's greedy growth, the order of nodes inside a clique) the corpora
use disjoint, complete cliques so the *set* is order-independent, and this restatement iterates
vertices in insertion order.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence

READ, WRITE, AUTH, CERT, PEER = 0x01, 0x02, 0x04, 0x08, 0x10  # quorum.go:10-16


@dataclass
class Vertex:
    id: int
    has_instance: bool = False
    edges: Dict[int, "Vertex"] = field(default_factory=dict)  # signer -> signee


@dataclass
class Clique:
    nodes: List[int]
    weight: int = 0


class Graph:
    def __init__(self):
        self.vertices: Dict[int, Vertex] = {}
        self.revoked: Dict[int, bool] = {}
        self.self_ids: List[int] = []

    def add_nodes(self, nodes: Iterable[tuple]) -> List[int]:
        """nodes: iterable of (id, signer_ids).  graph.go:46-75."""
        res = []
        for skid, signers in nodes:
            if skid in self.revoked:
                continue
            v = self.vertices.get(skid)
            if v is None:
                v = Vertex(skid, True)
                self.vertices[skid] = v
            else:
                v.has_instance = True
            for signer in signers:
                if signer in self.revoked:
                    continue
                sv = self.vertices.get(signer)
                if sv is None:
                    sv = Vertex(signer, False)
                    self.vertices[signer] = sv
                sv.edges[skid] = v
            res.append(skid)
        return res

    def set_self(self, ids: Sequence[int]):  # graph.go:77-88 (graph assumed built)
        for i in ids:
            if i not in self.vertices or not self.vertices[i].has_instance:
                self.add_nodes([(i, [])])
            self.self_ids.append(i)

    def get_self_id(self) -> int:
        return self.self_ids[0] if self.self_ids else 0

    def remove_nodes(self, ids: Iterable[int]):  # graph.go:90-107
        for i in ids:
            for v in self.vertices.values():
                v.edges.pop(i, None)
            self.vertices.pop(i, None)
            if i in self.self_ids:
                self.self_ids.remove(i)

    def revoke(self, i: int):  # graph.go:131-140
        if i in self.vertices:
            self.remove_nodes([i])
        self.revoked[i] = True

    def get_peers(self) -> List[int]:  # graph.go:117-125
        me = self.get_self_id()
        return [v.id for v in self.vertices.values() if v.has_instance and v.id != me]

    def _bfs(self, v: Vertex, proc):  # graph.go:420-438
        seen = {v.id}
        q = [(v, 0)]
        while q:
            vd = q.pop(0)
            if proc(vd):
                return
            for i, e in vd[0].edges.items():
                if i not in seen:
                    q.append((e, vd[1] + 1))
                    seen.add(i)

    def get_reachable_nodes(self, sid: int, distance: int) -> List[int]:  # graph.go:279-295
        nodes: List[int] = []
        v = self.vertices.get(sid)
        if v is None:
            return nodes

        def proc(vd):
            if distance >= 0 and vd[1] > distance:
                return True
            if vd[0].has_instance:
                nodes.append(vd[0].id)
            return False
        self._bfs(v, proc)
        return nodes

    @staticmethod
    def _bidirect(v: Vertex, clique: List[Vertex]) -> bool:  # graph.go:364-374
        for c in clique:
            if v.id not in c.edges:
                return False
            if c.id not in v.edges:
                return False
        return True

    def _find_maximal_clique(self, s: Vertex) -> Optional[Clique]:  # graph.go:333-362
        clique = [s]
        for v in self.vertices.values():
            if not v.has_instance or v is s:
                continue
            if self._bidirect(v, clique):
                clique.append(v)
        for v in self.vertices.values():
            if v.has_instance and v is not s and v not in clique and self._bidirect(v, [s]):
                return None
        return Clique([c.id for c in clique])

    def get_cliques(self, sid: int, distance: int) -> List[Clique]:  # graph.go:297-320
        cliques: List[Clique] = []
        v = self.vertices.get(sid)
        if v is None or not v.has_instance:
            return cliques

        def proc(vd):
            if distance >= 0 and vd[1] > distance:
                return True
            if vd[0].has_instance:
                if not any(vd[0].id in c.nodes for c in cliques):  # inClique :322-331
                    c = self._find_maximal_clique(vd[0])
                    if c is not None:
                        for i in v.edges:  # putWeight :385-393
                            if i in c.nodes:
                                c.weight += 1
                        cliques.append(c)
            return False
        self._bfs(v, proc)
        return cliques


@dataclass
class QC:  # wotqs.go:16-22
    nodes: List[int]
    f: int
    min: int
    threshold: int
    suff: int


def new_qc(nodes_in: List[int], weight: int, rw: int, self_id: int) -> Optional[QC]:  # wotqs.go:36-70
    if rw & PEER:
        nodes = [n for n in nodes_in if n != self_id]
    else:
        nodes = list(nodes_in)
    n = len(nodes)
    if n == 0:
        return None
    if rw == WRITE:
        return QC(nodes, 0, 0, 0, 0)
    f = (n - 1) // 3
    if f >= 1:
        mn = 3 * f + 1
        threshold = 2 * f + 1
        suff = f + (n - f) // 2 + 1
        if rw & (CERT | READ):
            threshold = f + 1
        if weight <= n - suff:
            suff = 0
        return QC(nodes, f, mn, threshold, suff)
    return None


def intersection(s1: Sequence[int], s2: Sequence[int]) -> List[int]:  # wotqs.go:195-206
    ret = []
    for n1 in s1:
        for n2 in s2:
            if n1 == n2:
                ret.append(n1)
                break
    return ret


class WotQ:  # wotqs.go:24-26, 132-193
    def __init__(self, qcs: Optional[List[QC]] = None):
        self.qcs: List[QC] = qcs or []

    def nodes(self) -> List[int]:  # :132-142 (Active()/Address() filters are transport concerns)
        return [n for qc in self.qcs for n in qc.nodes]

    def is_quorum(self, nodes: Sequence[int]) -> bool:  # :144-154
        if not self.qcs:
            return False
        for qc in self.qcs:
            if qc.f > 0 and len(intersection(nodes, qc.nodes)) < qc.min:
                return False
        return True

    def is_threshold(self, nodes: Sequence[int]) -> bool:  # :156-166
        if not self.qcs:
            return False
        for qc in self.qcs:
            if qc.threshold > 0 and len(intersection(nodes, qc.nodes)) < qc.threshold:
                return False
        return True

    def is_sufficient(self, nodes: Sequence[int]) -> bool:  # :168-175
        for qc in self.qcs:
            if qc.suff > 0 and len(intersection(nodes, qc.nodes)) >= qc.suff:
                return True
        return False

    def reject(self, nodes: Sequence[int]) -> bool:  # :177-184
        for qc in self.qcs:
            if qc.f == 0 or len(intersection(nodes, qc.nodes)) <= qc.f:
                return False
        return True

    def get_threshold(self) -> int:  # :186-192
        return sum(qc.threshold for qc in self.qcs)


class Wot:  # wotqs.go:12-14, 72-127
    def __init__(self, g: Graph):
        self.g = g

    def _complement(self, u: List[int], c: List[QC], e: List[QC], rw: int) -> List[QC]:  # :72-93
        nodes = [n1 for n1 in u if not any(n1 in qc.nodes for qc in c)]
        q = new_qc(nodes, 0, rw, self.g.get_self_id())
        if q is not None:
            e = e + [q]
        return e

    def _get_quorum_from(self, rw: int, s: int, distance: int) -> WotQ:  # :95-115
        q: List[QC] = []
        for c in self.g.get_cliques(s, distance):
            qc = new_qc(c.nodes, c.weight, rw | AUTH, self.g.get_self_id())
            if qc is not None:
                q.append(qc)
        if rw & (READ | WRITE):
            qcs = list(q)
            if rw & AUTH == 0:
                qcs = []
            qcs = self._complement(self.g.get_reachable_nodes(s, distance), q, qcs, READ)
            if rw & WRITE:
                qcs = self._complement(self.g.get_peers(), q + qcs, qcs, WRITE)
            q = qcs
        return WotQ(q)

    def choose_quorum(self, rw: int) -> WotQ:  # :117-127
        if rw & CERT:
            distance = 0
        elif rw & AUTH:
            distance = 1
        else:
            distance = 2
        return self._get_quorum_from(rw, self.g.get_self_id(), distance)
