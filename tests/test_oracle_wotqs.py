"""CPU: quorum arithmetic / predicates / graph cliques restatement (quorum/wotqs, node/graph) against the
tables of SURVEY.md Appendix C and the reference's own graph checkers (node/graph/graph_test.go:144-179)."""
import itertools

import pytest

from oracle import wotqs as W


@pytest.mark.parametrize("n,f,mn,thr,thr_rc,suff", [
    (4, 1, 4, 3, 2, 3), (6, 1, 4, 3, 2, 4), (10, 3, 10, 7, 4, 7), (64, 21, 64, 43, 22, 43), (256, 85, 256, 171, 86, 171)])
def test_new_qc_table(n, f, mn, thr, thr_rc, suff):
    nodes = list(range(100, 100 + n))
    qc = W.new_qc(nodes, n, W.AUTH, 0)
    assert (qc.f, qc.min, qc.threshold, qc.suff) == (f, mn, thr, suff)
    assert W.new_qc(nodes, n, W.AUTH | W.CERT, 0).threshold == thr_rc
    assert W.new_qc(nodes, n, W.READ, 0).threshold == thr_rc
    assert W.new_qc(nodes, n - suff, W.AUTH, 0).suff == 0       # Weight <= n - suff  => suff forced to 0 (wotqs.go:63-65)
    assert W.new_qc(nodes, n - suff + 1, W.AUTH, 0).suff == suff
    w = W.new_qc(nodes, n, W.WRITE, 0)                          # rw == WRITE exactly => all-zero qc (:52-54)
    assert (w.f, w.min, w.threshold, w.suff) == (0, 0, 0, 0)


def test_new_qc_small_and_peer():
    assert W.new_qc([1, 2, 3], 3, W.AUTH, 0) is None            # f < 1
    assert W.new_qc([], 0, W.AUTH, 0) is None
    assert W.new_qc([1, 2, 3, 4], 4, W.AUTH | W.PEER, 4) is None  # PEER drops self: n=3 => f=0
    q = W.new_qc([1, 2, 3, 4, 5], 5, W.AUTH | W.PEER, 5)
    assert q.nodes == [1, 2, 3, 4] and q.f == 1


def test_predicates_multiset_and_empty_quorum():
    q = W.WotQ([W.new_qc(list(range(4)), 4, W.AUTH, 99)])
    assert q.is_sufficient([0, 1, 2]) and not q.is_sufficient([0, 1])
    assert q.is_sufficient([0, 0, 0])                            # duplicates count repeatedly
    assert not q.is_sufficient([7, 8, 9, 10])
    assert q.is_threshold([0, 1, 2]) and not q.is_threshold([0, 1, 9])
    assert q.is_quorum([0, 1, 2, 3]) and not q.is_quorum([0, 1, 2])
    assert q.reject([0, 1]) and not q.reject([0])                # > f
    assert q.get_threshold() == 3
    e = W.WotQ([])
    assert not e.is_quorum([1]) and not e.is_threshold([1]) and not e.is_sufficient([1]) and e.reject([1])
    # OR over cliques for sufficiency, AND for threshold
    two = W.WotQ([W.new_qc(list(range(4)), 4, W.AUTH, 99), W.new_qc(list(range(10, 14)), 4, W.AUTH, 99)])
    assert two.is_sufficient([0, 1, 2]) and not two.is_threshold([0, 1, 2]) and two.is_threshold([0, 1, 2, 10, 11, 12])
    assert not two.reject([0, 1]) and two.reject([0, 1, 10, 11])
    zero = W.WotQ([W.QC([1, 2], 0, 0, 0, 0)])
    assert not zero.reject([1, 2]) and zero.is_threshold([]) and zero.is_quorum([])


def _cluster_graph(n_clique=10, n_rw=6, n_users=1):
    """scripts/setup.sh shape: a01..a10 a clique; clients certified by a07..a10; clients trust a01..a06 + rw*."""
    g = W.Graph()
    a = list(range(1, n_clique + 1))
    rw = list(range(101, 101 + n_rw))
    u = list(range(201, 201 + n_users))
    nodes = [(i, [j for j in a if j != i]) for i in a]            # everyone in the clique certifies everyone
    nodes += [(r, u) for r in rw]                                 # clients trust (sign) rw nodes
    nodes += [(x, a[-4:]) for x in u]                             # quorum certificate: a07..a10 certify the user
    g.add_nodes(nodes)
    for x in u:                                                   # clients trust a01..a06
        for i in a[:6]:
            g.vertices[x].edges[i] = g.vertices[i]
    return g, a, rw, u


def test_cliques_are_cliques_maximal_and_unique():
    g, a, rw, u = _cluster_graph()
    g.set_self([a[0]])
    cl = g.get_cliques(a[0], 1)
    assert len(cl) == 1 and sorted(cl[0].nodes) == a
    assert cl[0].weight == len(a) - 1                             # self certified the 9 other members
    # graph_test.go:144-179 checkers
    for c in cl:
        for x, y in itertools.permutations(c.nodes, 2):
            assert y in g.vertices[x].edges                       # clique-ness
        for v in g.vertices.values():
            if v.has_instance and v.id not in c.nodes:
                assert not all(v.id in g.vertices[m].edges and m in v.edges for m in c.nodes)   # maximality
    assert g.get_reachable_nodes(a[0], 0) == [a[0]]
    assert sorted(g.get_reachable_nodes(a[0], 1)) == a            # edges point signer -> signee (graph.go:61-71)
    assert sorted(g.get_reachable_nodes(a[0], 2)) == sorted(a + u)   # a07..a10 certified the client
    assert sorted(g.get_reachable_nodes(a[0], -1)) == sorted(a + u + rw)


def test_choose_quorum_call_sites():
    g, a, rw, u = _cluster_graph()
    # a server (clique member)
    g.set_self([a[2]])
    qs = W.Wot(g)
    q = qs.choose_quorum(W.AUTH)                                  # Server.write (server.go:300)
    assert len(q.qcs) == 1 and (q.qcs[0].f, q.qcs[0].threshold, q.qcs[0].suff) == (3, 7, 7)
    q = qs.choose_quorum(W.AUTH | W.CERT)                         # Server.sign quorum-cert check (server.go:211)
    assert q.qcs[0].threshold == 4
    # a client: weight = 6 certifications into the clique; suff = 7 > n - weight? 6 <= 10-7 is false => suff kept
    g2, a, rw, u = _cluster_graph()
    g2.set_self([u[0]])
    qc = W.Wot(g2).choose_quorum(W.AUTH | W.PEER)                 # collectSignatures (client.go:141)
    assert len(qc.qcs) == 1 and sorted(qc.qcs[0].nodes) == a and qc.qcs[0].suff == 7
    qw = W.Wot(g2).choose_quorum(W.WRITE)                         # writeWithTimestamp (client.go:101)
    assert all(q.suff == 0 for q in qw.qcs)
    assert any(q.f == 0 and q.threshold == 0 for q in qw.qcs) or len(qw.qcs) >= 1
    qr = W.Wot(g2).choose_quorum(W.READ)                          # Client.Read (client.go:238): R only
    assert all(set(q.nodes).isdisjoint(a) for q in qr.qcs)
    # R = reachable - cliques; GetReachableNodes includes the start vertex (graph.go:279-295), so the client
    # itself sits in R: n = 7, f = 2, threshold (READ rule) = f+1 = 3
    assert qr.qcs and sorted(qr.qcs[0].nodes) == sorted(rw + u) and (qr.qcs[0].f, qr.qcs[0].threshold, qr.qcs[0].suff) == (2, 3, 0)


def test_revoke_removes_vertex_and_edges():
    g, a, rw, u = _cluster_graph()
    g.set_self([a[0]])
    g.revoke(a[5])
    assert a[5] not in g.vertices and all(a[5] not in v.edges for v in g.vertices.values())
    assert g.add_nodes([(a[5], [])]) == []                        # revoked ids are refused (graph.go:49-51)
    assert sorted(g.get_cliques(a[0], 1)[0].nodes) == [i for i in a if i != a[5]]
