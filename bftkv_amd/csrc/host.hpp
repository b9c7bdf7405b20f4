// Host-side mirror of the reference code on either side of the GPU path (include/bftkv_host.h).
// Plain C++17, no device code.  Names follow the reference (packet / graph / wotqs / client).
#pragma once
#include <stdint.h>
#include <string.h>

#include <deque>
#include <string>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/bftkv_host.h"

namespace bftkv {
namespace host {

// ---------------------------------------------------------------------------------------------
// packet/packet.go
// ---------------------------------------------------------------------------------------------
inline void put_u64(std::string& o, uint64_t v) { for (int i = 7; i >= 0; --i) o.push_back((char)(v >> (8 * i))); }
inline void put_u32(std::string& o, uint32_t v) { for (int i = 3; i >= 0; --i) o.push_back((char)(v >> (8 * i))); }
inline void write_chunk(std::string& o, const uint8_t* p, uint64_t n) {   // WriteChunk, packet.go:117-124
  put_u64(o, n);
  if (n) o.append((const char*)p, n);
}
inline void write_signature(std::string& o, const bftkv_sigpkt* s) {      // writeSignature, packet.go:192-212
  bftkv_sigpkt z;
  memset(&z, 0, sizeof z);
  if (!s) s = &z;
  o.push_back((char)s->type);
  put_u32(o, s->version);
  o.push_back(s->completed ? 1 : 0);
  write_chunk(o, s->data, s->data_len);
  write_chunk(o, s->cert, s->cert_len);
}

// Reader with the io.EOF / io.ErrUnexpectedEOF distinction Parse relies on.
struct Reader {
  const uint8_t* p; uint64_t len, pos;
  enum { OK = 0, END = 1, SHORT = 2 };
  int take(uint64_t n, const uint8_t** out) {
    if (n == 0) { *out = p + pos; return OK; }
    if (pos >= len) return END;
    if (len - pos < n) { pos = len; return SHORT; }
    *out = p + pos; pos += n; return OK;
  }
  int u64(uint64_t* v) {
    const uint8_t* b; int r = take(8, &b);
    if (r) return r;
    *v = 0; for (int i = 0; i < 8; ++i) *v = (*v << 8) | b[i];
    return OK;
  }
  int chunk(const uint8_t** d, uint64_t* n) {   // ReadChunk, packet.go:126-140 (length 0 reads back as nil)
    uint64_t l; int r = u64(&l);
    if (r) return r;
    *n = l; *d = nullptr;
    if (l == 0) return OK;
    return take(l, d);
  }
  int signature(bftkv_sigpkt* s) {               // readSignature, packet.go:214-235
    const uint8_t* b; int r = take(1, &b);
    if (r) return r;
    s->type = b[0];
    if ((r = take(4, &b))) return r;
    s->version = ((uint32_t)b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3];
    if ((r = take(1, &b))) return r;
    s->completed = b[0] != 0;
    if ((r = chunk(&s->data, &s->data_len))) return r;
    if ((r = chunk(&s->cert, &s->cert_len))) return r;
    return OK;
  }
};

inline int packet_parse(const uint8_t* pkt, uint64_t len, bftkv_parsed* o) {   // Parse, packet.go:62-115
  memset(o, 0, sizeof *o);
  Reader r{pkt, len, 0};
  const uint8_t* d; uint64_t n; int rc;
  if ((rc = r.chunk(&d, &n))) return BFTKV_E_INVALID;          // the first field's error is returned as is
  o->x_off = n ? (uint64_t)(d - pkt) : 0; o->x_len = n;
  if ((rc = r.chunk(&d, &n))) return rc == Reader::END ? 0 : BFTKV_E_INVALID;
  o->v_off = n ? (uint64_t)(d - pkt) : 0; o->v_len = n;
  if ((rc = r.u64(&o->t))) { o->t = 0; return rc == Reader::END ? 0 : BFTKV_E_INVALID; }
  if ((rc = r.signature(&o->sig))) { memset(&o->sig, 0, sizeof o->sig); return rc == Reader::END ? 0 : BFTKV_E_INVALID; }
  o->has_sig = o->sig.type != 0;
  if ((rc = r.signature(&o->ss))) { memset(&o->ss, 0, sizeof o->ss); return rc == Reader::END ? 0 : BFTKV_E_INVALID; }
  o->has_ss = o->ss.type != 0;
  if ((rc = r.chunk(&d, &n))) return rc == Reader::END ? 0 : BFTKV_E_INVALID;
  o->auth_off = n ? (uint64_t)(d - pkt) : 0; o->auth_len = n;
  return 0;
}

// seek2tbs, packet.go:142-154, as it stands: the errors of binary.Read and of Seek are IGNORED there.  So a read that hits
// the end leaves `l` at its previous value -- and consumes the bytes that were there (io.ReadFull's short read); Seek moves by
// that stale `l`, may leave the position past the end (bytes.Reader allows it), and refuses only a negative target, leaving
// the position where it was.  What the callers then see: TBS / TBSS fail when the final position is past the end.
inline bool seek2tbs(const uint8_t* pkt, uint64_t len, uint64_t* off) {
  int64_t pos = 0, l = 0;
  const int64_t n = (int64_t)len;
  auto read8 = [&](bool keep) {
    if (pos >= n) return;                                   // io.EOF: nothing consumed, nothing stored
    if (pos + 8 > n) { pos = n; return; }                   // io.ErrUnexpectedEOF: the tail is consumed, nothing stored
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i) v = (v << 8) | pkt[pos + i];
    pos += 8;
    if (keep) l = (int64_t)v;
  };
  auto seek = [&] {
    const int64_t abs = (int64_t)((uint64_t)pos + (uint64_t)l);   // Go's int64 addition wraps
    if (abs >= 0) pos = abs;                                // "bytes.Reader.Seek: negative position": position unchanged
  };
  read8(true); seek();        // the variable
  read8(true); seek();        // the value
  read8(false);               // the timestamp
  *off = (uint64_t)pos;
  return pos <= n;
}
inline int packet_tbs(const uint8_t* pkt, uint64_t len, uint64_t* n) {          // TBS, packet.go:156-168
  return seek2tbs(pkt, len, n) ? 0 : BFTKV_E_INVALID;
}
inline int packet_tbss(const uint8_t* pkt, uint64_t len, uint64_t* n) {         // TBSS, packet.go:170-190
  uint64_t off;
  if (!seek2tbs(pkt, len, &off)) return BFTKV_E_INVALID;
  Reader r{pkt, len, off};
  bftkv_sigpkt s;
  if (r.signature(&s)) return BFTKV_E_INVALID;
  *n = r.pos;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// node/graph/graph.go  (vertices iterate in insertion order; Go's map order is unspecified)
// ---------------------------------------------------------------------------------------------
struct Vertex {
  uint64_t id;
  bool has_instance = false;
  std::vector<uint64_t> edge_order;               // signer -> signee
  std::unordered_set<uint64_t> edges;
  void add_edge(uint64_t to) { if (edges.insert(to).second) edge_order.push_back(to); }
  void del_edge(uint64_t to) {
    if (edges.erase(to)) for (size_t i = 0; i < edge_order.size(); ++i) if (edge_order[i] == to) { edge_order.erase(edge_order.begin() + i); break; }
  }
};

struct Clique { std::vector<uint64_t> nodes; int weight = 0; };

struct Graph {
  std::vector<uint64_t> order;                    // vertex insertion order
  std::unordered_map<uint64_t, Vertex> vertices;
  std::unordered_set<uint64_t> revoked;
  std::vector<uint64_t> self;
  // Selector cache: the reference recomputes the clique search on every ChooseQuorum; here every
  // mutation bumps `epoch` and GetCliques results are kept per (start, distance) until the epoch moves.
  uint64_t epoch = 0;
  bool caching = true;
  struct CliqueCacheEntry { uint64_t epoch; std::vector<Clique> cs; };
  std::map<std::pair<uint64_t, int>, CliqueCacheEntry> clique_cache;
  uint64_t cache_hits = 0, cache_misses = 0;

  Vertex* find(uint64_t id) { auto it = vertices.find(id); return it == vertices.end() ? nullptr : &it->second; }
  Vertex& get_or_add(uint64_t id, bool instance) {
    auto it = vertices.find(id);
    if (it == vertices.end()) { order.push_back(id); Vertex v; v.id = id; it = vertices.emplace(id, v).first; }
    if (instance) it->second.has_instance = true;
    return it->second;
  }
  bool add_node(uint64_t id, const uint64_t* signers, uint32_t n) {   // AddNodes, graph.go:46-75
    ++epoch;
    if (revoked.count(id)) return false;
    get_or_add(id, true);
    for (uint32_t i = 0; i < n; ++i) {
      if (revoked.count(signers[i])) continue;
      get_or_add(signers[i], false).add_edge(id);
    }
    return true;
  }
  void set_self(uint64_t id) {                                          // SetSelfNodes, graph.go:77-88
    ++epoch;
    Vertex* v = find(id);
    if (!v || !v->has_instance) add_node(id, nullptr, 0);
    self.push_back(id);
  }
  uint64_t self_id() const { return self.empty() ? 0 : self[0]; }
  void remove_node(uint64_t id) {                                       // RemoveNodes, graph.go:90-107
    ++epoch;
    for (auto& kv : vertices) kv.second.del_edge(id);
    if (vertices.erase(id)) for (size_t i = 0; i < order.size(); ++i) if (order[i] == id) { order.erase(order.begin() + i); break; }
    for (size_t i = 0; i < self.size(); ++i) if (self[i] == id) { self.erase(self.begin() + i); break; }
  }
  void revoke(uint64_t id) { ++epoch; if (find(id)) remove_node(id); revoked.insert(id); }   // Revoke, graph.go:131-140
  std::vector<uint64_t> peers() {                                       // GetPeers, graph.go:117-125
    std::vector<uint64_t> r; uint64_t me = self_id();
    for (uint64_t id : order) { Vertex& v = vertices[id]; if (v.has_instance && id != me) r.push_back(id); }
    return r;
  }
  template <typename F> void bfs(uint64_t sid, F proc) {               // bfs, graph.go:420-438
    std::unordered_set<uint64_t> seen{sid};
    std::deque<std::pair<uint64_t, int>> q{{sid, 0}};
    while (!q.empty()) {
      auto [id, d] = q.front(); q.pop_front();
      Vertex* v = find(id);
      if (!v) continue;
      if (proc(*v, d)) return;
      for (uint64_t to : v->edge_order) if (seen.insert(to).second) q.push_back({to, d + 1});
    }
  }
  std::vector<uint64_t> reachable(uint64_t sid, int distance) {         // GetReachableNodes, graph.go:279-295
    std::vector<uint64_t> r;
    if (!find(sid)) return r;
    bfs(sid, [&](Vertex& v, int d) { if (distance >= 0 && d > distance) return true; if (v.has_instance) r.push_back(v.id); return false; });
    return r;
  }
  bool bidirect(Vertex& v, const std::vector<Vertex*>& clique) {       // graph.go:364-374
    for (Vertex* c : clique) { if (!c->edges.count(v.id)) return false; if (!v.edges.count(c->id)) return false; }
    return true;
  }
  bool find_maximal_clique(Vertex& s, Clique* out) {                    // findMaximalClique, graph.go:333-362
    std::vector<Vertex*> clique{&s};
    for (uint64_t id : order) { Vertex& v = vertices[id]; if (!v.has_instance || &v == &s) continue; if (bidirect(v, clique)) clique.push_back(&v); }
    for (uint64_t id : order) {
      Vertex& v = vertices[id];
      if (!v.has_instance || &v == &s) continue;
      bool in = false;
      for (Vertex* c : clique) if (c == &v) { in = true; break; }
      if (!in && bidirect(v, {&s})) return false;                       // "found more than one maximal cliques"
    }
    out->nodes.clear(); out->weight = 0;
    for (Vertex* c : clique) out->nodes.push_back(c->id);
    return true;
  }
  const std::vector<Clique>& cliques_cached(uint64_t sid, int distance) {
    auto key = std::make_pair(sid, distance);
    auto it = clique_cache.find(key);
    if (caching && it != clique_cache.end() && it->second.epoch == epoch) { ++cache_hits; return it->second.cs; }
    ++cache_misses;
    CliqueCacheEntry& e = clique_cache[key];
    e.cs = cliques(sid, distance);
    e.epoch = epoch;
    return e.cs;
  }
  std::vector<Clique> cliques(uint64_t sid, int distance) {             // GetCliques, graph.go:297-320
    std::vector<Clique> cs;
    Vertex* s = find(sid);
    if (!s || !s->has_instance) return cs;
    bfs(sid, [&](Vertex& v, int d) {
      if (distance >= 0 && d > distance) return true;
      if (v.has_instance) {
        bool in = false;                                               // inClique, graph.go:322-331
        for (auto& c : cs) for (uint64_t n : c.nodes) if (n == v.id) in = true;
        if (!in) {
          Clique c;
          if (find_maximal_clique(v, &c)) {
            for (uint64_t e : s->edge_order) for (uint64_t n : c.nodes) if (n == e) ++c.weight;   // putWeight, graph.go:385-393
            cs.push_back(c);
          }
        }
      }
      return false;
    });
    return cs;
  }
};

// ---------------------------------------------------------------------------------------------
// quorum/wotqs/wotqs.go
// ---------------------------------------------------------------------------------------------
struct QC {
  std::vector<uint64_t> nodes;
  int f = 0, min = 0, threshold = 0, suff = 0;
  // membership index: intersection() emits every element of s1 found in s2 -- duplicates in s1 count again
  // (wotqs.go:195-206) -- so a tally is sum over s1 of member(id), O(|s1|) instead of O(|s1| * |s2|)
  mutable std::unordered_set<uint64_t> member;
  mutable bool indexed = false;
  bool has(uint64_t id) const {
    if (!indexed) { member.clear(); member.insert(nodes.begin(), nodes.end()); indexed = true; }
    return member.count(id) != 0;
  }
};

inline bool new_qc(const std::vector<uint64_t>& in, int weight, int rw, uint64_t self, QC* out) {   // newQC, wotqs.go:36-70
  out->nodes.clear();
  if (rw & BFTKV_Q_PEER) { for (uint64_t n : in) if (n != self) out->nodes.push_back(n); }
  else out->nodes = in;
  int n = (int)out->nodes.size();
  if (n == 0) return false;
  if (rw == BFTKV_Q_WRITE) { out->f = out->min = out->threshold = out->suff = 0; return true; }
  int f = (n - 1) / 3;
  if (f < 1) return false;
  out->f = f; out->min = 3 * f + 1; out->threshold = 2 * f + 1; out->suff = f + (n - f) / 2 + 1;
  if (rw & (BFTKV_Q_CERT | BFTKV_Q_READ)) out->threshold = f + 1;
  if (weight <= n - out->suff) out->suff = 0;
  return true;
}

inline int intersection_count(const uint64_t* s1, uint32_t n1, const QC& qc) {   // wotqs.go:195-206
  int c = 0;
  for (uint32_t i = 0; i < n1; ++i) c += qc.has(s1[i]);
  return c;
}

}  // namespace host
}  // namespace bftkv

struct bftkv_graph {
  bftkv::host::Graph g;
  // ChooseQuorum results per rw flag set, valid while g.epoch stands still
  struct QuorumCacheEntry { uint64_t epoch; std::vector<bftkv::host::QC> qcs; };
  std::map<int, QuorumCacheEntry> quorum_cache;
};

struct bftkv_quorum {
  std::vector<bftkv::host::QC> qcs;
  // cached GPU registration
  bftkv_gpu_ctx* ctx = nullptr;
  int handle = -1;

  bool is_quorum(const uint64_t* ids, uint32_t n) const {       // wotqs.go:144-154
    if (qcs.empty()) return false;
    for (auto& qc : qcs) if (qc.f > 0 && bftkv::host::intersection_count(ids, n, qc) < qc.min) return false;
    return true;
  }
  bool is_threshold(const uint64_t* ids, uint32_t n) const {    // wotqs.go:156-166
    if (qcs.empty()) return false;
    for (auto& qc : qcs) if (qc.threshold > 0 && bftkv::host::intersection_count(ids, n, qc) < qc.threshold) return false;
    return true;
  }
  bool is_sufficient(const uint64_t* ids, uint32_t n) const {   // wotqs.go:168-175
    for (auto& qc : qcs) if (qc.suff > 0 && bftkv::host::intersection_count(ids, n, qc) >= qc.suff) return true;
    return false;
  }
  bool reject(const uint64_t* ids, uint32_t n) const {          // wotqs.go:177-184
    for (auto& qc : qcs) if (qc.f == 0 || bftkv::host::intersection_count(ids, n, qc) <= qc.f) return false;
    return true;
  }
  int get_threshold() const { int t = 0; for (auto& qc : qcs) t += qc.threshold; return t; }   // wotqs.go:186-192
};

// Running tally of one growing node list against a quorum: the collectors call IsSufficient / Reject after every
// reply (client.go:153, crypto_pgp.go:493), which costs the reference O(k * n) per call; per-clique counters make
// each step O(#cliques) with identical answers (same multiplicity rule as intersection()).
struct bftkv_tally {
  const bftkv_quorum* q;
  std::vector<int> count;
  explicit bftkv_tally(const bftkv_quorum* quorum) : q(quorum), count(quorum->qcs.size(), 0) {}
  void add(uint64_t id) { for (size_t i = 0; i < q->qcs.size(); ++i) count[i] += q->qcs[i].has(id); }
  bool is_sufficient() const { for (size_t i = 0; i < count.size(); ++i) if (q->qcs[i].suff > 0 && count[i] >= q->qcs[i].suff) return true; return false; }
  bool is_threshold() const {
    if (count.empty()) return false;
    for (size_t i = 0; i < count.size(); ++i) if (q->qcs[i].threshold > 0 && count[i] < q->qcs[i].threshold) return false;
    return true;
  }
  bool is_quorum() const {
    if (count.empty()) return false;
    for (size_t i = 0; i < count.size(); ++i) if (q->qcs[i].f > 0 && count[i] < q->qcs[i].min) return false;
    return true;
  }
  bool reject() const { for (size_t i = 0; i < count.size(); ++i) if (q->qcs[i].f == 0 || count[i] <= q->qcs[i].f) return false; return true; }
};
