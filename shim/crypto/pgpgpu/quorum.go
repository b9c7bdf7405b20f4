package pgpgpu

/*
#include <stdlib.h>
#include "bftkv_gpu.h"
*/
import "C"

import (
	"encoding/binary"
	"errors"
	"unsafe"

	"github.com/yahoo/bftkv/quorum"
	"github.com/yahoo/bftkv/quorum/wotqs"
)

// cliqueLister is what shim/patches/0001-wotqs-export-cliques.patch adds to *wotqs.wotq: the per-clique numbers newQC
// computed (quorum/wotqs/wotqs.go:36-70), which the quorum.Quorum interface itself does not expose.
type cliqueLister interface {
	Cliques() []wotqs.Clique
}

const maxCachedQuorums = 64

var errForeignQuorum = errors.New("pgpgpu: quorum.Quorum is not a wotqs quorum (no Cliques())")

// quorumHandle flattens q into bftkv_gpu_qc descriptors.  ChooseQuorum builds a fresh value on every call
// (wotqs.go:117-127), so handles are cached by content; the oldest is destroyed beyond maxCachedQuorums.
func (g *gpu) quorumHandle(q quorum.Quorum) (C.int, error) {
	cl, ok := q.(cliqueLister)
	if !ok {
		return -1, errForeignQuorum
	}
	cs := cl.Cliques()
	key := make([]byte, 0, 64)
	var w [8]byte
	put := func(v uint64) { binary.BigEndian.PutUint64(w[:], v); key = append(key, w[:]...) }
	for _, c := range cs {
		put(uint64(c.F)<<48 | uint64(c.Min)<<32 | uint64(c.Threshold)<<16 | uint64(c.Suff))
		put(uint64(len(c.Nodes)))
		for _, n := range c.Nodes {
			put(n.Id())
		}
	}
	g.qmu.Lock()
	defer g.qmu.Unlock()
	if h, ok := g.quorum[string(key)]; ok {
		return h, nil
	}
	qcs := make([]C.bftkv_gpu_qc, len(cs))
	var frees []unsafe.Pointer
	defer func() {
		for _, p := range frees {
			C.free(p)
		}
	}()
	for i, c := range cs {
		qcs[i].f, qcs[i].min, qcs[i].threshold, qcs[i].suff = C.int32_t(c.F), C.int32_t(c.Min), C.int32_t(c.Threshold), C.int32_t(c.Suff)
		if len(c.Nodes) > 0 {
			ids := (*[1 << 28]C.uint64_t)(C.malloc(C.size_t(8 * len(c.Nodes))))
			frees = append(frees, unsafe.Pointer(ids))
			for j, n := range c.Nodes {
				ids[j] = C.uint64_t(n.Id())
			}
			qcs[i].node_ids = &ids[0]
			qcs[i].n_nodes = C.uint32_t(len(c.Nodes))
		}
	}
	var h C.int
	var p *C.bftkv_gpu_qc
	if len(qcs) > 0 {
		p = &qcs[0]
	}
	if err := g.infra(C.bftkv_gpu_quorum_create(g.ctx, p, C.uint32_t(len(qcs)), &h), "quorum_create"); err != nil {
		return -1, err
	}
	if len(g.qorder) >= maxCachedQuorums {
		old := g.qorder[0]
		g.qorder = g.qorder[1:]
		C.bftkv_gpu_quorum_destroy(g.ctx, g.quorum[old])
		delete(g.quorum, old)
	}
	g.quorum[string(key)] = h
	g.qorder = append(g.qorder, string(key))
	return h, nil
}
