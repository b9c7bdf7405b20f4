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

// qentry is one cached quorum handle.  refs counts the device calls that are using it right now: an entry evicted from the
// cache while a call holds it is destroyed by the last release, never under a call in flight.
type qentry struct {
	h    C.int
	refs int
	dead bool
}

// quorumAcquire flattens q into bftkv_gpu_qc descriptors and returns its handle, held until quorumRelease.  ChooseQuorum builds a
// fresh value on every call (wotqs.go:117-127), so handles are cached by content; the oldest is retired beyond maxCachedQuorums.
func (g *gpu) quorumAcquire(q quorum.Quorum) (*qentry, error) {
	cl, ok := q.(cliqueLister)
	if !ok {
		return nil, errForeignQuorum
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
	if e, ok := g.quorum[string(key)]; ok {
		e.refs++
		return e, nil
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
		return nil, err
	}
	if len(g.qorder) >= maxCachedQuorums {
		old := g.qorder[0]
		g.qorder = g.qorder[1:]
		if e := g.quorum[old]; e != nil {
			delete(g.quorum, old)
			e.dead = true
			if e.refs == 0 {
				C.bftkv_gpu_quorum_destroy(g.ctx, e.h)
			}
		}
	}
	e := &qentry{h: h, refs: 1}
	g.quorum[string(key)] = e
	g.qorder = append(g.qorder, string(key))
	return e, nil
}

// quorumRelease ends a call's hold on e; the last holder of a retired entry destroys the device-side quorum.
func (g *gpu) quorumRelease(e *qentry) {
	g.qmu.Lock()
	defer g.qmu.Unlock()
	e.refs--
	if e.dead && e.refs == 0 {
		C.bftkv_gpu_quorum_destroy(g.ctx, e.h)
	}
}
