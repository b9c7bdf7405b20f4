package pgpgpu

/*
#include "bftkv_gpu.h"
#include "bftkv_host.h"
*/
import "C"

import (
	"github.com/yahoo/bftkv/crypto"
	"github.com/yahoo/bftkv/node"
	"github.com/yahoo/bftkv/packet"
	"github.com/yahoo/bftkv/quorum"
)

// CollectiveSignature replaces pgp.PGPCollectiveSignature (crypto/pgp/crypto_pgp.go:476-519).
type CollectiveSignature struct {
	g       *gpu
	inner   crypto.CollectiveSignature
	keyring *keyring
}

// Verify replaces PGPCollectiveSignature.Verify (crypto_pgp.go:485-500); call sites protocol/server.go:182,237,300,473,
// protocol/client.go:165,470, api/api.go:130.
func (cs *CollectiveSignature) Verify(tbs []byte, ss *packet.SignaturePacket, q quorum.Quorum) error {
	if !cs.keyring.fresh() {
		return cs.inner.Verify(tbs, ss, q) // a device table not known to equal the keyring: the reference decides
	}
	qe, err := cs.g.quorumAcquire(q)
	if err != nil {
		return cs.inner.Verify(tbs, ss, q) // a quorum this package cannot describe: the reference decides
	}
	var e, fenced C.uint8_t
	rc := C.bftkv_gpu_batcher_collective_verify(cs.g.batcher, qe.h, ptr(tbs), C.uint64_t(len(tbs)), ptr(ss.Data), C.uint64_t(len(ss.Data)), &e, &fenced)
	cs.g.quorumRelease(qe)
	if rc != 0 {
		// infrastructure error: never a verdict.  (The status byte is a failure too -- the library fails closed.)
		return cs.g.infra(rc, "collective_verify")
	}
	if fenced != 0 {
		// ss.Data holds a shape the kernels do not follow (partial lengths, text mode, MD5, ...): x/crypto decides
		return cs.inner.Verify(tbs, ss, q)
	}
	if e == C.BFTKV_ERR_NONE {
		ss.Completed = true // crypto_pgp.go:494: the client serialises it afterwards (client.go:102), servers store it
		return nil
	}
	return crypto.ErrInsufficientNumberOfSignatures // crypto/crypto.go:19; compared by identity and by string (X-error)
}

// Request is one element of VerifyBatch: the bytes Server.write verifies (protocol/server.go:286-300): tbss = packet.TBSS(req),
// ss and -- when the caller still holds the parsed packet -- the signer's certificate sig.Cert, which is the TAIL of tbss
// (writeSignature ends in chunk(Cert), packet/packet.go:192-212).
type Request struct {
	TBSS []byte
	Cert []byte // optional: the tail of TBSS that every write of this client shares (sig.Cert); nil = send TBSS whole
	SS   *packet.SignaturePacket
}

// VerifyBatch is CollectiveSignature.Verify over many requests in ONE device call from host memory: what a bulk caller -- an
// audit of a stored database, the revoke sweep of client.go:304-353, a server draining a queue of writes -- uses instead of one
// Verify per goroutine.  Each distinct certificate crosses PCIe once (bftkv_gpu_collective_verify_segments): 146 MB instead of
// 226 MB per 10,000 writes at 64 replicas.  errs[i] is what cs.Verify(reqs[i].TBSS, reqs[i].SS, q) returns, ss.Completed
// included; fenced items and infrastructure errors go to the reference path one by one.
func (cs *CollectiveSignature) VerifyBatch(reqs []Request, q quorum.Quorum) []error {
	errs := make([]error, len(reqs))
	one := func(i int) { errs[i] = cs.inner.Verify(reqs[i].TBSS, reqs[i].SS, q) }
	if len(reqs) == 0 {
		return errs
	}
	qe, err := cs.g.quorumAcquire(q)
	if !cs.keyring.fresh() || err != nil {
		for i := range reqs {
			one(i)
		}
		return errs
	}
	defer cs.g.quorumRelease(qe)
	var prefix, shared, ssb []byte
	poff := make([]C.uint64_t, 1, len(reqs)+1)
	soff := make([]C.uint64_t, 1, len(reqs)+1)
	shoff := []C.uint64_t{0}
	seg := make([]C.uint32_t, len(reqs))
	tails := map[string]C.uint32_t{}
	for i, r := range reqs {
		t := r.TBSS
		seg[i] = 0xFFFFFFFF
		if n := len(r.Cert); n >= 256 && n <= len(t) && string(t[len(t)-n:]) == string(r.Cert) {
			g, ok := tails[string(r.Cert)]
			if !ok {
				g = C.uint32_t(len(tails))
				tails[string(r.Cert)] = g
				shared = append(shared, r.Cert...)
				shoff = append(shoff, C.uint64_t(len(shared)))
			}
			seg[i] = g
			t = t[:len(t)-n]
		}
		prefix = append(prefix, t...)
		poff = append(poff, C.uint64_t(len(prefix)))
		if r.SS != nil {
			ssb = append(ssb, r.SS.Data...)
		}
		soff = append(soff, C.uint64_t(len(ssb)))
	}
	e := make([]C.uint8_t, len(reqs))
	fenced := make([]C.uint8_t, len(reqs))
	rc := C.bftkv_gpu_collective_verify_segments(cs.g.ctx, qe.h, C.uint32_t(len(reqs)), ptr(prefix), &poff[0], ptr(shared), &shoff[0],
		C.uint32_t(len(shoff)-1), &seg[0], ptr(ssb), &soff[0], &e[0], nil, nil, &fenced[0])
	for i := range reqs {
		switch {
		case rc != 0 || fenced[i] != 0 || reqs[i].SS == nil:
			one(i) // never a verdict from a failed call (the library fails closed); fenced shapes: x/crypto decides
		case e[i] == C.BFTKV_ERR_NONE:
			reqs[i].SS.Completed = true // crypto_pgp.go:494
		default:
			errs[i] = crypto.ErrInsufficientNumberOfSignatures
		}
	}
	return errs
}

// Combine replaces crypto_pgp.go:506-515: append, then IsSufficient over the CLAIMED signers (the real check is Verify).
func (cs *CollectiveSignature) Combine(ss, s *packet.SignaturePacket, q quorum.Quorum) bool {
	if ss.Type == packet.SignatureTypeNil {
		ss.Type = s.Type
	} else if ss.Type != s.Type {
		return false
	}
	ss.Data = append(ss.Data, s.Data...)
	return q.IsSufficient(cs.Signers(ss))
}

// Sign needs the node's private key: crypto/pgp (crypto_pgp.go:502-504).
func (cs *CollectiveSignature) Sign(tbs []byte) (*packet.SignaturePacket, error) { return cs.inner.Sign(tbs) }

// Signers replaces crypto_pgp.go:517-519 -> PGPSignature.Signers (:373-390): parse-only walk, issuers looked up among the
// primary key ids of the keyring (getCertById).  Batches of streams (revoke's history walk, client.go:304-353) have
// bftkv_gpu_signers_fenced; one stream at a time is walked on the host.
func (cs *CollectiveSignature) Signers(ss *packet.SignaturePacket) []node.Node {
	return signers(cs.g, cs.keyring, ss, cs.inner.Signers)
}

func signers(g *gpu, kr *keyring, ss *packet.SignaturePacket, fallback func(*packet.SignaturePacket) []node.Node) []node.Node {
	if ss == nil || len(ss.Data) == 0 {
		return nil
	}
	// One stream, parse only: walked on this goroutine's thread by the library's HOST build of its signer walk
	// (bftkv_host_signers_walk: the code of the device kernel; no device round trip, no context lock) -- Combine asks for the
	// signers after every signature it appends (client.go:153), a few dozen packets each time.  The walk returns the issuer of
	// every version-4 signature Reader.Next yields before its first error; getCertById is the mirror keyring's, as in the reference.
	capIds := len(ss.Data)/12 + 1 // a signature packet is never shorter than 12 bytes
	ids := make([]C.uint64_t, capIds)
	var n C.uint32_t
	var fenced C.uint8_t
	if rc := C.bftkv_host_signers_walk(ptr(ss.Data), C.uint64_t(len(ss.Data)), &ids[0], C.uint32_t(capIds), &n, &fenced); rc != 0 || fenced != 0 {
		return fallback(ss) // a shape the walk does not follow (or a v4 signature without issuer, on which the reference dereferences nil): crypto/pgp decides
	}
	var nodes []node.Node
	for _, id := range ids[:int(n)] {
		if nd := kr.GetCertById(uint64(id)); nd != nil {
			nodes = append(nodes, nd)
		}
	}
	return nodes
}
