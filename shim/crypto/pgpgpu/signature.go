package pgpgpu

/*
#include "bftkv_gpu.h"
*/
import "C"

import (
	"golang.org/x/crypto/openpgp"

	"github.com/yahoo/bftkv/crypto"
	"github.com/yahoo/bftkv/node"
	"github.com/yahoo/bftkv/packet"
)

// Signature replaces pgp.PGPSignature's verifying half (crypto/pgp/crypto_pgp.go:319-344), for keyring entities and for the
// principals whose certificate travels in the request; Sign uses the private key and stays with crypto/pgp, and so do Signers and
// Certs.  Issuer (issuer.go) has the device make the checks openpgp.ReadEntity makes and assembles the *openpgp.Entity from
// parsed packets, so that a request's certificate is verified once, not on the CPU in Issuer and again in VerifyWithCertificate.
type Signature struct {
	g       *gpu
	inner   crypto.Signature
	keyring *keyring
}

func (s *Signature) verify(tbs []byte, sig *packet.SignaturePacket, certId *C.uint64_t, fallback func() error) error {
	if sig == nil {
		return crypto.ErrInvalidSignature
	}
	if !s.keyring.fresh() {
		return fallback() // the device table is not known to equal the keyring (a failed upload): crypto/pgp decides
	}
	var e, fenced C.uint8_t
	rc := C.bftkv_gpu_batcher_signature_verify(s.g.batcher, ptr(tbs), C.uint64_t(len(tbs)), ptr(sig.Data), C.uint64_t(len(sig.Data)), certId, &e, &fenced)
	if rc != 0 {
		return s.g.infra(rc, "signature_verify")
	}
	if fenced != 0 {
		return fallback()
	}
	if e != C.BFTKV_ERR_NONE {
		return crypto.ErrInvalidSignature // crypto/crypto.go:20
	}
	return nil
}

// Verify replaces crypto_pgp.go:319-330 (no caller inside protocol/ today; interface method).
func (s *Signature) Verify(tbs []byte, sig *packet.SignaturePacket) error {
	return s.verify(tbs, sig, nil, func() error { return s.inner.Verify(tbs, sig) })
}

// VerifyWithCertificate replaces crypto_pgp.go:332-344 (protocol/server.go:207, 468).  The reference verifies against the
// key material of the certificate it is HANDED.  Two GPU routes:
//   - cert IS one of the uploaded keyring entities (keyring.holds: same *openpgp.Entity, no id twin): the batched
//     Signature.Verify with the keyring of the item restricted to that entity;
//   - cert is a principal OUTSIDE the node keyring -- the normal case at server.go:199-207 and :460-468, where the issuer was
//     parsed out of the request's own sig.Cert (crypto_pgp.go:392-405): bftkv_gpu_batcher_cert_verify registers the FIRST
//     entity of sig.Cert as a certificate-only entity of the device table (bounded, recycled), checks what openpgp.ReadEntity
//     checks about it once per distinct certificate, and verifies sig.Data against it -- one micro-batched call.  The library
//     answers for the first entity of sig.Cert; the reference answers for the node it was handed.  They are the same key
//     material exactly when the handed entity's primary-key fingerprint is the one the library reports (the server hands in
//     Issuer(sig), which IS that entity); anything else -- another certificate, a fenced shape, a certificate ReadEntity would
//     refuse (the caller could then not hold a node parsed from it) -- goes to crypto/pgp.
func (s *Signature) VerifyWithCertificate(tbs []byte, sig *packet.SignaturePacket, cert node.Node) error {
	if cert == nil {
		return crypto.ErrInvalidSignature
	}
	ref := func() error { return s.inner.VerifyWithCertificate(tbs, sig, cert) }
	e, ok := cert.Instance().(*openpgp.Entity)
	if !ok {
		return ref()
	}
	if s.keyring.holds(e) {
		id := C.uint64_t(cert.Id())
		return s.verify(tbs, sig, &id, ref)
	}
	if sig == nil || len(sig.Cert) == 0 || e.PrimaryKey == nil || !s.keyring.fresh() {
		return ref()
	}
	var st, fenced C.uint8_t
	var issuer C.uint64_t
	var fp [20]C.uint8_t
	rc := C.bftkv_gpu_batcher_cert_verify(s.g.batcher, ptr(sig.Cert), C.uint64_t(len(sig.Cert)), ptr(tbs), C.uint64_t(len(tbs)),
		ptr(sig.Data), C.uint64_t(len(sig.Data)), &st, &fenced, &issuer, &fp[0])
	if rc != 0 {
		return s.g.infra(rc, "cert_verify")
	}
	if fenced != 0 || st == C.BFTKV_ERR_CERTIFICATE_NOT_FOUND {
		return ref()
	}
	for i := range fp {
		if byte(fp[i]) != e.PrimaryKey.Fingerprint[i] {
			return ref() // not the certificate the library verified against
		}
	}
	if len(sig.Data) == 0 {
		return crypto.ErrInvalidSignature // the reference's loop never runs (crypto_pgp.go:336-337); never sent as "issuer alone"
	}
	if st != C.BFTKV_ERR_NONE {
		return crypto.ErrInvalidSignature
	}
	return nil
}

func (s *Signature) Sign(tbs []byte) (*packet.SignaturePacket, error) { return s.inner.Sign(tbs) }
func (s *Signature) Signers(sig *packet.SignaturePacket) []node.Node  { return s.inner.Signers(sig) }
