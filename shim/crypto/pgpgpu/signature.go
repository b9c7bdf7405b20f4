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

// Signature replaces pgp.PGPSignature's verifying half (crypto/pgp/crypto_pgp.go:319-344); Sign / Signers / Issuer / Certs
// parse or use the private key and stay with crypto/pgp.
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
// key material of the certificate it is HANDED.  At server.go:199-207 and :461-468 that is sig.Cert parsed out of the
// request (crypto_pgp.go:392-405) -- normally a principal that is not in the node keyring at all, and even under a
// keyring id not necessarily the keyring's key.  The device only holds the node keyring, so the GPU path is taken only
// when cert IS one of the uploaded entities (keyring.holds: same *openpgp.Entity); every other certificate goes to
// crypto/pgp.  (include/bftkv_host.h's bftkv_host_server_sign_verify shows the batched form that registers request
// certificates as certificate-only entities; the library itself reports an entity it does not hold as fenced, never as a
// verdict.)
func (s *Signature) VerifyWithCertificate(tbs []byte, sig *packet.SignaturePacket, cert node.Node) error {
	if cert == nil {
		return crypto.ErrInvalidSignature
	}
	ref := func() error { return s.inner.VerifyWithCertificate(tbs, sig, cert) }
	e, ok := cert.Instance().(*openpgp.Entity)
	if !ok || !s.keyring.holds(e) {
		return ref()
	}
	id := C.uint64_t(cert.Id())
	return s.verify(tbs, sig, &id, ref)
}

func (s *Signature) Sign(tbs []byte) (*packet.SignaturePacket, error)        { return s.inner.Sign(tbs) }
func (s *Signature) Signers(sig *packet.SignaturePacket) []node.Node         { return s.inner.Signers(sig) }
func (s *Signature) Issuer(sig *packet.SignaturePacket) node.Node            { return s.inner.Issuer(sig) }
func (s *Signature) Certs(sig *packet.SignaturePacket) ([]node.Node, error)  { return s.inner.Certs(sig) }
