package pgpgpu

/*
#include "bftkv_gpu.h"
*/
import "C"

import (
	"github.com/yahoo/bftkv/crypto"
	"github.com/yahoo/bftkv/node"
	"github.com/yahoo/bftkv/packet"
)

// Signature replaces pgp.PGPSignature's verifying half (crypto/pgp/crypto_pgp.go:319-344); Sign / Signers / Issuer / Certs
// parse or use the private key and stay with crypto/pgp.
type Signature struct {
	g     *gpu
	inner crypto.Signature
}

func (s *Signature) verify(tbs []byte, sig *packet.SignaturePacket, certId *C.uint64_t, fallback func() error) error {
	if sig == nil {
		return crypto.ErrInvalidSignature
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

// VerifyWithCertificate replaces crypto_pgp.go:332-344 (protocol/server.go:207, 468).  The certificate's entity must be
// in the device table: entities of the node keyring are; for a certificate that only travels in the request
// (sig.Cert, crypto_pgp.go:392-405) the reference path is used -- include/bftkv_host.h's bftkv_host_server_sign_verify
// shows the batched form that registers request certificates as certificate-only entities.
func (s *Signature) VerifyWithCertificate(tbs []byte, sig *packet.SignaturePacket, cert node.Node) error {
	if cert == nil {
		return crypto.ErrInvalidSignature
	}
	id := C.uint64_t(cert.Id())
	return s.verify(tbs, sig, &id, func() error { return s.inner.VerifyWithCertificate(tbs, sig, cert) })
}

func (s *Signature) Sign(tbs []byte) (*packet.SignaturePacket, error)        { return s.inner.Sign(tbs) }
func (s *Signature) Signers(sig *packet.SignaturePacket) []node.Node         { return s.inner.Signers(sig) }
func (s *Signature) Issuer(sig *packet.SignaturePacket) node.Node            { return s.inner.Issuer(sig) }
func (s *Signature) Certs(sig *packet.SignaturePacket) ([]node.Node, error)  { return s.inner.Certs(sig) }
