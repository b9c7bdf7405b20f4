package pgpgpu

/*
#include "bftkv_gpu.h"
*/
import "C"

import (
	"bytes"
	"io"

	"golang.org/x/crypto/openpgp"
	pgppacket "golang.org/x/crypto/openpgp/packet"

	"github.com/yahoo/bftkv/crypto/pgp"
	"github.com/yahoo/bftkv/node"
	"github.com/yahoo/bftkv/packet"
)

// Issuer replaces PGPSignature.Issuer (crypto/pgp/crypto_pgp.go:392-405; protocol/server.go:199, 330-331, 460) for the principal
// whose certificate travels inside the request.  The reference parses sig.Cert with openpgp.ReadEntity, which VERIFIES the first
// entity's user-id self-signatures, subkey bindings (with the cross-signature of a signing subkey) and revocations on the CPU --
// and Server.sign / Server.register then call VerifyWithCertificate, whose GPU route (bftkv_gpu_batcher_cert_verify) makes the
// same checks again on the device.  Here they are made ONCE, on the device (remembered by certificate bytes):
// bftkv_gpu_batcher_cert_entity answers "ReadEntity returns the first entity" and says what ReadEntity does with each of its
// packets; the *openpgp.Entity is then assembled from packet.Read objects -- parsing only, no cryptography -- and wrapped with
// pgp.NewNode (shim/patches/0004).  Anything else -- a fence, a refusal, an infrastructure error, a stale key table, a packet
// stream that does not read back the way the library walked it -- goes to crypto/pgp, which decides as it always did.
func (s *Signature) Issuer(sig *packet.SignaturePacket) node.Node {
	if sig != nil && len(sig.Cert) != 0 && s.keyring.fresh() {
		if n := s.issuerFromLibrary(sig.Cert); n != nil {
			return n
		}
	}
	return s.inner.Issuer(sig)
}

// Certs returns EVERY entity of sig.Cert (crypto_pgp.go:388-390); nothing in protocol/ calls it but Issuer: crypto/pgp.
func (s *Signature) Certs(sig *packet.SignaturePacket) ([]node.Node, error) { return s.inner.Certs(sig) }

func (s *Signature) issuerFromLibrary(cert []byte) node.Node {
	var st, fenced C.uint8_t
	var issuer, off, ln C.uint64_t
	var fp [20]C.uint8_t
	var n C.uint32_t
	roles := make([]C.uint32_t, 64)
	for {
		rc := C.bftkv_gpu_batcher_cert_entity(s.g.batcher, ptr(cert), C.uint64_t(len(cert)), &st, &fenced, &issuer, &fp[0], &off, &ln,
			&roles[0], C.uint32_t(len(roles)), &n)
		if rc == C.BFTKV_E_NOMEM && int(n) > len(roles) {
			roles = make([]C.uint32_t, int(n)) // a certificate of more packets than the first guess: the count came back
			continue
		}
		if rc != 0 || fenced != 0 || st != C.BFTKV_ERR_NONE {
			return nil // no claim (the status byte is a failure whenever rc != 0: the library fails closed)
		}
		break
	}
	if uint64(off)+uint64(ln) > uint64(len(cert)) || uint64(issuer) == 0 {
		return nil
	}
	e := assemble(cert[uint64(off):uint64(off)+uint64(ln)], roles[:int(n)])
	if e == nil || e.PrimaryKey == nil || e.PrimaryKey.KeyId != uint64(issuer) {
		return nil
	}
	for i := range fp {
		if byte(fp[i]) != e.PrimaryKey.Fingerprint[i] {
			return nil // not the key the library verified the certificate under
		}
	}
	return pgp.NewNode(e)
}

// assemble builds the entity openpgp.ReadEntity would return for these bytes, given what ReadEntity does with each packet
// (include/bftkv_gpu.h BFTKV_ROLE_*).  nil whenever the packets do not read back as the roles say.
func assemble(ent []byte, roles []C.uint32_t) *openpgp.Entity {
	packets := pgppacket.NewReader(bytes.NewReader(ent))
	e := &openpgp.Entity{Identities: make(map[string]*openpgp.Identity)}
	var idents []*openpgp.Identity
	for _, r := range roles {
		p, err := packets.Next() // (skips the packet types x/crypto does not know, as the library's walk does)
		if err != nil {
			return nil
		}
		role, idx, chosen := uint32(r)&0xFF, int((uint32(r)>>8)&0xFFFF), uint32(r)>>24 != 0
		sig, isSig := p.(*pgppacket.Signature)
		switch role {
		case C.BFTKV_ROLE_PRIMARY_KEY:
			pk, ok := p.(*pgppacket.PublicKey)
			if !ok || e.PrimaryKey != nil {
				return nil
			}
			e.PrimaryKey = pk
		case C.BFTKV_ROLE_USER_ID:
			uid, ok := p.(*pgppacket.UserId)
			if !ok || idx != len(idents) {
				return nil
			}
			idents = append(idents, &openpgp.Identity{Name: uid.Id, UserId: uid})
		case C.BFTKV_ROLE_SELF_SIGNATURE:
			if !isSig || idx >= len(idents) {
				return nil
			}
			idents[idx].SelfSignature = sig
			e.Identities[idents[idx].Name] = idents[idx] // a later identity of the same name takes the earlier one's place
		case C.BFTKV_ROLE_IDENTITY_SIGNATURE:
			if !isSig || idx >= len(idents) || sig.IssuerKeyId == nil {
				return nil
			}
			idents[idx].Signatures = append(idents[idx].Signatures, sig)
		case C.BFTKV_ROLE_SUBKEY:
			pk, ok := p.(*pgppacket.PublicKey)
			if !ok || idx != len(e.Subkeys) {
				return nil
			}
			e.Subkeys = append(e.Subkeys, openpgp.Subkey{PublicKey: pk})
		case C.BFTKV_ROLE_SUBKEY_SIGNATURE:
			if !isSig || idx >= len(e.Subkeys) {
				return nil
			}
			if chosen {
				e.Subkeys[idx].Sig = sig // shouldReplaceSubkeySig's winner: a revocation, else the newest binding
			}
		case C.BFTKV_ROLE_REVOCATION:
			if !isSig {
				return nil
			}
			e.Revocations = append(e.Revocations, sig)
		case C.BFTKV_ROLE_IGNORED:
			// a version-3 signature, a stray signature outside any run: read and dropped
		default:
			return nil
		}
	}
	if _, err := packets.Next(); err != io.EOF {
		return nil // the entity's bytes hold a packet the library's walk did not account for
	}
	for _, sk := range e.Subkeys {
		if sk.Sig == nil {
			return nil
		}
	}
	if e.PrimaryKey == nil || len(e.Identities) == 0 {
		return nil
	}
	return e
}
