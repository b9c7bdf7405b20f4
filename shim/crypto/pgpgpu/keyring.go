package pgpgpu

/*
#include <stdlib.h>
#include "bftkv_gpu.h"
*/
import "C"

import (
	"crypto/dsa"
	"crypto/rsa"
	"math/big"
	"sync"
	"unsafe"

	"golang.org/x/crypto/openpgp"
	"golang.org/x/crypto/openpgp/packet"

	"github.com/yahoo/bftkv/crypto"
	"github.com/yahoo/bftkv/node"
)

// keyring wraps crypto/pgp's PGPKeyring.  PGPKeyring.getKeyring() -- secring entities first, then keyring
// (crypto_pgp.go:195-197) -- is unexported, so the wrapper watches Register / Remove (crypto_pgp.go:142-177) to keep the
// same two lists and re-uploads the device key table after every change.
type keyring struct {
	inner crypto.Keyring
	g     *gpu

	mu      sync.Mutex
	secring openpgp.EntityList
	pubring openpgp.EntityList
	// stale: the last upload failed, so the device table may differ from the rings (a REMOVED node could still verify
	// there).  While it is set every verifying method goes to crypto/pgp; the next successful sync clears it.
	stale bool
}

// fresh reports whether the device key table is known to equal getKeyring().
func (k *keyring) fresh() bool {
	k.mu.Lock()
	defer k.mu.Unlock()
	return !k.stale
}

// holds reports whether e IS one of the entities uploaded to the device (same object: Register stores the node's
// *openpgp.Entity itself, crypto_pgp.go:142-160).  A certificate that only travels inside a request is a different object
// even when its key id equals a keyring entity's, and its key material need not be the keyring's.
func (k *keyring) holds(e *openpgp.Entity) bool {
	k.mu.Lock()
	defer k.mu.Unlock()
	if k.stale {
		return false
	}
	// The device call names the certificate by its 64-bit id and resolves it to the FIRST uploaded entity with that id.  If
	// another uploaded entity (a different object) shares the id, the device might verify against that twin instead of the
	// entity handed in: such a certificate goes to crypto/pgp (ADVICE r03).
	found := false
	for _, ring := range []openpgp.EntityList{k.secring, k.pubring} {
		for _, r := range ring {
			if r == e {
				found = true
			} else if r.PrimaryKey.KeyId == e.PrimaryKey.KeyId {
				return false
			}
		}
	}
	return found
}

// upsert mirrors crypto/pgp's replace (crypto_pgp.go:125-140): an entity whose primary key id the ring already holds takes that
// entry's place (the first such entry), the others are appended.  NOT the same order in one respect: replace collects the new
// nodes in a Go MAP and appends what is left of it in map-iteration order, i.e. in a random order that differs from run to run;
// here they are appended in argument order.  Ring order is observable only through which candidate is asked first under one
// 64-bit key id (KeysByIdUsage walks the ring), so for distinct key ids the two cannot be told apart, and with id twins
// registered in ONE call the reference is itself nondeterministic -- which is one more reason holds() sends certificates with an
// id twin to crypto/pgp instead of answering for them.
func upsert(ring openpgp.EntityList, nodes []node.Node) openpgp.EntityList {
	for _, n := range nodes {
		e := n.Instance().(*openpgp.Entity)
		found := false
		for i := range ring {
			if ring[i].PrimaryKey.KeyId == e.PrimaryKey.KeyId {
				ring[i] = e
				found = true
				break
			}
		}
		if !found {
			ring = append(ring, e)
		}
	}
	return ring
}

// Register and Remove hold k.mu across the inner call AND the mirror update: a concurrent Register / Remove of one node
// can then not leave the mirror and the PGPKeyring in different orders.
func (k *keyring) Register(nodes []node.Node, priv bool, self bool) error {
	k.mu.Lock()
	defer k.mu.Unlock()
	if err := k.inner.Register(nodes, priv, self); err != nil {
		return err
	}
	if priv {
		k.secring = upsert(k.secring, nodes) // crypto_pgp.go:144-145 (replace; see upsert for the one difference)
	} else {
		k.pubring = upsert(k.pubring, nodes) // crypto_pgp.go:146-147
	}
	return k.syncLocked()
}

func (k *keyring) Remove(nodes []node.Node) {
	k.mu.Lock()
	defer k.mu.Unlock()
	k.inner.Remove(nodes)
	var kept openpgp.EntityList
	for _, e := range k.pubring {
		drop := false
		for _, n := range nodes {
			if n.Id() == e.PrimaryKey.KeyId {
				drop = true
			}
		}
		if !drop {
			kept = append(kept, e)
		}
	}
	k.pubring = kept
	// Remove has no error to return (crypto.Keyring, crypto/crypto.go:35-41).  If the upload fails the revoked node's keys
	// are still on the device: syncLocked marks the table stale and every Verify takes the reference path until an upload
	// succeeds -- revocation must not fail open.
	_ = k.syncLocked()
}

func (k *keyring) syncLocked() error {
	err := k.sync()
	k.stale = err != nil
	return err
}

func (k *keyring) GetCertById(id uint64) node.Node { return k.inner.GetCertById(id) }
func (k *keyring) GetKeyring() []node.Node         { return k.inner.GetKeyring() }

// private entities (the decrypt half of Message stays on the CPU)
func (k *keyring) privateKeys() openpgp.EntityList {
	k.mu.Lock()
	defer k.mu.Unlock()
	return append(openpgp.EntityList{}, k.secring...)
}

// usable reproduces what EntityList.KeysByIdUsage(id, packet.KeyFlagSign) filters on: entity not revoked, self-signature
// not a revocation, key flags absent or containing Sign.
func usable(e *openpgp.Entity, self *packet.Signature) bool {
	if len(e.Revocations) > 0 {
		return false
	}
	if self == nil {
		return true
	}
	if self.RevocationReason != nil {
		return false
	}
	return !self.FlagsValid || self.FlagSign
}

func primarySelfSig(e *openpgp.Entity) *packet.Signature {
	for _, id := range e.Identities {
		if id.SelfSignature != nil && id.SelfSignature.IsPrimaryId != nil && *id.SelfSignature.IsPrimaryId {
			return id.SelfSignature
		}
	}
	for _, id := range e.Identities { // first identity otherwise (Entity.primaryIdentity)
		return id.SelfSignature
	}
	return nil
}

// sync uploads getKeyring() order to the device (bftkv_gpu_keyring_set).  Key material is copied into C memory: a C
// struct that holds pointers may not point into Go memory.
func (k *keyring) sync() error {
	var keys []C.bftkv_gpu_pubkey
	var frees []unsafe.Pointer
	defer func() {
		for _, p := range frees {
			C.free(p)
		}
	}()
	cbytes := func(b []byte) (*C.uint8_t, C.uint32_t) {
		if len(b) == 0 {
			return nil, 0
		}
		p := C.CBytes(b)
		frees = append(frees, p)
		return (*C.uint8_t)(p), C.uint32_t(len(b))
	}
	add := func(e *openpgp.Entity, pk *packet.PublicKey, self *packet.Signature) {
		var rec C.bftkv_gpu_pubkey
		rec.key_id = C.uint64_t(pk.KeyId)
		rec.entity_id = C.uint64_t(e.PrimaryKey.KeyId)
		rec.pk_algo = C.uint8_t(pk.PubKeyAlgo)
		if usable(e, self) {
			rec.usable_sign = 1
		}
		switch pub := pk.PublicKey.(type) {
		case *rsa.PublicKey:
			rec.n, rec.n_len = cbytes(pub.N.Bytes())
			rec.e, rec.e_len = cbytes(big.NewInt(int64(pub.E)).Bytes())
		case *dsa.PublicKey:
			rec.n, rec.n_len = cbytes(pub.P.Bytes())
			rec.e, rec.e_len = cbytes(pub.Q.Bytes())
			rec.g, rec.g_len = cbytes(pub.G.Bytes())
			rec.y, rec.y_len = cbytes(pub.Y.Bytes())
		default:
			// ECDSA / ElGamal / unknown: uploaded without material; signatures naming the key are fenced
		}
		keys = append(keys, rec)
	}
	for _, ring := range []openpgp.EntityList{k.secring, k.pubring} {
		for _, e := range ring {
			add(e, e.PrimaryKey, primarySelfSig(e))
			for i := range e.Subkeys {
				add(e, e.Subkeys[i].PublicKey, e.Subkeys[i].Sig)
			}
		}
	}
	var p *C.bftkv_gpu_pubkey
	if len(keys) > 0 {
		// the array itself lives in Go memory; its pointer FIELDS point to C memory (allowed)
		p = &keys[0]
	}
	return k.g.infra(C.bftkv_gpu_keyring_set(k.g.ctx, p, C.uint32_t(len(keys))), "keyring_set")
}
