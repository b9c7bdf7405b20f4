package pgpgpu

/*
#include "bftkv_gpu.h"
*/
import "C"

import (
	"bytes"
	"encoding/base64"
	"io"
	"io/ioutil"
	"unsafe"

	pgperrors "golang.org/x/crypto/openpgp/errors"
	pgppacket "golang.org/x/crypto/openpgp/packet"

	"github.com/yahoo/bftkv/crypto"
	"github.com/yahoo/bftkv/node"
)

// Message replaces only the SIGNATURE CHECK of PGPMessage.Decrypt (crypto/pgp/crypto_pgp.go:453-471 -> openpgp.ReadMessage);
// Encrypt / EncryptStream (crypto_pgp.go:419-451) and the opening of the container -- session-key decryption with the
// node's private key, AES-CFB, MDC -- stay on the CPU.  Every request (protocol/server.go:563) and every reply
// (transport/transport.go:119) goes through Decrypt, so the one public-key verification per message is what is batched.
type Message struct {
	g       *gpu
	inner   crypto.Message
	keyring *keyring
}

func (m *Message) Encrypt(peers []node.Node, plain []byte, nonce []byte) ([]byte, error) {
	return m.inner.Encrypt(peers, plain, nonce)
}
func (m *Message) EncryptStream(out io.Writer, peerId uint64, nonce []byte) (io.WriteCloser, error) {
	return m.inner.EncryptStream(out, peerId, nonce)
}

// openContainer does what openpgp.ReadMessage does up to the point where it starts reading the signed packet sequence:
// collect the public-key encrypted session keys, decrypt one with a private key of the node, open the symmetrically
// encrypted packet.  Returns the plaintext packet sequence ([one-pass signature] [literal data] [signature]); the MDC is
// verified when the reader hits EOF (SymmetricallyEncrypted.Decrypt's ReadCloser).
func (m *Message) openContainer(raw []byte) ([]byte, error) {
	packets := pgppacket.NewReader(bytes.NewReader(raw))
	var pubKeys []*pgppacket.EncryptedKey
	for {
		p, err := packets.Next()
		if err != nil {
			return nil, err
		}
		switch p := p.(type) {
		case *pgppacket.EncryptedKey:
			pubKeys = append(pubKeys, p)
		case *pgppacket.SymmetricallyEncrypted:
			// candidates as ReadMessage collects them: EntityList.KeysById (primary keys and subkeys under the id the
			// session key was encrypted to); only the node's own entities hold private keys
			secring := m.keyring.privateKeys()
			for _, ek := range pubKeys {
				for _, k := range secring.KeysById(ek.KeyId) {
					if k.PrivateKey == nil || k.PrivateKey.Encrypted {
						continue
					}
					if err := ek.Decrypt(k.PrivateKey, nil); err != nil {
						continue
					}
					rc, err := p.Decrypt(ek.CipherFunc, ek.Key)
					if err != nil {
						continue
					}
					seq, err := ioutil.ReadAll(rc)
					if err != nil {
						return nil, err
					}
					if err := rc.Close(); err != nil { // MDC
						return nil, err
					}
					return seq, nil
				}
			}
			return nil, pgperrors.ErrKeyIncorrect
		default:
			// anything else before the encrypted data: the reference's ReadMessage would not report IsEncrypted
			return nil, pgperrors.StructuralError("not an encrypted message")
		}
	}
}

func (m *Message) Decrypt(body io.Reader) (plain []byte, nonce []byte, peer node.Node, err error) {
	raw, err := ioutil.ReadAll(body)
	if err != nil {
		return nil, nil, nil, crypto.ErrDecryptionFailed
	}
	if !m.keyring.fresh() {
		return m.inner.Decrypt(bytes.NewReader(raw)) // device table not known to equal the keyring (a failed upload)
	}
	seq, err := m.openContainer(raw)
	if err != nil {
		// unencrypted or undecryptable: let the reference classify it (ErrDecryptionFailed / ErrInvalidTransportSecurityData)
		return m.inner.Decrypt(bytes.NewReader(raw))
	}
	out := make([]byte, len(seq)+1)
	var st, fl C.uint8_t
	var signer, peerId, n C.uint64_t
	var fname [256]C.uint8_t
	rc := C.bftkv_gpu_batcher_message_verify(m.g.batcher, ptr(seq), C.uint64_t(len(seq)), &st, &signer, &peerId, ptr(out), C.uint64_t(len(out)), &n, &fname[0], &fl)
	if rc != 0 {
		return nil, nil, nil, m.g.infra(rc, "message_verify") // never a verdict
	}
	switch st {
	case C.BFTKV_MSG_READ_ERROR:
		return nil, nil, nil, crypto.ErrDecryptionFailed // crypto_pgp.go:455-457
	case C.BFTKV_MSG_NOT_SIGNED:
		return nil, nil, nil, crypto.ErrInvalidTransportSecurityData // crypto_pgp.go:458-460
	case C.BFTKV_MSG_UNSUPPORTED:
		return m.inner.Decrypt(bytes.NewReader(raw)) // fenced shape (compressed, text-mode, v3, ...): x/crypto decides
	}
	plain = out[:int(n)]
	nonce, err = base64.StdEncoding.DecodeString(C.GoStringN((*C.char)(unsafe.Pointer(&fname[0])), C.int(fl)))
	if err != nil {
		return nil, nil, nil, err // crypto_pgp.go:466-468
	}
	if peerId != 0 {
		peer = m.keyring.GetCertById(uint64(peerId)) // crypto_pgp.go:469 (may be nil)
	}
	if st == C.BFTKV_MSG_SIGNATURE_ERROR {
		return plain, nonce, peer, pgperrors.SignatureError("transport signature") // m.SignatureError != nil: server.go:564 refuses
	}
	// BFTKV_MSG_OK -- and BFTKV_MSG_UNVERIFIED exactly as the reference: signed by a key the keyring does not hold yet gives a
	// NIL error with peer == nil (crypto_pgp.go:458 comment; join requests rely on it)
	return plain, nonce, peer, nil
}
