// Package pgpgpu plugs libbftkv_gpu.so (include/bftkv_gpu.h) into bftkv's crypto.Crypto bundle
// (crypto/crypto.go:103-111): Signature, CollectiveSignature and Message come from the GPU library, everything
// that needs a private key -- and every input shape the library fences -- stays with crypto/pgp.
//
// Use it where the reference calls pgp.New() (cmd/bftkv/main.go:66, api/api.go:37, scripts/test.go:61):
//
//	crypt := pgpgpu.New(0)          // device ordinal
//
// This file set is written against bftkv @ go.mod:8 (golang.org/x/crypto v0.0.0-20191227163750-53104e6ec876) and
// needs shim/patches/0001-wotqs-export-cliques.patch.  It is NOT compiled in this repository (the build image has no
// Go toolchain); the same C entry points are driven by tests/ through ctypes and by tests/c_harness/harness.c.
package pgpgpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -lbftkv_gpu
#include <stdlib.h>
#include "bftkv_gpu.h"
*/
import "C"

import (
	gocrypto "crypto"
	"errors"
	"sync"
	"unsafe"

	"github.com/yahoo/bftkv/crypto"
	"github.com/yahoo/bftkv/crypto/pgp"
)

// gpu is one verifier context plus its micro-batcher (bftkv_gpu_batcher_*: the reference verifies one message per
// goroutine, transport/http/http.go:85,143; the batcher turns concurrent calls into device batches).
type gpu struct {
	ctx     *C.bftkv_gpu_ctx
	batcher *C.bftkv_gpu_batcher

	qmu    sync.Mutex
	quorum map[string]*qentry // quorum descriptor -> handle, see quorum.go
	qorder []string
}

var errNoDevice = errors.New("pgpgpu: no usable MI355X (bftkv_gpu_init failed); there is no CPU fallback inside the library")

// Options are the knobs a deployment may want to set before the first keyring upload.
type Options struct {
	// DSATableBudget bounds the HBM the DSA fixed-base tables may hold, in bytes (bftkv_gpu_set_dsa_table_budget).  0: the
	// library's free-memory policy (up to 45 % of the free HBM: 76.5 GB for 32 DSA keys on an idle MI355X).  A replica that
	// shares its GPU sets it: 8 GB gives 32 keys 14-bit tables (37 multiplications per signature instead of 29).
	DSATableBudget uint64
}

// New mirrors pgp.New() (crypto/pgp/crypto_pgp.go:583-593).
func New(device int) *crypto.Crypto {
	return NewWithOptions(device, Options{})
}

// NewWithOptions is New with the deployment's knobs.
func NewWithOptions(device int, opt Options) *crypto.Crypto {
	c := pgp.New()
	g := &gpu{quorum: make(map[string]*qentry)}
	if rc := C.bftkv_gpu_init(C.int(device), &g.ctx); rc != 0 {
		panic(errNoDevice)
	}
	if opt.DSATableBudget != 0 {
		if rc := C.bftkv_gpu_set_dsa_table_budget(g.ctx, C.uint64_t(opt.DSATableBudget)); rc != 0 {
			panic("pgpgpu: bftkv_gpu_set_dsa_table_budget failed")
		}
	}
	// at most 256 calls per batch; lanes = 0: the library's measured default (3 on MI355X, profiles/r03_serving_batcher_lanes*;
	// BFTKV_BATCHER_LANES overrides it)
	g.batcher = C.bftkv_gpu_batcher_create_lanes(g.ctx, 256, 0, 0)
	if g.batcher == nil {
		panic("pgpgpu: bftkv_gpu_batcher_create failed")
	}
	// MD5 / RIPEMD-160: openpgp's hashForSignature refuses a hash this BINARY does not link ("hash not available").  Only
	// the binary knows -- tell the library, which otherwise fences every signature naming them (bftkv_gpu_set_hash_policy).
	avail := func(h gocrypto.Hash) C.int {
		if h.Available() {
			return 1
		}
		return 2
	}
	C.bftkv_gpu_set_hash_policy(g.ctx, 1, avail(gocrypto.MD5))
	C.bftkv_gpu_set_hash_policy(g.ctx, 3, avail(gocrypto.RIPEMD160))
	kr := &keyring{inner: c.Keyring, g: g}
	c.Keyring = kr
	// crypto/pgp's other objects were built around the original keyring by pgp.New(); they keep using it
	// (same underlying *PGPKeyring), this wrapper only observes Register / Remove.
	c.Signature = &Signature{g: g, inner: c.Signature, keyring: kr}
	c.CollectiveSignature = &CollectiveSignature{g: g, inner: c.CollectiveSignature, keyring: kr}
	c.Message = &Message{g: g, inner: c.Message, keyring: kr}
	return c
}

// Close releases the device context (the reference has no teardown hook; call it from main's defer).
func Close(c *crypto.Crypto) {
	if kr, ok := c.Keyring.(*keyring); ok {
		C.bftkv_gpu_batcher_destroy(kr.g.batcher)
		C.bftkv_gpu_destroy(kr.g.ctx)
	}
}

// BatcherHandle returns the micro-batcher of a bundle built by New, as an opaque pointer (a *C.bftkv_gpu_batcher of THIS
// package is a different Go type in any other package: cgo types are package-local), or nil for a bundle of crypto/pgp.
// crypto/thresholdgpu hands its one-operation-per-call share combines to the same batcher, so that config-5 traffic and
// verification traffic share the lanes of one device.
func BatcherHandle(c *crypto.Crypto) unsafe.Pointer {
	if kr, ok := c.Keyring.(*keyring); ok {
		return unsafe.Pointer(kr.g.batcher)
	}
	return nil
}

// ptr returns the address of the first byte (nil for an empty slice).  The library reads the buffer for the duration of
// the call only and keeps nothing (cgo pointer rules).
func ptr(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// infra turns a negative return code into an error that is NOT one of the crypto.Err* identities: callers such as
// protocol.Server answer it as an internal failure, never as a verdict.
func (g *gpu) infra(rc C.int, what string) error {
	if rc == 0 {
		return nil
	}
	return errors.New("pgpgpu: " + what + ": " + C.GoString(C.bftkv_gpu_last_error(g.ctx)))
}
