// Package thresholdgpu puts BASELINE config 5 -- the share-combine arithmetic of crypto/threshold and crypto/sss -- behind
// the reference's own seam.  crypto.Threshold / crypto.ThresholdProcess (crypto/crypto.go:92-101) keep all of their
// bookkeeping in Go; shim/patches/0003-threshold-combine-hooks.patch makes the big-number arithmetic they end in
// replaceable, and Install points those hooks at libbftkv_gpu.so's micro-batcher, ONE operation per call:
//
//	rsaProc.ProcessResponse -> calculateSignature   (crypto/threshold/rsa/rsa.go:235-253, 318-329)  bftkv_gpu_batcher_modmul_product
//	dsaProc.ProcessResponse -> CalculateR           (crypto/threshold/dsa/dsa_core.go:333-341, dsa.go:33-52)  bftkv_gpu_batcher_dsa_calculate_r
//	dsaProc.ProcessResponse -> calculateS           (dsa_core.go:351-360, 389-403)                  bftkv_gpu_batcher_lagrange_combine
//	SSSProcess.ProcessResponse -> calculateSecret   (crypto/sss/sss.go:69-92)                       bftkv_gpu_batcher_lagrange_combine
//	rsaContext.Sign / CalculatePartialR             (rsa.go:161-171, dsa.go:27-31; Options.SecretExponents)  bftkv_gpu_batcher_modexp
//
// One goroutine per DistSign (protocol/client.go:509-546) and per distSign request (protocol/server.go:528-541) calls them;
// the batcher gathers the concurrent callers of one shape into a device call.  Every hook answers nil whenever the library
// makes no claim -- an infrastructure error (the library fails closed: the status byte starts out as BFTKV_TH_FAILED), an
// input it fences (status != BFTKV_TH_OK: no modular inverse, Lagrange integers beyond 2128 bits), an even or over-wide modulus,
// a negative or over-long operand -- and the reference's arithmetic then runs exactly as it always did.
//
//	crypt := pgpgpu.New(0)
//	if err := thresholdgpu.Install(crypt, thresholdgpu.Options{}); err != nil { ... }
//
// NOT compiled in this repository (no Go toolchain in the build image); the C entry points are driven by
// tests/test_gpu_threshold.py (256 threads, one operation per call, against oracle/c/threshold.c and the reference's known
// answers) and tests/c_harness/harness.c.
package thresholdgpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -lbftkv_gpu
#include "bftkv_gpu.h"
*/
import "C"

import (
	"errors"
	"math/big"
	"unsafe"

	"github.com/yahoo/bftkv/crypto"
	"github.com/yahoo/bftkv/crypto/sss"
	thdsa "github.com/yahoo/bftkv/crypto/threshold/dsa"
	thrsa "github.com/yahoo/bftkv/crypto/threshold/rsa"

	"github.com/yahoo/bftkv/crypto/pgpgpu"
)

// Options of Install.
type Options struct {
	// SecretExponents also routes m^d_i mod N (rsaContext.Sign) and g^a_i mod p (CalculatePartialR) to the device.  Their
	// exponents are a fragment of the private key and a share of the signing nonce: whether those may leave the host is the
	// deployment's decision, so the default keeps both on the CPU.
	SecretExponents bool
}

const (
	maxModBytes   = 256 // the kernels hold numbers of up to 2048 bits (BFTKV_E_UNSUPPORTED beyond)
	maxOrderBytes = 32  // CalculateR: q of up to 256 bits
	maxTerms      = 1024
)

var errNoBatcher = errors.New("thresholdgpu: the crypto bundle was not built by pgpgpu.New (no device batcher to hand operations to)")

// Install points the hooks of patch 0003 at the batcher of a bundle built by pgpgpu.New.  The hooks are package-level
// variables of the reference packages: install once per process, before the first DistSign.
func Install(crypt *crypto.Crypto, opt Options) error {
	h := pgpgpu.BatcherHandle(crypt)
	if h == nil {
		return errNoBatcher
	}
	b := (*C.bftkv_gpu_batcher)(h)
	sss.CombineHook = func(coords []*sss.Coordinate, m *big.Int) *big.Int { return lagrangeCombine(b, coords, m) }
	thrsa.CombineHook = func(psigs []*big.Int, N *big.Int) *big.Int { return product(b, psigs, N) }
	thdsa.CalculateRHook = func(rs []*thdsa.PartialR, p, q *big.Int) *big.Int { return calculateR(b, rs, p, q) }
	if opt.SecretExponents {
		thrsa.ModExpHook = func(base, exp, N *big.Int) *big.Int { return modExp(b, base, exp, N) }
		thdsa.ModExpHook = func(base, exp, p *big.Int) *big.Int { return modExp(b, base, exp, p) }
	}
	return nil
}

// Uninstall restores the reference's arithmetic (call it before pgpgpu.Close destroys the batcher).
func Uninstall() {
	sss.CombineHook = nil
	thrsa.CombineHook = nil
	thdsa.CalculateRHook = nil
	thrsa.ModExpHook = nil
	thdsa.ModExpHook = nil
}

// width returns the byte length of an odd modulus of at most limit bytes, or 0 (the hook then makes no claim).
func width(m *big.Int, limit int) int {
	if m == nil || m.Sign() <= 0 || m.Bit(0) == 0 || m.BitLen() < 2 {
		return 0
	}
	n := (m.BitLen() + 7) / 8
	if n > limit {
		return 0
	}
	return n
}

// put writes x mod m, big-endian and left-padded, into dst (len(dst) = the modulus' width).  The reference's arithmetic
// reduces every product and sum mod m, so a residue gives the same result as the value itself; a negative or missing
// operand is left to the reference (false).
func put(dst []byte, x, m *big.Int) bool {
	if x == nil || x.Sign() < 0 {
		return false
	}
	if x.Cmp(m) >= 0 {
		x = new(big.Int).Mod(x, m)
	}
	raw := x.Bytes() // (Go 1.13, go.mod:3: no FillBytes yet)
	copy(dst[len(dst)-len(raw):], raw)
	return true
}

func bptr(b []byte) *C.uint8_t { return (*C.uint8_t)(unsafe.Pointer(&b[0])) }

// product replaces the fold of calculateSignature (rsa.go:318-329): prod psigs mod N.
func product(b *C.bftkv_gpu_batcher, psigs []*big.Int, N *big.Int) *big.Int {
	nb := width(N, maxModBytes)
	k := len(psigs)
	if nb == 0 || k == 0 || k > maxTerms {
		return nil
	}
	factors := make([]byte, k*nb)
	for j, s := range psigs {
		if !put(factors[j*nb:(j+1)*nb], s, N) {
			return nil
		}
	}
	mod := N.Bytes() // exactly nb bytes
	out := make([]byte, nb)
	var st C.uint8_t
	rc := C.bftkv_gpu_batcher_modmul_product(b, C.uint32_t(k), bptr(factors), C.uint32_t(nb), bptr(mod), bptr(out), &st)
	if rc != 0 || st != C.BFTKV_TH_OK {
		return nil
	}
	return new(big.Int).SetBytes(out)
}

// lagrangeCombine replaces SSSProcess.calculateSecret (sss.go:81-92) and calculateS (dsa_core.go:389-403):
// sum_j Lagrange(x_j; xs) * y_j mod m.
func lagrangeCombine(b *C.bftkv_gpu_batcher, coords []*sss.Coordinate, m *big.Int) *big.Int {
	nb := width(m, maxModBytes)
	k := len(coords)
	if nb == 0 || k == 0 || k > maxTerms {
		return nil
	}
	xs := make([]C.int32_t, k)
	ys := make([]byte, k*nb)
	for j, c := range coords {
		if c == nil || c.X != int(int32(c.X)) || !put(ys[j*nb:(j+1)*nb], c.Y, m) {
			return nil
		}
		xs[j] = C.int32_t(c.X)
	}
	mod := m.Bytes()
	out := make([]byte, nb)
	var st C.uint8_t
	rc := C.bftkv_gpu_batcher_lagrange_combine(b, C.uint32_t(k), &xs[0], bptr(ys), C.uint32_t(nb), bptr(mod), bptr(out), &st)
	if rc != 0 || st != C.BFTKV_TH_OK {
		return nil // BFTKV_TH_NO_INVERSE: math/big's ModInverse returns nil there and sss.Lagrange dereferences it -- the reference's own outcome
	}
	return new(big.Int).SetBytes(out)
}

// calculateR replaces dsaGroupOperations.CalculateR (dsa.go:33-52).
func calculateR(b *C.bftkv_gpu_batcher, rs []*thdsa.PartialR, p, q *big.Int) *big.Int {
	pb, qb := width(p, maxModBytes), width(q, maxOrderBytes)
	k := len(rs)
	if pb == 0 || qb == 0 || k == 0 || k > maxTerms {
		return nil
	}
	xs := make([]C.int32_t, k)
	ri := make([]byte, k*pb)
	vi := make([]byte, k*qb)
	for j, r := range rs {
		if r == nil || r.X != int(int32(r.X)) {
			return nil
		}
		xs[j] = C.int32_t(r.X)
		if !put(ri[j*pb:(j+1)*pb], new(big.Int).SetBytes(r.Ri), p) || !put(vi[j*qb:(j+1)*qb], r.Vi, q) {
			return nil
		}
	}
	pm, qm := p.Bytes(), q.Bytes()
	out := make([]byte, qb)
	var st C.uint8_t
	rc := C.bftkv_gpu_batcher_dsa_calculate_r(b, C.uint32_t(k), &xs[0], bptr(ri), C.uint32_t(pb), bptr(vi), C.uint32_t(qb), bptr(pm), bptr(qm), bptr(out), &st)
	if rc != 0 || st != C.BFTKV_TH_OK {
		return nil
	}
	return new(big.Int).SetBytes(out)
}

// modExp replaces big.Int.Exp(base, exp, m) in rsaContext.Sign (rsa.go:161-171) and CalculatePartialR (dsa.go:27-31).
func modExp(b *C.bftkv_gpu_batcher, base, exp, m *big.Int) *big.Int {
	nb := width(m, maxModBytes)
	if nb == 0 || exp == nil || exp.Sign() < 0 {
		return nil
	}
	// exponents travel left-padded to a multiple of 32 bytes: callers whose exponents differ by a byte still share a device call
	// (operations are grouped by shape), and the chain is as long as the padded width, whatever the value
	raw := exp.Bytes()
	eb := make([]byte, (len(raw)/32+1)*32)
	if len(eb) > 1024 {
		return nil
	}
	copy(eb[len(eb)-len(raw):], raw)
	bs := make([]byte, nb)
	if !put(bs, base, m) {
		return nil
	}
	mod := m.Bytes()
	out := make([]byte, nb)
	var st C.uint8_t
	rc := C.bftkv_gpu_batcher_modexp(b, bptr(bs), C.uint32_t(nb), bptr(eb), C.uint32_t(len(eb)), bptr(mod), bptr(out), &st)
	if rc != 0 || st != C.BFTKV_TH_OK {
		return nil
	}
	return new(big.Int).SetBytes(out)
}
