//go:build mi355x

// Resident proving key for package kzg on an MI355X: the SRS points live in HBM (gmsm_bases_register / gmsm_bases_register_dump),
// every commitment is one gmsm_multiexp_bases over them - the 64 bytes per point of the proving key cross PCIe once, not once
// per Commit as through the MultiExp drop-in (ecc/bn254/multiexp_mi355x.go), which already serves the package's own
//
//	Commit(p, pk)                       kzg.go:159-176  (res.MultiExp(pk.G1[:len(p)], p, config))
//	Open / BatchOpenSinglePoint         kzg.go:180, :246 (Commit of the quotient)
//
// unchanged. This file adds a type next to ProvingKey and touches nothing of the reference (kzg_purego.go gives the same API
// over the plain ProvingKey for builds without the tag, so callers compile either way):
//
//	NewResidentProvingKey(pk, tables)   gmsm_bases_register (+ gmsm_bases_precompute: window tables, one bucket set per commitment)
//	ReadDumpResident(path, max...)      SRS.ReadDump's twin (kzg/marshal.go:98-113): Vk through ReadFrom, the []G1Affine memory of
//	                                    the dump streamed from the file into HBM (gmsm_bases_register_dump) - it never exists as
//	                                    host limbs
//	ReadFromResident(r, tables)         (*ProvingKey).ReadFrom's twin (kzg/marshal.go:140-147): the Encoder's default stream - uint32
//	                                    length, then COMPRESSED points - goes to the device as bytes; Y = sqrt(X^3 + b) and the
//	                                    subgroup checks of Decoder.Decode's []G1Affine case (marshal.go of the curve package; bn254's lines 220-277: unsafeComputeY in parallel) run
//	                                    there (gmsm_bases_register_compressed); a raw stream (WriteRawTo) takes
//	                                    gmsm_bases_register_raw. subgroupCheck = false is UnsafeReadFrom (:151-158)
//	(*ResidentProvingKey).Commit        gmsm_multiexp_bases; below MinDevicePoints the package's Commit when the host copy exists
//	(*ResidentProvingKey).CommitBatch   k polynomials of equal length in one call, two MSMs in flight (gmsm_multiexp_bases_batch)
//	(*ResidentProvingKey).Release       gmsm_bases_release (also the finalizer)
//
// NOT compiled in the build environment of this repository (no Go toolchain there); the C entry points it calls are covered
// by tests/ through the same C ABI, tests/test_go_stubs.py checks tags, package, symbols and the cited lines. See INTEGRATION.md 2b.
package kzg

/*
#cgo CFLAGS: -I${SRCDIR}/../../../third_party/gmsm/include
#cgo LDFLAGS: -L${SRCDIR}/../../../third_party/gmsm/lib -lgmsm -Wl,-rpath,${SRCDIR}/../../../third_party/gmsm/lib
#include <stdlib.h>
#include "gmsm.h"
*/
import "C"

import (
	"encoding/binary"
	"errors"
	"io"
	"os"
	"runtime"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

// compressedFlagMask: every compressed flag value has the top bit of the first byte set and no uncompressed one has
// (marshal.go:26-30: 10 / 11 against 00 / 01 = infinity, which Bytes() writes over a compressed-size buffer - a proving key
// holds no point at infinity; bls12-381/marshal.go:25-35: 100 / 101 / 110 against 000 / 010).
const compressedFlagMask = 0x80

// MinDevicePoints is the shortest polynomial that is committed on the device when a host copy of the key exists
// (same measured crossover as bn254.MinDevicePoints, INTEGRATION.md section 2c).
var MinDevicePoints = 32

// ResidentProvingKey is a ProvingKey whose G1 points are registered with the device.
type ResidentProvingKey struct {
	handle C.uint64_t
	n      int
	host   ProvingKey // the caller's key (shared, not copied); empty after ReadDumpResident
}

func gmsmErr() error { return errors.New("gmsm: " + C.GoString(C.gmsm_last_error())) }

// NewResidentProvingKey copies pk.G1 to the device once. windowTables: also build 2^(c w) P_i for every window next to the
// SRS (gmsm_bases_precompute: 15 copies for BN254 G1 at 2^20; commitments of 2^13..2^21 coefficients then fill one bucket
// set - 2^20: 1.94 -> 1.56 ms - and calls of up to 2^12 need no host-side fold). Call it before the key is shared.
func NewResidentProvingKey(pk ProvingKey, windowTables bool) (*ResidentProvingKey, error) {
	if len(pk.G1) == 0 {
		return nil, ErrMinSRSSize
	}
	rk := &ResidentProvingKey{n: len(pk.G1), host: pk}
	if rc := C.gmsm_bases_register(C.GMSM_BN254_G1, (*C.uint64_t)(unsafe.Pointer(&pk.G1[0])), nil, C.size_t(len(pk.G1)), &rk.handle); rc != 0 {
		return nil, gmsmErr()
	}
	if windowTables {
		if rc := C.gmsm_bases_precompute(rk.handle, 0); rc != 0 {
			C.gmsm_bases_release(rk.handle)
			return nil, gmsmErr()
		}
	}
	runtime.SetFinalizer(rk, func(k *ResidentProvingKey) { k.Release() })
	return rk, nil
}

type countingReader struct {
	r io.Reader
	n int64
}

func (c *countingReader) Read(p []byte) (int, error) {
	k, err := c.r.Read(p)
	c.n += int64(k)
	return k, err
}

// ReadDumpResident reads what (*SRS).WriteDump wrote (kzg/marshal.go:65-95): the VerifyingKey through ReadFrom, as ReadDump
// does, then marker | length | raw []G1Affine memory (utils/unsafe/dump_slice.go:16-32, :80) from the file straight into HBM.
// Like ReadDump it validates nothing. maxPkPoints as in ReadDump.
func ReadDumpResident(path string, windowTables bool, maxPkPoints ...int) (*ResidentProvingKey, *VerifyingKey, error) {
	f, err := os.Open(path)
	if err != nil {
		return nil, nil, err
	}
	defer f.Close()
	var vk VerifyingKey
	cr := &countingReader{r: f}
	if _, err := vk.ReadFrom(cr); err != nil { // ReadFrom reads exactly the key's encoding (an unbuffered decoder)
		return nil, nil, err
	}
	max := 0
	if len(maxPkPoints) > 0 && maxPkPoints[0] > 0 {
		max = maxPkPoints[0]
	}
	cpath := C.CString(path)
	defer C.free(unsafe.Pointer(cpath))
	rk := &ResidentProvingKey{}
	var n C.size_t
	var bad C.int64_t
	if rc := C.gmsm_bases_register_dump(C.GMSM_BN254_G1, cpath, C.uint64_t(cr.n), 1, C.size_t(max), 0, &rk.handle, &n, &bad); rc != 0 {
		return nil, nil, gmsmErr()
	}
	rk.n = int(n)
	if windowTables {
		if rc := C.gmsm_bases_precompute(rk.handle, 0); rc != 0 {
			C.gmsm_bases_release(rk.handle)
			return nil, nil, gmsmErr()
		}
	}
	runtime.SetFinalizer(rk, func(k *ResidentProvingKey) { k.Release() })
	return rk, &vk, nil
}

// ReadFromResident reads what (*ProvingKey).WriteTo / WriteRawTo wrote (kzg/marshal.go:16-32): a big-endian uint32 length and
// the points, compressed (the Encoder's default: X and the flag of Y's half, marshal.go:801-823) or raw. The bytes go to the
// device as they are; decompression - the square root per point that Decoder.Decode spreads over the cores (bn254's marshal.go:258-272)
// - and the subgroup checks run there. subgroupCheck = false is (*ProvingKey).UnsafeReadFrom. A stream that mixes both
// encodings (no Encoder writes one) is refused.
func ReadFromResident(r io.Reader, windowTables bool, subgroupCheck bool) (*ResidentProvingKey, int64, error) {
	var hdr [4]byte
	if _, err := io.ReadFull(r, hdr[:]); err != nil {
		return nil, 0, err
	}
	n := int(binary.BigEndian.Uint32(hdr[:]))
	if n == 0 {
		return nil, 4, ErrMinSRSSize
	}
	first := make([]byte, bn254.SizeOfG1AffineCompressed)
	if _, err := io.ReadFull(r, first); err != nil {
		return nil, 4, err
	}
	size := bn254.SizeOfG1AffineCompressed
	compressed := first[0]&compressedFlagMask != 0
	if !compressed {
		size = bn254.SizeOfG1AffineUncompressed
	}
	buf := make([]byte, n*size)
	copy(buf, first)
	if _, err := io.ReadFull(r, buf[len(first):]); err != nil {
		return nil, 4 + int64(len(first)), err
	}
	check := C.int(0)
	if subgroupCheck {
		check = 2
	}
	rk := &ResidentProvingKey{n: n}
	var bad C.int64_t
	var rc C.int
	if compressed {
		rc = C.gmsm_bases_register_compressed(C.GMSM_BN254_G1, (*C.uint8_t)(unsafe.Pointer(&buf[0])), C.size_t(n), check, &rk.handle, &bad)
	} else {
		rc = C.gmsm_bases_register_raw(C.GMSM_BN254_G1, (*C.uint8_t)(unsafe.Pointer(&buf[0])), C.size_t(n), check, &rk.handle, &bad)
	}
	if rc != 0 {
		return nil, 4 + int64(len(buf)), gmsmErr()
	}
	if windowTables {
		if rc := C.gmsm_bases_precompute(rk.handle, 0); rc != 0 {
			C.gmsm_bases_release(rk.handle)
			return nil, 4 + int64(len(buf)), gmsmErr()
		}
	}
	runtime.SetFinalizer(rk, func(k *ResidentProvingKey) { k.Release() })
	return rk, 4 + int64(len(buf)), nil
}

// Size is the number of registered points.
func (rk *ResidentProvingKey) Size() int { return rk.n }

// Commit commits to a polynomial over the resident SRS: Commit(p, pk) of kzg.go:159-176 without the copy of pk.
func (rk *ResidentProvingKey) Commit(p []fr.Element, nbTasks ...int) (Digest, error) {
	if len(p) == 0 || len(p) > rk.n {
		return Digest{}, ErrInvalidPolynomialSize
	}
	if len(p) < MinDevicePoints && len(rk.host.G1) >= len(p) {
		return Commit(p, rk.host, nbTasks...)
	}
	tasks := 0
	if len(nbTasks) > 0 {
		tasks = nbTasks[0]
	}
	var jac bn254.G1Jac
	if rc := C.gmsm_multiexp_bases(rk.handle, (*C.uint64_t)(unsafe.Pointer(&p[0])), C.size_t(len(p)), C.int(tasks),
		(*C.uint64_t)(unsafe.Pointer(&jac))); rc != 0 {
		if rc == 2 {
			return Digest{}, errors.New("invalid config: config.NbTasks > 1024")
		}
		return Digest{}, gmsmErr()
	}
	var res bn254.G1Affine
	res.FromJacobian(&jac)
	runtime.KeepAlive(rk)
	return res, nil
}

// CommitBatch commits to k polynomials of the same length in one blocking call that keeps two MultiExp in flight (what
// BatchOpenSinglePoint's callers and provers with many wires do one Commit at a time, kzg.go:246): the scalars of polynomial
// i + 1 cross PCIe while polynomial i is being accumulated.
func (rk *ResidentProvingKey) CommitBatch(ps [][]fr.Element) ([]Digest, error) {
	if len(ps) == 0 {
		return nil, nil
	}
	n := len(ps[0])
	if n == 0 || n > rk.n {
		return nil, ErrInvalidPolynomialSize
	}
	flat := make([]fr.Element, 0, n*len(ps))
	for _, p := range ps {
		if len(p) != n {
			return nil, ErrInvalidPolynomialSize
		}
		flat = append(flat, p...)
	}
	jacs := make([]bn254.G1Jac, len(ps))
	if rc := C.gmsm_multiexp_bases_batch(rk.handle, (*C.uint64_t)(unsafe.Pointer(&flat[0])), nil, C.size_t(n), C.size_t(len(ps)), nil,
		(*C.uint64_t)(unsafe.Pointer(&jacs[0]))); rc != 0 {
		return nil, gmsmErr()
	}
	runtime.KeepAlive(rk)
	return bn254.BatchJacobianToAffineG1(jacs), nil
}

// Release gives the device memory of the key back (gmsm_bases_release); calls that are still running keep their own reference.
func (rk *ResidentProvingKey) Release() {
	if rk.handle != 0 {
		C.gmsm_bases_release(rk.handle)
		rk.handle = 0
		runtime.SetFinalizer(rk, nil)
	}
}
