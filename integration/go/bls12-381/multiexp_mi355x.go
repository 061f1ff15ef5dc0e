//go:build mi355x

// MI355X drop-in for (*G1Jac).MultiExp / (*G2Jac).MultiExp of package bls12381
// (replaces the bodies in ecc/bls12-381/multiexp.go:32 and :355 after the two argument checks).
// NOT compiled in the build environment of this repository (no Go toolchain there); the behaviour of the C entry
// points it calls is covered by tests/ through the same C ABI.  See INTEGRATION.md.
package bls12381

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/gmsm/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/gmsm/lib -lgmsm -Wl,-rpath,${SRCDIR}/../../third_party/gmsm/lib
#include "gmsm.h"
*/
import "C"

import (
	"errors"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark-crypto/ecc/bls12-381/fr"
)

func gmsmError(rc C.int) error {
	switch rc {
	case 1:
		return errors.New("len(points) != len(scalars)")
	case 2:
		return errors.New("invalid config: config.NbTasks > 1024")
	default:
		return errors.New("gmsm: " + C.GoString(C.gmsm_last_error()))
	}
}

// MultiExp implements section 4 of https://eprint.iacr.org/2012/549.pdf on an MI355X.
//
// This call return an error if len(scalars) != len(points) or if provided config is invalid.
func (p *G1Jac) MultiExp(points []G1Affine, scalars []fr.Element, config ecc.MultiExpConfig) (*G1Jac, error) {
	if len(points) != len(scalars) {
		return nil, errors.New("len(points) != len(scalars)")
	}
	if config.NbTasks > 1024 {
		return nil, errors.New("invalid config: config.NbTasks > 1024")
	}
	if len(points) == 0 {
		p.Set(&g1Infinity)
		return p, nil
	}
	rc := C.gmsm_bls12_381_g1_multiexp(
		(*C.uint64_t)(unsafe.Pointer(&points[0])), C.size_t(len(points)),
		(*C.uint64_t)(unsafe.Pointer(&scalars[0])), C.size_t(len(scalars)),
		C.int(config.NbTasks), (*C.uint64_t)(unsafe.Pointer(p)))
	if rc != 0 {
		return nil, gmsmError(rc)
	}
	return p, nil
}

// MultiExp implements section 4 of https://eprint.iacr.org/2012/549.pdf on an MI355X.
func (p *G2Jac) MultiExp(points []G2Affine, scalars []fr.Element, config ecc.MultiExpConfig) (*G2Jac, error) {
	if len(points) != len(scalars) {
		return nil, errors.New("len(points) != len(scalars)")
	}
	if config.NbTasks > 1024 {
		return nil, errors.New("invalid config: config.NbTasks > 1024")
	}
	if len(points) == 0 {
		p.Set(&g2Infinity)
		return p, nil
	}
	rc := C.gmsm_bls12_381_g2_multiexp(
		(*C.uint64_t)(unsafe.Pointer(&points[0])), C.size_t(len(points)),
		(*C.uint64_t)(unsafe.Pointer(&scalars[0])), C.size_t(len(scalars)),
		C.int(config.NbTasks), (*C.uint64_t)(unsafe.Pointer(p)))
	if rc != 0 {
		return nil, gmsmError(rc)
	}
	return p, nil
}
