//go:build mi355x

// MI355X drop-in for (*G1Jac).MultiExp / (*G2Jac).MultiExp of package bw6761.
//
// Go build constraints select FILES, not methods, so the patch to the package has three parts (integration/go/README.md):
//  1. in the generated multiexp.go the two methods (*G1Jac).MultiExp (:32) and (*G2Jac).MultiExp (:306) are renamed
//     multiExpCPU (one line each in internal/generator/ecc/template/multiexp.go.tmpl:247; the recursive calls at :350/:353
//     follow the rename) - the reference's own code, always compiled;
//  2. multiexp_purego.go (//go:build !mi355x) keeps the exported methods as one-line wrappers around multiExpCPU;
//  3. this file (//go:build mi355x) provides them on the device and hands calls below MinDevicePoints to multiExpCPU.
//
// One precondition beyond the reference's (round 6): the device splits every scalar with the curve's endomorphism (the package's
// own mulGLV does the same for single multiplications), which is the integer combination only on points of the r-torsion -
// every point that passed IsInSubGroup / a Decoder with its default checks, and every curve point at all where the cofactor is 1
// (BN254 G1). A caller that feeds MultiExp curve points OUTSIDE the subgroup and relies on the plain combination adds
//
//	func init() { C.gmsm_set_option(C.GMSM_OPT_GLV, 0) }
//
// to this file (INTEGRATION.md section 1).
//
// NOT compiled in the build environment of this repository (no Go toolchain there); the behaviour of the C entry
// points it calls is covered by tests/ through the same C ABI.  See INTEGRATION.md.
package bw6761

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
	"github.com/consensys/gnark-crypto/ecc/bw6-761/fr"
)

// MinDevicePoints is the smallest MultiExp that goes to the GPU; shorter ones run the package's own Go code
// (multiExpCPU). The device call has a floor of about 0.15 ms (one fused kernel launch + the 254 host-side doublings of
// the window fold, INTEGRATION.md section 2c): measured on the MI355X box against the C port of this package's
// algorithm, the device is ahead from 2^5 points up with either NbTasks setting (bench.py small_n); below that a
// handful of scalar multiplications on one core is as fast and keeps the GPU out of the picture.
var MinDevicePoints = 32

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
	if len(points) < MinDevicePoints { // includes len == 0: the reference returns infinity
		return p.multiExpCPU(points, scalars, config)
	}
	rc := C.gmsm_bw6_761_g1_multiexp(
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
	if len(points) < MinDevicePoints {
		return p.multiExpCPU(points, scalars, config)
	}
	rc := C.gmsm_bw6_761_g2_multiexp(
		(*C.uint64_t)(unsafe.Pointer(&points[0])), C.size_t(len(points)),
		(*C.uint64_t)(unsafe.Pointer(&scalars[0])), C.size_t(len(scalars)),
		C.int(config.NbTasks), (*C.uint64_t)(unsafe.Pointer(p)))
	if rc != 0 {
		return nil, gmsmError(rc)
	}
	return p, nil
}

// UseDevices asks the library to spread every MultiExp of 2^17 points or more over the listed GPUs of the node (one
// entry per logical rank). Spreading is opt-in: a process that never calls this - and has no GMSM_DEVICES in its
// environment - runs on one device and touches no other. UseDevices(nil) goes back to that.
func UseDevices(devices []int) error {
	if len(devices) == 0 {
		if rc := C.gmsm_set_devices(nil, 0); rc != 0 {
			return gmsmError(rc)
		}
		return nil
	}
	list := make([]C.int, len(devices))
	for i, d := range devices {
		list[i] = C.int(d)
	}
	if rc := C.gmsm_set_devices(&list[0], C.int(len(list))); rc != 0 {
		return gmsmError(rc)
	}
	return nil
}

// TrimDeviceMemory gives the scratch the library grew for its largest call so far back to the devices (buffers of the
// workspaces that are idle right now, above keepBytes each); registered bases and FFT domains stay. Returns the bytes freed.
func TrimDeviceMemory(keepBytes uint64) (uint64, error) {
	var freed C.size_t
	if rc := C.gmsm_trim(C.size_t(keepBytes), &freed); rc != 0 {
		return 0, gmsmError(rc)
	}
	return uint64(freed), nil
}

// Shutdown releases everything the library holds on every device (call it when no MultiExp is running; outstanding
// tickets are refused). The library stays usable: its state reappears with the next call.
func Shutdown() error {
	if rc := C.gmsm_shutdown(); rc != 0 {
		return gmsmError(rc)
	}
	return nil
}
