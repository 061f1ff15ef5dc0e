//go:build !mi355x

// The exported MultiExp methods of the pure-Go build: one-line wrappers around the generated code, which
// integration/go/README.md renames to multiExpCPU so that the mi355x build can provide the same methods from
// multiexp_mi355x.go (Go build constraints select files, not methods: two files may not both define
// (*G1Jac).MultiExp unless their tags exclude each other).
package bls12381

import (
	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark-crypto/ecc/bls12-381/fr"
)

// MultiExp implements section 4 of https://eprint.iacr.org/2012/549.pdf
//
// This call return an error if len(scalars) != len(points) or if provided config is invalid.
func (p *G1Jac) MultiExp(points []G1Affine, scalars []fr.Element, config ecc.MultiExpConfig) (*G1Jac, error) {
	return p.multiExpCPU(points, scalars, config)
}

// MultiExp implements section 4 of https://eprint.iacr.org/2012/549.pdf
//
// This call return an error if len(scalars) != len(points) or if provided config is invalid.
func (p *G2Jac) MultiExp(points []G2Affine, scalars []fr.Element, config ecc.MultiExpConfig) (*G2Jac, error) {
	return p.multiExpCPU(points, scalars, config)
}
