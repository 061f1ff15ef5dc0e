//go:build !mi355x

// The API of kzg_mi355x.go for builds without the tag: a ResidentProvingKey is the plain ProvingKey, every method is the
// package's own function - callers compile either way and nothing of the reference changes.
package kzg

import (
	"io"
	"os"

	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

// ResidentProvingKey is the caller's ProvingKey in this build.
type ResidentProvingKey struct{ host ProvingKey }

// NewResidentProvingKey wraps pk (windowTables has no meaning here).
func NewResidentProvingKey(pk ProvingKey, windowTables bool) (*ResidentProvingKey, error) {
	if len(pk.G1) == 0 {
		return nil, ErrMinSRSSize
	}
	return &ResidentProvingKey{host: pk}, nil
}

// ReadDumpResident is (*SRS).ReadDump on the file (kzg/marshal.go:98-113).
func ReadDumpResident(path string, windowTables bool, maxPkPoints ...int) (*ResidentProvingKey, *VerifyingKey, error) {
	f, err := os.Open(path)
	if err != nil {
		return nil, nil, err
	}
	defer f.Close()
	var srs SRS
	if err := srs.ReadDump(f, maxPkPoints...); err != nil {
		return nil, nil, err
	}
	return &ResidentProvingKey{host: srs.Pk}, &srs.Vk, nil
}

// ReadFromResident is (*ProvingKey).ReadFrom, or UnsafeReadFrom without the subgroup checks (kzg/marshal.go:140-158).
func ReadFromResident(r io.Reader, windowTables bool, subgroupCheck bool) (*ResidentProvingKey, int64, error) {
	var pk ProvingKey
	var n int64
	var err error
	if subgroupCheck {
		n, err = pk.ReadFrom(r)
	} else {
		n, err = pk.UnsafeReadFrom(r)
	}
	if err != nil {
		return nil, n, err
	}
	return &ResidentProvingKey{host: pk}, n, nil
}

// Size is the number of points of the key.
func (rk *ResidentProvingKey) Size() int { return len(rk.host.G1) }

// Commit is Commit(p, pk) (kzg.go:159-176).
func (rk *ResidentProvingKey) Commit(p []fr.Element, nbTasks ...int) (Digest, error) {
	return Commit(p, rk.host, nbTasks...)
}

// CommitBatch commits one polynomial after the other.
func (rk *ResidentProvingKey) CommitBatch(ps [][]fr.Element) ([]Digest, error) {
	out := make([]Digest, len(ps))
	for i, p := range ps {
		d, err := Commit(p, rk.host)
		if err != nil {
			return nil, err
		}
		out[i] = d
	}
	return out, nil
}

// Release does nothing in this build.
func (rk *ResidentProvingKey) Release() {}
