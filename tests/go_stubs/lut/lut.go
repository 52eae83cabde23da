// Stand-in for go-tfhe's lut package on the GPU box (tests/go_stubs/README.md): the table type only (the generator is host code the
// GPU-box test does not call).
package lut

import "github.com/thedonutfactory/go-tfhe/trlwe"

type LookUpTable struct {
	Poly *trlwe.TRLWELv1
}

func NewLookUpTable() *LookUpTable {
	return &LookUpTable{Poly: trlwe.NewTRLWELv1()}
}
