// Stand-in for go-tfhe's trlwe package on the GPU box (tests/go_stubs/README.md).
package trlwe

import "github.com/thedonutfactory/go-tfhe/params"

type TRLWELv1 struct {
	A []params.Torus
	B []params.Torus
}

func NewTRLWELv1() *TRLWELv1 {
	n := params.GetTRGSWLv1().N
	return &TRLWELv1{A: make([]params.Torus, n), B: make([]params.Torus, n)}
}
