// Stand-in for go-tfhe's tlwe package on the GPU box (tests/go_stubs/README.md).
package tlwe

import "github.com/thedonutfactory/go-tfhe/params"

type TLWELv0 struct {
	P []params.Torus
}

func NewTLWELv0() *TLWELv0 {
	return &TLWELv0{P: make([]params.Torus, params.GetTLWELv0().N+1)}
}

func (t *TLWELv0) B() params.Torus { return t.P[len(t.P)-1] }

func (t *TLWELv0) SetB(val params.Torus) { t.P[len(t.P)-1] = val }

func (t *TLWELv0) Neg() *TLWELv0 {
	r := NewTLWELv0()
	for i := range r.P {
		r.P[i] = 0 - t.P[i]
	}
	return r
}

type TLWELv1 struct {
	P []params.Torus
}

func NewTLWELv1() *TLWELv1 {
	return &TLWELv1{P: make([]params.Torus, params.GetTRGSWLv1().N+1)}
}

func (t *TLWELv1) SetB(val params.Torus) { t.P[len(t.P)-1] = val }
