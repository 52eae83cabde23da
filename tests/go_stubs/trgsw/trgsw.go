// Stand-in for go-tfhe's trgsw package on the GPU box (tests/go_stubs/README.md).
package trgsw

import "github.com/thedonutfactory/go-tfhe/poly"

type TRLWELv1FFT struct {
	A poly.FourierPoly
	B poly.FourierPoly
}

type TRGSWLv1FFT struct {
	TRLWEFFT []TRLWELv1FFT
}
