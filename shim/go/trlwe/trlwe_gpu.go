// Package trlwe is the one function of go-tfhe's trlwe package that lies on the bootstrap path -- SampleExtractIndex[Assign]
// (trlwe/trlwe.go:114-128, trlwe/trlwe_ops.go:10-21) -- on the MI355X engine, for any index k, with the reference's
// signatures; the types are the reference's own (aliases).  Inside a bootstrap the engine fuses the extraction at index 0
// into the key switch; this is the seam on its own, for callers that extract other coefficients.
package trlwe

import (
	"github.com/thedonutfactory/go-tfhe-gpu/gpu"
	"github.com/thedonutfactory/go-tfhe/tlwe"
	reftrlwe "github.com/thedonutfactory/go-tfhe/trlwe"
)

// The reference's type (trlwe/trlwe.go:13-16).
type TRLWELv1 = reftrlwe.TRLWELv1

// SampleExtractIndex: the level-1 LWE sample of coefficient k.  Reference: trlwe/trlwe.go:114.
func SampleExtractIndex(trlwe *TRLWELv1, k int) *tlwe.TLWELv1 {
	return gpu.Scratch().SampleExtract([]*reftrlwe.TRLWELv1{trlwe}, k)[0]
}

// SampleExtractIndexAssign writes into output.  Reference: trlwe/trlwe_ops.go:10.
func SampleExtractIndexAssign(trlwe *TRLWELv1, k int, output *tlwe.TLWELv1) {
	copy(output.P, SampleExtractIndex(trlwe, k).P)
}
