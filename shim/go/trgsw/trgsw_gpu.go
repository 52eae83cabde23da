// Package trgsw is the bootstrap surface of go-tfhe's trgsw package (trgsw/trgsw.go:108-312, trgsw/keyswitch.go:10-37) on the
// MI355X engine: the functions gates.Batch* and the evaluator are themselves built on -- BatchBlindRotate, BlindRotate, CMUX,
// ExternalProductWithFFT, IdentityKeySwitching[Assign] -- with the reference's names and parameter lists, so a caller of
// trgsw.BatchBlindRotate switches by changing the import path.  The types are the reference's own (aliases): a
// *trgsw.TRGSWLv1FFT made by the reference's trgsw.NewTRGSWLv1FFT is what these functions take.
//
// Keys arrive with every call, as in the reference, and are uploaded to the GPUs once, on first use (gpu.Attached /
// gpu.AttachedKSK find them again by identity).  A free-standing TRGSW operand (ExternalProductWithFFT, CMUX) travels with the call.
// The polyEval arguments are kept for signature compatibility and ignored: the engine owns its transforms.
package trgsw

import (
	"github.com/thedonutfactory/go-tfhe-gpu/gpu"
	"github.com/thedonutfactory/go-tfhe/params"
	"github.com/thedonutfactory/go-tfhe/poly"
	"github.com/thedonutfactory/go-tfhe/tlwe"
	reftrgsw "github.com/thedonutfactory/go-tfhe/trgsw"
	"github.com/thedonutfactory/go-tfhe/trlwe"
)

// The reference's types (trgsw/trgsw.go:60-68), under the names its callers use.
type TRGSWLv1FFT = reftrgsw.TRGSWLv1FFT
type TRLWELv1FFT = reftrgsw.TRLWELv1FFT

// ExternalProductWithFFT: trgswFFT (x) trlweIn.  Reference: trgsw/trgsw.go:108 (which returns pooled buffers; this result
// owns its storage).  decompositionOffset is a kernel operand here: any value the caller passes is used.
func ExternalProductWithFFT(trgswFFT *TRGSWLv1FFT, trlweIn *trlwe.TRLWELv1, decompositionOffset params.Torus, polyEval *poly.Evaluator) *trlwe.TRLWELv1 {
	return gpu.Scratch().ExternalProductWith(trgswFFT, []*trlwe.TRLWELv1{trlweIn}, decompositionOffset)[0]
}

// CMUX: in1 where cond encrypts 0, in2 where it encrypts 1 -- in1 + cond (x) (in2 - in1).  Reference: trgsw/trgsw.go:173.
func CMUX(in1, in2 *trlwe.TRLWELv1, cond *TRGSWLv1FFT, decompositionOffset params.Torus, polyEval *poly.Evaluator) *trlwe.TRLWELv1 {
	return gpu.Scratch().CMuxWith(cond, []*trlwe.TRLWELv1{in1}, []*trlwe.TRLWELv1{in2}, decompositionOffset)[0]
}

func checkOffset(k *gpu.CloudKey, decompositionOffset params.Torus) {
	if decompositionOffset != k.DecompositionOffset() {
		panic("tfhe_hip: decompositionOffset is not the offset of the current parameters (cloudkey/cloudkey.go:60-71)")
	}
}

// BlindRotate: all n CMUX steps of one sample.  Reference: trgsw/trgsw.go:197.
func BlindRotate(src *tlwe.TLWELv0, blindRotateTestvec *trlwe.TRLWELv1, bootstrappingKey []*TRGSWLv1FFT, decompositionOffset params.Torus, polyEval *poly.Evaluator) *trlwe.TRLWELv1 {
	k := gpu.Attached(bootstrappingKey, nil).Pick()
	checkOffset(k, decompositionOffset)
	return k.BlindRotateBatch([]*tlwe.TLWELv0{src}, blindRotateTestvec)[0]
}

// BatchBlindRotate: the batch fan-out every gates.Batch* of the reference is built on -- one goroutine per input there
// (trgsw/trgsw.go:234-252), contiguous shards over all GPUs here, results in input order.
func BatchBlindRotate(srcs []*tlwe.TLWELv0, blindRotateTestvec *trlwe.TRLWELv1, bootstrappingKey []*TRGSWLv1FFT, decompositionOffset params.Torus) []*trlwe.TRLWELv1 {
	set := gpu.Attached(bootstrappingKey, nil)
	checkOffset(set.Replica(0), decompositionOffset)
	return set.BlindRotateBatch(srcs, blindRotateTestvec)
}

// IdentityKeySwitching: a level-1 sample under the level-0 key.  Reference: trgsw/trgsw.go:285.
func IdentityKeySwitching(src *tlwe.TLWELv1, keySwitchingKey []*tlwe.TLWELv0) *tlwe.TLWELv0 {
	return gpu.AttachedKSK(keySwitchingKey).Pick().KeySwitch([]*tlwe.TLWELv1{src})[0]
}

// IdentityKeySwitchingAssign writes into output.  Reference: trgsw/keyswitch.go:10.
func IdentityKeySwitchingAssign(src *tlwe.TLWELv1, keySwitchingKey []*tlwe.TLWELv0, output *tlwe.TLWELv0) {
	copy(output.P, IdentityKeySwitching(src, keySwitchingKey).P)
}

// BatchIdentityKeySwitching is IdentityKeySwitching over a batch, sharded over all GPUs (no counterpart in the reference).
func BatchIdentityKeySwitching(srcs []*tlwe.TLWELv1, keySwitchingKey []*tlwe.TLWELv0) []*tlwe.TLWELv0 {
	return gpu.AttachedKSK(keySwitchingKey).KeySwitch(srcs)
}
