// Package evaluator is the bootstrap surface of go-tfhe's evaluator package (evaluator/evaluator.go:50-157,
// evaluator/programmable_bootstrap.go:16-115, evaluator/gates_helper.go:10-63) on the MI355X engine, with the reference's
// method names and parameter lists: the keys arrive with every call, as in the reference, and are uploaded and replicated
// to the GPUs once, on first use (gpu.Attached finds them again by identity).
//
// The decompositionOffset argument is kept for signature compatibility and checked against the value the engine derives
// from the parameters (cloudkey/cloudkey.go:60-71) -- the kernels have it baked in.
package evaluator

import (
	"github.com/thedonutfactory/go-tfhe-gpu/gpu"
	"github.com/thedonutfactory/go-tfhe/lut"
	"github.com/thedonutfactory/go-tfhe/params"
	"github.com/thedonutfactory/go-tfhe/tlwe"
	"github.com/thedonutfactory/go-tfhe/trgsw"
	"github.com/thedonutfactory/go-tfhe/trlwe"
	"github.com/thedonutfactory/go-tfhe/utils"
)

// Evaluator mirrors evaluator.Evaluator (evaluator/evaluator.go:15-24).  It holds no buffers: the GPU context does, and
// unlike the reference's it may be shared by any number of goroutines.
type Evaluator struct {
	n int
}

// NewEvaluator mirrors evaluator.NewEvaluator(n) (evaluator/evaluator.go:27).
func NewEvaluator(n int) *Evaluator {
	return &Evaluator{n: n}
}

// ShallowCopy mirrors evaluator/evaluator.go:40 (the reference needs one evaluator per goroutine; here it is free).
func (e *Evaluator) ShallowCopy() *Evaluator {
	return &Evaluator{n: e.n}
}

func expectedOffset() params.Torus {
	g := params.GetTRGSWLv1()
	var offset params.Torus
	for i := 0; i < g.L; i++ {
		offset += params.Torus(g.BG/2) * params.Torus(uint32(1)<<(32-uint32(i+1)*g.BGBIT))
	}
	return offset
}

func checkOffset(decompositionOffset params.Torus) {
	if decompositionOffset != expectedOffset() {
		panic("tfhe_hip: decompositionOffset is not the offset of the current parameters (cloudkey/cloudkey.go:60-71)")
	}
}

// ExternalProductAssign: ctFourierGGSW (x) ctIn, into ctOut -- for ANY TRGSW operand, which travels with the call; the
// decomposition offset is a kernel operand (any value).  Reference: evaluator/evaluator.go:50.
func (e *Evaluator) ExternalProductAssign(ctFourierGGSW *trgsw.TRGSWLv1FFT, ctIn *trlwe.TRLWELv1, decompositionOffset params.Torus, ctOut *trlwe.TRLWELv1) {
	res := gpu.Scratch().ExternalProductWith(ctFourierGGSW, []*trlwe.TRLWELv1{ctIn}, decompositionOffset)
	copy(ctOut.A, res[0].A)
	copy(ctOut.B, res[0].B)
}

// CMuxAssign: ctOut = ct0 + ctCond (x) (ct1 - ct0); ctOut may be ct0 (the blind rotation's own use).
// Reference: evaluator/evaluator.go:85.
func (e *Evaluator) CMuxAssign(ctCond *trgsw.TRGSWLv1FFT, ct0, ct1 *trlwe.TRLWELv1, decompositionOffset params.Torus, ctOut *trlwe.TRLWELv1) {
	res := gpu.Scratch().CMuxWith(ctCond, []*trlwe.TRLWELv1{ct0}, []*trlwe.TRLWELv1{ct1}, decompositionOffset)
	copy(ctOut.A, res[0].A)
	copy(ctOut.B, res[0].B)
}

// BlindRotateAssign: the accumulator after all n CMUX steps, into ctOut.  Reference: evaluator/evaluator.go:110.
func (e *Evaluator) BlindRotateAssign(ctIn *tlwe.TLWELv0, testvec *trlwe.TRLWELv1, bsk []*trgsw.TRGSWLv1FFT, decompositionOffset params.Torus, ctOut *trlwe.TRLWELv1) {
	checkOffset(decompositionOffset)
	res := gpu.Attached(bsk, nil).Pick().BlindRotateBatch([]*tlwe.TLWELv0{ctIn}, testvec)
	copy(ctOut.A, res[0].A)
	copy(ctOut.B, res[0].B)
}

// BootstrapAssign: blind rotation, sample extraction and key switch in one call of the engine, into ctOut.
// Reference: evaluator/evaluator.go:139.
func (e *Evaluator) BootstrapAssign(ctIn *tlwe.TLWELv0, testvec *trlwe.TRLWELv1, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus, ctOut *tlwe.TLWELv0) {
	checkOffset(decompositionOffset)
	res := gpu.Attached(bsk, ksk).Pick().BootstrapBatch([]*tlwe.TLWELv0{ctIn}, testvec)
	copy(ctOut.P, res[0].P)
}

// Bootstrap returns a fresh sample.  Reference: evaluator/evaluator.go:152, which returns a
// pointer into a four-slot ring that is valid "until 4 more bootstrap calls"; this result owns its storage.
func (e *Evaluator) Bootstrap(ctIn *tlwe.TLWELv0, testvec *trlwe.TRLWELv1, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus) *tlwe.TLWELv0 {
	result := tlwe.NewTLWELv0()
	e.BootstrapAssign(ctIn, testvec, bsk, ksk, decompositionOffset, result)
	return result
}

// BootstrapLUTAssign: the table's polynomial is the test vector of an ordinary bootstrap.
// Reference: evaluator/programmable_bootstrap.go:93.
func (e *Evaluator) BootstrapLUTAssign(ctIn *tlwe.TLWELv0, lut *lut.LookUpTable, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus, ctOut *tlwe.TLWELv0) {
	e.BootstrapAssign(ctIn, lut.Poly, bsk, ksk, decompositionOffset, ctOut)
}

// BootstrapLUT.  Reference: evaluator/programmable_bootstrap.go:54.
func (e *Evaluator) BootstrapLUT(ctIn *tlwe.TLWELv0, lut *lut.LookUpTable, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus) *tlwe.TLWELv0 {
	result := tlwe.NewTLWELv0()
	e.BootstrapLUTAssign(ctIn, lut, bsk, ksk, decompositionOffset, result)
	return result
}

// BootstrapLUTTemp mirrors evaluator/programmable_bootstrap.go:71 (a pooled result there; an owned one here).
func (e *Evaluator) BootstrapLUTTemp(ctIn *tlwe.TLWELv0, lut *lut.LookUpTable, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus) *tlwe.TLWELv0 {
	return e.BootstrapLUT(ctIn, lut, bsk, ksk, decompositionOffset)
}

// BootstrapFunc builds the table of f over [0, messageModulus) on the host and bootstraps through it.
// Reference: evaluator/programmable_bootstrap.go:16.
func (e *Evaluator) BootstrapFunc(ctIn *tlwe.TLWELv0, f func(int) int, messageModulus int, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus) *tlwe.TLWELv0 {
	generator := lut.NewGenerator(messageModulus)
	lookupTable := generator.GenLookUpTable(f)
	return e.BootstrapLUT(ctIn, lookupTable, bsk, ksk, decompositionOffset)
}

// BootstrapFuncAssign mirrors evaluator/programmable_bootstrap.go:33.
func (e *Evaluator) BootstrapFuncAssign(ctIn *tlwe.TLWELv0, f func(int) int, messageModulus int, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus, ctOut *tlwe.TLWELv0) {
	generator := lut.NewGenerator(messageModulus)
	lookupTable := generator.GenLookUpTable(f)
	e.BootstrapLUTAssign(ctIn, lookupTable, bsk, ksk, decompositionOffset, ctOut)
}

// BatchBootstrapLUT is BootstrapLUT over a batch, sharded over all GPUs (the reference has no batch form of it; its batch
// path is trgsw.BatchBlindRotate for gates, trgsw/trgsw.go:234-252).
func (e *Evaluator) BatchBootstrapLUT(ctsIn []*tlwe.TLWELv0, lut *lut.LookUpTable, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus) []*tlwe.TLWELv0 {
	checkOffset(decompositionOffset)
	return gpu.Attached(bsk, ksk).BootstrapBatch(ctsIn, lut.Poly)
}

// combine is the linear step in front of a gate bootstrap: ca*a + cb*b on every word, with mu added to the body
// (evaluator/gates_helper.go:10-63 writes it out per gate; -x is (2^32 - 1)*x on the torus).
func combine(a, b *tlwe.TLWELv0, ca, cb params.Torus, mu float64) *tlwe.TLWELv0 {
	out := tlwe.NewTLWELv0()
	for i := range out.P {
		out.P[i] = ca*a.P[i] + cb*b.P[i]
	}
	out.SetB(out.B() + utils.F64ToTorus(mu))
	return out
}

// PrepareNAND: -(a + b) + 1/8.  Reference: evaluator/gates_helper.go:10.
func (e *Evaluator) PrepareNAND(a, b *tlwe.TLWELv0) *tlwe.TLWELv0 {
	minusOne := ^params.Torus(0)
	return combine(a, b, minusOne, minusOne, 0.125)
}

// PrepareAND: (a + b) - 1/8.  Reference: evaluator/gates_helper.go:24.
func (e *Evaluator) PrepareAND(a, b *tlwe.TLWELv0) *tlwe.TLWELv0 {
	return combine(a, b, 1, 1, -0.125)
}

// PrepareOR: (a + b) + 1/8.  Reference: evaluator/gates_helper.go:38.
func (e *Evaluator) PrepareOR(a, b *tlwe.TLWELv0) *tlwe.TLWELv0 {
	return combine(a, b, 1, 1, 0.125)
}

// PrepareXOR: (a + 2b) + 1/4.  Reference: evaluator/gates_helper.go:52.
func (e *Evaluator) PrepareXOR(a, b *tlwe.TLWELv0) *tlwe.TLWELv0 {
	return combine(a, b, 1, 2, 0.25)
}
