// Package evaluator is the bootstrap surface of go-tfhe's evaluator package (evaluator/evaluator.go:110-157,
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

// BlindRotateAssign performs blind rotation and writes to ctOut (evaluator/evaluator.go:110).
func (e *Evaluator) BlindRotateAssign(ctIn *tlwe.TLWELv0, testvec *trlwe.TRLWELv1, bsk []*trgsw.TRGSWLv1FFT, decompositionOffset params.Torus, ctOut *trlwe.TRLWELv1) {
	checkOffset(decompositionOffset)
	res := gpu.Attached(bsk, nil).Pick().BlindRotateBatch([]*tlwe.TLWELv0{ctIn}, testvec)
	copy(ctOut.A, res[0].A)
	copy(ctOut.B, res[0].B)
}

// BootstrapAssign performs full bootstrapping, blind rotate + sample extract + key switch, into ctOut
// (evaluator/evaluator.go:139).
func (e *Evaluator) BootstrapAssign(ctIn *tlwe.TLWELv0, testvec *trlwe.TRLWELv1, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus, ctOut *tlwe.TLWELv0) {
	checkOffset(decompositionOffset)
	res := gpu.Attached(bsk, ksk).Pick().BootstrapBatch([]*tlwe.TLWELv0{ctIn}, testvec)
	copy(ctOut.P, res[0].P)
}

// Bootstrap performs full bootstrapping and returns the result (evaluator/evaluator.go:152).  The reference returns a
// pointer into a four-slot ring that is valid "until 4 more bootstrap calls"; this result owns its storage.
func (e *Evaluator) Bootstrap(ctIn *tlwe.TLWELv0, testvec *trlwe.TRLWELv1, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus) *tlwe.TLWELv0 {
	result := tlwe.NewTLWELv0()
	e.BootstrapAssign(ctIn, testvec, bsk, ksk, decompositionOffset, result)
	return result
}

// BootstrapLUTAssign performs programmable bootstrapping with a lookup table (evaluator/programmable_bootstrap.go:93):
// the table's polynomial is the test vector.
func (e *Evaluator) BootstrapLUTAssign(ctIn *tlwe.TLWELv0, lut *lut.LookUpTable, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus, ctOut *tlwe.TLWELv0) {
	e.BootstrapAssign(ctIn, lut.Poly, bsk, ksk, decompositionOffset, ctOut)
}

// BootstrapLUT performs programmable bootstrapping with a pre-computed lookup table (evaluator/programmable_bootstrap.go:54).
func (e *Evaluator) BootstrapLUT(ctIn *tlwe.TLWELv0, lut *lut.LookUpTable, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus) *tlwe.TLWELv0 {
	result := tlwe.NewTLWELv0()
	e.BootstrapLUTAssign(ctIn, lut, bsk, ksk, decompositionOffset, result)
	return result
}

// BootstrapLUTTemp mirrors evaluator/programmable_bootstrap.go:71 (a pooled result there; an owned one here).
func (e *Evaluator) BootstrapLUTTemp(ctIn *tlwe.TLWELv0, lut *lut.LookUpTable, bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, decompositionOffset params.Torus) *tlwe.TLWELv0 {
	return e.BootstrapLUT(ctIn, lut, bsk, ksk, decompositionOffset)
}

// BootstrapFunc performs programmable bootstrapping with a function on [0, messageModulus)
// (evaluator/programmable_bootstrap.go:16).
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

// PrepareNAND prepares a NAND input for bootstrapping: -(a + b) + 1/8 (evaluator/gates_helper.go:10).
func (e *Evaluator) PrepareNAND(a, b *tlwe.TLWELv0) *tlwe.TLWELv0 {
	n := params.GetTLWELv0().N
	result := tlwe.NewTLWELv0()
	for i := 0; i < n; i++ {
		result.P[i] = -(a.P[i] + b.P[i])
	}
	result.P[n] = -(a.P[n] + b.P[n]) + utils.F64ToTorus(0.125)
	return result
}

// PrepareAND prepares an AND input for bootstrapping: (a + b) - 1/8 (evaluator/gates_helper.go:24).
func (e *Evaluator) PrepareAND(a, b *tlwe.TLWELv0) *tlwe.TLWELv0 {
	n := params.GetTLWELv0().N
	result := tlwe.NewTLWELv0()
	for i := 0; i < n; i++ {
		result.P[i] = a.P[i] + b.P[i]
	}
	result.P[n] = a.P[n] + b.P[n] + utils.F64ToTorus(-0.125)
	return result
}

// PrepareOR prepares an OR input for bootstrapping: (a + b) + 1/8 (evaluator/gates_helper.go:38).
func (e *Evaluator) PrepareOR(a, b *tlwe.TLWELv0) *tlwe.TLWELv0 {
	n := params.GetTLWELv0().N
	result := tlwe.NewTLWELv0()
	for i := 0; i < n; i++ {
		result.P[i] = a.P[i] + b.P[i]
	}
	result.P[n] = a.P[n] + b.P[n] + utils.F64ToTorus(0.125)
	return result
}

// PrepareXOR prepares an XOR input for bootstrapping: (a + 2b) + 1/4 (evaluator/gates_helper.go:52).
func (e *Evaluator) PrepareXOR(a, b *tlwe.TLWELv0) *tlwe.TLWELv0 {
	n := params.GetTLWELv0().N
	result := tlwe.NewTLWELv0()
	for i := 0; i < n; i++ {
		result.P[i] = a.P[i] + 2*b.P[i]
	}
	result.P[n] = a.P[n] + 2*b.P[n] + utils.F64ToTorus(0.25)
	return result
}
