// Stand-in for go-tfhe's poly package on the GPU box (tests/go_stubs/README.md).
package poly

type FourierPoly struct {
	Coeffs []float64
}

// Evaluator: opaque here (the shim only passes *poly.Evaluator through, for signature compatibility with trgsw.*).
type Evaluator struct{}
