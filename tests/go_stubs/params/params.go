// Stand-in for go-tfhe's params package on the GPU box (tests/go_stubs/README.md): the types the shim reads, and the current
// parameter set as two variables the test sets.
package params

type Torus uint32

type TLWELv0Params struct {
	N     int
	ALPHA float64
}

type TRGSWLv1Params struct {
	N         int
	NBIT      int
	BGBIT     uint32
	BG        uint32
	L         int
	BASEBIT   int
	IKS_T     int
	ALPHA     float64
	BlockSize int
}

var Lv0 = TLWELv0Params{N: 700, ALPHA: 2.0e-5}

var Lv1 = TRGSWLv1Params{N: 1024, NBIT: 10, BGBIT: 6, BG: 64, L: 3, BASEBIT: 2, IKS_T: 9, ALPHA: 2.0e-8, BlockSize: 1}

func GetTLWELv0() TLWELv0Params { return Lv0 }

func GetTRGSWLv1() TRGSWLv1Params { return Lv1 }

func KSKAlpha() float64 { return Lv0.ALPHA }

func BSKAlpha() float64 { return Lv1.ALPHA }
