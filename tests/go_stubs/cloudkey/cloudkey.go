// Stand-in for go-tfhe's cloudkey package on the GPU box (tests/go_stubs/README.md).
package cloudkey

import (
	"github.com/thedonutfactory/go-tfhe/params"
	"github.com/thedonutfactory/go-tfhe/tlwe"
	"github.com/thedonutfactory/go-tfhe/trgsw"
	"github.com/thedonutfactory/go-tfhe/trlwe"
)

type CloudKey struct {
	DecompositionOffset params.Torus
	BlindRotateTestvec  *trlwe.TRLWELv1
	KeySwitchingKey     []*tlwe.TLWELv0
	BootstrappingKey    []*trgsw.TRGSWLv1FFT
}
