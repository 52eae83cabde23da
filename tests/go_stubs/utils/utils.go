// Stand-in for go-tfhe's utils package on the GPU box (tests/go_stubs/README.md).
package utils

import (
	"math"

	"github.com/thedonutfactory/go-tfhe/params"
)

// F64ToTorus: the fractional part of d on the 32-bit torus.
func F64ToTorus(d float64) params.Torus {
	return params.Torus(int64(math.Mod(d, 1.0) * 4294967296.0))
}
