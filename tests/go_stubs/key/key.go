// Stand-in for go-tfhe's key package on the GPU box (tests/go_stubs/README.md).
package key

import "github.com/thedonutfactory/go-tfhe/params"

type SecretKey struct {
	KeyLv0 []params.Torus
	KeyLv1 []params.Torus
}
