// The cgo side of the drop-in boundary (include/tfhe_hip.h) as a Go module of its own: three packages that keep the
// reference's package names and signatures, so a caller switches by changing import paths only:
//
//	github.com/thedonutfactory/go-tfhe/gates      ->  github.com/thedonutfactory/go-tfhe-gpu/gates
//	github.com/thedonutfactory/go-tfhe/evaluator  ->  github.com/thedonutfactory/go-tfhe-gpu/evaluator
//
// Build (needs Go >= 1.21, a go-tfhe checkout and libtfhe_hip.so; none of which this repository's image has -- the files are
// checked statically against /root/reference by tests/test_go_shim_static.py and have never met a Go compiler):
//
//	cd shim/go && go mod edit -replace github.com/thedonutfactory/go-tfhe=<your checkout> && go vet ./... && go test ./gates
module github.com/thedonutfactory/go-tfhe-gpu

go 1.21

require github.com/thedonutfactory/go-tfhe v0.0.0

replace github.com/thedonutfactory/go-tfhe => ../../../go-tfhe
