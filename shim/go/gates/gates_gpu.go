// Package gates is go-tfhe's gates package (gates/gates.go) on the MI355X engine: the same names, the same signatures, the
// same panics -- switch by changing the import path.  The cloud key a caller passes (*cloudkey.CloudKey, Go pointer graphs)
// is uploaded and replicated to the GPUs once, on first use (gpu.Attached); scalar gates may be issued from any number of
// goroutines (the reference's share one evaluator that is not goroutine-safe, gates/gates.go:18-23): they are spread over
// the GPUs in turn and the concurrent callers of one GPU travel in one launch.  Batch gates shard contiguously over all
// GPUs, one goroutine per GPU, results in input order (the reference: one goroutine per input, trgsw/trgsw.go:234-252).
//
// Differences a maintainer should know (INTEGRATION.md section 4): BatchXNOR follows the tested scalar XNOR (+1/4), not
// the reference's BatchXNOR (-1/4, gates/gates.go:293), which computes XOR; every result owns its storage.
package gates

import (
	"github.com/thedonutfactory/go-tfhe-gpu/gpu"
	"github.com/thedonutfactory/go-tfhe/cloudkey"
	"github.com/thedonutfactory/go-tfhe/params"
	"github.com/thedonutfactory/go-tfhe/tlwe"
	"github.com/thedonutfactory/go-tfhe/utils"
)

// Ciphertext: the same alias the reference declares (gates/gates.go:16).
type Ciphertext = tlwe.TLWELv0

func keys(ck *cloudkey.CloudKey) *gpu.CloudKeySet {
	return gpu.Attached(ck.BootstrappingKey, ck.KeySwitchingKey)
}

// Release frees the GPU replicas of ck (the reference has nothing to release; a long-running service switching keys does).
func Release(ck *cloudkey.CloudKey) {
	gpu.Detach(ck.BootstrappingKey)
}

func gate(op int, tlweA, tlweB *Ciphertext, ck *cloudkey.CloudKey) *Ciphertext {
	out := keys(ck).Pick().GateBatch(op, []*tlwe.TLWELv0{tlweA}, []*tlwe.TLWELv0{tlweB}, nil)
	return out[0]
}

func batch(op int, inputs [][2]*Ciphertext, ck *cloudkey.CloudKey) []*Ciphertext {
	a := make([]*tlwe.TLWELv0, len(inputs))
	b := make([]*tlwe.TLWELv0, len(inputs))
	for i, pair := range inputs {
		a[i] = pair[0]
		b[i] = pair[1]
	}
	return keys(ck).GateBatch(op, a, b, nil)
}

// NAND: NOT(a AND b).  Reference: gates/gates.go:26.
func NAND(tlweA, tlweB *Ciphertext, ck *cloudkey.CloudKey) *Ciphertext {
	return gate(gpu.OpNAND, tlweA, tlweB, ck)
}

// OR.  Reference: gates/gates.go:34.
func OR(tlweA, tlweB *Ciphertext, ck *cloudkey.CloudKey) *Ciphertext {
	return gate(gpu.OpOR, tlweA, tlweB, ck)
}

// AND.  Reference: gates/gates.go:40.
func AND(tlweA, tlweB *Ciphertext, ck *cloudkey.CloudKey) *Ciphertext {
	return gate(gpu.OpAND, tlweA, tlweB, ck)
}

// XOR.  Reference: gates/gates.go:46.
func XOR(tlweA, tlweB *Ciphertext, ck *cloudkey.CloudKey) *Ciphertext {
	return gate(gpu.OpXOR, tlweA, tlweB, ck)
}

// XNOR, with the +1/4 of the reference's scalar gate.  Reference: gates/gates.go:52.
func XNOR(tlweA, tlweB *Ciphertext, ck *cloudkey.CloudKey) *Ciphertext {
	return gate(gpu.OpXNOR, tlweA, tlweB, ck)
}

// Constant is a trivial (noiseless, keyless) encryption of `value`: host code only.  Reference: gates/gates.go:61.
func Constant(value bool) *Ciphertext {
	out := tlwe.NewTLWELv0()
	eighth := utils.F64ToTorus(0.125)
	if value {
		out.SetB(eighth)
	} else {
		out.SetB(1 - eighth) // the reference's encoding of false: 1 - mu, not -mu (gates/gates.go:63-65)
	}
	return out
}

// NOR.  Reference: gates/gates.go:72.
func NOR(tlweA, tlweB *Ciphertext, ck *cloudkey.CloudKey) *Ciphertext {
	return gate(gpu.OpNOR, tlweA, tlweB, ck)
}

// ANDNY: (NOT a) AND b.  Reference: gates/gates.go:79.
func ANDNY(tlweA, tlweB *Ciphertext, ck *cloudkey.CloudKey) *Ciphertext {
	return gate(gpu.OpANDNY, tlweA, tlweB, ck)
}

// ANDYN: a AND (NOT b).  Reference: gates/gates.go:86.
func ANDYN(tlweA, tlweB *Ciphertext, ck *cloudkey.CloudKey) *Ciphertext {
	return gate(gpu.OpANDYN, tlweA, tlweB, ck)
}

// ORNY: (NOT a) OR b.  Reference: gates/gates.go:93.
func ORNY(tlweA, tlweB *Ciphertext, ck *cloudkey.CloudKey) *Ciphertext {
	return gate(gpu.OpORNY, tlweA, tlweB, ck)
}

// ORYN: a OR (NOT b).  Reference: gates/gates.go:100.
func ORYN(tlweA, tlweB *Ciphertext, ck *cloudkey.CloudKey) *Ciphertext {
	return gate(gpu.OpORYN, tlweA, tlweB, ck)
}

// MUX performs homomorphic multiplexer a ? b : c = OR(AND(a, b), AND(NOT(a), c)) (gates/gates.go:107): three bootstraps,
// issued as ONE call (the engine runs AND(a, b) and ANDNY(a, c) in one launch and the OR in a second).
func MUX(tlweA, tlweB, tlweC *Ciphertext, ck *cloudkey.CloudKey) *Ciphertext {
	out := keys(ck).Pick().GateBatch(gpu.OpMUX, []*tlwe.TLWELv0{tlweA}, []*tlwe.TLWELv0{tlweB}, []*tlwe.TLWELv0{tlweC})
	return out[0]
}

// NOT negates the sample; no bootstrap, no key.  Reference: gates/gates.go:117.
func NOT(tlweA *Ciphertext) *Ciphertext {
	return tlweA.Neg()
}

// Copy returns a sample that owns a copy of the words.  Reference: gates/gates.go:122.
func Copy(tlweA *Ciphertext) *Ciphertext {
	return &tlwe.TLWELv0{P: append([]params.Torus(nil), tlweA.P...)}
}

// BatchNAND: NAND over [][2]*Ciphertext, sharded over every GPU.  Reference: gates/gates.go:156.
func BatchNAND(inputs [][2]*Ciphertext, ck *cloudkey.CloudKey) []*Ciphertext {
	return batch(gpu.OpNAND, inputs, ck)
}

// BatchAND.  Reference: gates/gates.go:185.
func BatchAND(inputs [][2]*Ciphertext, ck *cloudkey.CloudKey) []*Ciphertext {
	return batch(gpu.OpAND, inputs, ck)
}

// BatchOR.  Reference: gates/gates.go:211.
func BatchOR(inputs [][2]*Ciphertext, ck *cloudkey.CloudKey) []*Ciphertext {
	return batch(gpu.OpOR, inputs, ck)
}

// BatchXOR.  Reference: gates/gates.go:237.
func BatchXOR(inputs [][2]*Ciphertext, ck *cloudkey.CloudKey) []*Ciphertext {
	return batch(gpu.OpXOR, inputs, ck)
}

// BatchNOR.  Reference: gates/gates.go:263.
func BatchNOR(inputs [][2]*Ciphertext, ck *cloudkey.CloudKey) []*Ciphertext {
	return batch(gpu.OpNOR, inputs, ck)
}

// BatchXNOR, with the scalar XNOR's +1/4 (the reference's batch form adds -1/4 and computes XOR).  Reference: gates/gates.go:289.
func BatchXNOR(inputs [][2]*Ciphertext, ck *cloudkey.CloudKey) []*Ciphertext {
	return batch(gpu.OpXNOR, inputs, ck)
}

// BatchMUX is MUX over a batch of (a, b, c) triples; the reference has no batch form of it.
func BatchMUX(inputs [][3]*Ciphertext, ck *cloudkey.CloudKey) []*Ciphertext {
	a := make([]*tlwe.TLWELv0, len(inputs))
	b := make([]*tlwe.TLWELv0, len(inputs))
	c := make([]*tlwe.TLWELv0, len(inputs))
	for i, t := range inputs {
		a[i] = t[0]
		b[i] = t[1]
		c[i] = t[2]
	}
	return keys(ck).GateBatch(gpu.OpMUX, a, b, c)
}
