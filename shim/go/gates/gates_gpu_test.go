// The reference's gate tests (gates/gates_test.go:23-480) run against this package -- and, because both implementations are
// in one process here, every result is also compared WORD FOR WORD with the reference's own gate on the same inputs and the
// same cloud key (at the N = 1024, L = 3, Bgbit = 6 sets the transforms are exact, so bit equality is the bar: DESIGN.md
// section 4).  Needs Go, a go-tfhe checkout and an MI355X; never run in this repository's image.
package gates_test

import (
	"testing"

	"github.com/thedonutfactory/go-tfhe-gpu/gates"
	"github.com/thedonutfactory/go-tfhe/cloudkey"
	refgates "github.com/thedonutfactory/go-tfhe/gates"
	"github.com/thedonutfactory/go-tfhe/key"
	"github.com/thedonutfactory/go-tfhe/params"
	"github.com/thedonutfactory/go-tfhe/tlwe"
)

func encrypt(val bool, sk *key.SecretKey) *gates.Ciphertext {
	return tlwe.NewTLWELv0().EncryptBool(val, params.GetTLWELv0().ALPHA, sk.KeyLv0)
}

func sameWords(a, b *gates.Ciphertext) bool {
	if len(a.P) != len(b.P) {
		return false
	}
	for i := range a.P {
		if a.P[i] != b.P[i] {
			return false
		}
	}
	return true
}

type gate2 func(a, b *gates.Ciphertext, ck *cloudkey.CloudKey) *gates.Ciphertext

func TestScalarGatesTruthTablesAndWordParity(t *testing.T) {
	sk := key.NewSecretKey()
	ck := cloudkey.NewCloudKey(sk)
	defer gates.Release(ck)
	cases := []struct {
		name string
		gpu  gate2
		ref  gate2
		want func(a, b bool) bool
	}{
		{"NAND", gates.NAND, refgates.NAND, func(a, b bool) bool { return !(a && b) }},
		{"AND", gates.AND, refgates.AND, func(a, b bool) bool { return a && b }},
		{"OR", gates.OR, refgates.OR, func(a, b bool) bool { return a || b }},
		{"XOR", gates.XOR, refgates.XOR, func(a, b bool) bool { return a != b }},
		{"XNOR", gates.XNOR, refgates.XNOR, func(a, b bool) bool { return a == b }},
		{"NOR", gates.NOR, refgates.NOR, func(a, b bool) bool { return !(a || b) }},
		{"ANDNY", gates.ANDNY, refgates.ANDNY, func(a, b bool) bool { return !a && b }},
		{"ANDYN", gates.ANDYN, refgates.ANDYN, func(a, b bool) bool { return a && !b }},
		{"ORNY", gates.ORNY, refgates.ORNY, func(a, b bool) bool { return !a || b }},
		{"ORYN", gates.ORYN, refgates.ORYN, func(a, b bool) bool { return a || !b }},
	}
	for _, c := range cases {
		for _, a := range []bool{false, true} {
			for _, b := range []bool{false, true} {
				ctA := encrypt(a, sk)
				ctB := encrypt(b, sk)
				got := c.gpu(ctA, ctB, ck)
				if dec := got.DecryptBool(sk.KeyLv0); dec != c.want(a, b) {
					t.Errorf("%s(%v, %v) = %v, expected %v", c.name, a, b, dec, c.want(a, b))
				}
				if ref := c.ref(ctA, ctB, ck); !sameWords(got, ref) {
					t.Errorf("%s(%v, %v): GPU ciphertext differs from the reference's", c.name, a, b)
				}
			}
		}
	}
}

func TestMUXNotCopyConstant(t *testing.T) {
	sk := key.NewSecretKey()
	ck := cloudkey.NewCloudKey(sk)
	defer gates.Release(ck)
	for _, a := range []bool{false, true} {
		for _, b := range []bool{false, true} {
			for _, c := range []bool{false, true} {
				ctA := encrypt(a, sk)
				ctB := encrypt(b, sk)
				ctC := encrypt(c, sk)
				got := gates.MUX(ctA, ctB, ctC, ck)
				want := c
				if a {
					want = b
				}
				if dec := got.DecryptBool(sk.KeyLv0); dec != want {
					t.Errorf("MUX(%v, %v, %v) = %v, expected %v", a, b, c, dec, want)
				}
				if ref := refgates.MUX(ctA, ctB, ctC, ck); !sameWords(got, ref) {
					t.Errorf("MUX(%v, %v, %v): GPU ciphertext differs from the reference's", a, b, c)
				}
			}
		}
	}
	ct := encrypt(true, sk)
	if gates.NOT(ct).DecryptBool(sk.KeyLv0) {
		t.Errorf("NOT(true) decrypts to true")
	}
	if !sameWords(gates.Copy(ct), ct) {
		t.Errorf("Copy changed the ciphertext")
	}
	if !gates.Constant(true).DecryptBool(sk.KeyLv0) || gates.Constant(false).DecryptBool(sk.KeyLv0) {
		t.Errorf("Constant decrypts wrongly")
	}
}

type batch2 func(inputs [][2]*gates.Ciphertext, ck *cloudkey.CloudKey) []*gates.Ciphertext

func TestBatchGates(t *testing.T) {
	sk := key.NewSecretKey()
	ck := cloudkey.NewCloudKey(sk)
	defer gates.Release(ck)
	testCases := [][2]bool{{false, false}, {false, true}, {true, false}, {true, true}}
	inputs := make([][2]*gates.Ciphertext, len(testCases))
	for i, tc := range testCases {
		inputs[i] = [2]*gates.Ciphertext{encrypt(tc[0], sk), encrypt(tc[1], sk)}
	}
	cases := []struct {
		name   string
		gpu    batch2
		scalar gate2
		want   func(a, b bool) bool
	}{
		{"BatchNAND", gates.BatchNAND, refgates.NAND, func(a, b bool) bool { return !(a && b) }},
		{"BatchAND", gates.BatchAND, refgates.AND, func(a, b bool) bool { return a && b }},
		{"BatchOR", gates.BatchOR, refgates.OR, func(a, b bool) bool { return a || b }},
		{"BatchXOR", gates.BatchXOR, refgates.XOR, func(a, b bool) bool { return a != b }},
		{"BatchNOR", gates.BatchNOR, refgates.NOR, func(a, b bool) bool { return !(a || b) }},
		{"BatchXNOR", gates.BatchXNOR, refgates.XNOR, func(a, b bool) bool { return a == b }},
	}
	for _, c := range cases {
		results := c.gpu(inputs, ck)
		if len(results) != len(testCases) {
			t.Fatalf("%s returned %d results, expected %d", c.name, len(results), len(testCases))
		}
		for i, result := range results {
			if dec := result.DecryptBool(sk.KeyLv0); dec != c.want(testCases[i][0], testCases[i][1]) {
				t.Errorf("%s[%d] = %v, expected %v", c.name, i, dec, c.want(testCases[i][0], testCases[i][1]))
			}
			// the reference's SCALAR gate is the yardstick (its BatchXNOR computes XOR, gates/gates.go:293)
			if ref := c.scalar(inputs[i][0], inputs[i][1], ck); !sameWords(result, ref) {
				t.Errorf("%s[%d]: GPU ciphertext differs from the reference's scalar gate", c.name, i)
			}
		}
	}
}
