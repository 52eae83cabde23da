// The seam functions of this package and of shim/go/trlwe against the reference's own functions of the same name, WORD FOR WORD, on
// the same operands and the same cloud key (both implementations are in one process here; at the N = 1024, L = 3, Bgbit = 6 sets the
// transforms are exact, so bit equality is the bar: DESIGN.md section 4).  Needs Go, a go-tfhe checkout and an MI355X; never run in this
// repository's image (executed there by the Go-subset interpreter with the C layer mocked: tests/golden/goref/shim_go_test_run.json).
package trgsw_test

import (
	"testing"

	"github.com/thedonutfactory/go-tfhe-gpu/gpu"
	"github.com/thedonutfactory/go-tfhe-gpu/trgsw"
	gputrlwe "github.com/thedonutfactory/go-tfhe-gpu/trlwe"
	"github.com/thedonutfactory/go-tfhe/cloudkey"
	"github.com/thedonutfactory/go-tfhe/key"
	"github.com/thedonutfactory/go-tfhe/params"
	"github.com/thedonutfactory/go-tfhe/poly"
	"github.com/thedonutfactory/go-tfhe/tlwe"
	reftrgsw "github.com/thedonutfactory/go-tfhe/trgsw"
	"github.com/thedonutfactory/go-tfhe/trlwe"
)

// arbitrary torus words (a linear congruential sequence: any words do, parity is on the arithmetic)
var lcg uint32 = 0x7F4E0601

func word() params.Torus {
	lcg = lcg*1664525 + 1013904223
	return params.Torus(lcg)
}

func randomTRLWE() *trlwe.TRLWELv1 {
	t := trlwe.NewTRLWELv1()
	for i := range t.A {
		t.A[i] = word()
		t.B[i] = word()
	}
	return t
}

func sameTorus(a, b []params.Torus) bool {
	if len(a) != len(b) {
		return false
	}
	for i := range a {
		if a[i] != b[i] {
			return false
		}
	}
	return true
}

func sameTRLWE(a, b *trlwe.TRLWELv1) bool {
	return sameTorus(a.A, b.A) && sameTorus(a.B, b.B)
}

func TestExternalProductAndCMUXWithAFreeStandingOperand(t *testing.T) {
	sk := key.NewSecretKey()
	ck := cloudkey.NewCloudKey(sk)
	pe := poly.NewEvaluator(params.GetTRGSWLv1().N)
	g := ck.BootstrappingKey[0] // any *TRGSWLv1FFT: handed over with the call, not looked up in a loaded key
	in0, in1 := randomTRLWE(), randomTRLWE()
	got := trgsw.ExternalProductWithFFT(g, in0, ck.DecompositionOffset, pe)
	want := reftrgsw.ExternalProductWithFFT(g, in0, ck.DecompositionOffset, pe)
	if !sameTRLWE(got, want) {
		t.Errorf("ExternalProductWithFFT differs from the reference's")
	}
	got = trgsw.CMUX(in0, in1, g, ck.DecompositionOffset, pe)
	want = reftrgsw.CMUX(in0, in1, g, ck.DecompositionOffset, pe)
	if !sameTRLWE(got, want) {
		t.Errorf("CMUX differs from the reference's")
	}
}

func TestBlindRotateExtractAndKeySwitchEqualTheReferences(t *testing.T) {
	sk := key.NewSecretKey()
	ck := cloudkey.NewCloudKey(sk)
	defer gpu.Detach(ck.BootstrappingKey)
	defer gpu.DetachKSK(ck.KeySwitchingKey) // the key switch of an extracted sample attached the key-switching key on its own
	pe := poly.NewEvaluator(params.GetTRGSWLv1().N)
	alpha := params.GetTLWELv0().ALPHA
	srcs := []*tlwe.TLWELv0{
		tlwe.NewTLWELv0().EncryptBool(true, alpha, sk.KeyLv0),
		tlwe.NewTLWELv0().EncryptBool(false, alpha, sk.KeyLv0),
		tlwe.NewTLWELv0().EncryptBool(true, alpha, sk.KeyLv0),
	}
	accs := trgsw.BatchBlindRotate(srcs, ck.BlindRotateTestvec, ck.BootstrappingKey, ck.DecompositionOffset)
	if len(accs) != len(srcs) {
		t.Fatalf("BatchBlindRotate returned %d accumulators for %d inputs", len(accs), len(srcs))
	}
	for i, src := range srcs {
		want := reftrgsw.BlindRotate(src, ck.BlindRotateTestvec, ck.BootstrappingKey, ck.DecompositionOffset, pe)
		if !sameTRLWE(accs[i], want) {
			t.Errorf("BatchBlindRotate[%d] differs from the reference's BlindRotate", i)
		}
	}
	one := trgsw.BlindRotate(srcs[1], ck.BlindRotateTestvec, ck.BootstrappingKey, ck.DecompositionOffset, pe)
	if !sameTRLWE(one, accs[1]) {
		t.Errorf("BlindRotate differs from BatchBlindRotate on the same input")
	}
	for _, k := range []int{0, 1, params.GetTRGSWLv1().N - 1} {
		got := gputrlwe.SampleExtractIndex(accs[0], k)
		want := trlwe.SampleExtractIndex(accs[0], k)
		if !sameTorus(got.P, want.P) {
			t.Errorf("SampleExtractIndex(%d) differs from the reference's", k)
		}
	}
	ext := tlwe.NewTLWELv1()
	gputrlwe.SampleExtractIndexAssign(accs[0], 0, ext)
	got := trgsw.IdentityKeySwitching(ext, ck.KeySwitchingKey)
	want := reftrgsw.IdentityKeySwitching(ext, ck.KeySwitchingKey)
	if !sameTorus(got.P, want.P) {
		t.Errorf("IdentityKeySwitching differs from the reference's")
	}
	out := tlwe.NewTLWELv0()
	trgsw.IdentityKeySwitchingAssign(ext, ck.KeySwitchingKey, out)
	if !sameTorus(out.P, want.P) {
		t.Errorf("IdentityKeySwitchingAssign differs from the reference's")
	}
	if !got.DecryptBool(sk.KeyLv0) {
		t.Errorf("the bootstrapped sample of an encryption of true does not decrypt to true")
	}
}
