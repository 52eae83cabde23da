// Package gpu binds libtfhe_hip.so (include/tfhe_hip.h), the MI355X gate-bootstrapping engine, for go-tfhe.
//
// It is the only package of the shim that imports "C".  The two packages above it, gates and evaluator, keep the
// reference's names and signatures (gates/gates.go:16-126,156-312, evaluator/evaluator.go:110-157,
// evaluator/programmable_bootstrap.go:16-115) and forward here.
//
// Types at the boundary: params.Torus is a DEFINED type (params/params.go:27, `type Torus uint32`), not an alias, so a
// []params.Torus is never assignable to or appendable into a []uint32.  Every ciphertext buffer in this package is a
// []params.Torus and crosses into C through torusPtr, which reinterprets the backing array (same size, same layout).
//
// Errors: every C entry point returns 0 or a negative code; the reference panics on every error on this path, and so does
// check.  tfhe_last_error() is thread-local and a goroutine may change OS threads between two cgo calls, so every sequence
// "call, then read the message" runs inside locked().
//
// Never compiled in this repository's image (no Go toolchain); tests/test_go_shim_static.py resolves every identifier used
// below against the reference sources and the C header.
package gpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../go-tfhe_amd/lib -ltfhe_hip -Wl,-rpath,${SRCDIR}/../../../go-tfhe_amd/lib
#include "tfhe_hip.h"
*/
import "C"

import (
	"runtime"
	"sync"
	"sync/atomic"
	"unsafe"

	"github.com/thedonutfactory/go-tfhe/cloudkey"
	"github.com/thedonutfactory/go-tfhe/key"
	"github.com/thedonutfactory/go-tfhe/params"
	"github.com/thedonutfactory/go-tfhe/tlwe"
	"github.com/thedonutfactory/go-tfhe/trgsw"
	"github.com/thedonutfactory/go-tfhe/trlwe"
)

// Op codes of tfhe_gate_batch (include/tfhe_hip.h; gates/gates.go:26-114).
const (
	OpNAND  = int(C.TFHE_OP_NAND)
	OpAND   = int(C.TFHE_OP_AND)
	OpOR    = int(C.TFHE_OP_OR)
	OpXOR   = int(C.TFHE_OP_XOR)
	OpXNOR  = int(C.TFHE_OP_XNOR)
	OpNOR   = int(C.TFHE_OP_NOR)
	OpANDNY = int(C.TFHE_OP_ANDNY)
	OpANDYN = int(C.TFHE_OP_ANDYN)
	OpORNY  = int(C.TFHE_OP_ORNY)
	OpORYN  = int(C.TFHE_OP_ORYN)
	OpMUX   = int(C.TFHE_OP_MUX)
)

// CloudKey is cloudkey.CloudKey (cloudkey/cloudkey.go:16-21) resident on ONE GPU: a context of the engine holding the
// bootstrapping key, the key-switching key, the decomposition offset and the gate test vector.
type CloudKey struct {
	ctx    *C.tfhe_ctx
	n1     int // words of one tlwe.TLWELv0: n + 1 (tlwe/tlwe.go:11-21)
	ringN  int // coefficients of one polynomial: params.GetTRGSWLv1().N
	device int
}

func locked(f func()) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	f()
}

func check(rc C.int) {
	if rc != 0 {
		panic("tfhe_hip: " + C.GoString(C.tfhe_last_error()))
	}
}

// torusPtr reinterprets the backing array of a Torus slice for C (params/params.go:27: Torus is a uint32).
func torusPtr(p []params.Torus) *C.uint32_t {
	if len(p) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&p[0]))
}

func currentParams() C.tfhe_params {
	g := params.GetTRGSWLv1()
	l0 := params.GetTLWELv0()
	return C.tfhe_params{n: C.int32_t(l0.N), N: C.int32_t(g.N), Nbit: C.int32_t(g.NBIT), L: C.int32_t(g.L),
		Bgbit: C.int32_t(g.BGBIT), basebit: C.int32_t(g.BASEBIT), t: C.int32_t(g.IKS_T)}
}

func newContext(device int) *CloudKey {
	p := currentParams()
	k := &CloudKey{n1: params.GetTLWELv0().N + 1, ringN: params.GetTRGSWLv1().N, device: device}
	locked(func() { check(C.tfhe_ctx_create(&p, C.int(device), &k.ctx)) })
	return k
}

// DeviceCount is the number of visible GPUs.
func DeviceCount() int {
	var n C.int
	locked(func() { check(C.tfhe_device_count(&n)) })
	return int(n)
}

// UploadKeys flattens the Go pointer graphs of the two keys once (cgo may not pass []*trgsw.TRGSWLv1FFT) and loads them
// on GPU `device`.  ksk may be nil (blind rotation only).
//
//	bsk []*trgsw.TRGSWLv1FFT -> [n][2L][2][N] float64: TRLWEFFT[r].A.Coeffs then .B.Coeffs, the reference FourierPoly
//	    layout as it is (trgsw/trgsw.go:60-68, poly/poly.go:54-62)
//	ksk []*tlwe.TLWELv0      -> [N*t*base][n+1] Torus, index base*t*i + base*j + k (trgsw/keyswitch.go:29)
func UploadKeys(bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0, device int) *CloudKey {
	g := params.GetTRGSWLv1()
	k := newContext(device)
	if len(bsk) != k.n1-1 {
		panic("tfhe_hip: bootstrapping key length does not match params.GetTLWELv0().N")
	}
	flat := make([]float64, 0, len(bsk)*2*g.L*2*g.N)
	for _, gsw := range bsk {
		for _, row := range gsw.TRLWEFFT {
			flat = append(flat, row.A.Coeffs...)
			flat = append(flat, row.B.Coeffs...)
		}
	}
	locked(func() { check(C.tfhe_load_bsk_fourier(k.ctx, (*C.double)(unsafe.Pointer(&flat[0])))) })
	if ksk != nil {
		k.LoadKSK(ksk)
	}
	return k
}

// LoadKSK loads (or replaces) the key-switching key of this context.
func (k *CloudKey) LoadKSK(ksk []*tlwe.TLWELv0) {
	rows := make([]params.Torus, 0, len(ksk)*k.n1)
	for _, row := range ksk {
		rows = append(rows, row.P...)
	}
	locked(func() { check(C.tfhe_load_ksk(k.ctx, torusPtr(rows))) })
}

// Upload is UploadKeys for a whole cloudkey.CloudKey (cloudkey/cloudkey.go:16-21).
func Upload(ck *cloudkey.CloudKey, device int) *CloudKey {
	return UploadKeys(ck.BootstrappingKey, ck.KeySwitchingKey, device)
}

// NewCloudKey replaces cloudkey.NewCloudKey(secretKey) (cloudkey/cloudkey.go:24-31): both keys are GENERATED on the GPU
// from the binary secret keys (milliseconds instead of the 5-10 s of the Go path; nothing but the two secret keys crosses
// PCIe).  The seed is 128 bits from the OS, like the reference's auto-seeded math/rand.
func NewCloudKey(sk *key.SecretKey, device int) *CloudKey {
	k := newContext(device)
	locked(func() {
		check(C.tfhe_keygen_cloud_seeded(k.ctx, torusPtr(sk.KeyLv0), torusPtr(sk.KeyLv1),
			C.double(params.KSKAlpha()), C.double(params.BSKAlpha()), nil))
	})
	return k
}

// CloneTo makes a replica of this key on another GPU, copied GPU to GPU (tfhe_ctx_clone_to: hipMemcpyPeerAsync over
// xGMI; a device-to-device copy when `device` is this key's own GPU).
func (k *CloudKey) CloneTo(device int) *CloudKey {
	r := &CloudKey{n1: k.n1, ringN: k.ringN, device: device}
	locked(func() { check(C.tfhe_ctx_clone_to(k.ctx, C.int(device), &r.ctx)) })
	return r
}

// Close releases the GPU context.
func (k *CloudKey) Close() {
	if k.ctx != nil {
		locked(func() { check(C.tfhe_ctx_destroy(k.ctx)) })
		k.ctx = nil
	}
}

// Device is the GPU this key lives on.
func (k *CloudKey) Device() int { return k.device }

func flatten(cts []*tlwe.TLWELv0, n1 int) []params.Torus {
	flat := make([]params.Torus, 0, len(cts)*n1)
	for _, c := range cts {
		if len(c.P) != n1 {
			panic("tfhe_hip: ciphertext length does not match params.GetTLWELv0().N + 1")
		}
		flat = append(flat, c.P...)
	}
	return flat
}

func unflatten(flat []params.Torus, n1 int) []*tlwe.TLWELv0 {
	res := make([]*tlwe.TLWELv0, len(flat)/n1)
	for i := range res {
		res[i] = &tlwe.TLWELv0{P: flat[i*n1 : (i+1)*n1 : (i+1)*n1]}
	}
	return res
}

func flattenTestvec(tv *trlwe.TRLWELv1, ringN int) []params.Torus {
	if len(tv.A) != ringN || len(tv.B) != ringN {
		panic("tfhe_hip: test vector length does not match params.GetTRGSWLv1().N")
	}
	flat := make([]params.Torus, 0, 2*ringN)
	flat = append(flat, tv.A...)
	flat = append(flat, tv.B...)
	return flat
}

// GateBatch replaces gates.Batch{NAND,AND,OR,XOR,NOR,XNOR} (gates/gates.go:156-312) and, with one item, the scalar
// gates (gates/gates.go:26-104): out[i] = op(a[i], b[i]) -- linear preparation (evaluator/gates_helper.go:10-63), blind
// rotation, sample extraction and key switching in one call.  op = OpMUX takes the third operand: out[i] = a[i] ? b[i] :
// c[i], three bootstraps each (gates/gates.go:107-114).  Every output owns its storage.
func (k *CloudKey) GateBatch(op int, a, b, c []*tlwe.TLWELv0) []*tlwe.TLWELv0 {
	if len(a) != len(b) || (c != nil && len(c) != len(a)) {
		panic("tfhe_hip: operand counts differ")
	}
	if op == OpMUX && c == nil {
		panic("tfhe_hip: MUX needs the third operand")
	}
	if len(a) == 0 {
		return []*tlwe.TLWELv0{}
	}
	fa := flatten(a, k.n1)
	fb := flatten(b, k.n1)
	var fc []params.Torus
	if c != nil {
		fc = flatten(c, k.n1)
	}
	out := make([]params.Torus, len(fa))
	locked(func() {
		check(C.tfhe_gate_batch(k.ctx, nil, C.int(op), torusPtr(fa), torusPtr(fb), torusPtr(fc), torusPtr(out), C.int(len(a))))
	})
	return unflatten(out, k.n1)
}

// GateBatchOps is GateBatch with one op code per item (a mixed gate stream; c may be nil when no op is OpMUX).
func (k *CloudKey) GateBatchOps(ops []uint8, a, b, c []*tlwe.TLWELv0) []*tlwe.TLWELv0 {
	if len(ops) != len(a) || len(a) != len(b) || (c != nil && len(c) != len(a)) {
		panic("tfhe_hip: operand counts differ")
	}
	if len(a) == 0 {
		return []*tlwe.TLWELv0{}
	}
	fa := flatten(a, k.n1)
	fb := flatten(b, k.n1)
	var fc []params.Torus
	if c != nil {
		fc = flatten(c, k.n1)
	}
	out := make([]params.Torus, len(fa))
	locked(func() {
		check(C.tfhe_gate_batch(k.ctx, (*C.uint8_t)(unsafe.Pointer(&ops[0])), 0, torusPtr(fa), torusPtr(fb), torusPtr(fc),
			torusPtr(out), C.int(len(a))))
	})
	return unflatten(out, k.n1)
}

// BootstrapBatch replaces Evaluator.BootstrapAssign / BootstrapLUTAssign over a batch (evaluator/evaluator.go:139-148,
// evaluator/programmable_bootstrap.go:93-115): out[i] = KeySwitch(SampleExtract(BlindRotate(cts[i], testvec))).
// testvec nil = the gate test vector (cloudkey/cloudkey.go:74-85).
func (k *CloudKey) BootstrapBatch(cts []*tlwe.TLWELv0, testvec *trlwe.TRLWELv1) []*tlwe.TLWELv0 {
	if len(cts) == 0 {
		return []*tlwe.TLWELv0{}
	}
	in := flatten(cts, k.n1)
	var tv []params.Torus
	if testvec != nil {
		tv = flattenTestvec(testvec, k.ringN)
	}
	out := make([]params.Torus, len(in))
	locked(func() {
		check(C.tfhe_bootstrap_batch(k.ctx, torusPtr(in), torusPtr(tv), 0, torusPtr(out), C.int(len(cts))))
	})
	return unflatten(out, k.n1)
}

// BootstrapBatchTables is BootstrapBatch with one lookup table per item.
func (k *CloudKey) BootstrapBatchTables(cts []*tlwe.TLWELv0, testvecs []*trlwe.TRLWELv1) []*tlwe.TLWELv0 {
	if len(cts) != len(testvecs) {
		panic("tfhe_hip: one test vector per ciphertext")
	}
	if len(cts) == 0 {
		return []*tlwe.TLWELv0{}
	}
	in := flatten(cts, k.n1)
	tv := make([]params.Torus, 0, len(cts)*2*k.ringN)
	for _, t := range testvecs {
		tv = append(tv, flattenTestvec(t, k.ringN)...)
	}
	out := make([]params.Torus, len(in))
	locked(func() {
		check(C.tfhe_bootstrap_batch(k.ctx, torusPtr(in), torusPtr(tv), 1, torusPtr(out), C.int(len(cts))))
	})
	return unflatten(out, k.n1)
}

// BlindRotateBatch replaces Evaluator.BlindRotateAssign / trgsw.BatchBlindRotate (evaluator/evaluator.go:110-135,
// trgsw/trgsw.go:234-252).  testvec nil = the gate test vector.
func (k *CloudKey) BlindRotateBatch(cts []*tlwe.TLWELv0, testvec *trlwe.TRLWELv1) []*trlwe.TRLWELv1 {
	res := make([]*trlwe.TRLWELv1, len(cts))
	if len(cts) == 0 {
		return res
	}
	in := flatten(cts, k.n1)
	var tv []params.Torus
	if testvec != nil {
		tv = flattenTestvec(testvec, k.ringN)
	}
	out := make([]params.Torus, len(cts)*2*k.ringN)
	locked(func() {
		check(C.tfhe_blind_rotate_batch(k.ctx, torusPtr(in), torusPtr(tv), 0, torusPtr(out), C.int(len(cts)), -1))
	})
	for i := range res {
		lo := i * 2 * k.ringN
		res[i] = &trlwe.TRLWELv1{A: out[lo : lo+k.ringN : lo+k.ringN], B: out[lo+k.ringN : lo+2*k.ringN : lo+2*k.ringN]}
	}
	return res
}

// ---- the trgsw / trlwe seams with caller-supplied operands (include/tfhe_hip.h: tfhe_external_product_with, tfhe_cmux_with,
// tfhe_sample_extract_batch, tfhe_keyswitch_batch; what shim/go/trgsw, shim/go/trlwe and Evaluator.ExternalProductAssign /
// CMuxAssign forward to) ----

// DecompositionOffset is cloudkey.CloudKey.DecompositionOffset (cloudkey/cloudkey.go:60-71) as the engine derived it.
func (k *CloudKey) DecompositionOffset() params.Torus {
	var off C.uint32_t
	locked(func() { check(C.tfhe_ctx_decomposition_offset(k.ctx, &off)) })
	return params.Torus(off)
}

// flattenTRGSW lays one TRGSWLv1FFT out as [2L][2][N] float64: TRLWEFFT[r].A.Coeffs then .B.Coeffs, the reference's own
// FourierPoly layout (trgsw/trgsw.go:60-68, poly/poly.go:54-62) -- one element of what UploadKeys flattens n of.
func flattenTRGSW(g *trgsw.TRGSWLv1FFT, ringN int) []float64 {
	flat := make([]float64, 0, len(g.TRLWEFFT)*2*ringN)
	for _, row := range g.TRLWEFFT {
		if len(row.A.Coeffs) != ringN || len(row.B.Coeffs) != ringN {
			panic("tfhe_hip: TRGSW row length does not match params.GetTRGSWLv1().N")
		}
		flat = append(flat, row.A.Coeffs...)
		flat = append(flat, row.B.Coeffs...)
	}
	if len(flat) != 2*params.GetTRGSWLv1().L*2*ringN {
		panic("tfhe_hip: TRGSW operand must hold 2*L rows")
	}
	return flat
}

func flattenTRLWE(in []*trlwe.TRLWELv1, ringN int) []params.Torus {
	flat := make([]params.Torus, 0, len(in)*2*ringN)
	for _, t := range in {
		flat = append(flat, flattenTestvec(t, ringN)...)
	}
	return flat
}

func unflattenTRLWE(flat []params.Torus, ringN int) []*trlwe.TRLWELv1 {
	res := make([]*trlwe.TRLWELv1, len(flat)/(2*ringN))
	for i := range res {
		lo := i * 2 * ringN
		res[i] = &trlwe.TRLWELv1{A: flat[lo : lo+ringN : lo+ringN], B: flat[lo+ringN : lo+2*ringN : lo+2*ringN]}
	}
	return res
}

// ExternalProductWith replaces trgsw.ExternalProductWithFFT / Evaluator.ExternalProductAssign (trgsw/trgsw.go:108-137,
// evaluator/evaluator.go:50-81) for ANY TRGSW operand: out[i] = g (x) in[i].  Needs no loaded key.
func (k *CloudKey) ExternalProductWith(g *trgsw.TRGSWLv1FFT, in []*trlwe.TRLWELv1, decompositionOffset params.Torus) []*trlwe.TRLWELv1 {
	if len(in) == 0 {
		return []*trlwe.TRLWELv1{}
	}
	gf := flattenTRGSW(g, k.ringN)
	fin := flattenTRLWE(in, k.ringN)
	out := make([]params.Torus, len(fin))
	locked(func() {
		check(C.tfhe_external_product_with(k.ctx, (*C.double)(unsafe.Pointer(&gf[0])), C.uint32_t(decompositionOffset), torusPtr(fin), torusPtr(out), C.int(len(in))))
	})
	return unflattenTRLWE(out, k.ringN)
}

// CMuxWith replaces trgsw.CMUX / Evaluator.CMuxAssign (trgsw/trgsw.go:173-194, evaluator/evaluator.go:85-106):
// out[i] = ct0[i] + cond (x) (ct1[i] - ct0[i]).
func (k *CloudKey) CMuxWith(cond *trgsw.TRGSWLv1FFT, ct0, ct1 []*trlwe.TRLWELv1, decompositionOffset params.Torus) []*trlwe.TRLWELv1 {
	if len(ct0) != len(ct1) {
		panic("tfhe_hip: operand counts differ")
	}
	if len(ct0) == 0 {
		return []*trlwe.TRLWELv1{}
	}
	gf := flattenTRGSW(cond, k.ringN)
	f0 := flattenTRLWE(ct0, k.ringN)
	f1 := flattenTRLWE(ct1, k.ringN)
	out := make([]params.Torus, len(f0))
	locked(func() {
		check(C.tfhe_cmux_with(k.ctx, (*C.double)(unsafe.Pointer(&gf[0])), C.uint32_t(decompositionOffset), torusPtr(f0), torusPtr(f1), torusPtr(out), C.int(len(ct0))))
	})
	return unflattenTRLWE(out, k.ringN)
}

// SampleExtract replaces trlwe.SampleExtractIndex (trlwe/trlwe.go:114-128, trlwe/trlwe_ops.go:10-21) for any index.
func (k *CloudKey) SampleExtract(in []*trlwe.TRLWELv1, index int) []*tlwe.TLWELv1 {
	res := make([]*tlwe.TLWELv1, len(in))
	if len(in) == 0 {
		return res
	}
	fin := flattenTRLWE(in, k.ringN)
	w := k.ringN + 1
	out := make([]params.Torus, len(in)*w)
	locked(func() {
		check(C.tfhe_sample_extract_batch(k.ctx, torusPtr(fin), C.int(index), torusPtr(out), C.int(len(in))))
	})
	for i := range res {
		res[i] = &tlwe.TLWELv1{P: out[i*w : (i+1)*w : (i+1)*w]}
	}
	return res
}

// KeySwitch replaces trgsw.IdentityKeySwitching (trgsw/trgsw.go:285-312, trgsw/keyswitch.go:10-37) on extracted samples.
func (k *CloudKey) KeySwitch(in []*tlwe.TLWELv1) []*tlwe.TLWELv0 {
	if len(in) == 0 {
		return []*tlwe.TLWELv0{}
	}
	w := k.ringN + 1
	fin := make([]params.Torus, 0, len(in)*w)
	for _, c := range in {
		if len(c.P) != w {
			panic("tfhe_hip: TLWELv1 length does not match params.GetTRGSWLv1().N + 1")
		}
		fin = append(fin, c.P...)
	}
	out := make([]params.Torus, len(in)*k.n1)
	locked(func() {
		check(C.tfhe_keyswitch_batch(k.ctx, torusPtr(fin), torusPtr(out), C.int(len(in))))
	})
	return unflatten(out, k.n1)
}

// BootstrapExtendedBatch: a table of ext*N entries (polyExtendFactor = ext: the Uint6/7/8 sets, params/params.go:399-402,
// which the reference leaves out); lutExt is [ext][2][N] flattened (include/tfhe_hip.h, tfhe_bootstrap_extended_batch).
func (k *CloudKey) BootstrapExtendedBatch(cts []*tlwe.TLWELv0, lutExt []params.Torus, ext int) []*tlwe.TLWELv0 {
	if len(lutExt) != ext*2*k.ringN {
		panic("tfhe_hip: extended table must hold ext*2*N words")
	}
	if len(cts) == 0 {
		return []*tlwe.TLWELv0{}
	}
	in := flatten(cts, k.n1)
	out := make([]params.Torus, len(in))
	locked(func() {
		check(C.tfhe_bootstrap_extended_batch(k.ctx, torusPtr(in), torusPtr(lutExt), 0, C.int(ext), torusPtr(out), C.int(len(cts))))
	})
	return unflatten(out, k.n1)
}

// Save / Load: the GPU-resident key as two self-describing blobs (the reference has no serialised cloud key).
// which: 0 = bootstrapping key, 1 = key-switching key.  Load panics -- like everything on this path -- if the blob belongs
// to another parameter set, is the other key, comes from a library with a different device layout, or is truncated.
func (k *CloudKey) Save(which int) []byte {
	var n C.size_t
	locked(func() { check(C.tfhe_key_size(k.ctx, C.int(which), &n)) })
	blob := make([]byte, int(n))
	locked(func() { check(C.tfhe_key_export(k.ctx, C.int(which), unsafe.Pointer(&blob[0]))) })
	return blob
}

func (k *CloudKey) Load(which int, blob []byte) {
	locked(func() { check(C.tfhe_key_import(k.ctx, C.int(which), unsafe.Pointer(&blob[0]), C.size_t(len(blob)))) })
}

// ClonePath says how CloneTo brought this replica's keys to its GPU: 0 not a clone, 1 same GPU (device-to-device copy),
// 2 peer copy GPU to GPU (xGMI), 3 staged through page-locked host memory (the devices are not peers).
func (k *CloudKey) ClonePath() int {
	var v C.int
	locked(func() { check(C.tfhe_ctx_get_option(k.ctx, C.TFHE_OPT_CLONE_PATH, &v)) })
	return int(v)
}

// CloudKeySet is one cloud key on several GPUs of a node, used from ONE process: the reference fans a batch out over
// goroutines that share the read-only keys (trgsw/trgsw.go:234-252); with one key copy per GPU the fan-out needs a replica
// per device first.  Replicas are made GPU to GPU (CloneTo); batch calls shard contiguously, one goroutine per replica,
// results in index order; scalar calls go round-robin, and the concurrent callers of one replica are combined into one
// launch by the engine (include/tfhe_hip.h, tfhe_gate_batch).  There is no exchange step: bootstraps are independent.
type CloudKeySet struct {
	replicas []*CloudKey
	next     uint32
}

// NewCloudKeySet replicates src onto `devices` (an index may repeat: two contexts on one GPU are two independent
// submitters); nil = every visible GPU.  src stays the caller's.
func NewCloudKeySet(src *CloudKey, devices []int) *CloudKeySet {
	if devices == nil {
		for d := 0; d < DeviceCount(); d++ {
			devices = append(devices, d)
		}
	}
	if len(devices) == 0 {
		panic("tfhe_hip: CloudKeySet needs at least one device")
	}
	s := &CloudKeySet{}
	for _, d := range devices {
		s.replicas = append(s.replicas, src.CloneTo(d))
	}
	return s
}

func (s *CloudKeySet) Len() int { return len(s.replicas) }

func (s *CloudKeySet) Replica(i int) *CloudKey { return s.replicas[i] }

// Pick returns the replicas in turn (scalar gates issued from many goroutines spread over the GPUs).
func (s *CloudKeySet) Pick() *CloudKey {
	i := atomic.AddUint32(&s.next, 1)
	return s.replicas[int(i)%len(s.replicas)]
}

func (s *CloudKeySet) Close() {
	for _, r := range s.replicas {
		r.Close()
	}
	s.replicas = nil
}

// shard runs f(g, lo, hi) for the contiguous ranges [g*B/G, (g+1)*B/G) of B items, one goroutine per replica, and
// re-panics the first panic of any of them on the caller's goroutine (SURVEY.md 8e).
func (s *CloudKeySet) shard(B int, f func(r *CloudKey, lo, hi int)) {
	G := len(s.replicas)
	panics := make([]interface{}, G)
	var wg sync.WaitGroup
	for g := 0; g < G; g++ {
		lo := B * g / G
		hi := B * (g + 1) / G
		if hi == lo {
			continue
		}
		wg.Add(1)
		go func(g, lo, hi int) {
			defer wg.Done()
			defer func() { panics[g] = recover() }()
			f(s.replicas[g], lo, hi)
		}(g, lo, hi)
	}
	wg.Wait()
	for _, p := range panics {
		if p != nil {
			panic(p)
		}
	}
}

// GateBatch is CloudKey.GateBatch over all replicas.
func (s *CloudKeySet) GateBatch(op int, a, b, c []*tlwe.TLWELv0) []*tlwe.TLWELv0 {
	if len(a) != len(b) || (c != nil && len(c) != len(a)) {
		panic("tfhe_hip: operand counts differ")
	}
	out := make([]*tlwe.TLWELv0, len(a))
	s.shard(len(a), func(r *CloudKey, lo, hi int) {
		var cs []*tlwe.TLWELv0
		if c != nil {
			cs = c[lo:hi]
		}
		copy(out[lo:hi], r.GateBatch(op, a[lo:hi], b[lo:hi], cs))
	})
	return out
}

// BootstrapBatch is CloudKey.BootstrapBatch over all replicas.
func (s *CloudKeySet) BootstrapBatch(cts []*tlwe.TLWELv0, testvec *trlwe.TRLWELv1) []*tlwe.TLWELv0 {
	out := make([]*tlwe.TLWELv0, len(cts))
	s.shard(len(cts), func(r *CloudKey, lo, hi int) {
		copy(out[lo:hi], r.BootstrapBatch(cts[lo:hi], testvec))
	})
	return out
}

// BlindRotateBatch is CloudKey.BlindRotateBatch over all replicas (trgsw.BatchBlindRotate, trgsw/trgsw.go:234-252).
func (s *CloudKeySet) BlindRotateBatch(cts []*tlwe.TLWELv0, testvec *trlwe.TRLWELv1) []*trlwe.TRLWELv1 {
	out := make([]*trlwe.TRLWELv1, len(cts))
	s.shard(len(cts), func(r *CloudKey, lo, hi int) {
		copy(out[lo:hi], r.BlindRotateBatch(cts[lo:hi], testvec))
	})
	return out
}

// KeySwitch is CloudKey.KeySwitch over all replicas.
func (s *CloudKeySet) KeySwitch(in []*tlwe.TLWELv1) []*tlwe.TLWELv0 {
	out := make([]*tlwe.TLWELv0, len(in))
	s.shard(len(in), func(r *CloudKey, lo, hi int) {
		copy(out[lo:hi], r.KeySwitch(in[lo:hi]))
	})
	return out
}

// The keys a caller hands to gates.* / evaluator.* / trgsw.* are Go pointer graphs; they are uploaded ONCE, on first use, and
// found again by identity: a cloud key by the first TRGSW of its bootstrapping key (and, once seen, the first row of its
// key-switching key), a key-switching key on its own (trgsw.IdentityKeySwitching) by its first row.
var registry = struct {
	sync.Mutex
	devices []int
	sets    map[*trgsw.TRGSWLv1FFT]*attachedKey
	kskOnly map[*tlwe.TLWELv0]*CloudKeySet
	scratch *CloudKey
	scratchP C.tfhe_params
}{sets: map[*trgsw.TRGSWLv1FFT]*attachedKey{}, kskOnly: map[*tlwe.TLWELv0]*CloudKeySet{}}

type attachedKey struct {
	set *CloudKeySet
	ksk *tlwe.TLWELv0 // identity of the key-switching key loaded beside it (nil: none yet)
}

// SetDevices chooses the GPUs that keys attached FROM NOW ON are replicated to (nil = every visible GPU).
func SetDevices(devices []int) {
	registry.Lock()
	defer registry.Unlock()
	registry.devices = devices
}

func targetDevices() []int {
	if len(registry.devices) > 0 {
		return registry.devices
	}
	devices := []int{}
	for d := 0; d < DeviceCount(); d++ {
		devices = append(devices, d)
	}
	return devices
}

// adoptSet makes the replica set of a freshly uploaded key: src itself serves its own GPU (no second copy of the keys there,
// no redundant device-to-device copy) and is cloned GPU to GPU to the others.  If a clone panics, what was made is closed.
func adoptSet(src *CloudKey, devices []int) *CloudKeySet {
	s := &CloudKeySet{}
	ok := false
	defer func() {
		if !ok {
			s.Close()
			src.Close()
		}
	}()
	used := false
	for _, d := range devices {
		if d == src.device && !used {
			s.replicas = append(s.replicas, src)
			used = true
		} else {
			s.replicas = append(s.replicas, src.CloneTo(d))
		}
	}
	if !used {
		src.Close()
	}
	ok = true
	return s
}

// Attached returns the GPU replicas of (bsk, ksk), uploading and replicating them on first use.  ksk may be nil for
// callers that only blind-rotate; it is loaded into every replica the first time a caller passes it.  A later call with a
// DIFFERENT key-switching key for the same bootstrapping key panics (two cloud keys cannot share a bootstrapping key).
func Attached(bsk []*trgsw.TRGSWLv1FFT, ksk []*tlwe.TLWELv0) *CloudKeySet {
	if len(bsk) == 0 {
		panic("tfhe_hip: empty bootstrapping key")
	}
	registry.Lock()
	defer registry.Unlock()
	e, ok := registry.sets[bsk[0]]
	if !ok {
		devices := targetDevices()
		src := UploadKeys(bsk, ksk, devices[0])
		e = &attachedKey{set: adoptSet(src, devices)}
		if ksk != nil {
			e.ksk = ksk[0]
		}
		registry.sets[bsk[0]] = e
	} else if ksk != nil && e.ksk == nil {
		for i := 0; i < e.set.Len(); i++ {
			e.set.Replica(i).LoadKSK(ksk)
		}
		e.ksk = ksk[0]
	} else if ksk != nil && e.ksk != ksk[0] {
		panic("tfhe_hip: this bootstrapping key is attached with a different key-switching key")
	}
	return e.set
}

// AttachedKSK returns GPU contexts holding the key-switching key ksk -- the replicas of the cloud key it belongs to when
// that is attached already, else contexts that hold this key alone (trgsw.IdentityKeySwitching takes nothing else).
func AttachedKSK(ksk []*tlwe.TLWELv0) *CloudKeySet {
	if len(ksk) == 0 {
		panic("tfhe_hip: empty key-switching key")
	}
	registry.Lock()
	defer registry.Unlock()
	for _, e := range registry.sets {
		if e.ksk == ksk[0] {
			return e.set
		}
	}
	s, ok := registry.kskOnly[ksk[0]]
	if !ok {
		devices := targetDevices()
		src := newContext(devices[0])
		src.LoadKSK(ksk)
		s = adoptSet(src, devices)
		registry.kskOnly[ksk[0]] = s
	}
	return s
}

// Scratch is a key-less context of the CURRENT parameter set on the first target GPU, for the operations that take their
// operands with the call (ExternalProductWith, CMuxWith, SampleExtract).  Re-made when the parameter set has been switched.
func Scratch() *CloudKey {
	registry.Lock()
	defer registry.Unlock()
	p := currentParams()
	if registry.scratch != nil && registry.scratchP != p {
		registry.scratch.Close()
		registry.scratch = nil
	}
	if registry.scratch == nil {
		registry.scratch = newContext(targetDevices()[0])
		registry.scratchP = p
	}
	return registry.scratch
}

// Detach releases the GPU replicas of a key.
func Detach(bsk []*trgsw.TRGSWLv1FFT) {
	if len(bsk) == 0 {
		return
	}
	registry.Lock()
	defer registry.Unlock()
	if e, ok := registry.sets[bsk[0]]; ok {
		e.set.Close()
		delete(registry.sets, bsk[0])
	}
}

// DetachKSK releases the contexts AttachedKSK made for a key-switching key on its own.
func DetachKSK(ksk []*tlwe.TLWELv0) {
	if len(ksk) == 0 {
		return
	}
	registry.Lock()
	defer registry.Unlock()
	if s, ok := registry.kskOnly[ksk[0]]; ok {
		s.Close()
		delete(registry.kskOnly, ksk[0])
	}
}
