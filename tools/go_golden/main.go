// go_golden -- dumps seeded inputs and outputs of the reference's own code on the gate-bootstrap path, so
// that the C oracle (oracle/) and the HIP engine can be pinned against go-tfhe ITSELF.
//
// The build image of this repository has no Go toolchain, so this file is shipped as source with a
// recipe and has not been compiled there.  On any machine with Go >= 1.21 and a checkout of
// github.com/thedonutfactory/go-tfhe:
//
//	cp -r tools/go_golden  <go-tfhe checkout>/cmd/go_golden
//	cd <go-tfhe checkout> && go run ./cmd/go_golden -out /tmp/go_golden
//	cp /tmp/go_golden/small/*.npy  <this repo>/tests/golden/go/          # ~0.6 MB, commit them
//	TFHE_GO_GOLDEN_BIG=/tmp/go_golden/big python -m pytest tests/test_go_golden.py      # full-key vectors
//
// Everything is written as NumPy .npy files (version 1.0, little endian), which tests/test_go_golden.py reads:
//
//	small/  (committable)
//	  params.npy                 int64 [8]: n, N, Nbit, L, Bgbit, basebit, t, decomposition offset
//	  extprod_trgsw.npy          float64 [2L][2][N]   bsk[0] in the reference FourierPoly layout (poly.go:57-62)
//	  extprod_in.npy             uint32 [2][N]        random TRLWE
//	  extprod_out.npy            uint32 [2][N]        Evaluator.ExternalProductAssign            (evaluator.go:50-81)
//	  cmux_trgsw.npy             float64 [K][2L][2][N] bsk[0..K)
//	  cmux_lwe.npy               uint32 [n+1]         the LWE sample whose first K mask words drive the chain
//	  cmux_acc.npy               uint32 [K+1][2][N]   accumulator after 0..K steps of BlindRotateAssign's loop (evaluator.go:110-130)
//	big/    (full 128-bit cloud key: 172 MB, not for the repository)
//	  key_lv0.npy, key_lv1.npy   uint32               the binary secret keys
//	  bsk_fourier.npy            float64 [n][2L][2][N] CloudKey.BootstrappingKey
//	  ksk.npy                    uint32 [N*t*base][n+1] CloudKey.KeySwitchingKey
//	  lwe_in.npy                 uint32 [B][n+1]      fresh encryptions of bits
//	  bits.npy                   uint8 [B]
//	  trlwe_acc.npy              uint32 [B][2][N]     BlindRotateAssign of lwe_in                    (evaluator.go:110-135)
//	  lwe_out.npy                uint32 [B][n+1]      BootstrapAssign of lwe_in                      (evaluator.go:139-148)
//	  gate_a.npy, gate_b.npy, gate_c.npy   uint32 [B][n+1]; gate_bits.npy uint8 [3][B]
//	  gate_<OP>.npy              uint32 [B][n+1]      gates.<OP>(a, b[, c], ck) for NAND AND OR XOR XNOR NOR ANDNY ANDYN ORNY ORYN MUX
//
// The programmable-bootstrap seam (BASELINE config 4; -uint5=true, the default), at params.SecurityUint5 after the
// 128-bit part (a fresh evaluator.NewEvaluator(N): gates.globalEval was sized at package init and is not touched again):
//
//	small/
//	  uint5_params.npy           int64 [9]: n, N, Nbit, L, Bgbit, basebit, t, decomposition offset, messageModulus (32)
//	  uint5_lut_identity.npy, uint5_lut_mod16.npy, uint5_lut_ge16.npy
//	                             uint32 [2][N]        lut.Generator.GenLookUpTableAssign (lut/generator.go:56-100) of x, x mod 16,
//	                                                  [x >= 16]: Poly.A then Poly.B -- exact integers: the oracle's generator and
//	                                                  go-tfhe_amd/lut.py must match them bit for bit
//	big/uint5/  (70 MB + 1.69 GB of keys: not for the repository)
//	  key_lv0.npy, key_lv1.npy, bsk_fourier.npy [n][2L][2][N], ksk.npy [N*t*base][n+1]
//	  pbs_msgs.npy               int64 [P]            the encrypted messages (tlwe/programmable_encrypt.go:12-26, modulus 32)
//	  pbs_lut.npy                int64 [P]            which table each item goes through: 0 identity, 1 mod 16, 2 >= 16
//	  pbs_in.npy                 uint32 [P][n+1]      EncryptLWEMessage(m, 32, alpha, KeyLv0)
//	  pbs_out.npy                uint32 [P][n+1]      Evaluator.BootstrapLUTAssign (evaluator/programmable_bootstrap.go:93-115)
//	  pbs_dec.npy                int64 [P]            DecryptLWEMessage(32, KeyLv0) of pbs_out
//	This parameter shape is in the fp64 TOLERANCE regime (values reach 2^58: two correct FFT pipelines differ in low bits and
//	digits flip downstream, SURVEY.md 8c(4)), so pbs_out is compared by decryption and phase distance, not word for word.
//
// Key generation in the reference draws from auto-seeded generators and fans out over goroutines, so the KEY is
// not reproducible from a seed; that is why the key itself is part of the dump.  Given the key, every function on
// the path is deterministic, which is all parity needs.
package main

import (
	"encoding/binary"
	"flag"
	"fmt"
	"math"
	"math/rand"
	"os"
	"path/filepath"
	"strings"

	"github.com/thedonutfactory/go-tfhe/cloudkey"
	"github.com/thedonutfactory/go-tfhe/evaluator"
	"github.com/thedonutfactory/go-tfhe/gates"
	"github.com/thedonutfactory/go-tfhe/key"
	"github.com/thedonutfactory/go-tfhe/lut"
	"github.com/thedonutfactory/go-tfhe/params"
	"github.com/thedonutfactory/go-tfhe/poly"
	"github.com/thedonutfactory/go-tfhe/tlwe"
	"github.com/thedonutfactory/go-tfhe/trgsw"
	"github.com/thedonutfactory/go-tfhe/trlwe"
)

func must(err error) {
	if err != nil {
		panic(err)
	}
}

// writeNpy writes a C-contiguous little-endian array: descr is "<u4", "<f8", "<i8" or "|u1".
func writeNpy(path string, descr string, shape []int, payload []byte) {
	dims := make([]string, len(shape))
	for i, d := range shape {
		dims[i] = fmt.Sprintf("%d", d)
	}
	shp := strings.Join(dims, ", ")
	if len(shape) == 1 {
		shp += ","
	}
	hdr := fmt.Sprintf("{'descr': '%s', 'fortran_order': False, 'shape': (%s), }", descr, shp)
	total := 10 + len(hdr) + 1
	pad := (64 - total%64) % 64
	hdr += strings.Repeat(" ", pad) + "\n"
	f, err := os.Create(path)
	must(err)
	defer f.Close()
	_, err = f.Write([]byte{0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0})
	must(err)
	must(binary.Write(f, binary.LittleEndian, uint16(len(hdr))))
	_, err = f.Write([]byte(hdr))
	must(err)
	_, err = f.Write(payload)
	must(err)
}

func u32Bytes(v []params.Torus) []byte {
	b := make([]byte, 4*len(v))
	for i, x := range v {
		binary.LittleEndian.PutUint32(b[4*i:], uint32(x))
	}
	return b
}

func f64Bytes(v []float64) []byte {
	b := make([]byte, 8*len(v))
	for i, x := range v {
		binary.LittleEndian.PutUint64(b[8*i:], math.Float64bits(x))
	}
	return b
}

func i64Bytes(v []int64) []byte {
	b := make([]byte, 8*len(v))
	for i, x := range v {
		binary.LittleEndian.PutUint64(b[8*i:], uint64(x))
	}
	return b
}

// one TRGSW in FFT form -> [2L][2][N] float64 (row r, part 0 = A / 1 = B), the layout tfhe_load_bsk_fourier takes
func flattenTRGSW(g *trgsw.TRGSWLv1FFT) []float64 {
	var out []float64
	for r := range g.TRLWEFFT {
		out = append(out, g.TRLWEFFT[r].A.Coeffs...)
		out = append(out, g.TRLWEFFT[r].B.Coeffs...)
	}
	return out
}

func flattenTRLWE(t *trlwe.TRLWELv1) []params.Torus {
	out := append([]params.Torus{}, t.A...)
	return append(out, t.B...)
}

func flattenLWEs(cts []*tlwe.TLWELv0) []params.Torus {
	var out []params.Torus
	for _, c := range cts {
		out = append(out, c.P...)
	}
	return out
}

type gate2 func(a, b *gates.Ciphertext, ck *cloudkey.CloudKey) *gates.Ciphertext

func main() {
	outDir := flag.String("out", "go_golden_out", "output directory")
	batch := flag.Int("batch", 8, "ciphertexts per full-key vector")
	steps := flag.Int("steps", 4, "CMUX steps in the small chain fixture")
	seed := flag.Int64("seed", 0x7F4E0020, "seed of the input generator (the key is dumped, not seeded)")
	uint5 := flag.Bool("uint5", true, "also dump the programmable-bootstrap seam at params.SecurityUint5 (1.8 GB of keys under big/uint5)")
	pbs := flag.Int("pbs", 8, "programmable bootstraps in the Uint5 vector")
	flag.Parse()
	small, big := filepath.Join(*outDir, "small"), filepath.Join(*outDir, "big")
	must(os.MkdirAll(small, 0o755))
	must(os.MkdirAll(big, 0o755))

	params.CurrentSecurityLevel = params.Security128Bit
	rand.Seed(*seed) // the reference seeds its local generators from the global source
	rng := rand.New(rand.NewSource(*seed + 1))

	lv0, g1 := params.GetTLWELv0(), params.GetTRGSWLv1()
	n, N, L := lv0.N, g1.N, g1.L
	base := 1 << g1.BASEBIT

	sk := key.NewSecretKey()
	ck := cloudkey.NewCloudKey(sk)
	eval := evaluator.NewEvaluator(N)
	writeNpy(filepath.Join(small, "params.npy"), "<i8", []int{8},
		i64Bytes([]int64{int64(n), int64(N), int64(g1.NBIT), int64(L), int64(g1.BGBIT), int64(g1.BASEBIT), int64(g1.IKS_T), int64(ck.DecompositionOffset)}))

	// ---- one external product (evaluator.go:50-81)
	in := trlwe.NewTRLWELv1()
	for i := 0; i < N; i++ {
		in.A[i] = params.Torus(rng.Uint32())
		in.B[i] = params.Torus(rng.Uint32())
	}
	out := trlwe.NewTRLWELv1()
	eval.ExternalProductAssign(ck.BootstrappingKey[0], in, ck.DecompositionOffset, out)
	writeNpy(filepath.Join(small, "extprod_trgsw.npy"), "<f8", []int{2 * L, 2, N}, f64Bytes(flattenTRGSW(ck.BootstrappingKey[0])))
	writeNpy(filepath.Join(small, "extprod_in.npy"), "<u4", []int{2, N}, u32Bytes(flattenTRLWE(in)))
	writeNpy(filepath.Join(small, "extprod_out.npy"), "<u4", []int{2, N}, u32Bytes(flattenTRLWE(out)))

	// ---- the first K iterations of BlindRotateAssign's loop, written out with the reference's own calls
	//      (evaluator.go:116-130: initial rotation by b~, then rotate by a~_i and CMuxAssign with bsk[i])
	K := *steps
	lwe := tlwe.NewTLWELv0()
	for i := range lwe.P {
		lwe.P[i] = params.Torus(rng.Uint32())
	}
	nBit := g1.NBIT
	acc1, acc2 := trlwe.NewTRLWELv1(), trlwe.NewTRLWELv1()
	bTilda := 2*N - ((int(lwe.B()) + (1 << (31 - nBit - 1))) >> (32 - nBit - 1))
	poly.PolyMulWithXKInPlace(ck.BlindRotateTestvec.A, bTilda, acc1.A)
	poly.PolyMulWithXKInPlace(ck.BlindRotateTestvec.B, bTilda, acc1.B)
	var chain []params.Torus
	var chainKey []float64
	chain = append(chain, flattenTRLWE(acc1)...)
	for i := 0; i < K; i++ {
		aTilda := int((lwe.P[i] + (1 << (31 - nBit - 1))) >> (32 - nBit - 1))
		poly.PolyMulWithXKInPlace(acc1.A, aTilda, acc2.A)
		poly.PolyMulWithXKInPlace(acc1.B, aTilda, acc2.B)
		eval.CMuxAssign(ck.BootstrappingKey[i], acc1, acc2, ck.DecompositionOffset, acc1)
		chain = append(chain, flattenTRLWE(acc1)...)
		chainKey = append(chainKey, flattenTRGSW(ck.BootstrappingKey[i])...)
	}
	writeNpy(filepath.Join(small, "cmux_trgsw.npy"), "<f8", []int{K, 2 * L, 2, N}, f64Bytes(chainKey))
	writeNpy(filepath.Join(small, "cmux_lwe.npy"), "<u4", []int{n + 1}, u32Bytes(lwe.P))
	writeNpy(filepath.Join(small, "cmux_acc.npy"), "<u4", []int{K + 1, 2, N}, u32Bytes(chain))

	// ---- full key + whole bootstraps and gates
	writeNpy(filepath.Join(big, "key_lv0.npy"), "<u4", []int{n}, u32Bytes(sk.KeyLv0))
	writeNpy(filepath.Join(big, "key_lv1.npy"), "<u4", []int{N}, u32Bytes(sk.KeyLv1))
	var bsk []float64
	for i := 0; i < n; i++ {
		bsk = append(bsk, flattenTRGSW(ck.BootstrappingKey[i])...)
	}
	writeNpy(filepath.Join(big, "bsk_fourier.npy"), "<f8", []int{n, 2 * L, 2, N}, f64Bytes(bsk))
	writeNpy(filepath.Join(big, "ksk.npy"), "<u4", []int{N * g1.IKS_T * base, n + 1}, u32Bytes(flattenLWEs(ck.KeySwitchingKey)))

	B := *batch
	bits := make([]byte, B)
	cts := make([]*tlwe.TLWELv0, B)
	var accs []params.Torus
	outs := make([]*tlwe.TLWELv0, B)
	for b := 0; b < B; b++ {
		bits[b] = byte(rng.Intn(2))
		cts[b] = tlwe.NewTLWELv0().EncryptBool(bits[b] == 1, lv0.ALPHA, sk.KeyLv0)
		acc := trlwe.NewTRLWELv1()
		eval.BlindRotateAssign(cts[b], ck.BlindRotateTestvec, ck.BootstrappingKey, ck.DecompositionOffset, acc)
		accs = append(accs, flattenTRLWE(acc)...)
		outs[b] = tlwe.NewTLWELv0()
		eval.BootstrapAssign(cts[b], ck.BlindRotateTestvec, ck.BootstrappingKey, ck.KeySwitchingKey, ck.DecompositionOffset, outs[b])
	}
	writeNpy(filepath.Join(big, "bits.npy"), "|u1", []int{B}, bits)
	writeNpy(filepath.Join(big, "lwe_in.npy"), "<u4", []int{B, n + 1}, u32Bytes(flattenLWEs(cts)))
	writeNpy(filepath.Join(big, "trlwe_acc.npy"), "<u4", []int{B, 2, N}, u32Bytes(accs))
	writeNpy(filepath.Join(big, "lwe_out.npy"), "<u4", []int{B, n + 1}, u32Bytes(flattenLWEs(outs)))

	gbits := make([]byte, 3*B)
	ga, gb, gc := make([]*tlwe.TLWELv0, B), make([]*tlwe.TLWELv0, B), make([]*tlwe.TLWELv0, B)
	for b := 0; b < B; b++ {
		for k, dst := range []*[]*tlwe.TLWELv0{&ga, &gb, &gc} {
			gbits[k*B+b] = byte(rng.Intn(2))
			(*dst)[b] = tlwe.NewTLWELv0().EncryptBool(gbits[k*B+b] == 1, lv0.ALPHA, sk.KeyLv0)
		}
	}
	writeNpy(filepath.Join(big, "gate_bits.npy"), "|u1", []int{3, B}, gbits)
	writeNpy(filepath.Join(big, "gate_a.npy"), "<u4", []int{B, n + 1}, u32Bytes(flattenLWEs(ga)))
	writeNpy(filepath.Join(big, "gate_b.npy"), "<u4", []int{B, n + 1}, u32Bytes(flattenLWEs(gb)))
	writeNpy(filepath.Join(big, "gate_c.npy"), "<u4", []int{B, n + 1}, u32Bytes(flattenLWEs(gc)))
	two := map[string]gate2{"NAND": gates.NAND, "AND": gates.AND, "OR": gates.OR, "XOR": gates.XOR, "XNOR": gates.XNOR,
		"NOR": gates.NOR, "ANDNY": gates.ANDNY, "ANDYN": gates.ANDYN, "ORNY": gates.ORNY, "ORYN": gates.ORYN}
	for name, fn := range two {
		res := make([]*tlwe.TLWELv0, B)
		for b := 0; b < B; b++ {
			r := fn(ga[b], gb[b], ck)
			res[b] = tlwe.NewTLWELv0()
			copy(res[b].P, r.P)
		}
		writeNpy(filepath.Join(big, "gate_"+name+".npy"), "<u4", []int{B, n + 1}, u32Bytes(flattenLWEs(res)))
	}
	mux := make([]*tlwe.TLWELv0, B)
	for b := 0; b < B; b++ {
		r := gates.MUX(ga[b], gb[b], gc[b], ck)
		mux[b] = tlwe.NewTLWELv0()
		copy(mux[b].P, r.P)
	}
	writeNpy(filepath.Join(big, "gate_MUX.npy"), "<u4", []int{B, n + 1}, u32Bytes(flattenLWEs(mux)))
	if *uint5 {
		dumpUint5(small, filepath.Join(big, "uint5"), *pbs, rng)
	}
	fmt.Println("wrote", small, "and", big)
}

// dumpUint5 is params/uint_params_test.go:46-71 with everything written out: the three lookup tables of the reference's
// nibble adder (examples/add_two_numbers/main.go:59-72) and P programmable bootstraps through them.
func dumpUint5(small, bigDir string, P int, rng *rand.Rand) {
	must(os.MkdirAll(bigDir, 0o755))
	params.CurrentSecurityLevel = params.SecurityUint5
	const messageModulus = 32
	lv0, g1 := params.GetTLWELv0(), params.GetTRGSWLv1()
	n, N, L := lv0.N, g1.N, g1.L
	base := 1 << g1.BASEBIT

	sk := key.NewSecretKey()
	ck := cloudkey.NewCloudKey(sk)
	eval := evaluator.NewEvaluator(N)
	writeNpy(filepath.Join(small, "uint5_params.npy"), "<i8", []int{9},
		i64Bytes([]int64{int64(n), int64(N), int64(g1.NBIT), int64(L), int64(g1.BGBIT), int64(g1.BASEBIT), int64(g1.IKS_T), int64(ck.DecompositionOffset), messageModulus}))

	gen := lut.NewGenerator(messageModulus)
	funcs := []func(int) int{
		func(x int) int { return x },
		func(x int) int { return x % 16 },
		func(x int) int {
			if x >= 16 {
				return 1
			}
			return 0
		},
	}
	names := []string{"identity", "mod16", "ge16"}
	tables := make([]*lut.LookUpTable, len(funcs))
	for i, f := range funcs {
		tables[i] = lut.NewLookUpTable()
		gen.GenLookUpTableAssign(f, tables[i])
		writeNpy(filepath.Join(small, "uint5_lut_"+names[i]+".npy"), "<u4", []int{2, N}, u32Bytes(flattenTRLWE(tables[i].Poly)))
	}

	writeNpy(filepath.Join(bigDir, "key_lv0.npy"), "<u4", []int{n}, u32Bytes(sk.KeyLv0))
	writeNpy(filepath.Join(bigDir, "key_lv1.npy"), "<u4", []int{N}, u32Bytes(sk.KeyLv1))
	var bsk []float64
	for i := 0; i < n; i++ {
		bsk = append(bsk, flattenTRGSW(ck.BootstrappingKey[i])...)
	}
	writeNpy(filepath.Join(bigDir, "bsk_fourier.npy"), "<f8", []int{n, 2 * L, 2, N}, f64Bytes(bsk))
	writeNpy(filepath.Join(bigDir, "ksk.npy"), "<u4", []int{N * g1.IKS_T * base, n + 1}, u32Bytes(flattenLWEs(ck.KeySwitchingKey)))

	msgs := make([]int64, P)
	which := make([]int64, P)
	dec := make([]int64, P)
	ins := make([]*tlwe.TLWELv0, P)
	outs := make([]*tlwe.TLWELv0, P)
	for i := 0; i < P; i++ {
		m := rng.Intn(messageModulus)
		w := i % len(tables)
		msgs[i] = int64(m)
		which[i] = int64(w)
		ins[i] = tlwe.NewTLWELv0()
		ins[i].EncryptLWEMessage(m, messageModulus, lv0.ALPHA, sk.KeyLv0)
		outs[i] = tlwe.NewTLWELv0()
		eval.BootstrapLUTAssign(ins[i], tables[w], ck.BootstrappingKey, ck.KeySwitchingKey, ck.DecompositionOffset, outs[i])
		dec[i] = int64(outs[i].DecryptLWEMessage(messageModulus, sk.KeyLv0))
	}
	writeNpy(filepath.Join(bigDir, "pbs_msgs.npy"), "<i8", []int{P}, i64Bytes(msgs))
	writeNpy(filepath.Join(bigDir, "pbs_lut.npy"), "<i8", []int{P}, i64Bytes(which))
	writeNpy(filepath.Join(bigDir, "pbs_in.npy"), "<u4", []int{P, n + 1}, u32Bytes(flattenLWEs(ins)))
	writeNpy(filepath.Join(bigDir, "pbs_out.npy"), "<u4", []int{P, n + 1}, u32Bytes(flattenLWEs(outs)))
	writeNpy(filepath.Join(bigDir, "pbs_dec.npy"), "<i8", []int{P}, i64Bytes(dec))
}
