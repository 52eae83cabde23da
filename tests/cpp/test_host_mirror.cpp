// C++ host-mirror test (runs on the GPU box): the reference's own gate tests re-expressed
// against tfhe::gates / tfhe::evaluator (gates/gates_test.go:23-366, 369-480), with the CPU
// oracle (test infrastructure) as key generator, encryptor/decryptor and bit-exact checker.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../go-tfhe_amd/host/tfhe_gpu.hpp"
#include "../../oracle/tfhe_oracle.h"

using namespace tfhe;

static int failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { failures++; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

int main()
{
    orc_params op;
    orc_get_params(0, &op);                                   // 80-bit set (via explicit params)
    params::Params p = params::Security80Bit();
    orc_rng rng;
    orc_rng_seed(&rng, 0x7F4E0021ull);
    std::vector<uint32_t> s0(op.n), s1(op.N);
    orc_keygen_secret(&op, &rng, s0.data(), s1.data());
    const size_t bsk_len = (size_t)op.n * 2 * op.L * 2 * op.N;
    std::vector<double> bsk(bsk_len);
    orc_keygen_bsk(&op, &rng, s0.data(), s1.data(), nullptr, bsk.data());
    std::vector<uint32_t> ksk((size_t)op.N * op.t * (1 << op.basebit) * (op.n + 1));
    orc_keygen_ksk(&op, &rng, s0.data(), s1.data(), ksk.data());
    orc_fft *fft = orc_fft_new(op.N);

    cloudkey::CloudKey ck(p, bsk.data(), ksk.data(), 0);
    auto enc = [&](int bit) { gates::Ciphertext c(op.n); orc_tlwe_encrypt_bool(&op, &rng, bit, s0.data(), c.P.data()); return c; };
    auto dec = [&](const gates::Ciphertext &c) { return orc_tlwe_decrypt_bool(&op, s0.data(), c.P.data()) != 0; };

    struct G { const char *name; gates::Ciphertext (*f)(const gates::Ciphertext &, const gates::Ciphertext &, const cloudkey::CloudKey &); int op; bool (*t)(bool, bool); };
    const G table[] = {
        {"NAND", gates::NAND, ORC_NAND, [](bool a, bool b) { return !(a && b); }},
        {"AND", gates::AND, ORC_AND, [](bool a, bool b) { return a && b; }},
        {"OR", gates::OR, ORC_OR, [](bool a, bool b) { return a || b; }},
        {"XOR", gates::XOR, ORC_XOR, [](bool a, bool b) { return a != b; }},
        {"XNOR", gates::XNOR, ORC_XNOR, [](bool a, bool b) { return a == b; }},
        {"NOR", gates::NOR, ORC_NOR, [](bool a, bool b) { return !(a || b); }},
        {"ANDNY", gates::ANDNY, ORC_ANDNY, [](bool a, bool b) { return !a && b; }},
        {"ANDYN", gates::ANDYN, ORC_ANDYN, [](bool a, bool b) { return a && !b; }},
        {"ORNY", gates::ORNY, ORC_ORNY, [](bool a, bool b) { return !a || b; }},
        {"ORYN", gates::ORYN, ORC_ORYN, [](bool a, bool b) { return a || !b; }},
    };
    for (const G &g : table)
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) {
                auto ca = enc(a), cb = enc(b);
                auto out = g.f(ca, cb, ck);
                EXPECT(dec(out) == g.t(a, b), "%s(%d,%d) decrypts wrong", g.name, a, b);
                std::vector<uint32_t> want(op.n + 1);
                orc_gate(&op, fft, bsk.data(), ksk.data(), g.op, ca.P.data(), cb.P.data(), nullptr, want.data());
                EXPECT(out.P == want, "%s(%d,%d) differs from the oracle", g.name, a, b);
            }
    // MUX, NOT, Copy, Constant (gates_test.go:273-366)
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++)
            for (int c = 0; c < 2; c++) {
                auto out = gates::MUX(enc(a), enc(b), enc(c), ck);
                EXPECT(dec(out) == (a ? b : c), "MUX(%d,%d,%d)", a, b, c);
            }
    EXPECT(dec(gates::NOT(enc(1))) == false && dec(gates::NOT(enc(0))) == true, "NOT");
    EXPECT(dec(gates::Copy(enc(1))) == true, "Copy");
    EXPECT(dec(gates::Constant(true, p)) == true && dec(gates::Constant(false, p)) == false, "Constant");
    // Batch AND/OR/XOR with 4 inputs (gates_test.go:369-480) + NAND/NOR/XNOR (untested upstream)
    gates::Pairs pairs;
    const int A[4] = {0, 0, 1, 1}, B[4] = {0, 1, 0, 1};
    for (int i = 0; i < 4; i++) pairs.push_back({enc(A[i]), enc(B[i])});
    auto r_and = gates::BatchAND(pairs, ck), r_or = gates::BatchOR(pairs, ck), r_xor = gates::BatchXOR(pairs, ck);
    auto r_nand = gates::BatchNAND(pairs, ck), r_nor = gates::BatchNOR(pairs, ck), r_xnor = gates::BatchXNOR(pairs, ck);
    for (int i = 0; i < 4; i++) {
        EXPECT(dec(r_and[i]) == (A[i] && B[i]), "BatchAND %d", i);
        EXPECT(dec(r_or[i]) == (A[i] || B[i]), "BatchOR %d", i);
        EXPECT(dec(r_xor[i]) == (A[i] != B[i]), "BatchXOR %d", i);
        EXPECT(dec(r_nand[i]) == !(A[i] && B[i]), "BatchNAND %d", i);
        EXPECT(dec(r_nor[i]) == !(A[i] || B[i]), "BatchNOR %d", i);
        EXPECT(dec(r_xnor[i]) == (A[i] == B[i]), "BatchXNOR %d", i);
    }
    // one cloud key on several devices from ONE process (here: two contexts on the one GPU of the box): the key
    // is replicated GPU to GPU behind the C ABI (tfhe_ctx_clone_to; on one GPU the peer copy degenerates to a device-to-device
    // copy: clone path 1), the batch is sharded contiguously over one thread per replica, and every output word equals the
    // single-context result
    {
        cloudkey::CloudKeySet set(ck, {0, 0});
        EXPECT(set.size() == 2 && cloudkey::CloudKeySet::AllDevices().size() >= 1, "CloudKeySet");
        EXPECT(set.ClonePath(0) == 1 && set.ClonePath(1) == 1, "clones onto the source's own GPU must take the device-to-device path");
        {
            // a clone is a full, independent context: its scalar gate equals the source's, word for word, and it outlives nothing
            auto solo = ck.CloneTo(0);
            auto x = enc(1), y = enc(0);
            EXPECT(gates::NAND(x, y, *solo).P == gates::NAND(x, y, ck).P, "a clone's NAND differs from its source's");
            bool bad = false;
            try { (void)ck.CloneTo(4096); } catch (const Panic &) { bad = true; }
            EXPECT(bad, "cloning onto a device that does not exist must panic");
        }
        gates::Pairs seven;
        std::vector<std::array<gates::Ciphertext, 3>> mux7;
        for (int i = 0; i < 7; i++) {                        // ragged: shards of 3 and 4
            seven.push_back({enc(i & 1), enc((i >> 1) & 1)});
            mux7.push_back({enc(i & 1), enc((i >> 1) & 1), enc((i >> 2) & 1)});
        }
        auto one = gates::BatchXOR(seven, ck), two = gates::BatchOnSet(TFHE_OP_XOR, seven, set);
        bool same = one.size() == two.size();
        for (size_t i = 0; same && i < one.size(); i++) same = one[i].P == two[i].P;
        EXPECT(same, "BatchOnSet(XOR) differs from the single-context batch");
        auto m = gates::BatchMUXOnSet(mux7, set);
        for (int i = 0; i < 7; i++) {
            EXPECT(dec(m[i]) == ((i & 1) ? ((i >> 1) & 1) : ((i >> 2) & 1)), "BatchMUXOnSet %d", i);
            EXPECT(m[i].P == gates::MUX(mux7[i][0], mux7[i][1], mux7[i][2], ck).P, "BatchMUXOnSet %d differs from scalar MUX", i);
        }
        bool threw = false;                                  // a blob of another parameter set is refused
        try { auto other = cloudkey::CloudKey::Empty(params::Security128Bit()); other->Import(0, ck.Export(0)); } catch (const Panic &) { threw = true; }
        EXPECT(threw, "importing an 80-bit key blob into a 128-bit context must panic");
    }
    // Prepare + Bootstrap seam (BASELINE config 1 goes through this)
    evaluator::Evaluator ev(ck);
    auto ca = enc(1), cb = enc(1);
    EXPECT(dec(ev.Bootstrap(ev.PrepareNAND(ca, cb))) == false, "PrepareNAND + Bootstrap");
    EXPECT(dec(ev.Bootstrap(ev.PrepareXOR(ca, cb))) == false, "PrepareXOR + Bootstrap");
    // BlindRotateAssign + test-vector variant are bit-exact vs the oracle
    trlwe::TRLWELv1 acc(op.N);
    auto prep = ev.PrepareAND(ca, cb);
    ev.BlindRotateAssign(prep, nullptr, acc);
    std::vector<uint32_t> tv(2 * op.N), want(2 * op.N);
    orc_gate_testvec(&op, tv.data());
    orc_blind_rotate(&op, fft, bsk.data(), prep.P.data(), tv.data(), -1, want.data());
    EXPECT(std::equal(acc.A.begin(), acc.A.end(), want.begin()) && std::equal(acc.B.begin(), acc.B.end(), want.begin() + op.N), "BlindRotateAssign");
    // trgsw / trlwe seams with caller-supplied operands (SURVEY 8b seam 3): one blind rotation rebuilt step by step OUTSIDE the engine's
    // persistent kernel -- X^a rotation by the oracle, CMuxAssign with bsk[i] handed over as a free-standing TRGSWLv1FFT -- must end
    // in the same words as BlindRotateAssign; then SampleExtractIndex + IdentityKeySwitching == the fused bootstrap output
    {
        const uint32_t off = orc_decomposition_offset(&op);
        const size_t gsw = (size_t)2 * op.L * 2 * op.N;
        auto accs = trgsw::BatchBlindRotate({prep}, [&] { trlwe::TRLWELv1 t(op.N); std::copy(tv.begin(), tv.begin() + op.N, t.A.begin()); std::copy(tv.begin() + op.N, tv.end(), t.B.begin()); return t; }(), off, ck);
        EXPECT(accs.size() == 1 && accs[0].A == acc.A && accs[0].B == acc.B, "trgsw::BatchBlindRotate differs from BlindRotateAssign");
        // first 3 CMUX steps by hand
        std::vector<uint32_t> cur(2 * op.N), rot(2 * op.N), wantp(2 * op.N);
        orc_blind_rotate(&op, fft, bsk.data(), prep.P.data(), tv.data(), 0, cur.data());
        for (int i = 0; i < 3; i++) {
            const int at = (int)((uint32_t)(prep.P[i] + (1u << (31 - op.Nbit - 1))) >> (32 - op.Nbit - 1));
            orc_poly_mul_xk(op.N, cur.data(), at, rot.data());
            orc_poly_mul_xk(op.N, cur.data() + op.N, at, rot.data() + op.N);
            trgsw::TRGSWLv1FFT g(bsk.data() + (size_t)i * gsw, p);
            trlwe::TRLWELv1 c0 = trgsw::detail_t::unflat(cur.data(), op.N), c1 = trgsw::detail_t::unflat(rot.data(), op.N), out(op.N);
            ev.CMuxAssign(g, c0, c1, off, out);
            std::copy(out.A.begin(), out.A.end(), cur.begin());
            std::copy(out.B.begin(), out.B.end(), cur.begin() + op.N);
        }
        orc_blind_rotate(&op, fft, bsk.data(), prep.P.data(), tv.data(), 3, wantp.data());
        EXPECT(cur == wantp, "three CMuxAssign steps with free-standing TRGSW operands differ from the oracle's blind-rotate prefix");
        trlwe::TRLWELv1 prod(op.N);
        trgsw::TRGSWLv1FFT g5(bsk.data() + 5 * gsw, p);
        ev.ExternalProductAssign(g5, acc, off, prod);
        std::vector<uint32_t> accf = trgsw::detail_t::flat(acc), wante(2 * op.N);
        orc_external_product(&op, fft, bsk.data() + 5 * gsw, accf.data(), wante.data());
        EXPECT(trgsw::detail_t::flat(prod) == wante, "ExternalProductAssign with a free-standing operand differs from the oracle");
        for (int k : {0, 1, 700}) {
            auto ext = trlwe::SampleExtractIndex(acc, k, ck);
            std::vector<uint32_t> wx(op.N + 1);
            orc_sample_extract(op.N, accf.data(), k, wx.data());
            EXPECT(ext.P == wx, "SampleExtractIndex(%d) differs from the oracle", k);
        }
        auto lv0 = trgsw::IdentityKeySwitching(trlwe::SampleExtractIndex(acc, 0, ck), ck);
        EXPECT(lv0.P == ev.Bootstrap(prep).P, "SampleExtractIndex + IdentityKeySwitching differs from the fused bootstrap");
        bool bad = false;
        try { (void)trgsw::BlindRotate(prep, trlwe::TRLWELv1(op.N), off + 1, ck); } catch (const Panic &e) { bad = e.code == TFHE_E_INVALID; }
        EXPECT(bad, "a decomposition offset other than the cloud key's must panic in BlindRotate");
    }
    // lut package + BootstrapFunc (lut_test.go:10-215, programmable_bootstrap_test.go:13-188)
    {
        lut::Encoder e4(4);
        for (int i = 0; i < 4; i++) EXPECT(e4.Decode(e4.Encode(i)) == i, "Encoder(4) round trip %d", i);
        EXPECT(e4.Encode(-1) == e4.Encode(3) && e4.Encode(4) == e4.Encode(0), "Encoder wrap-around");
        EXPECT(lut::F64ToTorus(-0.125) == 0xE0000000u && lut::F64ToTorus(0.5) == 0x80000000u, "F64ToTorus known answers");
        const int moduli[] = {2, 3, 4, 8};
        for (int m : moduli) {
            std::vector<int32_t> tab(m);
            for (int x = 0; x < m; x++) tab[x] = (3 * x + 1) % m;
            auto t = lut::Generator(p, m).GenLookUpTable([&](int x) { return tab[x]; });
            std::vector<uint32_t> ref(2 * op.N);
            orc_lut_generate(&op, tab.data(), m, ref.data());
            EXPECT(std::equal(t.Poly.A.begin(), t.Poly.A.end(), ref.begin()) && std::equal(t.Poly.B.begin(), t.Poly.B.end(), ref.begin() + op.N),
                   "Generator(%d) differs from the oracle table", m);
        }
        lut::Generator g2(p, 2);
        EXPECT(g2.ModSwitch(0) == 0 && g2.ModSwitch(1u << 30) == op.N / 4 && g2.ModSwitch(1u << 31) == op.N / 2 && g2.ModSwitch(0xFFFFFFFFu) == 0, "ModSwitch");
        int (*fs[3])(int) = {[](int x) { return x; }, [](int x) { return 1 - x; }, [](int) { return 1; }};
        for (auto f : fs)
            for (int m = 0; m < 2; m++) {
                gates::Ciphertext ct(op.n);
                orc_tlwe_encrypt_message(&op, &rng, m, 2, s0.data(), ct.P.data());
                auto out = ev.BootstrapFunc(ct, f, 2);
                EXPECT(orc_tlwe_decrypt_message(&op, 2, s0.data(), out.P.data()) == f(m), "BootstrapFunc decrypts wrong (m=%d)", m);
                const int32_t tab[2] = {f(0), f(1)};
                std::vector<uint32_t> tvf(2 * op.N), wantf(op.n + 1);
                orc_lut_generate(&op, tab, 2, tvf.data());
                orc_bootstrap(&op, fft, bsk.data(), ksk.data(), ct.P.data(), tvf.data(), wantf.data());
                EXPECT(out.P == wantf, "BootstrapFunc differs from the oracle (m=%d)", m);
            }
    }
    // extended tables (polyExtendFactor > 1) need the N = 2048 shape: a Uint5-ring context with a short LWE dimension,
    // key generated on the GPU from OS entropy (seed128 = nullptr); Uint6 = modulus 64 over a 4096-entry table
    {
        orc_params o5;
        orc_get_params(3, &o5);
        o5.n = 24;
        params::Params p5{o5.n, o5.N, o5.Nbit, o5.L, o5.Bgbit, o5.basebit, o5.t};
        std::vector<uint32_t> k0(o5.n), k1(o5.N);
        orc_keygen_secret(&o5, &rng, k0.data(), k1.data());
        auto ck5 = cloudkey::CloudKey::NewCloudKey(p5, k0, k1, o5.alpha_lv0, o5.alpha_lv1);
        evaluator::Evaluator ev5(*ck5);
        lut::Generator g6(p5, 64, 2);
        EXPECT(g6.LookUpTableSize == 4096, "extended generator size");
        const auto t6 = g6.GenLookUpTableExtended([](int x) { return 63 - x; });
        EXPECT(t6.size() == (size_t)2 * 2 * 2048, "extended table shape");
        for (int m : {0, 1, 31, 32, 62, 63}) {
            gates::Ciphertext ct(o5.n);
            orc_tlwe_encrypt_message(&o5, &rng, m, 64, k0.data(), ct.P.data());
            auto out = ev5.BootstrapLUTExtended(ct, t6);
            EXPECT(orc_tlwe_decrypt_message(&o5, 64, k0.data(), out.P.data()) == 63 - m, "BootstrapLUTExtended decrypts wrong (m=%d)", m);
        }
        bool threw6 = false;
        try { g6.GenLookUpTable([](int x) { return x; }); } catch (const Panic &) { threw6 = true; }
        EXPECT(threw6, "an extended generator must refuse to build an N-coefficient table");
        // the cloud key through its serialised form into a second context: identical ciphertexts; a blob of the wrong kind,
        // a truncated one and one offered to another parameter set are refused
        auto ck5b = cloudkey::CloudKey::Empty(p5);
        const auto blob0 = ck5->Export(0), blob1 = ck5->Export(1);
        ck5b->Import(0, blob0);
        ck5b->Import(1, blob1);
        evaluator::Evaluator ev5b(*ck5b);
        gates::Ciphertext ct(o5.n);
        orc_tlwe_encrypt_message(&o5, &rng, 17, 64, k0.data(), ct.P.data());
        EXPECT(ev5.BootstrapLUTExtended(ct, t6).P == ev5b.BootstrapLUTExtended(ct, t6).P, "imported key computes different ciphertexts");
        int refused = 0;
        try { ck5b->Import(0, blob1); } catch (const Panic &e) { refused += e.code == TFHE_E_INVALID; }
        try { ck5b->Import(1, std::vector<uint8_t>(blob1.begin(), blob1.end() - 4)); } catch (const Panic &e) { refused += e.code == TFHE_E_INVALID; }
        params::Params p5c = p5;
        p5c.n = 25;
        try { cloudkey::CloudKey::Empty(p5c)->Import(0, blob0); } catch (const Panic &e) { refused += e.code == TFHE_E_INVALID; }
        EXPECT(refused == 3, "mismatched key blobs must be refused (%d of 3 were)", refused);
    }
    // error behaviour: a Go panic is a thrown Panic
    bool threw = false;
    try { gates::Ciphertext bad(3); gates::NAND(bad, bad, ck); } catch (const Panic &) { threw = true; }
    EXPECT(threw, "wrong-length ciphertext must panic");
    threw = false;
    try { params::Params q = p; q.N = 512; cloudkey::CloudKey bad(q, nullptr, nullptr, 0); } catch (const Panic &e) { threw = e.code == TFHE_E_INVALID; }
    EXPECT(threw, "unsupported parameter shape must panic");
    orc_fft_free(fft);
    std::printf(failures ? "host mirror: %d FAILURES\n" : "host mirror: all checks passed\n", failures);
    return failures ? 1 : 0;
}
