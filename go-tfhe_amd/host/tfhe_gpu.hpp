// tfhe_gpu.hpp -- C++ host side above the C ABI, mirroring go-tfhe's operator packages.
//
// The reference is Go; this image has no Go toolchain, so the host layer a Go user would call
// is provided in C++ with the same package / function names, argument meaning and error
// behaviour (a Go panic becomes a thrown tfhe::Panic).  Namespaces = Go packages:
//
//   params::     parameter sets                     (params/params.go:83-112,117-146,151-180,362-391)
//   tlwe::       TLWELv0 sample + linear ops        (tlwe/tlwe.go:11-33,76-134)
//   trlwe::      TRLWELv1 sample, SampleExtractIndex[Assign]   (trlwe/trlwe.go:13-25,114-131, trlwe_ops.go:10)
//   trgsw::      TRGSWLv1FFT operand, ExternalProductWithFFT, CMUX, [Batch]BlindRotate,
//                IdentityKeySwitching[Assign]        (trgsw/trgsw.go:108-252,285, keyswitch.go:10)
//   cloudkey::   CloudKey resident on one GPU       (cloudkey/cloudkey.go:16-31)
//   lut::        LookUpTable, Encoder, Generator    (lut/lut.go:13-45, encoder.go:10-107, generator.go:10-173)
//   evaluator::  Evaluator {ExternalProductAssign, BlindRotateAssign, BootstrapAssign, Bootstrap,
//                BootstrapLUT[Assign], BootstrapFunc[Assign], Prepare*}
//                                                   (evaluator/evaluator.go:50-157, gates_helper.go:10-63,
//                                                    programmable_bootstrap.go:16-115)
//   gates::      NAND ... MUX, NOT, Copy, Constant, Batch*   (gates/gates.go:26-126,156-312)
//
// Header-only; link with -ltfhe_hip.  Everything that computes goes through include/tfhe_hip.h.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <thread>
#include <vector>

#include "../../include/tfhe_hip.h"

namespace tfhe {

struct Panic : std::runtime_error {
    int code;
    Panic(int c, const std::string &m) : std::runtime_error("tfhe_hip: " + m), code(c) {}
};

inline void check(int rc)
{
    if (rc != TFHE_OK) throw Panic(rc, tfhe_last_error());
}

namespace params {
using Torus = uint32_t;                                   // params.go:27
using Params = tfhe_params;
inline Params Security80Bit() { return {550, 1024, 10, 3, 6, 2, 7}; }
inline Params Security110Bit() { return {630, 1024, 10, 3, 6, 2, 8}; }
inline Params Security128Bit() { return {700, 1024, 10, 3, 6, 2, 9}; }
inline Params SecurityUint1() { return {700, 1024, 10, 2, 10, 2, 8}; }     // params.go:194-232
inline Params SecurityUint2() { return {687, 512, 9, 1, 18, 4, 3}; }       // params.go:236-265
inline Params SecurityUint3() { return {820, 1024, 10, 1, 23, 6, 2}; }     // params.go:277-313
inline Params SecurityUint4() { return {820, 2048, 11, 1, 22, 5, 3}; }     // params.go:318-354
inline Params SecurityUint5() { return {1071, 2048, 11, 1, 22, 6, 3}; }    // params.go:362-398
inline Params SecurityUint6() { return {1071, 2048, 11, 1, 22, 6, 3}; }    // params.go:403-439
inline Params SecurityUint7() { return {1160, 2048, 11, 1, 22, 7, 3}; }    // params.go:444-480
inline Params SecurityUint8() { return {1160, 2048, 11, 1, 22, 7, 3}; }    // params.go:485-521
} // namespace params

namespace tlwe {
// tlwe.go:11-33 : n+1 words, body last.
struct TLWELv0 {
    std::vector<params::Torus> P;
    TLWELv0() = default;
    explicit TLWELv0(int n) : P((size_t)n + 1, 0u) {}
    params::Torus B() const { return P.back(); }
    void SetB(params::Torus v) { P.back() = v; }
    // tlwe.go:76-134
    TLWELv0 Add(const TLWELv0 &o) const { TLWELv0 r = *this; for (size_t i = 0; i < P.size(); i++) r.P[i] = P[i] + o.P[i]; return r; }
    TLWELv0 Sub(const TLWELv0 &o) const { TLWELv0 r = *this; for (size_t i = 0; i < P.size(); i++) r.P[i] = P[i] - o.P[i]; return r; }
    TLWELv0 Neg() const { TLWELv0 r = *this; for (auto &x : r.P) x = 0u - x; return r; }
    TLWELv0 AddMul(const TLWELv0 &o, params::Torus m) const { TLWELv0 r = *this; for (size_t i = 0; i < P.size(); i++) r.P[i] = P[i] + o.P[i] * m; return r; }
    TLWELv0 SubMul(const TLWELv0 &o, params::Torus m) const { TLWELv0 r = *this; for (size_t i = 0; i < P.size(); i++) r.P[i] = P[i] - o.P[i] * m; return r; }
};
// tlwe.go:36-73: a level-1 sample, N mask words and the body last (what SampleExtractIndex produces, IdentityKeySwitching consumes)
struct TLWELv1 {
    std::vector<params::Torus> P;
    TLWELv1() = default;
    explicit TLWELv1(int N) : P((size_t)N + 1, 0u) {}
    params::Torus B() const { return P.back(); }
    void SetB(params::Torus v) { P.back() = v; }
};
} // namespace tlwe

namespace trlwe {
// trlwe.go:13-16
struct TRLWELv1 {
    std::vector<params::Torus> A, B;
    TRLWELv1() = default;
    explicit TRLWELv1(int N) : A((size_t)N, 0u), B((size_t)N, 0u) {}
};
} // namespace trlwe

namespace cloudkey {
// cloudkey.go:16-21.  DecompositionOffset and BlindRotateTestvec are derived from the
// parameters inside the context; the two keys are uploaded once from flat arrays.
class CloudKey {
  public:
    // bsk_fourier: [n][2L][2][N] float64 in the reference FourierPoly layout; ksk: [N*t*base][n+1].
    CloudKey(const params::Params &p, const double *bsk_fourier, const uint32_t *ksk, int device = 0) : P(p)
    {
        check(tfhe_ctx_create(&P, device, &ctx_));
        try {
            if (bsk_fourier) check(tfhe_load_bsk_fourier(ctx_, bsk_fourier));
            if (ksk) check(tfhe_load_ksk(ctx_, ksk));
        } catch (...) {          // a throwing constructor runs no destructor: release the context here
            tfhe_ctx_destroy(ctx_);
            ctx_ = nullptr;
            throw;
        }
    }
    // coefficient-domain bootstrapping key (trgsw.TRGSWLv1), [n][2L][2][N] uint32
    static std::unique_ptr<CloudKey> FromTorus(const params::Params &p, const uint32_t *bsk_torus, const uint32_t *ksk, int device = 0)
    {
        auto ck = std::make_unique<CloudKey>(p, nullptr, ksk, device);
        check(tfhe_load_bsk_torus(ck->ctx_, bsk_torus));
        return ck;
    }
    // cloudkey.NewCloudKey(secretKey) (cloudkey.go:24-31), generated on the GPU.  seed128 = nullptr (the default)
    // draws 128 bits from the OS entropy source, like the reference's auto-seeded generator; a fixed seed is for
    // tests only: the seed is secret key material (include/tfhe_hip.h).
    static std::unique_ptr<CloudKey> NewCloudKey(const params::Params &p, const std::vector<uint32_t> &keyLv0,
                                                 const std::vector<uint32_t> &keyLv1, double alphaLv0, double alphaLv1,
                                                 const uint64_t *seed128 = nullptr, int device = 0)
    {
        auto ck = std::make_unique<CloudKey>(p, nullptr, nullptr, device);
        if ((int)keyLv0.size() != p.n || (int)keyLv1.size() != p.N) throw Panic(TFHE_E_INVALID, "secret key has the wrong length");
        check(tfhe_keygen_cloud_seeded(ck->ctx_, keyLv0.data(), keyLv1.data(), alphaLv0, alphaLv1, seed128));
        return ck;
    }
    // The engine's serialised form of the two keys (the reference has none): which = 0 bootstrapping key, 1 key-switching
    // key; a blob carries a header (parameter set, key kind, device-layout version, length) that Import checks -- a blob
    // of another set / kind / library build, or a truncated one, is a Panic and installs nothing (include/tfhe_hip.h).
    std::vector<uint8_t> Export(int which) const
    {
        size_t bytes = 0;
        check(tfhe_key_size(ctx_, which, &bytes));
        std::vector<uint8_t> blob(bytes);
        check(tfhe_key_export(ctx_, which, blob.data()));
        return blob;
    }
    void Import(int which, const std::vector<uint8_t> &blob) { check(tfhe_key_import(ctx_, which, blob.data(), blob.size())); }
    // a replica of this key on another GPU (or a second context on the same one), copied GPU to GPU: tfhe_ctx_clone_to
    std::unique_ptr<CloudKey> CloneTo(int device) const
    {
        tfhe_ctx *c = nullptr;
        check(tfhe_ctx_clone_to(ctx_, device, &c));
        return std::unique_ptr<CloudKey>(new CloudKey(P, c));
    }
    // an empty context of the given parameters, to Import a cloud key into
    static std::unique_ptr<CloudKey> Empty(const params::Params &p, int device = 0) { return std::make_unique<CloudKey>(p, nullptr, nullptr, device); }
    ~CloudKey() { if (ctx_) tfhe_ctx_destroy(ctx_); }
    CloudKey(const CloudKey &) = delete;
    CloudKey &operator=(const CloudKey &) = delete;
    tfhe_ctx *ctx() const { return ctx_; }
    params::Params P;

  private:
    CloudKey(const params::Params &p, tfhe_ctx *adopted) : P(p), ctx_(adopted) {}      // CloneTo
    tfhe_ctx *ctx_ = nullptr;
};

// One cloud key on SEVERAL GPUs of a node, used from ONE process (a Go service: one goroutine per device instead of one
// process per GPU; the reference's own concurrency is one evaluator per goroutine, trgsw.go:227-252).  The key is replicated
// GPU to GPU behind the C ABI -- tfhe_ctx_clone_to: hipMemcpyPeerAsync between the device layouts, over xGMI between two GPUs,
// a device-to-device copy on one; no host copy of the 147 MB (round 4 went through Export -> host blob -> Import per device) --
// and batch calls shard contiguously, one thread per replica (gates::BatchOnSet below).  devices may repeat an index: two
// contexts on one GPU are two independent submitters (that is how tests/cpp exercises this on a one-GPU box).
class CloudKeySet {
  public:
    CloudKeySet(const CloudKey &src, const std::vector<int> &devices) : P(src.P)
    {
        if (devices.empty()) throw Panic(TFHE_E_INVALID, "CloudKeySet needs at least one device");
        for (int d : devices) replicas_.push_back(src.CloneTo(d));
    }
    // how replica i's keys arrived (TFHE_OPT_CLONE_PATH): 1 same GPU, 2 peer copy (xGMI), 3 host-staged (devices are not peers)
    int ClonePath(size_t i) const
    {
        int v = 0;
        check(tfhe_ctx_get_option(replicas_[i]->ctx(), TFHE_OPT_CLONE_PATH, &v));
        return v;
    }
    // every visible GPU once
    static std::vector<int> AllDevices()
    {
        int n = 0;
        check(tfhe_device_count(&n));
        std::vector<int> d(n);
        for (int i = 0; i < n; i++) d[i] = i;
        return d;
    }
    size_t size() const { return replicas_.size(); }
    const CloudKey &operator[](size_t i) const { return *replicas_[i]; }
    params::Params P;

  private:
    std::vector<std::unique_ptr<CloudKey>> replicas_;
};
} // namespace cloudkey

namespace detail {
inline std::vector<uint32_t> flatten(const std::vector<tlwe::TLWELv0> &v, size_t n1)
{
    std::vector<uint32_t> f(v.size() * n1);
    for (size_t i = 0; i < v.size(); i++) {
        if (v[i].P.size() != n1) throw Panic(TFHE_E_INVALID, "ciphertext has the wrong length");
        std::copy(v[i].P.begin(), v[i].P.end(), f.begin() + i * n1);
    }
    return f;
}
inline std::vector<tlwe::TLWELv0> unflatten(const std::vector<uint32_t> &f, size_t n1)
{
    std::vector<tlwe::TLWELv0> v(f.size() / n1);
    for (size_t i = 0; i < v.size(); i++) v[i].P.assign(f.begin() + i * n1, f.begin() + (i + 1) * n1);
    return v;
}
} // namespace detail

namespace lut {
// utils/utils.go:11-19
inline params::Torus F64ToTorus(double d) { return (params::Torus)(int64_t)(std::fmod(d, 1.0) * 4294967296.0); }
inline double TorusToF64(params::Torus t) { return (double)t / 4294967296.0; }

// lut.go:13-45 : a TRLWE whose B polynomial holds the table (A = 0 for generated tables)
struct LookUpTable {
    trlwe::TRLWELv1 Poly;
    explicit LookUpTable(int N) : Poly(N) {}
    LookUpTable Copy() const { return *this; }
    void CopyFrom(const LookUpTable &o) { Poly = o.Poly; }
    void Clear() { std::fill(Poly.A.begin(), Poly.A.end(), 0u); std::fill(Poly.B.begin(), Poly.B.end(), 0u); }
};

// encoder.go:10-107 : message i of a modulus-m space sits at i * Scale, Scale = 1/(2m) by default
struct Encoder {
    int MessageModulus;
    double Scale;
    explicit Encoder(int messageModulus) : MessageModulus(messageModulus), Scale(1.0 / (2.0 * messageModulus)) {}
    Encoder(int messageModulus, double scale) : MessageModulus(messageModulus), Scale(scale) {}
    int wrap(long m) const { m %= MessageModulus; return (int)(m < 0 ? m + MessageModulus : m); }
    params::Torus Encode(int message) const { return EncodeWithCustomScale(message, Scale); }
    params::Torus EncodeWithCustomScale(int message, double scale) const { return F64ToTorus((double)wrap(message) * scale); }
    int Decode(params::Torus v) const { return wrap((long)(TorusToF64(v) / Scale + 0.5)); }
    bool DecodeBool(params::Torus v) const { return Decode(v) != 0; }
};

// generator.go:10-173 for one parameter set.  polyExtendFactor > 1 gives the EXTENDED tables the Uint6/7/8 sets are
// specified for (LookUpTableSize = polyExtendFactor * N, params.go:399-402,440-443,481-484) and the reference does not
// implement (generator.go:19-20): GenLookUpTableExtended + Evaluator::BootstrapLUTExtended.
class Generator {
  public:
    Encoder Enc;
    int PolyDegree, PolyExtendFactor, LookUpTableSize;
    Generator(const params::Params &p, int messageModulus, int polyExtendFactor = 1)
        : Enc(messageModulus), PolyDegree(p.N), PolyExtendFactor(polyExtendFactor), LookUpTableSize(p.N * polyExtendFactor) {}
    Generator(const params::Params &p, int messageModulus, double scale)
        : Enc(messageModulus, scale), PolyDegree(p.N), PolyExtendFactor(1), LookUpTableSize(p.N) {}

    // The table of f over LookUpTableSize positions, de-interleaved: [ext][2][N] uint32, component k = coefficients of
    // Y^(i*ext + k) of the big polynomial (A parts zero) -- the layout tfhe_bootstrap_extended_batch takes.
    std::vector<params::Torus> GenLookUpTableExtended(const std::function<int(int)> &f) const
    {
        const std::vector<params::Torus> big = table(Enc.MessageModulus, [&](int x) { return Enc.Encode(f(x)); });
        const size_t N = (size_t)PolyDegree, ext = (size_t)PolyExtendFactor;
        std::vector<params::Torus> out(ext * 2 * N, 0u);
        for (size_t i = 0; i < N; i++)
            for (size_t k = 0; k < ext; k++) out[(k * 2 + 1) * N + i] = big[i * ext + k];
        return out;
    }

    LookUpTable GenLookUpTable(const std::function<int(int)> &f) const
    {
        return fill(Enc.MessageModulus, [&](int x) { return Enc.Encode(f(x)); });
    }
    void GenLookUpTableAssign(const std::function<int(int)> &f, LookUpTable &out) const { out = GenLookUpTable(f); }
    LookUpTable GenLookUpTableFull(const std::function<params::Torus(int)> &f) const { return fill(Enc.MessageModulus, f); }
    LookUpTable GenLookUpTableCustom(const std::function<int(int)> &f, int messageModulus, double scale) const
    {
        const Encoder e(messageModulus, scale);
        return fill(messageModulus, [&](int x) { return e.Encode(f(x)); });
    }
    // generator.go:159-168
    int ModSwitch(params::Torus x) const
    {
        const long r = std::lround((double)x / 4294967296.0 * LookUpTableSize) % LookUpTableSize;
        return (int)(r < 0 ? r + LookUpTableSize : r);
    }

  private:
    static long divRound(long a, long b) { return (a + b / 2) / b; }       // generator.go:171-173
    // Coefficient i of the table is the value of the message whose raw range contains (i + offset) mod N,
    // offset = divRound(N, 2m); the coefficients that wrapped around are negated  (generator.go:62-93).
    std::vector<params::Torus> table(int m, const std::function<params::Torus(int)> &value) const
    {
        const long N = LookUpTableSize, offset = divRound(N, 2L * m);
        std::vector<params::Torus> val((size_t)m), out((size_t)N);
        for (int x = 0; x < m; x++) val[(size_t)x] = value(x);
        int x = 0;
        for (long k = 0; k < N; k++) {                           // k walks the raw positions, in rotated order
            const long src = (k + offset) % N;
            if (src == 0) x = 0;
            while (x + 1 < m && src >= divRound((long)(x + 1) * N, m)) x++;
            const params::Torus v = val[(size_t)x];
            out[(size_t)k] = k >= N - offset ? 0u - v : v;
        }
        return out;
    }
    LookUpTable fill(int m, const std::function<params::Torus(int)> &value) const
    {
        if (PolyExtendFactor != 1) throw Panic(TFHE_E_INVALID, "a LookUpTable holds N coefficients: use GenLookUpTableExtended");
        LookUpTable out(PolyDegree);
        out.Poly.B = table(m, value);
        return out;
    }
};
} // namespace lut

namespace trgsw {
// trgsw.go:60-68: 2L TRLWE rows in the Fourier domain, kept flat: [2L][2][N] float64, row r = TRLWEFFT[r].A.Coeffs then .B.Coeffs in
// the reference's own FourierPoly layout (what one element of the flattened bootstrapping key is).
struct TRGSWLv1FFT {
    std::vector<double> Flat;
    TRGSWLv1FFT() = default;
    TRGSWLv1FFT(const double *src, const params::Params &p) : Flat(src, src + (size_t)2 * p.L * 2 * p.N) {}
};
namespace detail_t {
inline std::vector<uint32_t> flat(const trlwe::TRLWELv1 &t)
{
    std::vector<uint32_t> f(t.A);
    f.insert(f.end(), t.B.begin(), t.B.end());
    return f;
}
inline trlwe::TRLWELv1 unflat(const uint32_t *f, size_t N)
{
    trlwe::TRLWELv1 t;
    t.A.assign(f, f + N);
    t.B.assign(f + N, f + 2 * N);
    return t;
}
inline void want_offset(const cloudkey::CloudKey &ck, params::Torus off)
{
    uint32_t mine = 0;
    check(tfhe_ctx_decomposition_offset(ck.ctx(), &mine));
    if (mine != off) throw Panic(TFHE_E_INVALID, "decompositionOffset differs from the cloud key's (cloudkey.go:60-71): the blind rotation uses the context's");
}
} // namespace detail_t
// trgsw.go:108-137
inline trlwe::TRLWELv1 ExternalProductWithFFT(const TRGSWLv1FFT &g, const trlwe::TRLWELv1 &in, params::Torus decompositionOffset, const cloudkey::CloudKey &ck)
{
    const size_t N = (size_t)ck.P.N;
    std::vector<uint32_t> f = detail_t::flat(in), out(2 * N);
    check(tfhe_external_product_with(ck.ctx(), g.Flat.data(), decompositionOffset, f.data(), out.data(), 1));
    return detail_t::unflat(out.data(), N);
}
// trgsw.go:173-194: in1 where cond encrypts 0, in2 where it encrypts 1
inline trlwe::TRLWELv1 CMUX(const trlwe::TRLWELv1 &in1, const trlwe::TRLWELv1 &in2, const TRGSWLv1FFT &cond, params::Torus decompositionOffset, const cloudkey::CloudKey &ck)
{
    const size_t N = (size_t)ck.P.N;
    std::vector<uint32_t> f1 = detail_t::flat(in1), f2 = detail_t::flat(in2), out(2 * N);
    check(tfhe_cmux_with(ck.ctx(), cond.Flat.data(), decompositionOffset, f1.data(), f2.data(), out.data(), 1));
    return detail_t::unflat(out.data(), N);
}
// trgsw.go:234-252 (and :197-224 with one input): the bootstrapping key is the one resident in `ck`
inline std::vector<trlwe::TRLWELv1> BatchBlindRotate(const std::vector<tlwe::TLWELv0> &srcs, const trlwe::TRLWELv1 &testvec, params::Torus decompositionOffset,
                                                     const cloudkey::CloudKey &ck)
{
    detail_t::want_offset(ck, decompositionOffset);
    const size_t N = (size_t)ck.P.N, n1 = (size_t)ck.P.n + 1;
    std::vector<uint32_t> in(srcs.size() * n1), tv = detail_t::flat(testvec), out(srcs.size() * 2 * N);
    for (size_t i = 0; i < srcs.size(); i++) {
        if (srcs[i].P.size() != n1) throw Panic(TFHE_E_INVALID, "ciphertext length");
        std::copy(srcs[i].P.begin(), srcs[i].P.end(), in.begin() + i * n1);
    }
    check(tfhe_blind_rotate_batch(ck.ctx(), in.data(), tv.data(), 0, out.data(), (int)srcs.size(), -1));
    std::vector<trlwe::TRLWELv1> res;
    for (size_t i = 0; i < srcs.size(); i++) res.push_back(detail_t::unflat(out.data() + i * 2 * N, N));
    return res;
}
inline trlwe::TRLWELv1 BlindRotate(const tlwe::TLWELv0 &src, const trlwe::TRLWELv1 &testvec, params::Torus decompositionOffset, const cloudkey::CloudKey &ck)
{
    return BatchBlindRotate({src}, testvec, decompositionOffset, ck)[0];
}
// trgsw.go:285-312, keyswitch.go:10-37: the key-switching key is the one resident in `ck`
inline tlwe::TLWELv0 IdentityKeySwitching(const tlwe::TLWELv1 &src, const cloudkey::CloudKey &ck)
{
    if (src.P.size() != (size_t)ck.P.N + 1) throw Panic(TFHE_E_INVALID, "TLWELv1 length");
    tlwe::TLWELv0 out(ck.P.n);
    check(tfhe_keyswitch_batch(ck.ctx(), src.P.data(), out.P.data(), 1));
    return out;
}
inline void IdentityKeySwitchingAssign(const tlwe::TLWELv1 &src, const cloudkey::CloudKey &ck, tlwe::TLWELv0 &output) { output = IdentityKeySwitching(src, ck); }
} // namespace trgsw

namespace trlwe {
// trlwe.go:114-128, trlwe_ops.go:10-21 (any index k)
inline tlwe::TLWELv1 SampleExtractIndex(const TRLWELv1 &t, int k, const cloudkey::CloudKey &ck)
{
    tlwe::TLWELv1 out(ck.P.N);
    const std::vector<uint32_t> f = trgsw::detail_t::flat(t);
    check(tfhe_sample_extract_batch(ck.ctx(), f.data(), k, out.P.data(), 1));
    return out;
}
inline void SampleExtractIndexAssign(const TRLWELv1 &t, int k, const cloudkey::CloudKey &ck, tlwe::TLWELv1 &output) { output = SampleExtractIndex(t, k, ck); }
} // namespace trlwe

namespace evaluator {
// evaluator.go:14-35.  The bsk / ksk / decompositionOffset arguments of the Go methods are the
// ones resident in the CloudKey; outputs are caller-owned (the *Assign style).
class Evaluator {
  public:
    explicit Evaluator(const cloudkey::CloudKey &ck) : ck_(ck), n1_((size_t)ck.P.n + 1) {}

    // evaluator.go:50-81 : ctOut = bsk[keyIndex] (x) ctIn
    void ExternalProductAssign(int keyIndex, const trlwe::TRLWELv1 &ctIn, trlwe::TRLWELv1 &ctOut) const
    {
        const size_t N = (size_t)ck_.P.N;
        std::vector<uint32_t> in(2 * N), out(2 * N);
        std::copy(ctIn.A.begin(), ctIn.A.end(), in.begin());
        std::copy(ctIn.B.begin(), ctIn.B.end(), in.begin() + N);
        check(tfhe_external_product_batch(ck_.ctx(), keyIndex, in.data(), out.data(), 1));
        ctOut.A.assign(out.begin(), out.begin() + N);
        ctOut.B.assign(out.begin() + N, out.end());
    }
    // evaluator.go:50-81 with any operand, and evaluator.go:85-106: ctOut = ct0 + ctCond (x) (ct1 - ct0)
    void ExternalProductAssign(const trgsw::TRGSWLv1FFT &ctFourierGGSW, const trlwe::TRLWELv1 &ctIn, params::Torus decompositionOffset, trlwe::TRLWELv1 &ctOut) const
    {
        ctOut = trgsw::ExternalProductWithFFT(ctFourierGGSW, ctIn, decompositionOffset, ck_);
    }
    void CMuxAssign(const trgsw::TRGSWLv1FFT &ctCond, const trlwe::TRLWELv1 &ct0, const trlwe::TRLWELv1 &ct1, params::Torus decompositionOffset, trlwe::TRLWELv1 &ctOut) const
    {
        ctOut = trgsw::CMUX(ct0, ct1, ctCond, decompositionOffset, ck_);
    }
    // evaluator.go:110-135 ; testvec == nullptr selects the gate test vector (cloudkey.go:74-85)
    void BlindRotateAssign(const tlwe::TLWELv0 &ctIn, const trlwe::TRLWELv1 *testvec, trlwe::TRLWELv1 &ctOut) const
    {
        const size_t N = (size_t)ck_.P.N;
        std::vector<uint32_t> tv, out(2 * N);
        if (testvec) { tv = testvec->A; tv.insert(tv.end(), testvec->B.begin(), testvec->B.end()); }
        check(tfhe_blind_rotate_batch(ck_.ctx(), ctIn.P.data(), testvec ? tv.data() : nullptr, 0, out.data(), 1, -1));
        ctOut.A.assign(out.begin(), out.begin() + N);
        ctOut.B.assign(out.begin() + N, out.end());
    }
    // evaluator.go:139-148
    void BootstrapAssign(const tlwe::TLWELv0 &ctIn, const trlwe::TRLWELv1 *testvec, tlwe::TLWELv0 &ctOut) const
    {
        std::vector<uint32_t> tv;
        if (testvec) { tv = testvec->A; tv.insert(tv.end(), testvec->B.begin(), testvec->B.end()); }
        ctOut.P.resize(n1_);
        check(tfhe_bootstrap_batch(ck_.ctx(), ctIn.P.data(), testvec ? tv.data() : nullptr, 0, ctOut.P.data(), 1));
    }
    // evaluator.go:152-157 (returns an owned value, not a pointer into a 4-slot pool)
    tlwe::TLWELv0 Bootstrap(const tlwe::TLWELv0 &ctIn, const trlwe::TRLWELv1 *testvec = nullptr) const
    {
        tlwe::TLWELv0 out;
        BootstrapAssign(ctIn, testvec, out);
        return out;
    }
    // programmable_bootstrap.go:93-115 : the LUT is a TRLWE with A = 0, B = table
    void BootstrapLUTAssign(const tlwe::TLWELv0 &ctIn, const trlwe::TRLWELv1 &lut, tlwe::TLWELv0 &ctOut) const { BootstrapAssign(ctIn, &lut, ctOut); }
    tlwe::TLWELv0 BootstrapLUT(const tlwe::TLWELv0 &ctIn, const trlwe::TRLWELv1 &lut) const { return Bootstrap(ctIn, &lut); }
    tlwe::TLWELv0 BootstrapLUT(const tlwe::TLWELv0 &ctIn, const lut::LookUpTable &t) const { return Bootstrap(ctIn, &t.Poly); }
    // programmable_bootstrap.go:16-52 : table for f over [0, messageModulus), then bootstrap
    tlwe::TLWELv0 BootstrapFunc(const tlwe::TLWELv0 &ctIn, const std::function<int(int)> &f, int messageModulus) const
    {
        return BootstrapLUT(ctIn, lut::Generator(ck_.P, messageModulus).GenLookUpTable(f));
    }
    void BootstrapFuncAssign(const tlwe::TLWELv0 &ctIn, const std::function<int(int)> &f, int messageModulus, tlwe::TLWELv0 &ctOut) const
    {
        ctOut = BootstrapFunc(ctIn, f, messageModulus);
    }
    // extended tables: lut = Generator(P, m, ext).GenLookUpTableExtended(f), [ext][2][N]
    tlwe::TLWELv0 BootstrapLUTExtended(const tlwe::TLWELv0 &ctIn, const std::vector<params::Torus> &lut) const
    {
        const size_t per = 2 * (size_t)ck_.P.N;
        if (lut.empty() || lut.size() % per) throw Panic(TFHE_E_INVALID, "extended table must be [ext][2][N]");
        tlwe::TLWELv0 out;
        out.P.resize(n1_);
        check(tfhe_bootstrap_extended_batch(ck_.ctx(), ctIn.P.data(), lut.data(), 0, (int)(lut.size() / per), out.P.data(), 1));
        return out;
    }
    // batch forms (trgsw.go:234-252)
    std::vector<tlwe::TLWELv0> BatchBootstrap(const std::vector<tlwe::TLWELv0> &in, const trlwe::TRLWELv1 *testvec = nullptr) const
    {
        std::vector<uint32_t> tv, f = detail::flatten(in, n1_), out(f.size());
        if (testvec) { tv = testvec->A; tv.insert(tv.end(), testvec->B.begin(), testvec->B.end()); }
        check(tfhe_bootstrap_batch(ck_.ctx(), f.data(), testvec ? tv.data() : nullptr, 0, out.data(), (int)in.size()));
        return detail::unflatten(out, n1_);
    }
    // gates_helper.go:10-63 (the GPU path fuses these; kept for the Prepare + Bootstrap seam)
    tlwe::TLWELv0 PrepareNAND(const tlwe::TLWELv0 &a, const tlwe::TLWELv0 &b) const { auto r = a.Add(b).Neg(); r.SetB(r.B() + 0x20000000u); return r; }
    tlwe::TLWELv0 PrepareAND(const tlwe::TLWELv0 &a, const tlwe::TLWELv0 &b) const { auto r = a.Add(b); r.SetB(r.B() + 0xE0000000u); return r; }
    tlwe::TLWELv0 PrepareOR(const tlwe::TLWELv0 &a, const tlwe::TLWELv0 &b) const { auto r = a.Add(b); r.SetB(r.B() + 0x20000000u); return r; }
    tlwe::TLWELv0 PrepareXOR(const tlwe::TLWELv0 &a, const tlwe::TLWELv0 &b) const { auto r = a.AddMul(b, 2); r.SetB(r.B() + 0x40000000u); return r; }

  private:
    const cloudkey::CloudKey &ck_;
    size_t n1_;
};
} // namespace evaluator

namespace gates {
using Ciphertext = tlwe::TLWELv0;                         // gates.go:16

namespace detail_g {
inline std::vector<Ciphertext> run(int op, const std::vector<Ciphertext> &a, const std::vector<Ciphertext> &b,
                                   const std::vector<Ciphertext> *c, const cloudkey::CloudKey &ck)
{
    const size_t n1 = (size_t)ck.P.n + 1;
    if (a.size() != b.size() || (c && c->size() != a.size())) throw Panic(TFHE_E_INVALID, "operand counts differ");
    std::vector<uint32_t> fa = detail::flatten(a, n1), fb = detail::flatten(b, n1), fc, out(fa.size());
    if (c) fc = detail::flatten(*c, n1);
    check(tfhe_gate_batch(ck.ctx(), nullptr, op, fa.data(), fb.data(), c ? fc.data() : nullptr, out.data(), (int)a.size()));
    return detail::unflatten(out, n1);
}
inline Ciphertext one(int op, const Ciphertext &a, const Ciphertext &b, const cloudkey::CloudKey &ck) { return run(op, {a}, {b}, nullptr, ck)[0]; }
inline std::vector<Ciphertext> batch(int op, const std::vector<std::array<Ciphertext, 2>> &in, const cloudkey::CloudKey &ck)
{
    std::vector<Ciphertext> a, b;
    for (auto &p : in) { a.push_back(p[0]); b.push_back(p[1]); }
    return run(op, a, b, nullptr, ck);
}
// the same over the replicas of a CloudKeySet: contiguous shards [g B/G, (g+1) B/G), one thread per replica
inline std::vector<Ciphertext> run_on_set(int op, const std::vector<Ciphertext> &a, const std::vector<Ciphertext> &b,
                                          const std::vector<Ciphertext> *c, const cloudkey::CloudKeySet &set)
{
    const size_t n1 = (size_t)set.P.n + 1, B = a.size(), G = set.size();
    if (a.size() != b.size() || (c && c->size() != a.size())) throw Panic(TFHE_E_INVALID, "operand counts differ");
    std::vector<uint32_t> fa = detail::flatten(a, n1), fb = detail::flatten(b, n1), fc, out(fa.size());
    if (c) fc = detail::flatten(*c, n1);
    std::vector<int> rc(G, TFHE_OK);
    std::vector<std::string> err(G);
    std::vector<std::thread> th;
    for (size_t g = 0; g < G; g++) {
        const size_t lo = B * g / G, hi = B * (g + 1) / G;
        if (hi == lo) continue;
        th.emplace_back([&, g, lo, hi] {
            rc[g] = tfhe_gate_batch(set[g].ctx(), nullptr, op, fa.data() + lo * n1, fb.data() + lo * n1, c ? fc.data() + lo * n1 : nullptr,
                                    out.data() + lo * n1, (int)(hi - lo));
            if (rc[g]) err[g] = tfhe_last_error();          // the message is thread-local: take it where it was set
        });
    }
    for (auto &t : th) t.join();
    for (size_t g = 0; g < G; g++)
        if (rc[g]) throw Panic(rc[g], err[g]);
    return detail::unflatten(out, n1);
}
} // namespace detail_g

// gates.go:26-104
inline Ciphertext NAND(const Ciphertext &a, const Ciphertext &b, const cloudkey::CloudKey &ck) { return detail_g::one(TFHE_OP_NAND, a, b, ck); }
inline Ciphertext OR(const Ciphertext &a, const Ciphertext &b, const cloudkey::CloudKey &ck) { return detail_g::one(TFHE_OP_OR, a, b, ck); }
inline Ciphertext AND(const Ciphertext &a, const Ciphertext &b, const cloudkey::CloudKey &ck) { return detail_g::one(TFHE_OP_AND, a, b, ck); }
inline Ciphertext XOR(const Ciphertext &a, const Ciphertext &b, const cloudkey::CloudKey &ck) { return detail_g::one(TFHE_OP_XOR, a, b, ck); }
inline Ciphertext XNOR(const Ciphertext &a, const Ciphertext &b, const cloudkey::CloudKey &ck) { return detail_g::one(TFHE_OP_XNOR, a, b, ck); }
inline Ciphertext NOR(const Ciphertext &a, const Ciphertext &b, const cloudkey::CloudKey &ck) { return detail_g::one(TFHE_OP_NOR, a, b, ck); }
inline Ciphertext ANDNY(const Ciphertext &a, const Ciphertext &b, const cloudkey::CloudKey &ck) { return detail_g::one(TFHE_OP_ANDNY, a, b, ck); }
inline Ciphertext ANDYN(const Ciphertext &a, const Ciphertext &b, const cloudkey::CloudKey &ck) { return detail_g::one(TFHE_OP_ANDYN, a, b, ck); }
inline Ciphertext ORNY(const Ciphertext &a, const Ciphertext &b, const cloudkey::CloudKey &ck) { return detail_g::one(TFHE_OP_ORNY, a, b, ck); }
inline Ciphertext ORYN(const Ciphertext &a, const Ciphertext &b, const cloudkey::CloudKey &ck) { return detail_g::one(TFHE_OP_ORYN, a, b, ck); }
// gates.go:107-114 : a ? b : c, three bootstraps
inline Ciphertext MUX(const Ciphertext &a, const Ciphertext &b, const Ciphertext &c, const cloudkey::CloudKey &ck)
{
    std::vector<Ciphertext> cv{c};
    return detail_g::run(TFHE_OP_MUX, {a}, {b}, &cv, ck)[0];
}
inline Ciphertext NOT(const Ciphertext &a) { return a.Neg(); }                 // gates.go:117-119
inline Ciphertext Copy(const Ciphertext &a) { return a; }                      // gates.go:122-126
inline Ciphertext Constant(bool value, const params::Params &p)                // gates.go:61-69 (keeps the reference's 1 - mu)
{
    Ciphertext r(p.n);
    const uint32_t mu = 0x20000000u;
    r.SetB(value ? mu : 1u - mu);
    return r;
}
// gates.go:156-312 ([][2]*Ciphertext -> []*Ciphertext).  BatchXNOR follows the scalar XNOR sign.
using Pairs = std::vector<std::array<Ciphertext, 2>>;
inline std::vector<Ciphertext> BatchNAND(const Pairs &in, const cloudkey::CloudKey &ck) { return detail_g::batch(TFHE_OP_NAND, in, ck); }
inline std::vector<Ciphertext> BatchAND(const Pairs &in, const cloudkey::CloudKey &ck) { return detail_g::batch(TFHE_OP_AND, in, ck); }
inline std::vector<Ciphertext> BatchOR(const Pairs &in, const cloudkey::CloudKey &ck) { return detail_g::batch(TFHE_OP_OR, in, ck); }
inline std::vector<Ciphertext> BatchXOR(const Pairs &in, const cloudkey::CloudKey &ck) { return detail_g::batch(TFHE_OP_XOR, in, ck); }
inline std::vector<Ciphertext> BatchNOR(const Pairs &in, const cloudkey::CloudKey &ck) { return detail_g::batch(TFHE_OP_NOR, in, ck); }
inline std::vector<Ciphertext> BatchXNOR(const Pairs &in, const cloudkey::CloudKey &ck) { return detail_g::batch(TFHE_OP_XNOR, in, ck); }
// gates.Batch* over several GPUs from one process: op = TFHE_OP_NAND ... TFHE_OP_ORYN (MUX through BatchMUXOnSet)
inline std::vector<Ciphertext> BatchOnSet(int op, const Pairs &in, const cloudkey::CloudKeySet &set)
{
    std::vector<Ciphertext> a, b;
    for (auto &p : in) { a.push_back(p[0]); b.push_back(p[1]); }
    return detail_g::run_on_set(op, a, b, nullptr, set);
}
inline std::vector<Ciphertext> BatchMUXOnSet(const std::vector<std::array<Ciphertext, 3>> &in, const cloudkey::CloudKeySet &set)
{
    std::vector<Ciphertext> a, b, c;
    for (auto &t : in) { a.push_back(t[0]); b.push_back(t[1]); c.push_back(t[2]); }
    return detail_g::run_on_set(TFHE_OP_MUX, a, b, &c, set);
}
} // namespace gates

} // namespace tfhe
