// tfhe_hip.hip -- C ABI (include/tfhe_hip.h) over the HIP kernels in kernels.hpp.
//
// Host-side glue only: context/key residency, staging buffers, launches, HIP-event timing.
// There is NO CPU fallback: every entry point either runs the gfx950 kernels or fails with a
// TFHE_E_* code.
#include "../../include/tfhe_hip.h"
#include "tfhe_hip_internal.hpp"

#include <hip/hip_runtime.h>
#include <linux/futex.h>
#include <sys/random.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <deque>
#include <condition_variable>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "kernels.hpp"
#include "kernels_n2048.hpp"
#include "kernels_n512.hpp"
#include "kernels_quad.hpp"
#include "keyswitch_mfma.hpp"
#include "launch_blind_rotate.hpp"
#include "keygen.hpp"

using namespace tfhe;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(TFHE_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }      // locals are released on every early-return path
    bool fits(size_t bytes) const { return bytes <= cap; }
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return TFHE_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail(TFHE_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        cap = bytes;
        return TFHE_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

} // namespace

constexpr int kCombQuietUsDefault = 150;

struct tfhe_ctx {
    tfhe_params P{};
    int device = 0;
    int shape = 0;              // tfhe::Shape (launch_blind_rotate.hpp)
    uint32_t offset = 0;        // cloudkey.go:60-71
    int n1p = 0;                // row length of the packed KSK in words: n + 1 rounded up to 32 words = whole 128-byte lines, so
                                // that the 256-byte column slices the tiled key switches read are line-aligned in EVERY row (at
                                // Uint5, 1,072-word rows put every other row's slices across three lines instead of two and each
                                // XCD fetched the shared line again: 2.3 GB per launch for a 1.66 GB table, round 4)
    int num_cus = 256;          // hipDeviceProp_t.multiProcessorCount
    // host-pointer batches longer than one slab: transfers of slab s+1 / s-1 overlap the kernels of slab s
    hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
    hipEvent_t pipe_ev[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};      // [in, done, out][buffer]
    // One event per caller stream that "_dev" calls were enqueued on, re-recorded behind every such call: tfhe_ctx_sync
    // and tfhe_ctx_destroy wait on the EVENTS (context-owned, valid whatever became of the stream), never on a stream
    // handle the caller may have destroyed.  The handle value is only a lookup key.
    struct DevStreamMark { hipStream_t key; hipEvent_t ev; };
    std::vector<DevStreamMark> dev_marks;
    // Set once a "_dev" call has been enqueued on a capturing stream: the addresses of the intermediate buffers are
    // then recorded in the caller's hipGraph, so growing (= freeing and re-allocating) them would leave the graph
    // pointing at freed memory.  From then on a call that needs larger buffers fails with TFHE_E_INVALID instead
    // (tfhe_ctx_reserve before capturing; TFHE_OPT_FROZEN = 0 once the graphs are gone).
    bool frozen = false;
    // read lock-free by combine_cap (tfhe_gate_batch / tfhe_bootstrap_batch before they take any lock): atomics
    std::atomic<int> oct_limit{0};      // ... and of up to this many the eight-wave kernel (one bootstrap per CU)
    std::atomic<int> quad_limit{0};     // launches of up to this many bootstraps use the four-wave kernel (N = 1024 shapes)
    hipStream_t stream = nullptr;
    hipEvent_t ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    bool ev_valid[2] = {false, false};
    // cumulative timing (tfhe_timing_*): one event pair per launch while enabled
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tev[2];
    std::vector<hipEvent_t> ev_pool;
    DevBuf bsk, ksk, tw, gate_tv;
    DevBuf status;              // one int: kStatus* bits set by kernels (bad op codes on the _dev path)
    DevBuf kskB;                // byte-column copy of the key-switching key for the MFMA form (keyswitch_mfma.hpp; base-4 sets)
    DevBuf s_onehot;            // its per-launch one-hot digit matrix
    int ks_mfma_min = 0;        // batches of at least this many ciphertexts use it
    int ks_wide_ct = 0;         // k_keyswitch_wide: ciphertexts per wave, 0 = by batch size (TFHE_OPT_KS_WIDE_CT)
    DevBuf bskq, twq;           // four-wave layout of the key + its twiddles (N = 1024 shapes, kernels_quad.hpp)
    // read by tfhe_gate_batch / tfhe_bootstrap_batch before they take any lock (the header promises thread-safe calls): atomics
    std::atomic<bool> have_bsk{false}, have_ksk{false};
    // staging (grow-only)
    DevBuf s_in0, s_in1, s_in2, s_out, s_trlwe, s_tv, s_ops, s_idx, s_plan, s_t0, s_t1, s_t2, s_t3;
    DevBuf s_gsw_raw, s_gsw;    // one caller-supplied TRGSW operand (tfhe_external_product_with / tfhe_cmux_with): reference layout, wave-native layout
    std::recursive_mutex mu;    // host-pointer calls hold it for their whole duration, _dev calls while they reserve and enqueue
    // Flat combining of concurrent host-pointer callers (combine_request below).  One queue per kind of request -- gate batches
    // (tfhe_gate_batch) and programmable bootstraps (tfhe_bootstrap_batch) -- because a launch carries one kind.  Every queue is a ring
    // of three BATCHES: one accepting requests (callers claim rows of its page-locked staging with a compare-and-swap and copy their
    // operands in themselves), one whose launch is in flight, one whose callers are taking their rows out.
    struct GateReq {
        int kind;                       // 0: gates (ops / op_uniform, a, b, cc), 1: bootstraps through a table (a = in, b = tv, op_uniform = tv_per_item)
        const uint8_t *ops; int op_uniform; const uint32_t *a, *b, *cc; uint32_t *out; int B;
        int pos = 0;                    // first row of this request in its batch's staging
        int rc = TFHE_OK; std::string err;
    };
    struct CombBatch {
        static constexpr uint64_t kClosed = 1ull << 31;
        // rows claimed so far (bits 0-30), closed flag (bit 31), bootstraps those rows stand for (bits 32-63: a MUX row is two, see combine_weight)
        std::atomic<uint64_t> claim{kClosed};
        std::atomic<int> filled{0};     // rows whose operands are in staging and whose request is registered
        std::atomic<int> nreq{0};
        std::atomic<uint32_t> done{1};  // futex word: 0 from the moment the batch opens until its results (or errors) are final
        std::atomic<int> readers{0};    // followers that have yet to take their rows out: the slot is not re-opened before that
        std::atomic<bool> any_c{false}, full{false};
        std::vector<GateReq *> reqs;    // [cap_rows], fixed at allocation
        char *host = nullptr;           // page-locked: gates [a | b | c | out][cap_rows][n+1] + [cap_rows] op codes; bootstraps [in | out][cap_rows][n+1] + [cap_rows][2][N]
    };
    struct CombQueue {
        CombBatch ring[3];
        std::atomic<int> open{0};               // the batch that accepts requests
        std::atomic<uint32_t> open_gen{0};      // futex word: bumped whenever `open` moves (callers that found the batch closed or full sleep here)
        std::atomic<int> returning{0};          // requests of the most recent launch that have not been seen again yet (the gathering wait's target)
        std::atomic<uint32_t> gather{0};        // futex word the gathering leader sleeps on
        std::atomic<long long> last_done_ns{0}; // steady clock, when the most recent launch finished
        std::atomic<bool> ready{false};         // staging allocated (by the first caller that is about to FOLLOW: a lone caller never needs it)
        std::mutex alloc_mu;
        int cap_rows = 0;
    };
    CombQueue comb[2];
    // Host-pointer gate batches of more than one row per CU (up to one pipelined piece) from SEVERAL callers overlap: two slots of device
    // operand / result buffers, the upload of one call on the transfer stream while the kernels of the previous one run (gate_batch_overlapped)
    struct HostSlot {
        std::mutex mu;                  // held for a whole call: the slot's buffers are that call's
        DevBuf in0, in1, in2, out, ops, tv;
        hipEvent_t up = nullptr, done = nullptr;
    };
    HostSlot hslot[2];
    std::atomic<unsigned> hticket{0};
    std::mutex up_mu, down_mu;          // one uploader, one downloader at a time (PCIe is one pipe each way); kernels under `mu`
    std::atomic<int> combine_max{0};        // requests of at most this many items are combined (TFHE_OPT_COMBINE_MAX; 0 = off)
    std::atomic<long long> comb_launches{0}, comb_requests{0};     // combined launches issued / requests they carried (TFHE_OPT_COMBINE_*)
    // where the time between two combined launches goes (TFHE_OPT_COMBINE_US_*; nanoseconds, summed over the combined launches):
    // idle = previous launch done -> this one issued; gather = the part of it the leader spent waiting for returning callers;
    // launch = transfers + kernels + synchronisation
    std::atomic<long long> comb_ns_idle{0}, comb_ns_gather{0}, comb_ns_launch{0};
    std::atomic<long long> comb_exit[6];     // how the gathering waits ended (TFHE_OPT_COMBINE_EXIT_*): nobody to wait for, stale, all back, batch full, quiet window, deadline
    std::atomic<int> comb_quiet_us{kCombQuietUsDefault};     // TFHE_OPT_COMBINE_QUIET_US: how long the gathering leader waits without a new arrival before it launches
    std::vector<uint32_t> gate_tv_host;                 // the gate test vector (a combined bootstrap launch carries one table per item)
    void *hdr_host[2] = {nullptr, nullptr};             // page-locked key-blob headers (tfhe_key_export_dev): the asynchronous copy
                                                        // reads them after the call has returned
    int clone_path = 0;                                 // TFHE_OPT_CLONE_PATH: how tfhe_ctx_clone_to brought the keys here (0 = not a clone)
    bool clone_force_host = false;                      // TFHE_OPT_CLONE_FORCE_HOST (tests): clones OF this context take the host-staged path
    bool need_sync_all = false;                         // a stream's event could not be recorded: tfhe_ctx_sync falls back to hipDeviceSynchronize
};

namespace {

size_t bsk_elems(const tfhe_params &P) { return (size_t)P.n * 2 * P.L * 2 * (P.N / 2); }
size_t ksk_rows_ref(const tfhe_params &P) { return (size_t)P.N * P.t * (1u << P.basebit); }
size_t ksk_rows_packed(const tfhe_params &P) { return (size_t)P.N * P.t * ((1u << P.basebit) - 1); }  // + 1 zero row on device

// _dev entry points run on the caller's stream; NULL is HIP's default (null) stream, as for any hipStream_t.
hipStream_t pick(tfhe_ctx *, void *stream) { return (hipStream_t)stream; }

// Twiddle tables (negacyclic_fft.hpp layout), computed in long double.  One table per half h of
// the ring's root tree: N = 1024 has one (H = 1), N = 2048 two (H = 2, kernels_n2048.hpp).  The
// root index is u = h + H*(m + 8m' + 64m''); the level-k pre-twist is zeta^(stride * idx * (1 + 4 u_k))
// with u_k the part of u already fixed at that level and zeta = exp(i pi / N).
std::vector<cd> make_twiddles(int N)
{
    const int H = N / 1024;
    const long double pi = 3.14159265358979323846264338327950288L;
    auto zeta = [&](long e) {
        e %= 2L * N; if (e < 0) e += 2L * N;
        long double a = pi * (long double)e / (long double)N;
        return cd{(double)cosl(a), (double)sinl(a)};
    };
    if (N == 512) {                                       // layout: kernels_n512.hpp
        std::vector<cd> t(kTwCount512);
        for (int a = 0; a < 8; a++) {
            t[a] = zeta(32L * a);
            cd c = zeta(-32L * a);
            t[8 + a] = cd{c.re / 256.0, c.im / 256.0};
        }
        for (int hl = 0; hl < 32; hl++) {
            const int m = hl >> 2, i = hl & 3;
            for (int b = 0; b < 8; b++) t[kTw512Level2 + b * 32 + hl] = zeta(4L * b * (1 + 4 * m));
            for (int q = 0; q < 2; q++)
                for (int c = 0; c < 4; c++)
                    t[kTw512Level3 + (4 * q + c) * 32 + hl] = zeta((long)c * (1 + 4 * (m + 8 * (4 * q + i))));
        }
        return t;
    }
    std::vector<cd> t((size_t)kTwCount1024 * H);
    for (int h = 0; h < H; h++) {
        cd *T = t.data() + (size_t)h * kTwCount1024;
        for (int a = 0; a < 8; a++) {
            T[a] = zeta(64L * a * (1 + 4 * h));
            cd c = zeta(-64L * a * (1 + 4 * h));
            T[8 + a] = cd{c.re / (512.0 * H), c.im / (512.0 * H)};
        }
        for (int k = 0; k < 8; k++)
            for (int lane = 0; lane < 64; lane++) {
                const int m = lane >> 3, mp = lane & 7;
                T[kTwLevel2 + k * 64 + lane] = zeta(8L * k * (1 + 4 * (h + H * m)));
                T[kTwLevel3 + k * 64 + lane] = zeta((long)k * (1 + 4 * (h + H * (m + 8 * mp))));
            }
    }
    return t;
}

// Twiddle table of the four-wave kernel (kernels_quad.hpp layout), N = 1024: root index
// u = h + 2m + 8m' + 32m'' + 128m'''.
std::vector<cd> make_twiddles_quad()
{
    const int N = 1024;
    const long double pi = 3.14159265358979323846264338327950288L;
    auto zeta = [&](long e) {
        e %= 2L * N; if (e < 0) e += 2L * N;
        long double a = pi * (long double)e / (long double)N;
        return cd{(double)cosl(a), (double)sinl(a)};
    };
    std::vector<cd> t((size_t)2 * kTwQuadHalf);
    for (int h = 0; h < 2; h++) {
        cd *T = t.data() + (size_t)h * kTwQuadHalf;
        for (int a = 0; a < 4; a++) {
            T[a] = zeta(64L * a * (1 + 4 * h));
            const cd c = zeta(-64L * a * (1 + 4 * h));
            T[4 + a] = cd{c.re / 512.0, c.im / 512.0};
        }
        for (int lane = 0; lane < 64; lane++) {
            const int m = lane >> 4, mp = (lane >> 2) & 3, mpp = lane & 3;
            for (int a = 1; a < 4; a++) {
                T[8 + (0 * 3 + a - 1) * 64 + lane] = zeta(16L * a * (1 + 4 * (h + 2 * m)));
                T[8 + (1 * 3 + a - 1) * 64 + lane] = zeta(4L * a * (1 + 4 * (h + 2 * m + 8 * mp)));
                T[8 + (2 * 3 + a - 1) * 64 + lane] = zeta((long)a * (1 + 4 * (h + 2 * m + 8 * mp + 32 * mpp)));
            }
        }
    }
    return t;
}

// Event pair bracketing one launch of kernel `which` on stream st.  Nothing is recorded while the stream is being
// captured into a hipGraph (the events would belong to the graph, not to this context).
bool stream_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

// Intermediate buffers of the batch paths (s_trlwe, s_onehot, s_idx, s_plan, s_t0..s_t3) are grow-only, and growing
// means hipFree + hipMalloc: an implicit device synchronisation, illegal while `st` is being captured, and fatal for
// a hipGraph that already recorded the old address (tfhe_ctx::frozen).  In both cases the call is refused with a
// message that names the remedy instead of a raw HIP error or, worse, a graph replaying into freed memory.
int grow(tfhe_ctx *c, DevBuf &b, size_t bytes, hipStream_t st, const char *what,
         const char *remedy = "tfhe_ctx_reserve(ctx, max_batch, with_mux)")
{
    if (b.fits(bytes)) return TFHE_OK;
    const bool cap = stream_capturing(st);
    if (cap || c->frozen)
        return fail(TFHE_E_INVALID,
                    "the context's %s buffer would have to grow from %zu to %zu bytes %s: call %s for the largest batch BEFORE capturing%s",
                    what, b.cap, bytes,
                    cap ? "while the stream is being captured into a hipGraph (no allocation is possible there)"
                        : "but a captured hipGraph holds its current address (the context is frozen)",
                    remedy, cap ? "" : ", or clear TFHE_OPT_FROZEN once every such graph is destroyed");
    return b.reserve(bytes);
}

// Behind every "_dev" call: re-record the context-owned event of the caller's stream (see tfhe_ctx::dev_marks).
int mark_dev_stream(tfhe_ctx *c, hipStream_t st)
{
    if (stream_capturing(st)) return TFHE_OK;             // an event record would become a node of the caller's graph
    // The batch this mark follows HAS been enqueued: a failure to record the mark must not be reported as a failure of the
    // call (the caller would believe nothing ran).  It is remembered instead, and tfhe_ctx_sync / tfhe_ctx_destroy then wait
    // for the whole device.
    auto give_up = [&]() { (void)hipGetLastError(); c->need_sync_all = true; return TFHE_OK; };
    for (auto &m : c->dev_marks)
        if (m.key == st) return hipEventRecord(m.ev, st) == hipSuccess ? TFHE_OK : give_up();
    constexpr size_t kSoftMarks = 8;
    tfhe_ctx::DevStreamMark m{st, nullptr};
    if (c->dev_marks.size() >= kSoftMarks) {              // recycle a mark whose work is already done -- never wait for one here:
        for (size_t i = 0; i < c->dev_marks.size(); i++)  // this is an enqueue-only call holding the context mutex
            if (hipEventQuery(c->dev_marks[i].ev) == hipSuccess) {
                m.ev = c->dev_marks[i].ev;
                c->dev_marks.erase(c->dev_marks.begin() + (long)i);
                break;
            }
        (void)hipGetLastError();                          // hipErrorNotReady of the queries
    }
    if (!m.ev && hipEventCreateWithFlags(&m.ev, hipEventDisableTiming) != hipSuccess) return give_up();     // otherwise the list grows
    c->dev_marks.push_back(m);
    return hipEventRecord(m.ev, st) == hipSuccess ? TFHE_OK : give_up();
}

int timing_begin(tfhe_ctx *c, int which, hipStream_t st, hipEvent_t *stop)
{
    *stop = nullptr;
    if (stream_capturing(st)) return TFHE_OK;
    hipEvent_t a, b;
    if (c->timing) {
        for (hipEvent_t *e : {&a, &b}) {
            if (!c->ev_pool.empty()) { *e = c->ev_pool.back(); c->ev_pool.pop_back(); }
            else HIP_TRY(hipEventCreate(e));
        }
        c->tev[which].emplace_back(a, b);
    } else {
        a = c->ev[which][0]; b = c->ev[which][1];
    }
    HIP_TRY(hipEventRecord(a, st));
    *stop = b;
    return TFHE_OK;
}

int timing_end(tfhe_ctx *c, int which, hipStream_t st, hipEvent_t stop)
{
    if (!stop) return TFHE_OK;
    HIP_TRY(hipEventRecord(stop, st));
    c->ev_valid[which] = !c->timing;
    return TFHE_OK;
}

int check_ctx(tfhe_ctx *c)
{
    if (!c) return fail(TFHE_E_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    return TFHE_OK;
}

// After any key load / keygen on an N = 1024 shape: derive the four-wave layout from the two-wave one.
int make_quad_key(tfhe_ctx *c, hipStream_t st)
{
    if (!shape_is_1024(c->shape)) return TFHE_OK;
    const size_t elems = bsk_elems(c->P);
    int rc;
    if ((rc = c->bskq.reserve(elems * sizeof(cd)))) return rc;
    hipLaunchKernelGGL(k_bsk_quad_from_wave, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, c->bsk.as<cd>(),
                       c->bskq.as<cd>(), c->P.n, c->P.L);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

// Byte-column copy of the packed key-switching key for k_keyswitch_mfma (base-4 sets; keyswitch_mfma.hpp).
constexpr int kKsMfmaMinDefault = 24;       // from 24 ciphertexts on; below, the per-ciphertext gather dealt to up to 16 workgroups per
                                            // ciphertext is quicker (128-bit set: 0.026-0.030 vs 0.038-0.040 ms at 1...16, 0.042 vs 0.038 at 31;
                                            // at 1,024: 0.135 vs 0.51 ms for the tiled vector kernel) -- profiles/r04_t_small_launches_uint.txt
constexpr int kKsMfmaChunk = 1024;           // ciphertexts per one-hot matrix (38 MB at the 128-bit set)
int ks_mfma_pieces(const tfhe_params &P) { return P.t * P.N / (16 >> P.basebit); }      // 16-K pieces: K = N t base
bool ks_mfma_shape(const tfhe_params &P)
{
    return (P.basebit == 2 || P.basebit == 4) && P.N % 64 == 0 && 1 + P.basebit * P.t <= 32 && (ks_mfma_pieces(P) / 2) % (2 * kKsStage) == 0;
}
int ks_mfma_cols(const tfhe_params &P) { return (4 * (P.n + 1) + kKsGroup - 1) / kKsGroup * kKsGroup; }
int make_mfma_ksk(tfhe_ctx *c, hipStream_t st)
{
    if (!ks_mfma_shape(c->P) || c->ks_mfma_min <= 0) return TFHE_OK;
    const int colsP = ks_mfma_cols(c->P);
    const size_t pieces = (size_t)ks_mfma_pieces(c->P) * colsP;
    int rc;
    if ((rc = c->kskB.reserve(pieces * sizeof(uint4)))) return rc;
    const dim3 grid((unsigned)((pieces + 255) / 256));
    if (c->P.basebit == 2)
        hipLaunchKernelGGL(k_ksk_mfma_pack<2>, grid, dim3(256), 0, st, c->ksk.as<uint32_t>(), c->kskB.as<uint4>(), c->P.N, c->P.t,
                           c->P.n + 1, c->n1p, colsP);
    else
        hipLaunchKernelGGL(k_ksk_mfma_pack<4>, grid, dim3(256), 0, st, c->ksk.as<uint32_t>(), c->kskB.as<uint4>(), c->P.N, c->P.t,
                           c->P.n + 1, c->n1p, colsP);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

// Items of one blind-rotate request (see BlindRotateArgs): direct operands, or the list form of the MUX passes.
struct RotateJob {
    const uint32_t *in0 = nullptr, *in1 = nullptr;
    const uint8_t *ops = nullptr;
    int op_uniform = -1;
    const uint32_t *tv = nullptr;
    int tv_per_item = 0;
    uint32_t *out = nullptr;
    int first = 0, B = 0, nsteps = -1;
    const int *idx = nullptr, *count = nullptr;
    int split = 0;
    const uint32_t *list_in1 = nullptr;
    int list_in1_by_idx = 0, list_op = 0;
};

int launch_blind_rotate(tfhe_ctx *c, const RotateJob &j, hipStream_t st)
{
    if (!c->have_bsk) return fail(TFHE_E_NOKEY, "bootstrapping key not loaded");
    if (j.B <= 0) return TFHE_OK;
    BlindRotateArgs a{};
    a.bsk = c->bsk.as<cd>();
    a.tw = c->tw.as<cd>();
    a.bskq = c->bskq.as<cd>(); a.twq = c->twq.as<cd>();
    a.in0 = j.in0; a.in1 = j.in1; a.ops = j.ops; a.op_uniform = j.op_uniform;
    a.tv = j.tv ? j.tv : c->gate_tv.as<uint32_t>();
    a.tv_stride = (j.tv && j.tv_per_item) ? 2L * c->P.N : 0;
    a.out = j.out;
    a.n = c->P.n; a.Nbit = c->P.Nbit;
    a.nsteps = (j.nsteps < 0 || j.nsteps > c->P.n) ? c->P.n : j.nsteps;
    a.offset = c->offset;
    a.first = j.first;
    a.idx = j.idx; a.count = j.count; a.split = j.split;
    a.list_in1 = j.list_in1; a.list_in1_by_idx = j.list_in1_by_idx; a.list_op = j.list_op;
    a.status = c->status.as<int>();
    hipEvent_t stop;
    int trc = timing_begin(c, 0, st, &stop);
    if (trc) return trc;
    launch_blind_rotate(c->shape, a, j.B, c->num_cus, c->quad_limit, c->oct_limit, st);
    HIP_TRY(hipGetLastError());
    return timing_end(c, 0, st, stop);
}

// d_count: optional device-side item count (list launches); B is then the worst case the grids are sized for.
int launch_keyswitch(tfhe_ctx *c, const uint32_t *d_trlwe, uint32_t *d_out, int B, const int *d_count, hipStream_t st)
{
    if (!c->have_ksk) return fail(TFHE_E_NOKEY, "key-switching key not loaded");
    if (B <= 0) return TFHE_OK;
    KeySwitchArgs a{};
    a.trlwe = d_trlwe; a.ksk = c->ksk.as<uint32_t>(); a.out = d_out;
    a.n = c->P.n; a.N = c->P.N; a.t = c->P.t; a.basebit = c->P.basebit; a.n1p = c->n1p;
    a.count = d_count;
    const int ch = (c->n1p + 255) / 256;
    const bool mfma = c->kskB.p && c->ks_mfma_min > 0 && B >= c->ks_mfma_min;
    if (mfma) {                 // the one-hot matrix of one chunk (a no-op after reserve_scratch / the first call)
        const int Bc = B < kKsMfmaChunk ? B : kKsMfmaChunk, MpadMax = (Bc + kKsGroup - 1) / kKsGroup * kKsGroup;
        int rc = grow(c, c->s_onehot, (size_t)ks_mfma_pieces(c->P) * MpadMax * sizeof(uint4), st, "one-hot digit");
        if (rc) return rc;
    }
    hipEvent_t stop;
    int trc = timing_begin(c, 1, st, &stop);
    if (trc) return trc;
    // base-4 and base-16 sets: the exact int8 matrix-core form (keyswitch_mfma.hpp), in chunks of kKsMfmaChunk ciphertexts
    if (mfma) {
        const int N = c->P.N, t = c->P.t, n1 = c->P.n + 1, colsP = ks_mfma_cols(c->P);
        const size_t tot = (size_t)B * n1;
        hipLaunchKernelGGL(k_ks_init, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_trlwe, d_out, c->P.n, N, B, d_count);
        const uint32_t bias_word = (uint32_t)((unsigned long long)(128ull * N * t) * 0x01010101ull);
        const int pairs_total = ks_mfma_pieces(c->P) / 2 / (2 * kKsStage), n_groups = colsP / kKsGroup;
        for (int m_base = 0; m_base < B; m_base += kKsMfmaChunk) {
            const int M = B - m_base < kKsMfmaChunk ? B - m_base : kKsMfmaChunk;
            const int Mpad = (M + kKsGroup - 1) / kKsGroup * kKsGroup, m_groups = Mpad / kKsGroup;
            // K ranges: as many as give every CU at most ONE workgroup (that is all a CU holds: no second, partial round)
            int parts = c->num_cus / (m_groups * n_groups);
            if (parts < 1) parts = 1;
            if (parts > pairs_total) parts = pairs_total;
            const int per_part = (pairs_total + parts - 1) / parts;
            parts = (pairs_total + per_part - 1) / per_part;
            if (c->P.basebit == 2)
                hipLaunchKernelGGL(k_ks_onehot<2>, dim3(Mpad / 16, N / 64), dim3(256), 0, st, d_trlwe + (size_t)m_base * 2 * N,
                                   c->s_onehot.as<uint4>(), N, t, M, Mpad, d_count, m_base);
            else
                hipLaunchKernelGGL(k_ks_onehot<4>, dim3(Mpad / 16, N / 64), dim3(256), 0, st, d_trlwe + (size_t)m_base * 2 * N,
                                   c->s_onehot.as<uint4>(), N, t, M, Mpad, d_count, m_base);
            const int work = m_groups * n_groups * parts, grid = (work + 7) / 8 * 8;      // a multiple of the 8 XCDs (see the kernel)
            hipLaunchKernelGGL(k_keyswitch_mfma, dim3((unsigned)grid), dim3(512), 0, st,
                               c->s_onehot.as<uint4>(), c->kskB.as<uint4>(), d_out + (size_t)m_base * n1, Mpad, colsP, n1, M, d_count,
                               m_base, m_groups, n_groups, pairs_total, per_part, parts, bias_word);
        }
        HIP_TRY(hipGetLastError());
        return timing_end(c, 1, st, stop);
    }
    // base-4 sets (80/110/128-bit): tiles of 32 ciphertexts x 256 columns, two (i, j) steps per LDS read
    constexpr int kT = 32;
    if (c->P.basebit == 2 && B >= kT && c->P.N % 16 == 0) {
        const int ct_tiles = (B + kT - 1) / kT, col_slices = ((c->n1p >> 2) + 63) / 64;
        // coefficient ranges: a multiple of 8 (XCD decode), IC = N/ranges even and >= 8, about 1.5 single-wave
        // workgroups per SIMD (IC = 64 at B = 1024: 0.50 ms; IC = 32: 0.55 ms)
        int ranges = 8;
        while (ranges * 16 <= c->P.N && ct_tiles * col_slices * ranges < 6 * c->num_cus) ranges *= 2;
        const int IC = c->P.N / ranges;
        const size_t tot = (size_t)B * (c->P.n + 1);
        hipLaunchKernelGGL(k_ks_init, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_trlwe, d_out, c->P.n, c->P.N, B, d_count);
        hipLaunchKernelGGL((k_keyswitch_pair<kT>), dim3((unsigned)(ct_tiles * col_slices * ranges)), dim3(64), 0, st, a, B, IC,
                           ct_tiles, col_slices);
        HIP_TRY(hipGetLastError());
        return timing_end(c, 1, st, stop);
    }
    // larger bases (Uint sets), batches that fill at least one wave of ciphertexts: column-sliced tiles
    if (c->P.basebit >= 4 && c->P.basebit <= 7 && B >= 64) {
        // 256 ciphertexts per workgroup (64 per wave).  The form with 128 per wave (a key row crosses L2 once per 512 ciphertexts)
        // needs 296 registers, i.e. one wave per SIMD: 0.92 vs 0.57 ms at Uint5 x 512 (profiles/r04_e_keyswitch_hbm.txt); it stays
        // selectable for measurements (TFHE_OPT_KS_WIDE_CT = 128).
        const bool big = c->ks_wide_ct == 128 && c->P.basebit <= 6;      // measured slower (one wave per SIMD): only on request
        const int per_wg = big ? 512 : 256;
        const int ct_tiles = (B + per_wg - 1) / per_wg, col_blocks = (c->P.n + 1 + 63) / 64;
        // coefficient ranges: as many as fill k whole rounds of the resident workgroups (three per CU by registers, two at
        // base 128 by LDS and with 128 ciphertexts per wave by registers) -- the kernel's time goes with rounds x coefficients
        // per workgroup, so a grid that ends in a part-filled round wastes the difference (Uint5 x 512: 1,088 workgroups on 768
        // slots 0.60 ms, 3,060 on 4 x 768 0.50 ms; profiles/r03_k_keyswitch_rounds.txt).  k = 1...8 with at least 20
        // coefficients per workgroup: the best fill, the larger k on a tie (shorter workgroups even out the tail).
        const int slots = (big ? 1 : c->P.basebit >= 7 ? 2 : 3) * c->num_cus, units = ct_tiles * col_blocks;
        int ranges = 1;
        double best_fill = 0.0;
        for (int k = 1; k <= 8; k++) {
            int r = (int)((long long)k * slots / units);
            if (r > c->P.N / 20) r = c->P.N / 20;
            if (r < 1) r = 1;
            const long long wgs = (long long)units * r, rounds = (wgs + slots - 1) / slots;
            const double fill = (double)wgs / (double)(rounds * slots);
            if (fill >= best_fill - 0.005) { if (fill > best_fill) best_fill = fill; ranges = r; }
        }
        const size_t tot = (size_t)B * (c->P.n + 1);
        hipLaunchKernelGGL(k_ks_init, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_trlwe, d_out, c->P.n, c->P.N, B, d_count);
        const dim3 g((unsigned)((col_blocks * ranges + 7) / 8 * 8 * ct_tiles));                // XCD decode: see the kernel
#define KSW(BBv)                                                                                                            \
        if (big) hipLaunchKernelGGL((k_keyswitch_wide<BBv, 128>), g, dim3(256), 0, st, a, B, ranges, ct_tiles, col_blocks);     \
        else hipLaunchKernelGGL((k_keyswitch_wide<BBv, 64>), g, dim3(256), 0, st, a, B, ranges, ct_tiles, col_blocks)
        switch (c->P.basebit) {
        case 4: KSW(4); break;
        case 5: KSW(5); break;
        case 6: KSW(6); break;
        default: hipLaunchKernelGGL((k_keyswitch_wide<7, 64>), g, dim3(256), 0, st, a, B, ranges, ct_tiles, col_blocks); break;
        }
#undef KSW
        HIP_TRY(hipGetLastError());
        return timing_end(c, 1, st, stop);
    }
    // row indices fit 16 bits for the 2-bit key-switch base of the N=1024 sets (halves the LDS list)
    const bool small_idx = ksk_rows_packed(c->P) < 65535;
    // few ciphertexts: the digit list of each is dealt to `parts` workgroups (up to 16, at least 256 digits each), so that a
    // lone ciphertext's rows do not all go through one CU
    int parts = c->num_cus / B;
    if (parts > 16) parts = 16;
    while (parts > 1 && c->P.N * c->P.t / parts < 256) parts--;
    if (parts < 1) parts = 1;
    if (parts > 1) {
        const size_t tot = (size_t)B * (c->P.n + 1);
        hipLaunchKernelGGL(k_ks_init, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_trlwe, d_out, c->P.n, c->P.N, B, d_count);
    }
#define KS_LAUNCH(CH)                                                                                   \
    if (small_idx) hipLaunchKernelGGL((k_extract_keyswitch<CH, uint16_t>), dim3(B * parts), dim3(256), 0, st, a, parts); \
    else hipLaunchKernelGGL((k_extract_keyswitch<CH, uint32_t>), dim3(B * parts), dim3(256), 0, st, a, parts)
    switch (ch) {
    case 1: KS_LAUNCH(1); break;
    case 2: KS_LAUNCH(2); break;
    case 3: KS_LAUNCH(3); break;
    case 4: KS_LAUNCH(4); break;
    case 5: KS_LAUNCH(5); break;
    default: return fail(TFHE_E_INVALID, "LWE dimension %d too large for the key-switch kernel", c->P.n);
    }
#undef KS_LAUNCH
    HIP_TRY(hipGetLastError());
    return timing_end(c, 1, st, stop);
}

// Device work is issued in slabs of at most this many items, so the context's intermediate buffers have a fixed
// size (64 co-resident launches' worth = 65,536 bootstraps at N = 1024: 1.07 GB of TRLWE samples with the MUX passes)
// however large the batch -- after the first large call nothing is ever re-allocated under work in flight.  Large
// slabs because every MUX pass of a slab ends in one partly filled launch whose size the host does not know (the
// list lengths live on the device): with 16,384-item slabs those tails were 7 % of a mixed gate stream.
// Host-pointer batches are cut into pipe_items pieces (a quarter slab) so that transfers overlap the kernels.
int launch_items(const tfhe_ctx *c) { return (shape_is_512(c->shape) ? 8 : shape_is_1024(c->shape) ? 4 : 2) * c->num_cus; }
int slab_items(const tfhe_ctx *c) { return 64 * launch_items(c); }
int pipe_items(const tfhe_ctx *c) { return 16 * launch_items(c); }

// Scratch for slabs of up to `items` bootstraps (mux: with the three-pass MUX form).  hipMalloc is not allowed
// while a stream is being captured: callers that capture run one call (or tfhe_ctx_reserve) beforehand.
int reserve_scratch(tfhe_ctx *c, int items, bool mux, hipStream_t st)
{
    const int S = items < slab_items(c) ? items : slab_items(c);
    const size_t trl = (size_t)S * 2 * c->P.N * sizeof(uint32_t), rows = (size_t)S * (c->P.n + 1) * sizeof(uint32_t);
    int rc;
    if ((rc = grow(c, c->s_trlwe, mux ? 2 * trl : trl, st, "TRLWE accumulator"))) return rc;
    if (c->kskB.p) {            // one-hot digit matrix of the matrix-core key switch (one chunk; launch_keyswitch)
        const int Bc = S < kKsMfmaChunk ? S : kKsMfmaChunk, Mpad = (Bc + kKsGroup - 1) / kKsGroup * kKsGroup;
        if ((rc = grow(c, c->s_onehot, (size_t)ks_mfma_pieces(c->P) * Mpad * sizeof(uint4), st, "one-hot digit"))) return rc;
    }
    if (mux) {
        const int nb = (S + kPlanBlock - 1) / kPlanBlock;
        if ((rc = grow(c, c->s_idx, (size_t)S * sizeof(int), st, "MUX list")) ||
            (rc = grow(c, c->s_plan, (size_t)(2 * nb + 2) * sizeof(int), st, "MUX plan")) ||
            (rc = grow(c, c->s_t0, rows, st, "MUX temporary")) || (rc = grow(c, c->s_t1, rows, st, "MUX temporary")))
            return rc;
    }
    return TFHE_OK;
}

// Full gate batch on device pointers, enqueue-only.  MUX = OR(AND(a,b), AND(NOT a, c)) (gates.go:107-114;
// AND(NOT a, c) has exactly ANDNY's linear form -a + c - 1/8) in two bootstrap passes:
//   plan    compact list of the MUX items, built on the device (k_mux_*; all items when op_uniform = MUX)
//   pass 1  ONE blind-rotate request over S + Mx items: item v < S runs its own gate (MUX reads as AND(a,b)),
//           list entry k runs ANDNY(a[idx k], c[idx k]); key switch to out (first S) and to y (entries)
//   pass 2  OR(out[idx k], y[k]) over the list, scattered back to out[idx k]
// The host never learns Mx: list launches are sized for the worst case and the workgroups of entries past the
// device-side count exit at once.  Without a third operand no item can be a MUX and only pass 1's first half runs.
int gate_batch_device(tfhe_ctx *c, const uint8_t *d_ops, int op_uniform, const uint32_t *d_a, const uint32_t *d_b,
                      const uint32_t *d_c, uint32_t *d_out, int B, hipStream_t st)
{
    if (!d_ops && (op_uniform < 0 || op_uniform > TFHE_OP_MUX)) return fail(TFHE_E_INVALID, "bad op code %d", op_uniform);
    const bool mux = d_c && (d_ops || op_uniform == TFHE_OP_MUX);
    if (!d_ops && op_uniform == TFHE_OP_MUX && !d_c) return fail(TFHE_E_INVALID, "MUX needs the third operand");
    int rc;
    if ((rc = reserve_scratch(c, B, mux, st))) return rc;
    const size_t n1 = (size_t)c->P.n + 1, trlw = (size_t)2 * c->P.N;
    const int slab = slab_items(c);
    for (int base = 0; base < B; base += slab) {
        const int S = B - base < slab ? B - base : slab;
        const uint8_t *ops = d_ops ? d_ops + base : nullptr;
        const uint32_t *a = d_a + base * n1, *b = d_b + base * n1, *cc = d_c ? d_c + base * n1 : nullptr;
        uint32_t *out = d_out + base * n1, *trl = c->s_trlwe.as<uint32_t>();
        RotateJob j;
        j.in0 = a; j.in1 = b; j.ops = ops; j.op_uniform = op_uniform; j.out = trl; j.B = S;
        if (!mux) {
            if ((rc = launch_blind_rotate(c, j, st)) || (rc = launch_keyswitch(c, trl, out, S, nullptr, st))) return rc;
            continue;
        }
        int *idx = c->s_idx.as<int>(), *plan = c->s_plan.as<int>();
        const int nb = (S + kPlanBlock - 1) / kPlanBlock;
        int *count = plan + 2 * nb + 1;                    // plan: [nb] block counts, [nb + 1] offsets, count
        if (ops) {
            hipLaunchKernelGGL(k_mux_count, dim3(nb), dim3(256), 0, st, ops, S, plan, c->status.as<int>());
            hipLaunchKernelGGL(k_mux_scan, dim3(1), dim3(256), 0, st, (const int *)plan, nb, plan + nb, count);
            hipLaunchKernelGGL(k_mux_fill, dim3(nb), dim3(256), 0, st, ops, S, (const int *)(plan + nb), idx);
        } else {
            hipLaunchKernelGGL(k_mux_all, dim3((S + 255) / 256), dim3(256), 0, st, S, idx, count);
        }
        uint32_t *y = c->s_t0.as<uint32_t>(), *z = c->s_t1.as<uint32_t>();
        // pass 1: S direct items + up to S list entries in one request
        j.B = 2 * S; j.idx = idx; j.count = count; j.split = S;
        j.list_in1 = cc; j.list_in1_by_idx = 1; j.list_op = TFHE_OP_ANDNY;
        if ((rc = launch_blind_rotate(c, j, st))) return rc;
        if ((rc = launch_keyswitch(c, trl, out, S, nullptr, st))) return rc;
        if ((rc = launch_keyswitch(c, trl + (size_t)S * trlw, y, S, count, st))) return rc;
        // pass 2: OR(out[idx k], y[k]) -> z[k] -> out[idx k]
        RotateJob o;
        o.in0 = out; o.in1 = out; o.op_uniform = TFHE_OP_OR; o.out = trl; o.B = S;
        o.idx = idx; o.count = count; o.split = 0; o.list_in1 = y; o.list_in1_by_idx = 0; o.list_op = TFHE_OP_OR;
        if ((rc = launch_blind_rotate(c, o, st))) return rc;
        if ((rc = launch_keyswitch(c, trl, z, S, count, st))) return rc;
        hipLaunchKernelGGL(k_scatter_rows, dim3(S), dim3(256), 0, st, (const uint32_t *)z, (const int *)idx, out, (int)n1,
                           (const int *)count);
        HIP_TRY(hipGetLastError());
    }
    return TFHE_OK;
}

// Plain / LUT bootstraps of B items in slabs (see slab_items).
int bootstrap_device(tfhe_ctx *c, const uint32_t *d_in, const uint32_t *d_tv, int tv_per_item, uint32_t *d_out, int B,
                     hipStream_t st)
{
    int rc;
    if ((rc = reserve_scratch(c, B, false, st))) return rc;
    const size_t n1 = (size_t)c->P.n + 1, trlw = (size_t)2 * c->P.N;
    const int slab = slab_items(c);
    for (int base = 0; base < B; base += slab) {
        const int S = B - base < slab ? B - base : slab;
        RotateJob j;
        j.in0 = d_in + base * n1; j.out = c->s_trlwe.as<uint32_t>(); j.B = S;
        j.tv = d_tv ? d_tv + (tv_per_item ? base * trlw : 0) : nullptr; j.tv_per_item = tv_per_item;
        if ((rc = launch_blind_rotate(c, j, st)) || (rc = launch_keyswitch(c, c->s_trlwe.as<uint32_t>(), d_out + base * n1, S, nullptr, st)))
            return rc;
    }
    return TFHE_OK;
}

// Programmable bootstrap through an extended lookup table (kernels_n2048.hpp), N = 2048 shape only.
//   ext = 2 (Uint6): the persistent eight-wave kernel -- both accumulator components in LDS, all n steps in one launch;
//   other ext:       one launch per CMUX step over (item, component) pairs, accumulators double-buffered in global memory
//                    (a functional path for the experimental sets), in chunks of items so that the buffers have a
//                    fixed size whatever the batch.
// Scratch of the extended-table bootstrap for batches of up to `items`: polyExtendFactor 2 runs the persistent kernel over the
// ordinary slabs; the other factors keep two sets of ext accumulators and the mod-switched samples of one chunk.
int reserve_extended_scratch(tfhe_ctx *c, int items, int ext, hipStream_t st)
{
    if (ext == 2) return reserve_scratch(c, items, false, st);
    const int chunk = launch_items(c), Bc = items < chunk ? items : chunk;
    const size_t accw = (size_t)ext * Bc * 2 * 2048, n1 = (size_t)c->P.n + 1;
    static const char *kRemedy = "tfhe_ctx_reserve_extended(ctx, max_batch, ext)";
    int rc;
    if ((rc = grow(c, c->s_t2, 2 * accw * sizeof(uint32_t), st, "extended accumulator", kRemedy)) ||
        (rc = grow(c, c->s_t3, (size_t)Bc * n1 * sizeof(uint32_t), st, "mod-switched sample", kRemedy))) return rc;
    // the final sample extract + key switch runs on the accumulators themselves; its one-hot matrix is sized by reserve_scratch
    return reserve_scratch(c, Bc, false, st);
}

int bootstrap_extended_device(tfhe_ctx *c, const uint32_t *d_in, const uint32_t *d_lut, int lut_per_item, int ext,
                              uint32_t *d_out, int B, hipStream_t st)
{
    if (c->shape != kShapeN2048_L1_B22) return fail(TFHE_E_INVALID, "extended lookup tables need the N = 2048 parameter shape");
    if (ext < 1 || ext > 16) return fail(TFHE_E_INVALID, "polyExtendFactor %d out of range (1..16)", ext);
    const size_t N = 2048, n1 = (size_t)c->P.n + 1;
    int rc;
    if (ext == 2) {
        const int slab = slab_items(c);
        if ((rc = reserve_scratch(c, B, false, st))) return rc;
        for (int base = 0; base < B; base += slab) {
            const int S = B - base < slab ? B - base : slab;
            BlindRotateArgs a{};
            a.bsk = c->bsk.as<cd>(); a.tw = c->tw.as<cd>();
            a.in0 = d_in + (size_t)base * n1;
            a.tv = d_lut + (lut_per_item ? (size_t)base * ext * 2 * N : 0);
            a.tv_stride = lut_per_item ? (long)ext * 2 * N : 0;
            a.out = c->s_trlwe.as<uint32_t>();
            a.n = c->P.n; a.nsteps = c->P.n; a.Nbit = c->P.Nbit; a.offset = c->offset;
            a.op_uniform = -1;
            a.status = c->status.as<int>();
            hipEvent_t stop;
            if ((rc = timing_begin(c, 0, st, &stop))) return rc;
            launch_blind_rotate_ext2(a, S, st);
            HIP_TRY(hipGetLastError());
            if ((rc = timing_end(c, 0, st, stop))) return rc;
            if ((rc = launch_keyswitch(c, c->s_trlwe.as<uint32_t>(), d_out + (size_t)base * n1, S, nullptr, st))) return rc;
        }
        return TFHE_OK;
    }
    const int chunk = launch_items(c);                       // items per pass: the accumulators are 2 * ext * chunk TRLWE samples
    const int Bc = B < chunk ? B : chunk;
    const size_t accw = (size_t)ext * Bc * 2 * N;
    if ((rc = reserve_extended_scratch(c, B, ext, st))) return rc;
    uint32_t *acc[2] = {c->s_t2.as<uint32_t>(), c->s_t2.as<uint32_t>() + accw};
    for (int base = 0; base < B; base += chunk) {
        const int S = B - base < chunk ? B - base : chunk;
        ExtendedArgs a{};
        a.bsk = c->bsk.as<cd>(); a.tw = c->tw.as<cd>();
        a.in = d_in + (size_t)base * n1;
        a.lut = d_lut + (lut_per_item ? (size_t)base * ext * 2 * N : 0);
        a.lut_stride = lut_per_item ? (long)ext * 2 * N : 0;
        a.amod = c->s_t3.as<uint32_t>();
        a.n = c->P.n; a.ext = ext; a.B = S; a.offset = c->offset;
        a.acc_out = acc[0];
        hipEvent_t stop;
        int trc = timing_begin(c, 0, st, &stop);
        if (trc) return trc;
        hipLaunchKernelGGL(k_ext_init_2048, dim3(S), dim3(256), 0, st, a);
        for (int i = 0; i < c->P.n; i++) {
            a.step = i; a.acc_in = acc[i & 1]; a.acc_out = acc[(i + 1) & 1];
            hipLaunchKernelGGL((k_cmux_ext_2048<22>), dim3((unsigned)((size_t)S * ext)), dim3(128), 0, st, a);
        }
        HIP_TRY(hipGetLastError());
        if ((rc = timing_end(c, 0, st, stop))) return rc;
        // component 0 of the final accumulators is contiguous [S][2][N]: sample extract + key switch as usual
        if ((rc = launch_keyswitch(c, acc[c->P.n & 1], d_out + (size_t)base * n1, S, nullptr, st))) return rc;
    }
    return TFHE_OK;
}

// Host-pointer gate batch longer than one piece (pipe_items: 16,384 bootstraps at N = 1024): the operands of piece s+1
// go up and the results of piece s-1 come down on their own streams while the kernels of piece s run, through
// double-buffered staging of one piece each -- a Go caller can only hand over host memory, so for it this IS the
// throughput path.
int gate_batch_pipelined(tfhe_ctx *c, const uint8_t *ops, int op_uniform, const uint32_t *a, const uint32_t *b,
                         const uint32_t *cc, uint32_t *out, int B)
{
    const int S = pipe_items(c);
    const size_t n1 = (size_t)c->P.n + 1, slab_rows = (size_t)S * n1 * 4;
    int rc;
    if ((rc = c->s_in0.reserve(2 * slab_rows)) || (rc = c->s_in1.reserve(2 * slab_rows)) || (rc = c->s_out.reserve(2 * slab_rows))) return rc;
    if (cc && (rc = c->s_in2.reserve(2 * slab_rows))) return rc;
    if (ops && (rc = c->s_ops.reserve(2 * (size_t)S))) return rc;
    if ((rc = reserve_scratch(c, S, cc && (ops || op_uniform == TFHE_OP_MUX), c->stream))) return rc;
    if (!c->h2d_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->h2d_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
        for (auto &pr : c->pipe_ev)
            for (auto &e : pr) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));                    // earlier work of this context is done with the buffers
    const int ns = (B + S - 1) / S;
    auto dev = [&](DevBuf &d, int buf, size_t unit) { return static_cast<char *>(d.p) + (size_t)buf * unit; };
    for (int s = 0; s <= ns; s++) {
        const int buf = s & 1;
        if (s < ns) {
            const size_t base = (size_t)s * S, cnt = (size_t)(B - (int)base < S ? B - (int)base : S), bytes = cnt * n1 * 4;
            if (s >= 2) HIP_TRY(hipStreamWaitEvent(c->h2d_stream, c->pipe_ev[1][buf], 0));      // slab s-2's kernels have read this buffer
            HIP_TRY(hipMemcpyAsync(dev(c->s_in0, buf, slab_rows), a + base * n1, bytes, hipMemcpyHostToDevice, c->h2d_stream));
            HIP_TRY(hipMemcpyAsync(dev(c->s_in1, buf, slab_rows), b + base * n1, bytes, hipMemcpyHostToDevice, c->h2d_stream));
            if (cc) HIP_TRY(hipMemcpyAsync(dev(c->s_in2, buf, slab_rows), cc + base * n1, bytes, hipMemcpyHostToDevice, c->h2d_stream));
            if (ops) HIP_TRY(hipMemcpyAsync(dev(c->s_ops, buf, (size_t)S), ops + base, cnt, hipMemcpyHostToDevice, c->h2d_stream));
            HIP_TRY(hipEventRecord(c->pipe_ev[0][buf], c->h2d_stream));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->pipe_ev[0][buf], 0));
            if (s >= 2) HIP_TRY(hipStreamWaitEvent(c->stream, c->pipe_ev[2][buf], 0));           // slab s-2's results have left this buffer
            if ((rc = gate_batch_device(c, ops ? reinterpret_cast<const uint8_t *>(dev(c->s_ops, buf, (size_t)S)) : nullptr, op_uniform,
                                        reinterpret_cast<const uint32_t *>(dev(c->s_in0, buf, slab_rows)),
                                        reinterpret_cast<const uint32_t *>(dev(c->s_in1, buf, slab_rows)),
                                        cc ? reinterpret_cast<const uint32_t *>(dev(c->s_in2, buf, slab_rows)) : nullptr,
                                        reinterpret_cast<uint32_t *>(dev(c->s_out, buf, slab_rows)), (int)cnt, c->stream))) return rc;
            HIP_TRY(hipEventRecord(c->pipe_ev[1][buf], c->stream));
        }
        if (s >= 1) {                                            // results of slab s-1 (issued after slab s's uploads, so the
            const int pb = (s - 1) & 1;                          // host blocks on them while slab s computes)
            const size_t base = (size_t)(s - 1) * S, cnt = (size_t)(B - (int)base < S ? B - (int)base : S);
            HIP_TRY(hipStreamWaitEvent(c->d2h_stream, c->pipe_ev[1][pb], 0));
            HIP_TRY(hipMemcpyAsync(out + base * n1, dev(c->s_out, pb, slab_rows), cnt * n1 * 4, hipMemcpyDeviceToHost, c->d2h_stream));
            HIP_TRY(hipEventRecord(c->pipe_ev[2][pb], c->d2h_stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(c->d2h_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

// Host-pointer gate batches of 257 ... pipe_items gates (what a Go service submits from several goroutines: gates.Batch* on host memory):
// the three phases of a call -- operands up, kernels, results down -- hold three different locks, and a call works in one of two buffer
// slots, so the upload of caller B runs on the transfer stream while caller A's kernels run, and A's results go down while B computes.
// A lone caller goes through the same three phases back to back (two event waits more than the serial path).  The kernels themselves
// still run one call at a time under the context mutex (they share the scratch), so results are exactly the serial path's.
// An overlapped call that leaves on an error must not release its slot (or hand the caller's buffers back) while a copy or kernel of
// it is still in flight: drained here on every exit that did not reach the end.
struct SlotDrain {
    tfhe_ctx *c;
    bool armed = true;
    ~SlotDrain()
    {
        if (!armed) return;
        if (c->h2d_stream) (void)hipStreamSynchronize(c->h2d_stream);
        (void)hipStreamSynchronize(c->stream);
        if (c->d2h_stream) (void)hipStreamSynchronize(c->d2h_stream);
    }
};

// (called with up_mu held) the transfer streams and the slot's two events, made on first use
int overlap_prepare(tfhe_ctx *c, tfhe_ctx::HostSlot &S)
{
    if (!c->h2d_stream) {
        std::lock_guard<std::recursive_mutex> lk(c->mu);          // the pipelined path creates the same streams under this mutex
        if (!c->h2d_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&c->h2d_stream, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
            for (auto &pr : c->pipe_ev)
                for (auto &e : pr) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
    }
    if (!S.up) {
        HIP_TRY(hipEventCreateWithFlags(&S.up, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
    }
    return TFHE_OK;
}

int gate_batch_overlapped(tfhe_ctx *c, const uint8_t *ops, int op_uniform, const uint32_t *a, const uint32_t *b, const uint32_t *cc,
                          uint32_t *out, int B)
{
    int rc;
    HIP_TRY(hipSetDevice(c->device));
    tfhe_ctx::HostSlot &S = c->hslot[c->hticket.fetch_add(1, std::memory_order_relaxed) & 1];
    std::lock_guard<std::mutex> slot(S.mu);
    SlotDrain drain{c};
    const size_t rows = (size_t)B * (c->P.n + 1) * 4;
    {
        std::lock_guard<std::mutex> up(c->up_mu);
        if ((rc = overlap_prepare(c, S))) return rc;
        if ((rc = S.in0.reserve(rows)) || (rc = S.in1.reserve(rows)) || (rc = S.out.reserve(rows))) return rc;
        if (cc && (rc = S.in2.reserve(rows))) return rc;
        if (ops && (rc = S.ops.reserve((size_t)B))) return rc;
        HIP_TRY(hipMemcpyAsync(S.in0.p, a, rows, hipMemcpyHostToDevice, c->h2d_stream));
        HIP_TRY(hipMemcpyAsync(S.in1.p, b, rows, hipMemcpyHostToDevice, c->h2d_stream));
        if (cc) HIP_TRY(hipMemcpyAsync(S.in2.p, cc, rows, hipMemcpyHostToDevice, c->h2d_stream));
        if (ops) HIP_TRY(hipMemcpyAsync(S.ops.p, ops, (size_t)B, hipMemcpyHostToDevice, c->h2d_stream));
        HIP_TRY(hipEventRecord(S.up, c->h2d_stream));
    }
    {
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        HIP_TRY(hipStreamWaitEvent(c->stream, S.up, 0));
        if ((rc = gate_batch_device(c, ops ? S.ops.as<uint8_t>() : nullptr, op_uniform, S.in0.as<uint32_t>(), S.in1.as<uint32_t>(),
                                    cc ? S.in2.as<uint32_t>() : nullptr, S.out.as<uint32_t>(), B, c->stream))) return rc;
        HIP_TRY(hipEventRecord(S.done, c->stream));
    }
    {
        std::lock_guard<std::mutex> down(c->down_mu);
        HIP_TRY(hipStreamWaitEvent(c->d2h_stream, S.done, 0));
        HIP_TRY(hipMemcpyAsync(out, S.out.p, rows, hipMemcpyDeviceToHost, c->d2h_stream));
        HIP_TRY(hipStreamSynchronize(c->d2h_stream));
    }
    drain.armed = false;
    return TFHE_OK;
}

// The same for programmable bootstraps (evaluator.BootstrapLUT over host batches, one table or one per item).
int bootstrap_batch_overlapped(tfhe_ctx *c, const uint32_t *in, const uint32_t *tv, int tv_per_item, uint32_t *out, int B)
{
    int rc;
    HIP_TRY(hipSetDevice(c->device));
    tfhe_ctx::HostSlot &S = c->hslot[c->hticket.fetch_add(1, std::memory_order_relaxed) & 1];
    std::lock_guard<std::mutex> slot(S.mu);
    SlotDrain drain{c};
    const size_t rows = (size_t)B * (c->P.n + 1) * 4;
    const size_t tvb = tv ? (tv_per_item ? (size_t)B : 1) * 2 * c->P.N * 4 : 0;
    {
        std::lock_guard<std::mutex> up(c->up_mu);
        if ((rc = overlap_prepare(c, S))) return rc;
        if ((rc = S.in0.reserve(rows)) || (rc = S.out.reserve(rows)) || (tvb && (rc = S.tv.reserve(tvb)))) return rc;
        HIP_TRY(hipMemcpyAsync(S.in0.p, in, rows, hipMemcpyHostToDevice, c->h2d_stream));
        if (tv) HIP_TRY(hipMemcpyAsync(S.tv.p, tv, tvb, hipMemcpyHostToDevice, c->h2d_stream));
        HIP_TRY(hipEventRecord(S.up, c->h2d_stream));
    }
    {
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        HIP_TRY(hipStreamWaitEvent(c->stream, S.up, 0));
        if ((rc = bootstrap_device(c, S.in0.as<uint32_t>(), tv ? S.tv.as<uint32_t>() : nullptr, tv_per_item, S.out.as<uint32_t>(), B, c->stream))) return rc;
        HIP_TRY(hipEventRecord(S.done, c->stream));
    }
    {
        std::lock_guard<std::mutex> down(c->down_mu);
        HIP_TRY(hipStreamWaitEvent(c->d2h_stream, S.done, 0));
        HIP_TRY(hipMemcpyAsync(out, S.out.p, rows, hipMemcpyDeviceToHost, c->d2h_stream));
        HIP_TRY(hipStreamSynchronize(c->d2h_stream));
    }
    drain.armed = false;
    return TFHE_OK;
}

// The host-pointer gate batch of ONE caller: stage, launch, read back, under the context mutex.
int gate_batch_serial(tfhe_ctx *c, const uint8_t *ops, int op_uniform, const uint32_t *a, const uint32_t *b, const uint32_t *cc,
                      uint32_t *out, int B)
{
    int rc;
    HIP_TRY(hipSetDevice(c->device));        // here, not in tfhe_gate_batch: a caller that only FOLLOWS a combined launch never touches the HIP runtime
    if (B > c->num_cus && B <= pipe_items(c)) return gate_batch_overlapped(c, ops, op_uniform, a, b, cc, out, B);
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (B > pipe_items(c)) return gate_batch_pipelined(c, ops, op_uniform, a, b, cc, out, B);
    const size_t rows = (size_t)B * (c->P.n + 1) * 4;
    if ((rc = c->s_in0.reserve(rows)) || (rc = c->s_in1.reserve(rows)) || (rc = c->s_out.reserve(rows))) return rc;
    if (cc && (rc = c->s_in2.reserve(rows))) return rc;
    if (ops && (rc = c->s_ops.reserve(B))) return rc;
    HIP_TRY(hipMemcpyAsync(c->s_in0.p, a, rows, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->s_in1.p, b, rows, hipMemcpyHostToDevice, c->stream));
    if (cc) HIP_TRY(hipMemcpyAsync(c->s_in2.p, cc, rows, hipMemcpyHostToDevice, c->stream));
    if (ops) HIP_TRY(hipMemcpyAsync(c->s_ops.p, ops, B, hipMemcpyHostToDevice, c->stream));
    if ((rc = gate_batch_device(c, ops ? c->s_ops.as<uint8_t>() : nullptr, op_uniform, c->s_in0.as<uint32_t>(),
                                c->s_in1.as<uint32_t>(), cc ? c->s_in2.as<uint32_t>() : nullptr,
                                c->s_out.as<uint32_t>(), B, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, c->s_out.p, rows, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

// The host-pointer programmable bootstrap of ONE caller (evaluator.BootstrapLUT, programmable_bootstrap.go:93-115).
int bootstrap_batch_serial(tfhe_ctx *c, const uint32_t *in, const uint32_t *tv, int tv_per_item, uint32_t *out, int B)
{
    int rc;
    HIP_TRY(hipSetDevice(c->device));
    if (B > c->num_cus && B <= pipe_items(c)) return bootstrap_batch_overlapped(c, in, tv, tv_per_item, out, B);
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t inb = (size_t)B * (c->P.n + 1) * 4, trl = (size_t)B * 2 * c->P.N * 4;
    const size_t tvb = tv ? (tv_per_item ? trl : (size_t)2 * c->P.N * 4) : 0;
    if ((rc = c->s_in0.reserve(inb)) || (rc = c->s_out.reserve(inb)) || (tvb && (rc = c->s_tv.reserve(tvb)))) return rc;
    HIP_TRY(hipMemcpyAsync(c->s_in0.p, in, inb, hipMemcpyHostToDevice, c->stream));
    if (tv) HIP_TRY(hipMemcpyAsync(c->s_tv.p, tv, tvb, hipMemcpyHostToDevice, c->stream));
    if ((rc = bootstrap_device(c, c->s_in0.as<uint32_t>(), tv ? c->s_tv.as<uint32_t>() : nullptr, tv_per_item,
                               c->s_out.as<uint32_t>(), B, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, c->s_out.p, inb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

// Items one combined launch may carry, in BOOTSTRAPS.  At the N = 1024, L = 3, Bgbit = 6 shape (80 / 110 / 128-bit sets) every
// transform is exact: whatever kernel runs an item, its words are the same, so a launch may be as long as its
// staging (one full launch's worth of rows).  At every other shape (tolerance regime) the kernels of different launch shapes round
// differently -- gates included: the gate test vector goes through the same transforms -- so a combined launch stays within the kernel
// shape a lone small call runs: at most one bootstrap per CU AND within the four-/eight-wave limits of the context (TFHE_OPT_QUAD_MAX /
// TFHE_OPT_OCT_MAX overrides move those).  Requests heavier than the cap are not combined at all.
int combine_cap(const tfhe_ctx *c, int kind)
{
    (void)kind;
    if (c->shape == kShapeN1024_L3_B6) return 2 * c->num_cus;            // rows are bounded by the staging (one per CU); a MUX row is two bootstraps
    int cap = c->num_cus;
    if (shape_is_1024(c->shape)) {              // the small-launch kernels exist at the N = 1024 shapes only
        const int q = c->quad_limit.load(), o = c->oct_limit.load();
        if (q > 0 && q < cap) cap = q;
        if (o > 0 && o < cap) cap = o;
    }
    return cap < 1 ? 1 : cap;
}

// Bootstraps of the widest blind-rotate request a gate call puts into a launch: a row that may be a MUX adds its ANDNY(a, c) to pass 1
// (gate_batch_device: S + Mx items), so it weighs two (counting rows let 200 single-MUX callers form a 400-bootstrap
// launch -- a different kernel shape than the lone call's).
int combine_weight(const tfhe_ctx::GateReq &r)
{
    const bool may_mux = r.kind == 0 && r.cc && (r.ops || r.op_uniform == TFHE_OP_MUX);
    return may_mux ? 2 * r.B : r.B;
}

// Rows of one combined launch = rows of its page-locked staging: one per CU.  That is the scalar-caller case combining exists for (a
// launch of 1 ... one bootstrap per CU costs the same ~2.2 ms); larger staging only bought the merging of mid-size batches and cost
// 9 ms of hipHostMalloc at the first contention -- inside which the other callers went one by one (round 6 trace: 34 MB for 1,024 rows;
// 8.6 MB for 256 rows is 2.3 ms).  Requests of more rows take the serial path.
int combine_rows(const tfhe_ctx *c, int kind)
{
    const int cap = combine_cap(c, kind);
    return cap < c->num_cus ? cap : c->num_cus;
}

long long steady_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void futex_wait(std::atomic<uint32_t> &w, uint32_t seen, long timeout_us = 0)
{
    struct timespec ts{0, timeout_us * 1000};
    syscall(SYS_futex, reinterpret_cast<uint32_t *>(&w), FUTEX_WAIT_PRIVATE, seen, timeout_us ? &ts : nullptr, nullptr, 0);
}

void futex_wake_all(std::atomic<uint32_t> &w)
{
    syscall(SYS_futex, reinterpret_cast<uint32_t *>(&w), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}

// Page-locked staging of the three batches of a queue, allocated by the first caller that is about to FOLLOW another one (a context
// that only ever sees lone callers never allocates it).  Sized for the parameter shape's largest combined launch (combine_rows at
// the default limits: set_option may only lower the bootstraps per launch afterwards, never the row capacity).
int comb_ensure_staging(tfhe_ctx *c, int kind)
{
    tfhe_ctx::CombQueue &Q = c->comb[kind];
    if (Q.ready.load(std::memory_order_acquire)) return TFHE_OK;
    std::lock_guard<std::mutex> lk(Q.alloc_mu);
    if (Q.ready.load(std::memory_order_relaxed)) return TFHE_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t rows = (size_t)Q.cap_rows, n1 = (size_t)c->P.n + 1;
    const size_t bytes = kind == 0 ? 4 * rows * n1 * 4 + rows : 2 * rows * n1 * 4 + rows * 2 * c->P.N * 4;
    try {
        for (auto &b : Q.ring) b.reqs.assign(rows, nullptr);
    } catch (...) {
        return fail(TFHE_E_NOMEM, "out of host memory for the combining queue");
    }
    for (auto &b : Q.ring) {
        if (b.host) continue;
        void *h = nullptr;
        hipError_t e = hipHostMalloc(&h, bytes, hipHostMallocDefault);
        if (e != hipSuccess) return fail(TFHE_E_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        b.host = static_cast<char *>(h);
    }
    Q.ready.store(true, std::memory_order_release);
    return TFHE_OK;
}

struct CombPlanes {                 // where a batch's planes live in its staging
    char *a, *b, *c, *out, *tv;
    uint8_t *ops;
};
CombPlanes comb_planes(const tfhe_ctx *c, const tfhe_ctx::CombQueue &Q, const tfhe_ctx::CombBatch &bt, int kind)
{
    const size_t plane = (size_t)Q.cap_rows * ((size_t)c->P.n + 1) * 4;
    CombPlanes p{};
    p.a = bt.host;
    if (kind == 0) {
        p.b = p.a + plane; p.c = p.b + plane; p.out = p.c + plane;
        p.ops = reinterpret_cast<uint8_t *>(p.out + plane);
    } else {
        p.out = p.a + plane; p.tv = p.out + plane;
    }
    return p;
}

// A caller's operands into its rows of the batch's staging (done by the caller itself: 256 callers copy in parallel while the
// previous launch's other callers are still waking up, instead of one leader copying 256 requests in sequence).
void comb_copy_in(const tfhe_ctx *c, const tfhe_ctx::CombQueue &Q, tfhe_ctx::CombBatch &bt, const tfhe_ctx::GateReq &r)
{
    const size_t n1b = ((size_t)c->P.n + 1) * 4, at = (size_t)r.pos, bytes = (size_t)r.B * n1b;
    const CombPlanes p = comb_planes(c, Q, bt, r.kind);
    memcpy(p.a + at * n1b, r.a, bytes);
    if (r.kind == 0) {
        memcpy(p.b + at * n1b, r.b, bytes);
        if (r.cc) { memcpy(p.c + at * n1b, r.cc, bytes); bt.any_c.store(true, std::memory_order_relaxed); }
        else memset(p.c + at * n1b, 0, bytes);                           // never read by the kernels (none of this request's ops is MUX); defined all the same
        if (r.ops) memcpy(p.ops + at, r.ops, (size_t)r.B);
        else memset(p.ops + at, r.op_uniform, (size_t)r.B);
    } else {
        const size_t twb = (size_t)2 * c->P.N * 4;
        const uint32_t *tv = r.b ? r.b : c->gate_tv_host.data();
        if (r.b && r.op_uniform) memcpy(p.tv + at * twb, tv, (size_t)r.B * twb);
        else for (int k = 0; k < r.B; k++) memcpy(p.tv + (at + (size_t)k) * twb, tv, twb);
    }
}

// ONE launch for the `rows` rows of a closed batch: one transfer per operand plane, every item with its own op code (gates) or its own
// table (bootstraps), results into the staging's out plane.  A gate's (a bootstrap's) result depends on its own operands only -- not
// on its position in a batch, nor, within the limits of combine_cap, on the batch's size -- so every caller receives exactly the words
// the serial path would have given it (tests: batch-position invariance, combined == lone).
int run_combined(tfhe_ctx *c, tfhe_ctx::CombQueue &Q, tfhe_ctx::CombBatch &bt, int kind, int rows)
{
    int rc;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t n1 = (size_t)c->P.n + 1, plane = (size_t)rows * n1 * 4;
    const CombPlanes p = comb_planes(c, Q, bt, kind);
    if (kind == 0) {
        const bool any_c = bt.any_c.load(std::memory_order_relaxed);
        // (Measured and dropped, profiles/r06_c_combine.txt: letting the kernels of a small launch read the operand rows straight from the
        // page-locked staging instead of making these transfers -- the launch got 0.14 ms LONGER, 2.51 vs 2.37 ms at 256 rows.)
        if ((rc = c->s_in0.reserve(plane)) || (rc = c->s_in1.reserve(plane)) || (rc = c->s_out.reserve(plane))) return rc;
        if (any_c && (rc = c->s_in2.reserve(plane))) return rc;
        if ((rc = c->s_ops.reserve((size_t)rows))) return rc;
        HIP_TRY(hipMemcpyAsync(c->s_in0.p, p.a, plane, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->s_in1.p, p.b, plane, hipMemcpyHostToDevice, c->stream));
        if (any_c) HIP_TRY(hipMemcpyAsync(c->s_in2.p, p.c, plane, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->s_ops.p, p.ops, (size_t)rows, hipMemcpyHostToDevice, c->stream));
        if ((rc = gate_batch_device(c, c->s_ops.as<uint8_t>(), 0, c->s_in0.as<uint32_t>(), c->s_in1.as<uint32_t>(),
                                    any_c ? c->s_in2.as<uint32_t>() : nullptr, c->s_out.as<uint32_t>(), rows, c->stream))) return rc;
    } else {
        const size_t tabs = (size_t)rows * 2 * c->P.N * 4;
        if ((rc = c->s_in0.reserve(plane)) || (rc = c->s_out.reserve(plane)) || (rc = c->s_tv.reserve(tabs))) return rc;
        HIP_TRY(hipMemcpyAsync(c->s_in0.p, p.a, plane, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->s_tv.p, p.tv, tabs, hipMemcpyHostToDevice, c->stream));
        if ((rc = bootstrap_device(c, c->s_in0.as<uint32_t>(), c->s_tv.as<uint32_t>(), 1, c->s_out.as<uint32_t>(), rows, c->stream))) return rc;
    }
    HIP_TRY(hipMemcpyAsync(p.out, c->s_out.p, plane, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

// Flat combining (the reference's concurrency is goroutine fan-out over pooled evaluators, trgsw.go:227-252; its scalar gates.* serialise
// on one evaluator, gates.go:19-23,136-142).  A launch of 1 ... 256 bootstraps costs the same ~2.2 ms, so N threads issuing scalar gates
// one launch each would take N x 2.2 ms.  Instead (round 6: lock-free, every caller moves its own rows):
//   * a caller CLAIMS rows of the open batch's page-locked staging with one compare-and-swap -- no mutex: 255 callers coming back from
//     a launch within ~150 us formed a convoy on the queue mutex -- copies its operands in itself and sleeps on that batch's `done` word;
//   * the caller that claimed row 0 is the batch's LEADER.  It waits for the previous batch's launch, then for the callers that launch
//     carried to come back (`returning`: every claim counts one down; the wait ends when it reaches zero, when nobody has arrived for
//     150 us, when the batch is full, or after 600 us -- round 5 polled with 20 us sleeps for at most 200 us and compared against the
//     PREVIOUS launch's size only, so two cohorts formed at thread start-up never merged and took turns: 49-53 launches for 40 rounds);
//     closes the batch, opens the next one (arrivals from now on queue there), issues ONE launch, and wakes its followers through the
//     batch's own futex word (one system call, 0.55 us per sleeper measured on the GPU box; tools/ubench_wake.cpp);
//   * every follower takes its rows out of the staging itself; the slot is re-opened only after the last one has.
// A lone caller claims row 0 of an empty batch, finds nothing to wait for and runs the serial path on its own pointers: a few atomic
// operations more than round 5, no staging.  If a combined launch fails for a reason only the combination has (scratch a frozen context
// may not grow, ...), its requests are re-issued one by one from the staging, each with its own result.
#ifdef TFHE_TSAN_CONTROL
long g_tsan_control;         // tools/asan_host_check.sh control thread: a deliberate unsynchronised counter the ThreadSanitizer build must report
#endif
int combine_request(tfhe_ctx *c, tfhe_ctx::GateReq &me)
{
#ifdef TFHE_TSAN_CONTROL
    g_tsan_control++;
#endif
    using Batch = tfhe_ctx::CombBatch;
    tfhe_ctx::CombQueue &Q = c->comb[me.kind];
    auto serial = [&](const tfhe_ctx::GateReq &r, const uint8_t *ops, const uint32_t *a, const uint32_t *b, const uint32_t *cc, uint32_t *out) {
        return r.kind == 0 ? gate_batch_serial(c, ops, ops ? 0 : r.op_uniform, a, b, cc, out, r.B)
                           : bootstrap_batch_serial(c, a, b, r.op_uniform, out, r.B);
    };
    const int w = combine_weight(me), cap_w = combine_cap(c, me.kind), cap_rows = Q.cap_rows;
    if (w > cap_w || me.B > cap_rows) return serial(me, me.ops, me.a, me.b, me.cc, me.out);     // limits lowered since the caller's pre-check
    // ---- claim rows of the open batch
    int bi;
    uint64_t cl;
    for (;;) {
        const uint32_t og = Q.open_gen.load(std::memory_order_acquire);
        bi = Q.open.load(std::memory_order_acquire);
        Batch &bt = Q.ring[bi];
        cl = bt.claim.load(std::memory_order_acquire);
        const int rows = (int)(cl & (Batch::kClosed - 1)), wsum = (int)(cl >> 32);
        if ((cl & Batch::kClosed) || rows + me.B > cap_rows || wsum + w > cap_w) {
            if (!(cl & Batch::kClosed) && !bt.full.exchange(true)) {            // full: its leader need not wait for anybody else
                Q.gather.fetch_add(1, std::memory_order_release);
                futex_wake_all(Q.gather);
            }
            futex_wait(Q.open_gen, og, 2000);                 // until the next batch opens (bounded: a missed wake-up costs 2 ms, never a hang)
            continue;
        }
        if (rows > 0) {                                       // about to follow somebody: the staging must exist
            if (comb_ensure_staging(c, me.kind)) return serial(me, me.ops, me.a, me.b, me.cc, me.out);      // no page-locked memory: uncombined
        }
        if (bt.claim.compare_exchange_weak(cl, cl + (uint64_t)me.B + ((uint64_t)w << 32), std::memory_order_acq_rel)) break;
    }
    Batch &bt = Q.ring[bi];
    me.pos = (int)(cl & (Batch::kClosed - 1));
    if (Q.returning.load(std::memory_order_relaxed) > 0 && Q.returning.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        Q.gather.fetch_add(1, std::memory_order_release);     // the last caller of the previous launch is back: the gathering leader may go
        futex_wake_all(Q.gather);
    }
    if (me.pos > 0) {
        // ---- follower: operands in, sleep until the batch is done, rows out
        comb_copy_in(c, Q, bt, me);
        bt.reqs[(size_t)bt.nreq.fetch_add(1, std::memory_order_acq_rel)] = &me;
        bt.filled.fetch_add(me.B, std::memory_order_release);
        while (bt.done.load(std::memory_order_acquire) == 0) futex_wait(bt.done, 0);
        const int rc = me.rc;
        if (rc) { try { g_err = me.err; } catch (...) {} }
        else {
            const size_t n1b = ((size_t)c->P.n + 1) * 4;
            memcpy(me.out, comb_planes(c, Q, bt, me.kind).out + (size_t)me.pos * n1b, (size_t)me.B * n1b);
        }
        bt.readers.fetch_sub(1, std::memory_order_release);   // nothing of the batch is touched after this
        return rc;
    }
    // ---- leader of batch bi.  Nothing here may throw past this function (the ABI never throws, and every follower sleeps until the
    // batch is settled): the only allocations are error strings, guarded below.
    Batch &prev = Q.ring[(bi + 2) % 3];
    while (prev.done.load(std::memory_order_acquire) == 0) futex_wait(prev.done, 0);            // one launch in flight at a time
    // gathering wait: the callers the previous launch carried are waking up this very moment and will be back within ~0.2 ms; launching
    // without them makes two cohorts that take turns (each launch half as full as it could be at the same cost)
    long long ns_gather = 0;
    const long long prev_done_ns = Q.last_done_ns.load(std::memory_order_relaxed);
    if (Q.returning.load(std::memory_order_acquire) <= 0) {
        c->comb_exit[0]++;
    } else {
        const long long t_in = steady_ns();
        if (t_in - prev_done_ns > 1000000) {
            Q.returning.store(0, std::memory_order_relaxed);                                     // stale: those callers went elsewhere long ago
            c->comb_exit[1]++;
        } else {
            // the first kGatherSpinNs busy-polling (the GPU is idle and every caller of the context is waiting for this launch: a sleeping
            // leader adds a second thread wake-up -- 50-100 us out of an idle core -- to every round), then asleep on the futex word
            constexpr long long kGatherSpinNs = 120000;
            auto gathered = [&]() { return Q.returning.load(std::memory_order_acquire) <= 0 || bt.full.load(std::memory_order_acquire); };
            bool done = gathered();
            while (!done && steady_ns() - t_in < kGatherSpinNs) {
                for (int i = 0; i < 32; i++) __builtin_ia32_pause();
                done = gathered();
            }
            int why = 0;
            while (!done) {
                const uint32_t g = Q.gather.load(std::memory_order_acquire);
                const uint64_t seen = bt.claim.load(std::memory_order_acquire);
                if (gathered()) break;
                const long quiet_us = c->comb_quiet_us.load(std::memory_order_relaxed);
                futex_wait(Q.gather, g, quiet_us);
                if (gathered()) break;
                if (bt.claim.load(std::memory_order_acquire) == seen) { why = 4; break; }             // quiet for one window
                if (steady_ns() - t_in > 4000LL * quiet_us) { why = 5; break; }                      // four windows in all
            }
            if (!why) why = bt.full.load(std::memory_order_acquire) ? 3 : 2;
            c->comb_exit[why]++;
            Q.returning.store(0, std::memory_order_relaxed);
            ns_gather = steady_ns() - t_in;
        }
    }
    // close; open the next batch (its slot was the one before the previous: every follower of that launch has long taken its rows)
    const uint64_t closed = bt.claim.fetch_or(Batch::kClosed, std::memory_order_acq_rel);
    const int rows = (int)(closed & (Batch::kClosed - 1));
    {
        const int ni = (bi + 1) % 3;
        Batch &nb = Q.ring[ni];
        while (nb.readers.load(std::memory_order_acquire) > 0) std::this_thread::yield();
        nb.filled.store(0, std::memory_order_relaxed); nb.nreq.store(0, std::memory_order_relaxed);
        nb.any_c.store(false, std::memory_order_relaxed); nb.full.store(false, std::memory_order_relaxed);
        nb.done.store(0, std::memory_order_relaxed);
        nb.claim.store(0, std::memory_order_release);
        Q.open.store(ni, std::memory_order_release);
        Q.open_gen.fetch_add(1, std::memory_order_release);
        futex_wake_all(Q.open_gen);
    }
    int rc = TFHE_OK;
    std::string err;
    int nfollow = 0;
    if (rows == me.B) {
        rc = serial(me, me.ops, me.a, me.b, me.cc, me.out);          // alone: the caller's own pointers, no staging
        if (rc) { try { err = g_err; } catch (...) {} }
    } else {
        comb_copy_in(c, Q, bt, me);
        for (int spins = 0; bt.filled.load(std::memory_order_acquire) < rows - me.B; spins++)      // followers still copying (microseconds)
            if (spins > 64) std::this_thread::yield();
        nfollow = bt.nreq.load(std::memory_order_acquire);
        const long long t_issue = steady_ns();
        rc = run_combined(c, Q, bt, me.kind, rows);
        c->comb_ns_launch += steady_ns() - t_issue;
        if (t_issue - prev_done_ns < 5000000) { c->comb_ns_idle += t_issue - prev_done_ns; c->comb_ns_gather += ns_gather; }      // back-to-back rounds only
        const CombPlanes p = comb_planes(c, Q, bt, me.kind);
        const size_t n1b = ((size_t)c->P.n + 1) * 4;
        if (rc == TFHE_OK) {
            c->comb_launches++;
            c->comb_requests += nfollow + 1;
            memcpy(me.out, p.out, (size_t)me.B * n1b);
        } else {
            // A combined launch can fail for a reason that only the COMBINATION has -- the sum of the requests needs buffers a frozen
            // context (captured hipGraph) may not grow -- while each request on its own would succeed.  The header promises every caller
            // what a lone call returns: re-issue the requests one by one from the staging, each with its own result.
            auto redo = [&](tfhe_ctx::GateReq &r, uint32_t *out) {
                const size_t at = (size_t)r.pos;
                if (r.kind == 0) return serial(r, p.ops + at, (const uint32_t *)(p.a + at * n1b), (const uint32_t *)(p.b + at * n1b),
                                               r.cc ? (const uint32_t *)(p.c + at * n1b) : nullptr, out);
                tfhe_ctx::GateReq t{1, nullptr, 1, nullptr, nullptr, nullptr, nullptr, r.B};
                return serial(t, nullptr, (const uint32_t *)(p.a + at * n1b), (const uint32_t *)(p.tv + at * (size_t)2 * c->P.N * 4), nullptr, out);
            };
            rc = redo(me, me.out);
            if (rc) { try { err = g_err; } catch (...) {} }
            for (int i = 0; i < nfollow; i++) {
                tfhe_ctx::GateReq &r = *bt.reqs[(size_t)i];
                r.rc = redo(r, (uint32_t *)(p.out + (size_t)r.pos * n1b));
                if (r.rc) { try { r.err = g_err; } catch (...) {} }
            }
        }
    }
#ifdef TFHE_COMBINE_TRACE
    std::fprintf(stderr, "[combine] t=%lld us batch=%d rows=%d followers=%d gather_us=%lld tid=%ld\n", steady_ns() / 1000 % 100000000, bi, rows, nfollow,
                 ns_gather / 1000, (long)syscall(SYS_gettid));
#endif
    // settle: followers may read their request and their rows from here on
    bt.readers.store(nfollow, std::memory_order_relaxed);
    Q.last_done_ns.store(steady_ns(), std::memory_order_relaxed);
    Q.returning.store(nfollow + 1, std::memory_order_release);
    bt.done.store(1, std::memory_order_release);
    futex_wake_all(bt.done);      // ONE system call for all sleepers (0.55 us each on the GPU box).  Measured and dropped (profiles/r06_c_combine.txt):
                                  // dealing the sleepers over four words with one relay waker per word -- every relay hop costs a thread wake-up, 134 vs 122 ms
    if (rc) { try { g_err = err; } catch (...) {} }
    return rc;
}

// Reads and clears the device status word (after the caller has synchronised the stream the work ran on).
int check_status(tfhe_ctx *c)
{
    int st = 0;
    HIP_TRY(hipMemcpy(&st, c->status.p, sizeof st, hipMemcpyDeviceToHost));
    if (!st) return TFHE_OK;
    HIP_TRY(hipMemset(c->status.p, 0, sizeof st));
    return fail(TFHE_E_INVALID, "an earlier gate batch carried an op code outside TFHE_OP_NAND..TFHE_OP_MUX "
                                "(or a MUX without a third operand); those items ran as plain bootstraps of their first operand");
}

} // namespace

extern "C" {

const char *tfhe_last_error(void) { return g_err.c_str(); }

// "release", or the name of the deliberate-defect control this object was compiled with (tests/fuzz_gpu.py's and the ThreadSanitizer
// script's positive controls: they return WRONG ciphertexts / race on purpose).  Loaders refuse anything but "release" unless a test
// opts in, so a stray -D in a packaging build cannot ship silently.
const char *tfhe_build_flavor(void)
{
#if defined(TFHE_FUZZ_CONTROL)
#warning "TFHE_FUZZ_CONTROL: this build returns one wrong bit on purpose (fuzzer positive control) -- never ship it"
    return "control:fuzz";
#elif defined(TFHE_TSAN_CONTROL)
#warning "TFHE_TSAN_CONTROL: this build contains a deliberate data race (ThreadSanitizer positive control) -- never ship it"
    return "control:tsan";
#else
    return "release";
#endif
}

int tfhe_device_count(int *count)
{
    if (!count) return fail(TFHE_E_INVALID, "null count");
    HIP_TRY(hipGetDeviceCount(count));
    return TFHE_OK;
}

int tfhe_ctx_create(const tfhe_params *P, int device_id, tfhe_ctx **out)
{
    if (!P || !out) return fail(TFHE_E_INVALID, "null argument");
    int shape = 0;
    if (P->N == 1024 && P->Nbit == 10 && P->L == 3 && P->Bgbit == 6) shape = kShapeN1024_L3_B6;
    if (P->N == 1024 && P->Nbit == 10 && P->L == 2 && P->Bgbit == 10) shape = kShapeN1024_L2_B10;
    if (P->N == 1024 && P->Nbit == 10 && P->L == 1 && P->Bgbit == 23) shape = kShapeN1024_L1_B23;
    if (P->N == 2048 && P->Nbit == 11 && P->L == 1 && P->Bgbit == 22) shape = kShapeN2048_L1_B22;
    if (P->N == 512 && P->Nbit == 9 && P->L == 1 && P->Bgbit == 18) shape = kShapeN512_L1_B18;
    if (!shape)
        return fail(TFHE_E_INVALID,
                    "unsupported parameter shape N=%d L=%d Bgbit=%d (supported: N=1024 with (L,Bgbit) = (3,6), (2,10), (1,23); "
                    "N=2048 with (1,22); N=512 with (1,18))", P->N, P->L, P->Bgbit);
    if (P->n < 1 || P->n >= kMaxLweDim) return fail(TFHE_E_INVALID, "LWE dimension %d out of range", P->n);
    if (P->basebit < 1 || P->t < 1 || P->basebit * P->t > 31 || (size_t)P->N * P->t > 9216)
        return fail(TFHE_E_INVALID, "unsupported key-switch shape basebit=%d t=%d", P->basebit, P->t);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(TFHE_E_INVALID, "device %d not present (%d visible)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    // every failure path below releases what was created so far (tfhe_ctx_destroy is null-tolerant)
    struct Guard {
        tfhe_ctx *c;
        ~Guard() { if (c) tfhe_ctx_destroy(c); }
    } guard{new tfhe_ctx};
    tfhe_ctx *c = guard.c;
    c->P = *P; c->device = device_id; c->shape = shape;
    c->n1p = (P->n + 1 + 31) & ~31;      // whole 128-byte lines: see tfhe_ctx::n1p
    for (int i = 0; i < P->L; i++) c->offset += (1u << (P->Bgbit - 1)) * (1u << (32 - (i + 1) * P->Bgbit));
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device_id));
        c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // kernel-dispatch limits: defaults here, tfhe_ctx_set_option for measurements and tests -- never the environment
    c->quad_limit = c->num_cus;
    c->oct_limit = c->num_cus;
    c->ks_mfma_min = kKsMfmaMinDefault;
    c->combine_max = c->num_cus;           // requests of up to one bootstrap per CU are combined (tfhe_gate_batch): the staging's rows
    for (auto &e : c->comb_exit) e.store(0);
    for (int kind = 0; kind < 2; kind++) {  // the combining queues: batch 0 open, the other two idle (closed, done)
        c->comb[kind].cap_rows = combine_rows(c, kind);
        c->comb[kind].ring[0].done.store(0);
        c->comb[kind].ring[0].claim.store(0);
    }
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto &pair : c->ev)
        for (auto &e : pair) HIP_TRY(hipEventCreate(&e));
    std::vector<cd> tw = make_twiddles(P->N);
    int rc;
    if ((rc = c->tw.reserve(tw.size() * sizeof(cd)))) return rc;
    HIP_TRY(hipMemcpy(c->tw.p, tw.data(), tw.size() * sizeof(cd), hipMemcpyHostToDevice));
    if (shape_is_1024(shape)) {
        std::vector<cd> twq = make_twiddles_quad();
        if ((rc = c->twq.reserve(twq.size() * sizeof(cd)))) return rc;
        HIP_TRY(hipMemcpy(c->twq.p, twq.data(), twq.size() * sizeof(cd), hipMemcpyHostToDevice));
    }
    if ((rc = c->status.reserve(sizeof(int)))) return rc;
    HIP_TRY(hipMemset(c->status.p, 0, sizeof(int)));
    std::vector<uint32_t> tv(2 * (size_t)P->N, 0u);              // cloudkey.go:74-85
    for (int j = 0; j < P->N; j++) tv[P->N + j] = 0x20000000u;
    if ((rc = c->gate_tv.reserve(tv.size() * sizeof(uint32_t)))) return rc;
    HIP_TRY(hipMemcpy(c->gate_tv.p, tv.data(), tv.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->gate_tv_host = std::move(tv);
    guard.c = nullptr;
    *out = c;
    return TFHE_OK;
}

int tfhe_ctx_destroy(tfhe_ctx *c)
{
    if (!c) return TFHE_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &m : c->dev_marks) { (void)hipEventSynchronize(m.ev); (void)hipEventDestroy(m.ev); }      // "_dev" work on caller streams
    if (c->need_sync_all) (void)hipDeviceSynchronize();
    c->dev_marks.clear();
    for (DevBuf *b : {&c->bsk, &c->bskq, &c->twq, &c->status, &c->s_plan, &c->ksk, &c->tw, &c->gate_tv, &c->s_in0, &c->s_in1, &c->s_in2, &c->s_out, &c->s_trlwe,
                      &c->s_tv, &c->s_ops, &c->s_idx, &c->s_t0, &c->s_t1, &c->s_t2, &c->s_t3, &c->kskB, &c->s_onehot, &c->s_gsw_raw, &c->s_gsw})
        b->release();
    for (auto &pair : c->ev)
        for (auto &e : pair) if (e) (void)hipEventDestroy(e);
    for (auto &v : c->tev)
        for (auto &pr : v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto &Q : c->comb)
        for (auto &b : Q.ring) if (b.host) (void)hipHostFree(b.host);
    for (auto &S : c->hslot) {
        for (DevBuf *b : {&S.in0, &S.in1, &S.in2, &S.out, &S.ops, &S.tv}) b->release();
        for (hipEvent_t e : {S.up, S.done}) if (e) (void)hipEventDestroy(e);
    }
    for (void *h : c->hdr_host) if (h) (void)hipHostFree(h);
    for (auto &pr : c->pipe_ev)
        for (auto &e : pr) if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {c->h2d_stream, c->d2h_stream}) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return TFHE_OK;
}

int tfhe_ctx_params(const tfhe_ctx *c, tfhe_params *out)
{
    if (!c || !out) return fail(TFHE_E_INVALID, "null argument");
    *out = c->P;
    return TFHE_OK;
}

int tfhe_ctx_sync(tfhe_ctx *c)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto &m : c->dev_marks) HIP_TRY(hipEventSynchronize(m.ev));
    if (c->need_sync_all) {                                  // some stream's mark could not be recorded (mark_dev_stream)
        HIP_TRY(hipDeviceSynchronize());
        c->need_sync_all = false;
    }
    return check_status(c);
}

int tfhe_ctx_set_option(tfhe_ctx *c, int option, int value)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    switch (option) {
    case TFHE_OPT_QUAD_MAX: c->quad_limit = value < 0 ? c->num_cus : value; return TFHE_OK;
    case TFHE_OPT_OCT_MAX: c->oct_limit = value < 0 ? c->num_cus : value; return TFHE_OK;
    case TFHE_OPT_KS_MFMA_MIN:
        c->ks_mfma_min = value < 0 ? kKsMfmaMinDefault : value;
        if (c->ks_mfma_min > 0 && c->have_ksk && !c->kskB.p) {      // the byte-column key copy was skipped at load time
            if ((rc = make_mfma_ksk(c, c->stream))) return rc;
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        return TFHE_OK;
    case TFHE_OPT_FROZEN: c->frozen = value != 0; return TFHE_OK;
    case TFHE_OPT_COMBINE_MAX: c->combine_max = value < 0 ? c->num_cus : value; return TFHE_OK;
    case TFHE_OPT_COMBINE_QUIET_US: c->comb_quiet_us = value < 0 ? kCombQuietUsDefault : value < 20 ? 20 : value > 5000 ? 5000 : value; return TFHE_OK;
    case TFHE_OPT_CLONE_FORCE_HOST: c->clone_force_host = value != 0; return TFHE_OK;
    case TFHE_OPT_KS_WIDE_CT:
        if (value > 0 && value != 64 && value != 128) return fail(TFHE_E_INVALID, "TFHE_OPT_KS_WIDE_CT is 64, 128 or 0 / -1 (by batch size)");
        c->ks_wide_ct = value < 0 ? 0 : value;
        return TFHE_OK;
    default: return fail(TFHE_E_INVALID, "unknown option %d", option);
    }
}

int tfhe_ctx_get_option(tfhe_ctx *c, int option, int *value)
{
    if (!c || !value) return fail(TFHE_E_INVALID, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    switch (option) {
    case TFHE_OPT_QUAD_MAX: *value = c->quad_limit; return TFHE_OK;
    case TFHE_OPT_OCT_MAX: *value = c->oct_limit; return TFHE_OK;
    case TFHE_OPT_KS_MFMA_MIN: *value = c->ks_mfma_min; return TFHE_OK;
    case TFHE_OPT_FROZEN: *value = c->frozen ? 1 : 0; return TFHE_OK;
    case TFHE_OPT_COMBINE_MAX: *value = c->combine_max.load(); return TFHE_OK;
    case TFHE_OPT_COMBINE_QUIET_US: *value = c->comb_quiet_us.load(); return TFHE_OK;
    case TFHE_OPT_KS_WIDE_CT: *value = c->ks_wide_ct; return TFHE_OK;
    case TFHE_OPT_CLONE_PATH: *value = c->clone_path; return TFHE_OK;
    case TFHE_OPT_CLONE_FORCE_HOST: *value = c->clone_force_host ? 1 : 0; return TFHE_OK;
    // the counters are 64-bit; the option interface is int: saturate instead of wrapping
    case TFHE_OPT_COMBINE_LAUNCHES: *value = (int)std::min<long long>(c->comb_launches.load(), INT_MAX); return TFHE_OK;
    case TFHE_OPT_COMBINE_REQUESTS: *value = (int)std::min<long long>(c->comb_requests.load(), INT_MAX); return TFHE_OK;
    case TFHE_OPT_COMBINE_US_IDLE: *value = (int)std::min<long long>(c->comb_ns_idle.load() / 1000, INT_MAX); return TFHE_OK;
    case TFHE_OPT_COMBINE_US_GATHER: *value = (int)std::min<long long>(c->comb_ns_gather.load() / 1000, INT_MAX); return TFHE_OK;
    case TFHE_OPT_COMBINE_US_LAUNCH: *value = (int)std::min<long long>(c->comb_ns_launch.load() / 1000, INT_MAX); return TFHE_OK;
    case TFHE_OPT_COMBINE_EXIT_NONE: case TFHE_OPT_COMBINE_EXIT_NONE + 1: case TFHE_OPT_COMBINE_EXIT_NONE + 2: case TFHE_OPT_COMBINE_EXIT_NONE + 3:
    case TFHE_OPT_COMBINE_EXIT_NONE + 4: case TFHE_OPT_COMBINE_EXIT_NONE + 5:
        *value = (int)std::min<long long>(c->comb_exit[option - TFHE_OPT_COMBINE_EXIT_NONE].load(), INT_MAX); return TFHE_OK;
    default: return fail(TFHE_E_INVALID, "unknown option %d", option);
    }
}

int tfhe_ctx_reserve(tfhe_ctx *c, int max_batch, int with_mux)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (max_batch < 0) return fail(TFHE_E_INVALID, "bad batch size");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    return reserve_scratch(c, max_batch, with_mux != 0, c->stream);
}

int tfhe_ctx_reserve_extended(tfhe_ctx *c, int max_batch, int ext)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (max_batch < 0) return fail(TFHE_E_INVALID, "bad batch size");
    if (c->shape != kShapeN2048_L1_B22) return fail(TFHE_E_INVALID, "extended lookup tables need the N = 2048 parameter shape");
    if (ext < 1 || ext > 16) return fail(TFHE_E_INVALID, "polyExtendFactor %d out of range (1..16)", ext);
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    return reserve_extended_scratch(c, max_batch, ext, c->stream);
}

int tfhe_load_bsk_fourier(tfhe_ctx *c, const double *bsk)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!bsk) return fail(TFHE_E_INVALID, "null key");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t elems = bsk_elems(c->P), bytes = elems * sizeof(cd);
    DevBuf raw;
    if ((rc = raw.reserve(bytes)) || (rc = c->bsk.reserve(bytes))) { raw.release(); return rc; }
    HIP_TRY(hipMemcpyAsync(raw.p, bsk, bytes, hipMemcpyHostToDevice, c->stream));
    if (shape_is_1024(c->shape))
        hipLaunchKernelGGL(k_bsk_from_fourier, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, c->stream,
                           raw.as<double>(), c->bsk.as<cd>(), c->P.n, c->P.L);
    else if (shape_is_512(c->shape))
        hipLaunchKernelGGL(k_bsk_from_fourier_512, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, c->stream,
                           raw.as<double>(), c->bsk.as<cd>(), c->P.n);
    else
        hipLaunchKernelGGL(k_bsk_from_fourier_2048, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, c->stream,
                           raw.as<double>(), c->bsk.as<cd>(), c->P.n);
    HIP_TRY(hipGetLastError());
    if ((rc = make_quad_key(c, c->stream))) { raw.release(); return rc; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    raw.release();
    c->have_bsk = true;
    return TFHE_OK;
}

int tfhe_load_bsk_torus(tfhe_ctx *c, const uint32_t *bsk)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!bsk) return fail(TFHE_E_INVALID, "null key");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t polys = (size_t)c->P.n * 2 * c->P.L * 2, bytes = polys * c->P.N * sizeof(uint32_t);
    DevBuf raw;
    if ((rc = raw.reserve(bytes)) || (rc = c->bsk.reserve(bsk_elems(c->P) * sizeof(cd)))) { raw.release(); return rc; }
    HIP_TRY(hipMemcpyAsync(raw.p, bsk, bytes, hipMemcpyHostToDevice, c->stream));
    if (shape_is_1024(c->shape))
        hipLaunchKernelGGL(k_bsk_from_torus, dim3((unsigned)polys), dim3(64), 0, c->stream, raw.as<uint32_t>(),
                           c->bsk.as<cd>(), c->tw.as<cd>(), c->P.L);
    else if (shape_is_512(c->shape))     // reference order (i, row, part) is already the device order
        hipLaunchKernelGGL(k_spectra_512, dim3((unsigned)((polys + 1) / 2)), dim3(64), 0, c->stream, raw.as<uint32_t>(),
                           c->bsk.as<cd>(), c->tw.as<cd>(), (int)polys);
    else
        hipLaunchKernelGGL(k_bsk_from_torus_2048, dim3((unsigned)polys), dim3(64), 0, c->stream, raw.as<uint32_t>(),
                           c->bsk.as<cd>(), c->tw.as<cd>());
    HIP_TRY(hipGetLastError());
    if ((rc = make_quad_key(c, c->stream))) { raw.release(); return rc; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    raw.release();
    c->have_bsk = true;
    return TFHE_OK;
}

int tfhe_load_ksk(tfhe_ctx *c, const uint32_t *ksk)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!ksk) return fail(TFHE_E_INVALID, "null key");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const int n1 = c->P.n + 1, base = 1 << c->P.basebit;
    const size_t ref_bytes = ksk_rows_ref(c->P) * n1 * sizeof(uint32_t);
    const size_t rows_p = ksk_rows_packed(c->P) + 1, total = rows_p * c->n1p;    // + the all-zero padding row
    DevBuf raw;
    if ((rc = raw.reserve(ref_bytes)) || (rc = c->ksk.reserve(total * sizeof(uint32_t)))) { raw.release(); return rc; }
    HIP_TRY(hipMemcpyAsync(raw.p, ksk, ref_bytes, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_ksk_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, raw.as<uint32_t>(),
                       c->ksk.as<uint32_t>(), n1, c->n1p, base, rows_p);
    HIP_TRY(hipGetLastError());
    if ((rc = make_mfma_ksk(c, c->stream))) { raw.release(); return rc; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    raw.release();
    c->have_ksk = true;
    return TFHE_OK;
}

int tfhe_keygen_cloud_seeded(tfhe_ctx *c, const uint32_t *s0, const uint32_t *s1, double alpha_lv0, double alpha_lv1,
                             const uint64_t *seed128)
{
    Seed128 seed{};
    if (seed128) {
        seed.lo = seed128[0]; seed.hi = seed128[1];
    } else if (getrandom(&seed, sizeof seed, 0) != (ssize_t)sizeof seed) {
        return fail(TFHE_E_INVALID, "getrandom failed: no OS entropy for the key-generation seed");
    }
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!s0 || !s1) return fail(TFHE_E_INVALID, "null secret key");
    if (!(alpha_lv0 >= 0.0) || !(alpha_lv1 >= 0.0) || alpha_lv0 >= 0.25 || alpha_lv1 >= 0.25)
        return fail(TFHE_E_INVALID, "noise parameters out of range");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const tfhe_params &P = c->P;
    for (int i = 0; i < P.n; i++) if (s0[i] > 1) return fail(TFHE_E_INVALID, "level-0 key is not binary at %d", i);
    for (int i = 0; i < P.N; i++) if (s1[i] > 1) return fail(TFHE_E_INVALID, "level-1 key is not binary at %d", i);
    DevBuf d_s0, d_s1, d_spec;
    const size_t rows_p = ksk_rows_packed(P) + 1;
    if ((rc = d_s0.reserve((size_t)P.n * 4)) || (rc = d_s1.reserve((size_t)P.N * 4)) ||
        (rc = d_spec.reserve((size_t)(P.N / 2) * sizeof(cd))) || (rc = c->bsk.reserve(bsk_elems(P) * sizeof(cd))) ||
        (rc = c->ksk.reserve(rows_p * c->n1p * sizeof(uint32_t)))) {
        d_s0.release(); d_s1.release(); d_spec.release();
        return rc;
    }
    hipStream_t st = c->stream;
    HIP_TRY(hipMemcpyAsync(d_s0.p, s0, (size_t)P.n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_s1.p, s1, (size_t)P.N * 4, hipMemcpyHostToDevice, st));
    if (shape_is_1024(c->shape)) {
        hipLaunchKernelGGL(k_keygen_s1_spectrum, dim3(1), dim3(64), 0, st, d_s1.as<uint32_t>(), d_spec.as<cd>(), c->tw.as<cd>());
        const dim3 g(P.n * 2 * P.L);
        if (c->shape == kShapeN1024_L3_B6)
            hipLaunchKernelGGL((k_keygen_bsk<3, 6>), g, dim3(64), 0, st, c->bsk.as<cd>(), c->tw.as<cd>(), d_spec.as<cd>(),
                               d_s0.as<uint32_t>(), alpha_lv1, seed);
        else if (c->shape == kShapeN1024_L2_B10)
            hipLaunchKernelGGL((k_keygen_bsk<2, 10>), g, dim3(64), 0, st, c->bsk.as<cd>(), c->tw.as<cd>(), d_spec.as<cd>(),
                               d_s0.as<uint32_t>(), alpha_lv1, seed);
        else
            hipLaunchKernelGGL((k_keygen_bsk<1, 23>), g, dim3(64), 0, st, c->bsk.as<cd>(), c->tw.as<cd>(), d_spec.as<cd>(),
                               d_s0.as<uint32_t>(), alpha_lv1, seed);
    } else if (shape_is_512(c->shape)) {
        hipLaunchKernelGGL(k_spectra_512, dim3(1), dim3(64), 0, st, d_s1.as<uint32_t>(), d_spec.as<cd>(), c->tw.as<cd>(), 1);
        hipLaunchKernelGGL((k_keygen_bsk_512<18>), dim3(P.n), dim3(64), 0, st, c->bsk.as<cd>(), c->tw.as<cd>(),
                           d_spec.as<cd>(), d_s0.as<uint32_t>(), alpha_lv1, seed);
    } else {
        hipLaunchKernelGGL(k_keygen_s1_spectrum_2048, dim3(1), dim3(64), 0, st, d_s1.as<uint32_t>(), d_spec.as<cd>(), c->tw.as<cd>());
        hipLaunchKernelGGL((k_keygen_bsk_2048<22>), dim3(P.n * 2), dim3(64), 0, st, c->bsk.as<cd>(), c->tw.as<cd>(),
                           d_spec.as<cd>(), d_s0.as<uint32_t>(), alpha_lv1, seed);
    }
    hipLaunchKernelGGL(k_keygen_ksk, dim3((unsigned)rows_p), dim3(64), 0, st, c->ksk.as<uint32_t>(), d_s0.as<uint32_t>(),
                       d_s1.as<uint32_t>(), P.n, c->n1p, P.t, P.basebit, rows_p, alpha_lv0, Seed128{seed.lo ^ 0x9E3779B97F4A7C15ull, seed.hi});
    HIP_TRY(hipGetLastError());
    if ((rc = make_quad_key(c, st)) || (rc = make_mfma_ksk(c, st))) { d_s0.release(); d_s1.release(); d_spec.release(); return rc; }
    HIP_TRY(hipStreamSynchronize(st));
    d_s0.release(); d_s1.release(); d_spec.release();
    c->have_bsk = c->have_ksk = true;
    return TFHE_OK;
}

// ---- device-layout key blobs (replication across GPUs, save / restore) ------------------------------------
// A blob = a 64-byte header + the device layout of the key.  The header names what the payload is (magic, layout
// version of this library, which key, the parameter set, the payload length) and carries a checksum of those fields,
// so a blob of another parameter set, another key kind, another layout version -- or a truncated buffer -- is
// rejected (TFHE_E_INVALID) instead of being installed as garbage.
namespace {
constexpr uint64_t kKeyBlobMagic = 0x0159454B45484654ull;       // "TFHEKEY\x01", little-endian
constexpr uint32_t kKeyLayoutVersion = 4;                        // bump whenever a device key layout changes (4: key-switching key rows padded to 128-byte lines)
struct KeyBlobHeader {
    uint64_t magic;
    uint32_t layout, which;
    int32_t params[7];
    uint32_t n1p;
    uint64_t payload_bytes;
    uint64_t check;              // FNV-1a over everything above
};
static_assert(sizeof(KeyBlobHeader) == 64, "the key blob header is 64 bytes");

size_t ksk_device_bytes(const tfhe_ctx *c) { return (ksk_rows_packed(c->P) + 1) * (size_t)c->n1p * sizeof(uint32_t); }
size_t key_payload_bytes(const tfhe_ctx *c, int which) { return which == 0 ? bsk_elems(c->P) * sizeof(cd) : ksk_device_bytes(c); }

uint64_t header_check(const KeyBlobHeader &h)
{
    uint64_t x = 0xcbf29ce484222325ull;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(&h);
    for (size_t i = 0; i < offsetof(KeyBlobHeader, check); i++) { x ^= p[i]; x *= 0x100000001b3ull; }
    return x;
}

KeyBlobHeader make_header(const tfhe_ctx *c, int which)
{
    KeyBlobHeader h{};
    h.magic = kKeyBlobMagic; h.layout = kKeyLayoutVersion; h.which = (uint32_t)which;
    const int32_t pp[7] = {c->P.n, c->P.N, c->P.Nbit, c->P.L, c->P.Bgbit, c->P.basebit, c->P.t};
    memcpy(h.params, pp, sizeof pp);
    h.n1p = (uint32_t)c->n1p;
    h.payload_bytes = key_payload_bytes(c, which);
    h.check = header_check(h);
    return h;
}

int check_header(const tfhe_ctx *c, int which, const KeyBlobHeader &h, size_t bytes)
{
    if (bytes < sizeof(KeyBlobHeader)) return fail(TFHE_E_INVALID, "key blob of %zu bytes is shorter than its header", bytes);
    if (h.magic != kKeyBlobMagic || h.check != header_check(h)) return fail(TFHE_E_INVALID, "not a key blob of this library (bad magic or header checksum)");
    if (h.layout != kKeyLayoutVersion)
        return fail(TFHE_E_INVALID, "key blob has device-layout version %u, this library reads version %u: export it again with this build", h.layout, kKeyLayoutVersion);
    if ((int)h.which != which) return fail(TFHE_E_INVALID, "key blob holds key %u, asked to install it as key %d (0 = bootstrapping, 1 = key-switching)", h.which, which);
    const KeyBlobHeader want = make_header(c, which);
    if (memcmp(h.params, want.params, sizeof h.params) || h.n1p != want.n1p)
        return fail(TFHE_E_INVALID, "key blob was exported for another parameter set (n=%d N=%d L=%d Bgbit=%d basebit=%d t=%d), the context has n=%d N=%d L=%d Bgbit=%d basebit=%d t=%d",
                    h.params[0], h.params[1], h.params[3], h.params[4], h.params[5], h.params[6], c->P.n, c->P.N, c->P.L, c->P.Bgbit, c->P.basebit, c->P.t);
    if (h.payload_bytes != want.payload_bytes || bytes != sizeof(KeyBlobHeader) + want.payload_bytes)
        return fail(TFHE_E_INVALID, "key blob length %zu does not match header + payload = %zu bytes", bytes, sizeof(KeyBlobHeader) + (size_t)want.payload_bytes);
    return TFHE_OK;
}
}

int tfhe_key_size(tfhe_ctx *c, int which, size_t *bytes)
{
    if (!c || !bytes || which < 0 || which > 1) return fail(TFHE_E_INVALID, "bad argument");
    *bytes = sizeof(KeyBlobHeader) + key_payload_bytes(c, which);
    return TFHE_OK;
}

int tfhe_key_export_dev(tfhe_ctx *c, int which, void *d_dst, void *stream)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!d_dst || which < 0 || which > 1) return fail(TFHE_E_INVALID, "bad argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (which == 0 ? !c->have_bsk : !c->have_ksk) return fail(TFHE_E_NOKEY, "key not loaded");
    hipStream_t st = (hipStream_t)stream;
    if (stream_capturing(st)) return fail(TFHE_E_INVALID, "tfhe_key_export_dev on a stream that is being captured into a hipGraph");
    // the header travels from a context-owned page-locked buffer (one per key): an asynchronous copy may read its source
    // after this call has returned, which a stack object does not survive
    if (!c->hdr_host[which]) HIP_TRY(hipHostMalloc(&c->hdr_host[which], sizeof(KeyBlobHeader), hipHostMallocDefault));
    KeyBlobHeader *h = static_cast<KeyBlobHeader *>(c->hdr_host[which]);
    *h = make_header(c, which);
    HIP_TRY(hipMemcpyAsync(d_dst, h, sizeof *h, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(static_cast<char *>(d_dst) + sizeof *h, which == 0 ? c->bsk.p : c->ksk.p, h->payload_bytes,
                           hipMemcpyDeviceToDevice, st));
    return TFHE_OK;
}

int tfhe_key_import_dev(tfhe_ctx *c, int which, const void *d_src, size_t bytes, void *stream)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!d_src || which < 0 || which > 1) return fail(TFHE_E_INVALID, "bad argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    hipStream_t st = (hipStream_t)stream;
    if (stream_capturing(st))
        return fail(TFHE_E_INVALID, "tfhe_key_import_dev on a stream that is being captured into a hipGraph (the header is checked on the host)");
    KeyBlobHeader h{};
    if (bytes < sizeof h) return fail(TFHE_E_INVALID, "key blob of %zu bytes is shorter than its header", bytes);
    HIP_TRY(hipMemcpyAsync(&h, d_src, sizeof h, hipMemcpyDeviceToHost, st));           // the one synchronisation of an import:
    HIP_TRY(hipStreamSynchronize(st));                                                 // the header is checked on the host
    if ((rc = check_header(c, which, h, bytes))) return rc;
    const char *payload = static_cast<const char *>(d_src) + sizeof h;
    if (which == 0) {
        if ((rc = c->bsk.reserve(h.payload_bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(c->bsk.p, payload, h.payload_bytes, hipMemcpyDeviceToDevice, st));
        if ((rc = make_quad_key(c, st))) return rc;
        c->have_bsk = true;
    } else {
        if ((rc = c->ksk.reserve(h.payload_bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(c->ksk.p, payload, h.payload_bytes, hipMemcpyDeviceToDevice, st));
        if ((rc = make_mfma_ksk(c, st))) return rc;
        c->have_ksk = true;
    }
    return mark_dev_stream(c, st);      // the install is only ENQUEUED: tfhe_ctx_sync, tfhe_ctx_destroy and tfhe_ctx_clone_to wait for it through this mark
}

// Host-memory forms of the blobs (persist a GPU-generated cloud key, hand it to another process / machine).
int tfhe_key_export(tfhe_ctx *c, int which, void *dst)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!dst || which < 0 || which > 1) return fail(TFHE_E_INVALID, "bad argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (which == 0 ? !c->have_bsk : !c->have_ksk) return fail(TFHE_E_NOKEY, "key not loaded");
    const KeyBlobHeader h = make_header(c, which);
    memcpy(dst, &h, sizeof h);
    HIP_TRY(hipMemcpyAsync(static_cast<char *>(dst) + sizeof h, which == 0 ? c->bsk.p : c->ksk.p, h.payload_bytes,
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int tfhe_key_import(tfhe_ctx *c, int which, const void *src, size_t bytes)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!src || which < 0 || which > 1) return fail(TFHE_E_INVALID, "bad argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    KeyBlobHeader h{};
    if (bytes < sizeof h) return fail(TFHE_E_INVALID, "key blob of %zu bytes is shorter than its header", bytes);
    memcpy(&h, src, sizeof h);
    if ((rc = check_header(c, which, h, bytes))) return rc;
    DevBuf &dstbuf = which == 0 ? c->bsk : c->ksk;
    if ((rc = dstbuf.reserve(h.payload_bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(dstbuf.p, static_cast<const char *>(src) + sizeof h, h.payload_bytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = which == 0 ? make_quad_key(c, c->stream) : make_mfma_ksk(c, c->stream))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    (which == 0 ? c->have_bsk : c->have_ksk) = true;
    return TFHE_OK;
}

// ---- one cloud key on several GPUs of a node, from ONE process (trgsw.go:234-252: the reference fans a batch out over goroutines
// that share the read-only keys; here the keys are per device, so the fan-out needs a replica per GPU) -----------------------------
// The replica is made GPU to GPU: the two device-layout keys travel by hipMemcpyPeerAsync (over xGMI when the devices are peers;
// a plain device-to-device copy when source and target are the same GPU), and the target derives what a load derives (the
// four-/eight-wave key layout, the byte-column key-switching key) on its own stream.  No host copy of the 147 MB (1.8 GB at the
// Uint5 set) is ever made -- unless the devices are NOT peers (hipDeviceCanAccessPeer false): then the copy is staged through
// page-locked host memory in 32 MB pieces, which is what tfhe_key_export + tfhe_key_import would do, minus the full-size host blob.
namespace {
constexpr size_t kCloneStage = (size_t)32 << 20;

int peer_copy(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, bool peers, hipStream_t st, void *bounce)
{
    if (dst_dev == src_dev && !bounce) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
        return TFHE_OK;
    }
    if (peers) {
        HIP_TRY(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st));
        return TFHE_OK;
    }
    for (size_t at = 0; at < bytes; at += kCloneStage) {      // documented fallback: no peer access between the two devices
        const size_t len = bytes - at < kCloneStage ? bytes - at : kCloneStage;
        HIP_TRY(hipMemcpy(bounce, static_cast<const char *>(src) + at, len, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(static_cast<char *>(dst) + at, bounce, len, hipMemcpyHostToDevice));
    }
    return TFHE_OK;
}
}

int tfhe_ctx_clone_to(tfhe_ctx *src, int device_id, tfhe_ctx **out)
{
    if (!src || !out) return fail(TFHE_E_INVALID, "null argument");
    *out = nullptr;
    int rc = check_ctx(src);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(src->mu);            // no key load on the source while it is being read
    // Whatever the source still has in flight that WRITES its keys must be complete before they are read from another stream / device:
    // tfhe_key_import_dev only enqueues its copy and the derived layouts on the caller's stream (it records the stream's mark), so a
    // clone issued right behind an asynchronous import -- the RCCL-broadcast flow -- would otherwise replicate a half-written key
    // (ADVICE r05).  The source's device is current here (check_ctx).
    HIP_TRY(hipStreamSynchronize(src->stream));
    for (auto &m : src->dev_marks) HIP_TRY(hipEventSynchronize(m.ev));
    if (src->need_sync_all) HIP_TRY(hipDeviceSynchronize());
    tfhe_ctx *dst = nullptr;
    rc = tfhe_ctx_create(&src->P, device_id, &dst);               // validates device_id; leaves device_id current
    if (rc) return rc;
    struct Guard {
        tfhe_ctx *c;
        ~Guard() { if (c) tfhe_ctx_destroy(c); }
    } guard{dst};
    // the per-context limits travel with the key: a clone dispatches like its source
    dst->quad_limit = src->quad_limit.load(); dst->oct_limit = src->oct_limit.load(); dst->ks_mfma_min = src->ks_mfma_min;
    dst->ks_wide_ct = src->ks_wide_ct; dst->combine_max = src->combine_max.load(); dst->comb_quiet_us = src->comb_quiet_us.load();
    bool peers = false;
    void *bounce = nullptr;
    struct Bounce { void *&p; ~Bounce() { if (p) (void)hipHostFree(p); } } bounce_guard{bounce};
    if (src->clone_force_host) {                                  // tests: the fallback of devices that are not peers, on any box
        HIP_TRY(hipHostMalloc(&bounce, kCloneStage, hipHostMallocDefault));
    } else if (device_id != src->device) {
        int can = 0;
        HIP_TRY(hipDeviceCanAccessPeer(&can, device_id, src->device));
        if (can) {
            hipError_t e = hipDeviceEnablePeerAccess(src->device, 0);      // current device (the target) maps the source's memory
            if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
            peers = e == hipSuccess;
            if (!peers) (void)hipGetLastError();
        }
        if (!peers) HIP_TRY(hipHostMalloc(&bounce, kCloneStage, hipHostMallocDefault));
    }
    dst->clone_path = bounce ? 3 : device_id == src->device ? 1 : 2;
    hipStream_t st = dst->stream;
    if (src->have_bsk) {
        const size_t bytes = key_payload_bytes(src, 0);
        if ((rc = dst->bsk.reserve(bytes))) return rc;
        if ((rc = peer_copy(dst->bsk.p, device_id, src->bsk.p, src->device, bytes, peers, st, bounce))) return rc;
        if ((rc = make_quad_key(dst, st))) return rc;
    }
    if (src->have_ksk) {
        const size_t bytes = key_payload_bytes(src, 1);
        if ((rc = dst->ksk.reserve(bytes))) return rc;
        if ((rc = peer_copy(dst->ksk.p, device_id, src->ksk.p, src->device, bytes, peers, st, bounce))) return rc;
        if ((rc = make_mfma_ksk(dst, st))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    dst->have_bsk = src->have_bsk.load();
    dst->have_ksk = src->have_ksk.load();
    guard.c = nullptr;
    *out = dst;
    return TFHE_OK;
}

int tfhe_keygen_cloud(tfhe_ctx *c, const uint32_t *s0, const uint32_t *s1, double alpha_lv0, double alpha_lv1,
                      uint64_t seed)
{
    const uint64_t s[2] = {seed, 0};
    return tfhe_keygen_cloud_seeded(c, s0, s1, alpha_lv0, alpha_lv1, s);
}

// _dev entry points: reserve + enqueue under the context mutex (the grow-only buffers, the event lists and the
// last-stream note are host state), never synchronise, never read device memory.
#define DEV_PROLOGUE()                                              \
    int rc = check_ctx(c);                                          \
    if (rc) return rc;                                              \
    std::lock_guard<std::recursive_mutex> lk(c->mu);                \
    hipStream_t st = pick(c, stream)
// ... and behind the enqueued work the stream's event is re-recorded (tfhe_ctx_sync / tfhe_ctx_destroy wait on it)
// A call that HAS enqueued work on a capturing stream freezes the context: the graph now holds the intermediate buffers'
// addresses.  (Not before: a call rejected for its arguments or for a missing key must not freeze anything.)
#define DEV_RETURN(expr)                                            \
    do {                                                            \
        rc = (expr);                                                \
        if (rc) return rc;                                          \
        if (stream_capturing(st)) c->frozen = true;                 \
        return mark_dev_stream(c, st);                              \
    } while (0)

int tfhe_blind_rotate_batch_dev(tfhe_ctx *c, const uint32_t *d_in, const uint32_t *d_tv, int tv_per_item,
                                uint32_t *d_out, int B, int nsteps, void *stream)
{
    DEV_PROLOGUE();
    if (B < 0 || (B > 0 && (!d_in || !d_out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    RotateJob j;
    j.in0 = d_in; j.tv = d_tv; j.tv_per_item = tv_per_item; j.out = d_out; j.B = B; j.nsteps = nsteps;
    DEV_RETURN(launch_blind_rotate(c, j, st));
}

int tfhe_extract_keyswitch_batch_dev(tfhe_ctx *c, const uint32_t *d_in, uint32_t *d_out, int B, void *stream)
{
    DEV_PROLOGUE();
    if (B < 0 || (B > 0 && (!d_in || !d_out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    DEV_RETURN(launch_keyswitch(c, d_in, d_out, B, nullptr, st));
}

int tfhe_bootstrap_batch_dev(tfhe_ctx *c, const uint32_t *d_in, const uint32_t *d_tv, int tv_per_item,
                             uint32_t *d_out, int B, void *stream)
{
    DEV_PROLOGUE();
    if (B < 0 || (B > 0 && (!d_in || !d_out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (!c->have_bsk || !c->have_ksk) return fail(TFHE_E_NOKEY, "cloud key not loaded");
    if (B == 0) return TFHE_OK;
    DEV_RETURN(bootstrap_device(c, d_in, d_tv, tv_per_item, d_out, B, st));
}

int tfhe_bootstrap_extended_batch_dev(tfhe_ctx *c, const uint32_t *d_in, const uint32_t *d_lut, int lut_per_item, int ext,
                                      uint32_t *d_out, int B, void *stream)
{
    DEV_PROLOGUE();
    if (B < 0 || (B > 0 && (!d_in || !d_lut || !d_out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (!c->have_bsk || !c->have_ksk) return fail(TFHE_E_NOKEY, "cloud key not loaded");
    if (B == 0) return TFHE_OK;
    DEV_RETURN(bootstrap_extended_device(c, d_in, d_lut, lut_per_item, ext, d_out, B, st));
}

int tfhe_gate_batch_dev(tfhe_ctx *c, const uint8_t *d_ops, int op_uniform, const uint32_t *d_a, const uint32_t *d_b,
                        const uint32_t *d_c, uint32_t *d_out, int B, void *stream)
{
    DEV_PROLOGUE();
    if (B < 0 || (B > 0 && (!d_a || !d_b || !d_out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (!c->have_bsk || !c->have_ksk) return fail(TFHE_E_NOKEY, "cloud key not loaded");
    if (B == 0) return TFHE_OK;
    DEV_RETURN(gate_batch_device(c, d_ops, op_uniform, d_a, d_b, d_c, d_out, B, st));
}
#undef DEV_PROLOGUE
#undef DEV_RETURN

// ---- host-pointer variants: stage, run, copy back, synchronise -----------------------

int tfhe_blind_rotate_batch(tfhe_ctx *c, const uint32_t *in, const uint32_t *tv, int tv_per_item, uint32_t *out,
                            int B, int nsteps)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!in || !out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (B == 0) return TFHE_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t inb = (size_t)B * (c->P.n + 1) * 4, trl = (size_t)B * 2 * c->P.N * 4;
    const size_t tvb = tv ? (tv_per_item ? trl : (size_t)2 * c->P.N * 4) : 0;
    if ((rc = c->s_in0.reserve(inb)) || (rc = grow(c, c->s_trlwe, trl, c->stream, "TRLWE accumulator")) || (tvb && (rc = c->s_tv.reserve(tvb)))) return rc;
    HIP_TRY(hipMemcpyAsync(c->s_in0.p, in, inb, hipMemcpyHostToDevice, c->stream));
    if (tv) HIP_TRY(hipMemcpyAsync(c->s_tv.p, tv, tvb, hipMemcpyHostToDevice, c->stream));
    {
        RotateJob j;
        j.in0 = c->s_in0.as<uint32_t>(); j.tv = tv ? c->s_tv.as<uint32_t>() : nullptr; j.tv_per_item = tv_per_item;
        j.out = c->s_trlwe.as<uint32_t>(); j.B = B; j.nsteps = nsteps;
        if ((rc = launch_blind_rotate(c, j, c->stream))) return rc;
    }
    HIP_TRY(hipMemcpyAsync(out, c->s_trlwe.p, trl, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int tfhe_extract_keyswitch_batch(tfhe_ctx *c, const uint32_t *in, uint32_t *out, int B)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!in || !out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (B == 0) return TFHE_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t outb = (size_t)B * (c->P.n + 1) * 4, trl = (size_t)B * 2 * c->P.N * 4;
    if ((rc = c->s_out.reserve(outb)) || (rc = grow(c, c->s_trlwe, trl, c->stream, "TRLWE accumulator"))) return rc;
    HIP_TRY(hipMemcpyAsync(c->s_trlwe.p, in, trl, hipMemcpyHostToDevice, c->stream));
    if ((rc = launch_keyswitch(c, c->s_trlwe.as<uint32_t>(), c->s_out.as<uint32_t>(), B, nullptr, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, c->s_out.p, outb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int tfhe_bootstrap_batch(tfhe_ctx *c, const uint32_t *in, const uint32_t *tv, int tv_per_item, uint32_t *out, int B)
{
    if (!c) return fail(TFHE_E_INVALID, "null context");          // the device is made current where HIP is called (serial path / leader)
    if (B < 0 || (B > 0 && (!in || !out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (B == 0) return TFHE_OK;
    if (!c->have_bsk) return fail(TFHE_E_NOKEY, "bootstrapping key not loaded");
    if (!c->have_ksk) return fail(TFHE_E_NOKEY, "key-switching key not loaded");
    // concurrent callers are combined like those of tfhe_gate_batch (combine_request): one launch of 1 ... 256 bootstraps costs the same
    if (B > c->combine_max || B > combine_cap(c, 1)) return bootstrap_batch_serial(c, in, tv, tv_per_item, out, B);
    tfhe_ctx::GateReq me{1, nullptr, tv_per_item ? 1 : 0, in, tv, nullptr, out, B};
    return combine_request(c, me);
}

int tfhe_bootstrap_extended_batch(tfhe_ctx *c, const uint32_t *in, const uint32_t *lut, int lut_per_item, int ext,
                                  uint32_t *out, int B)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!in || !lut || !out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (ext < 1 || ext > 16) return fail(TFHE_E_INVALID, "polyExtendFactor %d out of range (1..16)", ext);
    if (B == 0) return TFHE_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (!c->have_bsk || !c->have_ksk) return fail(TFHE_E_NOKEY, "cloud key not loaded");
    const size_t inb = (size_t)B * (c->P.n + 1) * 4, lutb = (size_t)(lut_per_item ? B : 1) * ext * 2 * c->P.N * 4;
    if ((rc = c->s_in0.reserve(inb)) || (rc = c->s_out.reserve(inb)) || (rc = c->s_tv.reserve(lutb))) return rc;
    HIP_TRY(hipMemcpyAsync(c->s_in0.p, in, inb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->s_tv.p, lut, lutb, hipMemcpyHostToDevice, c->stream));
    if ((rc = bootstrap_extended_device(c, c->s_in0.as<uint32_t>(), c->s_tv.as<uint32_t>(), lut_per_item, ext,
                                        c->s_out.as<uint32_t>(), B, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, c->s_out.p, inb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int tfhe_gate_batch(tfhe_ctx *c, const uint8_t *ops, int op_uniform, const uint32_t *a, const uint32_t *b,
                    const uint32_t *cc, uint32_t *out, int B)
{
    if (!c) return fail(TFHE_E_INVALID, "null context");          // the device is made current where HIP is called (serial path / leader)
    if (B < 0 || (B > 0 && (!a || !b || !out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (!c->have_bsk || !c->have_ksk) return fail(TFHE_E_NOKEY, "cloud key not loaded");
    if (B == 0) return TFHE_OK;
    if (ops) {              // the op codes are on the host here: refuse bad ones before any work is issued
        for (int i = 0; i < B; i++) {
            if (ops[i] > TFHE_OP_MUX) return fail(TFHE_E_INVALID, "bad op code %d at item %d", ops[i], i);
            if (ops[i] == TFHE_OP_MUX && !cc) return fail(TFHE_E_INVALID, "MUX needs the third operand");
        }
    } else {
        if (op_uniform < 0 || op_uniform > TFHE_OP_MUX) return fail(TFHE_E_INVALID, "bad op code %d", op_uniform);
        if (op_uniform == TFHE_OP_MUX && !cc) return fail(TFHE_E_INVALID, "MUX needs the third operand");
    }
#ifdef TFHE_FUZZ_CONTROL
    // tests/fuzz_gpu.py's positive control (tools/build_variant_main.sh fuzzcontrol -DTFHE_FUZZ_CONTROL): ONE wrong bit in the last row of
    // a batch of exactly one more than the CU count -- the kind of dispatch-boundary defect the fuzzer's batch sizes are weighted to find
    if (B == c->num_cus + 1) {
        const int rc = gate_batch_serial(c, ops, op_uniform, a, b, cc, out, B);
        out[(size_t)(B - 1) * ((size_t)c->P.n + 1)] ^= 1u;
        return rc;
    }
#endif
    if (B > c->combine_max || B > combine_cap(c, 0)) return gate_batch_serial(c, ops, op_uniform, a, b, cc, out, B);
    tfhe_ctx::GateReq me{0, ops, op_uniform, a, b, cc, out, B};
    return combine_request(c, me);
}

int tfhe_external_product_batch(tfhe_ctx *c, int key_index, const uint32_t *in, uint32_t *out, int B)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!in || !out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (!c->have_bsk) return fail(TFHE_E_NOKEY, "bootstrapping key not loaded");
    if (key_index < 0 || key_index >= c->P.n) return fail(TFHE_E_INVALID, "key index %d out of range", key_index);
    if (B == 0) return TFHE_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t trl = (size_t)B * 2 * c->P.N * 4;
    if ((rc = grow(c, c->s_trlwe, trl, c->stream, "TRLWE accumulator")) || (rc = grow(c, c->s_t0, trl, c->stream, "temporary"))) return rc;
    HIP_TRY(hipMemcpyAsync(c->s_trlwe.p, in, trl, hipMemcpyHostToDevice, c->stream));
    launch_external_product(c->shape, c->bsk.as<cd>(), c->tw.as<cd>(), key_index, c->s_trlwe.as<uint32_t>(),
                            c->s_t0.as<uint32_t>(), c->offset, B, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->s_t0.p, trl, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

// ---- trgsw / trlwe seams with caller-supplied operands (SURVEY.md 8(b) seam 3) ------------------------------------------------
namespace {
// One TRGSWLv1FFT (trgsw.go:60-68) in the reference FourierPoly layout, [2L][2][N] float64, brought into the wave-native layout
// the external-product kernels read (the key-ingest kernels with n = 1): c->s_gsw.  Enqueued on the context stream.
int stage_trgsw(tfhe_ctx *c, const double *trgsw)
{
    const size_t elems = (size_t)2 * c->P.L * 2 * (c->P.N / 2), bytes = elems * sizeof(cd);
    int rc;
    if ((rc = c->s_gsw_raw.reserve(bytes)) || (rc = c->s_gsw.reserve(bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(c->s_gsw_raw.p, trgsw, bytes, hipMemcpyHostToDevice, c->stream));
    const dim3 grid((unsigned)((elems + 255) / 256));
    if (shape_is_1024(c->shape))
        hipLaunchKernelGGL(k_bsk_from_fourier, grid, dim3(256), 0, c->stream, c->s_gsw_raw.as<double>(), c->s_gsw.as<cd>(), 1, c->P.L);
    else if (shape_is_512(c->shape))
        hipLaunchKernelGGL(k_bsk_from_fourier_512, grid, dim3(256), 0, c->stream, c->s_gsw_raw.as<double>(), c->s_gsw.as<cd>(), 1);
    else
        hipLaunchKernelGGL(k_bsk_from_fourier_2048, grid, dim3(256), 0, c->stream, c->s_gsw_raw.as<double>(), c->s_gsw.as<cd>(), 1);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}
}

int tfhe_ctx_decomposition_offset(tfhe_ctx *c, uint32_t *offset)
{
    if (!c || !offset) return fail(TFHE_E_INVALID, "null argument");
    *offset = c->offset;
    return TFHE_OK;
}

int tfhe_external_product_with(tfhe_ctx *c, const double *trgsw, uint32_t decomposition_offset, const uint32_t *in, uint32_t *out, int B)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!trgsw) return fail(TFHE_E_INVALID, "null TRGSW operand");
    if (B < 0 || (B > 0 && (!in || !out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (B == 0) return TFHE_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t trl = (size_t)B * 2 * c->P.N * 4;
    if ((rc = grow(c, c->s_trlwe, trl, c->stream, "TRLWE accumulator")) || (rc = grow(c, c->s_t0, trl, c->stream, "temporary"))) return rc;
    if ((rc = stage_trgsw(c, trgsw))) return rc;
    HIP_TRY(hipMemcpyAsync(c->s_trlwe.p, in, trl, hipMemcpyHostToDevice, c->stream));
    launch_external_product(c->shape, c->s_gsw.as<cd>(), c->tw.as<cd>(), 0, c->s_trlwe.as<uint32_t>(), c->s_t0.as<uint32_t>(),
                            decomposition_offset, B, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->s_t0.p, trl, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int tfhe_cmux_with(tfhe_ctx *c, const double *trgsw, uint32_t decomposition_offset, const uint32_t *ct0, const uint32_t *ct1,
                   uint32_t *out, int B)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!trgsw) return fail(TFHE_E_INVALID, "null TRGSW operand");
    if (B < 0 || (B > 0 && (!ct0 || !ct1 || !out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (B == 0) return TFHE_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t words = (size_t)B * 2 * c->P.N, trl = words * 4;
    if ((rc = grow(c, c->s_trlwe, 2 * trl, c->stream, "TRLWE accumulator")) || (rc = grow(c, c->s_t0, trl, c->stream, "temporary")) ||
        (rc = grow(c, c->s_t1, trl, c->stream, "temporary"))) return rc;
    if ((rc = stage_trgsw(c, trgsw))) return rc;
    uint32_t *d0 = c->s_trlwe.as<uint32_t>(), *d1 = d0 + words, *diff = c->s_t0.as<uint32_t>(), *prod = c->s_t1.as<uint32_t>();
    HIP_TRY(hipMemcpyAsync(d0, ct0, trl, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d1, ct1, trl, hipMemcpyHostToDevice, c->stream));
    const dim3 grid((unsigned)((words + 255) / 256));
    hipLaunchKernelGGL(k_torus_addsub<true>, grid, dim3(256), 0, c->stream, (const uint32_t *)d1, (const uint32_t *)d0, diff, words);      // ct1 - ct0
    launch_external_product(c->shape, c->s_gsw.as<cd>(), c->tw.as<cd>(), 0, diff, prod, decomposition_offset, B, c->stream);
    hipLaunchKernelGGL(k_torus_addsub<false>, grid, dim3(256), 0, c->stream, (const uint32_t *)d0, (const uint32_t *)prod, prod, words);   // ct0 + product
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, prod, trl, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int tfhe_sample_extract_batch(tfhe_ctx *c, const uint32_t *in, int k, uint32_t *out, int B)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!in || !out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (k < 0 || k >= c->P.N) return fail(TFHE_E_INVALID, "sample-extract index %d outside [0, %d)", k, c->P.N);
    if (B == 0) return TFHE_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t trl = (size_t)B * 2 * c->P.N * 4, outb = (size_t)B * (c->P.N + 1) * 4;
    if ((rc = grow(c, c->s_trlwe, trl, c->stream, "TRLWE accumulator")) || (rc = grow(c, c->s_t0, outb, c->stream, "temporary"))) return rc;
    HIP_TRY(hipMemcpyAsync(c->s_trlwe.p, in, trl, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_sample_extract, dim3(B), dim3(256), 0, c->stream, (const uint32_t *)c->s_trlwe.as<uint32_t>(), c->s_t0.as<uint32_t>(), c->P.N, k);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->s_t0.p, outb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int tfhe_keyswitch_batch(tfhe_ctx *c, const uint32_t *in, uint32_t *out, int B)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!in || !out))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (B == 0) return TFHE_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t inb = (size_t)B * (c->P.N + 1) * 4, trl = (size_t)B * 2 * c->P.N * 4, outb = (size_t)B * (c->P.n + 1) * 4;
    if ((rc = c->s_out.reserve(outb)) || (rc = grow(c, c->s_trlwe, trl, c->stream, "TRLWE accumulator")) ||
        (rc = grow(c, c->s_t2, inb, c->stream, "temporary"))) return rc;
    HIP_TRY(hipMemcpyAsync(c->s_t2.p, in, inb, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_unextract, dim3(B), dim3(256), 0, c->stream, (const uint32_t *)c->s_t2.as<uint32_t>(), c->s_trlwe.as<uint32_t>(), c->P.N);
    HIP_TRY(hipGetLastError());
    if ((rc = launch_keyswitch(c, c->s_trlwe.as<uint32_t>(), c->s_out.as<uint32_t>(), B, nullptr, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, c->s_out.p, outb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int tfhe_to_fourier_batch(tfhe_ctx *c, const uint32_t *polys, double *spectra, int P)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (P < 0 || (P > 0 && (!polys || !spectra))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (P == 0) return TFHE_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t pb = (size_t)P * c->P.N * 4, sb = (size_t)P * c->P.N * 8;
    if ((rc = grow(c, c->s_t0, pb, c->stream, "temporary")) || (rc = grow(c, c->s_t1, sb, c->stream, "temporary"))) return rc;
    HIP_TRY(hipMemcpyAsync(c->s_t0.p, polys, pb, hipMemcpyHostToDevice, c->stream));
    if (shape_is_1024(c->shape))
        hipLaunchKernelGGL(k_to_fourier, dim3(P), dim3(64), 0, c->stream, c->s_t0.as<uint32_t>(), c->s_t1.as<double>(),
                           c->tw.as<cd>());
    else if (shape_is_512(c->shape))
        hipLaunchKernelGGL(k_to_fourier_512, dim3((P + 1) / 2), dim3(64), 0, c->stream, c->s_t0.as<uint32_t>(),
                           c->s_t1.as<double>(), c->tw.as<cd>(), P);
    else
        hipLaunchKernelGGL(k_to_fourier_2048, dim3(P), dim3(64), 0, c->stream, c->s_t0.as<uint32_t>(),
                           c->s_t1.as<double>(), c->tw.as<cd>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(spectra, c->s_t1.p, sb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int tfhe_to_poly_batch(tfhe_ctx *c, const double *spectra, uint32_t *polys, int P)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (P < 0 || (P > 0 && (!polys || !spectra))) return fail(TFHE_E_INVALID, "bad batch arguments");
    if (P == 0) return TFHE_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const size_t pb = (size_t)P * c->P.N * 4, sb = (size_t)P * c->P.N * 8;
    if ((rc = grow(c, c->s_t0, pb, c->stream, "temporary")) || (rc = grow(c, c->s_t1, sb, c->stream, "temporary"))) return rc;
    HIP_TRY(hipMemcpyAsync(c->s_t1.p, spectra, sb, hipMemcpyHostToDevice, c->stream));
    if (shape_is_1024(c->shape))
        hipLaunchKernelGGL(k_to_poly, dim3(P), dim3(64), 0, c->stream, c->s_t1.as<double>(), c->s_t0.as<uint32_t>(),
                           c->tw.as<cd>());
    else if (shape_is_512(c->shape))
        hipLaunchKernelGGL(k_to_poly_512, dim3((P + 1) / 2), dim3(64), 0, c->stream, c->s_t1.as<double>(),
                           c->s_t0.as<uint32_t>(), c->tw.as<cd>(), P);
    else
        hipLaunchKernelGGL(k_to_poly_2048, dim3(P), dim3(64), 0, c->stream, c->s_t1.as<double>(), c->s_t0.as<uint32_t>(),
                           c->tw.as<cd>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(polys, c->s_t0.p, pb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int tfhe_last_kernel_ms(tfhe_ctx *c, int which, float *ms)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!ms || which < 0 || which > 1) return fail(TFHE_E_INVALID, "bad argument");
    if (!c->ev_valid[which]) return fail(TFHE_E_INVALID, "no launch recorded");
    HIP_TRY(hipEventSynchronize(c->ev[which][1]));
    HIP_TRY(hipEventElapsedTime(ms, c->ev[which][0], c->ev[which][1]));
    return TFHE_OK;
}

int tfhe_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0) return fail(TFHE_E_INVALID, "bad argument");
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(TFHE_E_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return TFHE_OK;
}

int tfhe_host_free(void *p)
{
    if (!p) return TFHE_OK;
    HIP_TRY(hipHostFree(p));
    return TFHE_OK;
}

int tfhe_timing_enable(tfhe_ctx *c, int on)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    c->timing = on != 0;
    return TFHE_OK;
}

int tfhe_timing_read(tfhe_ctx *c, int which, int *launches, float *total_ms)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!launches || !total_ms || which < 0 || which > 1) return fail(TFHE_E_INVALID, "bad argument");
    float sum = 0.f;
    for (auto &pr : c->tev[which]) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(pr.second));
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        sum += ms;
        c->ev_pool.push_back(pr.first);
        c->ev_pool.push_back(pr.second);
    }
    *launches = (int)c->tev[which].size();
    *total_ms = sum;
    c->tev[which].clear();
    return TFHE_OK;
}

} // extern "C"
