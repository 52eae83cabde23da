// tools/combine_bench.cpp -- concurrent submitters on ONE context through the C ABI, without an interpreter lock:
//   T threads each issue the reference adder's 40 scalar gate calls in sequence (README.md:78-106: XOR, AND, XOR, AND, OR per
//   bit; every call waits for its result) on one tfhe_ctx; prints the wall time, the combined launches and the calls they carried,
//   and the same with combining switched off (TFHE_OPT_COMBINE_MAX = 0) for a few threads.
// Random key and random operands: timing does not depend on the values.
//   g++ -O2 -std=c++17 tools/combine_bench.cpp -o tools/combine_bench.bin -Lgo-tfhe_amd/lib -ltfhe_hip -Wl,-rpath,'$ORIGIN/../go-tfhe_amd/lib' -Wl,-rpath,/opt/rocm/lib -lpthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../include/tfhe_hip.h"
#include "../go-tfhe_amd/csrc/tfhe_hip_internal.hpp"      // the gather-exit counters (measurement-only option ids)

#define CK(x) do { int rc_ = (x); if (rc_) { std::printf("FAILED %s: %s\n", #x, tfhe_last_error()); std::exit(1); } } while (0)

// `combine_bench.bin T pbs`: T threads each issue a chain of 8 DEPENDENT programmable bootstraps at the Uint5 set (n = 1071, N = 2048),
// every thread through its own lookup table (evaluator.BootstrapLUT, programmable_bootstrap.go:93-115).
static int pbs_mode(int T)
{
    tfhe_params P{1071, 2048, 11, 1, 22, 6, 3};                // Uint5 (params.go:362-398)
    tfhe_ctx *ctx = nullptr;
    CK(tfhe_ctx_create(&P, 0, &ctx));
    std::mt19937_64 gen(9);
    std::vector<uint32_t> s0(P.n), s1(P.N);
    for (auto &v : s0) v = gen() & 1;
    for (auto &v : s1) v = gen() & 1;
    CK(tfhe_keygen_cloud(ctx, s0.data(), s1.data(), 1.0e-7, 1.0e-15, 13));
    const int n1 = P.n + 1;
    std::vector<std::vector<uint32_t>> in(T, std::vector<uint32_t>(n1)), lut(T, std::vector<uint32_t>((size_t)2 * P.N));
    for (auto &v : in) for (auto &w : v) w = (uint32_t)gen();
    for (auto &v : lut) for (auto &w : v) w = (uint32_t)gen();
    auto chain = [&](int t) {
        std::vector<uint32_t> a(in[t]), b(n1);
        for (int i = 0; i < 8; i++) {
            CK(tfhe_bootstrap_batch(ctx, a.data(), lut[t].data(), 0, b.data(), 1));
            a.swap(b);
        }
    };
    auto run = [&](int threads) {
        std::vector<std::thread> th;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < threads; t++) th.emplace_back(chain, t);
        for (auto &x : th) x.join();
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    chain(0);
    int l0, r0, l1, r1;
    CK(tfhe_ctx_get_option(ctx, TFHE_OPT_COMBINE_LAUNCHES, &l0));
    CK(tfhe_ctx_get_option(ctx, TFHE_OPT_COMBINE_REQUESTS, &r0));
    const double one = run(1), ms = run(T);
    CK(tfhe_ctx_get_option(ctx, TFHE_OPT_COMBINE_LAUNCHES, &l1));
    CK(tfhe_ctx_get_option(ctx, TFHE_OPT_COMBINE_REQUESTS, &r1));
    CK(tfhe_ctx_set_option(ctx, TFHE_OPT_COMBINE_MAX, 0));
    const int Ts = T < 8 ? T : 8;
    const double serial = run(Ts);
    std::printf("{\"mode\": \"pbs uint5\", \"threads\": %d, \"pbs_calls\": %d, \"ms\": %.1f, \"combined_launches\": %d, \"calls_carried\": %d, "
                "\"one_thread_8_pbs_ms\": %.1f, \"serialised_%d_threads_ms\": %.1f, \"serialised_all_threads_projected_ms\": %.0f}\n",
                T, 8 * T, ms, l1 - l0, r1 - r0, one, Ts, serial, serial / Ts * T);
    CK(tfhe_ctx_destroy(ctx));
    std::fflush(stdout);
    return 0;
}

// `combine_bench.bin T batch B [K] [pbs]`: T threads each issue K (default 6) host-pointer gate batches of B NAND gates from pageable memory on ONE
// context -- what several goroutines calling gates.BatchNAND get.  Batches of 257 ... 16,384 gates overlap: one caller's upload and download run
// while another's kernels do (gate_batch_overlapped in csrc/tfhe_hip.hip).
static int batch_mode(int T, int B, int K, bool pbs)
{
    tfhe_params P{700, 1024, 10, 3, 6, 2, 9};
    if (pbs) P = tfhe_params{1071, 2048, 11, 1, 22, 6, 3};          // Uint5: programmable bootstraps through one shared table
    tfhe_ctx *ctx = nullptr;
    CK(tfhe_ctx_create(&P, 0, &ctx));
    std::mt19937_64 gen(5);
    std::vector<uint32_t> s0(P.n), s1(P.N);
    for (auto &v : s0) v = gen() & 1;
    for (auto &v : s1) v = gen() & 1;
    CK(tfhe_keygen_cloud(ctx, s0.data(), s1.data(), pbs ? 1.0e-7 : 2.0e-5, pbs ? 1.0e-15 : 2.0e-8, 11));
    const size_t n1 = P.n + 1;
    std::vector<uint32_t> lut((size_t)2 * P.N);
    for (auto &w : lut) w = (uint32_t)gen();
    std::vector<std::vector<uint32_t>> a(T, std::vector<uint32_t>((size_t)B * n1)), b(a), o(a);
    for (auto &v : a) for (auto &w : v) w = (uint32_t)gen();
    for (auto &v : b) for (auto &w : v) w = (uint32_t)gen();
    auto call = [&](int t, uint32_t *dst) {
        if (pbs) CK(tfhe_bootstrap_batch(ctx, a[t].data(), lut.data(), 0, dst, B));
        else CK(tfhe_gate_batch(ctx, nullptr, TFHE_OP_NAND, a[t].data(), b[t].data(), nullptr, dst, B));
    };
    auto work = [&](int t) { for (int k = 0; k < K; k++) call(t, o[t].data()); };
    auto run = [&](int threads) {
        std::vector<std::thread> th;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < threads; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    work(0);
    const double one = run(1), ms = run(T);
    // every thread's result equals thread 0's recomputed alone from the same operands (same words whatever ran beside it)
    std::vector<uint32_t> ref((size_t)B * n1);
    bool same = true;
    for (int t = 0; t < T; t++) {
        call(t, ref.data());
        same = same && ref == o[t];
    }
    std::printf("{\"mode\": \"host batches%s\", \"threads\": %d, \"batch\": %d, \"calls_per_thread\": %d, \"one_thread_ms_per_call\": %.3f, \"one_thread_gates_per_s\": %.0f, "
                "\"all_threads_ms\": %.1f, \"all_threads_gates_per_s\": %.0f, \"outputs_equal_a_lone_call\": %s}\n",
                pbs ? " (Uint5 PBS)" : "", T, B, K, one / K, 1e3 * B * K / one, ms, 1e3 * (double)B * K * T / ms, same ? "true" : "false");
    CK(tfhe_ctx_destroy(ctx));
    return same ? 0 : 1;
}

int main(int argc, char **argv)
{
    const int T = argc > 1 ? std::atoi(argv[1]) : 256;
    if (argc > 2 && std::string(argv[2]) == "pbs") return pbs_mode(T);
    if (argc > 3 && std::string(argv[2]) == "batch") return batch_mode(T, std::atoi(argv[3]), argc > 4 ? std::atoi(argv[4]) : 6, argc > 5 && std::string(argv[5]) == "pbs");
    tfhe_params P{700, 1024, 10, 3, 6, 2, 9};                 // 128-bit set (params.go:151-180)
    tfhe_ctx *ctx = nullptr;
    CK(tfhe_ctx_create(&P, 0, &ctx));
    std::mt19937_64 gen(7);
    std::vector<uint32_t> s0(P.n), s1(P.N);
    for (auto &v : s0) v = gen() & 1;
    for (auto &v : s1) v = gen() & 1;
    CK(tfhe_keygen_cloud(ctx, s0.data(), s1.data(), 2.0e-5, 2.0e-8, 11));
    if (argc > 3) CK(tfhe_ctx_set_option(ctx, TFHE_OPT_COMBINE_QUIET_US, std::atoi(argv[3])));      // combine_bench T quick <quiet_us>
    const int n1 = P.n + 1;
    std::vector<std::vector<uint32_t>> in(T, std::vector<uint32_t>((size_t)16 * n1));
    for (auto &v : in) for (auto &w : v) w = (uint32_t)gen();

    auto adder = [&](int t) {
        std::vector<uint32_t> x(n1), g(n1), s(n1), u(n1), carry(n1, 0u), nc(n1);
        carry[P.n] = 0xE0000001u;                              // gates.Constant(false), gates.go:61-69
        for (int i = 0; i < 8; i++) {
            const uint32_t *a = in[t].data() + (size_t)i * n1, *b = in[t].data() + (size_t)(8 + i) * n1;
            CK(tfhe_gate_batch(ctx, nullptr, TFHE_OP_XOR, a, b, nullptr, x.data(), 1));
            CK(tfhe_gate_batch(ctx, nullptr, TFHE_OP_AND, a, b, nullptr, g.data(), 1));
            CK(tfhe_gate_batch(ctx, nullptr, TFHE_OP_XOR, x.data(), carry.data(), nullptr, s.data(), 1));
            CK(tfhe_gate_batch(ctx, nullptr, TFHE_OP_AND, x.data(), carry.data(), nullptr, u.data(), 1));
            CK(tfhe_gate_batch(ctx, nullptr, TFHE_OP_OR, g.data(), u.data(), nullptr, nc.data(), 1));
            carry.swap(nc);
        }
    };
    auto run = [&](int threads) {
        std::vector<std::thread> th;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < threads; t++) th.emplace_back(adder, t);
        for (auto &x : th) x.join();
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    adder(0);                                                  // warm-up
    int l0, r0, l1, r1;
    CK(tfhe_ctx_get_option(ctx, TFHE_OPT_COMBINE_LAUNCHES, &l0));
    CK(tfhe_ctx_get_option(ctx, TFHE_OPT_COMBINE_REQUESTS, &r0));
    const double one = run(1);
    const double ms = run(T);
    CK(tfhe_ctx_get_option(ctx, TFHE_OPT_COMBINE_LAUNCHES, &l1));
    CK(tfhe_ctx_get_option(ctx, TFHE_OPT_COMBINE_REQUESTS, &r1));
    int us_idle = 0, us_gather = 0, us_launch = 0;
    CK(tfhe_ctx_get_option(ctx, TFHE_OPT_COMBINE_US_IDLE, &us_idle));
    CK(tfhe_ctx_get_option(ctx, TFHE_OPT_COMBINE_US_GATHER, &us_gather));
    CK(tfhe_ctx_get_option(ctx, TFHE_OPT_COMBINE_US_LAUNCH, &us_launch));
    int ex[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) CK(tfhe_ctx_get_option(ctx, TFHE_OPT_COMBINE_EXIT_NONE + i, &ex[i]));
    const int nl = l1 - l0 > 0 ? l1 - l0 : 1;
    const bool quick = argc > 2 && std::string(argv[2]) == "quick";      // skip the serialised comparison (0.7 s)
    CK(tfhe_ctx_set_option(ctx, TFHE_OPT_COMBINE_MAX, 0));
    const int Ts = T < 8 ? T : 8;
    const double serial = quick ? 0.0 : run(Ts);
    std::printf("{\"threads\": %d, \"gate_calls\": %d, \"ms\": %.1f, \"combined_launches\": %d, \"calls_carried\": %d, "
                "\"per_combined_launch_us\": {\"idle_before\": %.0f, \"of_which_gathering\": %.0f, \"launch\": %.0f}, "
                "\"gather_exits\": {\"none\": %d, \"stale\": %d, \"all_back\": %d, \"full\": %d, \"quiet\": %d, \"deadline\": %d}, "
                "\"one_thread_40_gates_ms\": %.1f, \"serialised_%d_threads_ms\": %.1f, \"serialised_all_threads_projected_ms\": %.0f}\n",
                T, 40 * T, ms, l1 - l0, r1 - r0, (double)us_idle / nl, (double)us_gather / nl, (double)us_launch / nl,
                ex[0], ex[1], ex[2], ex[3], ex[4], ex[5], one, Ts, serial, serial / Ts * T);
    CK(tfhe_ctx_destroy(ctx));
    std::fflush(stdout);
    return 0;
}
