// tools/ubench_wake.cpp -- what waking T sleeping threads costs on this host (the flat-combining wake path of csrc/tfhe_hip.hip):
//   mode 0: ONE FUTEX_WAKE(INT_MAX) by the waker;  mode 1: the waker wakes 16, every woken thread forwards FUTEX_WAKE(2) (a wake tree).
//   g++ -O2 -std=c++17 tools/ubench_wake.cpp -o /tmp/ubench_wake -lpthread && /tmp/ubench_wake 255 0
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <climits>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
using clk = std::chrono::steady_clock;
static std::atomic<uint32_t> gen{0};
static std::atomic<int> ready{0}, done{0};
static std::atomic<long long> last_ns{0};
static clk::time_point t0;
int main(int argc, char **argv) {
    int T = argc > 1 ? atoi(argv[1]) : 255, mode = argc > 2 ? atoi(argv[2]) : 0, rounds = 30;
    std::vector<std::thread> th;
    std::atomic<bool> stop{false};
    for (int i = 0; i < T; i++) th.emplace_back([&] {
        uint32_t seen = 0;
        while (!stop) {
            ready++;
            while (gen.load() == seen && !stop) syscall(SYS_futex, (uint32_t*)&gen, FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
            if (stop) break;
            seen = gen.load();
            if (mode == 1) syscall(SYS_futex, (uint32_t*)&gen, FUTEX_WAKE_PRIVATE, 2, nullptr, nullptr, 0);
            long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count();
            long long prev = last_ns.load();
            while (ns > prev && !last_ns.compare_exchange_weak(prev, ns)) {}
            done++;
        }
    });
    double sum_wake = 0, sum_last = 0;
    for (int r = 0; r < rounds; r++) {
        while (ready.load() < T * (r + 1)) usleep(100);
        usleep(2000);
        last_ns = 0; done = 0;
        t0 = clk::now();
        gen++;
        syscall(SYS_futex, (uint32_t*)&gen, FUTEX_WAKE_PRIVATE, mode == 1 ? 16 : INT_MAX, nullptr, nullptr, 0);
        double wake_us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
        while (done.load() < T) usleep(50);
        sum_wake += wake_us; sum_last += last_ns.load() / 1e3;
    }
    printf("T=%d mode=%d: wake syscall %.0f us, last thread running after %.0f us (avg of %d)\n", T, mode, sum_wake / rounds, sum_last / rounds, rounds);
    stop = true; gen++;
    syscall(SYS_futex, (uint32_t*)&gen, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
    for (auto &t : th) t.join();
}
