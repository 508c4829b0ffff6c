// tests/native/inflight_threads.cpp — compiled host driver over the C ABI (GPU box): COUNT resident 2^20-term G1 MSMs from T NATIVE host threads — what a Rust
// host's rayon workers are — per "T,COUNT" argument, three passes each: total time, ms per call, the longest single call, results compared with the first
// call's.  bench.py runs it for `secondary.native_host_threads` (the headline's workload without an interpreter between the calls); __graft_entry__.build()
// compiles it (tests/test_gpu_cpp_api.py build_inflight_driver).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "dock_gpu.h"
static const char *G1_GEN_HEX = "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb";
int main(int argc, char **argv) {
    if (dgpu_init(0)) { std::puts("no device"); return 1; }
    const size_t n = (size_t)1 << 20;
    uint8_t gen[48]; for (int i = 0; i < 48; i++) { unsigned v; std::sscanf(G1_GEN_HEX + 2 * i, "%2x", &v); gen[i] = (uint8_t)v; }
    uint64_t gxy[12]; uint8_t ginf = 0;
    if (dgpu_g1_deserialize(gen, 1, 1, gxy, &ginf)) { std::puts("deserialize"); return 1; }
    std::vector<uint64_t> ks(n * 4), sc(n * 4);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (size_t i = 0; i < n; i++) { for (int k = 0; k < 4; k++) { ks[4 * i + k] = rnd(); sc[4 * i + k] = rnd(); } ks[4 * i + 3] >>= 2; sc[4 * i + 3] >>= 2; }
    uint64_t tab = 0, bases = 0, scal = 0;
    if (dgpu_window_table_g1(gxy, &tab) || dgpu_window_table_mul_to_bases_g1(tab, ks.data(), n, 0, &bases) || dgpu_bases_precompute_g1(bases, 0) || dgpu_scalars_upload(sc.data(), n, 0, &scal)) { std::puts("setup"); return 1; }
    uint64_t ref[18]; if (dgpu_msm_g1_resident(bases, 0, scal, 0, n, ref)) { std::puts("msm"); return 1; }
    for (int a = 1; a < argc; a++) {
        int T = 0, COUNT = 0; if (std::sscanf(argv[a], "%d,%d", &T, &COUNT) != 2) continue;
        for (int rep = 0; rep < 3; rep++) {
            std::atomic<int> next{0}, bad{0}; std::atomic<long> worst{0};
            std::atomic<bool> go{false};
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&] {
                while (!go.load()) std::this_thread::yield();
                for (;;) {
                    const int i = next.fetch_add(1); if (i >= COUNT) break;
                    uint64_t out[18];
                    const auto t0 = std::chrono::steady_clock::now();
                    if (dgpu_msm_g1_resident(bases, 0, scal, 0, n, out) || std::memcmp(out, ref, sizeof out)) bad++;
                    const long us = (long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                    long w = worst.load(); while (us > w && !worst.compare_exchange_weak(w, us)) {}
                }
            });
            const auto t0 = std::chrono::steady_clock::now();
            go = true;
            for (auto &t : th) t.join();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::printf("T=%2d count=%3d: %7.2f ms total, %.3f ms per call, longest call %.1f ms, mismatches %d\n", T, COUNT, ms, ms / COUNT, worst.load() / 1e3, bad.load());
        }
    }
    return 0;
}
