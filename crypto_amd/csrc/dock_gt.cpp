// crypto_amd/csrc/dock_gt.cpp — GT (Fp12) products of powers on the host cores (include/dock_gpu.h).
//
// The aggregation verifier folds every T, U, Z of the GIPA rounds with the round challenges
// (legogroth16/src/aggregation/groth16/verifier.rs:272-370: a rayon fold of `PairingOutput::mul_bigint` + `add_assign`);
// here each target is one multi-exponentiation prod a_i^{e_i}: 4-bit windows, the bases split over host threads, each
// thread sharing its 252 squarings among its bases.  Generic Fp12 arithmetic (no cyclotomic shortcuts): the values come
// from an untrusted proof and need not lie in the cyclotomic subgroup.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>
#include "../../include/dock_gpu.h"
#include "host_field.hpp"

namespace {
using hostf::Fq12;

Fq12 multi_pow_serial(const uint64_t *a, const uint64_t *e, size_t n) {
    std::vector<Fq12> tab(n * 15);                     // tab[i][d-1] = a_i^d, d = 1..15
    for (size_t i = 0; i < n; i++) {
        Fq12 b; memcpy(&b, a + 72 * i, sizeof b);
        tab[15 * i] = b;
        for (int d = 2; d <= 15; d++) tab[15 * i + d - 1] = (d & 1) ? tab[15 * i + d - 2] * b : tab[15 * i + d / 2 - 1].sqr();
    }
    Fq12 acc = Fq12::one();
    bool started = false;
    for (int w = 63; w >= 0; w--) {
        if (started) for (int k = 0; k < 4; k++) acc = acc.sqr();
        for (size_t i = 0; i < n; i++) {
            unsigned d = (unsigned)(e[4 * i + w / 16] >> (4 * (w % 16))) & 15u;
            if (d) { acc = acc * tab[15 * i + d - 1]; started = true; }
        }
    }
    return acc;
}
}  // namespace

extern "C" int32_t dgpu_fp12_multi_pow(const uint64_t *a, const uint64_t *e, size_t n, uint64_t out[72]) {
    if (!out || (n && (!a || !e))) return DGPU_E_BADARG;
    Fq12 r = Fq12::one();
    if (n) {
        size_t T = std::min<size_t>(std::max<size_t>(1, std::thread::hardware_concurrency()), (n + 1) / 2);
        T = std::min<size_t>(T, 32);
        if (T <= 1) r = multi_pow_serial(a, e, n);
        else {
            std::vector<Fq12> part(T);
            std::vector<std::thread> th;
            for (size_t t = 0; t < T; t++) {
                size_t lo = n * t / T, hi = n * (t + 1) / T;
                th.emplace_back([&, t, lo, hi] { part[t] = multi_pow_serial(a + 72 * lo, e + 4 * lo, hi - lo); });
            }
            for (auto &x : th) x.join();
            for (size_t t = 0; t < T; t++) r = r * part[t];
        }
    }
    memcpy(out, &r, sizeof r);
    return DGPU_OK;
}

// ---- Keccak-f[1600]: the permutation under the Merlin / STROBE-128 transcript of the aggregation protocol
// (merlin/src/strobe.rs:97-104 `run_f` calls keccak::f1600).  The transcript framing stays in the host language
// (crypto_amd/aggregation/transcript.py); only the 24-round permutation is native.
extern "C" int32_t dgpu_keccak_f1600(uint8_t state[200]) {
    if (!state) return DGPU_E_BADARG;
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL, 0x0000000080000001ULL,
        0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
        0x000000000000800AULL, 0x800000008000000AULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};   // index x + 5y
    uint64_t a[25];
    memcpy(a, state, 200);                                  // little-endian host
    auto rol = [](uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; };
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];
    }
    memcpy(state, a, 200);
    return DGPU_OK;
}
