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
#include "host_par.hpp"

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
    return dock::abi_guard([&]() -> int32_t {
    Fq12 r = Fq12::one();
    if (n) {
        size_t T = std::min<size_t>(std::max<size_t>(1, std::thread::hardware_concurrency()), (n + 1) / 2);
        T = std::min<size_t>(T, 32);
        if (T <= 1) r = multi_pow_serial(a, e, n);
        else {
            std::vector<Fq12> part(T);
            const int32_t rc = dock::par_run(T, [&](size_t t) -> int32_t {
                const size_t lo = n * t / T, hi = n * (t + 1) / T;
                part[t] = multi_pow_serial(a + 72 * lo, e + 4 * lo, hi - lo); return DGPU_OK; });
            if (rc) return rc;
            for (size_t t = 0; t < T; t++) r = r * part[t];
        }
    }
    memcpy(out, &r, sizeof r);
    return DGPU_OK;
    });
}

