// crypto_amd/csrc/dock_gt.cpp — GT (Fp12) products of powers on the host cores (include/dock_gpu.h).
//
// The aggregation verifier folds every T, U, Z of the GIPA rounds with the round challenges
// (legogroth16/src/aggregation/groth16/verifier.rs:272-370: a rayon fold of `PairingOutput::mul_bigint` + `add_assign`);
// here each target is one multi-exponentiation prod a_i^{e_i}: 4-bit windows, the bases split over host threads, each
// thread sharing its 252 squarings among its bases.  The values come from an untrusted proof and need not lie in the cyclotomic
// subgroup: the cyclotomic shortcuts (Granger-Scott squarings, conjugates for negative digits) are taken only for bases that
// pass the membership test of that subgroup (a Frobenius identity, ~5 us per base); GT membership itself: dgpu_gt_in_subgroup.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>
#include "../../include/dock_gpu.h"
#include "host_field.hpp"
#include "host_par.hpp"

namespace {
using hostf::Fq12;

// f in the cyclotomic subgroup G_{Phi_12(p)} (order p^4 - p^2 + 1): f^(p^4) f == f^(p^2).  There the inverse is the conjugate and the Granger-Scott
// squaring (hostf::Fq12::cyclotomic_sqr) is the square; every output of a final exponentiation lies in it, an arbitrary Fp12 value from an untrusted
// proof need not.  Three Frobenius maps and a product: ~5 us.
bool in_cyclotomic(const Fq12 &f) {
    if (f.is_zero()) return false;
    const Fq12 f2 = hostf::frob2(f), f4 = hostf::frob2(f2);
    const Fq12 l = f4 * f;
    return memcmp(&l, &f2, sizeof l) == 0;
}
// f in GT (order r): in the cyclotomic subgroup and f^p == f^x — x the curve parameter, p = x mod r, and the cofactor of r in Phi_12(p) shares nothing
// with p - x (M. Scott, "A note on group membership tests for G1, G2 and GT on BLS pairing-friendly curves", the test blst's blst_fp12_in_group
// runs): one exponentiation by the 64-bit |x| with cyclotomic squarings (~90 us) instead of f^r with generic ones (1.3 ms).
bool in_gt(const Fq12 &f) {
    if (!in_cyclotomic(f)) return false;
    const Fq12 l = hostf::frob1(f), r = hostf::exp_by_x(f);
    return memcmp(&l, &r, sizeof l) == 0;
}

// signed 4-bit digits of a 256-bit exponent (d in [-7, 8], 65 of them: the last holds the carry out of bit 255)
void signed_digits(const uint64_t e[4], int8_t d[65]) {
    unsigned carry = 0;
    for (int w = 0; w < 64; w++) {
        unsigned v = ((unsigned)(e[w / 16] >> (4 * (w % 16))) & 15u) + carry;
        carry = v > 8 ? 1 : 0;
        d[w] = (int8_t)(carry ? (int)v - 16 : (int)v);
    }
    d[64] = (int8_t)carry;
}

// prod a_i^{e_i}.  Bases of the cyclotomic subgroup: signed digits (a negative one multiplies by the conjugate of the table entry), eight table entries
// per base, cyclotomic squarings; anything else: unsigned digits, fifteen entries, generic squarings.  The same value either way.
Fq12 multi_pow_serial(const uint64_t *a, const uint64_t *e, size_t n) {
    bool cyc = true;
    for (size_t i = 0; i < n && cyc; i++) { Fq12 b; memcpy(&b, a + 72 * i, sizeof b); cyc = in_cyclotomic(b); }
    if (cyc) {
        std::vector<Fq12> tab(n * 8);                  // tab[i][d-1] = a_i^d, d = 1..8
        std::vector<int8_t> dig(n * 65);
        for (size_t i = 0; i < n; i++) {
            Fq12 b; memcpy(&b, a + 72 * i, sizeof b);
            tab[8 * i] = b;
            for (int d = 2; d <= 8; d++) tab[8 * i + d - 1] = (d & 1) ? tab[8 * i + d - 2] * b : tab[8 * i + d / 2 - 1].cyclotomic_sqr();
            signed_digits(e + 4 * i, &dig[65 * i]);
        }
        Fq12 acc = Fq12::one();
        bool started = false;
        for (int w = 64; w >= 0; w--) {
            if (started) for (int k = 0; k < 4; k++) acc = acc.cyclotomic_sqr();
            for (size_t i = 0; i < n; i++) {
                const int d = dig[65 * i + w];
                if (d > 0) { acc = acc * tab[8 * i + d - 1]; started = true; }
                else if (d < 0) { acc = acc * tab[8 * i - d - 1].conj(); started = true; }
            }
        }
        return acc;
    }
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

// `PairingOutput::mul_bigint` (the power in GT's multiplicative notation): any 256-bit exponent, any Fp12 base
// PairingOutput `+`: the Fp12 product (host code, like everything in this file; moved here from dock_pairing.hip in round 6 so that the host-only units link without it)
extern "C" int32_t dgpu_fp12_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) {
    if (!a || !b || !out) return DGPU_E_BADARG;
    hostf::Fq12 x, y; memcpy(&x, a, sizeof x); memcpy(&y, b, sizeof y);
    hostf::Fq12 r = x * y; memcpy(out, &r, sizeof r); return DGPU_OK;
}
extern "C" int32_t dgpu_fp12_pow(const uint64_t *a, const uint64_t *e, uint64_t *out) {
    if (!a || !e || !out) return DGPU_E_BADARG;
    return dock::abi_guard([&]() -> int32_t { const Fq12 r = multi_pow_serial(a, e, 1); memcpy(out, &r, sizeof r); return DGPU_OK; });
}
// ok[i] = a_i has order dividing r, i.e. is an element of GT: what `Valid::check` of ark-ec's PairingOutput answers with f^r == 1 when a proof is
// deserialized with Validate::Yes (the aggregation's proofs carry 2 + 6 log2(n) and more of them).  Host threads.
extern "C" int32_t dgpu_gt_in_subgroup(const uint64_t *a, size_t n, uint8_t *ok) {
    if (n && (!a || !ok)) return DGPU_E_BADARG;
    return dock::abi_guard([&]() -> int32_t {
        const size_t T = std::min<size_t>(std::min<size_t>(std::max<size_t>(1, std::thread::hardware_concurrency()), 32), (n + 1) / 2);
        return dock::par_run(std::max<size_t>(T, 1), [&](size_t t) -> int32_t {
            for (size_t i = t; i < n; i += std::max<size_t>(T, 1)) { Fq12 f; memcpy(&f, a + 72 * i, sizeof f); ok[i] = in_gt(f) ? 1 : 0; }
            return DGPU_OK; });
    });
}

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

