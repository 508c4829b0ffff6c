// crypto_amd/csrc/host_field.hpp — host-side (x86-64) BLS12-381 field and group arithmetic used by
// libdock_gpu.so for the O(window-count) tail of an MSM (Horner fold of the per-window sums, final
// normalisation) and for scalar preparation.  A lone wave on the GPU needs ~15 us per group addition;
// a host core needs ~0.3 us, so the few hundred strictly sequential operations at the end of an MSM
// belong here.  Product code: it does not use anything under oracle/.
//
// Representation = the C-ABI one: 6 x u64 little-endian, value * 2^384 mod p (ark-ff Fp<MontBackend,6>).
#pragma once
#include <stdint.h>
#include <string.h>
#include <immintrin.h>

namespace hostf {

typedef unsigned __int128 u128;

struct Fq {
    uint64_t l[6];
    static constexpr uint64_t P[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
    static constexpr uint64_t ONE[6] = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
    static constexpr uint64_t INV = 0x89f3fffcfffcfffdULL;

    static Fq zero() { Fq r; memset(&r, 0, sizeof r); return r; }
    static Fq one() { Fq r; memcpy(r.l, ONE, sizeof ONE); return r; }
    bool is_zero() const { uint64_t t = 0; for (int i = 0; i < 6; i++) t |= l[i]; return t == 0; }
    bool operator==(const Fq &o) const { uint64_t t = 0; for (int i = 0; i < 6; i++) t |= l[i] ^ o.l[i]; return t == 0; }

    static bool geq_p(const uint64_t *a) {
        for (int i = 5; i >= 0; i--) { if (a[i] > P[i]) return true; if (a[i] < P[i]) return false; }
        return true;
    }
    static void sub_p(uint64_t *a) {
        uint64_t br = 0;
        for (int i = 0; i < 6; i++) { u128 d = (u128)a[i] - P[i] - br; a[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    }
    // t in [0, 2p) -> t mod p without a branch: the borrow of t - p selects (the tower's additions are a third of an Fp12 product's time when
    // each ends in a data-dependent branch: 3.9 -> 3.3 us per product on the build container's Xeon)
    static void reduce_once(uint64_t *r, const uint64_t *t) {
        uint64_t u[6], br = 0;
        for (int i = 0; i < 6; i++) { u128 d = (u128)t[i] - P[i] - br; u[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
        const uint64_t keep = (uint64_t)0 - br;                 // all ones: t < p
        for (int i = 0; i < 6; i++) r[i] = (t[i] & keep) | (u[i] & ~keep);
    }
    Fq operator+(const Fq &b) const {
        Fq r; uint64_t t[6], c = 0;
        for (int i = 0; i < 6; i++) { u128 s = (u128)l[i] + b.l[i] + c; t[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }      // < 2p < 2^382: no carry out
        reduce_once(r.l, t);
        return r;
    }
    Fq operator-(const Fq &b) const {
        Fq r; uint64_t d[6], br = 0, c = 0;
        for (int i = 0; i < 6; i++) { u128 x = (u128)l[i] - b.l[i] - br; d[i] = (uint64_t)x; br = (uint64_t)(x >> 64) & 1; }
        const uint64_t addp = (uint64_t)0 - br;                 // all ones: a < b, add p back
        for (int i = 0; i < 6; i++) { u128 s = (u128)d[i] + (P[i] & addp) + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        return r;
    }
    Fq neg() const { return is_zero() ? *this : zero() - *this; }
    // Montgomery product, finely integrated operand scanning: the partial product row a * b_i and the reduction row m * p advance
    // together as two carry chains.  p < 2^381 leaves the top word three spare bits, so the running value never needs a seventh word
    // (t < 2p throughout) and one conditional subtraction finishes.
    Fq operator*(const Fq &b) const {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; i++) {
            const uint64_t bi = b.l[i];
            u128 A = (u128)l[0] * bi + t[0];
            const uint64_t m = (uint64_t)A * INV;
            u128 C = (u128)m * P[0] + (uint64_t)A;
            for (int j = 1; j < 6; j++) {
                A = (u128)l[j] * bi + t[j] + (uint64_t)(A >> 64);
                C = (u128)m * P[j] + (uint64_t)A + (uint64_t)(C >> 64);
                t[j - 1] = (uint64_t)C;
            }
            t[5] = (uint64_t)(C >> 64) + (uint64_t)(A >> 64);
        }
        Fq r; reduce_once(r.l, t);
        return r;
    }
    Fq sqr() const { return (*this) * (*this); }
    Fq dbl() const { return (*this) + (*this); }
    Fq inv() const {   // Fermat: a^(p-2)
        static constexpr uint64_t E[6] = {0xb9feffffffffaaa9ULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
        Fq acc = one();
        for (int i = 380; i >= 0; i--) { acc = acc.sqr(); if ((E[i / 64] >> (i % 64)) & 1) acc = acc * (*this); }
        return acc;
    }
};

// ---- double-width values: lazy reduction across the tower -----------------------------------------------------------------------
// An Fq12 product is 54 Fq products; reduced one by one (36 + 36 word products each) the 54 Montgomery reductions are half of it.  The
// sums and differences of the Karatsuba levels can be taken on the UNREDUCED 768-bit products instead, so that an Fq2 product ends in 2
// reductions instead of 3 and an Fq6 product in 6 instead of 18 (Aranha et al., "Faster explicit formulas for computing pairings over
// ordinary curves", §3): 18 x 36 + 6 x 36 = 864 word products per Fq6 product instead of 1296.
// Wide values are plain non-negative 768-bit integers congruent to the value mod p: additions are 12-word additions, a subtraction adds
// a multiple k p R of p R = p 2^384 first (k p into the high half, k >= the subtrahend's bound) — no comparison, no select.  p R =
// 0.1016 * 2^768, so sums up to 9 p R fit.  Montgomery reduction of T < B p R yields T / R mod p in [0, (B + 1) p), and log2(B + 1)
// conditional subtractions (4p, 2p, p) finish: the results are the same canonical residues as with eager reduction, bit for bit.
// Every formula below states its bounds in units of p R (p^2 = 0.1016 p R); with -DHOSTF_CHECK each value carries its bound and every
// operation asserts its precondition (tests/test_gt_host.py builds the tower that way once).
#ifdef HOSTF_CHECK
#include <assert.h>
#define HCHK(...) __VA_ARGS__
#else
#define HCHK(...)
#endif
// 64 x 64 -> 128 (mulx when the unit is compiled for BMI2: the library's host units are; a test shim built without it still compiles)
static inline unsigned long long mulx64(unsigned long long a, unsigned long long b, unsigned long long *hi) { const u128 p = (u128)a * b; *hi = (unsigned long long)(p >> 64); return (unsigned long long)p; }
struct FqD {
    uint64_t l[12];
    HCHK(double ub;)                                            // upper bound of the value, in units of p R
    static constexpr double P_OVER_R = 0.10158;                 // p / 2^384, rounded up
    // r = a * b; ua, ub_: how many times p each operand may be (sums of two canonical values enter unreduced)
    static FqD mul(const uint64_t *a, const uint64_t *b, int ua = 1, int ub_ = 1) {
        // row by row, the low and the high words of a row's six products as two carry chains (mulx + adc: what the u128 spelling of the
        // same loop is not compiled to)
        typedef unsigned long long ull;
        FqD r; ull t[12];
        {   ull hi[6], lo[6];
#pragma unroll
            for (int j = 0; j < 6; j++) lo[j] = mulx64(a[j], b[0], &hi[j]);
            t[0] = lo[0]; unsigned char c = 0;
#pragma unroll
            for (int j = 1; j < 6; j++) c = _addcarry_u64(c, lo[j], hi[j - 1], &t[j]);
            _addcarry_u64(c, hi[5], 0, &t[6]);
        }
#pragma unroll
        for (int i = 1; i < 6; i++) {
            ull hi[6], lo[6];
#pragma unroll
            for (int j = 0; j < 6; j++) lo[j] = mulx64(a[j], b[i], &hi[j]);
            unsigned char c = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) c = _addcarry_u64(c, t[i + j], lo[j], &t[i + j]);
            ull top; _addcarry_u64(c, 0, 0, &top);
            c = 0;
#pragma unroll
            for (int j = 0; j < 5; j++) c = _addcarry_u64(c, t[i + j + 1], hi[j], &t[i + j + 1]);
            _addcarry_u64(c, top, hi[5], &t[i + 6]);
        }
        memcpy(r.l, t, sizeof t);
        (void)ua; (void)ub_;
        HCHK(r.ub = ua * ub_ * P_OVER_R;)
        return r;
    }
    FqD operator+(const FqD &b) const {
        typedef unsigned long long ull;
        FqD r; ull t[12]; unsigned char c = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) c = _addcarry_u64(c, l[i], b.l[i], &t[i]);
        memcpy(r.l, t, sizeof t);
        HCHK(r.ub = ub + b.ub; assert(r.ub < 9.8 && c == 0);)
        return r;
    }
    // a - b + K p R, K >= b's bound: non-negative whatever a is
    template <int K> FqD sub(const FqD &b) const {
        static_assert(K >= 0 && K <= 8, "k p must fit six words");
        typedef unsigned long long ull;
        FqD r; ull t[12]; unsigned char br = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) br = _subborrow_u64(br, l[i], b.l[i], &t[i]);
        if (K > 0) {
            ull kp[6]; { ull cc = 0; for (int i = 0; i < 6; i++) { u128 x = (u128)Fq::P[i] * (ull)K + cc; kp[i] = (ull)x; cc = (ull)(x >> 64); } }    // (constant-folded)
            unsigned char c = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) c = _addcarry_u64(c, t[6 + i], kp[i], &t[6 + i]);
            HCHK(assert(c == br);)                             // the borrow of a negative difference is cancelled by the carry of the offset
        } else { HCHK(assert(br == 0);) }
        memcpy(r.l, t, sizeof t);
        HCHK(assert(K == 0 || (double)K >= b.ub); r.ub = ub + K; assert(r.ub < 9.8);)
        return r;
    }
    // T / 2^384 mod p, canonical; T < (2^STEPS - 1) p R
    template <int STEPS> Fq redc() const {
        HCHK(assert(ub + 1.0 <= (double)(1 << STEPS));)
        typedef unsigned long long ull;
        ull t[12]; memcpy(t, l, sizeof t);
        ull pend = 0;                                           // carry out of word i + 5 + 1 of the previous row, due at word i + 6
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const ull m = t[i] * Fq::INV;
            ull hi[6], lo[6];
#pragma unroll
            for (int j = 0; j < 6; j++) lo[j] = mulx64(m, Fq::P[j], &hi[j]);
            unsigned char c = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) c = _addcarry_u64(c, t[i + j], lo[j], &t[i + j]);
            c = _addcarry_u64(c, t[i + 6], pend, &t[i + 6]);
            pend = c;
            c = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) c = _addcarry_u64(c, t[i + j + 1], hi[j], &t[i + j + 1]);
            pend += c;
        }
        HCHK(assert(pend == 0);)                                // (T + m p) / R < 2^STEPS p < 2^384
        ull *h = t + 6;
#pragma unroll
        for (int k = STEPS - 1; k >= 0; k--) {                   // h in [0, 2^(k+1) p) -> [0, 2^k p)
            ull u[6]; unsigned char br = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const ull kp = (Fq::P[i] << k) | (i && k ? Fq::P[i - 1] >> (64 - k) : 0);      // word i of 2^k p (a constant)
                br = _subborrow_u64(br, h[i], kp, &u[i]);
            }
#pragma unroll
            for (int i = 0; i < 6; i++) h[i] = br ? h[i] : u[i];
        }
        Fq r; memcpy(r.l, h, sizeof r.l);
        return r;
    }
};
// a + b without the conditional subtraction (< 2p < 2^382: an operand of FqD::mul only)
static inline void fq_add_nored(uint64_t *r, const Fq &a, const Fq &b) {
    uint64_t c = 0;
    for (int i = 0; i < 6; i++) { u128 s = (u128)a.l[i] + b.l[i] + c; r[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
}
// an unreduced Fq2 value
struct Fq2D { FqD c0, c1; };

struct Fq2 {
    Fq c0, c1;
    static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
    static Fq2 one() { return {Fq::one(), Fq::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fq2 &o) const { return c0 == o.c0 && c1 == o.c1; }
    Fq2 operator+(const Fq2 &b) const { return {c0 + b.c0, c1 + b.c1}; }
    Fq2 operator-(const Fq2 &b) const { return {c0 - b.c0, c1 - b.c1}; }
    Fq2 neg() const { return {c0.neg(), c1.neg()}; }
    // Karatsuba on unreduced products: 3 wide products; the caller reduces (2 reductions) or keeps combining.  Operands canonical.
    // Bounds (units of p R): c0 in (0.89, 1.11), c1 < 0.21
    Fq2D mul_wide(const Fq2 &b) const {
        const FqD t0 = FqD::mul(c0.l, b.c0.l), t1 = FqD::mul(c1.l, b.c1.l);
        uint64_t sa[6], sb[6]; fq_add_nored(sa, c0, c1); fq_add_nored(sb, b.c0, b.c1);
        Fq2D r = {t0.sub<1>(t1), FqD::mul(sa, sb, 2, 2).sub<0>(t0).sub<0>(t1)};      // a0 b1 + a1 b0 >= 0
        HCHK(r.c1.ub = 2 * FqD::P_OVER_R;)
        return r;
    }
    // c0 < 0.21, c1 < 0.21
    Fq2D sqr_wide() const {
        uint64_t sa[6]; fq_add_nored(sa, c0, c1);
        const Fq d = c0 - c1;
        const FqD t = FqD::mul(c0.l, c1.l);
        return {FqD::mul(sa, d.l, 2, 1), t + t};
    }
#ifdef HOSTF_EAGER_REDUCTION
    Fq2 operator*(const Fq2 &b) const {
        Fq t0 = c0 * b.c0, t1 = c1 * b.c1, t2 = (c0 + c1) * (b.c0 + b.c1);
        return {t0 - t1, t2 - t0 - t1};
    }
#else
    Fq2 operator*(const Fq2 &b) const { const Fq2D d = mul_wide(b); return {d.c0.redc<2>(), d.c1.redc<1>()}; }
#endif
    Fq2 sqr() const { Fq t = c0 * c1; return {(c0 + c1) * (c0 - c1), t + t}; }
    Fq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    Fq2 inv() const { Fq n = (c0.sqr() + c1.sqr()).inv(); return {c0 * n, (c1 * n).neg()}; }
    Fq2 conj() const { return {c0, c1.neg()}; }
    Fq2 mul_xi() const { return {c0 - c1, c0 + c1}; }          // * (1 + u)
    Fq2 mul_fq(const Fq &k) const { return {c0 * k, c1 * k}; }
};

// Fq6 = Fq2[v]/(v^3 - (1+u)), Fq12 = Fq6[w]/(w^2 - v): host side of the pairing path — combining the per-step line
// products that the GPU returns (dgpu_multi_miller_loop) and the final exponentiation, which the reference runs once
// per batch (utils/src/randomized_pairing_check.rs:213, legogroth16/src/verifier.rs:78).
struct Fq6 {
    Fq2 c0, c1, c2;
    static Fq6 zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    static Fq6 one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
    Fq6 operator+(const Fq6 &b) const { return {c0 + b.c0, c1 + b.c1, c2 + b.c2}; }
    Fq6 operator-(const Fq6 &b) const { return {c0 - b.c0, c1 - b.c1, c2 - b.c2}; }
    Fq6 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
#ifdef HOSTF_EAGER_REDUCTION
    Fq6 operator*(const Fq6 &b) const {
        Fq2 t0 = c0 * b.c0, t1 = c1 * b.c1, t2 = c2 * b.c2;
        Fq2 r0 = ((c1 + c2) * (b.c1 + b.c2) - t1 - t2).mul_xi() + t0;
        Fq2 r1 = (c0 + c1) * (b.c0 + b.c1) - t0 - t1 + t2.mul_xi();
        Fq2 r2 = (c0 + c2) * (b.c0 + b.c2) - t0 - t2 + t1;
        return {r0, r1, r2};
    }
#else
    // the same Karatsuba formulas on unreduced Fq2 products: 18 wide products, 6 reductions.  With t = a_i b_i and M_ij = (a_i + a_j)(b_i + b_j)
    // (c0 parts < 1.11 p R, c1 parts < 0.21 p R) every output component is one signed sum; the offsets cover the subtrahends:
    //   r0 = xi (M12 - t1 - t2) + t0,   r1 = M01 - t0 - t1 + xi t2,   r2 = M02 - t0 - t2 + t1
    Fq6 operator*(const Fq6 &b) const {
        const Fq2D t0 = c0.mul_wide(b.c0), t1 = c1.mul_wide(b.c1), t2 = c2.mul_wide(b.c2);
        const Fq2D m12 = (c1 + c2).mul_wide(b.c1 + b.c2), m01 = (c0 + c1).mul_wide(b.c0 + b.c1), m02 = (c0 + c2).mul_wide(b.c0 + b.c2);
        Fq6 r;
        // r0.c0 = (M12.c0 - t1.c0 - t2.c0) - (M12.c1 - t1.c1 - t2.c1) + t0.c0     < 2.64 + 3
        r.c0.c0 = (m12.c0 + t1.c1 + t2.c1 + t0.c0).sub<3>(t1.c0 + t2.c0 + m12.c1).redc<3>();
        // r0.c1 = (M12.c0 - t1.c0 - t2.c0) + (M12.c1 - t1.c1 - t2.c1) + t0.c1     < 1.53 + 3
        r.c0.c1 = (m12.c0 + m12.c1 + t0.c1).sub<3>(t1.c0 + t2.c0 + t1.c1 + t2.c1).redc<3>();
        // r1.c0 = M01.c0 - t0.c0 - t1.c0 + t2.c0 - t2.c1                           < 2.22 + 3
        r.c1.c0 = (m01.c0 + t2.c0).sub<3>(t0.c0 + t1.c0 + t2.c1).redc<3>();
        // r1.c1 = M01.c1 - t0.c1 - t1.c1 + t2.c0 + t2.c1                           < 1.53 + 1
        r.c1.c1 = (m01.c1 + t2.c0 + t2.c1).sub<1>(t0.c1 + t1.c1).redc<2>();
        // r2.c0 = M02.c0 + t1.c0 - t0.c0 - t2.c0                                    < 2.22 + 3
        r.c2.c0 = (m02.c0 + t1.c0).sub<3>(t0.c0 + t2.c0).redc<3>();
        // r2.c1 = M02.c1 + t1.c1 - t0.c1 - t2.c1                                    < 0.42 + 1
        r.c2.c1 = (m02.c1 + t1.c1).sub<1>(t0.c1 + t2.c1).redc<2>();
        return r;
    }
#endif
    Fq6 mul_v() const { return {c2.mul_xi(), c0, c1}; }
    Fq6 inv() const {
        Fq2 t0 = c0.sqr() - (c1 * c2).mul_xi();
        Fq2 t1 = c2.sqr().mul_xi() - c0 * c1;
        Fq2 t2 = c1.sqr() - c0 * c2;
        Fq2 n = ((c2 * t1 + c1 * t2).mul_xi() + c0 * t0).inv();
        return {t0 * n, t1 * n, t2 * n};
    }
};
struct Fq12 {
    Fq6 c0, c1;
    static Fq12 one() { return {Fq6::one(), Fq6::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    Fq12 operator*(const Fq12 &b) const {
        Fq6 t0 = c0 * b.c0, t1 = c1 * b.c1;
        return {t0 + t1.mul_v(), (c0 + c1) * (b.c0 + b.c1) - t0 - t1};
    }
    // complex squaring: (c0 + c1 w)^2 = (c0 + c1)(c0 + v c1) - c0 c1 - v c0 c1  +  2 c0 c1 w      (2 Fq6 products instead of 3)
    Fq12 sqr() const {
        Fq6 ab = c0 * c1;
        Fq6 t = (c0 + c1) * (c0 + c1.mul_v());
        return {t - ab - ab.mul_v(), ab + ab};
    }
    // Granger-Scott squaring for elements of the cyclotomic subgroup (after the easy part of the final exponentiation):
    // with Fq4 = Fq2[s]/(s^2 - xi), sq4(a, b) = (a^2 + xi b^2, 2ab); 9 Fq2 squarings instead of 12 Fq2 products
    Fq12 cyclotomic_sqr() const {
        const Fq2 &z0 = c0.c0, &z4 = c0.c1, &z3 = c0.c2, &z2 = c1.c0, &z1 = c1.c1, &z5 = c1.c2;
#ifdef HOSTF_EAGER_REDUCTION
        auto sq4 = [](const Fq2 &a, const Fq2 &b, Fq2 &r0, Fq2 &r1) { Fq2 a2 = a.sqr(), b2 = b.sqr(); r0 = a2 + b2.mul_xi(); r1 = (a + b).sqr() - a2 - b2; };
#else
        auto sq4 = [](const Fq2 &a, const Fq2 &b, Fq2 &r0, Fq2 &r1) {       // three unreduced squarings (parts < 0.21 p R), four reductions instead of six
            const Fq2D a2 = a.sqr_wide(), b2 = b.sqr_wide(), s2 = (a + b).sqr_wide();
            r0 = {(a2.c0 + b2.c0).sub<1>(b2.c1).redc<2>(), (a2.c1 + b2.c0 + b2.c1).redc<1>()};           // a^2 + xi b^2
            r1 = {s2.c0.sub<1>(a2.c0 + b2.c0).redc<2>(), s2.c1.sub<1>(a2.c1 + b2.c1).redc<2>()}; };       // (a + b)^2 - a^2 - b^2
#endif
        Fq2 t0, t1, t2, t3, t4, t5;
        sq4(z0, z1, t0, t1); sq4(z2, z3, t2, t3); sq4(z4, z5, t4, t5);
        auto three_minus_two = [](const Fq2 &t, const Fq2 &z) { Fq2 d = t - z; return d + d + t; };   // 3t - 2z
        auto three_plus_two = [](const Fq2 &t, const Fq2 &z) { Fq2 d = t + z; return d + d + t; };    // 3t + 2z
        Fq12 r;
        r.c0.c0 = three_minus_two(t0, z0);
        r.c0.c1 = three_minus_two(t2, z4);
        r.c0.c2 = three_minus_two(t4, z3);
        r.c1.c0 = three_plus_two(t5.mul_xi(), z2);
        r.c1.c1 = three_plus_two(t1, z1);
        r.c1.c2 = three_plus_two(t3, z5);
        return r;
    }
    Fq12 conj() const { return {c0, c1.neg()}; }
    Fq12 inv() const { Fq6 n = (c0 * c0 - (c1 * c1).mul_v()).inv(); return {c0 * n, (c1 * n).neg()}; }
};

// Frobenius coefficients xi^(i (p-1)/6) and their norms, derived at first use
struct FrobTable {
    Fq2 g1[6]; Fq g2[6];
    FrobTable() {
        static constexpr uint64_t E[6] = {0x49aa7ffffffff1c7ULL, 0x051caaaa72e35555ULL, 0xe688231ad3c82906ULL, 0xe613e1eb7deb831fULL, 0x0c849bf3b5e1f223ULL, 0x045582fc5eeaa66fULL};   // (p-1)/6
        Fq2 xi = {Fq::one(), Fq::one()}, acc = Fq2::one(), base = xi;
        for (int i = 0; i < 384; i++) { if ((E[i / 64] >> (i % 64)) & 1) acc = acc * base; base = base.sqr(); }
        g1[0] = Fq2::one();
        for (int i = 1; i < 6; i++) g1[i] = g1[i - 1] * acc;
        for (int i = 0; i < 6; i++) g2[i] = g1[i].c0.sqr() + g1[i].c1.sqr();
    }
};
inline const FrobTable &frob_table() { static const FrobTable t; return t; }
inline Fq12 frob1(const Fq12 &a) {
    const FrobTable &t = frob_table();
    return {{a.c0.c0.conj(), a.c0.c1.conj() * t.g1[2], a.c0.c2.conj() * t.g1[4]}, {a.c1.c0.conj() * t.g1[1], a.c1.c1.conj() * t.g1[3], a.c1.c2.conj() * t.g1[5]}};
}
inline Fq12 frob2(const Fq12 &a) {
    const FrobTable &t = frob_table();
    return {{a.c0.c0, a.c0.c1.mul_fq(t.g2[2]), a.c0.c2.mul_fq(t.g2[4])}, {a.c1.c0.mul_fq(t.g2[1]), a.c1.c1.mul_fq(t.g2[3]), a.c1.c2.mul_fq(t.g2[5])}};
}
constexpr uint64_t BLS_X_ABS = 0xd201000000010000ULL;   // |x|, x < 0
inline Fq12 exp_by_x(const Fq12 &a) {
    Fq12 acc = Fq12::one();
    for (int i = 63; i >= 0; i--) { acc = acc.cyclotomic_sqr(); if ((BLS_X_ABS >> i) & 1) acc = acc * a; }   // a is in the cyclotomic subgroup
    return acc.conj();
}
// ark-ec Bls12::final_exponentiation: easy part, then the x-chain that raises to 3 (p^4 - p^2 + 1)/r
inline bool final_exponentiation(Fq12 &out, const Fq12 &f) {
    if (f.is_zero()) return false;
    Fq12 f1 = f.conj(), f2 = f.inv(), r = f1 * f2; f2 = r;
    r = frob2(r) * f2;
    Fq12 y0 = r.cyclotomic_sqr(), y1 = exp_by_x(r), y2 = r.conj();
    y1 = y1 * y2; y2 = exp_by_x(y1); y1 = y1.conj(); y1 = y1 * y2;
    y2 = exp_by_x(y1); y1 = frob1(y1); y1 = y1 * y2; r = r * y0;
    y0 = exp_by_x(y1); y2 = exp_by_x(y0); y0 = frob2(y1); y1 = y1.conj();
    y1 = y1 * y2; y1 = y1 * y0; r = r * y1;
    out = r; return true;
}

// Extended Jacobian point (x = X/ZZ, y = Y/ZZZ); identity: inf = true.
template <class F> struct HXyzz {
    F x, y, zz, zzz; bool inf;
    static HXyzz identity() { HXyzz r; r.x = F::zero(); r.y = F::zero(); r.zz = F::zero(); r.zzz = F::zero(); r.inf = true; return r; }
    void dbl_in_place() {
        if (inf) return;
        F U = y.dbl(), V = U.sqr(), W = U * V, S = x * V, M = x.sqr(); M = M + M + M;
        F X3 = M.sqr() - S - S, Y3 = M * (S - X3) - W * y;
        x = X3; y = Y3; zz = V * zz; zzz = W * zzz;
    }
    void add_in_place(const HXyzz &b) {
        if (b.inf) return;
        if (inf) { *this = b; return; }
        F U1 = x * b.zz, U2 = b.x * zz, S1 = y * b.zzz, S2 = b.y * zzz;
        F Pd = U2 - U1, Rd = S2 - S1;
        if (Pd.is_zero()) { if (Rd.is_zero()) dbl_in_place(); else *this = identity(); return; }
        F PP = Pd.sqr(), PPP = Pd * PP, Q = U1 * PP;
        F X3 = Rd.sqr() - PPP - Q - Q, Y3 = Rd * (Q - X3) - S1 * PPP;
        x = X3; y = Y3; zz = zz * b.zz * PP; zzz = zzz * b.zzz * PPP;
    }
    // canonical Jacobian triple: (x_aff, y_aff, 1) or (1, 1, 0) for the identity — ark-ec Projective::zero()
    void to_normalised_jacobian(F &X, F &Y, F &Z) const {
        if (inf) { X = F::one(); Y = F::one(); Z = F::zero(); return; }
        F i3 = zzz.inv();            // 1/ZZZ
        F i2 = i3 * i3 * zz * zz;    // ZZ^2/ZZZ^2 = ZZ^2/ZZ^3 = 1/ZZ
        X = x * i2; Y = y * i3; Z = F::one();
    }
};

// ---- Fr (scalar field): Montgomery (R = 2^256) -> canonical, i.e. ark-ff into_bigint ----
static inline void fr_from_mont(uint64_t out[4], const uint64_t a[4]) {
    static constexpr uint64_t MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    static constexpr uint64_t INV = 0xfffffffeffffffffULL;
    uint64_t t[6] = {a[0], a[1], a[2], a[3], 0, 0};
    for (int i = 0; i < 4; i++) {   // four reduction rounds: t = (t + m * MOD) / 2^64
        uint64_t m = t[0] * INV; u128 s = (u128)m * MOD[0] + t[0]; uint64_t c = (uint64_t)(s >> 64);
        for (int j = 1; j < 4; j++) { s = (u128)m * MOD[j] + t[j] + c; t[j - 1] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        s = (u128)t[4] + c; t[3] = (uint64_t)s; t[4] = (uint64_t)(s >> 64);
    }
    bool ge = t[4] != 0;
    if (!ge) { ge = true; for (int i = 3; i >= 0; i--) { if (t[i] > MOD[i]) break; if (t[i] < MOD[i]) { ge = false; break; } } }
    if (ge) { uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - MOD[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }
    for (int i = 0; i < 4; i++) out[i] = t[i];
}

// ---- Fr on the host (domain constants of the witness map: omega_D, g^D - 1, 1/D) ----
struct FrH {
    uint64_t l[4];
    static constexpr uint64_t MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    static constexpr uint64_t R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
    static constexpr uint64_t INV = 0xfffffffeffffffffULL;
    static FrH mont_mul(const FrH &a, const FrH &b) {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            uint64_t c = 0; u128 s;
            for (int j = 0; j < 4; j++) { s = (u128)a.l[j] * b.l[i] + t[j] + c; t[j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
            s = (u128)t[4] + c; t[4] = (uint64_t)s; t[5] = (uint64_t)(s >> 64);
            uint64_t m = t[0] * INV; s = (u128)m * MOD[0] + t[0]; c = (uint64_t)(s >> 64);
            for (int j = 1; j < 4; j++) { s = (u128)m * MOD[j] + t[j] + c; t[j - 1] = (uint64_t)s; c = (uint64_t)(s >> 64); }
            s = (u128)t[4] + c; t[3] = (uint64_t)s; t[4] = t[5] + (uint64_t)(s >> 64);
        }
        bool ge = t[4] != 0;
        if (!ge) { ge = true; for (int i = 3; i >= 0; i--) { if (t[i] > MOD[i]) break; if (t[i] < MOD[i]) { ge = false; break; } } }
        if (ge) { uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - MOD[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }
        FrH r; for (int i = 0; i < 4; i++) r.l[i] = t[i]; return r;
    }
    static FrH from_u64(uint64_t v) { FrH a{{v, 0, 0, 0}}, r2; memcpy(r2.l, R2, 32); return mont_mul(a, r2); }   // Montgomery form
    FrH operator*(const FrH &b) const { return mont_mul(*this, b); }
    FrH pow(const uint64_t e[4]) const {
        FrH acc = from_u64(1), base = *this;
        for (int i = 0; i < 256; i++) { if ((e[i / 64] >> (i % 64)) & 1) acc = acc * base; base = base * base; }
        return acc;
    }
    FrH inv() const { uint64_t e[4] = {MOD[0] - 2, MOD[1], MOD[2], MOD[3]}; return pow(e); }
    FrH sub_one() const {   // this - 1 (Montgomery)
        FrH one = from_u64(1), r; uint64_t br = 0;
        for (int i = 0; i < 4; i++) { u128 d = (u128)l[i] - one.l[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
        if (br) { uint64_t c = 0; for (int i = 0; i < 4; i++) { u128 s = (u128)r.l[i] + MOD[i] + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
        return r;
    }
    void to_canonical(uint64_t out[4]) const { FrH one{{1, 0, 0, 0}}; FrH c = mont_mul(*this, one); memcpy(out, c.l, 32); }
    // 7^((r-1)/2^logn): the radix-2 domain generator ark-poly uses (TWO_ADIC_ROOT_OF_UNITY^(2^(32-logn)))
    static FrH root_of_unity(int logn) {
        uint64_t e[4]; memcpy(e, MOD, 32); e[0] -= 1;
        for (int k = 0; k < logn; k++) { for (int i = 0; i < 3; i++) e[i] = (e[i] >> 1) | (e[i + 1] << 63); e[3] >>= 1; }
        return from_u64(7).pow(e);
    }
};


// ---- GLV decomposition of a G1 scalar: k mod r = k1 + k2 * lambda with k1, k2 < 2^128 ----
// lambda = x^2 - 1 (x the BLS parameter) is a cube root of unity mod r with phi(P) = (beta x, y) = lambda P on G1, and r = lambda^2 + lambda + 1,
// so plain Euclidean division gives both halves below 2^128: k2 = floor(k / lambda) <= r / lambda < 2^128, k1 = k - k2 lambda < lambda < 2^128.
// Barrett division with mu = floor(2^384 / lambda); the quotient estimate is off by at most 2.
static inline void glv_decompose(const uint64_t k_in[4], uint64_t k1[2], uint64_t k2[2]) {
    static constexpr uint64_t RM[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    static constexpr uint64_t LAM[2] = {0x00000000ffffffffULL, 0xac45a4010001a402ULL};
    static constexpr uint64_t MU[5] = {0xda5e4f8d896c72ddULL, 0x389f49a7268bf7a3ULL, 0x63f6e522f6cfee30ULL, 0x7c6becf1e01faaddULL, 0x1ULL};
    uint64_t k[4] = {k_in[0], k_in[1], k_in[2], k_in[3]};
    auto geq_r = [&]() { for (int i = 3; i >= 0; i--) { if (k[i] > RM[i]) return true; if (k[i] < RM[i]) return false; } return true; };
    while (geq_r()) { uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)k[i] - RM[i] - br; k[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }   // at most twice
    uint64_t prod[9] = {0};                                   // k * mu
    for (int i = 0; i < 4; i++) { uint64_t c = 0; for (int j = 0; j < 5; j++) { u128 t = (u128)k[i] * MU[j] + prod[i + j] + c; prod[i + j] = (uint64_t)t; c = (uint64_t)(t >> 64); } prod[i + 5] += c; }
    uint64_t q[3] = {prod[6], prod[7], prod[8]};              // >> 384
    // rem = k - q * lambda  (q < 2^129 in principle; the product fits 4 limbs plus a little)
    auto rem_of = [&](const uint64_t qq[3], uint64_t rem[5]) {
        uint64_t ql[5] = {0};
        for (int i = 0; i < 3; i++) { uint64_t c = 0; for (int j = 0; j < 2; j++) { u128 t = (u128)qq[i] * LAM[j] + ql[i + j] + c; ql[i + j] = (uint64_t)t; c = (uint64_t)(t >> 64); } if (i + 2 < 5) ql[i + 2] += c; }
        uint64_t br = 0;
        for (int i = 0; i < 5; i++) { u128 d = (u128)(i < 4 ? k[i] : 0) - ql[i] - br; rem[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    };
    uint64_t rem[5]; rem_of(q, rem);
    for (int it = 0; it < 4; it++) {                          // while rem >= lambda: q += 1, rem -= lambda
        bool ge = rem[4] == 0 && (rem[3] | rem[2]) != 0;
        if (rem[4] == 0 && !ge) ge = rem[1] > LAM[1] || (rem[1] == LAM[1] && rem[0] >= LAM[0]);
        if (rem[4] != 0 || !ge) break;                        // (rem[4] != 0 would mean an over-estimate, which floor(k mu / 2^384) cannot produce)
        uint64_t c = 1; for (int i = 0; i < 3 && c; i++) { q[i] += c; c = q[i] == 0; }
        uint64_t br = 0; for (int i = 0; i < 5; i++) { u128 d = (u128)rem[i] - (i < 2 ? LAM[i] : 0) - br; rem[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    }
    k1[0] = rem[0]; k1[1] = rem[1]; k2[0] = q[0]; k2[1] = q[1];
}

// Base-|x| digits of k mod r, x = -0xd201000000010000 the BLS parameter: k = d0 + d1 |x| + d2 |x|^2 + d3 |x|^3, d0..d2 < |x|,
// d3 = floor(k / |x|^3) < 2^64 (r < 2^255, |x|^3 > 2^191).  On the prime-order subgroup of the twist |x|^j P = (-psi)^j (P), so the
// four 64-bit chains of k_mul_add_g2_gls replace one 255-bit chain (Galbraith-Lin-Scott decomposition; fixed_kernels.hip.h).
static inline void gls4_decompose(const uint64_t k_in[4], uint64_t d[4]) {
    static constexpr uint64_t RM[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    uint64_t k[4] = {k_in[0], k_in[1], k_in[2], k_in[3]};
    auto geq_r = [&]() { for (int i = 3; i >= 0; i--) { if (k[i] > RM[i]) return true; if (k[i] < RM[i]) return false; } return true; };
    while (geq_r()) { uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 t = (u128)k[i] - RM[i] - br; k[i] = (uint64_t)t; br = (uint64_t)(t >> 64) & 1; } }   // at most twice
    for (int j = 0; j < 3; j++) {                              // k, d[j] = divmod(k, |x|)
        u128 rem = 0;
        for (int i = 3; i >= 0; i--) { const u128 cur = (rem << 64) | k[i]; k[i] = (uint64_t)(cur / BLS_X_ABS); rem = cur % BLS_X_ABS; }
        d[j] = (uint64_t)rem;
    }
    d[3] = k[0];                                               // (k[1..3] == 0 here)
}

}  // namespace hostf
