// crypto_amd/csrc/fp30s.hip.h — BLS12-381 base field for the G1 MSM kernels: 13 SIGNED limbs of 30 bits.
//
// Same design rules as fp29.hip.h (one multiply-add per limb product into a 64-bit column accumulator, lazy additions, no conditional
// subtraction, no carry chain on the hot path), different radix: 13 x 30 = 390 bits hold p (381 bits) with 9 bits to spare, so a Montgomery
// product is 2 * 13^2 = 338 v_mad_i64_i32 instead of 2 * 14^2 = 392 v_mad_u64_u32 (-13.8 %), 25 column hand-offs instead of 27.  What makes
// 30-bit limbs fit a 64-bit accumulator is the SIGN: digits are kept balanced, |d| <= 2^29, so a column of 13 operand products + 13
// reduction products is bounded by 26 * 2^58 < 2^63 in magnitude (unsigned 30-bit digits would need 2^64.7).  Measured on MI355X at the
// accumulation kernel's occupancy (tools/ubench/fp30_rate.hip): 82.9 G products/s against 69.1 for the 14 x 29-bit field (+20 %).
//
// Representation: value = sum l[i] * 2^(30 i), l[i] signed; Montgomery form x * 2^390 mod p; any representative of the residue class
// with |value| < 2^6 p may occur (negative values included: there is no "multiple of p that dominates the subtrahend", a subtraction is
// 13 v_sub_u32 and a negation is free).
// Digit classes used below:
//   B : |l[i]| <= 2^29 + 8 for i < 12 ("balanced": output of mul / sqr / bal)           — may enter a product on both sides
//   D : |l[i]| <  2^31            (sums / differences of a few B values)                 — may enter a product against a B operand if the
//                                                                                           worst-case column sum stays below 2^63 (checked)
// With -DFP29_CHECK (host only) every element carries worst-case digit-magnitude and value bounds, propagated by each operation
// independently of the data and asserted against the preconditions: one execution of a formula proves it overflow-free for all inputs of
// the same classes (tests/test_device_code_on_host.py).
#pragma once
#include "fp29.hip.h"

namespace bls29 {

constexpr int SN = 13;
constexpr int SB = 30;
constexpr uint32_t SMASK = (1u << SB) - 1;
constexpr int32_t SHALF = 1 << (SB - 1);
constexpr uint32_t SINV30 = 0x3ffcfffdu;    // -p^-1 mod 2^30
constexpr uint32_t SPINV30 = 0x30003u;      //  p^-1 mod 2^30

#define BLS30_P     {-21845, -402915328, 356515836, -352321620, -252304353, 55215067, 288093811, 316751073, -321428361, 517541167, -375082566, -91332614, 1704210}
#define BLS30_PU    {0x3fffaaabu, 0x27fbffffu, 0x153ffffbu, 0x2affffacu, 0x30f6241eu, 0x34a83dau, 0x112bf673u, 0x12e13ce1u, 0x2cd76477u, 0x1ed90d2eu, 0x29a4b1bau, 0x3a8e5ff9u, 0x1a0111u}
#define BLS30_ONE   {13762350, 433586176, -192935228, -301937177, 37952645, -425753694, -36732706, 162803105, -437337492, 366579475, 78814996, -442511456, 89578}
#define BLS30_CIN   {-192885889, -32767999, -532500185, -13403472, -35343150, 503405810, 6050829, -194530918, -181768796, 433626394, -273043482, -129447912, 620336}
#define BLS30_COUT  {196605, 405012480, 12582951, -50330895, 123255532, -496935601, -445360651, 370465813, -328370226, -362903204, 154517618, -251748295, 1439327}
#define BLS29_C422  {0x1a7f6bafu, 0x1e281d51u, 0x19fe47cbu, 0x16b3408eu, 0x136783f9u, 0xd3c8407u, 0x1e61f415u, 0x15deb981u, 0x10c53570u, 0x67aa9du, 0x256511cu, 0x8eb4399u, 0x4f0125u, 0x2u}
#define BLS29_C390  {0xd1ff2eu, 0x13b00000u, 0x12002b11u, 0x10066f36u, 0x431c84bu, 0x13f07441u, 0x13e03766u, 0x1a16d07bu, 0xec26c26u, 0x131e252fu, 0xa7c515du, 0x1e7d0096u, 0x15de9967u, 0x0u}

struct Fs {
    int32_t l[SN];
#ifdef FP29_CHECK
    uint64_t ubn[SN], ubp[SN];   // worst-case magnitude of each digit on the negative / positive side (a product's digits are [-2^29, 2^29 - 1])
    double vb;                   // worst-case |value|, in units of p
#endif
};

#ifdef FP29_CHECK
static const double P_OVER_2_360 = 1704209.92;    // p / 2^360 (rounded up)
static const double P_OVER_2_390 = 0.00158717;    // p / 2^390 (rounded up)
inline void schk_set_B(Fs &r, double vb) {
    for (int i = 0; i < SN - 1; i++) { r.ubn[i] = (uint64_t)SHALF; r.ubp[i] = (uint64_t)SHALF - 1; }
    r.ubn[SN - 1] = r.ubp[SN - 1] = (uint64_t)floor(vb * P_OVER_2_360) + 2;      // the lower digits are balanced: |sum| < 2^359.1, i.e. at most one unit of the top digit
    r.vb = vb;
}
inline uint64_t smag(const Fs &a, int i) { return a.ubn[i] > a.ubp[i] ? a.ubn[i] : a.ubp[i]; }
inline void schk_actual(const Fs &a) { for (int i = 0; i < SN; i++) { int64_t v = a.l[i]; assert(v < 0 ? (uint64_t)(-v) <= a.ubn[i] : (uint64_t)v <= a.ubp[i]); } }
inline void schk_fits(const Fs &a) { for (int i = 0; i < SN; i++) { assert(a.ubn[i] <= (1ull << 31)); assert(a.ubp[i] < (1ull << 31)); } }
// worst-case column sums of a (sum of) product(s) with the reduction products and the carry must stay below 2^63
inline void schk_columns(const Fs *const *a, const Fs *const *b, int nprod) {
    constexpr int32_t P_[SN] = BLS30_P;
    unsigned __int128 carry = 0;
    for (int k = 0; k < 2 * SN - 1; k++) {
        unsigned __int128 s = carry + ((unsigned __int128)1 << 29) + (k >= SN ? (unsigned __int128)1 << 59 : 0);      // (the rounding bias of the output columns; FS_ALT_HIGH: every other one also carries the next column's)
        for (int q = 0; q < nprod; q++)
            for (int i = 0; i < SN; i++) { int j = k - i; if (j < 0 || j >= SN) continue; s += (unsigned __int128)smag(*a[q], i) * smag(*b[q], j); }
        for (int i = 0; i < SN; i++) { int j = k - i; if (j < 0 || j >= SN) continue; s += (unsigned __int128)SHALF * (uint64_t)(P_[j] < 0 ? -(int64_t)P_[j] : P_[j]); }
        assert(s < ((unsigned __int128)1 << 63));
        carry = (s >> SB) + 1;
    }
}
#define SCHK(...) __VA_ARGS__
#else
#define SCHK(...)
#endif

FD int32_t sext30(uint32_t x) { return (int32_t)(x << 2) >> 2; }
// The multiply-add chain of an output column, started from the bias constant 2^29.  Written as `x * y + 2^29` the compiler reassociates the sum
// and the constant ends up in a 64-bit addition of its own after the chain, 12 per product.  Forcing it into the addend of the chain's first
// v_mad_i64_i32 with one inline-assembly statement per multiply-add removes those additions and was measured 1 % SLOWER on the mixed addition —
// the compiler schedules its own chains better than opaque ones, tools/ubench/madd_rate.hip.  What works is hiding only the CONSTANT: an empty
// asm statement that claims to modify an SGPR pair holding 2^29 makes it an ordinary loop-invariant value, the chain's first multiply-add takes
// it as its addend (src2 from the scalar registers) and every other instruction stays visible to the scheduler: 324 -> 237 v_lshl_add_u64 per
// mixed addition, 4416 -> 4329 instructions.
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_PIN(x) asm("" : "+v"(x))
#else
#define FS_PIN(x)
#endif
#ifdef FS_SERIAL_LOW
#define FS_PIN_LOW(x) FS_PIN(x)
#else
#define FS_PIN_LOW(x)
#endif
// FS_ALT_HIGH: an output column with an even index starts its chain from 2^29 + 2^59 — its own rounding bias and, through the carry
// ((s + 2^59) >> 30 = (s >> 30) + 2^29, the low 30 bits untouched), the next column's, which then needs no chain of its own
constexpr int64_t FS_BIAS2 = (int64_t)SHALF + ((int64_t)1 << 59);
struct FsChain {
    int64_t v;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FS_NO_BIAS_PIN)
    FD FsChain(int32_t x, int32_t y, int64_t bias = (int64_t)SHALF) { int64_t b = bias; asm("" : "+s"(b)); v = (int64_t)x * y + b; }
#else
    FD FsChain(int32_t x, int32_t y, int64_t bias = (int64_t)SHALF) : v((int64_t)x * y + bias) {}
#endif
    FD void vv(int32_t x, int32_t y) { v += (int64_t)x * y; }
    FD void vs(int32_t x, int32_t k) { v += (int64_t)x * k; }
};

FD void fs_zero(Fs &r) {
#pragma unroll
    for (int i = 0; i < SN; i++) r.l[i] = 0;
    SCHK(for (int i = 0; i < SN; i++) r.ubn[i] = r.ubp[i] = 0; r.vb = 0;)
}
FD void fs_set_one(Fs &r) {
    constexpr int32_t O_[SN] = BLS30_ONE;
#pragma unroll
    for (int i = 0; i < SN; i++) r.l[i] = O_[i];
    SCHK(schk_set_B(r, 1.0);)
}
FD void fs_add(Fs &r, const Fs &a, const Fs &b) {
    SCHK(for (int i = 0; i < SN; i++) { r.ubn[i] = a.ubn[i] + b.ubn[i]; r.ubp[i] = a.ubp[i] + b.ubp[i]; } r.vb = a.vb + b.vb; schk_fits(r);)
#pragma unroll
    for (int i = 0; i < SN; i++) r.l[i] = a.l[i] + b.l[i];
}
FD void fs_sub(Fs &r, const Fs &a, const Fs &b) {
    SCHK(for (int i = 0; i < SN; i++) { uint64_t n = a.ubn[i] + b.ubp[i], q = a.ubp[i] + b.ubn[i]; r.ubn[i] = n; r.ubp[i] = q; } r.vb = a.vb + b.vb; schk_fits(r);)
#pragma unroll
    for (int i = 0; i < SN; i++) r.l[i] = a.l[i] - b.l[i];
}
FD void fs_neg(Fs &r, const Fs &a) {
    SCHK(for (int i = 0; i < SN; i++) { uint64_t n = a.ubp[i], q = a.ubn[i]; r.ubn[i] = n; r.ubp[i] = q; } r.vb = a.vb; schk_fits(r);)
#pragma unroll
    for (int i = 0; i < SN; i++) r.l[i] = -a.l[i];
}
// r = neg ? -a : a without control flow: (d ^ m) - m with m = -neg.  Digit classes are symmetric, so the class of a is the class of r.
FD void fs_cond_neg(Fs &r, const Fs &a, bool neg) {
    SCHK(for (int i = 0; i < SN; i++) { uint64_t w = a.ubn[i] > a.ubp[i] ? a.ubn[i] : a.ubp[i]; r.ubn[i] = w; r.ubp[i] = w; } r.vb = a.vb; schk_fits(r);)
    const int32_t m = neg ? -1 : 0;
#pragma unroll
    for (int i = 0; i < SN; i++) r.l[i] = (a.l[i] ^ m) - m;
}
// one parallel carry pass: class D -> class B (|digit| <= 2^29 + carry of the neighbour); value unchanged.  Precondition: digit + 2^29 fits an int32.
FD void fs_bal(Fs &r, const Fs &a) {
    int32_t c[SN];
    uint32_t u[SN];
#pragma unroll
    for (int i = 0; i < SN - 1; i++) { u[i] = (uint32_t)a.l[i] + (uint32_t)SHALF; c[i] = (int32_t)u[i] >> SB; }
    const int32_t tl = a.l[SN - 1] + c[SN - 2];
#pragma unroll
    for (int i = SN - 2; i >= 1; i--) r.l[i] = (int32_t)(u[i] & SMASK) - SHALF + c[i - 1];
    r.l[0] = (int32_t)(u[0] & SMASK) - SHALF;
    r.l[SN - 1] = tl;
#ifdef FP29_CHECK
    {
        uint64_t cn[SN], cp[SN];       // carry magnitudes, negative / positive side
        for (int i = 0; i < SN - 1; i++) {
            assert(a.ubp[i] + (uint64_t)SHALF < (1ull << 31)); assert(a.ubn[i] <= (1ull << 31));
            cp[i] = (a.ubp[i] + (uint64_t)SHALF) >> SB; cn[i] = (a.ubn[i] + (uint64_t)SHALF - 1) >> SB;      // ceil((n - 2^29) / 2^30) for the negative side
            assert(cp[i] <= 8 && cn[i] <= 8);
        }
        const uint64_t tn = a.ubn[SN - 1] + cn[SN - 2], tp = a.ubp[SN - 1] + cp[SN - 2];
        const double vb = a.vb;
        for (int i = 1; i < SN - 1; i++) { r.ubn[i] = (uint64_t)SHALF + cn[i - 1]; r.ubp[i] = (uint64_t)SHALF - 1 + cp[i - 1]; }
        r.ubn[0] = (uint64_t)SHALF; r.ubp[0] = (uint64_t)SHALF - 1;
        const uint64_t topv = (uint64_t)floor(vb * P_OVER_2_360) + 2;
        r.ubn[SN - 1] = tn < topv ? tn : topv; r.ubp[SN - 1] = tp < topv ? tp : topv;
        r.vb = vb;
        schk_actual(r);
    }
#endif
}
// the same for digits of up to 31 bits (the four-term combination R^2 - PPP - 2 Q of the addition formulas): floor carry, then the remainder is
// folded into the balanced range; two more instructions per digit than fs_bal
FD void fs_bal_wide(Fs &r, const Fs &a) {
    int32_t c[SN], d[SN];
#pragma unroll
    for (int i = 0; i < SN - 1; i++) {
        const uint32_t rem = (uint32_t)a.l[i] & SMASK;            // in [0, 2^30)
        const int32_t h = (int32_t)(rem >> (SB - 1));             // 1 if the remainder belongs to the negative half
        c[i] = (a.l[i] >> SB) + h;
        d[i] = (int32_t)rem - (h << SB);
    }
    const int32_t tl = a.l[SN - 1] + c[SN - 2];
#pragma unroll
    for (int i = SN - 2; i >= 1; i--) r.l[i] = d[i] + c[i - 1];
    r.l[0] = d[0];
    r.l[SN - 1] = tl;
#ifdef FP29_CHECK
    {
        schk_fits(a);
        uint64_t cn[SN], cp[SN];
        for (int i = 0; i < SN - 1; i++) { cp[i] = (a.ubp[i] >> SB) + 1; cn[i] = (a.ubn[i] + SMASK) >> SB; assert(cp[i] <= 8 && cn[i] <= 8); }
        const uint64_t tn = a.ubn[SN - 1] + cn[SN - 2], tp = a.ubp[SN - 1] + cp[SN - 2];
        const double vb = a.vb;
        for (int i = 1; i < SN - 1; i++) { r.ubn[i] = (uint64_t)SHALF + cn[i - 1]; r.ubp[i] = (uint64_t)SHALF - 1 + cp[i - 1]; }
        r.ubn[0] = (uint64_t)SHALF; r.ubp[0] = (uint64_t)SHALF - 1;
        const uint64_t topv = (uint64_t)floor(vb * P_OVER_2_360) + 2;
        r.ubn[SN - 1] = tn < topv ? tn : topv; r.ubp[SN - 1] = tp < topv ? tp : topv;
        r.vb = vb;
        schk_actual(r);
    }
#endif
}

// Montgomery product, r = a b / 2^390 mod p; result class B, |value| < p (1/2 + |a||b| p / 2^390 + ...).
// Product scanning with the reduction interleaved.  (-DFS_SERIAL_LOW / -DFS_ALT_HIGH: the carry-seeded column chains of DESIGN.md section 10 — 195 fewer 64-bit
// additions per mixed addition, no faster on MI355X and slower in the latency-bound kernels; off by default, kept for the A/B.)  Output digit j is taken from column 13 + j: the column sum carries a bias of 2^29 (a constant
// folded into the start of the column's multiply-add chain), so that  (low 30 bits) - 2^29  is the balanced digit and the arithmetic shift is the carry.
FD void fs_mul(Fs &r, const Fs &a, const Fs &b) {
    constexpr int32_t P_[SN] = BLS30_P;
    SCHK({ const Fs *pa[1] = {&a}, *pb[1] = {&b}; schk_columns(pa, pb, 1); schk_actual(a); schk_actual(b); })
    int32_t m[SN], t[SN];
    int64_t acc = 0;
    // every column is written as its own chain of multiply-adds (started from the bias constant in the output half) that is joined with the
    // carry of the previous column by ONE 64-bit addition: the shape the compiler schedules best (the chains of neighbouring columns overlap)
#pragma unroll
    for (int k = 0; k < SN; k++) {
#ifdef FS_SERIAL_LOW
        // the carry of column k - 1 is the addend of this column's first multiply-add: no 64-bit addition to join them
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (int64_t)a.l[i] * b.l[k - i]; FS_PIN(acc); }
#pragma unroll
        for (int i = 0; i < k; i++) { acc += (int64_t)m[i] * P_[k - i]; FS_PIN(acc); }
#else
        int64_t part = 0;
#pragma unroll
        for (int i = 0; i <= k; i++) part += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) part += (int64_t)m[i] * P_[k - i];
        acc += part;
#endif
        m[k] = sext30((uint32_t)acc * SINV30);
        acc += (int64_t)m[k] * P_[0];
        acc >>= SB;
    }
#pragma unroll
    for (int k = SN; k < 2 * SN - 1; k++) {
#ifdef FS_ALT_HIGH
        if ((k - SN) & 1) {        // the carry brought this column's bias along (FS_BIAS2 one column earlier): the column continues the carry, no addition
#pragma unroll
            for (int i = k - SN + 1; i < SN; i++) { acc += (int64_t)a.l[i] * b.l[k - i]; FS_PIN(acc); }
#pragma unroll
            for (int i = k - SN + 1; i < SN; i++) { acc += (int64_t)m[i] * P_[k - i]; FS_PIN(acc); }
        } else {
            FsChain part(a.l[k - SN + 1], b.l[SN - 1], FS_BIAS2);
#pragma unroll
            for (int i = k - SN + 2; i < SN; i++) part.vv(a.l[i], b.l[k - i]);
#pragma unroll
            for (int i = k - SN + 1; i < SN; i++) part.vs(m[i], P_[k - i]);
            acc += part.v;
        }
#else
        FsChain part(a.l[k - SN + 1], b.l[SN - 1]);
#pragma unroll
        for (int i = k - SN + 2; i < SN; i++) part.vv(a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - SN + 1; i < SN; i++) part.vs(m[i], P_[k - i]);
        acc += part.v;
#endif
        t[k - SN] = (int32_t)((uint32_t)acc & SMASK) - SHALF;
        acc >>= SB;
    }
    t[SN - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < SN; i++) r.l[i] = t[i];
    SCHK(schk_set_B(r, 0.51 + a.vb * b.vb * P_OVER_2_390); schk_actual(r);)
}

// Fused r = (a b + c d) / 2^390 mod p: two operand products share one reduction (507 instead of 676 multiply-adds).  All four operands class B.
FD void fs_mul2(Fs &r, const Fs &a, const Fs &b, const Fs &c, const Fs &d) {
    constexpr int32_t P_[SN] = BLS30_P;
    SCHK({ const Fs *pa[2] = {&a, &c}, *pb[2] = {&b, &d}; schk_columns(pa, pb, 2); schk_actual(a); schk_actual(b); schk_actual(c); schk_actual(d); })
    int32_t m[SN], t[SN];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < SN; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (int64_t)a.l[i] * b.l[k - i]; FS_PIN_LOW(acc); acc += (int64_t)c.l[i] * d.l[k - i]; FS_PIN_LOW(acc); }
#pragma unroll
        for (int i = 0; i < k; i++) { acc += (int64_t)m[i] * P_[k - i]; FS_PIN_LOW(acc); }
        m[k] = sext30((uint32_t)acc * SINV30);
        acc += (int64_t)m[k] * P_[0];
        acc >>= SB;
    }
#pragma unroll
    for (int k = SN; k < 2 * SN - 1; k++) {
#ifdef FS_ALT_HIGH
        if ((k - SN) & 1) {
#pragma unroll
            for (int i = k - SN + 1; i < SN; i++) { acc += (int64_t)a.l[i] * b.l[k - i]; FS_PIN(acc); acc += (int64_t)c.l[i] * d.l[k - i]; FS_PIN(acc); }
#pragma unroll
            for (int i = k - SN + 1; i < SN; i++) { acc += (int64_t)m[i] * P_[k - i]; FS_PIN(acc); }
        } else {
            FsChain part(a.l[k - SN + 1], b.l[SN - 1], FS_BIAS2);
            part.vv(c.l[k - SN + 1], d.l[SN - 1]);
#pragma unroll
            for (int i = k - SN + 2; i < SN; i++) { part.vv(a.l[i], b.l[k - i]); part.vv(c.l[i], d.l[k - i]); }
#pragma unroll
            for (int i = k - SN + 1; i < SN; i++) part.vs(m[i], P_[k - i]);
            acc += part.v;
        }
#else
        FsChain part(a.l[k - SN + 1], b.l[SN - 1]);
        part.vv(c.l[k - SN + 1], d.l[SN - 1]);
#pragma unroll
        for (int i = k - SN + 2; i < SN; i++) { part.vv(a.l[i], b.l[k - i]); part.vv(c.l[i], d.l[k - i]); }
#pragma unroll
        for (int i = k - SN + 1; i < SN; i++) part.vs(m[i], P_[k - i]);
        acc += part.v;
#endif
        t[k - SN] = (int32_t)((uint32_t)acc & SMASK) - SHALF;
        acc >>= SB;
    }
    t[SN - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < SN; i++) r.l[i] = t[i];
    SCHK(schk_set_B(r, 0.51 + (a.vb * b.vb + c.vb * d.vb) * P_OVER_2_390); schk_actual(r);)
}

// r = a^2 / 2^390 mod p: 91 + 169 multiply-adds.  a class B.
FD void fs_sqr(Fs &r, const Fs &a) {
    constexpr int32_t P_[SN] = BLS30_P;
    SCHK({ const Fs *pa[1] = {&a}, *pb[1] = {&a}; schk_columns(pa, pb, 1); schk_actual(a); })
    int32_t m[SN], t[SN], a2[SN];
#pragma unroll
    for (int i = 0; i < SN; i++) a2[i] = a.l[i] * 2;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < SN; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) { acc += (int64_t)a.l[i] * a2[k - i]; FS_PIN_LOW(acc); }
        if ((k & 1) == 0) { acc += (int64_t)a.l[k / 2] * a.l[k / 2]; FS_PIN_LOW(acc); }
#pragma unroll
        for (int i = 0; i < k; i++) { acc += (int64_t)m[i] * P_[k - i]; FS_PIN_LOW(acc); }
        m[k] = sext30((uint32_t)acc * SINV30);
        acc += (int64_t)m[k] * P_[0];
        acc >>= SB;
    }
#pragma unroll
    for (int k = SN; k < 2 * SN - 1; k++) {
#ifdef FS_ALT_HIGH
        if ((k - SN) & 1) {
#pragma unroll
            for (int i = k - SN + 1; i < SN; i++) { acc += (int64_t)m[i] * P_[k - i]; FS_PIN(acc); }
#pragma unroll
            for (int i = k - SN + 1; 2 * i < k; i++) { acc += (int64_t)a.l[i] * a2[k - i]; FS_PIN(acc); }
            if ((k & 1) == 0) { acc += (int64_t)a.l[k / 2] * a.l[k / 2]; FS_PIN(acc); }
        } else {
            FsChain part(m[k - SN + 1], P_[SN - 1], FS_BIAS2);
#pragma unroll
            for (int i = k - SN + 2; i < SN; i++) part.vs(m[i], P_[k - i]);
#pragma unroll
            for (int i = k - SN + 1; 2 * i < k; i++) part.vv(a.l[i], a2[k - i]);
            if ((k & 1) == 0) part.vv(a.l[k / 2], a.l[k / 2]);
            acc += part.v;
        }
#else
        FsChain part(m[k - SN + 1], P_[SN - 1]);         // (started from a reduction product: the last column has no off-diagonal operand product)
#pragma unroll
        for (int i = k - SN + 2; i < SN; i++) part.vs(m[i], P_[k - i]);
#pragma unroll
        for (int i = k - SN + 1; 2 * i < k; i++) part.vv(a.l[i], a2[k - i]);
        if ((k & 1) == 0) part.vv(a.l[k / 2], a.l[k / 2]);
        acc += part.v;
#endif
        t[k - SN] = (int32_t)((uint32_t)acc & SMASK) - SHALF;
        acc >>= SB;
    }
    t[SN - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < SN; i++) r.l[i] = t[i];
    SCHK(schk_set_B(r, 0.51 + a.vb * a.vb * P_OVER_2_390); schk_actual(r);)
}

// ---- exact (slow-path) helpers: canonical representative in [0, p) as 13 unsigned 30-bit digits ----
FD void fs_canon(uint32_t t[SN], const Fs &a) {
    constexpr uint32_t PU_[SN] = BLS30_PU;
    SCHK(assert(a.vb < 1024.0);)
    // value + 2^10 p >= 0: sequential signed carry propagation, then conditional subtraction of 2^j p for j = 11 .. 0
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < SN - 1; i++) { c += (int64_t)a.l[i] + ((int64_t)PU_[i] << 10); t[i] = (uint32_t)c & SMASK; c >>= SB; }
    c += (int64_t)a.l[SN - 1] + ((int64_t)PU_[SN - 1] << 10);
    t[SN - 1] = (uint32_t)c;                         // < 2^11 p / 2^360 < 2^32
    for (int j = 11; j >= 0; j--) {
        uint32_t q[SN];
        uint64_t cc = 0;
#pragma unroll
        for (int i = 0; i < SN - 1; i++) { cc += ((uint64_t)PU_[i] << j); q[i] = (uint32_t)cc & SMASK; cc >>= SB; }
        cc += ((uint64_t)PU_[SN - 1] << j);
        q[SN - 1] = (uint32_t)cc;
        bool ge = true, decided = false;
#pragma unroll
        for (int i = SN - 1; i >= 0; i--) { if (!decided && t[i] != q[i]) { ge = t[i] > q[i]; decided = true; } }
        if (ge) {
            int64_t b2 = 0;
#pragma unroll
            for (int i = 0; i < SN - 1; i++) { int64_t v = (int64_t)t[i] - (int64_t)q[i] + b2; t[i] = (uint32_t)v & SMASK; b2 = v >> SB; }
            t[SN - 1] = (uint32_t)((int64_t)t[SN - 1] - (int64_t)q[SN - 1] + b2);
        }
    }
}
FD bool fs_is_zero_exact(const Fs &a) {
    uint32_t t[SN]; fs_canon(t, a);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < SN; i++) o |= t[i];
    return o == 0;
}
// cheap necessary condition for a == 0 mod p when |value| < 64 p: a = k p  =>  k = l0 p^-1 mod 2^30 with |k| < 64
FD bool fs_maybe_zero(const Fs &a) { const int32_t k = sext30((uint32_t)a.l[0] * SPINV30); return k > -64 && k < 64; }

// ---- C-ABI form (6 x u64 little-endian, value * 2^384 mod p) <-> Fs ----
FD void fs_from_abi(Fs &r, const uint32_t w[12]) {
    constexpr int32_t CIN_[SN] = BLS30_CIN;
    Fs t, cin;
#pragma unroll
    for (int i = 0; i < SN; i++) {
        const int bit = i * SB, wi = bit >> 5, sh = bit & 31;
        uint64_t v = (uint64_t)w[wi] >> sh;
        if (wi + 1 < 12) v |= ((uint64_t)w[wi + 1] << (32 - sh));
        t.l[i] = (int32_t)((uint32_t)v & SMASK);
    }
    t.l[SN - 1] = (int32_t)(w[11] >> ((12 * SB) & 31));     // bits 360 .. 383
#pragma unroll
    for (int i = 0; i < SN; i++) cin.l[i] = CIN_[i];
    SCHK(for (int i = 0; i < SN; i++) { t.ubn[i] = 0; t.ubp[i] = SMASK; } t.ubp[SN - 1] = (1u << 24) - 1; t.vb = 10.0; schk_set_B(cin, 1.0);)
    Fs tb; fs_bal(tb, t);
    fs_mul(r, tb, cin);          // x 2^384 * 2^396 / 2^390 = x 2^390
}
FD void fs_to_abi(uint32_t w[12], const Fs &a) {
    constexpr int32_t COUT_[SN] = BLS30_COUT;
    Fs ab, t, cout;
#pragma unroll
    for (int i = 0; i < SN; i++) cout.l[i] = COUT_[i];
    SCHK(schk_set_B(cout, 1.0);)
    fs_bal(ab, a);
    fs_mul(t, ab, cout);         // x 2^390 * 2^384 / 2^390 = x 2^384
    uint32_t c[SN]; fs_canon(c, t);
    uint32_t o[12];
#pragma unroll
    for (int i = 0; i < 12; i++) o[i] = 0;
#pragma unroll
    for (int i = 0; i < SN; i++) {
        const int bit = i * SB, wi = bit >> 5, sh = bit & 31;
        if (wi < 12) o[wi] |= c[i] << sh;
        if (sh + SB > 32 && wi + 1 < 12) o[wi + 1] |= c[i] >> (32 - sh);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = o[i];
}

// ---- Fs <-> Fp (the 14 x 29-bit field): used where a kernel needs the division-step inversion, which is written for Fp ----
FD void fp_from_fs(Fp &r, const Fs &a) {
    constexpr uint32_t C_[NL] = BLS29_C422;
    uint32_t c[SN]; fs_canon(c, a);
    Fp t, k;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        // bits [29 i, 29 i + 29) of the 390-bit string made of 13 30-bit digits
        const int bit = i * LB, di = bit / SB, sh = bit % SB;
        uint64_t v = (uint64_t)c[di] >> sh;
        if (di + 1 < SN) v |= ((uint64_t)c[di + 1] << (SB - sh));
        t.l[i] = (uint32_t)v & LMASK;
    }
#pragma unroll
    for (int i = 0; i < NL; i++) k.l[i] = C_[i];
    CHK(chk_set_N(t, 1.0); chk_set_N(k, 1.0);)
    fp_mul(r, t, k);             // x 2^390 * 2^422 / 2^406 = x 2^406
}
FD void fs_from_fp(Fs &r, const Fp &a) {
    constexpr uint32_t C_[NL] = BLS29_C390;
    Fp an, t, k, c;
#pragma unroll
    for (int i = 0; i < NL; i++) k.l[i] = C_[i];
    CHK(chk_set_N(k, 1.0);)
    fp_norm(an, a);
    fp_mul(t, an, k);            // x 2^406 * 2^390 / 2^406 = x 2^390
    fp_canon(c, t);
    Fs u;
#pragma unroll
    for (int i = 0; i < SN; i++) {
        const int bit = i * SB, di = bit / LB, sh = bit % LB;
        uint64_t v = (uint64_t)c.l[di] >> sh;
        if (di + 1 < NL) v |= ((uint64_t)c.l[di + 1] << (LB - sh));
        if (di + 2 < NL) v |= ((uint64_t)c.l[di + 2] << (2 * LB - sh));
        u.l[i] = (int32_t)((uint32_t)v & SMASK);
    }
    SCHK(for (int i = 0; i < SN; i++) { u.ubn[i] = 0; u.ubp[i] = SMASK; } u.ubp[SN - 1] = 1704210; u.vb = 1.0;)
    fs_bal(r, u);
}

// ---- uniform spellings used by the field-generic group law (ec29.hip.h) ----
FD void fcond_neg(Fs &y, bool neg) { fs_cond_neg(y, y, neg); }
FD void fzero(Fs &r) { fs_zero(r); }
FD void fset_one(Fs &r) { fs_set_one(r); }
FD void fadd(Fs &r, const Fs &a, const Fs &b) { fs_add(r, a, b); }
FD void fdbl(Fs &r, const Fs &a) { fs_add(r, a, a); }
template <int M> FD void fsub(Fs &r, const Fs &a, const Fs &b) { fs_sub(r, a, b); }      // (no multiple of p needed: digits are signed)
FD void fnorm(Fs &r, const Fs &a) { fs_bal(r, a); }
FD void fnormw(Fs &r, const Fs &a) { fs_bal_wide(r, a); }
FD void fmul(Fs &r, const Fs &a, const Fs &b) { fs_mul(r, a, b); }
FD void fsqr(Fs &r, const Fs &a) { fs_sqr(r, a); }
// r = a b - c d (class B): all four operands class B
template <int M> FD void fmul_sub(Fs &r, const Fs &a, const Fs &b, const Fs &c, const Fs &d) { Fs cn; fs_neg(cn, c); fs_mul2(r, a, b, cn, d); }
FD bool fmaybe_zero(const Fs &a) { return fs_maybe_zero(a); }
FD bool fis_zero_exact(const Fs &a) { return fs_is_zero_exact(a); }

}  // namespace bls29
