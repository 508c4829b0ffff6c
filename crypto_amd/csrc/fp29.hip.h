// crypto_amd/csrc/fp29.hip.h — BLS12-381 base field for gfx950, carry-free lazy representation.
//
// Replaces (on the device) what the reference reaches as ark_ff::Fp<MontBackend<FqConfig,6>,6>
// (third-party ark-ff 0.4; call sites e.g. legogroth16/src/prover.rs:286 via G1::msm_bigint).
//
// Why this shape (measured on MI355X, profiles/r01_instr_rate_ubench.txt): v_mad_u64_u32 issues at
// ~13 lanes/clk/SIMD and v_add_co/v_addc_co are no faster, so a 12x32-bit carry-chain Montgomery
// product costs ~2 instructions per limb product.  Here an element is 14 limbs of 29 bits held in
// 32-bit registers with 3 spare bits: every limb product is ONE v_mad_u64_u32 into a 64-bit column
// accumulator (28 products of < 2^59 fit), additions are 14 plain v_add_u32, subtractions add a
// limb-dominating multiple of p, and nothing is reduced below "a few p" until the value leaves the
// device (Montgomery radix 2^406 >> p gives 25 bits of slack).
//
// Representation: value = sum l[i] * 2^(29 i); Montgomery form x * 2^406 mod p.
// Limb classes used in the comments below:
//   N  : l[i] <= 2^29 + 7 for i < 13 (output of mul/sqr/norm)
//   L  : l[i] <  2^32 (anything; only norm() and add/sub bookkeeping accept it)
// With -DFP29_CHECK (host only) every element carries worst-case limb and value bounds that are
// propagated by each operation independently of the data and asserted against the preconditions,
// so one execution of a formula proves its overflow-freedom for all inputs of the same classes.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FD __host__ __device__ __forceinline__
#else
#define FD inline
#endif
#ifdef FP29_CHECK
#include <assert.h>
#include <math.h>
#endif

namespace bls29 {

constexpr int NL = 14;
constexpr int LB = 29;
constexpr uint32_t LMASK = (1u << LB) - 1;
constexpr uint32_t INV29 = 0x1ffcfffdu;   // -p^-1 mod 2^29
constexpr uint32_t PINV29 = 0x30003u;     //  p^-1 mod 2^29

#define BLS29_P     {0x1fffaaabu, 0xff7ffffu, 0x14ffffeeu, 0x17fffd62u, 0xf6241eau, 0x9507b58u, 0xafd9cc3u, 0x109e70a2u, 0x1764774bu, 0x121a5d66u, 0x12c6e9edu, 0x12ffcd34u, 0x111ea3u, 0xdu}
#define BLS29_ONE   {0x3a9fb84u, 0xba00690u, 0x71288f1u, 0xf59bcc5u, 0x126cb614u, 0x585bf36u, 0x1b85ac3du, 0x1cf856fau, 0x1891ecbdu, 0x1a7eec05u, 0x155a88f0u, 0x741ac6du, 0x1317c30fu, 0x9u}
#define BLS29_CIN   {0x1fddebbdu, 0x1a4f5474u, 0x291f399u, 0x14d03b3cu, 0xf6cad2cu, 0x1b4cabcau, 0x1592827cu, 0x21c6ac7u, 0x1ec52a84u, 0x16fd5ec4u, 0xc960da6u, 0xfd2af6bu, 0x13263591u, 0xbu}
#define BLS29_COUT  {0x2fffdu, 0x10480000u, 0x300009du, 0x8001788u, 0x158baebfu, 0xc2ba9e3u, 0x1d157d22u, 0xa6e0a4au, 0xd77ce58u, 0x1d12b763u, 0x1701c6a5u, 0x1501c926u, 0x1f65ec3fu, 0xau}
#define BLS29_K4    {0x9ffeaaacu, 0x9fdffffbu, 0x93ffffb5u, 0x9ffff586u, 0x9d8907a6u, 0x8541ed5du, 0x8bf67309u, 0x8279c285u, 0x9d91dd2au, 0x88697596u, 0x8b1ba7b2u, 0x8bff34ceu, 0x80447a8au, 0x30u}
#define BLS29_K8    {0x9ffd5558u, 0x9fbffffbu, 0x87ffff6fu, 0x9fffeb11u, 0x9b120f51u, 0x8a83dabfu, 0x97ece616u, 0x84f3850eu, 0x9b23ba58u, 0x90d2eb31u, 0x96374f68u, 0x97fe69a0u, 0x8088f518u, 0x64u}
#define BLS29_K16   {0x9ffaaab0u, 0x9f7ffffbu, 0x8ffffee3u, 0x9fffd626u, 0x96241ea7u, 0x9507b583u, 0x8fd9cc30u, 0x89e70a21u, 0x964774b4u, 0x81a5d667u, 0x8c6e9ed5u, 0x8ffcd345u, 0x8111ea35u, 0xccu}
#define BLS29_K64   {0x9feaaac0u, 0x9dfffffbu, 0x9ffffb9bu, 0x9fff58a5u, 0x98907aabu, 0x941ed61au, 0x9f6730ceu, 0x879c2891u, 0x991dd2ddu, 0x869759aau, 0x91ba7b60u, 0x9ff34d21u, 0x8447a8e1u, 0x33cu}
#define BLS29_K32   {0x9ff55560u, 0x9efffffbu, 0x9ffffdcbu, 0x9fffac50u, 0x8c483d53u, 0x8a0f6b0bu, 0x9fb39865u, 0x93ce1446u, 0x8c8ee96cu, 0x834bacd3u, 0x98dd3daeu, 0x9ff9a68eu, 0x8223d46eu, 0x19cu}

// constants are spelled as function-local constexpr arrays so that, after full unrolling, each
// limb becomes an immediate materialised into an SGPR (VOP3 on gfx9 cannot take a literal)
#define BLS29_DECL_P constexpr uint32_t P_[NL] = BLS29_P

struct Fp {
    uint32_t l[NL];
#ifdef FP29_CHECK
    uint64_t ub[NL];   // worst-case upper bound of each limb
    double vb;         // worst-case value bound, in units of p
#endif
};

#ifdef FP29_CHECK
static const double P_OVER_2_377 = 13.0021;   // p / 2^377 (rounded up)
inline void chk_set_N(Fp &r, double vb) {
    for (int i = 0; i < NL - 1; i++) r.ub[i] = (1ull << LB) + 7;
    r.ub[NL - 1] = (uint64_t)floor(vb * P_OVER_2_377) + 1;
    r.vb = vb;
}
inline void chk_actual(const Fp &a) { for (int i = 0; i < NL; i++) assert(a.l[i] <= a.ub[i]); }
#define CHK(x) x
#else
#define CHK(x)
#endif

FD void fp_zero(Fp &r) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = 0;
    CHK(for (int i = 0; i < NL; i++) r.ub[i] = 0; r.vb = 0;)
}
FD void fp_set_one(Fp &r) {
    constexpr uint32_t O_[NL] = BLS29_ONE;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = O_[i];
    CHK(chk_set_N(r, 1.0);)
}

// r = a + b (limb-wise, no carry).  Result class: sum of the operand classes.
FD void fp_add(Fp &r, const Fp &a, const Fp &b) {
    CHK(for (int i = 0; i < NL; i++) { uint64_t s = a.ub[i] + b.ub[i]; assert(s < (1ull << 32)); r.ub[i] = s; } r.vb = a.vb + b.vb;)
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
}
FD void fp_dbl(Fp &r, const Fp &a) { fp_add(r, a, a); }

// Redundant form of M*p whose limbs dominate any normalised-ish subtrahend: limbs i < 13 are 2^31 + d_i (d_i < 2^29),
// the top limb is whatever remains (~ 13 M - 4).  Generated at compile time for any M.
template <int M> struct KTab {
    uint32_t l[NL];
    constexpr KTab() : l{} {
        constexpr uint32_t P_[NL] = BLS29_P;
        long long v[NL + 1] = {};
        long long carry = 0;
        for (int i = 0; i < NL; i++) { long long t = (long long)P_[i] * M + carry; v[i] = t & LMASK; carry = t >> LB; }
        v[NL - 1] += carry << LB;                      // top limb keeps everything above bit 377
        for (int i = 0; i < NL - 1; i++) {
            // move 2^31 = 4 * 2^29 from limb i+1 down into limb i
            v[i] += (1ll << 31);
            v[i + 1] -= 4;
            for (int j = i + 1; j < NL - 1 && v[j] < 0; j++) { v[j] += (1ll << LB); v[j + 1] -= 1; }
        }
        for (int i = 0; i < NL; i++) l[i] = (uint32_t)v[i];
    }
};
// r = a - b + M*p.  Preconditions (asserted under FP29_CHECK): every limb of b is dominated by the matching limb of K_M
// and a.l[i] + K_M[i] < 2^32.
template <int M> FD void fp_sub(Fp &r, const Fp &a, const Fp &b) {
    constexpr KTab<M> K{};
    static_assert(M >= 2, "multiple too small");
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t k = K.l[i];
        CHK(assert(b.ub[i] <= k); assert(a.ub[i] + k < (1ull << 32));)
        CHK(uint64_t nub = a.ub[i] + k;)
        r.l[i] = a.l[i] + (k - b.l[i]);
        CHK(r.ub[i] = nub;)
    }
    CHK(r.vb = a.vb + M;)
}
static_assert(KTab<4>().l[0] == 0x9ffeaaacu && KTab<4>().l[13] == 0x30u && KTab<64>().l[5] == 0x941ed61au && KTab<64>().l[13] == 0x33cu, "KTab generator");

// one parallel carry pass: class L -> class N (limbs <= 2^29 + 7); value unchanged
FD void fp_norm(Fp &r, const Fp &a) {
    uint32_t c[NL];
#pragma unroll
    for (int i = 0; i < NL - 1; i++) c[i] = a.l[i] >> LB;
    CHK(uint64_t cub[NL]; for (int i = 0; i < NL - 1; i++) cub[i] = a.ub[i] >> LB; uint64_t top = a.ub[NL - 1] + cub[NL - 2]; double vb = a.vb;)
    uint32_t t0 = a.l[0] & LMASK;
    uint32_t tl = a.l[NL - 1] + c[NL - 2];
#pragma unroll
    for (int i = NL - 2; i >= 1; i--) r.l[i] = (a.l[i] & LMASK) + c[i - 1];
    r.l[0] = t0;
    r.l[NL - 1] = tl;
#ifdef FP29_CHECK
    for (int i = 1; i < NL - 1; i++) { assert(cub[i - 1] <= 7); r.ub[i] = LMASK + cub[i - 1]; }
    r.ub[0] = LMASK;
    uint64_t topv = (uint64_t)floor(vb * P_OVER_2_377) + 1;   // all lower limbs are >= 0, so top <= value / 2^377
    r.ub[NL - 1] = top < topv ? top : topv;
    assert(r.ub[NL - 1] < (1ull << 32));
    r.vb = vb;
#endif
}

// r = 12 a with ONE carry pass (the Miller loop's e = 3 b' c = 12 (1 + u) c: x 3, carry pass, x 4, carry pass cost three times as much).
// 3 a_i < 2^32 for any limb below 2^30.4; the carry of limb i is (3 a_i) >> 27, what stays is ((3 a_i) mod 2^27) << 2.
// Input class N; output limbs <= 2^29 - 4 + 12 (one more than class N: every use is followed by a lazy subtraction / addition and a carry pass).
FD void fp_mul12_norm(Fp &r, const Fp &a) {
    uint32_t x3[NL], c[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) { CHK(assert(3 * a.ub[i] < (1ull << 32));) x3[i] = a.l[i] + (a.l[i] << 1); }
#pragma unroll
    for (int i = 0; i < NL - 1; i++) c[i] = x3[i] >> 27;
    CHK(uint64_t ub[NL]; ub[0] = LMASK - 3; for (int i = 1; i < NL - 1; i++) ub[i] = (LMASK - 3) + ((3 * a.ub[i - 1]) >> 27);
        ub[NL - 1] = 12 * a.ub[NL - 1] + ((3 * a.ub[NL - 2]) >> 27); assert(ub[NL - 1] < (1ull << 32)); double vb = 12.0 * a.vb;)
    r.l[0] = (x3[0] & 0x7ffffffu) << 2;
#pragma unroll
    for (int i = 1; i < NL - 1; i++) r.l[i] = ((x3[i] & 0x7ffffffu) << 2) + c[i - 1];
    r.l[NL - 1] = (x3[NL - 1] << 2) + c[NL - 2];
    CHK(for (int i = 0; i < NL; i++) r.ub[i] = ub[i]; r.vb = vb; chk_actual(r);)
}

#ifdef FP29_CHECK
inline void chk_mul_pre(const Fp &a, const Fp &b) {
    BLS29_DECL_P;
    // worst-case column sums (operand products + m*p products + incoming carry) must fit in 64 bits
    unsigned __int128 carry = 0;
    for (int k = 0; k < 2 * NL - 1; k++) {
        unsigned __int128 s = carry;
        for (int i = 0; i < NL; i++) { int j = k - i; if (j < 0 || j >= NL) continue; s += (unsigned __int128)a.ub[i] * b.ub[j]; s += (unsigned __int128)LMASK * P_[j]; }
        assert(s < ((unsigned __int128)1 << 64));
        carry = s >> LB;
    }
    assert(a.vb * b.vb < 33554432.0 * 0.5);   // a*b < 2^24 p^2  =>  result < p (1 + va vb / 2^25)
}
#endif

// Montgomery product, r = a*b / 2^406 mod p, result class N with value < 2p.
// Precondition (asserted under FP29_CHECK): worst-case column sums < 2^64, e.g. a in N (or a sum of two
// N) and b in N.  Product-scanning with the reduction interleaved: 2 * 14^2 = 392 v_mad_u64_u32.
FD void fp_mul(Fp &r, const Fp &a, const Fp &b) {
    BLS29_DECL_P;
    CHK(chk_mul_pre(a, b); chk_actual(a); chk_actual(b);)
    uint32_t m[NL], t[NL];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P_[k - i];
        m[k] = ((uint32_t)acc * INV29) & LMASK;
        acc += (uint64_t)m[k] * P_[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)m[i] * P_[k - i];
        t[k - NL] = (uint32_t)acc & LMASK;
        acc >>= LB;
    }
    t[NL - 1] = (uint32_t)acc;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = t[i];
    CHK(chk_set_N(r, 1.0 + a.vb * b.vb / 33554432.0); chk_actual(r);)
}

// Fused  r = (a*b + c*d) / 2^406 mod p : two operand products share ONE Montgomery reduction (588 instead of 784
// v_mad_u64_u32).  All four operands class N: a column then holds 28 operand products + 14 reduction products < 42 * 2^58 < 2^64.
FD void fp_mul2(Fp &r, const Fp &a, const Fp &b, const Fp &c, const Fp &d) {
    BLS29_DECL_P;
#ifdef FP29_CHECK
    {
        unsigned __int128 carry = 0;
        for (int k = 0; k < 2 * NL - 1; k++) {
            unsigned __int128 s = carry;
            for (int i = 0; i < NL; i++) { int j = k - i; if (j < 0 || j >= NL) continue; s += (unsigned __int128)a.ub[i] * b.ub[j] + (unsigned __int128)c.ub[i] * d.ub[j]; s += (unsigned __int128)LMASK * P_[j]; }
            assert(s < ((unsigned __int128)1 << 64));
            carry = s >> LB;
        }
        assert(a.vb * b.vb + c.vb * d.vb < 33554432.0 * 0.5);
        chk_actual(a); chk_actual(b); chk_actual(c); chk_actual(d);
    }
#endif
    uint32_t m[NL], t[NL];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)a.l[i] * b.l[k - i]; acc += (uint64_t)c.l[i] * d.l[k - i]; }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P_[k - i];
        m[k] = ((uint32_t)acc * INV29) & LMASK;
        acc += (uint64_t)m[k] * P_[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) { acc += (uint64_t)a.l[i] * b.l[k - i]; acc += (uint64_t)c.l[i] * d.l[k - i]; }
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)m[i] * P_[k - i];
        t[k - NL] = (uint32_t)acc & LMASK;
        acc >>= LB;
    }
    t[NL - 1] = (uint32_t)acc;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = t[i];
    CHK(chk_set_N(r, 1.0 + (a.vb * b.vb + c.vb * d.vb) / 33554432.0); chk_actual(r);)
}

// r = a^2 / 2^406 mod p.  105 + 196 v_mad_u64_u32.  Precondition: a in N.
FD void fp_sqr(Fp &r, const Fp &a) {
    BLS29_DECL_P;
    CHK(chk_mul_pre(a, a); chk_actual(a);)
    uint32_t m[NL], t[NL], a2[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = a.l[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a.l[i] * a2[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P_[k - i];
        m[k] = ((uint32_t)acc * INV29) & LMASK;
        acc += (uint64_t)m[k] * P_[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; 2 * i < k; i++) acc += (uint64_t)a.l[i] * a2[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)m[i] * P_[k - i];
        t[k - NL] = (uint32_t)acc & LMASK;
        acc >>= LB;
    }
    t[NL - 1] = (uint32_t)acc;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = t[i];
    CHK(chk_set_N(r, 1.0 + a.vb * a.vb / 33554432.0); chk_actual(r);)
}

// ---- exact (slow-path) helpers: canonical representative in [0, p) with fully propagated limbs ----
FD void fp_canon(Fp &r, const Fp &a) {
    BLS29_DECL_P;
    uint32_t t[NL];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) { c += a.l[i]; t[i] = (uint32_t)c & LMASK; c >>= LB; }
    c += a.l[NL - 1];
    t[NL - 1] = (uint32_t)c;   // precondition: value < 2^12 p  =>  top limb < 2^16
    // subtract 2^j * p for j = 11..0 whenever the value stays non-negative
    for (int j = 11; j >= 0; j--) {
        uint32_t q[NL];
        uint64_t cc = 0;
#pragma unroll
        for (int i = 0; i < NL - 1; i++) { cc += ((uint64_t)P_[i] << j); q[i] = (uint32_t)cc & LMASK; cc >>= LB; }
        cc += ((uint64_t)P_[NL - 1] << j);
        q[NL - 1] = (uint32_t)cc;
        bool ge = true, decided = false;
#pragma unroll
        for (int i = NL - 1; i >= 0; i--) { if (!decided && t[i] != q[i]) { ge = t[i] > q[i]; decided = true; } }
        if (ge) {
            int64_t b2 = 0;
#pragma unroll
            for (int i = 0; i < NL - 1; i++) { int64_t v = (int64_t)t[i] - (int64_t)q[i] + b2; t[i] = (uint32_t)v & LMASK; b2 = v >> LB; }
            t[NL - 1] = (uint32_t)((int64_t)t[NL - 1] - (int64_t)q[NL - 1] + b2);
        }
    }
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = t[i];
    CHK(assert(a.vb < 4096.0); for (int i = 0; i < NL - 1; i++) r.ub[i] = LMASK; r.ub[NL - 1] = 13; r.vb = 1.0;)
}
FD bool fp_is_zero_exact(const Fp &a) {
    Fp c; fp_canon(c, a);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= c.l[i];
    return o == 0;
}
// cheap necessary condition for a == 0 mod p when 0 <= value < 64 p: a = k p  =>  k = l0 * p^-1 mod 2^29 < 64.
// False positives have probability 2^-23 on random data and only cost the exact check.
FD bool fp_maybe_zero(const Fp &a) { return ((a.l[0] * PINV29) & LMASK) < 64u; }

// ---- conversion to / from the C-ABI form: 6 x u64 little-endian, value * 2^384 mod p (ark-ff layout) ----
FD void fp_from_abi(Fp &r, const uint32_t w[12]) {
    constexpr uint32_t CIN_[NL] = BLS29_CIN;
    Fp t, cin;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = i * LB, wi = bit >> 5, sh = bit & 31;
        uint64_t v = (uint64_t)w[wi] >> sh;
        if (wi + 1 < 12) v |= ((uint64_t)w[wi + 1] << (32 - sh));
        t.l[i] = (uint32_t)v & LMASK;
    }
    t.l[NL - 1] = w[11] >> ((13 * LB) & 31);   // bits 377..383
#pragma unroll
    for (int i = 0; i < NL; i++) cin.l[i] = CIN_[i];
    CHK(chk_set_N(t, 10.0); t.ub[NL - 1] = 127; chk_set_N(cin, 1.0);)
    fp_mul(r, t, cin);   // x 2^384 * 2^428 / 2^406 = x 2^406
}
FD void fp_to_abi(uint32_t w[12], const Fp &a) {
    constexpr uint32_t COUT_[NL] = BLS29_COUT;
    Fp t, c, cout;
#pragma unroll
    for (int i = 0; i < NL; i++) cout.l[i] = COUT_[i];
    CHK(chk_set_N(cout, 1.0);)
    fp_norm(t, a);
    fp_mul(t, t, cout);   // x 2^406 * 2^384 / 2^406 = x 2^384
    fp_canon(c, t);
    uint32_t o[12];
#pragma unroll
    for (int i = 0; i < 12; i++) o[i] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = i * LB, wi = bit >> 5, sh = bit & 31;
        o[wi] |= c.l[i] << sh;
        if (sh + LB > 32 && wi + 1 < 12) o[wi + 1] |= c.l[i] >> (32 - sh);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = o[i];
}

// ---- uniform spellings used by the field-generic group law (ec29.hip.h) ----
FD void fzero(Fp &r) { fp_zero(r); }
FD void fset_one(Fp &r) { fp_set_one(r); }
FD void fadd(Fp &r, const Fp &a, const Fp &b) { fp_add(r, a, b); }
FD void fdbl(Fp &r, const Fp &a) { fp_add(r, a, a); }
template <int M> FD void fsub(Fp &r, const Fp &a, const Fp &b) { fp_sub<M>(r, a, b); }
FD void fnorm(Fp &r, const Fp &a) { fp_norm(r, a); }
FD void fmul(Fp &r, const Fp &a, const Fp &b) { fp_mul(r, a, b); }
FD void fsqr(Fp &r, const Fp &a) { fp_sqr(r, a); }
// r = a*b - c*d (class N); c has value < (M-1) p.  Base field: negate c lazily and fuse both products into one reduction.
template <int M> FD void fmul_sub(Fp &r, const Fp &a, const Fp &b, const Fp &c, const Fp &d) {
    Fp z, cn; fp_zero(z); fp_sub<M>(cn, z, c); fp_norm(cn, cn);
    fp_mul2(r, a, b, cn, d);
}
FD bool fmaybe_zero(const Fp &a) { return fp_maybe_zero(a); }
FD bool fis_zero_exact(const Fp &a) { return fp_is_zero_exact(a); }

}  // namespace bls29
