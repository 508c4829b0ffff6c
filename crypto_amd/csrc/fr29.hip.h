// crypto_amd/csrc/fr29.hip.h — BLS12-381 scalar field Fr for gfx950, same carry-free scheme as fp29.hip.h.
//
// Device counterpart of ark_bls12_381::Fr (ark-ff Fp<MontBackend<FrConfig,4>,4>) for the R1CS -> QAP witness map
// (/root/reference/legogroth16/src/r1cs_to_qap.rs:150-210: sparse A z, B z, C z; 3 iFFT + 3 coset FFT; (ab - c)/Z; coset iFFT).
//
// r has 255 bits; an element is 10 limbs of 29 bits (290 bits, Montgomery radix 2^290).  Nine limbs (261 bits) would
// leave only 6 bits of headroom above r, and an NTT butterfly chain adds a multiple of r per stage (lazy subtraction),
// so the tenth limb buys 2^35 of slack and removes every conditional subtraction from the transform.
//   add: limb-wise; sub: a + K512 - b with K512 = 512 r in limb-dominating redundant form (the smallest power-of-two multiple whose
//   redundant form has a non-negative top limb); norm: one parallel carry pass; mul: 2 * 10^2 v_mad_u64_u32.
// Contracts: fr_mul takes a with limbs < 2^31 and b with limbs <= 2^29 + 7 (twiddles / constants), value(a) * value(b) < 2^34 r^2,
// and returns limbs < 2^29 (top limb small), value < 2 r.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FRD __host__ __device__ __forceinline__
#else
#define FRD inline
#endif

namespace fr29 {

constexpr int NL = 10;
constexpr int LB = 29;
constexpr uint32_t LMASK = (1u << LB) - 1;
constexpr uint32_t INV29 = 0x1fffffffu;   // -r^-1 mod 2^29  (r = 1 mod 2^29)

#define FR29_R      {0x1u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0xc0404d0u, 0x1520cce7u, 0xa6533afu, 0x73eda7u, 0x0u}
#define FR29_ONE    {0xabdac49u, 0xa129d71u, 0x6a3eff5u, 0x168d894du, 0x15997df8u, 0x9407325u, 0xb7bc5dcu, 0x1ec6f83eu, 0x71e0beu, 0x0u}
#define FR29_CCANON {0x1a87a703u, 0xb8e0c5u, 0x9c2c11u, 0x471edf0u, 0x1f1ae7fbu, 0x1abfb833u, 0xc2840a2u, 0x6f2e13eu, 0x533266u, 0x0u}   /* 2^580: canonical -> internal */
#define FR29_CM_IN  {0x121c883du, 0x6d14718u, 0x1a2c9fe6u, 0xcbc06e6u, 0x570e741u, 0x1a1b3b01u, 0xb4a4f04u, 0xa0ab139u, 0x3f1c83u, 0x0u}   /* 2^324: x 2^256 -> internal */
#define FR29_CM_OUT {0x1ffffffeu, 0xfu, 0xd20080u, 0x96ff400u, 0x4ff5588u, 0x7f7f65eu, 0x15be6631u, 0xb3598a0u, 0x1824b1u, 0x0u}        /* 2^256: internal -> x 2^256 */
#define FR29_K512   {0x80000200u, 0x9fffeffcu, 0x8dff7ffbu, 0x900bfff5u, 0x80aa77b0u, 0x8809a1d4u, 0x8199cebcu, 0x8a675f4eu, 0x87db4ea2u, 0x3u}

struct Fr { uint32_t l[NL]; };

FRD void fr_zero(Fr &r) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = 0;
}
FRD void fr_const(Fr &r, const uint32_t (&c)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = c[i];
}
FRD void fr_one(Fr &r) { constexpr uint32_t O_[NL] = FR29_ONE; fr_const(r, O_); }
FRD void fr_add(Fr &r, const Fr &a, const Fr &b) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
}
// Redundant form of M r whose limbs dominate a normalised subtrahend: limbs i < 9 are 2^31 + d_i, the top limb is what remains
// (~ (M - 283) / 70).  Generated at compile time.
template <long long M, int SH = 31> struct FrKTab {
    uint32_t l[NL];
    constexpr FrKTab() : l{} {
        constexpr uint32_t P_[NL] = FR29_R;
        long long v[NL + 1] = {};
        unsigned __int128 carry = 0;
        for (int i = 0; i < NL; i++) { unsigned __int128 t = (unsigned __int128)P_[i] * (unsigned long long)M + carry; v[i] = (long long)(t & LMASK); carry = t >> LB; }
        v[NL - 1] += (long long)(carry << LB);
        for (int i = 0; i < NL - 1; i++) {
            v[i] += (1ll << SH);                       // limbs i < 9 become 2^SH + d_i ...
            v[i + 1] -= (1ll << (SH - LB));            // ... borrowed from the next limb
            for (int j = i + 1; j < NL - 1 && v[j] < 0; j++) { v[j] += (1ll << LB); v[j + 1] -= 1; }
        }
        for (int i = 0; i < NL; i++) l[i] = (uint32_t)v[i];
    }
};
static_assert(FrKTab<512>().l[0] == 0x80000200u && FrKTab<512>().l[9] == 0x3u, "FrKTab generator");
// r = a - b + M r, limb by limb in 32-bit arithmetic.  Exact when every limb of a + K_M - b stays in [0, 2^32): limbs i < 9 of b
// normalised (<= 2^29 + 7 < K's 2^31 + d_i), limbs of a < 2^30, and for the top limb value(b) <= ~M r (K_M's top limb is ~M / 70.4).
//   M = 512     : b is a product (value < 4 r)
//   M = 2^34    : b is any value an NTT can accumulate.  Decimation-in-frequency sums double per stage and the two operands of a
//                 butterfly need NOT be balanced (a circuit whose even rows are empty and whose odd rows are dense subtracts a partial
//                 sum of D/2 rows from nothing): with inputs < 16 r (k_csr_eval reduces longer rows) and D <= 2^28, value(b) < 2^32 r.
//                 The product that follows sees an operand < 1.5 * 2^34 r and returns < 2.4 r (Montgomery radix 2^290 = 2^35.1 r).
template <long long M = 512, int SH = 31> FRD void fr_sub(Fr &r, const Fr &a, const Fr &b) {
    constexpr FrKTab<M, SH> K{};
    static_assert(K.l[NL - 1] < (1u << 30), "multiple too large");
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + (K.l[i] - b.l[i]);
}
// the same with limbs 2^30 + d_i: the result of a (normalised) - b (a product) stays below 2^31 per limb, so it can enter a product or one more
// addition / subtraction without a carry pass (the radix-4 butterflies of the NTT)
static_assert(FrKTab<512, 30>().l[0] == 0x40000200u && FrKTab<512, 30>().l[9] == 0x5u, "FrKTab generator (2^30 form)");
constexpr long long FR_BIG = 1ll << 34;
FRD void fr_norm(Fr &r, const Fr &a) {
    uint32_t c[NL];
#pragma unroll
    for (int i = 0; i < NL - 1; i++) c[i] = a.l[i] >> LB;
    uint32_t t0 = a.l[0] & LMASK, tl = a.l[NL - 1] + c[NL - 2];
#pragma unroll
    for (int i = NL - 2; i >= 1; i--) r.l[i] = (a.l[i] & LMASK) + c[i - 1];
    r.l[0] = t0; r.l[NL - 1] = tl;
}
// Montgomery product a * b / 2^290 mod r (product scanning, reduction interleaved)
FRD void fr_mul(Fr &r, const Fr &a, const Fr &b) {
    constexpr uint32_t P_[NL] = FR29_R;
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
}
// canonical representative in [0, r), limbs fully propagated.  Precondition: value < 2^STEPS r (a product is < 2 r: STEPS = 2 covers it twice).
template <int STEPS = 24> FRD void fr_canon(Fr &r, const Fr &a) {
    constexpr uint32_t P_[NL] = FR29_R;
    uint32_t t[NL];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) { c += a.l[i]; t[i] = (uint32_t)c & LMASK; c >>= LB; }
    c += a.l[NL - 1];
    t[NL - 1] = (uint32_t)c;
    for (int j = STEPS - 1; j >= 0; j--) {
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
}
// 8 x u32 (256-bit little-endian) <-> internal.  mont = true: the words hold x * 2^256 mod r (ark-ff Fr), else canonical x.
FRD void fr_from_words(Fr &r, const uint32_t w[8], bool mont) {
    constexpr uint32_t CC_[NL] = FR29_CCANON; constexpr uint32_t CM_[NL] = FR29_CM_IN;
    Fr t, k;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = i * LB, wi = bit >> 5, sh = bit & 31;
        uint64_t v = 0;
        if (wi < 8) { v = (uint64_t)w[wi] >> sh; if (wi + 1 < 8) v |= ((uint64_t)w[wi + 1] << (32 - sh)); }
        t.l[i] = (uint32_t)v & LMASK;
    }
#pragma unroll
    for (int i = 0; i < NL; i++) k.l[i] = mont ? CM_[i] : CC_[i];
    fr_mul(r, t, k);
}
FRD void fr_to_words(uint32_t w[8], const Fr &a, bool mont) {
    constexpr uint32_t CM_[NL] = FR29_CM_OUT;
    Fr t, k, c;
    fr_zero(k); k.l[0] = 1;
    if (mont) fr_const(k, CM_);
    fr_norm(t, a);
    fr_mul(t, t, k);        // < r (1 + value(a) / 2^290) < 2 r
    fr_canon<2>(c, t);
    uint32_t o[9];
#pragma unroll
    for (int i = 0; i < 9; i++) o[i] = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {   // limb 9 of a canonical value is zero
        const int bit = i * LB, wi = bit >> 5, sh = bit & 31;
        o[wi] |= c.l[i] << sh;
        if (sh + LB > 32) o[wi + 1] |= c.l[i] >> (32 - sh);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = o[i];
}

}  // namespace fr29
