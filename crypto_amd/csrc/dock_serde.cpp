// crypto_amd/csrc/dock_serde.cpp — canonical point (de)serialisation, host code.
//
// ark-bls12-381 0.4 serialises group elements in the Zcash / IETF BLS12-381 format (ark_bls12_381::curves::util):
// big-endian coordinates; the top three bits of byte 0 are (compressed, infinity, y-is-lexicographically-largest);
// G1: 48 B compressed / 96 B uncompressed, G2: 96 / 192 B with c1 BEFORE c0.  This is what `CanonicalSerialize` emits for
// the reference's keys and proofs (legogroth16/src/data_structures.rs:7-186, utils/src/serde_utils.rs:8-33), so a driver
// can load a proving key written by the Rust side and hand the limbs to dgpu_bases_upload_*.
// Decompression needs a square root: p = 3 (mod 4) so sqrt(a) = a^((p+1)/4) in Fq, and the complex method in Fq2.
// Validation follows `deserialize_compressed` / `deserialize_uncompressed` (Validate::Yes): the point must be on the curve AND in the
// prime-order subgroup ([r]P = O, checked here by a plain double-and-add on the host — same verdict as arkworks' endomorphism test);
// mode bit 1 (DGPU_SERDE_NO_VALIDATE) is Validate::No and skips the subgroup test only.  An infinity encoding must be canonical: no
// other flag and no payload bit set (the Zcash / IETF rule; arkworks 0.4.0 ignores the payload there — stricter on malformed input).
#include <string.h>
#include "../../include/dock_gpu.h"
#include "host_field.hpp"

namespace {
using hostf::Fq;
using hostf::Fq2;

const uint64_t P_LIMBS[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
const uint64_t R2_LIMBS[6] = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};

Fq from_canonical(const uint64_t c[6]) { Fq a, r2; memcpy(a.l, c, 48); memcpy(r2.l, R2_LIMBS, 48); return a * r2; }
void to_canonical(uint64_t c[6], const Fq &a) { Fq one; memset(&one, 0, sizeof one); one.l[0] = 1; Fq t = a * one; memcpy(c, t.l, 48); }

bool lt_p(const uint64_t c[6]) { for (int i = 5; i >= 0; i--) { if (c[i] < P_LIMBS[i]) return true; if (c[i] > P_LIMBS[i]) return false; } return false; }
// canonical value > (p - 1) / 2 ?
bool is_high(const Fq &a) {
    static const uint64_t HALF[6] = {0xdcff7fffffffd555ULL, 0x0f55ffff58a9ffffULL, 0xb39869507b587b12ULL, 0xb23ba5c279c2895fULL, 0x258dd3db21a5d66bULL, 0x0d0088f51cbff34dULL};   // (p-1)/2
    uint64_t c[6]; to_canonical(c, a);
    for (int i = 5; i >= 0; i--) { if (c[i] > HALF[i]) return true; if (c[i] < HALF[i]) return false; }
    return false;
}
// arkworks' lexicographic order on Fq2: compare c1 first, then c0
bool is_high2(const Fq2 &a) { if (!a.c1.is_zero()) return is_high(a.c1); return is_high(a.c0); }

void be48_to_limbs(uint64_t c[6], const uint8_t *b) { for (int i = 0; i < 6; i++) { uint64_t v = 0; for (int k = 0; k < 8; k++) v = (v << 8) | b[(5 - i) * 8 + k]; c[i] = v; } }
void limbs_to_be48(uint8_t *b, const uint64_t c[6]) { for (int i = 0; i < 6; i++) for (int k = 0; k < 8; k++) b[(5 - i) * 8 + k] = (uint8_t)(c[i] >> (8 * (7 - k))); }

Fq fq_pow(const Fq &a, const uint64_t *e, int nl) { Fq acc = Fq::one(), base = a; for (int i = 0; i < nl * 64; i++) { if ((e[i / 64] >> (i % 64)) & 1) acc = acc * base; base = base.sqr(); } return acc; }
bool fq_sqrt(Fq &r, const Fq &a) {
    static const uint64_t E[6] = {0xee7fbfffffffeaabULL, 0x07aaffffac54ffffULL, 0xd9cc34a83dac3d89ULL, 0xd91dd2e13ce144afULL, 0x92c6e9ed90d2eb35ULL, 0x0680447a8e5ff9a6ULL};   // (p+1)/4
    Fq s = fq_pow(a, E, 6);
    if (!(s.sqr() == a)) return false;
    r = s; return true;
}
Fq fq_four() { Fq one = Fq::one(); Fq two = one + one; return two + two; }
bool fq2_sqrt(Fq2 &r, const Fq2 &a) {
    // complex method: |a| = sqrt(a0^2 + a1^2); x0 = sqrt((a0 + |a|)/2) (or with -|a|), x1 = a1 / (2 x0)
    if (a.c1.is_zero()) {
        Fq s;
        if (fq_sqrt(s, a.c0)) { r = {s, Fq::zero()}; return true; }
        if (fq_sqrt(s, a.c0.neg())) { r = {Fq::zero(), s}; return true; }     // sqrt(-1) = u
        return false;
    }
    Fq norm = a.c0.sqr() + a.c1.sqr(), alpha;
    if (!fq_sqrt(alpha, norm)) return false;
    Fq two_inv = (Fq::one() + Fq::one()).inv();
    Fq delta = (a.c0 + alpha) * two_inv, x0;
    if (!fq_sqrt(x0, delta)) { delta = (a.c0 - alpha) * two_inv; if (!fq_sqrt(x0, delta)) return false; }
    Fq x1 = a.c1 * (x0 + x0).inv();
    Fq2 cand = {x0, x1};
    if (!(cand.sqr() == a)) return false;
    r = cand; return true;
}

enum { FLAG_COMPRESSED = 0x80, FLAG_INF = 0x40, FLAG_LARGEST = 0x20 };

// [r]P == O ?   r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
template <class F> bool in_prime_subgroup(const F &x, const F &y) {
    static const uint64_t RM[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    hostf::HXyzz<F> P; P.x = x; P.y = y; P.zz = F::one(); P.zzz = F::one(); P.inf = false;
    hostf::HXyzz<F> acc = hostf::HXyzz<F>::identity();
    for (int i = 254; i >= 0; i--) { acc.dbl_in_place(); if ((RM[i / 64] >> (i % 64)) & 1) acc.add_in_place(P); }
    return acc.inf;
}
bool canonical_infinity(const uint8_t *b, size_t sz, uint8_t flags) {
    if (flags & FLAG_LARGEST) return false;
    if (b[0] & 0x1f) return false;
    for (size_t k = 1; k < sz; k++) if (b[k]) return false;
    return true;
}
}  // namespace

extern "C" {

int32_t dgpu_g1_serialize(const uint64_t *xy, const uint8_t *is_inf, size_t n, int32_t compressed, uint8_t *out) {
    if (n && (!xy || !out)) return DGPU_E_BADARG;
    const size_t sz = compressed ? 48 : 96;
    for (size_t i = 0; i < n; i++) {
        uint8_t *o = out + i * sz; memset(o, 0, sz);
        Fq x, y; memcpy(x.l, xy + 12 * i, 48); memcpy(y.l, xy + 12 * i + 6, 48);
        bool inf = (is_inf && is_inf[i]) || (x.is_zero() && y.is_zero());
        if (inf) { o[0] = (compressed ? FLAG_COMPRESSED : 0) | FLAG_INF; continue; }
        uint64_t c[6]; to_canonical(c, x); limbs_to_be48(o, c);
        if (compressed) { o[0] |= FLAG_COMPRESSED; if (is_high(y)) o[0] |= FLAG_LARGEST; }
        else { to_canonical(c, y); limbs_to_be48(o + 48, c); }
    }
    return DGPU_OK;
}
int32_t dgpu_g1_deserialize(const uint8_t *in, size_t n, int32_t mode, uint64_t *xy, uint8_t *is_inf) {
    if (n && (!in || !xy || !is_inf)) return DGPU_E_BADARG;
    const int compressed = mode & 1; const bool validate = !(mode & DGPU_SERDE_NO_VALIDATE);
    const size_t sz = compressed ? 48 : 96;
    for (size_t i = 0; i < n; i++) {
        const uint8_t *b = in + i * sz; uint8_t flags = b[0] & 0xe0;
        if (((flags & FLAG_COMPRESSED) != 0) != (compressed != 0)) return DGPU_E_BADARG;
        uint8_t tmp[48]; memcpy(tmp, b, 48); tmp[0] &= 0x1f;
        memset(xy + 12 * i, 0, 96); is_inf[i] = 0;
        if (flags & FLAG_INF) { if (!canonical_infinity(b, sz, flags)) return DGPU_E_BADARG; is_inf[i] = 1; continue; }
        if (!compressed && (flags & FLAG_LARGEST)) return DGPU_E_BADARG;
        uint64_t c[6]; be48_to_limbs(c, tmp); if (!lt_p(c)) return DGPU_E_BADARG;
        Fq x = from_canonical(c), y;
        if (compressed) {
            if (!fq_sqrt(y, x.sqr() * x + fq_four())) return DGPU_E_BADARG;     // not on the curve
            if (is_high(y) != ((flags & FLAG_LARGEST) != 0)) y = y.neg();
        } else {
            be48_to_limbs(c, b + 48); if (!lt_p(c)) return DGPU_E_BADARG;
            y = from_canonical(c);
            if (!(y.sqr() == x.sqr() * x + fq_four())) return DGPU_E_BADARG;
        }
        if (validate && !in_prime_subgroup<Fq>(x, y)) return DGPU_E_BADARG;                // Validate::Yes: on the curve but outside G1
        memcpy(xy + 12 * i, x.l, 48); memcpy(xy + 12 * i + 6, y.l, 48);
    }
    return DGPU_OK;
}
int32_t dgpu_g2_serialize(const uint64_t *xy, const uint8_t *is_inf, size_t n, int32_t compressed, uint8_t *out) {
    if (n && (!xy || !out)) return DGPU_E_BADARG;
    const size_t sz = compressed ? 96 : 192;
    for (size_t i = 0; i < n; i++) {
        uint8_t *o = out + i * sz; memset(o, 0, sz);
        Fq2 x, y; memcpy(&x, xy + 24 * i, 96); memcpy(&y, xy + 24 * i + 12, 96);
        bool inf = (is_inf && is_inf[i]) || (x.is_zero() && y.is_zero());
        if (inf) { o[0] = (compressed ? FLAG_COMPRESSED : 0) | FLAG_INF; continue; }
        uint64_t c[6];
        to_canonical(c, x.c1); limbs_to_be48(o, c); to_canonical(c, x.c0); limbs_to_be48(o + 48, c);       // c1 first
        if (compressed) { o[0] |= FLAG_COMPRESSED; if (is_high2(y)) o[0] |= FLAG_LARGEST; }
        else { to_canonical(c, y.c1); limbs_to_be48(o + 96, c); to_canonical(c, y.c0); limbs_to_be48(o + 144, c); }
    }
    return DGPU_OK;
}
int32_t dgpu_g2_deserialize(const uint8_t *in, size_t n, int32_t mode, uint64_t *xy, uint8_t *is_inf) {
    if (n && (!in || !xy || !is_inf)) return DGPU_E_BADARG;
    const int compressed = mode & 1; const bool validate = !(mode & DGPU_SERDE_NO_VALIDATE);
    const size_t sz = compressed ? 96 : 192;
    Fq four = fq_four(); Fq2 b2 = {four, four};       // 4 (1 + u)
    for (size_t i = 0; i < n; i++) {
        const uint8_t *b = in + i * sz; uint8_t flags = b[0] & 0xe0;
        if (((flags & FLAG_COMPRESSED) != 0) != (compressed != 0)) return DGPU_E_BADARG;
        uint8_t tmp[48]; memcpy(tmp, b, 48); tmp[0] &= 0x1f;
        memset(xy + 24 * i, 0, 192); is_inf[i] = 0;
        if (flags & FLAG_INF) { if (!canonical_infinity(b, sz, flags)) return DGPU_E_BADARG; is_inf[i] = 1; continue; }
        if (!compressed && (flags & FLAG_LARGEST)) return DGPU_E_BADARG;
        uint64_t c[6]; Fq2 x, y;
        be48_to_limbs(c, tmp); if (!lt_p(c)) return DGPU_E_BADARG; x.c1 = from_canonical(c);
        be48_to_limbs(c, b + 48); if (!lt_p(c)) return DGPU_E_BADARG; x.c0 = from_canonical(c);
        Fq2 rhs = x.sqr() * x + b2;
        if (compressed) {
            if (!fq2_sqrt(y, rhs)) return DGPU_E_BADARG;
            if (is_high2(y) != ((flags & FLAG_LARGEST) != 0)) y = y.neg();
        } else {
            be48_to_limbs(c, b + 96); if (!lt_p(c)) return DGPU_E_BADARG; y.c1 = from_canonical(c);
            be48_to_limbs(c, b + 144); if (!lt_p(c)) return DGPU_E_BADARG; y.c0 = from_canonical(c);
            if (!(y.sqr() == rhs)) return DGPU_E_BADARG;
        }
        if (validate && !in_prime_subgroup<Fq2>(x, y)) return DGPU_E_BADARG;               // Validate::Yes: on the twist but outside G2
        memcpy(xy + 24 * i, &x, 96); memcpy(xy + 24 * i + 12, &y, 96);
    }
    return DGPU_OK;
}

}  // extern "C"
