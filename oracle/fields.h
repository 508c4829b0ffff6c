/* oracle/fields.h — TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * Plain-C restatement of the BLS12-381 field tower that the reference reaches through
 * ark-ff 0.4 (`Fp<MontBackend<_,6>,6>`, little-endian u64 limbs, value stored x R, R = 2^384;
 * SURVEY.md Appendix A.5 / B).  ark-ff is a third-party crate that is NOT under /root/reference
 * (Cargo.toml:36 `ark-ff ^0.4.1`, no Cargo.lock) — this follows its published Montgomery
 * representation so that raw limbs crossing the C ABI mean the same thing on both sides.
 * Parity unpinned against a real arkworks run (no Rust toolchain here); pinned against
 * oracle/bls12_381_model.py and the golden fixtures in tests/golden/.
 */
#ifndef ORACLE_FIELDS_H
#define ORACLE_FIELDS_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fp;
typedef struct { fp c0, c1; } fp2;
typedef struct { fp2 c0, c1, c2; } fp6;
typedef struct { fp6 c0, c1; } fp12;

static const fp FP_P   = {{0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL}};
static const fp FP_ONE = {{0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}};
static const fp FP_R2  = {{0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL}};
static const uint64_t FP_INV = 0x89f3fffcfffcfffdULL;
static const uint64_t FP_PM2[6]   = {0xb9feffffffffaaa9ULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t FP_PM1_6[6] = {0x49aa7ffffffff1c7ULL, 0x051caaaa72e35555ULL, 0xe688231ad3c82906ULL, 0xe613e1eb7deb831fULL, 0x0c849bf3b5e1f223ULL, 0x045582fc5eeaa66fULL};

/* ---------------- Fp ---------------- */
static inline int fp_is_zero(const fp *a) { uint64_t t = 0; for (int i = 0; i < 6; i++) t |= a->l[i]; return t == 0; }
static inline int fp_eq(const fp *a, const fp *b) { uint64_t t = 0; for (int i = 0; i < 6; i++) t |= a->l[i] ^ b->l[i]; return t == 0; }
static inline void fp_zero(fp *a) { memset(a, 0, sizeof *a); }
static inline void fp_one(fp *a) { *a = FP_ONE; }
static inline int fp_geq_p(const fp *a) {
    for (int i = 5; i >= 0; i--) { if (a->l[i] > FP_P.l[i]) return 1; if (a->l[i] < FP_P.l[i]) return 0; }
    return 1;
}
static inline void fp_sub_p(fp *a) {
    uint64_t br = 0;
    for (int i = 0; i < 6; i++) { u128 d = (u128)a->l[i] - FP_P.l[i] - br; a->l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
static inline void fp_add(fp *r, const fp *a, const fp *b) {
    uint64_t c = 0;
    for (int i = 0; i < 6; i++) { u128 s = (u128)a->l[i] + b->l[i] + c; r->l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    if (fp_geq_p(r)) fp_sub_p(r);   /* p < 2^381: no carry out of 384 bits */
}
static inline void fp_sub(fp *r, const fp *a, const fp *b) {
    uint64_t br = 0;
    for (int i = 0; i < 6; i++) { u128 d = (u128)a->l[i] - b->l[i] - br; r->l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    if (br) { uint64_t c = 0; for (int i = 0; i < 6; i++) { u128 s = (u128)r->l[i] + FP_P.l[i] + c; r->l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
}
static inline void fp_neg(fp *r, const fp *a) { if (fp_is_zero(a)) { *r = *a; return; } fp z; fp_zero(&z); fp_sub(r, &z, a); }
static inline void fp_dbl(fp *r, const fp *a) { fp_add(r, a, a); }
/* Montgomery product, r = a*b/R mod p: coarsely integrated operand scanning in the "no-carry" form ark-ff 0.4 uses for moduli whose top
 * limb leaves a spare bit (p < 2^381: the running value never needs a seventh word), one 64 x 64 -> 128 multiply-add per limb pair (mulx
 * under -mbmi2).  This is the stand-in for ark-ff's Montgomery backend in the timed CPU baseline (bench.py reports ns per product next to it). */
static inline void fp_mul(fp *r, const fp *a, const fp *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
        const uint64_t bi = b->l[i];
        u128 A = (u128)a->l[0] * bi + t[0];
        const uint64_t m = (uint64_t)A * FP_INV;
        u128 C = (u128)m * FP_P.l[0] + (uint64_t)A;
        for (int j = 1; j < 6; j++) {
            A = (u128)a->l[j] * bi + t[j] + (uint64_t)(A >> 64);
            C = (u128)m * FP_P.l[j] + (uint64_t)A + (uint64_t)(C >> 64);
            t[j - 1] = (uint64_t)C;
        }
        t[5] = (uint64_t)(C >> 64) + (uint64_t)(A >> 64);
    }
    for (int i = 0; i < 6; i++) r->l[i] = t[i];
    if (fp_geq_p(r)) fp_sub_p(r);
}
static inline void fp_sqr(fp *r, const fp *a) { fp_mul(r, a, a); }
static inline void fp_pow(fp *r, const fp *a, const uint64_t *e, int nl) {
    fp acc = FP_ONE, base = *a;
    for (int i = 0; i < nl * 64; i++) { if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, &base); fp_sqr(&base, &base); }
    *r = acc;
}
static inline void fp_inv(fp *r, const fp *a) { fp_pow(r, a, FP_PM2, 6); }
static inline void fp_to_mont(fp *r, const fp *a) { fp_mul(r, a, &FP_R2); }
static inline void fp_from_mont(fp *r, const fp *a) { fp one = {{1, 0, 0, 0, 0, 0}}; fp_mul(r, a, &one); }

/* ---------------- Fp2 = Fp[u]/(u^2+1) ---------------- */
static inline int fp2_is_zero(const fp2 *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static inline int fp2_eq(const fp2 *a, const fp2 *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static inline void fp2_zero(fp2 *a) { memset(a, 0, sizeof *a); }
static inline void fp2_one(fp2 *a) { a->c0 = FP_ONE; fp_zero(&a->c1); }
static inline void fp2_add(fp2 *r, const fp2 *a, const fp2 *b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
static inline void fp2_sub(fp2 *r, const fp2 *a, const fp2 *b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
static inline void fp2_neg(fp2 *r, const fp2 *a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static inline void fp2_dbl(fp2 *r, const fp2 *a) { fp2_add(r, a, a); }
static inline void fp2_conj(fp2 *r, const fp2 *a) { r->c0 = a->c0; fp_neg(&r->c1, &a->c1); }
static inline void fp2_mul(fp2 *r, const fp2 *a, const fp2 *b) {
    fp t0, t1, t2, t3;
    fp_mul(&t0, &a->c0, &b->c0); fp_mul(&t1, &a->c1, &b->c1);
    fp_add(&t2, &a->c0, &a->c1); fp_add(&t3, &b->c0, &b->c1);
    fp_mul(&t2, &t2, &t3); fp_sub(&t2, &t2, &t0); fp_sub(&t2, &t2, &t1);
    fp_sub(&r->c0, &t0, &t1); r->c1 = t2;
}
static inline void fp2_sqr(fp2 *r, const fp2 *a) {
    fp t0, t1, t2;
    fp_add(&t0, &a->c0, &a->c1); fp_sub(&t1, &a->c0, &a->c1); fp_mul(&t2, &a->c0, &a->c1);
    fp_mul(&r->c0, &t0, &t1); fp_dbl(&r->c1, &t2);
}
static inline void fp2_mul_fp(fp2 *r, const fp2 *a, const fp *k) { fp_mul(&r->c0, &a->c0, k); fp_mul(&r->c1, &a->c1, k); }
static inline void fp2_mul_xi(fp2 *r, const fp2 *a) { fp t; fp_sub(&t, &a->c0, &a->c1); fp_add(&r->c1, &a->c0, &a->c1); r->c0 = t; }
static inline void fp2_inv(fp2 *r, const fp2 *a) {
    fp n, t; fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t); fp_inv(&n, &n);
    fp_mul(&r->c0, &a->c0, &n); fp_mul(&t, &a->c1, &n); fp_neg(&r->c1, &t);
}
static inline void fp2_pow(fp2 *r, const fp2 *a, const uint64_t *e, int nl) {
    fp2 acc, base = *a; fp2_one(&acc);
    for (int i = 0; i < nl * 64; i++) { if ((e[i / 64] >> (i % 64)) & 1) fp2_mul(&acc, &acc, &base); fp2_sqr(&base, &base); }
    *r = acc;
}

/* ---------------- Fp6 = Fp2[v]/(v^3 - xi), xi = 1+u ---------------- */
static inline void fp6_zero(fp6 *a) { memset(a, 0, sizeof *a); }
static inline void fp6_one(fp6 *a) { fp6_zero(a); a->c0.c0 = FP_ONE; }
static inline void fp6_add(fp6 *r, const fp6 *a, const fp6 *b) { fp2_add(&r->c0, &a->c0, &b->c0); fp2_add(&r->c1, &a->c1, &b->c1); fp2_add(&r->c2, &a->c2, &b->c2); }
static inline void fp6_sub(fp6 *r, const fp6 *a, const fp6 *b) { fp2_sub(&r->c0, &a->c0, &b->c0); fp2_sub(&r->c1, &a->c1, &b->c1); fp2_sub(&r->c2, &a->c2, &b->c2); }
static inline void fp6_neg(fp6 *r, const fp6 *a) { fp2_neg(&r->c0, &a->c0); fp2_neg(&r->c1, &a->c1); fp2_neg(&r->c2, &a->c2); }
static inline void fp6_mul(fp6 *r, const fp6 *a, const fp6 *b) {
    fp2 t0, t1, t2, s, u, c0, c1, c2;
    fp2_mul(&t0, &a->c0, &b->c0); fp2_mul(&t1, &a->c1, &b->c1); fp2_mul(&t2, &a->c2, &b->c2);
    /* c0 = t0 + xi*((a1+a2)(b1+b2) - t1 - t2) */
    fp2_add(&s, &a->c1, &a->c2); fp2_add(&u, &b->c1, &b->c2); fp2_mul(&c0, &s, &u); fp2_sub(&c0, &c0, &t1); fp2_sub(&c0, &c0, &t2); fp2_mul_xi(&c0, &c0); fp2_add(&c0, &c0, &t0);
    /* c1 = (a0+a1)(b0+b1) - t0 - t1 + xi*t2 */
    fp2_add(&s, &a->c0, &a->c1); fp2_add(&u, &b->c0, &b->c1); fp2_mul(&c1, &s, &u); fp2_sub(&c1, &c1, &t0); fp2_sub(&c1, &c1, &t1); fp2_mul_xi(&s, &t2); fp2_add(&c1, &c1, &s);
    /* c2 = (a0+a2)(b0+b2) - t0 - t2 + t1 */
    fp2_add(&s, &a->c0, &a->c2); fp2_add(&u, &b->c0, &b->c2); fp2_mul(&c2, &s, &u); fp2_sub(&c2, &c2, &t0); fp2_sub(&c2, &c2, &t2); fp2_add(&c2, &c2, &t1);
    r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
static inline void fp6_mul_v(fp6 *r, const fp6 *a) { fp2 t; fp2_mul_xi(&t, &a->c2); r->c2 = a->c1; r->c1 = a->c0; r->c0 = t; }
static inline void fp6_inv(fp6 *r, const fp6 *a) {
    fp2 t0, t1, t2, s, n;
    fp2_sqr(&t0, &a->c0); fp2_mul(&s, &a->c1, &a->c2); fp2_mul_xi(&s, &s); fp2_sub(&t0, &t0, &s);
    fp2_sqr(&t1, &a->c2); fp2_mul_xi(&t1, &t1); fp2_mul(&s, &a->c0, &a->c1); fp2_sub(&t1, &t1, &s);
    fp2_sqr(&t2, &a->c1); fp2_mul(&s, &a->c0, &a->c2); fp2_sub(&t2, &t2, &s);
    fp2_mul(&n, &a->c2, &t1); fp2_mul(&s, &a->c1, &t2); fp2_add(&n, &n, &s); fp2_mul_xi(&n, &n); fp2_mul(&s, &a->c0, &t0); fp2_add(&n, &n, &s);
    fp2_inv(&n, &n);
    fp2_mul(&r->c0, &t0, &n); fp2_mul(&r->c1, &t1, &n); fp2_mul(&r->c2, &t2, &n);
}

/* ---------------- Fp12 = Fp6[w]/(w^2 - v) ---------------- */
static inline void fp12_one(fp12 *a) { fp6_one(&a->c0); fp6_zero(&a->c1); }
static inline int fp12_eq(const fp12 *a, const fp12 *b) { return memcmp(a, b, sizeof *a) == 0; }   /* canonical limbs */
static inline int fp12_is_zero(const fp12 *a) { const uint64_t *w = (const uint64_t *)a; uint64_t t = 0; for (int i = 0; i < 72; i++) t |= w[i]; return t == 0; }
static inline void fp12_mul(fp12 *r, const fp12 *a, const fp12 *b) {
    fp6 t0, t1, s, u, c1;
    fp6_mul(&t0, &a->c0, &b->c0); fp6_mul(&t1, &a->c1, &b->c1);
    fp6_add(&s, &a->c0, &a->c1); fp6_add(&u, &b->c0, &b->c1); fp6_mul(&c1, &s, &u); fp6_sub(&c1, &c1, &t0); fp6_sub(&c1, &c1, &t1);
    fp6_mul_v(&s, &t1); fp6_add(&r->c0, &t0, &s); r->c1 = c1;
}
static inline void fp12_sqr(fp12 *r, const fp12 *a) { fp12_mul(r, a, a); }
static inline void fp12_conj(fp12 *r, const fp12 *a) { r->c0 = a->c0; fp6_neg(&r->c1, &a->c1); }
static inline void fp12_inv(fp12 *r, const fp12 *a) {
    fp6 n, t; fp6_mul(&n, &a->c0, &a->c0); fp6_mul(&t, &a->c1, &a->c1); fp6_mul_v(&t, &t); fp6_sub(&n, &n, &t); fp6_inv(&n, &n);
    fp6_mul(&r->c0, &a->c0, &n); fp6_mul(&t, &a->c1, &n); fp6_neg(&r->c1, &t);
}
/* f * (c0 + c1 v + c4 v w): ark-ff Fp12::mul_by_014, done here as a full product with the sparse operand */
static inline void fp12_mul_by_014(fp12 *f, const fp2 *c0, const fp2 *c1, const fp2 *c4) {
    fp12 s; memset(&s, 0, sizeof s); s.c0.c0 = *c0; s.c0.c1 = *c1; s.c1.c1 = *c4;
    fp12_mul(f, f, &s);
}
#endif
