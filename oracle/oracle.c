/* oracle/oracle.c — TEST INFRASTRUCTURE ONLY (CPU oracle + timed CPU baseline "port").
 *
 * Plain-C restatement of the hot path the reference (docknetwork/crypto) enters through arkworks:
 *   G::Group::msm_unchecked / msm_bigint      utils/src/pairs.rs:143-156, utils/src/owned_pairs.rs:93-106,
 *                                             legogroth16/src/prover.rs:286,299,585-594
 *   E::multi_miller_loop / final_exponentiation
 *                                             utils/src/randomized_pairing_check.rs:204-214,
 *                                             legogroth16/src/verifier.rs:62-84
 * The algorithms live in third-party crates absent from /root/reference (ark-ec/ark-ff ^0.4.1 ->
 * 0.4.2, ark-bls12-381 ^0.4.0; Cargo.toml:36-50, no Cargo.lock): they are restated from the
 * published sources as summarised in SURVEY.md Appendix A (A.1 msm_bigint_wnaf, A.2 make_digits,
 * A.3 Miller loop with M-twist line coefficients, A.4 final-exponentiation chain).
 *
 * PARITY UNPINNED against a real arkworks run: the reference keeps no known-answer vectors for this
 * path and has no buildable source here (Rust toolchain absent).  Pinned instead against the
 * independent big-integer model oracle/bls12_381_model.py (fixtures in tests/golden/) and the
 * algebraic identities the reference's own tests assert (utils/src/msm.rs:186-193,268-275).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
 */
#include <stdlib.h>
#include <pthread.h>
#include "fields.h"

/* ---------------- ark-ec window rule + signed digits (A.1 / A.2) ---------------- */
static int ark_log2(size_t x) {
    if (x == 0) return 0;
    int bl = 64 - __builtin_clzll((unsigned long long)x);
    return ((x & (x - 1)) == 0) ? bl - 1 : bl;
}
static int ark_window_c(size_t n) { return n < 32 ? 3 : ark_log2(n) * 69 / 100 + 2; }
static void ark_make_digits(int64_t *out, const uint64_t a[4], int w, int num_bits) {
    uint64_t radix = 1ULL << w, mask = radix - 1, carry = 0;
    int dc = (num_bits + w - 1) / w;
    for (int i = 0; i < dc; i++) {
        int bo = i * w, u = bo / 64, b = bo % 64;
        uint64_t buf;
        if (b < 64 - w || u == 3) buf = a[u] >> b;
        else buf = (a[u] >> b) | (a[u + 1] << (64 - b));
        uint64_t coef = carry + (buf & mask);
        carry = (coef + radix / 2) >> w;
        out[i] = (int64_t)coef - (int64_t)(carry << w);
    }
    out[dc - 1] += (int64_t)(carry << w);
}

#define FE fp
#define F(x) fp_##x
#define EC(x) g1_##x
#include "ec_tmpl.inc"
#undef FE
#undef F
#undef EC
#define FE fp2
#define F(x) fp2_##x
#define EC(x) g2_##x
#include "ec_tmpl.inc"
#undef FE
#undef F
#undef EC

static const g1_aff G1_GEN = {
    {{0x5cb38790fd530c16ULL, 0x7817fc679976fff5ULL, 0x154f95c7143ba1c1ULL, 0xf0ae6acdf3d0e747ULL, 0xedce6ecc21dbf440ULL, 0x120177419e0bfb75ULL}},
    {{0xbaac93d50ce72271ULL, 0x8c22631a7918fd8eULL, 0xdd595f13570725ceULL, 0x51ac582950405194ULL, 0x0e1c8c3fad0059c0ULL, 0x0bbc3efc5008a26aULL}}};
static const g2_aff G2_GEN = {
    {{{0xf5f28fa202940a10ULL, 0xb3f5fb2687b4961aULL, 0xa1a893b53e2ae580ULL, 0x9894999d1a3caee9ULL, 0x6f67b7631863366bULL, 0x058191924350bcd7ULL}},
     {{0xa5a9c0759e23f606ULL, 0xaaa0c59dbccd60c3ULL, 0x3bb17e18e2867806ULL, 0x1b1ab6cc8541b367ULL, 0xc2b6ed0ef2158547ULL, 0x11922a097360edf3ULL}}},
    {{{0x4c730af860494c4aULL, 0x597cfa1f5e369c5aULL, 0xe7e6856caa0a635aULL, 0xbbefb5e96e0d495fULL, 0x07d3a975f0ef25a2ULL, 0x0083fd8e7e80dae5ULL}},
     {{0xadc0fc92df64b05dULL, 0x18aa270a2b1461dcULL, 0x86adac6a3be4eba0ULL, 0x79495c4ec93da33aULL, 0xe7175850a43ccaedULL, 0x0b2bc2a163de1bf2ULL}}}};

/* ---------------- Fr (scalar field) Montgomery <-> canonical: Fr::into_bigint / from ---------------- */
static const uint64_t FR_MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static const uint64_t FR_R2[4]  = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
static const uint64_t FR_INV = 0xfffffffeffffffffULL;
static void fr_mont_mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t t[6] = {0};
    for (int i = 0; i < 4; i++) {
        uint64_t c = 0; u128 s;
        for (int j = 0; j < 4; j++) { s = (u128)a[j] * b[i] + t[j] + c; t[j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        s = (u128)t[4] + c; t[4] = (uint64_t)s; t[5] = (uint64_t)(s >> 64);
        uint64_t m = t[0] * FR_INV;
        s = (u128)m * FR_MOD[0] + t[0]; c = (uint64_t)(s >> 64);
        for (int j = 1; j < 4; j++) { s = (u128)m * FR_MOD[j] + t[j] + c; t[j - 1] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        s = (u128)t[4] + c; t[3] = (uint64_t)s; t[4] = t[5] + (uint64_t)(s >> 64);
    }
    int ge = t[4] != 0;
    if (!ge) { ge = 1; for (int i = 3; i >= 0; i--) { if (t[i] > FR_MOD[i]) break; if (t[i] < FR_MOD[i]) { ge = 0; break; } } }
    if (ge) { uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - FR_MOD[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }
    for (int i = 0; i < 4; i++) r[i] = t[i];
}

/* ---------------- Frobenius on Fp12 (coefficients derived, not copied) ---------------- */
static fp2 FROB1[6];   /* xi^(i (p-1)/6) */
static fp  FROB2[6];   /* xi^(i (p^2-1)/6) = norm(FROB1[i]) in Fp */
static pthread_once_t frob_once = PTHREAD_ONCE_INIT;
static void frob_init(void) {
    fp2 xi; xi.c0 = FP_ONE; xi.c1 = FP_ONE;
    fp2 g; fp2_pow(&g, &xi, FP_PM1_6, 6);
    fp2_one(&FROB1[0]);
    for (int i = 1; i < 6; i++) fp2_mul(&FROB1[i], &FROB1[i - 1], &g);
    for (int i = 0; i < 6; i++) { fp a, b; fp_sqr(&a, &FROB1[i].c0); fp_sqr(&b, &FROB1[i].c1); fp_add(&FROB2[i], &a, &b); }
}
static void fp12_frob1(fp12 *r, const fp12 *a) {
    pthread_once(&frob_once, frob_init);
    fp12 t;
    fp2_conj(&t.c0.c0, &a->c0.c0);
    fp2_conj(&t.c0.c1, &a->c0.c1); fp2_mul(&t.c0.c1, &t.c0.c1, &FROB1[2]);
    fp2_conj(&t.c0.c2, &a->c0.c2); fp2_mul(&t.c0.c2, &t.c0.c2, &FROB1[4]);
    fp2_conj(&t.c1.c0, &a->c1.c0); fp2_mul(&t.c1.c0, &t.c1.c0, &FROB1[1]);
    fp2_conj(&t.c1.c1, &a->c1.c1); fp2_mul(&t.c1.c1, &t.c1.c1, &FROB1[3]);
    fp2_conj(&t.c1.c2, &a->c1.c2); fp2_mul(&t.c1.c2, &t.c1.c2, &FROB1[5]);
    *r = t;
}
static void fp12_frob2(fp12 *r, const fp12 *a) {
    pthread_once(&frob_once, frob_init);
    fp12 t;
    t.c0.c0 = a->c0.c0;
    fp2_mul_fp(&t.c0.c1, &a->c0.c1, &FROB2[2]);
    fp2_mul_fp(&t.c0.c2, &a->c0.c2, &FROB2[4]);
    fp2_mul_fp(&t.c1.c0, &a->c1.c0, &FROB2[1]);
    fp2_mul_fp(&t.c1.c1, &a->c1.c1, &FROB2[3]);
    fp2_mul_fp(&t.c1.c2, &a->c1.c2, &FROB2[5]);
    *r = t;
}

/* ---------------- Miller loop (A.3) ---------------- */
#define X_ABS 0xd201000000010000ULL
#define N_COEFF 68
typedef struct { fp2 c0, c1, c2; } ell_coeff;

static fp TWO_INV;
static fp2 B_TWIST;
static pthread_once_t ml_once = PTHREAD_ONCE_INIT;
static void ml_init(void) {
    fp two; fp_add(&two, &FP_ONE, &FP_ONE); fp_inv(&TWO_INV, &two);
    fp four; fp_add(&four, &two, &two); B_TWIST.c0 = four; B_TWIST.c1 = four;
}
static void dbl_step(fp2 R[3], ell_coeff *co) {
    fp2 a, b, c, e, f, g, h, i, j, e2, t;
    fp2_mul(&a, &R[0], &R[1]); fp2_mul_fp(&a, &a, &TWO_INV);
    fp2_sqr(&b, &R[1]); fp2_sqr(&c, &R[2]);
    fp2_dbl(&t, &c); fp2_add(&t, &t, &c); fp2_mul(&e, &B_TWIST, &t);
    fp2_dbl(&f, &e); fp2_add(&f, &f, &e);
    fp2_add(&g, &b, &f); fp2_mul_fp(&g, &g, &TWO_INV);
    fp2_add(&h, &R[1], &R[2]); fp2_sqr(&h, &h); fp2_add(&t, &b, &c); fp2_sub(&h, &h, &t);
    fp2_sub(&i, &e, &b);
    fp2_sqr(&j, &R[0]);
    fp2_sqr(&e2, &e);
    fp2_sub(&t, &b, &f); fp2_mul(&R[0], &a, &t);
    fp2_sqr(&g, &g); fp2_dbl(&t, &e2); fp2_add(&t, &t, &e2); fp2_sub(&R[1], &g, &t);
    fp2_mul(&R[2], &b, &h);
    co->c0 = i; fp2_dbl(&t, &j); fp2_add(&co->c1, &t, &j); fp2_neg(&co->c2, &h);
}
static void add_step(fp2 R[3], const g2_aff *Q, ell_coeff *co) {
    fp2 theta, lam, c, d, e, f, g, h, j, t, u;
    fp2_mul(&t, &Q->y, &R[2]); fp2_sub(&theta, &R[1], &t);
    fp2_mul(&t, &Q->x, &R[2]); fp2_sub(&lam, &R[0], &t);
    fp2_sqr(&c, &theta); fp2_sqr(&d, &lam); fp2_mul(&e, &lam, &d);
    fp2_mul(&f, &R[2], &c); fp2_mul(&g, &R[0], &d);
    fp2_add(&h, &e, &f); fp2_dbl(&t, &g); fp2_sub(&h, &h, &t);
    fp2_sub(&t, &g, &h); fp2_mul(&t, &theta, &t); fp2_mul(&u, &e, &R[1]);
    fp2_mul(&R[0], &lam, &h);
    fp2_sub(&R[1], &t, &u);
    fp2_mul(&R[2], &R[2], &e);
    fp2_mul(&t, &theta, &Q->x); fp2_mul(&u, &lam, &Q->y); fp2_sub(&j, &t, &u);
    co->c0 = j; fp2_neg(&co->c1, &theta); co->c2 = lam;
}
static void g2_prepare(const g2_aff *Q, ell_coeff out[N_COEFF]) {
    pthread_once(&ml_once, ml_init);
    fp2 R[3]; R[0] = Q->x; R[1] = Q->y; fp2_one(&R[2]);
    int k = 0;
    for (int i = 62; i >= 0; i--) {
        dbl_step(R, &out[k++]);
        if ((X_ABS >> i) & 1) add_step(R, Q, &out[k++]);
    }
}
static void ell(fp12 *f, const ell_coeff *co, const g1_aff *P) {
    fp2 c1, c2; fp2_mul_fp(&c1, &co->c1, &P->x); fp2_mul_fp(&c2, &co->c2, &P->y);
    fp12_mul_by_014(f, &co->c0, &c1, &c2);
}
/* one rayon chunk (<= 4 pairs) of ark-ec's multi_miller_loop, without the final conjugation */
static void miller_chunk(fp12 *f, const g1_aff *ps, const ell_coeff *coeffs, size_t m) {
    fp12_one(f);
    int k = 0;
    for (int i = 62; i >= 0; i--) {
        fp12_sqr(f, f);
        for (size_t j = 0; j < m; j++) ell(f, &coeffs[j * N_COEFF + k], &ps[j]);
        k++;
        if ((X_ABS >> i) & 1) { for (size_t j = 0; j < m; j++) ell(f, &coeffs[j * N_COEFF + k], &ps[j]); k++; }
    }
}
typedef struct { const g1_aff *ps; const ell_coeff *co; size_t m; fp12 *partial; volatile int *next; int nchunks; } ml_job;
static void *ml_worker(void *arg) {
    ml_job *J = (ml_job *)arg;
    for (;;) {
        int ch = __sync_fetch_and_add(J->next, 1); if (ch >= J->nchunks) break;
        size_t lo = (size_t)ch * 4, hi = lo + 4 > J->m ? J->m : lo + 4;
        miller_chunk(&J->partial[ch], J->ps + lo, J->co + lo * N_COEFF, hi - lo);
    }
    return NULL;
}
static void cyclo_exp_x(fp12 *r, const fp12 *a) {   /* a^|x| then conj (x < 0) */
    fp12 acc; fp12_one(&acc);
    for (int i = 63; i >= 0; i--) { fp12_sqr(&acc, &acc); if ((X_ABS >> i) & 1) fp12_mul(&acc, &acc, a); }
    fp12_conj(r, &acc);
}

/* ======================= exported C API (ctypes) ======================= */
#define API __attribute__((visibility("default")))

API void orc_fp_to_mont(const uint64_t *in, uint64_t *out, size_t n) { for (size_t i = 0; i < n; i++) fp_to_mont((fp *)(out + 6 * i), (const fp *)(in + 6 * i)); }
API void orc_fp_from_mont(const uint64_t *in, uint64_t *out, size_t n) { for (size_t i = 0; i < n; i++) fp_from_mont((fp *)(out + 6 * i), (const fp *)(in + 6 * i)); }
API void orc_fr_to_mont(const uint64_t *in, uint64_t *out, size_t n) { for (size_t i = 0; i < n; i++) fr_mont_mul(out + 4 * i, in + 4 * i, FR_R2); }
API void orc_fr_from_mont(const uint64_t *in, uint64_t *out, size_t n) { uint64_t one[4] = {1, 0, 0, 0}; for (size_t i = 0; i < n; i++) fr_mont_mul(out + 4 * i, in + 4 * i, one); }
API int  orc_window_c(size_t n) { return ark_window_c(n); }
API void orc_make_digits(const uint64_t *s, int c, int64_t *out) { ark_make_digits(out, s, c, 255); }

/* SplitMix64 scalar stream, identical to bls12_381_model.SplitMix64.scalar() */
static uint64_t sm64(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static void rand_scalar(uint64_t *st, uint64_t out[4]) {
    for (;;) {
        for (int i = 0; i < 4; i++) out[i] = sm64(st);
        out[3] &= 0x7fffffffffffffffULL;
        int lt = 0; for (int i = 3; i >= 0; i--) { if (out[i] < FR_MOD[i]) { lt = 1; break; } if (out[i] > FR_MOD[i]) break; }
        if (lt) return;
    }
}
API void orc_rand_scalars(uint64_t seed, size_t n, uint64_t *out) { uint64_t st = seed; for (size_t i = 0; i < n; i++) rand_scalar(&st, out + 4 * i); }

/* ---- G1 ---- */
API void orc_g1_generator(uint64_t out[12]) { memcpy(out, &G1_GEN, 96); }
API void orc_g1_msm(const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars, size_t n, int threads, uint64_t out[18]) {
    g1_msm_bigint((g1_jac *)out, (const g1_aff *)bases, inf, scalars, n, threads);
}
API int orc_g1_to_affine(const uint64_t in[18], uint64_t out[12]) { return g1_to_affine((g1_aff *)out, (const g1_jac *)in); }
API void orc_g1_mul(const uint64_t base[12], int inf, const uint64_t k[4], uint64_t out[18]) { g1_mul((g1_jac *)out, (const g1_aff *)base, inf, k); }
API void orc_g1_add(const uint64_t a[18], const uint64_t b[18], uint64_t out[18]) { g1_jac r; g1_add(&r, (const g1_jac *)a, (const g1_jac *)b); memcpy(out, &r, sizeof r); }
API int orc_g1_on_curve(const uint64_t xy[12]) {
    const g1_aff *p = (const g1_aff *)xy; fp l, r, four;
    fp_sqr(&l, &p->y); fp_sqr(&r, &p->x); fp_mul(&r, &r, &p->x);
    fp_add(&four, &FP_ONE, &FP_ONE); fp_add(&four, &four, &four); fp_add(&r, &r, &four);
    return fp_eq(&l, &r);
}
/* ---- G2 ---- */
/* timed CPU baseline, calibration: ns per dependent Fp product and per G1 mixed addition on one core (bench.py prints them next to
 * cpu_baseline so that the stand-in can be placed against ark-ff's backend: ~25-30 ns per 381-bit Montgomery product on current x86) */
#include <time.h>
static double now_ns(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e9 + ts.tv_nsec; }
API double orc_bench_fp_mul_ns(size_t iters) {
    fp a = FP_R2, b = FP_ONE; b.l[0] ^= 0x1234567;
    double t0 = now_ns();
    for (size_t i = 0; i < iters; i++) { fp_mul(&a, &a, &b); fp_mul(&b, &b, &a); }
    double dt = now_ns() - t0;
    volatile uint64_t sink = a.l[0] ^ b.l[0]; (void)sink;
    return dt / (2.0 * iters);
}
API double orc_bench_g1_madd_ns(size_t iters) {
    g1_jac acc; g1_aff q = G1_GEN;
    uint64_t k[4] = {0x123456789abcdefULL, 0, 0, 0};
    g1_mul(&acc, &G1_GEN, 0, k);
    double t0 = now_ns();
    for (size_t i = 0; i < iters; i++) g1_madd(&acc, &acc, &q);
    double dt = now_ns() - t0;
    volatile uint64_t sink = acc.x.l[0]; (void)sink;
    return dt / iters;
}
API void orc_g2_generator(uint64_t out[24]) { memcpy(out, &G2_GEN, 192); }
API void orc_g2_msm(const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars, size_t n, int threads, uint64_t out[36]) {
    g2_msm_bigint((g2_jac *)out, (const g2_aff *)bases, inf, scalars, n, threads);
}
API int orc_g2_to_affine(const uint64_t in[36], uint64_t out[24]) { return g2_to_affine((g2_aff *)out, (const g2_jac *)in); }
API void orc_g2_mul(const uint64_t base[24], int inf, const uint64_t k[4], uint64_t out[36]) { g2_mul((g2_jac *)out, (const g2_aff *)base, inf, k); }
API void orc_g2_add(const uint64_t a[36], const uint64_t b[36], uint64_t out[36]) { g2_jac r; g2_add(&r, (const g2_jac *)a, (const g2_jac *)b); memcpy(out, &r, sizeof r); }
API int orc_g2_on_curve(const uint64_t xy[24]) {
    pthread_once(&ml_once, ml_init);
    const g2_aff *p = (const g2_aff *)xy; fp2 l, r;
    fp2_sqr(&l, &p->y); fp2_sqr(&r, &p->x); fp2_mul(&r, &r, &p->x); fp2_add(&r, &r, &B_TWIST);
    return fp2_eq(&l, &r);
}

/* ---- synthetic bases with known discrete logs: P_i = (k0 + i*d) * G, chunk-parallel ---- */
typedef struct { int g2; uint64_t k0[4], d[4]; size_t lo, hi; uint64_t *out; } gen_job;
static void fr_add_mod(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t c = 0; uint64_t t[4];
    for (int i = 0; i < 4; i++) { u128 s = (u128)a[i] + b[i] + c; t[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    int ge = 1; for (int i = 3; i >= 0; i--) { if (t[i] > FR_MOD[i]) break; if (t[i] < FR_MOD[i]) { ge = 0; break; } }
    if (c || ge) { uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 dd = (u128)t[i] - FR_MOD[i] - br; t[i] = (uint64_t)dd; br = (uint64_t)(dd >> 64) & 1; } }
    memcpy(r, t, 32);
}
static void fr_mul_small(uint64_t r[4], const uint64_t a[4], uint64_t k) {   /* a*k mod r via double-and-add */
    uint64_t acc[4] = {0, 0, 0, 0}, b[4]; memcpy(b, a, 32);
    while (k) { if (k & 1) fr_add_mod(acc, acc, b); fr_add_mod(b, b, b); k >>= 1; }
    memcpy(r, acc, 32);
}
static void *gen_worker(void *arg) {
    gen_job *J = (gen_job *)arg;
    size_t n = J->hi - J->lo; if (!n) return NULL;
    uint64_t ks[4], t[4]; fr_mul_small(t, J->d, J->lo); fr_add_mod(ks, J->k0, t);
    if (!J->g2) {
        g1_jac *pts = (g1_jac *)malloc(sizeof(g1_jac) * n); g1_jac D; g1_aff Da;
        g1_mul(&pts[0], &G1_GEN, 0, ks); g1_mul(&D, &G1_GEN, 0, J->d); g1_to_affine(&Da, &D);
        for (size_t i = 1; i < n; i++) g1_madd(&pts[i], &pts[i - 1], &Da);
        g1_batch_to_affine((g1_aff *)(J->out + 12 * J->lo), NULL, pts, n); free(pts);
    } else {
        g2_jac *pts = (g2_jac *)malloc(sizeof(g2_jac) * n); g2_jac D; g2_aff Da;
        g2_mul(&pts[0], &G2_GEN, 0, ks); g2_mul(&D, &G2_GEN, 0, J->d); g2_to_affine(&Da, &D);
        for (size_t i = 1; i < n; i++) g2_madd(&pts[i], &pts[i - 1], &Da);
        g2_batch_to_affine((g2_aff *)(J->out + 24 * J->lo), NULL, pts, n); free(pts);
    }
    return NULL;
}
static void gen_seq(int g2, const uint64_t k0[4], const uint64_t d[4], size_t n, int threads, uint64_t *out) {
    if (threads < 1) threads = 1; if (threads > 64) threads = 64;
    gen_job jobs[64]; pthread_t th[64];
    size_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        jobs[t].g2 = g2; memcpy(jobs[t].k0, k0, 32); memcpy(jobs[t].d, d, 32);
        jobs[t].lo = per * t > n ? n : per * t; jobs[t].hi = per * (t + 1) > n ? n : per * (t + 1); jobs[t].out = out;
        pthread_create(&th[t], NULL, gen_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}
/* P_i = (k0 + i d) G1 (no identity unless k0 + i d == 0 mod r, which the callers avoid) */
API void orc_g1_gen_seq(const uint64_t k0[4], const uint64_t d[4], size_t n, int threads, uint64_t *out_xy) { gen_seq(0, k0, d, n, threads, out_xy); }
API void orc_g2_gen_seq(const uint64_t k0[4], const uint64_t d[4], size_t n, int threads, uint64_t *out_xy) { gen_seq(1, k0, d, n, threads, out_xy); }


/* ---------------- R1CS -> QAP witness map (reference legogroth16/src/r1cs_to_qap.rs:150-210) ----------------
 * ark-poly Radix2EvaluationDomain semantics restated: omega_D = 7^((r-1)/D), coset generator g = Fr::GENERATOR = 7.
 * All Fr values below are in Montgomery form (R = 2^256) internally; the API takes / returns canonical limbs. */
typedef struct { uint64_t l[4]; } fr;
static const fr FR_ONE_M = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}};
static void fr_mul(fr *r, const fr *a, const fr *b) { fr_mont_mul(r->l, a->l, b->l); }
static void fr_add(fr *r, const fr *a, const fr *b) { fr_add_mod(r->l, a->l, b->l); }
static void fr_sub(fr *r, const fr *a, const fr *b) {
    uint64_t br = 0, t[4];
    for (int i = 0; i < 4; i++) { u128 d = (u128)a->l[i] - b->l[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    if (br) { uint64_t c = 0; for (int i = 0; i < 4; i++) { u128 s2 = (u128)t[i] + FR_MOD[i] + c; t[i] = (uint64_t)s2; c = (uint64_t)(s2 >> 64); } }
    memcpy(r->l, t, 32);
}
static void fr_pow(fr *r, const fr *a, const uint64_t *e, int nl) {
    fr acc = FR_ONE_M, base = *a;
    for (int i = 0; i < nl * 64; i++) { if ((e[i / 64] >> (i % 64)) & 1) fr_mul(&acc, &acc, &base); fr_mul(&base, &base, &base); }
    *r = acc;
}
static void fr_inv(fr *r, const fr *a) { uint64_t e[4] = {FR_MOD[0] - 2, FR_MOD[1], FR_MOD[2], FR_MOD[3]}; fr_pow(r, a, e, 4); }
static void fr_from_u64(fr *r, uint64_t v) { uint64_t t[4] = {v, 0, 0, 0}; fr_mont_mul(r->l, t, FR_R2); }
static void fr_root_of_unity(fr *w, int logn) {   /* 7^((r-1)/2^logn) */
    fr g; fr_from_u64(&g, 7);
    uint64_t e[4]; memcpy(e, FR_MOD, 32); e[0] -= 1;              /* r - 1 */
    for (int k = 0; k < logn; k++) { for (int i = 0; i < 3; i++) e[i] = (e[i] >> 1) | (e[i + 1] << 63); e[3] >>= 1; }
    fr_pow(w, &g, e, 4);
}
/* in-place radix-2 transform of a[0..n), natural order in and out; inverse divides by n */
static void fr_ntt(fr *a, int logn, int inverse) {
    size_t n = (size_t)1 << logn;
    for (size_t i = 0; i < n; i++) { size_t j = 0; for (int b = 0; b < logn; b++) if (i >> b & 1) j |= (size_t)1 << (logn - 1 - b); if (i < j) { fr t = a[i]; a[i] = a[j]; a[j] = t; } }
    fr w; fr_root_of_unity(&w, logn); if (inverse) fr_inv(&w, &w);
    for (int s = 1; s <= logn; s++) {
        size_t m = (size_t)1 << s, half = m >> 1;
        fr wm = w; for (int k = s; k < logn; k++) fr_mul(&wm, &wm, &wm);
        for (size_t k = 0; k < n; k += m) {
            fr tw = FR_ONE_M;
            for (size_t j = 0; j < half; j++) { fr t, u = a[k + j]; fr_mul(&t, &tw, &a[k + j + half]); fr_add(&a[k + j], &u, &t); fr_sub(&a[k + j + half], &u, &t); fr_mul(&tw, &tw, &wm); }
        }
    }
    if (inverse) { fr ninv; fr_from_u64(&ninv, n); fr_inv(&ninv, &ninv); for (size_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], &ninv); }
}
static void fr_coset_scale(fr *a, size_t n, const fr *g) { fr p = FR_ONE_M; for (size_t i = 0; i < n; i++) { fr_mul(&a[i], &a[i], &p); fr_mul(&p, &p, g); } }
API void orc_fr_ntt(uint64_t *vals /* n*4 canonical */, int logn, int inverse, int coset) {
    size_t n = (size_t)1 << logn; fr *a = (fr *)malloc(sizeof(fr) * n);
    for (size_t i = 0; i < n; i++) fr_mont_mul(a[i].l, vals + 4 * i, FR_R2);
    fr g; fr_from_u64(&g, 7);
    if (coset && !inverse) fr_coset_scale(a, n, &g);
    fr_ntt(a, logn, inverse);
    if (coset && inverse) { fr gi; fr_inv(&gi, &g); fr_coset_scale(a, n, &gi); }
    uint64_t one[4] = {1, 0, 0, 0};
    for (size_t i = 0; i < n; i++) fr_mont_mul(vals + 4 * i, a[i].l, one);
    free(a);
}
/* LibsnarkReduction::witness_map_from_matrices.  Matrices in CSR: rowptr[num_constraints+1], cols[], vals[] (canonical).
 * z = full assignment (canonical, num_inputs instance values first, z[0] = 1).  out_h: D canonical coefficients. */
static void csr_eval(fr *out, const uint64_t *rowptr, const uint32_t *cols, const uint64_t *vals, size_t rows, const fr *z) {
    for (size_t i = 0; i < rows; i++) {
        fr acc; memset(&acc, 0, sizeof acc);
        for (uint64_t k = rowptr[i]; k < rowptr[i + 1]; k++) { fr c, t; fr_mont_mul(c.l, vals + 4 * k, FR_R2); fr_mul(&t, &c, &z[cols[k]]); fr_add(&acc, &acc, &t); }
        out[i] = acc;
    }
}
API int orc_witness_map(const uint64_t *a_rowptr, const uint32_t *a_cols, const uint64_t *a_vals,
                        const uint64_t *b_rowptr, const uint32_t *b_cols, const uint64_t *b_vals,
                        const uint64_t *c_rowptr, const uint32_t *c_cols, const uint64_t *c_vals,
                        const uint64_t *z_canon, size_t num_vars, size_t num_inputs, size_t num_constraints, uint64_t *out_h) {
    int logn = 0; while (((size_t)1 << logn) < num_constraints + num_inputs) logn++;
    size_t D = (size_t)1 << logn;
    fr *z = (fr *)malloc(sizeof(fr) * num_vars), *a = (fr *)calloc(D, sizeof(fr)), *b = (fr *)calloc(D, sizeof(fr)), *c = (fr *)calloc(D, sizeof(fr));
    for (size_t i = 0; i < num_vars; i++) fr_mont_mul(z[i].l, z_canon + 4 * i, FR_R2);
    csr_eval(a, a_rowptr, a_cols, a_vals, num_constraints, z);
    csr_eval(b, b_rowptr, b_cols, b_vals, num_constraints, z);
    csr_eval(c, c_rowptr, c_cols, c_vals, num_constraints, z);
    for (size_t j = 0; j < num_inputs; j++) a[num_constraints + j] = z[j];
    fr g; fr_from_u64(&g, 7);
    fr *arr[3] = {a, b, c};
    for (int k = 0; k < 3; k++) { fr_ntt(arr[k], logn, 1); fr_coset_scale(arr[k], D, &g); fr_ntt(arr[k], logn, 0); }
    /* 1 / Z(g) = 1 / (g^D - 1) */
    fr gd = g; for (int k = 0; k < logn; k++) fr_mul(&gd, &gd, &gd);
    fr zi; fr_sub(&zi, &gd, &FR_ONE_M); fr_inv(&zi, &zi);
    for (size_t i = 0; i < D; i++) { fr t; fr_mul(&t, &a[i], &b[i]); fr_sub(&t, &t, &c[i]); fr_mul(&a[i], &t, &zi); }
    fr_ntt(a, logn, 1);
    fr gi; fr_inv(&gi, &g); fr_coset_scale(a, D, &gi);
    uint64_t one[4] = {1, 0, 0, 0};
    for (size_t i = 0; i < D; i++) fr_mont_mul(out_h + 4 * i, a[i].l, one);
    free(z); free(a); free(b); free(c);
    return logn;
}

/* The same witness map on `threads` host threads (the reference's ark-poly runs its transforms on rayon under the `parallel` feature,
 * legogroth16/Cargo.toml; r1cs_to_qap.rs:165-185 evaluates the rows with cfg_iter!): rows of the three matrices, the twiddle table, every
 * butterfly stage and every pointwise pass are split over the threads.  Used by bench.py's CPU legs (secondary.cpu.witness_map / .prove);
 * tests/test_oracle_golden.py checks it against the one-thread function above. */
typedef struct { int kind, tid, T, logn, s; size_t n; fr *a, *b, *c; const fr *tw, *z; fr g, k; const uint64_t *rp, *vals; const uint32_t *cols; uint64_t *out; } wm_job;
static void wm_range(const wm_job *J, size_t total, size_t *lo, size_t *hi) { *lo = total * (size_t)J->tid / (size_t)J->T; *hi = total * (size_t)(J->tid + 1) / (size_t)J->T; }
static void fr_pow_u64(fr *r, const fr *a, uint64_t e) { uint64_t ee[1] = {e}; fr_pow(r, a, ee, 1); }
static void *wm_worker(void *arg) {
    const wm_job *J = (const wm_job *)arg; size_t lo, hi;
    switch (J->kind) {
    case 0: {   /* twiddle table tw[i] = w^i, i < n/2 (J->g = w) */
        wm_range(J, J->n / 2, &lo, &hi); if (lo >= hi) break;
        fr p; fr_pow_u64(&p, &J->g, lo); fr *tw = (fr *)J->tw;
        for (size_t i = lo; i < hi; i++) { tw[i] = p; fr_mul(&p, &p, &J->g); }
    } break;
    case 1: {   /* bit reversal (each swap is done by the thread that owns the smaller index) */
        wm_range(J, J->n, &lo, &hi);
        for (size_t i = lo; i < hi; i++) { size_t j = 0; for (int b = 0; b < J->logn; b++) if (i >> b & 1) j |= (size_t)1 << (J->logn - 1 - b); if (i < j) { fr t = J->a[i]; J->a[i] = J->a[j]; J->a[j] = t; } }
    } break;
    case 2: {   /* butterflies of stage s: butterfly q = (block q / half, j = q % half) */
        const size_t half = (size_t)1 << (J->s - 1), stride = J->n >> J->s;
        wm_range(J, J->n / 2, &lo, &hi);
        for (size_t q = lo; q < hi; q++) {
            const size_t j = q & (half - 1), k = (q >> (J->s - 1)) << J->s;
            fr t, u = J->a[k + j]; fr_mul(&t, &J->tw[j * stride], &J->a[k + j + half]); fr_add(&J->a[k + j], &u, &t); fr_sub(&J->a[k + j + half], &u, &t);
        }
    } break;
    case 3: {   /* a[i] *= k * g^i */
        wm_range(J, J->n, &lo, &hi); if (lo >= hi) break;
        fr p; fr_pow_u64(&p, &J->g, lo); fr_mul(&p, &p, &J->k);
        for (size_t i = lo; i < hi; i++) { fr_mul(&J->a[i], &J->a[i], &p); fr_mul(&p, &p, &J->g); }
    } break;
    case 4: {   /* a = (a b - c) k */
        wm_range(J, J->n, &lo, &hi);
        for (size_t i = lo; i < hi; i++) { fr t; fr_mul(&t, &J->a[i], &J->b[i]); fr_sub(&t, &t, &J->c[i]); fr_mul(&J->a[i], &t, &J->k); }
    } break;
    case 5: {   /* rows of one CSR matrix */
        wm_range(J, J->n, &lo, &hi);
        for (size_t i = lo; i < hi; i++) {
            fr acc; memset(&acc, 0, sizeof acc);
            for (uint64_t k = J->rp[i]; k < J->rp[i + 1]; k++) { fr c, t; fr_mont_mul(c.l, J->vals + 4 * k, FR_R2); fr_mul(&t, &c, &J->z[J->cols[k]]); fr_add(&acc, &acc, &t); }
            J->a[i] = acc;
        }
    } break;
    case 6: {   /* canonical -> Montgomery (out = NULL) or Montgomery -> canonical */
        wm_range(J, J->n, &lo, &hi);
        const uint64_t one[4] = {1, 0, 0, 0};
        for (size_t i = lo; i < hi; i++) { if (J->out) fr_mont_mul(J->out + 4 * i, J->a[i].l, one); else fr_mont_mul(J->a[i].l, J->vals + 4 * i, FR_R2); }
    } break;
    }
    return NULL;
}
static void wm_run(wm_job proto, int T) {
    pthread_t th[64]; wm_job jobs[64];
    for (int t = 0; t < T; t++) { jobs[t] = proto; jobs[t].tid = t; jobs[t].T = T; }
    for (int t = 1; t < T; t++) pthread_create(&th[t], NULL, wm_worker, &jobs[t]);
    wm_worker(&jobs[0]);
    for (int t = 1; t < T; t++) pthread_join(th[t], NULL);
}
/* in-place transform on T threads with the twiddle table of w (forward) or 1/w (inverse); the inverse's 1/n is folded into the caller's scaling */
static void fr_ntt_mt(fr *a, int logn, const fr *tw, int T) {
    wm_job J; memset(&J, 0, sizeof J); J.n = (size_t)1 << logn; J.logn = logn; J.a = a; J.tw = tw;
    J.kind = 1; wm_run(J, T);
    J.kind = 2; for (int s = 1; s <= logn; s++) { J.s = s; wm_run(J, T); }
}
API int orc_witness_map_mt(const uint64_t *a_rowptr, const uint32_t *a_cols, const uint64_t *a_vals,
                           const uint64_t *b_rowptr, const uint32_t *b_cols, const uint64_t *b_vals,
                           const uint64_t *c_rowptr, const uint32_t *c_cols, const uint64_t *c_vals,
                           const uint64_t *z_canon, size_t num_vars, size_t num_inputs, size_t num_constraints, int threads, uint64_t *out_h) {
    int T = threads < 1 ? 1 : threads > 64 ? 64 : threads;
    int logn = 0; while (((size_t)1 << logn) < num_constraints + num_inputs) logn++;
    size_t D = (size_t)1 << logn;
    fr *z = (fr *)malloc(sizeof(fr) * num_vars), *a = (fr *)calloc(D, sizeof(fr)), *b = (fr *)calloc(D, sizeof(fr)), *c = (fr *)calloc(D, sizeof(fr));
    fr *twf = (fr *)malloc(sizeof(fr) * (D / 2 + 1)), *twi = (fr *)malloc(sizeof(fr) * (D / 2 + 1));
    wm_job J; memset(&J, 0, sizeof J);
    J.kind = 6; J.n = num_vars; J.a = z; J.vals = z_canon; J.out = NULL; wm_run(J, T);
    const uint64_t *rps[3] = {a_rowptr, b_rowptr, c_rowptr}, *vls[3] = {a_vals, b_vals, c_vals}; const uint32_t *cls[3] = {a_cols, b_cols, c_cols};
    fr *arr[3] = {a, b, c};
    for (int k = 0; k < 3; k++) { memset(&J, 0, sizeof J); J.kind = 5; J.n = num_constraints; J.a = arr[k]; J.rp = rps[k]; J.cols = cls[k]; J.vals = vls[k]; J.z = z; wm_run(J, T); }
    for (size_t j = 0; j < num_inputs; j++) a[num_constraints + j] = z[j];
    fr w, wi, g, gi, ninv; fr_root_of_unity(&w, logn); fr_inv(&wi, &w); fr_from_u64(&g, 7); fr_inv(&gi, &g); fr_from_u64(&ninv, D); fr_inv(&ninv, &ninv);
    memset(&J, 0, sizeof J); J.kind = 0; J.n = D; J.tw = twf; J.g = w; wm_run(J, T);
    J.tw = twi; J.g = wi; wm_run(J, T);
    for (int k = 0; k < 3; k++) {
        fr_ntt_mt(arr[k], logn, twi, T);                                                                        /* ifft (1/n below) */
        memset(&J, 0, sizeof J); J.kind = 3; J.n = D; J.a = arr[k]; J.g = g; J.k = ninv; wm_run(J, T);          /* coset shift, x 1/n */
        fr_ntt_mt(arr[k], logn, twf, T);                                                                        /* coset fft */
    }
    fr gd = g; for (int k = 0; k < logn; k++) fr_mul(&gd, &gd, &gd);
    fr zi; fr_sub(&zi, &gd, &FR_ONE_M); fr_inv(&zi, &zi);                                                        /* 1 / Z(g) = 1 / (g^D - 1) */
    memset(&J, 0, sizeof J); J.kind = 4; J.n = D; J.a = a; J.b = b; J.c = c; J.k = zi; wm_run(J, T);
    fr_ntt_mt(a, logn, twi, T);
    memset(&J, 0, sizeof J); J.kind = 3; J.n = D; J.a = a; J.g = gi; J.k = ninv; wm_run(J, T);                  /* coset ifft */
    memset(&J, 0, sizeof J); J.kind = 6; J.n = D; J.a = a; J.out = out_h; wm_run(J, T);
    free(z); free(a); free(b); free(c); free(twf); free(twi);
    return logn;
}

/* ---- pairings ---- */
API void orc_g2_prepare(const uint64_t q[24], uint64_t *out /* 68*36 u64 */) { g2_prepare((const g2_aff *)q, (ell_coeff *)out); }
API void orc_fp12_mul(const uint64_t a[72], const uint64_t b[72], uint64_t out[72]) { fp12 r; fp12_mul(&r, (const fp12 *)a, (const fp12 *)b); memcpy(out, &r, sizeof r); }
API void orc_fp12_one(uint64_t out[72]) { fp12_one((fp12 *)out); }
API void orc_fp12_pow(const uint64_t a[72], const uint64_t *e, int nl, uint64_t out[72]) {
    fp12 acc, base = *(const fp12 *)a; fp12_one(&acc);
    for (int i = 0; i < nl * 64; i++) { if ((e[i / 64] >> (i % 64)) & 1) fp12_mul(&acc, &acc, &base); fp12_sqr(&base, &base); }
    memcpy(out, &acc, sizeof acc);
}
/* E::multi_miller_loop: skip[i] != 0 marks a pair with an identity member (filtered like arkworks does) */
API void orc_multi_miller_loop(const uint64_t *p, const uint64_t *q, const uint8_t *skip, size_t n, int threads, uint64_t out[72]) {
    pthread_once(&ml_once, ml_init);
    g1_aff *ps = (g1_aff *)malloc(sizeof(g1_aff) * (n + 1));
    ell_coeff *co = (ell_coeff *)malloc(sizeof(ell_coeff) * N_COEFF * (n + 1));
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        if (skip && skip[i]) continue;
        ps[m] = *(const g1_aff *)(p + 12 * i);
        g2_prepare((const g2_aff *)(q + 24 * i), co + m * N_COEFF);
        m++;
    }
    int nch = (int)((m + 3) / 4);
    fp12 *partial = (fp12 *)malloc(sizeof(fp12) * (nch + 1));
    volatile int next = 0;
    ml_job J = {ps, co, m, partial, &next, nch};
    if (threads <= 1) ml_worker(&J);
    else {
        pthread_t th[64]; if (threads > 64) threads = 64; if (threads > nch) threads = nch > 0 ? nch : 1;
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, ml_worker, &J);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    }
    fp12 f; fp12_one(&f);
    for (int i = 0; i < nch; i++) fp12_mul(&f, &f, &partial[i]);
    fp12_conj(&f, &f);   /* x < 0 */
    memcpy(out, &f, sizeof f);
    free(partial); free(co); free(ps);
}
/* The verifier's call (legogroth16/src/verifier.rs:69-76): proof.b enters as an affine point (prepared inside the call), -delta and -gamma are
 * G2Prepared already (PreparedVerifyingKey, verifier.rs:22-23).  coeffs: n_prep x 68 x 3 Fp2 as orc_g2_prepare writes them. */
API void orc_multi_miller_loop_mixed(const uint64_t *p_aff, const uint64_t *q_aff, size_t n_aff, const uint64_t *p_prep, const uint64_t *coeffs, size_t n_prep,
                                     int threads, uint64_t out[72]) {
    pthread_once(&ml_once, ml_init);
    size_t m = n_aff + n_prep;
    g1_aff *ps = (g1_aff *)malloc(sizeof(g1_aff) * (m + 1));
    ell_coeff *co = (ell_coeff *)malloc(sizeof(ell_coeff) * N_COEFF * (m + 1));
    for (size_t i = 0; i < n_aff; i++) { ps[i] = *(const g1_aff *)(p_aff + 12 * i); g2_prepare((const g2_aff *)(q_aff + 24 * i), co + i * N_COEFF); }
    for (size_t i = 0; i < n_prep; i++) { ps[n_aff + i] = *(const g1_aff *)(p_prep + 12 * i); memcpy(co + (n_aff + i) * N_COEFF, coeffs + i * N_COEFF * 36, sizeof(ell_coeff) * N_COEFF); }
    int nch = (int)((m + 3) / 4);
    fp12 *partial = (fp12 *)malloc(sizeof(fp12) * (nch + 1));
    volatile int next = 0;
    ml_job J = {ps, co, m, partial, &next, nch};
    if (threads <= 1 || nch <= 1) ml_worker(&J);
    else {
        pthread_t th[64]; if (threads > 64) threads = 64; if (threads > nch) threads = nch;
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, ml_worker, &J);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    }
    fp12 f; fp12_one(&f);
    for (int i = 0; i < nch; i++) fp12_mul(&f, &f, &partial[i]);
    fp12_conj(&f, &f);
    memcpy(out, &f, sizeof f);
    free(partial); free(co); free(ps);
}
/* ---- batches on the host's cores: the per-equation work of RandomizedPairingChecker (utils/src/randomized_pairing_check.rs:116-214) as the reference
 * spreads it with rayon — `cfg_iter!(a).map(|a| a.mul_bigint(m))` scalings (:125-129,152-158), `G2Prepared::from` of every b (:132,163), the targets'
 * `out.mul_bigint(m)` (:136) — used by bench.py's CPU legs beside the batched verifier and the aggregation (test infrastructure, like everything here). ---- */
typedef struct { int kind; size_t n; const uint64_t *a; const uint8_t *f; const uint64_t *b; size_t bstride; const uint8_t *neg; uint64_t *out; uint8_t *outf; fp12 *acc; volatile long *next; } bt_job;
static void *bt_worker(void *arg) {
    bt_job *J = (bt_job *)arg;
    fp12 prod; fp12_one(&prod);
    for (;;) {
        long i = __sync_fetch_and_add(J->next, 1);
        if (i >= (long)J->n) break;
        if (J->kind == 0) {                                   /* out_i = (+-) s_i P_i, affine (mul_bigint, then into_affine: one inversion per point like G1Prepared::from(projective)) */
            g1_jac r; g1_aff o;
            const int inf = J->f && J->f[i];
            g1_mul(&r, (const g1_aff *)(J->a + 12 * i), inf, J->b + J->bstride * i);
            const int oinf = g1_to_affine(&o, &r);
            if (!oinf && J->neg && J->neg[i]) fp_neg(&o.y, &o.y);
            if (oinf) memset(J->out + 12 * i, 0, 96); else memcpy(J->out + 12 * i, &o, 96);
            J->outf[i] = (uint8_t)oinf;
        } else if (J->kind == 1) {                            /* G2Prepared::from(q_i) */
            g2_prepare((const g2_aff *)(J->a + 24 * i), (ell_coeff *)(J->out + (size_t)N_COEFF * 36 * i));
        } else {                                              /* prod *= a_i ^ e_i (255-bit square-and-multiply, PairingOutput::mul_bigint) */
            fp12 acc, base = *(const fp12 *)(J->a + (J->bstride ? 72 * i : 0)); fp12_one(&acc);
            const uint64_t *e = J->b + 4 * i;
            int top = 255; while (top >= 0 && !((e[top / 64] >> (top % 64)) & 1)) top--;
            for (int k = top; k >= 0; k--) { fp12_sqr(&acc, &acc); if ((e[k / 64] >> (k % 64)) & 1) fp12_mul(&acc, &acc, &base); }
            fp12_mul(&prod, &prod, &acc);
        }
    }
    if (J->kind == 2) { fp12 *slot = J->acc + __sync_fetch_and_add(J->next + 1, 1); *slot = prod; }
    return NULL;
}
static void bt_run(bt_job *J, int threads) {
    if (threads > 64) threads = 64; if (threads < 1) threads = 1; if ((size_t)threads > J->n) threads = J->n ? (int)J->n : 1;
    if (threads == 1) { bt_worker(J); return; }
    pthread_t th[64];
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, bt_worker, J);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}
API void orc_g1_scale_batch(const uint64_t *pts, const uint8_t *inf, const uint64_t *scalars, size_t scalar_stride, const uint8_t *negate, size_t n, int threads, uint64_t *out, uint8_t *out_inf) {
    volatile long next[2] = {0, 0};
    bt_job J = {0, n, pts, inf, scalars, scalar_stride, negate, out, out_inf, NULL, next};
    bt_run(&J, threads);
}
API void orc_g2_prepare_batch(const uint64_t *q, size_t n, int threads, uint64_t *out) {
    pthread_once(&ml_once, ml_init);
    volatile long next[2] = {0, 0};
    bt_job J = {1, n, q, NULL, NULL, 0, NULL, out, NULL, NULL, next};
    bt_run(&J, threads);
}
/* prod_i a_i ^ e_i; a_stride = 0: the same base for every exponent */
API void orc_fp12_multi_pow(const uint64_t *a, size_t a_stride, const uint64_t *e, size_t n, int threads, uint64_t out[72]) {
    pthread_once(&frob_once, frob_init);
    volatile long next[2] = {0, 0};
    fp12 parts[64];
    if (threads > 64) threads = 64; if (threads < 1) threads = 1; if ((size_t)threads > n) threads = n ? (int)n : 1;
    bt_job J = {2, n, a, NULL, e, a_stride, NULL, NULL, NULL, parts, next};
    bt_run(&J, threads);
    fp12 f; fp12_one(&f);
    for (long k = 0; k < next[1]; k++) fp12_mul(&f, &f, &parts[k]);
    memcpy(out, &f, sizeof f);
}
/* E::final_exponentiation (A.4 chain).  Returns 0 on success, -1 where arkworks returns None (f == 0). */
API int orc_final_exponentiation(const uint64_t in[72], uint64_t out[72]) {
    const fp12 *f = (const fp12 *)in;
    if (fp12_is_zero(f)) return -1;
    fp12 f1, f2, r, y0, y1, y2;
    fp12_conj(&f1, f); fp12_inv(&f2, f); fp12_mul(&r, &f1, &f2); f2 = r;
    fp12_frob2(&r, &r); fp12_mul(&r, &r, &f2);
    fp12_sqr(&y0, &r);
    cyclo_exp_x(&y1, &r);
    fp12_conj(&y2, &r);
    fp12_mul(&y1, &y1, &y2);
    cyclo_exp_x(&y2, &y1);
    fp12_conj(&y1, &y1);
    fp12_mul(&y1, &y1, &y2);
    cyclo_exp_x(&y2, &y1);
    fp12_frob1(&y1, &y1);
    fp12_mul(&y1, &y1, &y2);
    fp12_mul(&r, &r, &y0);
    cyclo_exp_x(&y0, &y1);
    cyclo_exp_x(&y2, &y0);
    fp12_frob2(&y0, &y1);
    fp12_conj(&y1, &y1);
    fp12_mul(&y1, &y1, &y2);
    fp12_mul(&y1, &y1, &y0);
    fp12_mul(&r, &r, &y1);
    memcpy(out, &r, sizeof r);
    return 0;
}
