// crypto_amd/csrc/fs2_pair.hip.h — Fp2 = Fp[u]/(u^2 + 1) over the 13 x 30-bit signed base field (fp30s.hip.h), for the G2 MSM kernels:
//   Fs2   one lane per element (base preparation and the table construction, where a lane owns whole points), and
//   Fs2H  one element per LANE PAIR (accumulation, fix-up, bucket reduction): the even lane holds every c0 component, the odd lane every c1
//         component, cross terms travel over DPP quad_perm — the layout of fp2_pair.hip.h, which stays the pairing kernels' field.
// A product is two fused two-product reductions (fs_mul2: 507 multiply-adds each, 588 over the 14 x 29-bit field), a square two products.
// Signed digits make the formulas shorter than their fp2_pair.hip.h counterparts: -a1 is 13 negations (no multiple of p, no carry pass).
#pragma once
#include "fp30s.hip.h"
#include "ec29.hip.h"

namespace bls29 {

// ---- one lane per element ----
struct Fs2 { Fs c0, c1; };
FD void fzero(Fs2 &r) { fs_zero(r.c0); fs_zero(r.c1); }
FD void fset_one(Fs2 &r) { fs_set_one(r.c0); fs_zero(r.c1); }
FD void fadd(Fs2 &r, const Fs2 &a, const Fs2 &b) { fs_add(r.c0, a.c0, b.c0); fs_add(r.c1, a.c1, b.c1); }
FD void fdbl(Fs2 &r, const Fs2 &a) { fs_add(r.c0, a.c0, a.c0); fs_add(r.c1, a.c1, a.c1); }
template <int M> FD void fsub(Fs2 &r, const Fs2 &a, const Fs2 &b) { fs_sub(r.c0, a.c0, b.c0); fs_sub(r.c1, a.c1, b.c1); }
FD void fnorm(Fs2 &r, const Fs2 &a) { fs_bal(r.c0, a.c0); fs_bal(r.c1, a.c1); }
FD void fnormw(Fs2 &r, const Fs2 &a) { fs_bal_wide(r.c0, a.c0); fs_bal_wide(r.c1, a.c1); }
FD bool fmaybe_zero(const Fs2 &a) { return fs_maybe_zero(a.c0) && fs_maybe_zero(a.c1); }
FD bool fis_zero_exact(const Fs2 &a) { return fs_is_zero_exact(a.c0) && fs_is_zero_exact(a.c1); }
// (Measurement only, -DFS2_KARATSUBA, tools/ubench/g2_madd_rate.hip.)  Karatsuba form of the one-lane product: the three operand products a0 b0, a1 b1, (a0 + a1)(b0 + b1) are formed column by column and
// combined BEFORE the two reductions (c0 columns = S - T, c1 columns = U - S - T): 3 * 169 + 2 * 169 = 845 multiply-adds against 1014.
// U alone may pass 2^63 (operand sums have 31-bit digits); the accumulators are unsigned so that it wraps, and what is shifted / reduced is
// U - S - T = the cross-term column, bounded exactly as in fs_mul2.  Measured on MI355X: the one-lane mixed addition built on it runs at
// 1.0 G additions/s (five live 64-bit column chains on top of a 104-register accumulator: 170-670 spilled registers), the one-lane form on
// fs_mul2 at 2.3 G/s, the lane-pair form (Fs2H below) at 2.5 G/s — which is why the accumulation keeps lane pairs and 1014 multiply-adds.
FD void fs2_mul_kara(Fs &r0, Fs &r1, const Fs &a0, const Fs &a1, const Fs &b0, const Fs &b1) {
    constexpr int32_t P_[SN] = BLS30_P;
    SCHK({ Fs n1; fs_neg(n1, a1); const Fs *pa[2] = {&a0, &n1}, *pb[2] = {&b0, &b1}; schk_columns(pa, pb, 2);
           const Fs *qa[2] = {&a0, &a1}, *qb[2] = {&b1, &b0}; schk_columns(qa, qb, 2); })
    int32_t sa[SN], sb[SN], m0[SN], m1[SN], t0[SN], t1[SN];
#pragma unroll
    for (int i = 0; i < SN; i++) { sa[i] = a0.l[i] + a1.l[i]; sb[i] = b0.l[i] + b1.l[i]; }
    uint64_t acc0 = 0, acc1 = 0;
#define MA(acc, x, y) acc += (uint64_t)((int64_t)(x) * (int64_t)(y))
#pragma unroll
    for (int k = 0; k < SN; k++) {
        uint64_t S = 0, T = 0, U = 0, R0 = 0, R1 = 0;
#pragma unroll
        for (int i = 0; i <= k; i++) { MA(S, a0.l[i], b0.l[k - i]); MA(T, a1.l[i], b1.l[k - i]); MA(U, sa[i], sb[k - i]); }
#pragma unroll
        for (int i = 0; i < k; i++) { MA(R0, m0[i], P_[k - i]); MA(R1, m1[i], P_[k - i]); }
        acc0 += S - T + R0; acc1 += U - S - T + R1;
        m0[k] = sext30((uint32_t)acc0 * SINV30); MA(acc0, m0[k], P_[0]); acc0 = (uint64_t)((int64_t)acc0 >> SB);
        m1[k] = sext30((uint32_t)acc1 * SINV30); MA(acc1, m1[k], P_[0]); acc1 = (uint64_t)((int64_t)acc1 >> SB);
    }
#pragma unroll
    for (int k = SN; k < 2 * SN - 1; k++) {
        uint64_t S = 0, T = 0, U = 0, R0 = (uint64_t)SHALF, R1 = (uint64_t)SHALF;
#pragma unroll
        for (int i = k - SN + 1; i < SN; i++) { MA(S, a0.l[i], b0.l[k - i]); MA(T, a1.l[i], b1.l[k - i]); MA(U, sa[i], sb[k - i]); }
#pragma unroll
        for (int i = k - SN + 1; i < SN; i++) { MA(R0, m0[i], P_[k - i]); MA(R1, m1[i], P_[k - i]); }
        acc0 += S - T + R0; acc1 += U - S - T + R1;
        t0[k - SN] = (int32_t)((uint32_t)acc0 & SMASK) - SHALF; acc0 = (uint64_t)((int64_t)acc0 >> SB);
        t1[k - SN] = (int32_t)((uint32_t)acc1 & SMASK) - SHALF; acc1 = (uint64_t)((int64_t)acc1 >> SB);
    }
#undef MA
    t0[SN - 1] = (int32_t)acc0; t1[SN - 1] = (int32_t)acc1;
    SCHK(const double v0 = 0.51 + (a0.vb * b0.vb + a1.vb * b1.vb) * P_OVER_2_390, v1 = 0.51 + (a0.vb * b1.vb + a1.vb * b0.vb) * P_OVER_2_390;)
#pragma unroll
    for (int i = 0; i < SN; i++) { r0.l[i] = t0[i]; r1.l[i] = t1[i]; }
    SCHK(schk_set_B(r0, v0); schk_set_B(r1, v1); schk_actual(r0); schk_actual(r1);)
}
// (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u; all components class B
FD void fmul(Fs2 &r, const Fs2 &a, const Fs2 &b) {
#ifdef FS2_KARATSUBA
    fs2_mul_kara(r.c0, r.c1, a.c0, a.c1, b.c0, b.c1);
#else
    Fs n1, c0, c1;
    fs_neg(n1, a.c1);
    fs_mul2(c0, a.c0, b.c0, n1, b.c1);
    fs_mul2(c1, a.c0, b.c1, a.c1, b.c0);
    r.c0 = c0; r.c1 = c1;
#endif
}
// (a0 + a1)(a0 - a1), 2 a0 a1; a class B
FD void fsqr(Fs2 &r, const Fs2 &a) {
    Fs s, d, db, a2, c0;
    fs_add(s, a.c0, a.c1);
    fs_sub(d, a.c0, a.c1); fs_bal(db, d);
    fs_add(a2, a.c0, a.c0);
    fs_mul(c0, s, db);                        // (class D) x (class B)
    fs_mul(r.c1, a2, a.c1);                   // 2 a0 a1 straight out of the product: exactly balanced digits, no carry pass
    r.c0 = c0;
}
template <int M> FD void fmul_sub(Fs2 &r, const Fs2 &a, const Fs2 &b, const Fs2 &c, const Fs2 &d) {
    Fs2 t, u; fmul(t, a, b); fmul(u, c, d); fsub<0>(t, t, u); fnorm(r, t);
}
template <> struct SubM<Fs2> { static constexpr int P = 0, R = 0, X = 0, D = 0, Y = 0, YN = 0, NEG = 0; };

#if defined(__HIPCC__)
// ---- one element per lane pair (device only) ----
struct Fs2H { Fs v; };   // this lane's half

__device__ __forceinline__ bool spair_odd() { return (threadIdx.x & 1u) != 0; }
__device__ __forceinline__ int32_t sxchg32(int32_t x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true); }
__device__ __forceinline__ void xchg(Fs &r, const Fs &a) {
#pragma unroll
    for (int i = 0; i < SN; i++) r.l[i] = sxchg32(a.l[i]);
}
__device__ __forceinline__ void sel(Fs &r, bool c, const Fs &a, const Fs &b) {   // r = c ? a : b
#pragma unroll
    for (int i = 0; i < SN; i++) r.l[i] = c ? a.l[i] : b.l[i];
}
__device__ __forceinline__ void fzero(Fs2H &r) { fs_zero(r.v); }
__device__ __forceinline__ void fcond_neg(Fs2H &y, bool neg) { fs_cond_neg(y.v, y.v, neg); }      // (both halves of -y: every lane negates its own)
__device__ __forceinline__ void fset_one(Fs2H &r) { Fs one, z; fs_set_one(one); fs_zero(z); sel(r.v, spair_odd(), z, one); }
__device__ __forceinline__ void fadd(Fs2H &r, const Fs2H &a, const Fs2H &b) { fs_add(r.v, a.v, b.v); }
__device__ __forceinline__ void fdbl(Fs2H &r, const Fs2H &a) { fs_add(r.v, a.v, a.v); }
template <int M> __device__ __forceinline__ void fsub(Fs2H &r, const Fs2H &a, const Fs2H &b) { fs_sub(r.v, a.v, b.v); }
__device__ __forceinline__ void fnorm(Fs2H &r, const Fs2H &a) { fs_bal(r.v, a.v); }
__device__ __forceinline__ void fnormw(Fs2H &r, const Fs2H &a) { fs_bal_wide(r.v, a.v); }
__device__ __forceinline__ bool fmaybe_zero(const Fs2H &a) {
    int32_t z = fs_maybe_zero(a.v) ? 1 : 0;
    return (z & sxchg32(z)) != 0;
}
__device__ __forceinline__ bool fis_zero_exact(const Fs2H &a) {
    int32_t z = fs_is_zero_exact(a.v) ? 1 : 0;
    return (z & sxchg32(z)) != 0;
}
// product: the even lane computes c0 = a0 b0 + a1 (-b1), the odd lane c1 = a1 b0 + a0 b1 — one fused two-product reduction per lane after
// exchanging both operands' halves.  All halves class B.
__device__ __forceinline__ void fmul(Fs2H &r, const Fs2H &a, const Fs2H &b) {
    const bool odd = spair_odd();
    Fs ao, bo, nbo, X, Z;
    xchg(ao, a.v); xchg(bo, b.v);
    fs_neg(nbo, bo);                          // even lane: -b1
    sel(X, odd, bo, b.v);                     // even: a0 b0 + a1 (-b1)     odd: a1 b0 + a0 b1  — the a side needs no selection
    sel(Z, odd, b.v, nbo);
    fs_mul2(r.v, a.v, X, ao, Z);
}
// TWO independent products at once, r1 = a b and r2 = c d: instead of half of each (two fused two-product reductions per lane: 2 x 507 multiply-adds) every
// lane of the pair computes ONE whole product by Karatsuba (fs2_mul_kara: 3 operand products + 2 reductions = 845) — the even lane a b, the odd lane c d —
// after the lanes have swapped the operand halves the other one needs, and the result halves are swapped back.  All halves class B.
__device__ __forceinline__ void fmul_two(Fs2H &r1, Fs2H &r2, const Fs2H &a, const Fs2H &b, const Fs2H &c, const Fs2H &d) {
    const bool odd = spair_odd();
    Fs s1, s2, g1, g2, own1, own2, X0, X1, Y0, Y1, R0, R1, back, got;
    sel(s1, odd, a.v, c.v); sel(s2, odd, b.v, d.v);               // what the partner needs: the odd lane gives its halves of a, b; the even lane its halves of c, d
    xchg(g1, s1); xchg(g2, s2);
    sel(own1, odd, c.v, a.v); sel(own2, odd, d.v, b.v);
    sel(X0, odd, g1, own1); sel(X1, odd, own1, g1);               // even: (a0, a1) ; odd: (c0, c1)
    sel(Y0, odd, g2, own2); sel(Y1, odd, own2, g2);
    fs2_mul_kara(R0, R1, X0, X1, Y0, Y1);
    sel(back, odd, R0, R1);                                       // even sends (a b)_1, odd sends (c d)_0
    xchg(got, back);
    sel(r1.v, odd, got, R0);                                      // a b: even keeps c0, odd receives c1
    sel(r2.v, odd, R1, got);                                      // c d: even receives c0, odd keeps c1
}
// square: input class B
__device__ __forceinline__ void fsqr(Fs2H &r, const Fs2H &a) {
    const bool odd = spair_odd();
    Fs ao, s, d, ao2, m1, m2;
    xchg(ao, a.v);
    fs_add(s, a.v, ao);                       // a0 + a1 (class D; used by the even lane)
    fs_sub(d, a.v, ao); fs_bal(d, d);         // even: a0 - a1
    fs_add(ao2, ao, ao);                      // odd: 2 a0
    sel(m1, odd, a.v, s);
    sel(m2, odd, ao2, d);
    fs_mul(r.v, m1, m2);                      // even: (a0 + a1)(a0 - a1)   odd: a1 (2 a0) — exactly balanced digits on both lanes
}
template <int M> __device__ __forceinline__ void fmul_sub(Fs2H &r, const Fs2H &a, const Fs2H &b, const Fs2H &c, const Fs2H &d) {
    Fs2H t, u; fmul(t, a, b); fmul(u, c, d); fsub<0>(t, t, u); fnorm(r, t);
}
template <> struct SubM<Fs2H> { static constexpr int P = 0, R = 0, X = 0, D = 0, Y = 0, YN = 0, NEG = 0; };
template <> struct MaddFormulaFirst<Fs2H> { static constexpr bool value = false; };      // (the early-return form, as for Fp2H: fewer live registers)
#endif

}  // namespace bls29
