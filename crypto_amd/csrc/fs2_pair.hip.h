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
// (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u; all components class B
FD void fmul(Fs2 &r, const Fs2 &a, const Fs2 &b) {
    Fs n1, c0, c1;
    fs_neg(n1, a.c1);
    fs_mul2(c0, a.c0, b.c0, n1, b.c1);
    fs_mul2(c1, a.c0, b.c1, a.c1, b.c0);
    r.c0 = c0; r.c1 = c1;
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
// product: the even lane computes c0 = a0 b0 + (-a1) b1, the odd lane c1 = a1 b0 + a0 b1 — one fused two-product reduction per lane after
// exchanging both operands' halves.  All halves class B.
__device__ __forceinline__ void fmul(Fs2H &r, const Fs2H &a, const Fs2H &b) {
    const bool odd = spair_odd();
    Fs ao, bo, nao, X, Y, Z;
    xchg(ao, a.v); xchg(bo, b.v);
    fs_neg(nao, ao);                          // even lane: -a1
    sel(X, odd, bo, b.v);
    sel(Y, odd, ao, nao);
    sel(Z, odd, b.v, bo);
    fs_mul2(r.v, a.v, X, Y, Z);
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
