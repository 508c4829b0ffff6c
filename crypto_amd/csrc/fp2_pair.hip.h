// crypto_amd/csrc/fp2_pair.hip.h — Fp2 arithmetic spread over a LANE PAIR (device only).
//
// An XYZZ accumulator over Fp2 is 112 registers and the fully inlined mixed addition needs > 256 VGPRs, so the
// one-lane-per-point G2 kernel runs at one wave per SIMD with spills (25 ms for n = 2^20, 7x G1).  Here lanes 2k and
// 2k+1 share one point: the even lane holds every c0 component, the odd lane every c1 component.  Each lane then has
// exactly the register footprint of the G1 kernel (2 waves/SIMD, no spills).  Cross terms travel over DPP quad_perm
// (full-rate VALU, no LDS).  A product costs 2 base-field products per lane (schoolbook, 4 in total instead of
// Karatsuba's 3), a square 1 per lane:
//     (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u
//     (a0 + a1 u)^2          = (a0 + a1)(a0 - a1) + 2 a0 a1 u
// Both lanes of a pair always follow the same control flow (same terms, same run boundaries).
#pragma once
#include "fp29.hip.h"
#include "ec29.hip.h"

namespace bls29 {

struct Fp2H { Fp v; };   // this lane's half of an Fp2 element

__device__ __forceinline__ bool pair_odd() { return (threadIdx.x & 1u) != 0; }
__device__ __forceinline__ uint32_t xchg32(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}
__device__ __forceinline__ void xchg(Fp &r, const Fp &a) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = xchg32(a.l[i]);
}
__device__ __forceinline__ void sel(Fp &r, bool c, const Fp &a, const Fp &b) {   // r = c ? a : b
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
}

__device__ __forceinline__ void fzero(Fp2H &r) { fp_zero(r.v); }
__device__ __forceinline__ void fset_one(Fp2H &r) { Fp one, z; fp_set_one(one); fp_zero(z); sel(r.v, pair_odd(), z, one); }
__device__ __forceinline__ void fadd(Fp2H &r, const Fp2H &a, const Fp2H &b) { fp_add(r.v, a.v, b.v); }
__device__ __forceinline__ void fdbl(Fp2H &r, const Fp2H &a) { fp_add(r.v, a.v, a.v); }
template <int M> __device__ __forceinline__ void fsub(Fp2H &r, const Fp2H &a, const Fp2H &b) { fp_sub<M>(r.v, a.v, b.v); }
__device__ __forceinline__ void fnorm(Fp2H &r, const Fp2H &a) { fp_norm(r.v, a.v); }
__device__ __forceinline__ bool fmaybe_zero(const Fp2H &a) {
    uint32_t z = fp_maybe_zero(a.v) ? 1u : 0u;
    return (z & xchg32(z)) != 0;
}
__device__ __forceinline__ bool fis_zero_exact(const Fp2H &a) {
    uint32_t z = fp_is_zero_exact(a.v) ? 1u : 0u;
    return (z & xchg32(z)) != 0;
}
// product: the even lane computes c0 = a0 b0 + (-a1) b1, the odd lane c1 = a1 b0 + a0 b1 — one fused two-product reduction
// (fp_mul2, 588 mads) per lane after exchanging both operands' halves.  Inputs class N (value < 500 p); output value < 2 p.
__device__ __forceinline__ void fmul(Fp2H &r, const Fp2H &a, const Fp2H &b) {
    const bool odd = pair_odd();
    Fp ao, bo, z, nao, X, Y, Z;
    xchg(ao, a.v); xchg(bo, b.v);
    fp_zero(z); fp_sub<512>(nao, z, ao); fp_norm(nao, nao);     // even lane: -a1
    sel(X, odd, bo, b.v);
    sel(Y, odd, ao, nao);
    sel(Z, odd, b.v, bo);
    fp_mul2(r.v, a.v, X, Y, Z);
}
// square: input class N with value < 60 p
__device__ __forceinline__ void fsqr(Fp2H &r, const Fp2H &a) {
    const bool odd = pair_odd();
    Fp ao, s, d, m1, m2, p, p2;
    xchg(ao, a.v);
    fp_add(s, a.v, ao);                       // a0 + a1 (both lanes; used by the even lane)
    fp_sub<64>(d, a.v, ao); fp_norm(d, d);    // even: a0 - a1
    sel(m1, odd, a.v, s);
    sel(m2, odd, ao, d);
    fp_mul(p, m1, m2);                        // even: (a0+a1)(a0-a1)   odd: a1 a0
    fp_add(p2, p, p); fp_norm(p2, p2);
    sel(r.v, odd, p2, p);
}

template <int M> __device__ __forceinline__ void fmul_sub(Fp2H &r, const Fp2H &a, const Fp2H &b, const Fp2H &c, const Fp2H &d) {
    Fp2H t, u; fmul(t, a, b); fmul(u, c, d); fsub<8>(t, t, u); fnorm(r, t);
}
// ---- helpers used by the Miller-loop line functions (pairing29.hip.h), lane-pair versions ----
template <int M> __device__ __forceinline__ void f2_sqr_m(Fp2H &r, const Fp2H &a) {
    const bool odd = pair_odd();
    Fp ao, s, d, m1, m2, p, p2;
    xchg(ao, a.v);
    fp_add(s, a.v, ao);
    fp_sub<M>(d, a.v, ao); fp_norm(d, d);
    sel(m1, odd, a.v, s);
    sel(m2, odd, ao, d);
    fp_mul(p, m1, m2);
    fp_add(p2, p, p); fp_norm(p2, p2);
    sel(r.v, odd, p2, p);
}
// r = a (1 + u) = (a0 - a1) + (a0 + a1) u, normalised
template <int M> __device__ __forceinline__ void f2_mul_xi_n(Fp2H &r, const Fp2H &a) {
    const bool odd = pair_odd();
    Fp ao, d, s, t;
    xchg(ao, a.v);
    fp_sub<M>(d, a.v, ao);       // even lane: a0 - a1 + M p
    fp_add(s, a.v, ao);          // odd lane:  a1 + a0
    sel(t, odd, s, d);
    fp_norm(r.v, t);
}
template <int M> __device__ __forceinline__ void f2_neg_n(Fp2H &r, const Fp2H &a) { Fp z; fp_zero(z); fp_sub<M>(r.v, z, a.v); fp_norm(r.v, r.v); }
__device__ __forceinline__ void f2_add_n(Fp2H &r, const Fp2H &a, const Fp2H &b) { fp_add(r.v, a.v, b.v); fp_norm(r.v, r.v); }
template <int M> __device__ __forceinline__ void f2_sub_n(Fp2H &r, const Fp2H &a, const Fp2H &b) { fp_sub<M>(r.v, a.v, b.v); fp_norm(r.v, r.v); }
__device__ __forceinline__ void fmul_fp(Fp2H &r, const Fp2H &a, const Fp &k) { fp_mul(r.v, a.v, k); }
__device__ __forceinline__ void fsel(Fp2H &r, bool c, const Fp2H &a, const Fp2H &b) { sel(r.v, c, a.v, b.v); }      // r = c ? a : b (pairing29.hip.h's role functions)

template <> struct SubM<Fp2H> {     // same value budgets as the one-lane Fp2 formulas
    static constexpr int P = 32;
    static constexpr int R = 16;
    static constexpr int X = 16;
    static constexpr int D = 32;
    static constexpr int Y = 8;
    static constexpr int YN = 16;
    static constexpr int NEG = 4;
};

}  // namespace bls29
