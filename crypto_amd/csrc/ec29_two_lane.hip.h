// crypto_amd/csrc/ec29_two_lane.hip.h — G1 XYZZ doubling and mixed addition shared by TWO adjacent lanes (device only).
//
// The batched scalar-multiplication kernels (k_g1_scale: RandomizedPairingChecker's `a.mul_bigint(m)`, utils/src/randomized_pairing_check.rs:125-127;
// k_mul_add<G1>: the aggregation's folding step) run one dependent chain of 255 doublings + ~128 additions per point and have far fewer
// lanes than the chip, so they last as long as the instruction stream of one lane.  Here lanes 2k (role A) and 2k+1 (role B) both hold
// the whole point and take one field operation of every round — the SAME operation on role-selected operands, so the wave stays
// convergent — and swap results over DPP quad_perm [1,0,3,2]:
// (and, with F = Fp2H, the two lane pairs of a quad share a G2 point the same way: Share4)
//   doubling (dbl-2008-s-1)          A                    B              mixed addition (madd-2008-s)   A                 B
//     round 1 (square)               V = U^2              X^2              round 1 (product)            U2 = X2 ZZ        S2 = Y2 ZZZ
//     round 2 (product)              W = U V              S = X V          round 2 (square)             PP = P^2          R^2
//     round 3 (square)               M^2                  -                round 3 (product)            PPP = P PP        Q = X1 PP
//     round 4 (product)              M (S - X3)           W Y              round 4 (product)            R (Q - X3)        Y1 PPP
//     round 5 (product)              ZZ' = V ZZ           ZZZ' = W ZZZ     round 5 (product)            ZZ' = ZZ PP       ZZZ' = ZZZ PPP
// 5 field operations per lane instead of 9 / 10.  Same formulas as ec29.hip.h (the group element is what is compared anyway).
#pragma once
#include "fp29.hip.h"
#include "ec29.hip.h"
#include "fp2_pair.hip.h"      // xchg, sel, pair_odd

namespace bls29 {

// Sharing policies: who the two cooperating parties are and how they swap / select a field value.
struct Share2 {       // two adjacent lanes share a G1 point (F = Fp): roles by lane parity, swap = DPP quad_perm [1,0,3,2]
    static __device__ __forceinline__ bool role() { return pair_odd(); }
    static __device__ __forceinline__ void swap(Fp &r, const Fp &a) { xchg(r, a); }
    static __device__ __forceinline__ void pick(Fp &r, bool b, const Fp &if_b, const Fp &if_a) { sel(r, b, if_b, if_a); }
};
struct Share4 {       // two lane pairs of a quad share a G2 point (F = Fp2H, halves on the lanes of a pair): roles by bit 1, swap = quad_perm [2,3,0,1]
    static __device__ __forceinline__ bool role() { return (threadIdx.x & 2u) != 0; }
    static __device__ __forceinline__ void swap(Fp2H &r, const Fp2H &a) {
#pragma unroll
        for (int i = 0; i < NL; i++) r.v.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.v.l[i], 0x4E, 0xF, 0xF, true);
    }
    static __device__ __forceinline__ void pick(Fp2H &r, bool b, const Fp2H &if_b, const Fp2H &if_a) { sel(r.v, b, if_b.v, if_a.v); }
};

// r = 2 a, a not the identity; both parties hold a and receive r
template <class F, class S> __device__ __forceinline__ void xyzz_dbl_shared(Xyzz<F> &r, const Xyzz<F> &a) {
    const bool B = S::role();
    F U, in, res, oth, V, X2, M, W, Sx, t, X3, u, v, Ya, Yb, Y3, ZZ3, ZZZ3;
    fadd(U, a.y, a.y); fnorm(U, U);
    S::pick(in, B, a.x, U); fsqr(res, in); S::swap(oth, res);                       // round 1: A: V = U^2, B: X^2
    S::pick(V, B, oth, res); S::pick(X2, B, res, oth);
    fadd(t, X2, X2); fadd(M, t, X2); fnorm(M, M);                                   // M = 3 X^2
    S::pick(u, B, a.x, U); fmul(res, u, V); S::swap(oth, res);                      // round 2: A: W = U V, B: S = X V
    S::pick(W, B, oth, res); S::pick(Sx, B, res, oth);
    fsqr(X3, M);                                                                    // round 3 (both parties: same operand)
    fadd(t, Sx, Sx); fsub<SubM<F>::X>(X3, X3, t); fnorm(X3, X3);
    fsub<SubM<F>::D>(t, Sx, X3); fnorm(t, t);
    S::pick(u, B, W, M); S::pick(v, B, a.y, t); fmul(res, u, v); S::swap(oth, res); // round 4: A: M (S - X3), B: W Y
    S::pick(Ya, B, oth, res); S::pick(Yb, B, res, oth);
    fsub<4>(Y3, Ya, Yb); fnorm(Y3, Y3);                                             // value < 6 p: the budget SubM<F>::R assumes for Y1
    S::pick(u, B, W, V); S::pick(v, B, a.zzz, a.zz); fmul(res, u, v); S::swap(oth, res);   // round 5: A: V ZZ, B: W ZZZ
    S::pick(ZZ3, B, oth, res); S::pick(ZZZ3, B, res, oth);
    r.x = X3; r.y = Y3; r.zz = ZZ3; r.zzz = ZZZ3;
}

// acc += (neg ? -q : q), q affine and not the identity; `inf` is acc's identity flag (same contract as xyzz_madd)
template <class F, class S> __device__ __forceinline__ void xyzz_madd_shared(Xyzz<F> &acc, bool &inf, const Aff<F> &q_in, bool neg) {
    const bool B = S::role();
    Aff<F> q = q_in;
    if (neg) { F z; fzero(z); fsub<SubM<F>::NEG>(q.y, z, q.y); fnorm(q.y, q.y); }
    F u, v, res, oth, U2, S2, Pd, Rd, PP, RR, PPP, Q, t, X3, Ya, Yb, Y3, ZZ3, ZZZ3;
    S::pick(u, B, q.y, q.x); S::pick(v, B, acc.zzz, acc.zz); fmul(res, u, v); S::swap(oth, res);      // round 1: A: U2, B: S2
    S::pick(U2, B, oth, res); S::pick(S2, B, res, oth);
    fsub<SubM<F>::P>(Pd, U2, acc.x); fnorm(Pd, Pd);
    fsub<SubM<F>::R>(Rd, S2, acc.y); fnorm(Rd, Rd);
    S::pick(u, B, Rd, Pd); fsqr(res, u); S::swap(oth, res);                                           // round 2: A: PP, B: R^2
    S::pick(PP, B, oth, res); S::pick(RR, B, res, oth);
    S::pick(u, B, acc.x, Pd); fmul(res, u, PP); S::swap(oth, res);                                    // round 3: A: PPP, B: Q
    S::pick(PPP, B, oth, res); S::pick(Q, B, res, oth);
    fadd(t, Q, Q); fadd(t, t, PPP);
    fsub<SubM<F>::X>(X3, RR, t); fnorm(X3, X3);
    fsub<SubM<F>::D>(t, Q, X3); fnorm(t, t);
    S::pick(u, B, acc.y, Rd); S::pick(v, B, PPP, t); fmul(res, u, v); S::swap(oth, res);              // round 4: A: R (Q - X3), B: Y1 PPP
    S::pick(Ya, B, oth, res); S::pick(Yb, B, res, oth);
    fsub<4>(Y3, Ya, Yb); fnorm(Y3, Y3);                                                               // value < 6 p (SubM<F>::R budget)
    S::pick(u, B, acc.zzz, acc.zz); S::pick(v, B, PPP, PP); fmul(res, u, v); S::swap(oth, res);       // round 5: A: ZZ PP, B: ZZZ PPP
    S::pick(ZZ3, B, oth, res); S::pick(ZZZ3, B, res, oth);
    const bool was_inf = inf;
    const bool special = !was_inf && fmaybe_zero(Pd);
    if (was_inf) { acc.x = q.x; acc.y = q.y; fset_one(acc.zz); fset_one(acc.zzz); inf = false; }
    else { acc.x = X3; acc.y = Y3; acc.zz = ZZ3; acc.zzz = ZZZ3; }
    if (special) {                                                                                    // P == +-Q: rare; every party takes the same branch
        if (fis_zero_exact(Pd)) {
            if (fis_zero_exact(Rd)) {
                Xyzz<F> one; one.x = q.x; one.y = q.y; fset_one(one.zz); fset_one(one.zzz);
                Xyzz<F> d; xyzz_dbl_shared<F, S>(d, one); acc = d;
            } else inf = true;
        }
    }
}
__device__ __forceinline__ void xyzz_dbl_2l(Xyzz<Fp> &r, const Xyzz<Fp> &a) { xyzz_dbl_shared<Fp, Share2>(r, a); }
__device__ __forceinline__ void xyzz_madd_2l(Xyzz<Fp> &acc, bool &inf, const Aff<Fp> &q, bool neg) { xyzz_madd_shared<Fp, Share2>(acc, inf, q, neg); }

// ---- GLV endomorphism on G1: phi(x, y) = (beta x, y) = lambda (x, y), lambda = x_BLS^2 - 1, beta the matching cube root of unity in Fq ----
// (beta chosen so that phi(G) == lambda G for the generator; the pairing of the two constants is re-checked on the device by
// tests/test_gpu_pairing_checker.py through every scaling it verifies against the oracle.)  Stored as beta * 2^406 mod p, 29-bit limbs.
#define BLS29_BETA {0x1195dfebu, 0x1b04e484u, 0x6026044u, 0x86070a2u, 0x1fd68858u, 0x137e9670u, 0x6871e67u, 0x1e736664u, 0x83b24f6u, 0x8a70373u, 0x2a012fdu, 0x112f94bu, 0x18a2733cu, 0x3u}
__device__ __forceinline__ void xyzz_phi(Xyzz<Fp> &p) {       // phi acts on X only: (beta X / ZZ, Y / ZZZ)
    constexpr uint32_t B_[NL] = BLS29_BETA;
    Fp beta;
#pragma unroll
    for (int i = 0; i < NL; i++) beta.l[i] = B_[i];
    Fp xn; fp_norm(xn, p.x);
    fp_mul(p.x, xn, beta);
}

}  // namespace bls29
