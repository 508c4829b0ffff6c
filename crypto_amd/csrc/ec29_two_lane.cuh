// crypto_amd/csrc/ec29_two_lane.cuh — G1 XYZZ doubling and mixed addition shared by TWO adjacent lanes (device only).
//
// The batched scalar-multiplication kernels (k_g1_scale: RandomizedPairingChecker's `a.mul_bigint(m)`, utils/src/randomized_pairing_check.rs:125-127;
// k_mul_add<G1>: the aggregation's folding step) run one dependent chain of 255 doublings + ~128 additions per point and have far fewer
// lanes than the chip, so they last as long as the instruction stream of one lane.  Here lanes 2k (role A) and 2k+1 (role B) both hold
// the whole point and take one field operation of every round — the SAME operation on role-selected operands, so the wave stays
// convergent — and swap results over DPP quad_perm [1,0,3,2]:
//   doubling (dbl-2008-s-1)          A                    B              mixed addition (madd-2008-s)   A                 B
//     round 1 (square)               V = U^2              X^2              round 1 (product)            U2 = X2 ZZ        S2 = Y2 ZZZ
//     round 2 (product)              W = U V              S = X V          round 2 (square)             PP = P^2          R^2
//     round 3 (square)               M^2                  -                round 3 (product)            PPP = P PP        Q = X1 PP
//     round 4 (product)              M (S - X3)           W Y              round 4 (product)            R (Q - X3)        Y1 PPP
//     round 5 (product)              ZZ' = V ZZ           ZZZ' = W ZZZ     round 5 (product)            ZZ' = ZZ PP       ZZZ' = ZZZ PPP
// 5 field operations per lane instead of 9 / 10.  Same formulas as ec29.cuh (the group element is what is compared anyway).
#pragma once
#include "fp29.cuh"
#include "ec29.cuh"
#include "fp2_pair.cuh"      // xchg, sel, pair_odd

namespace bls29 {

__device__ __forceinline__ void sel2(Fp &r, bool b, const Fp &if_b, const Fp &if_a) { sel(r, b, if_b, if_a); }

// r = 2 a, a not the identity; both lanes hold a and receive r
__device__ __forceinline__ void xyzz_dbl_2l(Xyzz<Fp> &r, const Xyzz<Fp> &a) {
    const bool B = pair_odd();
    Fp U, in, res, oth, V, X2, M, W, S, t, X3, u, v, Ya, Yb, ZZ3, ZZZ3;
    fp_add(U, a.y, a.y); fp_norm(U, U);
    sel2(in, B, a.x, U); fp_sqr(res, in); xchg(oth, res);                       // round 1: A: V = U^2, B: X^2
    sel2(V, B, oth, res); sel2(X2, B, res, oth);
    fp_add(t, X2, X2); fp_add(M, t, X2); fp_norm(M, M);                         // M = 3 X^2
    sel2(u, B, a.x, U); fp_mul(res, u, V); xchg(oth, res);                      // round 2: A: W = U V, B: S = X V
    sel2(W, B, oth, res); sel2(S, B, res, oth);
    fp_sqr(X3, M);                                                              // round 3 (both lanes: same operand)
    fp_add(t, S, S); fp_sub<SubM<Fp>::X>(X3, X3, t); fp_norm(X3, X3);
    fp_sub<SubM<Fp>::D>(t, S, X3); fp_norm(t, t);
    sel2(u, B, W, M); sel2(v, B, a.y, t); fp_mul(res, u, v); xchg(oth, res);    // round 4: A: M (S - X3), B: W Y
    sel2(Ya, B, oth, res); sel2(Yb, B, res, oth);
    Fp Y3; fp_sub<4>(Y3, Ya, Yb); fp_norm(Y3, Y3);      // value < 6 p: the budget SubM<Fp>::R assumes for Y1
    sel2(u, B, W, V); sel2(v, B, a.zzz, a.zz); fp_mul(res, u, v); xchg(oth, res);   // round 5: A: V ZZ, B: W ZZZ
    sel2(ZZ3, B, oth, res); sel2(ZZZ3, B, res, oth);
    r.x = X3; r.y = Y3; r.zz = ZZ3; r.zzz = ZZZ3;
}

// acc += (neg ? -q : q), q affine and not the identity; `inf` is acc's identity flag (same contract as xyzz_madd)
__device__ __forceinline__ void xyzz_madd_2l(Xyzz<Fp> &acc, bool &inf, const Aff<Fp> &q_in, bool neg) {
    const bool B = pair_odd();
    Aff<Fp> q = q_in;
    if (neg) { Fp z; fp_zero(z); fp_sub<SubM<Fp>::NEG>(q.y, z, q.y); fp_norm(q.y, q.y); }
    Fp u, v, res, oth, U2, S2, Pd, Rd, PP, RR, PPP, Q, t, X3, Ya, Yb, Y3, ZZ3, ZZZ3;
    sel2(u, B, q.y, q.x); sel2(v, B, acc.zzz, acc.zz); fp_mul(res, u, v); xchg(oth, res);      // round 1: A: U2, B: S2
    sel2(U2, B, oth, res); sel2(S2, B, res, oth);
    fp_sub<SubM<Fp>::P>(Pd, U2, acc.x); fp_norm(Pd, Pd);
    fp_sub<SubM<Fp>::R>(Rd, S2, acc.y); fp_norm(Rd, Rd);
    sel2(u, B, Rd, Pd); fp_sqr(res, u); xchg(oth, res);                                        // round 2: A: PP, B: R^2
    sel2(PP, B, oth, res); sel2(RR, B, res, oth);
    sel2(u, B, acc.x, Pd); fp_mul(res, u, PP); xchg(oth, res);                                 // round 3: A: PPP, B: Q
    sel2(PPP, B, oth, res); sel2(Q, B, res, oth);
    fp_add(t, Q, Q); fp_add(t, t, PPP);
    fp_sub<SubM<Fp>::X>(X3, RR, t); fp_norm(X3, X3);
    fp_sub<SubM<Fp>::D>(t, Q, X3); fp_norm(t, t);
    sel2(u, B, acc.y, Rd); sel2(v, B, PPP, t); fp_mul(res, u, v); xchg(oth, res);              // round 4: A: R (Q - X3), B: Y1 PPP
    sel2(Ya, B, oth, res); sel2(Yb, B, res, oth);
    fp_sub<4>(Y3, Ya, Yb); fp_norm(Y3, Y3);               // value < 6 p (SubM<Fp>::R budget)
    sel2(u, B, acc.zzz, acc.zz); sel2(v, B, PPP, PP); fp_mul(res, u, v); xchg(oth, res);       // round 5: A: ZZ PP, B: ZZZ PPP
    sel2(ZZ3, B, oth, res); sel2(ZZZ3, B, res, oth);
    const bool was_inf = inf;
    const bool special = !was_inf && fp_maybe_zero(Pd);
    if (was_inf) { acc.x = q.x; acc.y = q.y; fp_set_one(acc.zz); fp_set_one(acc.zzz); inf = false; }
    else { acc.x = X3; acc.y = Y3; acc.zz = ZZ3; acc.zzz = ZZZ3; }
    if (special) {                                                                             // P == +-Q: rare; both lanes take the same branch
        if (fp_is_zero_exact(Pd)) {
            if (fp_is_zero_exact(Rd)) {
                Xyzz<Fp> one; one.x = q.x; one.y = q.y; fp_set_one(one.zz); fp_set_one(one.zzz);
                Xyzz<Fp> d; xyzz_dbl_2l(d, one); acc = d;
            } else inf = true;
        }
    }
}


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
