// crypto_amd/csrc/fp_inv.hip.h — field inversion on the device and XYZZ -> affine.
//
// Only the per-element outputs need it: G::normalize_batch after FixedBase::msm (legogroth16/src/generator.rs:424-431),
// the affine points the Miller loop consumes after RandomizedPairingChecker's scalings
// (utils/src/randomized_pairing_check.rs:125-127).  One inversion per lane.
#pragma once
#include "fp29.hip.h"
#include "fp_safegcd.hip.h"
#include "fp2_29.hip.h"
#include "ec29.hip.h"
#include "fp2_pair.hip.h"

namespace bls29 {

// one inversion per lane: Bernstein-Yang division steps (fp_safegcd.hip.h), ~37 k instructions against ~250 k of the Fermat power a^(p-2)
__device__ __forceinline__ void fp_inv_device(Fp &r, const Fp &a) { fp_inv_safegcd(r, a); }
__device__ __forceinline__ void finv(Fp &r, const Fp &a) { fp_inv_device(r, a); }
// 1 / (c0 + c1 u) = (c0 - c1 u) / (c0^2 + c1^2)
__device__ __forceinline__ void finv(Fp2 &r, const Fp2 &a) {
    Fp n0, n1, t, ti, z;
    fp_sqr(n0, a.c0); fp_sqr(n1, a.c1); fp_add(t, n0, n1); fp_norm(t, t);
    fp_inv_device(ti, t);
    fp_mul(r.c0, a.c0, ti);
    fp_mul(n1, a.c1, ti); fp_zero(z); fp_sub<4>(r.c1, z, n1); fp_norm(r.c1, r.c1);
}
// lane-pair form: the norm c0^2 + c1^2 is assembled over the pair, both lanes invert it, each scales (and the odd lane negates) its half
__device__ __forceinline__ void finv(Fp2H &r, const Fp2H &a) {
    Fp sq, other, t, ti, m, z, neg;
    fp_sqr(sq, a.v); xchg(other, sq); fp_add(t, sq, other); fp_norm(t, t);
    fp_inv_device(ti, t);
    fp_mul(m, a.v, ti);
    fp_zero(z); fp_sub<4>(neg, z, m); fp_norm(neg, neg);
    sel(r.v, pair_odd(), neg, m);
}
// (X/ZZ, Y/ZZZ) of a non-identity point, using ZZ^3 == ZZZ^2: 1/ZZ = (ZZ/ZZZ)^2
template <class F> __device__ __forceinline__ void xyzz_to_affine(Aff<F> &r, const Xyzz<F> &a) {
    F i3, t, i2, xn, yn;
    finv(i3, a.zzz);
    fmul(t, a.zz, i3); fsqr(i2, t);
    fnorm(xn, a.x); fnorm(yn, a.y);
    fmul(r.x, xn, i2); fmul(r.y, yn, i3);
}

}  // namespace bls29
