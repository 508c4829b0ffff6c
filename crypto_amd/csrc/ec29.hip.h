// crypto_amd/csrc/ec29.hip.h — short-Weierstrass (a = 0) group law in extended Jacobian "XYZZ"
// coordinates over the lazy 29-bit-limb fields of fp29.hip.h / fp2_29.hip.h.
//
// Device replacement for ark-ec's Projective += Affine / Projective += Projective used inside
// VariableBaseMSM (third-party ark-ec 0.4; entered from utils/src/pairs.rs:145-155,
// legogroth16/src/prover.rs:286,299,592).  XYZZ (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) costs 8M+2S for a
// mixed add against 7M+4S Jacobian and needs no inversion; the group element is identical, only the
// projective representative differs — results are compared after normalisation to affine.
//
// The formulas are complete: P == Q falls into a doubling and P == -Q into the identity.  The test is a
// 3-instruction necessary condition (fp_maybe_zero) with an exact slow path, so the common path stays
// wave-uniform.  The identity is carried as an explicit flag next to the coordinates.
#pragma once
#include "fp29.hip.h"
#include "fp2_29.hip.h"

namespace bls29 {

template <class F> struct Aff { F x, y; };
template <class F> struct Xyzz { F x, y, zz, zzz; };

// value-growth budget of each subtraction (multiples of p); verified by the FP29_CHECK build
template <class F> struct SubM;
template <> struct SubM<Fp> {
    static constexpr int P = 16;   // U2 - X1      (X1 value < 10 p)
    static constexpr int R = 8;    // S2 - Y1      (Y1 value <  6 p)
    static constexpr int X = 8;    // RR - (PPP + 2Q)
    static constexpr int D = 16;   // Q - X3
    static constexpr int Y = 4;    // U2 - U1, S2 - S1 in the general addition (products)
    static constexpr int YN = 16;  // lazy negation of the c operand of R*(Q - X3) - c*PPP (c = Y1, S1 or W; value < 15 p)
    static constexpr int NEG = 4;  // 0 - y
};
template <> struct SubM<Fp2> {     // an Fp2 product leaves values < 6 p, so every budget is one notch wider
    static constexpr int P = 32;
    static constexpr int R = 16;
    static constexpr int X = 16;
    static constexpr int D = 32;
    static constexpr int Y = 8;
    static constexpr int YN = 16;
    static constexpr int NEG = 4;
};

// fnormw: the carry pass for the widest combination of the formulas (R^2 - PPP - 2 Q); the same as fnorm except for fields whose cheap pass
// has a narrower domain (fp30s.hip.h)
template <class F> FD void fnormw(F &r, const F &a) { fnorm(r, a); }

// q.y := -q.y when neg (the sign bit of a sorted entry).  Generic form: a subtraction from zero and a carry pass behind a branch.  The signed
// fields (fp30s.hip.h, fs2_pair.hip.h) overload it: the negative of a balanced digit is a balanced digit, so there is no carry pass, and
// (d ^ m) - m with m = -neg needs no branch — k_accumulate ran ~65 instructions behind a divergent branch here where 26 do (nearly every
// wave has a lane that subtracts)
template <class F> FD void fcond_neg(F &y, bool neg) {
    if (neg) { F z; fzero(z); fsub<SubM<F>::NEG>(y, z, y); fnorm(y, y); }
}

struct Fp2H;
struct Fs;
template <> struct SubM<Fs> {      // signed digits (fp30s.hip.h): a subtraction needs no multiple of p; the constants are ignored
    static constexpr int P = 0, R = 0, X = 0, D = 0, Y = 0, YN = 0, NEG = 0;
};
template <class F> struct MaddFormulaFirst { static constexpr bool value = true; };
template <> struct MaddFormulaFirst<Fp2H> { static constexpr bool value = false; };

// acc := 2 * (x, y) for an affine, non-identity point (mdbl-2008-s-1)
template <class F> FD void xyzz_dbl_affine(Xyzz<F> &r, const Aff<F> &p) {
    F U, V, W, S, M, t, X3, Y3;
    fdbl(U, p.y); fnorm(U, U);
    fsqr(V, U);
    fmul(W, U, V);
    fmul(S, p.x, V);
    fsqr(M, p.x); fadd(t, M, M); fadd(M, t, M); fnormw(M, M);
    fsqr(X3, M); fadd(t, S, S); fsub<SubM<F>::X>(X3, X3, t); fnormw(X3, X3);
    fsub<SubM<F>::D>(t, S, X3); fnorm(t, t);
    fmul_sub<SubM<F>::YN>(Y3, M, t, W, p.y);
    r.x = X3; r.y = Y3; r.zz = V; r.zzz = W;
}

// r := 2 * a (dbl-2008-s-1), a not the identity
template <class F> FD void xyzz_dbl(Xyzz<F> &r, const Xyzz<F> &a) {
    F U, V, W, S, M, t, X3, Y3;
    fdbl(U, a.y); fnorm(U, U);
    fsqr(V, U);
    fmul(W, U, V);
    fmul(S, a.x, V);
    fsqr(M, a.x); fadd(t, M, M); fadd(M, t, M); fnormw(M, M);
    fsqr(X3, M); fadd(t, S, S); fsub<SubM<F>::X>(X3, X3, t); fnormw(X3, X3);
    fsub<SubM<F>::D>(t, S, X3); fnorm(t, t);
    fmul_sub<SubM<F>::YN>(Y3, M, t, W, a.y);
    fmul(r.zz, V, a.zz); fmul(r.zzz, W, a.zzz);
    r.x = X3; r.y = Y3;
}

// Early-return form of the same addition: special cases tested before the generic formula.  Kept for the lane-pair Fp2 field,
// where the formula-first form needs 58 more spilled registers (k_accumulate<G2P>: 162 vs 104) and runs 7 % slower.
template <class F> FD void xyzz_madd_early(Xyzz<F> &acc, bool &inf, const Aff<F> &q_in, bool neg) {
    Aff<F> q = q_in;
    fcond_neg(q.y, neg);
    if (inf) { acc.x = q.x; acc.y = q.y; fset_one(acc.zz); fset_one(acc.zzz); inf = false; return; }
    F U2, S2, Pd, Rd;
    fmul(U2, q.x, acc.zz);
    fmul(S2, q.y, acc.zzz);
    fsub<SubM<F>::P>(Pd, U2, acc.x); fnorm(Pd, Pd);
    fsub<SubM<F>::R>(Rd, S2, acc.y); fnorm(Rd, Rd);
    if (fmaybe_zero(Pd)) {
        if (fis_zero_exact(Pd)) {
            if (fis_zero_exact(Rd)) xyzz_dbl_affine(acc, q);
            else inf = true;
            return;
        }
    }
    F PP, PPP, Q, t, X3, Y3;
    fsqr(PP, Pd);
    fmul(PPP, Pd, PP);
    fmul(Q, acc.x, PP);
    fsqr(X3, Rd);
    fadd(t, Q, Q); fadd(t, t, PPP);
    fsub<SubM<F>::X>(X3, X3, t); fnormw(X3, X3);
    fsub<SubM<F>::D>(t, Q, X3); fnorm(t, t);
    fmul_sub<SubM<F>::YN>(Y3, Rd, t, acc.y, PPP);
    fmul(acc.zz, acc.zz, PP);
    fmul(acc.zzz, acc.zzz, PPP);
    acc.x = X3; acc.y = Y3;
}

// acc += (neg ? -q : q), q affine and not the identity (madd-2008-s).  `inf` is acc's identity flag.
// The generic formula runs first and unconditionally (on an identity accumulator it works on zeros and the result is
// discarded by selects); the P == +-Q cases are a rare fix-up AFTER it.  With the special cases tested up front the main
// path sat behind two early returns and hipcc emitted ~740 extra instructions per addition into it (a second, 32-bit
// multiply-add chain next to every squaring: 3004 v_mad_u64_u32 + 492 v_mov where 2758 suffice) — 13 % of k_accumulate.
template <class F> FD void xyzz_madd(Xyzz<F> &acc, bool &inf, const Aff<F> &q_in, bool neg) {
    if constexpr (!MaddFormulaFirst<F>::value) { xyzz_madd_early(acc, inf, q_in, neg); return; }
    Aff<F> q = q_in;
    fcond_neg(q.y, neg);
    F U2, S2, Pd, Rd;
    fmul(U2, q.x, acc.zz);
    fmul(S2, q.y, acc.zzz);
    fsub<SubM<F>::P>(Pd, U2, acc.x); fnorm(Pd, Pd);
    fsub<SubM<F>::R>(Rd, S2, acc.y); fnorm(Rd, Rd);
    F PP, PPP, Q, t, X3, Y3, ZZ3, ZZZ3;
    fsqr(PP, Pd);
    fmul(PPP, Pd, PP);
    fmul(Q, acc.x, PP);
    fsqr(X3, Rd);
    fadd(t, Q, Q); fadd(t, t, PPP);
    fsub<SubM<F>::X>(X3, X3, t); fnormw(X3, X3);
    fsub<SubM<F>::D>(t, Q, X3); fnorm(t, t);
    fmul_sub<SubM<F>::YN>(Y3, Rd, t, acc.y, PPP);
    fmul(ZZ3, acc.zz, PP);
    fmul(ZZZ3, acc.zzz, PPP);
    const bool was_inf = inf;
    const bool special = !was_inf && fmaybe_zero(Pd);
    if (was_inf) { acc.x = q.x; acc.y = q.y; fset_one(acc.zz); fset_one(acc.zzz); inf = false; }
    else { acc.x = X3; acc.y = Y3; acc.zz = ZZ3; acc.zzz = ZZZ3; }
    if (special) {
        if (fis_zero_exact(Pd)) {
            if (fis_zero_exact(Rd)) xyzz_dbl_affine(acc, q);
            else inf = true;
        }
    }
}

// a += b, both XYZZ with identity flags (add-2008-s).  Same shape as xyzz_madd: formula first, special cases as a fix-up.
template <class F> FD void xyzz_add(Xyzz<F> &a, bool &ainf, const Xyzz<F> &b, bool binf) {
    F U1, U2, S1, S2, Pd, Rd;
    fmul(U1, a.x, b.zz);
    fmul(U2, b.x, a.zz);
    fmul(S1, a.y, b.zzz);
    fmul(S2, b.y, a.zzz);
    fsub<SubM<F>::Y>(Pd, U2, U1); fnorm(Pd, Pd);
    fsub<SubM<F>::Y>(Rd, S2, S1); fnorm(Rd, Rd);
    F PP, PPP, Q, t, X3, Y3, ZZ3, ZZZ3;
    fsqr(PP, Pd);
    fmul(PPP, Pd, PP);
    fmul(Q, U1, PP);
    fsqr(X3, Rd);
    fadd(t, Q, Q); fadd(t, t, PPP);
    fsub<SubM<F>::X>(X3, X3, t); fnormw(X3, X3);
    fsub<SubM<F>::D>(t, Q, X3); fnorm(t, t);
    fmul_sub<SubM<F>::YN>(Y3, Rd, t, S1, PPP);
    fmul(t, a.zz, b.zz); fmul(ZZ3, t, PP);
    fmul(t, a.zzz, b.zzz); fmul(ZZZ3, t, PPP);
    const bool both = !ainf && !binf;
    const bool special = both && fmaybe_zero(Pd);
    if (special) {
        if (fis_zero_exact(Pd)) {
            if (fis_zero_exact(Rd)) { Xyzz<F> d; xyzz_dbl(d, a); a = d; }
            else ainf = true;
            return;
        }
    }
    if (both) { a.x = X3; a.y = Y3; a.zz = ZZ3; a.zzz = ZZZ3; }
    else if (ainf && !binf) { a = b; ainf = false; }
}

// ---- the general addition / doubling regrouped into rounds of INDEPENDENT products ----------------------------------------------------
// A latency-bound kernel (k_reduce_top: a handful of waves on a chain of ~19 general additions) lasts as long as one lane's instruction
// stream; the 14 products of an addition are only four deep (U, S -> PP, R^2, Z Z' -> PPP, Q, ZZ3 -> Y3 terms, ZZZ3), the 9 of a doubling
// three.  xyzz_add_rounds / xyzz_dbl_rounds are the formulas above written round by round through a `Quad` policy object that decides
// who multiplies: QuadSerial (host, reference) computes all four products of a round itself; the device policy of msm_kernels.hip.h gives
// every point four lanes (lane pairs for G2), each multiplies ONE pair of role-selected operands and the results are broadcast.  Same
// field operations on the same values either way, so one host run under the bound tracker covers both; the only difference to
// xyzz_add is Y3 = R (Q - X3) - S1 PPP as two products and a subtraction instead of one fused two-product reduction.
struct QuadSerial {
    // r[k] = a[k] * b[k] for the four roles
    template <class F> FD void mul4(F (&r)[4], const F (&a)[4], const F (&b)[4], int /*used*/) const { for (int k = 0; k < 4; k++) fmul(r[k], a[k], b[k]); }
};
template <class F, class Q4> FD void xyzz_add_rounds(Xyzz<F> &a, bool &ainf, const Xyzz<F> &b, bool binf, const Q4 &q4) {
    F r[4], Pd, Rd, t, X3, Y3;
    { const F m1[4] = {a.x, b.x, a.y, b.y}, m2[4] = {b.zz, a.zz, b.zzz, a.zzz}; q4.mul4(r, m1, m2, 4); }      // U1 U2 S1 S2
    const F U1 = r[0], S1 = r[2];
    fsub<SubM<F>::Y>(Pd, r[1], r[0]); fnorm(Pd, Pd);
    fsub<SubM<F>::Y>(Rd, r[3], r[2]); fnorm(Rd, Rd);
    { const F m1[4] = {Pd, Rd, a.zz, a.zzz}, m2[4] = {Pd, Rd, b.zz, b.zzz}; q4.mul4(r, m1, m2, 4); }           // PP R^2 ZZ ZZ' ZZZ ZZZ'
    const F PP = r[0], RR = r[1], Z2 = r[2], Z3 = r[3];
    { const F m1[4] = {Pd, U1, Z2, Z2}, m2[4] = {PP, PP, PP, PP}; q4.mul4(r, m1, m2, 3); }                      // PPP Q ZZ3
    const F PPP = r[0], Qv = r[1], ZZ3 = r[2];
    fadd(t, Qv, Qv); fadd(t, t, PPP);
    fsub<SubM<F>::X>(X3, RR, t); fnormw(X3, X3);
    fsub<SubM<F>::D>(t, Qv, X3); fnorm(t, t);
    { const F m1[4] = {Rd, S1, Z3, Z3}, m2[4] = {t, PPP, PPP, PPP}; q4.mul4(r, m1, m2, 3); }                    // R (Q - X3), S1 PPP, ZZZ3
    const F ZZZ3 = r[2];
    fsub<SubM<F>::Y>(Y3, r[0], r[1]); fnorm(Y3, Y3);
    const bool both = !ainf && !binf;
    const bool special = both && fmaybe_zero(Pd);
    if (special) {
        if (fis_zero_exact(Pd)) {
            if (fis_zero_exact(Rd)) { Xyzz<F> d; xyzz_dbl(d, a); a = d; }
            else ainf = true;
            return;
        }
    }
    if (both) { a.x = X3; a.y = Y3; a.zz = ZZ3; a.zzz = ZZZ3; }
    else if (ainf && !binf) { a = b; ainf = false; }
}
template <class F, class Q4> FD void xyzz_dbl_rounds(Xyzz<F> &rr, const Xyzz<F> &a, const Q4 &q4) {
    F U, M, t, X3, Y3, r[4];
    fdbl(U, a.y); fnorm(U, U);
    { const F m1[4] = {U, a.x, U, U}, m2[4] = {U, a.x, U, U}; q4.mul4(r, m1, m2, 2); }                           // V = U^2, X^2
    const F V = r[0];
    fadd(t, r[1], r[1]); fadd(M, t, r[1]); fnormw(M, M);
    { const F m1[4] = {U, a.x, V, M}, m2[4] = {V, V, a.zz, M}; q4.mul4(r, m1, m2, 4); }                          // W S ZZ3 M^2
    const F W = r[0], S = r[1], ZZ3 = r[2];
    fadd(t, S, S); fsub<SubM<F>::X>(X3, r[3], t); fnormw(X3, X3);
    fsub<SubM<F>::D>(t, S, X3); fnorm(t, t);
    { const F m1[4] = {M, W, W, W}, m2[4] = {t, a.y, a.zzz, a.zzz}; q4.mul4(r, m1, m2, 3); }                     // M (S - X3), W Y, ZZZ3
    fsub<SubM<F>::Y>(Y3, r[0], r[1]); fnorm(Y3, Y3);
    rr.x = X3; rr.y = Y3; rr.zz = ZZ3; rr.zzz = r[2];
}

}  // namespace bls29
