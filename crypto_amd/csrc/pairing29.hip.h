// crypto_amd/csrc/pairing29.hip.h — Fp6 / Fp12 tower and BLS12-381 Miller-loop line functions over the lazy 29-bit-limb
// field (fp29.hip.h / fp2_29.hip.h), host+device.
//
// Device side of `Bls12_381::multi_miller_loop` (ark-ec 0.4 models/bls12/{mod,g2}.rs; the reference
// enters it at utils/src/randomized_pairing_check.rs:207 and legogroth16/src/verifier.rs:69-76).  The line
// coefficients are arkworks' exactly (M-twist: doubling (i, 3j, -h), addition (j, -theta, lambda), then c1 *= px,
// c2 *= py), so the raw Fp12 Miller output is bit-identical to the oracle for any partition of the pairs — Fp12
// products are exact and commute.
//
// Tower: Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3 - (1+u)), Fp12 = Fp6[w]/(w^2 - v).
// All functions take and return class-N elements (limbs <= 2^29+7); value bounds are carried by the FP29_CHECK
// build (tests/test_device_code_on_host.py) which proves the chosen subtraction multiples are sufficient.
#pragma once
#include "fp29.hip.h"
#include "fp2_29.hip.h"
#include "ec29.hip.h"

namespace bls29 {

#define BLS29_TWO_INV {0x1d4fdc2u, 0x15d00348u, 0x13894478u, 0x7acde62u, 0x9365b0au, 0x12c2df9bu, 0xdc2d61eu, 0x1e7c2b7du, 0x1c48f65eu, 0xd3f7602u, 0x1aad4478u, 0x13a0d636u, 0x198be187u, 0x4u}

template <class F2> struct Fp6T { F2 c0, c1, c2; };
template <class F2> struct Fp12T { Fp6T<F2> c0, c1; };
typedef Fp6T<Fp2> Fp6d;
typedef Fp12T<Fp2> Fp12d;

// ---- Fp2 extras ----
// square with a caller-chosen subtraction multiple (a.c1 may carry a value up to (M-1) p)
template <int M> FD void f2_sqr_m(Fp2 &r, const Fp2 &a) {
    Fp s, d, t;
    fp_add(s, a.c0, a.c1);
    fp_sub<M>(d, a.c0, a.c1); fp_norm(d, d);
    fp_mul(t, a.c0, a.c1);
    fp_mul(r.c0, s, d);
    fp_add(r.c1, t, t); fp_norm(r.c1, r.c1);
}
// r = a * (1 + u), normalised;  a.c1 value < (M-1) p
template <int M> FD void f2_mul_xi_n(Fp2 &r, const Fp2 &a) {
    Fp t0, t1;
    fp_sub<M>(t0, a.c0, a.c1); fp_add(t1, a.c0, a.c1);
    fp_norm(r.c0, t0); fp_norm(r.c1, t1);
}
template <int M> FD void f2_neg_n(Fp2 &r, const Fp2 &a) { Fp2 z; fzero(z); fsub<M>(r, z, a); fnorm(r, r); }
FD void f2_add_n(Fp2 &r, const Fp2 &a, const Fp2 &b) { fadd(r, a, b); fnorm(r, r); }
template <int M> FD void f2_sub_n(Fp2 &r, const Fp2 &a, const Fp2 &b) { fsub<M>(r, a, b); fnorm(r, r); }

// ---- Fp6 ----
template <class F2> FD void f6_zero(Fp6T<F2> &r) { fzero(r.c0); fzero(r.c1); fzero(r.c2); }
template <class F2> FD void f6_add_n(Fp6T<F2> &r, const Fp6T<F2> &a, const Fp6T<F2> &b) { f2_add_n(r.c0, a.c0, b.c0); f2_add_n(r.c1, a.c1, b.c1); f2_add_n(r.c2, a.c2, b.c2); }
// full product, 6 Fp2 products (Karatsuba).  Inputs: value < ~400 p per coefficient.  Outputs: value < 64 p.
template <class F2> FD void f6_mul(Fp6T<F2> &r, const Fp6T<F2> &a, const Fp6T<F2> &b) {
    F2 t0, t1, t2, s, u, m, x, o0, o1, o2;
    fmul(t0, a.c0, b.c0); fmul(t1, a.c1, b.c1); fmul(t2, a.c2, b.c2);
    f2_add_n(s, a.c1, a.c2); f2_add_n(u, b.c1, b.c2); fmul(m, s, u);
    fadd(x, t1, t2); f2_sub_n<16>(m, m, x);             // (a1+a2)(b1+b2) - t1 - t2
    f2_mul_xi_n<32>(x, m); fadd(o0, x, t0); fnorm(o0, o0);
    f2_add_n(s, a.c0, a.c1); f2_add_n(u, b.c0, b.c1); fmul(m, s, u);
    fadd(x, t0, t1); f2_sub_n<16>(m, m, x);
    f2_mul_xi_n<8>(x, t2); fadd(o1, m, x); fnorm(o1, o1);
    f2_add_n(s, a.c0, a.c2); f2_add_n(u, b.c0, b.c2); fmul(m, s, u);
    fadd(x, t0, t2); f2_sub_n<16>(m, m, x);
    fadd(o2, m, t1); fnorm(o2, o2);
    r.c0 = o0; r.c1 = o1; r.c2 = o2;
}
// r = a * v  (a coefficients value < 127 p)
template <class F2> FD void f6_mul_v(Fp6T<F2> &r, const Fp6T<F2> &a) { F2 t; f2_mul_xi_n<128>(t, a.c2); r.c2 = a.c1; r.c1 = a.c0; r.c0 = t; }
// a * (c0 + c1 v): 5 Fp2 products (ark-ff Fp6::mul_by_01)
template <class F2> FD void f6_mul_by_01(Fp6T<F2> &r, const Fp6T<F2> &a, const F2 &c0, const F2 &c1) {
    F2 aa, bb, s, u, m, x, o0, o1, o2;
    fmul(aa, a.c0, c0); fmul(bb, a.c1, c1);
    f2_add_n(s, a.c1, a.c2); fmul(m, s, c1); f2_sub_n<8>(m, m, bb);      // a1 c1 + a2 c1 - bb = a2 c1
    f2_mul_xi_n<16>(x, m); fadd(o0, x, aa); fnorm(o0, o0);               // xi a2 c1 + a0 c0
    f2_add_n(s, a.c0, a.c2); fmul(m, s, c0); f2_sub_n<8>(m, m, aa); fadd(o2, m, bb); fnorm(o2, o2);   // a2 c0 + a1 c1
    f2_add_n(s, a.c0, a.c1); f2_add_n(u, c0, c1); fmul(m, s, u); fadd(x, aa, bb); f2_sub_n<16>(o1, m, x);   // a0 c1 + a1 c0
    r.c0 = o0; r.c1 = o1; r.c2 = o2;
}
// a * (c1 v): 3 Fp2 products
template <class F2> FD void f6_mul_by_1(Fp6T<F2> &r, const Fp6T<F2> &a, const F2 &c1) {
    F2 t0, t1, t2;
    fmul(t0, a.c2, c1); fmul(t1, a.c0, c1); fmul(t2, a.c1, c1);
    f2_mul_xi_n<8>(r.c0, t0); r.c1 = t1; r.c2 = t2;
}

// ---- Fp12 ----
template <class F2> FD void f12_set_one(Fp12T<F2> &r) { f6_zero(r.c0); f6_zero(r.c1); fset_one(r.c0.c0); }
// 18 Fp2 products.  Inputs: coefficient values < 200 p.  Outputs: < 200 p.
template <class F2> FD void f12_mul(Fp12T<F2> &r, const Fp12T<F2> &a, const Fp12T<F2> &b) {
    Fp6T<F2> t0, t1, s, u, m, x;
    f6_mul(t0, a.c0, b.c0); f6_mul(t1, a.c1, b.c1);
    f6_add_n(s, a.c0, a.c1); f6_add_n(u, b.c0, b.c1); f6_mul(m, s, u);
    // c1 = m - t0 - t1
    fadd(x.c0, t0.c0, t1.c0); fadd(x.c1, t0.c1, t1.c1); fadd(x.c2, t0.c2, t1.c2);
    f2_sub_n<128>(r.c1.c0, m.c0, x.c0); f2_sub_n<128>(r.c1.c1, m.c1, x.c1); f2_sub_n<128>(r.c1.c2, m.c2, x.c2);
    // c0 = t0 + v t1
    f6_mul_v(x, t1);
    f6_add_n(r.c0, t0, x);
}
// f *= (c0 + c1 v + c4 v w)  — ark-ff Fp12::mul_by_014, 13 Fp2 products
template <class F2> FD void f12_mul_by_014(Fp12T<F2> &f, const F2 &c0, const F2 &c1, const F2 &c4) {
    Fp6T<F2> aa, bb, s, m, x; F2 o;
    f6_mul_by_01(aa, f.c0, c0, c1);
    f6_mul_by_1(bb, f.c1, c4);
    f2_add_n(o, c1, c4);
    f6_add_n(s, f.c0, f.c1);
    f6_mul_by_01(m, s, c0, o);
    fadd(x.c0, aa.c0, bb.c0); fadd(x.c1, aa.c1, bb.c1); fadd(x.c2, aa.c2, bb.c2);
    f2_sub_n<128>(f.c1.c0, m.c0, x.c0); f2_sub_n<128>(f.c1.c1, m.c1, x.c1); f2_sub_n<128>(f.c1.c2, m.c2, x.c2);
    f6_mul_v(x, bb);
    f6_add_n(f.c0, aa, x);
}
// ---- the dense Fp12 product as 18 independent Fp2 products ("roles") and 6 output combinations --------------------------------------
// f12_mul above is one long dependent sequence (18 Fp2 products behind each other on whoever computes the node); the product tree of the
// Miller loop is latency-bound, so k_product_tree18 (dock_pairing.hip) gives every node 18 lane pairs: lane pair r forms the operands of
// role r, multiplies, parks the product in LDS, and after a barrier the first 6 lane pairs combine the products into one output
// coefficient each.  Same formulas as f6_mul / f12_mul, only regrouped; the functions below are that regrouping, host + device, so that
// the FP29_CHECK build proves the bounds of exactly what the kernel runs (f12_mul_roles == f12_mul, tests/test_device_code_on_host.py).
//   role r = 6 R + k:  R = 0: a0 b0, 1: a1 b1, 2: (a0 + a1)(b0 + b1)   (Fp6 factors x, y)
//                      k = 0: x0 y0, 1: x1 y1, 2: x2 y2, 3: (x1 + x2)(y1 + y2), 4: (x0 + x1)(y0 + y1), 5: (x0 + x2)(y0 + y2)
// Coefficient order of an Fp12 everywhere below: c[0..5] = c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2.
FD int role_first(int k) { return k < 3 ? k : (k == 3 ? 1 : 0); }
FD int role_second(int k) { return k < 3 ? -1 : (k == 4 ? 1 : 2); }
// r = c ? a : b.  The condition is a function of the role (known where the formula is written down), so the check build carries the bounds of
// the operand that is selected: the other one was computed under its own asserted preconditions and is dropped.
FD void fsel(Fp &r, bool c, const Fp &a, const Fp &b) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
    CHK(for (int i = 0; i < NL; i++) r.ub[i] = c ? a.ub[i] : b.ub[i]; r.vb = c ? a.vb : b.vb;)
}
FD void fsel(Fp2 &r, bool c, const Fp2 &a, const Fp2 &b) { fsel(r.c0, c, a.c0, b.c0); fsel(r.c1, c, a.c1, b.c1); }
// operand of role (R, k): up to four coefficients of one factor, added limb by limb, one carry pass
template <class F2> FD void f12_role_operand(F2 &x, const F2 *c, int R, int k) {
    const int f = role_first(k), s = role_second(k), base = (R == 1) ? 3 : 0;
    F2 t = c[base + f];
    if (s >= 0) fadd(t, t, c[base + s]);
    if (R == 2) { fadd(t, t, c[3 + f]); if (s >= 0) fadd(t, t, c[3 + s]); }
    fnorm(x, t);
}
// coefficient k of an Fp6 product from four of its six role products (f6_mul's three formulas in one shape):
//   k = 0: xi (q12 - p1 - p2) + p0      (Q, A, B, C) = (P3, P1, P2, P0)
//   k = 1: (q01 - p0 - p1) + xi p2                     (P4, P0, P1, P2)
//   k = 2: (q02 - p0 - p2) + p1                        (P5, P0, P2, P1)
FD int role_q(int k) { return 3 + k; }
FD int role_a(int k) { return k == 0 ? 1 : 0; }
FD int role_b(int k) { return k == 1 ? 1 : 2; }
FD int role_c(int k) { return (3 - k) % 3; }
template <class F2> FD void f6_coeff_from_roles(F2 &z, const F2 &Q, const F2 &A, const F2 &B, const F2 &C, int k) {
    F2 x, m, mx, cx, d, c;
    fadd(x, A, B); f2_sub_n<16>(m, Q, x);
    f2_mul_xi_n<32>(mx, m);              // (k == 0)
    f2_mul_xi_n<8>(cx, C);               // (k == 1)
    fsel(d, k == 0, mx, m); fsel(c, k == 1, cx, C);
    fadd(x, d, c); fnorm(z, x);
}
// output coefficient q of the Fp12 product from coefficients of its three Fp6 products T_0 = a0 b0, T_1 = a1 b1, T_2 = (a0 + a1)(b0 + b1):
//   q < 3 : c0.q = T_0.q + (v T_1).q,  v T_1 = (xi T_1.2, T_1.0, T_1.1)        U = T_0.q, V = T_1.((q + 2) % 3)
//   q >= 3: c1.k = T_2.k - T_0.k - T_1.k  (k = q - 3)                           U = T_2.k, V = T_0.k, W = T_1.k
template <class F2> FD void f12_out_c0(F2 &o, const F2 &U, const F2 &V, int q) { F2 vx, v, t; f2_mul_xi_n<128>(vx, V); fsel(v, q == 0, vx, V); fadd(t, U, v); fnorm(o, t); }
template <class F2> FD void f12_out_c1(F2 &o, const F2 &U, const F2 &V, const F2 &W) { F2 x; fadd(x, V, W); f2_sub_n<128>(o, U, x); }
// the whole product through the role functions (reference for the kernel's dataflow; the kernel runs the 18 + 6 pieces on different lanes)
template <class F2> FD void f12_mul_roles(Fp12T<F2> &r, const Fp12T<F2> &a, const Fp12T<F2> &b) {
    const F2 *ac = reinterpret_cast<const F2 *>(&a), *bc = reinterpret_cast<const F2 *>(&b);
    F2 P[18], o[6];
    for (int rr = 0; rr < 18; rr++) { F2 x, y; f12_role_operand(x, ac, rr / 6, rr % 6); f12_role_operand(y, bc, rr / 6, rr % 6); fmul(P[rr], x, y); }
    auto z = [&](F2 &out, int R, int k) { f6_coeff_from_roles(out, P[6 * R + role_q(k)], P[6 * R + role_a(k)], P[6 * R + role_b(k)], P[6 * R + role_c(k)], k); };
    for (int q = 0; q < 3; q++) { F2 U, V; z(U, 0, q); z(V, 1, (q + 2) % 3); f12_out_c0(o[q], U, V, q); }
    for (int k = 0; k < 3; k++) { F2 U, V, W; z(U, 2, k); z(V, 0, k); z(W, 1, k); f12_out_c1(o[3 + k], U, V, W); }
    F2 *rc = reinterpret_cast<F2 *>(&r);
    for (int q = 0; q < 6; q++) rc[q] = o[q];
}
// dense Fp12 from a sparse 014 element
template <class F2> FD void f12_from_014(Fp12T<F2> &f, const F2 &c0, const F2 &c1, const F2 &c4) {
    f6_zero(f.c0); f6_zero(f.c1);
    f.c0.c0 = c0; f.c0.c1 = c1; f.c1.c1 = c4;
}

// ---- G2 line functions (homogeneous projective R = (X, Y, Z); ark-ec bls12/g2.rs double_in_place / add_in_place) ----
template <class F2> struct G2ProjT { F2 x, y, z; };
template <class F2> struct LineT { F2 c0, c1, c2; };
typedef G2ProjT<Fp2> G2Proj;
typedef LineT<Fp2> Line;

// r = a / 2 mod p without a multiplication: the parity of the redundant value is the parity of limb 0 (higher limbs weigh multiples of
// 2^29), so add p when it is odd and halve limb by limb, an odd limb i+1 handing 2^28 down to limb i.  ~60 cheap instructions instead of
// the 490 of a product with 1/2 (ark-ec double_in_place multiplies by TWO_INV twice per step).
FD void fp_half(Fp &r, const Fp &a) {
    BLS29_DECL_P;
    const uint32_t odd = a.l[0] & 1u;
    uint32_t t[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        CHK(assert(a.ub[i] + P_[i] < (1ull << 32));)
        t[i] = a.l[i] + (odd ? P_[i] : 0u);
    }
    Fp h;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) h.l[i] = (t[i] >> 1) + ((t[i + 1] & 1u) << 28);
    h.l[NL - 1] = t[NL - 1] >> 1;
    CHK(for (int i = 0; i < NL; i++) h.ub[i] = (a.ub[i] + P_[i]) / 2 + (i < NL - 1 ? (1u << 28) : 0u); h.vb = (a.vb + 1.0) / 2 + 1e-9; chk_actual(h);)
    fp_norm(r, h);
}
FD void fhalf(Fp2 &r, const Fp2 &a) { fp_half(r.c0, a.c0); fp_half(r.c1, a.c1); }

template <class F2> FD void line_dbl_step(G2ProjT<F2> &R, LineT<F2> &l) {
    F2 a, b, c, e, f, g, h, i, j, e2, t, d;
    fmul(a, R.x, R.y); fhalf(a, a);
    f2_sqr_m<64>(b, R.y);
    f2_sqr_m<64>(c, R.z);
    fadd(t, c, c); fadd(t, t, c); fnorm(t, t);              // 3c
    fdbl(t, t); fdbl(t, t); fnorm(t, t);                    // 12c
    f2_mul_xi_n<128>(e, t);                                 // e = 4(1+u) * 3c
    fadd(f, e, e); fadd(f, f, e); fnorm(f, f);              // f = 3e
    fadd(t, b, f); fhalf(g, t);                             // g = (b+f)/2
    f2_add_n(t, R.y, R.z); f2_sqr_m<64>(h, t); fadd(t, b, c); f2_sub_n<16>(h, h, t);   // h = (Y+Z)^2 - (b+c)
    f2_sub_n<8>(i, e, b);
    f2_sqr_m<64>(j, R.x);
    f2_sqr_m<256>(e2, e);
    f2_sub_n<1024>(d, b, f);
    fmul(R.x, a, d);
    f2_sqr_m<64>(t, g); fadd(d, e2, e2); fadd(d, d, e2); f2_sub_n<32>(R.y, t, d);
    fmul(R.z, b, h);
    l.c0 = i;
    fadd(t, j, j); fadd(t, t, j); fnorm(l.c1, t);
    f2_neg_n<32>(l.c2, h);
}
// ---- the doubling step as k_miller_lines_hex runs it (dock_pairing.hip: sixteen lanes per pair, one instruction stream for every role, so
// every instruction saved is saved on the critical path).  Same VALUES as line_dbl_step — so the raw Miller output does not change — with
// fewer carry passes and halvings:
//   * all five operations of the first round are squarings: X Y = ((X + Y)^2 - X^2 - Y^2) / 2;
//   * a squaring's imaginary part is (2 a0) a1 (no doubling + carry pass afterwards);
//   * 12 c in one scaled carry pass (fp_mul12_norm); f = 3 e, Y' and the three line coefficients are left un-normalised (their consumers
//     add / subtract lazily or carry-pass after loading: k_line_products, k_prepared_from_lines);
//   * X' = (X Y / 2)(b - f) = (X Y)(b - g): one halving instead of two ((b - f) / 2 = b - g).
// This one-lane form exists so that the FP29_CHECK build proves the bounds of exactly this sequence (tests/test_device_code_on_host.py
// runs it next to line_dbl_step); the kernel distributes it over the units of a 16-lane row op for op.
// R.y is left un-normalised (limbs < 2^32, value < 40 p): the next doubling step carry-passes it inside its operand sums; pass norm_y = true
// in front of anything else that consumes R (line_add_step).
template <int M> FD void f2_sqr_u(Fp2 &r, const Fp2 &a) {
    Fp s, d, t2, c0, c1;
    fp_add(s, a.c0, a.c1);
    fp_sub<M>(d, a.c0, a.c1); fp_norm(d, d);
    fp_add(t2, a.c0, a.c0);
    fp_mul(c0, s, d); fp_mul(c1, t2, a.c1);
    r.c0 = c0; r.c1 = c1;
}
FD void f2_mul12_n(Fp2 &r, const Fp2 &a) { fp_mul12_norm(r.c0, a.c0); fp_mul12_norm(r.c1, a.c1); }
template <class F2> FD void line_dbl_step_fast(G2ProjT<F2> &R, LineT<F2> &l, bool norm_y) {
    F2 in, b, c, hs, s1, j, t, e, f, g, h, a2, ah, d2, e2, g2, z;
    fzero(z);
    fadd(t, R.y, z); fnorm(in, t); f2_sqr_u<64>(b, in);                 // round 1 (the kernel: in = A + B, carry pass, square)
    fadd(t, R.z, z); fnorm(in, t); f2_sqr_u<64>(c, in);
    fadd(t, R.y, R.z); fnorm(in, t); f2_sqr_u<64>(hs, in);
    fadd(t, R.x, R.y); fnorm(in, t); f2_sqr_u<64>(s1, in);
    fadd(t, R.x, z); fnorm(in, t); f2_sqr_u<64>(j, in);
    f2_mul12_n(t, c); f2_mul_xi_n<128>(e, t);                           // e = 12 (1 + u) c
    fadd(f, e, e); fadd(f, f, e);                                       // f = 3 e
    fadd(t, b, f); fhalf(g, t);                                         // g = (b + f) / 2
    fsub<8>(l.c0, e, b);                                                // i = e - b
    fadd(t, b, c); f2_sub_n<16>(h, hs, t);                              // h = (Y + Z)^2 - (b + c)
    fadd(t, j, b); fsub<8>(a2, s1, t); fhalf(ah, a2);                   // X Y
    f2_sub_n<256>(d2, b, g);                                            // (b - f) / 2
    fadd(t, j, j); fadd(l.c1, t, j);                                    // 3 j
    fsub<32>(l.c2, z, h);                                               // -h
    F2 nx, nz;
    fmul(e2, e, e); fmul(g2, g, g); fmul(nz, b, h); fmul(nx, ah, d2);   // round 2
    fadd(t, e2, e2); fadd(t, t, e2); fsub<32>(R.y, g2, t);              // Y' = g^2 - 3 e^2
    if (norm_y) fnorm(R.y, R.y);
    R.x = nx; R.z = nz;
}
template <class F2> FD void line_add_step(G2ProjT<F2> &R, const Aff<F2> &Q, LineT<F2> &l) {
    F2 theta, lam, c, d, e, f, g, h, j, t, u;
    fmul(t, Q.y, R.z); f2_sub_n<8>(theta, R.y, t);
    fmul(t, Q.x, R.z); f2_sub_n<8>(lam, R.x, t);
    f2_sqr_m<64>(c, theta); f2_sqr_m<64>(d, lam);
    fmul(e, lam, d); fmul(f, R.z, c); fmul(g, R.x, d);
    fadd(t, e, f); fadd(u, g, g); f2_sub_n<16>(h, t, u);
    f2_sub_n<32>(t, g, h); fmul(t, theta, t); fmul(u, e, R.y);
    fmul(R.x, lam, h);
    f2_sub_n<8>(R.y, t, u);
    fmul(R.z, R.z, e);
    fmul(t, theta, Q.x); fmul(u, lam, Q.y); f2_sub_n<8>(j, t, u);
    l.c0 = j; f2_neg_n<64>(l.c1, theta); l.c2 = lam;
}
// ---- the steps as k_miller_lines_ws runs them (dock_pairing.hip: a wave per role, a lane quad per pair).  Same VALUES as line_dbl_step /
// line_add_step; what differs from line_dbl_step_fast is which products are formed:
//   * a general Fp2 product is FOUR separately reduced Fp products, one per lane of the quad (392 multiply-adds on the critical path instead of
//     the 588 of the fused two-product form): c0 = a0 b0 - a1 b1 + 4 p, c1 = a0 b1 + a1 b0, one carry pass (f2_mul_q);
//   * X Y is such a product of the step's operands (round 1) instead of ((X + Y)^2 - X^2 - Y^2) / 2: no halving in front of round 2;
//   * e^2 and g^2 are squarings (one product per lane) with a subtraction multiple that covers f = 3 e;
//   * Z' is squared as it is (a product's output is class N).
// One-lane forms so that the FP29_CHECK build proves the bounds of exactly these sequences (tests/test_device_code_on_host.py).
FD void f2_mul_q(Fp2 &r, const Fp2 &a, const Fp2 &b) {
    Fp p0, p1, p2, p3, d, s;
    fp_mul(p0, a.c0, b.c0); fp_mul(p1, a.c1, b.c1); fp_mul(p2, a.c0, b.c1); fp_mul(p3, a.c1, b.c0);
    fp_sub<4>(d, p0, p1); fp_add(s, p2, p3);
    fp_norm(r.c0, d); fp_norm(r.c1, s);
}
template <class F2> FD void line_dbl_step_ws(G2ProjT<F2> &R, LineT<F2> &l, bool norm_y) {
    F2 in, yn, b, c, hs, j, xy, t, e, f, g, h, d2, e2, g2, nx, nz, z;
    fzero(z);
    fadd(t, R.y, z); fnorm(yn, t); f2_sqr_u<64>(b, yn);                 // round 1: w0 (sub 0)
    f2_sqr_u<64>(c, R.z);                                               // w1
    fadd(t, R.y, R.z); fnorm(in, t); f2_sqr_u<64>(hs, in);              // w2
    fadd(t, R.x, z); fnorm(in, t); f2_sqr_u<64>(j, in);                 // w0 (sub 1)
    f2_mul_q(xy, R.x, yn);                                              // w3
    f2_mul12_n(t, c); f2_mul_xi_n<128>(e, t);                           // e = 12 (1 + u) c
    fadd(f, e, e); fadd(f, f, e);                                       // f = 3 e
    fadd(t, b, f); fhalf(g, t);                                         // g = (b + f) / 2
    fsub<8>(l.c0, e, b);                                                // i = e - b
    fadd(t, b, c); f2_sub_n<16>(h, hs, t);                              // h = (Y + Z)^2 - (b + c)
    fsub<512>(t, b, f); fhalf(d2, t);                                   // (b - f) / 2 (w3 halves the difference itself)
    fadd(t, j, j); fadd(l.c1, t, j);                                    // 3 j
    fsub<32>(l.c2, z, h);                                               // -h
    f2_sqr_u<512>(e2, e); f2_sqr_u<512>(g2, g); f2_mul_q(nz, b, h); f2_mul_q(nx, xy, d2);   // round 2
    fadd(t, e2, e2); fadd(t, t, e2); fsub<32>(R.y, g2, t);              // Y' = g^2 - 3 e^2
    if (norm_y) fnorm(R.y, R.y);
    R.x = nx; R.z = nz;
}
template <class F2> FD void line_add_step_ws(G2ProjT<F2> &R, const Aff<F2> &Q, LineT<F2> &l) {
    F2 theta, lam, c, d, e, f, g, h, j, t, u, m2, m3, r0, r1, r2, r3;
    f2_mul_q(t, Q.y, R.z); f2_sub_n<8>(theta, R.y, t);
    f2_mul_q(t, Q.x, R.z); f2_sub_n<8>(lam, R.x, t);
    f2_mul_q(c, theta, theta); f2_mul_q(d, lam, lam); f2_mul_q(m2, theta, Q.x); f2_mul_q(m3, lam, Q.y);
    f2_sub_n<8>(j, m2, m3);
    l.c0 = j; f2_neg_n<64>(l.c1, theta); l.c2 = lam;
    f2_mul_q(e, lam, d); f2_mul_q(f, R.z, c); f2_mul_q(g, R.x, d);
    fadd(t, e, f); fadd(u, g, g); f2_sub_n<16>(h, t, u);
    f2_sub_n<32>(t, g, h);
    f2_mul_q(r0, lam, h); f2_mul_q(r1, R.z, e); f2_mul_q(r2, theta, t); f2_mul_q(r3, e, R.y);
    R.x = r0; R.z = r1; f2_sub_n<8>(R.y, r2, r3);
}
// ark-ec `ell` for the M twist: c1 *= px, c2 *= py
template <class F2> FD void line_eval(LineT<F2> &l, const Fp &px, const Fp &py) { fmul_fp(l.c1, l.c1, px); fmul_fp(l.c2, l.c2, py); }

constexpr uint64_t BLS_X_ABS = 0xd201000000010000ULL;
constexpr int N_LINES = 68;

}  // namespace bls29
