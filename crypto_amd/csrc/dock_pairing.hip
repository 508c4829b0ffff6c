// crypto_amd/csrc/dock_pairing.hip — dgpu_multi_miller_loop / dgpu_final_exponentiation (include/dock_gpu.h).
//
// Bls12_381::multi_miller_loop(a, b) (utils/src/randomized_pairing_check.rs:207, legogroth16/src/verifier.rs:69-76)
// computes f = conj( prod_i f_i ), f_i the 63-step double-and-add Miller function of pair i.  arkworks shares the
// 63 squarings inside rayon chunks of 4 pairs.  Because squaring distributes over products, the same value is
//      f = conj( (...((L_0)^2 L_1)^2 ...) ),   L_s = prod_i line_{i,s}(P_i)        (68 steps: 63 doublings + 5 additions)
// so the batch splits into three parts that match the hardware:
//   K9  k_miller_lines    one lane per pair: G2Prepared::from(Q_i) fused with the evaluation at P_i -> 68 sparse lines
//   K10 k_line_products   one lane per (step, slice of pairs): sparse accumulation with mul_by_014
//   K11 k_product_tree    one block per step: dense Fp12 product tree through LDS -> L_s in ABI form
//   host                  131 Fp12 operations (63 squarings + 68 products) + conjugation: 0.25 ms on one core, where a
//                         lone GPU wave would need ~70 us per dense product
// Pairs with an identity member contribute the neutral line (1, 0, 0), which is what arkworks' filter amounts to.
#include "dock_ctx.hpp"
#include "host_field.hpp"
#include "pairing29.hip.h"
#include "fp2_pair.hip.h"
#include "sort_launch.hip.h"
#include "fixed_launch.hip.h"
#include <thread>
#include <functional>
#include <atomic>

namespace bls29 {
__device__ __forceinline__ void fhalf(Fp2H &r, const Fp2H &a) { fp_half(r.v, a.v); }   // lane-pair form of pairing29.hip.h's halving
}

namespace {
using namespace bls29;
using namespace dock;

constexpr int LW = 6 * NL;        // u32 per sparse line (3 Fp2)
constexpr int F12W = 12 * NL;     // u32 per dense Fp12
constexpr int MAX_SLICES = 64;

// lines[(s * LW + k) * n + i]
__global__ void __launch_bounds__(64) k_miller_lines(const uint32_t *__restrict__ p_abi, const uint32_t *__restrict__ q_abi, const uint8_t *__restrict__ skip, size_t n, uint32_t *__restrict__ lines, size_t stride) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool sk = skip && skip[i];
    uint32_t pw[24], qw[48]; uint32_t anyp = 0, anyq = 0;
    for (int k = 0; k < 24; k++) { pw[k] = p_abi[i * 24 + k]; anyp |= pw[k]; }
    for (int k = 0; k < 48; k++) { qw[k] = q_abi[i * 48 + k]; anyq |= qw[k]; }
    if (!anyp || !anyq) sk = true;              // all-zero words == identity
    if (sk) {
        Fp one; fp_set_one(one);
        for (int s = 0; s < N_LINES; s++)
            for (int k = 0; k < LW; k++) lines[((size_t)s * LW + k) * stride + i] = (k < NL) ? one.l[k] : 0u;
        return;
    }
    Fp px, py; fp_from_abi(px, pw); fp_from_abi(py, pw + 12);
    Aff<Fp2> Q; fp_from_abi(Q.x.c0, qw); fp_from_abi(Q.x.c1, qw + 12); fp_from_abi(Q.y.c0, qw + 24); fp_from_abi(Q.y.c1, qw + 36);
    G2Proj R; R.x = Q.x; R.y = Q.y; fset_one(R.z);
    int s = 0;
    for (int b = 62; b >= 0; b--) {
        Line l; line_dbl_step(R, l); line_eval(l, px, py);
        { const uint32_t *w = reinterpret_cast<const uint32_t *>(&l); for (int k = 0; k < LW; k++) lines[((size_t)s * LW + k) * stride + i] = w[k]; }
        s++;
        if ((BLS_X_ABS >> b) & 1) {
            line_add_step(R, Q, l); line_eval(l, px, py);
            const uint32_t *w = reinterpret_cast<const uint32_t *>(&l); for (int k = 0; k < LW; k++) lines[((size_t)s * LW + k) * stride + i] = w[k];
            s++;
        }
    }
}

// Lane-pair version of k_miller_lines (fp2_pair.hip.h): lanes 2i / 2i+1 hold the c0 / c1 halves of every Fp2 value of pair i,
// cross terms move over DPP.  Half the registers per lane (no spills, 2 waves/SIMD) and 2 instead of 3 base-field products per
// Fp2 product on the critical path of the 63 dependent doubling steps.
__global__ void __launch_bounds__(64) k_miller_lines_pair(const uint32_t *__restrict__ p_abi, const uint32_t *__restrict__ q_abi, const uint8_t *__restrict__ skip, size_t n, uint32_t *__restrict__ lines, size_t stride) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
    const uint32_t h = threadIdx.x & 1u;
    if (i >= n) return;
    bool sk = skip && skip[i];
    uint32_t pw[24]; uint32_t anyp = 0, anyq = 0;
    for (int k = 0; k < 24; k++) { pw[k] = p_abi[i * 24 + k]; anyp |= pw[k]; }
    uint32_t qx[12], qy[12];
    for (int k = 0; k < 12; k++) { qx[k] = q_abi[i * 48 + h * 12 + k]; qy[k] = q_abi[i * 48 + 24 + h * 12 + k]; anyq |= qx[k] | qy[k]; }
    anyq |= xchg32(anyq);
    if (!anyp || !anyq) sk = true;
    auto put = [&](int s, const LineT<Fp2H> &l) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&l);          // c0, c1, c2 halves: 3 x 14 words
        for (int c = 0; c < 3; c++) for (int j = 0; j < NL; j++) lines[((size_t)s * LW + (2 * c + h) * NL + j) * stride + i] = w[c * NL + j];
    };
    if (sk) {
        LineT<Fp2H> one; fset_one(one.c0); fzero(one.c1); fzero(one.c2);
        for (int s = 0; s < N_LINES; s++) put(s, one);
        return;
    }
    Fp px, py; fp_from_abi(px, pw); fp_from_abi(py, pw + 12);
    Aff<Fp2H> Q; fp_from_abi(Q.x.v, qx); fp_from_abi(Q.y.v, qy);
    G2ProjT<Fp2H> R; R.x = Q.x; R.y = Q.y; fset_one(R.z);
    int s = 0;
    for (int b = 62; b >= 0; b--) {
        LineT<Fp2H> l; line_dbl_step(R, l); line_eval(l, px, py); put(s++, l);
        if ((BLS_X_ABS >> b) & 1) { line_add_step(R, Q, l); line_eval(l, px, py); put(s++, l); }
    }
}

// ---- four lanes per (P, Q): two lane pairs share the doubling step ------------------------------------------------------------------
// The 63 doubling steps are one dependent chain per pair of points and the kernel has far fewer lanes than the chip (1024 pairs =
// 32 waves on 1024 SIMDs): its duration is the instruction count of one lane.  A doubling step is 3 Fp2 products + 6 Fp2 squarings
// (+ the two evaluations at P); here lane pair A (quad lanes 0,1) and lane pair B (lanes 2,3) each hold the whole state and take one
// operation of every round — the SAME operation on role-selected operands, so the wave stays convergent — and swap results over DPP
// quad_perm [2,3,0,1]:        A                 B
//   round 1 (square)          b = Y^2           c = Z^2
//   round 2 (square)          (Y + Z)^2         j = X^2
//   round 3 (square)          e^2               g^2
//   round 4 (product)         a = X Y           Z' = b h
//   round 5 (product)         X' = (a/2) d      -
//   round 6 (times px | py)   c1 = 3 j px       c2 = -h py
// Same formulas as line_dbl_step (ark-ec bls12/g2.rs double_in_place), so the raw Miller-loop output stays bit-identical.
// The five addition steps run redundantly on both pairs.
__device__ __forceinline__ bool quad_hi() { return (threadIdx.x & 2u) != 0; }
__device__ __forceinline__ void xq(Fp2H &r, const Fp2H &a) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.v.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.v.l[i], 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true);
}
__device__ __forceinline__ void selq(Fp2H &r, bool hi, const Fp2H &if_hi, const Fp2H &if_lo) { sel(r.v, hi, if_hi.v, if_lo.v); }

// on return l.c0 is complete on both pairs; l.c1 is valid on pair A, l.c2 on pair B (both already multiplied by px / py)
// EVAL = false: round 6 is left to the product kernel (k_line_products multiplies by px / py as it loads a line): the evaluation at P is
// not part of the dependent chain R -> 2R, and this kernel lasts as long as its chain
template <bool EVAL = true>
__device__ __forceinline__ void line_dbl_step_quad(G2ProjT<Fp2H> &R, LineT<Fp2H> &l, const Fp &px, const Fp &py) {
    const bool B = quad_hi();
    Fp2H in, res, oth, b, c, e, f, g, hh, h, i, j, e2, g2, d, t, u, v;
    selq(in, B, R.z, R.y); f2_sqr_m<64>(res, in); xq(oth, res);                    // round 1
    selq(b, B, oth, res); selq(c, B, res, oth);
    fadd(t, c, c); fadd(t, t, c); fnorm(t, t);
    fdbl(t, t); fdbl(t, t); fnorm(t, t);
    f2_mul_xi_n<128>(e, t);
    fadd(f, e, e); fadd(f, f, e); fnorm(f, f);
    fadd(t, b, f); fhalf(g, t);
    f2_sub_n<8>(i, e, b);
    f2_sub_n<1024>(d, b, f);
    f2_add_n(t, R.y, R.z); selq(in, B, R.x, t); f2_sqr_m<64>(res, in); xq(oth, res);   // round 2
    selq(hh, B, oth, res); selq(j, B, res, oth);
    fadd(t, b, c); f2_sub_n<16>(h, hh, t);
    selq(in, B, g, e); f2_sqr_m<256>(res, in); xq(oth, res);                      // round 3
    selq(e2, B, oth, res); selq(g2, B, res, oth);
    selq(u, B, b, R.x); selq(v, B, h, R.y); fmul(res, u, v);                      // round 4: A: X Y, B: b h
    Fp2H ah; fhalf(ah, res);
    fmul(t, ah, d);                                                               // round 5: A: X' (B's value is not used)
    Fp2H give; selq(give, B, res, t); xq(oth, give);                              // A hands X' over and receives Z'
    Fp2H nx, ny, nz;
    selq(nx, B, oth, t); selq(nz, B, res, oth);
    fadd(d, e2, e2); fadd(d, d, e2); f2_sub_n<32>(ny, g2, d);
    R.x = nx; R.y = ny; R.z = nz;
    l.c0 = i;
    Fp2H c1u, c2u; fadd(t, j, j); fadd(t, t, j); fnorm(c1u, t); f2_neg_n<32>(c2u, h);
    selq(in, B, c2u, c1u);
    if constexpr (EVAL) { Fp k; sel(k, B, py, px); fmul_fp(res, in, k); }         // round 6
    else res = in;
    l.c1 = res; l.c2 = res;
}

// The loop may be cut in two launches (dgpu_multi_miller_loop below): this launch runs the bits b_hi .. b_lo of |x| and writes the lines
// s_first ..; a launch that does not start at bit 62 takes R from `state`, one that does not end at bit 0 leaves it there
// (state[k * 4 n + lane]: 3 x NL words per lane, both lane pairs of a quad hold the whole R).
__device__ __host__ inline int ml_steps(int b_hi, int b_lo) { int c = 0; for (int b = b_hi; b >= b_lo; b--) c += 1 + (int)((BLS_X_ABS >> b) & 1); return c; }
// EVAL = false: the lines leave unevaluated (c1, c2 not yet multiplied by px, py) and the launch that starts the chain writes px, py in
// the internal form to pxy[(c * NL + k) * stride + i] (zeros for a skipped pair) for k_line_products.
template <bool EVAL = true>
__global__ void __launch_bounds__(64) k_miller_lines_quad(const uint32_t *__restrict__ p_abi, const uint32_t *__restrict__ q_abi, const uint8_t *__restrict__ skip, size_t n, uint32_t *__restrict__ lines, size_t stride,
                                                          int b_hi = 62, int b_lo = 0, int s_first = 0, uint32_t *__restrict__ state = nullptr, uint32_t *__restrict__ pxy = nullptr) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t i = gid >> 2;
    const uint32_t h = threadIdx.x & 1u;
    const bool B = quad_hi();
    if (i >= n) return;
    bool sk = skip && skip[i];
    uint32_t pw[24]; uint32_t anyp = 0, anyq = 0;
    if (p_abi) { for (int k = 0; k < 24; k++) { pw[k] = p_abi[i * 24 + k]; anyp |= pw[k]; } }
    else { for (int k = 0; k < 24; k++) pw[k] = 0; anyp = 1; }          // no P: the coefficients alone (dgpu_g2_prepare; EVAL = false, pxy = nullptr)
    uint32_t qx[12], qy[12];
    for (int k = 0; k < 12; k++) { qx[k] = q_abi[i * 48 + h * 12 + k]; qy[k] = q_abi[i * 48 + 24 + h * 12 + k]; anyq |= qx[k] | qy[k]; }
    anyq |= xchg32(anyq);
    if (!anyp || !anyq) sk = true;
    // pair A stores c0 and c1, pair B stores c2 (each lane its own half)
    auto put = [&](int s, const LineT<Fp2H> &l) {
        auto st = [&](int c, const Fp2H &x) { for (int j = 0; j < NL; j++) lines[((size_t)s * LW + (2 * c + h) * NL + j) * stride + i] = x.v.l[j]; };
        if (!B) { st(0, l.c0); st(1, l.c1); } else st(2, l.c2);
    };
    if (sk) {
        LineT<Fp2H> one; fset_one(one.c0); fzero(one.c1); fzero(one.c2);
        const int s_end = s_first + ml_steps(b_hi, b_lo);
        for (int s = s_first; s < s_end; s++) put(s, one);
        if constexpr (!EVAL) { if (pxy && b_hi == 62 && (gid & 3) < 2) for (int k = 0; k < NL; k++) pxy[((gid & 3) * NL + k) * stride + i] = 0; }
        return;
    }
    Fp px, py; fp_from_abi(px, pw); fp_from_abi(py, pw + 12);
    if constexpr (!EVAL) { if (pxy && b_hi == 62 && (gid & 3) < 2) { const Fp &c = (gid & 3) ? py : px; for (int k = 0; k < NL; k++) pxy[((gid & 3) * NL + k) * stride + i] = c.l[k]; } }
    Aff<Fp2H> Q; fp_from_abi(Q.x.v, qx); fp_from_abi(Q.y.v, qy);
    G2ProjT<Fp2H> R;
    const size_t lanes = 4 * n;
    if (b_hi == 62) { R.x = Q.x; R.y = Q.y; fset_one(R.z); }
    else {
        for (int k = 0; k < NL; k++) { R.x.v.l[k] = state[(size_t)k * lanes + gid]; R.y.v.l[k] = state[(size_t)(NL + k) * lanes + gid]; R.z.v.l[k] = state[(size_t)(2 * NL + k) * lanes + gid]; }
    }
    int s = s_first;
    for (int b = b_hi; b >= b_lo; b--) {
        LineT<Fp2H> l; line_dbl_step_quad<EVAL>(R, l, px, py); put(s++, l);
        if ((BLS_X_ABS >> b) & 1) { line_add_step(R, Q, l); if constexpr (EVAL) line_eval(l, px, py); put(s++, l); }
    }
    if (b_lo > 0)
        for (int k = 0; k < NL; k++) { state[(size_t)k * lanes + gid] = R.x.v.l[k]; state[(size_t)(NL + k) * lanes + gid] = R.y.v.l[k]; state[(size_t)(2 * NL + k) * lanes + gid] = R.z.v.l[k]; }
}

// ---- sixteen lanes per (P, Q): a doubling step two Fp2 operations deep -------------------------------------------------------------------
// k_miller_lines_quad lasts as long as ONE lane's instruction stream: 63 doubling steps x 5 rounds (3 squarings + 2 products) + 5 addition
// steps of 11 products, with 64 waves on 1024 SIMDs at 1024 pairs.  The nine Fp2 operations of a doubling step fall into TWO groups of
// mutually independent ones, so here a (P, Q) owns one 16-lane DPP row = eight lane pairs ("units" u0 .. u7, unit = lane pair, Fp2 halves on
// its two lanes as in fp2_pair.hip.h), six of which work:
//   round 1 (all squarings)   u0 b = Y^2   u1 c = Z^2   u2 (Y + Z)^2   u3 (X + Y)^2   u4 j = X^2            [X Y = ((X + Y)^2 - j - b) / 2]
//   round 2 (products)        u0 e^2       u1 g^2       u2 Z' = b h    u3 X' = (X Y / 2) d                  [e, g, h, d: linear in b, c]
// then Y' = g^2 - 3 e^2 on u1 (e^2 over DPP).  Every lane runs the same instruction stream; WHO computes WHAT is decided by which lane a value is
// fetched from: ds_bpermute lets each lane name its own source lane, so the operand of round 1 of the next step is in = A + B with
// (A, B) fetched from the units that hold X', Y', Z' (or from an idle unit that holds zero) — no role selects in front of the squaring.
// An addition step (5 of 68) is four rounds of products on u0 .. u3 with whole-row broadcasts in between.  Same formulas as
// line_dbl_step / line_add_step (pairing29.hip.h), i.e. the same field VALUES: the Miller output stays bit-identical (tests compare all forms).
// Only the unevaluated form exists (c1, c2 are multiplied by px, py in k_line_products; pxy written by the launch that starts the chain).
// state (two-launch form): R of pair i, half h at state[(comp * NL + k) * 2 n + 2 i + h].
// lane-pair forms of pairing29.hip.h's f2_sqr_u / fmul / f2_mul12_n (the operations line_dbl_step_fast is proved with), spelled for the
// fewest instructions: every lane of the wave runs them, whatever its role
template <int M> __device__ __forceinline__ void hx_sqr(Fp2H &r, const Fp2H &a) {      // even: (a0 + a1)(a0 - a1)   odd: (2 a0) a1
    const bool odd = pair_odd();
    Fp ao, t, U, d, V;
    xchg(ao, a.v);
    sel(t, odd, ao, a.v); fp_add(U, ao, t);
    fp_sub<M>(d, a.v, ao); fp_norm(d, d);
    sel(V, odd, a.v, d);
    fp_mul(r.v, U, V);
}
__device__ __forceinline__ void hx_mul(Fp2H &r, const Fp2H &a, const Fp2H &b) {        // even: a0 b0 + (-a1) b1   odd: a0 b1 + a1 b0
    const bool odd = pair_odd();
    Fp ao, bo, z, nao, P, Q;
    xchg(ao, a.v); xchg(bo, b.v);
    fp_zero(z); fp_sub<512>(nao, z, ao); fp_norm(nao, nao);
    sel(P, odd, ao, a.v); sel(Q, odd, a.v, nao);
    fp_mul2(r.v, P, b.v, Q, bo);
}
__device__ __forceinline__ void f2_mul12_n(Fp2H &r, const Fp2H &a) { fp_mul12_norm(r.v, a.v); }
__device__ __forceinline__ void hx_fetch(Fp2H &r, const Fp2H &a, int src_byte) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.v.l[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src_byte, (int)a.v.l[i]);
}
__global__ void __launch_bounds__(64) k_miller_lines_hex(const uint32_t *__restrict__ p_abi, const uint32_t *__restrict__ q_abi, const uint8_t *__restrict__ skip, size_t n, uint32_t *__restrict__ lines, size_t stride,
                                                         int b_hi, int b_lo, int s_first, uint32_t *__restrict__ state, uint32_t *__restrict__ pxy) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = gid >> 4;
    const uint32_t lane = threadIdx.x & 63u, h = lane & 1u, unit = (lane >> 1) & 7u;
    if (i >= n) return;
    bool sk = skip && skip[i];
    uint32_t anyp = 1, anyq = 0;
    uint32_t qx[12], qy[12];
    for (int k = 0; k < 12; k++) { qx[k] = q_abi[i * 48 + h * 12 + k]; qy[k] = q_abi[i * 48 + 24 + h * 12 + k]; anyq |= qx[k] | qy[k]; }
    anyq |= xchg32(anyq);
    uint32_t pw[12];                                            // this lane's coordinate of P (h = 0: x, h = 1: y): only to hand px, py to the product kernel
    if (p_abi) { uint32_t a = 0; for (int k = 0; k < 12; k++) { pw[k] = p_abi[i * 24 + h * 12 + k]; a |= pw[k]; } anyp = a | xchg32(a); }
    else for (int k = 0; k < 12; k++) pw[k] = 0;
    if (!anyp || !anyq) sk = true;
    // coefficient half (c, h) of line s is written by ONE lane: its 14 words
    auto st = [&](int s, uint32_t c, const Fp2H &x) { for (int j = 0; j < NL; j++) lines[((size_t)s * LW + (2 * c + h) * NL + j) * stride + i] = x.v.l[j]; };
    const bool starts = b_hi == 62;
    if (sk) {
        const int s_end = s_first + ml_steps(b_hi, b_lo);
        if (unit < 3) { Fp2H v; if (unit == 0) fset_one(v); else fzero(v); for (int s = s_first; s < s_end; s++) st(s, unit, v); }
        if (pxy && starts && unit == 0) for (int k = 0; k < NL; k++) pxy[(h * NL + k) * stride + i] = 0;
        return;
    }
    if (pxy && starts) { Fp c; fp_from_abi(c, pw); if (unit == 0) for (int k = 0; k < NL; k++) pxy[(h * NL + k) * stride + i] = c.l[k]; }
    Aff<Fp2H> Q; fp_from_abi(Q.x.v, qx); fp_from_abi(Q.y.v, qy);
    const int row = (int)((lane & 48u) + h) * 4;                 // byte address of this lane's half in unit 0 of its row
    auto from = [&](uint32_t u) { return row + 8 * (int)u; };
    // who holds what after a doubling step (val: X' on u3, Y' on u1, Z' on u2, zero on u4 .. u7) and what round 1 squares: in = A + B
    //                       u0: Y      u1: Z      u2: Y + Z   u3: X + Y   u4: X      u5 .. u7: 0
    const int srcA = from((0x55533121u >> (4 * unit)) & 7u), srcB = from((0x55551255u >> (4 * unit)) & 7u);
    const int srcJ = from(4);
    Fp2H X, Y, Z;                                                // the whole of R: valid at the start, around an addition step and at the end
    if (starts) { X = Q.x; Y = Q.y; fset_one(Z); }
    else {
        const size_t w = 2 * n, at = 2 * i + h;
        for (int k = 0; k < NL; k++) { X.v.l[k] = state[(size_t)k * w + at]; Y.v.l[k] = state[(size_t)(NL + k) * w + at]; Z.v.l[k] = state[(size_t)(2 * NL + k) * w + at]; }
    }
    // round-1 operand from a whole R (start, after an addition step): the same table, spelled with selects
    auto operand_from_R = [&](Fp2H &in) {
        Fp2H a, b, z, t; fzero(z);
        fsel(a, unit == 1, Z, Y); fsel(a, unit >= 3, X, a); fsel(a, unit >= 5, z, a);
        fsel(b, unit == 2, Z, z); fsel(b, unit == 3, Y, b);
        fadd(t, a, b); fnorm(in, t);
    };
    Fp2H in; operand_from_R(in);
    int s = s_first;
    for (int b = b_hi; b >= b_lo; b--) {
        // ---- doubling step (ark-ec double_in_place) as line_dbl_step_fast (pairing29.hip.h: bounds proved on the host) ----
        Fp2H res, B, C, J, t, e, f, g, hh, ii, d2, a2, ah, c1v, c2v, lv, eg, opA, opB, oth, yy, val, z;
        fzero(z);
        hx_sqr<64>(res, in);                                     // round 1
        hx_fetch(B, res, from(0)); hx_fetch(C, res, from(1)); hx_fetch(J, res, srcJ);
        f2_mul12_n(t, C); f2_mul_xi_n<128>(e, t);               // e = 12 (1 + u) c
        fadd(f, e, e); fadd(f, f, e);                           // f = 3e (lazy)
        fadd(t, B, f); fhalf(g, t);                             // g = (b + f) / 2
        fsub<8>(ii, e, B);                                      // line c0 (lazy: the consumer carry-passes)
        fadd(t, B, C); f2_sub_n<16>(hh, res, t);                // u2: h = (Y + Z)^2 - (b + c)
        fadd(t, J, B); fsub<8>(a2, res, t); fhalf(ah, a2);      // u3: X Y = ((X + Y)^2 - j - b) / 2
        f2_sub_n<256>(d2, B, g);                                // (b - f) / 2
        fadd(t, res, res); fadd(c1v, t, res);                   // u4: 3j
        fsub<32>(c2v, z, hh);                                   // u2: -h
        fsel(lv, unit == 4, c1v, ii); fsel(lv, unit == 2, c2v, lv);
        if (unit == 1 || unit == 2 || unit == 4) st(s, unit == 1 ? 0u : (unit == 4 ? 1u : 2u), lv);
        s++;
        fsel(eg, unit == 1, g, e);
        fsel(opA, unit == 2, B, eg); fsel(opA, unit == 3, ah, opA);
        fsel(opB, unit == 2, hh, eg); fsel(opB, unit == 3, d2, opB);
        hx_mul(res, opA, opB);                                   // round 2: u0 e^2, u1 g^2, u2 b h, u3 (X Y)(b - g)
        xq(oth, res);                                            // u1 receives e^2
        fadd(t, oth, oth); fadd(t, t, oth); fsub<32>(yy, res, t);              // u1: Y' = g^2 - 3 e^2 (lazy)
        fsel(val, unit == 1, yy, res); fsel(val, unit >= 4, z, val);
        const bool add = (BLS_X_ABS >> b) & 1;
        if (!add && b > b_lo) { Fp2H A2, B2; hx_fetch(A2, val, srcA); hx_fetch(B2, val, srcB); fadd(t, A2, B2); fnorm(in, t); continue; }
        hx_fetch(X, val, from(3)); hx_fetch(Y, val, from(1)); hx_fetch(Z, val, from(2)); fnorm(Y, Y);
        if (add) {
            // ---- addition step (ark-ec add_in_place): four rounds of products on u0 .. u3 ----
            Fp2H t1, t2, theta, lam, cc, dd, m2, m3, jj, ee, ff, gg, h2, r0, r1, r2, r3;
            fsel(opA, unit == 0, Q.y, Q.x); fmul(res, opA, Z);                                     // u0: Qy Z   u1: Qx Z
            hx_fetch(t1, res, from(0)); hx_fetch(t2, res, from(1));
            f2_sub_n<8>(theta, Y, t1); f2_sub_n<8>(lam, X, t2);
            fsel(opA, (unit & 1u) != 0, lam, theta);
            fsel(opB, unit == 1, lam, theta); fsel(opB, unit == 2, Q.x, opB); fsel(opB, unit == 3, Q.y, opB);
            fmul(res, opA, opB);                                                                   // u0: theta^2  u1: lam^2  u2: theta Qx  u3: lam Qy
            hx_fetch(cc, res, from(0)); hx_fetch(dd, res, from(1)); hx_fetch(m2, res, from(2)); hx_fetch(m3, res, from(3));
            f2_sub_n<8>(jj, m2, m3);
            { Fp2H nt; f2_neg_n<64>(nt, theta); fsel(lv, unit == 1, nt, jj); fsel(lv, unit == 2, lam, lv); }
            if (unit < 3) st(s, unit, lv);                                                         // (j, -theta, lam)
            s++;
            fsel(opA, unit == 1, Z, lam); fsel(opA, unit == 2, X, opA);
            fsel(opB, unit == 1, cc, dd);
            fmul(res, opA, opB);                                                                   // u0: e = lam d  u1: f = Z c  u2: g = X d
            hx_fetch(ee, res, from(0)); hx_fetch(ff, res, from(1)); hx_fetch(gg, res, from(2));
            fadd(t, ee, ff); { Fp2H u2; fadd(u2, gg, gg); f2_sub_n<16>(h2, t, u2); }
            f2_sub_n<32>(t, gg, h2);
            fsel(opA, unit == 1, Z, lam); fsel(opA, unit == 2, theta, opA); fsel(opA, unit == 3, ee, opA);
            fsel(opB, unit == 1, ee, h2); fsel(opB, unit == 2, t, opB); fsel(opB, unit == 3, Y, opB);
            fmul(res, opA, opB);                                                                   // u0: lam h  u1: Z e  u2: theta (g - h)  u3: e Y
            hx_fetch(r0, res, from(0)); hx_fetch(r1, res, from(1)); hx_fetch(r2, res, from(2)); hx_fetch(r3, res, from(3));
            X = r0; Z = r1; f2_sub_n<8>(Y, r2, r3);
        }
        operand_from_R(in);
    }
    if (b_lo > 0 && unit == 0) {
        const size_t w = 2 * n, at = 2 * i + h;
        for (int k = 0; k < NL; k++) { state[(size_t)k * w + at] = X.v.l[k]; state[(size_t)(NL + k) * w + at] = Y.v.l[k]; state[(size_t)(2 * NL + k) * w + at] = Z.v.l[k]; }
    }
}
// ---- four waves per sixteen pairs: every wave runs ONE role's instruction stream ------------------------------------------------------------
// k_miller_lines_hex gives a pair sixteen lanes of one wave, so every lane runs every role's linear work (e, f, g, h, X Y, the selects that
// route operands): 2584 wave-instructions per doubling step of which 1005 are multiply-adds, 4.8 us.  Here a role is a WAVE: a workgroup of four
// waves owns sixteen pairs, lane = 4 pair + 2 sub + half, and the roles talk through LDS (one 16-byte-wide slot per value and lane, two barriers
// per doubling step, waited for ~60 cycles each):
//   round 1   w0: Y^2 (sub 0) and X^2 (sub 1)   w1: Z^2, then e = 12 xi c, f = 3 e   w2: (Y + Z)^2     w3: X Y (four-lane product)
//   round 2   w0: e^2 (and the line's 3 j)      w1: g^2 (and e - b)                   w2: b h (and -h)  w3: (X Y)(b - g)
// so a wave runs its own operand's linear work only (branches on the role are scalar) and a general Fp2 product is one Fp product per lane of
// the quad (qx_mul) instead of a fused two-product reduction per lane of a pair.  An addition step is the hex kernel's four rounds with the
// same split.  The VALUES are those of line_dbl_step / line_add_step (pairing29.hip.h line_dbl_step_ws / line_add_step_ws: the one-lane forms
// the FP29_CHECK build proves the bounds of), so the Miller output and G2Prepared's bytes do not change (tests/test_gpu_pairing.py: every mode).
template <int CTRL> __device__ __forceinline__ void qfetch(Fp &r, const Fp &a) {       // DPP quad_perm: lane q of a quad reads lane (CTRL >> 2 q) & 3
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], CTRL, 0xF, 0xF, true);
}
// Fp2 product on a lane QUAD whose lanes q = 2 sub + half all hold their half of a and b (pairing29.hip.h f2_mul_q): one separately reduced Fp
// product per lane — q0: a0 b0, q1: a1 b1, q2: a0 b1, q3: a1 b0 — then c0 = P0 - P1 + 4 p on the even lanes, c1 = P2 + P3 on the odd ones, carry pass
__device__ __forceinline__ void qx_mul(Fp2H &r, const Fp2H &a, const Fp2H &b) {
    Fp bs, P, Xv, Yv, d, s2, t;
    qfetch<0xB4>(bs, b.v);                                       // [0, 1, 3, 2]: the lanes of sub 1 take the other half of b
    fp_mul(P, a.v, bs);
    qfetch<0x88>(Xv, P); qfetch<0xDD>(Yv, P);                    // [0, 2, 0, 2], [1, 3, 1, 3]: even lanes (P0, P1), odd lanes (P2, P3)
    fp_sub<4>(d, Xv, Yv); fp_add(s2, Xv, Yv);
    sel(t, pair_odd(), s2, d);
    fp_norm(r.v, t);
}
constexpr int WS_PAIRS = 16;
enum { WS_BJ = 0, WS_C = 1, WS_E = 2, WS_F = 3, WS_GG = 4, WS_T1 = 5, WS_T2 = 6, WS_E2 = 7, WS_G2 = 8, WS_NZ = 9, WS_NX = 10, WS_SLOTS = 11,
       WS_CC = 0, WS_DD = 1, WS_M2 = 2, WS_M3 = 3, WS_EE = 5, WS_FF = 6, WS_R0 = 7, WS_R1 = 8, WS_R2 = 9, WS_R3 = 10 };
typedef uint32_t WsSlot[4][64][4];                              // limb quad, lane, four limbs: a 16-byte access per lane, lanes side by side
__device__ __forceinline__ void ws_put(WsSlot &s, uint32_t lane, const Fp2H &x) {
    static_assert(NL == 14, "three quads and a pair");
#pragma unroll
    for (int q = 0; q < 3; q++) *reinterpret_cast<uint4 *>(s[q][lane]) = make_uint4(x.v.l[4 * q], x.v.l[4 * q + 1], x.v.l[4 * q + 2], x.v.l[4 * q + 3]);
    *reinterpret_cast<uint2 *>(s[3][lane]) = make_uint2(x.v.l[12], x.v.l[13]);
}
__device__ __forceinline__ void ws_get(Fp2H &r, const WsSlot &s, uint32_t lane) {
#pragma unroll
    for (int q = 0; q < 3; q++) { const uint4 v = *reinterpret_cast<const uint4 *>(s[q][lane]); r.v.l[4 * q] = v.x; r.v.l[4 * q + 1] = v.y; r.v.l[4 * q + 2] = v.z; r.v.l[4 * q + 3] = v.w; }
    const uint2 v = *reinterpret_cast<const uint2 *>(s[3][lane]); r.v.l[12] = v.x; r.v.l[13] = v.y;
}
__global__ void __launch_bounds__(256) k_miller_lines_ws(const uint32_t *__restrict__ p_abi, const uint32_t *__restrict__ q_abi, const uint8_t *__restrict__ skip, size_t n, uint32_t *__restrict__ lines, size_t stride,
                                                         int b_hi, int b_lo, int s_first, uint32_t *__restrict__ state, uint32_t *__restrict__ pxy) {
    __shared__ WsSlot L[WS_SLOTS];
    __builtin_amdgcn_s_setprio(3);                                // the chain is what a call waits for: its waves go first where a product kernel's wave shares their SIMD
    const uint32_t lane = threadIdx.x & 63u, h = lane & 1u;
    const bool sub = (lane & 2u) != 0;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lane_b = lane & ~2u, lane_j = lane | 2u;     // where w0 left b and j of this lane's pair and half
    const size_t i_raw = (size_t)blockIdx.x * WS_PAIRS + (lane >> 2);
    const bool inr = i_raw < n;
    const size_t i = inr ? i_raw : n - 1;                        // lanes past the end keep step with the barriers on the last pair's data and store nothing
    bool sk = skip && skip[i];
    uint32_t anyp = 1, anyq = 0;
    uint32_t qx[12], qy[12];
    for (int k = 0; k < 12; k++) { qx[k] = q_abi[i * 48 + h * 12 + k]; qy[k] = q_abi[i * 48 + 24 + h * 12 + k]; anyq |= qx[k] | qy[k]; }
    anyq |= xchg32(anyq);
    uint32_t pw[12];
    if (p_abi) { uint32_t a = 0; for (int k = 0; k < 12; k++) { pw[k] = p_abi[i * 24 + h * 12 + k]; a |= pw[k]; } anyp = a | xchg32(a); }
    else for (int k = 0; k < 12; k++) pw[k] = 0;
    if (!anyp || !anyq) sk = true;
    const bool live = inr && !sk;
    auto st = [&](int s, uint32_t c, const Fp2H &x) { for (int j = 0; j < NL; j++) lines[((size_t)s * LW + (2 * c + h) * NL + j) * stride + i] = x.v.l[j]; };
    const bool starts = b_hi == 62;
    if (inr && sk && !sub) {                                     // the neutral line at every step; the pair's lanes go on computing (barriers) and store nothing
        const int s_end = s_first + ml_steps(b_hi, b_lo);
        if (role < 3) { Fp2H v; if (role == 0) fset_one(v); else fzero(v); for (int s = s_first; s < s_end; s++) st(s, (uint32_t)role, v); }
        if (pxy && starts && role == 3) for (int k = 0; k < NL; k++) pxy[(h * NL + k) * stride + i] = 0;
    }
    if (pxy && starts && live && role == 3 && !sub) { Fp c; fp_from_abi(c, pw); for (int k = 0; k < NL; k++) pxy[(h * NL + k) * stride + i] = c.l[k]; }
    Aff<Fp2H> Q; fp_from_abi(Q.x.v, qx); fp_from_abi(Q.y.v, qy);
    Fp2H X, Y, Z;                                                // the whole of R (every wave): valid at the start, around an addition step and at the end
    if (starts) { X = Q.x; Y = Q.y; fset_one(Z); }
    else {
        const size_t w = 2 * n, at = 2 * i + h;
        for (int k = 0; k < NL; k++) { X.v.l[k] = state[(size_t)k * w + at]; Y.v.l[k] = state[(size_t)(NL + k) * w + at]; Z.v.l[k] = state[(size_t)(2 * NL + k) * w + at]; }
    }
    bool whole = true;
    Fp2H keep; fzero(keep);                                      // this wave's own round-2 product (w2: Z', w3: X')
    int s = s_first;
#ifdef WS_PROF
    uint64_t pc[6] = {0, 0, 0, 0, 0, 0}, pt;
#define WSP(k) { uint64_t now_ = __builtin_amdgcn_s_memtime(); pc[k] += now_ - pt; pt = now_; }
    pt = __builtin_amdgcn_s_memtime();
#else
#define WSP(k)
#endif
    for (int b = b_hi; b >= b_lo; b--) {
        Fp2H z; fzero(z);
        // ---- round 1: squarings of in = A + B (carry-passed; Z' as it is), w3: the product X Y.  One straight path per role (a wave runs one of
        // them): merging the roles' operands in front of a shared product costs more register copies than the product's code is worth ----
        Fp2H t, r1, res, lv;
        auto next_y = [&](Fp2H &yy, const WsSlot &g2slot) {      // Y' = g^2 - 3 e^2 (lazy)
            Fp2H v1, v2, u; ws_get(v1, g2slot, lane); ws_get(v2, L[WS_E2], lane);
            fadd(u, v2, v2); fadd(u, u, v2); fsub<32>(yy, v1, u);
        };
        if (role == 0) {
            Fp2H A, in;
            if (whole) fsel(A, sub, X, Y);
            else { Fp2H v1, v2, yy, u; ws_get(v1, L[sub ? WS_NX : WS_G2], lane); ws_get(v2, L[WS_E2], lane);
                   fadd(u, v2, v2); fadd(u, u, v2); fsub<32>(yy, v1, u); fsel(A, sub, v1, yy); }
            fadd(t, A, z); fnorm(in, t);
            hx_sqr<64>(r1, in);                                  // sub 0: b = Y^2   sub 1: j = X^2
            ws_put(L[WS_BJ], lane, r1);
        } else if (role == 1) {
            Fp2H in, e, f;
            if (whole) in = Z; else ws_get(in, L[WS_NZ], lane);
            hx_sqr<64>(r1, in);                                  // c = Z^2
            f2_mul12_n(t, r1); f2_mul_xi_n<128>(e, t);          // e = 12 (1 + u) c
            fadd(f, e, e); fadd(f, f, e);                       // f = 3 e (lazy)
            ws_put(L[WS_C], lane, r1); ws_put(L[WS_E], lane, e); ws_put(L[WS_F], lane, f);
        } else if (role == 2) {
            Fp2H in;
            if (whole) fadd(t, Y, Z); else { Fp2H yy; next_y(yy, L[WS_G2]); fadd(t, yy, keep); }
            fnorm(in, t);
            hx_sqr<64>(r1, in);                                  // (Y + Z)^2
        } else {
            if (whole) qx_mul(r1, X, Y);
            else { Fp2H yy, yn; next_y(yy, L[WS_G2]); fnorm(yn, yy); qx_mul(r1, keep, yn); }          // X Y
        }
        WSP(0)
        __syncthreads();
        WSP(1)
        // ---- round 2 ----
        if (role == 0) {
            Fp2H Ev; ws_get(Ev, L[WS_E], lane);
            fadd(t, r1, r1); fadd(lv, t, r1);                    // sub 1: the line's 3 j
            if (sub && live) st(s, 1, lv);
            hx_sqr<512>(res, Ev);                                // e^2
            ws_put(L[WS_E2], lane, res);
        } else if (role == 1) {
            Fp2H Bv, Fv, Ev, g;
            ws_get(Bv, L[WS_BJ], lane_b); ws_get(Fv, L[WS_F], lane); ws_get(Ev, L[WS_E], lane);
            fadd(t, Bv, Fv); fhalf(g, t);                        // g = (b + f) / 2
            fsub<8>(lv, Ev, Bv);                                 // line c0 = e - b (lazy)
            if (!sub && live) st(s, 0, lv);
            hx_sqr<512>(res, g);                                 // g^2
            ws_put(L[WS_G2], lane, res);
        } else if (role == 2) {
            Fp2H Bv, Cv, hh;
            ws_get(Bv, L[WS_BJ], lane_b); ws_get(Cv, L[WS_C], lane);
            fadd(t, Bv, Cv); f2_sub_n<16>(hh, r1, t);            // h = (Y + Z)^2 - (b + c)
            fsub<32>(lv, z, hh);                                 // -h
            if (!sub && live) st(s, 2, lv);
            qx_mul(res, Bv, hh);                                 // Z' = b h
            ws_put(L[WS_NZ], lane, res);
        } else {
            Fp2H Bv, Fv, d2;
            ws_get(Bv, L[WS_BJ], lane_b); ws_get(Fv, L[WS_F], lane);
            fsub<512>(t, Bv, Fv); fhalf(d2, t);                  // (b - f) / 2 = b - g, halved directly (g itself is w1's)
            qx_mul(res, r1, d2);                                 // X' = (X Y)(b - g)
            ws_put(L[WS_NX], lane, res);
        }
        s++;
        keep = res;
        WSP(2)
        __syncthreads();
        WSP(3)
        whole = false;
        const bool add = (BLS_X_ABS >> b) & 1;
        if (!add && b > b_lo) continue;
        { Fp2H g2, e2; ws_get(X, L[WS_NX], lane); ws_get(Z, L[WS_NZ], lane); ws_get(g2, L[WS_G2], lane); ws_get(e2, L[WS_E2], lane);
          fadd(t, e2, e2); fadd(t, t, e2); fsub<32>(Y, g2, t); fnorm(Y, Y); }
        whole = true;
        WSP(4)
        if (!add) continue;
        // ---- addition step (ark-ec add_in_place): four rounds, one product per wave and round ----
        // (kept as a ROLLED loop over the round index: the build with `#pragma unroll` here, -DWS_UNROLL, returns wrong lines from the first addition step on, for every
        // pair and the same ones run after run: a compile-time difference, not a race — DESIGN.md 10 — while this one passes every form's comparison and the soaks; the slot schedule: a round's outputs go to slots whose last readers sit behind a
        // barrier every wave has passed: T1 T2 -> CC DD M2 M3 (= BJ C E F) -> EE FF GG (EE FF = T1 T2) -> R0 .. R3 (= E2 G2 NZ NX))
        Fp2H theta, lam, cc, dd, ee, h2, gmh, opA, opB;
#ifdef WS_UNROLL
#pragma unroll
#else
#pragma nounroll
#endif
        for (int r = 0; r < 4; r++) {
            bool work = true;
            if (r == 0) { opA = role == 0 ? Q.y : Q.x; opB = Z; work = role < 2; }                      // w0: Qy Z   w1: Qx Z
            else if (r == 1) {                                                                         // w0: theta^2  w1: lam^2  w2: theta Qx  w3: lam Qy
                opA = (role & 1) ? lam : theta;
                opB = role == 0 ? theta : (role == 1 ? lam : (role == 2 ? Q.x : Q.y));
            } else if (r == 2) {                                                                       // w0: e = lam d  w1: f = Z c  w2: g = X d
                opA = role == 0 ? lam : (role == 1 ? Z : X);
                opB = role == 1 ? cc : dd; work = role < 3;
            } else {                                                                                   // w0: lam h  w1: Z e  w2: theta (g - h)  w3: e Y
                opA = role == 0 ? lam : (role == 1 ? Z : (role == 2 ? theta : ee));
                opB = role == 0 ? h2 : (role == 1 ? ee : (role == 2 ? gmh : Y));
            }
            if (work) {
                qx_mul(res, opA, opB);
                const int slot = r == 0 ? WS_T1 + role : (r == 1 ? WS_CC + role : (r == 2 ? (role == 2 ? WS_GG : WS_EE + role) : WS_R0 + role));
                ws_put(L[slot], lane, res);
            }
            __syncthreads();
            if (r == 0) {
                Fp2H t1, t2; ws_get(t1, L[WS_T1], lane); ws_get(t2, L[WS_T2], lane);
                f2_sub_n<8>(theta, Y, t1); f2_sub_n<8>(lam, X, t2);
            } else if (r == 1) {
                ws_get(cc, L[WS_CC], lane); ws_get(dd, L[WS_DD], lane);
                if (role == 0) { Fp2H m2, m3; ws_get(m2, L[WS_M2], lane); ws_get(m3, L[WS_M3], lane); f2_sub_n<8>(lv, m2, m3); }
                else if (role == 1) f2_neg_n<64>(lv, theta);
                else lv = lam;
                if (role < 3 && !sub && live) st(s, (uint32_t)role, lv);                               // (j, -theta, lam)
                s++;
            } else if (r == 2) {
                Fp2H ff, gg, u2; ws_get(ee, L[WS_EE], lane); ws_get(ff, L[WS_FF], lane); ws_get(gg, L[WS_GG], lane);
                fadd(t, ee, ff); fadd(u2, gg, gg); f2_sub_n<16>(h2, t, u2);
                f2_sub_n<32>(gmh, gg, h2);
            } else {
                Fp2H r2, r3; ws_get(X, L[WS_R0], lane); ws_get(Z, L[WS_R1], lane); ws_get(r2, L[WS_R2], lane); ws_get(r3, L[WS_R3], lane);
                f2_sub_n<8>(Y, r2, r3);
            }
        }
        WSP(5)
    }
#ifdef WS_PROF
    if (blockIdx.x == 0 && lane == 0) printf("role %d: r1 %llu waitA %llu r2 %llu waitB %llu whole-read %llu add %llu (cycles over %d steps)\n", role, (unsigned long long)pc[0], (unsigned long long)pc[1], (unsigned long long)pc[2], (unsigned long long)pc[3], (unsigned long long)pc[4], (unsigned long long)pc[5], s - s_first);
#endif
    if (b_lo > 0 && role == 0 && !sub && inr) {
        const size_t w = 2 * n, at = 2 * i + h;
        for (int k = 0; k < NL; k++) { state[(size_t)k * w + at] = X.v.l[k]; state[(size_t)(NL + k) * w + at] = Y.v.l[k]; state[(size_t)(2 * NL + k) * w + at] = Z.v.l[k]; }
    }
}
// one launcher for the line kernels that leave the evaluation to the product kernel: sixteen lanes per pair while the chip has room for them
// (dgpu_set_miller_pipeline bit 2), four otherwise.  Same arguments, same lines, same state size bound (3 NL 4 n words).
constexpr size_t ML_HEX_MAX = 4096;      // 16 lanes x 4096 pairs = 1024 waves: one per SIMD
static void launch_lines_uneval(hipStream_t s, const uint32_t *p_abi, const uint32_t *q_abi, const uint8_t *skip, size_t n, uint32_t *lines, size_t stride,
                                int b_hi, int b_lo, int s_first, uint32_t *state, uint32_t *pxy) {
    const int mode = gs.ml_mode.load();
    if ((mode & 8) && (mode & 4) && n <= ML_HEX_MAX)
        hipLaunchKernelGGL(k_miller_lines_ws, dim3((unsigned)((n + WS_PAIRS - 1) / WS_PAIRS)), dim3(256), 0, s, p_abi, q_abi, skip, n, lines, stride, b_hi, b_lo, s_first, state, pxy);
    else if ((mode & 4) && n <= ML_HEX_MAX)
        hipLaunchKernelGGL(k_miller_lines_hex, dim3((unsigned)((16 * n + 63) / 64)), dim3(64), 0, s, p_abi, q_abi, skip, n, lines, stride, b_hi, b_lo, s_first, state, pxy);
    else
        hipLaunchKernelGGL(k_miller_lines_quad<false>, dim3((unsigned)((4 * n + 63) / 64)), dim3(64), 0, s, p_abi, q_abi, skip, n, lines, stride, b_hi, b_lo, s_first, state, pxy);
}

// ---- G2Prepared (ark-ec bls12/g2.rs `G2Prepared::from`: the 68 coefficient triples before the evaluation at P) --------------------------
// The reference's verifier and pairing checker hold ONLY prepared G2 values (legogroth16/src/verifier.rs:69-76 passes
// pvk.delta_g2_neg_pc / gamma_g2_neg_pc, data_structures.rs:118-120; utils/src/randomized_pairing_check.rs:35 queues Vec<E::G2Prepared>),
// and a G2Prepared cannot be turned back into a point, so the boundary needs both directions:
//   k_g2_prepare           Q -> ell_coeffs in the ABI form (canonical 2^384 Montgomery limbs: the same bytes arkworks holds)
//   k_lines_from_prepared  (P, ell_coeffs) -> the evaluated sparse lines K10 consumes; every (pair, step) is independent here, so this is
//                          one lane per (pair, step) instead of a 63-step dependent chain per pair
constexpr int CW = 72;            // u32 per coefficient triple in the ABI (3 Fp2 = 6 x 12 words)
__global__ void __launch_bounds__(64) k_g2_prepare(const uint32_t *__restrict__ q_abi, const uint8_t *__restrict__ is_inf, size_t n, uint32_t *__restrict__ out, uint8_t *__restrict__ out_inf) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
    const uint32_t h = threadIdx.x & 1u;
    if (i >= n) return;
    uint32_t qx[12], qy[12], anyq = 0;
    for (int k = 0; k < 12; k++) { qx[k] = q_abi[i * 48 + h * 12 + k]; qy[k] = q_abi[i * 48 + 24 + h * 12 + k]; anyq |= qx[k] | qy[k]; }
    anyq |= xchg32(anyq);
    const bool inf = (is_inf && is_inf[i]) || !anyq;
    uint32_t *dst = out + i * (size_t)(N_LINES * CW);
    if (h == 0) out_inf[i] = inf ? 1 : 0;
    if (inf) {                                                       // arkworks: ell_coeffs = vec![], infinity = true
        for (int s = 0; s < N_LINES; s++) for (int c = 0; c < 3; c++) for (int k = 0; k < 12; k++) dst[s * CW + (2 * c + h) * 12 + k] = 0u;
        return;
    }
    auto put = [&](int s, const LineT<Fp2H> &l) {
        fp_to_abi(dst + s * CW + (0 + h) * 12, l.c0.v); fp_to_abi(dst + s * CW + (2 + h) * 12, l.c1.v); fp_to_abi(dst + s * CW + (4 + h) * 12, l.c2.v);
    };
    Aff<Fp2H> Q; fp_from_abi(Q.x.v, qx); fp_from_abi(Q.y.v, qy);
    G2ProjT<Fp2H> R; R.x = Q.x; R.y = Q.y; fset_one(R.z);
    int s = 0;
    for (int b = 62; b >= 0; b--) {
        LineT<Fp2H> l; line_dbl_step(R, l); put(s++, l);
        if ((BLS_X_ABS >> b) & 1) { line_add_step(R, Q, l); put(s++, l); }
    }
}
// The coefficient triples as k_miller_lines_quad<false> leaves them (K10's layout, internal limbs) -> arkworks' ell_coeffs bytes: one thread
// per (point, step, coefficient half).  dgpu_g2_prepare = the four-lanes-per-point chain (5 rounds per doubling step, no conversion inside
// the chain) + this fully parallel pass, instead of the lane-pair chain that converted three coefficients per step on its way (2.2 -> 1.2 ms
// for 1024 points, 0.4 ms of it the 20 MB going back to the host).
__global__ void __launch_bounds__(256) k_prepared_from_lines(const uint32_t *__restrict__ lines, const uint32_t *__restrict__ q_abi, const uint8_t *__restrict__ is_inf, size_t n,
                                                             uint32_t *__restrict__ out, uint8_t *__restrict__ out_inf) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * N_LINES * 6) return;
    const size_t i = t % n, r = t / n; const int ch = (int)(r % 6), s = (int)(r / 6);          // ch = 2 c + h
    uint32_t anyq = 0;
    for (int k = 0; k < 48; k += 4) { const uint4 v = *reinterpret_cast<const uint4 *>(q_abi + i * 48 + k); anyq |= v.x | v.y | v.z | v.w; }
    const bool inf = (is_inf && is_inf[i]) || !anyq;
    if (ch == 0 && s == 0) out_inf[i] = inf ? 1 : 0;
    uint32_t *dst = out + i * (size_t)(N_LINES * CW) + (size_t)s * CW + ch * 12;
    if (inf) { for (int k = 0; k < 12; k++) dst[k] = 0u; return; }                              // arkworks: ell_coeffs = vec![], infinity = true
    Fp f;
    for (int k = 0; k < NL; k++) f.l[k] = lines[((size_t)s * LW + ch * NL + k) * n + i];
    uint32_t w[12]; fp_to_abi(w, f);
    for (int k = 0; k < 12; k++) dst[k] = w[k];
}
// thread (s, i), i fastest: ark-ec `ell` (c1 *= px, c2 *= py) on coefficient triple s of pair i, written in K10's layout
// pxy_one != nullptr (a mixed call whose product kernel evaluates the affine pairs' lines): these pairs' lines are evaluated HERE, so their
// (px, py) for the product kernel is (1, 1) — pxy_one points at this kernel's first pair
__global__ void __launch_bounds__(256) k_lines_from_prepared(const uint32_t *__restrict__ p_abi, const uint32_t *__restrict__ coeffs, const uint8_t *__restrict__ skip, size_t n, uint32_t *__restrict__ lines, size_t stride,
                                                              uint32_t *__restrict__ pxy_one = nullptr) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * N_LINES) return;
    const size_t s = t / n, i = t % n;
    if (pxy_one && s == 0) { Fp one; fp_set_one(one); for (int k = 0; k < NL; k++) { pxy_one[(size_t)k * stride + i] = one.l[k]; pxy_one[(size_t)(NL + k) * stride + i] = one.l[k]; } }
    uint32_t pw[24], anyp = 0;
    for (int k = 0; k < 24; k += 4) { uint4 v = *reinterpret_cast<const uint4 *>(p_abi + i * 24 + k); pw[k] = v.x; pw[k + 1] = v.y; pw[k + 2] = v.z; pw[k + 3] = v.w; anyp |= v.x | v.y | v.z | v.w; }
    const uint32_t *src = coeffs + (i * N_LINES + s) * (size_t)CW;
    uint32_t cw[CW], anyc = 0;
    for (int k = 0; k < CW; k += 4) { uint4 v = *reinterpret_cast<const uint4 *>(src + k); cw[k] = v.x; cw[k + 1] = v.y; cw[k + 2] = v.z; cw[k + 3] = v.w; anyc |= v.x | v.y | v.z | v.w; }
    uint32_t any0 = 0;                                               // an identity Q is an all-zero block: its first triple decides (a real doubling line has c1 = 3 x^2 != 0)
    for (int k = 0; k < CW; k += 4) { uint4 v = *reinterpret_cast<const uint4 *>(coeffs + i * (size_t)(N_LINES * CW) + k); any0 |= v.x | v.y | v.z | v.w; }
    (void)anyc;
    const bool sk = (skip && skip[i]) || !anyp || !any0;
    uint32_t *dst = lines + (s * LW) * stride + i;
    if (sk) { Fp one; fp_set_one(one); for (int k = 0; k < LW; k++) dst[(size_t)k * stride] = (k < NL) ? one.l[k] : 0u; return; }
    Fp px, py; fp_from_abi(px, pw); fp_from_abi(py, pw + 12);
#pragma unroll
    for (int c = 0; c < 6; c++) {
        Fp f; fp_from_abi(f, cw + 12 * c);
        if (c >= 2) { Fp m; fp_mul(m, f, c < 4 ? px : py); f = m; }
        for (int k = 0; k < NL; k++) dst[(size_t)(c * NL + k) * stride] = f.l[k];
    }
}

// partial[(s * nsl + j) * F12W + k] = product of the lines of step s over slice j of the pairs (sparse Fp12::mul_by_014 chain).
// One chain per LANE PAIR (fp2_pair.hip.h): the chain is serial, so its duration is the instruction count of one lane, and the pair form
// of an Fp2 product is one fused two-product reduction per lane instead of two.  Word (2 q + h) * NL + j of an Fp12 is limb j of half h
// of its q-th Fp2 coefficient — the same order a one-lane Fp12d has in memory.
typedef Fp6T<Fp2H> Fp6p;
typedef Fp12T<Fp2H> Fp12p;
// seg_off != nullptr: nseg independent products over the pairs [seg_off[g], seg_off[g + 1]) of one line buffer (dgpu_multi_miller_loop_segments);
// partial (g * N_LINES + s) * nsl + j is slice j of step s of segment g, slices past the end of a segment are not written.
// s0, ns: the steps s0 .. s0 + ns - 1 only, of every segment (a call whose line kernel runs in several launches).
__global__ void __launch_bounds__(64) k_line_products(const uint32_t *__restrict__ lines, size_t n, int slice_len, int nsl, uint32_t *__restrict__ partial,
                                                      const uint32_t *__restrict__ seg_off, int nseg, int s0 = 0, int ns = N_LINES, const uint32_t *__restrict__ pxy = nullptr) {
    int t = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1);
    const uint32_t h = threadIdx.x & 1u;
    if (t >= ns * nsl * nseg) return;
    const int per = ns * nsl, g = t / per, rem = t % per, s = s0 + rem / nsl, j = rem % nsl;      // (segment, step of the range, slice)
    t = (g * N_LINES + s) * nsl + j;
    const size_t first = seg_off ? seg_off[g] : 0, last = seg_off ? seg_off[g + 1] : n;
    // slice j of the `have` slices of this (segment, step) is the pairs first + j, first + j + have, ...: neighbouring lane pairs read
    // neighbouring pairs of a row (full 128-B lines; contiguous slices made every lane of a wave touch its own line: 37 -> 9 ms of
    // k_line_products at 2^18 pairs).  The product does not depend on the order of its factors.
    const size_t have = (last - first + slice_len - 1) / slice_len;
    if ((size_t)j >= have) return;
    Fp12p f; f12_set_one(f);
    const size_t lo = first + j;
    for (size_t i = lo; i < last; i += have) {
        LineT<Fp2H> l;
        for (int c = 0; c < 3; c++) { Fp2H &x = c == 0 ? l.c0 : (c == 1 ? l.c1 : l.c2); for (int k = 0; k < NL; k++) x.v.l[k] = lines[((size_t)s * LW + (2 * c + h) * NL + k) * n + i]; }
        fnorm(l.c0, l.c0); fnorm(l.c1, l.c1); fnorm(l.c2, l.c2);       // (k_miller_lines_hex leaves its doubling lines un-normalised)
        if (pxy) {                                    // the lines came unevaluated (k_miller_lines_quad<false> / _hex): c1 *= px, c2 *= py here
            Fp px, py;
            uint32_t anyp = 0;
            for (int k = 0; k < NL; k++) { px.l[k] = pxy[(size_t)k * n + i]; py.l[k] = pxy[(size_t)(NL + k) * n + i]; anyp |= px.l[k] | py.l[k]; }
            line_eval(l, px, py);
            // px = py = 0 is no point of the curve: it is how a P that turned out to be the identity AFTER the chain of Q started arrives here (the scaled
            // Miller loop: [m] P = O for a P outside the prime-order subgroup) — the pair contributes one, as a pair skipped up front does
            if (!anyp) { fset_one(l.c0); fzero(l.c1); fzero(l.c2); }
        }
        if (i == lo) f12_from_014(f, l.c0, l.c1, l.c2); else f12_mul_by_014(f, l.c0, l.c1, l.c2);
    }
    const Fp2H *q = reinterpret_cast<const Fp2H *>(&f);              // c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2
    for (int c = 0; c < 6; c++) for (int k = 0; k < NL; k++) partial[(size_t)t * F12W + (2 * c + h) * NL + k] = q[c].v.l[k];
}

// ---- the sparse products of a launch that leaves the chip nearly empty: three waves per 32 slices -------------------------------------------------
// k_line_products inlines the thirteen Fp2 products of a mul_by_014 and the two of the evaluation: 107 KB of straight code, more than the
// instruction cache, so a lone wave per SIMD waits for nearly every instruction it runs (22 cycles each: 113 us for ONE product per slice at 1024
// pairs).  Here a block of three waves owns 32 slices (lane pair = slice, as before) and a wave runs a THIRD of the product — w0: aa = f.c0 x
// (c0, c1), w1: m = (f.c0 + f.c1) x (c0, c1 + c4), w2: bb = f.c1 x c4 and the line's evaluation is shared the same way — so each wave's code stays in the
// cache and the three parts run side by side; f, the line, aa and bb travel through LDS (k_miller_lines_ws's slots), two barriers per line.  The same
// operations on the same values as f12_mul_by_014 (pairing29.hip.h), so the same bounds; every slice runs slice_len rounds (a slice one pair short
// multiplies by the neutral line in its last one: the same value).  For the pieces of the pipelined Miller loop (ml_products); launches that fill the
// chip keep k_line_products, which is throughput-bound there.
enum { LP3_F = 0, LP3_L0 = 6, LP3_L1 = 7, LP3_L4 = 8, LP3_AA = 9, LP3_BB = 12, LP3_SLOTS = 15 };
__global__ void __launch_bounds__(192) k_line_products3(const uint32_t *__restrict__ lines, size_t n, int slice_len, int nsl, uint32_t *__restrict__ partial, int s0, int ns, const uint32_t *__restrict__ pxy,
                                                        const uint32_t *__restrict__ seg_off = nullptr, int nseg = 1) {
    __shared__ WsSlot L[LP3_SLOTS];
    const uint32_t lane = threadIdx.x & 63u, h = lane & 1u;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int per = ns * nsl, total = per * nseg;                   // (segment, step of the range, slice) as in k_line_products
    int t = (int)(blockIdx.x * 32u + (lane >> 1));
    bool inr = t < total;
    if (!inr) t = total - 1;
    const int g = t / per, rem = t % per, s = s0 + rem / nsl, j = rem % nsl;
    const size_t first = seg_off ? seg_off[g] : 0, last = seg_off ? seg_off[g + 1] : n;
    const size_t have = (last - first + slice_len - 1) / slice_len;  // slices of this segment (<= nsl); a lane pair beyond them keeps step with the barriers and stores nothing
    if ((size_t)j >= have) inr = false;
    const size_t lo = first + (size_t)j;
    Fp2H z; fzero(z);
    for (int it = 0; it < slice_len; it++) {
        const size_t i = lo + (size_t)it * have;
        const bool valid = inr && i < last;
        const size_t ic = valid ? i : n - 1;
        // ---- the line, evaluated: w2 takes c0, w0 c1 px, w1 c2 py ----
        {
            const int c = role == 2 ? 0 : (role == 0 ? 1 : 2);
            Fp2H x; for (int k = 0; k < NL; k++) x.v.l[k] = lines[((size_t)s * LW + (2 * c + h) * NL + k) * n + ic];
            fnorm(x, x);
            bool neutral = !valid;
            if (pxy) {
                Fp px, py; uint32_t anyp = 0;
                for (int k = 0; k < NL; k++) { px.l[k] = pxy[(size_t)k * n + ic]; py.l[k] = pxy[(size_t)(NL + k) * n + ic]; anyp |= px.l[k] | py.l[k]; }
                if (role == 0) fmul_fp(x, x, px); else if (role == 1) fmul_fp(x, x, py);
                if (!anyp) neutral = true;                             // (px = py = 0: the pair contributes one, see k_line_products)
            }
            if (neutral) { if (role == 2) fset_one(x); else fzero(x); }
            ws_put(L[role == 2 ? LP3_L0 : (role == 0 ? LP3_L1 : LP3_L4)], lane, x);
        }
        __syncthreads();
        if (it == 0) {                                                 // f = the line as a dense element (f12_from_014)
            Fp2H v;
            if (role == 0) { ws_get(v, L[LP3_L0], lane); ws_put(L[LP3_F + 0], lane, v); ws_get(v, L[LP3_L1], lane); ws_put(L[LP3_F + 1], lane, v); ws_put(L[LP3_F + 2], lane, z); }
            else if (role == 1) { ws_put(L[LP3_F + 3], lane, z); ws_get(v, L[LP3_L4], lane); ws_put(L[LP3_F + 4], lane, v); ws_put(L[LP3_F + 5], lane, z); }
            __syncthreads();
            continue;
        }
        Fp6T<Fp2H> m;
        if (role == 2) {
            Fp6T<Fp2H> b, bb; Fp2H l4;
            ws_get(b.c0, L[LP3_F + 3], lane); ws_get(b.c1, L[LP3_F + 4], lane); ws_get(b.c2, L[LP3_F + 5], lane); ws_get(l4, L[LP3_L4], lane);
            f6_mul_by_1(bb, b, l4);
            ws_put(L[LP3_BB + 0], lane, bb.c0); ws_put(L[LP3_BB + 1], lane, bb.c1); ws_put(L[LP3_BB + 2], lane, bb.c2);
        } else {
            Fp6T<Fp2H> a; Fp2H l0, l1;
            ws_get(a.c0, L[LP3_F + 0], lane); ws_get(a.c1, L[LP3_F + 1], lane); ws_get(a.c2, L[LP3_F + 2], lane);
            ws_get(l0, L[LP3_L0], lane); ws_get(l1, L[LP3_L1], lane);
            if (role == 1) {
                Fp6T<Fp2H> b, sum; Fp2H l4, o;
                ws_get(b.c0, L[LP3_F + 3], lane); ws_get(b.c1, L[LP3_F + 4], lane); ws_get(b.c2, L[LP3_F + 5], lane); ws_get(l4, L[LP3_L4], lane);
                f2_add_n(o, l1, l4);
                f6_add_n(sum, a, b);
                a = sum; l1 = o;
            }
            f6_mul_by_01(m, a, l0, l1);                                // w0: aa   w1: m
            if (role == 0) { ws_put(L[LP3_AA + 0], lane, m.c0); ws_put(L[LP3_AA + 1], lane, m.c1); ws_put(L[LP3_AA + 2], lane, m.c2); }
        }
        __syncthreads();
        if (role == 0) {                                               // f.c0 = aa + v bb
            Fp6T<Fp2H> bb, x, r;
            ws_get(bb.c0, L[LP3_BB + 0], lane); ws_get(bb.c1, L[LP3_BB + 1], lane); ws_get(bb.c2, L[LP3_BB + 2], lane);
            f6_mul_v(x, bb);
            f6_add_n(r, m, x);
            ws_put(L[LP3_F + 0], lane, r.c0); ws_put(L[LP3_F + 1], lane, r.c1); ws_put(L[LP3_F + 2], lane, r.c2);
        } else if (role == 1) {                                        // f.c1 = m - aa - bb
            Fp6T<Fp2H> aa, bb, x; Fp2H r;
            ws_get(aa.c0, L[LP3_AA + 0], lane); ws_get(aa.c1, L[LP3_AA + 1], lane); ws_get(aa.c2, L[LP3_AA + 2], lane);
            ws_get(bb.c0, L[LP3_BB + 0], lane); ws_get(bb.c1, L[LP3_BB + 1], lane); ws_get(bb.c2, L[LP3_BB + 2], lane);
            fadd(x.c0, aa.c0, bb.c0); fadd(x.c1, aa.c1, bb.c1); fadd(x.c2, aa.c2, bb.c2);
            f2_sub_n<128>(r, m.c0, x.c0); ws_put(L[LP3_F + 3], lane, r);
            f2_sub_n<128>(r, m.c1, x.c1); ws_put(L[LP3_F + 4], lane, r);
            f2_sub_n<128>(r, m.c2, x.c2); ws_put(L[LP3_F + 5], lane, r);
        }
        __syncthreads();
    }
    if (inr && role < 2) {
        for (int c = 0; c < 3; c++) {
            Fp2H v; ws_get(v, L[LP3_F + 3 * role + c], lane);
            for (int k = 0; k < NL; k++) partial[(((size_t)g * N_LINES + s) * nsl + j) * F12W + (2 * (3 * role + c) + h) * NL + k] = v.v.l[k];
        }
    }
}

// One block per (step, group of 64 partials): tree product through LDS.  A node product a * b is shared by three lane PAIRS — Karatsuba
// over Fp6: a0 b0, a1 b1, (a0 + a1)(b0 + b1) are independent Fp6 products of 6 Fp2 products each, and every Fp2 value sits on a lane pair.
// The tree is latency-bound (a dense Fp12 product is ~25 k instructions on one lane) and has few nodes, so spreading a node over six lanes
// shortens every level ~5x.  Wave r of the 192-thread block takes role r of the nodes p = (lane >> 1); roles are wave-uniform.
//   A: operands of node p (slots p and p + h) -> registers            | sync
//   B: t = Fp6 product; t0 -> slot(p+h).c0, t1 -> slot(p+h).c1, m -> slot(p).c1    | sync
//   C: role 0: slot(p).c0 = t0 + v t1;  role 1: slot(p).c1 = m - t0 - t1           (f12_mul of pairing29.hip.h, step for step)
// out_abi != nullptr: the group result of step s is L_s, written in the ABI form (last level); otherwise it is written back as a
// partial of the next level: next[(s * ngroups + g) * F12W + k].
constexpr int F6W = 6 * NL;
// seg_off != nullptr (one level, ngroups == 1): s runs over (segment, step) and the number of partials is the segment's own slice count.
__global__ void __launch_bounds__(192) k_product_tree(const uint32_t *__restrict__ partial, int nsl, int ngroups, uint32_t *__restrict__ next, uint32_t *__restrict__ out_abi,
                                                      const uint32_t *__restrict__ seg_off, int slice_len, int s0 = 0, int ns = N_LINES) {
    __shared__ uint32_t sh[F12W * MAX_SLICES];                        // word k of slot j at sh[k * 64 + j]
    const int node = blockIdx.x / ngroups, grp = blockIdx.x % ngroups, t = threadIdx.x;
    const int s = seg_off ? (node / ns) * N_LINES + s0 + node % ns : s0 + node;
    int have = nsl;
    if (seg_off) { const int g = s / N_LINES; have = (int)((seg_off[g + 1] - seg_off[g] + slice_len - 1) / slice_len); }
    const int cnt = min(MAX_SLICES, have - grp * MAX_SLICES);       // partials in this group
    if (cnt <= 0) return;                                           // an empty segment: the host writes the neutral element
    if (t < cnt) { const uint32_t *src = partial + ((size_t)s * nsl + grp * MAX_SLICES + t) * F12W; for (int k = 0; k < F12W; k++) sh[k * MAX_SLICES + t] = src[k]; }
    const int role = t >> 6, p = (t & 63) >> 1;
    const uint32_t hh = t & 1u;
    auto ld6 = [&](Fp6p &x, int slot, int half) {
        Fp2H *c = reinterpret_cast<Fp2H *>(&x);
        for (int q = 0; q < 3; q++) for (int k = 0; k < NL; k++) c[q].v.l[k] = sh[(half * F6W + (2 * q + hh) * NL + k) * MAX_SLICES + slot];
    };
    auto st6 = [&](const Fp6p &x, int slot, int half) {
        const Fp2H *c = reinterpret_cast<const Fp2H *>(&x);
        for (int q = 0; q < 3; q++) for (int k = 0; k < NL; k++) sh[(half * F6W + (2 * q + hh) * NL + k) * MAX_SLICES + slot] = c[q].v.l[k];
    };
    for (int h = MAX_SLICES / 2; h >= 1; h >>= 1) {
        if (h >= cnt) continue;                                     // (uniform) nothing to fold at this level
        const bool node = p < h && p + h < cnt;                     // slot p *= slot p + h
        Fp6p xa, xb;
        __syncthreads();
        if (node) {
            ld6(xa, p, role == 1); ld6(xb, p + h, role == 1);
            if (role == 2) { Fp6p ya, yb; ld6(ya, p, 1); ld6(yb, p + h, 1); f6_add_n(xa, xa, ya); f6_add_n(xb, xb, yb); }
        }
        __syncthreads();
        if (node) {
            Fp6p tt; f6_mul(tt, xa, xb);
            if (role == 0) st6(tt, p + h, 0); else if (role == 1) st6(tt, p + h, 1); else st6(tt, p, 1);
        }
        __syncthreads();
        if (node && role == 0) {                                    // c0 = t0 + v t1
            Fp6p t0, t1, x, r; ld6(t0, p + h, 0); ld6(t1, p + h, 1);
            f6_mul_v(x, t1); f6_add_n(r, t0, x); st6(r, p, 0);
        } else if (node && role == 1) {                             // c1 = m - t0 - t1
            Fp6p t0, t1, m, x, r; ld6(t0, p + h, 0); ld6(t1, p + h, 1); ld6(m, p, 1);
            fadd(x.c0, t0.c0, t1.c0); fadd(x.c1, t0.c1, t1.c1); fadd(x.c2, t0.c2, t1.c2);
            f2_sub_n<128>(r.c0, m.c0, x.c0); f2_sub_n<128>(r.c1, m.c1, x.c1); f2_sub_n<128>(r.c2, m.c2, x.c2);
            st6(r, p, 1);
        }
    }
    __syncthreads();
    if (t < 2) {                                                    // the pair of slot 0 writes the result, each lane its halves
        for (int q = 0; q < 6; q++) {
            Fp c;
            for (int k = 0; k < NL; k++) c.l[k] = sh[((2 * q + hh) * NL + k) * MAX_SLICES];
            if (out_abi) fp_to_abi(out_abi + ((size_t)s * 12 + 2 * q + hh) * 12, c);
            else for (int k = 0; k < NL; k++) next[((size_t)s * ngroups + grp) * F12W + (2 * q + hh) * NL + k] = c.l[k];
        }
    }
}

// ---- the same tree with 18 lane pairs per node ---------------------------------------------------------------------------------------
// A level of k_product_tree lasts ~32 us: six Fp2 products one after another on each of the node's three lane pairs, between three
// barriers.  Here lane pair r of a node computes ONE role product of pairing29.hip.h's regrouped Fp12 product (operands summed straight out
// of the factors' slots, product parked in LDS), and after a barrier lane pairs 0..8 turn the parked products into the nine Fp6 coefficients of
// the three Fp6 products, lane pairs 0..5 those into one output coefficient each: a level is one Fp2 product and two short combinations deep.  The instruction stream is the same for every role (operand
// and product indices are data: masked loads, selects), so a wave may hold any mixture of roles and nodes.  16 nodes per pass (576
// threads); the 32 nodes of a full group's first level take two passes.  Same interface and results as k_product_tree.
constexpr int T18_NODES = 16, T18_PAIRS = T18_NODES * 18, T18_THREADS = 2 * T18_PAIRS;
constexpr int T18_SLOT = F12W + 1;        // slot-major with an odd stride: the lanes of a node read DIFFERENT words of the SAME slot (word-major put them all in one bank)
constexpr size_t T18_LDS = ((size_t)T18_SLOT * MAX_SLICES + (size_t)NL * T18_THREADS) * 4;
__global__ void __launch_bounds__(T18_THREADS) k_product_tree18(const uint32_t *__restrict__ partial, int nsl, int ngroups, uint32_t *__restrict__ next, uint32_t *__restrict__ out_abi,
                                                                const uint32_t *__restrict__ seg_off, int slice_len, int s0 = 0, int ns = N_LINES) {
    extern __shared__ uint32_t lds18[];
    uint32_t *sh = lds18;                                            // word k of slot j at sh[j * T18_SLOT + k]
    uint32_t *pr = lds18 + T18_SLOT * MAX_SLICES;                        // word j of this pass's product of lane t at pr[j * T18_THREADS + t]
    const int node = blockIdx.x / ngroups, grp = blockIdx.x % ngroups, t = threadIdx.x;
    const int s = seg_off ? (node / ns) * N_LINES + s0 + node % ns : s0 + node;      // segments: steps s0 .. s0 + ns - 1 of every segment
    int have = nsl;
    if (seg_off) { const int g = s / N_LINES; have = (int)((seg_off[g + 1] - seg_off[g] + slice_len - 1) / slice_len); }
    const int cnt = min(MAX_SLICES, have - grp * MAX_SLICES);
    if (cnt <= 0) return;
    { const uint32_t *src = partial + ((size_t)s * nsl + grp * MAX_SLICES) * F12W;
      for (int e = t; e < cnt * F12W; e += T18_THREADS) { const int slot = e / F12W, k = e % F12W; sh[slot * T18_SLOT + k] = src[e]; } }
    const uint32_t hh = t & 1u;
    const int pi = t >> 1, nl = pi / 18, r = pi % 18, R = r / 6, k = r % 6;
    auto ldc = [&](Fp &v, int slot, int c) {
#pragma unroll
        for (int j = 0; j < NL; j++) v.l[j] = sh[slot * T18_SLOT + (2 * c + hh) * NL + j];
    };
    // operand of role (R, k) from the coefficients in `slot`: f12_role_operand with the optional terms masked instead of branched over
    auto operand = [&](Fp2H &x, int slot) {
        const int f = role_first(k), sc = role_second(k), base = (R == 1) ? 3 : 0;
        const bool two = sc >= 0, both = (R == 2);
        const int s2 = two ? sc : f;
        Fp a0, a1, a2, a3;
        ldc(a0, slot, base + f); ldc(a1, slot, base + s2); ldc(a2, slot, 3 + f); ldc(a3, slot, 3 + s2);
        Fp tsum;
#pragma unroll
        for (int j = 0; j < NL; j++) tsum.l[j] = a0.l[j] + (two ? a1.l[j] : 0u) + (both ? a2.l[j] : 0u) + ((both && two) ? a3.l[j] : 0u);
        fp_norm(x.v, tsum);
    };
    auto ldp = [&](Fp2H &v, int rr) {
        const int at = (nl * 18 + rr) * 2 + (int)hh;
#pragma unroll
        for (int j = 0; j < NL; j++) v.v.l[j] = pr[j * T18_THREADS + at];
    };
    auto zcoef = [&](Fp2H &z, int RR, int kk) {
        Fp2H Q, A, B, C;
        ldp(Q, 6 * RR + role_q(kk)); ldp(A, 6 * RR + role_a(kk)); ldp(B, 6 * RR + role_b(kk)); ldp(C, 6 * RR + role_c(kk));
        f6_coeff_from_roles(z, Q, A, B, C, kk);
    };
    for (int hs = MAX_SLICES / 2; hs >= 1; hs >>= 1) {
        if (hs >= cnt) continue;                                    // (uniform) nothing to fold at this level
        for (int base = 0; base < hs; base += T18_NODES) {
            const int p = base + nl;
            const bool node = p < hs && p + hs < cnt;               // slot p *= slot p + hs
            __syncthreads();
            if (node) {
                Fp2H x, y, m;
                operand(x, p); operand(y, p + hs);
                fmul(m, x, y);
#pragma unroll
                for (int j = 0; j < NL; j++) pr[j * T18_THREADS + t] = m.v.l[j];
            }
            __syncthreads();
            // the nine Fp6 coefficients z(R, k) of the three Fp6 products, one per lane pair (r = 3 R + k), parked where the products were
            Fp2H z;
            if (node && r < 9) zcoef(z, r / 3, r % 3);
            __syncthreads();
            if (node && r < 9) {
#pragma unroll
                for (int j = 0; j < NL; j++) pr[j * T18_THREADS + t] = z.v.l[j];
            }
            __syncthreads();
            if (node && r < 6) {                                   // the six Fp2 coefficients of the product
                const int q = r; const bool c0t = q < 3; const int kk = c0t ? q : q - 3;
                Fp2H U, V, W, o0, o1, o;
                ldp(U, c0t ? kk : 6 + kk); ldp(V, c0t ? 3 + (q + 2) % 3 : kk); ldp(W, 3 + kk);      // z(R, k) sits in lane pair 3 R + k's words
                f12_out_c0(o0, U, V, q); f12_out_c1(o1, U, V, W);
                fsel(o, c0t, o0, o1);
#pragma unroll
                for (int j = 0; j < NL; j++) sh[p * T18_SLOT + (2 * q + hh) * NL + j] = o.v.l[j];
            }
        }
    }
    __syncthreads();
    if (t < 12) {                                                   // twelve lanes, a coefficient half each (the conversion to the ABI form is a product and a canonical
        const int q = t >> 1;                                         // reduction: one lane pair doing all six behind each other was a third of a two-pass launch)
        Fp c;
        for (int j = 0; j < NL; j++) c.l[j] = sh[(2 * q + hh) * NL + j];
        if (out_abi) fp_to_abi(out_abi + ((size_t)s * 12 + 2 * q + hh) * 12, c);
        else for (int j = 0; j < NL; j++) next[((size_t)s * ngroups + grp) * F12W + (2 * q + hh) * NL + j] = c.l[j];
    }
}
// one launcher for both tree kernels (gs.ml_mode bit 1: the 18-role form)
static void launch_product_tree(hipStream_t st, unsigned blocks, const uint32_t *partial, int nsl, int ngroups, uint32_t *next, uint32_t *out_abi, const uint32_t *seg_off, int slice_len, int s0, int ns = N_LINES) {
    if (gs.ml_mode.load() & 2) {
        static std::atomic<uint32_t> done{0};
        { int dev = 0; (void)hipGetDevice(&dev); const uint32_t bit = 1u << (dev & 31);
          if (!(done.load() & bit)) { (void)hipFuncSetAttribute((const void *)k_product_tree18, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T18_LDS); done.fetch_or(bit); } }
        hipLaunchKernelGGL(k_product_tree18, dim3(blocks), dim3(T18_THREADS), T18_LDS, st, partial, nsl, ngroups, next, out_abi, seg_off, slice_len, s0, ns);
    } else
        hipLaunchKernelGGL(k_product_tree, dim3(blocks), dim3(192), 0, st, partial, nsl, ngroups, next, out_abi, seg_off, slice_len, s0, ns);
}

// Slices of pairs per step.  A lane multiplies its slice's lines into one partial (sparse products, serial), then 64-wide trees fold
// the partials (dense products, log depth): short slices keep both latency-bound phases short at small n and fill the chip at large n
// (a fixed 64 slices left k_line_products with 4352 lanes whatever n: 59 ms at 2^16 pairs).
inline int choose_slice_len(size_t n) {
    size_t len = n > 2048 ? 8 : (n > 512 ? 4 : 2);      // (measured with the 18-role tree: 2 wins up to 512 pairs, 4 at 1024, 8 from 4096)
    while ((n + len - 1) / len > 2048 && (n + len - 1) / len > 0) len *= 2;
    return (int)len;
}

}  // namespace

extern "C" {

// conj((...((L_0)^2 L_1)^2 ...)) over the 68 per-step products; may be taken in pieces (bits 62 .. b_lo, then on to 0)
struct MlTail {
    hostf::Fq12 f = hostf::Fq12::one(); int idx = 0, b = 62;
    void run(const hostf::Fq12 *L, int b_lo) { for (; b >= b_lo; b--) { f = f.sqr() * L[idx++]; if ((hostf::BLS_X_ABS >> b) & 1) f = f * L[idx++]; } }
    hostf::Fq12 result() const { return f.conj(); }      // x < 0
};
static hostf::Fq12 ml_host_tail(const hostf::Fq12 *L) { MlTail t; t.run(L, 0); return t.result(); }

struct MlGeom { int slice_len, nsl, ngroups; size_t base; };         // base: first word of this geometry's partials in sl.ml_partial
static size_t ml_geom_words(const MlGeom &g) { return (size_t)N_LINES * (g.nsl + g.ngroups) * F12W; }
static void ml_geom_set(MlGeom &g, size_t n, int slice_len, size_t base) {
    g.slice_len = slice_len;
    g.nsl = (int)((n + slice_len - 1) / slice_len);                  // <= 2048
    g.ngroups = (g.nsl + MAX_SLICES - 1) / MAX_SLICES;               // <= 32: the second tree level is one group
    g.base = base;
}
static int32_t ml_geometry(Slot &sl, size_t n, MlGeom &g) {
    int32_t rc;
    ml_geom_set(g, n, choose_slice_len(n), 0);
    if ((rc = sl.ml_partial.ensure(ml_geom_words(g) * 4))) return rc;
    if ((rc = sl.ml_out.ensure((size_t)N_LINES * 144 * 4))) return rc;
    return DGPU_OK;
}
// K10 + K11 for the steps s0 .. s0 + ns - 1 on stream s (partials and results are indexed by the step: disjoint for disjoint ranges)
constexpr unsigned LP3_MAX_BLOCKS = 512;      // two blocks of three waves per CU
static void ml_products(Slot &sl, hipStream_t s, size_t n, const MlGeom &g, int s0, int ns, bool timed, const uint32_t *pxy = nullptr) {
    const int nsl = g.nsl, ngroups = g.ngroups;
    auto products = [&] {
      const unsigned blocks3 = (unsigned)((ns * nsl + 31) / 32);
      const int mlm = gs.ml_mode.load();
      if (blocks3 <= (LP3_MAX_BLOCKS << ((mlm >> 28) & 3)) && (mlm & 16))
          hipLaunchKernelGGL(k_line_products3, dim3(blocks3), dim3(192), 0, s, sl.ml_lines.as<uint32_t>(), n, g.slice_len, nsl, sl.ml_partial.as<uint32_t>() + g.base, s0, ns, pxy);
      else
          hipLaunchKernelGGL(k_line_products, dim3((unsigned)((2 * ns * nsl + 63) / 64)), dim3(64), 0, s, sl.ml_lines.as<uint32_t>(), n, g.slice_len, nsl, sl.ml_partial.as<uint32_t>() + g.base, (const uint32_t *)nullptr, 1, s0, ns, pxy); };
    auto tree = [&] {
      uint32_t *lvl0 = sl.ml_partial.as<uint32_t>() + g.base, *lvl1 = lvl0 + (size_t)N_LINES * nsl * F12W;
      if (ngroups == 1) launch_product_tree(s, (unsigned)ns, lvl0, nsl, 1, (uint32_t *)nullptr, sl.ml_out.as<uint32_t>(), (const uint32_t *)nullptr, 0, s0);
      else {
          launch_product_tree(s, (unsigned)(ns * ngroups), lvl0, nsl, ngroups, lvl1, (uint32_t *)nullptr, (const uint32_t *)nullptr, 0, s0);
          launch_product_tree(s, (unsigned)ns, lvl1, ngroups, 1, (uint32_t *)nullptr, sl.ml_out.as<uint32_t>(), (const uint32_t *)nullptr, 0, s0);
      } };
    if (timed) { { StageTimer st(sl, "ml.products"); products(); } { StageTimer st(sl, "ml.tree"); tree(); } }      // (stage timers record on sl.stream)
    else { products(); tree(); }
}
// K10 + K11 + host tail on the lines already in sl.ml_lines
static int32_t ml_finish(Slot &sl, size_t n, uint64_t *out, const uint32_t *pxy = nullptr) {
    int32_t rc; MlGeom g;
    if ((rc = ml_geometry(sl, n, g))) return rc;
    hipStream_t s = sl.stream;
    ml_products(sl, s, n, g, 0, N_LINES, true, pxy);
    HIPCHK(hipGetLastError());
    std::vector<hostf::Fq12> L(N_LINES);
    HIPCHK(hipMemcpyAsync(L.data(), sl.ml_out.p, (size_t)N_LINES * 576, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (gs.prof) prof_flush(sl);
    const hostf::Fq12 f = ml_host_tail(L.data());
    memcpy(out, &f, sizeof f);
    return DGPU_OK;
}

// A call of up to 8192 pairs lasts as long as its chain: 68 dependent line steps (K9: the chip nearly empty), then the product levels of
// K10 / K11 (~0.18 ms whatever the number of steps), then 131 Fp12 operations on the host (~0.25 ms: as long as the chain itself since K9
// has sixteen lanes per pair).  The chain is therefore cut into ML_PIECES launches at the bits ML_CUTS of |x|: the products of a finished
// piece run on a stream of their own while the next piece is computed, its results land in pinned memory and the calling thread folds
// them into f as they arrive — what is left after the last launch is the product levels of the last piece and its share of the host's
// work.  Same values in the same order: bit-identical to the one-launch form, which stays for larger batches (throughput-bound) and
// while stage timers are on.  (Round 3: two pieces, cut at bit 17; 1.45 -> 1.05 ms at 1024 pairs.  Three pieces with sixteen lanes: 0.7.)
constexpr int ML_PIECES = 3;
constexpr int ML_CUTS[ML_PIECES] = {40, 17, 0};               // piece j runs the bits (ML_CUTS[j - 1] - 1, or 62) .. ML_CUTS[j]
// n pairs in the line buffer, the first n_aff of them affine (their chain is what gets cut); prepared(pxy) queues the line kernel of the
// prepared pairs n_aff .. n - 1, if any, on the slot's stream (it writes their neutral px, py)
// late(side_stream, pxy), if given, runs on the calling thread once every piece of the chain has been queued — host work that the chain hides
// (the verifier computes its third G1 operand meanwhile) — and may queue, on side_stream, whatever must precede the product kernels (the line
// kernel of prepared pairs whose P was not known before).
// p_late: the affine pairs' P are not known when the chain starts (dgpu_multi_miller_loop_scaled: they are being scaled meanwhile) — the line kernel gets
// no P and leaves px, py alone; `late` must write them (pxy[(c NL + k) n + i], the internal form) before the product kernels run
static int32_t ml_pipelined(Slot &sl, size_t n, size_t n_aff, const uint8_t *dskip, uint64_t *out, const std::function<void(uint32_t *)> &prepared,
                            const std::function<int32_t(hipStream_t, uint32_t *)> &late = nullptr, bool p_late = false) {
    int32_t rc; MlGeom g, g2;
    if ((rc = ml_geometry(sl, n, g))) return rc;
    // the last steps' products are all that is left when the chain ends: from 8-pair slices on they take half the slice length (fewer
    // sparse products in front of the tree, more tree for a quarter of the steps: 1.46 -> 1.39 ms at 4096 pairs; measured the other way
    // round at 1024 pairs, 4 -> 2: 1.06 -> 1.14)
    int tail_slice = g.slice_len >= 8 && (n + g.slice_len / 2 - 1) / (g.slice_len / 2) <= 2048 ? g.slice_len / 2 : g.slice_len;
    int cuts[ML_PIECES]; for (int j = 0; j < ML_PIECES; j++) cuts[j] = ML_CUTS[j];
    { const int m = gs.ml_mode.load();                               // development twin: dgpu_set_miller_pipeline's upper bits (cuts, slice length of the last piece)
      const int a = (m >> 8) & 63, b = (m >> 16) & 63, v = (m >> 24) & 15;
      if (a > b && b > 0 && a < 62) { cuts[0] = a; cuts[1] = b; }
      if (v >= 1 && (n + v - 1) / v <= 2048) tail_slice = v; }
    ml_geom_set(g2, n, tail_slice, ml_geom_words(g));
    if ((rc = sl.ml_partial.ensure((ml_geom_words(g) + ml_geom_words(g2)) * 4))) return rc;
    if ((rc = sl.ml_state.ensure(((size_t)3 * NL * 4 * n_aff + (size_t)2 * NL * n) * 4))) return rc;      // R of every lane, then px, py of every pair
    hipStream_t sa = sl.stream;
    hipStream_t side[2] = {sl.cstream, sl.xstream};                  // the pieces' products alternate between them (a piece's products outlast the next piece's chain)
    static_assert((size_t)N_LINES * 576 <= Slot::HPIN_BYTES, "pinned scratch");
    static_assert(2 * (ML_PIECES - 1) + 1 <= Slot::N_COPY_EV + 1, "events");
    hostf::Fq12 *L = (hostf::Fq12 *)sl.hpin;                          // pinned: the copies below are asynchronous for the host
    uint32_t *state = sl.ml_state.as<uint32_t>(), *pxy = state + (size_t)3 * NL * 4 * n_aff;
    // (a failed enqueue must not leave the slot with work in flight: every step is checked, every stream is drained before any return)
    rc = DGPU_OK;
    auto ok = [&](hipError_t e) { if (e != hipSuccess && !rc) rc = DGPU_E_HIP; return rc == DGPU_OK; };
    prepared(pxy);
    hipEvent_t done[ML_PIECES - 1] = {}, ready[ML_PIECES - 1] = {};
    int first_step[ML_PIECES], steps[ML_PIECES];
    { int s_first = 0, b_hi = 62;                                        // the whole chain first: its launches depend on nothing but each other
      for (int j = 0; j < ML_PIECES && !rc; j++) {
          const int b_lo = cuts[j], ns = ml_steps(b_hi, b_lo);
          first_step[j] = s_first; steps[j] = ns;
          launch_lines_uneval(sa, p_late ? (const uint32_t *)nullptr : sl.in_bases.as<uint32_t>(), sl.in_scalars.as<uint32_t>(), dskip, n_aff, sl.ml_lines.as<uint32_t>(), n, b_hi, b_lo, s_first, state,
                              p_late ? (uint32_t *)nullptr : pxy);
          if (j + 1 < ML_PIECES) { ready[j] = sl.copy_ev[sl.ev_next++ % (Slot::N_COPY_EV + 1)]; ok(hipEventRecord(ready[j], sa)); }
          s_first += ns; b_hi = b_lo - 1;
      } }
    hipEvent_t late_done = nullptr;
    if (late && !rc) {
        const int32_t lrc = late(side[0], pxy);
        if (lrc && !rc) rc = lrc;
        late_done = sl.copy_ev[sl.ev_next++ % (Slot::N_COPY_EV + 1)];
        if (!rc) { ok(hipEventRecord(late_done, side[0])); ok(hipStreamWaitEvent(side[1], late_done, 0)); ok(hipStreamWaitEvent(sa, late_done, 0)); }
    }
    for (int j = 0; j < ML_PIECES && !rc; j++) {
        const int s_first = first_step[j], ns = steps[j];
        if (j + 1 < ML_PIECES) {
            done[j] = sl.copy_ev[sl.ev_next++ % (Slot::N_COPY_EV + 1)];
            hipStream_t sp = side[j & 1];
            if (ok(hipStreamWaitEvent(sp, ready[j], 0))) {
                ml_products(sl, sp, n, g, s_first, ns, false, pxy);
                ok(hipMemcpyAsync(L + s_first, (const char *)sl.ml_out.p + (size_t)s_first * 576, (size_t)ns * 576, hipMemcpyDeviceToHost, sp));
                ok(hipEventRecord(done[j], sp));
            }
        } else {
            ml_products(sl, sa, n, g2, s_first, ns, false, pxy);
            ok(hipMemcpyAsync(L + s_first, (const char *)sl.ml_out.p + (size_t)s_first * 576, (size_t)ns * 576, hipMemcpyDeviceToHost, sa));
        }
    }
    ok(hipGetLastError());
    MlTail tail;
    for (int j = 0; j + 1 < ML_PIECES; j++) if (!rc && done[j] && ok(hipEventSynchronize(done[j]))) tail.run(L, cuts[j]);
    ok(hipStreamSynchronize(sa)); ok(hipStreamSynchronize(side[0])); ok(hipStreamSynchronize(side[1]));
    if (rc) return rc;
    tail.run(L, 0);
    const hostf::Fq12 f = tail.result();
    memcpy(out, &f, sizeof f);
    return DGPU_OK;
}
// how many Miller loops are in flight on the context (the two-launch form needs the slot's second stream and an event wait between the
// two; with more streams than hardware queues a waiting stream holds up whatever shares its queue — measured, 1024 pairs: 0.62 vs 0.70 ms
// per call with two calls in flight, but 0.49 vs 0.36 with six — so it is taken while at most two are in flight)
struct MlActive { std::atomic<int> &c; int v; explicit MlActive(std::atomic<int> &c_) : c(c_), v(++c_) {} ~MlActive() { --c; } MlActive(const MlActive &) = delete; };

int32_t dgpu_multi_miller_loop(const uint64_t *p, const uint64_t *q, const uint8_t *skip, size_t n, uint64_t *out) {
    if (!out || (n && (!p || !q))) return DGPU_E_BADARG;
    if (n == 0) { hostf::Fq12 one = hostf::Fq12::one(); memcpy(out, &one, sizeof one); return DGPU_OK; }
    if (n >= (1ull << 24)) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    SLOT_ACQUIRE(slot_lock, sl);
    HIPCHK(hipSetDevice(cur().device));
    int32_t rc;
    if ((rc = sl.in_bases.ensure(n * 96))) return rc;
    if ((rc = sl.in_scalars.ensure(n * 192))) return rc;
    if ((rc = sl.in_inf.ensure(n))) return rc;
    if ((rc = sl.ml_lines.ensure((size_t)N_LINES * LW * n * 4))) return rc;
    hipStream_t s = sl.stream;
    HIPCHK(hipMemcpyAsync(sl.in_bases.p, p, n * 96, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(sl.in_scalars.p, q, n * 192, hipMemcpyHostToDevice, s));
    const uint8_t *dskip = nullptr;
    if (skip) { HIPCHK(hipMemcpyAsync(sl.in_inf.p, skip, n, hipMemcpyHostToDevice, s)); dskip = sl.in_inf.as<uint8_t>(); }
    { StageTimer st(sl, "ml.lines");
#ifdef DGPU_DEV
      static const bool one_lane = getenv("DGPU_ML_ONE_LANE") != nullptr;     // development switches: one lane / one lane pair per (P, Q)
      static const bool two_lanes = getenv("DGPU_ML_TWO_LANES") != nullptr;
#else
      constexpr bool one_lane = false, two_lanes = false;
#endif
      if (!one_lane && !two_lanes && n <= 8192 && !gs.prof && (gs.ml_mode.load() & 1)) {
          // Either way the evaluation at P is left to the product kernel.
          MlActive act(cur().ml_active);
          if (act.v <= 2) return ml_pipelined(sl, n, n, dskip, out, [](uint32_t *) {});
          if ((rc = sl.ml_state.ensure((size_t)2 * NL * n * 4))) return rc;
          uint32_t *pxy = sl.ml_state.as<uint32_t>();
          launch_lines_uneval(s, sl.in_bases.as<uint32_t>(), sl.in_scalars.as<uint32_t>(), dskip, n, sl.ml_lines.as<uint32_t>(), n,
                             62, 0, 0, (uint32_t *)nullptr, pxy);
          return ml_finish(sl, n, out, pxy);
      }
      if (one_lane) hipLaunchKernelGGL(k_miller_lines, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, sl.in_bases.as<uint32_t>(), sl.in_scalars.as<uint32_t>(), dskip, n, sl.ml_lines.as<uint32_t>(), n);
      else if (two_lanes || n > 8192) hipLaunchKernelGGL(k_miller_lines_pair,     // (with the chip full, the pair form does less total work)
              dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, s, sl.in_bases.as<uint32_t>(), sl.in_scalars.as<uint32_t>(), dskip, n, sl.ml_lines.as<uint32_t>(), n);
      else hipLaunchKernelGGL(k_miller_lines_quad<true>, dim3((unsigned)((4 * n + 63) / 64)), dim3(64), 0, s, sl.in_bases.as<uint32_t>(), sl.in_scalars.as<uint32_t>(), dskip, n, sl.ml_lines.as<uint32_t>(), n);
    }
    return ml_finish(sl, n, out);
}

// nseg independent Miller loops in one call: segment g is the pairs [seg_end[g - 1], seg_end[g]) (seg_end ascending, seg_end[nseg - 1] == n;
// an empty segment yields one).  The ten `E::multi_pairing` calls of a GIPA round (legogroth16/src/aggregation/commitment.rs:30-31,54-67,
// aggregation/utils.rs:95-96, issued one after another in the reference) have 1 ... n/2 pairs each, and every launch of the line kernel lasts as
// long as its 68 dependent steps whatever the pair count: one line launch over all the pairs, one product launch and one tree launch over
// (segment, step), the nseg host tails on host threads.  out_f12: nseg x 72 words, each limb for limb what dgpu_multi_miller_loop
// returns for that segment alone.
static int32_t ml_segments(const uint64_t *p, const uint64_t *q, const uint8_t *skip, size_t n, const uint64_t *seg_end, size_t nseg, uint64_t *out, bool final_exp) {
    if (!out || !seg_end || nseg == 0 || (n && (!p || !q)) || n >= (1ull << 24) || nseg > 4096) return DGPU_E_BADARG;
    size_t maxlen = 0;
    { uint64_t prev = 0; for (size_t g = 0; g < nseg; g++) { if (seg_end[g] < prev || seg_end[g] > n) return DGPU_E_BADARG; maxlen = std::max<size_t>(maxlen, seg_end[g] - prev); prev = seg_end[g]; }
      if (prev != n) return DGPU_E_BADARG; }
    // (a Miller output of valid operands is never zero; arkworks' multi_pairing unwraps the Option — here a zero is DGPU_E_ZERO for the call)
    std::atomic<bool> zero{false};
    auto finish = [&](uint64_t *o) {
        if (!final_exp) return;
        hostf::Fq12 f, r; memcpy(&f, o, sizeof f);
        if (!hostf::final_exponentiation(r, f)) { zero = true; return; }
        memcpy(o, &r, sizeof r);
    };
    if (nseg == 1 || maxlen > 8192) {                      // long segments fill the chip on their own: one call each
        uint64_t prev = 0;
        for (size_t g = 0; g < nseg; g++) {
            const int32_t rc = dgpu_multi_miller_loop(p + prev * 12, q + prev * 24, skip ? skip + prev : nullptr, seg_end[g] - prev, out + g * 72);
            if (rc) return rc;
            finish(out + g * 72);
            prev = seg_end[g];
        }
        return zero ? DGPU_E_ZERO : DGPU_OK;
    }
    const hostf::Fq12 one = hostf::Fq12::one();
    if (n == 0) { for (size_t g = 0; g < nseg; g++) memcpy(out + g * 72, &one, sizeof one); return DGPU_OK; }
    if (!cur().ready) return DGPU_E_NODEVICE;
    SLOT_ACQUIRE(slot_lock, sl);
    HIPCHK(hipSetDevice(cur().device));
    int32_t rc;
    // Slices of a (segment, step): a lane pair multiplies its slice's lines one after the other, trees fold the slices.  Up to 64 slices of 4 (8) pairs fold in
    // one tree level; a longer segment keeps slices of 8 and takes a second level (<= 8192 pairs per segment here: <= 1024 slices, 16 groups) — with one level
    // its slices grew to 16 / 32 pairs, i.e. a quarter of the lanes on chains four times as long (the commitments of a 1024-proof aggregation: 2 x 2048 + 4 x 1024
    // pairs on 276 waves per piece).
    int slice_len = 4;
    if (maxlen > 4 * (size_t)MAX_SLICES) slice_len = 8;
    while ((maxlen + slice_len - 1) / slice_len > (size_t)MAX_SLICES * MAX_SLICES) slice_len *= 2;
    const int nsl = (int)((maxlen + slice_len - 1) / slice_len), ngroups = (nsl + MAX_SLICES - 1) / MAX_SLICES;
    // (K11 over the partials of steps s0 .. s0 + ns - 1 of every segment, one or two levels; the second level sees the groups as slices of 64 * slice_len pairs)
    auto seg_tree = [&](hipStream_t st, uint32_t *lvl0, const uint32_t *doff_, int s0, int ns) {
        if (ngroups == 1) { launch_product_tree(st, (unsigned)(ns * nseg), lvl0, nsl, 1, (uint32_t *)nullptr, sl.ml_out.as<uint32_t>(), doff_, slice_len, s0, ns); return; }
        uint32_t *lvl1 = lvl0 + (size_t)N_LINES * nseg * nsl * F12W;
        launch_product_tree(st, (unsigned)(ns * nseg * ngroups), lvl0, nsl, ngroups, lvl1, (uint32_t *)nullptr, doff_, slice_len, s0, ns);
        launch_product_tree(st, (unsigned)(ns * nseg), lvl1, ngroups, 1, (uint32_t *)nullptr, sl.ml_out.as<uint32_t>(), doff_, slice_len * MAX_SLICES, s0, ns);
    };
    std::vector<uint32_t> off(nseg + 1, 0);
    for (size_t g = 0; g < nseg; g++) off[g + 1] = (uint32_t)seg_end[g];
    if ((rc = sl.in_bases.ensure(n * 96))) return rc;
    if ((rc = sl.in_scalars.ensure(n * 192))) return rc;
    if ((rc = sl.in_inf.ensure(n + (nseg + 1) * 4 + 8))) return rc;
    if ((rc = sl.ml_lines.ensure((size_t)N_LINES * LW * n * 4))) return rc;
    if ((rc = sl.ml_partial.ensure((size_t)N_LINES * nseg * (nsl + (ngroups > 1 ? ngroups : 0)) * F12W * 4))) return rc;
    if ((rc = sl.ml_out.ensure((size_t)N_LINES * nseg * 144 * 4))) return rc;
    // pieces: like ml_pipelined, the chain runs in ML_PIECES launches; the products and trees of a finished piece (every segment's) run on a side stream
    // under the next piece, its results land in pinned memory, and the host tails of the segments advance piece by piece on the library's threads —
    // what is left when the chain ends is the last piece's share of the tail and the final exponentiations (a GIPA round: 1.8 -> 1.5 ms)
    const size_t pin_bytes = (size_t)N_LINES * nseg * 576;
    const bool pieces = n <= 8192 && (gs.ml_mode.load() & 1) && pin_bytes <= ((size_t)4 << 20);
    if ((rc = sl.ml_state.ensure(((size_t)(pieces ? 3 * NL * 4 : 0) * n + (size_t)2 * NL * n) * 4))) return rc;      // (R of every lane,) px, py of every pair for the product kernel
    uint32_t *state = sl.ml_state.as<uint32_t>(), *pxy = state + (size_t)(pieces ? 3 * NL * 4 : 0) * n;
    if (pieces && sl.hpin2_bytes < pin_bytes) {
        if (sl.hpin2) { (void)hipHostFree(sl.hpin2); sl.hpin2 = nullptr; sl.hpin2_bytes = 0; }
        if (hipHostMalloc(&sl.hpin2, pin_bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); sl.hpin2 = nullptr; return DGPU_E_OOM; }
        sl.hpin2_bytes = pin_bytes;
    }
    hipStream_t s = sl.stream;
    HIPCHK(hipMemcpyAsync(sl.in_bases.p, p, n * 96, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(sl.in_scalars.p, q, n * 192, hipMemcpyHostToDevice, s));
    const uint8_t *dskip = nullptr;
    if (skip) { HIPCHK(hipMemcpyAsync(sl.in_inf.p, skip, n, hipMemcpyHostToDevice, s)); dskip = sl.in_inf.as<uint8_t>(); }
    uint32_t *doff = (uint32_t *)(sl.in_inf.as<uint8_t>() + ((n + 7) & ~(size_t)7));
    HIPCHK(hipMemcpyAsync(doff, off.data(), (nseg + 1) * 4, hipMemcpyHostToDevice, s));
    if (pieces) {
        hipStream_t sa = s, side[2] = {sl.cstream, sl.xstream};
        hostf::Fq12 *Lp = (hostf::Fq12 *)sl.hpin2;                          // [segment][step]
        rc = DGPU_OK;
        auto ok = [&](hipError_t e) { if (e != hipSuccess && !rc) rc = DGPU_E_HIP; return rc == DGPU_OK; };
        hipEvent_t ready[ML_PIECES] = {}, done[ML_PIECES] = {};
        int first_step[ML_PIECES], steps[ML_PIECES];
        { int s_first = 0, b_hi = 62;
          for (int j = 0; j < ML_PIECES && !rc; j++) {
              const int b_lo = ML_CUTS[j], ns = ml_steps(b_hi, b_lo);
              first_step[j] = s_first; steps[j] = ns;
              launch_lines_uneval(sa, sl.in_bases.as<uint32_t>(), sl.in_scalars.as<uint32_t>(), dskip, n, sl.ml_lines.as<uint32_t>(), n, b_hi, b_lo, s_first, state, pxy);
              if (j + 1 < ML_PIECES) { ready[j] = sl.copy_ev[sl.ev_next++ % (Slot::N_COPY_EV + 1)]; ok(hipEventRecord(ready[j], sa)); }
              s_first += ns; b_hi = b_lo - 1;
          } }
        for (int j = 0; j < ML_PIECES && !rc; j++) {
            hipStream_t sp = j + 1 < ML_PIECES ? side[j & 1] : sa;
            if (j + 1 < ML_PIECES && !ok(hipStreamWaitEvent(sp, ready[j], 0))) break;
            const int s0 = first_step[j], ns = steps[j];
            const size_t blocks3 = ((size_t)ns * nsl * nseg + 31) / 32;
            const int mlm = gs.ml_mode.load();
            if (blocks3 <= ((size_t)LP3_MAX_BLOCKS << ((mlm >> 28) & 3)) && (mlm & 16))
                hipLaunchKernelGGL(k_line_products3, dim3((unsigned)blocks3), dim3(192), 0, sp, sl.ml_lines.as<uint32_t>(), n, slice_len, nsl, sl.ml_partial.as<uint32_t>(), s0, ns, (const uint32_t *)pxy, (const uint32_t *)doff, (int)nseg);
            else
                hipLaunchKernelGGL(k_line_products, dim3((unsigned)((2 * (size_t)ns * nsl * nseg + 63) / 64)), dim3(64), 0, sp, sl.ml_lines.as<uint32_t>(), n, slice_len, nsl, sl.ml_partial.as<uint32_t>(), doff, (int)nseg,
                                   s0, ns, (const uint32_t *)pxy);
            seg_tree(sp, sl.ml_partial.as<uint32_t>(), doff, s0, ns);
            ok(hipMemcpy2DAsync(Lp + s0, (size_t)N_LINES * 576, (const char *)sl.ml_out.p + (size_t)s0 * 576, (size_t)N_LINES * 576, (size_t)ns * 576, nseg, hipMemcpyDeviceToHost, sp));
            done[j] = sl.copy_ev[sl.ev_next++ % (Slot::N_COPY_EV + 1)];
            ok(hipEventRecord(done[j], sp));
        }
        ok(hipGetLastError());
        if (!rc) {
            const int device = cur().device;
            const size_t T = std::min<size_t>(std::min<size_t>(nseg, 16), std::max<size_t>(1, std::thread::hardware_concurrency()));
            const int32_t prc = par_run(T, [&](size_t k) -> int32_t {
                if (hipSetDevice(device) != hipSuccess) return DGPU_E_HIP;
                std::vector<MlTail> tails((nseg - k + T - 1) / T);
                for (int j = 0; j < ML_PIECES; j++) {
                    if (hipEventSynchronize(done[j]) != hipSuccess) return DGPU_E_HIP;
                    size_t m = 0;
                    for (size_t g = k; g < nseg; g += T, m++) if (off[g + 1] != off[g]) tails[m].run(Lp + g * N_LINES, ML_CUTS[j]);
                }
                size_t m = 0;
                for (size_t g = k; g < nseg; g += T, m++) { const hostf::Fq12 f = off[g + 1] == off[g] ? one : tails[m].result(); memcpy(out + g * 72, &f, sizeof f); finish(out + g * 72); }
                return DGPU_OK; });
            if (prc && !rc) rc = prc;
        }
        ok(hipStreamSynchronize(sa)); ok(hipStreamSynchronize(side[0])); ok(hipStreamSynchronize(side[1]));      // (nothing of this call stays in flight, whatever happened)
        if (gs.prof) prof_flush(sl);
        if (rc) return rc;
        return zero ? DGPU_E_ZERO : DGPU_OK;
    }
    { StageTimer st(sl, "ml.lines");
      if (n > 8192) hipLaunchKernelGGL(k_miller_lines_pair, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, s, sl.in_bases.as<uint32_t>(), sl.in_scalars.as<uint32_t>(), dskip, n, sl.ml_lines.as<uint32_t>(), n);
      else launch_lines_uneval(s, sl.in_bases.as<uint32_t>(), sl.in_scalars.as<uint32_t>(), dskip, n, sl.ml_lines.as<uint32_t>(), n,
                              62, 0, 0, (uint32_t *)nullptr, pxy); }       // (the evaluation at P is left to the product kernel: not part of the chain)
    { StageTimer st(sl, "ml.products");
      hipLaunchKernelGGL(k_line_products, dim3((unsigned)((2 * (size_t)N_LINES * nsl * nseg + 63) / 64)), dim3(64), 0, s, sl.ml_lines.as<uint32_t>(), n, slice_len, nsl, sl.ml_partial.as<uint32_t>(), doff, (int)nseg,
                         0, N_LINES, n > 8192 ? (const uint32_t *)nullptr : (const uint32_t *)pxy); }
    { StageTimer st(sl, "ml.tree");
      seg_tree(s, sl.ml_partial.as<uint32_t>(), doff, 0, N_LINES); }
    HIPCHK(hipGetLastError());
    std::vector<hostf::Fq12> L((size_t)N_LINES * nseg);
    HIPCHK(hipMemcpyAsync(L.data(), sl.ml_out.p, (size_t)N_LINES * nseg * 576, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (gs.prof) prof_flush(sl);
    auto tail = [&](size_t g) { const hostf::Fq12 f = off[g + 1] == off[g] ? one : ml_host_tail(&L[g * N_LINES]); memcpy(out + g * 72, &f, sizeof f); finish(out + g * 72); };
    const size_t T = std::min<size_t>(std::min<size_t>(nseg, 16), std::max<size_t>(1, std::thread::hardware_concurrency()));
    const int32_t prc = par_run(T, [&](size_t k) -> int32_t { for (size_t g = k; g < nseg; g += T) tail(g); return DGPU_OK; });
    if (prc) return prc;
    return zero ? DGPU_E_ZERO : DGPU_OK;
}

int32_t dgpu_multi_miller_loop_segments(const uint64_t *p, const uint64_t *q, const uint8_t *skip, size_t n, const uint64_t *seg_end, size_t nseg, uint64_t *out) {
    return ml_segments(p, q, skip, n, seg_end, nseg, out, false);
}
// E::multi_pairing per segment: the final exponentiation of every Miller output on the host thread that assembled it
int32_t dgpu_multi_pairing_segments(const uint64_t *p, const uint64_t *q, const uint8_t *skip, size_t n, const uint64_t *seg_end, size_t nseg, uint64_t *out) {
    return ml_segments(p, q, skip, n, seg_end, nseg, out, true);
}

// pairs chunked over the process's device contexts (SURVEY 8e "Miller loop": pairs are independent, the per-device raw outputs multiply —
// limb for limb the single-device value, because squaring distributes over the per-step line products and Fp12 products are exact)
int32_t dgpu_multi_miller_loop_sharded(const uint64_t *p, const uint64_t *q, const uint8_t *skip, size_t n, int32_t ngpus, uint64_t *out) {
    if (!out || (n && (!p || !q)) || ngpus < 0) return DGPU_E_BADARG;
    const std::vector<int> cx = ready_contexts(ngpus);
    if (cx.empty()) return DGPU_E_NODEVICE;
    if (ngpus > 0 && (int)cx.size() < ngpus) return DGPU_E_BADARG;
    const size_t G = cx.size();
    std::vector<hostf::Fq12> parts(G);
    const int32_t prc = par_run(G, [&](size_t k) -> int32_t {
        const size_t lo = k * (n / G) + std::min(k, n % G), hi = (k + 1) * (n / G) + std::min(k + 1, n % G);
        CtxScope here(cx[k]);
        return dgpu_multi_miller_loop(p + lo * 12, q + lo * 24, skip ? skip + lo : nullptr, hi - lo, (uint64_t *)&parts[k]);
    });
    if (prc) return prc;
    hostf::Fq12 f = parts[0];
    for (size_t k = 1; k < G; k++) f = f * parts[k];
    memcpy(out, &f, sizeof f);
    return DGPU_OK;
}

// E::G2Prepared::from for a batch (utils/src/randomized_pairing_check.rs:132 `b.into()`, legogroth16/src/verifier.rs:22-23,72)
int32_t dgpu_g2_prepare(const uint64_t *q, const uint8_t *is_inf, size_t n, uint64_t *out_coeffs, uint8_t *out_inf) {
    if (n && (!q || !out_coeffs || !out_inf)) return DGPU_E_BADARG;
    if (n == 0) return DGPU_OK;
    if (n > DGPU_MAX_PREPARED) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    SLOT_ACQUIRE(slot_lock, sl);
    HIPCHK(hipSetDevice(cur().device));
    int32_t rc;
    const size_t cbytes = n * (size_t)DGPU_G2_PREPARED_WORDS * 8;
    if ((rc = sl.in_scalars.ensure(n * 192))) return rc;
    if ((rc = sl.in_inf.ensure(2 * n))) return rc;
    if ((rc = sl.ml_coeffs.ensure(cbytes))) return rc;
    hipStream_t s = sl.stream;
    HIPCHK(hipMemcpyAsync(sl.in_scalars.p, q, n * 192, hipMemcpyHostToDevice, s));
    const uint8_t *dinf = nullptr;
    if (is_inf) { HIPCHK(hipMemcpyAsync(sl.in_inf.p, is_inf, n, hipMemcpyHostToDevice, s)); dinf = sl.in_inf.as<uint8_t>(); }
    { StageTimer st(sl, "ml.g2_prepare");
      if (n <= 8192 && (gs.ml_mode.load() & 1)) {
          if ((rc = sl.ml_lines.ensure((size_t)N_LINES * LW * n * 4))) return rc;
          launch_lines_uneval(s, (const uint32_t *)nullptr, sl.in_scalars.as<uint32_t>(), dinf, n, sl.ml_lines.as<uint32_t>(), n,
                             62, 0, 0, (uint32_t *)nullptr, (uint32_t *)nullptr);
          hipLaunchKernelGGL(k_prepared_from_lines, dim3((unsigned)((n * N_LINES * 6 + 255) / 256)), dim3(256), 0, s, sl.ml_lines.as<uint32_t>(), sl.in_scalars.as<uint32_t>(), dinf, n,
                             sl.ml_coeffs.as<uint32_t>(), sl.in_inf.as<uint8_t>() + n);
      } else
          hipLaunchKernelGGL(k_g2_prepare, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, s, sl.in_scalars.as<uint32_t>(), dinf, n, sl.ml_coeffs.as<uint32_t>(), sl.in_inf.as<uint8_t>() + n); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_coeffs, sl.ml_coeffs.p, cbytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_inf, sl.in_inf.as<uint8_t>() + n, n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (gs.prof) prof_flush(sl);
    return DGPU_OK;
}

// E::multi_miller_loop(a, b) with b already G2Prepared (what verifier.rs:69-76 and randomized_pairing_check.rs:207 pass)
int32_t dgpu_multi_miller_loop_prepared(const uint64_t *p, const uint64_t *coeffs, const uint8_t *skip, size_t n, uint64_t *out) {
    if (!out || (n && (!p || !coeffs))) return DGPU_E_BADARG;
    if (n == 0) { hostf::Fq12 one = hostf::Fq12::one(); memcpy(out, &one, sizeof one); return DGPU_OK; }
    if (n > DGPU_MAX_PREPARED) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    SLOT_ACQUIRE(slot_lock, sl);
    HIPCHK(hipSetDevice(cur().device));
    int32_t rc;
    const size_t cbytes = n * (size_t)DGPU_G2_PREPARED_WORDS * 8;
    if ((rc = sl.in_bases.ensure(n * 96))) return rc;
    if ((rc = sl.in_inf.ensure(n))) return rc;
    if ((rc = sl.ml_coeffs.ensure(cbytes))) return rc;
    if ((rc = sl.ml_lines.ensure((size_t)N_LINES * LW * n * 4))) return rc;
    hipStream_t s = sl.stream;
    HIPCHK(hipMemcpyAsync(sl.in_bases.p, p, n * 96, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(sl.ml_coeffs.p, coeffs, cbytes, hipMemcpyHostToDevice, s));
    const uint8_t *dskip = nullptr;
    if (skip) { HIPCHK(hipMemcpyAsync(sl.in_inf.p, skip, n, hipMemcpyHostToDevice, s)); dskip = sl.in_inf.as<uint8_t>(); }
    { StageTimer st(sl, "ml.lines_prepared");
      hipLaunchKernelGGL(k_lines_from_prepared, dim3((unsigned)((n * N_LINES + 255) / 256)), dim3(256), 0, s, sl.in_bases.as<uint32_t>(), sl.ml_coeffs.as<uint32_t>(), dskip, n, sl.ml_lines.as<uint32_t>(), n); }
    return ml_finish(sl, n, out);
}

// Affine and prepared G2 operands in ONE Miller loop: the line kernels of both forms fill one line buffer (the affine pairs first), the
// products / tree / host part run once.  This is what a verifier holds: proof.b affine, the key's -delta and -gamma prepared
// (legogroth16/src/verifier.rs:69-76: `[proof.b.into(), pvk.delta_g2_neg_pc.clone(), pvk.gamma_g2_neg_pc.clone()]`).  Preparing the affine
// members first costs a call of its own plus 19.5 KB per point down and up again.
int32_t dgpu_multi_miller_loop_mixed(const uint64_t *p_aff, const uint64_t *q_aff, const uint8_t *skip_aff, size_t n_aff,
                                     const uint64_t *p_prep, const uint64_t *coeffs, const uint8_t *skip_prep, size_t n_prep, uint64_t *out) {
    if (!out || (n_aff && (!p_aff || !q_aff)) || (n_prep && (!p_prep || !coeffs))) return DGPU_E_BADARG;
    const size_t n = n_aff + n_prep;
    if (n == 0) { hostf::Fq12 one = hostf::Fq12::one(); memcpy(out, &one, sizeof one); return DGPU_OK; }
    if (n_prep > DGPU_MAX_PREPARED || n >= (1ull << 24)) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    SLOT_ACQUIRE(slot_lock, sl);
    HIPCHK(hipSetDevice(cur().device));
    int32_t rc;
    const size_t cbytes = n_prep * (size_t)DGPU_G2_PREPARED_WORDS * 8;
    if ((rc = sl.in_bases.ensure(n * 96))) return rc;                 // P of the affine pairs, then P of the prepared pairs
    if ((rc = sl.in_scalars.ensure(n_aff * 192 + 16))) return rc;
    if ((rc = sl.in_inf.ensure(n))) return rc;
    if ((rc = sl.ml_coeffs.ensure(cbytes + 16))) return rc;
    if ((rc = sl.ml_lines.ensure((size_t)N_LINES * LW * n * 4))) return rc;
    hipStream_t s = sl.stream;
    uint32_t *dp = sl.in_bases.as<uint32_t>();
    uint8_t *dsk = sl.in_inf.as<uint8_t>();
    if (n_aff) {
        HIPCHK(hipMemcpyAsync(dp, p_aff, n_aff * 96, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(sl.in_scalars.p, q_aff, n_aff * 192, hipMemcpyHostToDevice, s));
        if (skip_aff) HIPCHK(hipMemcpyAsync(dsk, skip_aff, n_aff, hipMemcpyHostToDevice, s));
    }
    if (n_prep) {
        HIPCHK(hipMemcpyAsync(dp + n_aff * 24, p_prep, n_prep * 96, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(sl.ml_coeffs.p, coeffs, cbytes, hipMemcpyHostToDevice, s));
        if (skip_prep) HIPCHK(hipMemcpyAsync(dsk + n_aff, skip_prep, n_prep, hipMemcpyHostToDevice, s));
    }
    if (n_aff && n_aff <= 8192 && !gs.prof && (gs.ml_mode.load() & 1)) {
        // what a verifier calls (proof.b affine, the key's -delta and -gamma prepared): the affine pairs' chain is cut like
        // dgpu_multi_miller_loop's; the prepared pairs' lines are ready long before the first launch of the line kernel ends
        MlActive act(cur().ml_active);
        if (act.v <= 2) {
            uint32_t *lines = sl.ml_lines.as<uint32_t>();
            return ml_pipelined(sl, n, n_aff, skip_aff ? dsk : nullptr, out, [&](uint32_t *pxy) {
                if (n_prep)
                    hipLaunchKernelGGL(k_lines_from_prepared, dim3((unsigned)((n_prep * N_LINES + 255) / 256)), dim3(256), 0, s, dp + n_aff * 24, sl.ml_coeffs.as<uint32_t>(),
                                       skip_prep ? dsk + n_aff : (const uint8_t *)nullptr, n_prep, lines + n_aff, n, pxy + n_aff);
            });
        }
    }
    { StageTimer st(sl, "ml.lines");
      uint32_t *lines = sl.ml_lines.as<uint32_t>();
      if (n_aff) {
          const uint8_t *sk = skip_aff ? dsk : nullptr;
          if (n_aff > 8192) hipLaunchKernelGGL(k_miller_lines_pair, dim3((unsigned)((2 * n_aff + 63) / 64)), dim3(64), 0, s, dp, sl.in_scalars.as<uint32_t>(), sk, n_aff, lines, n);
          else hipLaunchKernelGGL(k_miller_lines_quad<true>, dim3((unsigned)((4 * n_aff + 63) / 64)), dim3(64), 0, s, dp, sl.in_scalars.as<uint32_t>(), sk, n_aff, lines, n);
      }
      if (n_prep)
          hipLaunchKernelGGL(k_lines_from_prepared, dim3((unsigned)((n_prep * N_LINES + 255) / 256)), dim3(256), 0, s, dp + n_aff * 24, sl.ml_coeffs.as<uint32_t>(),
                             skip_prep ? dsk + n_aff : (const uint8_t *)nullptr, n_prep, lines + n_aff, n);
    }
    return ml_finish(sl, n, out);
}

// ---- prod_i e(m_i P_i, Q_i) x prod_j e(P'_j, prepared_j): the scalings and the Miller loop of RandomizedPairingChecker as ONE call ----
// utils/src/randomized_pairing_check.rs:125-134: `a.mul_bigint(m)` for every source, then the pairs go to one multi_miller_loop (:204-214 in lazy mode).
// The line coefficients of a pair depend on Q alone and P enters only in the product kernels (the lines leave unevaluated, px and py travel beside
// them), so the chain of the Q_i starts at once and the 128-step scaling chains of the P_i (k_g1_scale_quad) run BESIDE it instead of in front of it;
// the scaled points never visit the host.  1024 pairs: 1.14 (scalings) + 0.75 (Miller loop) ms one after the other -> ~1.3 ms.
// scaled P (affine ABI words, on the device) -> px, py in the internal form for k_line_products
__global__ void __launch_bounds__(256) k_pxy_from_abi(const uint32_t *__restrict__ p_abi, size_t n, uint32_t *__restrict__ pxy, size_t stride) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n) return;
    const size_t i = t >> 1; const uint32_t h = (uint32_t)(t & 1);
    uint32_t w[12];
    for (int k = 0; k < 12; k++) w[k] = p_abi[i * 24 + h * 12 + k];
    Fp c; fp_from_abi(c, w);
    for (int k = 0; k < NL; k++) pxy[(h * NL + k) * stride + i] = c.l[k];
}
int32_t dgpu_multi_miller_loop_scaled(const uint64_t *p_aff, const uint64_t *scalars, size_t scalar_stride, const uint64_t *q_aff, const uint8_t *skip_aff, size_t n_aff,
                                      const uint64_t *p_prep, const uint64_t *coeffs, const uint8_t *skip_prep, size_t n_prep, uint64_t *out) {
    if (!out || (n_aff && (!p_aff || !q_aff || !scalars)) || (n_prep && (!p_prep || !coeffs)) || (scalar_stride != 0 && scalar_stride != 4)) return DGPU_E_BADARG;
    const size_t n = n_aff + n_prep;
    if (n_aff == 0) return dgpu_multi_miller_loop_mixed(nullptr, nullptr, nullptr, 0, p_prep, coeffs, skip_prep, n_prep, out);
    if (n_prep > DGPU_MAX_PREPARED || n >= (1ull << 24)) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    // GLV: every scalar goes to the device as (k1 | k2), k mod r = k1 + k2 lambda; a pair whose scaled point is the identity is skipped (m = 0 mod r or P = 0)
    const size_t nsc = scalar_stride ? n_aff : 1;
    std::vector<uint64_t> split(nsc * 4);
    for (size_t k = 0; k < nsc; k++) hostf::glv_decompose(scalars + 4 * k, &split[4 * k], &split[4 * k + 2]);
    auto zero = [](const uint64_t *w, int k) { uint64_t o = 0; for (int i = 0; i < k; i++) o |= w[i]; return o == 0; };
    std::vector<uint8_t> skip(n_aff);
    for (size_t i = 0; i < n_aff; i++)
        skip[i] = (skip_aff && skip_aff[i]) || zero(&split[scalar_stride ? 4 * i : 0], 4) || zero(p_aff + 12 * i, 12) ? 1 : 0;
    bool pipelined = n_aff <= 8192 && !gs.prof && (gs.ml_mode.load() & 1);
    if (pipelined) { MlActive probe(cur().ml_active); pipelined = probe.v <= 2; }
    if (!pipelined) {          // large batches (throughput-bound), stage timers on, or many loops in flight: the two calls one after the other
        std::vector<uint64_t> scaled(n_aff * 12); std::vector<uint8_t> sinf(n_aff);
        int32_t rc = dgpu_g1_scale_batch(p_aff, nullptr, scalars, scalar_stride, nullptr, n_aff, scaled.data(), sinf.data());
        if (rc) return rc;
        for (size_t i = 0; i < n_aff; i++) skip[i] |= sinf[i];
        return dgpu_multi_miller_loop_mixed(scaled.data(), q_aff, skip.data(), n_aff, p_prep, coeffs, skip_prep, n_prep, out);
    }
    SLOT_ACQUIRE(slot_lock, sl);
    HIPCHK(hipSetDevice(cur().device));
    int32_t rc;
    const size_t cbytes = n_prep * (size_t)DGPU_G2_PREPARED_WORDS * 8;
    const size_t sc_off = (n_aff * 96 + n_aff + 63) & ~(size_t)63;            // prepped: [scaled points | their flags | pad | split scalars]
    if ((rc = sl.in_bases.ensure(n * 96))) return rc;
    if ((rc = sl.in_scalars.ensure(n_aff * 192 + 16))) return rc;
    if ((rc = sl.in_inf.ensure(n))) return rc;
    if ((rc = sl.prepped.ensure(sc_off + nsc * 32))) return rc;
    if ((rc = sl.ml_coeffs.ensure(cbytes + 16))) return rc;
    if ((rc = sl.ml_lines.ensure((size_t)N_LINES * LW * n * 4))) return rc;
    hipStream_t s = sl.stream;
    uint32_t *dp = sl.in_bases.as<uint32_t>();
    uint8_t *dsk = sl.in_inf.as<uint8_t>();
    HIPCHK(hipMemcpyAsync(sl.in_scalars.p, q_aff, n_aff * 192, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dsk, skip.data(), n_aff, hipMemcpyHostToDevice, s));
    if (n_prep) {
        HIPCHK(hipMemcpyAsync(dp + n_aff * 24, p_prep, n_prep * 96, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(sl.ml_coeffs.p, coeffs, cbytes, hipMemcpyHostToDevice, s));
        if (skip_prep) HIPCHK(hipMemcpyAsync(dsk + n_aff, skip_prep, n_prep, hipMemcpyHostToDevice, s));
    }
    MlActive act(cur().ml_active);
    uint32_t *lines = sl.ml_lines.as<uint32_t>();
    uint32_t *scaled = sl.prepped.as<uint32_t>(); uint8_t *scaled_inf = sl.prepped.as<uint8_t>() + n_aff * 96;
    uint32_t *dsplit = (uint32_t *)(sl.prepped.as<uint8_t>() + sc_off);
    return ml_pipelined(sl, n, n_aff, dsk, out, [&](uint32_t *pxy) {
        if (n_prep)
            hipLaunchKernelGGL(k_lines_from_prepared, dim3((unsigned)((n_prep * N_LINES + 255) / 256)), dim3(256), 0, s, dp + n_aff * 24, sl.ml_coeffs.as<uint32_t>(),
                               skip_prep ? dsk + n_aff : (const uint8_t *)nullptr, n_prep, lines + n_aff, n, pxy + n_aff);
    }, [&](hipStream_t side, uint32_t *pxy) -> int32_t {
        // the chain of the Q_i is queued: the scalings run beside it on the side stream, their results go straight to px, py
        if (hipMemcpyAsync(dp, p_aff, n_aff * 96, hipMemcpyHostToDevice, side) != hipSuccess) return DGPU_E_HIP;
        if (hipMemcpyAsync(dsplit, split.data(), nsc * 32, hipMemcpyHostToDevice, side) != hipSuccess) return DGPU_E_HIP;
        msm::launch_g1_scale_quad(side, dp, nullptr, dsplit, (int)(scalar_stride * 2), nullptr, n_aff, scaled, scaled_inf);
        hipLaunchKernelGGL(k_pxy_from_abi, dim3((unsigned)((2 * n_aff + 255) / 256)), dim3(256), 0, side, (const uint32_t *)scaled, n_aff, pxy, n);
        return hipGetLastError() == hipSuccess ? DGPU_OK : DGPU_E_HIP;
    }, true);
}

// ---- the LegoGroth16 verifier as one call (legogroth16/src/verifier.rs:62-99 `verify_proof`: calculate_d :29-50,101-109, then verify_qap_proof) ----
// e(A, B) e(C, -delta) e(gamma_abc[0] + sum x_j gamma_abc[1 + j] + D, -gamma) == e(alpha, beta), with -delta, -gamma held prepared and e(alpha, beta)
// precomputed in the PreparedVerifyingKey (verifier.rs:17-25).  One call instead of calculate_d + multi_miller_loop + final_exponentiation lets
// the library hide the host's share under the device's: the chain of the one affine pair (A, B) is queued first, the third G1 operand (a
// 255-bit scalar multiplication per public input, ~0.15 ms of one host core) is computed while it runs, the two prepared pairs' lines follow on
// a side stream in front of the product kernels.  *ok = 1 / 0; DGPU_E_BADARG for a key too short for the inputs (`MalformedVerifyingKey`).
int32_t dgpu_legogroth16_verify(const uint64_t alpha_beta_gt[72], const uint64_t *delta_neg_pc, const uint64_t *gamma_neg_pc, const uint64_t *gamma_abc_g1, size_t gamma_abc_len,
                                const uint64_t proof_a[12], const uint64_t proof_b[24], const uint64_t proof_c[12], const uint64_t proof_d[12], const uint8_t *proof_inf /* 4 or NULL */,
                                const uint64_t *public_inputs, size_t n_pub, int32_t montgomery, int32_t *ok_out) {
    if (!alpha_beta_gt || !delta_neg_pc || !gamma_neg_pc || !gamma_abc_g1 || !proof_a || !proof_b || !proof_c || !proof_d || !ok_out || (n_pub && !public_inputs)) return DGPU_E_BADARG;
    if (n_pub + 1 > gamma_abc_len) return DGPU_E_BADARG;                       // verifier.rs:38-40 MalformedVerifyingKey
    if (n_pub + 2 > DGPU_MAX_LINCOMB) return DGPU_E_BADARG;                    // (more public inputs: calculate_d through dgpu_msm_g1, then dgpu_multi_miller_loop_mixed)
    if (!cur().ready) return DGPU_E_NODEVICE;
    auto all_zero = [](const uint64_t *w, int k) { uint64_t o = 0; for (int i = 0; i < k; i++) o |= w[i]; return o == 0; };
    const bool inf_a = (proof_inf && proof_inf[0]) || all_zero(proof_a, 12), inf_b = (proof_inf && proof_inf[1]) || all_zero(proof_b, 24);
    const bool inf_c = (proof_inf && proof_inf[2]) || all_zero(proof_c, 12), inf_d = (proof_inf && proof_inf[3]) || all_zero(proof_d, 12);
    // d = gamma_abc[0] + sum x_j gamma_abc[1 + j] + proof.d on a host core
    uint64_t dxy[12]; bool d_inf = true;
    auto compute_d = [&]() -> int32_t {
        uint64_t pts[DGPU_MAX_LINCOMB * 12], sc[DGPU_MAX_LINCOMB * 4], jac[18]; uint8_t inf[DGPU_MAX_LINCOMB] = {0};
        const size_t k = n_pub + 2;
        memcpy(pts, gamma_abc_g1, (n_pub + 1) * 96); memcpy(pts + (n_pub + 1) * 12, proof_d, 96); inf[n_pub + 1] = inf_d;
        memset(sc, 0, sizeof sc); sc[0] = 1; sc[4 * (n_pub + 1)] = 1;
        for (size_t j = 0; j < n_pub; j++) { if (montgomery) hostf::fr_from_mont(&sc[4 * (1 + j)], public_inputs + 4 * j); else memcpy(&sc[4 * (1 + j)], public_inputs + 4 * j, 32); }
        const int32_t e = dgpu_lincomb_g1(pts, inf, sc, k, jac);
        if (e) return e;
        d_inf = all_zero(jac + 12, 6);
        memcpy(dxy, jac, 96);                                                   // normalised: (x, y, 1)
        return DGPU_OK;
    };
    SLOT_ACQUIRE(slot_lock, sl);
    HIPCHK(hipSetDevice(cur().device));
    int32_t rc;
    const size_t n = 3, cbytes = 2 * (size_t)DGPU_G2_PREPARED_WORDS * 8;
    if ((rc = sl.in_bases.ensure(n * 96))) return rc;
    if ((rc = sl.in_scalars.ensure(192 + 16))) return rc;
    if ((rc = sl.in_inf.ensure(16))) return rc;
    if ((rc = sl.ml_coeffs.ensure(cbytes + 16))) return rc;
    if ((rc = sl.ml_lines.ensure((size_t)N_LINES * LW * n * 4))) return rc;
    hipStream_t s = sl.stream;
    uint32_t *dp = sl.in_bases.as<uint32_t>();
    uint8_t *dsk = sl.in_inf.as<uint8_t>();
    uint8_t *hsk = (uint8_t *)sl.hpin + Slot::HPIN_BYTES - 256;                 // pinned: the flags of the three pairs, and d's coordinates behind them
    hsk[0] = (inf_a || inf_b) ? 1 : 0; hsk[1] = inf_c ? 1 : 0; hsk[2] = 0;
    HIPCHK(hipMemcpyAsync(dp, proof_a, 96, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dp + 24, proof_c, 96, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(sl.in_scalars.p, proof_b, 192, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(sl.ml_coeffs.p, delta_neg_pc, cbytes / 2, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync((char *)sl.ml_coeffs.p + cbytes / 2, gamma_neg_pc, cbytes / 2, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dsk, hsk, 1, hipMemcpyHostToDevice, s));
    hostf::Fq12 f;
    uint32_t *lines = sl.ml_lines.as<uint32_t>();
    auto prepared_lines = [&](hipStream_t st, uint32_t *pxy) {
        hipLaunchKernelGGL(k_lines_from_prepared, dim3((unsigned)((2 * N_LINES + 255) / 256)), dim3(256), 0, st, dp + 24, sl.ml_coeffs.as<uint32_t>(), (const uint8_t *)(dsk + 1), (size_t)2, lines + 1, n, pxy ? pxy + 1 : nullptr);
    };
    MlActive act(cur().ml_active);
    if (!gs.prof && (gs.ml_mode.load() & 1) && act.v <= 2) {
        hipEvent_t in_ev = sl.copy_ev[sl.ev_next++ % (Slot::N_COPY_EV + 1)];
        HIPCHK(hipEventRecord(in_ev, s));
        rc = ml_pipelined(sl, n, 1, dsk, (uint64_t *)&f, [](uint32_t *) {}, [&](hipStream_t side, uint32_t *pxy) -> int32_t {
            const int32_t e = compute_d(); if (e) return e;                     // (the chain of (A, B) is running)
            memcpy(hsk + 16, dxy, 96); hsk[2] = d_inf ? 1 : 0;
            if (hipStreamWaitEvent(side, in_ev, 0) != hipSuccess) return DGPU_E_HIP;       // C, the coefficients and the first flag are on their way on `s`
            if (hipMemcpyAsync(dp + 48, hsk + 16, 96, hipMemcpyHostToDevice, side) != hipSuccess) return DGPU_E_HIP;
            if (hipMemcpyAsync(dsk + 1, hsk + 1, 2, hipMemcpyHostToDevice, side) != hipSuccess) return DGPU_E_HIP;
            prepared_lines(side, pxy);
            return DGPU_OK;
        });
        if (rc) return rc;
    } else {
        if ((rc = compute_d())) return rc;
        memcpy(hsk + 16, dxy, 96); hsk[2] = d_inf ? 1 : 0;
        HIPCHK(hipMemcpyAsync(dp + 48, hsk + 16, 96, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dsk + 1, hsk + 1, 2, hipMemcpyHostToDevice, s));
        if ((rc = sl.ml_state.ensure((size_t)2 * NL * n * 4))) return rc;
        uint32_t *pxy = sl.ml_state.as<uint32_t>();
        launch_lines_uneval(s, dp, sl.in_scalars.as<uint32_t>(), dsk, 1, lines, n, 62, 0, 0, (uint32_t *)nullptr, pxy);
        prepared_lines(s, pxy);
        if ((rc = ml_finish(sl, n, (uint64_t *)&f, pxy))) return rc;
    }
    hostf::Fq12 gt, want; memcpy(&want, alpha_beta_gt, sizeof want);
    if (!hostf::final_exponentiation(gt, f)) return DGPU_E_ZERO;               // (verifier.rs:78 `.ok_or(UnexpectedIdentity)`)
    *ok_out = memcmp(&gt, &want, sizeof gt) == 0 ? 1 : 0;
    return DGPU_OK;
}

int32_t dgpu_g1_scale_batch(const uint64_t *p, const uint8_t *is_inf, const uint64_t *scalars, size_t scalar_stride, const uint8_t *negate, size_t n, uint64_t *out, uint8_t *out_inf) {
    if ((n && (!p || !scalars || !out || !out_inf)) || (scalar_stride != 0 && scalar_stride != 4)) return DGPU_E_BADARG;
    if (n == 0) return DGPU_OK;
    if (!cur().ready) return DGPU_E_NODEVICE;
    SLOT_ACQUIRE(slot_lock, sl);
    HIPCHK(hipSetDevice(cur().device));
    int32_t rc;
    const size_t nsc = scalar_stride ? n : 1;
    if ((rc = sl.in_bases.ensure(n * 96))) return rc;
    if ((rc = sl.in_scalars.ensure(nsc * 32))) return rc;
    if ((rc = sl.in_inf.ensure(2 * n))) return rc;
    if ((rc = sl.prepped.ensure(n * 96 + n))) return rc;
    hipStream_t s = sl.stream;
    HIPCHK(hipMemcpyAsync(sl.in_bases.p, p, n * 96, hipMemcpyHostToDevice, s));
    // GLV: every scalar is sent as (k1 | k2), k mod r = k1 + k2 lambda, both below 2^128 (host_field.hpp)
    std::vector<uint64_t> split(nsc * 4);
    for (size_t k = 0; k < nsc; k++) hostf::glv_decompose(scalars + 4 * k, &split[4 * k], &split[4 * k + 2]);
    HIPCHK(hipMemcpyAsync(sl.in_scalars.p, split.data(), nsc * 32, hipMemcpyHostToDevice, s));
    const uint8_t *dinf = nullptr, *dneg = nullptr;
    if (is_inf) { HIPCHK(hipMemcpyAsync(sl.in_inf.p, is_inf, n, hipMemcpyHostToDevice, s)); dinf = sl.in_inf.as<uint8_t>(); }
    if (negate) { HIPCHK(hipMemcpyAsync(sl.in_inf.as<uint8_t>() + n, negate, n, hipMemcpyHostToDevice, s)); dneg = sl.in_inf.as<uint8_t>() + n; }
    uint8_t *dout_inf = sl.prepped.as<uint8_t>() + n * 96;
    { StageTimer st(sl, "pc.g1_scale");
      msm::launch_g1_scale_quad(s, sl.in_bases.as<uint32_t>(), dinf, sl.in_scalars.as<uint32_t>(), (int)(scalar_stride * 2), dneg, n, sl.prepped.as<uint32_t>(), dout_inf); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, sl.prepped.p, n * 96, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_inf, dout_inf, n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (gs.prof) prof_flush(sl);
    return DGPU_OK;
}
// E::final_exponentiation: once per batch, host code
int32_t dgpu_final_exponentiation(const uint64_t *in, uint64_t *out) {
    if (!in || !out) return DGPU_E_BADARG;
    hostf::Fq12 f, r; memcpy(&f, in, sizeof f);
    if (!hostf::final_exponentiation(r, f)) return DGPU_E_ZERO;
    memcpy(out, &r, sizeof r);
    return DGPU_OK;
}

}  // extern "C"
