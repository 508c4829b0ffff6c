// crypto_amd/csrc/fp_safegcd.hip.h — inversion in Fq by Bernstein–Yang division steps ("safegcd", ePrint 2019/266), host + device.
//
// The per-element outputs of the batched scalar-multiplication kernels end in ONE field inversion per lane (XYZZ -> affine:
// `into_affine` / `normalize_batch` in the reference, legogroth16/src/generator.rs:424-431, utils/src/randomized_pairing_check.rs:125-127,
// legogroth16/src/aggregation/utils.rs:34-49).  The Fermat power a^(p-2) is 380 squarings + ~190 products = ~250 k instructions of one
// dependent chain (0.5 ms on a lone wave, and a third of the fixed-base kernel's work); the division-step recurrence below is
// ~37 k instructions and data-independent (every lane runs the same 38 batches: no divergence inside a wave).
//
//   divstep(delta, f, g) = (1 - delta, g, (g - f) / 2)            if delta > 0 and g odd
//                          (1 + delta, f, (g + (g mod 2) f) / 2)  otherwise
// started at (1, p, a), 0 <= a < p: after floor((49 * 381 + 57) / 17) = 1101 steps g = 0 and f = +-gcd = +-1 (Theorem 11.2 of the paper,
// d = 381 bits).  Steps are taken 29 at a time on the low limbs only (the next 29 steps depend on nothing else), which yields a 2 x 2
// integer matrix t with (f', g') = t (f, g) / 2^29; the same matrix is applied to (d, e), kept modulo p with d a = f, e a = g (mod p),
// the exact division by 2^29 done by first adding the multiple of p that clears the low limb.  At the end d = +-1 / a.
// f, g, d, e: 14 signed limbs of 29 bits (limbs 0..12 in [0, 2^29), limb 13 carries the sign) — the limb width of fp29.hip.h.
#pragma once
#include "fp29.hip.h"

namespace bls29 {

constexpr int SG_STEPS = LB;                 // division steps per batch = limb width
constexpr int SG_BATCHES = 38;               // 38 * 29 = 1102 >= 1101
static_assert(SG_STEPS * SG_BATCHES >= (49 * 381 + 57) / 17, "division-step count below the proven bound");
// 2^(3 * 406) mod p: the Montgomery product with it turns (a 2^406)^-1 into a^-1 2^406
#define BLS29_R3 {0x9217d6au, 0x1d6118bau, 0x1114b11cu, 0x126aee7u, 0xa55e2c4u, 0x4d63ce0u, 0x154ff87du, 0x14555478u, 0x1d1bdc0du, 0x161f98d4u, 0x1d74e921u, 0x9b4345au, 0x1e5ecfb8u, 0xau}

struct SgMat { int32_t u, v, q, r; };
struct SgInt { int32_t l[NL]; };

// 29 division steps on the low limbs; eta = -delta.  Branch-free: both cases are formed with masks.
FD int32_t sg_divsteps(int32_t eta, uint32_t f, uint32_t g, SgMat &t) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
#pragma unroll
    for (int i = 0; i < SG_STEPS; i++) {
        uint32_t swap = (uint32_t)(eta >> 31);                 // delta > 0
        const uint32_t odd = 0u - (g & 1u);
        const uint32_t x = (f ^ swap) - swap, y = (u ^ swap) - swap, z = (v ^ swap) - swap;    // -f, -u, -v when delta > 0
        g += x & odd; q += y & odd; r += z & odd;             // g +- f (now even)
        swap &= odd;
        eta = (int32_t)(((uint32_t)eta ^ swap) - (swap + 1u)); // delta <- 1 - delta or 1 + delta
        f += g & swap; u += q & swap; v += r & swap;          // f <- old g
        g >>= 1; u <<= 1; v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return eta;
}
// (f, g) <- t (f, g) / 2^29 (exact)
FD void sg_update_fg(SgInt &f, SgInt &g, const SgMat &t) {
    int64_t cf = (int64_t)t.u * f.l[0] + (int64_t)t.v * g.l[0];
    int64_t cg = (int64_t)t.q * f.l[0] + (int64_t)t.r * g.l[0];
    cf >>= LB; cg >>= LB;
#pragma unroll
    for (int i = 1; i < NL; i++) {
        cf += (int64_t)t.u * f.l[i] + (int64_t)t.v * g.l[i];
        cg += (int64_t)t.q * f.l[i] + (int64_t)t.r * g.l[i];
        f.l[i - 1] = (int32_t)((uint32_t)cf & LMASK); cf >>= LB;
        g.l[i - 1] = (int32_t)((uint32_t)cg & LMASK); cg >>= LB;
    }
    f.l[NL - 1] = (int32_t)cf; g.l[NL - 1] = (int32_t)cg;
}
// (d, e) <- t (d, e) / 2^29 mod p, both kept in (-2p, p)
FD void sg_update_de(SgInt &d, SgInt &e, const SgMat &t) {
    BLS29_DECL_P;
    const int32_t sd = d.l[NL - 1] >> 31, se = e.l[NL - 1] >> 31;
    int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);      // a negative d / e first gains p (times its matrix entry)
    int64_t cd = (int64_t)t.u * d.l[0] + (int64_t)t.v * e.l[0];
    int64_t ce = (int64_t)t.q * d.l[0] + (int64_t)t.r * e.l[0];
    md -= (int32_t)((PINV29 * (uint32_t)cd + (uint32_t)md) & LMASK);         // cd + p md = 0 mod 2^29
    me -= (int32_t)((PINV29 * (uint32_t)ce + (uint32_t)me) & LMASK);
    cd += (int64_t)P_[0] * md; ce += (int64_t)P_[0] * me;
    cd >>= LB; ce >>= LB;
#pragma unroll
    for (int i = 1; i < NL; i++) {
        cd += (int64_t)t.u * d.l[i] + (int64_t)t.v * e.l[i] + (int64_t)P_[i] * md;
        ce += (int64_t)t.q * d.l[i] + (int64_t)t.r * e.l[i] + (int64_t)P_[i] * me;
        d.l[i - 1] = (int32_t)((uint32_t)cd & LMASK); cd >>= LB;
        e.l[i - 1] = (int32_t)((uint32_t)ce & LMASK); ce >>= LB;
    }
    d.l[NL - 1] = (int32_t)cd; e.l[NL - 1] = (int32_t)ce;
}
// x <- (negate ? -x : x) + (add_p ? p : 0), limbs carried
FD void sg_fix(SgInt &x, bool negate, bool add_p) {
    BLS29_DECL_P;
    const int32_t n = negate ? -1 : 0, m = add_p ? -1 : 0;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        const int32_t v = ((x.l[i] ^ n) - n) + ((int32_t)P_[i] & m) + c;
        x.l[i] = (int32_t)((uint32_t)v & LMASK); c = v >> LB;
    }
    x.l[NL - 1] = ((x.l[NL - 1] ^ n) - n) + ((int32_t)P_[NL - 1] & m) + c;
}

// r = 1 / a in the Montgomery domain of fp29.hip.h (a 2^406 -> a^-1 2^406); 0 -> 0 like the Fermat power.  Any lazy class of a.
FD void fp_inv_safegcd(Fp &r, const Fp &a) {
    BLS29_DECL_P;
    Fp ac; fp_canon(ac, a);
    SgInt f, g, d, e;
#pragma unroll
    for (int i = 0; i < NL; i++) { f.l[i] = (int32_t)P_[i]; g.l[i] = (int32_t)ac.l[i]; d.l[i] = 0; e.l[i] = 0; }
    e.l[0] = 1;
    int32_t eta = -1;
    for (int b = 0; b < SG_BATCHES; b++) {
        SgMat t;
        eta = sg_divsteps(eta, (uint32_t)f.l[0], (uint32_t)g.l[0], t);
        sg_update_de(d, e, t);
        sg_update_fg(f, g, t);
    }
    // g == 0 and f == +-1 (f == p when a == 0): d = sign(f) / a, d in (-2p, p)
    sg_fix(d, false, d.l[NL - 1] < 0);
    sg_fix(d, f.l[NL - 1] < 0, false);
    sg_fix(d, false, d.l[NL - 1] < 0);
    Fp dc, r3;
    constexpr uint32_t R3_[NL] = BLS29_R3;
#pragma unroll
    for (int i = 0; i < NL; i++) { dc.l[i] = (uint32_t)d.l[i]; r3.l[i] = R3_[i]; }
    CHK(chk_set_N(dc, 1.0); chk_set_N(r3, 1.0); chk_actual(dc);)
    fp_mul(r, dc, r3);
}

}  // namespace bls29
