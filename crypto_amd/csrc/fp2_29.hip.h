// crypto_amd/csrc/fp2_29.hip.h — Fp2 = Fp[u]/(u^2 + 1) over the lazy 29-bit-limb base field (fp29.hip.h).
// Device counterpart of ark_ff::Fp2<Fq2Config> as used by G2 (ark-bls12-381 0.4; reached from
// legogroth16/src/prover.rs:344 `b_g2_query` MSM and the Miller loop, utils/src/randomized_pairing_check.rs:207).
// ABI order of components: c0 then c1.
#pragma once
#include "fp29.hip.h"

namespace bls29 {

// FP2_NOINLINE: compile the Fp2 product / square as real device functions (smaller register footprint and code for the
// G2 kernels, whose fully inlined mixed addition does not fit the 256-VGPR budget)
#if defined(FP2_NOINLINE) && defined(__HIPCC__)
#define FD2 __host__ __device__ __noinline__
#else
#define FD2 FD
#endif

struct Fp2 { Fp c0, c1; };

FD void fzero(Fp2 &r) { fp_zero(r.c0); fp_zero(r.c1); }
FD void fset_one(Fp2 &r) { fp_set_one(r.c0); fp_zero(r.c1); }
FD void fadd(Fp2 &r, const Fp2 &a, const Fp2 &b) { fp_add(r.c0, a.c0, b.c0); fp_add(r.c1, a.c1, b.c1); }
FD void fdbl(Fp2 &r, const Fp2 &a) { fp_add(r.c0, a.c0, a.c0); fp_add(r.c1, a.c1, a.c1); }
template <int M> FD void fsub(Fp2 &r, const Fp2 &a, const Fp2 &b) { fp_sub<M>(r.c0, a.c0, b.c0); fp_sub<M>(r.c1, a.c1, b.c1); }
FD void fnorm(Fp2 &r, const Fp2 &a) { fp_norm(r.c0, a.c0); fp_norm(r.c1, a.c1); }
FD bool fmaybe_zero(const Fp2 &a) { return fp_maybe_zero(a.c0) && fp_maybe_zero(a.c1); }
FD bool fis_zero_exact(const Fp2 &a) { return fp_is_zero_exact(a.c0) && fp_is_zero_exact(a.c1); }

// (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u : each component is ONE fused two-product Montgomery reduction (fp_mul2), so a product
// costs 2 x 588 mads — the same multiplier work as Karatsuba's 3 x 392 but without its operand sums, subtractions and
// carry passes, and both outputs come out with value < 2 p.  Inputs class N with value < 500 p.
FD void fmul(Fp2 &r, const Fp2 &a, const Fp2 &b) {
    Fp z, n1, c0, c1;
    fp_zero(z); fp_sub<512>(n1, z, a.c1); fp_norm(n1, n1);      // -a1 (lazily: 512 p - a1)
    fp_mul2(c0, a.c0, b.c0, n1, b.c1);
    fp_mul2(c1, a.c0, b.c1, a.c1, b.c0);
    r.c0 = c0; r.c1 = c1;
}
// (a0 + a1)(a0 - a1), 2 a0 a1 : 2 base-field products.  Input class N with value < 60 p.
FD2 void fsqr(Fp2 &r, const Fp2 &a) {
    Fp s, d, t;
    fp_add(s, a.c0, a.c1);
    fp_sub<64>(d, a.c0, a.c1); fp_norm(d, d);
    fp_mul(t, a.c0, a.c1);
    fp_mul(r.c0, s, d);
    fp_add(r.c1, t, t); fp_norm(r.c1, r.c1);
}
// r = a*b - c*d (class N): two products and a lazy subtraction (no fused form over Fp2 yet)
template <int M> FD void fmul_sub(Fp2 &r, const Fp2 &a, const Fp2 &b, const Fp2 &c, const Fp2 &d) {
    Fp2 t, u; fmul(t, a, b); fmul(u, c, d); fsub<8>(t, t, u); fnorm(r, t);
}
// multiply both components by a base-field element (Miller loop line evaluation)
FD void fmul_fp(Fp2 &r, const Fp2 &a, const Fp &k) { fp_mul(r.c0, a.c0, k); fp_mul(r.c1, a.c1, k); }
// (a0 + a1 u)(1 + u) = (a0 - a1) + (a0 + a1) u
template <int M> FD void fmul_xi(Fp2 &r, const Fp2 &a) {
    Fp t; fp_sub<M>(t, a.c0, a.c1); fp_add(r.c1, a.c0, a.c1); r.c0 = t;
}
FD void fneg_c1(Fp2 &r, const Fp2 &a) { Fp z; fp_zero(z); r.c0 = a.c0; fp_sub<64>(r.c1, z, a.c1); }

}  // namespace bls29
