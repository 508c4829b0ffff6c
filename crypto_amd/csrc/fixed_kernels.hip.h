// crypto_amd/csrc/fixed_kernels.hip.h — batched fixed-base scalar multiplication: out_i = s_i * B for one base B.
//
// Device form of ark-ec's FixedBase::{get_window_table, msm} as the reference uses them: the six query MSMs of the
// LegoGroth16 CRS generator (legogroth16/src/generator.rs:335-399) and utils/src/msm.rs:8-62 (WindowTable::multiply_many,
// multiply_field_elems_with_same_group_elem).  arkworks picks the window from the batch size and keeps a ragged
// Vec<Vec<Affine>>; here the window is fixed at 8 bits so a scalar's digits are its bytes:
//
//   table   T[k][d-1] = d * 2^(8k) * B,  k < 32, d = 1..255, affine, 128 B (G1) / 256 B (G2) records — 1 / 2 MiB, stays in L2
//   mul     lane i: acc = sum_k T[k][byte_k(s_i)] (<= 32 mixed XYZZ additions), then one inversion -> affine ABI form
//
// The result is the group element s_i * B whatever the window; outputs are compared after normalisation.
#pragma once
#include <hip/hip_runtime.h>
#include "msm_kernels.hip.h"
#include "fp_inv.hip.h"
#include "ec29_two_lane.hip.h"

namespace msm {

constexpr int FB_WBITS = 8;
constexpr int FB_NW = 32;                      // 256 bits of scalar
constexpr int FB_ROW = (1 << FB_WBITS) - 1;    // digits 1..255
constexpr int FB_ENTRIES = FB_NW * FB_ROW;

template <class C> __device__ __forceinline__ void store_affine_abi(uint32_t *__restrict__ o, const Aff<typename C::F> &p) {
    const Fp *c = reinterpret_cast<const Fp *>(&p);
#pragma unroll
    for (int k = 0; k < 2 * C::NFP; k++) fp_to_abi(o + 12 * k, c[k]);
}

// window_bases: FB_NW prepared records of 2^(8k) * B (host doubles, k_prep_bases converts)
template <class C>
__global__ void __launch_bounds__(64) k_fb_table(const uint32_t *__restrict__ window_bases, uint32_t *__restrict__ table) {
    typedef typename C::F F;
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= FB_ENTRIES) return;
    const int k = e / FB_ROW, d = e % FB_ROW + 1;
    const uint32_t *rec = window_bases + (size_t)k * C::AFF_STRIDE;
    uint32_t *dst = table + (size_t)e * C::AFF_STRIDE;
    Aff<F> B; load_aff<C>(B, rec);
    Xyzz<F> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    if (!rec[2 * C::FW])
        for (int b = FB_WBITS - 1; b >= 0; b--) {
            if (!inf) { Xyzz<F> t; xyzz_dbl(t, acc); acc = t; }
            if ((d >> b) & 1) xyzz_madd(acc, inf, B, false);
        }
    Aff<F> a;
    if (inf) { fzero(a.x); fzero(a.y); } else xyzz_to_affine(a, acc);
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&a);
#pragma unroll
    for (int j = 0; j < 2 * C::FW; j++) dst[j] = w[j];
    dst[2 * C::FW] = inf ? 1u : 0u;
}

// scalars: canonical, 8 words each.  out_abi: affine x, y in the ABI form (zeros for the identity), out_inf: identity flags.
template <class C>
__global__ void __launch_bounds__(64) k_fb_mul(const uint32_t *__restrict__ table, const uint32_t *__restrict__ scalars, size_t n, uint32_t *__restrict__ out_abi, uint8_t *__restrict__ out_inf) {
    typedef typename C::F F;
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    uint32_t s[8];
    { const uint4 *sp = reinterpret_cast<const uint4 *>(scalars + i * 8); uint4 a = sp[0], b = sp[1]; s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w; }
    Xyzz<F> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (int k = 0; k < FB_NW; k++) {
        const uint32_t d = (s[k >> 2] >> (8 * (k & 3))) & 0xffu;
        if (!d) continue;
        const uint32_t *rec = table + (size_t)(k * FB_ROW + (int)d - 1) * C::AFF_STRIDE;
        if (rec[2 * C::FW]) continue;
        Aff<F> q; load_aff<C>(q, rec);
        xyzz_madd(acc, inf, q, false);
    }
    uint32_t *o = out_abi + i * (2 * C::ABI_W);
    out_inf[i] = inf ? 1 : 0;
    if (inf) {
#pragma unroll
        for (int j = 0; j < 2 * C::ABI_W; j++) o[j] = 0;
        return;
    }
    Aff<F> a; xyzz_to_affine(a, acc);
    store_affine_abi<C>(o, a);
}


// ---- out_i = A_i + s_i * P_i  (affine in, affine out) ------------------------------------------------------------------
// The GIPA folding step of the SnarkPack aggregation: `compress` (legogroth16/src/aggregation/utils.rs:34-49: vec[i] +=
// vec[i + split] * c), Key::compress and Key::scale (aggregation/key.rs:117-175) — per-element mul_bigint + add + into_affine
// under cfg_iter! in the reference.  One lane per element, double-and-add from the top set bit, one inversion.
// scalar_stride = 8 words: one canonical scalar per point; 0: the same scalar for all.  add_abi == nullptr: no addend.
template <class C>
__global__ void __launch_bounds__(64) k_mul_add(const uint32_t *__restrict__ p_abi, const uint8_t *__restrict__ p_inf, const uint32_t *__restrict__ scalars, int scalar_stride,
                                                const uint32_t *__restrict__ add_abi, const uint8_t *__restrict__ add_inf, size_t n, uint32_t *__restrict__ out_abi, uint8_t *__restrict__ out_inf) {
    typedef typename C::F F;
    constexpr int PW = 2 * C::ABI_W;
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    Aff<F> P;
    bool pinf;
    {
        const uint32_t *src = p_abi + i * PW;
        uint32_t any = 0;
        for (int k = 0; k < PW; k++) any |= src[k];
        pinf = (any == 0) || (p_inf && p_inf[i]);
        Fp *c = reinterpret_cast<Fp *>(&P);
#pragma unroll
        for (int k = 0; k < 2 * C::NFP; k++) fp_from_abi(c[k], src + 12 * k);
    }
    uint32_t s[8];
    for (int k = 0; k < 8; k++) s[k] = scalars[i * (size_t)scalar_stride + k];
    Xyzz<F> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    if (!pinf) {
        int top = -1;
        for (int k = 7; k >= 0; k--) if (s[k]) { top = 32 * k + 31 - __clz(s[k]); break; }
        for (int b = top; b >= 0; b--) {
            if (!inf) { Xyzz<F> d; xyzz_dbl(d, acc); acc = d; }
            if ((s[b >> 5] >> (b & 31)) & 1u) xyzz_madd(acc, inf, P, false);
        }
    }
    if (add_abi) {
        const uint32_t *src = add_abi + i * PW;
        uint32_t any = 0;
        for (int k = 0; k < PW; k++) any |= src[k];
        if (any != 0 && !(add_inf && add_inf[i])) {
            Aff<F> A; Fp *c = reinterpret_cast<Fp *>(&A);
#pragma unroll
            for (int k = 0; k < 2 * C::NFP; k++) fp_from_abi(c[k], src + 12 * k);
            xyzz_madd(acc, inf, A, false);
        }
    }
    uint32_t *o = out_abi + i * PW;
    out_inf[i] = inf ? 1 : 0;
    if (inf) { for (int j = 0; j < PW; j++) o[j] = 0; return; }
    Aff<F> a; xyzz_to_affine(a, acc);
    store_affine_abi<C>(o, a);
}


// G2 form of k_mul_add on lane pairs (fp2_pair.hip.h): lanes 2i and 2i+1 share point i, the even lane holds the c0 halves, the odd lane the
// c1 halves.  The kernel is latency-bound (one dependent chain of ~255 doublings per point whatever n), and a lane pair runs that chain in
// about half the instructions per lane of the one-lane Fp2 version.
__global__ void __launch_bounds__(64) k_mul_add_g2_pair(const uint32_t *__restrict__ p_abi, const uint8_t *__restrict__ p_inf, const uint32_t *__restrict__ scalars, int scalar_stride,
                                                        const uint32_t *__restrict__ add_abi, const uint8_t *__restrict__ add_inf, size_t n, uint32_t *__restrict__ out_abi, uint8_t *__restrict__ out_inf) {
    typedef Fp2H F;
    constexpr int PW = 48;
    const size_t gid = (size_t)blockIdx.x * 64 + threadIdx.x, i = gid >> 1;
    const uint32_t h = threadIdx.x & 1u;
    if (i >= n) return;
    auto load_half = [&](Aff<F> &A, const uint32_t *src) { fp_from_abi(A.x.v, src + 12 * h); fp_from_abi(A.y.v, src + 12 * (2 + h)); };
    auto all_zero = [&](const uint32_t *src) { uint32_t any = 0; for (int k = 0; k < PW; k++) any |= src[k]; return any == 0; };
    Aff<F> P; load_half(P, p_abi + i * PW);
    const bool pinf = all_zero(p_abi + i * PW) || (p_inf && p_inf[i]);
    uint32_t s[8];
    for (int k = 0; k < 8; k++) s[k] = scalars[i * (size_t)scalar_stride + k];
    Xyzz<F> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    if (!pinf) {
        int top = -1;
        for (int k = 7; k >= 0; k--) if (s[k]) { top = 32 * k + 31 - __clz(s[k]); break; }
        for (int b = top; b >= 0; b--) {
            if (!inf) { Xyzz<F> d; xyzz_dbl(d, acc); acc = d; }
            if ((s[b >> 5] >> (b & 31)) & 1u) xyzz_madd(acc, inf, P, false);
        }
    }
    if (add_abi) {
        const uint32_t *src = add_abi + i * PW;
        if (!all_zero(src) && !(add_inf && add_inf[i])) { Aff<F> A; load_half(A, src); xyzz_madd(acc, inf, A, false); }
    }
    uint32_t *o = out_abi + i * PW;
    if (h == 0) out_inf[i] = inf ? 1 : 0;
    if (inf) { for (int j = 0; j < 12; j++) { o[12 * h + j] = 0; o[12 * (2 + h) + j] = 0; } return; }
    Aff<F> a; xyzz_to_affine(a, acc);
    fp_to_abi(o + 12 * h, a.x.v); fp_to_abi(o + 12 * (2 + h), a.y.v);
}


// G1 form of k_mul_add with two adjacent lanes per point (ec29_two_lane.hip.h): both lanes hold the point and take one field operation of
// every round of the doubling / mixed addition, results swapped over DPP: 5 field operations per lane and step instead of 9 / 10.
template <class DUMMY>
__global__ void __launch_bounds__(64) k_mul_add_g1_2l(const uint32_t *__restrict__ p_abi, const uint8_t *__restrict__ p_inf, const uint32_t *__restrict__ scalars, int scalar_stride,
                                                      const uint32_t *__restrict__ add_abi, const uint8_t *__restrict__ add_inf, size_t n, uint32_t *__restrict__ out_abi, uint8_t *__restrict__ out_inf) {
    constexpr int PW = 24;
    const size_t i = ((size_t)blockIdx.x * 64 + threadIdx.x) >> 1;
    const bool B = (threadIdx.x & 1u) != 0;
    if (i >= n) return;
    auto all_zero = [&](const uint32_t *src) { uint32_t any = 0; for (int k = 0; k < PW; k++) any |= src[k]; return any == 0; };
    Aff<Fp> P; fp_from_abi(P.x, p_abi + i * PW); fp_from_abi(P.y, p_abi + i * PW + 12);
    const bool pinf = all_zero(p_abi + i * PW) || (p_inf && p_inf[i]);
    uint32_t s[8];
    for (int k = 0; k < 8; k++) s[k] = scalars[i * (size_t)scalar_stride + k];
    Xyzz<Fp> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    if (!pinf) {
        int top = -1;
        for (int k = 7; k >= 0; k--) if (s[k]) { top = 32 * k + 31 - __clz(s[k]); break; }
        for (int b = top; b >= 0; b--) {
            if (!inf) { Xyzz<Fp> d; xyzz_dbl_2l(d, acc); acc = d; }
            if ((s[b >> 5] >> (b & 31)) & 1u) xyzz_madd_2l(acc, inf, P, false);
        }
    }
    if (add_abi) {
        const uint32_t *src = add_abi + i * PW;
        if (!all_zero(src) && !(add_inf && add_inf[i])) { Aff<Fp> A; fp_from_abi(A.x, src); fp_from_abi(A.y, src + 12); xyzz_madd_2l(acc, inf, A, false); }
    }
    uint32_t *o = out_abi + i * PW;
    if (!B) out_inf[i] = inf ? 1 : 0;
    if (inf) { for (int j = 0; j < 12; j++) o[12 * (B ? 1 : 0) + j] = 0; return; }
    Aff<Fp> a; xyzz_to_affine(a, acc);
    if (!B) fp_to_abi(o, a.x); else fp_to_abi(o + 12, a.y);
}


// G2 form of k_mul_add with FOUR lanes per point: the two lane pairs of a quad both hold the point (halves on the lanes of a pair,
// fp2_pair.hip.h) and share every doubling / mixed addition (ec29_two_lane.hip.h, Share4): 5 Fp2 operations per pair and step instead of 9 / 10.
template <class DUMMY>
__global__ void __launch_bounds__(64) k_mul_add_g2_quad(const uint32_t *__restrict__ p_abi, const uint8_t *__restrict__ p_inf, const uint32_t *__restrict__ scalars, int scalar_stride,
                                                        const uint32_t *__restrict__ add_abi, const uint8_t *__restrict__ add_inf, size_t n, uint32_t *__restrict__ out_abi, uint8_t *__restrict__ out_inf) {
    typedef Fp2H F;
    constexpr int PW = 48;
    const size_t gid = (size_t)blockIdx.x * 64 + threadIdx.x, i = gid >> 2;
    const uint32_t h = threadIdx.x & 1u;
    const bool second_pair = (threadIdx.x & 2u) != 0;
    if (i >= n) return;
    auto load_half = [&](Aff<F> &A, const uint32_t *src) { fp_from_abi(A.x.v, src + 12 * h); fp_from_abi(A.y.v, src + 12 * (2 + h)); };
    auto all_zero = [&](const uint32_t *src) { uint32_t any = 0; for (int k = 0; k < PW; k++) any |= src[k]; return any == 0; };
    Aff<F> P; load_half(P, p_abi + i * PW);
    const bool pinf = all_zero(p_abi + i * PW) || (p_inf && p_inf[i]);
    uint32_t s[8];
    for (int k = 0; k < 8; k++) s[k] = scalars[i * (size_t)scalar_stride + k];
    Xyzz<F> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    if (!pinf) {
        int top = -1;
        for (int k = 7; k >= 0; k--) if (s[k]) { top = 32 * k + 31 - __clz(s[k]); break; }
        for (int b = top; b >= 0; b--) {
            if (!inf) { Xyzz<F> d; xyzz_dbl_shared<F, Share4>(d, acc); acc = d; }
            if ((s[b >> 5] >> (b & 31)) & 1u) xyzz_madd_shared<F, Share4>(acc, inf, P, false);
        }
    }
    if (add_abi) {
        const uint32_t *src = add_abi + i * PW;
        if (!all_zero(src) && !(add_inf && add_inf[i])) { Aff<F> A; load_half(A, src); xyzz_madd_shared<F, Share4>(acc, inf, A, false); }
    }
    if (second_pair) return;                             // the first pair writes the result
    uint32_t *o = out_abi + i * PW;
    if (h == 0) out_inf[i] = inf ? 1 : 0;
    if (inf) { for (int j = 0; j < 12; j++) { o[12 * h + j] = 0; o[12 * (2 + h) + j] = 0; } return; }
    Aff<F> a; xyzz_to_affine(a, acc);
    fp_to_abi(o + 12 * h, a.x.v); fp_to_abi(o + 12 * (2 + h), a.y.v);
}

// G2 form of k_mul_add with SIXTEEN lanes per point: the scalar arrives as four base-|x| digits, k = k0 + k1 |x| + k2 |x|^2 + k3 |x|^3
// (x = -0xd201000000010000 the BLS parameter, every digit < 2^64; hostf::gls4_decompose), and quad j of the 16 lanes runs the 64-step
// double-and-add of k_j B_j with B_j = |x|^j P on its two lane pairs (Share4).  B_j costs two Fp2 products: the untwist-Frobenius-twist
// map psi acts on the prime-order subgroup as multiplication by p = x (mod r), so |x|^j P = (-psi)^j (P) = (c_j(X) AX_j, c_j(Y) AY_j) with
// c_j the Fp2 conjugation for odd j and the constants below (xi^-((p-1)/3), xi^-((p-1)/2) and their conjugate products, the sign folded
// into AY_j; each one re-derived and checked against |x|^j Q by oracle/bls12_381_model.py).  A quarter of the dependent steps of the
// 255-bit chain, and 2/3 of its total work.  Points must lie in the prime-order subgroup (the invariant of arkworks' G2Affine).
__device__ const uint32_t GLS_BASE[4][4][NL] = {
  {{0x3a9fb84u, 0xba00690u, 0x71288f1u, 0xf59bcc5u, 0x126cb614u, 0x585bf36u, 0x1b85ac3du, 0x1cf856fau, 0x1891ecbdu, 0x1a7eec05u, 0x155a88f0u, 0x741ac6du, 0x1317c30fu, 0x9u},
   {0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u},
   {0x3a9fb84u, 0xba00690u, 0x71288f1u, 0xf59bcc5u, 0x126cb614u, 0x585bf36u, 0x1b85ac3du, 0x1cf856fau, 0x1891ecbdu, 0x1a7eec05u, 0x155a88f0u, 0x741ac6du, 0x1317c30fu, 0x9u},
   {0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u}},
  {{0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u},
   {0x154030c4u, 0x16aceb14u, 0x1814e947u, 0x1fba3004u, 0x2e0fc81u, 0xfb3da4fu, 0x170f2de1u, 0xacd4cbcu, 0x9689a69u, 0x110b9212u, 0x533b200u, 0x1554d884u, 0xba917a7u, 0x0u},
   {0x16620abdu, 0x12fd467cu, 0xd1f4f6fu, 0x18780c70u, 0x3a0bc76u, 0x1c749a28u, 0x9efbfa9u, 0x91b1f3u, 0xe4ddb2au, 0x286f628u, 0xe8943au, 0x981f0b0u, 0xe14367cu, 0x0u},
   {0x99d9feeu, 0x1cfab983u, 0x7e0b07eu, 0x1f87f0f2u, 0xbc18573u, 0xcdbe130u, 0x10ddd19u, 0x100cbeafu, 0x9169c21u, 0xf93673eu, 0x11de55b3u, 0x97ddc84u, 0x11fce827u, 0xcu}},
  {{0x1195dfebu, 0x1b04e484u, 0x6026044u, 0x86070a2u, 0x1fd68858u, 0x137e9670u, 0x6871e67u, 0x1e736664u, 0x83b24f6u, 0x8a70373u, 0x2a012fdu, 0x112f94bu, 0x18a2733cu, 0x3u},
   {0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u},
   {0x1c55af27u, 0x457f96fu, 0xded76fdu, 0x8a6409du, 0x1cf58bd6u, 0x3cabc21u, 0xf77f086u, 0x13a619a7u, 0x1ed28a8du, 0x179b7160u, 0x1d6c60fcu, 0xbbe20c6u, 0xcf95b94u, 0x3u},
   {0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u}},
  {{0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u, 0x0u},
   {0x1c55af27u, 0x457f96fu, 0xded76fdu, 0x8a6409du, 0x1cf58bd6u, 0x3cabc21u, 0xf77f086u, 0x13a619a7u, 0x1ed28a8du, 0x179b7160u, 0x1d6c60fcu, 0xbbe20c6u, 0xcf95b94u, 0x3u},
   {0x99d9feeu, 0x1cfab983u, 0x7e0b07eu, 0x1f87f0f2u, 0xbc18573u, 0xcdbe130u, 0x10ddd19u, 0x100cbeafu, 0x9169c21u, 0xf93673eu, 0x11de55b3u, 0x97ddc84u, 0x11fce827u, 0xcu},
   {0x16620abdu, 0x12fd467cu, 0xd1f4f6fu, 0x18780c70u, 0x3a0bc76u, 0x1c749a28u, 0x9efbfa9u, 0x91b1f3u, 0xe4ddb2au, 0x286f628u, 0xe8943au, 0x981f0b0u, 0xe14367cu, 0x0u}}};
template <class DUMMY>
__global__ void __launch_bounds__(64) k_mul_add_g2_gls(const uint32_t *__restrict__ p_abi, const uint8_t *__restrict__ p_inf, const uint32_t *__restrict__ digits, int scalar_stride,
                                                       const uint32_t *__restrict__ add_abi, const uint8_t *__restrict__ add_inf, size_t n, uint32_t *__restrict__ out_abi, uint8_t *__restrict__ out_inf) {
    typedef Fp2H F;
    constexpr int PW = 48;
    const size_t gid = (size_t)blockIdx.x * 64 + threadIdx.x, i = gid >> 4;
    const uint32_t h = threadIdx.x & 1u, j = (threadIdx.x >> 2) & 3u;
    if (i >= n) return;                                  // (all sixteen lanes of a point leave together)
    auto load_half = [&](Aff<F> &A, const uint32_t *src) { fp_from_abi(A.x.v, src + 12 * h); fp_from_abi(A.y.v, src + 12 * (2 + h)); };
    auto all_zero = [&](const uint32_t *src) { uint32_t any = 0; for (int k = 0; k < PW; k++) any |= src[k]; return any == 0; };
    Aff<F> P; load_half(P, p_abi + i * PW);
    const bool pinf = all_zero(p_abi + i * PW) || (p_inf && p_inf[i]);
    {   // B_j = (c_j(X) AX_j, c_j(Y) AY_j)
        const bool cj = (j & 1u) != 0 && h != 0;         // conjugation: the c1 half changes sign
        Fp z, nx, ny; fp_zero(z);
        fp_sub<4>(nx, z, P.x.v); fp_norm(nx, nx); fp_sub<4>(ny, z, P.y.v); fp_norm(ny, ny);
        sel(P.x.v, cj, nx, P.x.v); sel(P.y.v, cj, ny, P.y.v);
        F ax, ay;
#pragma unroll
        for (int k = 0; k < NL; k++) { ax.v.l[k] = GLS_BASE[j][h][k]; ay.v.l[k] = GLS_BASE[j][2 + h][k]; }
        F bx, by; fmul(bx, P.x, ax); fmul(by, P.y, ay);
        P.x = bx; P.y = by;
    }
    const uint32_t *s = digits + i * (size_t)scalar_stride + 2 * j;
    const uint32_t s0 = s[0], s1 = s[1];
    Xyzz<F> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    if (!pinf)
        for (int b = 63; b >= 0; b--) {
            if (!inf) { Xyzz<F> d; xyzz_dbl_shared<F, Share4>(d, acc); acc = d; }
            if (((b >= 32 ? s1 : s0) >> (b & 31)) & 1u) xyzz_madd_shared<F, Share4>(acc, inf, P, false);
        }
    // k0 B0 + k1 B1 + k2 B2 + k3 B3: quads 2, 3 hand their sums to quads 0, 1, then quad 1 to quad 0 (shuffles stay inside the 16 lanes)
#pragma unroll
    for (int d = 8; d >= 4; d >>= 1) {
        Xyzz<F> o; bool oinf;
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&acc);
        uint32_t *q = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
        for (int k = 0; k < 4 * NL; k++) q[k] = __shfl_down(w[k], d, 16);
        oinf = __shfl_down((int)inf, d, 16) != 0;
        xyzz_add(acc, inf, o, oinf);
    }
    if (j != 0) return;
    if (add_abi) {
        const uint32_t *src = add_abi + i * PW;
        if (!all_zero(src) && !(add_inf && add_inf[i])) { Aff<F> A; load_half(A, src); xyzz_madd_shared<F, Share4>(acc, inf, A, false); }
    }
    if ((threadIdx.x & 2u) != 0) return;                 // the first pair writes the result
    uint32_t *o = out_abi + i * PW;
    if (h == 0) out_inf[i] = inf ? 1 : 0;
    if (inf) { for (int k = 0; k < 12; k++) { o[12 * h + k] = 0; o[12 * (2 + h) + k] = 0; } return; }
    Aff<F> a; xyzz_to_affine(a, acc);
    fp_to_abi(o + 12 * h, a.x.v); fp_to_abi(o + 12 * (2 + h), a.y.v);
}

}  // namespace msm
