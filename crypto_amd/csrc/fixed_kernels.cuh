// crypto_amd/csrc/fixed_kernels.cuh — batched fixed-base scalar multiplication: out_i = s_i * B for one base B.
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
#include "msm_kernels.cuh"
#include "fp_inv.cuh"
#include "ec29_two_lane.cuh"

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


// G2 form of k_mul_add on lane pairs (fp2_pair.cuh): lanes 2i and 2i+1 share point i, the even lane holds the c0 halves, the odd lane the
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


// G1 form of k_mul_add with two adjacent lanes per point (ec29_two_lane.cuh): both lanes hold the point and take one field operation of
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
// fp2_pair.cuh) and share every doubling / mixed addition (ec29_two_lane.cuh, Share4): 5 Fp2 operations per pair and step instead of 9 / 10.
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

}  // namespace msm
