// crypto_amd/csrc/pre_kernels.hip.h — precomputed-multiples tables for resident bases (a proving-key query is fixed for the life of the key:
// legogroth16/src/data_structures.rs:151-168; every proof runs msm_bigint over the same bases, prover.rs:286,299,592).
//
// table[w][i] = 2^(c w) P_i in the prepared affine record form, w < W.  With it digit w of scalar i adds table[w][i] into ONE bucket set shared
// by all windows: n * W additions into 2^(c-1) buckets instead of W sets of them, so wide windows (c = 20: W = 13 instead of 16 additions per
// term) no longer multiply the bucket-reduction work, and the host's Horner fold over the windows disappears.
// Built once per key, one window step at a time:  k_pre_dbl (c doublings per point, XYZZ) then k_pre_norm (back to affine with ONE field
// inversion per 8 points: Montgomery's trick inside a lane).
#pragma once
#include <hip/hip_runtime.h>
#include "msm_kernels.hip.h"
#include "fp_inv.hip.h"

namespace msm {
using namespace bls29;

constexpr int PRE_GROUP = 8;

// the base-field value whose inverse yields 1 / a, and 1 / a from it
__device__ __forceinline__ void inv_key(Fp &k, const Fp &a) { fp_norm(k, a); }
__device__ __forceinline__ void inv_key(Fp &k, const Fp2 &a) { Fp n0, n1, t; fp_sqr(n0, a.c0); fp_sqr(n1, a.c1); fp_add(t, n0, n1); fp_norm(k, t); }
__device__ __forceinline__ void inv_from_key(Fp &r, const Fp &, const Fp &kinv) { r = kinv; }
// the signed 30-bit field borrows the 14 x 29-bit field's division-step inversion (fp_safegcd.hip.h): one conversion each way per inverted value
__device__ __forceinline__ void inv_key(Fp &k, const Fs &a) { fp_from_fs(k, a); }
__device__ __forceinline__ void inv_from_key(Fs &r, const Fs &, const Fp &kinv) { fs_from_fp(r, kinv); }
__device__ __forceinline__ void inv_key(Fp &k, const Fs2 &a) { Fs n0, n1, t; fs_sqr(n0, a.c0); fs_sqr(n1, a.c1); fs_add(t, n0, n1); fs_bal(t, t); fp_from_fs(k, t); }     // the norm c0^2 + c1^2
__device__ __forceinline__ void inv_from_key(Fs2 &r, const Fs2 &a, const Fp &kinv) {
    Fs ki, n1; fs_from_fp(ki, kinv); fs_mul(r.c0, a.c0, ki); fs_mul(n1, a.c1, ki); fs_neg(r.c1, n1);
}
__device__ __forceinline__ void inv_from_key(Fp2 &r, const Fp2 &a, const Fp &kinv) {
    Fp n1, z; fp_mul(r.c0, a.c0, kinv); fp_mul(n1, a.c1, kinv); fp_zero(z); fp_sub<4>(r.c1, z, n1); fp_norm(r.c1, r.c1);
}

// tmp[i] = 2^c * prev[i] (XYZZ, AoS: C::XW words per point); identity records are skipped (k_pre_norm copies the flag)
template <class C>
__global__ void __launch_bounds__(64) k_pre_dbl(const uint32_t *__restrict__ prev, size_t n, int c, uint32_t *__restrict__ tmp) {
    typedef typename C::F F;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *rec = prev + i * C::AFF_STRIDE;
    if (rec[C::FLAGW] != 0) return;
    Aff<F> p; load_aff<C>(p, rec);
    Xyzz<F> acc; xyzz_dbl_affine(acc, p);
    for (int k = 1; k < c; k++) { Xyzz<F> d; xyzz_dbl(d, acc); acc = d; }
    uint32_t *dst = tmp + i * C::XW;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&acc);
#pragma unroll
    for (int k = 0; k < C::XW; k += 4) *reinterpret_cast<uint4 *>(dst + k) = make_uint4(w[k], w[k + 1], w[k + 2], w[k + 3]);
}

template <class C> __device__ __forceinline__ void load_xyzz_aos(Xyzz<typename C::F> &p, const uint32_t *__restrict__ src) {
    uint32_t *w = reinterpret_cast<uint32_t *>(&p);
#pragma unroll
    for (int k = 0; k < C::XW; k += 4) { uint4 v = *reinterpret_cast<const uint4 *>(src + k); w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w; }
}

// out[i] = affine record of tmp[i]; thread t owns points [8 t, 8 t + 8)
template <class C>
__global__ void __launch_bounds__(64) k_pre_norm(const uint32_t *__restrict__ prev, const uint32_t *__restrict__ tmp, size_t n, uint32_t *__restrict__ out) {
    typedef typename C::F F;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i0 = t * PRE_GROUP;
    if (i0 >= n) return;
    Fp pre[PRE_GROUP], run; fp_set_one(run);
    bool live[PRE_GROUP];
#pragma unroll
    for (int k = 0; k < PRE_GROUP; k++) {
        const size_t i = i0 + k;
        live[k] = i < n && prev[i * C::AFF_STRIDE + C::FLAGW] == 0;
        pre[k] = run;
        if (live[k]) {
            F zzz; const uint32_t *src = tmp + i * C::XW + 3 * C::FW; uint32_t *w = reinterpret_cast<uint32_t *>(&zzz);
            for (int j = 0; j < C::FW; j++) w[j] = src[j];
            Fp key; inv_key(key, zzz);
            Fp nr; fp_mul(nr, run, key); run = nr;
        }
    }
    Fp inv; fp_inv_device(inv, run);
#pragma unroll
    for (int k = PRE_GROUP - 1; k >= 0; k--) {
        const size_t i = i0 + k;
        if (i >= n) continue;
        uint32_t *dst = out + i * C::AFF_STRIDE;
        if (!live[k]) { for (int j = 0; j < C::AFF_STRIDE; j++) dst[j] = prev[i * C::AFF_STRIDE + j]; continue; }     // identity stays identity
        Xyzz<F> p; load_xyzz_aos<C>(p, tmp + i * C::XW);
        Fp key, kinv, ni; inv_key(key, p.zzz);
        fp_mul(kinv, inv, pre[k]);                 // 1 / key_k
        fp_mul(ni, inv, key); inv = ni;            // drop key_k from the running inverse
        F i3, tt, i2, xn, yn; Aff<F> a;
        inv_from_key(i3, p.zzz, kinv);
        fmul(tt, p.zz, i3); fsqr(i2, tt);          // 1 / ZZ = (ZZ / ZZZ)^2
        fnorm(xn, p.x); fnorm(yn, p.y);
        fmul(a.x, xn, i2); fmul(a.y, yn, i3);
        store_aff_record<C>(dst, a);
    }
}

}  // namespace msm
