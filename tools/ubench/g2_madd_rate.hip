// tools/ubench/g2_madd_rate.hip — throughput of the Fp2 XYZZ mixed addition on gfx950: one lane per point (Fs2; -DFS2_KARATSUBA selects the
// three-product form) against one point per lane pair (Fs2H, what k_accumulate<G2P> runs), as a function of the register budget.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../crypto_amd/csrc/fs2_pair.hip.h"
using namespace bls29;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_madd_one(const uint32_t *pts, uint32_t *o, int iters) {
    Xyzz<Fs2> acc; int t = blockIdx.x * blockDim.x + threadIdx.x; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (int it = 0; it < iters; it++) {
        Aff<Fs2> p; const uint32_t *q = pts + (size_t)((t * 31 + it * 7) & 1023) * 64;
        uint32_t w[56];
#pragma unroll
        for (int k = 0; k < 56; k += 4) { uint4 v = *reinterpret_cast<const uint4 *>(q + k); w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w; }
        uint32_t *d = reinterpret_cast<uint32_t *>(&p);
#pragma unroll
        for (int k = 0; k < 52; k++) d[k] = (uint32_t)((int32_t)(w[k] & 0x1fffffffu) - (1 << 28));
        xyzz_madd(acc, inf, p, (it & 1));
    }
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&acc);
    for (int i = 0; i < 104; i++) o[(size_t)i * gridDim.x * blockDim.x + t] = inf ? 0 : w[i];
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_madd_pair(const uint32_t *pts, uint32_t *o, int iters) {
    Xyzz<Fs2H> acc; int t = blockIdx.x * blockDim.x + threadIdx.x; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (int it = 0; it < iters; it++) {
        Aff<Fs2H> p; const uint32_t *q = pts + (size_t)(((t >> 1) * 31 + it * 7) & 1023) * 64 + (t & 1) * 28;
        uint32_t w[28];
#pragma unroll
        for (int k = 0; k < 28; k += 4) { uint4 v = *reinterpret_cast<const uint4 *>(q + k); w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w; }
        uint32_t *d = reinterpret_cast<uint32_t *>(&p);
#pragma unroll
        for (int k = 0; k < 26; k++) d[k] = (uint32_t)((int32_t)(w[k] & 0x1fffffffu) - (1 << 28));
        xyzz_madd(acc, inf, p, (it & 1));
    }
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&acc);
    for (int i = 0; i < 52; i++) o[(size_t)i * gridDim.x * blockDim.x + t] = inf ? 0 : w[i];
}
// the lane-pair mixed addition with its four PAIRS of independent products computed as whole Karatsuba products, one per lane (fs2_pair.hip.h fmul_two):
// (U2, S2), (PPP, Q), (R (Q - X3), Y1 PPP), (ZZ PP, ZZZ PPP) — 4 x 845 + 2 x 338 multiply-adds per lane instead of 8 x 507 + 2 x 338 (-14 %)
__device__ __forceinline__ void xyzz_madd_two(Xyzz<Fs2H> &acc, bool &inf, const Aff<Fs2H> &q_in, bool neg) {
    typedef Fs2H F;
    Aff<F> q = q_in;
    fcond_neg(q.y, neg);
    if (inf) { acc.x = q.x; acc.y = q.y; fset_one(acc.zz); fset_one(acc.zzz); inf = false; return; }
    F U2, S2, Pd, Rd;
    fmul_two(U2, S2, q.x, acc.zz, q.y, acc.zzz);
    fsub<0>(Pd, U2, acc.x); fnorm(Pd, Pd);
    fsub<0>(Rd, S2, acc.y); fnorm(Rd, Rd);
    if (fmaybe_zero(Pd)) {
        if (fis_zero_exact(Pd)) {
            if (fis_zero_exact(Rd)) xyzz_dbl_affine(acc, q);
            else inf = true;
            return;
        }
    }
    F PP, PPP, Q, t, X3, Y3, t1, t2;
    fsqr(PP, Pd);
    fmul_two(PPP, Q, Pd, PP, acc.x, PP);
    fsqr(X3, Rd);
    fadd(t, Q, Q); fadd(t, t, PPP);
    fsub<0>(X3, X3, t); fnormw(X3, X3);
    fsub<0>(t, Q, X3); fnorm(t, t);
    fmul_two(t1, t2, Rd, t, acc.y, PPP);
    fsub<0>(t1, t1, t2); fnorm(Y3, t1);
    fmul_two(acc.zz, acc.zzz, acc.zz, PP, acc.zzz, PPP);
    acc.x = X3; acc.y = Y3;
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_madd_pair_two(const uint32_t *pts, uint32_t *o, int iters) {
    Xyzz<Fs2H> acc; int t = blockIdx.x * blockDim.x + threadIdx.x; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (int it = 0; it < iters; it++) {
        Aff<Fs2H> p; const uint32_t *q = pts + (size_t)(((t >> 1) * 31 + it * 7) & 1023) * 64 + (t & 1) * 28;
        uint32_t w[28];
#pragma unroll
        for (int k = 0; k < 28; k += 4) { uint4 v = *reinterpret_cast<const uint4 *>(q + k); w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w; }
        uint32_t *d = reinterpret_cast<uint32_t *>(&p);
#pragma unroll
        for (int k = 0; k < 26; k++) d[k] = (uint32_t)((int32_t)(w[k] & 0x1fffffffu) - (1 << 28));
        xyzz_madd_two(acc, inf, p, (it & 1));
    }
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&acc);
    for (int i = 0; i < 52; i++) o[(size_t)i * gridDim.x * blockDim.x + t] = inf ? 0 : w[i];
}
template <class K> static float timeit(K launch, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < reps; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
    std::vector<uint32_t> h(1024 * 64); uint64_t s = 88172645463325252ULL;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s; }
    uint32_t *d, *o; CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, (size_t)8192 * 256 * 104 * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
#ifdef FS2_KARATSUBA
    const char *form = "three-product";
#else
    const char *form = "two fused pairs";
#endif
    for (int blocks : {1024, 2048, 4096}) {
        int iters = 32; float ms;
        ms = timeit([&] { hipLaunchKernelGGL(k_madd_one<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("one lane (%s) <1w> blocks=%4d  %.3f ms  %.3f G additions/s\n", form, blocks, ms, (double)blocks * 256 * iters / ms * 1e-6);
        ms = timeit([&] { hipLaunchKernelGGL(k_madd_one<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("one lane (%s) <2w> blocks=%4d  %.3f ms  %.3f G additions/s\n", form, blocks, ms, (double)blocks * 256 * iters / ms * 1e-6);
        ms = timeit([&] { hipLaunchKernelGGL(k_madd_pair<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("lane pair <2w>            blocks=%4d  %.3f ms  %.3f G additions/s\n", blocks, ms, (double)blocks * 128 * iters / ms * 1e-6);
        ms = timeit([&] { hipLaunchKernelGGL(k_madd_pair<3>, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("lane pair <3w>            blocks=%4d  %.3f ms  %.3f G additions/s\n", blocks, ms, (double)blocks * 128 * iters / ms * 1e-6);
        // the paired-Karatsuba form, and whether it computes the same points (same residues: the outputs are balanced products, digit for digit equal)
        std::vector<uint32_t> ref((size_t)blocks * 256 * 52), got(ref.size());
        hipLaunchKernelGGL(k_madd_pair<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters); CK(hipMemcpy(ref.data(), o, ref.size() * 4, hipMemcpyDeviceToHost));
        ms = timeit([&] { hipLaunchKernelGGL(k_madd_pair_two<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        CK(hipMemcpy(got.data(), o, got.size() * 4, hipMemcpyDeviceToHost));
        size_t diff = 0; for (size_t i = 0; i < ref.size(); i++) diff += ref[i] != got[i];
        printf("lane pair, two whole Karatsuba products <2w> blocks=%4d  %.3f ms  %.3f G additions/s   (words that differ from the schoolbook form: %zu)\n", blocks, ms, (double)blocks * 128 * iters / ms * 1e-6, diff);
        ms = timeit([&] { hipLaunchKernelGGL(k_madd_pair_two<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("lane pair, two whole Karatsuba products <1w> blocks=%4d  %.3f ms  %.3f G additions/s\n", blocks, ms, (double)blocks * 128 * iters / ms * 1e-6);
    }
    return 0;
}
