// tools/ubench/madd_rate.hip — throughput of fp_mul and of the XYZZ mixed addition on gfx950 as a function of the
// register budget (launch bounds), isolating compute from the gather: points come from a 1024-entry table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../crypto_amd/csrc/fp29.hip.h"
#include "../../crypto_amd/csrc/ec29.hip.h"
#include "../../crypto_amd/csrc/fp30s.hip.h"
using namespace bls29;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_mul(const uint32_t *a, uint32_t *o, int iters) {
    Fp x, y; int t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < NL; i++) { x.l[i] = a[(t & 1023) * 32 + i] & LMASK; y.l[i] = a[(t & 1023) * 32 + 14 + i] & LMASK; }
    for (int it = 0; it < iters; it++) { fp_mul(x, x, y); }
    for (int i = 0; i < NL; i++) o[(size_t)t * NL + i] = x.l[i];
}
__global__ void __launch_bounds__(256) k_sqr(const uint32_t *a, uint32_t *o, int iters) {
    Fp x; int t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < NL; i++) { x.l[i] = a[(t & 1023) * 32 + i] & LMASK; }
    for (int it = 0; it < iters; it++) { fp_sqr(x, x); }
    for (int i = 0; i < NL; i++) o[(size_t)t * NL + i] = x.l[i];
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_madd(const uint32_t *pts, uint32_t *o, int iters) {
    Xyzz<Fp> acc; int t = blockIdx.x * blockDim.x + threadIdx.x; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (int it = 0; it < iters; it++) {
        Aff<Fp> p; const uint32_t *q = pts + (size_t)((t * 31 + it * 7) & 1023) * 32;
#pragma unroll
        for (int k = 0; k < 28; k += 4) { uint4 v = *reinterpret_cast<const uint4 *>(q + k); uint32_t *w = reinterpret_cast<uint32_t *>(&p); w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w; }
        xyzz_madd(acc, inf, p, (it & 1));
    }
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&acc);
    for (int i = 0; i < 56; i++) o[(size_t)i * gridDim.x * blockDim.x + t] = inf ? 0 : w[i];
}
// the same loop over the 13 x 30-bit signed field (fp30s.hip.h): records of 28 words (x[13] y[13] pad[2])
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_madd_s(const uint32_t *pts, uint32_t *o, int iters) {
    Xyzz<Fs> acc; int t = blockIdx.x * blockDim.x + threadIdx.x; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (int it = 0; it < iters; it++) {
        Aff<Fs> p; const uint32_t *q = pts + (size_t)((t * 31 + it * 7) & 1023) * 32;
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
template <class K> static float timeit(K launch, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < reps; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
    // table of 1024 valid affine points is not needed for throughput: random limbs (the arithmetic is data independent)
    std::vector<uint32_t> h(1024 * 32); uint64_t s = 88172645463325252ULL;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s & LMASK; }
    uint32_t *d, *o; CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, (size_t)8192 * 256 * 56 * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int blocks : {1024, 2048, 4096}) {
        int iters = 64;
        float ms = timeit([&] { hipLaunchKernelGGL(k_mul, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("fp_mul   blocks=%4d  %.3f ms  %.2f Gmul/s  (%.2f Tmad/s)\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6, (double)blocks * 256 * iters * 392 / ms * 1e-9);
        ms = timeit([&] { hipLaunchKernelGGL(k_sqr, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("fp_sqr   blocks=%4d  %.3f ms  %.2f Gsqr/s  (%.2f Tmad/s)\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6, (double)blocks * 256 * iters * 301 / ms * 1e-9);
    }
    for (int blocks : {1024, 2048, 4096}) {
        int iters = 32; double mads = (double)blocks * 256 * iters * (8 * 392 + 2 * 301);
        float ms;
        ms = timeit([&] { hipLaunchKernelGGL(k_madd<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("madd<1w> blocks=%4d  %.3f ms  %.3f Gmadd/s  (%.2f Tmad/s)\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6, mads / ms * 1e-9);
        ms = timeit([&] { hipLaunchKernelGGL(k_madd<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("madd<2w> blocks=%4d  %.3f ms  %.3f Gmadd/s  (%.2f Tmad/s)\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6, mads / ms * 1e-9);
        ms = timeit([&] { hipLaunchKernelGGL(k_madd_s<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("madd_s<2w> blocks=%4d  %.3f ms  %.3f Gmadd/s  (13 x 30-bit signed field)\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6);
        ms = timeit([&] { hipLaunchKernelGGL(k_madd_s<3>, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("madd_s<3w> blocks=%4d  %.3f ms  %.3f Gmadd/s  (13 x 30-bit signed field)\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6);
        ms = timeit([&] { hipLaunchKernelGGL(k_madd<3>, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("madd<3w> blocks=%4d  %.3f ms  %.3f Gmadd/s  (%.2f Tmad/s)\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6, mads / ms * 1e-9);
        ms = timeit([&] { hipLaunchKernelGGL(k_madd<4>, dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 3);
        printf("madd<4w> blocks=%4d  %.3f ms  %.3f Gmadd/s  (%.2f Tmad/s)\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6, mads / ms * 1e-9);
    }
    return 0;
}
