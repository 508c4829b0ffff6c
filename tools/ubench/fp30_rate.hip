// tools/ubench/fp30_rate.hip — go / no-go measurement for a 13 x 30-bit SIGNED-limb base field against the production 14 x 29-bit unsigned one:
// dependent chains of Montgomery products at the accumulation kernel's occupancy (2 waves / SIMD).  338 instead of 392 multiply-adds per product
// (v_mad_i64_i32 instead of v_mad_u64_u32), 25 instead of 27 column hand-offs.  Column bound: 26 products of magnitude <= 2^58 < 2^63.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../crypto_amd/csrc/fp29.hip.h"
using namespace bls29;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

namespace f30 {
constexpr int N = 13, B = 30;
constexpr uint32_t INV30 = 0x3ffcfffdu;      // -p^-1 mod 2^30
#define F30_P {-21845, -402915328, 356515836, -352321620, -252304353, 55215067, 288093811, 316751073, -321428361, 517541167, -375082566, -91332614, 1704210}
struct Fs { int32_t l[N]; };
__device__ __forceinline__ int32_t sext30(uint32_t x) { return (int32_t)(x << 2) >> 2; }
// r = a b / 2^390 mod p, signed digits; inputs |l| <= 2^29 (+ small), output digits in [-2^29, 2^29) except the top one
__device__ __forceinline__ void mul(Fs &r, const Fs &a, const Fs &b) {
    constexpr int32_t P_[N] = F30_P;
    int32_t m[N], t[N];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (int64_t)m[i] * P_[k - i];
        m[k] = sext30((uint32_t)acc * INV30);
        acc += (int64_t)m[k] * P_[0];
        acc >>= B;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - N + 1; i < N; i++) acc += (int64_t)m[i] * P_[k - i];
        t[k - N] = sext30((uint32_t)acc);
        acc = (acc - t[k - N]) >> B;
    }
    t[N - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
}
// variant: unsigned low digits for the output (mask instead of sign extension; carry = plain arithmetic shift): digits in [0, 2^30)
__device__ __forceinline__ void mul_u(Fs &r, const Fs &a, const Fs &b) {
    constexpr int32_t P_[N] = F30_P;
    int32_t m[N], t[N];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (int64_t)m[i] * P_[k - i];
        m[k] = sext30((uint32_t)acc * INV30);
        acc += (int64_t)m[k] * P_[0];
        acc >>= B;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - N + 1; i < N; i++) acc += (int64_t)m[i] * P_[k - i];
        t[k - N] = (int32_t)((uint32_t)acc & 0x3fffffffu);
        acc >>= B;
    }
    t[N - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
}
}  // namespace f30

template <int WAVES> __global__ void __launch_bounds__(256, WAVES) k_mul29(const uint32_t *a, uint32_t *o, int iters) {
    Fp x, y; int t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < NL; i++) { x.l[i] = a[(t & 1023) * 32 + i] & LMASK; y.l[i] = a[(t & 1023) * 32 + 14 + i] & LMASK; }
    for (int it = 0; it < iters; it++) { fp_mul(x, x, y); }
    for (int i = 0; i < NL; i++) o[(size_t)t * NL + i] = x.l[i];
}
template <int WAVES, int V> __global__ void __launch_bounds__(256, WAVES) k_mul30(const uint32_t *a, uint32_t *o, int iters) {
    f30::Fs x, y; int t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < f30::N; i++) { x.l[i] = (int32_t)(a[(t & 1023) * 32 + i] & 0x1fffffffu) - (1 << 28); y.l[i] = (int32_t)(a[(t & 1023) * 32 + 14 + i] & 0x1fffffffu) - (1 << 28); }
    for (int it = 0; it < iters; it++) { if (V == 0) f30::mul(x, x, y); else f30::mul_u(x, x, y); }
    for (int i = 0; i < f30::N; i++) o[(size_t)t * NL + i] = (uint32_t)x.l[i];
}
template <class K> static float timeit(K launch, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < reps; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
    std::vector<uint32_t> h(1024 * 32); uint64_t s = 88172645463325252ULL;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s & LMASK; }
    uint32_t *d, *o; CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, (size_t)8192 * 256 * 16 * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // one product checked on the host: x y / 2^390 mod p (signed-digit value)
    {
        hipLaunchKernelGGL((k_mul30<2, 0>), dim3(1), dim3(256), 0, 0, d, o, 1);
        std::vector<uint32_t> out(256 * NL); CK(hipMemcpy(out.data(), o, out.size() * 4, hipMemcpyDeviceToHost));
        printf("check t=0:");
        for (int i = 0; i < 13; i++) printf(" %d", (int32_t)out[i]);
        printf("\nin x:");
        for (int i = 0; i < 13; i++) printf(" %d", (int32_t)(h[i] & 0x1fffffffu) - (1 << 28));
        printf("\nin y:");
        for (int i = 0; i < 13; i++) printf(" %d", (int32_t)(h[14 + i] & 0x1fffffffu) - (1 << 28));
        printf("\n");
    }
    for (int blocks : {2048, 4096}) {
        int iters = 64; float ms;
        ms = timeit([&] { hipLaunchKernelGGL((k_mul29<2>), dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 5);
        printf("fp29 mul      <2w> blocks=%4d  %.3f ms  %.2f Gmul/s\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6);
        ms = timeit([&] { hipLaunchKernelGGL((k_mul30<2, 0>), dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 5);
        printf("fp30s mul     <2w> blocks=%4d  %.3f ms  %.2f Gmul/s\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6);
        ms = timeit([&] { hipLaunchKernelGGL((k_mul30<2, 1>), dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 5);
        printf("fp30s mul_u   <2w> blocks=%4d  %.3f ms  %.2f Gmul/s\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6);
        ms = timeit([&] { hipLaunchKernelGGL((k_mul29<4>), dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 5);
        printf("fp29 mul      <4w> blocks=%4d  %.3f ms  %.2f Gmul/s\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6);
        ms = timeit([&] { hipLaunchKernelGGL((k_mul30<4, 0>), dim3(blocks), dim3(256), 0, 0, d, o, iters); }, 5);
        printf("fp30s mul     <4w> blocks=%4d  %.3f ms  %.2f Gmul/s\n", blocks, ms, (double)blocks * 256 * iters / ms * 1e-6);
    }
    return 0;
}
