// tools/ubench/clock_under_load.hip — the shader clock the chip actually holds while a kernel runs: s_memtime (core clock ticks) against
// s_memrealtime (constant-rate wall clock) inside the kernel.  Question behind it: the mixed addition loses 4.5 % of its
// instructions (carry-seeded column chains, fp30s.hip.h FS_SERIAL_LOW / FS_ALT_HIGH) and runs no faster — is k_accumulate bound by instruction
// issue at the nominal 2.4 GHz, or by the clock the power limit allows under a stream of 64-bit multiply-adds?
// Build / run on the GPU box:  hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/clock_under_load.hip -o /tmp/clk && /tmp/clk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../crypto_amd/csrc/fp29.hip.h"
#include "../../crypto_amd/csrc/ec29.hip.h"
#include "../../crypto_amd/csrc/fp30s.hip.h"
using namespace bls29;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Stamp { long long c0, w0, c1, w1; };
__device__ __forceinline__ void stamp_begin(Stamp *s) { if (threadIdx.x == 0) { s[blockIdx.x].c0 = clock64(); s[blockIdx.x].w0 = wall_clock64(); } }
__device__ __forceinline__ void stamp_end(Stamp *s) { if (threadIdx.x == 0) { s[blockIdx.x].c1 = clock64(); s[blockIdx.x].w1 = wall_clock64(); } }

// the mixed addition of k_accumulate (13 x 30-bit signed field), operands from a small table: compute only
__global__ void __launch_bounds__(256, 2) k_madd_s(const uint32_t *pts, uint32_t *o, int iters, Stamp *st) {
    stamp_begin(st);
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
    stamp_end(st);
}
// nothing but v_mad_i64_i32, eight independent chains per lane
__global__ void __launch_bounds__(256, 2) k_mad_only(uint32_t *o, int iters, int a, int b, Stamp *st) {
    stamp_begin(st);
    long long acc[8]; int x[8];
    for (int i = 0; i < 8; i++) { acc[i] = threadIdx.x + i; x[i] = a + (int)threadIdx.x * (i + 1) + b; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 48; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] += (long long)x[i] * (int)acc[(i + 3) & 7];      // (the multiplier is data: nothing to fold)
    }
    long long s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
    stamp_end(st);
}
// nothing but 32-bit additions (v_add_u32 / v_xor), eight chains per lane
__global__ void __launch_bounds__(256, 2) k_add_only(uint32_t *o, int iters, uint32_t a, Stamp *st) {
    stamp_begin(st);
    uint32_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x * (i + 1) + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 48; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = (acc[i] + a) ^ (uint32_t)(r + i);
    }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
    stamp_end(st);
}

int main() {
    int wall_khz = 0; CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    printf("device: nominal shader clock %d kHz, wall clock %d kHz\n", clk_khz, wall_khz);
    std::vector<uint32_t> h(1024 * 32); uint64_t s = 88172645463325252ULL;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s; }
    const int blocks = 2048;
    uint32_t *d, *o; Stamp *st;
    CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, (size_t)blocks * 256 * 56 * 4)); CK(hipMalloc(&st, blocks * sizeof(Stamp)));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<Stamp> hs(blocks);
    auto report = [&](const char *name, float ms, double unit_per_launch, const char *unit) {
        hipMemcpy(hs.data(), st, blocks * sizeof(Stamp), hipMemcpyDeviceToHost);
        double f = 0; int n = 0;
        for (auto &q : hs) { const double dw = (double)(q.w1 - q.w0), dc = (double)(q.c1 - q.c0); if (dw > 0) { f += dc / (dw / (wall_khz * 1e3)); n++; } }
        printf("%-28s %8.3f ms   %8.3f G%s/s   shader clock while running: %.0f MHz (mean over %d blocks)\n", name, ms, unit_per_launch / ms * 1e-6, unit, f / n * 1e-6, n);
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 3; round++) {
        for (int iters : {64, 512}) {
            float ms;
            hipLaunchKernelGGL(k_madd_s, dim3(blocks), dim3(256), 0, 0, d, o, iters, st); hipDeviceSynchronize();
            hipEventRecord(e0); hipLaunchKernelGGL(k_madd_s, dim3(blocks), dim3(256), 0, 0, d, o, iters, st); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            char nm[64]; snprintf(nm, sizeof nm, "mixed addition x %d", iters); report(nm, ms, (double)blocks * 256 * iters, "madd");
            hipEventRecord(e0); hipLaunchKernelGGL(k_mad_only, dim3(blocks), dim3(256), 0, 0, o, iters * 8, 12345, 777, st); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            snprintf(nm, sizeof nm, "v_mad_i64_i32 only x %d", iters * 8); report(nm, ms, (double)blocks * 256 * iters * 8 * 384, "mad");
            hipEventRecord(e0); hipLaunchKernelGGL(k_add_only, dim3(blocks), dim3(256), 0, 0, o, iters * 8, 12345u, st); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            snprintf(nm, sizeof nm, "v_add / v_xor only x %d", iters * 8); report(nm, ms, (double)blocks * 256 * iters * 8 * 768, "op");
        }
    }
    return 0;
}
