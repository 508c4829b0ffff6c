// tools/ubench/instr_rate.hip — gfx950 integer/fp instruction-throughput probe used to choose the
// limb representation of the device field library (DESIGN.md "Field arithmetic").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048;

#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

__global__ void __launch_bounds__(256) k_mad_u64_u32(uint64_t *out, uint32_t a, uint32_t b) {
    uint64_t acc[8]; uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
        BODY8(OP)
#undef OP
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_mul_lo(uint64_t *out, uint32_t a, uint32_t b) {
    uint32_t acc[8]; uint32_t y = b + blockIdx.x;
    for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + a;
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
        BODY8(OP)
#undef OP
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_mul_hi(uint64_t *out, uint32_t a, uint32_t b) {
    uint32_t acc[8]; uint32_t y = b + blockIdx.x;
    for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + a;
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
        BODY8(OP)
#undef OP
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_mad_u32_u24(uint64_t *out, uint32_t a, uint32_t b) {
    uint32_t acc[8]; uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
        BODY8(OP)
#undef OP
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_mul_hi_u24(uint64_t *out, uint32_t a, uint32_t b) {
    uint32_t acc[8]; uint32_t y = b + blockIdx.x;
    for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + a;
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
        BODY8(OP)
#undef OP
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_add64(uint64_t *out, uint32_t a, uint32_t b) {
    uint32_t lo[8], hi[8]; uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int i = 0; i < 8; i++) { lo[i] = i + threadIdx.x; hi[i] = i; }
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo[i]), "+v"(hi[i]) : "v"(x), "v"(y) : "vcc");
        BODY8(OP)
#undef OP
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s += lo[i] + hi[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_add32(uint64_t *out, uint32_t a, uint32_t b) {
    uint32_t acc[8]; uint32_t y = b + blockIdx.x;
    for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + a;
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
        BODY8(OP)
#undef OP
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_lshr64(uint64_t *out, uint32_t a, uint32_t b) {
    uint64_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = ((uint64_t)(a + i) << 40) + threadIdx.x + b;
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
        BODY8(OP)
#undef OP
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_lshl_add64(uint64_t *out, uint32_t a, uint32_t b) {
    uint64_t acc[8]; uint64_t y = ((uint64_t)b << 32) + blockIdx.x + 1;
    for (int i = 0; i < 8; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(y));
        BODY8(OP)
#undef OP
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_alignbit(uint64_t *out, uint32_t a, uint32_t b) {
    uint32_t acc[8]; uint32_t y = b + blockIdx.x;
    for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + a;
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(acc[i]) : "v"(y));
        BODY8(OP)
#undef OP
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_and32(uint64_t *out, uint32_t a, uint32_t b) {
    uint32_t acc[8]; uint32_t y = ~(b + blockIdx.x);
    for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x + a;
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
        BODY8(OP)
#undef OP
    }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_fma_f64(uint64_t *out, uint32_t a, uint32_t b) {
    double acc[8]; double x = 1.0 + 1e-9 * (a + threadIdx.x), y = 1e-9 * (b + blockIdx.x);
    for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
        BODY8(OP)
#undef OP
    }
    double s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void __launch_bounds__(256) k_fma_f32(uint64_t *out, uint32_t a, uint32_t b) {
    float acc[8]; float x = 1.0f + 1e-6f * (a + threadIdx.x), y = 1e-6f * (b + blockIdx.x);
    for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
        BODY8(OP)
#undef OP
    }
    float s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
// dependent chain latency of v_mad_u64_u32 (single accumulator)
__global__ void __launch_bounds__(64) k_mad_dep(uint64_t *out, uint32_t a, uint32_t b) {
    uint64_t acc = threadIdx.x; uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int it = 0; it < ITERS * 8; it++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

typedef void (*kern_t)(uint64_t *, uint32_t, uint32_t);
static int run(const char *name, kern_t k, uint64_t *d, int blocks, int threads, double ops_per_thread) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 3u, 5u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 3u, 5u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double ops = ops_per_thread * blocks * threads;
    printf("%-16s blocks=%5d thr=%3d  %8.3f ms  %8.2f Gop/s  (%.2f lanes/clk/SIMD @2.4GHz)\n", name, blocks, threads, ms, ops / ms * 1e-6,
           ops / (ms * 1e-3) / (1024.0 * 2.4e9));
    return 0;
}
int main() {
    uint64_t *d; CK(hipMalloc(&d, sizeof(uint64_t) * 8192 * 256));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("device %s CUs=%d clock=%d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
    double opt = (double)ITERS * 8;
    for (int blocks : {1024, 2048}) {
        run("fma_f32", k_fma_f32, d, blocks, 256, opt);
        run("add_u32", k_add32, d, blocks, 256, opt);
        run("add64(co+addc)", k_add64, d, blocks, 256, opt);
        run("mad_u64_u32", k_mad_u64_u32, d, blocks, 256, opt);
        run("mul_lo_u32", k_mul_lo, d, blocks, 256, opt);
        run("mul_hi_u32", k_mul_hi, d, blocks, 256, opt);
        run("mad_u32_u24", k_mad_u32_u24, d, blocks, 256, opt);
        run("mul_hi_u32_u24", k_mul_hi_u24, d, blocks, 256, opt);
        run("fma_f64", k_fma_f64, d, blocks, 256, opt);
        run("lshrrev_b64", k_lshr64, d, blocks, 256, opt);
        run("lshl_add_u64", k_lshl_add64, d, blocks, 256, opt);
        run("alignbit_b32", k_alignbit, d, blocks, 256, opt);
        run("and_b32", k_and32, d, blocks, 256, opt);
    }
    run("mad_u64 dep 1w", k_mad_dep, d, 1, 64, opt);
    return 0;
}
