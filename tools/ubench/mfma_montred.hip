// tools/ubench/mfma_montred.hip — the study VERDICT r4 item 10 asks for: could the REDUCTION half of a field product (m * p: multiplication by the
// constant modulus, a Toeplitz matrix-vector product per lane, i.e. a dense contraction across a wave's 64 lanes) run on the matrix cores?
// `north_star` says no MFMA on this path; this file is the measurement behind "costed and dropped" in DESIGN.md section 10, not product code.
//
// What is measured, per wave of 64 lane-products (one Montgomery reduction each):
//   A. valu      the reduction as the MSM kernels run it today: 13 x 13 = 169 v_mad_i64_i32 in 13 column chains + 13 v_mul_lo_u32 for the quotient digits
//   B. mfma      the raw matrix-core time of the same contraction in 8-bit digits: m = 49 bytes per lane, p = 48 bytes, product columns 0 .. 96:
//                D[64 x 128] = A[64 x 64] . B[64 x 128] (zero-padded Toeplitz) = 2 x 4 x 4 = 32 v_mfma_i32_32x32x16_i8 (K = 16 per instruction), or 16 of the
//                gfx950 double-K form v_mfma_i32_32x32x32_i8 where the compiler has it — NOTHING ELSE: operands already in MFMA layout, results left in it
//   C. convert   what B leaves out and the real kernel cannot: 13 x 30-bit digits -> 49 bytes (v_bfe / v_perm), the lane <-> row exchange of the A layout
//                (lanes 32..63 supply k = 8..15 of rows 0..31), the 32 x 32 output tiles back to one lane per row through LDS (97 dwords per lane
//                out, 97 in), and 97 byte-spaced columns folded into 25 thirty-bit columns with 64-bit shift-adds
// Build / run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_montred.hip -o /tmp/mfma_montred && /tmp/mfma_montred
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 512;
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

// ---- A: the VALU reduction (operand values irrelevant: the instruction stream is what is timed) ----
__global__ void __launch_bounds__(256, 2) k_valu(uint64_t *out, int32_t seed) {
    int32_t m[13], P[13];
    for (int i = 0; i < 13; i++) { m[i] = seed + (int)threadIdx.x * (i + 3); P[i] = seed * (i + 7) + 1; }
    int64_t acc = 0; uint64_t sink = 0;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < 25; k++) {
            int64_t part = 0;
#pragma unroll
            for (int i = 0; i < 13; i++) { const int j = k - i; if (j >= 0 && j < 13) part += (int64_t)m[i] * P[j]; }
            acc += part;
            if (k < 13) m[k] = (int32_t)((uint32_t)acc * 0x3ffcfffdu) >> 2;        // quotient digit of column k (v_mul_lo_u32 + shift)
            acc >>= 30;
        }
        sink += (uint64_t)acc;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink + (uint64_t)m[5];
}

// ---- B: the matrix cores alone ----
__global__ void __launch_bounds__(256, 2) k_mfma(int *out, long a0, long b0) {
    v16i acc[8];
    for (int t = 0; t < 8; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0;
    long a[4], b[4];
    for (int i = 0; i < 4; i++) { a[i] = a0 + threadIdx.x * (i + 1); b[i] = b0 ^ (threadIdx.x * (i + 5)); }
    for (int it = 0; it < ITERS; it++) {
        // one wave-reduction = 2 row tiles x 4 column tiles x 4 K-steps of 16
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int k = 0; k < 4; k++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a[k], b[(k + t) & 3], acc[t], 0, 0, 0);
    }
    int s = 0; for (int t = 0; t < 8; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
#if __has_builtin(__builtin_amdgcn_mfma_i32_32x32x32_i8)
__global__ void __launch_bounds__(256, 2) k_mfma_k32(int *out, int a0, int b0) {
    v16i acc[8];
    for (int t = 0; t < 8; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0;
    v4i a[2], b[2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) { a[i][j] = a0 + threadIdx.x * (i + j + 1); b[i][j] = b0 ^ (threadIdx.x * (i + 5 + j)); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int k = 0; k < 2; k++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[(k + t) & 1], acc[t], 0, 0, 0);
    }
    int s = 0; for (int t = 0; t < 8; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
#define HAVE_K32 1
#else
#define HAVE_K32 0
#endif

// ---- C: the conversions around B (per lane-product), without the MFMAs ----
__global__ void __launch_bounds__(256, 2) k_convert(uint64_t *out, int32_t seed) {
    __shared__ uint32_t tile[256 * 97 + 32];        // 97 dwords per lane, padded rows would be 98: the write below is strided by 97 (odd: conflict-free)
    int32_t m[13];
    for (int i = 0; i < 13; i++) m[i] = seed + (int)threadIdx.x * (i + 3);
    uint64_t sink = 0;
    uint32_t *mine = tile + threadIdx.x * 97;
    for (int it = 0; it < ITERS; it++) {
        // (1) 13 x 30-bit -> 49 bytes packed four to a register (the MFMA A operand is 8 bytes per lane and K-step)
        uint32_t by[13];
#pragma unroll
        for (int i = 0; i < 13; i++) {                 // 13 registers of four bytes: digit i spans bits 30 i .. 30 i + 29 of the 390-bit string
            const int bit = 32 * i, d0 = bit / 30, sh = bit % 30;
            uint64_t v = (uint64_t)(uint32_t)m[d0] >> sh;
            if (d0 + 1 < 13) v |= (uint64_t)(uint32_t)m[d0 + 1] << (30 - sh);
            if (d0 + 2 < 13) v |= (uint64_t)(uint32_t)m[d0 + 2] << (60 - sh);
            by[i] = (uint32_t)v;
        }
        // (2) the A layout: lanes 32 .. 63 hold k = 8 .. 15 of rows 0 .. 31 -> half of the registers change lanes
#pragma unroll
        for (int i = 0; i < 13; i += 2) by[i] = (uint32_t)__shfl_xor((int)by[i], 32, 64);
        // (3) the D tiles (column = lane % 32, sixteen rows per register set) back to one lane per row: through LDS, 97 dwords out, 97 in
#pragma unroll
        for (int c = 0; c < 97; c++) mine[c] = by[c % 13] + (uint32_t)c;
        __syncthreads();
        uint32_t col[97];
        const uint32_t *theirs = tile + ((threadIdx.x & 192) + ((threadIdx.x * 37) & 63)) * 97;
#pragma unroll
        for (int c = 0; c < 97; c++) col[c] = theirs[c];
        __syncthreads();
        // (4) 97 byte-spaced column sums -> 25 columns of 30-bit spacing: value = sum col[c] 2^(8 c)
        uint64_t wide[25];
#pragma unroll
        for (int k = 0; k < 25; k++) wide[k] = 0;
#pragma unroll
        for (int c = 0; c < 97; c++) { const int bit = 8 * c, k = bit / 30, sh = bit % 30; wide[k] += (uint64_t)col[c] << sh; }
        uint64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 25; k++) { carry += wide[k]; if (k < 13) m[k] = (int32_t)(carry & 0x3fffffff); carry >>= 30; }
        sink += carry;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink + (uint64_t)m[3];
}

template <class F> static float time_ms(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    const int blocks = 256 * 8, threads = 256;                 // two waves per SIMD resident, eight rounds of the chip
    void *buf; CK(hipMalloc(&buf, (size_t)blocks * threads * 8));
    const double wave_products = (double)blocks * (threads / 64) * ITERS;       // wave-reductions (64 lane-products each) per launch
    const double simds = 256.0 * 4;
    auto report = [&](const char *name, float ms) {
        const double ns_per_wave_red = ms * 1e6 / (wave_products / simds);       // time one SIMD spends per wave-reduction
        printf("%-34s %8.3f ms   %7.1f ns per wave-reduction per SIMD   (%.2f G lane-reductions/s)\n", name, ms, ns_per_wave_red, wave_products * 64 / (ms * 1e-3) / 1e9);
    };
    report("A valu (169 mads + 13 mul_lo)", time_ms([&] { hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(threads), 0, 0, (uint64_t *)buf, 12345); }));
    report("B mfma 32x32x16 i8 (32 / wave)", time_ms([&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, (int *)buf, 0x0102030405060708L, 0x1112131415161718L); }));
#if HAVE_K32
    report("B' mfma 32x32x32 i8 (16 / wave)", time_ms([&] { hipLaunchKernelGGL(k_mfma_k32, dim3(blocks), dim3(threads), 0, 0, (int *)buf, 0x01020304, 0x11121314); }));
#else
    printf("B' mfma 32x32x32 i8: builtin not available in this compiler\n");
#endif
    report("C conversions around B (no mfma)", time_ms([&] { hipLaunchKernelGGL(k_convert, dim3(blocks), dim3(threads), 0, 0, (uint64_t *)buf, 777); }));
    printf("verdict: the matrix path is B (or B') + C per reduction against A; it pays only if B + C < A\n");
    (void)hipFree(buf);
    return 0;
}
