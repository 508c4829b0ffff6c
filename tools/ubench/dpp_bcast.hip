// tools/ubench/dpp_bcast.hip — quad_perm broadcasts of values that were just produced by 64-bit multiply-adds (the pattern of a
// four-lanes-per-point field product): every lane must see lane K's value.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int K> __device__ __forceinline__ int32_t bc(int32_t x) { return __builtin_amdgcn_update_dpp(0, x, K * 0x55, 0xF, 0xF, true); }
__global__ void k(const int32_t *in, int32_t *out, int n) {
    int t = threadIdx.x;
    int32_t a[13], r0[13], r1[13], r2[13], r3[13];
    for (int i = 0; i < 13; i++) a[i] = in[t * 13 + i];
    int64_t acc = 0; int32_t p[13];
    for (int i = 0; i < 13; i++) { for (int j = 0; j <= i; j++) acc += (int64_t)a[j] * a[i - j]; p[i] = (int32_t)((uint32_t)acc & 0x3fffffffu) - (1 << 29); acc >>= 30; }
    for (int i = 0; i < 13; i++) { r0[i] = bc<0>(p[i]); r1[i] = bc<1>(p[i]); r2[i] = bc<2>(p[i]); r3[i] = bc<3>(p[i]); }
    for (int i = 0; i < 13; i++) { out[(t * 13 + i) * 5 + 0] = p[i]; out[(t * 13 + i) * 5 + 1] = r0[i]; out[(t * 13 + i) * 5 + 2] = r1[i]; out[(t * 13 + i) * 5 + 3] = r2[i]; out[(t * 13 + i) * 5 + 4] = r3[i]; }
}
int main() {
    const int T = 256; int32_t *in, *out; hipMallocManaged(&in, T * 13 * 4); hipMallocManaged(&out, T * 13 * 5 * 4);
    for (int i = 0; i < T * 13; i++) in[i] = (int32_t)((i * 2654435761u) >> 3) - (1 << 28);
    hipLaunchKernelGGL(k, dim3(1), dim3(T), 0, 0, in, out, T); hipDeviceSynchronize();
    int bad = 0;
    for (int t = 0; t < T; t++) for (int i = 0; i < 13; i++) for (int K = 0; K < 4; K++) {
        int src = (t & ~3) + K;
        if (out[(t * 13 + i) * 5 + 1 + K] != out[(src * 13 + i) * 5]) bad++;
    }
    printf("quad_perm broadcast after multiply-adds: %d mismatches of %d\n", bad, T * 13 * 4);
    return 0;
}
