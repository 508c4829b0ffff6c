// tools/ubench/affine_rate.hip — why the bucket accumulation stays on XYZZ mixed additions (8M + 2S, no inversion) instead of batched
// affine additions (5M + 1S + a shared inversion, the CPU provers' trick): throughput of both on gfx950, compute isolated from the gather.
//
// An affine addition P + Q needs 1 / (x2 - x1).  Montgomery's trick shares one inversion among K independent additions at 3 extra products
// each, but the K prefix products must stay alive between the forward and the backward sweep (14 registers each) and the operands are read
// twice.  Variants measured here, K additions per lane per round:
//   mode 0  one inversion (fp_inv_device) per lane and round                      (inversion amortised over K only)
//   mode 1  inversion replaced by a copy                                  (upper bound: an inversion that costs nothing)
//   mode 2  one inversion per 256-lane block: product tree through LDS, one wave inverts while three wait   (the realistic sharing)
// against madd<2w> of madd_rate.hip (the production kernel's core).  Arithmetic is data independent, so random limbs serve as points.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../crypto_amd/csrc/fp29.hip.h"
#include "../../crypto_amd/csrc/ec29.hip.h"
#include "../../crypto_amd/csrc/fp_inv.hip.h"
using namespace bls29;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void ld(Fp &r, const uint32_t *q) { for (int i = 0; i < NL; i++) r.l[i] = q[i]; }

template <int K, int MODE>
__global__ void __launch_bounds__(256) k_affine(const uint32_t *pts, uint32_t *o, int rounds) {
    __shared__ uint32_t tree[MODE == 2 ? 512 * NL : 1];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp accx, accy; fp_zero(accx); fp_zero(accy);
    for (int r = 0; r < rounds; r++) {
        Fp pre[K], run; fp_set_one(run);
        // forward sweep: dx_j = x2 - x1, prefix products
#pragma unroll
        for (int j = 0; j < K; j++) {
            const uint32_t *p = pts + (size_t)((t * 31 + (r * K + j) * 7) & 1023) * 32, *q = pts + (size_t)((t * 17 + (r * K + j) * 5 + 3) & 1023) * 32;
            Fp x1, x2, dx; ld(x1, p); ld(x2, q);
            fp_sub<4>(dx, x2, x1); fp_norm(dx, dx);
            pre[j] = run; Fp nr; fp_mul(nr, run, dx); run = nr;
        }
        Fp inv;
        if (MODE == 0) fp_inv_device(inv, run);
        else if (MODE == 1) inv = run;
        else {
            // block product tree: level 0 = the 256 lane totals at tree[256 + tid]; node i = node 2i * node 2i+1; root at 1
            for (int i = 0; i < NL; i++) tree[(256 + threadIdx.x) * NL + i] = run.l[i];
            __syncthreads();
            for (int w = 128; w >= 1; w >>= 1) {
                if ((int)threadIdx.x < w) { Fp a, b, c; ld(a, tree + (2 * (w + threadIdx.x)) * NL); ld(b, tree + (2 * (w + threadIdx.x) + 1) * NL); fp_mul(c, a, b); for (int i = 0; i < NL; i++) tree[(w + threadIdx.x) * NL + i] = c.l[i]; }
                __syncthreads();
            }
            if (threadIdx.x < 64) { Fp root, ri; ld(root, tree + NL); fp_inv_device(ri, root); if (threadIdx.x == 0) for (int i = 0; i < NL; i++) tree[NL + i] = ri.l[i]; }   // one wave inverts
            __syncthreads();
            for (int w = 1; w <= 128; w <<= 1) {                      // inverse of a child = inverse of the parent * sibling
                if ((int)threadIdx.x < w) { Fp ip, a, b, ia, ib; ld(ip, tree + (w + threadIdx.x) * NL); ld(a, tree + (2 * (w + threadIdx.x)) * NL); ld(b, tree + (2 * (w + threadIdx.x) + 1) * NL);
                    fp_mul(ia, ip, b); fp_mul(ib, ip, a);
                    for (int i = 0; i < NL; i++) { tree[(2 * (w + threadIdx.x)) * NL + i] = ia.l[i]; tree[(2 * (w + threadIdx.x) + 1) * NL + i] = ib.l[i]; } }
                __syncthreads();
            }
            ld(inv, tree + (256 + threadIdx.x) * NL);
            __syncthreads();
        }
        // backward sweep: 1 / dx_j, then the affine addition (operands read a second time)
#pragma unroll
        for (int j = K - 1; j >= 0; j--) {
            const uint32_t *p = pts + (size_t)((t * 31 + (r * K + j) * 7) & 1023) * 32, *q = pts + (size_t)((t * 17 + (r * K + j) * 5 + 3) & 1023) * 32;
            Fp x1, y1, x2, y2, dx, dy, idx, ni, lam, l2, x3, y3, tt;
            ld(x1, p); ld(y1, p + NL); ld(x2, q); ld(y2, q + NL);
            fp_sub<4>(dx, x2, x1); fp_norm(dx, dx);
            fp_mul(idx, inv, pre[j]); fp_mul(ni, inv, dx); inv = ni;
            fp_sub<4>(dy, y2, y1); fp_norm(dy, dy);
            fp_mul(lam, dy, idx);
            fp_sqr(l2, lam);
            fp_add(tt, x1, x2); fp_sub<8>(x3, l2, tt); fp_norm(x3, x3);
            fp_sub<16>(tt, x1, x3); fp_norm(tt, tt);
            fp_mul(y3, lam, tt); fp_sub<4>(y3, y3, y1); fp_norm(y3, y3);
            fp_add(accx, accx, x3); fp_norm(accx, accx); fp_add(accy, accy, y3); fp_norm(accy, accy);     // keep the results alive
        }
    }
    for (int i = 0; i < NL; i++) { o[(size_t)t * 2 * NL + i] = accx.l[i]; o[(size_t)t * 2 * NL + NL + i] = accy.l[i]; }
}
__global__ void __launch_bounds__(256) k_inv(const uint32_t *a, uint32_t *o, int iters) {
    Fp x; int t = blockIdx.x * blockDim.x + threadIdx.x; ld(x, a + (t & 1023) * 32);
    for (int it = 0; it < iters; it++) { Fp y; fp_inv_device(y, x); x = y; }
    for (int i = 0; i < NL; i++) o[(size_t)t * NL + i] = x.l[i];
}
template <class K> static float timeit(K launch, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < reps; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
template <int K, int MODE> static void run(const uint32_t *d, uint32_t *o, int blocks) {
    const int rounds = 64 / K;
    float ms = timeit([&] { hipLaunchKernelGGL((k_affine<K, MODE>), dim3(blocks), dim3(256), 0, 0, d, o, rounds); }, 3);
    printf("affine K=%2d mode %d  blocks=%4d  %.3f ms  %.3f Gadd/s\n", K, MODE, blocks, ms, (double)blocks * 256 * rounds * K / ms * 1e-6);
}
int main() {
    std::vector<uint32_t> h(1024 * 32); uint64_t s = 88172645463325252ULL;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s & LMASK; }
    uint32_t *d, *o; CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, (size_t)4096 * 256 * 2 * NL * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int blocks : {1024, 2048}) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_inv, dim3(blocks), dim3(256), 0, 0, d, o, 2); }, 2);
        printf("fp_inv (div. steps)     blocks=%4d  %.3f ms  %.4f Ginv/s  = %.0f fp_mul-times per inversion at 63 Gmul/s\n", blocks, ms, (double)blocks * 256 * 2 / ms * 1e-6, 63.0 / ((double)blocks * 256 * 2 / ms * 1e-6));
    }
    for (int blocks : {2048, 4096}) {
        run<4, 1>(d, o, blocks); run<8, 1>(d, o, blocks); run<16, 1>(d, o, blocks);
        run<4, 0>(d, o, blocks); run<8, 0>(d, o, blocks);
        run<4, 2>(d, o, blocks); run<8, 2>(d, o, blocks); run<16, 2>(d, o, blocks);
    }
    return 0;
}
