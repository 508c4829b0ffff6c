// crypto_amd/csrc/k_g2_pre.hip — G2 precomputed-multiples table kernels
#include "pre_kernels.hip.h"
#include "msm_launch.hip.h"
namespace msm {
template <class C> void launch_pre_step(hipStream_t s, const uint32_t *prev, size_t n, int c, uint32_t *tmp, uint32_t *out) {
    hipLaunchKernelGGL((k_pre_dbl<typename C::MSM>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, prev, n, c, tmp);
    const size_t groups = (n + PRE_GROUP - 1) / PRE_GROUP;
    hipLaunchKernelGGL((k_pre_norm<typename C::MSM>), dim3((unsigned)((groups + 63) / 64)), dim3(64), 0, s, prev, (const uint32_t *)tmp, n, out);
}
template void launch_pre_step<G2>(hipStream_t, const uint32_t *, size_t, int, uint32_t *, uint32_t *);
}  // namespace msm
