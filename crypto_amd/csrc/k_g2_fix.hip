// crypto_amd/csrc/k_g2_fix.hip — G2 fix-up kernels
#include "msm_launch_impl.hip.h"
namespace msm {
template void launch_fixup<G2>(hipStream_t, uint32_t, uint32_t *, uint8_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint8_t *, size_t, const uint32_t *, uint32_t, const uint32_t *);
template void launch_fixup_heavy<G2>(hipStream_t, const uint32_t *, uint32_t, const uint32_t *, uint32_t, uint32_t, uint32_t *, uint8_t *, const uint32_t *, const uint32_t *, const uint8_t *, size_t, uint32_t *, uint32_t *, uint8_t *);
}  // namespace msm
