// crypto_amd/csrc/k_g1_small.hip — G1 kernels of the small-MSM path (small_kernels.hip.h)
#include "small_kernels.hip.h"
namespace msm {
template void launch_small_table<G1>(hipStream_t, const uint32_t *, size_t, uint32_t *, uint8_t *);
template void launch_small_subtable<G1>(hipStream_t, const uint32_t *, size_t, uint32_t *, uint8_t *);
template void launch_small_tree<G1>(hipStream_t, const uint32_t *, const uint8_t *, int, const uint32_t *, size_t, uint32_t *, uint8_t *, uint32_t *, uint32_t *, uint8_t *, uint8_t *);
}  // namespace msm
