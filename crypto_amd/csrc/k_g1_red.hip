// crypto_amd/csrc/k_g1_red.hip — G1 fix-up and bucket-reduction kernels
#include "msm_launch_impl.hip.h"
namespace msm {
template void launch_fixup<G1>(hipStream_t, uint32_t, uint32_t *, uint8_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint8_t *, size_t, const uint32_t *, uint32_t, const uint32_t *);
template void launch_fixup_heavy<G1>(hipStream_t, const uint32_t *, uint32_t, const uint32_t *, uint32_t, uint32_t, uint32_t *, uint8_t *, const uint32_t *, const uint32_t *, const uint8_t *, size_t, uint32_t *, uint32_t *, uint8_t *);
template void launch_merge_buckets<G1>(hipStream_t, uint32_t, uint32_t *, uint8_t *, const uint32_t *, const uint8_t *);
template void launch_reduce_l0<G1>(hipStream_t, unsigned, const uint32_t *, const uint8_t *, uint32_t, int, uint32_t *, uint8_t *);
template void launch_reduce_top<G1>(hipStream_t, unsigned, const uint32_t *, const uint8_t *, int, int, uint32_t *, uint8_t *, int);
template void launch_reduce_top_s<G1>(hipStream_t, unsigned, const uint32_t *, const uint8_t *, int, int, uint32_t *, uint8_t *, uint32_t *, uint8_t *, int);
template int launch_reduce_marginals<G1>(hipStream_t, unsigned, const uint32_t *, const uint8_t *, uint32_t, int, uint32_t *, uint8_t *, uint32_t *, uint8_t *, bool);
template size_t reduce_marginals_points<G1>(size_t);
}  // namespace msm
