// crypto_amd/csrc/k_g2_red.hip — G2 bucket-reduction kernels
#include "msm_launch_impl.hip.h"
namespace msm {
template void launch_merge_buckets<G2>(hipStream_t, uint32_t, uint32_t *, uint8_t *, const uint32_t *, const uint8_t *);
template void launch_reduce_l0<G2>(hipStream_t, unsigned, const uint32_t *, const uint8_t *, uint32_t, int, uint32_t *, uint8_t *);
template void launch_reduce_top<G2>(hipStream_t, unsigned, const uint32_t *, const uint8_t *, int, int, uint32_t *, uint8_t *, int);
template void launch_reduce_top_s<G2>(hipStream_t, unsigned, const uint32_t *, const uint8_t *, int, int, uint32_t *, uint8_t *, uint32_t *, uint8_t *, int);
template int launch_reduce_marginals<G2>(hipStream_t, unsigned, const uint32_t *, const uint8_t *, uint32_t, int, uint32_t *, uint8_t *, uint32_t *, uint8_t *, bool);
template size_t reduce_marginals_points<G2>(size_t);
}  // namespace msm
