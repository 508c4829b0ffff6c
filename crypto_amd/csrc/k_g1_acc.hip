// crypto_amd/csrc/k_g1_acc.hip — G1 base preparation + chunked bucket accumulation kernels
#include "msm_launch_impl.hip.h"
namespace msm {
template void launch_prep_bases<G1>(hipStream_t, const uint32_t *, const uint8_t *, size_t, uint32_t *);
template void launch_prep_bases_fp<G1>(hipStream_t, const uint32_t *, const uint8_t *, size_t, uint32_t *);
template void launch_prep_bases_raw<G1>(hipStream_t, const uint8_t *, size_t, size_t, size_t, size_t, const uint8_t *, size_t, uint32_t *);
template void launch_accumulate<G1>(hipStream_t, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t, uint32_t *, uint8_t *, uint32_t *, uint32_t *, uint32_t *, uint32_t *, uint8_t *, size_t, uint32_t, uint32_t, const uint32_t *);
template void launch_accumulate_skip_identity<G1>(hipStream_t, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t, uint32_t *, uint8_t *, uint32_t *, uint32_t *, uint32_t *, uint32_t *, uint8_t *, size_t, uint32_t, const uint32_t *, const RowMap &);
}  // namespace msm
