// crypto_amd/csrc/dock_g2.hip — BLS12-381 G2 entry points of include/dock_gpu.h.
#include "msm_driver.hip.h"
using namespace dock;

namespace dock {
// a host view of G2 bases as a handle for the duration of a larger call (dock_prover.cpp: dgpu_legogroth16_prove_host)
int32_t view_acquire_g2(const void *p, size_t stride, size_t x_off, size_t y_off, size_t inf_off, size_t n, int table_c, uint64_t *handle, void **pin) {
    return view_acquire<G2>(RawBases{(const uint8_t *)p, stride, x_off, y_off, inf_off, nullptr}, n, 2, table_c, handle, pin);
}
}  // namespace dock

extern "C" {
int32_t dgpu_fold_g2(const uint64_t *xyz, size_t k, uint64_t out[36]) { return host_fold_jacobian<hostf::Fq2>(xyz, k, out); }
int32_t dgpu_lincomb_g2(const uint64_t *p, const uint8_t *inf, const uint64_t *s, size_t k, uint64_t out[36]) { return host_lincomb<hostf::Fq2>(p, inf, s, k, out); }
int32_t dgpu_msm_g2(const uint64_t *b, const uint8_t *inf, const uint64_t *s, size_t n, uint64_t out[36]) { return msm_oneshot<G2, hostf::Fq2>(RawBases::packed<G2>(b, inf), s, n, false, out); }
int32_t dgpu_msm_g2_mont(const uint64_t *b, const uint8_t *inf, const uint64_t *s, size_t n, uint64_t out[36]) { return msm_oneshot<G2, hostf::Fq2>(RawBases::packed<G2>(b, inf), s, n, true, out); }
int32_t dgpu_msm_g2_strided(const void *b, size_t stride, size_t x_off, size_t y_off, size_t inf_off, const uint64_t *s, size_t n, int32_t mont, uint64_t out[36]) {
    return msm_oneshot<G2, hostf::Fq2>(RawBases{(const uint8_t *)b, stride, x_off, y_off, inf_off, nullptr}, s, n, mont != 0, out); }
int32_t dgpu_bases_upload_g2_strided(const void *b, size_t stride, size_t x_off, size_t y_off, size_t inf_off, size_t n, uint64_t *h) {
    return bases_upload<G2>(RawBases{(const uint8_t *)b, stride, x_off, y_off, inf_off, nullptr}, n, h, 2); }
int32_t dgpu_reserve_g2(size_t n) { CtxScope here(cur_index()); return reserve_slots<G2>(1, n, 200, nullptr); }
int32_t dgpu_bases_upload_g2(const uint64_t *b, const uint8_t *inf, size_t n, uint64_t *h) { return bases_upload<G2>(RawBases::packed<G2>(b, inf), n, h, 2); }
int32_t dgpu_msm_g2_handle(uint64_t b, size_t off, const uint64_t *s, size_t n, int32_t mont, uint64_t out[36]) { return msm_handle<G2, hostf::Fq2>(b, off, s, n, mont, out, 2); }
int32_t dgpu_msm_g2_sharded(const uint64_t *b, const uint8_t *inf, const uint64_t *s, size_t n, int32_t ngpus, uint64_t out[36]) { return msm_sharded_oneshot<G2, hostf::Fq2>(b, inf, s, n, ngpus, false, out); }
int32_t dgpu_bases_upload_g2_sharded(const uint64_t *b, const uint8_t *inf, size_t n, int32_t ngpus, uint64_t *h) { return bases_upload_sharded<G2>(b, inf, n, ngpus, h, 2); }
int32_t dgpu_msm_g2_sharded_handle(uint64_t b, const uint64_t *s, size_t n, int32_t mont, uint64_t out[36]) { return msm_sharded_handle<G2, hostf::Fq2>(b, s, n, mont, out, 2); }
int32_t dgpu_msm_g2_sharded_resident(uint64_t b, uint64_t s, uint64_t out[36]) { return msm_sharded_resident<G2, hostf::Fq2>(b, s, out, 2); }
int32_t dgpu_bases_precompute_g2(uint64_t h, int32_t window_bits) { return bases_precompute<G2>(h, window_bits, 2); }
int32_t dgpu_msm_g2_sorted(uint64_t table, uint64_t sorted, size_t row_shift, uint64_t out[36]) { return msm_sorted<G2, hostf::Fq2>(table, sorted, row_shift, out, 2); }
int32_t dgpu_msm_g2_resident(uint64_t b, size_t boff, uint64_t s, size_t soff, size_t n, uint64_t out[36]) { return msm_resident<G2, hostf::Fq2>(b, boff, s, soff, n, out, 2); }
}  // extern "C"
