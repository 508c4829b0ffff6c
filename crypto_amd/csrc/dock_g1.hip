// crypto_amd/csrc/dock_g1.hip — BLS12-381 G1 entry points of include/dock_gpu.h (+ the device self-tests).
#include "msm_driver.hip.h"
using namespace dock;
namespace dock {
// the one-shot pipeline without the size threshold, for callers inside the library (dock_prover.cpp: the g_d MSM of a proof)
int32_t msm_g1_nothreshold(const uint64_t *b, const uint8_t *inf, const uint64_t *s, size_t n, uint64_t out[18]) {
    if (!out || (n && (!b || !s))) return DGPU_E_BADARG;
    return msm_oneshot_here<G1, hostf::Fq>(RawBases::packed<G1>(b, inf), s, n, false, out);
}
}  // namespace dock

namespace dock {
// a host view of G1 bases as a handle for the duration of a larger call (dock_prover.cpp: dgpu_legogroth16_prove_host)
int32_t view_acquire_g1(const void *p, size_t stride, size_t x_off, size_t y_off, size_t inf_off, size_t n, int table_c, uint64_t *handle, void **pin) {
    return view_acquire<G1>(RawBases{(const uint8_t *)p, stride, x_off, y_off, inf_off, nullptr}, n, 1, table_c, handle, pin);
}
void view_release_any(void *pin) { view_release(pin); }
bool view_verify_any(void *pin) { return view_verify(pin); }
}  // namespace dock

extern "C" {
int32_t dgpu_fold_g1(const uint64_t *xyz, size_t k, uint64_t out[18]) { return host_fold_jacobian<hostf::Fq>(xyz, k, out); }
int32_t dgpu_lincomb_g1(const uint64_t *p, const uint8_t *inf, const uint64_t *s, size_t k, uint64_t out[18]) { return host_lincomb<hostf::Fq>(p, inf, s, k, out); }
int32_t dgpu_msm_g1(const uint64_t *b, const uint8_t *inf, const uint64_t *s, size_t n, uint64_t out[18]) { return msm_oneshot<G1, hostf::Fq>(RawBases::packed<G1>(b, inf), s, n, false, out); }
int32_t dgpu_msm_g1_mont(const uint64_t *b, const uint8_t *inf, const uint64_t *s, size_t n, uint64_t out[18]) { return msm_oneshot<G1, hostf::Fq>(RawBases::packed<G1>(b, inf), s, n, true, out); }
int32_t dgpu_msm_g1_strided(const void *b, size_t stride, size_t x_off, size_t y_off, size_t inf_off, const uint64_t *s, size_t n, int32_t mont, uint64_t out[18]) {
    return msm_oneshot<G1, hostf::Fq>(RawBases{(const uint8_t *)b, stride, x_off, y_off, inf_off, nullptr}, s, n, mont != 0, out); }
int32_t dgpu_bases_upload_g1_strided(const void *b, size_t stride, size_t x_off, size_t y_off, size_t inf_off, size_t n, uint64_t *h) {
    return bases_upload<G1>(RawBases{(const uint8_t *)b, stride, x_off, y_off, inf_off, nullptr}, n, h, 1); }
int32_t dgpu_reserve_g1(size_t n) { CtxScope here(cur_index()); return reserve_slots<G1>(1, n, 104, nullptr); }
int32_t dgpu_bases_upload_g1(const uint64_t *b, const uint8_t *inf, size_t n, uint64_t *h) { return bases_upload<G1>(RawBases::packed<G1>(b, inf), n, h, 1); }
int32_t dgpu_msm_g1_handle(uint64_t b, size_t off, const uint64_t *s, size_t n, int32_t mont, uint64_t out[18]) { return msm_handle<G1, hostf::Fq>(b, off, s, n, mont, out, 1); }
int32_t dgpu_msm_g1_sharded(const uint64_t *b, const uint8_t *inf, const uint64_t *s, size_t n, int32_t ngpus, uint64_t out[18]) { return msm_sharded_oneshot<G1, hostf::Fq>(b, inf, s, n, ngpus, false, out); }
int32_t dgpu_bases_upload_g1_sharded(const uint64_t *b, const uint8_t *inf, size_t n, int32_t ngpus, uint64_t *h) { return bases_upload_sharded<G1>(b, inf, n, ngpus, h, 1); }
int32_t dgpu_msm_g1_sharded_handle(uint64_t b, const uint64_t *s, size_t n, int32_t mont, uint64_t out[18]) { return msm_sharded_handle<G1, hostf::Fq>(b, s, n, mont, out, 1); }
int32_t dgpu_msm_g1_sharded_resident(uint64_t b, uint64_t s, uint64_t out[18]) { return msm_sharded_resident<G1, hostf::Fq>(b, s, out, 1); }
int32_t dgpu_bases_precompute_g1(uint64_t h, int32_t window_bits) { return bases_precompute<G1>(h, window_bits, 1); }
// shape of a precomputed table (dgpu_bases_precompute_* may have left a short handle plain: then, and for anything that is not a table, DGPU_E_BADARG)
int32_t dgpu_bases_table_shape(uint64_t handle, size_t *rows, int32_t *window_bits, int32_t *windows) {
    if (!rows || !window_bits || !windows) return DGPU_E_BADARG;
    HandleRef hb(handle);
    if (!hb.ok || (hb.h.kind != 10 && hb.h.kind != 11)) return DGPU_E_BADARG;
    const PreTable &pt = *(const PreTable *)hb.h.p;
    *rows = pt.n; *window_bits = pt.c; *windows = pt.W;
    return DGPU_OK;
}
int32_t dgpu_scalars_sort(uint64_t table, size_t boff, uint64_t s, size_t soff, size_t n, uint64_t *sorted) { return scalars_sort(table, boff, s, soff, n, sorted); }
int32_t dgpu_msm_g1_sorted(uint64_t table, uint64_t sorted, size_t row_shift, uint64_t out[18]) { return msm_sorted<G1, hostf::Fq>(table, sorted, row_shift, out, 1); }
int32_t dgpu_msm_g1_resident(uint64_t b, size_t boff, uint64_t s, size_t soff, size_t n, uint64_t out[18]) { return msm_resident<G1, hostf::Fq>(b, boff, s, soff, n, out, 1); }

}  // extern "C"
