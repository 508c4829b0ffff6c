// crypto_amd/csrc/sort_launch.cuh — host-callable launchers of the curve-independent kernels (k_sort.hip): digit recoding,
// LDS counting sort, histogram scan, batched G1 scaling, device self-tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace msm {
// curve independent (k_sort.hip)
void launch_digit_codes(hipStream_t s, bool wide, const uint32_t *scalars, const uint32_t *bases, int aff_stride, int flag_word, size_t n, size_t n_pad, int c, int W, void *dig);
void launch_sort_sweep(hipStream_t s, bool wide, bool scatter, unsigned grid, size_t lds_bytes, const void *dig, size_t n, size_t n_pad, int W, int RANGES, int rb_log, uint32_t B,
                       uint32_t *cnt, const uint32_t *off, uint32_t *entries, uint32_t heavy_thr, uint32_t *heavy, uint32_t heavy_cap);
void launch_scan(hipStream_t s, const uint32_t *cnt, uint32_t *off, uint32_t *cursor, uint32_t *bsums, size_t NB);
size_t scan_blocks(size_t NB);
void launch_g1_scale(hipStream_t s, const uint32_t *p_abi, const uint8_t *is_inf, const uint32_t *scalars, int scalar_stride, const uint8_t *negate, size_t n, uint32_t *out_abi, uint8_t *out_inf);
void launch_selftest_fp_mul(hipStream_t s, const uint32_t *a, const uint32_t *b, size_t n, uint32_t *out);
void launch_selftest_g1_sum(hipStream_t s, const uint32_t *pts, const uint8_t *neg, size_t n, uint32_t *out, uint8_t *out_inf);

}  // namespace msm
