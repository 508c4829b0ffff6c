// crypto_amd/csrc/k_sort.hip — translation unit of the curve-independent kernels (digits, counting sort, scan, self-tests).
#include "sort_kernels.hip.h"
#include "sort_launch.hip.h"

namespace msm {
void launch_digit_codes(hipStream_t s, bool wide, const uint32_t *scalars, const uint32_t *bases, int aff_stride, int flag_word, size_t n, size_t n_pad, int c, int W, void *dig, uint32_t *bad) {
    dim3 grid((unsigned)((n_pad + 255) / 256));
    if (!wide) hipLaunchKernelGGL((k_digit_codes<uint16_t>), grid, dim3(256), 0, s, scalars, bases, aff_stride, flag_word, n, n_pad, c, W, (uint16_t *)dig, bad);
    else hipLaunchKernelGGL((k_digit_codes<uint32_t>), grid, dim3(256), 0, s, scalars, bases, aff_stride, flag_word, n, n_pad, c, W, (uint32_t *)dig, bad);
}
void launch_sort_sweep(hipStream_t s, bool wide, bool scatter, unsigned grid, size_t lds_bytes, const void *dig, size_t n, size_t n_pad, int W, int RANGES, int rb_log, uint32_t B,
                       uint32_t *cnt, const uint32_t *off, uint32_t *entries, uint32_t heavy_thr, uint32_t *heavy, uint32_t heavy_cap) {
#define SWEEP(CODE, SC) hipLaunchKernelGGL((k_sort_sweep<CODE, SC>), dim3(grid), dim3(1024), lds_bytes, s, (const CODE *)dig, n, n_pad, W, RANGES, rb_log, B, cnt, off, entries, heavy_thr, heavy, heavy_cap)
    if (!wide) { if (scatter) SWEEP(uint16_t, true); else SWEEP(uint16_t, false); }
    else { if (scatter) SWEEP(uint32_t, true); else SWEEP(uint32_t, false); }
#undef SWEEP
}
size_t scan_blocks(size_t NB) { return (NB + SCAN_B - 1) / SCAN_B; }
void launch_scan(hipStream_t s, const uint32_t *cnt, uint32_t *off, uint32_t *cursor, uint32_t *bsums, size_t NB) {
    const size_t nblk = scan_blocks(NB);
    hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nblk), dim3(SCAN_T), 0, s, cnt, off, bsums, NB);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, bsums, nblk);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)((NB + 1 + 255) / 256)), dim3(256), 0, s, off, cursor, bsums, NB, nblk);
}
void launch_dyn_chunk(hipStream_t s, const uint32_t *total, uint32_t fixed_ch, uint32_t min_chunk, uint32_t max_chunks, uint32_t lanes_per_chunk, uint32_t T_max, uint32_t *dyn, uint32_t nb_shared) {
    hipLaunchKernelGGL(k_dyn_chunk, dim3(1), dim3(64), 0, s, total, fixed_ch, min_chunk, max_chunks, lanes_per_chunk, T_max, dyn, nb_shared);
}
void launch_flag_heavy(hipStream_t s, const uint32_t *off, uint32_t NB, const uint32_t *dyn, uint32_t *heavy, uint32_t heavy_cap) {
    hipLaunchKernelGGL(k_flag_heavy, dim3((NB + 255) / 256), dim3(256), 0, s, off, NB, dyn, heavy, heavy_cap);
}
void launch_g1_scale(hipStream_t s, const uint32_t *p_abi, const uint8_t *is_inf, const uint32_t *scalars, int scalar_stride, const uint8_t *negate, size_t n, uint32_t *out_abi, uint8_t *out_inf,
                     const uint32_t *add_abi, const uint8_t *add_inf) {
    hipLaunchKernelGGL(k_g1_scale, dim3((unsigned)((4 * n + 63) / 64)), dim3(64), 0, s, p_abi, is_inf, scalars, scalar_stride, negate, n, out_abi, out_inf, add_abi, add_inf);
}
void launch_selftest_fp_mul(hipStream_t s, const uint32_t *a, const uint32_t *b, size_t n, uint32_t *out) {
    hipLaunchKernelGGL(k_selftest_fp_mul, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, a, b, n, out);
}
void launch_selftest_g1_sum(hipStream_t s, const uint32_t *pts, const uint8_t *neg, size_t n, uint32_t *out, uint8_t *out_inf) {
    hipLaunchKernelGGL(k_selftest_g1_sum, dim3(1), dim3(64), 0, s, pts, neg, n, out, out_inf);
}
}  // namespace msm
