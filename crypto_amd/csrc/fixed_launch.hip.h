// crypto_amd/csrc/fixed_launch.hip.h — host-callable launchers of the fixed-base kernels (k_fixed.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace msm {
constexpr int FIXED_TABLE_ENTRIES = 32 * 255;
template <class C> void launch_fb_table(hipStream_t s, const uint32_t *window_bases, uint32_t *table);
template <class C> void launch_fb_mul(hipStream_t s, const uint32_t *table, const uint32_t *scalars, size_t n, uint32_t *out_abi, uint8_t *out_inf);
template <class C> void launch_mul_add(hipStream_t s, const uint32_t *p_abi, const uint8_t *p_inf, const uint32_t *scalars, int scalar_stride, const uint32_t *add_abi, const uint8_t *add_inf,
                                       size_t n, uint32_t *out_abi, uint8_t *out_inf);
// G2 with the scalars as four base-|x| digits each (hostf::gls4_decompose): sixteen lanes per point
void launch_mul_add_g2_gls(hipStream_t s, const uint32_t *p_abi, const uint8_t *p_inf, const uint32_t *digits, int scalar_stride, const uint32_t *add_abi, const uint8_t *add_inf,
                           size_t n, uint32_t *out_abi, uint8_t *out_inf);
}  // namespace msm
