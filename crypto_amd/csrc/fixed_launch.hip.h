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
// out_i = [A_i +] s_i P_i, G1, the scalars as (k1 | k2) of the GLV split (hostf::glv_decompose): ONE joint chain on four lanes per point (fold_kernels.hip.h);
// same arguments and results as launch_g1_scale (sort_launch.hip.h), which it replaces at the call sites
void launch_g1_scale_quad(hipStream_t s, const uint32_t *p_abi, const uint8_t *is_inf, const uint32_t *scalars, int scalar_stride, const uint8_t *negate, size_t n, uint32_t *out_abi, uint8_t *out_inf,
                          const uint32_t *add_abi = nullptr, const uint8_t *add_inf = nullptr);
// G2 with the scalars as four base-|x| digits each (hostf::gls4_decompose): sixteen lanes per point
void launch_mul_add_g2_gls(hipStream_t s, const uint32_t *p_abi, const uint8_t *p_inf, const uint32_t *digits, int scalar_stride, const uint32_t *add_abi, const uint8_t *add_inf,
                           size_t n, uint32_t *out_abi, uint8_t *out_inf);
// out_i = A_i + c P_i with the doubling chains of the P_i done before c exists (fold_kernels.hip.h): the table of a point set (XYZZ entries as ABI words),
// then per scalar the tree over the entries `leaves` names (G1: entry 128 + k = phi(2^k P)) and the conversion to affine
constexpr size_t FOLD_TABLE_WORDS_G1 = 128 * 56, FOLD_TABLE_WORDS_G2 = 256 * 112;      // u32 words per point (entries x four coordinates of 14 limbs, G2: eight halves)
void launch_fold_chain(hipStream_t s, const uint32_t *p1, size_t n1, uint32_t *tab1, uint8_t *inf1, const uint32_t *p2, size_t n2, uint32_t *tab2, uint8_t *inf2);      // G1 and G2 point sets in one launch (either may be empty)
void launch_fold_apply(hipStream_t s, bool g2, const uint32_t *tab, const uint8_t *tab_inf, const uint16_t *leaves, int T, const uint32_t *add_abi, size_t n, uint32_t *xyzz, uint8_t *out_inf, uint32_t *out_abi);
}  // namespace msm
