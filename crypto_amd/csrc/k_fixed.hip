// crypto_amd/csrc/k_fixed.hip — translation unit of the fixed-base kernels (G1 and G2).
#include <cstdlib>
#include "fixed_kernels.hip.h"
#include "fold_kernels.hip.h"
#include "fixed_launch.hip.h"

namespace msm {
static_assert(FIXED_TABLE_ENTRIES == FB_ENTRIES, "table size");
template <class C> void launch_fb_table(hipStream_t s, const uint32_t *window_bases, uint32_t *table) {
    hipLaunchKernelGGL((k_fb_table<C>), dim3((FB_ENTRIES + 63) / 64), dim3(64), 0, s, window_bases, table);
}
template <class C> void launch_fb_mul(hipStream_t s, const uint32_t *table, const uint32_t *scalars, size_t n, uint32_t *out_abi, uint8_t *out_inf) {
    hipLaunchKernelGGL((k_fb_mul<C>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, table, scalars, n, out_abi, out_inf);
}
template <class C> void launch_mul_add(hipStream_t s, const uint32_t *p_abi, const uint8_t *p_inf, const uint32_t *scalars, int scalar_stride, const uint32_t *add_abi, const uint8_t *add_inf,
                                       size_t n, uint32_t *out_abi, uint8_t *out_inf) {
#ifdef DGPU_DEV
    static const bool one_lane = getenv("DGPU_MULADD_ONE_LANE") != nullptr;      // development switches (compile with -DDGPU_DEV): one-lane kernels,
    static const bool two_lanes = getenv("DGPU_MULADD_G2_PAIR") != nullptr;       // one lane pair per G2 point
#else
    constexpr bool one_lane = false, two_lanes = false;
#endif
    if constexpr (C::NFP == 2) {
        if (!one_lane && two_lanes) { hipLaunchKernelGGL(k_mul_add_g2_pair, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, s, p_abi, p_inf, scalars, scalar_stride, add_abi, add_inf, n, out_abi, out_inf); return; }
        if (!one_lane) { hipLaunchKernelGGL((k_mul_add_g2_quad<C>), dim3((unsigned)((4 * n + 63) / 64)), dim3(64), 0, s, p_abi, p_inf, scalars, scalar_stride, add_abi, add_inf, n, out_abi, out_inf); return; }
    }
    if constexpr (C::NFP == 1) {
        if (!one_lane) { hipLaunchKernelGGL((k_mul_add_g1_2l<C>), dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, s, p_abi, p_inf, scalars, scalar_stride, add_abi, add_inf, n, out_abi, out_inf); return; }
    }
    hipLaunchKernelGGL((k_mul_add<C>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, p_abi, p_inf, scalars, scalar_stride, add_abi, add_inf, n, out_abi, out_inf);
}
void launch_mul_add_g2_gls(hipStream_t s, const uint32_t *p_abi, const uint8_t *p_inf, const uint32_t *digits, int scalar_stride, const uint32_t *add_abi, const uint8_t *add_inf,
                           size_t n, uint32_t *out_abi, uint8_t *out_inf) {
    hipLaunchKernelGGL((k_mul_add_g2_gls<G2>), dim3((unsigned)((16 * n + 63) / 64)), dim3(64), 0, s, p_abi, p_inf, digits, scalar_stride, add_abi, add_inf, n, out_abi, out_inf);
}
template void launch_mul_add<G1>(hipStream_t, const uint32_t *, const uint8_t *, const uint32_t *, int, const uint32_t *, const uint8_t *, size_t, uint32_t *, uint8_t *);
template void launch_mul_add<G2>(hipStream_t, const uint32_t *, const uint8_t *, const uint32_t *, int, const uint32_t *, const uint8_t *, size_t, uint32_t *, uint8_t *);
template void launch_fb_table<G1>(hipStream_t, const uint32_t *, uint32_t *);
template void launch_fb_table<G2>(hipStream_t, const uint32_t *, uint32_t *);
template void launch_fb_mul<G1>(hipStream_t, const uint32_t *, const uint32_t *, size_t, uint32_t *, uint8_t *);
template void launch_fb_mul<G2>(hipStream_t, const uint32_t *, const uint32_t *, size_t, uint32_t *, uint8_t *);
void launch_g1_scale_quad(hipStream_t s, const uint32_t *p_abi, const uint8_t *is_inf, const uint32_t *scalars, int scalar_stride, const uint8_t *negate, size_t n, uint32_t *out_abi, uint8_t *out_inf,
                          const uint32_t *add_abi, const uint8_t *add_inf) {
    // up to 4096 points (a block of sixteen per CU in one round of the chip) the chain is what a call lasts: four quads per point in four waves; beyond, the
    // one-quad form (a quarter of the lanes per point)
    if (n <= 4096) hipLaunchKernelGGL(k_g1_scale_oct, dim3((unsigned)((n + 15) / 16)), dim3(64 * SCO_WAVES), 0, s, p_abi, is_inf, scalars, scalar_stride, negate, n, out_abi, out_inf, add_abi, add_inf);
    else hipLaunchKernelGGL(k_g1_scale_quad, dim3((unsigned)((4 * n + 63) / 64)), dim3(64), 0, s, p_abi, is_inf, scalars, scalar_stride, negate, n, out_abi, out_inf, add_abi, add_inf);
}
// the folding step with its doubling chains done ahead of the scalar (fold_kernels.hip.h)
static_assert(FOLD_TABLE_WORDS_G1 == FOLD_E1 * FOLD_PW1 && FOLD_TABLE_WORDS_G2 == FOLD_E2 * FOLD_PW2, "fold table size");
void launch_fold_chain(hipStream_t s, const uint32_t *p1, size_t n1, uint32_t *tab1, uint8_t *inf1, const uint32_t *p2, size_t n2, uint32_t *tab2, uint8_t *inf2) {
    const unsigned blocks1 = (unsigned)((2 * n1 + 63) / 64), blocks2 = (unsigned)((16 * n2 + 63) / 64);
    if (blocks1 + blocks2 == 0) return;
    hipLaunchKernelGGL((k_fold_chain<G1>), dim3(blocks1 + blocks2), dim3(64), 0, s, p1, n1, tab1, inf1, blocks1, p2, n2, tab2, inf2);
}
void launch_fold_apply(hipStream_t s, bool g2, const uint32_t *tab, const uint8_t *tab_inf, const uint16_t *leaves, int T, const uint32_t *add_abi, size_t n, uint32_t *xyzz, uint8_t *out_inf, uint32_t *out_abi) {
    if (g2) {
        if (n <= 256) hipLaunchKernelGGL((k_fold_tree<G2P, 64>), dim3((unsigned)n), dim3(512), 0, s, tab, tab_inf, FOLD_E2, leaves, T, 1 << 30, add_abi, n, xyzz, out_inf);
        else hipLaunchKernelGGL((k_fold_tree<G2P, 16>), dim3((unsigned)((n + 3) / 4)), dim3(512), 0, s, tab, tab_inf, FOLD_E2, leaves, T, 1 << 30, add_abi, n, xyzz, out_inf);
        hipLaunchKernelGGL((k_fold_affine_g2<G2>), dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, s, (const uint32_t *)xyzz, (const uint8_t *)out_inf, n, out_abi);
    } else {
        if (n <= 256) hipLaunchKernelGGL((k_fold_tree<G1S, 64>), dim3((unsigned)n), dim3(256), 0, s, tab, tab_inf, FOLD_E1, leaves, T, FOLD_E1, add_abi, n, xyzz, out_inf);
        else hipLaunchKernelGGL((k_fold_tree<G1S, 16>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, tab, tab_inf, FOLD_E1, leaves, T, FOLD_E1, add_abi, n, xyzz, out_inf);
        hipLaunchKernelGGL((k_fold_affine_g1<G1>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, (const uint32_t *)xyzz, (const uint8_t *)out_inf, n, out_abi);
    }
}
}  // namespace msm
