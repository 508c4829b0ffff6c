// crypto_amd/csrc/msm_launch.hip.h — host-callable launchers of the MSM kernels.  The kernels are compiled in their own
// translation units (k_*.hip, one per curve and stage, built in parallel); the drivers only see these declarations.
#pragma once
#include <hip/hip_runtime.h>
#include "msm_kernels.hip.h"

namespace msm {

// per curve (k_g1_*.hip / k_g2_*.hip)
template <class C> void launch_prep_bases(hipStream_t s, const uint32_t *abi, const uint8_t *is_inf, size_t n, uint32_t *out);
template <class C> void launch_prep_bases_fp(hipStream_t s, const uint32_t *abi, const uint8_t *is_inf, size_t n, uint32_t *out);   // 14 x 29-bit records (fixed-base kernels)
template <class C> void launch_prep_bases_raw(hipStream_t s, const uint8_t *raw, size_t stride, size_t x_off, size_t y_off, size_t inf_off, const uint8_t *is_inf, size_t n, uint32_t *out);
template <class C> void launch_accumulate(hipStream_t s, const uint32_t *bases, const uint32_t *entries, const uint32_t *off, uint32_t NB, uint32_t *bucket, uint8_t *bucket_inf,
                                          uint32_t *head, uint32_t *tail, uint32_t *head_b, uint32_t *tail_b, uint8_t *part_inf, size_t T, uint32_t CH, uint32_t dbg_mask, const uint32_t *dyn = nullptr);
template <class C> void launch_accumulate_skip_identity(hipStream_t s, const uint32_t *bases, const uint32_t *entries, const uint32_t *off, uint32_t NB, uint32_t *bucket, uint8_t *bucket_inf,
                                          uint32_t *head, uint32_t *tail, uint32_t *head_b, uint32_t *tail_b, uint8_t *part_inf, size_t T, uint32_t CH, const uint32_t *dyn, const RowMap &map);
template <class C> void launch_fixup(hipStream_t s, uint32_t NB, uint32_t *bucket, uint8_t *bucket_inf, const uint32_t *head, const uint32_t *tail, const uint32_t *head_b,
                                     const uint32_t *tail_b, const uint8_t *part_inf, size_t T, const uint32_t *off, uint32_t heavy_thr, const uint32_t *dyn = nullptr);
template <class C> void launch_fixup_heavy(hipStream_t s, const uint32_t *heavy, uint32_t heavy_cap, const uint32_t *off, uint32_t CH, uint32_t NB, uint32_t *bucket, uint8_t *bucket_inf,
                                           const uint32_t *head, const uint32_t *tail, const uint8_t *part_inf, size_t T, uint32_t *dyn, uint32_t *hpart, uint8_t *hpart_inf);
template <class C> void launch_merge_buckets(hipStream_t s, uint32_t NB, uint32_t *dst, uint8_t *dst_inf, const uint32_t *src, const uint8_t *src_inf);
template <class C> void launch_reduce_l0(hipStream_t s, unsigned NG, const uint32_t *bucket, const uint8_t *bucket_inf, uint32_t NB, int mshift, uint32_t *l1, uint8_t *l1_inf);
// the shared bucket set by bit marginals (reduce_kernels.hip.h): level 0 + the class folds; leaves P and one M per bit of the lane index in win_abi /
// win_inf (ABI XYZZ form, point 0 = P, point 1 + t = M_t) and returns the number of marginals.  cls / cls_inf: reduce_m_points(NG, ..) points / bytes.
template <class C> int launch_reduce_marginals(hipStream_t s, unsigned NG, const uint32_t *bucket, const uint8_t *bucket_inf, uint32_t NB, int mshift, uint32_t *cls, uint8_t *cls_inf, uint32_t *win_abi, uint8_t *win_inf, bool quad);
template <class C> size_t reduce_marginals_points(size_t NG);
template <class C> void launch_reduce_top(hipStream_t s, unsigned W, const uint32_t *l1, const uint8_t *l1_inf, int G, int gshift, uint32_t *win_abi, uint8_t *win_inf, int lanes);

// the small-MSM path (small_kernels.hip.h; k_g1_small.hip / k_g2_small.hip): 64 signed 4-bit windows, eight multiples per base (and, for a table kept
// with a resident handle, per sub-table 2^(64 s) P, s < 4), a tree per (super-)window
constexpr int SMALL_MSM_C = 4, SMALL_MSM_W = 64, SMALL_MSM_E = 8, SMALL_MSM_LEAVES = 128, SMALL_MSM_S = 4;
constexpr size_t SMALL_MSM_MAX_N = 8192;
// leaves per group of k_small_tree over L leaves of `subtables` sub-tables: at most 4 x subtables blocks per (super-)window, i.e. 256 blocks = one block
// per CU (the kernel holds one wave per SIMD): one more block per window would run behind the others and double the kernel's length
inline int small_per_group(size_t L, int subtables) { const int g = (int)((L + 256 * (size_t)subtables - 1) / (256 * (size_t)subtables)); return g < 2 ? 2 : g; }
template <class C> void launch_small_table(hipStream_t s, const uint32_t *bases, size_t n, uint32_t *tab, uint8_t *tab_inf);
template <class C> void launch_small_subtable(hipStream_t s, const uint32_t *bases, size_t n, uint32_t *tab, uint8_t *tab_inf);
template <class C> void launch_small_tree(hipStream_t s, const uint32_t *tab, const uint8_t *tab_inf, int subtables, const uint32_t *scalars, size_t n, uint32_t *partial, uint8_t *partial_inf,
                                          uint32_t *count, uint32_t *win_abi, uint8_t *win_inf, uint8_t *win_bad);

// precomputed-multiples tables (pre_kernels.hip.h; k_g1_pre.hip / k_g2_pre.hip)
template <class C> void launch_pre_step(hipStream_t s, const uint32_t *prev, size_t n, int c, uint32_t *tmp, uint32_t *out);
// reduce_top that also hands back the plain sum S of every pseudo-window (win_s_abi / win_s_inf), for the shared-bucket-set fold
template <class C> void launch_reduce_top_s(hipStream_t s, unsigned W, const uint32_t *l1, const uint8_t *l1_inf, int G, int gshift, uint32_t *win_abi, uint8_t *win_inf, uint32_t *win_s_abi, uint8_t *win_s_inf, int lanes);

}  // namespace msm
