// crypto_amd/csrc/msm_launch_impl.hip.h — definitions of the per-curve launchers; included by the kernel translation
// units, which explicitly instantiate the ones they own.
#pragma once
#include "msm_launch.hip.h"
#include "reduce_kernels.hip.h"

namespace msm {
// the launchers take the curve as the driver names it (G1 / G2) and instantiate the kernels for the description the MSM pipeline runs it as
// (C::MSM: G1 -> G1S, the 13 x 30-bit signed field; G2 -> G2)
template <class C> void launch_prep_bases(hipStream_t s, const uint32_t *abi, const uint8_t *is_inf, size_t n, uint32_t *out) {
    hipLaunchKernelGGL((k_prep_bases<typename C::MSM>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, abi, is_inf, n, out);
}
template <class C> void launch_prep_bases_raw(hipStream_t s, const uint8_t *raw, size_t stride, size_t x_off, size_t y_off, size_t inf_off, const uint8_t *is_inf, size_t n, uint32_t *out) {
    hipLaunchKernelGGL((k_prep_bases_raw<typename C::MSM>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, raw, stride, x_off, y_off, inf_off, is_inf, n, out);
}
// base records in the 14 x 29-bit form, for the fixed-base kernels (fixed_kernels.hip.h), which run G1 over Fp
template <class C> void launch_prep_bases_fp(hipStream_t s, const uint32_t *abi, const uint8_t *is_inf, size_t n, uint32_t *out) {
    hipLaunchKernelGGL((k_prep_bases<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, abi, is_inf, n, out);
}
template <class C> void launch_accumulate(hipStream_t s, const uint32_t *bases, const uint32_t *entries, const uint32_t *off, uint32_t NB, uint32_t *bucket, uint8_t *bucket_inf,
                                          uint32_t *head, uint32_t *tail, uint32_t *head_b, uint32_t *tail_b, uint8_t *part_inf, size_t T, uint32_t CH, uint32_t dbg_mask, const uint32_t *dyn) {
    typedef typename C::ACC A;
    hipLaunchKernelGGL((k_accumulate<A>), dim3((unsigned)((T * A::LPP + 255) / 256)), dim3(256), 0, s, bases, entries, off, NB, bucket, bucket_inf, head, tail, head_b, tail_b, part_inf, T, CH, dbg_mask, dyn, RowMap{});
}
template <class C> void launch_accumulate_skip_identity(hipStream_t s, const uint32_t *bases, const uint32_t *entries, const uint32_t *off, uint32_t NB, uint32_t *bucket, uint8_t *bucket_inf,
                                          uint32_t *head, uint32_t *tail, uint32_t *head_b, uint32_t *tail_b, uint8_t *part_inf, size_t T, uint32_t CH, const uint32_t *dyn, const RowMap &map) {
    typedef typename C::ACC A;
    hipLaunchKernelGGL((k_accumulate<A, true>), dim3((unsigned)((T * A::LPP + 255) / 256)), dim3(256), 0, s, bases, entries, off, NB, bucket, bucket_inf, head, tail, head_b, tail_b, part_inf, T, CH, 0xffffffffu, dyn, map);
}
template <class C> void launch_fixup(hipStream_t s, uint32_t NB, uint32_t *bucket, uint8_t *bucket_inf, const uint32_t *head, const uint32_t *tail, const uint32_t *head_b,
                                     const uint32_t *tail_b, const uint8_t *part_inf, size_t T, const uint32_t *off, uint32_t heavy_thr, const uint32_t *dyn) {
    if constexpr (C::NFP == 2) hipLaunchKernelGGL((k_fixup_pair<G2P>), dim3((unsigned)((2 * T + 255) / 256)), dim3(256), 0, s, NB, bucket, bucket_inf, head, tail, head_b, tail_b, part_inf, T, off, heavy_thr, dyn);   // G2: lane pairs
    else hipLaunchKernelGGL((k_fixup<typename C::MSM>), dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, NB, bucket, bucket_inf, head, tail, head_b, tail_b, part_inf, T, off, heavy_thr, dyn);
}
template <class C> void launch_fixup_heavy(hipStream_t s, const uint32_t *heavy, uint32_t heavy_cap, const uint32_t *off, uint32_t CH, uint32_t NB, uint32_t *bucket, uint8_t *bucket_inf,
                                           const uint32_t *head, const uint32_t *tail, const uint8_t *part_inf, size_t T, uint32_t *dyn, uint32_t *hpart, uint8_t *hpart_inf) {
    typedef typename C::ACC A;        // G1 -> G1; G2 -> G2P: the folds run on lane pairs like the accumulation (one general G2 addition: ~25 us instead of ~50)
    hipLaunchKernelGGL((k_fixup_heavy<A>), dim3(512), dim3(A::HEAVY_T), 0, s, heavy, heavy_cap, off, CH, NB, bucket, bucket_inf, head, tail, part_inf, T, dyn);
    hipLaunchKernelGGL((k_fixup_heavy_ranges<A>), dim3(64, 32), dim3(A::HEAVY_T), 0, s, off, head, tail, part_inf, (const uint32_t *)dyn, hpart, hpart_inf);
    hipLaunchKernelGGL((k_fixup_heavy_join<A>), dim3(32), dim3(A::HEAVY_T), 0, s, off, NB, bucket, bucket_inf, (const uint32_t *)dyn, (const uint32_t *)hpart, (const uint8_t *)hpart_inf);
}
template <class C> void launch_merge_buckets(hipStream_t s, uint32_t NB, uint32_t *dst, uint8_t *dst_inf, const uint32_t *src, const uint8_t *src_inf) {
    typedef typename C::ACC A;
    hipLaunchKernelGGL((k_merge_buckets<A>), dim3((unsigned)(((size_t)NB * A::LPP + 255) / 256)), dim3(256), 0, s, NB, dst, dst_inf, src, src_inf);
}
template <class C> void launch_reduce_l0(hipStream_t s, unsigned NG, const uint32_t *bucket, const uint8_t *bucket_inf, uint32_t NB, int mshift, uint32_t *l1, uint8_t *l1_inf) {
    const unsigned blocks = (NG + 3) / 4;          // four groups (waves) per block: whole CUs (msm_kernels.hip.h)
    if constexpr (C::NFP == 2) hipLaunchKernelGGL((k_reduce_l0_pair<G2P>), dim3(blocks), dim3(256), 0, s, bucket, bucket_inf, NB, mshift, l1, l1_inf, NG);     // G2: lane pairs
    else hipLaunchKernelGGL((k_reduce_l0<typename C::MSM>), dim3(blocks), dim3(256), 0, s, bucket, bucket_inf, NB, mshift, l1, l1_inf, NG);
}
// geometry of the class folds: quad = four members per value and 64 values per block on either curve (k_reduce_cls_quad); else one (pair of) lane(s) per value, 64 / 32 per wave
template <class C> inline void red_levels(bool quad, int &PL, int &KK) { typedef RedGeom<typename C::ACC> RG; PL = quad ? 64 : RG::PL; KK = quad ? 6 : RG::K; }
template <class C> size_t reduce_marginals_points(size_t NG) {           // (the larger of the two forms)
    typedef RedGeom<typename C::ACC> RG;
    size_t best = 0;
    for (int quad = 0; quad < 2; quad++) {
        int PL, KK; red_levels<C>(quad != 0, PL, KK);
        size_t total = (size_t)(RG::K + 2) * NG, cnt = NG; int nm = RG::K;
        while (cnt > 1) { const int steps = std::min(KK, red_log2((unsigned)cnt)); const size_t chunks = std::max<size_t>(1, cnt / PL); nm += steps; total += (size_t)(nm + 2) * chunks; cnt = chunks; }
        best = std::max(best, total + 64);
    }
    return best;
}
template <class C> int launch_reduce_marginals(hipStream_t s, unsigned NG, const uint32_t *bucket, const uint8_t *bucket_inf, uint32_t NB, int mshift, uint32_t *cls, uint8_t *cls_inf, uint32_t *win_abi, uint8_t *win_inf, bool quad) {
    typedef typename C::ACC A; typedef RedGeom<A> RG;
    hipLaunchKernelGGL((k_reduce_m0<A>), dim3((NG + 3) / 4), dim3(256), 0, s, bucket, bucket_inf, NB, mshift, cls, cls_inf, NG);
    int PL, KK; red_levels<C>(quad, PL, KK);
    unsigned cnt = NG; int nm = RG::K;
    uint32_t *in = cls; uint8_t *in_inf = cls_inf;
    while (cnt > 1) {
        const int steps = std::min(KK, red_log2(cnt));
        const unsigned chunks = std::max(1u, cnt / (unsigned)PL), npairs = 1u + (unsigned)(nm + 1) / 2u, waves = npairs * chunks;
        uint32_t *out = in + (size_t)(nm + 2) * cnt * A::XW; uint8_t *out_inf = in_inf + (size_t)(nm + 2) * cnt;
        if (quad) hipLaunchKernelGGL((k_reduce_cls_quad<A>), dim3(waves), dim3(256 * A::LPP), 0, s, (const uint32_t *)in, (const uint8_t *)in_inf, cnt, nm, out, out_inf, chunks == 1 ? 1 : 0, win_abi, win_inf);
        else hipLaunchKernelGGL((k_reduce_cls<A>), dim3((waves + 3) / 4), dim3(256), 0, s, (const uint32_t *)in, (const uint8_t *)in_inf, cnt, nm, out, out_inf, chunks == 1 ? 1 : 0, win_abi, win_inf);
        nm += steps; cnt = chunks; in = out; in_inf = out_inf;
    }
    return nm;
}
template <class C> void launch_reduce_top(hipStream_t s, unsigned W, const uint32_t *l1, const uint8_t *l1_inf, int G, int gshift, uint32_t *win_abi, uint8_t *win_inf, int lanes) {
    if (lanes == 4) {          // four members per point: the chain of general additions is 4 products deep instead of 14
        typedef typename C::ACC A;
        hipLaunchKernelGGL((k_reduce_top_quad<A>), dim3(W), dim3(256 * A::LPP), 0, s, l1, l1_inf, G, gshift, win_abi, win_inf, (uint32_t *)nullptr, (uint8_t *)nullptr);
        return;
    }
    if constexpr (C::NFP == 2) hipLaunchKernelGGL((k_reduce_top_pair<G2P>), dim3(W), dim3(64), 0, s, l1, l1_inf, G, gshift, win_abi, win_inf, (uint32_t *)nullptr, (uint8_t *)nullptr);
    else hipLaunchKernelGGL((k_reduce_top<typename C::MSM>), dim3(W), dim3(64), 0, s, l1, l1_inf, G, gshift, win_abi, win_inf, (uint32_t *)nullptr, (uint8_t *)nullptr);
}
template <class C> void launch_reduce_top_s(hipStream_t s, unsigned W, const uint32_t *l1, const uint8_t *l1_inf, int G, int gshift, uint32_t *win_abi, uint8_t *win_inf, uint32_t *win_s_abi, uint8_t *win_s_inf, int lanes) {
    if (lanes == 4) {
        typedef typename C::ACC A;
        hipLaunchKernelGGL((k_reduce_top_quad<A>), dim3(W), dim3(256 * A::LPP), 0, s, l1, l1_inf, G, gshift, win_abi, win_inf, win_s_abi, win_s_inf);
        return;
    }
    if constexpr (C::NFP == 2) hipLaunchKernelGGL((k_reduce_top_pair<G2P>), dim3(W), dim3(64), 0, s, l1, l1_inf, G, gshift, win_abi, win_inf, win_s_abi, win_s_inf);
    else hipLaunchKernelGGL((k_reduce_top<typename C::MSM>), dim3(W), dim3(64), 0, s, l1, l1_inf, G, gshift, win_abi, win_inf, win_s_abi, win_s_inf);
}
}  // namespace msm
