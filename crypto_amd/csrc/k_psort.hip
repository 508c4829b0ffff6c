// crypto_amd/csrc/k_psort.hip — translation unit of the two-level partition sort (psort_kernels.hip.h)
#include <atomic>
#include "psort_kernels.hip.h"
#include "sort_launch.hip.h"

namespace msm {
void launch_id_flags(hipStream_t s, const uint32_t *bases, int aff_stride, int flag_word, size_t n, uint8_t *out) {
    if (n) hipLaunchKernelGGL(k_id_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bases, aff_stride, flag_word, n, out);
}
void launch_raw_record_hash(hipStream_t s, const uint8_t *raw, size_t stride, size_t x_off, size_t y_off, size_t inf_off, const uint8_t *is_inf, int words, size_t n, uint64_t *out) {
    if (n) hipLaunchKernelGGL(k_raw_record_hash, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, raw, stride, x_off, y_off, inf_off, is_inf, words, n, out);
}
void launch_psort(hipStream_t s, const PsParams &q, uint32_t NB, uint32_t *cnt1, uint32_t *off1, uint32_t *bsums, void *pairs, uint32_t *off, uint32_t *entries,
                  uint32_t heavy_thr, uint32_t *heavy, uint32_t heavy_cap, const uint32_t *dyn_args, uint32_t *dyn) {
    const size_t lds1 = (size_t)q.P * 4;
    const size_t lds3 = ((size_t)4 * q.P + 2) * 4 + (size_t)PS_TILE * (size_t)q.W * 8;      // (W = 13: 69 KB, two blocks per CU; sized for PS_MAX_W it was 81 KB and one)
    // (the attribute belongs to the function ON THE CURRENT DEVICE: a process that drives several GPUs sets it once per device)
    { static std::atomic<uint32_t> done{0}; int dev = 0; (void)hipGetDevice(&dev); const uint32_t bit = 1u << (dev & 31);
      if (!(done.load() & bit)) {
          (void)hipFuncSetAttribute((const void *)k_ps_scatter1<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
          (void)hipFuncSetAttribute((const void *)k_ps_scatter1<20, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
          (void)hipFuncSetAttribute((const void *)k_ps_scatter1<17, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
          (void)hipFuncSetAttribute((const void *)k_ps_scatter1<16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
          done.fetch_or(bit); } }
    // the shapes of the per-key tables (c = 20 at n >= 2^19, c = 17 for the prover's sparse queries) and of the plain pipeline at 2^17 .. 2^22 terms (c = 16) have
    // their digits extracted with constant shifts
    const int shape = (q.c == 20 && q.W == 13) ? 1 : ((q.c == 17 && q.W == 16) ? 2 : ((q.c == 16 && q.W == 16) ? 3 : 0));
    if (shape == 1) hipLaunchKernelGGL((k_ps_count1<20, 13>), dim3(q.ntiles), dim3(PS_TILE), lds1, s, q, cnt1);
    else if (shape == 2) hipLaunchKernelGGL((k_ps_count1<17, 16>), dim3(q.ntiles), dim3(PS_TILE), lds1, s, q, cnt1);
    else if (shape == 3) hipLaunchKernelGGL((k_ps_count1<16, 16>), dim3(q.ntiles), dim3(PS_TILE), lds1, s, q, cnt1);
    else hipLaunchKernelGGL((k_ps_count1<0, 0>), dim3(q.ntiles), dim3(PS_TILE), lds1, s, q, cnt1);
    launch_scan(s, cnt1, off1, nullptr, bsums, (size_t)q.P * q.ntiles);
    // the pair total is known: chunking and heavy-bucket threshold of the accumulation follow from it (dyn_args = {fixed_ch, min_chunk, max_chunks, lanes_per_chunk, T_max, nb_shared})
    if (dyn) launch_dyn_chunk(s, off1 + (size_t)q.P * q.ntiles, dyn_args[0], dyn_args[1], dyn_args[2], dyn_args[3], dyn_args[4], dyn, dyn_args[5]);
    if (shape == 1) hipLaunchKernelGGL((k_ps_scatter1<20, 13>), dim3(q.ntiles), dim3(PS_TILE), lds3, s, q, (const uint32_t *)cnt1, (const uint32_t *)off1, (uint2 *)pairs);
    else if (shape == 2) hipLaunchKernelGGL((k_ps_scatter1<17, 16>), dim3(q.ntiles), dim3(PS_TILE), lds3, s, q, (const uint32_t *)cnt1, (const uint32_t *)off1, (uint2 *)pairs);
    else if (shape == 3) hipLaunchKernelGGL((k_ps_scatter1<16, 16>), dim3(q.ntiles), dim3(PS_TILE), lds3, s, q, (const uint32_t *)cnt1, (const uint32_t *)off1, (uint2 *)pairs);
    else hipLaunchKernelGGL((k_ps_scatter1<0, 0>), dim3(q.ntiles), dim3(PS_TILE), lds3, s, q, (const uint32_t *)cnt1, (const uint32_t *)off1, (uint2 *)pairs);
    // P4 with write combining when the average partition is long (n >= 2^23 at c = 20): decided from the worst-case pair count, the direct path is right for sparse vectors
    const int wc = ((uint64_t)q.n * (uint64_t)q.W) / q.P >= PS_WC_MIN_PAIRS ? 1 : 0;
    const size_t lds4 = wc ? (size_t)2 * PS_PART * 4 + (size_t)PS_WC_TILE * 8 : (size_t)PS_DIRECT_CAP * 4;
    if (!wc) { static std::atomic<uint32_t> done5{0}; int dev = 0; (void)hipGetDevice(&dev); const uint32_t bit = 1u << (dev & 31);
      if (!(done5.load() & bit)) { (void)hipFuncSetAttribute((const void *)k_ps_bucket<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4); done5.fetch_or(bit); } }
    if (wc) { static std::atomic<uint32_t> done4{0}; int dev = 0; (void)hipGetDevice(&dev); const uint32_t bit = 1u << (dev & 31);
      if (!(done4.load() & bit)) { (void)hipFuncSetAttribute((const void *)k_ps_bucket<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4); done4.fetch_or(bit); } }
    if (wc) hipLaunchKernelGGL(k_ps_bucket<true>, dim3(q.P), dim3(1024), lds4, s, (const uint2 *)pairs, off1, q.ntiles, q.P, NB, q.part_log, off, entries, heavy_thr, heavy, heavy_cap, (const uint32_t *)dyn);
    else hipLaunchKernelGGL(k_ps_bucket<false>, dim3(q.P), dim3(1024), lds4, s, (const uint2 *)pairs, off1, q.ntiles, q.P, NB, q.part_log, off, entries, heavy_thr, heavy, heavy_cap, (const uint32_t *)dyn);
}
}  // namespace msm
