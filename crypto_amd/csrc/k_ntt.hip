// crypto_amd/csrc/k_ntt.hip — translation unit of the Fr NTT / witness-map kernels.
#include <atomic>
#include "ntt_kernels.cuh"
#include "qap_launch.cuh"
#include <cstdlib>
namespace ntt {
static inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + 255) / 256)); }
void launch_fr_mont_to_canonical(hipStream_t s, uint32_t *words, size_t n) { hipLaunchKernelGGL(k_fr_mont_to_canonical, grid_for(n), dim3(256), 0, s, words, n); }
void launch_fr_load(hipStream_t s, const uint32_t *words, size_t n, int mont, uint32_t *out, size_t D) { hipLaunchKernelGGL(k_fr_load, grid_for(D), dim3(256), 0, s, words, n, mont, out, D); }
void launch_fr_powers(hipStream_t s, const uint32_t *bw, const uint32_t *sw, size_t count, uint32_t *out) { hipLaunchKernelGGL(k_fr_powers, grid_for(count), dim3(256), 0, s, bw, sw, count, out); }
void launch_tw_compact(hipStream_t s, uint32_t *tw, size_t H) {
    if (H > 1) hipLaunchKernelGGL(k_tw_compact, grid_for(H), dim3(256), 0, s, tw, H, 0);
}
void launch_csr_eval(hipStream_t s, const uint64_t *rowptr, const uint32_t *cols, const uint32_t *vals_soa, size_t nnz, const uint32_t *z_words, int z_mont, size_t nvars, size_t rows, size_t extra, uint32_t *out, size_t D) {
    hipLaunchKernelGGL(k_csr_eval, grid_for(D), dim3(256), 0, s, rowptr, cols, vals_soa, nnz, z_words, z_mont, nvars, rows, extra, out, D);
}
void launch_ntt(hipStream_t s, uint32_t *buf, int logn, const uint32_t *tw, int dif, const uint32_t *pre) {
    const size_t D = (size_t)1 << logn, H = D >> 1;
#ifdef DGPU_DEV
    static const bool unfused = getenv("DGPU_NTT_UNFUSED") != nullptr;     // development switch (compile with -DDGPU_DEV)
#else
    constexpr bool unfused = false;
#endif
    if (logn < FUSE_TILE_LOG || unfused) {               // tiny domains: one pass per stage
        if (pre) hipLaunchKernelGGL(k_coset_scale, grid_for(D), dim3(256), 0, s, buf, logn, pre, (uint32_t *)nullptr, 1);
        for (int st = 0; st < logn; st++) hipLaunchKernelGGL(k_ntt_stage, grid_for(H), dim3(256), 0, s, buf, logn, st, tw, dif);
        return;
    }
    const size_t lds_bytes = (size_t)NL * (1u << FUSE_TILE_LOG) * 4;      // 80 KB
    // (the attribute belongs to the function ON THE CURRENT DEVICE: a process that drives several GPUs sets it once per device)
    { static std::atomic<uint32_t> done{0}; int dev = 0; (void)hipGetDevice(&dev); const uint32_t bit = 1u << (dev & 31);
      if (!(done.load() & bit)) { (void)hipFuncSetAttribute((const void *)k_ntt_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); done.fetch_or(bit); } }
    // groups of up to 7 stages; the short group goes where its L is harmless (first for DIF, last for DIT), see k_ntt_fused
    const int SMAX = 7;
    int groups[8], ng = 0, rest = logn % SMAX;
    if (dif) { if (rest) groups[ng++] = rest; for (int k = 0; k < logn / SMAX; k++) groups[ng++] = SMAX; }
    else { for (int k = 0; k < logn / SMAX; k++) groups[ng++] = SMAX; if (rest) groups[ng++] = rest; }
    // a short group of S < 4 stages next to L = 0 would give 0 < L < log2(columns): merge it with its neighbour by splitting 7 + S evenly
    if (ng >= 2) {
        int &shortg = dif ? groups[0] : groups[ng - 1]; int &nb = dif ? groups[1] : groups[ng - 2];
        if (shortg < 4) { int tot = shortg + nb; shortg = tot / 2; nb = tot - shortg; }
    }
    int s0 = 0;
    const unsigned tiles = (unsigned)(D >> FUSE_TILE_LOG);
    for (int gidx = 0; gidx < ng; gidx++) { hipLaunchKernelGGL(k_ntt_fused, dim3(tiles), dim3(FUSE_THREADS), lds_bytes, s, buf, logn, s0, groups[gidx], tw, dif, gidx == 0 ? pre : (const uint32_t *)nullptr); s0 += groups[gidx]; }
}
void launch_coset_scale(hipStream_t s, uint32_t *buf, int logn, const uint32_t *pw, uint32_t *out_words, int pw_in_data_order) { hipLaunchKernelGGL(k_coset_scale, grid_for((size_t)1 << logn), dim3(256), 0, s, buf, logn, pw, out_words, pw_in_data_order); }
void launch_bitrev_table(hipStream_t s, const uint32_t *src, uint32_t *dst, int logn) { hipLaunchKernelGGL(k_bitrev_table, grid_for((size_t)1 << logn), dim3(256), 0, s, src, dst, logn); }
void launch_pointwise(hipStream_t s, uint32_t *a, const uint32_t *b, const uint32_t *c, size_t D, const uint32_t *zinv_words) { hipLaunchKernelGGL(k_pointwise, grid_for(D), dim3(256), 0, s, a, b, c, D, zinv_words); }
}  // namespace ntt
