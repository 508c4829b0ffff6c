// crypto_amd/csrc/k_ntt.hip — translation unit of the Fr NTT / witness-map kernels.
#include <atomic>
#include "ntt_kernels.hip.h"
#include "qap_launch.hip.h"
#include <cstdlib>
#include <cstdio>
namespace ntt {
static inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + 255) / 256)); }
void launch_fr_mont_to_canonical(hipStream_t s, uint32_t *words, size_t n) { hipLaunchKernelGGL(k_fr_mont_to_canonical, grid_for(n), dim3(256), 0, s, words, n); }
void launch_fr_canonical_to_mont(hipStream_t s, uint32_t *words, size_t n) { hipLaunchKernelGGL(k_fr_canonical_to_mont, grid_for(n), dim3(256), 0, s, words, n); }
void launch_fr_load(hipStream_t s, const uint32_t *words, size_t n, int mont, uint32_t *out, size_t D) { hipLaunchKernelGGL(k_fr_load, grid_for(D), dim3(256), 0, s, words, n, mont, out, D); }
void launch_fr_powers(hipStream_t s, const uint32_t *bw, const uint32_t *sw, size_t count, uint32_t *out) { hipLaunchKernelGGL(k_fr_powers, grid_for(count), dim3(256), 0, s, bw, sw, count, out); }
void launch_tw_compact(hipStream_t s, uint32_t *tw, size_t H) {
    if (H > 1) hipLaunchKernelGGL(k_tw_compact, grid_for(H), dim3(256), 0, s, tw, H, 0);
}
void launch_csr_eval(hipStream_t s, const uint64_t *rowptr, const uint32_t *cols, const uint32_t *vals_soa, size_t nnz, const uint32_t *z_words, int z_mont, size_t nvars, size_t rows, size_t extra, uint32_t *out, size_t D) {
    hipLaunchKernelGGL(k_csr_eval, grid_for(D), dim3(256), 0, s, rowptr, cols, vals_soa, nnz, z_words, z_mont, nvars, rows, extra, out, D);
}
// in_pointwise: bufs = {a, b, c}, pre = zinv words, the transform runs over a only (see k_ntt_r4); post / out_words: the last pass writes scalars
static bool run_passes(hipStream_t s, uint32_t *const *bufs, int nbuf, int logn, const uint32_t *tw, int dif, const uint32_t *pre, bool in_pointwise, const uint32_t *post, uint32_t *out_words) {
    const size_t D = (size_t)1 << logn, H = D >> 1;
#ifdef DGPU_DEV
    static const bool unfused = getenv("DGPU_NTT_UNFUSED") != nullptr;     // development switches (compile with -DDGPU_DEV)
    static const bool unpiped = getenv("DGPU_NTT_STAGED") != nullptr;       // the load / stages / store kernel (k_ntt_fused)
#else
    constexpr bool unfused = false, unpiped = false;
#endif
    if ((in_pointwise || post) && (logn < PIPE_TILE_LOG || logn > PIPE_MAX_LOGN || unfused || unpiped)) return false;   // the caller runs the separate kernels
    if (logn < PIPE_TILE_LOG || unfused) {               // tiny domains: one pass per stage
        for (int b = 0; b < nbuf; b++) {
            if (pre) hipLaunchKernelGGL(k_coset_scale, grid_for(D), dim3(256), 0, s, bufs[b], logn, pre, (uint32_t *)nullptr, 1);
            for (int st = 0; st < logn; st++) hipLaunchKernelGGL(k_ntt_stage, grid_for(H), dim3(256), 0, s, bufs[b], logn, st, tw, dif);
        }
        return true;
    }
    if (unpiped || logn > PIPE_MAX_LOGN) {                // (32-bit buffer offsets: arrays beyond 4 GB take the staged kernel)
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
        const unsigned tiles = (unsigned)(D >> FUSE_TILE_LOG);
        for (int b = 0; b < nbuf; b++) { int s0 = 0; for (int gidx = 0; gidx < ng; gidx++) { hipLaunchKernelGGL(k_ntt_fused, dim3(tiles), dim3(FUSE_THREADS), lds_bytes, s, bufs[b], logn, s0, groups[gidx], tw, dif, gidx == 0 ? pre : (const uint32_t *)nullptr); s0 += groups[gidx]; } }
        return true;
    }
    // Pipelined passes.  One pass is "flat" (L = 0: a tile is one contiguous run, up to TILE_LOG stages), the others are strided and a tile
    // holds 2^(TILE_LOG - S) consecutive columns: S = TILE_LOG - 5 keeps every access a full 128-byte line (32 columns), one more stage halves it.
    constexpr int tile_log = PIPE_TILE_LOG;
    const int pref = tile_log - 5, maxs = tile_log - 4;
    int groups[12], ng = 0;
    {
        const int over = logn > tile_log ? logn - tile_log : 0;
        const int n_str = (over + maxs - 1) / maxs;
        int flat = logn - pref * n_str;
        if (flat > tile_log) flat = tile_log;
        if (flat < 1) flat = 1;
        int strided[12], rest = logn - flat;
        for (int k = 0; k < n_str; k++) { strided[k] = rest / (n_str - k); rest -= strided[k]; }
        if (dif) { for (int k = n_str - 1; k >= 0; k--) groups[ng++] = strided[k]; groups[ng++] = flat; }
        else { groups[ng++] = flat; for (int k = 0; k < n_str; k++) groups[ng++] = strided[k]; }
    }
#ifdef DGPU_DEV
    if (const char *e = getenv("DGPU_NTT_SPLIT")) {      // e.g. "6,6,8": strided passes then the flat one (the order is reversed for DIT)
        int v[12], n = 0, sum = 0; const char *q = e;
        while (*q && n < 12) { v[n] = atoi(q); sum += v[n++]; while (*q && *q != ',') q++; if (*q == ',') q++; }
        if (sum == logn) { ng = 0; if (dif) for (int k = 0; k < n; k++) groups[ng++] = v[k]; else for (int k = n - 1; k >= 0; k--) groups[ng++] = v[k]; }
    }
#endif
    const size_t lds_bytes = (size_t)NL * ((size_t)1 << tile_log) * 4;
    { static std::atomic<uint32_t> done{0}; int dev = 0; (void)hipGetDevice(&dev); const uint32_t bit = 1u << (dev & 31);
      if (!(done.load() & bit)) {
          (void)hipFuncSetAttribute((const void *)k_ntt_r4<true, PIPE_TILE_LOG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
          (void)hipFuncSetAttribute((const void *)k_ntt_r4<false, PIPE_TILE_LOG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
          done.fetch_or(bit); } }
    NttBatch B; for (int b = 0; b < 3; b++) B.buf[b] = bufs[b < nbuf ? b : 0];
    const unsigned tiles = (unsigned)(D >> tile_log), total = tiles * (unsigned)(in_pointwise ? 1 : nbuf);
    const dim3 blk(1u << (tile_log - 2));
    int s0 = 0;
    for (int gidx = 0; gidx < ng; gidx++) {
        const uint32_t *pr = gidx == 0 ? pre : (const uint32_t *)nullptr;
        const int im = (gidx == 0 && in_pointwise) ? NTT_IN_POINTWISE : 0, om = (gidx == ng - 1 && post) ? NTT_OUT_WORDS : 0;
        if (dif) hipLaunchKernelGGL((k_ntt_r4<true, PIPE_TILE_LOG>), dim3(total), blk, lds_bytes, s, B, logn, s0, groups[gidx], tw, pr, im, om, post, out_words);
        else hipLaunchKernelGGL((k_ntt_r4<false, PIPE_TILE_LOG>), dim3(total), blk, lds_bytes, s, B, logn, s0, groups[gidx], tw, pr, im, om, post, out_words);
        s0 += groups[gidx];
    }
    return true;
}
void launch_ntt_batch(hipStream_t s, uint32_t *const *bufs, int nbuf, int logn, const uint32_t *tw, int dif, const uint32_t *pre) {
    (void)run_passes(s, bufs, nbuf, logn, tw, dif, pre, false, nullptr, nullptr);
}
// h = coset iFFT of (a b - c) / Z(g), written as canonical scalars in natural order: a <- iDFT((a b - c) zinv) (bit-reversed), out[k] = a[rev k] pw[rev k]
void launch_ntt_final(hipStream_t s, uint32_t *a, uint32_t *b, uint32_t *c, int logn, const uint32_t *tw_i, const uint32_t *zinv_words, const uint32_t *pw_data_order, uint32_t *out_words) {
    uint32_t *bufs[3] = {a, b, c};
    if (run_passes(s, bufs, 3, logn, tw_i, 1, zinv_words, true, pw_data_order, out_words)) return;
    launch_pointwise(s, a, b, c, (size_t)1 << logn, zinv_words);
    launch_ntt(s, a, logn, tw_i, 1);
    launch_coset_scale(s, a, logn, pw_data_order, out_words, 1);
}
void launch_ntt(hipStream_t s, uint32_t *buf, int logn, const uint32_t *tw, int dif, const uint32_t *pre) { uint32_t *b[1] = {buf}; launch_ntt_batch(s, b, 1, logn, tw, dif, pre); }
void launch_coset_scale(hipStream_t s, uint32_t *buf, int logn, const uint32_t *pw, uint32_t *out_words, int pw_in_data_order) { hipLaunchKernelGGL(k_coset_scale, grid_for((size_t)1 << logn), dim3(256), 0, s, buf, logn, pw, out_words, pw_in_data_order); }
void launch_bitrev_table(hipStream_t s, const uint32_t *src, uint32_t *dst, int logn) { hipLaunchKernelGGL(k_bitrev_table, grid_for((size_t)1 << logn), dim3(256), 0, s, src, dst, logn); }
void launch_pointwise(hipStream_t s, uint32_t *a, const uint32_t *b, const uint32_t *c, size_t D, const uint32_t *zinv_words) { hipLaunchKernelGGL(k_pointwise, grid_for(D), dim3(256), 0, s, a, b, c, D, zinv_words); }
}  // namespace ntt
