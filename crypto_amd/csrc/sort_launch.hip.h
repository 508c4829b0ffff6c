// crypto_amd/csrc/sort_launch.hip.h — host-callable launchers of the curve-independent kernels (k_sort.hip): digit recoding,
// LDS counting sort, histogram scan, batched G1 scaling, device self-tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace msm {
// curve independent (k_sort.hip)
void launch_digit_codes(hipStream_t s, bool wide, const uint32_t *scalars, const uint32_t *bases, int aff_stride, int flag_word, size_t n, size_t n_pad, int c, int W, void *dig, uint32_t *bad /* set to 1 if a scalar has bit 255 set */);
void launch_sort_sweep(hipStream_t s, bool wide, bool scatter, unsigned grid, size_t lds_bytes, const void *dig, size_t n, size_t n_pad, int W, int RANGES, int rb_log, uint32_t B,
                       uint32_t *cnt, const uint32_t *off, uint32_t *entries, uint32_t heavy_thr, uint32_t *heavy, uint32_t heavy_cap);
void launch_scan(hipStream_t s, const uint32_t *cnt, uint32_t *off, uint32_t *cursor, uint32_t *bsums, size_t NB);
// chunking of the accumulation from the pair count the sort produced (dyn_chunk.hip.h): dyn[0..3] = chunk length, chunks, heavy threshold, pairs
void launch_dyn_chunk(hipStream_t s, const uint32_t *total, uint32_t fixed_ch, uint32_t min_chunk, uint32_t max_chunks, uint32_t lanes_per_chunk, uint32_t T_max, uint32_t *dyn,
                      uint32_t nb_shared = 0 /* buckets of the ONE set all windows share (table pipeline), 0: a set per window */);
void launch_flag_heavy(hipStream_t s, const uint32_t *off, uint32_t NB, const uint32_t *dyn, uint32_t *heavy, uint32_t heavy_cap);
size_t scan_blocks(size_t NB);
// ---- two-level partition sort (k_psort.hip, psort_kernels.hip.h) ----
constexpr int PS_TILE = 512;            // scalars per tile = threads per block of P1 / P3
constexpr int PS_PART_LOG_MAX = 11;     // at most 2048 buckets per partition (P4's LDS histogram)
constexpr int PS_PART = 1 << PS_PART_LOG_MAX;
constexpr int PS_WC_TILE = 16384;       // pairs per write-combining tile of P4 (2 x PS_PART counters + 128 KB of staged pairs in LDS)
#ifdef PS_NO_WC
constexpr uint32_t PS_WC_MIN_PAIRS = 0xffffffffu;      // (development A/B build: the direct placement at every size)
#else
constexpr uint32_t PS_WC_MIN_PAIRS = 65536;
#endif   // partitions at least this long (on average) take the write-combining path
constexpr int PS_DIRECT_CAP = 16384;     // slots of a partition that P4's direct form stages in LDS (64 KB: two blocks per CU)
constexpr int PS_MAX_W = 16;            // pairs staged per scalar (LDS: PS_TILE * PS_MAX_W * 8 B)
struct PsParams {
    const uint32_t *scalars;            // n x 8 words, canonical
    const uint8_t *idflag;              // idflag[flag_base + i] != 0 -> the base of term i is the identity, the term is skipped (the table's own byte per base: reading
                                        // the flag word inside every 128 / 256-byte record pulled the whole first table row through HBM twice per sort); nullptr: keep all
    size_t n;
    uint32_t flag_base;
    int c, W;
    uint32_t key_wstride;
    uint32_t val_base, val_wstride;
    int part_log;                       // log2 buckets per partition: ps_part_log(NB)
    uint32_t P;                         // partitions = ceil(NB / 2^part_log)
    uint32_t ntiles;
    uint32_t *bad;                      // *bad |= 1 if a scalar has bit 255 set (the call is refused: sort_kernels.hip.h k_digit_codes)
};

// about a thousand partitions (P4 runs one block per partition) while a partition keeps at least 32 buckets
inline int ps_part_log(uint32_t NB) { int lg = 0; while ((1u << lg) < NB) lg++; int pl = lg - 10; return pl < 5 ? 5 : (pl > PS_PART_LOG_MAX ? PS_PART_LOG_MAX : pl); }
// cnt1 / off1: P * ntiles + 1 words; bsums: scan_blocks(P * ntiles) + 2 words; pairs: n * W x 8 B; off: NB + 1; entries: n * W
void launch_psort(hipStream_t s, const PsParams &q, uint32_t NB, uint32_t *cnt1, uint32_t *off1, uint32_t *bsums, void *pairs, uint32_t *off, uint32_t *entries,
                  uint32_t heavy_thr, uint32_t *heavy, uint32_t heavy_cap, const uint32_t *dyn_args = nullptr, uint32_t *dyn = nullptr);
// out[i] = (bases[i * aff_stride + flag_word] != 0): the identity flags of n prepared records as one byte each (built once per table)
void launch_id_flags(hipStream_t s, const uint32_t *bases, int aff_stride, int flag_word, size_t n, uint8_t *out);
// out[i] = fingerprint of raw point i (psort_kernels.hip.h k_raw_record_hash; the host-side twin is rec_fingerprint of bases_cache.hpp); words = 64-bit words per coordinate
void launch_raw_record_hash(hipStream_t s, const uint8_t *raw, size_t stride, size_t x_off, size_t y_off, size_t inf_off, const uint8_t *is_inf, int words, size_t n, uint64_t *out);
void launch_g1_scale(hipStream_t s, const uint32_t *p_abi, const uint8_t *is_inf, const uint32_t *scalars, int scalar_stride, const uint8_t *negate, size_t n, uint32_t *out_abi, uint8_t *out_inf,
                     const uint32_t *add_abi = nullptr, const uint8_t *add_inf = nullptr);
void launch_selftest_fp_mul(hipStream_t s, const uint32_t *a, const uint32_t *b, size_t n, uint32_t *out);
void launch_selftest_g1_sum(hipStream_t s, const uint32_t *pts, const uint8_t *neg, size_t n, uint32_t *out, uint8_t *out_inf);

}  // namespace msm
