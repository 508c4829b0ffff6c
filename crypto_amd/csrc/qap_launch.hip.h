// crypto_amd/csrc/qap_launch.hip.h — host-callable launchers of the NTT / witness-map kernels (k_ntt.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
namespace ntt {
constexpr int FR_WORDS = 10;   // u32 per internal Fr element
void launch_fr_mont_to_canonical(hipStream_t s, uint32_t *words, size_t n);
void launch_fr_canonical_to_mont(hipStream_t s, uint32_t *words, size_t n);
void launch_fr_load(hipStream_t s, const uint32_t *words, size_t n, int mont, uint32_t *out, size_t D);
void launch_fr_powers(hipStream_t s, const uint32_t *base_words, const uint32_t *scale_words, size_t count, uint32_t *out);
// twiddle buffers hold 2H elements: T_0 (H powers) followed by the per-stage compacted tables built by launch_tw_compact
void launch_tw_compact(hipStream_t s, uint32_t *tw, size_t H);
void launch_csr_eval(hipStream_t s, const uint64_t *rowptr, const uint32_t *cols, const uint32_t *vals_soa, size_t nnz, const uint32_t *z_words /* 8 u32 per variable */, int z_mont, size_t nvars, size_t rows, size_t extra, uint32_t *out, size_t D);
void launch_ntt(hipStream_t s, uint32_t *buf, int logn, const uint32_t *tw, int dif, const uint32_t *pre = nullptr);   // all logn stages; pre: per-position factors applied on the way in
void launch_ntt_batch(hipStream_t s, uint32_t *const *bufs, int nbuf, int logn, const uint32_t *tw, int dif, const uint32_t *pre = nullptr);   // the same transform over up to 3 arrays, one launch per pass
// the last transform of the witness map with the (ab - c) / Z(g) step on the way in and scaling + un-reversal + canonical scalars on the way out
void launch_ntt_final(hipStream_t s, uint32_t *a, uint32_t *b, uint32_t *c, int logn, const uint32_t *tw_i, const uint32_t *zinv_words, const uint32_t *pw_data_order, uint32_t *out_words);
void launch_coset_scale(hipStream_t s, uint32_t *buf, int logn, const uint32_t *pw, uint32_t *out_words, int pw_in_data_order = 0);
void launch_bitrev_table(hipStream_t s, const uint32_t *src, uint32_t *dst, int logn);
void launch_pointwise(hipStream_t s, uint32_t *a, const uint32_t *b, const uint32_t *c, size_t D, const uint32_t *zinv_words);
}  // namespace ntt
