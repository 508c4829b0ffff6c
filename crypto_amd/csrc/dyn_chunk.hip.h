// crypto_amd/csrc/dyn_chunk.hip.h — chunking of the bucket accumulation, decided on the device.
//
// The host cannot know how many (bucket, term) pairs an MSM has before the sort has run: zero digits are dropped, so a Groth16 witness (mostly
// 0 / 1 / small values) leaves a quarter of the n * W pairs of a dense scalar vector.  A chunk length derived from n * W then leaves most of
// the chip idle: the few lanes that have work each run a full-length chain of dependent additions.  k_dyn_chunk applies the rule of
// choose_chunk (dock_core.hip) to the ACTUAL pair count E right after the sort's scan, and the accumulation / fix-up kernels read the result:
//     dyn[DYN_CH] terms per lane, dyn[DYN_T] lanes with work, dyn[DYN_HEAVY] = 16 chunks: buckets at least that long are folded by k_fixup_heavy.
// The launch grids and the partial-slot buffers stay sized for the worst case (n * W pairs); lanes beyond dyn[DYN_T] leave at once.
#pragma once
#include <stdint.h>
namespace msm {
constexpr int DYN_CH = 0, DYN_T = 1, DYN_HEAVY = 2, DYN_E = 3, DYN_NMULTI = 4, DYN_MULTI = 5;   // dyn[DYN_MULTI + i]: buckets longer than one fold range (msm_kernels.hip.h)
constexpr uint32_t HEAVY_RANGE = 512;
inline size_t dyn_words(size_t T) { return DYN_MULTI + T / HEAVY_RANGE + 2; }
constexpr uint32_t RESIDENT_ACC_LANES = 131072;      // 256 CUs x 4 SIMDs x 2 waves x 64 lanes (k_accumulate: 2 waves per SIMD)
}  // namespace msm
