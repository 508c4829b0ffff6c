// crypto_amd/csrc/sort_kernels.hip.h — curve-independent kernels of the MSM pipeline: signed-digit recoding, the LDS
// counting sort (K2/K4), the histogram scan (K3) and the device self-tests.  Included by k_sort.hip only.
#pragma once
#include "dyn_chunk.hip.h"
#include <hip/hip_runtime.h>
#include "fp29.hip.h"
#include "ec29.hip.h"
#include "fp_inv.hip.h"
#include "ec29_two_lane.hip.h"
#include "digit_codes.hip.h"

namespace msm {
using namespace bls29;

// ---- digits ----------------------------------------------------------------------------------------
// Window w covers scalar bits [w c, w c + c); W = 255 / c + 1 windows, so the top window holds fewer than c
// bits and never carries out.  digit in [-(B-1), B], B = 2^(c-1): magnitude-1 is the bucket index.

// K2a: every scalar's W signed digits, stored window-major as codes: code = (|d| - 1) | sign << (CB-1), all-ones = zero digit
// (|d| - 1 <= 2^(c-1) - 1 needs c - 1 bits; a negative digit has |d| <= 2^(c-1) - 1, so the all-ones pattern is free).
template <class CODE>
__global__ void __launch_bounds__(256) k_digit_codes(const uint32_t *__restrict__ scalars, const uint32_t *__restrict__ bases, int aff_stride, int flag_word,
                                                     size_t n, size_t n_pad, int c, int W, CODE *__restrict__ dig, uint32_t *__restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    // padding / identity base contributes nothing.  bases == nullptr: the base records are not there yet (one-shot calls sort the scalars
    // while the bases are still crossing PCIe): identity bases are passed over by the SKIP_ID accumulation instead.
    const bool skip = (i >= n) || (bases && bases[i * (size_t)aff_stride + flag_word] != 0);
    digit_codes_one<CODE>(scalars, i, n, skip, n_pad, c, W, dig, bad);
}

// K2b / K4: counting sort without global atomics.  Block (w, r) owns the RB = 2^rb_log buckets [r RB, (r+1) RB) of window w,
// sweeps the whole digit column of that window (L2-resident: 2 B per term) and keeps its histogram / cursors in LDS.
// blockIdx -> (w, r) is XCD-aware: the blocks of one window are spaced 8 apart so they share one XCD's L2.
__device__ __forceinline__ void sort_block_coords(int W, int RANGES, int &w, int &r) {
    int b = blockIdx.x, x = b & 7, q = b >> 3;          // x = XCD (observed round-robin placement; speed only)
    int wpx = (W + 7) >> 3;                             // windows per XCD
    r = q % RANGES;
    w = x + 8 * (q / RANGES);
    if (q / RANGES >= wpx) w = W;                       // padding block
}
template <class CODE, bool SCATTER>
__global__ void __launch_bounds__(1024) k_sort_sweep(const CODE *__restrict__ dig, size_t n, size_t n_pad, int W, int RANGES, int rb_log, uint32_t B,
                                                     uint32_t *__restrict__ cnt, const uint32_t *__restrict__ off, uint32_t *__restrict__ entries,
                                                     uint32_t heavy_thr, uint32_t *__restrict__ heavy /* [0] = count, [1..cap] = keys */, uint32_t heavy_cap) {
    extern __shared__ uint32_t lds[];
    int w, r; sort_block_coords(W, RANGES, w, r);
    if (w >= W) return;
    const uint32_t RB = 1u << rb_log;
    const size_t kbase = (size_t)w * B + (size_t)r * RB;
    for (uint32_t j = threadIdx.x; j < RB; j += blockDim.x) lds[j] = SCATTER ? off[kbase + j] : 0u;
    __syncthreads();
    constexpr CODE ZERO = (CODE)~(CODE)0;
    constexpr int SIGN = sizeof(CODE) * 8 - 1;
    constexpr int PER = 16 / sizeof(CODE);              // codes per 16-byte load
    const CODE *col = dig + (size_t)w * n_pad;
    for (size_t base = (size_t)threadIdx.x * PER; base < n_pad; base += (size_t)blockDim.x * PER) {
        uint4 v = *reinterpret_cast<const uint4 *>(col + base);
        CODE cs[PER];
        memcpy(cs, &v, 16);
#pragma unroll
        for (int k = 0; k < PER; k++) {
            CODE cd = cs[k];
            if (cd == ZERO) continue;
            uint32_t idx = (uint32_t)cd & ((1u << SIGN) - 1u);
            if ((idx >> rb_log) != (uint32_t)r) continue;
            uint32_t j = idx & (RB - 1);
            if (!SCATTER) atomicAdd(&lds[j], 1u);
            else { uint32_t pos = atomicAdd(&lds[j], 1u); entries[pos] = (uint32_t)(base + k) | ((uint32_t)(cd >> SIGN) << 31); }
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < RB; j += blockDim.x) {
            uint32_t v = lds[j];
            cnt[kbase + j] = v;
            if (v >= heavy_thr) { uint32_t k = atomicAdd(&heavy[0], 1u); if (k < heavy_cap) heavy[1 + k] = (uint32_t)(kbase + j); }   // rare
        }
    }
}

// ---- K3: exclusive scan (u32), 4096 elements per block ----------------------------------------------
constexpr int SCAN_T = 256, SCAN_E = 16, SCAN_B = SCAN_T * SCAN_E;
__global__ void __launch_bounds__(SCAN_T) k_scan_block(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t *__restrict__ block_sums, size_t n) {
    __shared__ uint32_t sh[SCAN_T];
    size_t base = (size_t)blockIdx.x * SCAN_B + (size_t)threadIdx.x * SCAN_E;
    uint32_t v[SCAN_E], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_E; k++) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < SCAN_T; d <<= 1) {
        uint32_t t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t excl = sh[threadIdx.x] - s;
    if (threadIdx.x == SCAN_T - 1) block_sums[blockIdx.x] = sh[threadIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_E; k++) { if (base + k < n) out[base + k] = excl; excl += v[k]; }
}
// single block: in-place exclusive scan of the block sums, total written to block_sums[nb]
__global__ void __launch_bounds__(1024) k_scan_sums(uint32_t *__restrict__ block_sums, size_t nb) {
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (size_t base = 0; base < nb; base += 1024) {
        size_t i = base + threadIdx.x;
        uint32_t v = i < nb ? block_sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            uint32_t t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        uint32_t c0 = carry_s;
        if (i < nb) block_sums[i] = c0 + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = c0 + sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[nb] = carry_s;
}
// out[i] += block_sums[i / SCAN_B]; also writes out[n] = total and copies to cursor
__global__ void __launch_bounds__(256) k_scan_add(uint32_t *__restrict__ out, uint32_t *__restrict__ cursor, const uint32_t *__restrict__ block_sums, size_t n, size_t nb) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint32_t v = out[i] + block_sums[i / SCAN_B]; out[i] = v; if (cursor) cursor[i] = v; }
    if (i == n) out[n] = block_sums[nb];
}

// ---- chunking from the actual pair count (dyn_chunk.hip.h) ------------------------------------------------------------
// one thread: choose_chunk's rule (dock_core.hip) on E = *total.  fixed_ch != 0: a chunk length forced by the host (tuning knobs) is kept.
__global__ void k_dyn_chunk(const uint32_t *__restrict__ total, uint32_t fixed_ch, uint32_t min_chunk, uint32_t max_chunks, uint32_t lanes_per_chunk, uint32_t T_max, uint32_t *__restrict__ dyn, uint32_t nb_shared) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t E = *total;
    uint32_t ch = fixed_ch;
    if (!ch) {
        ch = min_chunk;
        while (E / ch > max_chunks && ch < 4096) ch *= 2;
        const uint32_t resident = RESIDENT_ACC_LANES / lanes_per_chunk;
        const uint32_t chunks = (E + ch - 1) / ch;
        if (chunks > resident) {            // whole rounds of the chip
            uint32_t rounds = (chunks + resident / 2) / resident;
            if (rounds < 1) rounds = 1;
            const uint32_t len = (uint32_t)(((uint64_t)E + (uint64_t)rounds * resident - 1) / ((uint64_t)rounds * resident));
            if (len >= 16 && len <= 4096) ch = len;
        }
        // one bucket set shared by all windows (nb_shared buckets): the runs are long, and a chunk that ends inside a run leaves a partial for the
        // fix-up to fold.  When the average run is at least half the chunk of ONE whole round, the launch is one round instead of two
        // (choose_chunk, dock_core.hip, has the measurements)
        if (nb_shared) {
            const uint32_t run = E / nb_shared, one = (uint32_t)(((uint64_t)E + resident - 1) / resident);
            if (one > ch && 2 * run >= one && one <= 4096) ch = one;
        }
    }
    if (T_max && (E + ch - 1) / ch > T_max) ch = (E + T_max - 1) / T_max;      // never more chunks than partial slots
    dyn[DYN_CH] = ch; dyn[DYN_T] = (E + ch - 1) / ch; dyn[DYN_HEAVY] = 16u * ch; dyn[DYN_E] = E; dyn[DYN_NMULTI] = 0;
}
// buckets with at least dyn[DYN_HEAVY] terms -> heavy[] (the sweep sort counts before the pair total is known, so it cannot flag them itself)
__global__ void __launch_bounds__(256) k_flag_heavy(const uint32_t *__restrict__ off, uint32_t NB, const uint32_t *__restrict__ dyn, uint32_t *__restrict__ heavy, uint32_t heavy_cap) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= NB) return;
    if (off[b + 1] - off[b] >= dyn[DYN_HEAVY]) { uint32_t h = atomicAdd(&heavy[0], 1u); if (h < heavy_cap) heavy[1 + h] = b; }
}

// ---- self-test kernels (tests/: device arithmetic vs oracle without the MSM plumbing) ------------------------
__global__ void k_selftest_fp_mul(const uint32_t *a, const uint32_t *b, size_t n, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp x, y, r; fp_from_abi(x, a + 12 * i); fp_from_abi(y, b + 12 * i);
    fp_mul(r, x, y);
    fp_to_abi(out + 12 * i, r);
}
// one thread: sum of (+/-) points by mixed additions, result in ABI XYZZ form
__global__ void k_selftest_g1_sum(const uint32_t *pts_abi, const uint8_t *neg, size_t n, uint32_t *out, uint8_t *out_inf) {
    if (blockIdx.x || threadIdx.x) return;
    Xyzz<Fp> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (size_t i = 0; i < n; i++) {
        Aff<Fp> p; fp_from_abi(p.x, pts_abi + 24 * i); fp_from_abi(p.y, pts_abi + 24 * i + 12);
        xyzz_madd(acc, inf, p, neg && neg[i]);
    }
    *out_inf = inf;
    if (!inf) { fp_to_abi(out, acc.x); fp_to_abi(out + 12, acc.y); fp_to_abi(out + 24, acc.zz); fp_to_abi(out + 36, acc.zzz); }
}

// ---- batched G1 scalar multiplication: out_i = s_i * P_i (affine, ABI form) --------------------------------------
// RandomizedPairingChecker scales every G1 source by a power of the batching randomness before the Miller loop
// (utils/src/randomized_pairing_check.rs:125-127,152-158: `a.mul_bigint(m)` in a cfg_iter!); one lane per point,
// double-and-add over the 255 scalar bits, then one Fermat inversion per lane for the affine form the line
// evaluation needs.
// Four lanes per point.  The scalar arrives split by the GLV decomposition k = k1 + k2 lambda (k1, k2 < 2^128, done on the host by
// hostf::glv_decompose): lanes 0,1 of a quad run the 128-step double-and-add of k1 P, lanes 2,3 that of k2 P — each chain on two lanes
// (ec29_two_lane.hip.h) — and the result is k1 P + phi(k2 P).  Half the dependent steps of the 255-bit chain: the kernel is latency-bound
// (RandomizedPairingChecker scales a handful of points), 3.5 -> ~2.3 ms.
// add_abi != nullptr: out_i = A_i + s_i * P_i (the aggregation's folding step, dgpu_g1_mul_add_batch).  Points are elements of the
// prime-order subgroup (the invariant of arkworks' G1Affine): phi(P) = lambda P only holds there.
__global__ void __launch_bounds__(64) k_g1_scale(const uint32_t *__restrict__ p_abi, const uint8_t *__restrict__ is_inf, const uint32_t *__restrict__ scalars, int scalar_stride,
                                                 const uint8_t *__restrict__ negate, size_t n, uint32_t *__restrict__ out_abi, uint8_t *__restrict__ out_inf,
                                                 const uint32_t *__restrict__ add_abi, const uint8_t *__restrict__ add_inf) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const bool B = (threadIdx.x & 1u) != 0;
    const uint32_t chain = (threadIdx.x >> 1) & 1u;
    if (i >= n) return;
    uint32_t any = 0;
    for (int k = 0; k < 24; k++) any |= p_abi[i * 24 + k];
    bool pinf = (any == 0) || (is_inf && is_inf[i]);
    Aff<Fp> P; fp_from_abi(P.x, p_abi + i * 24); fp_from_abi(P.y, p_abi + i * 24 + 12);
    const uint32_t *s = scalars + i * (size_t)scalar_stride + 4 * chain;          // (k1 | k2), four words each
    Xyzz<Fp> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    if (!pinf)
        for (int b = 127; b >= 0; b--) {
            if (!inf) { Xyzz<Fp> d; xyzz_dbl_2l(d, acc); acc = d; }
            if ((s[b >> 5] >> (b & 31)) & 1u) xyzz_madd_2l(acc, inf, P, false);
        }
    // the k2 chain hands its result to the k1 lanes (quad_perm [2,3,0,1]; every lane of the quad takes part in the exchange)
    Xyzz<Fp> oth; bool oinf;
    {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&acc);
        uint32_t *q = reinterpret_cast<uint32_t *>(&oth);
#pragma unroll
        for (int k = 0; k < 4 * NL; k++) q[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[k], 0x4E, 0xF, 0xF, true);
        oinf = __builtin_amdgcn_update_dpp(0, (int)inf, 0x4E, 0xF, 0xF, true) != 0;
    }
    if (chain) return;
    if (!oinf) xyzz_phi(oth);
    xyzz_add(acc, inf, oth, oinf);
    if (add_abi) {
        const uint32_t *src = add_abi + i * 24;
        uint32_t nz = 0;
        for (int k = 0; k < 24; k++) nz |= src[k];
        if (nz != 0 && !(add_inf && add_inf[i])) { Aff<Fp> A; fp_from_abi(A.x, src); fp_from_abi(A.y, src + 12); xyzz_madd_2l(acc, inf, A, false); }
    }
    if (!B) out_inf[i] = inf;
    uint32_t *o = out_abi + i * 24;
    if (inf) { if (!B) for (int k = 0; k < 24; k++) o[k] = 0; return; }
    Fp i3, t, i2, x, y;
    fp_inv_device(i3, acc.zzz);                 // 1 / ZZZ   (both lanes: the inversion is one serial chain either way)
    fp_mul(t, acc.zz, i3); fp_sqr(i2, t);       // (ZZ / ZZZ)^2 = 1 / ZZ      (ZZ^3 == ZZZ^2)
    Fp xn, yn; fp_norm(xn, acc.x); fp_norm(yn, acc.y);
    fp_mul(x, xn, i2); fp_mul(y, yn, i3);
    if (negate && negate[i]) { Fp z; fp_zero(z); fp_sub<4>(y, z, y); fp_norm(y, y); }
    if (!B) fp_to_abi(o, x); else fp_to_abi(o + 12, y);
}

}  // namespace msm
