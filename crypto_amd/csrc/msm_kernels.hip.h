// crypto_amd/csrc/msm_kernels.hip.h — Pippenger bucket MSM for gfx950, written for the chip rather than
// translated from arkworks' rayon loop (ark-ec 0.4 VariableBaseMSM::msm_bigint_wnaf; the
// reference enters it at utils/src/pairs.rs:145-155 and legogroth16/src/prover.rs:286,299,592).
//
// arkworks parallelises over the ~17 windows only.  Here every (window, bucket) pair of all windows is one
// flat key space and the work is cut into equal chunks of the key-sorted term list, so all 256 CUs stay
// busy whatever the scalar distribution:
//
//   K1 prep_bases   ABI affine (2^384 Montgomery, 32-bit words) -> 128 B / 256 B records of 29-bit limbs
//   K2 digits+hist  signed radix-2^c digits of every scalar stored as 2-byte codes; histogram of (window, |digit|-1)
//                   keys by blocks that each own a bucket range and count in LDS (no global atomics)
//   K3 scan         exclusive prefix sum of the histogram (3 small kernels)
//   K4 scatter      same sweep with LDS cursors: term index (+ sign bit) written at its key's slot (counting sort)
//   K5 accumulate   thread t owns terms [t*CH, (t+1)*CH) of the sorted list: mixed XYZZ additions, runs that
//                   lie inside the chunk go straight to the bucket array, the (at most two) runs cut by a
//                   chunk border go to head/tail partial slots
//   K6 fixup        one thread per cut bucket folds its partials
//   K7 reduce       sum_k k*B_k per window: 2^s buckets serial per lane, then a wave64 suffix scan (shuffles)
//   K8 reduce_top   one wave per window folds the group results
//   host            Horner over the <= 64 window sums + normalisation (host_field.hpp)
//
// Any c, any chunking and any order of additions give the same group element; only the projective
// representative differs, and the ABI returns the normalised one.
#pragma once
#include "dyn_chunk.hip.h"
#include <hip/hip_runtime.h>
#include "fp29.hip.h"
#include "fp2_29.hip.h"
#include "ec29.hip.h"
#include "fp2_pair.hip.h"
#include "fp30s.hip.h"
#include "fs2_pair.hip.h"
#include <type_traits>

namespace msm {
using namespace bls29;


// ---- curve descriptions -------------------------------------------------------------------------
struct G1;
struct G1S;
struct G2;
struct G2P;
// G1 over the 13 x 30-bit signed field (fp30s.hip.h): the curve description the MSM pipeline instantiates for G1 (G1::MSM).  Records keep the
// strides of the 14 x 29-bit form (128-byte base records with the identity flag in word 28); an XYZZ record is 52 words.
struct G1S {
    typedef Fs F;
    static constexpr int FW = SN;            // 13 words per coordinate
    static constexpr int ABI_W = 12;
    static constexpr int NFP = 1;
    static constexpr int AFF_STRIDE = 32;    // x[13] y[13] pad[2] flag pad[3]
    static constexpr int FLAGW = 28;
    static constexpr int XW = 4 * FW;        // 52 words = 13 x 16 bytes
    static constexpr int HEAVY_T = 256;
    static constexpr int LPP = 1;
    typedef G1S ACC;
    typedef G1S MSM;
    static constexpr int ACC_WAVES = 2;
};
struct G1 {
    typedef Fp F;
    typedef G1S MSM;                         // what the MSM kernels (prep, accumulate, fix-up, reduce, table construction) run G1 as
    static constexpr int FLAGW = 28;         // word of a base record that holds the identity flag
    static constexpr int FW = NL;            // u32 words per coordinate (device form)
    static constexpr int ABI_W = 12;         // u32 words per Fp in the ABI
    static constexpr int NFP = 1;            // Fp components per coordinate
    static constexpr int AFF_STRIDE = 32;    // u32 per prepared base record (128 B): x[14] y[14] flag pad[3]
    static constexpr int XW = 4 * FW;        // u32 per XYZZ point
    static constexpr int HEAVY_T = 256;      // threads per block in k_fixup_heavy (XW * HEAVY_T * 4 B of LDS)
    static constexpr int LPP = 1;            // lanes per point in k_accumulate
    typedef G1S ACC;                         // traits used by the accumulate kernel
    static constexpr int ACC_WAVES = 2;      // waves/SIMD of k_accumulate: 256 VGPRs, no spills (tools/ubench/madd_rate: 6.35 vs 5.1 Gmadd/s at 3)
};
// G2 over the signed field, one lane per point (base preparation, table construction).  In memory every base-field component keeps a slot of
// PS = 14 words (13 used): the lane-pair kernels move a half with 8-byte accesses, which the 56-byte slots keep aligned.
constexpr int PS = NL;       // words per component slot in G2 records
struct G2S {
    typedef Fs2 F;
    typedef G2S MSM;
    static constexpr int FLAGW = 56;
    static constexpr int FW = 2 * SN;        // words per coordinate in registers (the XYZZ scratch of the table construction is the packed struct)
    static constexpr int ABI_W = 24;
    static constexpr int NFP = 2;
    static constexpr int AFF_STRIDE = 64;    // four component slots, flag, padding
    static constexpr int XW = 4 * FW;        // 104
    static constexpr int HEAVY_T = 128;
    static constexpr int LPP = 1;
};
struct G2 {
    typedef Fp2 F;
    typedef G2S MSM;
    static constexpr int FLAGW = 56;
    static constexpr int FW = 2 * NL;
    static constexpr int ABI_W = 24;
    static constexpr int NFP = 2;
    static constexpr int AFF_STRIDE = 64;    // 256 B: x[28] y[28] flag pad[7]
    static constexpr int XW = 4 * FW;
    static constexpr int HEAVY_T = 128;
    static constexpr int LPP = 1;
    typedef G2P ACC;                         // k_accumulate runs the lane-pair formulation (fp2_pair.hip.h)
    static constexpr int ACC_WAVES = 2;
};
// G2 with one point per lane pair: the even lane holds the c0 halves, the odd lane the c1 halves
struct G2P {
    typedef Fs2H F;
    static constexpr int FLAGW = 56;
    static constexpr int FW = SN;            // words per coordinate HALF held by one lane
    static constexpr int AFF_STRIDE = 64;
    static constexpr int XW = 8 * PS;        // words per full XYZZ point in the record arrays (eight component slots)
    static constexpr int LPP = 2;
    static constexpr int ACC_WAVES = 2;
    static constexpr int HEAVY_T = 256;      // 128 lane pairs per block in the heavy-bucket folds
};

template <class F> __device__ __forceinline__ uint32_t *limbs(F &f) { return reinterpret_cast<uint32_t *>(&f); }
template <class F> __device__ __forceinline__ const uint32_t *limbs(const F &f) { return reinterpret_cast<const uint32_t *>(&f); }

// the two coordinates of an affine point from their ABI words (2 * C::ABI_W words) into a base record (device form of C)
template <class C> __device__ __forceinline__ void store_coords_from_abi(uint32_t *__restrict__ dst, const uint32_t *w) {
    if constexpr (std::is_same<typename C::F, Fs>::value) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            Fs f; fs_from_abi(f, w + 12 * k);
#pragma unroll
            for (int j = 0; j < SN; j++) dst[k * SN + j] = (uint32_t)f.l[j];
        }
        dst[2 * SN] = 0; dst[2 * SN + 1] = 0;
    } else if constexpr (std::is_same<typename C::F, Fs2>::value) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            Fs f; fs_from_abi(f, w + 12 * k);
#pragma unroll
            for (int j = 0; j < SN; j++) dst[k * PS + j] = (uint32_t)f.l[j];
            dst[k * PS + SN] = 0;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 2 * C::NFP; k++) {
            Fp f; fp_from_abi(f, w + 12 * k);
#pragma unroll
            for (int j = 0; j < NL; j++) dst[k * NL + j] = f.l[j];
        }
    }
}
// a coordinate tuple of `count` base-field components (x, y[, zz, zzz]; Fp2: c0, c1 per coordinate) to ABI words, 12 per component
template <class F> __device__ __forceinline__ void coords_to_abi(uint32_t *__restrict__ dst, const void *pt, int count) {
    if constexpr (std::is_same<F, Fs>::value || std::is_same<F, Fs2>::value) { const Fs *f = reinterpret_cast<const Fs *>(pt); for (int k = 0; k < count; k++) fs_to_abi(dst + 12 * k, f[k]); }
    else { const Fp *f = reinterpret_cast<const Fp *>(pt); for (int k = 0; k < count; k++) fp_to_abi(dst + 12 * k, f[k]); }
}

// ---- K1: base preparation ------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(256) k_prep_bases(const uint32_t *__restrict__ abi, const uint8_t *__restrict__ is_inf, size_t n, uint32_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *src = abi + i * (2 * C::ABI_W);
    uint32_t w[2 * C::ABI_W];
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 2 * C::ABI_W; k += 4) {
        uint4 v = *reinterpret_cast<const uint4 *>(src + k);
        w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
        any |= v.x | v.y | v.z | v.w;
    }
    uint32_t flag = (any == 0) ? 1u : 0u;
    if (is_inf && is_inf[i]) flag = 1u;
    uint32_t *dst = out + i * C::AFF_STRIDE;
    store_coords_from_abi<C>(dst, w);
    dst[C::FLAGW] = flag;
}

// The same from the caller's own array of structures (dgpu_msm_*_strided): point i sits at raw + i * stride with its x at x_off and its y at
// y_off (each C::NFP * 48 bytes of ark-ff Montgomery limbs, 8-byte aligned) and, if inf_off != NO_INF_OFF, a flag byte at inf_off — ark-ec's
// in-memory `Affine { x, y, infinity }` (104 B for G1, 200 B for G2) goes over PCIe as it is, no host-side repacking.
constexpr size_t NO_INF_OFF = ~(size_t)0;
template <class C>
__global__ void __launch_bounds__(256) k_prep_bases_raw(const uint8_t *__restrict__ raw, size_t stride, size_t x_off, size_t y_off, size_t inf_off,
                                                        const uint8_t *__restrict__ is_inf, size_t n, uint32_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *pt = raw + i * stride;
    uint32_t w[2 * C::ABI_W];
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < C::ABI_W; k += 2) {
        const uint2 vx = *reinterpret_cast<const uint2 *>(pt + x_off + 4 * k), vy = *reinterpret_cast<const uint2 *>(pt + y_off + 4 * k);
        w[k] = vx.x; w[k + 1] = vx.y; w[C::ABI_W + k] = vy.x; w[C::ABI_W + k + 1] = vy.y;
        any |= vx.x | vx.y | vy.x | vy.y;
    }
    uint32_t flag = (any == 0) ? 1u : 0u;
    if (is_inf && is_inf[i]) flag = 1u;
    if (inf_off != NO_INF_OFF && pt[inf_off]) flag = 1u;
    uint32_t *dst = out + i * C::AFF_STRIDE;
    store_coords_from_abi<C>(dst, w);
    dst[C::FLAGW] = flag;
}

// ---- XYZZ <-> memory -------------------------------------------------------------------------------
// One record per point (array of structures): C::XW consecutive words (224 B for G1, 448 B for G2), moved as 16-byte pieces.  A lane
// that closes a bucket writes its record alone (lanes of a wave close runs at different times), so a word-major layout turned every
// store into 56 lone 4-byte writes on 56 cache lines: 5.6x write amplification at the HBM (1.04 GB written for 0.18 GB of payload,
// profiles/r01h_pmc_summary.txt).  A record is two cache lines written once.  The readers (fix-up, bucket reduction) take consecutive
// records per lane, so each of their lines is used in full as well.  Arrays are padded to a multiple of 64 points.
constexpr size_t SOA_TILE = 64;
__host__ __device__ inline size_t soa_points(size_t n) { return (n + SOA_TILE - 1) / SOA_TILE * SOA_TILE; }
template <class C> __device__ __forceinline__ void store_soa(uint32_t *__restrict__ base, size_t /*count*/, size_t b, const Xyzz<typename C::F> &p) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&p);
    uint32_t *t = base + b * C::XW;
#pragma unroll
    for (int k = 0; k < C::XW; k += 4) *reinterpret_cast<uint4 *>(t + k) = make_uint4(w[k], w[k + 1], w[k + 2], w[k + 3]);
}
template <class C> __device__ __forceinline__ void load_soa(Xyzz<typename C::F> &p, const uint32_t *__restrict__ base, size_t /*count*/, size_t b) {
    uint32_t *w = reinterpret_cast<uint32_t *>(&p);
    const uint32_t *t = base + b * C::XW;
#pragma unroll
    for (int k = 0; k < C::XW; k += 4) { const uint4 v = *reinterpret_cast<const uint4 *>(t + k); w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w; }
}
template <class C> __device__ __forceinline__ void load_aff(Aff<typename C::F> &p, const uint32_t *__restrict__ rec) {
    uint32_t *w = reinterpret_cast<uint32_t *>(&p);
#pragma unroll
    for (int k = 0; k < 2 * C::FW; k += 4) {
        uint4 v = *reinterpret_cast<const uint4 *>(rec + k);
        w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
    }
}

template <> __device__ __forceinline__ void load_aff<G1S>(Aff<Fs> &p, const uint32_t *__restrict__ rec) {
    uint32_t t[28];                                               // x[13] y[13] and two words of padding: seven 16-byte loads
#pragma unroll
    for (int k = 0; k < 28; k += 4) { const uint4 v = *reinterpret_cast<const uint4 *>(rec + k); t[k] = v.x; t[k + 1] = v.y; t[k + 2] = v.z; t[k + 3] = v.w; }
    uint32_t *w = reinterpret_cast<uint32_t *>(&p);
#pragma unroll
    for (int k = 0; k < 2 * SN; k++) w[k] = t[k];
}
template <> __device__ __forceinline__ void load_aff<G2S>(Aff<Fs2> &p, const uint32_t *__restrict__ rec) {
    uint32_t *w = reinterpret_cast<uint32_t *>(&p);
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < SN; j++) w[k * SN + j] = rec[k * PS + j];
}
// an affine point into a base record (coordinates only; the caller sets the flag word)
template <class C> __device__ __forceinline__ void store_aff_record(uint32_t *__restrict__ dst, const Aff<typename C::F> &a) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&a);
    if constexpr (std::is_same<typename C::F, Fs2>::value) {
        for (int k = 0; k < 4; k++) { for (int j = 0; j < SN; j++) dst[k * PS + j] = w[k * SN + j]; dst[k * PS + SN] = 0u; }
        for (int j = 4 * PS; j < C::AFF_STRIDE; j++) dst[j] = 0u;
    } else {
        for (int j = 0; j < 2 * C::FW; j++) dst[j] = w[j];
        for (int j = 2 * C::FW; j < C::AFF_STRIDE; j++) dst[j] = 0u;
    }
}
// lane-pair variants: this lane moves only its half (c0 on even lanes, c1 on odd lanes) of every coordinate: 13 words out of a 14-word slot
__device__ __forceinline__ void load_half(uint32_t *__restrict__ w, const uint32_t *__restrict__ src) {
#pragma unroll
    for (int j = 0; j + 1 < SN; j += 2) { const uint2 v = *reinterpret_cast<const uint2 *>(src + j); w[j] = v.x; w[j + 1] = v.y; }
    w[SN - 1] = src[SN - 1];
}
__device__ __forceinline__ void store_half(uint32_t *__restrict__ dst, const uint32_t *__restrict__ w) {
#pragma unroll
    for (int j = 0; j + 1 < SN; j += 2) *reinterpret_cast<uint2 *>(dst + j) = make_uint2(w[j], w[j + 1]);
    *reinterpret_cast<uint2 *>(dst + SN - 1) = make_uint2(w[SN - 1], 0u);
}
template <> __device__ __forceinline__ void store_soa<G2P>(uint32_t *__restrict__ base, size_t /*count*/, size_t b, const Xyzz<Fs2H> &p) {
    const uint32_t h = threadIdx.x & 1u;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&p);      // x, y, zz, zzz halves: 4 x 13 words
    uint32_t *t = base + b * G2P::XW;                               // G2 word order: coordinate k, half h in the slot (2k + h) * PS (56-byte slots, 8-byte aligned)
#pragma unroll
    for (int k = 0; k < 4; k++) store_half(t + (2 * k + h) * PS, w + k * SN);
}
template <> __device__ __forceinline__ void load_soa<G2P>(Xyzz<Fs2H> &p, const uint32_t *__restrict__ base, size_t /*count*/, size_t b) {
    const uint32_t h = threadIdx.x & 1u;
    uint32_t *w = reinterpret_cast<uint32_t *>(&p);
    const uint32_t *t = base + b * G2P::XW;
#pragma unroll
    for (int k = 0; k < 4; k++) load_half(w + k * SN, t + (2 * k + h) * PS);
}
template <> __device__ __forceinline__ void load_aff<G2P>(Aff<Fs2H> &p, const uint32_t *__restrict__ rec) {
    const uint32_t h = threadIdx.x & 1u;
    uint32_t *w = reinterpret_cast<uint32_t *>(&p);
#pragma unroll
    for (int k = 0; k < 2; k++) load_half(w + k * SN, rec + (2 * k + h) * PS);
}

// ---- K5: chunked bucket accumulation ------------------------------------------------------------------
// off[0..NB] = exclusive scan of the histogram (off[NB] = number of terms E).  Thread t owns terms
// [t*CH, min((t+1)*CH, E)).  head_b[t] / tail_b[t] = bucket of the partial left in the head / tail slot, or
// 0xffffffff.  bucket_inf[b] must be pre-set to 1; it is cleared by whoever writes bucket b.
// SKIP_ID: terms whose base record carries the identity flag are passed over here (the sorted list came from a sort shared by several
// tables, which cannot drop them per table: dgpu_scalars_sort / dgpu_msm_*_sorted)
// RowMap (SKIP_ID instance only): the list was sorted for tables of rows_src rows per window; this table has rows_dst = rows_src - shift rows
// and holds the points of rows shift .. rows_src - 1 (the l_query of a proof against the list sorted for its a_query): entry w rows_src + r
// becomes w rows_dst + (r - shift), entries with r < shift are passed over.  rows_src == 0: entries are used as they are.
struct RowMap { uint32_t rows_src = 0, rows_dst = 0, shift = 0; uint64_t magic = 0; /* ceil(2^64 / rows_src) */ };
template <class C, bool SKIP_ID = false>
__global__ void __launch_bounds__(256, C::ACC_WAVES) k_accumulate(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ entries, const uint32_t *__restrict__ off,
                                                    uint32_t NB, uint32_t *__restrict__ bucket, uint8_t *__restrict__ bucket_inf,
                                                    uint32_t *__restrict__ head, uint32_t *__restrict__ tail, uint32_t *__restrict__ head_b, uint32_t *__restrict__ tail_b,
                                                    uint8_t *__restrict__ part_inf, size_t T, uint32_t CH, uint32_t dbg_mask, const uint32_t *__restrict__ dyn, RowMap map) {
    typedef typename C::F F;
    size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / C::LPP;   // C::LPP lanes cooperate on one chunk
    if (dyn) { CH = dyn[DYN_CH]; T = dyn[DYN_T]; }      // chunking chosen on the device from the number of terms the sort produced
    if (t >= T) return;
    const uint32_t E = off[NB];
    uint64_t start64 = (uint64_t)t * CH;
    uint32_t hb = 0xffffffffu, tb = 0xffffffffu;
    if (start64 >= E) { head_b[t] = hb; tail_b[t] = tb; return; }
    uint32_t start = (uint32_t)start64;
    uint32_t end = (E - start > (uint32_t)CH) ? start + CH : E;
    // bucket containing `start`: largest b with off[b] <= start
    uint32_t lo = 0, hi = NB;   // invariant off[lo] <= start < off[hi]
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (off[mid] <= start) lo = mid; else hi = mid; }
    uint32_t b = lo;
    uint32_t bend = off[b + 1];
    bool started_before = off[b] < start;
    Xyzz<F> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (uint32_t pos = start; pos < end; pos++) {
        if (pos == bend) {
            // the run of bucket b ended inside this chunk
            if (started_before) { store_soa<C>(head, T, t, acc); part_inf[2 * t] = inf; hb = b; started_before = false; }
            else { store_soa<C>(bucket, NB, b, acc); bucket_inf[b] = inf; }
            inf = true;
            b++; bend = off[b + 1];
            if (bend == pos) {
                // empty buckets follow: find the bucket that holds `pos` (largest b with off[b] <= pos) by bisection — with a handful of
                // hot buckets (all scalars equal) a linear walk crossed tens of thousands of empty ones on a dependent load each (14 ms)
                uint32_t l2 = b, h2 = NB;            // off[l2] <= pos < off[h2] (= E)
                while (h2 - l2 > 1) { uint32_t mid = (l2 + h2) >> 1; if (off[mid] <= pos) l2 = mid; else h2 = mid; }
                b = l2; bend = off[b + 1];
            }
        }
        uint32_t e = entries[pos];
        uint32_t row = e & 0x7fffffffu & dbg_mask;
        if constexpr (SKIP_ID) {
            if (map.rows_src) {
                const uint32_t w = (uint32_t)__umul64hi((uint64_t)row, map.magic);      // row / rows_src (exact: row rows_src < 2^64)
                const uint32_t r = row - w * map.rows_src;
                if (r < map.shift) continue;
                row = w * map.rows_dst + (r - map.shift);
            }
        }
        const uint32_t *rec = bases + (size_t)row * C::AFF_STRIDE;
        if constexpr (SKIP_ID) { if (rec[C::FLAGW] != 0) continue; }
        Aff<F> p; load_aff<C>(p, rec);
        xyzz_madd(acc, inf, p, (e >> 31) != 0);
    }
    // the last run reaches the chunk end
    bool complete = (end == bend);
    if (started_before) { store_soa<C>(head, T, t, acc); part_inf[2 * t] = inf; hb = b; }   // cut on the left (and maybe on the right too)
    else if (complete) { store_soa<C>(bucket, NB, b, acc); bucket_inf[b] = inf; }
    else { store_soa<C>(tail, T, t, acc); part_inf[2 * t + 1] = inf; tb = b; }              // cut on the right only
    head_b[t] = hb; tail_b[t] = tb;
}

// ---- K6: fold the partials of buckets that were cut by chunk borders ------------------------------------
// The leftmost piece of a cut bucket is always a tail slot (its run starts at the bucket start); every later
// piece is a head slot of the following chunks.
template <class C>
__global__ void __launch_bounds__(256) k_fixup(uint32_t NB, uint32_t *__restrict__ bucket, uint8_t *__restrict__ bucket_inf,
                                               const uint32_t *__restrict__ head, const uint32_t *__restrict__ tail, const uint32_t *__restrict__ head_b,
                                               const uint32_t *__restrict__ tail_b, const uint8_t *__restrict__ part_inf, size_t T,
                                               const uint32_t *__restrict__ off, uint32_t heavy_thr, const uint32_t *__restrict__ dyn) {
    typedef typename C::F F;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (dyn) { T = dyn[DYN_T]; heavy_thr = dyn[DYN_HEAVY]; }
    if (t >= T) return;
    uint32_t b = tail_b[t];
    if (b == 0xffffffffu) return;
    if (off[b + 1] - off[b] >= heavy_thr) return;   // folded by k_fixup_heavy
    Xyzz<F> acc; load_soa<C>(acc, tail, T, t);
    bool inf = part_inf[2 * t + 1] != 0;
    for (size_t k = t + 1; k < T && head_b[k] == b; k++) {
        Xyzz<F> o; load_soa<C>(o, head, T, k);
        xyzz_add(acc, inf, o, part_inf[2 * k] != 0);
    }
    store_soa<C>(bucket, NB, b, acc);
    bucket_inf[b] = inf;
}

// Buckets with >= heavy_thr terms span many chunks; their pieces (the tail slot of the chunk holding the bucket's first term, then the head
// slots of every following chunk up to the one holding its last term) are folded by whole blocks: strided serial sums per thread, then a
// tree through LDS.  k_fixup_heavy folds a bucket of up to HEAVY_RANGE pieces with one block and puts a longer one (all scalars equal; the
// `1` of a Groth16 witness: 10^5 .. 10^7 terms in one bucket) on the list dyn[DYN_MULTI ..].  Those are cut at multiples of HEAVY_RANGE chunk
// indices: k_fixup_heavy_ranges, one block per (bucket, range), leaves the range's sum in hpart[2 k + (the bucket starts in range k)] (at most
// one long bucket ends and one starts inside a range); k_fixup_heavy_join folds a bucket's ranges.  Worst case ~2 x (HEAVY_RANGE / threads +
// log2(threads)) dependent additions instead of pieces / threads.
template <class C> __device__ __forceinline__ void block_tree_fold(Xyzz<typename C::F> &acc, bool &inf, size_t items, uint32_t *sh, uint8_t *shinf) {
    // C::LPP lanes hold one point (G2P: the two halves of every Fp2 coordinate sit on a lane pair): the tree pairs up POINTS, so partner lanes
    // are s2 * LPP threads apart and a pair always moves together
    typedef typename C::F F;
    constexpr int BT = C::HEAVY_T, BI = BT / C::LPP, TW = (int)(sizeof(Xyzz<F>) / 4);
    const int it = (int)threadIdx.x / C::LPP;
    for (int s2 = BI / 2; s2 >= 1; s2 >>= 1) {
        if ((size_t)s2 >= items) continue;             // (uniform per block) points s2 .. 2 s2 - 1 hold nothing yet
        __syncthreads();
        if (it >= s2 && it < 2 * s2) {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(&acc);
            for (int k = 0; k < TW; k++) sh[k * BT + threadIdx.x] = w[k];
            shinf[threadIdx.x] = inf;
        }
        __syncthreads();
        if (it < s2) {
            Xyzz<F> o; uint32_t *w = reinterpret_cast<uint32_t *>(&o);
            for (int k = 0; k < TW; k++) w[k] = sh[k * BT + threadIdx.x + s2 * C::LPP];
            xyzz_add(acc, inf, o, shinf[threadIdx.x + s2 * C::LPP] != 0);
        }
    }
    __syncthreads();
}
// sum of the pieces lo .. up of the bucket whose first piece is t0 (left on thread 0)
template <class C> __device__ __forceinline__ void fold_pieces(Xyzz<typename C::F> &acc, bool &inf, size_t t0, size_t lo, size_t up, const uint32_t *__restrict__ head, const uint32_t *__restrict__ tail,
                                                               const uint8_t *__restrict__ part_inf, size_t T, uint32_t *sh, uint8_t *shinf) {
    typedef typename C::F F;
    inf = true; fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (size_t t = lo + threadIdx.x / C::LPP; t <= up; t += C::HEAVY_T / C::LPP) {
        Xyzz<F> o; bool oinf;
        if (t == t0) { load_soa<C>(o, tail, T, t); oinf = part_inf[2 * t + 1] != 0; }
        else { load_soa<C>(o, head, T, t); oinf = part_inf[2 * t] != 0; }
        xyzz_add(acc, inf, o, oinf);
    }
    block_tree_fold<C>(acc, inf, up - lo + 1, sh, shinf);
}
template <class C>
__global__ void __launch_bounds__(C::HEAVY_T) k_fixup_heavy(const uint32_t *__restrict__ heavy, uint32_t heavy_cap, const uint32_t *__restrict__ off, uint32_t CH,
                                                     uint32_t NB, uint32_t *__restrict__ bucket, uint8_t *__restrict__ bucket_inf,
                                                     const uint32_t *__restrict__ head, const uint32_t *__restrict__ tail, const uint8_t *__restrict__ part_inf, size_t T, uint32_t *__restrict__ dyn) {
    typedef typename C::F F;
    constexpr int BT = C::HEAVY_T;
    CH = dyn[DYN_CH]; T = dyn[DYN_T];
    __shared__ uint32_t sh[(sizeof(Xyzz<typename C::F>) / 4) * BT];
    __shared__ uint8_t shinf[BT];
    uint32_t nh = heavy[0]; if (nh > heavy_cap) nh = heavy_cap;
    for (uint32_t hi = blockIdx.x; hi < nh; hi += gridDim.x) {
        const uint32_t b = heavy[1 + hi];
        const size_t t0 = off[b] / CH, t1 = (off[b + 1] - 1) / CH;
        if (t1 - t0 + 1 > HEAVY_RANGE) { if (threadIdx.x == 0) { const uint32_t m = atomicAdd(&dyn[DYN_NMULTI], 1u); dyn[DYN_MULTI + m] = b; } continue; }   // (at most T / HEAVY_RANGE of them)
        Xyzz<F> acc; bool inf;
        fold_pieces<C>(acc, inf, t0, t0, t1, head, tail, part_inf, T, sh, shinf);
        if (threadIdx.x < C::LPP) { store_soa<C>(bucket, NB, b, acc); if (threadIdx.x == 0) bucket_inf[b] = inf; }
    }
}
template <class C>
__global__ void __launch_bounds__(C::HEAVY_T) k_fixup_heavy_ranges(const uint32_t *__restrict__ off, const uint32_t *__restrict__ head, const uint32_t *__restrict__ tail, const uint8_t *__restrict__ part_inf,
                                                            const uint32_t *__restrict__ dyn, uint32_t *__restrict__ hpart, uint8_t *__restrict__ hpart_inf) {
    typedef typename C::F F;
    constexpr int BT = C::HEAVY_T;
    const uint32_t CH = dyn[DYN_CH], nm = dyn[DYN_NMULTI];
    const size_t T = dyn[DYN_T];
    __shared__ uint32_t sh[(sizeof(Xyzz<typename C::F>) / 4) * BT];
    __shared__ uint8_t shinf[BT];
    // ranges along x: consecutive workgroup ids go to different XCDs / CUs.  (With the buckets along x the few blocks that have work sat 32
    // ids apart and were all dispatched to the same eight CUs: 64 ranges took 8 rounds, 1.17 ms instead of 0.2.)
    for (uint32_t mi = blockIdx.y; mi < nm; mi += gridDim.y) {
        const uint32_t b = dyn[DYN_MULTI + mi];
        const size_t t0 = off[b] / CH, t1 = (off[b + 1] - 1) / CH, k0 = t0 / HEAVY_RANGE, k1 = t1 / HEAVY_RANGE;
        for (size_t k = k0 + blockIdx.x; k <= k1; k += gridDim.x) {
            const size_t lo = k * HEAVY_RANGE > t0 ? k * HEAVY_RANGE : t0, up = (k + 1) * HEAVY_RANGE - 1 < t1 ? (k + 1) * HEAVY_RANGE - 1 : t1;
            Xyzz<F> acc; bool inf;
            fold_pieces<C>(acc, inf, t0, lo, up, head, tail, part_inf, T, sh, shinf);
            if (threadIdx.x < C::LPP) { const size_t slot = 2 * k + (k == k0 ? 1 : 0); store_soa<C>(hpart, 0, slot, acc); if (threadIdx.x == 0) hpart_inf[slot] = inf; }
        }
    }
}
template <class C>
__global__ void __launch_bounds__(C::HEAVY_T) k_fixup_heavy_join(const uint32_t *__restrict__ off, uint32_t NB, uint32_t *__restrict__ bucket, uint8_t *__restrict__ bucket_inf, const uint32_t *__restrict__ dyn,
                                                          const uint32_t *__restrict__ hpart, const uint8_t *__restrict__ hpart_inf) {
    typedef typename C::F F;
    constexpr int BT = C::HEAVY_T;
    const uint32_t CH = dyn[DYN_CH], nm = dyn[DYN_NMULTI];
    __shared__ uint32_t sh[(sizeof(Xyzz<typename C::F>) / 4) * BT];
    __shared__ uint8_t shinf[BT];
    for (uint32_t mi = blockIdx.x; mi < nm; mi += gridDim.x) {
        const uint32_t b = dyn[DYN_MULTI + mi];
        const size_t t0 = off[b] / CH, t1 = (off[b + 1] - 1) / CH, k0 = t0 / HEAVY_RANGE, k1 = t1 / HEAVY_RANGE;
        Xyzz<F> acc; bool inf = true;
        fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
        for (size_t k = k0 + threadIdx.x / C::LPP; k <= k1; k += BT / C::LPP) {
            const size_t slot = 2 * k + (k == k0 ? 1 : 0);
            Xyzz<F> o; load_soa<C>(o, hpart, 0, slot);
            xyzz_add(acc, inf, o, hpart_inf[slot] != 0);
        }
        block_tree_fold<C>(acc, inf, k1 - k0 + 1, sh, shinf);
        if (threadIdx.x < C::LPP) { store_soa<C>(bucket, NB, b, acc); if (threadIdx.x == 0) bucket_inf[b] = inf; }
    }
}

// ---- K7/K8: sum_k (k+1) * B_k ----------------------------------------------------------------------------
template <class C> __device__ __forceinline__ void shfl_down_xyzz(Xyzz<typename C::F> &o, bool &oinf, const Xyzz<typename C::F> &x, bool xinf, int d) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&x);
    uint32_t *q = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
    for (int k = 0; k < C::XW; k++) q[k] = __shfl_down(w[k], d, 64);
    int fi = __shfl_down((int)xinf, d, 64);
    oinf = (fi != 0) || ((int)(threadIdx.x & 63) + d >= 64);
}
// Lane l holds (S_l, A_l): the plain and the locally weighted sum of its own items, lane l's items having
// global weights offset by l * 2^shift.  Returns on lane 0: S = sum S_l, A = sum (A_l + l 2^shift S_l).
template <class C> __device__ __forceinline__ void wave_weighted_sum(Xyzz<typename C::F> &S, bool &sinf, Xyzz<typename C::F> &A, bool &ainf, int shift) {
    typedef typename C::F F;
    const int lane = threadIdx.x & 63;
    // suffix scan: S_l <- sum_{j >= l} S_j
    for (int d = 1; d < 64; d <<= 1) {
        Xyzz<F> o; bool oinf; shfl_down_xyzz<C>(o, oinf, S, sinf, d);
        xyzz_add(S, sinf, o, oinf);
    }
    // sum_l l S_l = sum_{l >= 1} suffix_l ;  fold 2^shift * suffix_l into A_l before the reduction
    Xyzz<F> y = S; bool yinf = (lane == 0) ? true : sinf;
    for (int k = 0; k < shift; k++) { if (!yinf) { Xyzz<F> d2; xyzz_dbl(d2, y); y = d2; } }
    xyzz_add(A, ainf, y, yinf);
    for (int d = 32; d >= 1; d >>= 1) {
        Xyzz<F> o; bool oinf; shfl_down_xyzz<C>(o, oinf, A, ainf, d);
        xyzz_add(A, ainf, o, oinf);
    }
}

// One wave per group of 64 * m consecutive buckets of one window (m = 2^mshift buckets per lane).
// Writes (S_g, A_g) to l1[2 g], l1[2 g + 1] (AoS, C::XW words each) and their flags to l1_inf.
// Blocks of four waves (one per SIMD of a CU, each wave a group of its own; no block-level synchronisation): a launch that does not fill
// the chip then leaves WHOLE CUs free, and the 256-thread blocks of another call's k_accumulate (which need a slot on all four SIMDs of a
// CU) can run beside it — single-wave blocks are spread over every CU and lock them all.
template <class C>
__global__ void __launch_bounds__(256, 1) k_reduce_l0(const uint32_t *__restrict__ bucket, const uint8_t *__restrict__ bucket_inf, uint32_t NB, int mshift,
                                                  uint32_t *__restrict__ l1, uint8_t *__restrict__ l1_inf, unsigned NG) {
    typedef typename C::F F;
    const int lane = threadIdx.x & 63;
    const uint32_t m = 1u << mshift;
    size_t g = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= NG) return;
    size_t b0 = (g * 64 + lane) * (size_t)m;
    // a group without a single filled bucket (short scalars: 16-bit values leave 15 of 16 pseudo-windows empty) is two identity flags
    { bool any = false; for (uint32_t k = 0; k < m; k++) any = any || bucket_inf[b0 + k] == 0;
      if (!__any(any)) { if (lane == 0) { l1_inf[2 * g] = 1; l1_inf[2 * g + 1] = 1; } return; } }
    Xyzz<F> run, tot; bool rinf = true, tinf = true;
    fzero(run.x); fzero(run.y); fzero(run.zz); fzero(run.zzz); tot = run;
    for (int k = (int)m - 1; k >= 0; k--) {
        size_t b = b0 + k;
        bool binf = bucket_inf[b] != 0;
        Xyzz<F> p;
        if (!binf) load_soa<C>(p, bucket, NB, b); else p = run;
        xyzz_add(run, rinf, p, binf);
        xyzz_add(tot, tinf, run, rinf);
    }
    wave_weighted_sum<C>(run, rinf, tot, tinf, mshift);
    if (lane == 0) {
        uint32_t *dst = l1 + g * 2 * C::XW;
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&run);
        const uint32_t *v = reinterpret_cast<const uint32_t *>(&tot);
        for (int k = 0; k < C::XW; k++) { dst[k] = w[k]; dst[C::XW + k] = v[k]; }
        l1_inf[2 * g] = rinf; l1_inf[2 * g + 1] = tinf;
    }
}
// One wave per window: lane g holds group g's (S_g, A_g) (G <= 64 groups per window), group weights offset by
// g * 2^gshift.  Converts the window sum to the ABI form (X, Y, ZZ, ZZZ; 2^384 Montgomery) for the host.
template <class C>
__global__ void __launch_bounds__(64) k_reduce_top(const uint32_t *__restrict__ l1, const uint8_t *__restrict__ l1_inf, int G, int gshift,
                                                   uint32_t *__restrict__ win_abi, uint8_t *__restrict__ win_inf,
                                                   uint32_t *__restrict__ win_s_abi = nullptr, uint8_t *__restrict__ win_s_inf = nullptr) {
    typedef typename C::F F;
    const int lane = threadIdx.x & 63;
    size_t w = blockIdx.x;
    Xyzz<F> S, A; bool sinf = true, ainf = true;
    fzero(S.x); fzero(S.y); fzero(S.zz); fzero(S.zzz); A = S;
    if (lane < G) {
        const uint32_t *src = l1 + (w * G + lane) * 2 * C::XW;
        uint32_t *ps = reinterpret_cast<uint32_t *>(&S), *pa = reinterpret_cast<uint32_t *>(&A);
        for (int k = 0; k < C::XW; k++) { ps[k] = src[k]; pa[k] = src[C::XW + k]; }
        sinf = l1_inf[2 * (w * G + lane)] != 0; ainf = l1_inf[2 * (w * G + lane) + 1] != 0;
    }
    if (!__any(!sinf || !ainf)) {        // an empty (pseudo-)window
        if (lane == 0) { win_inf[w] = 1; if (win_s_abi) win_s_inf[w] = 1; }
        return;
    }
    wave_weighted_sum<C>(S, sinf, A, ainf, gshift);
    if (lane == 0) {
        uint32_t *dst = win_abi + w * 4 * C::ABI_W;
        win_inf[w] = ainf;
        if (!ainf) coords_to_abi<typename C::F>(dst, &A, 4 * C::NFP);
        if (win_s_abi) {                 // the plain sum of the window's buckets: the next level of the weighted sum is taken on the host
            win_s_inf[w] = sinf;
            if (!sinf) coords_to_abi<typename C::F>(win_s_abi + w * 4 * C::ABI_W, &S, 4 * C::NFP);
        }
    }
}


// ---- bucket sets of an MSM taken in term ranges (one-shot / fresh-scalar calls whose operands are still crossing PCIe, msm_driver.hip.h) ----
// dst[b] += src[b] for every bucket: C::LPP lanes per bucket (G2P: a lane pair).  About NB general additions: 0.1 ms at 2^19 buckets.
template <class C>
__global__ void __launch_bounds__(256) k_merge_buckets(uint32_t NB, uint32_t *__restrict__ dst, uint8_t *__restrict__ dst_inf, const uint32_t *__restrict__ src, const uint8_t *__restrict__ src_inf) {
    typedef typename C::F F;
    const size_t b = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / C::LPP;
    if (b >= NB) return;
    if (src_inf[b] != 0) return;
    Xyzz<F> o; load_soa<C>(o, src, NB, b);
    bool inf = dst_inf[b] != 0;
    Xyzz<F> acc;
    if (inf) { acc = o; inf = false; }
    else { load_soa<C>(acc, dst, NB, b); xyzz_add(acc, inf, o, false); }
    store_soa<C>(dst, NB, b, acc);
    if ((threadIdx.x % C::LPP) == 0) dst_inf[b] = inf;
}

// ---- K7/K8 for G2 on lane pairs (fp2_pair.hip.h) ---------------------------------------------------------------------------------
// Same group geometry and l1 layout as k_reduce_l0 / k_reduce_top, but a point lives on a lane pair (even lane: c0 halves, odd lane: c1
// halves), so a wave holds 32 points and every point-lane takes twice the items.  The one-lane Fp2 addition needs > 256 VGPRs (spills)
// and is ~3x the instructions of a half: the pair form runs the same dependent chain in about half the time (3.0 -> 1.6 ms at 2^20).
__device__ __forceinline__ void load_soa_pair(Xyzz<Fs2H> &p, const uint32_t *__restrict__ base, size_t b) { load_soa<G2P>(p, base, 0, b); }
__device__ __forceinline__ void shfl_down_pair(Xyzz<Fs2H> &o, bool &oinf, const Xyzz<Fs2H> &x, bool xinf, int d /* point-lanes */) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&x);
    uint32_t *q = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
    for (int k = 0; k < 4 * SN; k++) q[k] = __shfl_down(w[k], 2 * d, 64);
    int fi = __shfl_down((int)xinf, 2 * d, 64);
    oinf = (fi != 0) || ((int)((threadIdx.x & 63) >> 1) + d >= 32);
}
// point-lane q holds (S_q, A_q), its items' global weights offset by q * 2^shift; on return point-lane 0 holds S = sum S_q, A = sum (A_q + q 2^shift S_q)
__device__ __forceinline__ void wave_weighted_sum_pair(Xyzz<Fs2H> &S, bool &sinf, Xyzz<Fs2H> &A, bool &ainf, int shift) {
    const int q = (threadIdx.x & 63) >> 1;
    for (int d = 1; d < 32; d <<= 1) {
        Xyzz<Fs2H> o; bool oinf; shfl_down_pair(o, oinf, S, sinf, d);
        xyzz_add(S, sinf, o, oinf);
    }
    Xyzz<Fs2H> y = S; bool yinf = (q == 0) ? true : sinf;
    for (int k = 0; k < shift; k++) { if (!yinf) { Xyzz<Fs2H> d2; xyzz_dbl(d2, y); y = d2; } }
    xyzz_add(A, ainf, y, yinf);
    for (int d = 16; d >= 1; d >>= 1) {
        Xyzz<Fs2H> o; bool oinf; shfl_down_pair(o, oinf, A, ainf, d);
        xyzz_add(A, ainf, o, oinf);
    }
}
__device__ __forceinline__ void store_l1_pair(uint32_t *__restrict__ dst, const Xyzz<Fs2H> &p) {       // AoS, G2 word order: coordinate k, half h at (2k + h) * NL
    const uint32_t h = threadIdx.x & 1u;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&p);
    for (int k = 0; k < 4; k++) store_half(dst + (2 * k + h) * PS, w + k * SN);
}
__device__ __forceinline__ void load_l1_pair(Xyzz<Fs2H> &p, const uint32_t *__restrict__ src) {
    const uint32_t h = threadIdx.x & 1u;
    uint32_t *w = reinterpret_cast<uint32_t *>(&p);
    for (int k = 0; k < 4; k++) load_half(w + k * SN, src + (2 * k + h) * PS);
}
// one wave per group of 64 * m buckets: 32 point-lanes x 2m buckets each
template <class PAIR /* = G2P: a template only so that the header may be included by several translation units */>
__global__ void __launch_bounds__(256, 1) k_reduce_l0_pair(const uint32_t *__restrict__ bucket, const uint8_t *__restrict__ bucket_inf, uint32_t NB, int mshift,
                                                       uint32_t *__restrict__ l1, uint8_t *__restrict__ l1_inf, unsigned NG) {
    typedef Fs2H F;
    const int q = (threadIdx.x & 63) >> 1;
    const uint32_t m2 = 2u << mshift;                    // buckets per point-lane
    const size_t g = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // (four-wave blocks: see k_reduce_l0)
    if (g >= NG) return;
    const size_t b0 = (g * 32 + q) * (size_t)m2;
    { bool any = false; for (uint32_t k = 0; k < m2; k++) any = any || bucket_inf[b0 + k] == 0;
      if (!__any(any)) { if ((threadIdx.x & 63) == 0) { l1_inf[2 * g] = 1; l1_inf[2 * g + 1] = 1; } return; } }
    Xyzz<F> run, tot; bool rinf = true, tinf = true;
    fzero(run.x); fzero(run.y); fzero(run.zz); fzero(run.zzz); tot = run;
    for (int k = (int)m2 - 1; k >= 0; k--) {
        const size_t b = b0 + k;
        const bool binf = bucket_inf[b] != 0;
        Xyzz<F> p;
        if (!binf) load_soa_pair(p, bucket, b); else p = run;
        xyzz_add(run, rinf, p, binf);
        xyzz_add(tot, tinf, run, rinf);
    }
    wave_weighted_sum_pair(run, rinf, tot, tinf, mshift + 1);
    if (q == 0) {
        uint32_t *dst = l1 + g * 2 * G2P::XW;
        store_l1_pair(dst, run); store_l1_pair(dst + G2P::XW, tot);
        if ((threadIdx.x & 1u) == 0) { l1_inf[2 * g] = rinf; l1_inf[2 * g + 1] = tinf; }
    }
}
// one wave per window: point-lane q first folds groups 2q and 2q + 1 (weights offset by 2^gshift), then the wave sum over 32 point-lanes
template <class PAIR>
__global__ void __launch_bounds__(64) k_reduce_top_pair(const uint32_t *__restrict__ l1, const uint8_t *__restrict__ l1_inf, int G, int gshift,
                                                        uint32_t *__restrict__ win_abi, uint8_t *__restrict__ win_inf,
                                                        uint32_t *__restrict__ win_s_abi = nullptr, uint8_t *__restrict__ win_s_inf = nullptr) {
    typedef Fs2H F;
    const int q = (threadIdx.x & 63) >> 1;
    const uint32_t h = threadIdx.x & 1u;
    const size_t w = blockIdx.x;
    const bool has0 = 2 * q < G, has1 = 2 * q + 1 < G;
    const size_t g0 = w * G + 2 * q, g1 = g0 + 1;
    auto zero = [](Xyzz<F> &p) { fzero(p.x); fzero(p.y); fzero(p.zz); fzero(p.zzz); };
    // A = A_0 + A_1 + 2^gshift S_1 ;  S = S_0 + S_1     (one operand pair live at a time: four points at once would spill)
    Xyzz<F> A; bool ainf = true; zero(A);
    if (has0) { load_l1_pair(A, l1 + g0 * 2 * G2P::XW + G2P::XW); ainf = l1_inf[2 * g0 + 1] != 0; }
    { Xyzz<F> t; bool tinf = true; zero(t);
      if (has1) { load_l1_pair(t, l1 + g1 * 2 * G2P::XW + G2P::XW); tinf = l1_inf[2 * g1 + 1] != 0; }
      xyzz_add(A, ainf, t, tinf); }
    Xyzz<F> S1; bool s1inf = true; zero(S1);
    if (has1) { load_l1_pair(S1, l1 + g1 * 2 * G2P::XW); s1inf = l1_inf[2 * g1] != 0; }
    { Xyzz<F> y = S1; bool yinf = s1inf;
      for (int k = 0; k < gshift; k++) { if (!yinf) { Xyzz<F> d2; xyzz_dbl(d2, y); y = d2; } }
      xyzz_add(A, ainf, y, yinf); }
    Xyzz<F> S; bool sinf = true; zero(S);
    if (has0) { load_l1_pair(S, l1 + g0 * 2 * G2P::XW); sinf = l1_inf[2 * g0] != 0; }
    xyzz_add(S, sinf, S1, s1inf);
    wave_weighted_sum_pair(S, sinf, A, ainf, gshift + 1);
    if (q == 0) {
        uint32_t *dst = win_abi + w * 4 * G2::ABI_W;
        if (h == 0) win_inf[w] = ainf;
        if (!ainf) {
            const Fs *f = reinterpret_cast<const Fs *>(&A);          // x, y, zz, zzz halves
            for (int k = 0; k < 4; k++) fs_to_abi(dst + 12 * (2 * k + h), f[k]);
        }
        if (win_s_abi) {
            if (h == 0) win_s_inf[w] = sinf;
            if (!sinf) { const Fs *f = reinterpret_cast<const Fs *>(&S); for (int k = 0; k < 4; k++) fs_to_abi(win_s_abi + w * 4 * G2::ABI_W + 12 * (2 * k + h), f[k]); }
        }
    }
}


// ---- K8 with four lanes (G2: four lane pairs) per point ------------------------------------------------------------------------------
// k_reduce_top is a handful of lone waves on a chain of ~19 general additions (6 of the suffix scan, the doublings by 2^gshift, 6 of the
// tree): 0.34 ms of a 3.3 ms G1 call, 0.64 of an 8.8 ms G2 call, and no other work of the call can hide it.  A general addition is only
// four products deep (ec29.hip.h xyzz_add_rounds), a doubling three: here every group's (S, A) is held by FOUR members (lanes; lane pairs
// for G2) in identical copies, each member multiplies ONE role-selected operand pair per round and the four results are broadcast — DPP
// quad_perm for G1, ds_bpermute inside the group of eight lanes for G2.  The exchange between groups (the shuffles of
// wave_weighted_sum) goes through LDS: member r parks coordinate r.  Same group geometry, l1 layout and outputs as k_reduce_top(_pair).
// one of four values by the member's role, without control flow: nested ternaries on a lane-dependent value came out of the compiler as a
// branch per limb (s_cbranch_execz: ~60 per product), which made the four-member addition 2.5x as long as its instruction count
template <class T> __device__ __forceinline__ T pick4(int role, T a, T b, T c, T d) {
    const uint32_t m1 = (role & 1) ? 0xffffffffu : 0u, m2 = (role & 2) ? 0xffffffffu : 0u;
    const uint32_t lo = ((uint32_t)b & m1) | ((uint32_t)a & ~m1), hi = ((uint32_t)d & m1) | ((uint32_t)c & ~m1);
    return (T)((hi & m2) | (lo & ~m2));
}
template <int LPP> struct QuadLanes;
template <> struct QuadLanes<1> {                       // members = the four lanes of a quad
    int role; int src[4];
    __device__ QuadLanes() : role((int)(threadIdx.x & 3u)) {
        const int lane = (int)(threadIdx.x & 63u), base = lane & ~3;
        for (int k = 0; k < 4; k++) src[k] = (base + k) * 4;
    }
    __device__ __forceinline__ void mul4(Fs (&r)[4], const Fs (&a)[4], const Fs (&b)[4], int used) const {
        Fs m1, m2, p;
#pragma unroll
        for (int i = 0; i < SN; i++) { m1.l[i] = pick4(role, a[0].l[i], a[1].l[i], a[2].l[i], a[3].l[i]); m2.l[i] = pick4(role, b[0].l[i], b[1].l[i], b[2].l[i], b[3].l[i]); }
        fs_mul(p, m1, m2);
#pragma unroll
        for (int i = 0; i < SN; i++) {
            r[0].l[i] = __builtin_amdgcn_ds_bpermute(src[0], p.l[i]); r[1].l[i] = __builtin_amdgcn_ds_bpermute(src[1], p.l[i]);
            if (used > 2) r[2].l[i] = __builtin_amdgcn_ds_bpermute(src[2], p.l[i]);
            if (used > 3) r[3].l[i] = __builtin_amdgcn_ds_bpermute(src[3], p.l[i]);
        }
    }
};
template <> struct QuadLanes<2> {                       // members = the four lane pairs of a group of eight lanes
    int role; int src[4];
    __device__ QuadLanes() : role((int)((threadIdx.x >> 1) & 3u)) {
        const int lane = (int)(threadIdx.x & 63u), base = lane & ~7, h = lane & 1;
        for (int k = 0; k < 4; k++) src[k] = (base + 2 * k + h) * 4;
    }
    __device__ __forceinline__ void mul4(Fs2H (&r)[4], const Fs2H (&a)[4], const Fs2H (&b)[4], int used) const {
        Fs2H m1, m2, p;
#pragma unroll
        for (int i = 0; i < SN; i++) { m1.v.l[i] = pick4(role, a[0].v.l[i], a[1].v.l[i], a[2].v.l[i], a[3].v.l[i]); m2.v.l[i] = pick4(role, b[0].v.l[i], b[1].v.l[i], b[2].v.l[i], b[3].v.l[i]); }
        fmul(p, m1, m2);
#pragma unroll
        for (int i = 0; i < SN; i++) {
            r[0].v.l[i] = __builtin_amdgcn_ds_bpermute(src[0], p.v.l[i]); r[1].v.l[i] = __builtin_amdgcn_ds_bpermute(src[1], p.v.l[i]);
            if (used > 2) r[2].v.l[i] = __builtin_amdgcn_ds_bpermute(src[2], p.v.l[i]);
            if (used > 3) r[3].v.l[i] = __builtin_amdgcn_ds_bpermute(src[3], p.v.l[i]);
        }
    }
};
template <class C>
__global__ void __launch_bounds__(256 * C::LPP) k_reduce_top_quad(const uint32_t *__restrict__ l1, const uint8_t *__restrict__ l1_inf, int G, int gshift,
                                                                  uint32_t *__restrict__ win_abi, uint8_t *__restrict__ win_inf,
                                                                  uint32_t *__restrict__ win_s_abi = nullptr, uint8_t *__restrict__ win_s_inf = nullptr) {
    typedef typename C::F F;
    constexpr int LPP = C::LPP, GL = 4 * LPP, PW_ = 4 * SN;          // lanes per group; words of a point held by one lane (4 coordinates or coordinate halves)
    __shared__ uint32_t xs[64 * LPP * PW_];
    __shared__ uint8_t fl[64];
    const int t = (int)threadIdx.x, gi = t / GL, h = t % LPP;
    const QuadLanes<LPP> q4;
    const size_t w = blockIdx.x;
    Xyzz<F> S, A; bool sinf = true, ainf = true;
    fzero(S.x); fzero(S.y); fzero(S.zz); fzero(S.zzz); A = S;
    if (gi < G) {
        const uint32_t *src = l1 + (w * G + gi) * 2 * C::XW;
        if constexpr (LPP == 1) {
            uint32_t *ps = reinterpret_cast<uint32_t *>(&S), *pa = reinterpret_cast<uint32_t *>(&A);
            for (int k = 0; k < PW_; k++) { ps[k] = src[k]; pa[k] = src[C::XW + k]; }
        } else { load_l1_pair(S, src); load_l1_pair(A, src + C::XW); }
        sinf = l1_inf[2 * (w * G + gi)] != 0; ainf = l1_inf[2 * (w * G + gi) + 1] != 0;
    }
    if (!__syncthreads_or((!sinf || !ainf) ? 1 : 0)) {               // an empty (pseudo-)window
        if (t == 0) { win_inf[w] = 1; if (win_s_abi) win_s_inf[w] = 1; }
        return;
    }
    // o = the point of group gi + d (identity past the last group): member r parks coordinate r, everybody reads all four
    auto from_group = [&](Xyzz<F> &o, bool &oinf, const Xyzz<F> &x, bool xinf, int d) __attribute__((always_inline)) {
        __syncthreads();
        { const uint32_t *wx = reinterpret_cast<const uint32_t *>(&x);      // coordinate `role` (picked without a lane-dependent register index: that goes to scratch)
          uint32_t *dst = xs + ((gi * 4 + q4.role) * LPP + h) * SN;
#pragma unroll
          for (int j = 0; j < SN; j++) dst[j] = pick4(q4.role, wx[j], wx[SN + j], wx[2 * SN + j], wx[3 * SN + j]);
          if (t % GL == 0) fl[gi] = xinf; }
        __syncthreads();
        const int sg = gi + d;
        oinf = true;
        if (sg < 64) {
            uint32_t *ov = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t *sv = xs + ((sg * 4 + c) * LPP + h) * SN;
#pragma unroll
                for (int j = 0; j < SN; j++) ov[c * SN + j] = sv[j];
            }
            oinf = fl[sg] != 0;
        } else o = x;
    };
    int Gp = 1; while (Gp < G) Gp <<= 1;                              // groups G .. 63 are identities: the scan and the tree need log2(Gp) levels only
#pragma unroll 1
    for (int d = 1; d < Gp; d <<= 1) {                               // suffix scan: S_g <- sum_{j >= g} S_j
        Xyzz<F> o; bool oinf; from_group(o, oinf, S, sinf, d);
        xyzz_add_rounds(S, sinf, o, oinf, q4);
    }
    Xyzz<F> y = S; bool yinf = (gi == 0) ? true : sinf;              // sum_g g S_g = sum_{g >= 1} suffix_g, times 2^gshift
#pragma unroll 1
    for (int k = 0; k < gshift; k++) { if (!yinf) { Xyzz<F> d2; xyzz_dbl_rounds(d2, y, q4); y = d2; } }
    xyzz_add_rounds(A, ainf, y, yinf, q4);
#pragma unroll 1
    for (int d = Gp >> 1; d >= 1; d >>= 1) {
        Xyzz<F> o; bool oinf; from_group(o, oinf, A, ainf, d);
        xyzz_add_rounds(A, ainf, o, oinf, q4);
    }
    if (gi == 0) {                                                    // member r converts coordinate r (G2: each lane its half)
        const int r = q4.role;
        if (t == 0) { win_inf[w] = ainf; if (win_s_abi) win_s_inf[w] = sinf; }
        const Fs *fa = reinterpret_cast<const Fs *>(&A), *fsv = reinterpret_cast<const Fs *>(&S);
        constexpr int WS = 4 * 12 * LPP;                             // ABI words per window sum
        Fs ma, ms;
#pragma unroll
        for (int k = 0; k < SN; k++) { ma.l[k] = pick4(r, fa[0].l[k], fa[1].l[k], fa[2].l[k], fa[3].l[k]); ms.l[k] = pick4(r, fsv[0].l[k], fsv[1].l[k], fsv[2].l[k], fsv[3].l[k]); }
        if (!ainf) fs_to_abi(win_abi + w * WS + 12 * (LPP * r + h), ma);
        if (win_s_abi && !sinf) fs_to_abi(win_s_abi + w * WS + 12 * (LPP * r + h), ms);
    }
}

// K6 for G2 on lane pairs: lanes 2t and 2t + 1 fold the partials of the bucket whose tail slot is t (same logic as k_fixup)
template <class PAIR>
__global__ void __launch_bounds__(256) k_fixup_pair(uint32_t NB, uint32_t *__restrict__ bucket, uint8_t *__restrict__ bucket_inf,
                                                    const uint32_t *__restrict__ head, const uint32_t *__restrict__ tail, const uint32_t *__restrict__ head_b,
                                                    const uint32_t *__restrict__ tail_b, const uint8_t *__restrict__ part_inf, size_t T,
                                                    const uint32_t *__restrict__ off, uint32_t heavy_thr, const uint32_t *__restrict__ dyn) {
    const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
    if (dyn) { T = dyn[DYN_T]; heavy_thr = dyn[DYN_HEAVY]; }
    if (t >= T) return;
    const uint32_t b = tail_b[t];
    if (b == 0xffffffffu) return;
    if (off[b + 1] - off[b] >= heavy_thr) return;   // folded by k_fixup_heavy
    Xyzz<Fs2H> acc; load_soa_pair(acc, tail, t);
    bool inf = part_inf[2 * t + 1] != 0;
    for (size_t k = t + 1; k < T && head_b[k] == b; k++) {
        Xyzz<Fs2H> o; load_soa_pair(o, head, k);
        xyzz_add(acc, inf, o, part_inf[2 * k] != 0);
    }
    store_soa<G2P>(bucket, NB, b, acc);
    if ((threadIdx.x & 1u) == 0) bucket_inf[b] = inf;
}

}  // namespace msm
