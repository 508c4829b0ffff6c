// crypto_amd/csrc/ntt_kernels.hip.h — NTT over Fr (radix-4 passes through LDS) and the R1CS -> QAP witness map for gfx950.
//
// Device side of LibsnarkReduction::witness_map_from_matrices (/root/reference/legogroth16/src/r1cs_to_qap.rs:150-210,
// ark-poly Radix2EvaluationDomain semantics):
//     a_i = <A_i, z>, b_i = <B_i, z>, c_i = <C_i, z>   (i < m),   a_{m+j} = z_j (j < num_inputs)
//     a, b, c <- iFFT_D ; coset FFT over g H (g = 7) ; ab_i = (a_i b_i - c_i) / (g^D - 1) ; h <- coset iFFT
// SURVEY 8f-1 expected this to be the one HBM-bound piece of the prover; measured, it is bound by instruction issue (see k_ntt_r4).  Layout: limb-major SoA, word l of element i at buf[l * D + i], so
// every butterfly access is a coalesced row.  Bit reversal is never materialised: inverse transforms run
// decimation-in-frequency (natural in, bit-reversed out), the coset shift g^k / D is applied at position p with k = bitrev(p),
// forward transforms run decimation-in-time (bit-reversed in, natural out); the last pass un-reverses while it converts to the
// canonical 4x64-bit scalars the MSM consumes.
// Kernels: k_ntt_stage (one pass per stage: domains below 2^10), k_ntt_fused (staged passes: only for arrays beyond 4 GB, whose offsets do not fit
// the buffer addressing of the main kernel), k_ntt_r4 (everything else).
#pragma once
#include <hip/hip_runtime.h>
#include "fr29.hip.h"

namespace ntt {
using namespace fr29;

__device__ __forceinline__ void ld(Fr &r, const uint32_t *__restrict__ buf, size_t D, size_t i) {
#pragma unroll
    for (int l = 0; l < NL; l++) r.l[l] = buf[(size_t)l * D + i];
}
__device__ __forceinline__ void st(uint32_t *__restrict__ buf, size_t D, size_t i, const Fr &a) {
#pragma unroll
    for (int l = 0; l < NL; l++) buf[(size_t)l * D + i] = a.l[l];
}
__device__ __forceinline__ uint32_t bitrev(uint32_t x, int logn) { return __brev(x) >> (32 - logn); }

// words (8 x u32 per element, canonical or Montgomery) -> internal SoA; elements [n, D) are zeroed
__global__ void __launch_bounds__(256) k_fr_load(const uint32_t *__restrict__ words, size_t n, int mont, uint32_t *__restrict__ out, size_t D) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    Fr x; fr_zero(x);
    if (i < n) { uint32_t w[8]; for (int k = 0; k < 8; k++) w[k] = words[i * 8 + k]; fr_from_words(x, w, mont != 0); }
    st(out, D, i, x);
}
// Fr::into_bigint for a whole scalar vector, in place: x * 2^256 mod r  ->  x   (what ark-ec msm_unchecked does on rayon before the MSM)
__global__ void __launch_bounds__(256) k_fr_mont_to_canonical(uint32_t *__restrict__ words, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[8];
    const uint4 *p = reinterpret_cast<const uint4 *>(words + i * 8);
    uint4 a = p[0], b = p[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    Fr x; fr_from_words(x, w, true); fr_to_words(w, x, false);
    uint4 *q = reinterpret_cast<uint4 *>(words + i * 8);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]); q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// the inverse, in place: x -> x * 2^256 mod r (`Fr::from_bigint`): the witness map's h handed back as the &[Fr] the reference's QAP::witness_map returns
__global__ void __launch_bounds__(256) k_fr_canonical_to_mont(uint32_t *__restrict__ words, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[8];
    const uint4 *p = reinterpret_cast<const uint4 *>(words + i * 8);
    uint4 a = p[0], b = p[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    Fr x; fr_from_words(x, w, false); fr_to_words(w, x, true);
    uint4 *q = reinterpret_cast<uint4 *>(words + i * 8);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]); q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// pw[k] = base^k * scale for k < count (square-and-multiply per lane; built once per domain size and cached)
__global__ void __launch_bounds__(256) k_fr_powers(const uint32_t *__restrict__ base_words, const uint32_t *__restrict__ scale_words, size_t count, uint32_t *__restrict__ out) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    uint32_t bw[8], sw[8];
    for (int i = 0; i < 8; i++) { bw[i] = base_words[i]; sw[i] = scale_words[i]; }
    Fr b, acc; fr_from_words(b, bw, false); fr_from_words(acc, sw, false);
    for (size_t e = k; e; e >>= 1) { if (e & 1) fr_mul(acc, acc, b); fr_mul(b, b, b); }
    st(out, count, k, acc);
}
// Per-stage twiddle tables.  A stage whose twiddle exponents are j << sigma reads T_sigma[j] = w^(j << sigma), H >> sigma entries stored
// contiguously (limb-major, stride H >> sigma) behind the full table T_0: consecutive butterflies read consecutive words.  Indexing T_0
// with the stride 2^sigma made every lane of a wave touch its own cache line (0.6 of the 2.4 ms of the seven transforms at D = 2^20).
// Word offset of T_sigma inside the buffer: NL * (2H - 2 (H >> sigma)); the whole buffer holds < 2H elements.
__host__ __device__ inline size_t tw_stage_offset(size_t H, int sigma) { return (size_t)NL * (2 * H - 2 * (H >> sigma)); }
__global__ void __launch_bounds__(256) k_tw_compact(uint32_t *__restrict__ tw, size_t H, int logh) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // t in [0, H - 1): position inside T_1 .. T_logh
    if (t + 1 >= H) return;
    int sigma = 1; size_t base = 0;
    while (t - base >= (H >> sigma)) { base += H >> sigma; sigma++; }
    const size_t j = t - base, hs = H >> sigma;
    uint32_t *dst = tw + tw_stage_offset(H, sigma);
#pragma unroll
    for (int l = 0; l < NL; l++) dst[(size_t)l * hs + j] = tw[(size_t)l * H + (j << sigma)];
    (void)logh;
}
// sparse rows: out[i] = sum_k vals[k] * z[cols[k]] over row i (i < rows); out[rows + j] = z[j] for j < extra (matrix A only).
// z is gathered from the caller's scalar words (8 x u32 = 32 contiguous bytes per variable: one cache line per gather) and converted on
// the fly; gathering from a limb-major copy touched ten cache lines per variable and made this kernel 3x slower than the transforms'
// share of the witness map warranted.
__device__ __forceinline__ void ld_words(Fr &r, const uint32_t *__restrict__ words, size_t i, bool mont) {
    uint32_t w[8];
    const uint4 *p = reinterpret_cast<const uint4 *>(words + i * 8);
    uint4 a = p[0], b = p[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    fr_from_words(r, w, mont);
}
__global__ void __launch_bounds__(256) k_csr_eval(const uint64_t *__restrict__ rowptr, const uint32_t *__restrict__ cols, const uint32_t *__restrict__ vals_soa, size_t nnz,
                                                  const uint32_t *__restrict__ z_words, int z_mont, size_t nvars, size_t rows, size_t extra, uint32_t *__restrict__ out, size_t D) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    Fr acc; fr_zero(acc);
    if (i < rows) {
        for (uint64_t k = rowptr[i]; k < rowptr[i + 1]; k++) {
            Fr c, zz, t; ld(c, vals_soa, nnz, k); ld_words(zz, z_words, cols[k], z_mont != 0);
            fr_mul(t, zz, c); fr_add(acc, acc, t); fr_norm(acc, acc);
        }
        // a long row leaves a sum of (row length) products of < 2 r each: one product with the Montgomery one brings it back under 2 r, so
        // that what the inverse transform accumulates is bounded by the domain size alone (fr_sub's M = 2^34 case)
        if (rowptr[i + 1] - rowptr[i] > 8) { Fr one; fr_one(one); fr_mul(acc, acc, one); }
    } else if (i < rows + extra) ld_words(acc, z_words, i - rows, z_mont != 0);
    (void)nvars;
    st(out, D, i, acc);
}
// one radix-2 stage over the whole array.  dif != 0: (x, y) -> (x + y, (x - y) w^(j << s)), half = D >> (s+1)
//                                          dif == 0: (x, y) -> (x + y w, x - y w),       half = 1 << s, w^(j << (logn-1-s))
__global__ void __launch_bounds__(256) k_ntt_stage(uint32_t *__restrict__ buf, int logn, int s, const uint32_t *__restrict__ tw, int dif) {
    const size_t D = (size_t)1 << logn, H = D >> 1;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= H) return;
    const size_t half = dif ? (D >> (s + 1)) : ((size_t)1 << s);
    const size_t j = t & (half - 1), i0 = ((t - j) << 1) + j, i1 = i0 + half;
    const size_t e = dif ? (j << s) : (j << (logn - 1 - s));
    Fr x, y, w, u, v; ld(x, buf, D, i0); ld(y, buf, D, i1); ld(w, tw, H, e);
    if (dif) {
        fr_add(u, x, y); fr_norm(u, u);
        fr_sub<FR_BIG>(v, x, y); fr_norm(v, v); fr_mul(v, v, w);     // y: an unreduced partial sum (up to 2^32 r)
    } else {
        Fr yw; fr_mul(yw, y, w);
        fr_add(u, x, yw); fr_norm(u, u);
        fr_sub(v, x, yw); fr_norm(v, v);
    }
    st(buf, D, i0, u); st(buf, D, i1, v);
}
// Several consecutive radix-2 stages in one pass through LDS.  A tile is 2048 elements = 2^S values of the S index bits the
// pass resolves ("mid") x 2^(11-S) columns; a column fixes the other index bits: i = hi << (L + S) | mid << L | lo.
//   dif != 0: stages s0 .. s0+S-1, L = logn - s0 - S, half_mid = 2^(S-1-st), twiddle w^(j << s)
//   dif == 0: stages s0 .. s0+S-1, L = s0,            half_mid = 2^st,       twiddle w^(j << (logn-1-s))      j = jm << L | lo
// L is either 0 (a tile is one contiguous run of 2048 elements) or >= log2(columns) (the columns of a tile are consecutive lo:
// every global access is a contiguous run of 2^(11-S) words per limb) — the launcher orders the stage groups accordingly.
// Per pass every element is read and written once (2 x 40 B) instead of once per stage; twiddles come from the L2-resident table.
constexpr int FUSE_TILE_LOG = 11;
constexpr int FUSE_THREADS = 512;     // 2 butterflies per lane per stage: two 80-KB blocks per CU overlap their load / compute / store phases
// pre != nullptr: every element is multiplied by pre[position] while the tile is loaded (the coset shift g^k / D, table in the order of the
// data: fused here it costs one product per element instead of a pass over the array).
__global__ void __launch_bounds__(FUSE_THREADS) k_ntt_fused(uint32_t *__restrict__ buf, int logn, int s0, int S, const uint32_t *__restrict__ tw, int dif, const uint32_t *__restrict__ pre) {
    extern __shared__ uint32_t lds[];                      // [NL][2048]
    constexpr int TILE = 1 << FUSE_TILE_LOG;
    const size_t D = (size_t)1 << logn, H = D >> 1;
    const int L = dif ? (logn - s0 - S) : s0;
    const int cols_log = FUSE_TILE_LOG - S;
    const size_t c0 = (size_t)blockIdx.x << cols_log;      // first column of this tile; column c = hi << L | lo
    const size_t lo_mask = ((size_t)1 << L) - 1;
    auto addr = [&](uint32_t mid, uint32_t cc) -> size_t {
        size_t c = c0 + cc, hi = c >> L, lo = c & lo_mask;
        return (hi << (L + S)) | ((size_t)mid << L) | lo;
    };
    for (uint32_t e = threadIdx.x; e < TILE; e += FUSE_THREADS) {
        uint32_t mid, cc;
        if (L == 0) { cc = e >> S; mid = e & ((1u << S) - 1); } else { mid = e >> cols_log; cc = e & ((1u << cols_log) - 1); }
        size_t a = addr(mid, cc);
        uint32_t slot = (mid << cols_log) | cc;
        if (pre) {
            Fr x, g2; ld(x, buf, D, a); ld(g2, pre, D, a); fr_mul(x, x, g2);
#pragma unroll
            for (int l = 0; l < NL; l++) lds[l * TILE + slot] = x.l[l];
        } else {
#pragma unroll
            for (int l = 0; l < NL; l++) lds[l * TILE + slot] = buf[(size_t)l * D + a];
        }
    }
    __syncthreads();
    // TILE / 2 = 1024 butterflies per stage, two per lane (t and t + 512)
    for (int st = 0; st < S; st++) {
        const int s = s0 + st;
        const uint32_t half_m = dif ? (1u << (S - 1 - st)) : (1u << st);
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
            const uint32_t t = threadIdx.x + rep * FUSE_THREADS;
            const uint32_t cc = t & ((1u << cols_log) - 1), b = t >> cols_log;
            const size_t lo = (c0 + cc) & lo_mask;
            const uint32_t jm = b & (half_m - 1), m0 = ((b - jm) << 1) + jm, m1 = m0 + half_m;
            const size_t j = ((size_t)jm << L) | lo;
            const int sigma = dif ? s : (logn - 1 - s);
            const uint32_t p0 = (m0 << cols_log) | cc, p1 = (m1 << cols_log) | cc;
            Fr x, y, w, u, v;
            ld(w, tw + tw_stage_offset(H, sigma), H >> sigma, j);
#pragma unroll
            for (int l = 0; l < NL; l++) { x.l[l] = lds[l * TILE + p0]; y.l[l] = lds[l * TILE + p1]; }
            if (dif) {
                fr_add(u, x, y); fr_norm(u, u);
                fr_sub<FR_BIG>(v, x, y); fr_norm(v, v); fr_mul(v, v, w);
            } else {
                Fr yw; fr_mul(yw, y, w);
                fr_add(u, x, yw); fr_norm(u, u);
                fr_sub(v, x, yw); fr_norm(v, v);
            }
#pragma unroll
            for (int l = 0; l < NL; l++) { lds[l * TILE + p0] = u.l[l]; lds[l * TILE + p1] = v.l[l]; }
        }
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < TILE; e += FUSE_THREADS) {
        uint32_t mid, cc2;
        if (L == 0) { cc2 = e >> S; mid = e & ((1u << S) - 1); } else { mid = e >> cols_log; cc2 = e & ((1u << cols_log) - 1); }
        size_t a = addr(mid, cc2);
        uint32_t slot = (mid << cols_log) | cc2;
#pragma unroll
        for (int l = 0; l < NL; l++) buf[(size_t)l * D + a] = lds[l * TILE + slot];
    }
}

// ---- passes without separate load / store phases ----------------------------------------------------------------------------------
// k_ntt_fused runs  load tile -> S stages -> store tile  per block, and a D = 2^20 transform is exactly one round of blocks: every block
// of the chip loads, computes and stores at the same time, so a pass lasts load + compute + store (~66 us against ~38 us of issue time).
// k_ntt_r4 below has no staging loops: the FIRST group of stages takes its operands straight from HBM, the LAST one stores straight to
// HBM; the lane -> element mapping follows the direction that is contiguous in HBM (along the columns when L > 0, along `mid` when the
// tile is one contiguous run, L = 0), so these direct accesses are coalesced and the LDS layout never needs a transposition; rows are
// addressed through buffer descriptors; the a, b, c arrays of the witness map go through every pass together (one launch per pass).
struct NttBatch { uint32_t *buf[3]; };
// Rows of a limb-major array through a buffer descriptor: descriptor (4 SGPRs, uniform base) + row offset l * stride in an SGPR + ONE 32-bit
// element offset in a VGPR shared by the ten rows — no per-limb 64-bit address arithmetic on the vector ALU (hipcc otherwise re-associates
// base + l * stride + i into ten v_lshl_add_u64 and ten address register pairs).  Offsets are 32-bit: arrays of up to 4 GB (logn <= 26).
constexpr int PIPE_MAX_LOGN = 26;
constexpr int PIPE_TILE_LOG = 10;      // 1024 elements = 40 KB of LDS, 256 lanes (one radix-4 unit each), four blocks per CU
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_of(const uint32_t *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(base), 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ void ldg(Fr &r, __amdgpu_buffer_rsrc_t rows, uint32_t stride_bytes, uint32_t i) {
    const uint32_t off = i << 2;
#pragma unroll
    for (int l = 0; l < NL; l++) r.l[l] = __builtin_amdgcn_raw_buffer_load_b32(rows, off, l * stride_bytes, 0);
}
__device__ __forceinline__ void stg(__amdgpu_buffer_rsrc_t rows, uint32_t stride_bytes, uint32_t i, const Fr &a) {
    const uint32_t off = i << 2;
#pragma unroll
    for (int l = 0; l < NL; l++) __builtin_amdgcn_raw_buffer_store_b32(a.l[l], rows, off, l * stride_bytes, 0);
}
template <bool DIF> __device__ __forceinline__ void butterfly(Fr &x, Fr &y, const Fr &w) {
    Fr u, v;
    if (DIF) {
        fr_add(u, x, y); fr_norm(u, u);
        fr_sub<FR_BIG>(v, x, y); fr_norm(v, v); fr_mul(v, v, w);     // y: an unreduced partial sum (up to 2^32 r)
    } else {
        Fr yw; fr_mul(yw, y, w);
        fr_add(u, x, yw); fr_norm(u, u);
        fr_sub(v, x, yw); fr_norm(v, v);
    }
    x = u; y = v;
}
// Barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter (s_waitcnt vmcnt(0)): here that would make
// every stage wait for the prefetched operands of the next tile, for the twiddle requested one butterfly ahead and — at the end of a tile —
// for the results just stored to HBM.  Nothing a lane reads from HBM inside this kernel was written by another lane of the same launch.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ---- radix-4 form: two stages per trip through LDS ------------------------------------------------------------------------------
// The pass kernels are bound by instruction issue (every VALU instruction costs a quad-cycle, v_mad_u64_u32 1.2 of them; measured:
// one block per CU already reaches 88 % of the two-block rate, and prefetching the next tile changes nothing), so what counts is the
// instruction count per butterfly: 180 multiply-adds inside ~390 vector + ~40 LDS / memory instructions.  A lane that keeps FOUR elements
// and runs two stages on them halves the LDS traffic, the slot / address arithmetic and the barriers, loads 3 twiddles for 4 butterflies,
// and needs no carry pass between the two stages: limbs are 29 bits in 32-bit words, a sum of two normalised values or a difference
// formed with the 2^30-limbed multiple of r stays below 2^31 and may enter a product or one more addition as it is.
//   DIT pair of stages (st, st+1), distances ha = 2^st, hb = 2 ha:   m00, m01 = m00 + ha, m10 = m00 + hb, m11
//        A: (m00, m01) and (m10, m11) share one twiddle;  B: (m00, m10) twiddle jm, (m01, m11) twiddle jm + ha
//   DIF pair of stages, distances ha = 2^(S-1-st), hb = ha / 2:      m00, m01 = m00 + hb, m10 = m00 + ha, m11
//        A: (m00, m10) twiddle jm, (m01, m11) twiddle jm + hb;  B: (m00, m01) and (m10, m11) share one twiddle
// An odd S starts with one plain radix-2 stage.  The first group of a pass reads its operands from HBM (times pre[] if given), the last
// one stores to HBM; one tile per block.
//   in_mode  = NTT_IN_POINTWISE : the tile is first filled with (a b - c) * zinv from the three arrays (buf[0..2], pre = zinv words): the
//                                 (ab - c) / Z step of the witness map costs no pass of its own
//   out_mode = NTT_OUT_WORDS    : results leave as canonical 4 x 64-bit scalars times post[position], at the bit-reversed position (the h
//                                 coefficients as the MSM reads them): no separate scaling / un-reversing pass
constexpr int NTT_IN_POINTWISE = 2, NTT_OUT_WORDS = 1;
template <bool DIF, int TILE_LOG>
__global__ void __launch_bounds__(1 << (TILE_LOG - 2)) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_ntt_r4(NttBatch B, int logn, int s0, int S, const uint32_t *__restrict__ tw, const uint32_t *__restrict__ pre, int in_mode, int out_mode,
         const uint32_t *__restrict__ post, uint32_t *__restrict__ out_words) {
    extern __shared__ uint32_t lds[];                      // [NL][TILE]
    constexpr int TILE = 1 << TILE_LOG, THREADS = TILE >> 2;
    const size_t D = (size_t)1 << logn, H = D >> 1;
    const int L = DIF ? (logn - s0 - S) : s0;
    const int cols_log = TILE_LOG - S;
    const uint32_t lo_mask = (1u << L) - 1;
    const bool flat = (L == 0);
    const int tpb_log = logn - TILE_LOG;
    const uint32_t tile = blockIdx.x, bufk = in_mode == NTT_IN_POINTWISE ? 0u : (tile >> tpb_log);
    uint32_t *const buf = bufk == 0 ? B.buf[0] : (bufk == 1 ? B.buf[1] : B.buf[2]);
    const uint32_t c0 = (tile & ((1u << tpb_log) - 1)) << cols_log;
    const uint32_t Db = (uint32_t)(D << 2);                // bytes per row
    const __amdgpu_buffer_rsrc_t rows = rows_of(buf);
    const bool staged_in = (in_mode == NTT_IN_POINTWISE), staged_out = (out_mode == NTT_OUT_WORDS);
    auto gaddr = [&](uint32_t mid, uint32_t c_) -> uint32_t {
        const uint32_t c = c0 + c_, hi = c >> L, lo = c & lo_mask;
        return (hi << (L + S)) | (mid << L) | lo;
    };
    auto slot = [&](uint32_t mid, uint32_t c_) -> uint32_t { return flat ? ((c_ << S) | mid) : ((mid << cols_log) | c_); };
    auto twiddle = [&](Fr &w, int st, uint32_t c_, uint32_t jm) {
        const int s = s0 + st, sigma = DIF ? s : (logn - 1 - s);
        const uint32_t j = (jm << L) | ((c0 + c_) & lo_mask);
        ldg(w, rows_of(tw + tw_stage_offset(H, sigma)), (uint32_t)((H >> sigma) << 2), j);
    };
    auto load = [&](Fr &x, bool from_global, uint32_t mid, uint32_t c_) {
        if (from_global) {
            const uint32_t a = gaddr(mid, c_);
            ldg(x, rows, Db, a);
            if (pre && !staged_in) { Fr g; ldg(g, rows_of(pre), Db, a); fr_mul(x, x, g); }
        } else {
            const uint32_t p = slot(mid, c_);
#pragma unroll
            for (int l = 0; l < NL; l++) x.l[l] = lds[l * TILE + p];
        }
    };
    auto store = [&](bool to_global, uint32_t mid, uint32_t c_, const Fr &x) {
        if (to_global) stg(rows, Db, gaddr(mid, c_), x);
        else {
            const uint32_t p = slot(mid, c_);
#pragma unroll
            for (int l = 0; l < NL; l++) lds[l * TILE + p] = x.l[l];
        }
    };
    // element e of the tile <-> (mid, column), in the order that is contiguous in HBM
    auto elem = [&](uint32_t e, uint32_t &mid, uint32_t &c_) {
        if (flat) { mid = e & ((1u << S) - 1); c_ = e >> S; } else { c_ = e & ((1u << cols_log) - 1); mid = e >> cols_log; }
    };
    if (staged_in) {
        uint32_t zw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) zw[k] = pre[k];
        Fr zi; fr_from_words(zi, zw, false);
        const __amdgpu_buffer_rsrc_t rb = rows_of(B.buf[1]), rc = rows_of(B.buf[2]);
        for (uint32_t e = threadIdx.x; e < TILE; e += THREADS) {
            uint32_t mid, c_; elem(e, mid, c_);
            const uint32_t a = gaddr(mid, c_);
            Fr x, y, z, t; ldg(x, rows, Db, a); ldg(y, rb, Db, a); ldg(z, rc, Db, a);
            fr_mul(t, x, y); fr_sub<FR_BIG>(t, t, z); fr_norm(t, t); fr_mul(t, t, zi);      // z is an un-reduced transform output
            const uint32_t p = slot(mid, c_);
#pragma unroll
            for (int l = 0; l < NL; l++) lds[l * TILE + p] = t.l[l];
        }
        lds_barrier();
    }
    int st = 0;
    if (S & 1) {                                           // one radix-2 stage, two butterflies per lane
        const bool to_global = (S == 1) && !staged_out;
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
            const uint32_t t = threadIdx.x + rep * THREADS;
            uint32_t c_, b;
            if (flat) { b = t & ((1u << (S - 1)) - 1); c_ = t >> (S - 1); } else { c_ = t & ((1u << cols_log) - 1); b = t >> cols_log; }
            const uint32_t half_m = DIF ? (1u << (S - 1)) : 1u;
            const uint32_t jm = b & (half_m - 1), m0 = ((b - jm) << 1) + jm, m1 = m0 + half_m;
            Fr w, x, y; twiddle(w, 0, c_, jm);
            load(x, !staged_in, m0, c_); load(y, !staged_in, m1, c_);
            butterfly<DIF>(x, y, w);
            store(to_global, m0, c_, x); store(to_global, m1, c_, y);
        }
        st = 1;
        if (st < S || staged_out) lds_barrier();
    }
    // radix-4 groups: unit q of column c_
    uint32_t c_, q;
    if (S < 2) { q = 0; c_ = 0; } else if (flat) { q = threadIdx.x & ((1u << (S - 2)) - 1); c_ = threadIdx.x >> (S - 2); } else { c_ = threadIdx.x & ((1u << cols_log) - 1); q = threadIdx.x >> cols_log; }
    for (; st + 1 < S; st += 2) {
        const bool from_global = (st == 0) && !staged_in, to_global = (st + 2 == S) && !staged_out;
        Fr x00, x01, x10, x11, t;
        if (DIF) {
            const int pos = S - 2 - st;                    // bit position of hb
            const uint32_t hb = 1u << pos, ha = hb << 1;
            const uint32_t jm = q & (hb - 1), m00 = ((q >> pos) << (pos + 2)) | jm;
            Fr wa0, wa1, wb;
            twiddle(wa0, st, c_, jm); twiddle(wa1, st, c_, jm + hb); twiddle(wb, st + 1, c_, jm);
            load(x00, from_global, m00, c_); load(x10, from_global, m00 + ha, c_);
            load(x01, from_global, m00 + hb, c_); load(x11, from_global, m00 + ha + hb, c_);
            // A
            fr_sub<FR_BIG>(t, x00, x10); fr_add(x00, x00, x10); fr_norm(t, t); fr_mul(x10, t, wa0);
            fr_sub<FR_BIG>(t, x01, x11); fr_add(x01, x01, x11); fr_norm(t, t); fr_mul(x11, t, wa1);
            // B (x00, x01 carry limbs < 2^30 + 16: dominated by the subtraction constant, and their sum fits a word)
            fr_sub<FR_BIG>(t, x00, x01); fr_add(x00, x00, x01); fr_norm(x00, x00); fr_norm(t, t); fr_mul(x01, t, wb);
            fr_sub<FR_BIG>(t, x10, x11); fr_add(x10, x10, x11); fr_norm(x10, x10); fr_norm(t, t); fr_mul(x11, t, wb);
            store(to_global, m00, c_, x00); store(to_global, m00 + hb, c_, x01);
            store(to_global, m00 + ha, c_, x10); store(to_global, m00 + ha + hb, c_, x11);
        } else {
            const uint32_t ha = 1u << st, hb = ha << 1;
            const uint32_t jm = q & (ha - 1), m00 = ((q >> st) << (st + 2)) | jm;
            Fr wa, wb0, wb1;
            twiddle(wa, st, c_, jm); twiddle(wb0, st + 1, c_, jm); twiddle(wb1, st + 1, c_, jm + ha);
            load(x00, from_global, m00, c_); load(x01, from_global, m00 + ha, c_);
            load(x10, from_global, m00 + hb, c_); load(x11, from_global, m00 + ha + hb, c_);
            // A: no carry pass; sums < 2^30 + 16, differences < 2^31 per limb
            fr_mul(t, x01, wa); fr_sub<512, 30>(x01, x00, t); fr_add(x00, x00, t);
            fr_mul(t, x11, wa); fr_sub<512, 30>(x11, x10, t); fr_add(x10, x10, t);
            // B
            fr_mul(t, x10, wb0); fr_sub<512, 30>(x10, x00, t); fr_add(x00, x00, t); fr_norm(x00, x00); fr_norm(x10, x10);
            fr_mul(t, x11, wb1); fr_sub<512, 30>(x11, x01, t); fr_add(x01, x01, t); fr_norm(x01, x01); fr_norm(x11, x11);
            store(to_global, m00, c_, x00); store(to_global, m00 + ha, c_, x01);
            store(to_global, m00 + hb, c_, x10); store(to_global, m00 + ha + hb, c_, x11);
        }
        if (!to_global) lds_barrier();
    }
    if (staged_out) {
        const __amdgpu_buffer_rsrc_t rp = rows_of(post);
        for (uint32_t e = threadIdx.x; e < TILE; e += THREADS) {
            uint32_t mid, c_; elem(e, mid, c_);
            const uint32_t a = gaddr(mid, c_), p = slot(mid, c_);
            Fr x, g;
#pragma unroll
            for (int l = 0; l < NL; l++) x.l[l] = lds[l * TILE + p];
            ldg(g, rp, Db, a);
            fr_mul(x, x, g);
            uint32_t w[8]; fr_to_words(w, x, false);
            uint4 *q = reinterpret_cast<uint4 *>(out_words + (size_t)bitrev(a, logn) * 8);
            q[0] = make_uint4(w[0], w[1], w[2], w[3]); q[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
}

// element at position p (bit-reversed order, coefficient k = bitrev(p)) *= pw[k]; optionally written out un-reversed as
// canonical words (h coefficients for the MSM)
// pw_in_data_order != 0: pw[p] already holds the factor of position p (a bit-reversed copy of the table: coalesced reads; indexing the natural
// table with bitrev(p) made every lane of a wave touch ten cache lines of its own: 97 us per call at D = 2^20)
__global__ void __launch_bounds__(256) k_coset_scale(uint32_t *__restrict__ buf, int logn, const uint32_t *__restrict__ pw, uint32_t *__restrict__ out_words, int pw_in_data_order) {
    const size_t D = (size_t)1 << logn;
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D) return;
    size_t k = bitrev((uint32_t)p, logn);
    Fr x, g; ld(x, buf, D, p); ld(g, pw, D, pw_in_data_order ? p : k);
    fr_mul(x, x, g);
    if (out_words) { uint32_t w[8]; fr_to_words(w, x, false); for (int i = 0; i < 8; i++) out_words[k * 8 + i] = w[i]; }
    else st(buf, D, p, x);
}
// dst[p] = src[bitrev(p)] (limb-major tables of D elements): the coset-shift tables in the order of bit-reversed data
__global__ void __launch_bounds__(256) k_bitrev_table(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int logn) {
    const size_t D = (size_t)1 << logn;
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D) return;
    size_t k = bitrev((uint32_t)p, logn);
#pragma unroll
    for (int l = 0; l < NL; l++) dst[(size_t)l * D + p] = src[(size_t)l * D + k];
}
// a_i <- (a_i b_i - c_i) * zinv
__global__ void __launch_bounds__(256) k_pointwise(uint32_t *__restrict__ a, const uint32_t *__restrict__ b, const uint32_t *__restrict__ c, size_t D, const uint32_t *__restrict__ zinv_words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    uint32_t zw[8]; for (int k = 0; k < 8; k++) zw[k] = zinv_words[k];
    Fr x, y, z, zi, t; ld(x, a, D, i); ld(y, b, D, i); ld(z, c, D, i); fr_from_words(zi, zw, false);
    fr_mul(t, x, y); fr_sub<FR_BIG>(t, t, z); fr_norm(t, t); fr_mul(t, t, zi);   // z is an un-reduced transform output
    st(a, D, i, t);
}

}  // namespace ntt
