// crypto_amd/csrc/ntt_kernels.cuh — radix-2 NTT over Fr and the R1CS -> QAP witness map for gfx950.
//
// Device side of LibsnarkReduction::witness_map_from_matrices (/root/reference/legogroth16/src/r1cs_to_qap.rs:150-210,
// ark-poly Radix2EvaluationDomain semantics):
//     a_i = <A_i, z>, b_i = <B_i, z>, c_i = <C_i, z>   (i < m),   a_{m+j} = z_j (j < num_inputs)
//     a, b, c <- iFFT_D ; coset FFT over g H (g = 7) ; ab_i = (a_i b_i - c_i) / (g^D - 1) ; h <- coset iFFT
// The only HBM-bound piece of the prover.  Layout: limb-major SoA, word l of element i at buf[l * D + i], so
// every butterfly access is a coalesced row.  Bit reversal is never materialised: inverse transforms run
// decimation-in-frequency (natural in, bit-reversed out), the coset shift g^k / D is applied at position p with k = bitrev(p),
// forward transforms run decimation-in-time (bit-reversed in, natural out); the last pass un-reverses while it converts to the
// canonical 4x64-bit scalars the MSM consumes.
// v1: one global-memory pass per stage (2 x 40 B per element per stage); fusing 8-10 stages per pass through LDS is the next step.
#pragma once
#include <hip/hip_runtime.h>
#include "fr29.cuh"

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

// ---- the pipelined form of the fused pass -------------------------------------------------------------------------------------
// k_ntt_fused runs  load tile -> S stages -> store tile  per block, and a D = 2^20 transform is exactly one round of blocks (512 tiles on
// 256 CUs x 2): every block of the chip loads, computes and stores at the same time, so a pass lasts load + compute + store
// (~66 us against ~38 us of butterfly issue time).  k_ntt_pipe removes the two memory phases:
//   * blocks are persistent and walk the tiles of up to three arrays (the a, b, c rows of the witness map go through every pass together);
//   * the FIRST stage of a tile takes its operands from registers that were loaded from HBM while the previous tile was being computed,
//     the LAST stage stores its results straight to HBM — no staging loop, no barrier around it;
//   * the lane -> butterfly mapping follows the direction that is contiguous in HBM (along the columns when L > 0, along `mid` when the
//     tile is one contiguous run, L = 0), so these direct accesses are coalesced and the LDS layout never needs a transposition;
//   * rows are addressed as (uniform row base in SGPRs) + (one 32-bit element offset per access): no per-limb 64-bit address arithmetic.
struct NttBatch { uint32_t *buf[3]; };
// Rows of a limb-major array through a buffer descriptor: descriptor (4 SGPRs, uniform base) + row offset l * stride in an SGPR + ONE 32-bit
// element offset in a VGPR shared by the ten rows — no per-limb 64-bit address arithmetic on the vector ALU (hipcc otherwise re-associates
// base + l * stride + i into ten v_lshl_add_u64 and ten address register pairs).  Offsets are 32-bit: arrays of up to 4 GB (logn <= 26).
constexpr int PIPE_MAX_LOGN = 26;
constexpr int PIPE_TILE_LOG = 11;      // 2048 elements = 80 KB of LDS, 512 lanes, two blocks per CU
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
template <bool DIF, int TILE_LOG>
__global__ void __launch_bounds__(1 << (TILE_LOG - 2)) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_ntt_pipe(NttBatch B, int nbuf, int logn, int s0, int S, const uint32_t *__restrict__ tw, const uint32_t *__restrict__ pre) {
    extern __shared__ uint32_t lds[];                      // [NL][TILE]
    constexpr int TILE = 1 << TILE_LOG, THREADS = TILE >> 2;
    const size_t D = (size_t)1 << logn, H = D >> 1;
    const int L = DIF ? (logn - s0 - S) : s0;
    const int cols_log = TILE_LOG - S;
    const uint32_t lo_mask = (1u << L) - 1;
    const bool flat = (L == 0);
    const int tpb_log = logn - TILE_LOG;
    const uint32_t total = (uint32_t)nbuf << tpb_log;
    // two butterflies per lane and stage: (column cc, butterfly bb of that column)
    uint32_t cc[2], bb[2];
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        const uint32_t t = threadIdx.x + rep * THREADS;
        if (flat) { bb[rep] = t & ((1u << (S - 1)) - 1); cc[rep] = t >> (S - 1); }
        else { cc[rep] = t & ((1u << cols_log) - 1); bb[rep] = t >> cols_log; }
    }
    auto gaddr = [&](uint32_t c0, uint32_t mid, uint32_t c_) -> uint32_t {
        const uint32_t c = c0 + c_, hi = c >> L, lo = c & lo_mask;
        return (hi << (L + S)) | (mid << L) | lo;
    };
    auto slot = [&](uint32_t mid, uint32_t c_) -> uint32_t { return flat ? ((c_ << S) | mid) : ((mid << cols_log) | c_); };
    auto geom = [&](int st, uint32_t b, uint32_t &m0, uint32_t &m1, uint32_t &jm) {
        const uint32_t half_m = DIF ? (1u << (S - 1 - st)) : (1u << st);
        jm = b & (half_m - 1); m0 = ((b - jm) << 1) + jm; m1 = m0 + half_m;
    };
    auto twiddle = [&](Fr &w, int st, uint32_t c0, uint32_t c_, uint32_t jm) {
        const int s = s0 + st, sigma = DIF ? s : (logn - 1 - s);
        const uint32_t j = (jm << L) | ((c0 + c_) & lo_mask);
        ldg(w, rows_of(tw + tw_stage_offset(H, sigma)), (uint32_t)((H >> sigma) << 2), j);
    };
    auto pick = [&](uint32_t tile) -> uint32_t * { const uint32_t k = tile >> tpb_log; return k == 0 ? B.buf[0] : (k == 1 ? B.buf[1] : B.buf[2]); };

    Fr X0, Y0, X1, Y1;                                     // first-stage operands of the lane's two butterflies
    const uint32_t Db = (uint32_t)(D << 2);                // bytes per row
    auto fetch1 = [&](const uint32_t *buf, uint32_t c0, int rep, Fr &X, Fr &Y) {
        uint32_t m0, m1, jm; geom(0, bb[rep], m0, m1, jm);
        const __amdgpu_buffer_rsrc_t rows = rows_of(buf);
        ldg(X, rows, Db, gaddr(c0, m0, cc[rep])); ldg(Y, rows, Db, gaddr(c0, m1, cc[rep]));
    };
    auto fetch = [&](uint32_t tile) {
        const uint32_t *buf = pick(tile);
        const uint32_t c0 = (tile & ((1u << tpb_log) - 1)) << cols_log;
        fetch1(buf, c0, 0, X0, Y0); fetch1(buf, c0, 1, X1, Y1);
    };
    // Twiddles run one butterfly ahead: the factor of the NEXT butterfly of this lane (other rep, next stage, or the first stage of the
    // next tile) is requested before the current one is computed, so its L2 latency is never waited for.
    Fr wn;
    auto tw_req = [&](int st, uint32_t c0, int rep) { uint32_t m0, m1, jm; geom(st, bb[rep], m0, m1, jm); twiddle(wn, st, c0, cc[rep], jm); };
    auto first = [&](uint32_t *buf, uint32_t c0, int rep, Fr &X, Fr &Y, int nst, uint32_t nc0, int nrep) {
        uint32_t m0, m1, jm; geom(0, bb[rep], m0, m1, jm);
        const Fr w = wn;
        if (nst >= 0) tw_req(nst, nc0, nrep);
        if (pre) {
            const __amdgpu_buffer_rsrc_t prows = rows_of(pre);
            Fr g; ldg(g, prows, Db, gaddr(c0, m0, cc[rep])); fr_mul(X, X, g);
            ldg(g, prows, Db, gaddr(c0, m1, cc[rep])); fr_mul(Y, Y, g);
        }
        butterfly<DIF>(X, Y, w);
        if (S == 1) { const __amdgpu_buffer_rsrc_t rows = rows_of(buf); stg(rows, Db, gaddr(c0, m0, cc[rep]), X); stg(rows, Db, gaddr(c0, m1, cc[rep]), Y); return; }
        const uint32_t p0 = slot(m0, cc[rep]), p1 = slot(m1, cc[rep]);
#pragma unroll
        for (int l = 0; l < NL; l++) { lds[l * TILE + p0] = X.l[l]; lds[l * TILE + p1] = Y.l[l]; }
    };
    auto later = [&](uint32_t *buf, uint32_t c0, int st, bool last, int rep, int nst, uint32_t nc0, int nrep) {
        uint32_t m0, m1, jm; geom(st, bb[rep], m0, m1, jm);
        Fr x, y; const Fr w = wn;
        const uint32_t p0 = slot(m0, cc[rep]), p1 = slot(m1, cc[rep]);
#pragma unroll
        for (int l = 0; l < NL; l++) { x.l[l] = lds[l * TILE + p0]; y.l[l] = lds[l * TILE + p1]; }
        if (nst >= 0) tw_req(nst, nc0, nrep);
        butterfly<DIF>(x, y, w);
        if (last) { const __amdgpu_buffer_rsrc_t rows = rows_of(buf); stg(rows, Db, gaddr(c0, m0, cc[rep]), x); stg(rows, Db, gaddr(c0, m1, cc[rep]), y); }
        else {
#pragma unroll
            for (int l = 0; l < NL; l++) { lds[l * TILE + p0] = x.l[l]; lds[l * TILE + p1] = y.l[l]; }
        }
    };
    auto c0_of = [&](uint32_t tile) -> uint32_t { return (tile & ((1u << tpb_log) - 1)) << cols_log; };
    uint32_t tile = blockIdx.x;
    if (tile < total) { fetch(tile); tw_req(0, c0_of(tile), 0); }
    for (; tile < total; tile += gridDim.x) {
        uint32_t *buf = pick(tile);
        const uint32_t c0 = c0_of(tile), ntile = tile + gridDim.x;
        const bool more = ntile < total;
        const uint32_t nc0 = more ? c0_of(ntile) : 0;
        first(buf, c0, 0, X0, Y0, 0, c0, 1);                       // first stage: operands already in registers
        first(buf, c0, 1, X1, Y1, S > 1 ? 1 : (more ? 0 : -1), S > 1 ? c0 : nc0, 0);
        lds_barrier();
        if (more) fetch(ntile);                                     // lands while the remaining stages run
        for (int st = 1; st < S; st++) {
            const bool last = (st == S - 1);
            later(buf, c0, st, last, 0, st, c0, 1);
            later(buf, c0, st, last, 1, last ? (more ? 0 : -1) : st + 1, last ? nc0 : c0, 0);
            lds_barrier();
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
