// crypto_amd/csrc/small_kernels.hip.h — variable-base MSM for 1 <= n <= 2^13 terms in two launches (after the digit codes).
//
// Replaces, for the small calls, what the reference reaches as `G::msm_bigint` / `msm_unchecked` at the halving MSMs of the aggregation
// (legogroth16/src/aggregation/utils.rs:51-81), utils/src/randomized_mult_checker.rs:93-101 and most of the 166 call sites above the
// 512-term cut-off.  The bucket pipeline (msm_kernels.hip.h) is ~15 launches whose kernels each last as long as ONE
// lane's chain of group additions — 16 sequential mixed additions in the accumulation, ~32 general additions in the bucket reduction —
// so a call costs 0.7 - 0.9 ms from 2^4 to 2^13 terms whatever n.  With this few terms the chip is empty: work is free, depth is not.
//
//   no buckets, no sort: window w of term i contributes d_iw P_i with a signed 4-bit digit |d| <= 8 (64 windows), i.e. ONE entry of a
//   per-call table of the eight multiples 1 P_i .. 8 P_i (k_small_table), negated when the digit is; the window sum
//   S_w = sum_i (+-) T[i][|d_iw| - 1] is a plain TREE over i (k_small_tree): a few leaves per group, 64 groups per block folded through
//   LDS (6 levels), the <= 4 blocks of a window folded by whichever of them finishes last (2 more levels).  Every point has FOUR members
//   (lanes; lane pairs for G2: QuadLanes of msm_kernels.hip.h), each multiplies one role-selected operand pair per round, so an addition
//   is four products deep (ec29.hip.h xyzz_add_rounds) instead of fourteen.  Depth: 1 doubling + 6 additions (table) + 1 + 6 + <= 6
//   additions (tree) ~ 20 x 3.5 us; work n x 64 additions.  The host's Horner fold over the 64 window sums is the one the bucket
//   pipeline ends with (host_fold).
//
// Any digit set, any order of additions gives the same group element; the ABI returns the normalised representative, so the result is bit
// for bit the bucket pipeline's (tests/test_gpu_small_msm.py compares both and the oracle).
#pragma once
#include "msm_launch.hip.h"
#include "digit_codes.hip.h"

namespace msm {

constexpr int SMALL_C = SMALL_MSM_C;               // window width: signed digits in [-7, 8]
constexpr int SMALL_W = SMALL_MSM_W;               // 64 windows (the top one holds three bits and never carries out)
constexpr int SMALL_E = SMALL_MSM_E;               // table entries per base: 1 P .. 8 P
constexpr int SMALL_LEAVES = SMALL_MSM_LEAVES;     // terms per block of k_small_tree (two per group)
static_assert(SMALL_W == 255 / SMALL_C + 1 && SMALL_E == 1 << (SMALL_C - 1) && SMALL_MSM_MAX_N == (size_t)64 * SMALL_LEAVES, "<= 64 blocks per window, folded by one block");

__device__ __forceinline__ void neg_in_place(Fs &a) { fs_neg(a, a); }           // signed digits: a negation is thirteen v_sub
__device__ __forceinline__ void neg_in_place(Fs2H &a) { fs_neg(a.v, a.v); }

// T[i][e] = (e + 1) P_i for e < 8, and the term's digit codes (window-major, k_digit_codes' format): group g = base i, its four members hold identical copies; member 0 stores.
// An identity base (flag word of its record) gets identity entries (tab_inf), and its digit codes are all "zero" anyway (k_digit_codes).
template <class A>
__global__ void __launch_bounds__(256 * A::LPP) k_small_table(const uint32_t *__restrict__ bases, size_t n, uint32_t *__restrict__ tab, uint8_t *__restrict__ tab_inf,
                                                              const uint32_t *__restrict__ scalars, size_t n_pad, uint16_t *__restrict__ codes, uint32_t *__restrict__ bad) {
    typedef typename A::F F;
    constexpr int LPP = A::LPP, GL = 4 * LPP;
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / GL;
    if (i >= n) return;
    const QuadLanes<LPP> q4;
    const uint32_t *rec = bases + i * A::AFF_STRIDE;
    const bool inf = rec[A::FLAGW] != 0;
    const bool writer = ((threadIdx.x % GL) / LPP) == 0;          // member 0 (G2: both lanes of its pair, each its half)
    // the term's 64 digit codes, by the group's first lane (k_digit_codes' loop: a launch of its own costs more than these ~800 instructions)
    if (threadIdx.x % GL == 0) digit_codes_one<uint16_t>(scalars, i, n, inf, n_pad, SMALL_C, SMALL_W, codes, bad);
    if (inf) {
        if (threadIdx.x % GL == 0) for (int e = 0; e < SMALL_E; e++) tab_inf[i * SMALL_E + e] = 1;
        return;
    }
    Aff<F> p; load_aff<A>(p, rec);
    Xyzz<F> one_p, cur; one_p.x = p.x; one_p.y = p.y; fset_one(one_p.zz); fset_one(one_p.zzz);
    if (writer) store_soa<A>(tab, 0, i * SMALL_E, one_p);
    xyzz_dbl_rounds(cur, one_p, q4);                             // 2 P (both curves have odd order: no point with y = 0)
    bool cinf = false;
    if (writer) store_soa<A>(tab, 0, i * SMALL_E + 1, cur);
#pragma unroll 1
    for (int e = 2; e < SMALL_E; e++) {                           // 3 P .. 8 P: complete additions (k P = -P only for points of tiny order: then the entry is the identity)
        xyzz_add_rounds(cur, cinf, one_p, false, q4);
        if (writer) store_soa<A>(tab, 0, i * SMALL_E + e, cur);
        if (threadIdx.x % GL == 0) tab_inf[i * SMALL_E + e] = cinf;
    }
    if (threadIdx.x % GL == 0) { tab_inf[i * SMALL_E] = 0; tab_inf[i * SMALL_E + 1] = 0; }
}

// block (j, w): S = sum over terms 64 g j .. 64 g (j + 1) - 1 of window w's leaves (g = per_group = small_per_group(n): two, or as many as keep
// the blocks of a window at four); the last block of a window to finish folds the window's
// partials and writes the window sum in the form host_fold reads (k_reduce_top_quad's).  count[w] must be zero at launch and is zero again
// at the end.
template <class A>
__global__ void __launch_bounds__(256 * A::LPP) k_small_tree(const uint32_t *__restrict__ tab, const uint8_t *__restrict__ tab_inf, const uint16_t *__restrict__ codes, size_t n, size_t n_pad,
                                                             uint32_t *__restrict__ partial, uint8_t *__restrict__ partial_inf, uint32_t *__restrict__ count,
                                                             uint32_t *__restrict__ win_abi, uint8_t *__restrict__ win_inf, int per_group) {
    typedef typename A::F F;
    constexpr int LPP = A::LPP, GL = 4 * LPP, PW_ = 4 * SN;
    __shared__ uint32_t xs[64 * LPP * PW_];
    __shared__ uint8_t fl[64];
    __shared__ uint32_t last_flag;
    const int t = (int)threadIdx.x, gi = t / GL, h = t % LPP;
    const QuadLanes<LPP> q4;
    const unsigned j = blockIdx.x, w = blockIdx.y, nblk = gridDim.x;
    auto zero = [](Xyzz<F> &p) __attribute__((always_inline)) { fzero(p.x); fzero(p.y); fzero(p.zz); fzero(p.zzz); };
    auto leaf = [&](Xyzz<F> &p, bool &pinf, size_t i) __attribute__((always_inline)) {
        pinf = true; zero(p);
        if (i >= n) return;
        const uint32_t code = codes[(size_t)w * n_pad + i];
        if (code == 0xffffu) return;
        const size_t at = i * SMALL_E + (code & 0x7fffu);
        if (tab_inf[at]) return;
        load_soa<A>(p, tab, 0, at);
        pinf = false;
        if (code >> 15) neg_in_place(p.y);
    };
    // o = the point of group gi + d: member r parks coordinate r, everybody reads all four (k_reduce_top_quad's exchange)
    auto from_group = [&](Xyzz<F> &o, bool &oinf, const Xyzz<F> &x, bool xinf, int d) __attribute__((always_inline)) {
        __syncthreads();
        { const uint32_t *wx = reinterpret_cast<const uint32_t *>(&x);      // coordinate `role`, picked with selects (a register array indexed by a lane-dependent value goes to scratch)
          const int r = q4.role;
          uint32_t *dst = xs + ((gi * 4 + r) * LPP + h) * SN;
#pragma unroll
          for (int k = 0; k < SN; k++) dst[k] = pick4(r, wx[k], wx[SN + k], wx[2 * SN + k], wx[3 * SN + k]);
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
                for (int k = 0; k < SN; k++) ov[c * SN + k] = sv[k];
            }
            oinf = fl[sg] != 0;
        } else o = x;
    };
    auto tree = [&](Xyzz<F> &a, bool &ainf, int groups) __attribute__((always_inline)) {
        int gp = 1; while (gp < groups) gp <<= 1;
#pragma unroll 1
        for (int d = gp >> 1; d >= 1; d >>= 1) {
            Xyzz<F> o; bool oinf; from_group(o, oinf, a, ainf, d);
            xyzz_add_rounds(a, ainf, o, oinf, q4);
        }
    };
    auto write_window = [&](const Xyzz<F> &a, bool ainf) __attribute__((always_inline)) {      // group 0: member r converts coordinate r (G2: each lane its half)
        if (gi != 0) return;
        const int r = q4.role;
        if (t == 0) win_inf[w] = ainf;
        const Fs *fa = reinterpret_cast<const Fs *>(&a);
        constexpr int WS = 4 * 12 * LPP;
        Fs mine;
#pragma unroll
        for (int k = 0; k < SN; k++) mine.l[k] = pick4(r, fa[0].l[k], fa[1].l[k], fa[2].l[k], fa[3].l[k]);
        if (!ainf) fs_to_abi(win_abi + (size_t)w * WS + 12 * (LPP * r + h), mine);
    };
    // (the lambdas above are forced inline: an outlined one takes the accumulator by reference, i.e. through scratch memory — 0.43 instead of 0.1 ms)
    // a group's own leaves first (per_group of them, strided by 64 so that neighbouring groups read neighbouring codes), then the tree
    Xyzz<F> acc, o; bool ainf, oinf;
    const size_t first = (size_t)j * 64 * per_group, i0 = first + gi;
    leaf(acc, ainf, i0);
#pragma unroll 1
    for (int k = 1; k < per_group; k++) { leaf(o, oinf, i0 + (size_t)64 * k); xyzz_add_rounds(acc, ainf, o, oinf, q4); }
    const size_t here = n - first;                                // terms of this block (>= 1)
    tree(acc, ainf, (int)(here < 64 ? here : 64));
    if (nblk == 1) { write_window(acc, ainf); return; }
    if (gi == 0) {
        if (q4.role == 0) store_soa<A>(partial, 0, (size_t)w * nblk + j, acc);
        if (t == 0) partial_inf[(size_t)w * nblk + j] = ainf;
    }
    __threadfence();
    __syncthreads();
    if (t == 0) last_flag = (atomicAdd(&count[w], 1u) == nblk - 1) ? 1u : 0u;
    __syncthreads();
    if (!last_flag) return;
    __threadfence();
    ainf = true; zero(acc);
    if ((unsigned)gi < nblk) {
        ainf = partial_inf[(size_t)w * nblk + gi] != 0;
        if (!ainf) load_soa<A>(acc, partial, 0, (size_t)w * nblk + gi);
    }
    tree(acc, ainf, (int)nblk);
    write_window(acc, ainf);
    if (t == 0) count[w] = 0;
}

// launchers (instantiated by k_g1_small.hip / k_g2_small.hip; the drivers see the declarations in msm_launch.hip.h)
template <class C> void launch_small_table(hipStream_t s, const uint32_t *bases, size_t n, uint32_t *tab, uint8_t *tab_inf, const uint32_t *scalars, size_t n_pad, void *codes, uint32_t *bad) {
    typedef typename C::ACC A;
    hipLaunchKernelGGL((k_small_table<A>), dim3((unsigned)((n + 63) / 64)), dim3(256 * A::LPP), 0, s, bases, n, tab, tab_inf, scalars, n_pad, (uint16_t *)codes, bad);
}
template <class C> void launch_small_tree(hipStream_t s, const uint32_t *tab, const uint8_t *tab_inf, const void *codes, size_t n, size_t n_pad, uint32_t *partial, uint8_t *partial_inf,
                                          uint32_t *count, uint32_t *win_abi, uint8_t *win_inf) {
    typedef typename C::ACC A;
    const int per_group = small_per_group(n);
    const unsigned nblk = (unsigned)((n + 64 * (size_t)per_group - 1) / (64 * (size_t)per_group));
    hipLaunchKernelGGL((k_small_tree<A>), dim3(nblk, SMALL_W), dim3(256 * A::LPP), 0, s, tab, tab_inf, (const uint16_t *)codes, n, n_pad, partial, partial_inf, count, win_abi, win_inf, per_group);
}

}  // namespace msm
