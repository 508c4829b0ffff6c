// crypto_amd/csrc/small_kernels.hip.h — variable-base MSM for 1 <= n <= 2^13 terms without buckets: one table launch and one tree launch per call, or
// the tree launch alone over a table kept with a resident bases handle.
//
// Replaces, for the small calls, what the reference reaches as `G::msm_bigint` / `msm_unchecked` at the halving MSMs of the aggregation
// (legogroth16/src/aggregation/utils.rs:51-81), utils/src/randomized_mult_checker.rs:93-101 and most of the 166 call sites above the
// 256-term cut-off.  The bucket pipeline (msm_kernels.hip.h) is ~15 launches whose kernels each last as long as ONE
// lane's chain of group additions — 16 sequential mixed additions in the accumulation, ~32 general additions in the bucket reduction —
// so a call costs 0.7 - 0.9 ms from 2^4 to 2^13 terms whatever n.  With this few terms the chip is empty: work is free, depth is not.
//
//   no buckets, no sort: window w of term i contributes d_iw P_i with a signed 4-bit digit |d| <= 8 (64 windows), i.e. ONE entry of a table of
//   the eight multiples 1 P_i .. 8 P_i (k_small_table), negated when the digit is; the window sum S_w = sum_i (+-) T[i][|d_iw| - 1] is a plain
//   TREE over i (k_small_tree): a few leaves per group, 64 groups per block folded through LDS (6 levels), the blocks of a window folded by
//   whichever of them finishes last.  Every point has FOUR members (lanes; lane pairs for G2: QuadLanes of msm_kernels.hip.h), each multiplies
//   one role-selected operand pair per round, so an addition is four products deep (ec29.hip.h xyzz_add_rounds) instead of fourteen.  The
//   host's Horner fold over the 64 window sums (host_fold: 252 doublings, ~0.1 ms) is the one the bucket pipeline ends with.
//
//   S = 4 (a table kept with a resident handle: k_small_subtable, built once): the table also holds the multiples of 2^64 P_i, 2^128 P_i and
//   2^192 P_i, so window 16 s + v of term i is a leaf (i, s) of super-window v: 16 trees over 4 n leaves, no table launch, and the host's fold
//   shrinks to 60 doublings (~15 us).
//
//   A leaf computes its own digit from the scalar (no digit pass, no code buffer): with digits in [-7, 8] the carry into window w is
//   [(k mod 16^w) > 0x88...8 (w nibbles)] — a comparison of at most eight words — and the digit raw_w + carry, minus 16 above 8.
//   The per-window block counters clean up after themselves (the last block of a window zeroes its counter; the flag for a scalar >= 2^255
//   travels in the counter's high bits), so a call is ONE launch (resident table) or two, plus the copy of the window sums.
//
// Any digit set, any order of additions gives the same group element; the ABI returns the normalised representative, so the result is bit
// for bit the bucket pipeline's (tests/test_gpu_small_msm.py compares both and the oracle).
#pragma once
#include "msm_launch.hip.h"

namespace msm {

constexpr int SMALL_C = SMALL_MSM_C;               // window width: signed digits in [-7, 8]
constexpr int SMALL_W = SMALL_MSM_W;               // 64 windows (the top one holds three bits and never carries out)
constexpr int SMALL_E = SMALL_MSM_E;               // table entries per base and sub-table: 1 P .. 8 P
constexpr int SMALL_LEAVES = SMALL_MSM_LEAVES;
constexpr int SMALL_S = SMALL_MSM_S;               // sub-tables of a resident table: multiples of 2^(64 s) P, s < 4
static_assert(SMALL_W == 255 / SMALL_C + 1 && SMALL_E == 1 << (SMALL_C - 1) && SMALL_W % SMALL_S == 0 && SMALL_W / SMALL_S * SMALL_C == 64, "window layout");

__device__ __forceinline__ void neg_in_place(Fs &a) { fs_neg(a, a); }           // signed digits: a negation is thirteen v_sub
__device__ __forceinline__ void neg_in_place(Fs2H &a) { fs_neg(a.v, a.v); }

// the signed digit of window w (0 .. 63) of the canonical scalar at `sc` (8 words): magnitude 0 .. 8 and sign; *bad |= scalar >= 2^255
__device__ __forceinline__ void small_digit(const uint32_t *__restrict__ sc, int w, uint32_t &mag, bool &neg, uint32_t &bad) {
    const uint4 lo = *reinterpret_cast<const uint4 *>(sc), hi = *reinterpret_cast<const uint4 *>(sc + 4);
    const uint32_t s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w & 0x7fffffffu};
    if (w == SMALL_W - 1) bad |= hi.w >> 31;                      // (one leaf per scalar looks: the one of the top window)
    const int jw = w >> 3, sh = 4 * (w & 7);
    const uint32_t m = (1u << sh) - 1u;                            // the bits of word jw below the window
    uint32_t raw = 0; bool gt = false, decided = false;
#pragma unroll
    for (int j = 7; j >= 0; j--) {                                 // (static indices: the words stay in registers)
        const uint32_t mask = j < jw ? 0xffffffffu : (j == jw ? m : 0u);
        const uint32_t a = s[j] & mask, c8 = 0x88888888u & mask;
        if (j == jw) raw = (s[j] >> sh) & 15u;
        if (!decided && a != c8) { gt = a > c8; decided = true; }
    }
    const uint32_t d = raw + (gt ? 1u : 0u);                       // 0 .. 16
    neg = d > 8u;
    mag = neg ? 16u - d : d;
}

// T[i][e] = (e + 1) P_i for e < 8: group g = base i, its four members hold identical copies; member 0 stores.
// An identity base (flag word of its record) gets identity entries (tab_inf).
template <class A>
__global__ void __launch_bounds__(256 * A::LPP) k_small_table(const uint32_t *__restrict__ bases, size_t n, uint32_t *__restrict__ tab, uint8_t *__restrict__ tab_inf) {
    typedef typename A::F F;
    constexpr int LPP = A::LPP, GL = 4 * LPP;
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / GL;
    if (i >= n) return;
    const QuadLanes<LPP> q4;
    const uint32_t *rec = bases + i * A::AFF_STRIDE;
    const bool inf = rec[A::FLAGW] != 0;
    const bool writer = ((threadIdx.x % GL) / LPP) == 0;          // member 0 (G2: both lanes of its pair, each its half)
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

// The table of a resident handle: T[(i S + s)][e] = (e + 1) 2^(64 s) P_i.  Group g = (base i, sub-table s): 64 s doublings (a chain: the one-time
// cost of the table, ~1.2 ms whatever n), then the eight multiples as above.  The group order of both curves is odd, so no doubling meets the identity.
template <class A>
__global__ void __launch_bounds__(256 * A::LPP) k_small_subtable(const uint32_t *__restrict__ bases, size_t n, uint32_t *__restrict__ tab, uint8_t *__restrict__ tab_inf) {
    typedef typename A::F F;
    constexpr int LPP = A::LPP, GL = 4 * LPP;
    const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / GL;
    if (g >= n * SMALL_S) return;
    const size_t i = g / SMALL_S; const int sub = (int)(g % SMALL_S);
    const QuadLanes<LPP> q4;
    const uint32_t *rec = bases + i * A::AFF_STRIDE;
    const bool writer = ((threadIdx.x % GL) / LPP) == 0;
    const size_t at = g * SMALL_E;
    if (rec[A::FLAGW] != 0) {
        if (threadIdx.x % GL == 0) for (int e = 0; e < SMALL_E; e++) tab_inf[at + e] = 1;
        return;
    }
    Aff<F> p; load_aff<A>(p, rec);
    Xyzz<F> one_p, cur; one_p.x = p.x; one_p.y = p.y; fset_one(one_p.zz); fset_one(one_p.zzz);
#pragma unroll 1
    for (int k = 0; k < 64 * sub; k++) { xyzz_dbl_rounds(cur, one_p, q4); one_p = cur; }
    if (writer) store_soa<A>(tab, 0, at, one_p);
    xyzz_dbl_rounds(cur, one_p, q4);
    bool cinf = false;
    if (writer) store_soa<A>(tab, 0, at + 1, cur);
#pragma unroll 1
    for (int e = 2; e < SMALL_E; e++) {
        xyzz_add_rounds(cur, cinf, one_p, false, q4);
        if (writer) store_soa<A>(tab, 0, at + e, cur);
        if (threadIdx.x % GL == 0) tab_inf[at + e] = cinf;
    }
    if (threadIdx.x % GL == 0) { tab_inf[at] = 0; tab_inf[at + 1] = 0; }
}

// block (j, v): S = sum over the leaves 64 g j .. 64 g (j + 1) - 1 of (super-)window v (g = per_group); leaf l = (sub-table s = l / n, term i = l mod n)
// is +- T[i S + s][|d| - 1] with d the digit of window (64 / S) s + v of scalar i.  The last block of a window to finish folds the window's partials and
// writes the window sum in the form host_fold reads (k_reduce_top_quad's), the window's identity flag and its bad-scalar flag.  count[v] must be zero
// at launch and is zero again at the end (low 16 bits: blocks done; bit 16 up: a block saw a scalar >= 2^255).
template <class A, int S>
__global__ void __launch_bounds__(256 * A::LPP) k_small_tree(const uint32_t *__restrict__ tab, const uint8_t *__restrict__ tab_inf, const uint32_t *__restrict__ scalars, size_t n,
                                                             uint32_t *__restrict__ partial, uint8_t *__restrict__ partial_inf, uint32_t *__restrict__ count,
                                                             uint32_t *__restrict__ win_abi, uint8_t *__restrict__ win_inf, uint8_t *__restrict__ win_bad, int per_group) {
    typedef typename A::F F;
    constexpr int LPP = A::LPP, GL = 4 * LPP, PW_ = 4 * SN, WPS = SMALL_W / S;
    __shared__ uint32_t xs[64 * LPP * PW_];
    __shared__ uint8_t fl[64];
    __shared__ uint32_t last_flag;
    const int t = (int)threadIdx.x, gi = t / GL, h = t % LPP;
    const QuadLanes<LPP> q4;
    const unsigned j = blockIdx.x, v = blockIdx.y, nblk = gridDim.x;
    const size_t L = n * S;
    uint32_t bad = 0;
    auto zero = [](Xyzz<F> &p) __attribute__((always_inline)) { fzero(p.x); fzero(p.y); fzero(p.zz); fzero(p.zzz); };
    auto leaf = [&](Xyzz<F> &p, bool &pinf, size_t l) __attribute__((always_inline)) {
        pinf = true; zero(p);
        if (l >= L) return;
        const unsigned sub = S == 1 ? 0u : (uint32_t)l / (uint32_t)n;
        const size_t i = l - (size_t)sub * n;
        uint32_t mag; bool neg;
        small_digit(scalars + i * 8, (int)(WPS * sub + v), mag, neg, bad);
        if (mag == 0) return;
        const size_t at = (i * S + sub) * SMALL_E + (mag - 1);
        if (tab_inf[at]) return;
        load_soa<A>(p, tab, 0, at);
        pinf = false;
        if (neg) neg_in_place(p.y);
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
    auto write_window = [&](const Xyzz<F> &a, bool ainf, uint32_t any_bad) __attribute__((always_inline)) {      // group 0: member r converts coordinate r (G2: each lane its half)
        if (gi != 0) return;
        const int r = q4.role;
        if (t == 0) { win_inf[v] = ainf; win_bad[v] = any_bad != 0; }
        const Fs *fa = reinterpret_cast<const Fs *>(&a);
        constexpr int WS = 4 * 12 * LPP;
        Fs mine;
#pragma unroll
        for (int k = 0; k < SN; k++) mine.l[k] = pick4(r, fa[0].l[k], fa[1].l[k], fa[2].l[k], fa[3].l[k]);
        if (!ainf) fs_to_abi(win_abi + (size_t)v * WS + 12 * (LPP * r + h), mine);
    };
    // (the lambdas above are forced inline: an outlined one takes the accumulator by reference, i.e. through scratch memory — 0.43 instead of 0.1 ms)
    // a group's own leaves first (per_group of them, strided by 64 so that neighbouring groups read neighbouring scalars), then the tree
    Xyzz<F> acc, o; bool ainf, oinf;
    const size_t first = (size_t)j * 64 * per_group, l0 = first + gi;
    leaf(acc, ainf, l0);
#pragma unroll 1
    for (int k = 1; k < per_group; k++) { leaf(o, oinf, l0 + (size_t)64 * k); xyzz_add_rounds(acc, ainf, o, oinf, q4); }
    const size_t here = L - first;                                // leaves of this block (>= 1)
    tree(acc, ainf, (int)(here < 64 ? here : 64));
    const uint32_t block_bad = (uint32_t)__syncthreads_or((int)bad);
    if (nblk == 1) { write_window(acc, ainf, block_bad); return; }
    if (gi == 0) {
        if (q4.role == 0) store_soa<A>(partial, 0, (size_t)v * nblk + j, acc);
        if (t == 0) partial_inf[(size_t)v * nblk + j] = ainf;
    }
    __threadfence();
    __syncthreads();
    if (t == 0) last_flag = atomicAdd(&count[v], 1u + (block_bad ? 0x10000u : 0u)) + 1u + (block_bad ? 0x10000u : 0u);
    __syncthreads();
    const uint32_t total = last_flag;
    if ((total & 0xffffu) != nblk) return;
    __threadfence();
    ainf = true; zero(acc);
    if ((unsigned)gi < nblk) {
        ainf = partial_inf[(size_t)v * nblk + gi] != 0;
        if (!ainf) load_soa<A>(acc, partial, 0, (size_t)v * nblk + gi);
    }
    tree(acc, ainf, (int)nblk);
    write_window(acc, ainf, total >> 16);
    if (t == 0) count[v] = 0;
}

// launchers (instantiated by k_g1_small.hip / k_g2_small.hip; the drivers see the declarations in msm_launch.hip.h)
template <class C> void launch_small_table(hipStream_t s, const uint32_t *bases, size_t n, uint32_t *tab, uint8_t *tab_inf) {
    typedef typename C::ACC A;
    hipLaunchKernelGGL((k_small_table<A>), dim3((unsigned)((n + 63) / 64)), dim3(256 * A::LPP), 0, s, bases, n, tab, tab_inf);
}
template <class C> void launch_small_subtable(hipStream_t s, const uint32_t *bases, size_t n, uint32_t *tab, uint8_t *tab_inf) {
    typedef typename C::ACC A;
    hipLaunchKernelGGL((k_small_subtable<A>), dim3((unsigned)((n * SMALL_S + 63) / 64)), dim3(256 * A::LPP), 0, s, bases, n, tab, tab_inf);
}
template <class C> void launch_small_tree(hipStream_t s, const uint32_t *tab, const uint8_t *tab_inf, int subtables, const uint32_t *scalars, size_t n, uint32_t *partial, uint8_t *partial_inf,
                                          uint32_t *count, uint32_t *win_abi, uint8_t *win_inf, uint8_t *win_bad) {
    typedef typename C::ACC A;
    const size_t L = n * (size_t)subtables;
    const int per_group = small_per_group(L, subtables);
    const unsigned nblk = (unsigned)((L + 64 * (size_t)per_group - 1) / (64 * (size_t)per_group));
    if (subtables == 1) hipLaunchKernelGGL((k_small_tree<A, 1>), dim3(nblk, SMALL_W), dim3(256 * A::LPP), 0, s, tab, tab_inf, scalars, n, partial, partial_inf, count, win_abi, win_inf, win_bad, per_group);
    else hipLaunchKernelGGL((k_small_tree<A, SMALL_S>), dim3(nblk, SMALL_W / SMALL_S), dim3(256 * A::LPP), 0, s, tab, tab_inf, scalars, n, partial, partial_inf, count, win_abi, win_inf, win_bad, per_group);
}

}  // namespace msm
