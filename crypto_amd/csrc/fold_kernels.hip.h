// crypto_amd/csrc/fold_kernels.hip.h — out_i = A_i + c P_i for ONE scalar c and many points, with the scalar-independent part done ahead of time.
//
// The GIPA folding step of the SnarkPack aggregation (`compress`, legogroth16/src/aggregation/utils.rs:34-49: vec[i] += vec[i + split] * c; Key::compress,
// aggregation/key.rs:160-184) is one shared challenge c against the right halves of five to seven vectors.  k_g1_scale / k_mul_add_g2_gls
// (sort_kernels.hip.h, fixed_kernels.hip.h) run it as a double-and-add chain per point: 128 (G1, GLV) or 64 (G2, GLS) dependent steps, 1.5 ms
// whatever the vector length — and in the protocol that chain sits between the challenge and the next round's pairings, ten times per aggregation.
// But the DOUBLINGS of the chain do not depend on c:
//
//   prepare (k_fold_chain_*):  T_i[k] = 2^k P_i, k < 128 (G1), or 2^k (|x|^j P_i), j < 4, k < 64 (G2: the four GLS bases, fixed_kernels.hip.h) — the same
//                              chain as before, written out; it runs while the round's multi-pairings do, before the challenge exists
//   apply (k_fold_tree):       with c = k1 + k2 lambda (G1) or four base-|x| digits (G2) the product is the SUM of the table entries at the set bits
//                              (phi applied to the k2 ones: one product by beta): a tree over <= 256 leaves per point — a block per point, 64 groups of
//                              four members (small_kernels.hip.h's tree: xyzz_add_rounds, LDS exchange), the addend A_i last: ~9 additions deep
//                              instead of ~190 operations; then k_fold_affine_* (one inversion per point)
//
// The table is the interface between the two field representations: the chains run in the 14 x 29-bit field (two lanes per G1 point, sixteen per G2
// point: ec29_two_lane.hip.h) and store their limbs as they are; the tree runs in the 13 x 30-bit signed field (msm_kernels.hip.h QuadLanes): a leaf's
// members convert one coordinate each (a product by 2^384 / 2^406 in the old field, the packing into twelve words, fs_from_abi) and exchange them.  Same group element as the chain kernels give, and the affine output is its unique representative: bit for bit the
// result of dgpu_g*_mul_add_batch (tests/test_gpu_fold.py compares the two and the oracle).
#pragma once
#include "fixed_kernels.hip.h"

namespace msm {

constexpr int FOLD_E1 = 128, FOLD_E2 = 256;        // table entries per point
constexpr int FOLD_PW1 = 4 * NL, FOLD_PW2 = 8 * NL;  // words per XYZZ entry: the 14 limbs of x, y, zz, zzz as the chain holds them (G2: c0 then c1 of each)
constexpr int FOLD_XW1 = 48, FOLD_XW2 = 96;        // ABI words of a sum on its way from the tree to the affine conversion
constexpr int FOLD_MAX_LEAVES = 256;

// A table entry's coordinate: value * 2^384 mod p as twelve words like the ABI's, but only reduced below 2 p (< 2^384), not to the canonical
// representative — fs_from_abi takes any 384-bit value, and the twelve conditional subtractions of fp_to_abi's fp_canon cost more than the
// product in front of them (a chain step: 11.5 -> 8 us)
FD void fp_to_words_loose(uint32_t w[12], const Fp &a) {
    constexpr uint32_t COUT_[NL] = BLS29_COUT;
    Fp t, cout;
#pragma unroll
    for (int i = 0; i < NL; i++) cout.l[i] = COUT_[i];
    CHK(chk_set_N(cout, 1.0);)
    fp_norm(t, a);
    fp_mul(t, t, cout);                                   // < 2 p: limbs of class N, carried exactly below
    uint32_t c[NL]; uint64_t cy = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) { cy += t.l[i]; c[i] = (uint32_t)cy & LMASK; cy >>= LB; }
    c[NL - 1] = (uint32_t)(cy + t.l[NL - 1]);
    uint32_t o[12];
#pragma unroll
    for (int i = 0; i < 12; i++) o[i] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = i * LB, wi = bit >> 5, sh = bit & 31;
        o[wi] |= c[i] << sh;
        if (sh + LB > 32 && wi + 1 < 12) o[wi + 1] |= c[i] >> (32 - sh);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = o[i];
}

// ---- prepare --------------------------------------------------------------------------------------------------------------------------------
// The chains store their limbs as they are (a step is the doubling and 28 stores); a leaf of the tree converts what it loads.
__device__ __forceinline__ void fold_store(uint32_t *o, const Fp &a) {
#pragma unroll
    for (int k = 0; k < NL; k++) o[k] = a.l[k];
}
// G1: two adjacent lanes per point, lane 0 stores x and zz of every multiple, lane 1 y and zzz
__device__ __forceinline__ void fold_chain_g1(unsigned bid, const uint32_t *__restrict__ p_abi, size_t n, uint32_t *__restrict__ tab, uint8_t *__restrict__ tab_inf) {
    const size_t i = ((size_t)bid * 64 + threadIdx.x) >> 1;
    const bool B = (threadIdx.x & 1u) != 0;
    if (i >= n) return;
    uint32_t any = 0;
    for (int k = 0; k < 24; k++) any |= p_abi[i * 24 + k];
    if (!B) tab_inf[i] = any == 0;
    if (any == 0) return;
    Xyzz<Fp> acc; fp_from_abi(acc.x, p_abi + i * 24); fp_from_abi(acc.y, p_abi + i * 24 + 12); fset_one(acc.zz); fset_one(acc.zzz);
    uint32_t *o = tab + i * (size_t)FOLD_E1 * FOLD_PW1;
#pragma unroll 1
    for (int k = 0; k < FOLD_E1; k++) {
        if (!B) { fold_store(o, acc.x); fold_store(o + 2 * NL, acc.zz); } else { fold_store(o + NL, acc.y); fold_store(o + 3 * NL, acc.zzz); }
        o += FOLD_PW1;
        if (k + 1 < FOLD_E1) { Xyzz<Fp> d; xyzz_dbl_2l(d, acc); acc = d; }
    }
}
// G2: sixteen lanes per point; quad j holds B_j = |x|^j P (the GLS bases of k_mul_add_g2_gls) on its two lane pairs (Share4) and stores entry 64 j + k;
// pair 0 of the quad stores x and zz, pair 1 y and zzz (each lane its half)
__device__ __forceinline__ void fold_chain_g2(unsigned bid, const uint32_t *__restrict__ p_abi, size_t n, uint32_t *__restrict__ tab, uint8_t *__restrict__ tab_inf) {
    typedef Fp2H F;
    constexpr int PW = 48;
    const size_t gid = (size_t)bid * 64 + threadIdx.x, i = gid >> 4;
    const uint32_t h = threadIdx.x & 1u, j = (threadIdx.x >> 2) & 3u;
    const bool second_pair = (threadIdx.x & 2u) != 0;
    if (i >= n) return;
    uint32_t any = 0;
    for (int k = 0; k < PW; k++) any |= p_abi[i * PW + k];
    if ((threadIdx.x & 15u) == 0) tab_inf[i] = any == 0;
    if (any == 0) return;
    Aff<F> P; fp_from_abi(P.x.v, p_abi + i * PW + 12 * h); fp_from_abi(P.y.v, p_abi + i * PW + 12 * (2 + h));
    {   // B_j = (c_j(X) AX_j, c_j(Y) AY_j)   (fixed_kernels.hip.h)
        const bool cj = (j & 1u) != 0 && h != 0;
        Fp z, nx, ny; fp_zero(z);
        fp_sub<4>(nx, z, P.x.v); fp_norm(nx, nx); fp_sub<4>(ny, z, P.y.v); fp_norm(ny, ny);
        sel(P.x.v, cj, nx, P.x.v); sel(P.y.v, cj, ny, P.y.v);
        F ax, ay;
#pragma unroll
        for (int k = 0; k < NL; k++) { ax.v.l[k] = GLS_BASE[j][h][k]; ay.v.l[k] = GLS_BASE[j][2 + h][k]; }
        F bx, by; fmul(bx, P.x, ax); fmul(by, P.y, ay);
        P.x = bx; P.y = by;
    }
    Xyzz<F> acc; acc.x = P.x; acc.y = P.y; fzero(acc.zz); fzero(acc.zzz);
    if (h == 0) { fset_one(acc.zz.v); fset_one(acc.zzz.v); }          // one = (1, 0): the c1 halves stay zero
    uint32_t *o = tab + (i * (size_t)FOLD_E2 + 64 * j) * FOLD_PW2;
#pragma unroll 1
    for (int k = 0; k < 64; k++) {
        if (!second_pair) { fold_store(o + NL * h, acc.x.v); fold_store(o + NL * (4 + h), acc.zz.v); } else { fold_store(o + NL * (2 + h), acc.y.v); fold_store(o + NL * (6 + h), acc.zzz.v); }
        o += FOLD_PW2;
        if (k + 1 < 64) { Xyzz<F> d; xyzz_dbl_shared<F, Share4>(d, acc); acc = d; }
    }
}
// both groups' right halves of a round in ONE launch (two launches on one stream would run one after the other: 1.1 + 0.65 ms; blocks of one launch do not)
template <class DUMMY>
__global__ void __launch_bounds__(64) k_fold_chain(const uint32_t *__restrict__ p1, size_t n1, uint32_t *__restrict__ tab1, uint8_t *__restrict__ inf1, unsigned blocks1,
                                                  const uint32_t *__restrict__ p2, size_t n2, uint32_t *__restrict__ tab2, uint8_t *__restrict__ inf2) {
    if (blockIdx.x < blocks1) fold_chain_g1(blockIdx.x, p1, n1, tab1, inf1);
    else fold_chain_g2(blockIdx.x - blocks1, p2, n2, tab2, inf2);
}

// ---- apply ----------------------------------------------------------------------------------------------------------------------------------
// GPP groups of four members per point, 64 / GPP points per block (GPP = 64: a block per point, the shallowest tree, for the short vectors of the later
// rounds; GPP = 16: a wave (G2: two) per point, four points per block, so that thousands of points do not queue block by block):
// sum of the T leaves[] entries of the point (entries >= phi_from: x times beta first), plus the addend; the sum leaves as XYZZ in ABI words
template <class A, int GPP>
__global__ void __launch_bounds__(256 * A::LPP) k_fold_tree(const uint32_t *__restrict__ tab, const uint8_t *__restrict__ tab_inf, int entries, const uint16_t *__restrict__ leaves, int T,
                                                            int phi_from, const uint32_t *__restrict__ add_abi, size_t n,
                                                            uint32_t *__restrict__ out_xyzz, uint8_t *__restrict__ out_inf) {
    typedef typename A::F F;
    constexpr int LPP = A::LPP, GL = 4 * LPP, PW_ = 4 * SN, EW = 48 * LPP, AW = 24 * LPP;     // EW: words of the sum this kernel writes (ABI form)
    __shared__ uint32_t xs[64 * LPP * PW_];
    __shared__ uint8_t fl[64];
    const int t = (int)threadIdx.x, gi = t / GL, h = t % LPP, g = gi % GPP;
    const QuadLanes<LPP> q4;
    const size_t i_raw = (size_t)blockIdx.x * (64 / GPP) + gi / GPP;
    const bool live = i_raw < n;                                      // (a block's last points may be padding: they follow the barriers and write nothing)
    const size_t i = live ? i_raw : n - 1;
    const bool pinf_all = !live || tab_inf[i] != 0;
    auto coord = [&](F &r, const uint32_t *w) __attribute__((always_inline)) { fs_from_abi(*reinterpret_cast<Fs *>(&r), w + 12 * h); };      // this lane's half (G1: the whole)
    auto zero = [](Xyzz<F> &p) __attribute__((always_inline)) { fzero(p.x); fzero(p.y); fzero(p.zz); fzero(p.zzz); };
    F beta; fzero(beta);
    if constexpr (LPP == 1) {                                         // beta (ec29_two_lane.hip.h xyzz_phi's constant) from the 29-bit form into this field, once per lane
        constexpr uint32_t B_[NL] = BLS29_BETA;
        Fp b29; uint32_t w[12];
#pragma unroll
        for (int k = 0; k < NL; k++) b29.l[k] = B_[k];
        CHK(chk_set_N(b29, 1.0);)
        fp_to_abi(w, b29); fs_from_abi(beta, w);
    }
    // a leaf: member r loads coordinate r (G2: its half) as the chain left it, brings it into this field (one product in each field) and hands it to the
    // other members (the four of a group take the same path: k, T and the point's flag are the group's)
    auto leaf = [&](Xyzz<F> &p, bool &pinf, int k) __attribute__((always_inline)) {
        pinf = true; zero(p);
        if (k >= T || pinf_all) return;
        const int e = (int)leaves[k];
        const bool phi = e >= phi_from;
        const uint32_t *w = tab + (i * (size_t)entries + (size_t)(phi ? e - phi_from : e)) * (4 * LPP * NL) + (q4.role * LPP + h) * NL;
        Fp a29;
#pragma unroll
        for (int c = 0; c < NL; c++) a29.l[c] = w[c];
        CHK(chk_set_N(a29, 64.0);)
        uint32_t w12[12]; fp_to_words_loose(w12, a29);
        Fs mine; fs_from_abi(mine, w12);
        Fs *pc = reinterpret_cast<Fs *>(&p);
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int l = 0; l < SN; l++) pc[c].l[l] = __builtin_amdgcn_ds_bpermute(q4.src[c], mine.l[l]);
        if constexpr (LPP == 1) { if (phi) { F bx; fmul(bx, p.x, beta); p.x = bx; } }
        pinf = false;
    };
    auto from_group = [&](Xyzz<F> &o, bool &oinf, const Xyzz<F> &x, bool xinf, int d) __attribute__((always_inline)) {
        __syncthreads();
        { const uint32_t *wx = reinterpret_cast<const uint32_t *>(&x);
          const int r = q4.role;
          uint32_t *dst = xs + ((gi * 4 + r) * LPP + h) * SN;
#pragma unroll
          for (int k = 0; k < SN; k++) dst[k] = pick4(r, wx[k], wx[SN + k], wx[2 * SN + k], wx[3 * SN + k]);
          if (t % GL == 0) fl[gi] = xinf; }
        __syncthreads();
        const int sg = gi + d;
        oinf = true;
        if (g + d < GPP) {
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
    Xyzz<F> acc, o; bool ainf, oinf;
    leaf(acc, ainf, g);
#pragma unroll 1
    for (int k = g + GPP; k < (T + GPP - 1) / GPP * GPP; k += GPP) { leaf(o, oinf, k); xyzz_add_rounds(acc, ainf, o, oinf, q4); }      // (every group runs the same number of rounds)
    const int groups = T < GPP ? (T < 1 ? 1 : T) : GPP;
    int gp = 1; while (gp < groups) gp <<= 1;
#pragma unroll 1
    for (int d = gp >> 1; d >= 1; d >>= 1) { from_group(o, oinf, acc, ainf, d); xyzz_add_rounds(acc, ainf, o, oinf, q4); }
    if (g != 0 || !live) return;
    if (add_abi) {                                                    // A_i (affine; all-zero words: the identity)
        const uint32_t *w = add_abi + i * AW;
        uint32_t any = 0;
        for (int k = 0; k < AW; k++) any |= w[k];
        if (any != 0) {
            zero(o); coord(o.x, w); coord(o.y, w + 12 * LPP);
            if (h == 0) { fset_one(*reinterpret_cast<Fs *>(&o.zz)); fset_one(*reinterpret_cast<Fs *>(&o.zzz)); }
            xyzz_add_rounds(acc, ainf, o, false, q4);
        }
    }
    if (t % GL == 0) out_inf[i] = ainf;
    const int r = q4.role;
    const Fs *fa = reinterpret_cast<const Fs *>(&acc);
    Fs mine;
#pragma unroll
    for (int k = 0; k < SN; k++) mine.l[k] = pick4(r, fa[0].l[k], fa[1].l[k], fa[2].l[k], fa[3].l[k]);
    if (!ainf) fs_to_abi(out_xyzz + i * EW + 12 * (LPP * r + h), mine);
}

// XYZZ (ABI words) -> affine (ABI words): one lane per G1 point, one lane pair per G2 point; identity: zero words
template <class DUMMY>
__global__ void __launch_bounds__(64) k_fold_affine_g1(const uint32_t *__restrict__ xyzz, const uint8_t *__restrict__ inf, size_t n, uint32_t *__restrict__ out_abi) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    uint32_t *o = out_abi + i * 24;
    if (inf[i]) { for (int k = 0; k < 24; k++) o[k] = 0; return; }
    Xyzz<Fp> p; const uint32_t *w = xyzz + i * FOLD_XW1;
    fp_from_abi(p.x, w); fp_from_abi(p.y, w + 12); fp_from_abi(p.zz, w + 24); fp_from_abi(p.zzz, w + 36);
    Aff<Fp> a; xyzz_to_affine(a, p);
    fp_to_abi(o, a.x); fp_to_abi(o + 12, a.y);
}
template <class DUMMY>
__global__ void __launch_bounds__(64) k_fold_affine_g2(const uint32_t *__restrict__ xyzz, const uint8_t *__restrict__ inf, size_t n, uint32_t *__restrict__ out_abi) {
    typedef Fp2H F;
    const size_t i = ((size_t)blockIdx.x * 64 + threadIdx.x) >> 1;
    const uint32_t h = threadIdx.x & 1u;
    if (i >= n) return;
    uint32_t *o = out_abi + i * 48;
    if (inf[i]) { for (int k = 0; k < 12; k++) { o[12 * h + k] = 0; o[12 * (2 + h) + k] = 0; } return; }
    Xyzz<F> p; const uint32_t *w = xyzz + i * FOLD_XW2;
    fp_from_abi(p.x.v, w + 12 * h); fp_from_abi(p.y.v, w + 24 + 12 * h); fp_from_abi(p.zz.v, w + 48 + 12 * h); fp_from_abi(p.zzz.v, w + 72 + 12 * h);
    Aff<F> a; xyzz_to_affine(a, p);
    fp_to_abi(o + 12 * h, a.x.v); fp_to_abi(o + 12 * (2 + h), a.y.v);
}

// ---- out_i = [A_i +] s_i P_i with a scalar of its own per point: one joint chain on four lanes ------------------------------------------------
// The call of RandomizedPairingChecker's scalings (utils/src/randomized_pairing_check.rs:125-129,152-158: `a.mul_bigint(m)` per source) and of the batch
// verifier's r^i A_i (dgpu_legogroth16_verify_batch).  k_g1_scale (sort_kernels.hip.h) runs k1 P and k2 P as TWO 128-step chains on two lanes each and
// joins them at the end; in a wave of sixteen points some chain always has its bit set, so every step is a doubling AND a mixed addition, five products
// deep each: 1280 product-depths.  Here the four lanes of a point run ONE chain over both halves of the GLV split through the round forms of ec29.hip.h
// (a doubling three products deep, an addition four) with TWO bits of each half per addition: the addend of a step is T[d1 + 4 d2] = d1 P + d2 phi(P),
// d1, d2 < 4, from a table of fifteen sums the quad builds first (one doubling, ten additions, three products by beta) and keeps in LDS —
// 64 x (2 x 3 + 4) = 640 product-depths, and the additions fall on the same steps for every point of the wave.  13 x 30-bit signed field.  Points are
// elements of the prime-order subgroup (phi(P) = lambda P holds there, and no table entry is the identity).
// 53 KB of LDS per 64-lane block: three blocks per CU, i.e. 12 288 points in one round of the chip — the call sites scale hundreds to a few thousand points.
constexpr int SCQ_ENTRY = 4 * SN, SCQ_STRIDE = 16 * SCQ_ENTRY + 1;          // words per entry / per quad (odd: the sixteen quads of a wave read different banks)
__global__ void __launch_bounds__(64) k_g1_scale_quad(const uint32_t *__restrict__ p_abi, const uint8_t *__restrict__ is_inf, const uint32_t *__restrict__ scalars, int scalar_stride,
                                                      const uint8_t *__restrict__ negate, size_t n, uint32_t *__restrict__ out_abi, uint8_t *__restrict__ out_inf,
                                                      const uint32_t *__restrict__ add_abi, const uint8_t *__restrict__ add_inf) {
    typedef Fs F;
    __shared__ uint32_t tab[16 * SCQ_STRIDE];
    const size_t i = ((size_t)blockIdx.x * 64 + threadIdx.x) >> 2;
    if (i >= n) return;                                                      // (whole quads leave together; nothing below is a block-wide barrier)
    const QuadLanes<1> q4;
    uint32_t *const mine = tab + (threadIdx.x >> 2) * SCQ_STRIDE;
    uint32_t any = 0;
    for (int k = 0; k < 24; k++) any |= p_abi[i * 24 + k];
    const bool pinf = (any == 0) || (is_inf && is_inf[i]);
    // member r parks coordinate r of entry e (picked without a lane-dependent register index)
    auto park = [&](int e, const Xyzz<F> &x) __attribute__((always_inline)) {
        const uint32_t *wx = reinterpret_cast<const uint32_t *>(&x);
        uint32_t *dst = mine + e * SCQ_ENTRY + q4.role * SN;
#pragma unroll
        for (int j = 0; j < SN; j++) dst[j] = pick4(q4.role, wx[j], wx[SN + j], wx[2 * SN + j], wx[3 * SN + j]);
    };
    auto fetch = [&](Xyzz<F> &x, int e) __attribute__((always_inline)) {
        uint32_t *o = reinterpret_cast<uint32_t *>(&x);
        const uint32_t *src = mine + e * SCQ_ENTRY;
#pragma unroll
        for (int j = 0; j < SCQ_ENTRY; j++) o[j] = src[j];
    };
    Xyzz<F> acc;
    if (!pinf) {
        F beta;
        {   // beta (ec29_two_lane.hip.h xyzz_phi's constant) from the 29-bit form into this field
            constexpr uint32_t B_[NL] = BLS29_BETA;
            Fp b29; uint32_t w[12];
#pragma unroll
            for (int k = 0; k < NL; k++) b29.l[k] = B_[k];
            CHK(chk_set_N(b29, 1.0);)
            fp_to_abi(w, b29); fs_from_abi(beta, w);
        }
        Xyzz<F> P1, M;
        fs_from_abi(P1.x, p_abi + i * 24); fs_from_abi(P1.y, p_abi + i * 24 + 12); fset_one(P1.zz); fset_one(P1.zzz);
        park(1, P1);
        xyzz_dbl_rounds(M, P1, q4); park(2, M);                              // 2 P
        { bool f = false; xyzz_add_rounds(M, f, P1, false, q4); } park(3, M); // 3 P
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // (one wave per block: the table is written and read by the same wave, in order)
#pragma unroll 1
        for (int d2 = 1; d2 < 4; d2++) {                                     // d2 phi(P) = phi(d2 P): x times beta
            Xyzz<F> Q; fetch(Q, d2);
            { F xn; fnorm(xn, Q.x); fmul(Q.x, xn, beta); }
            park(4 * d2, Q);
#pragma unroll 1
            for (int d1 = 1; d1 < 4; d1++) {
                Xyzz<F> S; fetch(S, d1);
                bool f = false; xyzz_add_rounds(S, f, Q, false, q4);         // (d1 + d2 lambda) P: never the identity, and d1 P != +- d2 phi(P)
                park(4 * d2 + d1, S);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // (one wave per block: the table is written and read by the same wave, in order)
    }
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    bool inf = true;
    const uint32_t *s = scalars + i * (size_t)scalar_stride;                 // (k1 | k2), four words each
    if (!pinf) {
#pragma unroll 1
        for (int b = 126; b >= 0; b -= 2) {
            { Xyzz<F> d; xyzz_dbl_rounds(d, acc, q4); xyzz_dbl_rounds(acc, d, q4); }        // (on an identity accumulator: zeros in, zeros out)
            const uint32_t sel = ((s[b >> 5] >> (b & 31)) & 3u) | (((s[4 + (b >> 5)] >> (b & 31)) & 3u) << 2);
            Xyzz<F> B; fetch(B, sel ? (int)sel : 1);
            xyzz_add_rounds(acc, inf, B, sel == 0, q4);
        }
    }
    if (add_abi) {
        const uint32_t *src = add_abi + i * 24;
        uint32_t nz = 0;
        for (int k = 0; k < 24; k++) nz |= src[k];
        Xyzz<F> A; fs_from_abi(A.x, src); fs_from_abi(A.y, src + 12); fset_one(A.zz); fset_one(A.zzz);
        xyzz_add_rounds(acc, inf, A, nz == 0 || (add_inf && add_inf[i]), q4);
    }
    uint32_t *o = out_abi + i * 24;
    if (q4.role == 0) out_inf[i] = inf;
    if (inf) { if (q4.role == 0) for (int k = 0; k < 24; k++) o[k] = 0; return; }
    // (X / ZZ, Y / ZZZ) with 1 / ZZ = (ZZ / ZZZ)^2; the inversion is the division-step one of the 14 x 29-bit field (every member runs it: one chain either way)
    Fp z29, i29; F i3, t, i2, xn, yn, x, y;
    fnorm(t, acc.zzz); fp_from_fs(z29, t);
    fp_inv_device(i29, z29);
    fs_from_fp(i3, i29);
    fnorm(t, acc.zz); fmul(t, t, i3); fsqr(i2, t);
    fnorm(xn, acc.x); fnorm(yn, acc.y);
    fmul(x, xn, i2); fmul(y, yn, i3);
    fcond_neg(y, negate && negate[i]);
    if (q4.role == 0) fs_to_abi(o, x); else if (q4.role == 1) fs_to_abi(o + 12, y);
}

// ---- the same with FOUR quads per point, in different waves: the pieces of the chain side by side ----------------------------------------------
// A chain of 64 two-bit steps is 128 doublings whatever is added between them.  Cut into pieces: the quad that owns the steps [L, H) runs them and then
// 2 L bare doublings (its digits are worth 4^L times what they say) — 10 (H - L) + 6 L product-depths; with the cuts at steps 40, 56, 62 the four pieces
// cost 400, 400, 396, 392 instead of the 640 of one quad.  The quads of a point sit in DIFFERENT waves of a 256-lane block of sixteen points (in one wave
// they would run in lockstep through each other's additions); every wave builds P, 2 P, 3 P, waves 1 .. 3 one row d2 phi(P) + d1 P of the table each
// (20 depths instead of 44); the pieces meet through LDS and wave 3 adds them, inverts and writes.  62 KB of LDS per block.
constexpr int SCO_WAVES = 4;
__device__ constexpr int SCO_LO[SCO_WAVES] = {62, 56, 40, 0}, SCO_HI[SCO_WAVES] = {64, 62, 56, 40};      // wave w owns the steps [LO, HI): bits 2 LO .. 2 HI - 1
__global__ void __launch_bounds__(64 * SCO_WAVES) k_g1_scale_oct(const uint32_t *__restrict__ p_abi, const uint8_t *__restrict__ is_inf, const uint32_t *__restrict__ scalars, int scalar_stride,
                                                                 const uint8_t *__restrict__ negate, size_t n, uint32_t *__restrict__ out_abi, uint8_t *__restrict__ out_inf,
                                                                 const uint32_t *__restrict__ add_abi, const uint8_t *__restrict__ add_inf) {
    typedef Fs F;
    constexpr int HW = SCQ_ENTRY + 1;
    __shared__ uint32_t tab[16 * SCQ_STRIDE];
    __shared__ uint32_t hand[(SCO_WAVES - 1) * 16 * HW];                     // the pieces of waves 0 .. 2 of every point (+ identity flags)
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6), pq = lane >> 2;
    const size_t i_raw = (size_t)blockIdx.x * 16 + pq;
    const bool live = i_raw < n;                                             // (padding quads follow the barriers and write nothing)
    const size_t i = live ? i_raw : 0;
    const QuadLanes<1> q4;
    uint32_t *const mine = tab + pq * SCQ_STRIDE;
    uint32_t any = 0;
    for (int k = 0; k < 24; k++) any |= p_abi[i * 24 + k];
    const bool pinf = !live || (any == 0) || (is_inf && is_inf[i]);
    auto park_at = [&](uint32_t *base, const Xyzz<F> &x) __attribute__((always_inline)) {
        const uint32_t *wx = reinterpret_cast<const uint32_t *>(&x);
        uint32_t *dst = base + q4.role * SN;
#pragma unroll
        for (int j = 0; j < SN; j++) dst[j] = pick4(q4.role, wx[j], wx[SN + j], wx[2 * SN + j], wx[3 * SN + j]);
    };
    auto fetch_at = [&](Xyzz<F> &x, const uint32_t *src) __attribute__((always_inline)) {
        uint32_t *o = reinterpret_cast<uint32_t *>(&x);
#pragma unroll
        for (int j = 0; j < SCQ_ENTRY; j++) o[j] = src[j];
    };
    if (!pinf) {
        Xyzz<F> P1, P2, P3;
        fs_from_abi(P1.x, p_abi + i * 24); fs_from_abi(P1.y, p_abi + i * 24 + 12); fset_one(P1.zz); fset_one(P1.zzz);
        xyzz_dbl_rounds(P2, P1, q4);
        P3 = P2; { bool f = false; xyzz_add_rounds(P3, f, P1, false, q4); }
        if (wave == 0) { park_at(mine + 1 * SCQ_ENTRY, P1); park_at(mine + 2 * SCQ_ENTRY, P2); park_at(mine + 3 * SCQ_ENTRY, P3); }
        else {                                                               // row d2 = wave: phi(d2 P) (x times beta), then + d1 P
            F beta;
            {
                constexpr uint32_t B_[NL] = BLS29_BETA;
                Fp b29; uint32_t w[12];
#pragma unroll
                for (int k = 0; k < NL; k++) b29.l[k] = B_[k];
                CHK(chk_set_N(b29, 1.0);)
                fp_to_abi(w, b29); fs_from_abi(beta, w);
            }
            Xyzz<F> Q = wave == 1 ? P1 : (wave == 2 ? P2 : P3);
            { F xn; fnorm(xn, Q.x); fmul(Q.x, xn, beta); }
            park_at(mine + 4 * wave * SCQ_ENTRY, Q);
#pragma unroll 1
            for (int d1 = 1; d1 < 4; d1++) {
                Xyzz<F> S = d1 == 1 ? P1 : (d1 == 2 ? P2 : P3);
                bool f = false; xyzz_add_rounds(S, f, Q, false, q4);         // (d1 + d2 lambda) P: never the identity, and d1 P != +- d2 phi(P)
                park_at(mine + (4 * wave + d1) * SCQ_ENTRY, S);
            }
        }
    }
    __syncthreads();                                                         // (the table is complete; every lane of the block is here)
    Xyzz<F> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    const uint32_t *s = scalars + i * (size_t)scalar_stride;                 // (k1 | k2), four words each
    if (!pinf) {
        const int lo = SCO_LO[wave], hi = SCO_HI[wave];
#pragma unroll 1
        for (int j = hi - 1; j >= lo; j--) {                                 // two doublings, then the entry the bits 2 j + 1, 2 j of k1 and k2 name
            { Xyzz<F> d; xyzz_dbl_rounds(d, acc, q4); xyzz_dbl_rounds(acc, d, q4); }
            const int b = 2 * j;
            const uint32_t sel = ((s[b >> 5] >> (b & 31)) & 3u) | (((s[4 + (b >> 5)] >> (b & 31)) & 3u) << 2);
            Xyzz<F> B; fetch_at(B, mine + (sel ? (int)sel : 1) * SCQ_ENTRY);
            xyzz_add_rounds(acc, inf, B, sel == 0, q4);
        }
#pragma unroll 1
        for (int k = 0; k < lo; k++) { Xyzz<F> d; xyzz_dbl_rounds(d, acc, q4); xyzz_dbl_rounds(acc, d, q4); }
    }
    if (wave < SCO_WAVES - 1) {
        uint32_t *h = hand + (wave * 16 + pq) * HW;
        park_at(h, acc);
        if (q4.role == 0) h[SCQ_ENTRY] = inf ? 1u : 0u;
    }
    __syncthreads();
    if (wave != SCO_WAVES - 1 || !live) return;
#pragma unroll 1
    for (int w = 0; w < SCO_WAVES - 1; w++) {
        const uint32_t *h = hand + (w * 16 + pq) * HW;
        Xyzz<F> part; fetch_at(part, h);
        xyzz_add_rounds(acc, inf, part, h[SCQ_ENTRY] != 0, q4);
    }
    if (add_abi) {
        const uint32_t *src = add_abi + i * 24;
        uint32_t nz = 0;
        for (int k = 0; k < 24; k++) nz |= src[k];
        Xyzz<F> A; fs_from_abi(A.x, src); fs_from_abi(A.y, src + 12); fset_one(A.zz); fset_one(A.zzz);
        xyzz_add_rounds(acc, inf, A, nz == 0 || (add_inf && add_inf[i]), q4);
    }
    uint32_t *o = out_abi + i * 24;
    if (q4.role == 0) out_inf[i] = inf;
    if (inf) { if (q4.role == 0) for (int k = 0; k < 24; k++) o[k] = 0; return; }
    Fp z29, i29; F i3, t, i2, xn, yn, x, y;
    fnorm(t, acc.zzz); fp_from_fs(z29, t);
    fp_inv_device(i29, z29);
    fs_from_fp(i3, i29);
    fnorm(t, acc.zz); fmul(t, t, i3); fsqr(i2, t);
    fnorm(xn, acc.x); fnorm(yn, acc.y);
    fmul(x, xn, i2); fmul(y, yn, i3);
    fcond_neg(y, negate && negate[i]);
    if (q4.role == 0) fs_to_abi(o, x); else if (q4.role == 1) fs_to_abi(o + 12, y);
}

}  // namespace msm
