// crypto_amd/csrc/reduce_kernels.hip.h — bucket reduction of the shared bucket set by BIT MARGINALS (round 5).
//
// The table pipeline (msm_driver.hip.h, pre_finish) ends in  sum_b (b + 1) B_b  over ONE set of NB = 2^(c-1) buckets — what arkworks'
// VariableBaseMSM does per window with a running sum on one core (ark-ec 0.4 msm_bigint_wnaf; entered from
// legogroth16/src/prover.rs:286,299,592 and utils/src/pairs.rs:145-155).  k_reduce_l0 / k_reduce_top_quad (msm_kernels.hip.h) spend 16
// general additions per lane on its 8 buckets and then 16 more in the wave (a suffix scan, doublings by the lane weight, a tree), and the top
// kernel repeats the scan over the groups: 3.8 additions per bucket where the running sum needs 2.
//
// Here a point-lane L still owns m consecutive buckets and takes  S_L = sum B,  A_L = sum (k + 1) B  with the running sum (2 (m - 1)
// additions; the first bucket is a copy), so that the result is  sum_L A_L + m sum_L L S_L.  The weighted sum over the lanes is NOT scanned:
//
//      sum_L L S_L  =  sum_t 2^t M_t ,      M_t = sum of the S_L whose index L has bit t set,
//
// and every M_t, the total T = sum S_L and P = sum A_L are PLAIN sums.  One butterfly network of log2(lanes) steps produces all of them at
// once, one addition per lane and step (tree_step): at step s (distance d = 2^(s-1))
//   * lanes whose low s bits are zero add their partner's S-tree value (the total),
//   * the partners (low s bits == d) keep theirs: these 2^(6-s) values are the leaves of M_(s-1), summed by the later steps on lanes whose
//     low bits are exactly 2^t — a tree per class, in disjoint lanes,
//   * lanes whose low s bits are all ones add their partner's A-tree value (P).
// A lane gives its S register when its bit s-1 is set and its A register otherwise, and takes from lane ^ d: one exchange, one addition,
// whatever the lane's role.  After six steps lane 0 holds T, lane 2^t holds M_t, lane 63 holds P: 6 additions instead of 16, no doubling.
// The classes of all waves are again plain sums (and the wave totals T_g a weighted sum over g: the same network one level up):
// k_reduce_cls folds 64 values of two classes per wave until one value per class is left — P and one M per bit of the lane index — and
// the host finishes with a Horner pass over ~17 points (host_fold_marginals).  G2 runs the same code on lane pairs (32 point-lanes per wave).
//
// Any order of additions gives the same group element; the ABI returns the normalised representative, so results are bit-identical to the
// scan form (tests sweep both).
#pragma once
#include "msm_kernels.hip.h"

namespace msm {

template <class C> struct RedGeom {
    static constexpr int LPP = C::LPP;
    static constexpr int PL = 64 / LPP;                 // point-lanes per wave
    static constexpr int K = LPP == 1 ? 6 : 5;          // log2(PL): marginal classes a full wave produces
    static constexpr int PW = (int)(sizeof(Xyzz<typename C::F>) / 4);      // words of a point held by one lane
};
__host__ __device__ inline int red_log2(unsigned v) { int r = 0; while ((1u << (r + 1)) <= v) r++; return r; }
// points the class buffers of all levels hold (level 0: (K + 2) classes of NG values; every later level: two more than it has marginals, one value per wave)
inline size_t reduce_m_points(size_t NG, int PL, int K) {
    size_t total = (size_t)(K + 2) * NG; size_t cnt = NG; int nm = K;
    while (cnt > 1) { const int steps = std::min(K, red_log2((unsigned)cnt)); const size_t chunks = std::max<size_t>(1, cnt / PL); nm += steps; total += (size_t)(nm + 2) * chunks; cnt = chunks; }
    return total + 64;
}

// o <- the point held by point-lane (pl ^ d)
template <class C> __device__ __forceinline__ void shfl_xor_point(Xyzz<typename C::F> &o, bool &oinf, const Xyzz<typename C::F> &x, bool xinf, int d) {
    constexpr int PW_ = RedGeom<C>::PW;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&x);
    uint32_t *q = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
    for (int k = 0; k < PW_; k++) q[k] = (uint32_t)__shfl_xor((int)w[k], d * C::LPP, 64);
    oinf = __shfl_xor((int)xinf, d * C::LPP, 64) != 0;
}
// One step of the network (see the header).  S / A: this lane's two registers; pl: point-lane index inside the wave; weighted: the S values
// carry a weight (their marginal classes are wanted) — false for a pair of plain classes, whose totals alone are formed (lane 0 and lane 2d - 1).
template <class C> __device__ __forceinline__ void tree_step(Xyzz<typename C::F> &S, bool &sinf, Xyzz<typename C::F> &A, bool &ainf, int pl, int d, bool weighted) {
    typedef typename C::F F;
    constexpr int PW_ = RedGeom<C>::PW;
    const bool hi = (pl & d) != 0;
    uint32_t *ws = reinterpret_cast<uint32_t *>(&S), *wa = reinterpret_cast<uint32_t *>(&A);
    // hi lanes work on A and give S; the others work on S and give A: swap in place so that the addition below is one instruction stream
#pragma unroll
    for (int k = 0; k < PW_; k++) { const uint32_t s = ws[k], a = wa[k]; ws[k] = hi ? a : s; wa[k] = hi ? s : a; }
    { const bool s = sinf, a = ainf; sinf = hi ? a : s; ainf = hi ? s : a; }
    Xyzz<F> o; bool oinf;
    shfl_xor_point<C>(o, oinf, A, ainf, d);
    const int low = pl & (d - 1);
    const bool active = hi ? (low == d - 1) : (low == 0 || (weighted && (low & (low - 1)) == 0));
    xyzz_add(S, sinf, o, oinf || !active);
#pragma unroll
    for (int k = 0; k < PW_; k++) { const uint32_t s = ws[k], a = wa[k]; ws[k] = hi ? a : s; wa[k] = hi ? s : a; }
    { const bool s = sinf, a = ainf; sinf = hi ? a : s; ainf = hi ? s : a; }
}

// ---- level 0: buckets -> (T, P, M_0 .. M_(K-1)) per wave -------------------------------------------------------------------------------
// cls[c * NG + g] (C::XW words each, the layout of the bucket array) / cls_inf: class c of wave g; c = 0: T, 1: P, 2 + t: M_t.
// Four waves per block (whole CUs, see k_reduce_l0); the waves do not synchronise.
template <class C>
__global__ void __launch_bounds__(256, 1) k_reduce_m0(const uint32_t *__restrict__ bucket, const uint8_t *__restrict__ bucket_inf, uint32_t NB, int mshift,
                                                      uint32_t *__restrict__ cls, uint8_t *__restrict__ cls_inf, unsigned NG) {
    typedef typename C::F F;
    constexpr int LPP = RedGeom<C>::LPP, PL = RedGeom<C>::PL, K = RedGeom<C>::K;
    const int lane = threadIdx.x & 63, pl = lane / LPP;
    const uint32_t m = (uint32_t)LPP << mshift;                    // buckets per point-lane: a wave covers 64 * 2^mshift buckets on either curve
    const size_t g = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= NG) return;
    const size_t b0 = (g * PL + pl) * (size_t)m;
    { bool any = false; for (uint32_t k = 0; k < m; k++) any = any || bucket_inf[b0 + k] == 0;
      if (!__any(any)) { if (lane < K + 2) cls_inf[(size_t)lane * NG + g] = 1; return; } }     // a wave of empty buckets: K + 2 identity flags
    Xyzz<F> run, tot; bool rinf, tinf;
    {   // the last bucket is copied, not added
        const size_t b = b0 + m - 1;
        rinf = bucket_inf[b] != 0;
        fzero(run.x); fzero(run.y); fzero(run.zz); fzero(run.zzz);
        if (!rinf) load_soa<C>(run, bucket, NB, b);
        tot = run; tinf = rinf;
    }
    // ONE addition in ONE loop for the running sums (run += B_k, then tot += run: 2 (m - 1) iterations) and for the K steps of the network: with
    // an addition inlined per use the loop body of the running sums alone was 98 KB of straight-line code, which the 64-KB instruction cache (shared by two
    // CUs) re-fetched on every iteration, and a wave that is alone on its SIMD waits for every line (6 cycles per instruction where k_accumulate,
    // whose body fits, runs at 4.4).  Which register is added to is a per-lane (network) or per-iteration (running sums) swap in front of the addition.
    const int nserial = 2 * ((int)m - 1);
    constexpr int PW_ = RedGeom<C>::PW;
    uint32_t *wr = reinterpret_cast<uint32_t *>(&run), *wt = reinterpret_cast<uint32_t *>(&tot);
#pragma unroll 1
    for (int it = 0; it < nserial + K; it++) {
        const bool serial = it < nserial;
        const int d = serial ? 0 : 1 << (it - nserial);
        const bool hi = serial ? (it & 1) != 0 : (pl & d) != 0;       // work on `tot` (and give `run`)
#pragma unroll
        for (int k = 0; k < PW_; k++) { const uint32_t r = wr[k], t = wt[k]; wr[k] = hi ? t : r; wt[k] = hi ? r : t; }
        { const bool r = rinf, t = tinf; rinf = hi ? t : r; tinf = hi ? r : t; }
        Xyzz<F> o; bool oinf;
        if (serial) {
            if ((it & 1) == 0) { const size_t b = b0 + (m - 2 - (uint32_t)(it >> 1)); oinf = bucket_inf[b] != 0; if (!oinf) load_soa<C>(o, bucket, NB, b); else o = run; }
            else { o = tot; oinf = tinf; }                              // (after the swap `tot` holds the running sum)
        } else {
            shfl_xor_point<C>(o, oinf, tot, tinf, d);
            const int low = pl & (d - 1);
            const bool active = hi ? (low == d - 1) : (low == 0 || (low & (low - 1)) == 0);
            oinf = oinf || !active;
        }
        xyzz_add(run, rinf, o, oinf);
#pragma unroll
        for (int k = 0; k < PW_; k++) { const uint32_t r = wr[k], t = wt[k]; wr[k] = hi ? t : r; wt[k] = hi ? r : t; }
        { const bool r = rinf, t = tinf; rinf = hi ? t : r; tinf = hi ? r : t; }
    }
    const bool first = (lane % LPP) == 0;
    if (pl == 0) { store_soa<C>(cls, NG, g, run); if (first) cls_inf[g] = rinf; }
    else if ((pl & (pl - 1)) == 0) { const size_t c = 2 + red_log2((unsigned)pl); store_soa<C>(cls + c * NG * C::XW, NG, g, run); if (first) cls_inf[c * NG + g] = rinf; }
    if (pl == PL - 1) { store_soa<C>(cls + (size_t)NG * C::XW, NG, g, tot); if (first) cls_inf[NG + g] = tinf; }
}

// ---- levels >= 1: 64 (32) values of two classes per wave ------------------------------------------------------------------------------
// in: classes [T, P, M_0 .. M_(nm-1)] of cnt values each (cnt a power of two).  Wave (pair, j): pair 0 = (T weighted, P), pair p >= 1 = (M_(2p-2),
// M_(2p-1)); j = which PL consecutive values.  out: classes [T', P', M_0 .. M_(nm-1), M_nm .. M_(nm+steps-1)] of chunks = max(1, cnt / PL) values:
// the new marginals are those of the bits of the wave index g the input T values carried.
// final_ (chunks == 1): instead of `out`, the ABI form goes to win_abi / win_inf: point 0 = P, point 1 + t = M_t (T is not needed any more).
template <class C>
__global__ void __launch_bounds__(256, 1) k_reduce_cls(const uint32_t *__restrict__ in, const uint8_t *__restrict__ in_inf, unsigned cnt, int nm,
                                                       uint32_t *__restrict__ out, uint8_t *__restrict__ out_inf, int final_,
                                                       uint32_t *__restrict__ win_abi, uint8_t *__restrict__ win_inf) {
    typedef typename C::F F;
    constexpr int LPP = RedGeom<C>::LPP, PL = RedGeom<C>::PL, K = RedGeom<C>::K;
    const int lane = threadIdx.x & 63, pl = lane / LPP, h = lane % LPP;
    const unsigned chunks = cnt / PL ? cnt / PL : 1u;
    const int steps = red_log2(cnt) < K ? red_log2(cnt) : K;
    const unsigned npairs = 1u + (unsigned)(nm + 1) / 2u;
    const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned pair = w / chunks, j = w % chunks;
    if (pair >= npairs) return;
    const unsigned c0 = pair == 0 ? 0u : 2u * pair, c1 = c0 + 1;
    const bool has1 = c1 < 2u + (unsigned)nm;
    const unsigned idx = j * PL + pl;
    Xyzz<F> S, A; bool sinf = true, ainf = true;
    fzero(S.x); fzero(S.y); fzero(S.zz); fzero(S.zzz); A = S;
    if (idx < cnt) {
        sinf = in_inf[(size_t)c0 * cnt + idx] != 0;
        if (!sinf) load_soa<C>(S, in + (size_t)c0 * cnt * C::XW, cnt, idx);
        if (has1) { ainf = in_inf[(size_t)c1 * cnt + idx] != 0; if (!ainf) load_soa<C>(A, in + (size_t)c1 * cnt * C::XW, cnt, idx); }
    }
#pragma unroll 1
    for (int s = 0; s < steps; s++) tree_step<C>(S, sinf, A, ainf, pl, 1 << s, pair == 0);
    // who holds what: S total on point-lane 0, A total on point-lane 2^steps - 1, new marginal t on point-lane 2^t (pair 0 only)
    const bool outS = pl == 0, outA = has1 && pl == (1 << steps) - 1;
    const bool outM = pair == 0 && pl < (1 << steps) && pl != 0 && (pl & (pl - 1)) == 0;
    const unsigned cm = 2u + (unsigned)nm + (unsigned)red_log2((unsigned)pl);
    if (!final_) {
        if (outS) { store_soa<C>(out + (size_t)c0 * chunks * C::XW, chunks, j, S); if (h == 0) out_inf[(size_t)c0 * chunks + j] = sinf; }
        if (outA) { store_soa<C>(out + (size_t)c1 * chunks * C::XW, chunks, j, A); if (h == 0) out_inf[(size_t)c1 * chunks + j] = ainf; }
        if (outM) { store_soa<C>(out + (size_t)cm * chunks * C::XW, chunks, j, S); if (h == 0) out_inf[(size_t)cm * chunks + j] = sinf; }
        return;
    }
    // class c -> ABI point: P (c = 1) -> 0, M_t (c = 2 + t) -> 1 + t; T (c = 0) is dropped
    constexpr int WS = 4 * 12 * LPP;                                 // ABI words per point
    auto emit = [&](unsigned c, const Xyzz<F> &pt, bool inf) __attribute__((always_inline)) {
        if (c == 0) return;
        const unsigned slot = c - 1;
        if (h == 0) win_inf[slot] = inf;
        if (inf) return;
        const Fs *f = reinterpret_cast<const Fs *>(&pt);             // x, y, zz, zzz (G2: this lane's halves)
#pragma unroll 1
        for (int k = 0; k < 4; k++) {                                // (the coordinate is picked with selects: an index into the registers would go through scratch)
            Fs v;
#pragma unroll
            for (int i = 0; i < SN; i++) v.l[i] = pick4(k, f[0].l[i], f[1].l[i], f[2].l[i], f[3].l[i]);
            fs_to_abi(win_abi + (size_t)slot * WS + 12 * (LPP * k + h), v);
        }
    };
    if (outS || outM) emit(outS ? c0 : cm, S, sinf);
    if (outA) emit(c1, A, ainf);
}

// ---- levels >= 1 with four members per point (the form of k_reduce_top_quad) ----------------------------------------------------------------
// The class folds are a handful of waves on a chain of 4 .. 6 additions: with one lane per point each of those waves first has to FETCH the
// ~55 KB of straight-line code of an inlined addition through a cold instruction cache (measured: 0.17 - 0.3 ms per launch for 6 additions).
// Here every value is held by four members (lanes; lane pairs for G2) that multiply one role-selected operand pair per round
// (ec29.hip.h xyzz_add_rounds through QuadLanes): a quarter of the code and of the chain.  One block = 64 values of a class pair
// (64 groups of 4 * LPP lanes); the exchange of a network step goes through LDS (member r parks coordinate r of the register its group gives).
// Same class bookkeeping as k_reduce_cls with PL = 64 on either curve.
template <class C>
__global__ void __launch_bounds__(256 * C::LPP) k_reduce_cls_quad(const uint32_t *__restrict__ in, const uint8_t *__restrict__ in_inf, unsigned cnt, int nm,
                                                                  uint32_t *__restrict__ out, uint8_t *__restrict__ out_inf, int final_,
                                                                  uint32_t *__restrict__ win_abi, uint8_t *__restrict__ win_inf) {
    typedef typename C::F F;
    constexpr int LPP = C::LPP, GL = 4 * LPP, PW_ = 4 * SN, PLQ = 64;
    __shared__ uint32_t xs[64 * LPP * PW_];
    __shared__ uint8_t fl[64];
    const int t = (int)threadIdx.x, gi = t / GL, h = t % LPP;
    const QuadLanes<LPP> q4;
    const unsigned chunks = cnt / PLQ ? cnt / PLQ : 1u;
    const int steps = red_log2(cnt) < 6 ? red_log2(cnt) : 6;
    const unsigned pair = blockIdx.x / chunks, j = blockIdx.x % chunks;
    const unsigned c0 = pair == 0 ? 0u : 2u * pair, c1 = c0 + 1;
    const bool has1 = c1 < 2u + (unsigned)nm;
    const unsigned idx = j * PLQ + (unsigned)gi;
    Xyzz<F> S, A; bool sinf = true, ainf = true;
    fzero(S.x); fzero(S.y); fzero(S.zz); fzero(S.zzz); A = S;
    if (idx < cnt) {
        sinf = in_inf[(size_t)c0 * cnt + idx] != 0;
        if (!sinf) load_soa<C>(S, in + (size_t)c0 * cnt * C::XW, cnt, idx);
        if (has1) { ainf = in_inf[(size_t)c1 * cnt + idx] != 0; if (!ainf) load_soa<C>(A, in + (size_t)c1 * cnt * C::XW, cnt, idx); }
    }
    uint32_t *ws = reinterpret_cast<uint32_t *>(&S), *wa = reinterpret_cast<uint32_t *>(&A);
#pragma unroll 1
    for (int s = 0; s < steps; s++) {
        const int d = 1 << s;
        const bool hi = (gi & d) != 0;
#pragma unroll
        for (int k = 0; k < PW_; k++) { const uint32_t a = ws[k], b = wa[k]; ws[k] = hi ? b : a; wa[k] = hi ? a : b; }
        { const bool a = sinf, b = ainf; sinf = hi ? b : a; ainf = hi ? a : b; }
        // every group parks the register it gives (now in A), then takes its partner's
        __syncthreads();
        { uint32_t *dst = xs + ((gi * 4 + q4.role) * LPP + h) * SN;
#pragma unroll
          for (int k = 0; k < SN; k++) dst[k] = pick4(q4.role, wa[k], wa[SN + k], wa[2 * SN + k], wa[3 * SN + k]);
          if (t % GL == 0) fl[gi] = ainf; }
        __syncthreads();
        Xyzz<F> o; bool oinf;
        { const int sg = gi ^ d;
          uint32_t *ov = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
          for (int c = 0; c < 4; c++) {
              const uint32_t *sv = xs + ((sg * 4 + c) * LPP + h) * SN;
#pragma unroll
              for (int k = 0; k < SN; k++) ov[c * SN + k] = sv[k];
          }
          oinf = fl[sg] != 0; }
        const int low = gi & (d - 1);
        const bool active = hi ? (low == d - 1) : (low == 0 || (pair == 0 && (low & (low - 1)) == 0));
        xyzz_add_rounds(S, sinf, o, oinf || !active, q4);
#pragma unroll
        for (int k = 0; k < PW_; k++) { const uint32_t a = ws[k], b = wa[k]; ws[k] = hi ? b : a; wa[k] = hi ? a : b; }
        { const bool a = sinf, b = ainf; sinf = hi ? b : a; ainf = hi ? a : b; }
    }
    const bool outS = gi == 0, outA = has1 && gi == (1 << steps) - 1;
    const bool outM = pair == 0 && gi < (1 << steps) && gi != 0 && (gi & (gi - 1)) == 0;
    const unsigned cm = 2u + (unsigned)nm + (unsigned)red_log2((unsigned)gi);
    const bool first = t % GL == 0;              // one member writes the flag; members with role 0 write the record (G2: both lanes of the pair, each its halves)
    if (!final_) {
        if (q4.role == 0) {
            if (outS) { store_soa<C>(out + (size_t)c0 * chunks * C::XW, chunks, j, S); if (first) out_inf[(size_t)c0 * chunks + j] = sinf; }
            if (outA) { store_soa<C>(out + (size_t)c1 * chunks * C::XW, chunks, j, A); if (first) out_inf[(size_t)c1 * chunks + j] = ainf; }
            if (outM) { store_soa<C>(out + (size_t)cm * chunks * C::XW, chunks, j, S); if (first) out_inf[(size_t)cm * chunks + j] = sinf; }
        }
        return;
    }
    constexpr int WS = 4 * 12 * LPP;
    auto emit = [&](unsigned c, const Xyzz<F> &pt, bool inf) __attribute__((always_inline)) {       // member r converts coordinate r (G2: each lane its half)
        if (c == 0) return;
        const unsigned slot = c - 1;
        if (first) win_inf[slot] = inf;
        if (inf) return;
        const Fs *f = reinterpret_cast<const Fs *>(&pt);
        Fs v;
#pragma unroll
        for (int i = 0; i < SN; i++) v.l[i] = pick4(q4.role, f[0].l[i], f[1].l[i], f[2].l[i], f[3].l[i]);
        fs_to_abi(win_abi + (size_t)slot * WS + 12 * (LPP * q4.role + h), v);
    };
    if (outS || outM) emit(outS ? c0 : cm, S, sinf);
    if (outA) emit(c1, A, ainf);
}

}  // namespace msm
