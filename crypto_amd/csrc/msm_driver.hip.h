// crypto_amd/csrc/msm_driver.hip.h — host driver of the MSM pipeline (templated on the curve), included by
// dock_g1.hip and dock_g2.hip so the two curves compile in parallel.
#pragma once
#include <chrono>
#include <thread>
#include "dock_ctx.hpp"
#include "bases_cache.hpp"
#include "host_field.hpp"
#include "msm_launch.hip.h"
#include "sort_launch.hip.h"
#include "qap_launch.hip.h"

namespace dock {
using namespace msm;

// host tail: Horner over window sums (ABI XYZZ form), normalised Jacobian out
template <class HF>
void host_fold(const uint64_t *win_abi, const uint8_t *win_inf, int W, int c, uint64_t *out_xyz) {
    typedef hostf::HXyzz<HF> PT;
    PT acc = PT::identity();
    const size_t FWORDS = sizeof(HF) / 8;
    for (int w = W - 1; w >= 0; w--) {
        if (!acc.inf) for (int k = 0; k < c; k++) acc.dbl_in_place();
        if (!win_inf[w]) {
            PT t; t.inf = false;
            const uint64_t *src = win_abi + (size_t)w * 4 * FWORDS;
            memcpy(&t.x, src, sizeof(HF)); memcpy(&t.y, src + FWORDS, sizeof(HF)); memcpy(&t.zz, src + 2 * FWORDS, sizeof(HF)); memcpy(&t.zzz, src + 3 * FWORDS, sizeof(HF));
            acc.add_in_place(t);
        }
    }
    HF X, Y, Z; acc.to_normalised_jacobian(X, Y, Z);
    memcpy(out_xyz, &X, sizeof(HF)); memcpy(out_xyz + FWORDS, &Y, sizeof(HF)); memcpy(out_xyz + 2 * FWORDS, &Z, sizeof(HF));
}

// sum of k Jacobian triples (host): partial results gathered from the other ranks
template <class HF>
int32_t host_fold_jacobian(const uint64_t *xyz, size_t k, uint64_t *out_xyz) {
    if (!out_xyz || (k && !xyz)) return DGPU_E_BADARG;
    typedef hostf::HXyzz<HF> PT;
    const size_t FWORDS = sizeof(HF) / 8;
    PT acc = PT::identity();
    for (size_t i = 0; i < k; i++) {
        HF X, Y, Z;
        memcpy(&X, xyz + i * 3 * FWORDS, sizeof(HF)); memcpy(&Y, xyz + i * 3 * FWORDS + FWORDS, sizeof(HF)); memcpy(&Z, xyz + i * 3 * FWORDS + 2 * FWORDS, sizeof(HF));
        if (Z.is_zero()) continue;
        PT t; t.inf = false; t.x = X; t.y = Y; t.zz = Z * Z; t.zzz = t.zz * Z;   // Jacobian (X, Y, Z) == XYZZ (X, Y, Z^2, Z^3)
        acc.add_in_place(t);
    }
    HF X, Y, Z; acc.to_normalised_jacobian(X, Y, Z);
    memcpy(out_xyz, &X, sizeof(HF)); memcpy(out_xyz + FWORDS, &Y, sizeof(HF)); memcpy(out_xyz + 2 * FWORDS, &Z, sizeof(HF));
    return DGPU_OK;
}

// sum_i s_i P_i over k <= DGPU_MAX_LINCOMB affine points on the HOST (4-bit windows, one table of 15 multiples per point, joint doublings).
// This is not the MSM path: it is the O(1) group arithmetic around it that the reference does with `mul_bigint` / FixedBase on the CPU — the
// r delta, s g_a + r g1_b, -rs delta - v eta/delta of a proof (prover.rs:309-313, 350-355, 585-594) — next to
// dgpu_fold_* and dgpu_final_exponentiation.  A 2..4-term product costs 0.15 - 0.35 ms of one host core and no device launch; the same
// through the bucket pipeline is ~0.75 ms of launch latency per call and queues behind the accumulation kernels of the large MSMs.
template <class HF>
int32_t host_lincomb(const uint64_t *points_xy, const uint8_t *is_inf, const uint64_t *scalars, size_t k, uint64_t *out_xyz) {
    if (!out_xyz || k > DGPU_MAX_LINCOMB || (k && (!points_xy || !scalars))) return DGPU_E_BADARG;
    typedef hostf::HXyzz<HF> PT;
    const size_t FWORDS = sizeof(HF) / 8;
    std::vector<PT> tab(k * 15);
    std::vector<uint8_t> live(k, 0);
    for (size_t i = 0; i < k; i++) {
        HF X, Y;
        memcpy(&X, points_xy + i * 2 * FWORDS, sizeof(HF)); memcpy(&Y, points_xy + i * 2 * FWORDS + FWORDS, sizeof(HF));
        const uint64_t *sc = scalars + 4 * i;
        uint64_t any = 0; for (size_t w = 0; w < 2 * FWORDS; w++) any |= points_xy[i * 2 * FWORDS + w];      // all-zero coordinates: the ABI's other spelling of the identity
        if ((is_inf && is_inf[i]) || !any || !(sc[0] | sc[1] | sc[2] | sc[3])) continue;
        live[i] = 1;
        PT p; p.inf = false; p.x = X; p.y = Y; p.zz = HF::one(); p.zzz = HF::one();
        tab[i * 15] = p;
        for (int m = 1; m < 15; m++) { PT t = tab[i * 15 + m - 1]; if (m == 1) t.dbl_in_place(); else t.add_in_place(p); tab[i * 15 + m] = t; }
    }
    PT acc = PT::identity();
    for (int w = 63; w >= 0; w--) {
        for (int d = 0; d < 4; d++) acc.dbl_in_place();
        for (size_t i = 0; i < k; i++) {
            if (!live[i]) continue;
            const unsigned nib = (unsigned)(scalars[4 * i + (w >> 4)] >> ((w & 15) * 4)) & 15u;
            if (nib) acc.add_in_place(tab[i * 15 + nib - 1]);
        }
    }
    HF X, Y, Z; acc.to_normalised_jacobian(X, Y, Z);
    memcpy(out_xyz, &X, sizeof(HF)); memcpy(out_xyz + FWORDS, &Y, sizeof(HF)); memcpy(out_xyz + 2 * FWORDS, &Z, sizeof(HF));
    return DGPU_OK;
}

template <class HF> void write_identity(uint64_t *out_xyz) {
    typedef hostf::HXyzz<HF> PT; PT id = PT::identity(); HF X, Y, Z; id.to_normalised_jacobian(X, Y, Z);
    const size_t FWORDS = sizeof(HF) / 8; memcpy(out_xyz, &X, sizeof(HF)); memcpy(out_xyz + FWORDS, &Y, sizeof(HF)); memcpy(out_xyz + 2 * FWORDS, &Z, sizeof(HF));
}

constexpr size_t PLAIN_PSORT_MIN_N = (size_t)1 << 17;
// ---- geometry + workspace of the plain pipeline ---------------------------------------------------------------------------------------
struct PlainGeom {
    int c, W; uint32_t B, NB; int mshift, G; size_t NG, Emax; int CH; size_t T, nblk; bool wide; size_t n_pad; int RANGES, rb_log; unsigned sort_grid; size_t lds_bytes;
    uint32_t min_chunk, max_chunks, lanes_per_chunk, HEAVY_CAP;
    bool psort;            // the two-level partition sort (psort_kernels.hip.h) instead of the per-window sweeps: large n
};
template <class C> int32_t plain_geometry(size_t n, PlainGeom &g) {
    if (n >= (1ull << 31)) return DGPU_E_BADARG;
    g.c = choose_c(n, C::NFP == 2);
    g.W = 255 / g.c + 1;
    g.B = 1u << (g.c - 1);
    if ((uint64_t)g.W * g.B >= (1ull << 31) || (uint64_t)n * g.W >= (1ull << 32)) return DGPU_E_BADARG;
    g.NB = (uint32_t)g.W * g.B;
    g.mshift = std::max(0, g.c - 1 - 12);
    g.G = (int)(g.B >> (6 + g.mshift));          // groups per window (<= 64), B >= 64 because c >= 7
    g.NG = (size_t)g.W * g.G;
    g.Emax = (size_t)n * g.W;
    g.CH = C::NFP == 2 ? choose_chunk(g.Emax, 32, 150000, 2) : choose_chunk(g.Emax, 16, 300000, 1);
    g.T = (g.Emax + g.CH - 1) / g.CH;
    g.nblk = scan_blocks(g.NB);
    // counting sort of the n*W (key, term) pairs: digit codes -> LDS histograms per (window, bucket range) -> scan -> LDS cursors
    g.wide = g.c > 16;
    g.n_pad = (n + 7) & ~(size_t)7;
    g.RANGES = 1; while ((g.B / g.RANGES) * 4 > 64 * 1024 || g.W * g.RANGES < 256) { if (g.B / g.RANGES <= 64) break; g.RANGES *= 2; }
    g.rb_log = 0; while ((1u << g.rb_log) < g.B / g.RANGES) g.rb_log++;
    g.sort_grid = (unsigned)(8 * ((g.W + 7) / 8) * g.RANGES);
    g.lds_bytes = ((size_t)1 << g.rb_log) * 4;
    // chunk length / heavy-bucket threshold of the accumulation are fixed on the device once the pair count is known (dyn_chunk.hip.h): CH and T
    // only size the launch and the partial slots
    g.min_chunk = C::NFP == 2 ? 32u : 16u; g.max_chunks = C::NFP == 2 ? 150000u : 300000u; g.lanes_per_chunk = C::NFP == 2 ? 2u : 1u;
    // a heavy bucket has >= 16 chunk lengths of terms and a chunk is never shorter than 16 terms (k_dyn_chunk, forced_chunk), whatever min_chunk says
    g.HEAVY_CAP = (uint32_t)(g.Emax / (16u * 16u)) + 1;
    // From 2^17 terms on the sweeps (every digit column re-read once per bucket range: 0.38 ms at n = 2^20) give way to the partition sort the table
    // pipeline uses (0.2 ms), with one bucket set per window in the key: key = w B + |digit| - 1
    g.psort = g.W <= PS_MAX_W && n >= PLAIN_PSORT_MIN_N;
    return DGPU_OK;
}
// grow-only workspace of one slot for the plain pipeline of that geometry (no-ops once the slot has seen the size: dgpu_reserve_*, uploads)
template <class C> int32_t ws_plain(Slot &sl, const PlainGeom &g) {
    int32_t rc;
    const size_t NB = g.NB, T = g.T;
    if ((rc = sl.flags.ensure(64))) return rc;
    if ((rc = sl.cnt.ensure((NB + 1) * 4))) return rc;
    if ((rc = sl.off.ensure((NB + 1) * 4))) return rc;
    if ((rc = sl.cursor.ensure((NB + 1) * 4))) return rc;
    if ((rc = sl.bsums.ensure((g.nblk + 2) * 4))) return rc;
    if ((rc = sl.entries.ensure(g.Emax * 4))) return rc;
    if ((rc = sl.bucket.ensure(soa_points(NB) * C::XW * 4))) return rc;
    if ((rc = sl.bucket_inf.ensure(NB))) return rc;
    if ((rc = sl.head.ensure(soa_points(T) * C::XW * 4))) return rc;
    if ((rc = sl.tail.ensure(soa_points(T) * C::XW * 4))) return rc;
    if ((rc = sl.head_b.ensure(T * 4))) return rc;
    if ((rc = sl.tail_b.ensure(T * 4))) return rc;
    if ((rc = sl.part_inf.ensure(T * 2))) return rc;
    if ((rc = sl.l1.ensure(g.NG * 2 * C::XW * 4))) return rc;
    if ((rc = sl.l1_inf.ensure(g.NG * 2))) return rc;
    if ((rc = sl.win.ensure((size_t)g.W * 4 * C::ABI_W * 4))) return rc;
    if ((rc = sl.win_inf.ensure(g.W))) return rc;
    if ((rc = sl.digits.ensure((size_t)g.W * g.n_pad * (g.wide ? 4 : 2)))) return rc;
    if (g.psort) {                                 // cnt1 / off1 per (partition, tile), the (key, value) pairs
        const size_t n_terms = g.Emax / g.W;
        const uint32_t P = (g.NB + (1u << ps_part_log(g.NB)) - 1) >> ps_part_log(g.NB);
        const size_t n1 = (size_t)P * ((n_terms + PS_TILE - 1) / PS_TILE);
        if ((rc = sl.cnt.ensure((n1 + 1) * 4))) return rc;
        if ((rc = sl.cursor.ensure((n1 + 1) * 4))) return rc;
        if ((rc = sl.bsums.ensure((scan_blocks(n1) + 2) * 4))) return rc;
        if ((rc = sl.digits.ensure(g.Emax * 8))) return rc;
    }
    if ((rc = sl.heavy.ensure(((size_t)g.HEAVY_CAP + 1) * 4))) return rc;
    if ((rc = sl.dyn.ensure(msm::dyn_words(T) * 4))) return rc;
    { const size_t hslots = 2 * (T / msm::HEAVY_RANGE + 2); if ((rc = sl.hpart.ensure(hslots * C::XW * 4))) return rc; if ((rc = sl.hpart_inf.ensure(hslots))) return rc; }
    return DGPU_OK;
}

// An MSM whose operands are still crossing PCIe is taken in K TERM RANGES: range k is sorted, accumulated and fixed up into a bucket set of its
// own as soon as its operands have landed (the kernels of range k run under the copy of range k + 1); the sets are then merged
// (k_merge_buckets: one general addition per bucket and extra set) and reduced once.  K = 1 is the plain case.
inline size_t range_count(size_t n, bool operands_in_flight) { return operands_in_flight && n >= ((size_t)1 << 19) ? 2 : 1; }
inline size_t range_lo(size_t n, size_t K, size_t k) { return ((n * k / K) + 7) & ~(size_t)7; }       // (multiples of 8 terms, except the end)
inline void range_bounds(size_t n, size_t K, size_t k, size_t &lo, size_t &hi) { lo = k == 0 ? 0 : std::min(n, range_lo(n, K, k)); hi = k + 1 == K ? n : std::min(n, range_lo(n, K, k + 1)); }
template <class C> int32_t ws_bucket_sets(Slot &sl, uint32_t NB, size_t K) {
    int32_t rc;
    if ((rc = sl.bucket.ensure(K * soa_points(NB) * C::XW * 4))) return rc;
    return sl.bucket_inf.ensure(K * (size_t)NB);
}

// ---- the small path: up to SMALL_MSM_MAX_N terms over plain bases (small_kernels.hip.h) ----------------------------------------------------
// the table of a resident handle (Handle::aux): (e + 1) 2^(64 s) P_i for s < 4, e < 8, and one identity flag per entry behind it
struct SmallSub { const uint32_t *tab; const uint8_t *tab_inf; };
template <class C> inline size_t small_sub_tab_bytes(size_t n) { return n * SMALL_MSM_S * SMALL_MSM_E * (size_t)C::ACC::XW * 4; }
template <class C> inline size_t small_sub_bytes(size_t n) { return small_sub_tab_bytes<C>(n) + n * SMALL_MSM_S * SMALL_MSM_E; }
template <class C> inline SmallSub small_sub_at(const void *aux, size_t handle_n, size_t offset) {
    const uint8_t *base = (const uint8_t *)aux;
    return SmallSub{(const uint32_t *)base + offset * SMALL_MSM_S * SMALL_MSM_E * (size_t)C::ACC::XW, base + small_sub_tab_bytes<C>(handle_n) + offset * SMALL_MSM_S * SMALL_MSM_E};
}
template <class C> int32_t ws_small(Slot &sl, size_t n) {
    int32_t rc;
    typedef typename C::ACC A;
    if ((rc = sl.bucket.ensure(n * SMALL_MSM_E * A::XW * 4))) return rc;
    if ((rc = sl.bucket_inf.ensure(n * SMALL_MSM_E))) return rc;
    if ((rc = sl.head.ensure((size_t)256 * A::XW * 4))) return rc;                // at most 4 blocks x 64 windows or 16 x 16 partials
    if ((rc = sl.part_inf.ensure(256))) return rc;
    if (!sl.small_cnt.p) {                                                          // the per-window block counters: zero once, every launch leaves them zero
        if ((rc = sl.small_cnt.ensure((size_t)SMALL_MSM_W * 4))) return rc;
        if (hipMemsetAsync(sl.small_cnt.p, 0, (size_t)SMALL_MSM_W * 4, sl.stream) != hipSuccess) { (void)hipGetLastError(); sl.small_cnt.release(); return DGPU_E_HIP; }
    }
    // window sums, their identity flags and their bad-scalar flags in ONE buffer: one copy back
    return sl.win.ensure((size_t)SMALL_MSM_W * 4 * C::ABI_W * 4 + 2 * SMALL_MSM_W);
}
// the table of eight multiples per base (unless the handle brought one: `sub`) -> one tree per (super-)window -> the host's Horner fold.  ready_scalars /
// ready_bases queue whatever still has to bring the operands to the device (one-shot calls: the copies and the conversion of the raw points).
template <class C, class HF, class ReadyS, class ReadyB>
int32_t msm_device_small(Slot &sl, const uint32_t *d_bases, const uint32_t *d_scalars, size_t n, uint64_t *out_xyz, ReadyS ready_scalars, ReadyB ready_bases, const SmallSub *sub = nullptr) {
    int32_t rc;
    if ((rc = ws_small<C>(sl, sub ? 1 : n))) return rc;
    hipStream_t s = sl.stream;
    constexpr int c = SMALL_MSM_C;
    const int S = sub ? SMALL_MSM_S : 1, W = SMALL_MSM_W / S;
    if ((rc = ready_scalars(0, 0, n))) return rc;
    if ((rc = ready_bases(0, 0, n))) return rc;
    constexpr size_t WBYTES = (size_t)SMALL_MSM_W * 4 * C::ABI_W * 4;
    static_assert(WBYTES + 2 * SMALL_MSM_W <= Slot::HPIN_BYTES, "pinned scratch");
    // the tree's last block writes the window sums and their flags straight into the slot's pinned host scratch (hipHostMalloc memory is the device's to address): a call of
    // a few hundred terms is one kernel and a synchronisation, the copy kernel that used to bring the 2.6 KB back cost a tenth of it
    uint8_t *const hbuf = (uint8_t *)sl.hpin;
    uint8_t *const wbuf = hbuf;
    uint8_t *const d_inf = wbuf + WBYTES, *const d_bad = d_inf + SMALL_MSM_W;
    if (!sub) {
        StageTimer st(sl, "msm.small_table");
        launch_small_table<C>(s, d_bases, n, sl.bucket.as<uint32_t>(), sl.bucket_inf.as<uint8_t>());
    }
    {
        StageTimer st(sl, "msm.small_tree");
        launch_small_tree<C>(s, sub ? sub->tab : sl.bucket.as<uint32_t>(), sub ? sub->tab_inf : sl.bucket_inf.as<uint8_t>(), S, d_scalars, n, sl.head.as<uint32_t>(), sl.part_inf.as<uint8_t>(),
                             sl.small_cnt.as<uint32_t>(), (uint32_t *)wbuf, d_inf, d_bad);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    if (gs.prof) prof_flush(sl);
    const uint64_t *hwin = (const uint64_t *)hbuf; const uint8_t *hinf = hbuf + WBYTES, *hbad = hinf + SMALL_MSM_W;
    for (int w = 0; w < W; w++) if (hbad[w]) return DGPU_E_BADARG;      // a scalar >= 2^255
    host_fold<HF>(hwin, hinf, W, c, out_xyz);
    return DGPU_OK;
}

// d_bases: prepared records; d_scalars: canonical 8 x u32 per scalar.  Caller holds the slot.
// bases_pending: the base records are written by work that `ready_bases(k, lo, hi)` queues on this stream between the sort and the accumulation
// of range k (one-shot calls: the bases cross PCIe while the scalars are sorted): the sort does not look at them and the accumulation passes over
// identity records itself.  `ready_scalars(k, lo, hi)` is called before the sort of range k.
template <class C, class HF, class ReadyS, class ReadyB>
int32_t msm_device_ranges(Slot &sl, const uint32_t *d_bases, const uint32_t *d_scalars, size_t n, size_t K, uint64_t *out_xyz, bool bases_pending, ReadyS ready_scalars, ReadyB ready_bases, const SmallSub *sub = nullptr) {
    if (n == 0) { write_identity<HF>(out_xyz); return DGPU_OK; }
    if (K == 1 && n <= gs.small_max.load() && n <= SMALL_MSM_MAX_N) return msm_device_small<C, HF>(sl, d_bases, d_scalars, n, out_xyz, ready_scalars, ready_bases, sub);
    PlainGeom g; int32_t rc;
    if ((rc = plain_geometry<C>(n, g))) return rc;
    if ((rc = ws_plain<C>(sl, g))) return rc;
    if (K > 1 && (rc = ws_bucket_sets<C>(sl, g.NB, K))) return rc;
    const int c = g.c, W = g.W; const uint32_t NB = g.NB; const size_t T = g.T; const int CH = g.CH;
    hipStream_t s = sl.stream;
    const uint32_t heavy_thr = 0xffffffffu /* the sweeps flag nothing: k_flag_heavy does, after the scan */, HEAVY_CAP = g.HEAVY_CAP;
    uint32_t *const dyn = sl.dyn.as<uint32_t>();
    const size_t set_words = soa_points(NB) * C::XW;
    HIPCHK(hipMemsetAsync(sl.bucket_inf.p, 1, K * (size_t)NB, s));
    HIPCHK(hipMemsetAsync(sl.flags.p, 0, 4, s));
    for (size_t k = 0; k < K; k++) {
        size_t lo, hi; range_bounds(n, K, k, lo, hi);
        const size_t nk = hi - lo, nk_pad = (nk + 7) & ~(size_t)7;
        if (nk == 0) continue;
        uint32_t *const bucket = sl.bucket.as<uint32_t>() + k * set_words;
        uint8_t *const bucket_inf = sl.bucket_inf.as<uint8_t>() + k * (size_t)NB;
        const uint32_t *const bases_k = d_bases + lo * C::AFF_STRIDE, *const scalars_k = d_scalars + lo * 8;
        if ((rc = ready_scalars(k, lo, hi))) return rc;
        if (g.psort) {
            StageTimer st(sl, "msm.psort");
            PsParams q;
            q.scalars = scalars_k; q.idflag = nullptr /* the accumulation passes over identity records itself */; q.n = nk; q.flag_base = 0;
            q.c = c; q.W = W; q.key_wstride = g.B; q.val_base = 0; q.val_wstride = 0;
            q.part_log = ps_part_log(NB); q.P = (NB + (1u << q.part_log) - 1) >> q.part_log; q.ntiles = (uint32_t)((nk + PS_TILE - 1) / PS_TILE);
            q.bad = sl.flags.as<uint32_t>();
            const uint32_t dyn_args[6] = {(uint32_t)forced_chunk(), g.min_chunk, g.max_chunks, g.lanes_per_chunk, (uint32_t)T, 0u};
            HIPCHK(hipMemsetAsync(sl.heavy.p, 0, 4, s));
            launch_psort(s, q, NB, sl.cnt.as<uint32_t>(), sl.cursor.as<uint32_t>(), sl.bsums.as<uint32_t>(), sl.digits.p, sl.off.as<uint32_t>(), sl.entries.as<uint32_t>(),
                         16u * (uint32_t)CH /* replaced on the device, dyn_chunk.hip.h */, sl.heavy.as<uint32_t>(), HEAVY_CAP, dyn_args, dyn);
        } else {
        {
            StageTimer st(sl, "msm.count");
            HIPCHK(hipMemsetAsync(sl.heavy.p, 0, 4, s));
            launch_digit_codes(s, g.wide, scalars_k, bases_pending ? nullptr : bases_k, C::AFF_STRIDE, C::FLAGW, nk, nk_pad, c, W, sl.digits.p, sl.flags.as<uint32_t>());
            launch_sort_sweep(s, g.wide, false, g.sort_grid, g.lds_bytes, sl.digits.p, nk, nk_pad, W, g.RANGES, g.rb_log, g.B, sl.cnt.as<uint32_t>(), nullptr, nullptr, heavy_thr, sl.heavy.as<uint32_t>(), HEAVY_CAP);
        }
        {
            StageTimer st(sl, "msm.scan");
            launch_scan(s, sl.cnt.as<uint32_t>(), sl.off.as<uint32_t>(), sl.cursor.as<uint32_t>(), sl.bsums.as<uint32_t>(), (size_t)NB);
            launch_dyn_chunk(s, sl.off.as<uint32_t>() + NB, (uint32_t)forced_chunk(), g.min_chunk, g.max_chunks, g.lanes_per_chunk, (uint32_t)T, dyn);
            launch_flag_heavy(s, sl.off.as<uint32_t>(), NB, dyn, sl.heavy.as<uint32_t>(), HEAVY_CAP);
        }
        {
            StageTimer st(sl, "msm.scatter");
            launch_sort_sweep(s, g.wide, true, g.sort_grid, g.lds_bytes, sl.digits.p, nk, nk_pad, W, g.RANGES, g.rb_log, g.B, nullptr, sl.off.as<uint32_t>(), sl.entries.as<uint32_t>(), heavy_thr, sl.heavy.as<uint32_t>(), HEAVY_CAP);
        }
        }
        if ((rc = ready_bases(k, lo, hi))) return rc;
        {
            StageTimer st(sl, "msm.accumulate");
#ifdef DGPU_DEV
            static const uint32_t dbg_mask = getenv("DGPU_DBG_NOGATHER") ? 1023u : 0xffffffffu;   // development experiment (wrong results by design): L2-resident points
#else
            constexpr uint32_t dbg_mask = 0xffffffffu;
#endif
            if (bases_pending || g.psort) launch_accumulate_skip_identity<C>(s, bases_k, sl.entries.as<uint32_t>(), sl.off.as<uint32_t>(), NB, bucket, bucket_inf,
                               sl.head.as<uint32_t>(), sl.tail.as<uint32_t>(), sl.head_b.as<uint32_t>(), sl.tail_b.as<uint32_t>(), sl.part_inf.as<uint8_t>(), T, (uint32_t)CH, dyn, msm::RowMap{});
            else launch_accumulate<C>(s, bases_k, sl.entries.as<uint32_t>(), sl.off.as<uint32_t>(), NB, bucket, bucket_inf,
                               sl.head.as<uint32_t>(), sl.tail.as<uint32_t>(), sl.head_b.as<uint32_t>(), sl.tail_b.as<uint32_t>(), sl.part_inf.as<uint8_t>(), T, (uint32_t)CH, dbg_mask, dyn);
        }
        {
            StageTimer st(sl, "msm.fixup");
            launch_fixup<C>(s, NB, bucket, bucket_inf, sl.head.as<uint32_t>(), sl.tail.as<uint32_t>(),
                               sl.head_b.as<uint32_t>(), sl.tail_b.as<uint32_t>(), sl.part_inf.as<uint8_t>(), T, sl.off.as<uint32_t>(), heavy_thr, dyn);
            launch_fixup_heavy<C>(s, sl.heavy.as<uint32_t>(), HEAVY_CAP, sl.off.as<uint32_t>(), (uint32_t)CH, NB, bucket, bucket_inf,
                               sl.head.as<uint32_t>(), sl.tail.as<uint32_t>(), sl.part_inf.as<uint8_t>(), T, dyn, sl.hpart.as<uint32_t>(), sl.hpart_inf.as<uint8_t>());
            if (k > 0) launch_merge_buckets<C>(s, NB, sl.bucket.as<uint32_t>(), sl.bucket_inf.as<uint8_t>(), bucket, bucket_inf);
        }
    }
    {
        StageTimer st(sl, "msm.reduce");
        launch_reduce_l0<C>(s, (unsigned)g.NG, sl.bucket.as<uint32_t>(), sl.bucket_inf.as<uint8_t>(), NB, g.mshift, sl.l1.as<uint32_t>(), sl.l1_inf.as<uint8_t>());
        launch_reduce_top<C>(s, (unsigned)W, sl.l1.as<uint32_t>(), sl.l1_inf.as<uint8_t>(), g.G, 6 + g.mshift, sl.win.as<uint32_t>(), sl.win_inf.as<uint8_t>(), gs.reduce_lanes.load() == 1 ? 1 : 4);
    }
    HIPCHK(hipGetLastError());
    std::vector<uint64_t> hwin((size_t)W * 2 * C::ABI_W);
    std::vector<uint8_t> hinf(W);
    // The results come back into the slot's PINNED scratch: a copy to pageable memory is synchronous inside hipMemcpyAsync (the runtime waits for the
    // stream, then stages the bytes under a lock of its own) — with several calls in flight their host threads queued up there, and the default bench line
    // fell to half its rate on loaded hosts (round 5: 420 -> 190 - 240 MSM/s with kernels of unchanged length).  Pinned: three asynchronous copies, one wait.
    const size_t wb = (size_t)W * 4 * C::ABI_W * 4, ib = ((size_t)W + 7) & ~(size_t)7;
    static_assert((size_t)64 * 4 * 24 * 4 + 64 + 16 <= Slot::HPIN_BYTES, "pinned scratch");
    std::vector<uint8_t> big;                        // (W <= 37 today, c >= 7: the pinned scratch always fits — a future window rule that breaks this falls back to pageable memory instead of overflowing it)
    if (wb + ib + 4 > Slot::HPIN_BYTES) big.resize(wb + ib + 4);
    uint8_t *const hp = big.empty() ? (uint8_t *)sl.hpin : big.data();
    HIPCHK(hipMemcpyAsync(hp, sl.win.p, wb, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hp + wb, sl.win_inf.p, W, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hp + wb + ib, sl.flags.p, 4, hipMemcpyDeviceToHost, s));
    auto tsync0 = std::chrono::steady_clock::now();
    HIPCHK(hipStreamSynchronize(s));
    auto tsync1 = std::chrono::steady_clock::now();
    if (gs.prof) prof_flush(sl);
    uint32_t hbad; memcpy(&hbad, hp + wb + ib, 4);
    if (hbad) return DGPU_E_BADARG;                  // a scalar >= 2^255 (sort_kernels.hip.h k_digit_codes)
    memcpy(hwin.data(), hp, wb); memcpy(hinf.data(), hp + wb, W);
    host_fold<HF>(hwin.data(), hinf.data(), W, c, out_xyz);
    if (gs.prof) {
        auto t2 = std::chrono::steady_clock::now();
        prof_add_host("msm.host_wait", std::chrono::duration<double, std::milli>(tsync1 - tsync0).count());
        prof_add_host("msm.host_fold", std::chrono::duration<double, std::milli>(t2 - tsync1).count());
    }
    return DGPU_OK;
}
template <class C, class HF>
int32_t msm_device(Slot &sl, const uint32_t *d_bases, const uint32_t *d_scalars, size_t n, uint64_t *out_xyz, const SmallSub *sub = nullptr) {
    auto nothing = [](size_t, size_t, size_t) { return (int32_t)DGPU_OK; };
    return msm_device_ranges<C, HF>(sl, d_bases, d_scalars, n, 1, out_xyz, false, nothing, nothing, sub);
}

// The small-path table of a plain resident handle (Handle::aux): built by the upload of up to 8192 bases (bases_upload, so that the handle's calls
// allocate nothing), by dgpu_bases_precompute_*, or — for a handle that came into being another way — when it meets the small path for the second
// time: one allocation, released with the handle.  Returns false
// when there is none (yet, or because the allocation failed: the call then builds its eight multiples per base itself).  Caller holds a slot.
template <class C>
bool small_sub_for(Slot &sl, uint64_t handle_id, const Handle &h, size_t offset, size_t n, SmallSub &out, bool force = false) {
    if (h.n == 0 || h.n > SMALL_MSM_MAX_N || n > gs.small_max.load()) return false;
    void *aux = nullptr;
    {
        std::lock_guard<std::mutex> lk(gs.mu);
        auto it = gs.handles.find(handle_id);
        if (it == gs.handles.end()) return false;
        aux = it->second.aux;
        if (!aux && !force && ++it->second.small_uses < 2) return false;
    }
    if (!aux) {
        std::lock_guard<std::mutex> build(gs.small_mu);                 // one build at a time; whoever comes second finds the table
        { std::lock_guard<std::mutex> lk(gs.mu); auto it = gs.handles.find(handle_id); if (it == gs.handles.end()) return false; aux = it->second.aux; }
        if (!aux) {
            void *p = nullptr;
            if (dev_malloc(&p, small_sub_bytes<C>(h.n)) != hipSuccess) { (void)hipGetLastError(); return false; }
            {
                StageTimer st(sl, "msm.small_subtable");
                launch_small_subtable<C>(sl.stream, (const uint32_t *)h.p, h.n, (uint32_t *)p, (uint8_t *)p + small_sub_tab_bytes<C>(h.n));
            }
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(sl.stream) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return false; }
            std::lock_guard<std::mutex> lk(gs.mu);
            auto it = gs.handles.find(handle_id);
            if (it == gs.handles.end()) { (void)hipFree(p); return false; }
            it->second.aux = aux = p;
        }
    }
    out = small_sub_at<C>(aux, h.n, offset);
    return true;
}

inline void shard_bounds(size_t n, size_t parts, std::vector<size_t> &lo) {
    lo.resize(parts + 1);
    const size_t base = n / parts, rem = n % parts;
    for (size_t k = 0; k <= parts; k++) lo[k] = k * base + std::min(k, rem);       // contiguous, balanced (== sharded.chunk_bounds)
}
template <class F> int32_t run_shards(size_t parts, F body) { return par_run(parts, body); }
// ---- shared-bucket-set pipeline over a precomputed-multiples table (pre_kernels.hip.h, psort_kernels.hip.h) ------------------------------------

// Window width of a table for n bases (measured, tests/perf/pre_perf.py at 2^16 .. 2^21, profiles/r02a_table_sizes.txt): 20 bits from
// 2^18.3 terms on (2^17.5 until round 4's re-measurement) (W = 13: fewer additions, and runs short enough that few buckets are cut by chunk borders), 16 bits for 2^15 .. 2^18.3
// (W = 16, but one set of 2^15 buckets instead of sixteen: +25 % MSM/s at 2^16), and below 2^15 terms no table at all (0): the plain
// pipeline's narrow windows (c = 8 .. 10) keep the bucket count proportionate to the terms there.
inline int choose_c_pre(size_t n) {
    if (gs.window_bits.load() >= 16 && gs.window_bits.load() <= 22) return gs.window_bits.load();
    if (n < (1u << 15)) return 0;
    return n >= 320000 ? 20 : 16;      // (re-measured in round 4, profiles/r04zz_table_widths.txt: at 2^18 terms width 16 is 1.32 ms against 1.44 for width 20 and equal with four in flight; at 2^19 width 20 wins both ways)
}

// sum_j A_j + 2^lb * sum_j j S_j over the PW pseudo-windows (bucket b = j 2^lb + k of the one bucket set weighs b + 1 = (k + 1) + j 2^lb)
template <class HF>
void host_fold_shared(const uint64_t *a_abi, const uint8_t *a_inf, const uint64_t *s_abi, const uint8_t *s_inf, int PW, int lb, uint64_t *out_xyz) {
    typedef hostf::HXyzz<HF> PT;
    const size_t FWORDS = sizeof(HF) / 8;
    auto load = [&](const uint64_t *src, bool inf) { PT t = PT::identity(); if (!inf) { t.inf = false; memcpy(&t.x, src, sizeof(HF)); memcpy(&t.y, src + FWORDS, sizeof(HF)); memcpy(&t.zz, src + 2 * FWORDS, sizeof(HF)); memcpy(&t.zzz, src + 3 * FWORDS, sizeof(HF)); } return t; };
    PT suffix = PT::identity(), weighted = PT::identity(), total = PT::identity();
    for (int j = PW - 1; j >= 1; j--) { suffix.add_in_place(load(s_abi + (size_t)j * 4 * FWORDS, s_inf[j] != 0)); weighted.add_in_place(suffix); }   // sum_{j>=1} j S_j
    for (int k = 0; k < lb; k++) weighted.dbl_in_place();
    for (int j = 0; j < PW; j++) total.add_in_place(load(a_abi + (size_t)j * 4 * FWORDS, a_inf[j] != 0));
    total.add_in_place(weighted);
    HF X, Y, Z; total.to_normalised_jacobian(X, Y, Z);
    memcpy(out_xyz, &X, sizeof(HF)); memcpy(out_xyz + FWORDS, &Y, sizeof(HF)); memcpy(out_xyz + 2 * FWORDS, &Z, sizeof(HF));
}

// P + 2^shift * sum_t 2^t M_t (reduce_kernels.hip.h): pts[0] = P, pts[1 + t] = M_t, nm marginals
template <class HF>
void host_fold_marginals(const uint64_t *pts, const uint8_t *inf, int nm, int shift, uint64_t *out_xyz) {
    typedef hostf::HXyzz<HF> PT;
    const size_t FWORDS = sizeof(HF) / 8;
    auto load = [&](int i) { PT t = PT::identity(); if (!inf[i]) { const uint64_t *src = pts + (size_t)i * 4 * FWORDS; t.inf = false; memcpy(&t.x, src, sizeof(HF)); memcpy(&t.y, src + FWORDS, sizeof(HF)); memcpy(&t.zz, src + 2 * FWORDS, sizeof(HF)); memcpy(&t.zzz, src + 3 * FWORDS, sizeof(HF)); } return t; };
    PT acc = PT::identity();
    for (int t = nm - 1; t >= 0; t--) { if (!acc.inf) acc.dbl_in_place(); acc.add_in_place(load(1 + t)); }
    if (!acc.inf) for (int k = 0; k < shift; k++) acc.dbl_in_place();
    acc.add_in_place(load(0));
    HF X, Y, Z; acc.to_normalised_jacobian(X, Y, Z);
    memcpy(out_xyz, &X, sizeof(HF)); memcpy(out_xyz + FWORDS, &Y, sizeof(HF)); memcpy(out_xyz + 2 * FWORDS, &Z, sizeof(HF));
}

// ---- the shared-bucket-set pipeline over a precomputed-multiples table, in two stages -------------------------------------------------
// Stage 1 (pre_sort): scalars -> the key-sorted row list `entries` and the bucket offsets `off` (off[NB] = number of pairs).  The result
// depends on the scalars and on the table's SHAPE (c, W, rows, first row) only, not on the curve or on the points: the MSMs of a proof
// that multiply one assignment vector by several tables (A, B in G1, B in G2 of create_proof) can share it (dgpu_scalars_sort_*).
// Stage 2 (pre_tail): accumulate / fix-up / reduce / host fold on any table of that shape.
struct PreGeom { uint32_t NB; int mshift, lb, PW, G; size_t NG, Emax, T; int CH; uint32_t min_chunk, max_chunks, lanes_per_chunk, HEAVY_CAP; };
template <class C> int32_t pre_geometry(const PreTable &pt, size_t n, PreGeom &g) {
    const int c = pt.c, W = pt.W;
    g.NB = 1u << (c - 1);
    // buckets per lane of k_reduce_l0 = 2^mshift: 8 once the bucket set fills the chip with one wave per SIMD (2^19 buckets = 1024 waves, the
    // kernel is work-bound), fewer for small sets, where the serial part of every lane is pure latency (2^15 buckets: 1 per lane, 512 waves)
    g.mshift = g.NB >= (1u << 18) ? 3 : (g.NB >= (1u << 17) ? 2 : (g.NB >= (1u << 16) ? 1 : 0));
    // With three or more calls in flight on the context the reduction takes 16 buckets per lane (512 waves in four-wave blocks = half the CUs, 22 %
    // fewer additions; the other CUs go to the other calls' kernels): 2.51 -> 2.48 ms per MSM with six in flight, and 0.2 ms more for a call that
    // runs alone, which therefore keeps 8 (profiles/r04z_reduce_block_ab.txt).  Same result limb for limb (tests sweep the shift).
    if (g.NB >= (1u << 19) && cur().busy.load() >= 3) g.mshift = 4;      // (G2, four in flight: 6.9 - 7.05 -> 6.8 ms per MSM)
    { const int f = gs.reduce_shift.load(); if (f >= 0 && c - 1 >= 6 + f) g.mshift = f; }
    g.lb = std::min(c - 1, 12 + g.mshift);        // log2 buckets per pseudo-window (64 groups of 64 * 2^mshift buckets)
    g.PW = (int)(g.NB >> g.lb);
    g.G = 1 << (g.lb - 6 - g.mshift);             // groups per pseudo-window (<= 64)
    g.NG = (size_t)g.PW * g.G;
    g.Emax = (size_t)n * W;
    if (g.Emax >= (1ull << 32) || (uint64_t)W * pt.n >= (1ull << 31) || W > PS_MAX_W) return DGPU_E_BADARG;
    g.CH = C::NFP == 2 ? choose_chunk(g.Emax, 32, 150000, 2) : choose_chunk(g.Emax, 16, 300000, 1);     // (sizes the partial slots: the rule WITHOUT the run-length term gives the most chunks any E <= Emax can have)
    g.T = (g.Emax + g.CH - 1) / g.CH;
    g.min_chunk = C::NFP == 2 ? 32u : 16u; g.max_chunks = C::NFP == 2 ? 150000u : 300000u; g.lanes_per_chunk = C::NFP == 2 ? 2u : 1u;
    g.HEAVY_CAP = (uint32_t)(g.Emax / (16u * 16u)) + 1;         // (see plain_geometry)
    return DGPU_OK;
}
// grow-only workspace of one slot for the table pipeline (sort + tail) of n terms on a table of pt's shape
template <class C> int32_t ws_pre(Slot &sl, const PreTable &pt, const PreGeom &g, size_t n) {
    const uint32_t NB = g.NB; const size_t T = g.T; const int PW = g.PW;
    const uint32_t P = (NB + (1u << ps_part_log(NB)) - 1) >> ps_part_log(NB);
    const size_t n1 = (size_t)P * ((n + PS_TILE - 1) / PS_TILE);
    int32_t rc;
    if ((rc = sl.flags.ensure(64))) return rc;
    if ((rc = sl.cnt.ensure((n1 + 1) * 4))) return rc;
    if ((rc = sl.cursor.ensure((n1 + 1) * 4))) return rc;                 // off1
    if ((rc = sl.bsums.ensure((scan_blocks(n1) + 2) * 4))) return rc;
    if ((rc = sl.digits.ensure(g.Emax * 8))) return rc;                   // (key, val) pairs
    if ((rc = sl.heavy.ensure(((size_t)g.HEAVY_CAP + 1) * 4))) return rc;
    if ((rc = sl.off.ensure(((size_t)NB + 1) * 4))) return rc;
    if ((rc = sl.entries.ensure(g.Emax * 4))) return rc;
    if ((rc = sl.bucket.ensure(soa_points(NB) * C::XW * 4))) return rc;
    if ((rc = sl.bucket_inf.ensure(NB))) return rc;
    if ((rc = sl.head.ensure(soa_points(T) * C::XW * 4))) return rc;
    if ((rc = sl.tail.ensure(soa_points(T) * C::XW * 4))) return rc;
    if ((rc = sl.head_b.ensure(T * 4))) return rc;
    if ((rc = sl.tail_b.ensure(T * 4))) return rc;
    if ((rc = sl.part_inf.ensure(T * 2))) return rc;
    // (sized for the reduction's shape of a call that runs alone as well: pre_geometry picks fewer, longer groups when the context is busy, and a
    //  slot reserved under load must not allocate when it is later used by a lone call)
    const size_t NGw = std::max(g.NG, (size_t)(NB >> 9)), PWw = std::max((size_t)PW, (size_t)(NB >> 15));
    const size_t mpts = reduce_marginals_points<C>(NGw);                            // class buffers of the marginal reduction (reduce_kernels.hip.h)
    if ((rc = sl.l1.ensure(std::max(NGw * 2, mpts) * C::XW * 4))) return rc;
    if ((rc = sl.l1_inf.ensure(std::max(NGw * 2, mpts)))) return rc;
    if ((rc = sl.win.ensure(std::max((size_t)2 * PWw, (size_t)32) * 4 * C::ABI_W * 4))) return rc;        // A_j then S_j; or P and the marginals
    if ((rc = sl.win_inf.ensure(std::max((size_t)2 * PWw, (size_t)32)))) return rc;
    if ((rc = sl.dyn.ensure(msm::dyn_words(T) * 4))) return rc;
    { const size_t hslots = 2 * (T / msm::HEAVY_RANGE + 2); if ((rc = sl.hpart.ensure(hslots * C::XW * 4))) return rc; if ((rc = sl.hpart_inf.ensure(hslots))) return rc; }
    (void)pt;
    return DGPU_OK;
}
// sizes of `off` / `entries` for a sort kept outside a slot
// a table's allocation: W rows of n prepared records, then one identity-flag byte per base (what the sort reads instead of the records' flag words)
template <class C> inline size_t pre_rows_bytes(size_t n, int W) { return (size_t)W * n * C::AFF_STRIDE * 4; }
template <class C> inline size_t pre_tab_bytes(size_t n, int W) { return pre_rows_bytes<C>(n, W) + ((n + 15) & ~(size_t)15); }
template <class C> inline const uint8_t *pre_idflags(const PreTable &pt) { return (const uint8_t *)pt.tab + pre_rows_bytes<C>(pt.n, pt.W); }
inline size_t pre_off_bytes(const PreTable &pt) { return (((size_t)1 << (pt.c - 1)) + 1) * 4; }
inline size_t pre_entries_bytes(const PreTable &pt, size_t n) { return (size_t)n * pt.W * 4; }

// dyn != nullptr: the chunking and the heavy-bucket list of the accumulation are produced on the way (curve-specific: dyn_args); nullptr: sort only
template <class C>
int32_t pre_sort(Slot &sl, const PreTable &pt, const PreGeom &g, size_t boff, const uint32_t *d_scalars, size_t n, uint32_t *off, uint32_t *entries, uint32_t *dyn, bool reset_flag = true) {
    PsParams q;
    q.scalars = d_scalars; q.idflag = dyn ? pre_idflags<C>(pt) : nullptr /* shared sort: no per-table identity filter */; q.n = n; q.flag_base = (uint32_t)boff;
    q.c = pt.c; q.W = pt.W; q.key_wstride = 0; q.val_base = (uint32_t)boff; q.val_wstride = (uint32_t)pt.n;
    q.part_log = ps_part_log(g.NB); q.P = (g.NB + (1u << q.part_log) - 1) >> q.part_log; q.ntiles = (uint32_t)((n + PS_TILE - 1) / PS_TILE);
    int32_t rc;
    if ((rc = ws_pre<C>(sl, pt, g, n))) return rc;
    q.bad = sl.flags.as<uint32_t>();
    const uint32_t heavy_thr = dyn ? 16u * (uint32_t)g.CH /* replaced on the device, dyn_chunk.hip.h */ : 0xffffffffu /* nothing flagged */;
    const uint32_t dyn_args[6] = {(uint32_t)forced_chunk(), g.min_chunk, g.max_chunks, g.lanes_per_chunk, (uint32_t)g.T, g.NB};
    StageTimer st(sl, "msm.psort");
    HIPCHK(hipMemsetAsync(sl.heavy.p, 0, 4, sl.stream));
    if (reset_flag) HIPCHK(hipMemsetAsync(sl.flags.p, 0, 4, sl.stream));
    launch_psort(sl.stream, q, g.NB, sl.cnt.as<uint32_t>(), sl.cursor.as<uint32_t>(), sl.bsums.as<uint32_t>(), sl.digits.p, off, entries,
                 heavy_thr, sl.heavy.as<uint32_t>(), g.HEAVY_CAP, dyn_args, dyn);
    return DGPU_OK;
}
// Stage 2a (pre_acc): accumulate + fix-up of one sorted list into bucket set `set` (merged into set 0 when set > 0).
// derive_dyn: `off` / `entries` come from a shared sort — chunking and heavy-bucket list are derived here from off[] (as the plain pipeline does)
template <class C>
int32_t pre_acc(Slot &sl, const PreTable &pt, const PreGeom &g, const uint32_t *off, const uint32_t *entries, bool derive_dyn, const msm::RowMap &map, size_t set) {
    const uint32_t NB = g.NB; const size_t T = g.T;
    uint32_t *const dyn = sl.dyn.as<uint32_t>();
    const uint32_t heavy_thr = 16u * (uint32_t)g.CH;      // (replaced on the device by dyn[])
    hipStream_t s = sl.stream;
    uint32_t *const bucket = sl.bucket.as<uint32_t>() + set * soa_points(NB) * C::XW;
    uint8_t *const bucket_inf = sl.bucket_inf.as<uint8_t>() + set * (size_t)NB;
    HIPCHK(hipMemsetAsync(bucket_inf, 1, NB, s));
    if (derive_dyn) {
        HIPCHK(hipMemsetAsync(sl.heavy.p, 0, 4, s));
        launch_dyn_chunk(s, off + NB, (uint32_t)forced_chunk(), g.min_chunk, g.max_chunks, g.lanes_per_chunk, (uint32_t)T, dyn, NB);
        launch_flag_heavy(s, off, NB, dyn, sl.heavy.as<uint32_t>(), g.HEAVY_CAP);
    }
    {
        StageTimer st(sl, "msm.accumulate");
        if (derive_dyn) launch_accumulate_skip_identity<C>(s, (const uint32_t *)pt.tab, entries, off, NB, bucket, bucket_inf,
                           sl.head.as<uint32_t>(), sl.tail.as<uint32_t>(), sl.head_b.as<uint32_t>(), sl.tail_b.as<uint32_t>(), sl.part_inf.as<uint8_t>(), T, (uint32_t)g.CH, dyn, map);
        else launch_accumulate<C>(s, (const uint32_t *)pt.tab, entries, off, NB, bucket, bucket_inf,
                           sl.head.as<uint32_t>(), sl.tail.as<uint32_t>(), sl.head_b.as<uint32_t>(), sl.tail_b.as<uint32_t>(), sl.part_inf.as<uint8_t>(), T, (uint32_t)g.CH, 0xffffffffu, dyn);
    }
    {
        StageTimer st(sl, "msm.fixup");
        launch_fixup<C>(s, NB, bucket, bucket_inf, sl.head.as<uint32_t>(), sl.tail.as<uint32_t>(),
                           sl.head_b.as<uint32_t>(), sl.tail_b.as<uint32_t>(), sl.part_inf.as<uint8_t>(), T, off, heavy_thr, dyn);
        launch_fixup_heavy<C>(s, sl.heavy.as<uint32_t>(), g.HEAVY_CAP, off, (uint32_t)g.CH, NB, bucket, bucket_inf,
                           sl.head.as<uint32_t>(), sl.tail.as<uint32_t>(), sl.part_inf.as<uint8_t>(), T, dyn, sl.hpart.as<uint32_t>(), sl.hpart_inf.as<uint8_t>());
        if (set > 0) launch_merge_buckets<C>(s, NB, sl.bucket.as<uint32_t>(), sl.bucket_inf.as<uint8_t>(), bucket, bucket_inf);
    }
    return DGPU_OK;
}
// Stage 2b (pre_finish): reduction of bucket set 0, read-back, host fold.  check_flag: the slot's flag word (a scalar >= 2^255 seen by this call's sort)
template <class C, class HF>
int32_t pre_finish(Slot &sl, const PreGeom &g, bool check_flag, uint64_t *out_xyz) {
    const uint32_t NB = g.NB; const int PW = g.PW;
    hipStream_t s = sl.stream;
    uint32_t *win_a = sl.win.as<uint32_t>(), *win_s = win_a + (size_t)PW * 4 * C::ABI_W;
    uint8_t *inf_a = sl.win_inf.as<uint8_t>(), *inf_s = inf_a + PW;
    // reduce_lanes 0 (default): bit marginals (reduce_kernels.hip.h); 1 / 4: the scan form (k_reduce_l0 + k_reduce_top / _quad), kept for the comparison tests
    const int rl = gs.reduce_lanes.load();
    const bool marginals = (rl == 0 || rl == 2) && g.NG >= 2;        // (2: the class folds with one lane per value, for the comparison)
    int nm = 0;
    {
        StageTimer st(sl, "msm.reduce");
        if (marginals) nm = launch_reduce_marginals<C>(s, (unsigned)g.NG, sl.bucket.as<uint32_t>(), sl.bucket_inf.as<uint8_t>(), NB, g.mshift, sl.l1.as<uint32_t>(), sl.l1_inf.as<uint8_t>(), win_a, inf_a, rl == 0);
        else {
            launch_reduce_l0<C>(s, (unsigned)g.NG, sl.bucket.as<uint32_t>(), sl.bucket_inf.as<uint8_t>(), NB, g.mshift, sl.l1.as<uint32_t>(), sl.l1_inf.as<uint8_t>());
            launch_reduce_top_s<C>(s, (unsigned)PW, sl.l1.as<uint32_t>(), sl.l1_inf.as<uint8_t>(), g.G, 6 + g.mshift, win_a, inf_a, win_s, inf_s, gs.reduce_lanes.load() == 1 ? 1 : 4);
        }
    }
    HIPCHK(hipGetLastError());
    const size_t npts = marginals ? (size_t)nm + 1 : (size_t)2 * PW;
    std::vector<uint64_t> hwin(npts * 2 * C::ABI_W);
    std::vector<uint8_t> hinf(npts);
    // (into the slot's pinned scratch: see the plain pipeline's read-back above)
    const size_t wb = npts * 4 * C::ABI_W * 4, ib = (npts + 7) & ~(size_t)7;
    std::vector<uint8_t> big;                        // (a forced geometry with hundreds of pseudo-windows — a knob of the twin — does not fit the pinned scratch: pageable then)
    if (wb + ib + 4 > Slot::HPIN_BYTES) big.resize(wb + ib + 4);
    uint8_t *const hp = big.empty() ? (uint8_t *)sl.hpin : big.data();
    memset(hp + wb + ib, 0, 4);
    HIPCHK(hipMemcpyAsync(hp, sl.win.p, wb, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hp + wb, sl.win_inf.p, npts, hipMemcpyDeviceToHost, s));
    if (check_flag) HIPCHK(hipMemcpyAsync(hp + wb + ib, sl.flags.p, 4, hipMemcpyDeviceToHost, s));      // (a shared sort was checked by dgpu_scalars_sort)
    auto tsync0 = std::chrono::steady_clock::now();
    HIPCHK(hipStreamSynchronize(s));
    auto tsync1 = std::chrono::steady_clock::now();
    if (gs.prof) prof_flush(sl);
    uint32_t hbad; memcpy(&hbad, hp + wb + ib, 4);
    if (hbad) return DGPU_E_BADARG;                  // a scalar >= 2^255 (sort_kernels.hip.h k_digit_codes)
    memcpy(hwin.data(), hp, wb); memcpy(hinf.data(), hp + wb, npts);
    if (marginals) host_fold_marginals<HF>(hwin.data(), hinf.data(), nm, g.mshift + (C::NFP == 2 ? 1 : 0), out_xyz);
    else host_fold_shared<HF>(hwin.data(), hinf.data(), hwin.data() + (size_t)PW * 2 * C::ABI_W, hinf.data() + PW, PW, g.lb, out_xyz);
    if (gs.prof) {
        auto t2 = std::chrono::steady_clock::now();
        prof_add_host("msm.host_wait", std::chrono::duration<double, std::milli>(tsync1 - tsync0).count());
        prof_add_host("msm.host_fold", std::chrono::duration<double, std::milli>(t2 - tsync1).count());
    }
    return DGPU_OK;
}
template <class C, class HF>
int32_t pre_tail(Slot &sl, const PreTable &pt, const PreGeom &g, const uint32_t *off, const uint32_t *entries, bool derive_dyn, uint64_t *out_xyz, const msm::RowMap &map = msm::RowMap{}) {
    int32_t rc;
    if ((rc = ws_pre<C>(sl, pt, g, g.Emax / pt.W))) return rc;
    if ((rc = pre_acc<C>(sl, pt, g, off, entries, derive_dyn, map, 0))) return rc;
    return pre_finish<C, HF>(sl, g, !derive_dyn, out_xyz);
}
// d_scalars: canonical 8 x u32 per scalar; terms i < n use table rows at column boff + i.  Caller holds the slot.  K > 1: the scalars are still
// crossing PCIe and are taken in K ranges (see msm_device_ranges); `ready_scalars(k, lo, hi)` queues the wait for range k.
template <class C, class HF, class ReadyS>
int32_t msm_device_pre_ranges(Slot &sl, const PreTable &pt, size_t boff, const uint32_t *d_scalars, size_t n, size_t K, uint64_t *out_xyz, ReadyS ready_scalars) {
    if (n == 0) { write_identity<HF>(out_xyz); return DGPU_OK; }
    PreGeom g; int32_t rc;
    if ((rc = pre_geometry<C>(pt, n, g))) return rc;
    if ((rc = ws_pre<C>(sl, pt, g, n))) return rc;
    if (K > 1 && (rc = ws_bucket_sets<C>(sl, g.NB, K))) return rc;
    HIPCHK(hipMemsetAsync(sl.flags.p, 0, 4, sl.stream));
    for (size_t k = 0; k < K; k++) {
        size_t lo, hi; range_bounds(n, K, k, lo, hi);
        if (hi == lo) continue;
        if ((rc = ready_scalars(k, lo, hi))) return rc;
        if ((rc = pre_sort<C>(sl, pt, g, boff + lo, d_scalars + lo * 8, hi - lo, sl.off.as<uint32_t>(), sl.entries.as<uint32_t>(), sl.dyn.as<uint32_t>(), false))) return rc;
        if ((rc = pre_acc<C>(sl, pt, g, sl.off.as<uint32_t>(), sl.entries.as<uint32_t>(), false, msm::RowMap{}, k))) return rc;
    }
    return pre_finish<C, HF>(sl, g, true, out_xyz);
}
template <class C, class HF>
int32_t msm_device_pre(Slot &sl, const PreTable &pt, size_t boff, const uint32_t *d_scalars, size_t n, uint64_t *out_xyz) {
    return msm_device_pre_ranges<C, HF>(sl, pt, boff, d_scalars, n, 1, out_xyz, [](size_t, size_t, size_t) { return (int32_t)DGPU_OK; });
}

// ---- one sort for several tables (dgpu_scalars_sort / dgpu_msm_*_sorted) ---------------------------------------------------------------
// A, B-in-G1 and B-in-G2 of a LegoGroth16 proof multiply the SAME assignment by three proving-key queries of equal length
// (legogroth16/src/prover.rs:325-344 -> calculate_coeff :585-594): with the queries held as tables of one shape their partition sorts are
// identical.  The sorted list lives in a handle of its own (kind 12, buffers from the scalar pool), read-only while MSMs use it.
inline int32_t scalars_sort(uint64_t table, size_t boff, uint64_t scalars, size_t soff, size_t n, uint64_t *sorted) {
    if (!sorted || n == 0) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    HandleRef hb(table), hs(scalars);
    if (!hb.ok || !hs.ok || (hb.h.kind != 10 && hb.h.kind != 11) || hs.h.kind != 3 || hb.h.ctx != hs.h.ctx) return DGPU_E_BADARG;
    if (boff > hb.h.n || n > hb.h.n - boff || soff > hs.h.n || n > hs.h.n - soff) return DGPU_E_BADARG;
    CtxScope on_owner(hb.h.ctx);
    SLOT_ACQUIRE(L, sl);
    HIPCHK(hipSetDevice(cur().device));
    const PreTable &pt = *(const PreTable *)hb.h.p;
    PreGeom g; int32_t rc;
    if ((rc = pre_geometry<G1>(pt, n, g))) return rc;                     // (only the curve-independent fields are used by the sort)
    SortedScalars *ss = new SortedScalars{nullptr, nullptr, pre_off_bytes(pt), pre_entries_bytes(pt, n), n, pt.n, boff, pt.c, pt.W};
    ss->off = scalar_alloc(ss->off_bytes); ss->entries = scalar_alloc(ss->entries_bytes);
    auto drop = [&]() { scalar_release(cur_index(), ss->off, ss->off_bytes); scalar_release(cur_index(), ss->entries, ss->entries_bytes); delete ss; };
    if (!ss->off || !ss->entries) { drop(); return DGPU_E_OOM; }
    rc = pre_sort<G1>(sl, pt, g, boff, (const uint32_t *)hs.h.p + soff * 8, n, (uint32_t *)ss->off, (uint32_t *)ss->entries, nullptr);
    uint32_t hbad = 0;
    if (!rc && hipMemcpyAsync(&hbad, sl.flags.p, 4, hipMemcpyDeviceToHost, sl.stream) != hipSuccess) rc = DGPU_E_HIP;
    if (!rc && hipStreamSynchronize(sl.stream) != hipSuccess) { gs.last_hip = (int32_t)hipGetLastError(); rc = DGPU_E_HIP; }
    if (gs.prof) prof_flush(sl);
    if (!rc && hbad) rc = DGPU_E_BADARG;             // a scalar >= 2^255
    if (rc) { drop(); return rc; }
    *sorted = register_handle(ss, n, 12);
    return DGPU_OK;
}
// row_shift > 0: the table holds the points of rows row_shift .. rows - 1 of the shape the list was sorted for (RowMap, msm_kernels.hip.h)
template <class C, class HF>
int32_t msm_sorted(uint64_t table, uint64_t sorted, size_t row_shift, uint64_t *out, int kind) {
    if (!out) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    HandleRef hb(table), hs(sorted);
    if (!hb.ok || !hs.ok || hb.h.kind != kind + 9 || hs.h.kind != 12 || hb.h.ctx != hs.h.ctx) return DGPU_E_BADARG;
    const PreTable &pt = *(const PreTable *)hb.h.p;
    const SortedScalars &ss = *(const SortedScalars *)hs.h.p;
    if (pt.c != ss.c || pt.W != ss.W || row_shift >= ss.rows || pt.n != ss.rows - row_shift) return DGPU_E_BADARG;      // not the shape the list was sorted for
    CtxScope on_owner(hb.h.ctx);
    SLOT_ACQUIRE(L, sl);
    HIPCHK(hipSetDevice(cur().device));
    PreGeom g; int32_t rc;
    if ((rc = pre_geometry<C>(pt, ss.n, g))) return rc;
    msm::RowMap map;
    if (row_shift) {
        map.rows_src = (uint32_t)ss.rows; map.rows_dst = (uint32_t)pt.n; map.shift = (uint32_t)row_shift;
        map.magic = ~(uint64_t)0 / ss.rows + 1;           // floor((2^64 - 1) / rows) + 1 >= 2^64 / rows, off by < 1: the quotient of any row < 2^32 is exact
    }
    return pre_tail<C, HF>(sl, pt, g, (const uint32_t *)ss.off, (const uint32_t *)ss.entries, true, out, map);
}

template <class C> int32_t reserve_slots(int what, size_t n, size_t stride, const PreTable *pt);      // (defined below)

// In-place: the bases behind `handle` (kind 1 / 2, or every part of a sharded handle 7 / 8) become precomputed-multiples tables.
// The handle keeps its id; while the table is being built other calls on it fail with DGPU_E_BADARG.
template <class C>
int32_t bases_precompute(uint64_t handle, int32_t window_bits, int kind /* 1 | 2 */) {
    if (window_bits != 0 && (window_bits < 16 || window_bits > 22)) return DGPU_E_BADARG;
    Handle hd;
    { Handle peek; if (!lookup_handle(handle, peek)) return DGPU_E_BADARG;
      if (peek.kind == kind + 6) {                       // sharded: convert every part
          HandleRef ref(handle); if (!ref.ok) return DGPU_E_BADARG;
          const ShardSet &ss = *(const ShardSet *)ref.h.p;
          return run_shards(ss.sub.size(), [&](size_t k) { return bases_precompute<C>(ss.sub[k], window_bits, kind); });
      }
      if (peek.kind == kind + 9) return DGPU_OK; }       // already a table
    if (!take_handle(handle, [kind](int k) { return k == kind; }, hd)) return DGPU_E_BADARG;
    // (a handle that stays plain keeps its small-MSM table; one that becomes a bucket table drops it)
    auto put_back = [&](void *p, int k) { std::lock_guard<std::mutex> lk(gs.mu); gs.handles[handle] = Handle{p, hd.n, k, hd.ctx, 0, k == kind ? hd.aux : nullptr, hd.small_uses}; };
    CtxScope on_owner(hd.ctx);
    if (!cur().ready) { put_back(hd.p, kind); return DGPU_E_NODEVICE; }
    const size_t n = hd.n;
    const int c = window_bits ? window_bits : choose_c_pre(n);
    if (c == 0) {                                                   // too few terms for a bucket table to pay: the handle stays plain ...
        put_back(hd.p, kind);
        if (n && n <= SMALL_MSM_MAX_N) {                            // ... and gets the small path's table now instead of at its second small call
            HandleRef ref(handle);
            if (ref.ok) { SlotLock L; if (L.ok && hipSetDevice(cur().device) == hipSuccess) { SmallSub sub; (void)small_sub_for<C>(*L.s, handle, ref.h, 0, 0, sub, true); } }
        }
        return DGPU_OK;
    }
    const int W = 255 / c + 1;
    if (n == 0 || (uint64_t)W * n >= (1ull << 31)) { put_back(hd.p, kind); return n == 0 ? DGPU_OK : DGPU_E_BADARG; }
    void *tab = nullptr, *tmp = nullptr;
    int32_t rc = DGPU_OK;
    {
        SLOT_ACQUIRE(L, sl);
        if (hipSetDevice(cur().device) != hipSuccess) rc = DGPU_E_HIP;
        const size_t rec = (size_t)C::AFF_STRIDE * 4;
        if (!rc && dev_malloc(&tab, pre_tab_bytes<C>(n, W)) != hipSuccess) { (void)hipGetLastError(); rc = DGPU_E_OOM; }
        if (!rc && dev_malloc(&tmp, n * (size_t)C::XW * 4) != hipSuccess) { (void)hipGetLastError(); rc = DGPU_E_OOM; }
        if (!rc && hipMemcpyAsync(tab, hd.p, n * rec, hipMemcpyDeviceToDevice, sl.stream) != hipSuccess) rc = DGPU_E_HIP;
        if (!rc) {
            StageTimer st(sl, "msm.precompute");
            for (int w = 1; w < W; w++) launch_pre_step<C>(sl.stream, (const uint32_t *)tab + (size_t)(w - 1) * n * C::AFF_STRIDE, n, c, (uint32_t *)tmp, (uint32_t *)tab + (size_t)w * n * C::AFF_STRIDE);
            launch_id_flags(sl.stream, (const uint32_t *)tab, C::AFF_STRIDE, C::FLAGW, n, (uint8_t *)tab + pre_rows_bytes<C>(n, W));
        }
        if (!rc && (hipGetLastError() != hipSuccess || hipStreamSynchronize(sl.stream) != hipSuccess)) rc = DGPU_E_HIP;
        if (gs.prof) prof_flush(sl);
        if (tmp) (void)hipFree(tmp);
    }
    if (rc) { if (tab) (void)hipFree(tab); (void)hipGetLastError(); put_back(hd.p, kind); return rc; }
    (void)hipFree(hd.p);
    if (hd.aux) (void)hipFree(hd.aux);
    PreTable *npt = new PreTable{tab, n, c, W};
    put_back(npt, kind + 9);
    (void)reserve_slots<C>(3, n, 0, npt);           // (the handle cannot be freed under us: the caller still owns it)
    return DGPU_OK;
}

// The bases as the caller holds them: `stride` bytes per point, x at x_off and y at y_off (C::NFP x 48 bytes of Montgomery limbs each, 8-byte
// aligned), identity flags as a byte inside the point (inf_off != NO_INF_OFF) and / or as a separate array (is_inf).  packed() = the ABI's own
// layout (x then y, 96 / 192 bytes per point); the strided entry points describe ark-ec's in-memory `Affine { x, y, infinity }`.
struct RawBases {
    const uint8_t *p; size_t stride, x_off, y_off, inf_off; const uint8_t *is_inf;
    template <class C> static RawBases packed(const uint64_t *xy, const uint8_t *is_inf) {
        return RawBases{(const uint8_t *)xy, (size_t)2 * C::ABI_W * 4, 0, (size_t)C::ABI_W * 4, msm::NO_INF_OFF, is_inf};
    }
    template <class C> bool ok() const {
        const size_t fb = (size_t)C::ABI_W * 4;
        return stride >= 2 * fb && stride % 8 == 0 && x_off % 8 == 0 && y_off % 8 == 0 && x_off + fb <= stride && y_off + fb <= stride &&
               (inf_off == msm::NO_INF_OFF || inf_off < stride) && ((uintptr_t)p % 8) == 0 && stride <= 4096;
    }
};
constexpr size_t STAGE_CHUNK_BYTES = (size_t)16 << 20;      // H2D piece after which the conversion kernel of that piece may start

template <class C> int32_t ws_stage_bases(Slot &sl, const RawBases &rb, size_t n) {
    int32_t rc;
    if ((rc = sl.in_bases.ensure(n * rb.stride + 16))) return rc;
    if (rb.is_inf && (rc = sl.in_inf.ensure(n + 16))) return rc;
    return DGPU_OK;
}
// Queue: the raw points cross PCIe in pieces on the slot's COPY stream; the conversion of piece k to prepared records (k_prep_bases_raw) runs on
// the compute stream as soon as that piece has landed, i.e. under the copy of piece k + 1 and under whatever the compute stream was doing
// before (a one-shot MSM sorts its scalars meanwhile).  Pageable host memory: hipMemcpyAsync returns when the piece is on its way.
// (points [lo, hi) of the caller's array into records [lo, hi) of d_out; every piece takes the next of the slot's copy events — an event may be
// recorded again once a wait on its previous record has been queued)
template <class C>
int32_t stage_bases(Slot &sl, const RawBases &rb, size_t lo0, size_t hi0, uint32_t *d_out) {
    if (hi0 <= lo0) return DGPU_OK;
    const size_t n = hi0 - lo0;
    uint8_t *draw = sl.in_bases.as<uint8_t>();
    const uint8_t *dinf = nullptr;
    if (rb.is_inf) { HIPCHK(hipMemcpyAsync(sl.in_inf.as<uint8_t>() + lo0, rb.is_inf + lo0, n, hipMemcpyHostToDevice, sl.cstream)); dinf = sl.in_inf.as<uint8_t>(); }
    const size_t per = std::max<size_t>(1, STAGE_CHUNK_BYTES / rb.stride);
    size_t pieces = (n + per - 1) / per;
    if (pieces > 6) pieces = 6;
    const size_t len = (n + pieces - 1) / pieces;
    StageTimer st(sl, "msm.prep_bases");
    for (size_t lo = lo0; lo < hi0; lo += len) {
        const size_t cnt = std::min(len, hi0 - lo);
        hipEvent_t ev = sl.copy_ev[sl.ev_next++ % (Slot::N_COPY_EV + 1)];
        HIPCHK(hipMemcpyAsync(draw + lo * rb.stride, rb.p + lo * rb.stride, cnt * rb.stride, hipMemcpyHostToDevice, sl.cstream));
        HIPCHK(hipEventRecord(ev, sl.cstream));
        HIPCHK(hipStreamWaitEvent(sl.stream, ev, 0));
        launch_prep_bases_raw<C>(sl.stream, draw + lo * rb.stride, rb.stride, rb.x_off, rb.y_off, rb.inf_off, dinf ? dinf + lo : nullptr, cnt, d_out + lo * C::AFF_STRIDE);
    }
    return DGPU_OK;
}
template <class C> int32_t stage_bases(Slot &sl, const RawBases &rb, size_t n, uint32_t *d_out) { return stage_bases<C>(sl, rb, 0, n, d_out); }
// canonical scalars in d_out once the compute stream gets there (no host wait: the pipeline that follows is queued behind it)
inline int32_t stage_scalars(Slot &sl, const uint64_t *h, size_t lo, size_t hi, bool mont, uint32_t *d_out) {      // scalars [lo, hi)
    if (hi <= lo) return DGPU_OK;
    hipEvent_t ev = sl.copy_ev[sl.ev_next++ % (Slot::N_COPY_EV + 1)];
    HIPCHK(hipMemcpyAsync(d_out + lo * 8, h + lo * 4, (hi - lo) * 32, hipMemcpyHostToDevice, sl.cstream));
    HIPCHK(hipEventRecord(ev, sl.cstream));
    HIPCHK(hipStreamWaitEvent(sl.stream, ev, 0));
    if (mont) ntt::launch_fr_mont_to_canonical(sl.stream, d_out + lo * 8, hi - lo);     // Fr::into_bigint on the device (ark-ec msm_unchecked does it on rayon)
    return DGPU_OK;
}
inline int32_t stage_scalars(Slot &sl, const uint64_t *h, size_t n, bool mont, uint32_t *d_out) { return stage_scalars(sl, h, 0, n, mont, d_out); }

// ---- workspaces sized ahead of the calls (no hipMalloc on an MSM path in steady state) ---------------------------------------------------
// what: 1 = one-shot call of n terms (raw bases of `stride` bytes + scalars + prepared records + plain pipeline), 2 = fresh host scalars on a
// plain handle of n terms, 3 = the same on a table (pt)
template <class C> int32_t ws_for(Slot &sl, int what, size_t n, size_t stride, const PreTable *pt) {
    if (n == 0) return DGPU_OK;
    int32_t rc;
    if ((rc = sl.in_scalars.ensure(n * 32))) return rc;
    if (what == 1) {
        if ((rc = sl.in_bases.ensure(n * stride + 16))) return rc;
        if ((rc = sl.in_inf.ensure(n + 16))) return rc;
        if ((rc = sl.prepped.ensure(n * C::AFF_STRIDE * 4))) return rc;
    }
    // (the bucket sets of the range-wise form, which calls with operands still in flight take from n = 2^19 on)
    if (what == 3) { PreGeom g; if ((rc = pre_geometry<C>(*pt, n, g))) return rc; if ((rc = ws_pre<C>(sl, *pt, g, n))) return rc; return ws_bucket_sets<C>(sl, g.NB, range_count(n, true)); }
    if (n <= SMALL_MSM_MAX_N && (rc = ws_small<C>(sl, n))) return rc;     // (both paths: dgpu_set_small_msm_max may switch between them)
    PlainGeom g; if ((rc = plain_geometry<C>(n, g))) return rc;
    if ((rc = ws_plain<C>(sl, g))) return rc;
    return ws_bucket_sets<C>(sl, g.NB, range_count(n, true));
}
// every slot of the current context (the caller holds none of them); a failure (out of memory) is not an error of the call that triggered
// the reservation: the slot grows on its first use instead
template <class C> int32_t reserve_slots(int what, size_t n, size_t stride, const PreTable *pt) {
    Ctx &cx = cur();
    if (!cx.ready) return DGPU_E_NODEVICE;
    int32_t first = DGPU_OK;
    for (int k = 0; k < N_SLOTS; k++) {
        std::lock_guard<std::mutex> lk(cx.slots[k].mu);
        if (hipSetDevice(cx.device) != hipSuccess) return DGPU_E_HIP;
        const int32_t rc = ws_for<C>(cx.slots[k], what, n, stride, pt);
        if (rc && !first) first = rc;
    }
    return first;
}
// the same for the slots that are idle right now (called by a one-shot call that had to grow its own slot: the other host threads of the
// caller — rayon workers — will come with the same size next, and would each stall the device in hipMalloc / hipFree)
template <class C> void reserve_idle_slots(const Slot *mine, int what, size_t n, size_t stride) {
    Ctx &cx = cur();
    for (int k = 0; k < N_SLOTS; k++) {
        Slot &o = cx.slots[k];
        if (&o == mine || !o.mu.try_lock()) continue;
        (void)ws_for<C>(o, what, n, stride, nullptr);
        o.mu.unlock();
    }
}

// one-shot MSM on the calling thread's context (no size threshold: the callers apply it)
template <class C, class HF>
int32_t msm_oneshot_here(const RawBases &rb, const uint64_t *scalars, size_t n, bool mont, uint64_t *out) {
    if (!cur().ready) return DGPU_E_NODEVICE;
    SLOT_ACQUIRE(L, sl);
    HIPCHK(hipSetDevice(cur().device));
    if (n == 0) { write_identity<HF>(out); return DGPU_OK; }
    int32_t rc;
    const uint64_t allocs0 = g_dev_allocs.load();
    if ((rc = ws_for<C>(sl, 1, n, rb.stride, nullptr))) return rc;
    const size_t K = range_count(n, true);
    const bool grew = g_dev_allocs.load() != allocs0;
    // per range: the scalars first (their digits are sorted while the range's bases, 3/4 of its bytes, are still crossing PCIe); a range's
    // kernels run under the next range's copies
    rc = msm_device_ranges<C, HF>(sl, sl.prepped.as<uint32_t>(), sl.in_scalars.as<uint32_t>(), n, K, out, true,
                                  [&](size_t, size_t lo, size_t hi) { return stage_scalars(sl, scalars, lo, hi, mont, sl.in_scalars.as<uint32_t>()); },
                                  [&](size_t, size_t lo, size_t hi) { return stage_bases<C>(sl, rb, lo, hi, sl.prepped.as<uint32_t>()); });
    if (rc) { (void)hipStreamSynchronize(sl.cstream); (void)hipStreamSynchronize(sl.stream); }      // nothing of ours may still read the caller's buffers
    if (grew) reserve_idle_slots<C>(&sl, 1, n, rb.stride);
    return rc;
}
template <class C, class HF> bool msm_oneshot_cached(const RawBases &rb, const uint64_t *scalars, size_t n, bool mont, uint64_t *out, int kind, int32_t &rc);      // (defined below)
// one-shot MSM on the calling thread's context: from the resident-bases cache if the caller's points are (or now become) resident, else from host memory
template <class C, class HF>
int32_t msm_oneshot_ctx(const RawBases &rb, const uint64_t *scalars, size_t n, bool mont, uint64_t *out) {
    if (gcache.enabled.load(std::memory_order_relaxed) && n >= gcache.min_n.load(std::memory_order_relaxed)) {
        int32_t rc;
        if (msm_oneshot_cached<C, HF>(rb, scalars, n, mont, out, C::NFP, rc)) return rc;
    }
    return msm_oneshot_here<C, HF>(rb, scalars, n, mont, out);
}
template <class C, class HF>
int32_t msm_oneshot(const RawBases &rb, const uint64_t *scalars, size_t n, bool mont, uint64_t *out) {
    if (!out || (n && (!rb.p || !scalars)) || n >= (1ull << 31) || !rb.ok<C>()) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;            // (before the size threshold: a missing device is never answered with "too small")
    if (!tl_no_min && n < gs.min_gpu_n) return DGPU_E_TOO_SMALL;
    // A process that drives several devices (dgpu_init_devices) and asked for it (dgpu_set_auto_shard_min_n): the unmodified call is sharded over them —
    // context k takes the contiguous balanced chunk k of the terms on a host thread of its own, from ITS resident copy of that chunk once the cache holds
    // it, and the partial points are folded on the host (point-chunk sharding; the same group element as on one device).
    if (n >= gs.auto_shard_min_n.load(std::memory_order_relaxed)) {
        const std::vector<int> cx = ready_contexts(0);
        if (cx.size() > 1) {
            std::vector<size_t> lo; shard_bounds(n, cx.size(), lo);
            const size_t JW = 3 * sizeof(HF) / 8;
            std::vector<uint64_t> parts(cx.size() * JW);
            const int32_t rc = run_shards(cx.size(), [&](size_t k) {
                CtxScope here(cx[k]);
                RawBases part = rb; part.p = rb.p + lo[k] * rb.stride; if (rb.is_inf) part.is_inf = rb.is_inf + lo[k];
                return msm_oneshot_ctx<C, HF>(part, scalars + lo[k] * 4, lo[k + 1] - lo[k], mont, parts.data() + k * JW);
            });
            if (rc) return rc;
            return host_fold_jacobian<HF>(parts.data(), cx.size(), out);
        }
    }
    return msm_oneshot_ctx<C, HF>(rb, scalars, n, mont, out);
}

// rec_hash != nullptr (the resident-bases cache, bases_cache.hpp): the fingerprint of every raw record, computed on the device from the staged bytes
template <class C>
int32_t bases_upload(const RawBases &rb, size_t n, uint64_t *handle, int kind, std::vector<uint64_t> *rec_hash = nullptr) {
    if (!handle || (n && !rb.p) || n >= (1ull << 31) || !rb.ok<C>()) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    void *p = nullptr;
    {
        SLOT_ACQUIRE(L, sl);
        HIPCHK(hipSetDevice(cur().device));
        if (dev_malloc(&p, std::max<size_t>(n, 1) * C::AFF_STRIDE * 4) != hipSuccess) { (void)hipGetLastError(); return DGPU_E_OOM; }
        int32_t rc = ws_stage_bases<C>(sl, rb, n);
        if (!rc) rc = stage_bases<C>(sl, rb, n, (uint32_t *)p);
        if (!rc && rec_hash && n) {               // (the compute stream has waited for every piece of the copy: stage_bases)
            rc = sl.digits.ensure(n * 8);
            if (!rc) {
                launch_raw_record_hash(sl.stream, sl.in_bases.as<uint8_t>(), rb.stride, rb.x_off, rb.y_off, rb.inf_off, rb.is_inf ? sl.in_inf.as<uint8_t>() : nullptr, C::ABI_W / 2, n, sl.digits.as<uint64_t>());
                if (hipMemcpyAsync(rec_hash->data(), sl.digits.p, n * 8, hipMemcpyDeviceToHost, sl.stream) != hipSuccess) { (void)hipGetLastError(); rc = DGPU_E_HIP; }
            }
        }
        if (hipStreamSynchronize(sl.cstream) != hipSuccess || hipStreamSynchronize(sl.stream) != hipSuccess) { if (!rc) rc = DGPU_E_HIP; }
        if (gs.prof) prof_flush(sl);
        if (rc) { (void)hipFree(p); return rc; }
    }
    *handle = register_handle(p, n, kind);
    (void)reserve_slots<C>(2, n, 0, nullptr);       // every slot is ready for an MSM over this query before the first proof arrives
    if (n && n <= SMALL_MSM_MAX_N && n <= gs.small_max.load()) {     // ... and a handle the small path will serve gets its table now (best effort: ~1 ms once, no allocation at its calls)
        HandleRef ref(*handle);
        if (ref.ok) { SlotLock L; if (L.ok && hipSetDevice(cur().device) == hipSuccess) { SmallSub sub; (void)small_sub_for<C>(*L.s, *handle, ref.h, 0, 0, sub, true); } }
    }
    return DGPU_OK;
}

// `check_min`: the sharded entry points apply the size threshold to the whole MSM, not to a shard
template <class C, class HF>
int32_t msm_handle(uint64_t bases, size_t offset, const uint64_t *scalars, size_t n, int mont, uint64_t *out, int kind, bool check_min = true) {
    if (!out || (n && !scalars)) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    // (bases behind a handle have their small-path table with them: one launch, 0.14 ms at 16 terms against 0.64 ms of one CPU thread)
    if (check_min && !tl_no_min && n < std::min<size_t>(gs.min_gpu_n, DGPU_MIN_GPU_N_HANDLE)) return DGPU_E_TOO_SMALL;
    HandleRef hb(bases);
    if (!hb.ok || (hb.h.kind != kind && hb.h.kind != kind + 9) || offset > hb.h.n || n > hb.h.n - offset) return DGPU_E_BADARG;
    CtxScope on_owner(hb.h.ctx);                    // run where the bases live
    SLOT_ACQUIRE(L, sl);
    HIPCHK(hipSetDevice(cur().device));
    int32_t rc;
    if ((rc = sl.in_scalars.ensure(std::max<size_t>(n, 1) * 32))) return rc;
    // the scalars cross PCIe in ranges; a range's kernels run under the next range's copy (msm_device_ranges)
    const size_t K = range_count(n, true);
    auto ready = [&](size_t, size_t lo, size_t hi) { return stage_scalars(sl, scalars, lo, hi, mont != 0, sl.in_scalars.as<uint32_t>()); };
    auto nothing = [](size_t, size_t, size_t) { return (int32_t)DGPU_OK; };
    if (hb.h.kind == kind + 9) rc = msm_device_pre_ranges<C, HF>(sl, *(const PreTable *)hb.h.p, offset, sl.in_scalars.as<uint32_t>(), n, K, out, ready);
    else {
        SmallSub sub; const bool have = K == 1 && n && small_sub_for<C>(sl, bases, hb.h, offset, n, sub);
        rc = msm_device_ranges<C, HF>(sl, (const uint32_t *)hb.h.p + offset * C::AFF_STRIDE, sl.in_scalars.as<uint32_t>(), n, K, out, false, ready, nothing, have ? &sub : nullptr);
    }
    if (rc) { (void)hipStreamSynchronize(sl.cstream); (void)hipStreamSynchronize(sl.stream); }      // nothing of ours may still read the caller's scalars
    return rc;
}

// ---- the resident-bases cache (bases_cache.hpp) --------------------------------------------------------------------------------------------
// device bytes of an entry of n points as a table of width c (0: the automatic choice; a handle too short for a table stays plain)
template <class C> inline size_t cache_entry_bytes(size_t n, int c) {
    if (c == 0) c = choose_c_pre(n);
    if (c == 0) return n * (size_t)C::AFF_STRIDE * 4 + (n <= SMALL_MSM_MAX_N ? small_sub_bytes<C>(n) : 0);
    return pre_tab_bytes<C>(n, 255 / c + 1);
}
// The cache's state machine for one sighting of `rb` (n points): true = `e` is a resident entry that still matches the caller's memory (records
// [off, off + n) of its handle; the shared_ptr pins it), false = not resident (first sighting, another thread is filling the entry, a stale or oversized
// key, a failed fill): the caller takes the points from host memory.  table_c: window width of the table a fill builds (0 = automatic).
// verify_now = false: the caller checks the entry against the host memory itself (msm_oneshot_cached in the exact mode: beside the MSM, not in front of it)
template <class C>
bool cache_acquire(const RawBases &rb, size_t n, int kind, int table_c, std::shared_ptr<CacheEntry> &e, size_t &off, bool verify_now = true, bool *was_resident = nullptr) {
    const CacheKey key{rb.p, n, rb.stride, rb.x_off, rb.y_off, rb.inf_off, rb.is_inf, kind, cur_index()};
    constexpr int words = C::ABI_W / 2;
    std::vector<std::shared_ptr<CacheEntry>> dropped;        // (destroyed after the locks are released: an entry's destructor frees its handle)
    e.reset(); off = 0;
    bool fill = false;
    {
        std::lock_guard<std::mutex> lk(gcache.mu);
        for (auto &c : gcache.entries) if (c->state == CacheEntry::READY && key.inside(c->k, &off)) { e = c; e->last_use = ++gcache.tick; break; }
    }
    if (was_resident) *was_resident = (bool)e;
    if (e && verify_now && !cache_verify(*e, key, off, words)) {            // the host memory behind the entry changed: forget it; this sighting is the new contents' first
        std::lock_guard<std::mutex> lk(gcache.mu);
        cache_remove_locked(e.get(), dropped);
        gcache.stale++; e.reset();
    }
    if (e) return true;
    off = 0;
    const uint64_t fp = range_fingerprint(key, words);
    size_t budget_auto = 0;
    if (gcache.budget.load() == CACHE_BUDGET_AUTO) {          // a quarter of what the device has free now (resolved once; dgpu_set_bases_cache_bytes overrides)
        size_t fr = 0, tot = 0;
        if (hipSetDevice(cur().device) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess) budget_auto = fr / 4; else (void)hipGetLastError();
    }
    {
        std::lock_guard<std::mutex> lk(gcache.mu);
        if (gcache.budget.load() == CACHE_BUDGET_AUTO) gcache.budget = budget_auto;
        std::shared_ptr<CacheEntry> seen;
        for (auto &c : gcache.entries) if (c->state != CacheEntry::READY && c->k.same(key)) { seen = c; break; }
        if (!seen) {                                          // first sighting: remember the fingerprint
            size_t n_seen = 0, oldest = gcache.entries.size();
            for (size_t i = 0; i < gcache.entries.size(); i++) if (gcache.entries[i]->state == CacheEntry::SEEN) { n_seen++; if (oldest == gcache.entries.size() || gcache.entries[i]->last_use < gcache.entries[oldest]->last_use) oldest = i; }
            if (n_seen >= CACHE_MAX_SEEN) { dropped.push_back(std::move(gcache.entries[oldest])); gcache.entries.erase(gcache.entries.begin() + oldest); }
            auto c = std::make_shared<CacheEntry>(); c->k = key; c->fp = fp; c->last_use = ++gcache.tick;
            gcache.entries.push_back(std::move(c));
        } else if (seen->state == CacheEntry::SEEN) {
            seen->last_use = ++gcache.tick;
            if (seen->fp != fp) seen->fp = fp;                // other contents at the same address: a first sighting again
            else if (cache_make_room_locked(cache_entry_bytes<C>(n, table_c), seen.get(), dropped)) { seen->state = CacheEntry::FILLING; e = seen; fill = true; }
        }                                                     // (FILLING: another thread is uploading this key right now)
    }
    if (!fill) { gcache.misses++; return false; }
    // second sighting: upload once (+ the per-record fingerprints), make it a table
    e->rec_hash.resize(n);
    uint64_t h = 0;
    int32_t frc = bases_upload<C>(rb, n, &h, kind, &e->rec_hash);
    if (!frc) {
        frc = bases_precompute<C>(h, table_c, kind);
        if (frc) { (void)dgpu_bases_free(h); h = 0; }
    }
    std::lock_guard<std::mutex> lk(gcache.mu);
    if (frc) { cache_remove_locked(e.get(), dropped); e.reset(); gcache.misses++; return false; }
    e->handle = h; e->bytes = cache_entry_bytes<C>(n, table_c); gcache.fills++;
    bool listed = false;
    for (auto &c : gcache.entries) if (c.get() == e.get()) listed = true;
    // (not listed any more: dgpu_bases_cache_clear ran meanwhile — this call still uses the table, which goes when the call lets go of it)
    if (listed) {
        e->state = CacheEntry::READY; e->last_use = ++gcache.tick; gcache.used += e->bytes;
        for (size_t i = 0; i < gcache.entries.size();) {      // an older entry that lies wholly inside the new one is redundant
            size_t o; CacheEntry &c = *gcache.entries[i];
            if (&c != e.get() && c.state == CacheEntry::READY && c.k.inside(e->k, &o)) { gcache.used -= c.bytes; dropped.push_back(std::move(gcache.entries[i])); gcache.entries.erase(gcache.entries.begin() + i); }
            else i++;
        }
    }
    return true;
}
// true: the call was served from a resident entry and rc is its answer; false: run it one-shot
template <class C, class HF>
bool msm_oneshot_cached(const RawBases &rb, const uint64_t *scalars, size_t n, bool mont, uint64_t *out, int kind, int32_t &rc) {
    std::shared_ptr<CacheEntry> e; size_t off = 0;
    // DGPU_CACHE_VERIFY_FULL: re-fingerprinting every record of a 2^20-point slice is 1.5 ms of host work — it runs BESIDE the MSM on the resident copy (whose
    // result is thrown away if the check fails) instead of in front of it: 4.8 -> 3.3 ms per call, the sampled check's latency with the exact answer
    const bool full = gcache.verify_samples.load(std::memory_order_relaxed) < 0;
    bool was_resident = false;
    if (!cache_acquire<C>(rb, n, kind, 0, e, off, !full, &was_resident)) return false;
    if (full && was_resident) {
        const CacheKey key{rb.p, n, rb.stride, rb.x_off, rb.y_off, rb.inf_off, rb.is_inf, kind, cur_index()};
        bool same = true; int32_t mrc = DGPU_OK;
        const int32_t prc = par_run(2, [&](size_t part) -> int32_t {
            if (part == 0) mrc = msm_handle<C, HF>(e->handle, off, scalars, n, mont, out, kind, false);
            else same = cache_verify(*e, key, off, C::ABI_W / 2);
            return DGPU_OK;
        });
        if (prc) same = false;                                // (the two parts could not be run: nothing was checked and nothing may be taken from `out`)
        if (!same) {                                          // the key changed under the entry: forget it, the one-shot path answers (and notes the new contents at its next call)
            std::vector<std::shared_ptr<CacheEntry>> dropped;
            std::lock_guard<std::mutex> lk(gcache.mu);
            cache_remove_locked(e.get(), dropped);
            gcache.stale++; return false;
        }
        rc = mrc;
    } else
    rc = msm_handle<C, HF>(e->handle, off, scalars, n, mont, out, kind, false);
    if (rc != DGPU_OK && rc != DGPU_E_BADARG) {               // a device-side failure on the resident path: forget the entry, let the one-shot path answer
        std::vector<std::shared_ptr<CacheEntry>> dropped;
        std::lock_guard<std::mutex> lk(gcache.mu);
        cache_remove_locked(e.get(), dropped);
        gcache.misses++; return false;
    }
    gcache.hits++;
    return true;
}
// A view of host bases as a handle for the duration of a larger call (dgpu_legogroth16_prove_host): the cache's entry when the WHOLE view is one
// (pinned by *pin), else a temporary upload that view_release frees.  Never fails for want of a cache: the temporary upload is the one-shot path.
// `check`: the entry was taken WITHOUT the stale-key check (the exact mode: view_verify runs it beside the proof, dock_prover.cpp)
struct ViewPin { std::shared_ptr<CacheEntry> e; uint64_t temp = 0; bool check = false; CacheKey key{}; int words = 0; };
template <class C>
int32_t view_acquire(const RawBases &rb, size_t n, int kind, int table_c, uint64_t *handle, void **pin) {
    if (!handle || !pin || (n && !rb.p) || n >= (1ull << 31) || !rb.ok<C>()) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    ViewPin *vp = new ViewPin();
    size_t off = 0;
    const bool full = gcache.verify_samples.load(std::memory_order_relaxed) < 0;
    bool was_resident = false;
    if (n && gcache.enabled.load() && n >= gcache.min_n.load() && cache_acquire<C>(rb, n, kind, table_c, vp->e, off, !full, &was_resident)) {
        if (off == 0 && vp->e->k.n == n) {
            gcache.hits++; *handle = vp->e->handle; *pin = vp;
            if (full && was_resident) { vp->check = true; vp->key = CacheKey{rb.p, n, rb.stride, rb.x_off, rb.y_off, rb.inf_off, rb.is_inf, kind, cur_index()}; vp->words = C::ABI_W / 2; }
            return DGPU_OK;
        }
        vp->e.reset();                        // (a sub-range of a larger entry: the prover addresses its queries from row 0 — take the points from the host)
    }
    const int32_t rc = bases_upload<C>(rb, n, &vp->temp, kind);
    if (rc) { delete vp; return rc; }
    *handle = vp->temp; *pin = vp;
    return DGPU_OK;
}
// the deferred stale-key check of a view (exact mode).  False: the host memory behind the entry changed — the entry is forgotten and whatever was computed
// from it must be thrown away
inline bool view_verify(void *pin) {
    ViewPin *vp = (ViewPin *)pin;
    if (!vp || !vp->check || !vp->e) return true;
    if (cache_verify(*vp->e, vp->key, 0, vp->words)) return true;
    std::vector<std::shared_ptr<CacheEntry>> dropped;
    std::lock_guard<std::mutex> lk(gcache.mu);
    cache_remove_locked(vp->e.get(), dropped);
    gcache.stale++;
    return false;
}
inline void view_release(void *pin) {
    ViewPin *vp = (ViewPin *)pin;
    if (!vp) return;
    if (vp->temp) (void)dgpu_bases_free(vp->temp);
    delete vp;
}

template <class C, class HF>
int32_t msm_resident(uint64_t bases, size_t boff, uint64_t scalars, size_t soff, size_t n, uint64_t *out, int kind, bool check_min = true) {
    if (!out) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    if (check_min && !tl_no_min && n < std::min<size_t>(gs.min_gpu_n, DGPU_MIN_GPU_N_HANDLE)) return DGPU_E_TOO_SMALL;
    HandleRef hb(bases), hs(scalars);
    if (!hb.ok || !hs.ok || (hb.h.kind != kind && hb.h.kind != kind + 9) || hs.h.kind != 3 || hb.h.ctx != hs.h.ctx) return DGPU_E_BADARG;    // both operands on one device
    if (boff > hb.h.n || n > hb.h.n - boff || soff > hs.h.n || n > hs.h.n - soff) return DGPU_E_BADARG;
    CtxScope on_owner(hb.h.ctx);
    SLOT_ACQUIRE(L, sl);
    HIPCHK(hipSetDevice(cur().device));
    if (hb.h.kind == kind + 9) return msm_device_pre<C, HF>(sl, *(const PreTable *)hb.h.p, boff, (const uint32_t *)hs.h.p + soff * 8, n, out);
    SmallSub sub; const bool have = n && small_sub_for<C>(sl, bases, hb.h, boff, n, sub);
    return msm_device<C, HF>(sl, (const uint32_t *)hb.h.p + boff * C::AFF_STRIDE, (const uint32_t *)hs.h.p + soff * 8, n, out, have ? &sub : nullptr);
}

// ---- several GPUs behind the ABI (SURVEY.md 8b `dgpu_msm_g1_sharded`, 8e point-chunk sharding) ----------------------------------------
// One process, one context per device, one host thread per device inside the call: device k runs the whole pipeline on the terms
// [lo_k, lo_{k+1}) and hands back one normalised Jacobian point (144 / 288 B); the partials are folded on the host.  No collective is
// needed inside a process; the multi-process form (one rank per GPU, RCCL all_gather of the same partials) stays above the ABI.
template <class C, class HF>
int32_t msm_sharded_oneshot(const uint64_t *bases, const uint8_t *is_inf, const uint64_t *scalars, size_t n, int32_t ngpus, bool mont, uint64_t *out) {
    if (!out || (n && (!bases || !scalars)) || n >= (1ull << 31) || ngpus < 0) return DGPU_E_BADARG;
    const std::vector<int> cx = ready_contexts(ngpus);
    if (cx.empty()) return DGPU_E_NODEVICE;
    if (ngpus > 0 && (int)cx.size() < ngpus) return DGPU_E_BADARG;
    if (!tl_no_min && n < gs.min_gpu_n) return DGPU_E_TOO_SMALL;
    std::vector<size_t> lo; shard_bounds(n, cx.size(), lo);
    const size_t JW = 3 * sizeof(HF) / 8, BW = 2 * sizeof(HF) / 8;
    std::vector<uint64_t> parts(cx.size() * JW);
    int32_t rc = run_shards(cx.size(), [&](size_t k) {
        CtxScope here(cx[k]);
        const size_t cnt = lo[k + 1] - lo[k];
        return msm_oneshot_here<C, HF>(RawBases::packed<C>(bases + lo[k] * BW, is_inf ? is_inf + lo[k] : nullptr), scalars + lo[k] * 4, cnt, mont, parts.data() + k * JW);
    });
    if (rc) return rc;
    return host_fold_jacobian<HF>(parts.data(), cx.size(), out);
}
template <class C>
int32_t bases_upload_sharded(const uint64_t *bases, const uint8_t *is_inf, size_t n, int32_t ngpus, uint64_t *handle, int kind /* 1 | 2 */) {
    if (!handle || (n && !bases) || n >= (1ull << 31) || ngpus < 0) return DGPU_E_BADARG;
    const std::vector<int> cx = ready_contexts(ngpus);
    if (cx.empty()) return DGPU_E_NODEVICE;
    if (ngpus > 0 && (int)cx.size() < ngpus) return DGPU_E_BADARG;
    ShardSet *ss = new ShardSet();
    ss->n = n; ss->sub.assign(cx.size(), 0); shard_bounds(n, cx.size(), ss->lo);
    const size_t BW = 2 * C::ABI_W / 2;           // u64 words per affine point
    int32_t rc = run_shards(cx.size(), [&](size_t k) {
        CtxScope here(cx[k]);
        return bases_upload<C>(RawBases::packed<C>(bases + ss->lo[k] * BW, is_inf ? is_inf + ss->lo[k] : nullptr), ss->lo[k + 1] - ss->lo[k], &ss->sub[k], kind);
    });
    if (rc) { for (uint64_t h : ss->sub) if (h) (void)dgpu_bases_free(h); delete ss; return rc; }
    *handle = register_handle(ss, n, kind + 6);       // 7 = G1 sharded, 8 = G2 sharded
    return DGPU_OK;
}
// fresh host scalars against a sharded bases handle: shard k uploads and uses scalars [lo_k, min(lo_{k+1}, n))
template <class C, class HF>
int32_t msm_sharded_handle(uint64_t bases, const uint64_t *scalars, size_t n, int mont, uint64_t *out, int kind) {
    if (!out || (n && !scalars)) return DGPU_E_BADARG;
    HandleRef hb(bases);
    if (!hb.ok || hb.h.kind != kind + 6 || n > hb.h.n) return DGPU_E_BADARG;
    if (!tl_no_min && n < gs.min_gpu_n) return DGPU_E_TOO_SMALL;
    const ShardSet &ss = *(const ShardSet *)hb.h.p;
    const size_t G = ss.sub.size(), JW = 3 * sizeof(HF) / 8;
    std::vector<uint64_t> parts(G * JW);
    int32_t rc = run_shards(G, [&](size_t k) {
        const size_t lo = std::min(ss.lo[k], n), hi = std::min(ss.lo[k + 1], n);
        return msm_handle<C, HF>(ss.sub[k], 0, scalars + lo * 4, hi - lo, mont, parts.data() + k * JW, kind, false);
    });
    if (rc) return rc;
    return host_fold_jacobian<HF>(parts.data(), G, out);
}
// both operands resident on their devices (inputs pre-sharded: BASELINE config 5's timed region)
template <class C, class HF>
int32_t msm_sharded_resident(uint64_t bases, uint64_t scalars, uint64_t *out, int kind) {
    if (!out) return DGPU_E_BADARG;
    HandleRef hb(bases), hs(scalars);
    if (!hb.ok || !hs.ok || hb.h.kind != kind + 6 || hs.h.kind != 9) return DGPU_E_BADARG;
    const ShardSet &sb = *(const ShardSet *)hb.h.p, &sv = *(const ShardSet *)hs.h.p;
    if (sb.sub.size() != sv.sub.size() || sv.n > sb.n) return DGPU_E_BADARG;
    for (size_t k = 0; k < sb.sub.size(); k++) if (sv.lo[k] != std::min(sb.lo[k], sv.n) || sv.lo[k + 1] != std::min(sb.lo[k + 1], sv.n)) return DGPU_E_BADARG;
    if (sv.n < gs.min_gpu_n) return DGPU_E_TOO_SMALL;
    const size_t G = sb.sub.size(), JW = 3 * sizeof(HF) / 8;
    std::vector<uint64_t> parts(G * JW);
    int32_t rc = run_shards(G, [&](size_t k) { return msm_resident<C, HF>(sb.sub[k], 0, sv.sub[k], 0, sv.lo[k + 1] - sv.lo[k], parts.data() + k * JW, kind, false); });
    if (rc) return rc;
    return host_fold_jacobian<HF>(parts.data(), G, out);
}


}  // namespace dock
