// crypto_amd/csrc/msm_driver.cuh — host driver of the MSM pipeline (templated on the curve), included by
// dock_g1.hip and dock_g2.hip so the two curves compile in parallel.
#pragma once
#include <chrono>
#include "dock_ctx.hpp"
#include "host_field.hpp"
#include "msm_launch.cuh"
#include "sort_launch.cuh"

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

// d_bases: prepared records; d_scalars: canonical 8 x u32 per scalar.  Caller holds the slot.
template <class C, class HF>
int32_t msm_device(Slot &sl, const uint32_t *d_bases, const uint32_t *d_scalars, size_t n, uint64_t *out_xyz) {
    if (n == 0) { typedef hostf::HXyzz<HF> PT; PT id = PT::identity(); HF X, Y, Z; id.to_normalised_jacobian(X, Y, Z);
        const size_t FWORDS = sizeof(HF) / 8; memcpy(out_xyz, &X, sizeof(HF)); memcpy(out_xyz + FWORDS, &Y, sizeof(HF)); memcpy(out_xyz + 2 * FWORDS, &Z, sizeof(HF)); return DGPU_OK; }
    if (n >= (1ull << 31)) return DGPU_E_BADARG;
    const int c = choose_c(n, C::NFP == 2);
    const int W = 255 / c + 1;
    const uint32_t B = 1u << (c - 1);
    if ((uint64_t)W * B >= (1ull << 31) || (uint64_t)n * W >= (1ull << 32)) return DGPU_E_BADARG;
    const uint32_t NB = (uint32_t)W * B;
    const int mshift = std::max(0, c - 1 - 12);
    const int G = (int)(B >> (6 + mshift));          // groups per window (<= 64), B >= 64 because c >= 7
    const size_t NG = (size_t)W * G;
    const size_t Emax = (size_t)n * W;
    const int CH = C::NFP == 2 ? choose_chunk(Emax, 32, 150000) : choose_chunk(Emax, 16, 300000);
    const size_t T = (Emax + CH - 1) / CH;
    const size_t nblk = scan_blocks(NB);

    int32_t rc;
    if ((rc = sl.cnt.ensure(((size_t)NB + 1) * 4))) return rc;
    if ((rc = sl.off.ensure(((size_t)NB + 1) * 4))) return rc;
    if ((rc = sl.cursor.ensure(((size_t)NB + 1) * 4))) return rc;
    if ((rc = sl.bsums.ensure((nblk + 2) * 4))) return rc;
    if ((rc = sl.entries.ensure(Emax * 4))) return rc;
    if ((rc = sl.bucket.ensure(soa_points(NB) * C::XW * 4))) return rc;
    if ((rc = sl.bucket_inf.ensure(NB))) return rc;
    if ((rc = sl.head.ensure(soa_points(T) * C::XW * 4))) return rc;
    if ((rc = sl.tail.ensure(soa_points(T) * C::XW * 4))) return rc;
    if ((rc = sl.head_b.ensure(T * 4))) return rc;
    if ((rc = sl.tail_b.ensure(T * 4))) return rc;
    if ((rc = sl.part_inf.ensure(T * 2))) return rc;
    if ((rc = sl.l1.ensure(NG * 2 * C::XW * 4))) return rc;
    if ((rc = sl.l1_inf.ensure(NG * 2))) return rc;
    if ((rc = sl.win.ensure((size_t)W * 4 * C::ABI_W * 4))) return rc;
    if ((rc = sl.win_inf.ensure(W))) return rc;

    hipStream_t s = sl.stream;
    // counting sort of the n*W (key, term) pairs: digit codes -> LDS histograms per (window, bucket range) -> scan -> LDS cursors
    const bool wide = c > 16;
    const size_t n_pad = (n + 7) & ~(size_t)7;
    int RANGES = 1; while ((B / RANGES) * 4 > 64 * 1024 || W * RANGES < 256) { if (B / RANGES <= 64) break; RANGES *= 2; }
    int rb_log = 0; while ((1u << rb_log) < B / RANGES) rb_log++;
    const int wpx = (W + 7) / 8;
    const unsigned sort_grid = (unsigned)(8 * wpx * RANGES);
    const size_t lds_bytes = ((size_t)1 << rb_log) * 4;
    if ((rc = sl.digits.ensure((size_t)W * n_pad * (wide ? 4 : 2)))) return rc;
    const uint32_t heavy_thr = 16u * (uint32_t)CH, HEAVY_CAP = (uint32_t)(Emax / heavy_thr) + 1;   // at most E / thr buckets can be heavy
    if ((rc = sl.heavy.ensure(((size_t)HEAVY_CAP + 1) * 4))) return rc;
    {
        StageTimer st(sl, "msm.count");
        HIPCHK(hipMemsetAsync(sl.bucket_inf.p, 1, NB, s));
        HIPCHK(hipMemsetAsync(sl.heavy.p, 0, 4, s));
        launch_digit_codes(s, wide, d_scalars, d_bases, C::AFF_STRIDE, 2 * C::FW, n, n_pad, c, W, sl.digits.p);
        launch_sort_sweep(s, wide, false, sort_grid, lds_bytes, sl.digits.p, n, n_pad, W, RANGES, rb_log, B, sl.cnt.as<uint32_t>(), nullptr, nullptr, heavy_thr, sl.heavy.as<uint32_t>(), HEAVY_CAP);
    }
    {
        StageTimer st(sl, "msm.scan");
        launch_scan(s, sl.cnt.as<uint32_t>(), sl.off.as<uint32_t>(), sl.cursor.as<uint32_t>(), sl.bsums.as<uint32_t>(), (size_t)NB);
    }
    {
        StageTimer st(sl, "msm.scatter");
        launch_sort_sweep(s, wide, true, sort_grid, lds_bytes, sl.digits.p, n, n_pad, W, RANGES, rb_log, B, nullptr, sl.off.as<uint32_t>(), sl.entries.as<uint32_t>(), heavy_thr, sl.heavy.as<uint32_t>(), HEAVY_CAP);
    }
    {
        StageTimer st(sl, "msm.accumulate");
        static const uint32_t dbg_mask = getenv("DGPU_DBG_NOGATHER") ? 1023u : 0xffffffffu;   // experiment: L2-resident points
        launch_accumulate<C>(s, d_bases, sl.entries.as<uint32_t>(), sl.off.as<uint32_t>(), NB, sl.bucket.as<uint32_t>(), sl.bucket_inf.as<uint8_t>(),
                           sl.head.as<uint32_t>(), sl.tail.as<uint32_t>(), sl.head_b.as<uint32_t>(), sl.tail_b.as<uint32_t>(), sl.part_inf.as<uint8_t>(), T, (uint32_t)CH, dbg_mask);
    }
    {
        StageTimer st(sl, "msm.fixup");
        launch_fixup<C>(s, NB, sl.bucket.as<uint32_t>(), sl.bucket_inf.as<uint8_t>(), sl.head.as<uint32_t>(), sl.tail.as<uint32_t>(),
                           sl.head_b.as<uint32_t>(), sl.tail_b.as<uint32_t>(), sl.part_inf.as<uint8_t>(), T, sl.off.as<uint32_t>(), heavy_thr);
        launch_fixup_heavy<C>(s, sl.heavy.as<uint32_t>(), HEAVY_CAP, sl.off.as<uint32_t>(), (uint32_t)CH, NB, sl.bucket.as<uint32_t>(), sl.bucket_inf.as<uint8_t>(),
                           sl.head.as<uint32_t>(), sl.tail.as<uint32_t>(), sl.part_inf.as<uint8_t>(), T);
    }
    {
        StageTimer st(sl, "msm.reduce");
        launch_reduce_l0<C>(s, (unsigned)NG, sl.bucket.as<uint32_t>(), sl.bucket_inf.as<uint8_t>(), NB, mshift, sl.l1.as<uint32_t>(), sl.l1_inf.as<uint8_t>());
        launch_reduce_top<C>(s, (unsigned)W, sl.l1.as<uint32_t>(), sl.l1_inf.as<uint8_t>(), G, 6 + mshift, sl.win.as<uint32_t>(), sl.win_inf.as<uint8_t>());
    }
    HIPCHK(hipGetLastError());
    std::vector<uint64_t> hwin((size_t)W * 2 * C::ABI_W);
    std::vector<uint8_t> hinf(W);
    HIPCHK(hipMemcpyAsync(hwin.data(), sl.win.p, (size_t)W * 4 * C::ABI_W * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hinf.data(), sl.win_inf.p, W, hipMemcpyDeviceToHost, s));
    auto tsync0 = std::chrono::steady_clock::now();
    HIPCHK(hipStreamSynchronize(s));
    auto tsync1 = std::chrono::steady_clock::now();
    if (g.prof) prof_flush(sl);
    host_fold<HF>(hwin.data(), hinf.data(), W, c, out_xyz);
    if (g.prof) {
        auto t2 = std::chrono::steady_clock::now();
        prof_add_host("msm.host_wait", std::chrono::duration<double, std::milli>(tsync1 - tsync0).count());
        prof_add_host("msm.host_fold", std::chrono::duration<double, std::milli>(t2 - tsync1).count());
    }
    return DGPU_OK;
}

template <class C>
int32_t prep_bases(Slot &sl, const uint64_t *h_bases, const uint8_t *h_inf, size_t n, uint32_t *d_out) {
    int32_t rc;
    const size_t bytes = n * 2 * C::ABI_W * 4;
    if ((rc = sl.in_bases.ensure(bytes ? bytes : 16))) return rc;
    HIPCHK(hipMemcpyAsync(sl.in_bases.p, h_bases, bytes, hipMemcpyHostToDevice, sl.stream));
    uint8_t *dinf = nullptr;
    if (h_inf) { if ((rc = sl.in_inf.ensure(n))) return rc; HIPCHK(hipMemcpyAsync(sl.in_inf.p, h_inf, n, hipMemcpyHostToDevice, sl.stream)); dinf = sl.in_inf.as<uint8_t>(); }
    StageTimer st(sl, "msm.prep_bases");
    launch_prep_bases<C>(sl.stream, sl.in_bases.as<uint32_t>(), dinf, n, d_out);
    return DGPU_OK;
}

template <class C, class HF>
int32_t msm_oneshot(const uint64_t *bases, const uint8_t *is_inf, const uint64_t *scalars, size_t n, bool mont, uint64_t *out) {
    if (!out || (n && (!bases || !scalars)) || n >= (1ull << 31)) return DGPU_E_BADARG;
    if (n < g.min_gpu_n) return DGPU_E_TOO_SMALL;
    if (!g.ready) return DGPU_E_NODEVICE;
    SlotLock L; Slot &sl = *L.s;
    HIPCHK(hipSetDevice(g.device));
    int32_t rc;
    if (n) {
        if ((rc = sl.prepped.ensure(n * C::AFF_STRIDE * 4))) return rc;
        if ((rc = sl.in_scalars.ensure(n * 32))) return rc;
        if ((rc = prep_bases<C>(sl, bases, is_inf, n, sl.prepped.as<uint32_t>()))) return rc;
        if ((rc = upload_scalars(sl, scalars, n, mont, sl.in_scalars.as<uint32_t>()))) return rc;
    }
    return msm_device<C, HF>(sl, sl.prepped.as<uint32_t>(), sl.in_scalars.as<uint32_t>(), n, out);
}

template <class C>
int32_t bases_upload(const uint64_t *bases, const uint8_t *is_inf, size_t n, uint64_t *handle, int kind) {
    if (!handle || (n && !bases) || n >= (1ull << 31)) return DGPU_E_BADARG;
    if (!g.ready) return DGPU_E_NODEVICE;
    void *p = nullptr;
    {
        SlotLock L; Slot &sl = *L.s;
        HIPCHK(hipSetDevice(g.device));
        if (hipMalloc(&p, std::max<size_t>(n, 1) * C::AFF_STRIDE * 4) != hipSuccess) { (void)hipGetLastError(); return DGPU_E_OOM; }
        int32_t rc = n ? prep_bases<C>(sl, bases, is_inf, n, (uint32_t *)p) : DGPU_OK;
        if (rc == DGPU_OK && hipStreamSynchronize(sl.stream) != hipSuccess) rc = DGPU_E_HIP;
        if (rc) { (void)hipFree(p); return rc; }
    }
    std::lock_guard<std::mutex> lk(g.mu);
    uint64_t h = g.next_handle++;
    g.handles[h] = Handle{p, n, kind};
    *handle = h;
    return DGPU_OK;
}

template <class C, class HF>
int32_t msm_handle(uint64_t bases, size_t offset, const uint64_t *scalars, size_t n, int mont, uint64_t *out, int kind) {
    if (!out || (n && !scalars)) return DGPU_E_BADARG;
    if (n < g.min_gpu_n) return DGPU_E_TOO_SMALL;
    if (!g.ready) return DGPU_E_NODEVICE;
    Handle hb;
    if (!lookup_handle(bases, hb) || hb.kind != kind || offset > hb.n || n > hb.n - offset) return DGPU_E_BADARG;
    SlotLock L; Slot &sl = *L.s;
    HIPCHK(hipSetDevice(g.device));
    int32_t rc;
    if ((rc = sl.in_scalars.ensure(std::max<size_t>(n, 1) * 32))) return rc;
    if (n && (rc = upload_scalars(sl, scalars, n, mont != 0, sl.in_scalars.as<uint32_t>()))) return rc;
    return msm_device<C, HF>(sl, (const uint32_t *)hb.p + offset * C::AFF_STRIDE, sl.in_scalars.as<uint32_t>(), n, out);
}

template <class C, class HF>
int32_t msm_resident(uint64_t bases, size_t boff, uint64_t scalars, size_t soff, size_t n, uint64_t *out, int kind) {
    if (!out) return DGPU_E_BADARG;
    if (n < g.min_gpu_n) return DGPU_E_TOO_SMALL;
    if (!g.ready) return DGPU_E_NODEVICE;
    Handle hb, hs;
    if (!lookup_handle(bases, hb) || !lookup_handle(scalars, hs) || hb.kind != kind || hs.kind != 3) return DGPU_E_BADARG;
    if (boff > hb.n || n > hb.n - boff || soff > hs.n || n > hs.n - soff) return DGPU_E_BADARG;
    SlotLock L; Slot &sl = *L.s;
    HIPCHK(hipSetDevice(g.device));
    return msm_device<C, HF>(sl, (const uint32_t *)hb.p + boff * C::AFF_STRIDE, (const uint32_t *)hs.p + soff * 8, n, out);
}


}  // namespace dock
