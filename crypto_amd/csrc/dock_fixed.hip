// crypto_amd/csrc/dock_fixed.hip — fixed-base batch multiplication entry points of include/dock_gpu.h
// (WindowTable of utils/src/msm.rs:8-62; FixedBase::msm of legogroth16/src/generator.rs:335-399).
#include "msm_driver.hip.h"
#include "fixed_launch.hip.h"
using namespace dock;

namespace {

// 2^(8k) * B for k < 32 in affine ABI form: 248 doublings and ONE field inversion on the host (the chain is serial; a GPU lane would take ~3 ms)
template <class HF> void window_bases_host(const uint64_t *base_abi, uint64_t *out_abi) {
    constexpr int NW = 32;
    constexpr size_t FW64 = sizeof(HF) / 8;
    hostf::HXyzz<HF> P; memcpy(&P.x, base_abi, sizeof(HF)); memcpy(&P.y, base_abi + FW64, sizeof(HF));
    P.zz = HF::one(); P.zzz = HF::one(); P.inf = false;
    std::vector<hostf::HXyzz<HF>> B(NW);
    for (int k = 0; k < NW; k++) { B[k] = P; if (k + 1 < NW) for (int j = 0; j < 8; j++) P.dbl_in_place(); }
    // Montgomery's trick over the ZZZ coordinates
    std::vector<HF> pre(NW); HF acc = HF::one();
    for (int k = 0; k < NW; k++) { pre[k] = acc; acc = acc * B[k].zzz; }
    HF inv = acc.inv();
    for (int k = NW - 1; k >= 0; k--) {
        HF i3 = inv * pre[k]; inv = inv * B[k].zzz;
        HF t = B[k].zz * i3, i2 = t * t;
        HF x = B[k].x * i2, y = B[k].y * i3;
        memcpy(out_abi + (size_t)k * 2 * FW64, &x, sizeof(HF)); memcpy(out_abi + (size_t)k * 2 * FW64 + FW64, &y, sizeof(HF));
    }
}

template <class C, class HF> int32_t table_build(const uint64_t *base, uint64_t *handle, int kind) {
    if (!base || !handle) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    constexpr size_t W64 = C::ABI_W;                     // u64 words per affine point (2 coordinates x ABI_W u32)
    uint64_t any = 0; for (size_t k = 0; k < W64; k++) any |= base[k];
    std::vector<uint64_t> wb(32 * W64, 0);
    if (any) window_bases_host<HF>(base, wb.data());     // identity base: all-zero records -> flagged as identity by k_prep_bases
    void *tab = nullptr;
    {
        SLOT_ACQUIRE(L, sl);
        HIPCHK(hipSetDevice(cur().device));
        int32_t rc;
        if ((rc = sl.prepped.ensure(32 * C::AFF_STRIDE * 4))) return rc;
        if (dev_malloc(&tab, (size_t)msm::FIXED_TABLE_ENTRIES * C::AFF_STRIDE * 4) != hipSuccess) { (void)hipGetLastError(); return DGPU_E_OOM; }
        // (the fixed-base kernels run over Fp: their window bases are 14 x 29-bit records, not the MSM pipeline's form)
        rc = sl.in_bases.ensure(32 * 2 * C::ABI_W * 4 + 16);
        if (rc == DGPU_OK && hipMemcpyAsync(sl.in_bases.p, wb.data(), 32 * 2 * C::ABI_W * 4, hipMemcpyHostToDevice, sl.stream) != hipSuccess) rc = DGPU_E_HIP;
        if (rc == DGPU_OK) msm::launch_prep_bases_fp<C>(sl.stream, sl.in_bases.as<uint32_t>(), nullptr, 32, sl.prepped.as<uint32_t>());
        if (rc == DGPU_OK) {
            StageTimer st(sl, "fixed.table");
            msm::launch_fb_table<C>(sl.stream, sl.prepped.as<uint32_t>(), (uint32_t *)tab);
        }
        if (rc == DGPU_OK && (hipGetLastError() != hipSuccess || hipStreamSynchronize(sl.stream) != hipSuccess)) rc = DGPU_E_HIP;
        if (gs.prof) prof_flush(sl);
        if (rc) { (void)hipFree(tab); return rc; }
    }
    *handle = register_handle(tab, (size_t)msm::FIXED_TABLE_ENTRIES, kind);
    return DGPU_OK;
}

// out / out_inf: host arrays, or (bases_handle != nullptr) the products stay in HBM as an MSM bases handle
template <class C> int32_t table_mul(uint64_t table, const uint64_t *scalars, size_t n, int32_t mont, uint64_t *out, uint8_t *out_inf, int kind, uint64_t *bases_handle = nullptr) {
    if ((n && !scalars) || (!bases_handle && n && (!out || !out_inf)) || n >= (1ull << 31)) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    HandleRef tref(table);
    if (!tref.ok || tref.h.kind != kind) return DGPU_E_BADARG;
    const Handle &ht = tref.h;
    CtxScope on_owner(ht.ctx);                           // the table's device
    if (n == 0 && !bases_handle) return DGPU_OK;
    void *keep = nullptr;
    if (n == 0) {          // an empty query (a circuit whose witnesses are all committed has no l_query entries): an empty bases handle, no launch
        HIPCHK(hipSetDevice(cur().device));
        if (dev_malloc(&keep, (size_t)C::AFF_STRIDE * 4) != hipSuccess) { (void)hipGetLastError(); return DGPU_E_OOM; }
        *bases_handle = register_handle(keep, 0, kind - 4);
        return DGPU_OK;
    }
    {
    SLOT_ACQUIRE(L, sl);
    HIPCHK(hipSetDevice(cur().device));
    int32_t rc;
    const size_t pt_bytes = 2 * C::ABI_W * 4;
    if ((rc = sl.in_scalars.ensure(n * 32))) return rc;
    if ((rc = sl.in_bases.ensure(n * pt_bytes + n))) return rc;
    if ((rc = upload_scalars(sl, scalars, n, mont != 0, sl.in_scalars.as<uint32_t>()))) return rc;
    uint8_t *dinf = sl.in_bases.as<uint8_t>() + n * pt_bytes;
    { StageTimer st(sl, "fixed.mul");
      msm::launch_fb_mul<C>(sl.stream, (const uint32_t *)ht.p, sl.in_scalars.as<uint32_t>(), n, sl.in_bases.as<uint32_t>(), dinf); }
    HIPCHK(hipGetLastError());
    if (bases_handle) {
        if (dev_malloc(&keep, std::max<size_t>(n, 1) * C::AFF_STRIDE * 4) != hipSuccess) { (void)hipGetLastError(); return DGPU_E_OOM; }
        if (n) msm::launch_prep_bases<C>(sl.stream, sl.in_bases.as<uint32_t>(), dinf, n, (uint32_t *)keep);
    } else {
        HIPCHK(hipMemcpyAsync(out, sl.in_bases.p, n * pt_bytes, hipMemcpyDeviceToHost, sl.stream));
        HIPCHK(hipMemcpyAsync(out_inf, dinf, n, hipMemcpyDeviceToHost, sl.stream));
    }
    if (hipStreamSynchronize(sl.stream) != hipSuccess) { (void)hipGetLastError(); if (keep) (void)hipFree(keep); return DGPU_E_HIP; }
    if (gs.prof) prof_flush(sl);
    }
    if (bases_handle) { *bases_handle = register_handle(keep, n, kind - 4);      // 5 -> 1 (G1 bases), 6 -> 2 (G2 bases)
        (void)reserve_slots<C>(2, n, 0, nullptr); }
    return DGPU_OK;
}

template <class C, class HF> int32_t fixed_base(const uint64_t *base, const uint64_t *scalars, size_t n, int32_t mont, uint64_t *out, uint8_t *out_inf, int kind) {
    uint64_t h = 0;
    int32_t rc = table_build<C, HF>(base, &h, kind);
    if (rc) return rc;
    rc = table_mul<C>(h, scalars, n, mont, out, out_inf, kind);
    (void)dgpu_bases_free(h);
    return rc;
}

// out_i = addend_i + s_i * P_i
template <class C> int32_t mul_add(const uint64_t *p, const uint8_t *p_inf, const uint64_t *scalars, size_t scalar_stride, const uint64_t *addend, const uint8_t *add_inf, size_t n, uint64_t *out, uint8_t *out_inf) {
    if ((n && (!p || !scalars || !out || !out_inf)) || (scalar_stride != 0 && scalar_stride != 4) || (add_inf && !addend) || n >= (1ull << 31)) return DGPU_E_BADARG;
    if (n == 0) return DGPU_OK;
    if (!cur().ready) return DGPU_E_NODEVICE;
    SLOT_ACQUIRE(L, sl);
    HIPCHK(hipSetDevice(cur().device));
    int32_t rc;
    const size_t pt = 2 * C::ABI_W * 4, nsc = scalar_stride ? n : 1;
    // in_bases: [points | addends], prepped: [out | out_inf], in_inf: [p_inf | add_inf]
    if ((rc = sl.in_bases.ensure(2 * n * pt))) return rc;
    if ((rc = sl.in_scalars.ensure(nsc * 32))) return rc;
    if ((rc = sl.in_inf.ensure(2 * n))) return rc;
    if ((rc = sl.prepped.ensure(n * pt + n))) return rc;
    hipStream_t s = sl.stream;
    uint8_t *dp = sl.in_bases.as<uint8_t>();
    HIPCHK(hipMemcpyAsync(dp, p, n * pt, hipMemcpyHostToDevice, s));
#ifdef DGPU_DEV
    static const bool plain = getenv("DGPU_MULADD_PLAIN") != nullptr;     // development switch: the 255-bit chains (no endomorphism)
#else
    constexpr bool plain = false;
#endif
    // every scalar is sent split: G1 (k1 | k2) with k = k1 + k2 lambda (GLV), G2 four base-|x| digits (GLS) — host_field.hpp
    std::vector<uint64_t> split(nsc * 4);
    if (!plain)
        for (size_t k = 0; k < nsc; k++) {
            if constexpr (C::NFP == 1) hostf::glv_decompose(scalars + 4 * k, &split[4 * k], &split[4 * k + 2]);
            else hostf::gls4_decompose(scalars + 4 * k, &split[4 * k]);
        }
    HIPCHK(hipMemcpyAsync(sl.in_scalars.p, plain ? scalars : split.data(), nsc * 32, hipMemcpyHostToDevice, s));
    const uint32_t *dadd = nullptr; const uint8_t *dpinf = nullptr, *dainf = nullptr;
    if (addend) { HIPCHK(hipMemcpyAsync(dp + n * pt, addend, n * pt, hipMemcpyHostToDevice, s)); dadd = (const uint32_t *)(dp + n * pt); }
    if (p_inf) { HIPCHK(hipMemcpyAsync(sl.in_inf.p, p_inf, n, hipMemcpyHostToDevice, s)); dpinf = sl.in_inf.as<uint8_t>(); }
    if (add_inf) { HIPCHK(hipMemcpyAsync(sl.in_inf.as<uint8_t>() + n, add_inf, n, hipMemcpyHostToDevice, s)); dainf = sl.in_inf.as<uint8_t>() + n; }
    uint8_t *dout_inf = sl.prepped.as<uint8_t>() + n * pt;
    { StageTimer st(sl, "fixed.mul_add");
      if (plain) msm::launch_mul_add<C>(s, (const uint32_t *)dp, dpinf, sl.in_scalars.as<uint32_t>(), (int)(scalar_stride * 2), dadd, dainf, n, sl.prepped.as<uint32_t>(), dout_inf);
      else if constexpr (C::NFP == 1) msm::launch_g1_scale_quad(s, (const uint32_t *)dp, dpinf, sl.in_scalars.as<uint32_t>(), (int)(scalar_stride * 2), nullptr, n, sl.prepped.as<uint32_t>(), dout_inf, dadd, dainf);
      else msm::launch_mul_add_g2_gls(s, (const uint32_t *)dp, dpinf, sl.in_scalars.as<uint32_t>(), (int)(scalar_stride * 2), dadd, dainf, n, sl.prepped.as<uint32_t>(), dout_inf); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, sl.prepped.p, n * pt, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_inf, dout_inf, n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (gs.prof) prof_flush(sl);
    return DGPU_OK;
}


// ---- the folding step with the doubling chains done ahead of the scalar (fold_kernels.hip.h) -------------------------------------------------
// handle kind 13 / 14: the table of n G1 / G2 points (XYZZ entries as ABI words) and one identity flag per point behind it
inline size_t fold_words(bool g2) { return g2 ? msm::FOLD_TABLE_WORDS_G2 : msm::FOLD_TABLE_WORDS_G1; }
// the tables of an aggregation halve from round to round: a buffer is kept for the next, smaller one (a hipMalloc / hipFree pair per round costs more
// than the kernels, and hipFree waits for every other call in flight)
void *fold_alloc(size_t bytes, size_t &got) {
    {
        std::lock_guard<std::mutex> lk(gs.mu);
        auto &pool = cur().fold_pool;
        size_t best = pool.size();
        for (size_t k = 0; k < pool.size(); k++) if (pool[k].second >= bytes && (best == pool.size() || pool[k].second < pool[best].second)) best = k;
        if (best != pool.size()) { void *p = pool[best].first; got = pool[best].second; pool[best] = pool.back(); pool.pop_back(); return p; }
    }
    void *p = nullptr;
    if (dev_malloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    got = bytes;
    return p;
}
void fold_release(int ctx, void *p, size_t bytes) {
    void *drop = p;                                   // up to four buffers are kept; a larger one displaces the smallest
    {
        std::lock_guard<std::mutex> lk(gs.mu);
        auto &pool = ctxs[ctx].fold_pool;
        if (ctxs[ctx].ready.load()) {
            if (pool.size() < 4) { pool.push_back({p, bytes}); drop = nullptr; }
            else {
                size_t least = 0;
                for (size_t k = 1; k < pool.size(); k++) if (pool[k].second < pool[least].second) least = k;
                if (pool[least].second < bytes) { drop = pool[least].first; pool[least] = {p, bytes}; }
            }
        }
    }
    if (drop) (void)hipFree(drop);
}
// the chains of a G1 and a G2 point set in one launch (either set may be empty: then its handle stays 0)
int32_t fold_prepare(const uint64_t *p1, size_t n1, uint64_t *h1, const uint64_t *p2, size_t n2, uint64_t *h2) {
    if ((n1 && (!p1 || !h1)) || (n2 && (!p2 || !h2)) || n1 + n2 == 0 || n1 >= (1ull << 24) || n2 >= (1ull << 24)) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    const size_t bytes1 = n1 * fold_words(false) * 4 + n1, bytes2 = n2 * fold_words(true) * 4 + n2;
    size_t got1 = 0, got2 = 0;
    void *tab1 = nullptr, *tab2 = nullptr;
    auto drop = [&] { if (tab1) fold_release(cur_index(), tab1, got1); if (tab2) fold_release(cur_index(), tab2, got2); };
    {
        SLOT_ACQUIRE(L, sl);
        HIPCHK(hipSetDevice(cur().device));
        int32_t rc;
        if ((rc = sl.in_bases.ensure(n1 * 96 + n2 * 192))) return rc;
        if (n1 && !(tab1 = fold_alloc(bytes1, got1))) return DGPU_E_OOM;
        if (n2 && !(tab2 = fold_alloc(bytes2, got2))) { drop(); return DGPU_E_OOM; }
        hipStream_t s = sl.stream;
        rc = DGPU_OK;
        uint8_t *d1 = sl.in_bases.as<uint8_t>(), *d2 = d1 + n1 * 96;
        if (n1 && hipMemcpyAsync(d1, p1, n1 * 96, hipMemcpyHostToDevice, s) != hipSuccess) rc = DGPU_E_HIP;
        if (!rc && n2 && hipMemcpyAsync(d2, p2, n2 * 192, hipMemcpyHostToDevice, s) != hipSuccess) rc = DGPU_E_HIP;
        if (!rc) { StageTimer st(sl, "fixed.fold_chain");
                   msm::launch_fold_chain(s, (const uint32_t *)d1, n1, (uint32_t *)tab1, (uint8_t *)tab1 + n1 * fold_words(false) * 4,
                                          (const uint32_t *)d2, n2, (uint32_t *)tab2, (uint8_t *)tab2 + n2 * fold_words(true) * 4); }
        if (!rc && (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) rc = DGPU_E_HIP;
        if (gs.prof) prof_flush(sl);
        if (rc) { gs.last_hip = (int32_t)hipGetLastError(); drop(); return rc; }
    }
    FoldTab *f1 = n1 ? new (std::nothrow) FoldTab{tab1, got1, n1} : nullptr, *f2 = n2 ? new (std::nothrow) FoldTab{tab2, got2, n2} : nullptr;
    if ((n1 && !f1) || (n2 && !f2)) { delete f1; delete f2; drop(); return DGPU_E_OOM; }
    if (n1) *h1 = register_handle(f1, n1, 13);
    if (n2) *h2 = register_handle(f2, n2, 14);
    return DGPU_OK;
}
int32_t fold_apply(bool g2, uint64_t handle, const uint64_t *scalar, const uint64_t *addend, uint64_t *out, uint8_t *out_inf) {
    if (!scalar || !out || !out_inf) return DGPU_E_BADARG;
    HandleRef hb(handle);
    if (!hb.ok || hb.h.kind != (g2 ? 14 : 13)) return DGPU_E_BADARG;
    const FoldTab &ft = *(const FoldTab *)hb.h.p;
    const size_t n = ft.n, pt = g2 ? 192 : 96, xw = g2 ? 96 : 48;
    // the set bits of the split scalar name the table entries: G1 k = k1 + k2 lambda (entry k of k1, entry 128 + k — phi of entry k — of k2);
    // G2 four base-|x| digits (entry 64 j + k of digit j)
    uint16_t leaves[256]; int T = 0;
    uint64_t split[4];
    if (g2) hostf::gls4_decompose(scalar, split); else hostf::glv_decompose(scalar, &split[0], &split[2]);
    for (int w = 0; w < 4; w++) for (int b = 0; b < 64; b++) if ((split[w] >> b) & 1) leaves[T++] = (uint16_t)(64 * w + b);
    CtxScope on_owner(hb.h.ctx);
    if (!cur().ready) return DGPU_E_NODEVICE;
    SLOT_ACQUIRE(L, sl);
    HIPCHK(hipSetDevice(cur().device));
    int32_t rc;
    // in_bases: the addends; in_scalars: the leaf list; prepped: [sum as XYZZ | affine out | flags]
    if ((rc = sl.in_bases.ensure(n * pt))) return rc;
    if ((rc = sl.in_scalars.ensure(512))) return rc;
    if ((rc = sl.prepped.ensure(n * (xw * 4 + pt + 1)))) return rc;
    hipStream_t s = sl.stream;
    const uint32_t *dadd = nullptr;
    if (addend) { HIPCHK(hipMemcpyAsync(sl.in_bases.p, addend, n * pt, hipMemcpyHostToDevice, s)); dadd = sl.in_bases.as<uint32_t>(); }
    HIPCHK(hipMemcpyAsync(sl.in_scalars.p, leaves, 512, hipMemcpyHostToDevice, s));
    uint32_t *dx = sl.prepped.as<uint32_t>(), *dout = dx + n * xw; uint8_t *dinf = (uint8_t *)(dout + n * pt / 4);
    { StageTimer st(sl, "fixed.fold_apply");
      msm::launch_fold_apply(s, g2, (const uint32_t *)ft.tab, (const uint8_t *)ft.tab + n * fold_words(g2) * 4, sl.in_scalars.as<uint16_t>(), T, dadd, n, dx, dinf, dout); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout, n * pt, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_inf, dinf, n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (gs.prof) prof_flush(sl);
    return DGPU_OK;
}

}  // namespace

extern "C" {
int32_t dgpu_g1_mul_add_batch(const uint64_t *p, const uint8_t *p_inf, const uint64_t *sc, size_t stride, const uint64_t *add, const uint8_t *add_inf, size_t n, uint64_t *out, uint8_t *out_inf) {
    return mul_add<G1>(p, p_inf, sc, stride, add, add_inf, n, out, out_inf);
}
int32_t dgpu_g2_mul_add_batch(const uint64_t *p, const uint8_t *p_inf, const uint64_t *sc, size_t stride, const uint64_t *add, const uint8_t *add_inf, size_t n, uint64_t *out, uint8_t *out_inf) {
    return mul_add<G2>(p, p_inf, sc, stride, add, add_inf, n, out, out_inf);
}
// (nothing unwinds through the ABI: the pool and the handle table allocate)
int32_t dgpu_g1_fold_prepare(const uint64_t *p, size_t n, uint64_t *handle) { return abi_guard([&] { return n ? fold_prepare(p, n, handle, nullptr, 0, nullptr) : (int32_t)DGPU_E_BADARG; }); }
int32_t dgpu_g2_fold_prepare(const uint64_t *p, size_t n, uint64_t *handle) { return abi_guard([&] { return n ? fold_prepare(nullptr, 0, nullptr, p, n, handle) : (int32_t)DGPU_E_BADARG; }); }
int32_t dgpu_fold_prepare_pair(const uint64_t *g1_xy, size_t n1, uint64_t *g1_handle, const uint64_t *g2_xy, size_t n2, uint64_t *g2_handle) {
    if (g1_handle) *g1_handle = 0;
    if (g2_handle) *g2_handle = 0;
    return abi_guard([&] { return fold_prepare(g1_xy, n1, g1_handle, g2_xy, n2, g2_handle); });
}
int32_t dgpu_g1_fold_apply(uint64_t handle, const uint64_t scalar[4], const uint64_t *addend, uint64_t *out, uint8_t *out_inf) { return abi_guard([&] { return fold_apply(false, handle, scalar, addend, out, out_inf); }); }
int32_t dgpu_g2_fold_apply(uint64_t handle, const uint64_t scalar[4], const uint64_t *addend, uint64_t *out, uint8_t *out_inf) { return abi_guard([&] { return fold_apply(true, handle, scalar, addend, out, out_inf); }); }
int32_t dgpu_fold_free(uint64_t handle) {
    return abi_guard([&]() -> int32_t {
    Handle h;
    if (!take_handle(handle, [](int k) { return k == 13 || k == 14; }, h)) return DGPU_E_BADARG;
    FoldTab *ft = (FoldTab *)h.p;
    { CtxScope on_owner(h.ctx); if (cur().device >= 0) (void)hipSetDevice(cur().device); fold_release(h.ctx, ft->tab, ft->bytes); }
    delete ft;
    return DGPU_OK;
    });
}
int32_t dgpu_window_table_g1(const uint64_t base_xy[12], uint64_t *handle) { return table_build<G1, hostf::Fq>(base_xy, handle, 5); }
int32_t dgpu_window_table_g2(const uint64_t base_xy[24], uint64_t *handle) { return table_build<G2, hostf::Fq2>(base_xy, handle, 6); }
int32_t dgpu_window_table_free(uint64_t handle) {
    Handle h;
    if (!lookup_handle(handle, h) || (h.kind != 5 && h.kind != 6)) return DGPU_E_BADARG;
    return dgpu_bases_free(handle);
}
int32_t dgpu_window_table_mul_g1(uint64_t t, const uint64_t *s, size_t n, int32_t mont, uint64_t *out, uint8_t *out_inf) { return table_mul<G1>(t, s, n, mont, out, out_inf, 5); }
int32_t dgpu_window_table_mul_g2(uint64_t t, const uint64_t *s, size_t n, int32_t mont, uint64_t *out, uint8_t *out_inf) { return table_mul<G2>(t, s, n, mont, out, out_inf, 6); }
int32_t dgpu_window_table_mul_to_bases_g1(uint64_t t, const uint64_t *s, size_t n, int32_t mont, uint64_t *bases) { return bases ? table_mul<G1>(t, s, n, mont, nullptr, nullptr, 5, bases) : DGPU_E_BADARG; }
int32_t dgpu_window_table_mul_to_bases_g2(uint64_t t, const uint64_t *s, size_t n, int32_t mont, uint64_t *bases) { return bases ? table_mul<G2>(t, s, n, mont, nullptr, nullptr, 6, bases) : DGPU_E_BADARG; }
int32_t dgpu_fixed_base_g1(const uint64_t base_xy[12], const uint64_t *s, size_t n, int32_t mont, uint64_t *out, uint8_t *out_inf) { return fixed_base<G1, hostf::Fq>(base_xy, s, n, mont, out, out_inf, 5); }
int32_t dgpu_fixed_base_g2(const uint64_t base_xy[24], const uint64_t *s, size_t n, int32_t mont, uint64_t *out, uint8_t *out_inf) { return fixed_base<G2, hostf::Fq2>(base_xy, s, n, mont, out, out_inf, 6); }
}  // extern "C"
