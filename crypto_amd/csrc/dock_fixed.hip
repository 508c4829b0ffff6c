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
      else if constexpr (C::NFP == 1) msm::launch_g1_scale(s, (const uint32_t *)dp, dpinf, sl.in_scalars.as<uint32_t>(), (int)(scalar_stride * 2), nullptr, n, sl.prepped.as<uint32_t>(), dout_inf, dadd, dainf);
      else msm::launch_mul_add_g2_gls(s, (const uint32_t *)dp, dpinf, sl.in_scalars.as<uint32_t>(), (int)(scalar_stride * 2), dadd, dainf, n, sl.prepped.as<uint32_t>(), dout_inf); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, sl.prepped.p, n * pt, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_inf, dout_inf, n, hipMemcpyDeviceToHost, s));
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
