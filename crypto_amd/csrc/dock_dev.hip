// crypto_amd/csrc/dock_dev.hip — the DEVELOPMENT surface (include/dock_gpu_dev.h): tuning knobs, stage timers and self-test hooks.  Linked into
// crypto_amd/libdock_gpu_dev.so only (the twin the tests, tools/ and bench.py's stage / roofline leg load); the product library libdock_gpu.so
// exports none of these symbols (tests/test_abi_host.py asserts both) and runs every knob at its default.
#include "msm_driver.hip.h"
#include "../../include/dock_gpu_dev.h"

using namespace dock;

extern "C" {

int32_t dgpu_set_window_bits(int32_t c) { if (c != 0 && (c < 7 || c > 22)) return DGPU_E_BADARG; gs.window_bits = c; return DGPU_OK; }
int32_t dgpu_set_chunk(int32_t terms) { if (terms != 0 && (terms < 16 || terms > 4096)) return DGPU_E_BADARG; gs.chunk = terms; return DGPU_OK; }
int32_t dgpu_set_reduce_lanes(int32_t lanes) { if (lanes != 0 && lanes != 1 && lanes != 2 && lanes != 4) return DGPU_E_BADARG; gs.reduce_lanes = lanes; return DGPU_OK; }
int32_t dgpu_set_reduce_shift(int32_t sh) { if (sh < -1 || sh > 6) return DGPU_E_BADARG; gs.reduce_shift = sh; return DGPU_OK; }
int32_t dgpu_set_miller_pipeline(int32_t mode) {       // bits 0-4: forms; bits 8-13 / 16-21: where the chain is cut (0: default); bits 24-27: slice length of the last piece's products (0: automatic); bits 28-29: log2 of the factor on the block limit of k_line_products3
    if (mode < 0 || (mode & ~0x3F3F3F1F)) return DGPU_E_BADARG;
    const int a = (mode >> 8) & 63, b = (mode >> 16) & 63;
    if ((a || b) && !(a > b && b > 0 && a < 62)) return DGPU_E_BADARG;
    gs.ml_mode = mode; return DGPU_OK;
}

int32_t dgpu_prof_enable(int32_t on) { gs.prof = on != 0; return DGPU_OK; }
int32_t dgpu_prof_reset(void) { std::lock_guard<std::mutex> lk(gs.mu); gs.prof_tab.clear(); gs.allocs_at_reset = g_dev_allocs.load(); gs.alloc_ns_at_reset = g_dev_alloc_ns.load(); return DGPU_OK; }
int32_t dgpu_prof_read(const char **names, double *total_ms, uint64_t *calls, int32_t cap) {
    std::lock_guard<std::mutex> lk(gs.mu);
    int32_t k = 0;
    for (auto &t : gs.prof_tab) { if (k >= cap) break; names[k] = t.name; total_ms[k] = t.ms; calls[k] = t.calls; k++; }
    // device allocations since the last dgpu_prof_reset (counted whether or not the stage timers are enabled): 0 calls in steady state
    if (k < cap) { names[k] = "hipMalloc"; total_ms[k] = (double)(g_dev_alloc_ns.load() - gs.alloc_ns_at_reset) * 1e-6; calls[k] = g_dev_allocs.load() - gs.allocs_at_reset; k++; }
    return k;
}

int32_t dgpu_selftest_fp_mul(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out) {
    if (!cur().ready) return DGPU_E_NODEVICE;
    SLOT_ACQUIRE(L, sl);
    HIPCHK(hipSetDevice(cur().device));
    void *da, *db, *dout;
    HIPCHK(dev_malloc(&da, n * 48 + 16)); HIPCHK(dev_malloc(&db, n * 48 + 16)); HIPCHK(dev_malloc(&dout, n * 48 + 16));
    HIPCHK(hipMemcpy(da, a, n * 48, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(db, b, n * 48, hipMemcpyHostToDevice));
    launch_selftest_fp_mul(sl.stream, (const uint32_t *)da, (const uint32_t *)db, n, (uint32_t *)dout);
    HIPCHK(hipStreamSynchronize(sl.stream));
    HIPCHK(hipMemcpy(out, dout, n * 48, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return DGPU_OK;
}
int32_t dgpu_selftest_g1_sum(const uint64_t *pts, const uint8_t *neg, size_t n, uint64_t out[18]) {
    if (!cur().ready) return DGPU_E_NODEVICE;
    SLOT_ACQUIRE(L, sl);
    HIPCHK(hipSetDevice(cur().device));
    void *dp, *dn, *dout, *dinf;
    HIPCHK(dev_malloc(&dp, n * 96 + 16)); HIPCHK(dev_malloc(&dn, n + 16)); HIPCHK(dev_malloc(&dout, 4 * 48)); HIPCHK(dev_malloc(&dinf, 16));
    HIPCHK(hipMemcpy(dp, pts, n * 96, hipMemcpyHostToDevice));
    if (neg) HIPCHK(hipMemcpy(dn, neg, n, hipMemcpyHostToDevice)); else HIPCHK(hipMemset(dn, 0, n + 16));
    launch_selftest_g1_sum(sl.stream, (const uint32_t *)dp, (const uint8_t *)dn, n, (uint32_t *)dout, (uint8_t *)dinf);
    HIPCHK(hipStreamSynchronize(sl.stream));
    uint64_t w[24]; uint8_t inf;
    HIPCHK(hipMemcpy(w, dout, 4 * 48, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(&inf, dinf, 1, hipMemcpyDeviceToHost));
    (void)hipFree(dp); (void)hipFree(dn); (void)hipFree(dout); (void)hipFree(dinf);
    uint8_t finf = inf;
    host_fold<hostf::Fq>(w, &finf, 1, 1, out);
    return DGPU_OK;
}
// host self-test hook: the GLV split the scaling kernel is fed with (k mod r = k1 + k2 lambda, both < 2^128)
int32_t dgpu_selftest_glv_decompose(const uint64_t k[4], uint64_t k1[2], uint64_t k2[2]) {
    if (!k || !k1 || !k2) return DGPU_E_BADARG;
    hostf::glv_decompose(k, k1, k2);
    return DGPU_OK;
}

}  // extern "C"
