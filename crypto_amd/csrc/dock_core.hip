// crypto_amd/csrc/dock_core.hip — lifecycle, handles, error strings and instrumentation of libdock_gpu.so
// (implements the curve-independent part of include/dock_gpu.h).
#include "dock_ctx.hpp"
#include "host_field.hpp"
#include "qap_launch.cuh"

namespace dock {
Ctx g;

int choose_c(size_t n, bool g2) {
    if (g.window_bits >= 7 && g.window_bits <= 22) return g.window_bits;
    const char *e = getenv("DGPU_WINDOW_BITS");
    if (e) { int v = atoi(e); if (v >= 7 && v <= 22) return v; }
    // Window width by size, from sweeps on MI355X (tests/perf/c_sweep.py, one call in flight).  c <= 16: digit codes stay 2 bytes and the
    // LDS counting sort sweeps W * RANGES * n * 2 B (at c = 20 the 4-byte codes and 32 ranges cost more than the saved additions).
    // From 2^17 terms on 16 wins clearly (2^18: 1.99 ms vs 2.50 at c = 14; 2^19: 2.81 vs 3.83): the bucket reduction is a latency-bound
    // tail whose length hardly depends on the bucket count, while every window saved removes n additions.  Below that the tail
    // dominates and narrow windows with few buckets win (2^14: c = 9 for G1, c = 8 for G2, whose additions are 3x dearer).
    int lg = 0; while (((size_t)1 << lg) < n && lg < 40) lg++;          // ceil(log2 n)
    if (lg > 0 && n - ((size_t)1 << (lg - 1)) < ((size_t)1 << lg) - n) lg--;   // round to the nearer power of two
    int bc;
    if (lg >= 17) bc = 16;
    else if (lg == 16) bc = 12;
    else if (lg == 15) bc = 10;
    else if (lg == 14) bc = g2 ? 8 : 9;
    else bc = 8;
    return bc;
}
int choose_chunk(size_t E, int min_chunk, size_t max_chunks) {
    if (g.chunk) return g.chunk;
    const char *e = getenv("DGPU_CHUNK");
    if (e) { int v = atoi(e); if (v >= 16 && v <= 4096) return v; }
    // terms per lane.  A lane's chunk is one dependent chain of mixed additions (~12 us each), so short chunks win as long as the
    // partial slots they create stay cheap to fold: 16 terms up to ~300 k lanes (two rounds of the chip's 131 072 lanes at
    // 2 waves/SIMD), then doubling — measured at n = 2^12 .. 2^18: 16 beats 64 by 13-35 % (tools: DGPU_CHUNK sweep), at 2^20 64 and
    // 128 tie, and at n = 2^24 a fixed 64 left 7 of 8 chunks inside one bucket and the fix-up pass ran at 1/8 lane efficiency.
    int ch = min_chunk;       // 16 for G1; 32 for G2, whose partial slots are folded by the one-lane Fp2 addition of k_fixup
    while (E / (size_t)ch > max_chunks && ch < 4096) ch *= 2;      // G1: 300 k chunks; G2 (two lanes per chunk): 150 k
    return ch;
}

int32_t upload_scalars(Slot &sl, const uint64_t *h, size_t n, bool mont, uint32_t *d_out) {
    HIPCHK(hipMemcpyAsync(d_out, h, n * 32, hipMemcpyHostToDevice, sl.stream));
    if (mont) ntt::launch_fr_mont_to_canonical(sl.stream, d_out, n);     // Fr::into_bigint on the device (ark-ec msm_unchecked does it on rayon)
    HIPCHK(hipStreamSynchronize(sl.stream));
    return DGPU_OK;
}


}  // namespace dock
using namespace dock;

extern "C" {

int32_t dgpu_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }

int32_t dgpu_init(int32_t device) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.ready) return g.device == device ? DGPU_OK : DGPU_E_BADARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return DGPU_E_NODEVICE; }
    if (device < 0 || device >= n) return DGPU_E_NODEVICE;
    HIPCHK(hipSetDevice(device));
    for (int i = 0; i < N_SLOTS; i++) HIPCHK(hipStreamCreateWithFlags(&g.slots[i].stream, hipStreamNonBlocking));
    g.device = device; g.ready = true;
    return DGPU_OK;
}

int32_t dgpu_shutdown(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.ready) return DGPU_OK;
    g.ready = false;
    (void)hipSetDevice(g.device);
    for (int i = 0; i < N_SLOTS; i++) {
        std::lock_guard<std::mutex> sk(g.slots[i].mu);
        (void)hipStreamSynchronize(g.slots[i].stream);
        g.slots[i].release_all();
        (void)hipStreamDestroy(g.slots[i].stream);
        g.slots[i].stream = nullptr;
    }
    for (auto &h : g.handles) if (h.second.kind != 4) (void)hipFree(h.second.p);   // (R1CS handles own several allocations: freed by dgpu_r1cs_free)
    g.handles.clear();
    for (auto &d : g.ntt_domains) { void *ps[] = {d.second.tw_f, d.second.tw_i, d.second.pw_f, d.second.pw_i, d.second.zinv}; for (void *p : ps) if (p) (void)hipFree(p); }
    g.ntt_domains.clear();
    g.prof_tab.clear();
    g.device = -1;
    return DGPU_OK;
}

const char *dgpu_strerror(int32_t code) {
    switch (code) {
        case DGPU_OK: return "ok";
        case DGPU_E_NODEVICE: return "no usable HIP device (dgpu_init not called or failed)";
        case DGPU_E_OOM: return "out of memory";
        case DGPU_E_BADARG: return "bad argument";
        case DGPU_E_HIP: return "HIP runtime error";
        case DGPU_E_ZERO: return "final exponentiation of zero";
        case DGPU_E_TOO_SMALL: return "n below the GPU threshold";
        case DGPU_E_LENGTH: return "length mismatch";
        default: return "unknown error";
    }
}
int32_t dgpu_last_hip_error(void) { return g.last_hip.load(); }
int32_t dgpu_set_min_gpu_n(size_t n) { g.min_gpu_n = n; return DGPU_OK; }
int32_t dgpu_set_window_bits(int32_t c) { if (c != 0 && (c < 7 || c > 22)) return DGPU_E_BADARG; g.window_bits = c; return DGPU_OK; }


static int32_t free_handle(uint64_t h, bool scalars) {
    std::lock_guard<std::mutex> lk(g.mu);
    auto it = g.handles.find(h);
    if (it == g.handles.end() || it->second.kind == 4 || ((it->second.kind == 3) != scalars)) return DGPU_E_BADARG;
    if (g.ready) { (void)hipSetDevice(g.device); (void)hipDeviceSynchronize(); }
    (void)hipFree(it->second.p);
    g.handles.erase(it);
    return DGPU_OK;
}
int32_t dgpu_bases_free(uint64_t h) { return free_handle(h, false); }
int32_t dgpu_scalars_free(uint64_t h) { return free_handle(h, true); }
int32_t dgpu_scalars_upload(const uint64_t *s, size_t n, int32_t mont, uint64_t *handle) {
    if (!handle || (n && !s)) return DGPU_E_BADARG;
    if (!g.ready) return DGPU_E_NODEVICE;
    void *p = nullptr;
    {
        SlotLock L; Slot &sl = *L.s;
        HIPCHK(hipSetDevice(g.device));
        if (hipMalloc(&p, std::max<size_t>(n, 1) * 32) != hipSuccess) { (void)hipGetLastError(); return DGPU_E_OOM; }
        int32_t rc = n ? upload_scalars(sl, s, n, mont != 0, (uint32_t *)p) : DGPU_OK;
        if (rc) { (void)hipFree(p); return rc; }
    }
    std::lock_guard<std::mutex> lk(g.mu);
    uint64_t h = g.next_handle++;
    g.handles[h] = Handle{p, n, 3};
    *handle = h;
    return DGPU_OK;
}


// several host arrays -> one resident scalar vector (the prover's `assignment` = inputs[1..] ++ witnesses, prover.rs:319-321,
// without a host-side concatenation)
int32_t dgpu_scalars_upload_parts(const uint64_t *const *parts, const size_t *counts, size_t n_parts, int32_t mont, uint64_t *handle) {
    if (!handle || (n_parts && (!parts || !counts))) return DGPU_E_BADARG;
    size_t n = 0;
    for (size_t k = 0; k < n_parts; k++) { if (counts[k] && !parts[k]) return DGPU_E_BADARG; n += counts[k]; }
    if (!g.ready) return DGPU_E_NODEVICE;
    void *p = nullptr;
    {
        SlotLock L; Slot &sl = *L.s;
        HIPCHK(hipSetDevice(g.device));
        if (hipMalloc(&p, std::max<size_t>(n, 1) * 32) != hipSuccess) { (void)hipGetLastError(); return DGPU_E_OOM; }
        size_t at = 0;
        hipError_t e = hipSuccess;
        for (size_t k = 0; k < n_parts && e == hipSuccess; k++) {
            if (counts[k]) e = hipMemcpyAsync((uint8_t *)p + at * 32, parts[k], counts[k] * 32, hipMemcpyHostToDevice, sl.stream);
            at += counts[k];
        }
        if (e == hipSuccess && mont && n) ntt::launch_fr_mont_to_canonical(sl.stream, (uint32_t *)p, n);
        if (e == hipSuccess) e = hipStreamSynchronize(sl.stream);
        if (e != hipSuccess) { g.last_hip = (int32_t)e; (void)hipGetLastError(); (void)hipFree(p); return DGPU_E_HIP; }
    }
    std::lock_guard<std::mutex> lk(g.mu);
    uint64_t h = g.next_handle++;
    g.handles[h] = Handle{p, n, 3};
    *handle = h;
    return DGPU_OK;
}

int32_t dgpu_prof_enable(int32_t on) { g.prof = on != 0; return DGPU_OK; }
int32_t dgpu_prof_reset(void) { std::lock_guard<std::mutex> lk(g.mu); g.prof_tab.clear(); return DGPU_OK; }
int32_t dgpu_prof_read(const char **names, double *total_ms, uint64_t *calls, int32_t cap) {
    std::lock_guard<std::mutex> lk(g.mu);
    int32_t k = 0;
    for (auto &t : g.prof_tab) { if (k >= cap) break; names[k] = t.name; total_ms[k] = t.ms; calls[k] = t.calls; k++; }
    return k;
}

}  // extern "C"
