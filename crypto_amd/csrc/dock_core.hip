// crypto_amd/csrc/dock_core.hip — lifecycle, handles, error strings and instrumentation of libdock_gpu.so
// (implements the curve-independent part of include/dock_gpu.h).
#include <cstdlib>
#include "dock_ctx.hpp"
#include "bases_cache.hpp"
#include "host_field.hpp"
#include "qap_launch.hip.h"
#include <chrono>
#include <thread>

namespace dock {
Shared gs;
BasesCache gcache;
#ifdef DGPU_DEV
// the twin's handles start far from the product's: a process that has both libraries loaded (tests, bench.py) cannot free or use one library's
// object through the other by accident — the foreign number is simply unknown there (DGPU_E_BADARG)
static const bool g_dev_handle_base = (gs.next_handle = 1ull << 40, true);
#endif
Ctx ctxs[MAX_CTX];
thread_local int tl_ctx = -1;
thread_local bool tl_no_min = false;
std::atomic<uint64_t> g_dev_allocs{0}, g_dev_alloc_ns{0}, g_dev_alloc_bytes{0};
#ifdef DGPU_DEV
// fault injection (development build only: `make dev` -> libdock_gpu_dev.so; tests/test_gpu_fault_paths.py): the k-th hipMalloc from now and the
// count - 1 after it fail (Buf::ensure retries a failed allocation once with the exact size: count = 2 defeats the retry, count = 1 exercises it)
static std::atomic<int64_t> g_fail_after{-1}, g_fail_count{0}, g_fail_left{0};
extern "C" __attribute__((visibility("default"))) int32_t dgpu_dev_fail_alloc_after(int64_t k, int64_t count) { g_fail_left = 0; g_fail_count = count; g_fail_after = k; return DGPU_OK; }
#endif
#ifdef DGPU_DEV
static bool injected_failure() {
    if (g_fail_left.load() > 0) return g_fail_left.fetch_sub(1) > 0;
    if (g_fail_after.load() >= 0 && g_fail_after.fetch_sub(1) == 0) { g_fail_left = g_fail_count.load() - 1; return true; }
    return false;
}
#endif
static hipError_t raw_malloc(void **p, size_t bytes) {
#ifdef DGPU_DEV
    if (injected_failure()) { *p = nullptr; return hipErrorOutOfMemory; }
#endif
    return hipMalloc(p, bytes);
}
hipError_t dev_malloc(void **p, size_t bytes) {
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = raw_malloc(p, bytes);
    // The resident-bases cache holds memory nobody asked for by name: when the device is full it gives way.  Least recently used entries leave (an entry a call is
    // using goes when that call returns) until something was released, then the allocation is tried again — a host's explicit upload, a workspace or a table
    // build never fails because cached keys are in the way.  (No lock is held here that an entry's release takes: callers hold at most a slot.  The release
    // may select another device: the caller's is restored.)
    while (e != hipSuccess) {
        (void)hipGetLastError();
        int dev = -1; (void)hipGetDevice(&dev);
        const bool released = cache_release_lru(bytes);
        if (dev >= 0) (void)hipSetDevice(dev);
        if (!released) break;
        e = raw_malloc(p, bytes);
    }
    g_dev_alloc_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    g_dev_allocs++; if (e == hipSuccess) g_dev_alloc_bytes += bytes;
    return e;
}

int choose_c(size_t n, bool g2) {
    { const int wb = gs.window_bits.load(); if (wb >= 7 && wb <= 22) return wb; }
#ifdef DGPU_DEV
    { const char *e = getenv("DGPU_WINDOW_BITS"); if (e) { int v = atoi(e); if (v >= 7 && v <= 22) return v; } }     // (any width gives the same point)
#endif
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
// a chunk length forced by a tuning knob (dgpu_set_chunk; DGPU_CHUNK in -DDGPU_DEV builds only: any chunk length gives the same point, tests/test_gpu_msm.py sweeps it), else 0
int forced_chunk() {
    if (int f = gs.chunk.load()) return f;
#ifdef DGPU_DEV
    const char *e = getenv("DGPU_CHUNK");
    if (e) { int v = atoi(e); if (v >= 16 && v <= 4096) return v; }
#endif
    return 0;
}
int choose_chunk(size_t E, int min_chunk, size_t max_chunks, int lanes_per_chunk, size_t nb_shared) {
    if (int f = forced_chunk()) return f;
    // terms per lane.  A lane's chunk is one dependent chain of mixed additions (~12 us each), so short chunks win as long as the
    // partial slots they create stay cheap to fold: 16 terms up to ~300 k lanes (two rounds of the chip's 131 072 lanes at
    // 2 waves/SIMD), then doubling — measured at n = 2^12 .. 2^18: 16 beats 64 by 13-35 % (tools: DGPU_CHUNK sweep), at 2^20 64 and
    // 128 tie, and at n = 2^24 a fixed 64 left 7 of 8 chunks inside one bucket and the fix-up pass ran at 1/8 lane efficiency.
    int ch = min_chunk;       // 16 for G1; 32 for G2, whose partial slots are folded by the one-lane Fp2 addition of k_fixup
    while (E / (size_t)ch > max_chunks && ch < 4096) ch *= 2;      // G1: 300 k chunks; G2 (two lanes per chunk): 150 k
    // Whole rounds of the chip: k_accumulate keeps 2 waves per SIMD = 131 072 lanes resident, and a launch lasts as long as its rounds,
    // full or not (13 windows x 2^20 terms in chunks of 64 are 1.62 rounds and took the time of 2).  Once the launch exceeds one round the
    // chunk count is snapped to a multiple of the resident chunks and the chunk length follows from it (need not be a power of two).
    const size_t resident = (size_t)131072 / (size_t)lanes_per_chunk;
    const size_t chunks = (E + ch - 1) / ch;
    if (chunks > resident) {
        size_t rounds = (chunks + resident / 2) / resident;
        if (rounds < 1) rounds = 1;
        size_t len = (E + rounds * resident - 1) / (rounds * resident);
        if (len >= 16 && len <= 4096) ch = (int)len;
    }
    // The table pipeline (ONE set of nb_shared buckets for all windows): runs are E / nb_shared terms long on average, and every chunk that ends
    // inside a run leaves a partial for the fix-up.  When the average run is at least HALF the chunk of ONE whole round of the chip (the
    // proportion of the tuned dense case: runs of 26, chunks of 52), the launch is one round instead of two: a witness-shaped MSM on a width-17 table (4.4 M terms, 2^16 buckets: runs of 68) goes from chunks of 17
    // to 34 (fix-up 0.49 -> 0.33 ms).  NOT further: chunks of 68 are one wave per SIMD (65 536 lanes = one 256-thread block per CU), where a lone
    // wave leaves the gather's latency uncovered (dense scalars on that table: 4.5 ms against 2.3) and one lane too many puts a second block on
    // some CUs and doubles the kernel (chunks of 66: 1.12 ms against 0.74 for 68; tests/perf/witness_msm_perf.py CHUNK=).  Dense scalars on a
    // width-20 table (runs of 26 < chunks of 52) are untouched.
    if (nb_shared) {
        const size_t run = E / nb_shared, one = (E + resident - 1) / resident;
        if (one > (size_t)ch && 2 * run >= one && one <= 4096) ch = (int)one;
    }
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

// dgpu_runtime_hints — process-level settings of the ROCm runtime that suit this library, applied ONLY when the host asks (the library itself never touches
// its host's environment).  DGPU_HINT_EIGHT_HW_QUEUES: ROCm reads GPU_MAX_HW_QUEUES when the runtime comes up (the process's first HIP call) and maps every
// stream onto that many hardware queues, 4 by default.  A slot has three streams and six slots run at once; two streams on one queue execute one after the
// other, which is what the latency chains of the pairing and aggregation paths (line chain | scalings | product pieces side by side) cannot afford.  Eight
// queues, same-box A/B of the whole bench line (profiles/r05_hwq_bench_ab.txt): 1024-pair Miller loop 0.71 -> 0.66 ms, 1024 proofs aggregated 37.1 -> 35.3 ms,
// one proof 10.0 -> 9.8 ms, the MSM rate unchanged.  The hint exports GPU_MAX_HW_QUEUES=8 unless the variable is already set.  It works only BEFORE the process's
// first HIP call and must be made while no other thread can be inside getenv / setenv (a host's start-up code): DGPU_E_BADARG once this library has touched HIP.
static std::atomic<bool> g_hip_touched{false};
int32_t dgpu_runtime_hints(uint32_t flags) {
    if (flags & ~(uint32_t)DGPU_HINT_EIGHT_HW_QUEUES) return DGPU_E_BADARG;
    if (g_hip_touched.load()) return DGPU_E_BADARG;
    if (flags & DGPU_HINT_EIGHT_HW_QUEUES) (void)setenv("GPU_MAX_HW_QUEUES", "8", 0);
    return DGPU_OK;
}
int32_t dgpu_device_count(void) { g_hip_touched = true; int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }

// streams, events, pinned scratch and workspaces of a context's slots (shutdown; a failed bring-up).  The caller holds each slot's mutex or
// knows that no call can reach the context.
static void destroy_slot(Slot &sl) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    if (sl.cstream) (void)hipStreamSynchronize(sl.cstream);
    if (sl.xstream) { (void)hipStreamSynchronize(sl.xstream); (void)hipStreamDestroy(sl.xstream); sl.xstream = nullptr; }
    sl.release_all();
    if (sl.stream) (void)hipStreamDestroy(sl.stream);
    if (sl.cstream) (void)hipStreamDestroy(sl.cstream);
    for (hipEvent_t &e : sl.copy_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (sl.hpin) (void)hipHostFree(sl.hpin);
    if (sl.hpin2) (void)hipHostFree(sl.hpin2);
    sl.hpin2 = nullptr; sl.hpin2_bytes = 0;
    sl.stream = nullptr; sl.cstream = nullptr; sl.hpin = nullptr;
}
static int32_t init_ctx_slots(Ctx &c) {
    // The runtime hands its hardware queues (GPU_MAX_HW_QUEUES, 4 unless the environment says otherwise) to streams in creation order, round
    // robin; kernels of two streams on one hardware queue run one after the other.  The six compute streams are created first, so that they
    // are spread over the queues as evenly as their number allows (created slot by slot next to their copy streams, the compute streams of
    // slots 0 / 5, 1 / 4 and 2 / 3 shared a queue and the prover's h-query MSM waited 2.7 ms behind another MSM's bucket reduction).
    for (int i = 0; i < N_SLOTS; i++) HIPCHK(hipStreamCreateWithFlags(&c.slots[i].stream, hipStreamNonBlocking));
    for (int i = 0; i < N_SLOTS; i++) HIPCHK(hipStreamCreateWithFlags(&c.slots[i].cstream, hipStreamNonBlocking));
    for (int i = 0; i < N_SLOTS; i++) HIPCHK(hipStreamCreateWithFlags(&c.slots[i].xstream, hipStreamNonBlocking));
    for (int i = 0; i < N_SLOTS; i++) {
        for (hipEvent_t &e : c.slots[i].copy_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHK(hipHostMalloc(&c.slots[i].hpin, Slot::HPIN_BYTES, hipHostMallocDefault));
        HIPCHK(c.slots[i].flags.ensure(64) ? hipErrorOutOfMemory : hipSuccess);
    }
    return DGPU_OK;
}
// bring up context `idx` on physical device `device` (gs.mu held)
static int32_t init_ctx_locked(int idx, int device) {
    Ctx &c = ctxs[idx];
    if (c.ready) return c.device == device ? DGPU_OK : DGPU_E_BADARG;
    if (!__builtin_cpu_supports("bmi2") || !__builtin_cpu_supports("adx")) return DGPU_E_NODEVICE;   // the host field code (host_field.hpp) is compiled for mulx / adcx / adox
    HIPCHK(hipSetDevice(device));
    { const int32_t rc = init_ctx_slots(c);
      if (rc) { for (int i = 0; i < N_SLOTS; i++) destroy_slot(c.slots[i]); return rc; } }      // nothing half-built is left behind
    // xGMI: peer access between this device and the ones already in use (best effort: without it hipMemcpyPeerAsync stages through the host)
    for (int k = 0; k < MAX_CTX; k++) {
        if (k == idx || !ctxs[k].ready || ctxs[k].device == device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device, ctxs[k].device) == hipSuccess && can) { (void)hipSetDevice(device); (void)hipDeviceEnablePeerAccess(ctxs[k].device, 0); }
        if (hipDeviceCanAccessPeer(&can, ctxs[k].device, device) == hipSuccess && can) { (void)hipSetDevice(ctxs[k].device); (void)hipDeviceEnablePeerAccess(device, 0); }
        (void)hipGetLastError();                       // (hipErrorPeerAccessAlreadyEnabled when two contexts share a pair of devices)
        (void)hipSetDevice(device);
    }
    c.device = device; c.ready = true;
    if (gs.default_ctx < 0) gs.default_ctx = idx;
    return DGPU_OK;
}
// logical context k runs on physical device physical[k].  The list may name a device more than once (two contexts, each with its own
// streams and workspaces, on one GPU): that is how the sharded entry points are exercised on a one-GPU box.
int32_t dgpu_init_device_list(const int32_t *physical, int32_t count) {
    if (!physical || count <= 0 || count > MAX_CTX) return DGPU_E_BADARG;
    std::lock_guard<std::mutex> lk(gs.mu);
    g_hip_touched = true;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return DGPU_E_NODEVICE; }
    for (int k = 0; k < count; k++) if (physical[k] < 0 || physical[k] >= n) return DGPU_E_NODEVICE;
    for (int k = 0; k < count; k++) { int32_t rc = init_ctx_locked(k, physical[k]); if (rc) return rc; }
    return DGPU_OK;
}
int32_t dgpu_init_devices(uint32_t mask) {
    int32_t phys[MAX_CTX]; int cnt = 0;
    for (int d = 0; d < 32 && cnt < MAX_CTX; d++) if (mask & (1u << d)) phys[cnt++] = d;
    return cnt ? dgpu_init_device_list(phys, cnt) : DGPU_E_BADARG;
}
// single-device form (one process per GPU): context 0 on `device`
int32_t dgpu_init(int32_t device) { return dgpu_init_device_list(&device, 1); }
int32_t dgpu_context_count(void) { int k = 0; for (int i = 0; i < MAX_CTX; i++) if (ctxs[i].ready) k++; return k; }
// the calling thread's context for the entry points that take host pointers (thread-local, like hipSetDevice); handle-based calls
// always run on the context that owns the handle
int32_t dgpu_set_device(int32_t ctx) {
    if (ctx < 0 || ctx >= MAX_CTX || !ctxs[ctx].ready) return DGPU_E_BADARG;
    tl_ctx = ctx; return DGPU_OK;
}

int32_t dgpu_shutdown(void) {
    cache_clear();                 // (the resident-bases cache's entries are handles like any other: released first, through dgpu_bases_free)
    // Lock order: a call in flight holds its slot and may take gs.mu (handle pins, the scalar pool); so the slots are drained WITHOUT gs.mu held.
    // 1. no new call gets past its `ready` check; 2. every slot is taken once, i.e. every call in flight has returned; 3. handles and pools
    // are released under gs.mu once nothing pins them.
    bool was_ready[MAX_CTX];
    {
        std::lock_guard<std::mutex> lk(gs.mu);
        bool any = false;
        for (int i = 0; i < MAX_CTX; i++) { was_ready[i] = ctxs[i].ready; any = any || was_ready[i]; ctxs[i].ready = false; }
        if (!any) return DGPU_OK;
    }
    for (int i = 0; i < MAX_CTX; i++) {
        Ctx &c = ctxs[i];
        if (!was_ready[i]) continue;
        (void)hipSetDevice(c.device);
        for (int k = 0; k < N_SLOTS; k++) {
            std::lock_guard<std::mutex> sk(c.slots[k].mu);
            destroy_slot(c.slots[k]);
        }
    }
    std::unique_lock<std::mutex> lk(gs.mu);
    gs.cv.wait(lk, [] { for (auto &h : gs.handles) if (h.second.inflight) return false; return true; });      // (pins are dropped by the calls that just returned)
    for (auto &h : gs.handles) {
        const Handle &hd = h.second;
        if (hd.ctx >= 0 && hd.ctx < MAX_CTX && ctxs[hd.ctx].device >= 0) (void)hipSetDevice(ctxs[hd.ctx].device);
        if (hd.kind == 4) free_r1cs_object(hd.p);                       // DevR1cs owns several allocations
        else if (hd.kind == 10 || hd.kind == 11) { PreTable *pt = (PreTable *)hd.p; (void)hipFree(pt->tab); delete pt; }
        else if (hd.kind == 12) { SortedScalars *ss = (SortedScalars *)hd.p; (void)hipFree(ss->off); (void)hipFree(ss->entries); delete ss; }
        else if (hd.kind == 13 || hd.kind == 14) { FoldTab *ft = (FoldTab *)hd.p; (void)hipFree(ft->tab); delete ft; }
        else if (hd.kind >= 7) delete (ShardSet *)hd.p;                 // its per-device parts are table entries of their own
        else { if (hd.aux) (void)hipFree(hd.aux); (void)hipFree(hd.p); }
    }
    gs.handles.clear();
    for (int i = 0; i < MAX_CTX; i++) {
        Ctx &c = ctxs[i];
        if (c.device >= 0) (void)hipSetDevice(c.device);
        for (auto &e : c.scalar_pool) (void)hipFree(e.first);
        c.scalar_pool.clear(); c.scalar_pool_bytes = 0;
        for (auto &e : c.fold_pool) (void)hipFree(e.first);
        c.fold_pool.clear();
        for (auto &d : c.ntt_domains) { void *ps[] = {d.second.tw_f, d.second.tw_i, d.second.pw_f, d.second.pw_i, d.second.zinv, d.second.pwr_f, d.second.pwr_i}; for (void *p : ps) if (p) (void)hipFree(p); }
        c.ntt_domains.clear();
        c.device = -1;
    }
    gs.prof_tab.clear();
    gs.default_ctx = -1;
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
int32_t dgpu_last_hip_error(void) { return gs.last_hip.load(); }
int32_t dgpu_set_min_gpu_n(size_t n) { gs.min_gpu_n = n; return DGPU_OK; }
int32_t dgpu_set_auto_shard_min_n(size_t n) { gs.auto_shard_min_n = n == 0 ? ~(size_t)0 : n; return DGPU_OK; }
size_t dgpu_get_min_gpu_n(void) { return gs.min_gpu_n.load(); }
int32_t dgpu_set_small_msm_max(size_t n) { if (n > 8192) return DGPU_E_BADARG; gs.small_max = n; return DGPU_OK; }
uint64_t dgpu_device_alloc_count(void) { return g_dev_allocs.load(); }

// ---- the resident-bases cache behind the one-shot MSM entry points (bases_cache.hpp) ----
int32_t dgpu_set_bases_cache_bytes(size_t bytes) {
    gcache.budget = bytes; gcache.enabled = bytes != 0;            // (DGPU_CACHE_BYTES_AUTO == CACHE_BUDGET_AUTO: resolved again at the next use)
    if (bytes == 0) { cache_clear(); return DGPU_OK; }
    std::vector<std::shared_ptr<CacheEntry>> dropped;
    std::lock_guard<std::mutex> lk(gcache.mu);
    (void)cache_make_room_locked(0, nullptr, dropped);         // a smaller budget: least recently used entries go (`dropped` outlives the lock: their handles are freed after it)
    return DGPU_OK;
}
int32_t dgpu_set_bases_cache_min_n(size_t n) { gcache.min_n = n < 2 ? 2 : n; return DGPU_OK; }
int32_t dgpu_set_bases_cache_verify(int32_t samples) {
    if (samples != DGPU_CACHE_VERIFY_FULL && (samples < 2 || samples > 4096)) return DGPU_E_BADARG;
    gcache.verify_samples = samples; return DGPU_OK;
}
int32_t dgpu_bases_cache_clear(void) { cache_clear(); return DGPU_OK; }
int32_t dgpu_bases_cache_invalidate(const void *p, size_t bytes) {
    if (!p) return DGPU_E_BADARG;
    std::vector<std::shared_ptr<CacheEntry>> dropped;
    std::unique_lock<std::mutex> lk(gcache.mu);
    for (size_t i = 0; i < gcache.entries.size();) {
        CacheEntry &c = *gcache.entries[i];
        if (c.state != CacheEntry::FILLING && c.k.overlaps(p, bytes)) { if (c.state == CacheEntry::READY) gcache.used -= c.bytes; dropped.push_back(std::move(gcache.entries[i])); gcache.entries.erase(gcache.entries.begin() + i); }
        else i++;
    }
    lk.unlock();
    return DGPU_OK;
}
int32_t dgpu_bases_cache_stats(uint64_t out[8]) {
    if (!out) return DGPU_E_BADARG;
    std::lock_guard<std::mutex> lk(gcache.mu);
    size_t ready = 0; for (auto &c : gcache.entries) if (c->state == CacheEntry::READY) ready++;
    out[0] = gcache.hits; out[1] = gcache.misses; out[2] = gcache.fills; out[3] = gcache.stale; out[4] = gcache.evictions;
    out[5] = gcache.used; out[6] = gcache.budget.load() == CACHE_BUDGET_AUTO ? 0 : gcache.budget.load(); out[7] = ready;
    return DGPU_OK;
}


// A free waits until no call uses the handle (HandleRef pins), then releases the memory outside the table lock: every entry point
// synchronises its stream before it returns, so nothing on the device still reads the allocation.
static void release_parts(const Handle &hd) {
    if (hd.kind >= 7 && hd.kind <= 9) {
        ShardSet *ss = (ShardSet *)hd.p;
        for (uint64_t sub : ss->sub) { Handle part; if (take_handle(sub, [](int) { return true; }, part)) release_parts(part); }
        delete ss;
        return;
    }
    CtxScope on_owner(hd.ctx);
    if (cur().device >= 0) (void)hipSetDevice(cur().device);
    if (hd.kind == 10 || hd.kind == 11) { PreTable *pt = (PreTable *)hd.p; (void)hipFree(pt->tab); delete pt; return; }
    if (hd.kind == 3) { scalar_release(hd.ctx, hd.p, scalar_bytes(hd.n)); return; }      // recycled: no device-wide wait per proof
    if (hd.kind == 12) { SortedScalars *ss = (SortedScalars *)hd.p; scalar_release(hd.ctx, ss->off, ss->off_bytes); scalar_release(hd.ctx, ss->entries, ss->entries_bytes); delete ss; return; }
    if (hd.aux) (void)hipFree(hd.aux);                 // the small-MSM table of a plain bases handle
    (void)hipFree(hd.p);
}
static int32_t free_handle(uint64_t h, bool scalars) {
    Handle hd;
    auto ok = [scalars](int k) { return k != 4 && k < 13 && ((k == 3 || k == 9 || k == 12) == scalars); };       // (13 / 14: dgpu_fold_free)
    if (!take_handle(h, ok, hd)) return DGPU_E_BADARG;
    release_parts(hd);
    return DGPU_OK;
}
// number of elements behind a bases / scalars / sorted-scalars handle (points, scalars), whatever its kind
int32_t dgpu_handle_len(uint64_t handle, size_t *n) {
    Handle hd;
    if (!n || !lookup_handle(handle, hd) || hd.kind == 5 || hd.kind == 6) return DGPU_E_BADARG;
    *n = hd.n; return DGPU_OK;                       // (a resident circuit: its number of constraints)
}
// the device context that owns a handle
int32_t dgpu_handle_context(uint64_t handle, int32_t *ctx) {
    Handle hd;
    if (!ctx || !lookup_handle(handle, hd)) return DGPU_E_BADARG;
    *ctx = hd.ctx; return DGPU_OK;
}
// layout of a sharded handle (dgpu_bases_upload_*_sharded, dgpu_scalars_upload_sharded): number of parts; part k = elements [lo, hi) as a
// handle of its own on context `ctx`.  count = 0 for a handle that is not sharded.
int32_t dgpu_shard_count(uint64_t handle, int32_t *count) {
    Handle hd;
    if (!count || !lookup_handle(handle, hd)) return DGPU_E_BADARG;
    *count = (hd.kind >= 7 && hd.kind <= 9) ? (int32_t)((const ShardSet *)hd.p)->sub.size() : 0;
    return DGPU_OK;
}
int32_t dgpu_shard_part(uint64_t handle, size_t k, uint64_t *sub, size_t *lo, size_t *hi, int32_t *ctx) {
    HandleRef ref(handle);
    if (!sub || !lo || !hi || !ctx || !ref.ok || ref.h.kind < 7 || ref.h.kind > 9) return DGPU_E_BADARG;
    const ShardSet &ss = *(const ShardSet *)ref.h.p;
    if (k >= ss.sub.size()) return DGPU_E_BADARG;
    Handle part; if (!lookup_handle(ss.sub[k], part)) return DGPU_E_BADARG;
    *sub = ss.sub[k]; *lo = ss.lo[k]; *hi = ss.lo[k + 1]; *ctx = part.ctx;
    return DGPU_OK;
}
int32_t dgpu_bases_free(uint64_t h) { return free_handle(h, false); }
int32_t dgpu_scalars_free(uint64_t h) { return free_handle(h, true); }
int32_t dgpu_scalars_upload(const uint64_t *s, size_t n, int32_t mont, uint64_t *handle) {
    if (!handle || (n && !s)) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    void *p = nullptr;
    {
        SLOT_ACQUIRE(L, sl);
        HIPCHK(hipSetDevice(cur().device));
        if (!(p = scalar_alloc(scalar_bytes(n)))) return DGPU_E_OOM;
        int32_t rc = n ? upload_scalars(sl, s, n, mont != 0, (uint32_t *)p) : DGPU_OK;
        if (rc) { scalar_release(cur_index(), p, scalar_bytes(n)); return rc; }
    }
    *handle = register_handle(p, n, 3);
    return DGPU_OK;
}


// scalars [lo, hi) of a resident vector as a vector of its own on context `dst_ctx`: device to device — over xGMI when the two contexts sit on
// different GPUs (hipMemcpyPeerAsync; peer access is enabled between the process's devices when they are brought up), a plain device copy
// when they share one.  The sharded prover hands every shard its slice of h this way instead of through the host (32 B x D down and up PCIe).
int32_t dgpu_scalars_copy_range(uint64_t src, size_t lo, size_t hi, int32_t dst_ctx, uint64_t *handle) {
    if (!handle || lo > hi || dst_ctx < 0 || dst_ctx >= MAX_CTX || !ctxs[dst_ctx].ready) return DGPU_E_BADARG;
    HandleRef hs(src);
    if (!hs.ok || hs.h.kind != 3 || hi > hs.h.n) return DGPU_E_BADARG;
    const size_t n = hi - lo;
    const int src_dev = ctxs[hs.h.ctx].device, dst_dev = ctxs[dst_ctx].device;
    CtxScope on_dst(dst_ctx);
    void *p = nullptr;
    {
        SLOT_ACQUIRE(L, sl);
        HIPCHK(hipSetDevice(dst_dev));
        if (!(p = scalar_alloc(scalar_bytes(n)))) return DGPU_E_OOM;
        hipError_t e = hipSuccess;
        if (n) {
            const char *from = (const char *)hs.h.p + lo * 32;
            e = src_dev == dst_dev ? hipMemcpyAsync(p, from, n * 32, hipMemcpyDeviceToDevice, sl.stream) : hipMemcpyPeerAsync(p, dst_dev, from, src_dev, n * 32, sl.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(sl.stream);
        }
        if (e != hipSuccess) { gs.last_hip = (int32_t)e; (void)hipGetLastError(); scalar_release(dst_ctx, p, scalar_bytes(n)); return DGPU_E_HIP; }
    }
    *handle = register_handle(p, n, 3);
    return DGPU_OK;
}

// several host arrays -> one resident scalar vector (the prover's `assignment` = inputs[1..] ++ witnesses, prover.rs:319-321,
// without a host-side concatenation)
int32_t dgpu_scalars_upload_parts(const uint64_t *const *parts, const size_t *counts, size_t n_parts, int32_t mont, uint64_t *handle) {
    if (!handle || (n_parts && (!parts || !counts))) return DGPU_E_BADARG;
    size_t n = 0;
    for (size_t k = 0; k < n_parts; k++) { if (counts[k] && !parts[k]) return DGPU_E_BADARG; n += counts[k]; }
    if (!cur().ready) return DGPU_E_NODEVICE;
    void *p = nullptr;
    {
        SLOT_ACQUIRE(L, sl);
        HIPCHK(hipSetDevice(cur().device));
        if (!(p = scalar_alloc(scalar_bytes(n)))) return DGPU_E_OOM;
        size_t at = 0;
        hipError_t e = hipSuccess;
        for (size_t k = 0; k < n_parts && e == hipSuccess; k++) {
            if (counts[k]) e = hipMemcpyAsync((uint8_t *)p + at * 32, parts[k], counts[k] * 32, hipMemcpyHostToDevice, sl.stream);
            at += counts[k];
        }
        if (e == hipSuccess && mont && n) ntt::launch_fr_mont_to_canonical(sl.stream, (uint32_t *)p, n);
        if (e == hipSuccess) e = hipStreamSynchronize(sl.stream);
        if (e != hipSuccess) { gs.last_hip = (int32_t)e; (void)hipGetLastError(); scalar_release(cur_index(), p, scalar_bytes(n)); return DGPU_E_HIP; }
    }
    *handle = register_handle(p, n, 3);
    return DGPU_OK;
}

// scalars split the way the sharded bases handle `like` is split (scalar i sits next to base i); n <= the number of bases
int32_t dgpu_scalars_upload_sharded(const uint64_t *sc, size_t n, int32_t mont, uint64_t like, uint64_t *handle) {
    if (!handle || (n && !sc)) return DGPU_E_BADARG;
    HandleRef hb(like);
    if (!hb.ok || (hb.h.kind != 7 && hb.h.kind != 8) || n > hb.h.n) return DGPU_E_BADARG;
    const ShardSet &sb = *(const ShardSet *)hb.h.p;
    const size_t G = sb.sub.size();
    ShardSet *ss = new ShardSet();
    ss->n = n; ss->sub.assign(G, 0); ss->lo.resize(G + 1);
    for (size_t k = 0; k <= G; k++) ss->lo[k] = std::min(sb.lo[k], n);
    const int32_t prc = par_run(G, [&](size_t k) -> int32_t {
        Handle part; if (!lookup_handle(sb.sub[k], part)) return (int32_t)DGPU_E_BADARG;
        CtxScope here(part.ctx);
        return dgpu_scalars_upload(sc + ss->lo[k] * 4, ss->lo[k + 1] - ss->lo[k], mont, &ss->sub[k]);
    });
    if (prc) { for (uint64_t h : ss->sub) if (h) (void)dgpu_scalars_free(h); delete ss; return prc; }
    *handle = register_handle(ss, n, 9);
    return DGPU_OK;
}

}  // extern "C"
