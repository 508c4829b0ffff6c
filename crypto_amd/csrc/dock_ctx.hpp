// crypto_amd/csrc/dock_ctx.hpp — process-wide state of libdock_gpu.so shared by its translation units
// (dock_core.hip: lifecycle/handles/instrumentation; dock_g1.hip / dock_g2.hip: the per-curve pipelines).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/dock_gpu.h"
#include "host_par.hpp"

namespace dock {

// Every hipMalloc the library issues goes through dev_malloc: the count and the time spent are reported by dgpu_device_alloc_count and as
// the "hipMalloc" row of dgpu_prof_read.  A steady-state MSM / proof must not allocate (hipMalloc costs 0.1 - 1 ms, and the hipFree that
// growing a buffer implies waits for the whole device, i.e. for every other call in flight): tests/test_gpu_reserve.py asserts a delta of 0.
extern std::atomic<uint64_t> g_dev_allocs, g_dev_alloc_ns, g_dev_alloc_bytes;
hipError_t dev_malloc(void **p, size_t bytes);

struct Buf {
    void *p = nullptr; size_t cap = 0;
    int32_t ensure(size_t bytes) {
        if (bytes <= cap) return DGPU_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        if (dev_malloc(&p, want) != hipSuccess) { (void)hipGetLastError(); if (dev_malloc(&p, bytes) != hipSuccess) { p = nullptr; return DGPU_E_OOM; } want = bytes; }
        cap = want; return DGPU_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

// kind: 1 = G1 bases, 2 = G2 bases, 3 = scalars, 4 = R1CS (DevR1cs*), 5 / 6 = G1 / G2 window table, 7 / 8 = G1 / G2 bases sharded over
// several devices (ShardSet*), 9 = scalars sharded like a bases set (ShardSet*), 10 / 11 = G1 / G2 precomputed-multiples table (PreTable*),
// 12 = sorted scalars (SortedScalars*), 13 / 14 = G1 / G2 fold table (FoldTab*, dock_fixed.hip).  `ctx` = the device context that owns the allocation;
// `inflight` = calls currently using it (a free waits for them: no use-after-free when dgpu_*_free races an MSM on the same handle).
struct Handle { void *p; size_t n; int kind; int ctx; int inflight; void *aux = nullptr; int small_uses = 0; };   // aux: the small-MSM table of a plain bases handle (msm_driver.hip.h small_sub_for)
struct PreTable { void *tab; size_t n; int c, W; };      // kind 10 / 11: tab[w * n + i] = prepared record of 2^(c w) P_i (pre_kernels.hip.h)
struct SortedScalars { void *off, *entries; size_t off_bytes, entries_bytes; size_t n, rows, boff; int c, W; };   // kind 12: the partition sort of n scalars for tables of `rows` rows, width c, first row boff
struct FoldTab { void *tab; size_t bytes; size_t n; };    // kind 13 / 14: the doubling chains of n points (fold_kernels.hip.h), identity flags behind them
struct ShardSet { std::vector<uint64_t> sub; std::vector<size_t> lo; size_t n = 0; };   // sub[k] covers [lo[k], lo[k+1]) (lo has sub.size() + 1 entries)
struct NttDomain { void *tw_f = nullptr, *tw_i = nullptr, *pw_f = nullptr, *pw_i = nullptr, *zinv = nullptr, *pwr_f = nullptr, *pwr_i = nullptr; };   // pwr_*: pw_* in bit-reversed order   // per log2(D), built once per device

struct ProfEntry { const char *name; double ms; uint64_t calls; };

struct Shared;
extern Shared gs;
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { dock::gs.last_hip = (int32_t)e_; (void)hipGetLastError(); return DGPU_E_HIP; } } while (0)

// One in-flight call = one Slot: its own HIP stream and grow-only workspace.  Several host threads (rayon workers in the
// reference, verifiable_encryption/src/tz_21/rdkgith.rs:140-147) can therefore have calls in flight at once; the
// latency-bound tail of one MSM (bucket reduction, host fold) overlaps the throughput-bound bulk of the next.
struct Slot {
    std::mutex mu;
    hipStream_t stream = nullptr;
    // H2D pieces travel on their own stream, ordered against the compute stream by these events: the kernels of a call run under its copies
    static constexpr int N_COPY_EV = 15;
    hipStream_t cstream = nullptr;
    hipStream_t xstream = nullptr;  // a third stream: the products of a Miller loop's middle piece (dock_pairing.hip: ml_pipelined)
    void *hpin = nullptr;           // pinned host scratch (HPIN_BYTES): results a host thread consumes while the slot's streams keep running
    static constexpr size_t HPIN_BYTES = 64 * 1024;
    void *hpin2 = nullptr; size_t hpin2_bytes = 0;      // a second, grow-only pinned buffer: the per-step products of a segmented Miller loop (dock_pairing.hip ml_segments)
    hipEvent_t copy_ev[N_COPY_EV + 1] = {};
    unsigned ev_next = 0;           // next event to record (taken in turn)
    Buf flags, in_bases, in_inf, in_scalars, prepped, digits, heavy, cnt, off, cursor, bsums, entries, bucket, bucket_inf, head, tail, head_b, tail_b, part_inf, l1, l1_inf, win, win_inf, ml_lines, ml_partial, ml_out, ml_coeffs, ml_state, dyn, hpart, hpart_inf, small_cnt;
    Buf q[16];      // witness-map workspace (dock_qap.hip)
    std::vector<std::pair<const char *, std::pair<hipEvent_t, hipEvent_t>>> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    void release_all() {
        Buf *bufs[] = {&flags, &in_bases, &in_inf, &in_scalars, &prepped, &digits, &heavy, &cnt, &off, &cursor, &bsums, &entries, &bucket, &bucket_inf, &head, &tail, &head_b, &tail_b, &part_inf, &l1, &l1_inf, &win, &win_inf, &ml_lines, &ml_partial, &ml_out, &ml_coeffs, &ml_state, &dyn, &hpart, &hpart_inf, &small_cnt};
        for (Buf *b : bufs) b->release();
        for (Buf &b : q) b.release();
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        ev_pool.clear(); prof_pending.clear();
    }
};
constexpr int N_SLOTS = 6;      // calls in flight per device context (a LegoGroth16 proof issues the witness map + five MSMs + the commitment's MSM together)

constexpr int MAX_CTX = 16;
// One Ctx per device the library was told to use (dgpu_init / dgpu_init_devices / dgpu_init_device_list): its streams, workspaces and NTT
// tables.  A host thread works on the context dgpu_set_device chose for it (thread-local, like hipSetDevice; default = the first one
// initialised); calls on a handle run on the context that owns the handle.
struct Ctx {
    std::atomic<bool> ready{false};
    int device = -1;               // physical HIP device
    Slot slots[N_SLOTS];
    std::atomic<unsigned> rr{0};
    // the slots are handed out first come, first served (SlotLock): tickets, the one being served, which slots are out
    std::mutex slot_mu; std::condition_variable slot_cv; uint64_t slot_next = 0, slot_serving = 0; bool slot_taken[N_SLOTS] = {};
    std::atomic<int> busy{0};       // calls holding one of the context's slots right now (SlotLock): the table MSM picks its bucket reduction's shape by it (msm_driver.hip.h pre_geometry)
    std::atomic<int> ml_active{0};  // dgpu_multi_miller_loop calls in flight on this context (dock_pairing.hip picks the kernel form by it)
    std::map<int, NttDomain> ntt_domains;
    // Recycled scalar vectors (guarded by gs.mu).  A proof uploads one assignment and receives one h vector, both released when it is done:
    // hipMalloc / hipFree per proof cost ~0.3 ms and hipFree waits for the whole device, i.e. for every other call in flight.
    std::vector<std::pair<void *, size_t>> scalar_pool;
    size_t scalar_pool_bytes = 0;
    std::vector<std::pair<void *, size_t>> fold_pool;   // tables of dgpu_g*_fold_prepare kept for the next one (guarded by gs.mu; dock_fixed.hip)
};
constexpr size_t SCALAR_POOL_MAX_ENTRIES = 32, SCALAR_POOL_MAX_BYTES = (size_t)4 << 30;
// process-wide state shared by all contexts
struct Shared {
    std::mutex mu;                 // lifecycle, handle table, profile table, NTT-domain tables
    std::mutex small_mu;           // one small-MSM table build at a time (msm_driver.hip.h small_sub_for)
    std::condition_variable cv;    // signalled when a handle's in-flight count drops
    std::map<uint64_t, Handle> handles;
    uint64_t next_handle = 1;
    std::atomic<int32_t> last_hip{0};
    std::atomic<size_t> min_gpu_n{DGPU_DEFAULT_MIN_GPU_N};
    std::atomic<size_t> auto_shard_min_n{~(size_t)0};      // dgpu_set_auto_shard_min_n: one-shot MSMs of at least this many terms are sharded over the process's device contexts (off by default)
    std::atomic<int> window_bits{0};
    std::atomic<size_t> small_max{8192};      // dgpu_set_small_msm_max: MSMs of up to this many terms on plain bases take the two-launch tree path (small_kernels.hip.h); 0 = never
    std::atomic<int> chunk{0};
    std::atomic<int> reduce_lanes{0};         // dgpu_set_reduce_lanes: 0 = the shared bucket set by bit marginals (reduce_kernels.hip.h; the plain pipeline then takes 4); 1 / 4: the scan form with k_reduce_top / k_reduce_top_quad (members per point in the last kernel)
    std::atomic<int> reduce_shift{-1};        // dgpu_set_reduce_shift: log2 buckets per lane of k_reduce_l0 on the table pipeline (-1 = automatic)
    std::atomic<int> ml_mode{31};             // dgpu_set_miller_pipeline: bit 0 the two-launch line kernel of small Miller loops, bit 1 the 18-role product tree, bit 2 sixteen lanes per pair in the line kernel, bit 3 (with bit 2) a wave per role, bit 4 three waves per sparse product while the chip is nearly empty (dock_pairing.hip)
    uint64_t allocs_at_reset = 0, alloc_ns_at_reset = 0;
    int default_ctx = -1;
    std::atomic<bool> prof{false};
    std::vector<ProfEntry> prof_tab;
};
extern Ctx ctxs[MAX_CTX];
extern thread_local int tl_ctx;
extern thread_local bool tl_no_min;      // set on the library's own threads (dock_prover.cpp): the caller-facing size threshold does not apply to calls the library makes itself
inline int cur_index() { int i = tl_ctx >= 0 ? tl_ctx : gs.default_ctx; return (i < 0 || i >= MAX_CTX) ? 0 : i; }
inline Ctx &cur() { return ctxs[cur_index()]; }
struct CtxScope { int prev; explicit CtxScope(int c) : prev(tl_ctx) { tl_ctx = c; } ~CtxScope() { tl_ctx = prev; } CtxScope(const CtxScope &) = delete; };

// RAII: a slot of the calling thread's context, first come, first served.  Callers beyond the N_SLOTS in flight queue with a ticket and are served in order:
// with `try_lock` round the slots and a blocking lock on ONE of them (rounds 1 - 6) a thread that had just released a slot took it again before the sleeper
// it had woken could run — eight native host threads kept the device exactly as busy as six, but two of them waited until the others had nothing left to do
// (tools/dev/native/inflight_threads.cpp: the longest of 60 calls = the whole 135-ms run), i.e. one MSM of a proof could wait for all MSMs of everybody else.
// `ok` is false when the context was shut down between the caller's `ready` check and the moment the slot was acquired (dgpu_shutdown clears `ready` first,
// then takes every slot): the slot's streams and buffers are gone then, and the call must answer DGPU_E_NODEVICE instead of running on a null stream
// (SLOT_ACQUIRE).  A slot's own mutex still guards it against the maintenance paths that take slots directly (reserve_slots, dgpu_shutdown).
struct SlotLock {
    Slot *s = nullptr; bool ok = false; Ctx *cxp = nullptr; int idx = -1;
    SlotLock() {
        Ctx &cx = cur();
        const unsigned start = cx.rr.fetch_add(1);
        auto free_slot = [&]() -> int { for (int k = 0; k < N_SLOTS; k++) { const int i = (int)((start + k) % N_SLOTS); if (!cx.slot_taken[i]) return i; } return -1; };
        {
            std::unique_lock<std::mutex> lk(cx.slot_mu);
            const uint64_t my = cx.slot_next++;
            cx.slot_cv.wait(lk, [&] { return cx.slot_serving == my && free_slot() >= 0; });
            idx = free_slot(); cx.slot_taken[idx] = true; cx.slot_serving++;
        }
        cx.slot_cv.notify_all();                         // (the next ticket may find another slot free)
        Slot *got = &cx.slots[idx];
        got->mu.lock();
        cxp = &cx;
        if (!cx.ready.load() || !got->stream) { got->mu.unlock(); give_back(); return; }
        s = got; ok = true; cx.busy.fetch_add(1);
    }
    ~SlotLock() { if (s) { cxp->busy.fetch_sub(1); s->mu.unlock(); give_back(); } }
    SlotLock(const SlotLock &) = delete;
private:
    void give_back() { { std::lock_guard<std::mutex> lk(cxp->slot_mu); cxp->slot_taken[idx] = false; } cxp->slot_cv.notify_all(); }
};
#define SLOT_ACQUIRE(lockname, slotname) dock::SlotLock lockname; if (!lockname.ok) return DGPU_E_NODEVICE; dock::Slot &slotname = *lockname.s

// device memory for a resident scalar vector of `bytes` on the current context: a recycled buffer of exactly that size, else hipMalloc
inline void *scalar_alloc(size_t bytes) {
    {
        std::lock_guard<std::mutex> lk(gs.mu);
        auto &pool = cur().scalar_pool;
        for (size_t k = 0; k < pool.size(); k++)
            if (pool[k].second == bytes) { void *p = pool[k].first; pool[k] = pool.back(); pool.pop_back(); cur().scalar_pool_bytes -= bytes; return p; }
    }
    void *p = nullptr;
    if (dev_malloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
// give it back (context `ctx` owns it; every call that used it has returned, so no stream still touches it)
inline void scalar_release(int ctx, void *p, size_t bytes) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(gs.mu);
        Ctx &c = ctxs[ctx];
        if (c.ready && c.scalar_pool.size() < SCALAR_POOL_MAX_ENTRIES && c.scalar_pool_bytes + bytes <= SCALAR_POOL_MAX_BYTES) {
            c.scalar_pool.emplace_back(p, bytes); c.scalar_pool_bytes += bytes; return;
        }
    }
    (void)hipFree(p);
}
inline size_t scalar_bytes(size_t n) { return (n ? n : 1) * 32; }

inline hipEvent_t ev_get(Slot &sl) {
    if (!sl.ev_pool.empty()) { hipEvent_t e = sl.ev_pool.back(); sl.ev_pool.pop_back(); return e; }
    hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; return e;
}
struct StageTimer {
    Slot &sl; const char *name; hipEvent_t a = nullptr, b = nullptr; bool on;
    StageTimer(Slot &s, const char *n) : sl(s), name(n), on(gs.prof.load()) { if (on) { a = ev_get(sl); b = ev_get(sl); if (a) (void)hipEventRecord(a, sl.stream); } }
    ~StageTimer() { if (on && a && b) { (void)hipEventRecord(b, sl.stream); sl.prof_pending.push_back({name, {a, b}}); } }
};
inline void prof_add_host(const char *name, double ms) {
    std::lock_guard<std::mutex> lk(gs.mu);
    for (auto &t : gs.prof_tab) if (t.name == name) { t.ms += ms; t.calls++; return; }
    gs.prof_tab.push_back({name, ms, 1});
}
inline void prof_flush(Slot &sl) {
    for (auto &pe : sl.prof_pending) {
        float ms = 0; (void)hipEventSynchronize(pe.second.second);
        if (hipEventElapsedTime(&ms, pe.second.first, pe.second.second) == hipSuccess) prof_add_host(pe.first, (double)ms);
        sl.ev_pool.push_back(pe.second.first); sl.ev_pool.push_back(pe.second.second);
    }
    sl.prof_pending.clear();
}
// peek (no pin): for argument checks only
inline bool lookup_handle(uint64_t h, Handle &out) {
    std::lock_guard<std::mutex> lk(gs.mu);
    auto it = gs.handles.find(h);
    if (it == gs.handles.end()) return false;
    out = it->second; return true;
}
// pin a handle for the duration of a call: dgpu_*_free of the same handle blocks until the call has returned
struct HandleRef {
    uint64_t id = 0; Handle h{}; bool ok = false;
    HandleRef() {}
    explicit HandleRef(uint64_t id_) { acquire(id_); }
    bool acquire(uint64_t id_) {
        std::lock_guard<std::mutex> lk(gs.mu);
        auto it = gs.handles.find(id_);
        if (it == gs.handles.end()) return false;
        it->second.inflight++; h = it->second; id = id_; ok = true; return true;
    }
    ~HandleRef() {
        if (!ok) return;
        { std::lock_guard<std::mutex> lk(gs.mu); auto it = gs.handles.find(id); if (it != gs.handles.end()) it->second.inflight--; }
        gs.cv.notify_all();
    }
    HandleRef(const HandleRef &) = delete;
    HandleRef &operator=(const HandleRef &) = delete;
};
inline uint64_t register_handle(void *p, size_t n, int kind) {
    std::lock_guard<std::mutex> lk(gs.mu);
    uint64_t h = gs.next_handle++;
    gs.handles[h] = Handle{p, n, kind, cur_index(), 0};
    return h;
}
// remove handle `id` from the table once nothing uses it; kind_ok decides whether the caller may free this kind
template <class Pred> inline bool take_handle(uint64_t id, Pred kind_ok, Handle &out) {
    std::unique_lock<std::mutex> lk(gs.mu);
    for (;;) {
        auto it = gs.handles.find(id);
        if (it == gs.handles.end() || !kind_ok(it->second.kind)) return false;
        if (it->second.inflight == 0) { out = it->second; gs.handles.erase(it); return true; }
        gs.cv.wait(lk);
    }
}
void free_r1cs_object(void *p);      // dock_qap.hip
// contexts a sharded call runs on: the first `ngpus` initialised ones (0 = all)
inline std::vector<int> ready_contexts(int ngpus) {
    std::vector<int> v;
    for (int i = 0; i < MAX_CTX; i++) if (ctxs[i].ready) { v.push_back(i); if (ngpus > 0 && (int)v.size() == ngpus) break; }
    return v;
}

int choose_c(size_t n, bool g2 = false);
int choose_chunk(size_t E, int min_chunk = 16, size_t max_chunks = 300000, int lanes_per_chunk = 1, size_t nb_shared = 0);
int forced_chunk();
int32_t upload_scalars(Slot &sl, const uint64_t *h, size_t n, bool mont, uint32_t *d_out);

}  // namespace dock
