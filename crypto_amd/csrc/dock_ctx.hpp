// crypto_amd/csrc/dock_ctx.hpp — process-wide state of libdock_gpu.so shared by its translation units
// (dock_core.hip: lifecycle/handles/instrumentation; dock_g1.hip / dock_g2.hip: the per-curve pipelines).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/dock_gpu.h"

namespace dock {

struct Buf {
    void *p = nullptr; size_t cap = 0;
    int32_t ensure(size_t bytes) {
        if (bytes <= cap) return DGPU_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; return DGPU_E_OOM; } want = bytes; }
        cap = want; return DGPU_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

struct Handle { void *p; size_t n; int kind; };
struct NttDomain { void *tw_f = nullptr, *tw_i = nullptr, *pw_f = nullptr, *pw_i = nullptr, *zinv = nullptr; };   // per log2(D), built once   // kind: 1 = G1 bases, 2 = G2 bases, 3 = scalars

struct ProfEntry { const char *name; double ms; uint64_t calls; };

struct Ctx;
extern Ctx g;
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { dock::g.last_hip = (int32_t)e_; (void)hipGetLastError(); return DGPU_E_HIP; } } while (0)

// One in-flight call = one Slot: its own HIP stream and grow-only workspace.  Several host threads (rayon workers in the
// reference, verifiable_encryption/src/tz_21/rdkgith.rs:140-147) can therefore have calls in flight at once; the
// latency-bound tail of one MSM (bucket reduction, host fold) overlaps the throughput-bound bulk of the next.
struct Slot {
    std::mutex mu;
    hipStream_t stream = nullptr;
    Buf in_bases, in_inf, in_scalars, prepped, digits, heavy, cnt, off, cursor, bsums, entries, bucket, bucket_inf, head, tail, head_b, tail_b, part_inf, l1, l1_inf, win, win_inf, ml_lines, ml_partial, ml_out, ml_coeffs;
    Buf q[16];      // witness-map workspace (dock_qap.hip)
    std::vector<std::pair<const char *, std::pair<hipEvent_t, hipEvent_t>>> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    void release_all() {
        Buf *bufs[] = {&in_bases, &in_inf, &in_scalars, &prepped, &digits, &heavy, &cnt, &off, &cursor, &bsums, &entries, &bucket, &bucket_inf, &head, &tail, &head_b, &tail_b, &part_inf, &l1, &l1_inf, &win, &win_inf, &ml_lines, &ml_partial, &ml_out, &ml_coeffs};
        for (Buf *b : bufs) b->release();
        for (Buf &b : q) b.release();
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        ev_pool.clear(); prof_pending.clear();
    }
};
constexpr int N_SLOTS = 4;

struct Ctx {
    std::mutex mu;                 // lifecycle, handle table, profile table
    std::atomic<bool> ready{false};
    int device = -1;
    std::atomic<int32_t> last_hip{0};
    size_t min_gpu_n = 0;
    int window_bits = 0;
    int chunk = 0;
    Slot slots[N_SLOTS];
    std::atomic<unsigned> rr{0};
    std::map<uint64_t, Handle> handles;
    uint64_t next_handle = 1;
    std::map<int, NttDomain> ntt_domains;
    std::atomic<bool> prof{false};
    std::vector<ProfEntry> prof_tab;
};

// RAII: pick a free slot (round-robin try_lock), or wait for one
struct SlotLock {
    Slot *s;
    SlotLock() {
        unsigned start = g.rr.fetch_add(1);
        for (int k = 0; k < N_SLOTS; k++) { Slot &c = g.slots[(start + k) % N_SLOTS]; if (c.mu.try_lock()) { s = &c; return; } }
        s = &g.slots[start % N_SLOTS]; s->mu.lock();
    }
    ~SlotLock() { s->mu.unlock(); }
    SlotLock(const SlotLock &) = delete;
};

inline hipEvent_t ev_get(Slot &sl) {
    if (!sl.ev_pool.empty()) { hipEvent_t e = sl.ev_pool.back(); sl.ev_pool.pop_back(); return e; }
    hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; return e;
}
struct StageTimer {
    Slot &sl; const char *name; hipEvent_t a = nullptr, b = nullptr; bool on;
    StageTimer(Slot &s, const char *n) : sl(s), name(n), on(g.prof.load()) { if (on) { a = ev_get(sl); b = ev_get(sl); if (a) (void)hipEventRecord(a, sl.stream); } }
    ~StageTimer() { if (on && a && b) { (void)hipEventRecord(b, sl.stream); sl.prof_pending.push_back({name, {a, b}}); } }
};
inline void prof_add_host(const char *name, double ms) {
    std::lock_guard<std::mutex> lk(g.mu);
    for (auto &t : g.prof_tab) if (t.name == name) { t.ms += ms; t.calls++; return; }
    g.prof_tab.push_back({name, ms, 1});
}
inline void prof_flush(Slot &sl) {
    for (auto &pe : sl.prof_pending) {
        float ms = 0; (void)hipEventSynchronize(pe.second.second);
        if (hipEventElapsedTime(&ms, pe.second.first, pe.second.second) == hipSuccess) prof_add_host(pe.first, (double)ms);
        sl.ev_pool.push_back(pe.second.first); sl.ev_pool.push_back(pe.second.second);
    }
    sl.prof_pending.clear();
}
inline bool lookup_handle(uint64_t h, Handle &out) {
    std::lock_guard<std::mutex> lk(g.mu);
    auto it = g.handles.find(h);
    if (it == g.handles.end()) return false;
    out = it->second; return true;
}

int choose_c(size_t n, bool g2 = false);
int choose_chunk(size_t E, int min_chunk = 16, size_t max_chunks = 300000);
int32_t upload_scalars(Slot &sl, const uint64_t *h, size_t n, bool mont, uint32_t *d_out);

}  // namespace dock
