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

struct Handle { void *p; size_t n; int kind; };   // kind: 1 = G1 bases, 2 = G2 bases, 3 = scalars

struct ProfEntry { const char *name; double ms; uint64_t calls; };

struct Ctx;
extern Ctx g;
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { dock::g.last_hip = (int32_t)e_; (void)hipGetLastError(); return DGPU_E_HIP; } } while (0)

struct Ctx {
    std::mutex mu;
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    std::atomic<int32_t> last_hip{0};
    size_t min_gpu_n = 0;
    int window_bits = 0;
    int chunk = 0;
    // workspace (grow-only)
    Buf in_bases, in_inf, in_scalars, prepped, digits, heavy, cnt, off, cursor, bsums, entries, bucket, bucket_inf, head, tail, head_b, tail_b, part_inf, l1, l1_inf, win, win_inf, ml_lines, ml_partial, ml_out;
    std::map<uint64_t, Handle> handles;
    uint64_t next_handle = 1;
    // profiling
    bool prof = false;
    std::vector<ProfEntry> prof_tab;
    std::vector<std::pair<const char *, std::pair<hipEvent_t, hipEvent_t>>> prof_pending;
    std::vector<hipEvent_t> ev_pool;
};


inline hipEvent_t ev_get() {
    if (!g.ev_pool.empty()) { hipEvent_t e = g.ev_pool.back(); g.ev_pool.pop_back(); return e; }
    hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; return e;
}
struct StageTimer {
    const char *name; hipEvent_t a = nullptr, b = nullptr;
    explicit StageTimer(const char *n) : name(n) { if (g.prof) { a = ev_get(); b = ev_get(); if (a) (void)hipEventRecord(a, g.stream); } }
    ~StageTimer() { if (g.prof && a && b) { (void)hipEventRecord(b, g.stream); g.prof_pending.push_back({name, {a, b}}); } }
};
inline void prof_flush() {
    for (auto &pe : g.prof_pending) {
        float ms = 0; (void)hipEventSynchronize(pe.second.second);
        if (hipEventElapsedTime(&ms, pe.second.first, pe.second.second) == hipSuccess) {
            bool found = false;
            for (auto &t : g.prof_tab) if (t.name == pe.first) { t.ms += ms; t.calls++; found = true; break; }
            if (!found) g.prof_tab.push_back({pe.first, (double)ms, 1});
        }
        g.ev_pool.push_back(pe.second.first); g.ev_pool.push_back(pe.second.second);
    }
    g.prof_pending.clear();
}


inline void prof_add_host(const char *name, double ms) {
    for (auto &t : g.prof_tab) if (t.name == name) { t.ms += ms; t.calls++; return; }
    g.prof_tab.push_back({name, ms, 1});
}
int choose_c(size_t n);
int choose_chunk();
int32_t upload_scalars(const uint64_t *h, size_t n, bool mont, uint32_t *d_out);

}  // namespace dock
