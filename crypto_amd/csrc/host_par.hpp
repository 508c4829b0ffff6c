// crypto_amd/csrc/host_par.hpp — the library's host-thread helpers.  Nothing may cross the C ABI by unwinding (include/dock_gpu.h: "never
// unwind, abort or print"): par_run returns only when every part has run, runs a part on the calling thread when it cannot be handed to a
// worker, and maps an exception of a part to an error code; abi_guard does the same for a whole entry point.
//
// The parts run on a process-wide pool of worker threads created at first use (creating a thread costs 30 - 50 us on the GPU box's host, more
// when fifty are created at once: the aggregation verifier's five concurrent GT multi-exponentiations took 3.3 ms instead of 1.1, a GIPA round's
// fourteen Miller-loop tails 0.3 ms more than their work).  A caller that waits for its parts helps: it runs those of ITS OWN parts that are
// still queued instead of sleeping (never somebody else's: a 0.3-ms tail must not pick up a 5-ms shard), so nested par_run calls cannot
// deadlock — every waiter can finish its own group alone — and a process whose workers are gone (a forked child) still finishes, on the
// calling thread.  A task runs under the SUBMITTER's thread-local context selection (tl_ctx: dgpu_set_device; tl_no_min),
// whichever thread executes it, and the executing thread's own values are restored afterwards.  DGPU_HOST_POOL=0 in the environment (read once)
// goes back to one new thread per part: measured on the GPU box, the pool takes the aggregation verifier from 11.5 to 8.3 ms and a GIPA round's
// Miller-loop tails from 1.25 to 1.1 ms, and costs 0.1 - 0.2 ms where fourteen 0.75-ms parts (tail + final exponentiation) are woken next to
// each other instead of being spread over the machine by the scheduler's fork balancing.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <pthread.h>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include "../../include/dock_gpu.h"

namespace dock {

extern thread_local int tl_ctx;            // dock_core.hip
extern thread_local bool tl_no_min;

template <class F> inline int32_t abi_guard(F &&f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc &) { return DGPU_E_OOM; }
    catch (...) { return DGPU_E_HIP; }
}

class HostPool {
    std::mutex m_;
    std::condition_variable cv_;
    struct Entry { const void *group; std::function<void()> f; };
    std::deque<Entry> q_;
    std::vector<std::thread> workers_;
    bool stop_ = false, started_ = false;
    void loop() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;                        // stop_ and nothing left
                f = std::move(q_.front().f); q_.pop_front();
            }
            f();                                               // (tasks never throw: par_run wraps them)
        }
    }
    void start_locked() {
        started_ = true;
        // (at least 16: most parts issue a device call and wait for it — they need a thread, not a core.  On a host whose affinity mask shows two cores the batch
        //  verifier's second MSM waited 0.8 ms for the worker that was computing a GT power: profiles/r06_timeline_batch.txt)
        unsigned w = std::thread::hardware_concurrency();
        w = w < 16 ? 16 : (w > 64 ? 64 : w);
        try { workers_.reserve(w); for (unsigned i = 0; i < w; i++) workers_.emplace_back([this] { loop(); }); }
        catch (...) {}                                         // fewer workers (or none): callers run what is left themselves
    }
    // fork(): the child has the forking thread only.  prepare takes the pool's mutex (no worker or submitter holds it across the fork, so the child never
    // inherits a locked mutex it cannot unlock); the child then starts from an empty, not-yet-started pool — the parent's workers do not exist there,
    // their std::thread objects and whatever was queued are abandoned (overwritten without running destructors: joining or detaching a thread that
    // does not exist is undefined) and new workers are created at the child's first par_run.  This makes fork + exec safe, NOT the use of the library in the
    // child: HIP state does not survive fork (streams, events and device memory of the parent are invalid there) and tasks that were queued by other threads of the
    // parent vanish — a child that wants the library calls dgpu_shutdown + dgpu_init* first (include/dock_gpu.h, lifecycle).
    static void atfork_prepare() { get().m_.lock(); }
    static void atfork_parent() { get().m_.unlock(); }
    static void atfork_child() {
        HostPool &p = get();
        new (&p.m_) std::mutex();
        new (&p.cv_) std::condition_variable();
        new (&p.q_) std::deque<Entry>();
        new (&p.workers_) std::vector<std::thread>();
        p.stop_ = false; p.started_ = false;
    }
    HostPool() { (void)pthread_atfork(&HostPool::atfork_prepare, &HostPool::atfork_parent, &HostPool::atfork_child); }
public:
    // Never destroyed: a static destructor would join workers that may still be inside HIP calls after the runtime's own teardown (and run during
    // exit() of a forked child whose workers do not exist).  The workers end with the process.
    static HostPool &get() { static HostPool *p = new HostPool; return *p; }
    // false: not queued (out of memory, shutting down) — the caller runs the task itself
    bool submit(const void *group, std::function<void()> f) noexcept {
        try {
            std::lock_guard<std::mutex> lk(m_);
            if (stop_) return false;
            if (!started_) start_locked();
            q_.push_back(Entry{group, std::move(f)});
        } catch (...) { return false; }
        cv_.notify_one();
        return true;
    }
    // run one queued task of `group` on the calling thread
    bool run_one_of(const void *group) noexcept {
        std::function<void()> f;
        {
            std::lock_guard<std::mutex> lk(m_);
            auto it = q_.begin();
            while (it != q_.end() && it->group != group) ++it;
            if (it == q_.end()) return false;
            f = std::move(it->f); q_.erase(it);
        }
        f();
        return true;
    }
};

// body(k) -> int32_t for k = 0 .. parts - 1, part 0 on the calling thread, the others on the pool; the first non-zero code wins.
template <class F> inline int32_t par_run(size_t parts, F body) noexcept {
    if (parts == 0) return DGPU_OK;
    if (parts == 1) return abi_guard([&] { return (int32_t)body((size_t)0); });
    struct Group { std::atomic<size_t> left{0}; std::mutex m; std::condition_variable cv; std::vector<int32_t> rc; };
    std::shared_ptr<Group> g;
    try { g = std::make_shared<Group>(); g->rc.assign(parts, DGPU_OK); }
    catch (...) { return DGPU_E_OOM; }
    g->left.store(parts - 1);
    const int ctx = tl_ctx; const bool no_min = tl_no_min;
    auto run_part = [g, ctx, no_min, &body](size_t k) {
        const int keep_ctx = tl_ctx; const bool keep_min = tl_no_min;
        tl_ctx = ctx; tl_no_min = no_min;
        g->rc[k] = abi_guard([&] { return (int32_t)body(k); });
        tl_ctx = keep_ctx; tl_no_min = keep_min;
        if (g->left.fetch_sub(1) == 1) { std::lock_guard<std::mutex> lk(g->m); g->cv.notify_all(); }
    };
    static const bool use_pool = []{ const char *e = getenv("DGPU_HOST_POOL"); return !(e && e[0] == '0'); }();
    if (!use_pool) {
        struct Joiner { std::vector<std::thread> th; ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); } } j;
        for (size_t k = 1; k < parts; k++) {
            try { j.th.emplace_back([run_part, k] { run_part(k); }); }
            catch (...) { run_part(k); }                   // no thread to be had: this part runs here
        }
        g->rc[0] = abi_guard([&] { return (int32_t)body((size_t)0); });
        for (auto &t : j.th) t.join();
        for (size_t k = 0; k < parts; k++) if (g->rc[k]) return g->rc[k];
        return DGPU_OK;
    }
    HostPool &pool = HostPool::get();
    for (size_t k = 1; k < parts; k++) {
        bool queued = false;
        try { queued = pool.submit(g.get(), [run_part, k] { run_part(k); }); } catch (...) { queued = false; }
        if (!queued) run_part(k);
    }
    g->rc[0] = abi_guard([&] { return (int32_t)body((size_t)0); });
    while (pool.run_one_of(g.get())) {}                          // help: our own parts that nobody has picked up yet (no new ones can appear)
    {
        std::unique_lock<std::mutex> lk(g->m);
        g->cv.wait(lk, [&] { return g->left.load() == 0; });
    }
    for (size_t k = 0; k < parts; k++) if (g->rc[k]) return g->rc[k];
    return DGPU_OK;
}

}  // namespace dock
