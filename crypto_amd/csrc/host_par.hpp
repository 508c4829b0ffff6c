// crypto_amd/csrc/host_par.hpp — the library's host-thread helpers.  Nothing may cross the C ABI by unwinding (include/dock_gpu.h: "never
// unwind, abort or print"): par_run joins whatever it started before it returns, runs a part on the calling thread when a thread cannot be
// created, and maps an exception of a part to an error code; abi_guard does the same for a whole entry point.
#pragma once
#include <cstdint>
#include <new>
#include <thread>
#include <vector>
#include "../../include/dock_gpu.h"

namespace dock {

template <class F> inline int32_t abi_guard(F &&f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc &) { return DGPU_E_OOM; }
    catch (...) { return DGPU_E_HIP; }
}

// body(k) -> int32_t for k = 0 .. parts - 1, part 0 on the calling thread, the others on threads of their own; the first non-zero code wins.
template <class F> inline int32_t par_run(size_t parts, F body) noexcept {
    if (parts == 0) return DGPU_OK;
    if (parts == 1) return abi_guard([&] { return (int32_t)body((size_t)0); });
    struct Joiner {                                    // joins on every path out of this function
        std::vector<std::thread> th;
        ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); }
    } j;
    std::vector<int32_t> rcs;
    try { rcs.assign(parts, DGPU_OK); j.th.reserve(parts - 1); }
    catch (...) { return DGPU_E_OOM; }
    int32_t *rc = rcs.data();
    for (size_t k = 1; k < parts; k++) {
        try { j.th.emplace_back([rc, k, &body] { rc[k] = abi_guard([&] { return (int32_t)body(k); }); }); }
        catch (...) { rc[k] = abi_guard([&] { return (int32_t)body(k); }); }      // no thread to be had: this part runs here
    }
    rc[0] = abi_guard([&] { return (int32_t)body((size_t)0); });
    for (auto &t : j.th) t.join();
    for (size_t k = 0; k < parts; k++) if (rc[k]) return rc[k];
    return DGPU_OK;
}

}  // namespace dock
