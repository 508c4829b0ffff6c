// tests/native/host_sanitize_driver.cpp — the library's HOST-side concurrency under ThreadSanitizer / AddressSanitizer+UBSan, no GPU needed.
//
// Built and run by tests/test_host_sanitizers.py (g++ -fsanitize=thread and -fsanitize=address,undefined) from the PRODUCT's sources:
//   * crypto_amd/csrc/dock_prover.cpp (dgpu_legogroth16_prove: seven host threads per proof, one more per shard in the sharded form),
//     crypto_amd/csrc/dock_gt.cpp (dgpu_fp12_multi_pow: a thread per chunk), crypto_amd/csrc/dock_serde.cpp (for_points: a thread per slice),
//     crypto_amd/csrc/host_par.hpp (par_run and its worker pool), crypto_amd/csrc/dock_aggregation.cpp (dgpu_snarkpack_aggregate / _verify: nested parallel
//     sections, the caller's transcript called back, fold tables) — compiled as they are;
//   * crypto_amd/csrc/dock_ctx.hpp's slot / handle machinery (SlotLock, HandleRef, register_handle, take_handle, scalar_alloc / scalar_release)
//     — included as it is, with the handful of HIP runtime calls it names stubbed out below (nothing here touches a device).
// The device entry points dock_prover.cpp calls are replaced by stand-ins that return the identity after a short, varying delay and FAIL on
// request (every k-th call), so that the error paths — a job failing while its siblings are still running, handles freed while other
// jobs still use them — run under the sanitizers too.
//
// Exit code 0 and no sanitizer report = pass.  Prints a one-line summary.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
#define __HIP_PLATFORM_AMD__ 1
#include "../../crypto_amd/csrc/dock_ctx.hpp"
#include "../../crypto_amd/csrc/host_field.hpp"

// ---- HIP runtime stand-ins (only what dock_ctx.hpp's inline functions name) ----------------------------------------------------------------
extern "C" {
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
}
namespace dock {
Shared gs;
Ctx ctxs[MAX_CTX];
thread_local int tl_ctx = -1;
thread_local bool tl_no_min = false;
std::atomic<uint64_t> g_dev_allocs{0}, g_dev_alloc_ns{0}, g_dev_alloc_bytes{0};
hipError_t dev_malloc(void **p, size_t bytes) { *p = malloc(bytes ? bytes : 1); g_dev_allocs++; return *p ? hipSuccess : hipErrorOutOfMemory; }
void free_r1cs_object(void *) {}
int32_t msm_g1_nothreshold(const uint64_t *, const uint8_t *, const uint64_t *, size_t, uint64_t out[18]) { memset(out, 0, 144); return DGPU_OK; }
// stand-ins for the host-key prover's views (dock_g1.hip / dock_g2.hip view_acquire_*: the resident-bases cache or an upload for the call): a handle of the
// view's length that reads the caller's memory like the real one, released through the pin
std::atomic<int> g_live_views{0};
static int32_t view_acquire_any(int kind, const void *p, size_t stride, size_t n, uint64_t *handle, void **pin) {
    if (n) { volatile uint8_t t = ((const uint8_t *)p)[(n - 1) * stride]; (void)t; }
    *handle = register_handle(malloc(16), n, kind); *pin = new uint64_t(*handle); g_live_views++;
    return DGPU_OK;
}
int32_t view_acquire_g1(const void *p, size_t stride, size_t, size_t, size_t, size_t n, int, uint64_t *handle, void **pin) { return view_acquire_any(1, p, stride, n, handle, pin); }
int32_t view_acquire_g2(const void *p, size_t stride, size_t, size_t, size_t, size_t n, int, uint64_t *handle, void **pin) { return view_acquire_any(2, p, stride, n, handle, pin); }
bool view_verify_any(void *) { return true; }
void view_release_any(void *pin) {
    if (!pin) return;
    uint64_t *h = (uint64_t *)pin; Handle hd;
    if (take_handle(*h, [](int) { return true; }, hd)) { free(hd.p); g_live_views--; }
    delete h;
}
}  // namespace dock
using namespace dock;

// ---- stand-ins for the device entry points of the prover ----------------------------------------------------------------------------------
static std::atomic<uint64_t> g_calls{0};
static std::atomic<uint64_t> g_fail_every{0};          // 0: never
static std::atomic<int> g_live_scalars{0}, g_live_sorted{0};
static int32_t step(const char *) {
    const uint64_t k = ++g_calls;
    std::this_thread::sleep_for(std::chrono::microseconds(20 + (k * 37) % 180));
    const uint64_t f = g_fail_every.load();
    return (f && k % f == 0) ? DGPU_E_OOM : DGPU_OK;
}
static void identity1(uint64_t *o) { hostf::Fq one = hostf::Fq::one(); memcpy(o, &one, 48); memcpy(o + 6, &one, 48); memset(o + 12, 0, 48); }
static void identity2(uint64_t *o) { memset(o, 0, 288); hostf::Fq one = hostf::Fq::one(); memcpy(o, &one, 48); memcpy(o + 12, &one, 48); }
extern "C" {
int32_t dgpu_scalars_upload(const uint64_t *sc, size_t n, int32_t, uint64_t *h) {
    int32_t rc = step("upload"); if (rc) return rc;
    volatile uint64_t touch = n ? sc[4 * (n - 1)] : 0; (void)touch;                 // reads the caller's buffer like the real one
    void *p = malloc(32 * (n ? n : 1)); *h = register_handle(p, n, 3); g_live_scalars++; return DGPU_OK;
}
int32_t dgpu_scalars_upload_parts(const uint64_t *const *parts, const size_t *counts, size_t n_parts, int32_t, uint64_t *h) {
    int32_t rc = step("upload"); if (rc) return rc;
    size_t n = 0;
    for (size_t k = 0; k < n_parts; k++) { if (counts[k]) { volatile uint64_t touch = parts[k][4 * (counts[k] - 1)]; (void)touch; } n += counts[k]; }
    void *p = malloc(32 * (n ? n : 1)); *h = register_handle(p, n, 3); g_live_scalars++; return DGPU_OK;
}
int32_t dgpu_scalars_free(uint64_t h) { Handle hd; if (!take_handle(h, [](int k) { return k == 3 || k == 12; }, hd)) return DGPU_E_BADARG; if (hd.kind == 3) g_live_scalars--; else g_live_sorted--; free(hd.p); return DGPU_OK; }
int32_t dgpu_handle_len(uint64_t h, size_t *n) { Handle hd; if (!lookup_handle(h, hd)) return DGPU_E_BADARG; *n = hd.n; return DGPU_OK; }
int32_t dgpu_handle_context(uint64_t h, int32_t *c) { Handle hd; if (!lookup_handle(h, hd)) return DGPU_E_BADARG; *c = hd.ctx; return DGPU_OK; }
int32_t dgpu_set_device(int32_t c) { tl_ctx = c; return DGPU_OK; }
int32_t dgpu_r1cs_shape(uint64_t h, size_t *v, size_t *i, size_t *c) { Handle hd; if (!lookup_handle(h, hd) || hd.kind != 4) return DGPU_E_BADARG; if (v) *v = hd.n; if (i) *i = 2; if (c) *c = hd.n - 1; return DGPU_OK; }
int32_t dgpu_witness_map_r1cs_resident(uint64_t r1cs, uint64_t z, uint64_t *, uint64_t *out_handle, size_t *out_len) {
    HandleRef a(r1cs), b(z); if (!a.ok || !b.ok) return DGPU_E_BADARG;
    int32_t rc = step("wm"); if (rc) return rc;
    size_t D = 1; while (D < a.h.n + 1) D <<= 1;
    void *p = calloc(D, 32); *out_handle = register_handle(p, D, 3); g_live_scalars++; *out_len = D; return DGPU_OK;
}
int32_t dgpu_witness_map_r1cs(uint64_t r1cs, const uint64_t *, size_t, int32_t, uint64_t *out_h, uint64_t *out_handle, size_t *out_len) {
    HandleRef a(r1cs); if (!a.ok) return DGPU_E_BADARG;
    int32_t rc = step("wm"); if (rc) return rc;
    size_t D = 1; while (D < a.h.n + 1) D <<= 1;
    if (out_h) memset(out_h, 0, D * 32);
    if (out_handle) { *out_handle = register_handle(calloc(D, 32), D, 3); g_live_scalars++; }
    *out_len = D; return DGPU_OK;
}
int32_t dgpu_scalars_copy_range(uint64_t src, size_t lo, size_t hi, int32_t, uint64_t *h) {
    HandleRef a(src); if (!a.ok || a.h.kind != 3 || lo > hi || hi > a.h.n) return DGPU_E_BADARG;
    int32_t rc = step("copy"); if (rc) return rc;
    void *p = malloc(32 * (hi - lo ? hi - lo : 1)); memcpy(p, (const char *)a.h.p + 32 * lo, 32 * (hi - lo));
    *h = register_handle(p, hi - lo, 3); g_live_scalars++; return DGPU_OK;
}
int32_t dgpu_bases_table_shape(uint64_t h, size_t *rows, int32_t *c, int32_t *w) { Handle hd; if (!lookup_handle(h, hd) || (hd.kind != 10 && hd.kind != 11)) return DGPU_E_BADARG; *rows = hd.n; *c = 17; *w = 15; return DGPU_OK; }
int32_t dgpu_scalars_sort(uint64_t t, size_t, uint64_t s, size_t, size_t n, uint64_t *out) {
    HandleRef a(t), b(s); if (!a.ok || !b.ok) return DGPU_E_BADARG;
    int32_t rc = step("sort"); if (rc) return rc;
    *out = register_handle(malloc(64), n, 12); g_live_sorted++; return DGPU_OK;
}
static int32_t msm_like(uint64_t a, uint64_t b, uint64_t *out, bool g2) { HandleRef x(a), y(b); if (!x.ok || !y.ok) return DGPU_E_BADARG; int32_t rc = step("msm"); if (rc) return rc; if (g2) identity2(out); else identity1(out); return DGPU_OK; }
int32_t dgpu_msm_g1_resident(uint64_t b, size_t, uint64_t s, size_t, size_t, uint64_t *o) { return msm_like(b, s, o, false); }
int32_t dgpu_msm_g2_resident(uint64_t b, size_t, uint64_t s, size_t, size_t, uint64_t *o) { return msm_like(b, s, o, true); }
int32_t dgpu_msm_g1_sorted(uint64_t t, uint64_t s, size_t, uint64_t *o) { return msm_like(t, s, o, false); }
int32_t dgpu_msm_g2_sorted(uint64_t t, uint64_t s, size_t, uint64_t *o) { return msm_like(t, s, o, true); }
int32_t dgpu_lincomb_g1(const uint64_t *, const uint8_t *, const uint64_t *, size_t, uint64_t *o) { identity1(o); return step("lin"); }
int32_t dgpu_lincomb_g2(const uint64_t *, const uint8_t *, const uint64_t *, size_t, uint64_t *o) { identity2(o); return step("lin"); }
int32_t dgpu_fold_g1(const uint64_t *, size_t, uint64_t *o) { identity1(o); return DGPU_OK; }
int32_t dgpu_fold_g2(const uint64_t *, size_t, uint64_t *o) { identity2(o); return DGPU_OK; }
int32_t dgpu_shard_count(uint64_t h, int32_t *c) { Handle hd; if (!lookup_handle(h, hd)) return DGPU_E_BADARG; *c = (hd.kind >= 7 && hd.kind <= 9) ? (int32_t)((const ShardSet *)hd.p)->sub.size() : 0; return DGPU_OK; }
int32_t dgpu_shard_part(uint64_t h, size_t k, uint64_t *sub, size_t *lo, size_t *hi, int32_t *ctx) {
    HandleRef r(h); if (!r.ok || r.h.kind < 7 || r.h.kind > 9) return DGPU_E_BADARG;
    const ShardSet &ss = *(const ShardSet *)r.h.p; if (k >= ss.sub.size()) return DGPU_E_BADARG;
    *sub = ss.sub[k]; *lo = ss.lo[k]; *hi = ss.lo[k + 1]; *ctx = (int32_t)k; return DGPU_OK;
}
// ---- stand-ins for the device entry points of the aggregation (dock_aggregation.cpp): zeros / ones after a delay, failures on request ----
static hostf::Fq12 gt_of(uint64_t seed) { hostf::Fq12 f = hostf::Fq12::one(); f.c0.c0.c0.l[0] ^= seed * 0x9e3779b97f4a7c15ULL; return f; }      // some value that depends on the call
int32_t dgpu_g1_mul_add_batch(const uint64_t *p, const uint8_t *, const uint64_t *, size_t, const uint64_t *, const uint8_t *, size_t n, uint64_t *out, uint8_t *oi) { memcpy(out, p, n * 96); memset(oi, 0, n); return step("muladd"); }
int32_t dgpu_g2_mul_add_batch(const uint64_t *p, const uint8_t *, const uint64_t *, size_t, const uint64_t *, const uint8_t *, size_t n, uint64_t *out, uint8_t *oi) { memcpy(out, p, n * 192); memset(oi, 0, n); return step("muladd"); }
int32_t dgpu_msm_g1(const uint64_t *b, const uint8_t *, const uint64_t *sc, size_t n, uint64_t *o) { volatile uint64_t t = b[12 * n - 1] ^ sc[4 * n - 1]; (void)t; identity1(o); return step("msm"); }
int32_t dgpu_msm_g2(const uint64_t *b, const uint8_t *, const uint64_t *sc, size_t n, uint64_t *o) { volatile uint64_t t = b[24 * n - 1] ^ sc[4 * n - 1]; (void)t; identity2(o); return step("msm"); }
int32_t dgpu_multi_pairing_segments(const uint64_t *p, const uint64_t *q, const uint8_t *, size_t n, const uint64_t *ends, size_t nseg, uint64_t *out) {
    volatile uint64_t t = p[12 * n - 1] ^ q[24 * n - 1] ^ ends[nseg - 1]; (void)t;
    for (size_t g = 0; g < nseg; g++) { hostf::Fq12 f = gt_of(g + n); memcpy(out + 72 * g, &f, 576); }
    return step("pairings");
}
int32_t dgpu_multi_miller_loop(const uint64_t *p, const uint64_t *q, const uint8_t *, size_t n, uint64_t *out) { volatile uint64_t t = p[12 * n - 1] ^ q[24 * n - 1]; (void)t; hostf::Fq12 f = gt_of(n); memcpy(out, &f, 576); return step("ml"); }
int32_t dgpu_final_exponentiation(const uint64_t *in, uint64_t *out) { memcpy(out, in, 576); return step("fe"); }
int32_t dgpu_multi_miller_loop_mixed(const uint64_t *pa, const uint64_t *qa, const uint8_t *sa, size_t na, const uint64_t *pp, const uint64_t *co, const uint8_t *sp, size_t np, uint64_t *out) {
    volatile uint64_t t = (na ? pa[12 * na - 1] ^ qa[24 * na - 1] ^ sa[na - 1] : 0) ^ (np ? pp[12 * np - 1] ^ co[(size_t)DGPU_G2_PREPARED_WORDS * np - 1] ^ sp[np - 1] : 0); (void)t;      // (reads the last word of every operand: sizes as the caller promised)
    memset(out, 0, 576); out[0] = 1; return step("mixed");
}
int32_t dgpu_multi_miller_loop_scaled(const uint64_t *pa, const uint64_t *sc, size_t stride, const uint64_t *qa, const uint8_t *sa, size_t na, const uint64_t *pp, const uint64_t *co, const uint8_t *sp, size_t np, uint64_t *out) {
    volatile uint64_t t = (na ? pa[12 * na - 1] ^ qa[24 * na - 1] ^ sc[stride ? 4 * na - 1 : 3] ^ (sa ? sa[na - 1] : 0) : 0) ^ (np ? pp[12 * np - 1] ^ co[(size_t)DGPU_G2_PREPARED_WORDS * np - 1] ^ (sp ? sp[np - 1] : 0) : 0); (void)t;
    memset(out, 0, 576); out[0] = 1; return step("scaled");
}
int32_t dgpu_g1_scale_batch(const uint64_t *p, const uint8_t *, const uint64_t *, size_t, const uint8_t *, size_t n, uint64_t *out, uint8_t *oi) { memcpy(out, p, n * 96); memset(oi, 0, n); return step("scale"); }
static std::atomic<int> g_live_fold{0};
int32_t dgpu_fold_prepare_pair(const uint64_t *p1, size_t n1, uint64_t *h1, const uint64_t *p2, size_t n2, uint64_t *h2) {
    int32_t rc = step("prepare"); if (rc) return rc;
    uint64_t *t1 = (uint64_t *)malloc(n1 * 96), *t2 = (uint64_t *)malloc(n2 * 192); memcpy(t1, p1, n1 * 96); memcpy(t2, p2, n2 * 192);
    *h1 = register_handle(t1, n1, 13); *h2 = register_handle(t2, n2, 14); g_live_fold += 2; return DGPU_OK;
}
static int32_t fold_apply_like(uint64_t h, int kind, size_t pt, uint64_t *out, uint8_t *oi) { HandleRef r(h); if (!r.ok || r.h.kind != kind) return DGPU_E_BADARG; memcpy(out, r.h.p, r.h.n * pt); memset(oi, 0, r.h.n); return step("apply"); }
int32_t dgpu_g1_fold_apply(uint64_t h, const uint64_t *, const uint64_t *, uint64_t *out, uint8_t *oi) { return fold_apply_like(h, 13, 96, out, oi); }
int32_t dgpu_g2_fold_apply(uint64_t h, const uint64_t *, const uint64_t *, uint64_t *out, uint8_t *oi) { return fold_apply_like(h, 14, 192, out, oi); }
int32_t dgpu_fold_free(uint64_t h) { Handle hd; if (!take_handle(h, [](int k) { return k == 13 || k == 14; }, hd)) return DGPU_E_BADARG; free(hd.p); g_live_fold--; return DGPU_OK; }
int32_t dgpu_g2_serialize(const uint64_t *xy, const uint8_t *is_inf, size_t n, int32_t compressed, uint8_t *out);
// the entry points under test that live in host-only units of the product
int32_t dgpu_fp12_multi_pow(const uint64_t *a, const uint64_t *e, size_t n, uint64_t out[72]);
int32_t dgpu_g1_serialize(const uint64_t *xy, const uint8_t *is_inf, size_t n, int32_t compressed, uint8_t *out);
int32_t dgpu_g1_deserialize(const uint8_t *in, size_t n, int32_t mode, uint64_t *xy, uint8_t *is_inf);
}

// ---- the tests ----------------------------------------------------------------------------------------------------------------------------
static std::atomic<int> failures{0};
#define EXPECT(c) do { if (!(c)) { fprintf(stderr, "EXPECT failed: %s (line %d)\n", #c, __LINE__); failures++; } } while (0)

// 1. slots and handles: lockers, pinners and a freer race; a free never completes while a pin is held
static void test_slots_and_handles() {
    Ctx &c = ctxs[0];
    for (int i = 0; i < N_SLOTS; i++) c.slots[i].stream = (hipStream_t)(uintptr_t)(0x1000 + i);
    c.device = 0; c.ready = true; gs.default_ctx = 0;
    std::atomic<int> in_slot[N_SLOTS]; for (auto &x : in_slot) x = 0;
    std::atomic<bool> stop{false}, expect_ok{true};
    std::atomic<uint64_t> shared_handle{register_handle(malloc(64), 1, 3)};
    std::atomic<int> pins{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 10; t++) th.emplace_back([&, t] {
        std::mt19937 rng(t);
        while (!stop) {
            { SlotLock L; if (expect_ok) EXPECT(L.ok); if (L.ok) { const int k = (int)(L.s - c.slots); EXPECT(in_slot[k].fetch_add(1) == 0); std::this_thread::sleep_for(std::chrono::microseconds(rng() % 50)); in_slot[k]--; } }
            { HandleRef r(shared_handle.load()); if (r.ok) { pins++; volatile char seen = ((volatile char *)r.h.p)[0]; (void)seen; std::this_thread::sleep_for(std::chrono::microseconds(rng() % 30)); pins--; } }     // (pins are shared: readers)
            void *p = scalar_alloc(scalar_bytes(64 + t)); EXPECT(p != nullptr); scalar_release(0, p, scalar_bytes(64 + t));
        }
    });
    for (int round = 0; round < 200; round++) {                      // the freer: takes the handle (waits for the pins), replaces it
        Handle hd; const uint64_t h = shared_handle.load();
        EXPECT(take_handle(h, [](int k) { return k == 3; }, hd));
        memset(hd.p, 0xEE, 64); free(hd.p);                          // nobody may still be writing into it (ASAN / TSAN see it if somebody is)
        shared_handle = register_handle(malloc(64), 1, 3);
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    // shutdown race: ready cleared while lockers keep coming -> they report !ok instead of using the slot
    expect_ok = false;
    c.ready = false;
    for (int i = 0; i < 50; i++) { SlotLock L; EXPECT(!L.ok); }
    c.ready = true;
    stop = true;
    for (auto &t : th) t.join();
    { Handle hd; EXPECT(take_handle(shared_handle.load(), [](int k) { return k == 3; }, hd)); free(hd.p); }
    std::lock_guard<std::mutex> lk(gs.mu);
    for (auto &e : c.scalar_pool) free(e.first);
    c.scalar_pool.clear(); c.scalar_pool_bytes = 0;
}

// 2. the prover: concurrent proofs, single-context and sharded keys, with and without failing stages
static uint64_t make(int kind, size_t n, void *p = nullptr) { return register_handle(p ? p : malloc(16), n, kind); }
static void drop(uint64_t h) { Handle hd; if (take_handle(h, [](int) { return true; }, hd)) { if (hd.kind >= 7 && hd.kind <= 9) delete (ShardSet *)hd.p; else free(hd.p); } }
static void test_prover() {
    const size_t V = 40, n_inst = 2, cw = 1;
    std::vector<uint64_t> z(4 * V, 0); for (size_t i = 0; i < V; i++) z[4 * i] = i + 1;
    hostf::Fq one = hostf::Fq::one();
    uint64_t g1pt[12], g2pt[24]; memcpy(g1pt, &one, 48); memcpy(g1pt + 6, &one, 48); memset(g2pt, 0, 192); memcpy(g2pt, &one, 48); memcpy(g2pt + 12, &one, 48);
    std::vector<uint64_t> gabc(12 * (n_inst + cw)); for (size_t k = 0; k < n_inst + cw; k++) memcpy(&gabc[12 * k], g1pt, 96);
    auto key = [&](bool tables, bool sharded) {
        dgpu_lego_pk pk; memset(&pk, 0, sizeof pk);
        auto query = [&](int kind, size_t n) -> uint64_t {
            if (!sharded) return make(tables ? kind + 9 : kind, n);
            ShardSet *ss = new ShardSet(); ss->n = n; ss->lo = {0, n / 2, n};
            ss->sub = {make(tables ? kind + 9 : kind, n / 2), make(tables ? kind + 9 : kind, n - n / 2)};
            return make(kind + 6, n, ss);
        };
        pk.a_query = query(1, V); pk.b_g1_query = query(1, V); pk.b_g2_query = query(2, V); pk.h_query = query(1, 63); pk.l_query = query(1, V - n_inst - cw);
        pk.alpha_g1 = pk.beta_g1 = pk.delta_g1 = pk.eta_delta_inv_g1 = pk.eta_gamma_inv_g1 = pk.a0 = pk.b1_0 = g1pt;
        pk.beta_g2 = pk.delta_g2 = pk.b2_0 = g2pt;
        pk.gamma_abc_g1 = gabc.data(); pk.gamma_abc_len = n_inst + cw; pk.commit_witness_count = cw;
        return pk;
    };
    auto drop_key = [&](const dgpu_lego_pk &pk) {
        for (uint64_t h : {pk.a_query, pk.b_g1_query, pk.b_g2_query, pk.h_query, pk.l_query}) {
            Handle hd; if (lookup_handle(h, hd) && hd.kind >= 7 && hd.kind <= 9) for (uint64_t s : ((ShardSet *)hd.p)->sub) drop(s);
            drop(h);
        }
    };
    const uint64_t r[4] = {5, 0, 0, 0}, s[4] = {7, 0, 0, 0}, v[4] = {9, 0, 0, 0}, r0[4] = {0, 0, 0, 0};
    for (int variant = 0; variant < 3; variant++) {                  // plain queries, tables (shared sort), sharded tables
        const dgpu_lego_pk pk = key(variant >= 1, variant == 2);
        const uint64_t circuit = make(4, V);
        for (uint64_t fail_every : {(uint64_t)0, (uint64_t)7, (uint64_t)3}) {
            g_fail_every = fail_every;
            std::atomic<int> ok{0}, bad{0};
            std::vector<std::thread> th;
            for (int t = 0; t < 6; t++) th.emplace_back([&, t] {
                for (int k = 0; k < 6; k++) {
                    uint64_t a[12], b[24], c[12], d[12]; uint8_t inf[4];
                    const int32_t rc = dgpu_legogroth16_prove(&pk, circuit, 0, z.data(), V, n_inst, 0, (t + k) % 3 ? r : r0, s, v, a, b, c, d, inf);
                    if (rc == DGPU_OK) ok++; else if (rc == DGPU_E_OOM) bad++; else { fprintf(stderr, "unexpected rc %d\n", rc); failures++; }
                }
            });
            for (auto &x : th) x.join();
            g_fail_every = 0;
            if (!fail_every) EXPECT(ok == 36 && bad == 0); else EXPECT(bad > 0);
            EXPECT(g_live_scalars.load() == 0 && g_live_sorted.load() == 0);          // every z / h / sort handle of every call was freed, failed or not
        }
        // round 6: the same schedule for a key held as host slices (dgpu_legogroth16_prove_host: five views acquired side by side, released on every path out),
        // h from the host and from the resident circuit, six callers at once, with and without injected failures
        if (variant == 0) {
            std::vector<uint64_t> q1(13 * V, 1), q2(25 * V, 1), hq(13 * 63, 1), hh(4 * 64, 3);
            dgpu_lego_pk_host hk; memset(&hk, 0, sizeof hk);
            auto view = [](const std::vector<uint64_t> &m, size_t words, size_t n) { dgpu_bases_view v_; v_.p = m.data(); v_.stride = words * 8; v_.x_off = 0; v_.y_off = (words - 1) * 4; v_.inf_off = (words - 1) * 8; v_.n = n; return v_; };
            hk.a_query = view(q1, 13, V); hk.b_g1_query = view(q1, 13, V); hk.b_g2_query = view(q2, 25, V); hk.h_query = view(hq, 13, 63); hk.l_query = view(q1, 13, V - n_inst - cw);
            hk.alpha_g1 = hk.beta_g1 = hk.delta_g1 = hk.eta_delta_inv_g1 = hk.eta_gamma_inv_g1 = hk.a0 = hk.b1_0 = g1pt;
            hk.beta_g2 = hk.delta_g2 = hk.b2_0 = g2pt;
            hk.gamma_abc_g1 = gabc.data(); hk.gamma_abc_len = n_inst + cw; hk.commit_witness_count = cw;
            for (uint64_t fail_every : {(uint64_t)0, (uint64_t)5}) {
                g_fail_every = fail_every;
                std::atomic<int> ok{0}, bad{0};
                std::vector<std::thread> th;
                for (int t = 0; t < 6; t++) th.emplace_back([&, t] {
                    for (int k = 0; k < 6; k++) {
                        uint64_t a[12], b[24], c[12], d[12]; uint8_t inf[4];
                        const bool from_host = (t + k) & 1;
                        const int32_t rc = dgpu_legogroth16_prove_host(&hk, from_host ? 0 : circuit, from_host ? hh.data() : nullptr, from_host ? 64 : 0, 0, z.data(), n_inst, z.data() + 4 * n_inst, V - n_inst, 0,
                                                                       r, s, v, a, b, c, d, inf);
                        if (rc == DGPU_OK) ok++; else if (rc == DGPU_E_OOM) bad++; else { fprintf(stderr, "prove_host: unexpected rc %d\n", rc); failures++; }
                    }
                });
                for (auto &x : th) x.join();
                g_fail_every = 0;
                if (!fail_every) EXPECT(ok == 36 && bad == 0); else EXPECT(bad > 0);
                EXPECT(g_live_scalars.load() == 0 && g_live_sorted.load() == 0 && dock::g_live_views.load() == 0);
            }
            { uint64_t a[12], b[24], c[12], d[12]; uint8_t inf[4];            // both sources of h / neither: refused before anything is acquired
              EXPECT(dgpu_legogroth16_prove_host(&hk, circuit, hh.data(), 64, 0, z.data(), n_inst, z.data() + 4 * n_inst, V - n_inst, 0, r, s, v, a, b, c, d, inf) == DGPU_E_BADARG);
              EXPECT(dgpu_legogroth16_prove_host(&hk, 0, nullptr, 0, 0, z.data(), n_inst, z.data() + 4 * n_inst, V - n_inst, 0, r, s, v, a, b, c, d, inf) == DGPU_E_BADARG); }
        }
        // argument check added in round 4: n_inst must agree with the resident circuit
        { uint64_t a[12], b[24], c[12], d[12]; uint8_t inf[4]; EXPECT(dgpu_legogroth16_prove(&pk, circuit, 0, z.data(), V, n_inst + 1, 0, r, s, v, a, b, c, d, inf) == DGPU_E_BADARG); }
        drop(circuit); drop_key(pk);
    }
}

// 3. the threaded host entry points of dock_gt.cpp / dock_serde.cpp from several callers at once
static void test_gt_and_serde() {
    const size_t n = 96;
    std::vector<uint64_t> a(72 * n, 0), e(4 * n);
    hostf::Fq12 one = hostf::Fq12::one();
    for (size_t i = 0; i < n; i++) { memcpy(&a[72 * i], &one, 576); e[4 * i] = i + 3; e[4 * i + 1] = e[4 * i + 2] = e[4 * i + 3] = 0; }
    std::vector<std::thread> th;
    for (int t = 0; t < 4; t++) th.emplace_back([&] {
        uint64_t out[72];
        EXPECT(dgpu_fp12_multi_pow(a.data(), e.data(), n, out) == DGPU_OK);
        EXPECT(memcmp(out, &one, 576) == 0);
        // G1 identity points round-trip through the (threaded, validating) codec
        std::vector<uint64_t> xy(12 * 64, 0), back(12 * 64, 1); std::vector<uint8_t> inf(64, 1), inf2(64, 0), bytes(48 * 64);
        EXPECT(dgpu_g1_serialize(xy.data(), inf.data(), 64, 1, bytes.data()) == DGPU_OK);
        EXPECT(dgpu_g1_deserialize(bytes.data(), 64, 1, back.data(), inf2.data()) == DGPU_OK);
        for (int i = 0; i < 64; i++) EXPECT(inf2[i] == 1);
    });
    for (auto &x : th) x.join();
}

// 4. the aggregation (dock_aggregation.cpp: protocol steps as parallel sections on the pool, nested inside each other, the caller's transcript called back,
//    fold tables freed on every path) from several callers at once, with and without failing device calls
struct ToyTranscript {
    uint64_t h = 0xcbf29ce484222325ULL; std::atomic<int> calls{0};
    static void append(void *c, const uint8_t *l, size_t ll, const uint8_t *b, size_t n) { ToyTranscript *t = (ToyTranscript *)c; t->calls++; for (size_t i = 0; i < ll; i++) t->h = (t->h ^ l[i]) * 0x100000001b3ULL; for (size_t i = 0; i < n; i++) t->h = (t->h ^ b[i]) * 0x100000001b3ULL; }
    static void challenge(void *c, const uint8_t *l, size_t ll, uint64_t out[4]) { ToyTranscript *t = (ToyTranscript *)c; t->calls++; for (size_t i = 0; i < ll; i++) t->h = (t->h ^ l[i]) * 0x100000001b3ULL; for (int k = 0; k < 4; k++) { t->h ^= t->h >> 29; t->h *= 0xff51afd7ed558ccdULL; out[k] = t->h; } out[3] &= 0x3fffffffffffffffULL; out[0] |= 1; }
};
static void test_aggregation() {
    const size_t n = 16;
    std::vector<uint64_t> g1(12 * 2 * n), g2(24 * 2 * n);
    for (size_t i = 0; i < g1.size(); i++) g1[i] = 0x1111 * (i + 1);
    for (size_t i = 0; i < g2.size(); i++) g2[i] = 0x2222 * (i + 3);
    dgpu_snarkpack_prover_srs srs{n, g1.data(), g1.data(), g2.data(), g2.data(), g2.data(), g2.data(), g1.data() + 12 * n, g1.data() + 12 * n};
    dgpu_snarkpack_verifier_srs vs{n, g1.data(), g2.data(), g1.data() + 12, g1.data() + 24, g2.data() + 24, g2.data() + 48};
    dgpu_groth16_vk vk{g1.data(), g2.data(), g2.data() + 24, g2.data() + 48, g1.data() + 36, 2};
    std::vector<uint64_t> pub(4 * n, 7);
    for (int with_d = 0; with_d < 2; with_d++)
        for (uint64_t fail_every : {(uint64_t)0, (uint64_t)23}) {
            g_fail_every = fail_every;
            std::atomic<int> ok{0}, bad{0};
            std::vector<std::thread> th;
            for (int t = 0; t < 4; t++) th.emplace_back([&] {
                for (int rep = 0; rep < 3; rep++) {
                    ToyTranscript tp; dgpu_transcript tr{&tp, ToyTranscript::append, ToyTranscript::challenge};
                    const size_t cap = dgpu_snarkpack_proof_words(n, with_d); std::vector<uint64_t> proof(cap); size_t len = 0;
                    int32_t rc = dgpu_snarkpack_aggregate(&srs, g1.data(), g2.data(), g1.data() + 24, with_d ? g1.data() + 48 : nullptr, n, &tr, proof.data(), cap, &len);
                    if (rc) { EXPECT(rc == DGPU_E_OOM); bad++; continue; }
                    EXPECT(len == cap && proof[0] == n && proof[1] == (uint64_t)(with_d ? 2 : 1) && tp.calls.load() > 20);
                    ToyTranscript tv; dgpu_transcript trv{&tv, ToyTranscript::append, ToyTranscript::challenge};
                    const uint64_t rnd[4] = {5, 0, 0, 0}; int32_t okv = -1;
                    rc = dgpu_snarkpack_verify(&vs, &vk, pub.data(), n, 1, proof.data(), len, with_d ? 1 : 0, nullptr, rnd, &trv, DGPU_SNARKPACK_VALIDATE_GT, &okv);
                    if (rc) { EXPECT(rc == DGPU_E_OOM); bad++; continue; }
                    EXPECT(okv == 0 || okv == 1);                          // (the stand-ins compute nothing: either answer, never a crash or a leak)
                    // malformed proofs: refused before any call
                    EXPECT(dgpu_snarkpack_verify(&vs, &vk, pub.data(), n, 1, proof.data(), len - 1, with_d ? 1 : 0, nullptr, rnd, &trv, 0, &okv) == DGPU_E_BADARG);
                    EXPECT(dgpu_snarkpack_verify(&vs, &vk, pub.data(), n - 1, 1, proof.data(), len, with_d ? 1 : 0, nullptr, rnd, &trv, 0, &okv) == DGPU_E_BADARG);
                    ok++;
                }
            });
            for (auto &x : th) x.join();
            // many proofs of one key in one call (dgpu_legogroth16_verify_batch): the four concurrent pieces, operand sizes, error paths
            {
                std::vector<uint64_t> pc(2 * (size_t)DGPU_G2_PREPARED_WORDS, 3), gt(72, 9);
                const uint64_t rnd[4] = {5, 0, 0, 0}, zero[4] = {0, 0, 0, 0}; int32_t okv = -1;
                for (int rep = 0; rep < 6; rep++) {
                    const int32_t rc = dgpu_legogroth16_verify_batch(gt.data(), pc.data(), pc.data() + DGPU_G2_PREPARED_WORDS, g1.data(), 2, g1.data(), g2.data(), g1.data() + 12, g1.data() + 24, n, pub.data(), 1, rep & 1, rnd, &okv);
                    EXPECT(rc == DGPU_OK ? (okv == 0 || okv == 1) : rc == DGPU_E_OOM);
                }
                EXPECT(dgpu_legogroth16_verify_batch(gt.data(), pc.data(), pc.data(), g1.data(), 2, g1.data(), g2.data(), g1.data(), g1.data(), n, pub.data(), 1, 0, zero, &okv) == DGPU_E_BADARG);
                EXPECT(dgpu_legogroth16_verify_batch(gt.data(), pc.data(), pc.data(), g1.data(), 1, g1.data(), g2.data(), g1.data(), g1.data(), n, pub.data(), 1, 0, rnd, &okv) == DGPU_E_BADARG);
                EXPECT(dgpu_legogroth16_verify_batch(gt.data(), pc.data(), pc.data(), g1.data(), 2, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 1, 0, rnd, &okv) == DGPU_OK && okv == 1);
            }
            g_fail_every = 0;
            if (!fail_every) EXPECT(ok == 12 && bad == 0); else EXPECT(bad > 0);
            EXPECT(g_live_fold.load() == 0);                                // every fold table of every call was freed, failed or not
        }
}

// fork() while the pool exists (Python's multiprocessing forks a process that has used the library): the child's pool starts empty and its first
// par_run creates workers of its own; a fork issued while OTHER threads are hammering the pool must not leave the child a locked mutex
#include <sys/wait.h>
#include <unistd.h>
static void test_fork_with_a_live_pool() {
    std::atomic<bool> stop{false};
    std::thread hammer([&] { while (!stop.load()) (void)dock::par_run(8, [](size_t) -> int32_t { return DGPU_OK; }); });
    for (int round = 0; round < 8; round++) {
        const pid_t pid = fork();
        if (pid == 0) {
            std::atomic<int> sum{0};
            const int32_t rc = dock::par_run(16, [&](size_t k) -> int32_t { sum += (int)k; return DGPU_OK; });
            _exit(rc == DGPU_OK && sum.load() == 120 ? 0 : 1);
        }
        int st = 0;
        if (pid < 0 || waitpid(pid, &st, 0) != pid || !WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "forked child failed (round %d, status %d)\n", round, st); failures++; }
    }
    stop = true; hammer.join();
}

int main() {
#if !defined(__SANITIZE_THREAD__) && !defined(__SANITIZE_ADDRESS__)      // (the sanitizer runtimes do not support new threads in the child of a multi-threaded fork: the plain build of tests/test_host_sanitizers.py runs this)
    test_fork_with_a_live_pool();
#endif
    test_slots_and_handles();
    test_prover();
    test_gt_and_serde();
    test_aggregation();
    { std::lock_guard<std::mutex> lk(gs.mu); if (!gs.handles.empty()) { fprintf(stderr, "%zu handles left\n", gs.handles.size()); failures++; } }
    printf("host_sanitize_driver: %s (%llu stand-in device calls)\n", failures.load() ? "FAILED" : "ok", (unsigned long long)g_calls.load());
    return failures.load() ? 1 : 0;
}
