// include/dock_gpu.hpp — C++ host side above the C ABI (include/dock_gpu.h): the reference's call surface for the hot path,
// same names, argument meaning and error behaviour, for a compiled host (the reference is Rust; no Rust toolchain exists in the
// build image, so the compiled-language mirror is C++ and INTEGRATION.md shows the Rust binding).  Header-only; link -ldock_gpu.
//
//   ark-ec VariableBaseMSM::{msm, msm_unchecked, msm_bigint}     -> dock_gpu::VariableBaseMSM<G>      (called from
//       utils/src/pairs.rs:143-156, legogroth16/src/prover.rs:286,299,363,592, utils/src/randomized_mult_checker.rs:100)
//   dock_crypto_utils::pairs::Pairs                               -> dock_gpu::Pairs<G>
//   dock_crypto_utils::msm::WindowTable, multiply_field_elems_with_same_group_elem (utils/src/msm.rs:8-62)
//                                                                 -> dock_gpu::WindowTable<G>, multiply_field_elems_with_same_group_elem
//   ark-ec Pairing::{multi_miller_loop, final_exponentiation, multi_pairing} (utils/src/randomized_pairing_check.rs:207,213,
//       legogroth16/src/verifier.rs:69-78)                        -> dock_gpu::multi_miller_loop / final_exponentiation / multi_pairing
//   ProvingKey queries kept on the device (legogroth16/src/data_structures.rs:151-168) -> dock_gpu::DeviceBases<G>
//   dock_crypto_utils::randomized_pairing_check::RandomizedPairingChecker, randomized_mult_checker::RandomizedMultChecker -> the same names
//   legogroth16 aggregation (aggregate_proofs / verify_aggregate_proof)                -> dock_gpu::aggregation::*
//
// Errors: arkworks' MSM has no failure mode; here a negative ABI code throws dock_gpu::Error (a Rust shim falls back to arkworks
// instead).  `msm` keeps arkworks' checked-length contract: Err(min_len) when the lengths differ.
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "dock_gpu.h"

namespace dock_gpu {

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const char *what) : std::runtime_error(std::string(what) + ": " + dgpu_strerror(c)), code(c) {}
};
inline void check(int32_t rc, const char *what) { if (rc != DGPU_OK) throw Error(rc, what); }
// this mirror has no CPU path to stay on, so the size threshold (DGPU_E_TOO_SMALL below DGPU_DEFAULT_MIN_GPU_N terms) is lowered to 0
inline void init(int device = 0) { check(dgpu_init(device), "dgpu_init"); check(dgpu_set_min_gpu_n(0), "dgpu_set_min_gpu_n"); }
// several GPUs in one process: context k on HIP device physical[k]
inline void init_devices(const std::vector<int32_t> &physical) { check(dgpu_init_device_list(physical.data(), (int32_t)physical.size()), "dgpu_init_device_list"); check(dgpu_set_min_gpu_n(0), "dgpu_set_min_gpu_n"); }

// the resident-bases cache behind the one-shot MSM calls (include/dock_gpu.h dgpu_set_bases_cache_*): what makes VariableBaseMSM::msm_bigint below — the
// unmodified call on the caller's own vector of Affine structs — run on a resident table from its second use of that vector on
namespace bases_cache {
struct Stats { uint64_t hits, misses, fills, stale, evictions, bytes, budget, entries; };
inline void set_bytes(size_t bytes) { check(dgpu_set_bases_cache_bytes(bytes), "set_bases_cache_bytes"); }
inline void set_min_n(size_t n) { check(dgpu_set_bases_cache_min_n(n), "set_bases_cache_min_n"); }
inline void verify_samples(int32_t samples) { check(dgpu_set_bases_cache_verify(samples), "set_bases_cache_verify"); }      // DGPU_CACHE_VERIFY_FULL: every record
template <class T> void invalidate(const std::vector<T> &v) { check(dgpu_bases_cache_invalidate(v.data(), v.size() * sizeof(T)), "bases_cache_invalidate"); }
inline void clear() { check(dgpu_bases_cache_clear(), "bases_cache_clear"); }
inline Stats stats() { uint64_t w[8]; check(dgpu_bases_cache_stats(w), "bases_cache_stats"); return Stats{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]}; }
}  // namespace bases_cache

using Fq = std::array<uint64_t, 6>;        // Montgomery limbs, R = 2^384 (ark-ff Fp384 layout)
using BigInt256 = std::array<uint64_t, 4>; // canonical scalar (Fr::into_bigint)
struct Fr { std::array<uint64_t, 4> mont; };   // Montgomery limbs, R = 2^256 (what ark-ff's Fr holds)
using Fq12 = std::array<uint64_t, 72>;     // c0.c0.c0 ... c1.c2.c1

struct G1 {
    static constexpr size_t AW = 12;        // u64 per affine point
    struct Affine { Fq x{}, y{}; bool infinity = true; };
    struct Projective { Fq x{}, y{}, z{}; bool is_zero() const { for (auto w : z) if (w) return false; return true; } };
    static int32_t msm(const uint64_t *b, const uint8_t *i, const uint64_t *s, size_t n, uint64_t *o) { return dgpu_msm_g1(b, i, s, n, o); }
    static int32_t msm_mont(const uint64_t *b, const uint8_t *i, const uint64_t *s, size_t n, uint64_t *o) { return dgpu_msm_g1_mont(b, i, s, n, o); }
    static int32_t upload(const uint64_t *b, const uint8_t *i, size_t n, uint64_t *h) { return dgpu_bases_upload_g1(b, i, n, h); }
    static int32_t msm_handle(uint64_t h, size_t off, const uint64_t *s, size_t n, int32_t mont, uint64_t *o) { return dgpu_msm_g1_handle(h, off, s, n, mont, o); }
    static int32_t msm_strided(const void *b, size_t st, size_t xo, size_t yo, size_t io, const uint64_t *s, size_t n, int32_t mont, uint64_t *o) { return dgpu_msm_g1_strided(b, st, xo, yo, io, s, n, mont, o); }
    static int32_t upload_strided(const void *b, size_t st, size_t xo, size_t yo, size_t io, size_t n, uint64_t *h) { return dgpu_bases_upload_g1_strided(b, st, xo, yo, io, n, h); }
    static int32_t table(const uint64_t *b, uint64_t *h) { return dgpu_window_table_g1(b, h); }
    static int32_t table_mul(uint64_t t, const uint64_t *s, size_t n, int32_t mont, uint64_t *o, uint8_t *oi) { return dgpu_window_table_mul_g1(t, s, n, mont, o, oi); }
};
struct G2 {
    static constexpr size_t AW = 24;
    struct Affine { std::array<Fq, 2> x{}, y{}; bool infinity = true; };      // Fq2 = c0, c1
    struct Projective { std::array<Fq, 2> x{}, y{}, z{}; bool is_zero() const { for (auto &c : z) for (auto w : c) if (w) return false; return true; } };
    static int32_t msm(const uint64_t *b, const uint8_t *i, const uint64_t *s, size_t n, uint64_t *o) { return dgpu_msm_g2(b, i, s, n, o); }
    static int32_t msm_mont(const uint64_t *b, const uint8_t *i, const uint64_t *s, size_t n, uint64_t *o) { return dgpu_msm_g2_mont(b, i, s, n, o); }
    static int32_t upload(const uint64_t *b, const uint8_t *i, size_t n, uint64_t *h) { return dgpu_bases_upload_g2(b, i, n, h); }
    static int32_t msm_handle(uint64_t h, size_t off, const uint64_t *s, size_t n, int32_t mont, uint64_t *o) { return dgpu_msm_g2_handle(h, off, s, n, mont, o); }
    static int32_t msm_strided(const void *b, size_t st, size_t xo, size_t yo, size_t io, const uint64_t *s, size_t n, int32_t mont, uint64_t *o) { return dgpu_msm_g2_strided(b, st, xo, yo, io, s, n, mont, o); }
    static int32_t upload_strided(const void *b, size_t st, size_t xo, size_t yo, size_t io, size_t n, uint64_t *h) { return dgpu_bases_upload_g2_strided(b, st, xo, yo, io, n, h); }
    static int32_t table(const uint64_t *b, uint64_t *h) { return dgpu_window_table_g2(b, h); }
    static int32_t table_mul(uint64_t t, const uint64_t *s, size_t n, int32_t mont, uint64_t *o, uint8_t *oi) { return dgpu_window_table_mul_g2(t, s, n, mont, o, oi); }
};

// ark-ec's Affine { x, y, infinity } is not a flat array.  The MSM entry points take the array of structs as it is (dgpu_msm_*_strided:
// stride and field offsets are arguments — a Rust shim passes size_of::<G1Affine>() and offset_of!(G1Affine, x / y / infinity), this mirror
// sizeof / offsetof of its own Affine, which has the same 104 / 200-byte layout); the pairing and fixed-base entry points, whose operand
// counts are small, still take the packed ABI form (coordinates + one flag byte per point):
template <class G> struct Packed {
    std::vector<uint64_t> xy; std::vector<uint8_t> inf;
    explicit Packed(const std::vector<typename G::Affine> &pts, size_t n) : xy(n * G::AW, 0), inf(n, 0) {
        for (size_t i = 0; i < n; i++) {
            if (pts[i].infinity) { inf[i] = 1; continue; }
            std::memcpy(&xy[i * G::AW], &pts[i].x, G::AW * 4);
            std::memcpy(&xy[i * G::AW + G::AW / 2], &pts[i].y, G::AW * 4);
        }
    }
};
template <class G> typename G::Projective projective_from_abi(const uint64_t *o) {
    typename G::Projective p;
    std::memcpy(&p.x, o, G::AW * 4); std::memcpy(&p.y, o + G::AW / 2, G::AW * 4); std::memcpy(&p.z, o + G::AW, G::AW * 4);
    return p;
}
template <class G> std::vector<typename G::Affine> affine_from_abi(const std::vector<uint64_t> &xy, const std::vector<uint8_t> &inf) {
    std::vector<typename G::Affine> out(inf.size());
    for (size_t i = 0; i < inf.size(); i++) {
        out[i].infinity = inf[i] != 0;
        if (out[i].infinity) continue;
        std::memcpy(&out[i].x, &xy[i * G::AW], G::AW * 4); std::memcpy(&out[i].y, &xy[i * G::AW + G::AW / 2], G::AW * 4);
    }
    return out;
}

template <class G> struct VariableBaseMSM {
    using Affine = typename G::Affine; using Projective = typename G::Projective;
    // msm_bigint(bases, bigints): truncates to the shorter operand (legogroth16/src/prover.rs:286 relies on it)
    static Projective msm_bigint(const std::vector<Affine> &bases, const std::vector<BigInt256> &bigints) {
        size_t n = std::min(bases.size(), bigints.size());
        std::array<uint64_t, G::AW * 3 / 2> out{};
        check(G::msm_strided(bases.data(), sizeof(Affine), offsetof(Affine, x), offsetof(Affine, y), offsetof(Affine, infinity), n ? bigints[0].data() : nullptr, n, 0, out.data()), "msm_bigint");
        return projective_from_abi<G>(out.data());
    }
    // msm_unchecked(bases, scalars): scalars in Montgomery form, converted on the device
    static Projective msm_unchecked(const std::vector<Affine> &bases, const std::vector<Fr> &scalars) {
        size_t n = std::min(bases.size(), scalars.size());
        std::array<uint64_t, G::AW * 3 / 2> out{};
        check(G::msm_strided(bases.data(), sizeof(Affine), offsetof(Affine, x), offsetof(Affine, y), offsetof(Affine, infinity), n ? scalars[0].mont.data() : nullptr, n, 1, out.data()), "msm_unchecked");
        return projective_from_abi<G>(out.data());
    }
    // msm(bases, scalars): Err(min_len) on a length mismatch — returned as {nullopt, min_len}
    static std::pair<std::optional<Projective>, size_t> msm(const std::vector<Affine> &bases, const std::vector<Fr> &scalars) {
        size_t n = std::min(bases.size(), scalars.size());
        if (bases.size() != scalars.size()) return {std::nullopt, n};
        return {msm_unchecked(bases, scalars), n};
    }
};

// utils/src/pairs.rs:143-156
template <class G> struct Pairs {
    std::vector<typename G::Affine> left; std::vector<Fr> right;
    typename G::Projective msm() const { return VariableBaseMSM<G>::msm_unchecked(left, right); }
    typename G::Projective msm_bigint(const std::vector<BigInt256> &right_bigint) const { return VariableBaseMSM<G>::msm_bigint(left, right_bigint); }
};

// a proving-key query kept in HBM (ProvingKeyCommon, legogroth16/src/data_structures.rs:151-168); offset = 1 is &query[1..]
template <class G> class DeviceBases {
    uint64_t h_ = 0; size_t n_ = 0;
public:
    explicit DeviceBases(const std::vector<typename G::Affine> &bases) : n_(bases.size()) {
        using A = typename G::Affine;
        check(G::upload_strided(bases.data(), sizeof(A), offsetof(A, x), offsetof(A, y), offsetof(A, infinity), n_, &h_), "bases_upload");
    }
    DeviceBases(const DeviceBases &) = delete;
    ~DeviceBases() { if (h_) dgpu_bases_free(h_); }
    size_t len() const { return n_; }
    uint64_t handle() const { return h_; }
    // per-key setup: MSMs over this query run on a precomputed-multiples table from now on (dgpu_bases_precompute_*); same results.
    // window_bits: 0 = automatic (full-width scalars: an h query), DGPU_TABLE_C_WITNESS for queries that meet a witness (a / b / l)
    void precompute(int32_t window_bits = 0) { check(G::AW == 12 ? dgpu_bases_precompute_g1(h_, window_bits) : dgpu_bases_precompute_g2(h_, window_bits), "bases_precompute"); }
    typename G::Projective msm_bigint(const std::vector<BigInt256> &bigints, size_t offset = 0) const {
        size_t n = std::min(n_ - offset, bigints.size());
        std::array<uint64_t, G::AW * 3 / 2> out{};
        check(G::msm_handle(h_, offset, n ? bigints[0].data() : nullptr, n, 0, out.data()), "msm_handle");
        return projective_from_abi<G>(out.data());
    }
};

// utils/src/msm.rs:8-52.  `num_multiplications` only sizes arkworks' window; the device table is fixed (accepted, ignored).
template <class G> class WindowTable {
    uint64_t h_ = 0;
public:
    WindowTable(size_t /*num_multiplications*/, const typename G::Affine &group_elem) {
        std::vector<typename G::Affine> one{group_elem}; Packed<G> p(one, 1);
        check(G::table(p.xy.data(), &h_), "window_table");
    }
    WindowTable(const WindowTable &) = delete;
    ~WindowTable() { if (h_) dgpu_window_table_free(h_); }
    std::vector<typename G::Affine> multiply_many(const std::vector<Fr> &elements) const {
        std::vector<uint64_t> xy(elements.size() * G::AW); std::vector<uint8_t> inf(elements.size());
        check(G::table_mul(h_, elements.empty() ? nullptr : elements[0].mont.data(), elements.size(), 1, xy.data(), inf.data()), "multiply_many");
        return affine_from_abi<G>(xy, inf);
    }
    typename G::Affine multiply(const Fr &element) const { return multiply_many({element})[0]; }
};
template <class G> std::vector<typename G::Affine> multiply_field_elems_with_same_group_elem(const typename G::Affine &group_elem, const std::vector<Fr> &elements) {
    return WindowTable<G>(elements.size(), group_elem).multiply_many(elements);
}

// Pairing::multi_miller_loop: equal lengths required (arkworks' zip_eq panics); pairs with an identity member are skipped
inline Fq12 multi_miller_loop(const std::vector<G1::Affine> &a, const std::vector<G2::Affine> &b) {
    if (a.size() != b.size()) throw Error(DGPU_E_LENGTH, "multi_miller_loop");
    Packed<G1> p(a, a.size()); Packed<G2> q(b, b.size());
    std::vector<uint8_t> skip(a.size());
    for (size_t i = 0; i < a.size(); i++) skip[i] = p.inf[i] | q.inf[i];
    Fq12 out{};
    check(dgpu_multi_miller_loop(p.xy.data(), q.xy.data(), skip.data(), a.size(), out.data()), "multi_miller_loop");
    return out;
}
// Pairing::final_exponentiation: None for a zero input
inline std::optional<Fq12> final_exponentiation(const Fq12 &f) {
    Fq12 out{};
    int32_t rc = dgpu_final_exponentiation(f.data(), out.data());
    if (rc == DGPU_E_ZERO) return std::nullopt;
    check(rc, "final_exponentiation");
    return out;
}
inline Fq12 multi_pairing(const std::vector<G1::Affine> &a, const std::vector<G2::Affine> &b) { return *final_exponentiation(multi_miller_loop(a, b)); }


// ---- RandomizedPairingChecker (utils/src/randomized_pairing_check.rs:24-215) over the C ABI ---------------------------------------------------
// Same state and laziness as the reference's type: `left` the product of Miller outputs so far (:27), `right` the GT target (`right +=
// out.mul_bigint(m)` — GT is written additively in arkworks, it is the Fp12 product, :30,:136), `pending` the (G1, G2) pairs queued for ONE
// multi_miller_loop in verify() when lazy (:34,:204-214), `random` / `current_random` = r, r^k (:36-38): equation k is scaled by r^k.
// The G1 scalings `a.mul_bigint(m)` (:125-127,:152-158) run batched on the device (dgpu_g1_scale_batch: lazy mode scales everything
// queued in one launch), the Miller loops through dgpu_multi_miller_loop, GT arithmetic and the single final exponentiation on the host
// inside the library.  crypto_amd/pairing_check.py is the same type for the Python tests; tests/native/cpp_api_driver.cpp drives this one.
namespace detail {
inline constexpr BigInt256 FR_MODULUS = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
inline bool geq(const BigInt256 &a, const BigInt256 &b) { for (int i = 3; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; } return true; }
inline BigInt256 add_mod(const BigInt256 &a, const BigInt256 &b) {          // a, b < r
    BigInt256 s{}; unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) { c += (unsigned __int128)a[i] + b[i]; s[i] = (uint64_t)c; c >>= 64; }
    if (c || geq(s, FR_MODULUS)) { unsigned __int128 br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)s[i] - FR_MODULUS[i] - (uint64_t)br; s[i] = (uint64_t)d; br = (d >> 64) & 1; } }
    return s;
}
inline BigInt256 neg_mod(const BigInt256 &a) { BigInt256 z{}; if (a == z) return a; BigInt256 r = FR_MODULUS; unsigned __int128 br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)r[i] - a[i] - (uint64_t)br; r[i] = (uint64_t)d; br = (d >> 64) & 1; } return r; }          // a < r
inline BigInt256 mul_mod(const BigInt256 &a, const BigInt256 &b) {          // double-and-add: a handful of calls per checker
    BigInt256 acc{};
    for (int i = 255; i >= 0; i--) { acc = add_mod(acc, acc); if ((b[i / 64] >> (i % 64)) & 1) acc = add_mod(acc, a); }
    return acc;
}
inline Fq12 fq12_one() { Fq12 o{}; const Fq one = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}; std::memcpy(o.data(), one.data(), 48); return o; }
inline Fq12 fq12_mul(const Fq12 &a, const Fq12 &b) { Fq12 o{}; check(dgpu_fp12_mul(a.data(), b.data(), o.data()), "fp12_mul"); return o; }
}  // namespace detail

class RandomizedPairingChecker {
    Fq12 left_ = detail::fq12_one(), right_ = detail::fq12_one();
    bool lazy_;
    struct Queued { std::vector<G1::Affine> a; BigInt256 m; bool negate; std::vector<G2::Affine> b; };
    std::vector<Queued> pending_;
    std::vector<std::pair<Fq12, BigInt256>> pending_targets_;
    BigInt256 random_, current_random_{1, 0, 0, 0};
    // prod e([+-m_i] a_i, b_i) as ONE call (dgpu_multi_miller_loop_scaled: the scalings run beside the chain of the b_i); a negated source is scaled by r - m
    static Fq12 scaled_miller_loop(const std::vector<G1::Affine> &a, const std::vector<BigInt256> &m, const std::vector<uint8_t> &neg, const std::vector<G2::Affine> &b) {
        if (a.size() != b.size()) throw Error(DGPU_E_LENGTH, "scaled_miller_loop");
        Packed<G1> p(a, a.size()); Packed<G2> q(b, b.size());
        std::vector<uint8_t> skip(a.size());
        for (size_t i = 0; i < a.size(); i++) skip[i] = p.inf[i] | q.inf[i];
        std::vector<BigInt256> ms(m);
        if (!neg.empty()) {
            if (ms.size() == 1) ms.assign(a.size(), m[0]);
            for (size_t i = 0; i < ms.size(); i++) if (neg[i]) ms[i] = detail::neg_mod(ms[i]);
        }
        Fq12 out{};
        check(dgpu_multi_miller_loop_scaled(p.xy.data(), ms.empty() ? nullptr : ms[0].data(), ms.size() == 1 ? 0 : 4, q.xy.data(), skip.data(), a.size(), nullptr, nullptr, nullptr, 0, out.data()),
              "multi_miller_loop_scaled");
        return out;
    }
    void advance() { current_random_ = detail::mul_mod(current_random_, random_); }
public:
    RandomizedPairingChecker(const BigInt256 &random, bool lazy) : lazy_(lazy), random_(random) {}                     // new(random, lazy)  :44-53
    // e(a, b) == out   :61-77
    void add_sources_and_target(const G1::Affine &a, const G2::Affine &b, const Fq12 &out) { add_multiple_sources_and_target({a}, {b}, out); }
    // e(a, b) == e(c, d)   :104-113
    void add_sources(const G1::Affine &a, const G2::Affine &b, const G1::Affine &c, const G2::Affine &d) { add_multiple_sources({a}, {b}, {c}, {d}); }
    // prod e(a_i, b_i) == out   :116-138
    void add_multiple_sources_and_target(const std::vector<G1::Affine> &a, const std::vector<G2::Affine> &b, const Fq12 &out) {
        if (a.size() != b.size()) throw Error(DGPU_E_LENGTH, "add_multiple_sources_and_target");
        const BigInt256 m = current_random_;
        if (lazy_) { pending_.push_back({a, m, false, b}); pending_targets_.push_back({out, m}); }
        else {
            left_ = detail::fq12_mul(left_, scaled_miller_loop(a, {m}, {}, b));
            Fq12 pw{}; check(dgpu_fp12_pow(out.data(), m.data(), pw.data()), "fp12_pow");
            right_ = detail::fq12_mul(right_, pw);
        }
        advance();
    }
    // prod e(a_i, b_i) == prod e(c_i, d_i)   :142-173
    void add_multiple_sources(const std::vector<G1::Affine> &a, const std::vector<G2::Affine> &b, const std::vector<G1::Affine> &c, const std::vector<G2::Affine> &d) {
        if (a.size() != b.size() || c.size() != d.size()) throw Error(DGPU_E_LENGTH, "add_multiple_sources");
        const BigInt256 m = current_random_;
        if (lazy_) { pending_.push_back({a, m, false, b}); pending_.push_back({c, m, true, d}); }
        else {
            left_ = detail::fq12_mul(left_, scaled_miller_loop(a, {m}, {}, b));
            left_ = detail::fq12_mul(left_, scaled_miller_loop(c, {m}, std::vector<uint8_t>(c.size(), 1), d));
        }
        advance();
    }
    // :204-214
    bool verify() {
        Fq12 left = left_;
        if (!pending_targets_.empty()) {
            std::vector<uint64_t> bases, exps;
            for (auto &t : pending_targets_) { bases.insert(bases.end(), t.first.begin(), t.first.end()); exps.insert(exps.end(), t.second.begin(), t.second.end()); }
            Fq12 pw{}; check(dgpu_fp12_multi_pow(bases.data(), exps.data(), pending_targets_.size(), pw.data()), "fp12_multi_pow");
            right_ = detail::fq12_mul(right_, pw);
            pending_targets_.clear();
        }
        if (!pending_.empty()) {
            std::vector<G1::Affine> pts; std::vector<G2::Affine> qs; std::vector<BigInt256> ms; std::vector<uint8_t> neg;
            for (auto &q : pending_) for (size_t i = 0; i < q.a.size(); i++) { pts.push_back(q.a[i]); qs.push_back(q.b[i]); ms.push_back(q.m); neg.push_back(q.negate ? 1 : 0); }
            left = detail::fq12_mul(scaled_miller_loop(pts, ms, neg, qs), left);
            pending_.clear();
        }
        const auto gt = final_exponentiation(left);
        if (!gt) throw Error(DGPU_E_ZERO, "final_exponentiation");              // arkworks: .unwrap() panics
        return *gt == right_;
    }
};

// ---- RandomizedMultChecker (utils/src/randomized_mult_checker.rs:11-118) over the MSM entry points ------------------------------------------------
// Many claimed scalar multiplications / small MSMs `sum_i s_i P_i == T` batched with the powers of a random scalar into ONE variable-base MSM whose
// result must be the identity (`G::Group::msm_unchecked(&points, &scalars).is_zero()`, :100 — one of the reference's large-n MSM call sites).  Same
// state and merging rule: `args` maps a point's x coordinate to (scalar, point), so a point and its negative share an entry (:104-117); the identity
// is ignored.  crypto_amd/mult_checker.py is the same type for the Python tests.
template <class G> class RandomizedMultChecker {
    using Affine = typename G::Affine;
    struct Entry { BigInt256 s; Affine p; };
    std::vector<Entry> args_;                                   // (a linear scan over x: the reference's BTreeMap keyed by x, same merging)
    BigInt256 random_, current_random_{1, 0, 0, 0};
    static BigInt256 neg_mod(const BigInt256 &a) { BigInt256 z{}; if (a == z) return a; BigInt256 r = detail::FR_MODULUS; unsigned __int128 br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)r[i] - a[i] - (uint64_t)br; r[i] = (uint64_t)d; br = (d >> 64) & 1; } return r; }
    void add(const Affine &p, const BigInt256 &s) {             // :104-117
        if (p.infinity) return;
        for (auto &e : args_)
            if (std::memcmp(&e.p.x, &p.x, sizeof p.x) == 0) { e.s = detail::add_mod(e.s, std::memcmp(&e.p.y, &p.y, sizeof p.y) == 0 ? s : neg_mod(s)); return; }      // same x, other y: the stored point is -p
        args_.push_back({s, p});
    }
public:
    explicit RandomizedMultChecker(const BigInt256 &random) : random_(random) {}                       // new(random)  :20-26
    size_t len() const { return args_.size(); }
    // sum b_i a_i == t   :64-75 (add_1 / add_2 / add_3 are this with one, two, three terms, :32-61); scalars canonical (Fr::into_bigint)
    void add_many(const std::vector<Affine> &a, const std::vector<BigInt256> &b, const Affine &t) {
        if (a.size() != b.size()) throw Error(DGPU_E_LENGTH, "add_many");
        for (size_t i = 0; i < a.size(); i++) add(a[i], detail::mul_mod(current_random_, b[i]));
        add(t, neg_mod(current_random_));
        current_random_ = detail::mul_mod(current_random_, random_);
    }
    void add_1(const Affine &p, const BigInt256 &s, const Affine &t) { add_many({p}, {s}, t); }
    // :78-86: one MSM, result must be the identity
    bool verify() const {
        if (args_.empty()) return true;
        std::vector<Affine> pts; std::vector<BigInt256> sc;
        for (auto &e : args_) { pts.push_back(e.p); sc.push_back(e.s); }
        return VariableBaseMSM<G>::msm_bigint(pts, sc).is_zero();
    }
};

// ---- Circom's binary `.r1cs` (legogroth16/src/circom/r1cs_reader.rs:16-140) -> the CSR matrices dgpu_r1cs_upload takes --------------------------------
// magic "r1cs", version 1, sections {1: header, 2: constraints, 3: wire2label}; 32-byte little-endian field elements; a constraint is three linear
// combinations (A, B, C) of (wire id, coefficient) terms.  Wire order is Circom's (0 = one, public outputs, public inputs, private inputs,
// intermediates), i.e. the assignment order (1, instance..., witness...) of the witness map.  Errors are the reader's: std::runtime_error with its message.
namespace circom {
struct Csr { std::vector<uint64_t> rowptr{0}; std::vector<uint32_t> cols; std::vector<uint64_t> vals; };       // vals: 4 canonical words per entry
struct R1CSFile {
    BigInt256 prime{}; uint32_t n_wires = 0, n_pub_out = 0, n_pub_in = 0, n_prv_in = 0, n_constraints = 0; uint64_t n_labels = 0;
    Csr a, b, c; std::vector<uint64_t> wire_mapping;
    size_t num_inputs() const { return 1 + (size_t)n_pub_out + n_pub_in; }           // instance variables incl. the constant one
    static R1CSFile parse(const uint8_t *d, size_t len) {
        auto fail = [](const char *m) -> void { throw std::runtime_error(m); };
        auto need = [&](size_t off, size_t cnt) { if (off > len || cnt > len - off) fail("unexpected end of file"); };
        auto u32 = [&](size_t off) { need(off, 4); uint32_t v; std::memcpy(&v, d + off, 4); return v; };
        auto u64 = [&](size_t off) { need(off, 8); uint64_t v; std::memcpy(&v, d + off, 8); return v; };
        need(0, 12);
        if (std::memcmp(d, "r1cs", 4) != 0) fail("Invalid magic number");
        if (u32(4) != 1) fail("Unsupported version");
        size_t off = 12, sec_off[4] = {0, 0, 0, 0}, sec_size[4] = {0, 0, 0, 0}; bool have[4] = {false, false, false, false};
        for (uint32_t k = 0, nsec = u32(8); k < nsec; k++) {
            const uint32_t typ = u32(off); const uint64_t size = u64(off + 4);
            off += 12; need(off, size);
            if (typ < 4) { sec_off[typ] = off; sec_size[typ] = size; have[typ] = true; }
            off += size;
        }
        if (!have[1] || !have[2]) fail("missing header or constraint section");
        R1CSFile f;
        const size_t h = sec_off[1];
        if (u32(h) != 32 || sec_size[1] != 64) fail("This parser only supports 32-byte fields");
        need(h + 4, 60);
        std::memcpy(f.prime.data(), d + h + 4, 32);
        f.n_wires = u32(h + 36); f.n_pub_out = u32(h + 40); f.n_pub_in = u32(h + 44); f.n_prv_in = u32(h + 48); f.n_labels = u64(h + 52); f.n_constraints = u32(h + 60);
        size_t p = sec_off[2];
        Csr *m[3] = {&f.a, &f.b, &f.c};
        for (uint32_t r = 0; r < f.n_constraints; r++)
            for (int k = 0; k < 3; k++) {
                const uint32_t nt = u32(p); p += 4;
                need(p, (size_t)nt * 36);
                for (uint32_t t = 0; t < nt; t++, p += 36) { m[k]->cols.push_back(u32(p)); uint64_t w[4]; std::memcpy(w, d + p + 4, 32); m[k]->vals.insert(m[k]->vals.end(), w, w + 4); }
                m[k]->rowptr.push_back(m[k]->cols.size());
            }
        if (!have[3]) fail("No section offset for wire2label type found");           // (the reference looks the section up unconditionally, :91-96)
        if (sec_size[3] != (size_t)f.n_wires * 8) fail("Invalid map section size");
        f.wire_mapping.resize(f.n_wires);
        if (f.n_wires) std::memcpy(f.wire_mapping.data(), d + sec_off[3], (size_t)f.n_wires * 8);
        if (f.n_wires && f.wire_mapping[0] != 0) fail("Wire 0 should always be mapped to 0");
        return f;
    }
    bool is_bls12_381() const { return prime == detail::FR_MODULUS; }                 // r1cs_reader.rs:186-200: anything else is IncompatibleWithCurve
    // the circuit resident in HBM (handle for dgpu_witness_map_r1cs / legogroth16::create_proof_with_reduction; dgpu_r1cs_free releases it)
    uint64_t upload() const {
        if (!is_bls12_381()) throw Error(DGPU_E_BADARG, "r1cs: the file was compiled for another field");
        uint64_t h = 0;
        check(dgpu_r1cs_upload(a.rowptr.data(), a.cols.data(), a.vals.data(), a.cols.size(), b.rowptr.data(), b.cols.data(), b.vals.data(), b.cols.size(),
                               c.rowptr.data(), c.cols.data(), c.vals.data(), c.cols.size(), n_wires, num_inputs(), n_constraints, 0, &h), "r1cs_upload");
        return h;
    }
};
}  // namespace circom

// ---- legogroth16::create_proof_with_reduction (legogroth16/src/prover.rs:153-180 -> :267-383) over dgpu_legogroth16_prove ----
namespace legogroth16 {
// ProvingKey (legogroth16/src/data_structures.rs:55-70,151-168): the five queries live on the device, the O(1) elements on the host
struct ProvingKey {
    DeviceBases<G1> a_query, b_g1_query, h_query, l_query; DeviceBases<G2> b_g2_query;
    G1::Affine alpha_g1, beta_g1, delta_g1, eta_delta_inv_g1, eta_gamma_inv_g1, a0, b1_0; G2::Affine beta_g2, delta_g2, b2_0;
    std::vector<G1::Affine> gamma_abc_g1; size_t commit_witness_count;
    ProvingKey(const std::vector<G1::Affine> &a, const std::vector<G1::Affine> &b1, const std::vector<G2::Affine> &b2, const std::vector<G1::Affine> &h, const std::vector<G1::Affine> &l)
        : a_query(a), b_g1_query(b1), h_query(h), l_query(l), b_g2_query(b2), a0(a.at(0)), b1_0(b1.at(0)), b2_0(b2.at(0)), commit_witness_count(0) {}
};
struct Proof { G1::Affine a, c, d; G2::Affine b; };
// the resident circuit (ConstraintMatrices of the synthesised system as CSR) is a dgpu_r1cs_upload handle; z = (1, instance..., witness...)
inline Proof create_proof_with_reduction(const ProvingKey &pk, uint64_t r1cs, const std::vector<BigInt256> &z, size_t n_inst,
                                         const BigInt256 &r, const BigInt256 &s, const BigInt256 &v) {
    auto flat = [](const G1::Affine &p, uint64_t *o) { if (p.infinity) std::memset(o, 0, 96); else { std::memcpy(o, &p.x, 48); std::memcpy(o + 6, &p.y, 48); } };
    auto flat2 = [](const G2::Affine &p, uint64_t *o) { if (p.infinity) std::memset(o, 0, 192); else { std::memcpy(o, &p.x, 96); std::memcpy(o + 12, &p.y, 96); } };
    uint64_t g1s[7][12], g2s[3][24];
    const G1::Affine *g1p[7] = {&pk.alpha_g1, &pk.beta_g1, &pk.delta_g1, &pk.eta_delta_inv_g1, &pk.eta_gamma_inv_g1, &pk.a0, &pk.b1_0};
    for (int i = 0; i < 7; i++) flat(*g1p[i], g1s[i]);
    flat2(pk.beta_g2, g2s[0]); flat2(pk.delta_g2, g2s[1]); flat2(pk.b2_0, g2s[2]);
    std::vector<uint64_t> gabc(pk.gamma_abc_g1.size() * 12);
    for (size_t i = 0; i < pk.gamma_abc_g1.size(); i++) flat(pk.gamma_abc_g1[i], &gabc[12 * i]);
    dgpu_lego_pk k{};
    k.a_query = pk.a_query.handle(); k.b_g1_query = pk.b_g1_query.handle(); k.b_g2_query = pk.b_g2_query.handle(); k.h_query = pk.h_query.handle(); k.l_query = pk.l_query.handle();
    k.alpha_g1 = g1s[0]; k.beta_g1 = g1s[1]; k.delta_g1 = g1s[2]; k.eta_delta_inv_g1 = g1s[3]; k.eta_gamma_inv_g1 = g1s[4]; k.a0 = g1s[5]; k.b1_0 = g1s[6];
    k.beta_g2 = g2s[0]; k.delta_g2 = g2s[1]; k.b2_0 = g2s[2];
    k.gamma_abc_g1 = gabc.data(); k.gamma_abc_len = pk.gamma_abc_g1.size(); k.commit_witness_count = pk.commit_witness_count;
    uint64_t a[12], b[24], c[12], d[12]; uint8_t inf[4];
    check(dgpu_legogroth16_prove(&k, r1cs, 0, z.empty() ? nullptr : z[0].data(), z.size(), n_inst, 0, r.data(), s.data(), v.data(), a, b, c, d, inf), "legogroth16_prove");
    Proof pr;
    auto unflat = [](G1::Affine &p, const uint64_t *o, bool i) { p.infinity = i; std::memcpy(&p.x, o, 48); std::memcpy(&p.y, o + 6, 48); };
    unflat(pr.a, a, inf[0]); unflat(pr.c, c, inf[2]); unflat(pr.d, d, inf[3]);
    pr.b.infinity = inf[1]; std::memcpy(&pr.b.x, b, 96); std::memcpy(&pr.b.y, b + 12, 96);
    return pr;
}
// The same prover for a key held the way the reference holds it — ProvingKeyCommon's queries as vectors of Affine structs in HOST memory
// (legogroth16/src/data_structures.rs:151-168), nothing uploaded by the caller: dgpu_legogroth16_prove_host resolves the five views through the library's
// resident-bases cache (a key's second proof makes them resident tables).  h: the coefficients QAP::witness_map returned (create_proof_with_assignment,
// prover.rs:237-265), or nullptr with `r1cs` a resident circuit (create_proof_with_reduction: the witness map runs inside the call).
struct HostProvingKey {
    std::vector<G1::Affine> a_query, b_g1_query, h_query, l_query; std::vector<G2::Affine> b_g2_query;
    G1::Affine alpha_g1, beta_g1, delta_g1, eta_delta_inv_g1, eta_gamma_inv_g1; G2::Affine beta_g2, delta_g2;
    std::vector<G1::Affine> gamma_abc_g1; size_t commit_witness_count = 0;
};
inline Proof create_proof_host(const HostProvingKey &pk, uint64_t r1cs, const std::vector<BigInt256> *h, const std::vector<BigInt256> &instance, const std::vector<BigInt256> &witness,
                               const BigInt256 &r, const BigInt256 &s, const BigInt256 &v) {
    auto flat = [](const G1::Affine &p, uint64_t *o) { if (p.infinity) std::memset(o, 0, 96); else { std::memcpy(o, &p.x, 48); std::memcpy(o + 6, &p.y, 48); } };
    auto flat2 = [](const G2::Affine &p, uint64_t *o) { if (p.infinity) std::memset(o, 0, 192); else { std::memcpy(o, &p.x, 96); std::memcpy(o + 12, &p.y, 96); } };
    uint64_t g1s[7][12], g2s[3][24];
    const G1::Affine *g1p[7] = {&pk.alpha_g1, &pk.beta_g1, &pk.delta_g1, &pk.eta_delta_inv_g1, &pk.eta_gamma_inv_g1, &pk.a_query.at(0), &pk.b_g1_query.at(0)};
    for (int i = 0; i < 7; i++) flat(*g1p[i], g1s[i]);
    flat2(pk.beta_g2, g2s[0]); flat2(pk.delta_g2, g2s[1]); flat2(pk.b_g2_query.at(0), g2s[2]);
    std::vector<uint64_t> gabc(pk.gamma_abc_g1.size() * 12);
    for (size_t i = 0; i < pk.gamma_abc_g1.size(); i++) flat(pk.gamma_abc_g1[i], &gabc[12 * i]);
    auto view1 = [](const std::vector<G1::Affine> &q) { return dgpu_bases_view{q.data(), sizeof(G1::Affine), offsetof(G1::Affine, x), offsetof(G1::Affine, y), offsetof(G1::Affine, infinity), q.size()}; };
    dgpu_lego_pk_host k{};
    k.a_query = view1(pk.a_query); k.b_g1_query = view1(pk.b_g1_query); k.h_query = view1(pk.h_query); k.l_query = view1(pk.l_query);
    k.b_g2_query = dgpu_bases_view{pk.b_g2_query.data(), sizeof(G2::Affine), offsetof(G2::Affine, x), offsetof(G2::Affine, y), offsetof(G2::Affine, infinity), pk.b_g2_query.size()};
    k.alpha_g1 = g1s[0]; k.beta_g1 = g1s[1]; k.delta_g1 = g1s[2]; k.eta_delta_inv_g1 = g1s[3]; k.eta_gamma_inv_g1 = g1s[4]; k.a0 = g1s[5]; k.b1_0 = g1s[6];
    k.beta_g2 = g2s[0]; k.delta_g2 = g2s[1]; k.b2_0 = g2s[2];
    k.gamma_abc_g1 = gabc.data(); k.gamma_abc_len = pk.gamma_abc_g1.size(); k.commit_witness_count = pk.commit_witness_count;
    uint64_t a[12], b[24], c[12], d[12]; uint8_t inf[4];
    check(dgpu_legogroth16_prove_host(&k, r1cs, h && !h->empty() ? (*h)[0].data() : nullptr, h ? h->size() : 0, 0, instance.empty() ? nullptr : instance[0].data(), instance.size(),
                                      witness.empty() ? nullptr : witness[0].data(), witness.size(), 0, r.data(), s.data(), v.data(), a, b, c, d, inf), "legogroth16_prove_host");
    Proof pr;
    auto unflat = [](G1::Affine &p, const uint64_t *o, bool i) { p.infinity = i; std::memcpy(&p.x, o, 48); std::memcpy(&p.y, o + 6, 48); };
    unflat(pr.a, a, inf[0]); unflat(pr.c, c, inf[2]); unflat(pr.d, d, inf[3]);
    pr.b.infinity = inf[1]; std::memcpy(&pr.b.x, b, 96); std::memcpy(&pr.b.y, b + 12, 96);
    return pr;
}
// ---- the verifier (legogroth16/src/verifier.rs) over dgpu_legogroth16_verify / dgpu_legogroth16_verify_batch ----
// VerifyingKey (data_structures.rs:55-70) and PreparedVerifyingKey (:112-120): e(alpha, beta) and the two negated, prepared G2 members
struct VerifyingKey { G1::Affine alpha_g1; G2::Affine beta_g2, gamma_g2, delta_g2; std::vector<G1::Affine> gamma_abc_g1; size_t commit_witness_count = 0; };
struct PreparedVerifyingKey { VerifyingKey vk; Fq12 alpha_g1_beta_g2{}; std::vector<uint64_t> gamma_g2_neg_pc, delta_g2_neg_pc, gamma_abc_words; };
namespace detail {
inline constexpr Fq FQ_MODULUS = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
inline Fq fq_neg(const Fq &a) {               // p - a (Montgomery form is closed under it); 0 stays 0
    Fq z{}; if (a == z) return a;
    Fq r{}; unsigned __int128 br = 0;
    for (int i = 0; i < 6; i++) { const unsigned __int128 d = (unsigned __int128)FQ_MODULUS[i] - a[i] - (uint64_t)br; r[i] = (uint64_t)d; br = (d >> 64) & 1; }
    return r;
}
inline G2::Affine g2_neg(G2::Affine q) { if (!q.infinity) { q.y[0] = fq_neg(q.y[0]); q.y[1] = fq_neg(q.y[1]); } return q; }
inline void g1_words(const G1::Affine &p, uint64_t *o) { if (p.infinity) std::memset(o, 0, 96); else { std::memcpy(o, &p.x, 48); std::memcpy(o + 6, &p.y, 48); } }
inline void g2_words(const G2::Affine &p, uint64_t *o) { if (p.infinity) std::memset(o, 0, 192); else { std::memcpy(o, &p.x, 96); std::memcpy(o + 12, &p.y, 96); } }
}  // namespace detail
// prepare_verifying_key (verifier.rs:18-25): one pairing, two G2 preparations (one dgpu_g2_prepare call for both)
inline PreparedVerifyingKey prepare_verifying_key(const VerifyingKey &vk) {
    PreparedVerifyingKey p; p.vk = vk;
    p.alpha_g1_beta_g2 = multi_pairing({vk.alpha_g1}, {vk.beta_g2});
    uint64_t q[48]; uint8_t qi[2] = {vk.gamma_g2.infinity, vk.delta_g2.infinity}, oi[2];
    detail::g2_words(detail::g2_neg(vk.gamma_g2), q); detail::g2_words(detail::g2_neg(vk.delta_g2), q + 24);
    std::vector<uint64_t> pc(2 * DGPU_G2_PREPARED_WORDS);
    check(dgpu_g2_prepare(q, qi, 2, pc.data(), oi), "g2_prepare");
    p.gamma_g2_neg_pc.assign(pc.begin(), pc.begin() + DGPU_G2_PREPARED_WORDS); p.delta_g2_neg_pc.assign(pc.begin() + DGPU_G2_PREPARED_WORDS, pc.end());
    p.gamma_abc_words.resize(vk.gamma_abc_g1.size() * 12);
    for (size_t i = 0; i < vk.gamma_abc_g1.size(); i++) detail::g1_words(vk.gamma_abc_g1[i], &p.gamma_abc_words[12 * i]);
    return p;
}
// verify_proof (verifier.rs:87-99): Ok(()) -> true, Err(InvalidProof) -> false; MalformedVerifyingKey / UnexpectedIdentity throw (DGPU_E_BADARG / DGPU_E_ZERO)
inline bool verify_proof(const PreparedVerifyingKey &pvk, const Proof &proof, const std::vector<BigInt256> &public_inputs) {
    uint64_t a[12], b[24], c[12], d[12]; const uint8_t inf[4] = {proof.a.infinity, proof.b.infinity, proof.c.infinity, proof.d.infinity};
    detail::g1_words(proof.a, a); detail::g2_words(proof.b, b); detail::g1_words(proof.c, c); detail::g1_words(proof.d, d);
    int32_t ok = 0;
    check(dgpu_legogroth16_verify(pvk.alpha_g1_beta_g2.data(), pvk.delta_g2_neg_pc.data(), pvk.gamma_g2_neg_pc.data(), pvk.gamma_abc_words.data(), pvk.vk.gamma_abc_g1.size(),
                                  a, b, c, d, inf, public_inputs.empty() ? nullptr : public_inputs[0].data(), public_inputs.size(), 0, &ok), "legogroth16_verify");
    return ok != 0;
}
// n proofs under one key in one call: what a verifier holding many statements reaches through one lazy RandomizedPairingChecker
// (utils/src/randomized_pairing_check.rs:116-138,204-214), with the pairs that share -delta / -gamma merged before the pairing.  `random`: drawn
// after the proofs are fixed, non-zero mod r.  Every row of public_inputs has the same length.
inline bool verify_proofs_batch(const PreparedVerifyingKey &pvk, const std::vector<Proof> &proofs, const std::vector<std::vector<BigInt256>> &public_inputs, const BigInt256 &random) {
    const size_t n = proofs.size();
    if (public_inputs.size() != n) throw Error(DGPU_E_LENGTH, "verify_proofs_batch");
    const size_t np = n ? public_inputs[0].size() : 0;
    std::vector<uint64_t> a(12 * n), b(24 * n), c(12 * n), d(12 * n), x(4 * n * np);
    for (size_t i = 0; i < n; i++) {
        if (public_inputs[i].size() != np) throw Error(DGPU_E_LENGTH, "verify_proofs_batch: rows of public inputs differ in length");
        detail::g1_words(proofs[i].a, &a[12 * i]); detail::g2_words(proofs[i].b, &b[24 * i]); detail::g1_words(proofs[i].c, &c[12 * i]); detail::g1_words(proofs[i].d, &d[12 * i]);
        for (size_t j = 0; j < np; j++) std::memcpy(&x[4 * (i * np + j)], public_inputs[i][j].data(), 32);
    }
    int32_t ok = 0;
    check(dgpu_legogroth16_verify_batch(pvk.alpha_g1_beta_g2.data(), pvk.delta_g2_neg_pc.data(), pvk.gamma_g2_neg_pc.data(), pvk.gamma_abc_words.data(), pvk.vk.gamma_abc_g1.size(),
                                        a.data(), b.data(), c.data(), d.data(), n, x.data(), np, 0, random.data(), &ok), "legogroth16_verify_batch");
    return ok != 0;
}
}  // namespace legogroth16

// ---- SnarkPack aggregation (legogroth16/src/aggregation/) over dgpu_snarkpack_aggregate / dgpu_snarkpack_verify ---------------------------------
// aggregate_proofs (groth16/prover.rs:47-147; legogroth16/prover.rs:38-127 when the commitments d are given) and verify_aggregate_proof
// (groth16/verifier.rs:36-100, legogroth16/verifier.rs:34-96, legogroth16/using_groth16.rs:45-128).  The transcript is the caller's, as in the
// reference (`&mut impl Transcript`, utils/src/transcript.rs:45-63): any type with
//     void append_message(const uint8_t *label, size_t label_len, const uint8_t *bytes, size_t len);
//     BigInt256 challenge_scalar(const uint8_t *label, size_t label_len);
// Points travel as affine ABI words (12 per G1 point, 24 per G2 point, identity all zero), the aggregate proof as the flat words of
// include/dock_gpu.h; the protocol itself runs inside the library (crypto_amd/csrc/dock_aggregation.cpp).
namespace aggregation {
using Words = std::vector<uint64_t>;
template <class T> dgpu_transcript bind(T &t) {
    dgpu_transcript d;
    d.ctx = &t;
    d.append_message = [](void *c, const uint8_t *l, size_t ll, const uint8_t *b, size_t n) { static_cast<T *>(c)->append_message(l, ll, b, n); };
    d.challenge_scalar = [](void *c, const uint8_t *l, size_t ll, uint64_t out[4]) { const BigInt256 v = static_cast<T *>(c)->challenge_scalar(l, ll); std::memcpy(out, v.data(), 32); };
    return d;
}
struct ProverSRS {                          // srs.rs:60-93; specialize (:180-237) gives vkey = (h^{alpha^i}, h^{beta^i}), wkey = (g^{alpha^{n+i}}, g^{beta^{n+i}}), i < n
    size_t n = 0;
    Words g_alpha_powers_table, g_beta_powers_table, h_alpha_powers_table, h_beta_powers_table, vkey_a, vkey_b, wkey_a, wkey_b;
    // from the four power vectors of a GenericSRS (2n entries each at least)
    static ProverSRS specialize(size_t n, const Words &g_alpha, const Words &h_alpha, const Words &g_beta, const Words &h_beta) {
        if (g_alpha.size() < 24 * n || g_beta.size() < 24 * n || h_alpha.size() < 48 * n || h_beta.size() < 48 * n) throw Error(DGPU_E_LENGTH, "specialize");
        ProverSRS s; s.n = n;
        s.g_alpha_powers_table.assign(g_alpha.begin(), g_alpha.begin() + 24 * n); s.g_beta_powers_table.assign(g_beta.begin(), g_beta.begin() + 24 * n);
        s.h_alpha_powers_table.assign(h_alpha.begin(), h_alpha.begin() + 24 * n); s.h_beta_powers_table.assign(h_beta.begin(), h_beta.begin() + 24 * n);
        s.vkey_a = s.h_alpha_powers_table; s.vkey_b = s.h_beta_powers_table;
        s.wkey_a.assign(g_alpha.begin() + 12 * n, g_alpha.begin() + 24 * n); s.wkey_b.assign(g_beta.begin() + 12 * n, g_beta.begin() + 24 * n);
        return s;
    }
    dgpu_snarkpack_prover_srs view() const {
        return {n, g_alpha_powers_table.data(), g_beta_powers_table.data(), h_alpha_powers_table.data(), h_beta_powers_table.data(), vkey_a.data(), vkey_b.data(), wkey_a.data(), wkey_b.data()};
    }
};
struct VerifierSRS {                        // srs.rs:95-110
    size_t n = 0; std::array<uint64_t, 12> g{}, g_alpha{}, g_beta{}; std::array<uint64_t, 24> h{}, h_alpha{}, h_beta{};
    static VerifierSRS specialize(size_t n, const Words &g_alpha, const Words &h_alpha, const Words &g_beta, const Words &h_beta) {
        VerifierSRS v; v.n = n;
        std::memcpy(v.g.data(), &g_alpha[0], 96); std::memcpy(v.g_alpha.data(), &g_alpha[12], 96); std::memcpy(v.g_beta.data(), &g_beta[12], 96);
        std::memcpy(v.h.data(), &h_alpha[0], 192); std::memcpy(v.h_alpha.data(), &h_alpha[24], 192); std::memcpy(v.h_beta.data(), &h_beta[24], 192);
        return v;
    }
};
struct VerifyingKey { std::array<uint64_t, 12> alpha_g1{}; std::array<uint64_t, 24> beta_g2{}, gamma_g2{}, delta_g2{}; Words gamma_abc_g1; };
enum class Variant : int32_t { Groth16 = 0, LegoGroth16 = 1, LegoGroth16UsingGroth16 = 2 };

// a, c (, d): n x 12 words, b: n x 24; d == nullptr: Groth16 proofs.
// The library absorbs into / squeezes from the transcript as the protocol goes and may fail half-way (an allocation, a HIP error): it runs on a
// COPY of `transcript` (T must be copyable, like merlin::Transcript is Clone), which replaces the caller's only when the call succeeded — after a
// throw the caller's transcript is what it was, so a CPU fallback produces a proof every verifier can reproduce.
template <class T> Words aggregate_proofs(const ProverSRS &srs, T &transcript, const Words &a, const Words &b, const Words &c, const Words *d = nullptr) {
    const size_t n = a.size() / 12;
    if (b.size() != 24 * n || c.size() != 12 * n || (d && d->size() != 12 * n)) throw Error(DGPU_E_LENGTH, "aggregate_proofs");
    if (srs.n != n || srs.g_alpha_powers_table.size() != 24 * n || srs.g_beta_powers_table.size() != 24 * n || srs.h_alpha_powers_table.size() != 24 * n || srs.h_beta_powers_table.size() != 24 * n ||
        srs.vkey_a.size() != 24 * n || srs.vkey_b.size() != 24 * n || srs.wkey_a.size() != 12 * n || srs.wkey_b.size() != 12 * n) throw Error(DGPU_E_LENGTH, "aggregate_proofs: the SRS is not specialised to this many proofs");
    const size_t cap = dgpu_snarkpack_proof_words(n, d ? 1 : 0);
    if (!cap) throw Error(DGPU_E_BADARG, "aggregate_proofs: the number of proofs is not a power of two >= 2");
    Words out(cap); size_t len = 0;
    const dgpu_snarkpack_prover_srs v = srs.view();
    T work = transcript;
    const dgpu_transcript t = bind(work);
    check(dgpu_snarkpack_aggregate(&v, a.data(), b.data(), c.data(), d ? d->data() : nullptr, n, &t, out.data(), cap, &len), "snarkpack_aggregate");
    transcript = std::move(work);
    out.resize(len);
    return out;
}
// public_inputs: one row of `inputs_per_proof` scalars per proof.  true: the aggregate verifies; throws Error(DGPU_E_BADARG) for a malformed
// proof / key (AggregationError::InvalidProof / MalformedVerifyingKey before any group operation) and for random == 0 mod r.
// validate: the words came from an untrusted source — every G1 / G2 member must be on its curve and in the prime-order subgroup, every GT member of
// order r (what CanonicalDeserialize with Validate::Yes checks before the reference ever sees an AggregateProof); leave it on unless the words were
// produced by aggregate_proofs in this process.  Runs on a copy of the transcript like aggregate_proofs.
template <class T> bool verify_aggregate_proof(const VerifierSRS &srs, const VerifyingKey &vk, const std::vector<BigInt256> &public_inputs, size_t inputs_per_proof,
                                               const Words &proof, const BigInt256 &random, T &transcript, Variant variant = Variant::Groth16,
                                               const Words *d_list = nullptr, bool validate = true) {
    if (inputs_per_proof && public_inputs.size() % inputs_per_proof) throw Error(DGPU_E_LENGTH, "verify_aggregate_proof");
    const size_t rows = inputs_per_proof ? public_inputs.size() / inputs_per_proof : (proof.empty() ? 0 : (size_t)proof[0]);
    // the library reads 12 * rows words of d_list in the LegoGroth16-under-Groth16 variant and none otherwise
    if ((variant == Variant::LegoGroth16UsingGroth16) != (d_list != nullptr)) throw Error(DGPU_E_BADARG, "verify_aggregate_proof: d_list goes with Variant::LegoGroth16UsingGroth16 and only with it");
    if (d_list && d_list->size() != 12 * rows) throw Error(DGPU_E_LENGTH, "verify_aggregate_proof: d_list must hold one commitment per proof");
    const dgpu_snarkpack_verifier_srs s{srs.n, srs.g.data(), srs.h.data(), srs.g_alpha.data(), srs.g_beta.data(), srs.h_alpha.data(), srs.h_beta.data()};
    const dgpu_groth16_vk k{vk.alpha_g1.data(), vk.beta_g2.data(), vk.gamma_g2.data(), vk.delta_g2.data(), vk.gamma_abc_g1.data(), vk.gamma_abc_g1.size() / 12};
    T work = transcript;
    const dgpu_transcript t = bind(work);
    int32_t ok = 0;
    check(dgpu_snarkpack_verify(&s, &k, public_inputs.empty() ? nullptr : public_inputs[0].data(), rows, inputs_per_proof, proof.data(), proof.size(), (int32_t)variant,
                                d_list ? d_list->data() : nullptr, random.data(), &t, validate ? (DGPU_SNARKPACK_VALIDATE_GT | DGPU_SNARKPACK_VALIDATE_POINTS) : 0, &ok), "snarkpack_verify");
    transcript = std::move(work);
    return ok != 0;
}
}  // namespace aggregation

}  // namespace dock_gpu
