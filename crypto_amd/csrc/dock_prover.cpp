// crypto_amd/csrc/dock_prover.cpp — the LegoGroth16 prover as ONE entry point of the C ABI (include/dock_gpu.h: dgpu_legogroth16_prove).
//
// Replaces legogroth16/src/prover.rs:267-383 `create_proof_and_committed_witnesses_with_assignment` (+ `calculate_coeff` :585-594 and, with a
// resident circuit, the `witness_map` call of `create_proof_with_reduction` :153-180) at the granularity the reference calls it: a Rust shim
// hands over the proving key's handles, the assignment and (r, s, v) and gets (A, B, C, D) back.  The schedule that used to live in Python
// above the ABI is host threads inside the library, like dgpu_multi_pairing_segments:
//
//   z uploaded ONCE (the witness map reads it in place; the four assignment MSMs use the same handle at scalar offset 1; h never leaves HBM)
//   job H   witness map -> h -> MSM over h_query (D - 1 points against D scalars: the truncation of prover.rs:286)
//   one partition sort for A, B-in-G1, B-in-G2 (and l, through row_shift) when their queries are tables of one shape (dgpu_scalars_sort)
//   job B2  the G2 MSM first: it is the longest call and ends in latency-bound kernels that then run under the G1 MSMs
//   jobs A, B1, L; job K: the O(1) group arithmetic that depends on no MSM (r delta + a_0 + alpha, ..., g_d) on a host core meanwhile
//   s g_a + r g1_b as soon as A and B1 are back, while B2 is still in flight; the final fold
//
// Everything device-side goes through the library's own entry points (each call takes one of the context's slots), so this file is plain
// host C++: no kernel, no HIP call.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include "../../include/dock_gpu.h"
#include "host_field.hpp"
#include "host_par.hpp"

namespace dock {
extern thread_local bool tl_no_min;          // dock_core.hip: the size threshold (DGPU_E_TOO_SMALL) is for callers, not for the library's own calls
int32_t msm_g1_nothreshold(const uint64_t *bases, const uint8_t *is_inf, const uint64_t *scalars, size_t n, uint64_t out[18]);   // dock_g1.hip
}

namespace {
using hostf::FrH;

// A host thread of the call.  Nothing may leave the C ABI by unwinding: the body maps std::bad_alloc / anything else to an error code, a thread
// that cannot be created (std::system_error) runs its body on the calling thread instead, and the destructor joins so that an early return
// never destroys a joinable std::thread (std::terminate).
struct Job {
    std::thread th; int32_t rc = DGPU_OK;
    template <class F> static int32_t guarded(F &f) noexcept {
        try { return f(); } catch (const std::bad_alloc &) { return DGPU_E_OOM; } catch (...) { return DGPU_E_HIP; }
    }
    template <class F> void start(F f) {
        try { th = std::thread([this, f]() mutable { dock::tl_no_min = true; rc = guarded(f); }); }
        catch (...) { const bool keep = dock::tl_no_min; dock::tl_no_min = true; rc = guarded(f); dock::tl_no_min = keep; }
    }
    int32_t join() { if (th.joinable()) th.join(); return rc; }
    Job() = default;
    Job(const Job &) = delete; Job &operator=(const Job &) = delete;
    ~Job() { if (th.joinable()) th.join(); }
};

inline bool is_zero4(const uint64_t a[4]) { return !(a[0] | a[1] | a[2] | a[3]); }
inline void fr_neg(uint64_t out[4], const uint64_t a[4]) {          // -a mod r, a canonical
    if (is_zero4(a)) { memset(out, 0, 32); return; }
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) { hostf::u128 d = (hostf::u128)FrH::MOD[i] - a[i] - br; out[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
inline void fr_mul(uint64_t out[4], const uint64_t a[4], const uint64_t b[4]) {      // canonical in, canonical out
    FrH x, y, r2; memcpy(x.l, a, 32); memcpy(y.l, b, 32); memcpy(r2.l, FrH::R2, 32);
    FrH p = FrH::mont_mul(FrH::mont_mul(x, r2), y);         // (a R)(b) / R = a b
    memcpy(out, p.l, 32);
}
// any value below 2^256 -> its residue (the entry point accepts what `into_bigint` yields, but r, s, v are caller data)
inline void fr_reduce(uint64_t out[4], const uint64_t a[4]) { const uint64_t one[4] = {1, 0, 0, 0}; fr_mul(out, a, one); }

const uint64_t ONE4[4] = {1, 0, 0, 0};

// sum of k normalised Jacobian G1 triples / affine out (identity: zero words + flag)
int32_t fold_to_affine_g1(const uint64_t *parts, size_t k, uint64_t out[12], uint8_t *inf) {
    uint64_t j[18];
    int32_t rc = dgpu_fold_g1(parts, k, j);
    if (rc) return rc;
    bool z = true; for (int i = 12; i < 18; i++) z = z && j[i] == 0;
    *inf = z; if (z) memset(out, 0, 96); else memcpy(out, j, 96);
    return DGPU_OK;
}

// z = the full assignment, given as consecutive host arrays (one for dgpu_legogroth16_prove; instance and witness for _prove_host)
struct ProofInputs {
    const dgpu_lego_pk *pk; const uint64_t *const *z_parts; const size_t *z_counts; size_t z_nparts; size_t n_inst; int32_t montgomery; uint64_t r[4], s[4], v[4]; bool with_b1;
    const uint64_t *zat(size_t i) const { for (size_t k = 0; k < z_nparts; k++) { if (i < z_counts[k]) return z_parts[k] + 4 * i; i -= z_counts[k]; } return nullptr; }
    const uint64_t *z() const { return z_parts[0]; }          // (the sharded form: one part)
};
struct ProofConsts { uint64_t rest_a[18], rest_b1[18], rest_c[18], rest_b2[36], g_d[18]; };
// what depends on no MSM (host arithmetic; the tiny g_d MSM goes to the device only when it has more than 15 terms)
int32_t proof_constants(const ProofInputs &in, ProofConsts &c) {
    const dgpu_lego_pk *pk = in.pk; const size_t cw = pk->commit_witness_count;
    int32_t e;
    { uint64_t p[36]; memcpy(p, pk->delta_g1, 96); memcpy(p + 12, pk->a0, 96); memcpy(p + 24, pk->alpha_g1, 96);
      uint64_t k[12]; memcpy(k, in.r, 32); memcpy(k + 4, ONE4, 32); memcpy(k + 8, ONE4, 32);
      if ((e = dgpu_lincomb_g1(p, nullptr, k, 3, c.rest_a))) return e; }                                  // r delta + a_0 + alpha  (:585-594)
    { uint64_t p[72]; memcpy(p, pk->delta_g2, 192); memcpy(p + 24, pk->b2_0, 192); memcpy(p + 48, pk->beta_g2, 192);
      uint64_t k[12]; memcpy(k, in.s, 32); memcpy(k + 4, ONE4, 32); memcpy(k + 8, ONE4, 32);
      if ((e = dgpu_lincomb_g2(p, nullptr, k, 3, c.rest_b2))) return e; }
    if (in.with_b1) { uint64_t p[36]; memcpy(p, pk->delta_g1, 96); memcpy(p + 12, pk->b1_0, 96); memcpy(p + 24, pk->beta_g1, 96);
      uint64_t k[12]; memcpy(k, in.s, 32); memcpy(k + 4, ONE4, 32); memcpy(k + 8, ONE4, 32);
      if ((e = dgpu_lincomb_g1(p, nullptr, k, 3, c.rest_b1))) return e; }
    { uint64_t p[24]; memcpy(p, pk->delta_g1, 96); memcpy(p + 12, pk->eta_delta_inv_g1, 96);
      uint64_t k[8], rs[4]; fr_mul(rs, in.r, in.s); fr_neg(k, rs); fr_neg(k + 4, in.v);
      if ((e = dgpu_lincomb_g1(p, nullptr, k, 2, c.rest_c))) return e; }                                  // -rs delta - v eta/delta  (:350-355)
    // g_d = msm(gamma_abc[n_inst .. n_inst + cw], committed witnesses) + v eta/gamma   (:361-368)
    std::vector<uint64_t> pts((cw + 1) * 12), sc((cw + 1) * 4);
    if (cw) memcpy(pts.data(), pk->gamma_abc_g1 + in.n_inst * 12, cw * 96);
    memcpy(pts.data() + cw * 12, pk->eta_gamma_inv_g1, 96);
    for (size_t i = 0; i < cw; i++) { const uint64_t *zi = in.zat(in.n_inst + i); if (in.montgomery) hostf::fr_from_mont(&sc[4 * i], zi); else memcpy(&sc[4 * i], zi, 32); }
    memcpy(&sc[4 * cw], in.v, 32);
    if (cw + 1 <= DGPU_MAX_LINCOMB) return dgpu_lincomb_g1(pts.data(), nullptr, sc.data(), cw + 1, c.g_d);
    return dock::msm_g1_nothreshold(pts.data(), nullptr, sc.data(), cw + 1, c.g_d);
}
// g_a and g1_b from their MSM parts, and s g_a + r g1_b (computed while the G2 MSM is still in flight)
struct ProofAB { uint64_t ga[36]; uint64_t sa_rb[18]; uint8_t inf_a = 1, inf_b1 = 1; };
int32_t proof_ab(const ProofInputs &in, const ProofConsts &c, const uint64_t acc_a[18], const uint64_t acc_b1[18], ProofAB &o) {
    memset(o.ga, 0, sizeof o.ga);
    int32_t e;
    uint64_t parts[36]; memcpy(parts, acc_a, 144); memcpy(parts + 18, c.rest_a, 144);
    if ((e = fold_to_affine_g1(parts, 2, o.ga, &o.inf_a))) return e;
    if (in.with_b1) { memcpy(parts, acc_b1, 144); memcpy(parts + 18, c.rest_b1, 144); if ((e = fold_to_affine_g1(parts, 2, o.ga + 12, &o.inf_b1))) return e; }
    const uint8_t inf_two[2] = {o.inf_a, o.inf_b1};
    uint64_t k[8]; memcpy(k, in.s, 32); memcpy(k + 4, in.r, 32);
    return dgpu_lincomb_g1(o.ga, inf_two, k, 2, o.sa_rb);
}
int32_t proof_finish(const ProofConsts &c, const ProofAB &ab, const uint64_t acc_b2[36], const uint64_t acc_l[18], const uint64_t acc_h[18],
                     uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12], uint64_t out_d[12], uint8_t out_inf[4]) {
    int32_t rc;
    {
        uint64_t parts[72], jb[36];
        memcpy(parts, acc_b2, 288); memcpy(parts + 36, c.rest_b2, 288);
        if ((rc = dgpu_fold_g2(parts, 2, jb))) return rc;
        bool zb = true; for (int i = 24; i < 36; i++) zb = zb && jb[i] == 0;
        out_inf[1] = zb; if (zb) memset(out_b, 0, 192); else memcpy(out_b, jb, 192);
    }
    out_inf[0] = ab.inf_a; memcpy(out_a, ab.ga, 96);
    {
        uint64_t parts[72]; memcpy(parts, ab.sa_rb, 144); memcpy(parts + 18, c.rest_c, 144); memcpy(parts + 36, acc_l, 144); memcpy(parts + 54, acc_h, 144);
        if ((rc = fold_to_affine_g1(parts, 4, out_c, &out_inf[2]))) return rc;            // g_c = s g_a + r g1_b - rs delta + l_aux + h_acc - v eta/delta
    }
    return fold_to_affine_g1(c.g_d, 1, out_d, &out_inf[3]);
}
int32_t prove_sharded(const ProofInputs &in, size_t G, uint64_t r1cs, uint64_t h_scalars, size_t num_vars,
                      uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12], uint64_t out_d[12], uint8_t out_inf[4]);
}  // namespace



// z: the assignment as consecutive host arrays; h_host != nullptr: the D coefficients of h in host memory (uploaded beside z), else r1cs / h_scalars
static int32_t prove_impl(const dgpu_lego_pk *pk, uint64_t r1cs, uint64_t h_scalars, const uint64_t *h_host, size_t h_len, int32_t h_montgomery,
                          const uint64_t *const *z_parts, const size_t *z_counts, size_t z_nparts, size_t n_inst,
                          int32_t montgomery, const uint64_t r_in[4], const uint64_t s_in[4], const uint64_t v_in[4],
                          uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12], uint64_t out_d[12], uint8_t out_inf[4]);
extern "C" int32_t dgpu_legogroth16_prove(const dgpu_lego_pk *pk, uint64_t r1cs, uint64_t h_scalars, const uint64_t *z, size_t num_vars, size_t n_inst,
                                          int32_t montgomery, const uint64_t r_in[4], const uint64_t s_in[4], const uint64_t v_in[4],
                                          uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12], uint64_t out_d[12], uint8_t out_inf[4]) {
    try { return prove_impl(pk, r1cs, h_scalars, nullptr, 0, 0, &z, &num_vars, 1, n_inst, montgomery, r_in, s_in, v_in, out_a, out_b, out_c, out_d, out_inf); }
    catch (const std::bad_alloc &) { return DGPU_E_OOM; }
    catch (...) { return DGPU_E_HIP; }
}
static int32_t prove_impl(const dgpu_lego_pk *pk, uint64_t r1cs, uint64_t h_scalars, const uint64_t *h_host, size_t h_len, int32_t h_montgomery,
                          const uint64_t *const *z_parts, const size_t *z_counts, size_t z_nparts, size_t n_inst,
                          int32_t montgomery, const uint64_t r_in[4], const uint64_t s_in[4], const uint64_t v_in[4],
                          uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12], uint64_t out_d[12], uint8_t out_inf[4]) {
    if (!pk || !z_parts || !z_counts || z_nparts == 0 || !r_in || !s_in || !v_in || !out_a || !out_b || !out_c || !out_d || !out_inf) return DGPU_E_BADARG;
    size_t num_vars = 0;
    for (size_t k = 0; k < z_nparts; k++) { if (z_counts[k] && !z_parts[k]) return DGPU_E_BADARG; num_vars += z_counts[k]; }
    if ((r1cs != 0) + (h_scalars != 0) + (h_host != nullptr) != 1) return DGPU_E_BADARG;       // exactly one source of h
    const size_t cw = pk->commit_witness_count;
    if (n_inst == 0 || n_inst + cw > num_vars || pk->gamma_abc_len < n_inst + cw) return DGPU_E_BADARG;
    if (!pk->alpha_g1 || !pk->beta_g1 || !pk->delta_g1 || !pk->eta_delta_inv_g1 || !pk->eta_gamma_inv_g1 || !pk->beta_g2 || !pk->delta_g2 ||
        !pk->a0 || !pk->b1_0 || !pk->b2_0 || (cw && !pk->gamma_abc_g1)) return DGPU_E_BADARG;
    if (r1cs) {                                                                      // the circuit's own shape decides D, the gamma_abc / l offsets: refuse a caller that disagrees
        size_t cv = 0, ci = 0;
        if (dgpu_r1cs_shape(r1cs, &cv, &ci, nullptr) != DGPU_OK || cv != num_vars || ci != n_inst) return DGPU_E_BADARG;
    }
    ProofInputs in{pk, z_parts, z_counts, z_nparts, n_inst, montgomery, {0}, {0}, {0}, false};
    fr_reduce(in.r, r_in); fr_reduce(in.s, s_in); fr_reduce(in.v, v_in);
    in.with_b1 = !is_zero4(in.r);                                                    // prover.rs:330-336
    { int32_t shards = 0;                                                           // a key resident across several device contexts (dgpu_bases_upload_*_sharded)
      if (dgpu_shard_count(pk->a_query, &shards) == DGPU_OK && shards > 0) return (z_nparts != 1 || h_host) ? (int32_t)DGPU_E_BADARG : prove_sharded(in, (size_t)shards, r1cs, h_scalars, num_vars, out_a, out_b, out_c, out_d, out_inf); }
    size_t n_a = 0, n_b1 = 0, n_b2 = 0, n_h = 0, n_l = 0;
    if (dgpu_handle_len(pk->a_query, &n_a) || dgpu_handle_len(pk->b_g1_query, &n_b1) || dgpu_handle_len(pk->b_g2_query, &n_b2) ||
        dgpu_handle_len(pk->h_query, &n_h) || dgpu_handle_len(pk->l_query, &n_l) || n_a == 0 || n_b1 == 0 || n_b2 == 0) return DGPU_E_BADARG;
    const bool with_b1 = in.with_b1;

    // ---- z resident, once ----
    uint64_t zh = 0;
    int32_t rc = dgpu_scalars_upload_parts(z_parts, z_counts, z_nparts, montgomery, &zh);
    if (rc) return rc;
    const size_t n_assign = num_vars - 1;                                            // assignment = z[1..]
    const size_t n_aux = num_vars - n_inst - cw, aux_at = n_inst + cw;              // aux = witnesses after the committed ones (prover.rs:292-299)

    // ---- job H: witness map + the h_query MSM ----
    uint64_t h_owned = 0;
    uint64_t acc_h[18], acc_a[18], acc_b1[18], acc_l[18], acc_b2[36];
    Job jH, jB2, jA, jB1, jL, jK;
    jH.start([&]() -> int32_t {
        uint64_t hh = h_scalars; size_t D = 0;
        if (h_host) { int32_t e = dgpu_scalars_upload(h_host, h_len, h_montgomery, &h_owned); if (e) return e; hh = h_owned; D = h_len; }
        else if (r1cs) { int32_t e = dgpu_witness_map_r1cs_resident(r1cs, zh, nullptr, &h_owned, &D); if (e) return e; hh = h_owned; }
        else if (dgpu_handle_len(hh, &D)) return DGPU_E_BADARG;
        return dgpu_msm_g1_resident(pk->h_query, 0, hh, 0, std::min(n_h, D), acc_h);          // :286 (h_query has D - 1 points)
    });

    // ---- one partition sort for the MSMs that multiply z[1..] by tables of one shape ----
    size_t rows_a = 0, rows_b1 = 0, rows_b2 = 0, rows_l = 0; int32_t c_a = 0, w_a = 0, c_x = 0, w_x = 0;
    const bool tab_a = dgpu_bases_table_shape(pk->a_query, &rows_a, &c_a, &w_a) == DGPU_OK;
    bool share = tab_a && n_assign >= 1;
    share = share && dgpu_bases_table_shape(pk->b_g2_query, &rows_b2, &c_x, &w_x) == DGPU_OK && rows_b2 == rows_a && c_x == c_a;
    if (share && with_b1) share = dgpu_bases_table_shape(pk->b_g1_query, &rows_b1, &c_x, &w_x) == DGPU_OK && rows_b1 == rows_a && c_x == c_a;
    const size_t n_coeff = std::min(n_assign, n_a - 1);                             // msm(query[1..], assignment) of calculate_coeff (:592)
    uint64_t sorted = 0;
    if (share && n_coeff > 0) { if (dgpu_scalars_sort(pk->a_query, 1, zh, 1, n_coeff, &sorted) != DGPU_OK) { sorted = 0; share = false; } }
    else share = false;
    const size_t l_shift = n_inst + cw;
    const bool l_shares = share && dgpu_bases_table_shape(pk->l_query, &rows_l, &c_x, &w_x) == DGPU_OK && c_x == c_a && rows_l + l_shift == rows_a &&
                          n_aux == n_l && n_coeff == n_a - 1;
    auto coeff = [&](int g2, uint64_t query, size_t nq, uint64_t *out) -> int32_t {
        if (share) return g2 ? dgpu_msm_g2_sorted(query, sorted, 0, out) : dgpu_msm_g1_sorted(query, sorted, 0, out);
        const size_t n = std::min(n_assign, nq - 1);
        return g2 ? dgpu_msm_g2_resident(query, 1, zh, 1, n, out) : dgpu_msm_g1_resident(query, 1, zh, 1, n, out);
    };
    jB2.start([&] { return coeff(1, pk->b_g2_query, n_b2, acc_b2); });                 // :343-344, first in
    jA.start([&] { return coeff(0, pk->a_query, n_a, acc_a); });                       // :325-326
    if (with_b1) jB1.start([&] { return coeff(0, pk->b_g1_query, n_b1, acc_b1); });
    jL.start([&]() -> int32_t {                                                      // :299
        if (l_shares) return dgpu_msm_g1_sorted(pk->l_query, sorted, l_shift, acc_l);
        return dgpu_msm_g1_resident(pk->l_query, 0, zh, aux_at, std::min(n_l, n_aux), acc_l);
    });

    // ---- job K: what depends on no MSM ----
    ProofConsts cst;
    jK.start([&] { return proof_constants(in, cst); });

    // ---- assemble ----
    int32_t first = DGPU_OK;
    auto note = [&](int32_t e) { if (e && !first) first = e; };
    note(jK.join()); note(jA.join()); if (with_b1) note(jB1.join());
    ProofAB ab;
    if (!first) note(proof_ab(in, cst, acc_a, acc_b1, ab));                          // (while the G2 MSM is still in flight)
    note(jB2.join()); note(jL.join()); note(jH.join());
    if (sorted) (void)dgpu_scalars_free(sorted);
    if (h_owned) (void)dgpu_scalars_free(h_owned);
    (void)dgpu_scalars_free(zh);
    if (first) return first;
    return proof_finish(cst, ab, acc_b2, acc_l, acc_h, out_a, out_b, out_c, out_d, out_inf);
}

// ---- the same proof with the key resident across G device contexts (one process): every query is a sharded handle whose part g lives on
// context g.  Context g multiplies its rows of the five queries by the matching slice of z / h (one host thread per context inside this call, the
// schedule of the single-device form inside it: shared sort, G2 first); the witness map runs once, on the circuit's context, and its h coefficients
// go to the shards device to device (peer copies over xGMI between GPUs); the per-shard partial points (144 / 288 B each) are folded on the host.
namespace {
struct ShardRange { uint64_t sub; size_t lo, hi; int32_t ctx; };
int32_t shard_part(uint64_t handle, size_t g, ShardRange &o) { return dgpu_shard_part(handle, g, &o.sub, &o.lo, &o.hi, &o.ctx); }

int32_t prove_sharded(const ProofInputs &in, size_t G, uint64_t r1cs, uint64_t h_scalars, size_t num_vars,
                      uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12], uint64_t out_d[12], uint8_t out_inf[4]) {
    const dgpu_lego_pk *pk = in.pk; const size_t cw = pk->commit_witness_count, n_inst = in.n_inst;
    const size_t aux_at = n_inst + cw;
    std::vector<ShardRange> A(G), B1(G), B2(G), Hq(G), L(G);
    for (size_t g = 0; g < G; g++) {
        if (shard_part(pk->a_query, g, A[g]) || shard_part(pk->b_g1_query, g, B1[g]) || shard_part(pk->b_g2_query, g, B2[g]) ||
            shard_part(pk->h_query, g, Hq[g]) || shard_part(pk->l_query, g, L[g])) return DGPU_E_BADARG;
        if (A[g].ctx != B1[g].ctx || A[g].ctx != B2[g].ctx || A[g].ctx != Hq[g].ctx || A[g].ctx != L[g].ctx ||
            A[g].lo != B1[g].lo || A[g].hi != B1[g].hi || A[g].lo != B2[g].lo || A[g].hi != B2[g].hi) return DGPU_E_BADARG;     // the five queries split over the same contexts, a / b alike
    }
    // h: from the caller (a resident vector cannot be split here: it must be a host-visible source) or from the witness map on the circuit's context
    if (!r1cs) return DGPU_E_BADARG;                 // the sharded form takes the circuit (h_scalars lives on one device only)
    (void)h_scalars;
    // h = the witness map's output, resident on the circuit's context; every shard receives its slice device to device (dgpu_scalars_copy_range:
    // a peer copy over xGMI between two GPUs).  Rounds 1-3 moved it through the host: D x 32 B down PCIe and 1 / G of it up again per shard.
    size_t D = 0;
    uint64_t h_dev = 0;
    Job jW;
    jW.start([&]() -> int32_t {
        int32_t rctx = 0; if (dgpu_handle_context(r1cs, &rctx)) return DGPU_E_BADARG;
        if (dgpu_set_device(rctx)) return DGPU_E_BADARG;
        return dgpu_witness_map_r1cs(r1cs, in.z(), num_vars, in.montgomery, nullptr, &h_dev, &D);
    });
    ProofConsts cst; Job jK;
    jK.start([&] { return proof_constants(in, cst); });
    // per shard: the A / B1 / B2 / L partial sums now, the H partial sum once h is there
    std::vector<uint64_t> pa(G * 18), pb1(G * 18), pl(G * 18), ph(G * 18), pb2(G * 36);
    std::vector<Job> jobs(G);
    std::vector<int32_t> wm_rc(1, DGPU_OK);
    // (the witness map is awaited by shard threads through this join-once helper)
    std::mutex wm_mu; bool wm_joined = false;
    auto wait_h = [&]() -> int32_t { std::lock_guard<std::mutex> lk(wm_mu); if (!wm_joined) { wm_rc[0] = jW.join(); wm_joined = true; } return wm_rc[0]; };
    for (size_t g = 0; g < G; g++) jobs[g].start([&, g]() -> int32_t {
        if (dgpu_set_device(A[g].ctx)) return DGPU_E_BADARG;
        int32_t e = DGPU_OK;
        auto identity1 = [](uint64_t *o) { uint64_t none[18]; (void)dgpu_fold_g1(none, 0, o); };
        auto identity2 = [](uint64_t *o) { uint64_t none[36]; (void)dgpu_fold_g2(none, 0, o); };
        // rows [lo, hi) of the (V + 1)-row queries; row 0 is query[0] (added by proof_constants): terms are rows max(lo, 1) .. hi, paired with z[row]
        const size_t lo = std::max<size_t>(A[g].lo, 1), hi = std::min(A[g].hi, num_vars);
        uint64_t zs = 0, srt = 0;
        if (hi > lo) {
            const size_t n = hi - lo, boff = lo - A[g].lo;
            if ((e = dgpu_scalars_upload(in.z() + 4 * lo, n, in.montgomery, &zs))) return e;
            size_t ra = 0, rx = 0; int32_t ca = 0, cx = 0, wa = 0, wx = 0;
            bool share = dgpu_bases_table_shape(A[g].sub, &ra, &ca, &wa) == DGPU_OK && dgpu_bases_table_shape(B2[g].sub, &rx, &cx, &wx) == DGPU_OK && rx == ra && cx == ca;
            if (share && in.with_b1) share = dgpu_bases_table_shape(B1[g].sub, &rx, &cx, &wx) == DGPU_OK && rx == ra && cx == ca;
            if (share && dgpu_scalars_sort(A[g].sub, boff, zs, 0, n, &srt) != DGPU_OK) { srt = 0; share = false; }
            Job j2, j1;
            j2.start([&] { return share ? dgpu_msm_g2_sorted(B2[g].sub, srt, 0, &pb2[36 * g]) : dgpu_msm_g2_resident(B2[g].sub, boff, zs, 0, n, &pb2[36 * g]); });      // G2 first
            if (in.with_b1) j1.start([&] { return share ? dgpu_msm_g1_sorted(B1[g].sub, srt, 0, &pb1[18 * g]) : dgpu_msm_g1_resident(B1[g].sub, boff, zs, 0, n, &pb1[18 * g]); });
            else identity1(&pb1[18 * g]);
            e = share ? dgpu_msm_g1_sorted(A[g].sub, srt, 0, &pa[18 * g]) : dgpu_msm_g1_resident(A[g].sub, boff, zs, 0, n, &pa[18 * g]);
            const int32_t e1 = j1.join(), e2 = j2.join();
            if (!e) e = e1; if (!e) e = e2;
            if (srt) (void)dgpu_scalars_free(srt);
            (void)dgpu_scalars_free(zs);
            if (e) return e;
        } else { identity1(&pa[18 * g]); identity1(&pb1[18 * g]); identity2(&pb2[36 * g]); }
        // l_query rows [lo, hi) pair with z[aux_at + row]
        { const size_t llo = L[g].lo, lhi = std::min(L[g].hi, num_vars - std::min(num_vars, aux_at));
          if (lhi > llo) { uint64_t zl = 0; if ((e = dgpu_scalars_upload(in.z() + 4 * (aux_at + llo), lhi - llo, in.montgomery, &zl))) return e;
                           e = dgpu_msm_g1_resident(L[g].sub, 0, zl, 0, lhi - llo, &pl[18 * g]); (void)dgpu_scalars_free(zl); if (e) return e; }
          else identity1(&pl[18 * g]); }
        // h_query rows [lo, hi) pair with h[row] (canonical, from the witness map)
        if ((e = wait_h())) return e;
        { const size_t hlo = Hq[g].lo, hhi = std::min(Hq[g].hi, D);
          if (hhi > hlo) { uint64_t hs = 0; if ((e = dgpu_scalars_copy_range(h_dev, hlo, hhi, Hq[g].ctx, &hs))) return e;
                           e = dgpu_msm_g1_resident(Hq[g].sub, 0, hs, 0, hhi - hlo, &ph[18 * g]); (void)dgpu_scalars_free(hs); if (e) return e; }
          else identity1(&ph[18 * g]); }
        return DGPU_OK;
    });
    int32_t first = DGPU_OK;
    auto note = [&](int32_t e) { if (e && !first) first = e; };
    note(jK.join());
    for (size_t g = 0; g < G; g++) note(jobs[g].join());
    note(wait_h());
    if (h_dev) (void)dgpu_scalars_free(h_dev);
    if (first) return first;
    uint64_t acc_a[18], acc_b1[18], acc_l[18], acc_h[18], acc_b2[36];
    if ((first = dgpu_fold_g1(pa.data(), G, acc_a)) || (first = dgpu_fold_g1(pb1.data(), G, acc_b1)) || (first = dgpu_fold_g1(pl.data(), G, acc_l)) ||
        (first = dgpu_fold_g1(ph.data(), G, acc_h)) || (first = dgpu_fold_g2(pb2.data(), G, acc_b2))) return first;
    ProofAB ab;
    if ((first = proof_ab(in, cst, acc_a, acc_b1, ab))) return first;
    return proof_finish(cst, ab, acc_b2, acc_l, acc_h, out_a, out_b, out_c, out_d, out_inf);
}
}  // namespace

// ---- the same prover for a host that holds its proving key as ark-ec slices (the reference: legogroth16/src/prover.rs:267-383 receives
// `pk_common: &ProvingKeyCommon<E>`, whose queries are Vec<G1Affine> / Vec<G2Affine>, and `h: &[E::ScalarField]` from QAP::witness_map) ----
// Every query is a view of host memory that the resident-bases cache resolves (bases_cache.hpp): at a key's second proof its queries are uploaded once
// and become tables, later proofs run dgpu_legogroth16_prove's schedule on them — only z and h cross PCIe.  A view the cache does not hold (the first
// proof, the cache off or full) is uploaded for the duration of the call.
namespace dock {
int32_t view_acquire_g1(const void *p, size_t stride, size_t x_off, size_t y_off, size_t inf_off, size_t n, int table_c, uint64_t *handle, void **pin);      // dock_g1.hip
int32_t view_acquire_g2(const void *p, size_t stride, size_t x_off, size_t y_off, size_t inf_off, size_t n, int table_c, uint64_t *handle, void **pin);      // dock_g2.hip
void view_release_any(void *pin);
bool view_verify_any(void *pin);        // the exact mode's stale-key check of a view, deferred so that it runs beside the proof
}
static int32_t prove_host_once(const dgpu_lego_pk_host *pk, uint64_t r1cs, const uint64_t *h, size_t h_len, int32_t h_montgomery,
                               const uint64_t *instance, size_t n_inst, const uint64_t *witness, size_t n_wit, int32_t montgomery,
                               const uint64_t r_in[4], const uint64_t s_in[4], const uint64_t v_in[4],
                               uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12], uint64_t out_d[12], uint8_t out_inf[4], bool &stale);
extern "C" int32_t dgpu_legogroth16_prove_host(const dgpu_lego_pk_host *pk, uint64_t r1cs, const uint64_t *h, size_t h_len, int32_t h_montgomery,
                                               const uint64_t *instance, size_t n_inst, const uint64_t *witness, size_t n_wit, int32_t montgomery,
                                               const uint64_t r_in[4], const uint64_t s_in[4], const uint64_t v_in[4],
                                               uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12], uint64_t out_d[12], uint8_t out_inf[4]) {
    // (exact stale-key mode: the views' records are re-fingerprinted BESIDE the proof; a proof computed from a key whose host memory had changed is thrown
    //  away and made again — the stale entries are gone by then, their views are uploaded for the call)
    for (int attempt = 0; attempt < 2; attempt++) {
        bool stale = false;
        const int32_t rc = prove_host_once(pk, r1cs, h, h_len, h_montgomery, instance, n_inst, witness, n_wit, montgomery, r_in, s_in, v_in, out_a, out_b, out_c, out_d, out_inf, stale);
        if (!stale) return rc;
    }
    return DGPU_E_BADARG;          // the key changed under two consecutive attempts: the caller is writing to it during the call
}
static int32_t prove_host_once(const dgpu_lego_pk_host *pk, uint64_t r1cs, const uint64_t *h, size_t h_len, int32_t h_montgomery,
                                               const uint64_t *instance, size_t n_inst, const uint64_t *witness, size_t n_wit, int32_t montgomery,
                                               const uint64_t r_in[4], const uint64_t s_in[4], const uint64_t v_in[4],
                               uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12], uint64_t out_d[12], uint8_t out_inf[4], bool &stale) {
    if (!pk || ((h != nullptr) == (r1cs != 0)) || !instance || n_inst == 0 || (n_wit && !witness)) return DGPU_E_BADARG;      // exactly one source of h
    try {
        struct Pins { void *p[5] = {}; ~Pins() { for (void *q : p) dock::view_release_any(q); } } pins;
        dgpu_lego_pk k{};
        const dgpu_bases_view *v[5] = {&pk->a_query, &pk->b_g1_query, &pk->b_g2_query, &pk->h_query, &pk->l_query};
        uint64_t *hd[5] = {&k.a_query, &k.b_g1_query, &k.b_g2_query, &k.h_query, &k.l_query};
        // the queries that meet the witness take the narrower table (include/dock_gpu.h DGPU_TABLE_C_WITNESS), the h query the automatic width
        // (a resident view costs microseconds; the uploads and table builds of a cold key run side by side, on the calling thread's device context)
        const int32_t arc = dock::par_run(5, [&](size_t i) -> int32_t {
            const int c = i == 3 ? 0 : (v[i]->n >= ((size_t)1 << 17) ? DGPU_TABLE_C_WITNESS : 0);
            return i == 2 ? dock::view_acquire_g2(v[i]->p, v[i]->stride, v[i]->x_off, v[i]->y_off, v[i]->inf_off, v[i]->n, c, hd[i], &pins.p[i])
                          : dock::view_acquire_g1(v[i]->p, v[i]->stride, v[i]->x_off, v[i]->y_off, v[i]->inf_off, v[i]->n, c, hd[i], &pins.p[i]);
        });
        if (arc) return arc;
        k.alpha_g1 = pk->alpha_g1; k.beta_g1 = pk->beta_g1; k.delta_g1 = pk->delta_g1; k.eta_delta_inv_g1 = pk->eta_delta_inv_g1; k.eta_gamma_inv_g1 = pk->eta_gamma_inv_g1;
        k.beta_g2 = pk->beta_g2; k.delta_g2 = pk->delta_g2; k.a0 = pk->a0; k.b1_0 = pk->b1_0; k.b2_0 = pk->b2_0;
        k.gamma_abc_g1 = pk->gamma_abc_g1; k.gamma_abc_len = pk->gamma_abc_len; k.commit_witness_count = pk->commit_witness_count;
        const uint64_t *parts[2] = {instance, witness}; const size_t counts[2] = {n_inst, n_wit};
        // the deferred checks (a no-op unless the exact mode took entries unchecked) on a thread of their own, beside the proof
        std::atomic<bool> fresh{true};      // (declared before the job that writes it: the job is joined first if anything below throws)
        Job jV;
        jV.start([&]() -> int32_t { return dock::par_run(5, [&](size_t i) -> int32_t { if (!dock::view_verify_any(pins.p[i])) fresh = false; return DGPU_OK; }); });
        const int32_t rc = prove_impl(&k, r1cs, 0, h, h_len, h_montgomery, parts, counts, n_wit ? 2 : 1, n_inst, montgomery, r_in, s_in, v_in, out_a, out_b, out_c, out_d, out_inf);
        const int32_t vrc = jV.join();
        stale = !fresh.load() || vrc != DGPU_OK;      // (a check that could not run is a check that failed: the proof is made again from what the host holds)
        return rc;
    }
    catch (const std::bad_alloc &) { return DGPU_E_OOM; }
    catch (...) { return DGPU_E_HIP; }
}
