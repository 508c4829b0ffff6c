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
#include <cstring>
#include <thread>
#include <vector>
#include "../../include/dock_gpu.h"
#include "host_field.hpp"

namespace dock {
extern thread_local bool tl_no_min;          // dock_core.hip: the size threshold (DGPU_E_TOO_SMALL) is for callers, not for the library's own calls
int32_t msm_g1_nothreshold(const uint64_t *bases, const uint8_t *is_inf, const uint64_t *scalars, size_t n, uint64_t out[18]);   // dock_g1.hip
}

namespace {
using hostf::FrH;

struct Job {
    std::thread th; int32_t rc = DGPU_OK;
    template <class F> void start(F f) { th = std::thread([this, f] { dock::tl_no_min = true; rc = f(); }); }
    int32_t join() { if (th.joinable()) th.join(); return rc; }
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
}  // namespace

extern "C" int32_t dgpu_handle_len(uint64_t handle, size_t *n);

extern "C" int32_t dgpu_legogroth16_prove(const dgpu_lego_pk *pk, uint64_t r1cs, uint64_t h_scalars, const uint64_t *z, size_t num_vars, size_t n_inst,
                                          int32_t montgomery, const uint64_t r_in[4], const uint64_t s_in[4], const uint64_t v_in[4],
                                          uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12], uint64_t out_d[12], uint8_t out_inf[4]) {
    if (!pk || !z || !r_in || !s_in || !v_in || !out_a || !out_b || !out_c || !out_d || !out_inf) return DGPU_E_BADARG;
    if ((r1cs != 0) == (h_scalars != 0)) return DGPU_E_BADARG;                      // exactly one source of h
    const size_t cw = pk->commit_witness_count;
    if (n_inst == 0 || n_inst + cw > num_vars || pk->gamma_abc_len < n_inst + cw) return DGPU_E_BADARG;
    if (!pk->alpha_g1 || !pk->beta_g1 || !pk->delta_g1 || !pk->eta_delta_inv_g1 || !pk->eta_gamma_inv_g1 || !pk->beta_g2 || !pk->delta_g2 ||
        !pk->a0 || !pk->b1_0 || !pk->b2_0 || (cw && !pk->gamma_abc_g1)) return DGPU_E_BADARG;
    size_t n_a = 0, n_b1 = 0, n_b2 = 0, n_h = 0, n_l = 0;
    if (dgpu_handle_len(pk->a_query, &n_a) || dgpu_handle_len(pk->b_g1_query, &n_b1) || dgpu_handle_len(pk->b_g2_query, &n_b2) ||
        dgpu_handle_len(pk->h_query, &n_h) || dgpu_handle_len(pk->l_query, &n_l) || n_a == 0 || n_b1 == 0 || n_b2 == 0) return DGPU_E_BADARG;
    uint64_t r[4], s[4], v[4];
    fr_reduce(r, r_in); fr_reduce(s, s_in); fr_reduce(v, v_in);
    const bool with_b1 = !is_zero4(r);                                              // prover.rs:330-336

    // ---- z resident, once ----
    uint64_t zh = 0;
    int32_t rc = dgpu_scalars_upload(z, num_vars, montgomery, &zh);
    if (rc) return rc;
    const size_t n_assign = num_vars - 1;                                            // assignment = z[1..]
    const size_t n_aux = num_vars - n_inst - cw, aux_at = n_inst + cw;              // aux = witnesses after the committed ones (prover.rs:292-299)

    // ---- job H: witness map + the h_query MSM ----
    uint64_t h_owned = 0;
    uint64_t acc_h[18], acc_a[18], acc_b1[18], acc_l[18], acc_b2[36];
    Job jH, jB2, jA, jB1, jL, jK;
    jH.start([&]() -> int32_t {
        uint64_t hh = h_scalars; size_t D = 0;
        if (r1cs) { int32_t e = dgpu_witness_map_r1cs_resident(r1cs, zh, nullptr, &h_owned, &D); if (e) return e; hh = h_owned; }
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

    // ---- job K: what depends on no MSM (host arithmetic; the tiny g_d MSM goes to the device only when it has more than 15 terms) ----
    uint64_t rest_a[18], rest_b1[18], rest_c[18], rest_b2[36], g_d[18];
    jK.start([&]() -> int32_t {
        int32_t e;
        { uint64_t p[36]; memcpy(p, pk->delta_g1, 96); memcpy(p + 12, pk->a0, 96); memcpy(p + 24, pk->alpha_g1, 96);
          uint64_t k[12]; memcpy(k, r, 32); memcpy(k + 4, ONE4, 32); memcpy(k + 8, ONE4, 32);
          if ((e = dgpu_lincomb_g1(p, nullptr, k, 3, rest_a))) return e; }                                  // r delta + a_0 + alpha  (:585-594)
        { uint64_t p[72]; memcpy(p, pk->delta_g2, 192); memcpy(p + 24, pk->b2_0, 192); memcpy(p + 48, pk->beta_g2, 192);
          uint64_t k[12]; memcpy(k, s, 32); memcpy(k + 4, ONE4, 32); memcpy(k + 8, ONE4, 32);
          if ((e = dgpu_lincomb_g2(p, nullptr, k, 3, rest_b2))) return e; }
        if (with_b1) { uint64_t p[36]; memcpy(p, pk->delta_g1, 96); memcpy(p + 12, pk->b1_0, 96); memcpy(p + 24, pk->beta_g1, 96);
          uint64_t k[12]; memcpy(k, s, 32); memcpy(k + 4, ONE4, 32); memcpy(k + 8, ONE4, 32);
          if ((e = dgpu_lincomb_g1(p, nullptr, k, 3, rest_b1))) return e; }
        { uint64_t p[24]; memcpy(p, pk->delta_g1, 96); memcpy(p + 12, pk->eta_delta_inv_g1, 96);
          uint64_t k[8], rs[4]; fr_mul(rs, r, s); fr_neg(k, rs); fr_neg(k + 4, v);
          if ((e = dgpu_lincomb_g1(p, nullptr, k, 2, rest_c))) return e; }                                  // -rs delta - v eta/delta  (:350-355)
        // g_d = msm(gamma_abc[n_inst .. n_inst + cw], committed witnesses) + v eta/gamma   (:361-368)
        std::vector<uint64_t> pts((cw + 1) * 12), sc((cw + 1) * 4);
        if (cw) memcpy(pts.data(), pk->gamma_abc_g1 + n_inst * 12, cw * 96);
        memcpy(pts.data() + cw * 12, pk->eta_gamma_inv_g1, 96);
        for (size_t i = 0; i < cw; i++) { if (montgomery) hostf::fr_from_mont(&sc[4 * i], z + 4 * (n_inst + i)); else memcpy(&sc[4 * i], z + 4 * (n_inst + i), 32); }
        memcpy(&sc[4 * cw], v, 32);
        if (cw + 1 <= DGPU_MAX_LINCOMB) return dgpu_lincomb_g1(pts.data(), nullptr, sc.data(), cw + 1, g_d);
        return dock::msm_g1_nothreshold(pts.data(), nullptr, sc.data(), cw + 1, g_d);
    });

    // ---- assemble ----
    int32_t first = DGPU_OK;
    auto note = [&](int32_t e) { if (e && !first) first = e; };
    note(jK.join()); note(jA.join()); if (with_b1) note(jB1.join());
    uint64_t ga[36], sa_rb[18];                     // g_a then g1_b as affine points, for s g_a + r g1_b
    uint8_t inf_a = 1, inf_b1 = 1, inf_two[2] = {1, 1};
    memset(ga, 0, sizeof ga);
    if (!first) {
        uint64_t parts[36]; memcpy(parts, acc_a, 144); memcpy(parts + 18, rest_a, 144);
        note(fold_to_affine_g1(parts, 2, ga, &inf_a));
        if (with_b1) { memcpy(parts, acc_b1, 144); memcpy(parts + 18, rest_b1, 144); note(fold_to_affine_g1(parts, 2, ga + 12, &inf_b1)); }
        inf_two[0] = inf_a; inf_two[1] = inf_b1;
        uint64_t k[8]; memcpy(k, s, 32); memcpy(k + 4, r, 32);
        if (!first) note(dgpu_lincomb_g1(ga, inf_two, k, 2, sa_rb));                    // (while the G2 MSM is still in flight)
    }
    note(jB2.join()); note(jL.join()); note(jH.join());
    if (sorted) (void)dgpu_scalars_free(sorted);
    if (h_owned) (void)dgpu_scalars_free(h_owned);
    (void)dgpu_scalars_free(zh);
    if (first) return first;
    {
        uint64_t parts[72], jb[36];
        memcpy(parts, acc_b2, 288); memcpy(parts + 36, rest_b2, 288);
        if ((rc = dgpu_fold_g2(parts, 2, jb))) return rc;
        bool zb = true; for (int i = 24; i < 36; i++) zb = zb && jb[i] == 0;
        out_inf[1] = zb; if (zb) memset(out_b, 0, 192); else memcpy(out_b, jb, 192);
    }
    out_inf[0] = inf_a; memcpy(out_a, ga, 96);
    {
        uint64_t parts[72]; memcpy(parts, sa_rb, 144); memcpy(parts + 18, rest_c, 144); memcpy(parts + 36, acc_l, 144); memcpy(parts + 54, acc_h, 144);
        if ((rc = fold_to_affine_g1(parts, 4, out_c, &out_inf[2]))) return rc;            // g_c = s g_a + r g1_b - rs delta + l_aux + h_acc - v eta/delta
    }
    return fold_to_affine_g1(g_d, 1, out_d, &out_inf[3]);
}
