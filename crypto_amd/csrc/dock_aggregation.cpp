// crypto_amd/csrc/dock_aggregation.cpp — SnarkPack aggregation of Groth16 / LegoGroth16 proofs as two entry points of the C ABI
// (include/dock_gpu.h: dgpu_snarkpack_aggregate, dgpu_snarkpack_verify).
//
// Replaces, at the granularity the reference calls them,
//   legogroth16/src/aggregation/groth16/prover.rs    aggregate_proofs :47-147, prove_tipp_mipp :156-206, gipa_tipp_mipp :212-382
//   legogroth16/src/aggregation/legogroth16/prover.rs :38-127 (one more MIPP, for the commitments d), using_groth16.rs :26-128
//   legogroth16/src/aggregation/groth16/verifier.rs  verify_aggregate_proof :36-100, verify_tipp_mipp :102-192, gipa_verify_tipp_mipp :194-400
//   legogroth16/src/aggregation/kzg.rs               :30-343, commitment.rs :23-69, key.rs :96-187, utils.rs :34-265
// The protocol's group and pairing work goes through the library's own entry points — the segmented multi-pairing (ten to eighteen independent
// `E::multi_pairing`s of a GIPA round in ONE call), the folding kernel (dgpu_g*_mul_add_batch), the small-MSM path, the GT multi-exponentiation —
// issued from host threads where the reference issues them one after another.  The Fiat-Shamir transcript stays the CALLER's: the reference
// takes `&mut impl Transcript` (utils/src/transcript.rs:45-63), so the entry points take two callbacks (append_message, challenge_scalar) and
// hand over exactly the bytes `Transcript::append` would serialize (`serialize_compressed`: Zcash points, canonical little-endian field
// elements).  crypto_amd/aggregation/*.py is the same protocol in Python (the test driver this file is compared with, element by element).
//
// Plain host C++: no kernel, no HIP call.  Points are affine ABI words (identity: all-zero words), scalars canonical 4 x u64 at the boundary
// and Montgomery inside.  Nothing unwinds through the ABI: failures travel as Fail / Reject to the entry point.
#include <algorithm>
#include <array>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include "../../include/dock_gpu.h"
#include "host_field.hpp"
#include "host_par.hpp"        // (dock::tl_no_min: the size threshold DGPU_E_TOO_SMALL is for callers, not for the library's own calls)

namespace dock { bool g1_words_valid(const uint64_t xy[12]); bool g2_words_valid(const uint64_t xy[24]); }      // dock_serde.cpp

namespace {
using hostf::FrH;
typedef uint64_t W;
typedef std::vector<W> Vec;
typedef std::array<W, 72> Gt;

struct Fail { int32_t rc; };                  // an ABI error code on its way to the entry point
struct Reject {};                             // the proof does not verify (AggregationError::InvalidProof): *ok = 0
inline void ck(int32_t rc) { if (rc) throw Fail{rc}; }
inline void need(bool c) { if (!c) throw Fail{DGPU_E_BADARG}; }

// ---- Fr, Montgomery form ----------------------------------------------------------------------------------------------------------------
struct Fr {
    FrH v;
    static Fr raw(const FrH &x) { Fr r; r.v = x; return r; }
    static Fr from_canon(const W a[4]) { FrH x, r2; memcpy(x.l, a, 32); memcpy(r2.l, FrH::R2, 32); return raw(FrH::mont_mul(x, r2)); }   // any a < 2^256
    static Fr from_u64(uint64_t x) { return raw(FrH::from_u64(x)); }
    static Fr zero() { Fr r; memset(&r, 0, sizeof r); return r; }
    static Fr one() { return from_u64(1); }
    void canon(W out[4]) const { v.to_canonical(out); }
    Fr operator*(const Fr &b) const { return raw(v * b.v); }
    Fr operator+(const Fr &b) const {
        Fr r; uint64_t c = 0;
        for (int i = 0; i < 4; i++) { hostf::u128 s = (hostf::u128)v.l[i] + b.v.l[i] + c; r.v.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        bool ge = c != 0;
        if (!ge) { ge = true; for (int i = 3; i >= 0; i--) { if (r.v.l[i] > FrH::MOD[i]) break; if (r.v.l[i] < FrH::MOD[i]) { ge = false; break; } } }
        if (ge) { uint64_t br = 0; for (int i = 0; i < 4; i++) { hostf::u128 d = (hostf::u128)r.v.l[i] - FrH::MOD[i] - br; r.v.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }
        return r;
    }
    Fr operator-(const Fr &b) const {
        Fr r; uint64_t br = 0;
        for (int i = 0; i < 4; i++) { hostf::u128 d = (hostf::u128)v.l[i] - b.v.l[i] - br; r.v.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
        if (br) { uint64_t c = 0; for (int i = 0; i < 4; i++) { hostf::u128 s = (hostf::u128)r.v.l[i] + FrH::MOD[i] + c; r.v.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
        return r;
    }
    Fr neg() const { return zero() - *this; }
    Fr inv() const { return raw(v.inv()); }
    Fr pow(uint64_t e) const { Fr acc = one(), base = *this; for (; e; e >>= 1) { if (e & 1) acc = acc * base; base = base * base; } return acc; }
    bool is_zero() const { return !(v.l[0] | v.l[1] | v.l[2] | v.l[3]); }
};
typedef std::vector<Fr> Frs;

Vec canon_words(const Frs &s) { Vec o(4 * s.size()); for (size_t i = 0; i < s.size(); i++) s[i].canon(&o[4 * i]); return o; }
Frs powers(const Fr &r, size_t n) { Frs o(n); Fr acc = Fr::one(); for (size_t i = 0; i < n; i++) { o[i] = acc; acc = acc * r; } return o; }

// ---- independent ABI calls from host threads (the library keeps six calls in flight per device; further callers queue for a slot) ----------
void par(std::vector<std::function<void()>> thunks) {
    const size_t k = thunks.size();
    std::vector<uint8_t> rej(k, 0);
    const int32_t rc = dock::par_run(k, [&](size_t i) -> int32_t {       // (the parts inherit this thread's tl_no_min = true: host_par.hpp)
        try { thunks[i](); } catch (const Fail &f) { return f.rc; } catch (const Reject &) { rej[i] = 1; }
        return DGPU_OK;
    });
    if (rc) throw Fail{rc};
    for (size_t i = 0; i < k; i++) if (rej[i]) throw Reject{};
}

// ---- group helpers over the ABI ---------------------------------------------------------------------------------------------------------
inline bool is_id(const W *p, int aw) { W t = 0; for (int i = 0; i < aw; i++) t |= p[i]; return t == 0; }
inline int aw_of(bool g2) { return g2 ? 24 : 12; }

// [addend_i + s_i P_i]: one scalar for all (stride 0) or one per point
Vec mul_add(bool g2, const W *pts, size_t n, const Vec &sc, size_t stride, const W *addend) {
    const int aw = aw_of(g2);
    Vec out(n * aw, 0);
    if (n == 0) return out;
    std::vector<uint8_t> oinf(n);
    ck(g2 ? dgpu_g2_mul_add_batch(pts, nullptr, sc.data(), stride, addend, nullptr, n, out.data(), oinf.data())
          : dgpu_g1_mul_add_batch(pts, nullptr, sc.data(), stride, addend, nullptr, n, out.data(), oinf.data()));
    for (size_t i = 0; i < n; i++) if (oinf[i]) memset(&out[i * aw], 0, aw * 8);
    return out;
}
// sum s_i P_i as an affine point.  Up to DGPU_MAX_LINCOMB terms (the `mul_bigint`s and two-term combinations of the KZG checks and of the final
// verification: CPU scalar multiplications in the reference): host arithmetic; more: the MSM entry point.
Vec msm(bool g2, const W *pts, size_t n, const Frs &sc) {
    const int aw = aw_of(g2);
    Vec out(aw, 0);
    if (n == 0) return out;
    need(sc.size() >= n);
    Vec s = canon_words(sc), jac(aw * 3 / 2);
    std::vector<uint8_t> inf(n);
    for (size_t i = 0; i < n; i++) inf[i] = is_id(pts + i * aw, aw);
    const bool keep = dock::tl_no_min; dock::tl_no_min = true;
    int32_t rc;
    if (n <= DGPU_MAX_LINCOMB) rc = g2 ? dgpu_lincomb_g2(pts, inf.data(), s.data(), n, jac.data()) : dgpu_lincomb_g1(pts, inf.data(), s.data(), n, jac.data());
    else rc = g2 ? dgpu_msm_g2(pts, inf.data(), s.data(), n, jac.data()) : dgpu_msm_g1(pts, inf.data(), s.data(), n, jac.data());
    dock::tl_no_min = keep;
    ck(rc);
    if (!is_id(&jac[aw], aw / 2)) memcpy(out.data(), jac.data(), aw * 8);
    return out;
}
Vec neg_point(bool g2, const W *p) {
    const int aw = aw_of(g2), h = aw / 2;
    Vec o(p, p + aw);
    if (is_id(p, aw)) return o;
    for (int k = 0; k < h / 6; k++) { hostf::Fq y; memcpy(y.l, p + h + 6 * k, 48); y = y.neg(); memcpy(&o[h + 6 * k], y.l, 48); }
    return o;
}

// the mutually independent multi-pairings of one step as segments of ONE call (dgpu_multi_pairing_segments); a segment is built from parts
struct PairJobs {
    Vec p, q; std::vector<uint64_t> ends;
    void part(const W *ps, const W *qs, size_t n) { p.insert(p.end(), ps, ps + 12 * n); q.insert(q.end(), qs, qs + 24 * n); }
    void close() { ends.push_back(p.size() / 12); }
    void one(const W *ps, const W *qs, size_t n) { part(ps, qs, n); close(); }
    std::vector<Gt> run() const {
        std::vector<Gt> out(ends.size());
        if (ends.empty()) return out;
        ck(dgpu_multi_pairing_segments(p.data(), q.data(), nullptr, p.size() / 12, ends.data(), ends.size(), out[0].data()));
        return out;
    }
};
Gt gt_one() { Gt o{}; memcpy(o.data(), hostf::Fq::ONE, 48); return o; }

// ---- the caller's transcript ----------------------------------------------------------------------------------------------------------------
struct Tr {
    const dgpu_transcript *t;
    void bytes(const char *label, const uint8_t *b, size_t n) const { t->append_message(t->ctx, (const uint8_t *)label, strlen(label), b, n); }
    static void gt_bytes(uint8_t *o, const Gt &f) {                // PairingOutput / Fp12: twelve Fp, c0.c0.c0 first, 48 bytes little-endian canonical each
        hostf::Fq one_raw = hostf::Fq::zero(); one_raw.l[0] = 1;
        for (int i = 0; i < 12; i++) { hostf::Fq a; memcpy(a.l, &f[6 * i], 48); a = a * one_raw; memcpy(o + 48 * i, a.l, 48); }
    }
    void gt(const char *label, const Gt &f) const { uint8_t b[576]; gt_bytes(b, f); bytes(label, b, 576); }
    void pc(const char *label, const Gt &t_, const Gt &u_) const { uint8_t b[1152]; gt_bytes(b, t_); gt_bytes(b + 576, u_); bytes(label, b, 1152); }   // PairCommitment (t, u)
    void g1(const char *label, const W *p) const { uint8_t b[48]; ck(dgpu_g1_serialize(p, nullptr, 1, 1, b)); bytes(label, b, 48); }
    void g2(const char *label, const W *p) const { uint8_t b[96]; ck(dgpu_g2_serialize(p, nullptr, 1, 1, b)); bytes(label, b, 96); }
    void fr(const char *label, const Fr &x) const { W w[4]; x.canon(w); bytes(label, (const uint8_t *)w, 32); }
    Fr challenge(const char *label) const { W w[4] = {0, 0, 0, 0}; t->challenge_scalar(t->ctx, (const uint8_t *)label, strlen(label), w); return Fr::from_canon(w); }
};

// ---- the proof ------------------------------------------------------------------------------------------------------------------------------
struct PairCommitment { Gt t, u; };
struct Mipp {                                  // per committed G1 vector (C; D for LegoGroth16): commitment, aggregate, the GIPA elements
    PairCommitment com; Vec z;                 // z: 12 words
    std::vector<std::array<PairCommitment, 2>> comms; std::vector<std::array<std::array<W, 12>, 2>> zs; std::array<W, 12> final_;
};
struct Proof {
    size_t n = 0; int nm = 1;                  // nproofs; number of MIPP instances (1: Groth16, 2: LegoGroth16)
    PairCommitment com_ab; Gt z_ab; Mipp m[2];
    std::vector<std::array<PairCommitment, 2>> comms_ab; std::vector<std::array<Gt, 2>> zs_ab;
    std::array<W, 12> final_a; std::array<W, 24> final_b;
    std::array<W, 24> final_vkey[2]; std::array<W, 12> final_wkey[2];
    std::array<W, 24> vkey_opening[2]; std::array<W, 12> wkey_opening[2];
};
size_t log2_exact(size_t n) { size_t l = 0; while (((size_t)1 << l) < n) l++; return l; }
size_t proof_words(size_t n, int nm) {
    const size_t L = log2_exact(n);
    return 2 + 144 + 72 + (size_t)nm * (144 + 12) + L * (288 + 144 + (size_t)nm * (288 + 24)) + 12 + 24 + (size_t)nm * 12 + 48 + 24 + 48 + 24;
}
// word layout (the field order of AggregateProof / GipaProof / TippMippProof, groth16/proof.rs:12-24,76-84,117-121; the LegoGroth16 proof carries the
// D members right after the C ones, legogroth16/proof.rs:12-26,79-95):
//   nproofs, n_mipp | com_ab(t,u) com_c [com_d] | z_ab z_c [z_d] | comms_ab[L](l.t l.u r.t r.u) comms_c[L] [comms_d[L]] | z_ab[L](l r) z_c[L] [z_d[L]]
//   | final_a final_b final_c [final_d] final_vkey(2) final_wkey(2) | vkey_opening(2) wkey_opening(2)
struct Wr { W *o; template <class A> void put(const A &a) { memcpy(o, a.data(), a.size() * 8); o += a.size(); } void pc(const PairCommitment &c) { put(c.t); put(c.u); } };
struct Rd { const W *i; template <class A> void get(A &a) { memcpy(a.data(), i, a.size() * 8); i += a.size(); } void pc(PairCommitment &c) { get(c.t); get(c.u); } };
void write_proof(const Proof &p, W *out) {
    Wr w{out};
    *w.o++ = p.n; *w.o++ = (W)p.nm;
    w.pc(p.com_ab); for (int k = 0; k < p.nm; k++) w.pc(p.m[k].com);
    w.put(p.z_ab); for (int k = 0; k < p.nm; k++) w.put(p.m[k].z);
    for (auto &c : p.comms_ab) { w.pc(c[0]); w.pc(c[1]); }
    for (int k = 0; k < p.nm; k++) for (auto &c : p.m[k].comms) { w.pc(c[0]); w.pc(c[1]); }
    for (auto &z : p.zs_ab) { w.put(z[0]); w.put(z[1]); }
    for (int k = 0; k < p.nm; k++) for (auto &z : p.m[k].zs) { w.put(z[0]); w.put(z[1]); }
    w.put(p.final_a); w.put(p.final_b); for (int k = 0; k < p.nm; k++) w.put(p.m[k].final_);
    w.put(p.final_vkey[0]); w.put(p.final_vkey[1]); w.put(p.final_wkey[0]); w.put(p.final_wkey[1]);
    w.put(p.vkey_opening[0]); w.put(p.vkey_opening[1]); w.put(p.wkey_opening[0]); w.put(p.wkey_opening[1]);
}
// parsing_check (groth16/proof.rs:29-58, legogroth16/proof.rs:44-75): the length bounds, the power of two, every vector log2(n) long — here the
// vector lengths are implied by the total length, which has to match
void read_proof(Proof &p, const W *in, size_t len, int want_nm) {
    need(in && len >= 2);
    p.n = (size_t)in[0]; p.nm = (int)in[1];
    if (p.n < 2 || p.n > DGPU_SNARKPACK_MAX_SRS_SIZE || (p.n & (p.n - 1)) || p.nm != want_nm || len != proof_words(p.n, p.nm)) throw Fail{DGPU_E_BADARG};
    const size_t L = log2_exact(p.n);
    Rd r{in + 2};
    r.pc(p.com_ab); for (int k = 0; k < p.nm; k++) r.pc(p.m[k].com);
    r.get(p.z_ab); for (int k = 0; k < p.nm; k++) { p.m[k].z.resize(12); r.get(p.m[k].z); }
    p.comms_ab.resize(L); for (auto &c : p.comms_ab) { r.pc(c[0]); r.pc(c[1]); }
    for (int k = 0; k < p.nm; k++) { p.m[k].comms.resize(L); for (auto &c : p.m[k].comms) { r.pc(c[0]); r.pc(c[1]); } }
    p.zs_ab.resize(L); for (auto &z : p.zs_ab) { r.get(z[0]); r.get(z[1]); }
    for (int k = 0; k < p.nm; k++) { p.m[k].zs.resize(L); for (auto &z : p.m[k].zs) { r.get(z[0]); r.get(z[1]); } }
    r.get(p.final_a); r.get(p.final_b); for (int k = 0; k < p.nm; k++) r.get(p.m[k].final_);
    r.get(p.final_vkey[0]); r.get(p.final_vkey[1]); r.get(p.final_wkey[0]); r.get(p.final_wkey[1]);
    r.get(p.vkey_opening[0]); r.get(p.vkey_opening[1]); r.get(p.wkey_opening[0]); r.get(p.wkey_opening[1]);
}

const char *const MIPP_TAG[2] = {"c", "d"};
struct Labels { std::string comm, z_l, z_r, tu_l, tu_r, commitment; };
Labels labels(int k) {
    const std::string t = MIPP_TAG[k];
    return {"comm-" + t, "z" + t + "_l", "z" + t + "_r", "tu" + t + "_l", "tu" + t + "_r", std::string(k == 0 ? "C" : "D") + "-commitment"};
}

// ---- KZG openings of the final commitment keys (kzg.rs) ---------------------------------------------------------------------------------------
// prod_i (1 + x_i (z r)^(2^i))   kzg.rs:238-256
Fr poly_eval_product_form(const Frs &tr, const Fr &z, const Fr &r_shift) {
    Fr power_zr = z * r_shift, res = Fr::one() + tr[0] * power_zr;
    for (size_t i = 1; i < tr.size(); i++) { power_zr = power_zr * power_zr; res = res * (Fr::one() + tr[i] * power_zr); }
    return res;
}
// the coefficients of the same polynomial   kzg.rs:258-292
Frs poly_coefficients(const Frs &tr, const Fr &r_shift) {
    Frs c{Fr::one()};
    c.reserve((size_t)1 << tr.size());
    Fr power_2_r = r_shift;
    for (size_t i = 0; i < tr.size(); i++) {
        if (i > 0) power_2_r = power_2_r * power_2_r;
        const Fr k = tr[i] * power_2_r;
        const size_t m = c.size();
        for (size_t j = 0; j < m; j++) c.push_back(c[j] * k);
    }
    return c;
}
// create_kzg_opening (kzg.rs:182-236): quotient of (poly - eval) by (X - z), committed under both SRS tables
void kzg_opening(bool g2, const W *alpha_tab, const W *beta_tab, size_t tab_len, Frs poly, const Fr &eval, const Fr &z, W *out_a, W *out_b) {
    need(poly.size() == tab_len && tab_len >= 1);
    poly[0] = poly[0] - eval;
    Frs q(tab_len, Fr::zero());
    Fr carry = Fr::zero();
    for (size_t i = tab_len - 1; i >= 1; i--) { carry = poly[i] + carry * z; q[i - 1] = carry; }
    Vec a, b;
    par({[&] { a = msm(g2, alpha_tab, tab_len, q); }, [&] { b = msm(g2, beta_tab, tab_len, q); }});
    memcpy(out_a, a.data(), a.size() * 8); memcpy(out_b, b.data(), b.size() * 8);
}

// ---- prover ---------------------------------------------------------------------------------------------------------------------------------
struct Key { Vec a, b; };                       // VKey over G2, WKey over G1 (key.rs:41-57)

void aggregate(const dgpu_snarkpack_prover_srs *srs, const W *pa, const W *pb, const W *const pv[2], int nm, size_t n, const Tr &tr, Proof &P) {
    need(n >= 2 && (n & (n - 1)) == 0 && n <= DGPU_SNARKPACK_MAX_SRS_SIZE && srs->n == n);     // prover.rs:52-66 (has_correct_len: the SRS is specialised to n)
    const size_t L = log2_exact(n);
    P.n = n; P.nm = nm;
    Key vkey{Vec(srs->vkey_a, srs->vkey_a + 24 * n), Vec(srs->vkey_b, srs->vkey_b + 24 * n)}, wkey{Vec(srs->wkey_a, srs->wkey_a + 12 * n), Vec(srs->wkey_b, srs->wkey_b + 12 * n)};
    // commitments to (A, B), C (and D)   prover.rs:77-88
    {
        PairJobs j;
        j.part(pa, vkey.a.data(), n); j.part(wkey.a.data(), pb, n); j.close();
        j.part(pa, vkey.b.data(), n); j.part(wkey.b.data(), pb, n); j.close();
        for (int k = 0; k < nm; k++) { j.one(pv[k], vkey.a.data(), n); j.one(pv[k], vkey.b.data(), n); }
        const std::vector<Gt> g = j.run();
        P.com_ab = {g[0], g[1]};
        for (int k = 0; k < nm; k++) P.m[k].com = {g[2 + 2 * k], g[3 + 2 * k]};
    }
    tr.pc("AB-commitment", P.com_ab.t, P.com_ab.u);
    for (int k = 0; k < nm; k++) tr.pc(labels(k).commitment.c_str(), P.m[k].com.t, P.m[k].com.u);
    const Fr r = tr.challenge("r-random-fiatshamir");
    Frs m_r = powers(r, n);
    const Frs r_inv = powers(r.inv(), n);         // 1, r^-1, r^-2, ... (the reference batch-inverts r_vec, :101-103)
    // B^{r^i}, Z_AB = prod e(A_i, B_i^{r^i}), w^{r^-i}, the aggregates sum r^i C_i   :107-120, mutually independent
    Vec m_a(pa, pa + 12 * n), m_b, m_v[2];
    for (int k = 0; k < nm; k++) m_v[k].assign(pv[k], pv[k] + 12 * n);
    {
        std::vector<std::function<void()>> th;
        th.push_back([&] {
            m_b = mul_add(true, pb, n, canon_words(m_r), 4, nullptr);
            PairJobs j; j.one(pa, m_b.data(), n);
            P.z_ab = j.run()[0];
        });
        th.push_back([&] {                      // Key::scale (key.rs:117-139): both vectors in one launch
            Vec both(wkey.a); both.insert(both.end(), wkey.b.begin(), wkey.b.end());
            Frs s2(r_inv); s2.insert(s2.end(), r_inv.begin(), r_inv.end());
            const Vec o = mul_add(false, both.data(), 2 * n, canon_words(s2), 4, nullptr);
            wkey.a.assign(o.begin(), o.begin() + 12 * n); wkey.b.assign(o.begin() + 12 * n, o.end());
        });
        for (int k = 0; k < nm; k++) th.push_back([&, k] { P.m[k].z = msm(false, pv[k], n, m_r); });
        par(th);
    }
    // ---- gipa_tipp_mipp (groth16/prover.rs:212-382; legogroth16/prover.rs:176-350 runs one more MIPP, for D) ----
    tr.gt("inner-product-ab", P.z_ab);
    for (int k = 0; k < nm; k++) tr.g1(labels(k).comm.c_str(), P.m[k].z.data());
    Fr c_inv = tr.challenge("first-challenge"), ch = c_inv.inv();
    Frs challenges, challenges_inv;
    P.comms_ab.resize(L); P.zs_ab.resize(L);
    for (int k = 0; k < nm; k++) { P.m[k].comms.resize(L); P.m[k].zs.resize(L); }
    size_t len = n;
    for (size_t i = 0; len > 1; i++) {
        const size_t s = len / 2;
        const W *a_l = m_a.data(), *a_r = a_l + 12 * s, *b_l = m_b.data(), *b_r = b_l + 24 * s;
        const W *vka_l = vkey.a.data(), *vka_r = vka_l + 24 * s, *vkb_l = vkey.b.data(), *vkb_r = vkb_l + 24 * s;
        const W *wka_l = wkey.a.data(), *wka_r = wka_l + 12 * s, *wkb_l = wkey.b.data(), *wkb_r = wkb_l + 12 * s;
        // TIPP (utils.rs:83-118) and the MIPPs (utils.rs:51-81): 6 + 4 per MIPP multi-pairings and two MSMs per MIPP, all independent
        PairJobs j;
        j.part(a_r, vka_l, s); j.part(wka_r, b_l, s); j.close();        // tab_l = double(vk_left, wk_right, a_right, b_left)   commitment.rs:36-69
        j.part(a_r, vkb_l, s); j.part(wkb_r, b_l, s); j.close();
        j.part(a_l, vka_r, s); j.part(wka_l, b_r, s); j.close();        // tab_r = double(vk_right, wk_left, a_left, b_right)
        j.part(a_l, vkb_r, s); j.part(wkb_l, b_r, s); j.close();
        j.one(a_r, b_l, s); j.one(a_l, b_r, s);                          // zab_l, zab_r
        for (int k = 0; k < nm; k++) {
            const W *v_l = m_v[k].data(), *v_r = v_l + 12 * s;
            j.one(v_r, vka_l, s); j.one(v_r, vkb_l, s);                  // single(vk_left, v_right)    commitment.rs:23-33
            j.one(v_l, vka_r, s); j.one(v_l, vkb_r, s);                  // single(vk_right, v_left)
        }
        std::vector<Gt> g;
        Vec z_l[2], z_r[2];
        const Frs r_l(m_r.begin(), m_r.begin() + s), r_r(m_r.begin() + s, m_r.begin() + 2 * s);
        // what the round's folding step (below) will multiply by the challenge: the right halves.  Their doubling chains do not depend on the challenge and
        // run now, beside the pairings (dgpu_fold_prepare_pair: ~0.8 ms, the pairings take ~1.5); the step itself is then a shallow tree (fold_kernels.hip.h)
        Vec right, left, right2, left2;
        {
            auto cat = [&](Vec &dst, const W *p, size_t words) { dst.insert(dst.end(), p, p + words); };
            cat(right, a_r, 12 * s); cat(right, wka_r, 12 * s); cat(right, wkb_r, 12 * s);
            cat(left, a_l, 12 * s); cat(left, wka_l, 12 * s); cat(left, wkb_l, 12 * s);
            for (int k = 0; k < nm; k++) { cat(right, m_v[k].data() + 12 * s, 12 * s); cat(left, m_v[k].data(), 12 * s); }
            cat(right2, b_r, 24 * s); cat(right2, vka_r, 24 * s); cat(right2, vkb_r, 24 * s);
            cat(left2, b_l, 24 * s); cat(left2, vka_l, 24 * s); cat(left2, vkb_l, 24 * s);
        }
        uint64_t prep1 = 0, prep2 = 0;
        struct FreeFold { uint64_t &h; ~FreeFold() { if (h) (void)dgpu_fold_free(h); } } free1{prep1}, free2{prep2};
        {
            std::vector<std::function<void()>> th;
            th.push_back([&] { g = j.run(); });
            th.push_back([&] { if (dgpu_fold_prepare_pair(right.data(), (3 + nm) * s, &prep1, right2.data(), 3 * s, &prep2)) { prep1 = 0; prep2 = 0; } });      // (no table: the chain kernels do the step)
            for (int k = 0; k < nm; k++) {
                th.push_back([&, k] { z_l[k] = msm(false, m_v[k].data() + 12 * s, s, r_l); });
                th.push_back([&, k] { z_r[k] = msm(false, m_v[k].data(), s, r_r); });
            }
            par(th);
        }
        P.comms_ab[i] = {PairCommitment{g[0], g[1]}, PairCommitment{g[2], g[3]}};
        P.zs_ab[i] = {g[4], g[5]};
        for (int k = 0; k < nm; k++) {
            P.m[k].comms[i] = {PairCommitment{g[6 + 4 * k], g[7 + 4 * k]}, PairCommitment{g[8 + 4 * k], g[9 + 4 * k]}};
            memcpy(P.m[k].zs[i][0].data(), z_l[k].data(), 96); memcpy(P.m[k].zs[i][1].data(), z_r[k].data(), 96);
        }
        if (i > 0) {                             // prover.rs:292-326 (the first round uses the challenge drawn before the loop)
            tr.fr("c_inv", c_inv);
            tr.gt("zab_l", P.zs_ab[i][0]); tr.gt("zab_r", P.zs_ab[i][1]);
            for (int k = 0; k < nm; k++) { const Labels lb = labels(k); tr.g1(lb.z_l.c_str(), P.m[k].zs[i][0].data()); tr.g1(lb.z_r.c_str(), P.m[k].zs[i][1].data()); }
            tr.pc("tab_l", P.comms_ab[i][0].t, P.comms_ab[i][0].u); tr.pc("tab_r", P.comms_ab[i][1].t, P.comms_ab[i][1].u);
            for (int k = 0; k < nm; k++) {
                const Labels lb = labels(k);
                tr.pc(lb.tu_l.c_str(), P.m[k].comms[i][0].t, P.m[k].comms[i][0].u); tr.pc(lb.tu_r.c_str(), P.m[k].comms[i][1].t, P.m[k].comms[i][1].u);
            }
            c_inv = tr.challenge("challenge_i"); ch = c_inv.inv();
        }
        // folding (prover.rs:328-351): A, the MIPP vectors and both w vectors take the challenge, B and both v vectors its inverse — one launch per group
        // instead of `compress` x (2 + n_mipp) + Key::compress x 2
        Vec g1_new, g2_new;
        {
            Vec chw(4), ciw(4); ch.canon(chw.data()); c_inv.canon(ciw.data());
            auto fold = [&](bool g2, uint64_t prep, const Vec &rt, const Vec &lf, size_t cnt, const Vec &cw) {
                if (!prep) return mul_add(g2, rt.data(), cnt, cw, 0, lf.data());
                const int aw = aw_of(g2);
                Vec out(cnt * aw, 0); std::vector<uint8_t> oinf(cnt);
                ck(g2 ? dgpu_g2_fold_apply(prep, cw.data(), lf.data(), out.data(), oinf.data()) : dgpu_g1_fold_apply(prep, cw.data(), lf.data(), out.data(), oinf.data()));
                return out;                                            // (identity outputs are zero words already)
            };
            par({[&] { g1_new = fold(false, prep1, right, left, (3 + nm) * s, chw); },
                 [&] { g2_new = fold(true, prep2, right2, left2, 3 * s, ciw); }});
        }
        m_a.assign(g1_new.begin(), g1_new.begin() + 12 * s);
        wkey.a.assign(g1_new.begin() + 12 * s, g1_new.begin() + 24 * s); wkey.b.assign(g1_new.begin() + 24 * s, g1_new.begin() + 36 * s);
        for (int k = 0; k < nm; k++) m_v[k].assign(g1_new.begin() + (3 + k) * 12 * s, g1_new.begin() + (4 + k) * 12 * s);
        m_b.assign(g2_new.begin(), g2_new.begin() + 24 * s);
        vkey.a.assign(g2_new.begin() + 24 * s, g2_new.begin() + 48 * s); vkey.b.assign(g2_new.begin() + 48 * s, g2_new.end());
        Frs r_new(s);
        for (size_t t = 0; t < s; t++) r_new[t] = r_l[t] + r_r[t] * c_inv;
        m_r.swap(r_new);
        challenges.push_back(ch); challenges_inv.push_back(c_inv);
        len = s;
    }
    memcpy(P.final_a.data(), m_a.data(), 96); memcpy(P.final_b.data(), m_b.data(), 192);
    for (int k = 0; k < nm; k++) memcpy(P.m[k].final_.data(), m_v[k].data(), 96);
    memcpy(P.final_vkey[0].data(), vkey.a.data(), 192); memcpy(P.final_vkey[1].data(), vkey.b.data(), 192);
    memcpy(P.final_wkey[0].data(), wkey.a.data(), 96); memcpy(P.final_wkey[1].data(), wkey.b.data(), 96);
    // ---- prove_tipp_mipp (prover.rs:156-206): the KZG openings of the final keys ----
    std::reverse(challenges.begin(), challenges.end()); std::reverse(challenges_inv.begin(), challenges_inv.end());
    const Fr r_inverse = r.inv();                 // r_shift = r_vec[1] = r
    tr.fr("kzg-challenge", challenges[0]);
    tr.g2("vkey0", P.final_vkey[0].data()); tr.g2("vkey1", P.final_vkey[1].data());
    tr.g1("wkey0", P.final_wkey[0].data()); tr.g1("wkey1", P.final_wkey[1].data());
    const Fr z = tr.challenge("z-challenge");
    par({[&] {                                   // prove_commitment_v (kzg.rs:294-312)
             kzg_opening(true, srs->h_alpha_powers_table, srs->h_beta_powers_table, n, poly_coefficients(challenges_inv, Fr::one()),
                         poly_eval_product_form(challenges_inv, z, Fr::one()), z, P.vkey_opening[0].data(), P.vkey_opening[1].data());
         },
         [&] {                                   // prove_commitment_w (kzg.rs:314-343): f_w(X) = X^n f(X); only the dropped remainder depends on z^n
             const Frs f = poly_coefficients(challenges, r_inverse);
             Frs fw(f.size(), Fr::zero()); fw.insert(fw.end(), f.begin(), f.end());
             const Fr fwz = poly_eval_product_form(challenges, z, r_inverse) * z.pow(2 * n);
             kzg_opening(false, srs->g_alpha_powers_table, srs->g_beta_powers_table, 2 * n, fw, fwz, z, P.wkey_opening[0].data(), P.wkey_opening[1].data());
         }});
}

// ---- verifier -------------------------------------------------------------------------------------------------------------------------------
// RandomizedPairingChecker in its lazy mode (utils/src/randomized_pairing_check.rs:116-138,204-214): equation k scaled by random^k, the G1 scalings
// of everything queued in one launch, one multi_miller_loop, one final exponentiation, the targets as one GT multi-exponentiation
struct Checker {
    Fr random, cur = Fr::one();
    Vec pts, qs, sc; std::vector<Gt> targets; Frs target_exp;
    void add(const std::vector<const W *> &a, const std::vector<const W *> &b, const Gt &out) {     // prod e(a_i, b_i) == out
        need(a.size() == b.size());
        W m[4]; cur.canon(m);
        for (size_t i = 0; i < a.size(); i++) { pts.insert(pts.end(), a[i], a[i] + 12); qs.insert(qs.end(), b[i], b[i] + 24); sc.insert(sc.end(), m, m + 4); }
        if (out != gt_one()) { targets.push_back(out); target_exp.push_back(cur); }     // a target of one (the KZG checks) contributes nothing
        cur = cur * random;
    }
    bool verify() {
        Gt right = gt_one(), left = gt_one();
        const size_t n = pts.size() / 12;
        par({[&] { if (!targets.empty()) { const Vec e = canon_words(target_exp); ck(dgpu_fp12_multi_pow(targets[0].data(), e.data(), targets.size(), right.data())); } },   // host cores
             [&] {                                                                                                                                                              // the device
                 if (!n) return;
                 // the scalings beside the line chain, in one call (a pair with an identity member, before or after the scaling, contributes one)
                 ck(dgpu_multi_miller_loop_scaled(pts.data(), sc.data(), 4, qs.data(), nullptr, n, nullptr, nullptr, nullptr, 0, left.data()));
             }});
        Gt gt;
        const int32_t rc = dgpu_final_exponentiation(left.data(), gt.data());
        if (rc == DGPU_E_ZERO) return false;
        ck(rc);
        return gt == right;
    }
};

struct VerifyIn {
    const dgpu_snarkpack_verifier_srs *srs; const dgpu_groth16_vk *vk;
    const W *pub; size_t l; const W *d_list; int variant;             // 0 Groth16, 1 LegoGroth16 (MIPP for D), 2 LegoGroth16 proofs under the Groth16 aggregator + the list of d
};

void verify(const VerifyIn &in, const Proof &P, const Fr &random, const Tr &tr, bool validate_gt, bool validate_points) {
    const int nm = P.nm; const size_t n = P.n, L = log2_exact(n), l = in.l;
    const bool with_d = in.variant == 1;
    if (validate_points) {                        // the other half of Validate::Yes: every G1 / G2 member on its curve and in the prime-order subgroup
        std::vector<const W *> g1s{P.final_a.data(), P.final_wkey[0].data(), P.final_wkey[1].data(), P.wkey_opening[0].data(), P.wkey_opening[1].data()};
        std::vector<const W *> g2s{P.final_b.data(), P.final_vkey[0].data(), P.final_vkey[1].data(), P.vkey_opening[0].data(), P.vkey_opening[1].data()};
        for (int k = 0; k < nm; k++) { g1s.push_back(P.m[k].z.data()); g1s.push_back(P.m[k].final_.data()); for (auto &z : P.m[k].zs) { g1s.push_back(z[0].data()); g1s.push_back(z[1].data()); } }
        if (in.variant == 2) for (size_t i = 0; i < n; i++) g1s.push_back(in.d_list + 12 * i);
        std::vector<uint8_t> bad(g1s.size() + g2s.size(), 0);
        ck(dock::par_run(bad.size(), [&](size_t i) -> int32_t { bad[i] = i < g1s.size() ? !dock::g1_words_valid(g1s[i]) : !dock::g2_words_valid(g2s[i - g1s.size()]); return DGPU_OK; }));
        for (uint8_t b : bad) if (b) throw Reject{};
    }
    // (verifier.rs:50-64) public inputs: a rectangle of n rows; the key has to cover them
    if (with_d || in.variant == 2) need(l + 1 <= in.vk->gamma_abc_len); else need(l + 1 == in.vk->gamma_abc_len);
    if (validate_gt) {                            // what CanonicalDeserialize with Validate::Yes does for a proof that arrives as bytes: f^r == 1 for every GT element
        std::vector<const Gt *> all{&P.com_ab.t, &P.com_ab.u, &P.z_ab};
        for (int k = 0; k < nm; k++) { all.push_back(&P.m[k].com.t); all.push_back(&P.m[k].com.u); }
        for (auto &c : P.comms_ab) for (int s = 0; s < 2; s++) { all.push_back(&c[s].t); all.push_back(&c[s].u); }
        for (auto &z : P.zs_ab) { all.push_back(&z[0]); all.push_back(&z[1]); }
        for (int k = 0; k < nm; k++) for (auto &c : P.m[k].comms) for (int s = 0; s < 2; s++) { all.push_back(&c[s].t); all.push_back(&c[s].u); }
        std::vector<W> flat(72 * all.size()); std::vector<uint8_t> okv(all.size());
        for (size_t i = 0; i < all.size(); i++) memcpy(&flat[72 * i], all[i]->data(), 576);
        ck(dgpu_gt_in_subgroup(flat.data(), all.size(), okv.data()));
        for (uint8_t o : okv) if (!o) throw Reject{};
    }
    tr.pc("AB-commitment", P.com_ab.t, P.com_ab.u);
    for (int k = 0; k < nm; k++) tr.pc(labels(k).commitment.c_str(), P.m[k].com.t, P.m[k].com.u);
    const Fr r = tr.challenge("r-random-fiatshamir");
    Checker chk; chk.random = random;
    // ---- gipa_verify_tipp_mipp (groth16/verifier.rs:194-400): replay the challenges, fold T, U, Z with them ----
    tr.gt("inner-product-ab", P.z_ab);
    for (int k = 0; k < nm; k++) tr.g1(labels(k).comm.c_str(), P.m[k].z.data());
    Fr c_inv = tr.challenge("first-challenge"), ch = c_inv.inv();
    Frs challenges, challenges_inv;
    for (size_t i = 0; i < L; i++) {
        if (i > 0) {
            tr.fr("c_inv", c_inv);
            tr.gt("zab_l", P.zs_ab[i][0]); tr.gt("zab_r", P.zs_ab[i][1]);
            for (int k = 0; k < nm; k++) { const Labels lb = labels(k); tr.g1(lb.z_l.c_str(), P.m[k].zs[i][0].data()); tr.g1(lb.z_r.c_str(), P.m[k].zs[i][1].data()); }
            tr.pc("tab_l", P.comms_ab[i][0].t, P.comms_ab[i][0].u); tr.pc("tab_r", P.comms_ab[i][1].t, P.comms_ab[i][1].u);
            for (int k = 0; k < nm; k++) {
                const Labels lb = labels(k);
                tr.pc(lb.tu_l.c_str(), P.m[k].comms[i][0].t, P.m[k].comms[i][0].u); tr.pc(lb.tu_r.c_str(), P.m[k].comms[i][1].t, P.m[k].comms[i][1].u);
            }
            c_inv = tr.challenge("challenge_i"); ch = c_inv.inv();
        }
        challenges.push_back(ch); challenges_inv.push_back(c_inv);
    }
    Frs exps{Fr::one()};
    for (size_t i = 0; i < L; i++) { exps.push_back(challenges[i]); exps.push_back(challenges_inv[i]); }
    const Vec exps_w = canon_words(exps);
    // z_k + sum (c z_l + c^-1 z_r)  (:262-270); T, U, Z folded with the challenges (:272-370): left entries to the challenge, right entries to its inverse
    Vec zfold[2]; Gt tab, uab, zab, tk[2], uk[2];
    {
        std::vector<std::function<void()>> th;
        for (int k = 0; k < nm; k++) th.push_back([&, k] {
            Vec pts(P.m[k].z);
            for (auto &z : P.m[k].zs) { pts.insert(pts.end(), z[0].begin(), z[0].end()); pts.insert(pts.end(), z[1].begin(), z[1].end()); }
            zfold[k] = msm(false, pts.data(), 2 * L + 1, exps);
        });
        auto gt_fold = [&](Gt &out, const Gt &first, std::function<const Gt &(size_t, int)> at) {
            std::vector<Gt> bases{first};
            for (size_t i = 0; i < L; i++) { bases.push_back(at(i, 0)); bases.push_back(at(i, 1)); }
            ck(dgpu_fp12_multi_pow(bases[0].data(), exps_w.data(), bases.size(), out.data()));
        };
        th.push_back([&] { gt_fold(tab, P.com_ab.t, [&](size_t i, int s) -> const Gt & { return P.comms_ab[i][s].t; }); });
        th.push_back([&] { gt_fold(uab, P.com_ab.u, [&](size_t i, int s) -> const Gt & { return P.comms_ab[i][s].u; }); });
        th.push_back([&] { gt_fold(zab, P.z_ab, [&](size_t i, int s) -> const Gt & { return P.zs_ab[i][s]; }); });
        for (int k = 0; k < nm; k++) {
            th.push_back([&, k] { gt_fold(tk[k], P.m[k].com.t, [&, k](size_t i, int s) -> const Gt & { return P.m[k].comms[i][s].t; }); });
            th.push_back([&, k] { gt_fold(uk[k], P.m[k].com.u, [&, k](size_t i, int s) -> const Gt & { return P.m[k].comms[i][s].u; }); });
        }
        par(th);
    }
    std::reverse(challenges.begin(), challenges.end()); std::reverse(challenges_inv.begin(), challenges_inv.end());
    const Fr final_r = poly_eval_product_form(challenges_inv, r, Fr::one());
    // ---- verify_tipp_mipp (groth16/verifier.rs:102-192) ----
    tr.fr("kzg-challenge", challenges[0]);
    tr.g2("vkey0", P.final_vkey[0].data()); tr.g2("vkey1", P.final_vkey[1].data());
    tr.g1("wkey0", P.final_wkey[0].data()); tr.g1("wkey1", P.final_wkey[1].data());
    const Fr z = tr.challenge("z-challenge");
    const dgpu_snarkpack_verifier_srs &S = *in.srs;
    const Gt one = gt_one();
    // keep every queued operand alive until chk.verify()
    std::vector<Vec> hold; hold.reserve(64);
    auto keep = [&](Vec v) -> const W * { hold.push_back(std::move(v)); return hold.back().data(); };
    auto two = [&](bool g2, const W *p0, const W *p1, const Fr &s1) {       // p0 + s1 p1
        const int aw = aw_of(g2); Vec pts(p0, p0 + aw); pts.insert(pts.end(), p1, p1 + aw);
        return msm(g2, pts.data(), 2, Frs{Fr::one(), s1});
    };
    {   // verify_kzg_v (kzg.rs:30-76): e(-g, C_f - y h) e(v - x g, pi) == 1 for both halves of the final v key; verify_kzg_w (kzg.rs:78-125): the same with
        // the groups exchanged.  The eight two-term combinations are independent host computations (0.15 / 0.5 ms each in G1 / G2): side by side
        const Fr y = poly_eval_product_form(challenges_inv, z, Fr::one());
        const Fr fwz = poly_eval_product_form(challenges, z, r.inv()) * z.pow(S.n);
        const W *cfv[2] = {P.final_vkey[0].data(), P.final_vkey[1].data()}, *vk[2] = {S.g_alpha, S.g_beta}, *piv[2] = {P.vkey_opening[0].data(), P.vkey_opening[1].data()};
        const W *cfw[2] = {P.final_wkey[0].data(), P.final_wkey[1].data()}, *wk[2] = {S.h_alpha, S.h_beta}, *piw[2] = {P.wkey_opening[0].data(), P.wkey_opening[1].data()};
        Vec vb[2], vc[2], wa[2], wd[2];
        std::vector<std::function<void()>> th;
        for (int k = 0; k < 2; k++) {
            th.push_back([&, k] { vb[k] = two(true, cfv[k], S.h, y.neg()); });       // C_f - y h
            th.push_back([&, k] { vc[k] = two(false, vk[k], S.g, z.neg()); });       // v - x g
            th.push_back([&, k] { wa[k] = two(false, cfw[k], S.g, fwz.neg()); });    // C_f - y g
            th.push_back([&, k] { wd[k] = two(true, wk[k], S.h, z.neg()); });        // w - x h
        }
        par(th);
        const Vec ng = neg_point(false, S.g), nh = neg_point(true, S.h);
        for (int k = 0; k < 2; k++) chk.add({ng.data(), vc[k].data()}, {vb[k].data(), piv[k]}, one);
        for (int k = 0; k < 2; k++) chk.add({wa[k].data(), piw[k]}, {nh.data(), wd[k].data()}, one);
    }
    const W *fa = P.final_a.data(), *fb = P.final_b.data(), *v0 = P.final_vkey[0].data(), *v1 = P.final_vkey[1].data(), *w0 = P.final_wkey[0].data(), *w1 = P.final_wkey[1].data();
    chk.add({fa}, {fb}, zab);
    chk.add({fa, w0}, {v0, fb}, tab);
    chk.add({fa, w1}, {v1, fb}, uab);
    bool bad_final_z = false;
    for (int k = 0; k < nm; k++) {               // MIPP: Z == final^final_r, T = e(final, v1), U = e(final, v2)
        const W *fk = P.m[k].final_.data();
        const Vec final_z = msm(false, fk, 1, Frs{final_r});
        chk.add({fk}, {v0}, tk[k]);
        chk.add({fk}, {v1}, uk[k]);
        if (final_z != zfold[k]) bad_final_z = true;
    }
    if (bad_final_z) throw Reject{};             // "tipp verify: INVALID final_z check"
    // ---- final_verification_check (utils.rs:218-265; using_groth16.rs:95-126 folds the d_i itself) ----
    const Frs r_powers = powers(r, n);
    Fr r_sum = Fr::zero(); for (auto &x : r_powers) r_sum = r_sum + x;
    Frs sc; Vec pts;
    if (in.variant == 2) { pts.assign(in.d_list, in.d_list + 12 * n); sc = r_powers; }
    pts.insert(pts.end(), in.vk->gamma_abc_g1, in.vk->gamma_abc_g1 + 12 * (l + 1));
    sc.push_back(r_sum);
    for (size_t i = 0; i < l; i++) {             // aggregate_public_inputs (utils.rs:120-158): sum_j r^j x_{j,i}
        Fr s = Fr::zero();
        for (size_t jj = 0; jj < n; jj++) s = s + Fr::from_canon(in.pub + 4 * (jj * l + i)) * r_powers[jj];
        sc.push_back(s);
    }
    const W *inputs = keep(msm(false, pts.data(), sc.size(), sc));
    const W *alpha_r = keep(msm(false, in.vk->alpha_g1, 1, Frs{r_sum}));
    std::vector<const W *> s1, s2;
    if (with_d) { s1.push_back(P.m[1].z.data()); s2.push_back(in.vk->gamma_g2); }
    s1.insert(s1.end(), {alpha_r, inputs, P.m[0].z.data()});
    s2.insert(s2.end(), {in.vk->beta_g2, in.vk->gamma_g2, in.vk->delta_g2});
    chk.add(s1, s2, P.z_ab);
    if (!chk.verify()) throw Reject{};
}

template <class F> int32_t guarded(F f) noexcept {
    const bool keep = dock::tl_no_min; dock::tl_no_min = true;
    int32_t rc;
    try { rc = f(); }
    catch (const Fail &e) { rc = e.rc; }
    catch (const std::bad_alloc &) { rc = DGPU_E_OOM; }
    catch (...) { rc = DGPU_E_HIP; }
    dock::tl_no_min = keep;
    return rc;
}
}  // namespace

extern "C" size_t dgpu_snarkpack_proof_words(size_t n, int32_t with_d) {
    if (n < 2 || (n & (n - 1)) || n > DGPU_SNARKPACK_MAX_SRS_SIZE) return 0;
    return proof_words(n, with_d ? 2 : 1);
}

extern "C" int32_t dgpu_snarkpack_aggregate(const dgpu_snarkpack_prover_srs *srs, const uint64_t *a, const uint64_t *b, const uint64_t *c, const uint64_t *d,
                                            size_t n, const dgpu_transcript *transcript, uint64_t *proof, size_t cap_words, size_t *len_words) {
    return guarded([&]() -> int32_t {
        if (!srs || !a || !b || !c || !transcript || !transcript->append_message || !transcript->challenge_scalar || !proof || !len_words) return DGPU_E_BADARG;
        if (n < 2 || (n & (n - 1)) || n > DGPU_SNARKPACK_MAX_SRS_SIZE) return DGPU_E_BADARG;          // "invalid proof size" (prover.rs:52-60)
        if (!srs->vkey_a || !srs->vkey_b || !srs->wkey_a || !srs->wkey_b || !srs->g_alpha_powers_table || !srs->g_beta_powers_table || !srs->h_alpha_powers_table || !srs->h_beta_powers_table) return DGPU_E_BADARG;
        const int nm = d ? 2 : 1;
        const size_t need_words = proof_words(n, nm);
        *len_words = need_words;
        if (cap_words < need_words) return DGPU_E_LENGTH;
        const W *pv[2] = {c, d};
        Proof P;
        aggregate(srs, a, b, pv, nm, n, Tr{transcript}, P);
        write_proof(P, proof);
        return DGPU_OK;
    });
}

extern "C" int32_t dgpu_snarkpack_verify(const dgpu_snarkpack_verifier_srs *srs, const dgpu_groth16_vk *vk, const uint64_t *public_inputs, size_t n_rows, size_t inputs_per_proof,
                                         const uint64_t *proof, size_t len_words, int32_t variant, const uint64_t *d_list, const uint64_t random[4],
                                         const dgpu_transcript *transcript, int32_t flags, int32_t *ok) {
    return guarded([&]() -> int32_t {
        if (ok) *ok = 0;
        if (!srs || !vk || !proof || !random || !transcript || !transcript->append_message || !transcript->challenge_scalar || !ok) return DGPU_E_BADARG;
        // the batching scalar of the pairing checker: the reference draws it from an RNG inside RandomizedPairingChecker; here it is the caller's, and 0
        // (mod r) would scale every equation after the first by 0^k = 0 — the KZG, T / U / Z and final Groth16 checks would drop out of the product
        { W c[4]; Fr::from_canon(random).canon(c); if (!(c[0] | c[1] | c[2] | c[3])) return DGPU_E_BADARG; }
        if (variant < 0 || variant > 2 || (variant == 2) != (d_list != nullptr) || (inputs_per_proof && !public_inputs)) return DGPU_E_BADARG;
        if (!srs->g || !srs->h || !srs->g_alpha || !srs->g_beta || !srs->h_alpha || !srs->h_beta) return DGPU_E_BADARG;
        if (!vk->alpha_g1 || !vk->beta_g2 || !vk->gamma_g2 || !vk->delta_g2 || !vk->gamma_abc_g1) return DGPU_E_BADARG;
        Proof P;
        read_proof(P, proof, len_words, variant == 1 ? 2 : 1);
        if (n_rows != P.n) return DGPU_E_BADARG;                 // "public inputs len != number of proofs"
        try {
            verify(VerifyIn{srs, vk, public_inputs, inputs_per_proof, d_list, variant}, P, Fr::from_canon(random), Tr{transcript}, (flags & DGPU_SNARKPACK_VALIDATE_GT) != 0, (flags & DGPU_SNARKPACK_VALIDATE_POINTS) != 0);
            *ok = 1;
        } catch (const Reject &) { *ok = 0; }
        return DGPU_OK;
    });
}

// ---- many LegoGroth16 proofs of one circuit in ONE call: the classical Groth16 batch check ------------------------------------------------
// The reference batches through RandomizedPairingChecker (utils/src/randomized_pairing_check.rs:116-138,204-214; proof_system/src/verifier.rs hands
// every statement's three pairs to one lazy checker): 3 N pairs, N GT powers.  All proofs of ONE verifying key share -delta and -gamma, so the pairs can
// be merged BEFORE the pairing: with m_i = random^i
//     prod_i e(m_i A_i, B_i) * e(sum_i m_i C_i, -delta) * e(sum_i m_i (gamma_abc_0 + sum_j x_ij gamma_abc_j + d_i), -gamma) == e(alpha, beta)^(sum_i m_i)
// i.e. N scalings, two variable-base MSMs (N and N + k + 1 terms: this library's hot path), ONE Miller loop over N + 2 pairs — two of them on the
// prepared key — one final exponentiation and one GT power.  Accepts exactly the batches whose every proof dgpu_legogroth16_verify accepts, up to the
// 2^-255 soundness error of the random combination (the checker's own).  The three device pieces and the GT power run side by side from host threads.
extern "C" int32_t dgpu_legogroth16_verify_batch(const uint64_t alpha_beta_gt[72], const uint64_t *delta_neg_pc, const uint64_t *gamma_neg_pc, const uint64_t *gamma_abc_g1, size_t gamma_abc_len,
                                                 const uint64_t *proofs_a, const uint64_t *proofs_b, const uint64_t *proofs_c, const uint64_t *proofs_d, size_t n,
                                                 const uint64_t *public_inputs, size_t n_pub, int32_t montgomery, const uint64_t random[4], int32_t *ok) {
    return guarded([&]() -> int32_t {
        if (ok) *ok = 0;
        if (!ok || !alpha_beta_gt || !delta_neg_pc || !gamma_neg_pc || !gamma_abc_g1 || !random) return DGPU_E_BADARG;
        if (n && (!proofs_a || !proofs_b || !proofs_c || !proofs_d)) return DGPU_E_BADARG;
        if (n_pub + 1 > gamma_abc_len || (n && n_pub && !public_inputs)) return DGPU_E_BADARG;              // MalformedVerifyingKey (verifier.rs:101-109)
        if (n == 0) { *ok = 1; return DGPU_OK; }
        const Fr rnd = Fr::from_canon(random);
        if (rnd.is_zero()) return DGPU_E_BADARG;                                                             // (every proof after the first would drop out of the check)
        const Frs m = powers(rnd, n);
        Fr m_sum = Fr::zero(); for (auto &x : m) m_sum = m_sum + x;
        const Vec m_words = canon_words(m);
        // scalars of the d-side MSM: [sum m_i] for gamma_abc_0, [sum_i m_i x_ij] for gamma_abc_j, m_i for d_i
        Frs d_sc(1 + n_pub + n);
        d_sc[0] = m_sum;
        for (size_t j = 0; j < n_pub; j++) {
            Fr sj = Fr::zero();
            for (size_t i = 0; i < n; i++) {
                const W *w = public_inputs + 4 * (i * n_pub + j);
                Fr x; if (montgomery) memcpy(x.v.l, w, 32); else x = Fr::from_canon(w);
                sj = sj + m[i] * x;
            }
            d_sc[1 + j] = sj;
        }
        for (size_t i = 0; i < n; i++) d_sc[1 + n_pub + i] = m[i];
        Vec d_pts(gamma_abc_g1, gamma_abc_g1 + 12 * (1 + n_pub));
        d_pts.insert(d_pts.end(), proofs_d, proofs_d + 12 * n);
        Vec c_sum, d_sum;
        Gt rhs;
        std::vector<uint8_t> skip_prep(2);
        Vec co(delta_neg_pc, delta_neg_pc + DGPU_G2_PREPARED_WORDS); co.insert(co.end(), gamma_neg_pc, gamma_neg_pc + DGPU_G2_PREPARED_WORDS);
        // Three independent chains, side by side (round 6; the kernel trace of round 5's order — MSMs, THEN the loop — showed the longest chain of the call, the
        // 0.9-ms scalings, waiting 0.7 ms for two MSMs it does not depend on: profiles/r06_timeline_batch.txt):
        //   1. prod e([m_i] A_i, B_i): ONE Miller loop over the n affine pairs, the scalings running beside the chain of the B_i inside the call
        //   2. the two MSMs, then the Miller loop of the two prepared pairs e(sum m_i C_i, -delta) e(d, -gamma) they feed
        //   3. the GT power of the right-hand side
        // A Miller loop's output is the product of its pairs' own outputs (the squarings distribute over the product of the lines, the final conjugation too),
        // so f = f_1 f_2 limb for limb what the one loop over all n + 2 pairs returns (dgpu_multi_miller_loop_sharded multiplies partial outputs the same way).
        Gt f, f1, f2, gt;
        par({
            [&] { ck(dgpu_multi_miller_loop_scaled(proofs_a, m_words.data(), 4, proofs_b, nullptr, n, nullptr, nullptr, nullptr, 0, f1.data())); },
            [&] {
                par({ [&] { c_sum = msm(false, proofs_c, n, m); }, [&] { d_sum = msm(false, d_pts.data(), d_sc.size(), d_sc); } });
                Vec p_prep(c_sum); p_prep.insert(p_prep.end(), d_sum.begin(), d_sum.end());
                skip_prep[0] = is_id(c_sum.data(), 12); skip_prep[1] = is_id(d_sum.data(), 12);
                ck(dgpu_multi_miller_loop_mixed(nullptr, nullptr, nullptr, 0, p_prep.data(), co.data(), skip_prep.data(), 2, f2.data()));
            },
            [&] { W e[4]; m_sum.canon(e); ck(dgpu_fp12_pow(alpha_beta_gt, e, rhs.data())); },
        });
        ck(dgpu_fp12_mul(f1.data(), f2.data(), f.data()));
        ck(dgpu_final_exponentiation(f.data(), gt.data()));                                                   // DGPU_E_ZERO: UnexpectedIdentity
        *ok = gt == rhs ? 1 : 0;
        return DGPU_OK;
    });
}
