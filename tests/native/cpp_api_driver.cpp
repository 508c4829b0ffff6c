// tests/native/cpp_api_driver.cpp — GPU parity driver for the C++ host mirror (include/dock_gpu.hpp), written like the reference's
// own tests (utils/src/msm.rs:186-231: msm == sum of mul_bigint, multiply_many[i] == g * s_i; utils/src/randomized_pairing_check.rs
// tests: pairing products).  The CPU oracle (oracle/liboracle.so, test infrastructure) is the checker.  Exit code 0 = all equal.
#include <cstdio>
#include <cstdlib>
#include "../../include/dock_gpu.hpp"

extern "C" {   // oracle/oracle.c
void orc_rand_scalars(uint64_t seed, size_t n, uint64_t *out);
void orc_fr_to_mont(const uint64_t *a, uint64_t *out, size_t n);
void orc_g1_generator(uint64_t *out); void orc_g2_generator(uint64_t *out);
void orc_g1_gen_seq(const uint64_t *k0, const uint64_t *d, size_t n, int threads, uint64_t *out);
void orc_g2_gen_seq(const uint64_t *k0, const uint64_t *d, size_t n, int threads, uint64_t *out);
void orc_g1_msm(const uint64_t *b, const uint8_t *inf, const uint64_t *s, size_t n, int threads, uint64_t *out);
void orc_g2_msm(const uint64_t *b, const uint8_t *inf, const uint64_t *s, size_t n, int threads, uint64_t *out);
int orc_g1_to_affine(const uint64_t *jac, uint64_t *out); int orc_g2_to_affine(const uint64_t *jac, uint64_t *out);
void orc_g1_mul(const uint64_t *base, int inf, const uint64_t *k, uint64_t *out); void orc_g2_mul(const uint64_t *base, int inf, const uint64_t *k, uint64_t *out);
void orc_multi_miller_loop(const uint64_t *p, const uint64_t *q, const uint8_t *skip, size_t n, int threads, uint64_t *out);
int orc_final_exponentiation(const uint64_t *f, uint64_t *out);
}
using namespace dock_gpu;
static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

template <class G> std::vector<typename G::Affine> affine_vec(const std::vector<uint64_t> &xy, size_t n) { return affine_from_abi<G>(xy, std::vector<uint8_t>(n, 0)); }
template <class G> bool same_point(const typename G::Projective &got, const uint64_t *oracle_jac, int (*to_aff)(const uint64_t *, uint64_t *)) {
    uint64_t a[G::AW], b[G::AW], j[G::AW * 3 / 2];
    std::memcpy(j, &got.x, G::AW * 4); std::memcpy(j + G::AW / 2, &got.y, G::AW * 4); std::memcpy(j + G::AW, &got.z, G::AW * 4);
    int i1 = to_aff(j, a), i2 = to_aff(oracle_jac, b);
    return i1 == i2 && (i1 || std::memcmp(a, b, sizeof a) == 0);
}

int main() {
    init(0);
    const size_t n = 3000;
    std::vector<uint64_t> k0(4), d(4), sc(n * 4), scm(n * 4), b1(n * 12), b2(n * 24);
    orc_rand_scalars(1, 1, k0.data()); orc_rand_scalars(2, 1, d.data()); orc_rand_scalars(3, n, sc.data());
    orc_fr_to_mont(sc.data(), scm.data(), n);
    orc_g1_gen_seq(k0.data(), d.data(), n, 8, b1.data()); orc_g2_gen_seq(d.data(), k0.data(), n, 8, b2.data());
    auto P1 = affine_vec<G1>(b1, n); auto P2 = affine_vec<G2>(b2, n);
    P1[5].infinity = true; P2[7].infinity = true;                         // identity bases are allowed
    std::vector<uint8_t> inf1(n, 0), inf2(n, 0); inf1[5] = 1; inf2[7] = 1;
    std::vector<BigInt256> big(n); std::vector<Fr> fr(n);
    for (size_t i = 0; i < n; i++) { std::memcpy(big[i].data(), &sc[4 * i], 32); std::memcpy(fr[i].mont.data(), &scm[4 * i], 32); }

    // the MSM entry points read the caller's array of structs in place (dgpu_msm_*_strided): Affine { x, y, infinity } is 104 / 200 bytes,
    // the layout of ark-ec's G1Affine / G2Affine
    static_assert(sizeof(G1::Affine) == 104 && sizeof(G2::Affine) == 200, "Affine layout");
    uint64_t e1[18], e2[36];
    orc_g1_msm(b1.data(), inf1.data(), sc.data(), n, 8, e1); orc_g2_msm(b2.data(), inf2.data(), sc.data(), n, 8, e2);
    {   // strided == packed, limb for limb; an identity flag inside the struct is honoured; DGPU_NO_INF_OFF ignores it
        uint64_t a1[18], a2[18], a3[18];
        EXPECT(dgpu_msm_g1(b1.data(), inf1.data(), sc.data(), n, a1) == DGPU_OK);
        EXPECT(dgpu_msm_g1_strided(P1.data(), sizeof(G1::Affine), offsetof(G1::Affine, x), offsetof(G1::Affine, y), offsetof(G1::Affine, infinity), sc.data(), n, 0, a2) == DGPU_OK && std::memcmp(a1, a2, sizeof a1) == 0);
        EXPECT(dgpu_msm_g1(b1.data(), nullptr, sc.data(), n, a1) == DGPU_OK);
        EXPECT(dgpu_msm_g1_strided(P1.data(), sizeof(G1::Affine), offsetof(G1::Affine, x), offsetof(G1::Affine, y), DGPU_NO_INF_OFF, sc.data(), n, 0, a3) == DGPU_OK && std::memcmp(a1, a3, sizeof a1) == 0);
        EXPECT(dgpu_msm_g1_strided(P1.data(), 100, 0, 48, 96, sc.data(), n, 0, a3) == DGPU_E_BADARG);          // stride not a multiple of 8
    }
    EXPECT(same_point<G1>(VariableBaseMSM<G1>::msm_bigint(P1, big), e1, orc_g1_to_affine));
    EXPECT(same_point<G1>(VariableBaseMSM<G1>::msm_unchecked(P1, fr), e1, orc_g1_to_affine));
    EXPECT(same_point<G2>(VariableBaseMSM<G2>::msm_bigint(P2, big), e2, orc_g2_to_affine));
    EXPECT(same_point<G1>(Pairs<G1>{P1, fr}.msm(), e1, orc_g1_to_affine));
    // checked msm: Err(min_len) on mismatch; truncation for msm_bigint
    auto shorter = std::vector<Fr>(fr.begin(), fr.begin() + 100);
    auto r = VariableBaseMSM<G1>::msm(P1, shorter);
    EXPECT(!r.first.has_value() && r.second == 100);
    orc_g1_msm(b1.data(), inf1.data(), sc.data(), 100, 8, e1);
    EXPECT(same_point<G1>(VariableBaseMSM<G1>::msm_bigint(P1, std::vector<BigInt256>(big.begin(), big.begin() + 100)), e1, orc_g1_to_affine));
    EXPECT(VariableBaseMSM<G1>::msm_bigint({}, {}).is_zero());            // n = 0 -> identity
    // resident query, &query[1..]
    DeviceBases<G1> q(P1);
    orc_g1_msm(b1.data() + 12, inf1.data() + 1, sc.data(), n - 1, 8, e1);
    EXPECT(same_point<G1>(q.msm_bigint(big, 1), e1, orc_g1_to_affine));
    // WindowTable: multiply_many[i] == g * s_i
    G1::Affine g; g.infinity = false; uint64_t gen[12]; orc_g1_generator(gen); std::memcpy(&g.x, gen, 48); std::memcpy(&g.y, gen + 6, 48);
    auto prod = multiply_field_elems_with_same_group_elem<G1>(g, std::vector<Fr>(fr.begin(), fr.begin() + 50));
    for (size_t i = 0; i < 50; i++) {
        uint64_t j[18], a[12]; orc_g1_mul(gen, 0, &sc[4 * i], j); orc_g1_to_affine(j, a);
        EXPECT(!prod[i].infinity && std::memcmp(&prod[i].x, a, 48) == 0 && std::memcmp(&prod[i].y, a + 6, 48) == 0);
    }
    // pairing: raw Miller-loop output and GT equal to the oracle's; unequal lengths rejected
    const size_t np = 33;
    std::vector<G1::Affine> a(P1.begin() + 10, P1.begin() + 10 + np); std::vector<G2::Affine> b(P2.begin() + 10, P2.begin() + 10 + np);
    Fq12 ml = multi_miller_loop(a, b), eml{}, egt{};
    orc_multi_miller_loop(b1.data() + 120, b2.data() + 240, nullptr, np, 4, eml.data());
    EXPECT(ml == eml);
    orc_final_exponentiation(eml.data(), egt.data());
    EXPECT(multi_pairing(a, b) == egt);
    bool threw = false; try { a.pop_back(); multi_miller_loop(a, b); } catch (const Error &e) { threw = e.code == DGPU_E_LENGTH; }
    EXPECT(threw);
    EXPECT(!final_exponentiation(Fq12{}).has_value());                    // zero -> None
    // prepared and mixed operands straight through the C ABI (verifier.rs:69-76 shape), the host lincomb, and the per-key table
    {
        a = std::vector<G1::Affine>(P1.begin() + 10, P1.begin() + 10 + np);
        std::vector<uint64_t> co(np * DGPU_G2_PREPARED_WORDS); std::vector<uint8_t> cinf(np);
        EXPECT(dgpu_g2_prepare(b2.data() + 240, nullptr, np, co.data(), cinf.data()) == DGPU_OK);
        Fq12 f1{}, f2{};
        EXPECT(dgpu_multi_miller_loop_prepared(b1.data() + 120, co.data(), cinf.data(), np, f1.data()) == DGPU_OK && f1 == eml);
        const size_t na = 5;                  // first five pairs affine, the rest prepared
        EXPECT(dgpu_multi_miller_loop_mixed(b1.data() + 120, b2.data() + 240, nullptr, na, b1.data() + 120 + 12 * na, co.data() + na * DGPU_G2_PREPARED_WORDS, cinf.data() + na, np - na, f2.data()) == DGPU_OK && f2 == eml);
        EXPECT(dgpu_multi_miller_loop_mixed(nullptr, nullptr, nullptr, 0, b1.data() + 120, co.data(), cinf.data(), np, f2.data()) == DGPU_OK && f2 == eml);
        EXPECT(dgpu_multi_miller_loop_mixed(b1.data() + 120, b2.data() + 240, nullptr, np, nullptr, nullptr, nullptr, 0, f2.data()) == DGPU_OK && f2 == eml);
        EXPECT(dgpu_multi_miller_loop_mixed(nullptr, nullptr, nullptr, 3, nullptr, nullptr, nullptr, 0, f2.data()) == DGPU_E_BADARG);
        // three independent products in one call (an empty one in the middle), raw and finally exponentiated
        {
            const uint64_t ends[3] = {na, na, np};
            std::vector<uint64_t> seg(3 * 72), gt(3 * 72); Fq12 s0{}, s2{}, g2{};
            EXPECT(dgpu_multi_miller_loop_segments(b1.data() + 120, b2.data() + 240, nullptr, np, ends, 3, seg.data()) == DGPU_OK);
            EXPECT(dgpu_multi_miller_loop(b1.data() + 120, b2.data() + 240, nullptr, na, s0.data()) == DGPU_OK && std::memcmp(seg.data(), s0.data(), 576) == 0);
            EXPECT(dgpu_multi_miller_loop(b1.data() + 120 + 12 * na, b2.data() + 240 + 24 * na, nullptr, np - na, s2.data()) == DGPU_OK && std::memcmp(seg.data() + 144, s2.data(), 576) == 0);
            EXPECT(dgpu_multi_pairing_segments(b1.data() + 120, b2.data() + 240, nullptr, np, ends, 3, gt.data()) == DGPU_OK);
            EXPECT(dgpu_final_exponentiation(s2.data(), g2.data()) == DGPU_OK && std::memcmp(gt.data() + 144, g2.data(), 576) == 0);
            EXPECT(std::memcmp(gt.data() + 72, seg.data() + 72, 576) == 0);          // the empty product: one, before and after
            const uint64_t bad[2] = {na, np - 1};
            EXPECT(dgpu_multi_miller_loop_segments(b1.data() + 120, b2.data() + 240, nullptr, np, bad, 2, seg.data()) == DGPU_E_BADARG);
        }
        uint64_t lc[18], el[18];
        EXPECT(dgpu_lincomb_g1(b1.data(), inf1.data(), sc.data(), 9, lc) == DGPU_OK);
        orc_g1_msm(b1.data(), inf1.data(), sc.data(), 9, 1, el);
        { uint64_t x[12], y[12]; int i1 = orc_g1_to_affine(lc, x), i2 = orc_g1_to_affine(el, y); EXPECT(i1 == i2 && std::memcmp(x, y, sizeof x) == 0); }
        EXPECT(dgpu_lincomb_g1(b1.data(), nullptr, sc.data(), DGPU_MAX_LINCOMB + 1, lc) == DGPU_E_BADARG);
        uint64_t h = 0, plain[18], tab[18];
        EXPECT(dgpu_bases_upload_g1(b1.data(), inf1.data(), n, &h) == DGPU_OK);
        EXPECT(dgpu_msm_g1_handle(h, 0, sc.data(), n, 0, plain) == DGPU_OK);
        EXPECT(dgpu_bases_precompute_g1(h, 16) == DGPU_OK);
        EXPECT(dgpu_msm_g1_handle(h, 0, sc.data(), n, 0, tab) == DGPU_OK && std::memcmp(plain, tab, sizeof plain) == 0);
        // one sort, then the MSM of the table on the sorted list (dgpu_scalars_sort / dgpu_msm_g1_sorted): the same point
        {
            uint64_t hsc = 0, srt = 0, viaSort[18]; size_t rows = 0; int32_t cbits = 0, wins = 0;
            EXPECT(dgpu_bases_table_shape(h, &rows, &cbits, &wins) == DGPU_OK && rows == n && cbits == 16 && wins == 16);
            EXPECT(dgpu_scalars_upload(sc.data(), n, 0, &hsc) == DGPU_OK);
            EXPECT(dgpu_scalars_sort(h, 0, hsc, 0, n, &srt) == DGPU_OK);
            EXPECT(dgpu_msm_g1_sorted(h, srt, 0, viaSort) == DGPU_OK && std::memcmp(plain, viaSort, sizeof plain) == 0);
            EXPECT(dgpu_msm_g1_sorted(h, hsc, 0, viaSort) == DGPU_E_BADARG);
            EXPECT(dgpu_scalars_free(srt) == DGPU_OK && dgpu_scalars_free(hsc) == DGPU_OK);
        }
        EXPECT(dgpu_bases_free(h) == DGPU_OK);
    }
    // the LegoGroth16 prover as ONE call (dgpu_legogroth16_prove through legogroth16::create_proof_with_reduction, prover.rs:153-180 -> :267-383)
    // on a synthetic key of 1000 variables, against the same equations evaluated piece by piece through OTHER entry points (one-shot MSMs
    // from host memory, host lincombs, folds); then again with every query a precomputed table (one shared partition sort inside the call)
    {
        const size_t NV = 1000, NI = 2, CW = 2, NC = 1022, D = 1024;
        std::vector<G1::Affine> qa(P1.begin(), P1.begin() + NV), qb1(P1.begin() + 1000, P1.begin() + 1000 + NV), qh(P1.begin() + 1500, P1.begin() + 1500 + D - 1), ql(P1.begin() + 2000, P1.begin() + 2000 + NV - NI - CW);
        std::vector<G2::Affine> qb2(P2.begin(), P2.begin() + NV);
        qa[5].infinity = true;                                                 // (P1[5] is already flagged; its coordinates stay: the flag must win)
        legogroth16::ProvingKey pk(qa, qb1, qb2, qh, ql);
        pk.alpha_g1 = P1[2900]; pk.beta_g1 = P1[2901]; pk.delta_g1 = P1[2902]; pk.eta_delta_inv_g1 = P1[2903]; pk.eta_gamma_inv_g1 = P1[2904];
        pk.beta_g2 = P2[2900]; pk.delta_g2 = P2[2901];
        pk.gamma_abc_g1.assign(P1.begin() + 2910, P1.begin() + 2910 + NI + CW); pk.commit_witness_count = CW;
        std::vector<uint64_t> rp(NC + 1), va(NC * 4, 0); std::vector<uint32_t> ca(NC), cb(NC), cc(NC);
        for (size_t i = 0; i < NC; i++) { rp[i] = i; ca[i] = (uint32_t)(i % NV); cb[i] = (uint32_t)((7 * i + 3) % NV); cc[i] = (uint32_t)((5 * i + 1) % NV); va[4 * i] = 1 + i % 3; }
        rp[NC] = NC;
        uint64_t circ = 0;
        EXPECT(dgpu_r1cs_upload(rp.data(), ca.data(), va.data(), NC, rp.data(), cb.data(), va.data(), NC, rp.data(), cc.data(), va.data(), NC, NV, NI, NC, 0, &circ) == DGPU_OK);
        std::vector<BigInt256> z(big.begin(), big.begin() + NV);
        z[0] = BigInt256{1, 0, 0, 0};
        const BigInt256 r{3, 0, 0, 0}, s{5, 0, 0, 0}, v{7, 0, 0, 0};
        // reference, piece by piece
        std::vector<uint64_t> h(D * 4); size_t hl = 0;
        EXPECT(dgpu_witness_map_r1cs(circ, z[0].data(), NV, 0, h.data(), nullptr, &hl) == DGPU_OK && hl == D);
        auto flat1 = [&](const G1::Affine &p, uint64_t *o) { std::memcpy(o, &p.x, 48); std::memcpy(o + 6, &p.y, 48); };
        auto msm1 = [&](const std::vector<G1::Affine> &q, size_t off, const uint64_t *sc_, size_t cnt, uint64_t *out) {
            EXPECT(dgpu_msm_g1_strided(q.data() + off, sizeof(G1::Affine), offsetof(G1::Affine, x), offsetof(G1::Affine, y), offsetof(G1::Affine, infinity), sc_, cnt, 0, out) == DGPU_OK); };
        const uint64_t one4[4] = {1, 0, 0, 0};
        auto coeff1 = [&](const std::vector<G1::Affine> &q, const G1::Affine &vkp, const BigInt256 &k, uint64_t *out) {     // k delta + q[0] + vk + msm(q[1..], z[1..])
            uint64_t parts[36], pts[36], ks[12]; uint8_t fl[3] = {pk.delta_g1.infinity, q[0].infinity, vkp.infinity};
            flat1(pk.delta_g1, pts); flat1(q[0], pts + 12); flat1(vkp, pts + 24);
            std::memcpy(ks, k.data(), 32); std::memcpy(ks + 4, one4, 32); std::memcpy(ks + 8, one4, 32);
            msm1(q, 1, z[1].data(), NV - 1, parts);
            EXPECT(dgpu_lincomb_g1(pts, fl, ks, 3, parts + 18) == DGPU_OK);
            EXPECT(dgpu_fold_g1(parts, 2, out) == DGPU_OK); };
        uint64_t eA[18], eB1[18], eB2[36], eL[18], eH[18];
        coeff1(qa, pk.alpha_g1, r, eA); coeff1(qb1, pk.beta_g1, s, eB1);
        {
            uint64_t parts[72], pts[72], ks[12];
            std::memcpy(pts, &pk.delta_g2.x, 96); std::memcpy(pts + 12, &pk.delta_g2.y, 96); std::memcpy(pts + 24, &qb2[0].x, 96); std::memcpy(pts + 36, &qb2[0].y, 96); std::memcpy(pts + 48, &pk.beta_g2.x, 96); std::memcpy(pts + 60, &pk.beta_g2.y, 96);
            std::memcpy(ks, s.data(), 32); std::memcpy(ks + 4, one4, 32); std::memcpy(ks + 8, one4, 32);
            EXPECT(dgpu_msm_g2_strided(qb2.data() + 1, sizeof(G2::Affine), offsetof(G2::Affine, x), offsetof(G2::Affine, y), offsetof(G2::Affine, infinity), z[1].data(), NV - 1, 0, parts) == DGPU_OK);
            EXPECT(dgpu_lincomb_g2(pts, nullptr, ks, 3, parts + 36) == DGPU_OK);
            EXPECT(dgpu_fold_g2(parts, 2, eB2) == DGPU_OK);
        }
        msm1(ql, 0, z[NI + CW].data(), NV - NI - CW, eL); msm1(qh, 0, h.data(), D - 1, eH);
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1) { pk.a_query.precompute(16); pk.b_g1_query.precompute(16); pk.b_g2_query.precompute(16); pk.h_query.precompute(16); pk.l_query.precompute(16); }
            legogroth16::Proof pr = legogroth16::create_proof_with_reduction(pk, circ, z, NI, r, s, v);
            EXPECT(!pr.a.infinity && std::memcmp(&pr.a.x, eA, 96) == 0);
            EXPECT(!pr.b.infinity && std::memcmp(&pr.b.x, eB2, 192) == 0);
            // C + rs delta + v eta/delta == s A + r B1 + L + H
            uint64_t lhs[36], rhs[72], pts[36], ks[12], L[18], Rr[18]; uint8_t fl[3] = {0, 0, 0};
            flat1(pr.c, pts); flat1(pk.delta_g1, pts + 12); flat1(pk.eta_delta_inv_g1, pts + 24);
            const uint64_t k15[4] = {15, 0, 0, 0};
            std::memcpy(ks, one4, 32); std::memcpy(ks + 4, k15, 32); std::memcpy(ks + 8, v.data(), 32);
            EXPECT(dgpu_lincomb_g1(pts, fl, ks, 3, L) == DGPU_OK);
            std::memcpy(pts, eA, 96); std::memcpy(pts + 12, eB1, 96);
            std::memcpy(ks, s.data(), 32); std::memcpy(ks + 4, r.data(), 32);
            EXPECT(dgpu_lincomb_g1(pts, fl, ks, 2, rhs) == DGPU_OK);
            std::memcpy(rhs + 18, eL, 144); std::memcpy(rhs + 36, eH, 144);
            EXPECT(dgpu_fold_g1(rhs, 3, Rr) == DGPU_OK);
            (void)lhs;
            EXPECT(std::memcmp(L, Rr, sizeof L) == 0);
            // D = msm(gamma_abc[NI .. NI + CW], z[NI .. NI + CW]) + v eta/gamma
            uint64_t dp[36], dk[12], eD[18];
            flat1(pk.gamma_abc_g1[NI], dp); flat1(pk.gamma_abc_g1[NI + 1], dp + 12); flat1(pk.eta_gamma_inv_g1, dp + 24);
            std::memcpy(dk, z[NI].data(), 32); std::memcpy(dk + 4, z[NI + 1].data(), 32); std::memcpy(dk + 8, v.data(), 32);
            EXPECT(dgpu_lincomb_g1(dp, fl, dk, 3, eD) == DGPU_OK);
            EXPECT(!pr.d.infinity && std::memcmp(&pr.d.x, eD, 96) == 0);
            // the same proof for a key the host HOLDS (legogroth16::create_proof_host -> dgpu_legogroth16_prove_host): vectors of Affine structs, no handle — the key's
            // first proof uploads the views for the call, the second makes them resident through the library's cache, the third runs on the resident copies; h from
            // the host (create_proof_with_assignment) and the circuit resident (create_proof_with_reduction)
            if (pass == 0) {
                legogroth16::HostProvingKey hk;
                hk.a_query = qa; hk.b_g1_query = qb1; hk.b_g2_query = qb2; hk.h_query = qh; hk.l_query = ql;
                hk.alpha_g1 = pk.alpha_g1; hk.beta_g1 = pk.beta_g1; hk.delta_g1 = pk.delta_g1; hk.eta_delta_inv_g1 = pk.eta_delta_inv_g1; hk.eta_gamma_inv_g1 = pk.eta_gamma_inv_g1;
                hk.beta_g2 = pk.beta_g2; hk.delta_g2 = pk.delta_g2; hk.gamma_abc_g1 = pk.gamma_abc_g1; hk.commit_witness_count = CW;
                std::vector<BigInt256> inst(z.begin(), z.begin() + NI), wit(z.begin() + NI, z.end()), hv(D);
                std::memcpy(hv[0].data(), h.data(), D * 32);
                bases_cache::clear(); bases_cache::set_min_n(64);
                const bases_cache::Stats c0 = bases_cache::stats();
                for (int k = 0; k < 3; k++) {
                    legogroth16::Proof ph = legogroth16::create_proof_host(hk, 0, &hv, inst, wit, r, s, v);
                    legogroth16::Proof pc = legogroth16::create_proof_host(hk, circ, nullptr, inst, wit, r, s, v);
                    for (const legogroth16::Proof *q : {&ph, &pc}) {
                        EXPECT(q->a.infinity == pr.a.infinity && std::memcmp(&q->a.x, &pr.a.x, 96) == 0 && std::memcmp(&q->b.x, &pr.b.x, 192) == 0);
                        EXPECT(std::memcmp(&q->c.x, &pr.c.x, 96) == 0 && std::memcmp(&q->d.x, &pr.d.x, 96) == 0);
                    }
                }
                const bases_cache::Stats c1 = bases_cache::stats();
                EXPECT(c1.fills - c0.fills == 5 && c1.hits - c0.hits >= 20 && c1.stale == c0.stale);
                // the unmodified one-shot call on one of those vectors: resident already (same pointer, same layout) — and a stale key is noticed
                {
                    auto ref = VariableBaseMSM<G1>::msm_bigint(qh, hv);
                    hk.h_query[0] = hk.h_query[7];                                      // an edit of the FIRST record of the cached vector (always among the samples)
                    const uint64_t stale0 = bases_cache::stats().stale;
                    auto ed = VariableBaseMSM<G1>::msm_bigint(hk.h_query, hv);
                    EXPECT(bases_cache::stats().stale == stale0 + 1);
                    std::vector<G1::Affine> copy = hk.h_query;                           // the same points at another address: never cached, one-shot
                    bases_cache::set_bytes(0);
                    auto ed2 = VariableBaseMSM<G1>::msm_bigint(copy, hv);
                    EXPECT(std::memcmp(&ed, &ed2, sizeof ed) == 0 && std::memcmp(&ed, &ref, sizeof ed) != 0);
                    bases_cache::set_bytes(DGPU_CACHE_BYTES_AUTO);
                }
                bases_cache::clear(); bases_cache::set_min_n((size_t)1 << 16);
            }
        }
        EXPECT(dgpu_r1cs_free(circ) == DGPU_OK);
    }
    // RandomizedMultChecker: thirty claims s_i P_i == T_i (T by the oracle) and one two-term claim in ONE MSM; a wrong claim is caught; P and -P share an entry
    {
        RandomizedMultChecker<G1> chk(BigInt256{0x1234567, 99, 0, 0});
        auto oracle_mul = [&](size_t i, const uint64_t *k) { uint64_t j[18]; G1::Affine t; orc_g1_mul(b1.data() + 12 * i, 0, k, j); uint64_t a12[12]; t.infinity = orc_g1_to_affine(j, a12) != 0; std::memcpy(&t.x, a12, 48); std::memcpy(&t.y, a12 + 6, 48); return t; };
        for (size_t i = 20; i < 50; i++) chk.add_1(P1[i], big[i], oracle_mul(i, &sc[4 * i]));
        { uint64_t two[18], pts[24], ks[8]; std::memcpy(pts, b1.data() + 12 * 60, 96); std::memcpy(pts + 12, b1.data() + 12 * 61, 96); std::memcpy(ks, &sc[4 * 60], 32); std::memcpy(ks + 4, &sc[4 * 61], 32);
          EXPECT(dgpu_lincomb_g1(pts, nullptr, ks, 2, two) == DGPU_OK);
          G1::Affine t; t.infinity = false; std::memcpy(&t.x, two, 48); std::memcpy(&t.y, two + 6, 48);
          chk.add_many({P1[60], P1[61]}, {big[60], big[61]}, t); }
        EXPECT(chk.verify());
        G1::Affine minus = P1[20]; { uint64_t j[18], a12[12]; BigInt256 m1 = detail::FR_MODULUS; m1[0] -= 1; orc_g1_mul(b1.data() + 12 * 20, 0, m1.data(), j); orc_g1_to_affine(j, a12); std::memcpy(&minus.x, a12, 48); std::memcpy(&minus.y, a12 + 6, 48); }
        const size_t before = chk.len();
        chk.add_1(minus, BigInt256{5, 0, 0, 0}, oracle_mul(20, (BigInt256{detail::FR_MODULUS[0] - 5, detail::FR_MODULUS[1], detail::FR_MODULUS[2], detail::FR_MODULUS[3]}).data()));      // 5 (-P) == (r - 5) P
        EXPECT(chk.len() <= before + 1 && chk.verify());        // -P merged into P's entry (only the target may be new)
        chk.add_1(P1[70], big[70], P1[71]);                     // a false claim
        EXPECT(!chk.verify());
    }
    // SnarkPack aggregation through the C++ mirror (aggregation::aggregate_proofs / verify_aggregate_proof over dgpu_snarkpack_*): eight Groth16
    // statements with known discrete logs (e(A, B) = e(alpha, beta) e(k0 + x k1, gamma) e(C, delta) holds exactly), a toy transcript of the
    // caller's (the library only ever sees the two callbacks), then the reference's rejection cases (aggregation/tests.rs:117-330)
    {
        using detail::add_mod; using detail::mul_mod;
        const size_t na = 8;
        std::vector<uint64_t> rs((7 + 3 * na) * 4); orc_rand_scalars(77, 7 + 3 * na, rs.data());
        auto S = [&](size_t i) { BigInt256 v; std::memcpy(v.data(), &rs[4 * i], 32); return v; };
        auto inv = [&](const BigInt256 &a) { BigInt256 e = detail::FR_MODULUS; e[0] -= 2; BigInt256 acc{1, 0, 0, 0}; for (int i = 255; i >= 0; i--) { acc = mul_mod(acc, acc); if ((e[i / 64] >> (i % 64)) & 1) acc = mul_mod(acc, a); } return acc; };
        auto neg = [&](const BigInt256 &a) { BigInt256 r = detail::FR_MODULUS; unsigned __int128 br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)r[i] - a[i] - (uint64_t)br; r[i] = (uint64_t)d; br = (d >> 64) & 1; } return a == BigInt256{} ? a : r; };
        uint64_t g1g[12], g2g[24]; orc_g1_generator(g1g); orc_g2_generator(g2g);
        auto mul1 = [&](const BigInt256 &k, uint64_t *out12) { uint64_t j[18]; orc_g1_mul(g1g, 0, k.data(), j); if (orc_g1_to_affine(j, out12)) std::memset(out12, 0, 96); };
        auto mul2 = [&](const BigInt256 &k, uint64_t *out24) { uint64_t j[36]; orc_g2_mul(g2g, 0, k.data(), j); if (orc_g2_to_affine(j, out24)) std::memset(out24, 0, 192); };
        const BigInt256 alpha = S(0), beta = S(1), gamma = S(2), delta = S(3), k0s = S(4), k1s = S(5), srs_a = S(6), srs_b = mul_mod(S(6), S(5));
        aggregation::VerifyingKey vk; vk.gamma_abc_g1.resize(24);
        mul1(alpha, vk.alpha_g1.data()); mul2(beta, vk.beta_g2.data()); mul2(gamma, vk.gamma_g2.data()); mul2(delta, vk.delta_g2.data());
        mul1(k0s, &vk.gamma_abc_g1[0]); mul1(k1s, &vk.gamma_abc_g1[12]);
        aggregation::Words A(12 * na), B(24 * na), C(12 * na);
        std::vector<BigInt256> pub(na);
        const BigInt256 dinv = inv(delta), ab = mul_mod(alpha, beta);
        for (size_t i = 0; i < na; i++) {
            const BigInt256 a_ = S(7 + 3 * i), b_ = S(8 + 3 * i), x = S(9 + 3 * i);
            const BigInt256 sg = mul_mod(add_mod(k0s, mul_mod(x, k1s)), gamma);
            const BigInt256 c_ = mul_mod(add_mod(mul_mod(a_, b_), neg(add_mod(ab, sg))), dinv);
            mul1(a_, &A[12 * i]); mul2(b_, &B[24 * i]); mul1(c_, &C[12 * i]); pub[i] = x;
        }
        aggregation::Words ga(24 * na), gb(24 * na), ha(48 * na), hb(48 * na);
        { BigInt256 pa{1, 0, 0, 0}, pb{1, 0, 0, 0};
          for (size_t i = 0; i < 2 * na; i++) { mul1(pa, &ga[12 * i]); mul1(pb, &gb[12 * i]); mul2(pa, &ha[24 * i]); mul2(pb, &hb[24 * i]); pa = mul_mod(pa, srs_a); pb = mul_mod(pb, srs_b); } }
        const auto psrs = aggregation::ProverSRS::specialize(na, ga, ha, gb, hb);
        const auto vsrs = aggregation::VerifierSRS::specialize(na, ga, ha, gb, hb);
        struct ToyTranscript {
            uint64_t h;
            explicit ToyTranscript(uint64_t seed) : h(0xcbf29ce484222325ULL ^ seed) {}
            void absorb(const uint8_t *p, size_t n) { for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ULL; } }
            void append_message(const uint8_t *l, size_t ll, const uint8_t *b, size_t n) { absorb(l, ll); absorb(b, n); }
            BigInt256 challenge_scalar(const uint8_t *l, size_t ll) {
                absorb(l, ll);
                BigInt256 v;
                for (int k = 0; k < 4; k++) { h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 29; v[k] = h; absorb((const uint8_t *)&h, 8); }
                v[3] &= 0x3fffffffffffffffULL; v[0] |= 1;             // below 2^254 < r, never zero
                return v;
            }
        };
        const BigInt256 rnd{0x5eed, 7, 0, 0};
        ToyTranscript tp(1);
        const aggregation::Words proof = aggregation::aggregate_proofs(psrs, tp, A, B, C);
        EXPECT(proof.size() == dgpu_snarkpack_proof_words(na, 0) && proof[0] == na && proof[1] == 1);
        { ToyTranscript t(1); EXPECT(aggregation::verify_aggregate_proof(vsrs, vk, pub, 1, proof, rnd, t)); }
        { ToyTranscript t(1); EXPECT(aggregation::verify_aggregate_proof(vsrs, vk, pub, 1, proof, rnd, t, aggregation::Variant::Groth16, nullptr, true)); }
        { ToyTranscript t(2); EXPECT(!aggregation::verify_aggregate_proof(vsrs, vk, pub, 1, proof, rnd, t)); }                       // another transcript
        { ToyTranscript t(1); auto bad = pub; bad[3][0] ^= 1; EXPECT(!aggregation::verify_aggregate_proof(vsrs, vk, bad, 1, proof, rnd, t)); }   // a wrong public input
        { ToyTranscript t(1); auto bad = proof; std::memcpy(&bad[2 + 144 + 144 + 72], &A[0], 96); EXPECT(!aggregation::verify_aggregate_proof(vsrs, vk, pub, 1, bad, rnd, t)); }   // z_c replaced
        { ToyTranscript t1(1), t2(1); auto Cw = C; std::memcpy(&Cw[12], &A[12], 96);                                                  // one wrong proof in the batch
          EXPECT(!aggregation::verify_aggregate_proof(vsrs, vk, pub, 1, aggregation::aggregate_proofs(psrs, t1, A, B, Cw), rnd, t2)); }
        { ToyTranscript t(1); auto cut = proof; cut.pop_back(); bool bad_arg = false;
          try { aggregation::verify_aggregate_proof(vsrs, vk, pub, 1, cut, rnd, t); } catch (const Error &e) { bad_arg = e.code == DGPU_E_BADARG; }
          EXPECT(bad_arg); }
        // the same proofs with a commitment d each through the LegoGroth16 aggregator: two MIPP instances in the proof words (the statement with
        // arbitrary d is not a valid LegoGroth16 one: the verifier has to say no, not fail)
        { ToyTranscript t1(3), t2(3);
          const aggregation::Words pl = aggregation::aggregate_proofs(psrs, t1, A, B, C, &A);
          EXPECT(pl.size() == dgpu_snarkpack_proof_words(na, 1) && pl[1] == 2);
          EXPECT(!aggregation::verify_aggregate_proof(vsrs, vk, pub, 1, pl, rnd, t2, aggregation::Variant::LegoGroth16)); }

        // the LegoGroth16 verifier through the mirror (legogroth16::prepare_verifying_key / verify_proof / verify_proofs_batch, verifier.rs:18-25,
        // :87-109): twelve statements with known discrete logs and two public inputs each,
        //     e(A, B) = e(alpha, beta) e(C, delta) e(g0 + x1 g1 + x2 g2 + D, gamma),
        // accepted one by one and in one call; then the reference's rejections (a wrong input, a swapped commitment, a foreign key)
        {
            const size_t nl = 12, np = 2;
            std::vector<uint64_t> r2((3 + 5 * nl) * 4); orc_rand_scalars(91, 3 + 5 * nl, r2.data());
            auto T = [&](size_t i) { BigInt256 v; std::memcpy(v.data(), &r2[4 * i], 32); return v; };
            auto g1a = [&](const BigInt256 &k) { uint64_t w[12]; mul1(k, w); G1::Affine q; q.infinity = false; std::memcpy(&q.x, w, 48); std::memcpy(&q.y, w + 6, 48); return q; };
            auto g2a = [&](const BigInt256 &k) { uint64_t w[24]; mul2(k, w); G2::Affine q; q.infinity = false; std::memcpy(&q.x, w, 96); std::memcpy(&q.y, w + 12, 96); return q; };
            const BigInt256 gk[4] = {T(0), T(1), T(2), k1s};
            legogroth16::VerifyingKey lvk;
            lvk.alpha_g1 = g1a(alpha); lvk.beta_g2 = g2a(beta); lvk.gamma_g2 = g2a(gamma); lvk.delta_g2 = g2a(delta);
            for (const auto &k : gk) lvk.gamma_abc_g1.push_back(g1a(k));
            lvk.commit_witness_count = 1;
            const legogroth16::PreparedVerifyingKey pvk = legogroth16::prepare_verifying_key(lvk);
            {   // the prepared key is the reference's: e(alpha, beta) by the oracle, and a Miller loop over the prepared -gamma equals the one over the affine point
                uint64_t pa[12], qb[24], ml[72], fe[72]; mul1(alpha, pa); mul2(beta, qb);
                orc_multi_miller_loop(pa, qb, nullptr, 1, 1, ml); EXPECT(orc_final_exponentiation(ml, fe) == 0);
                EXPECT(std::memcmp(fe, pvk.alpha_g1_beta_g2.data(), 576) == 0);
                uint64_t ng[24], m1[72], m2[72]; mul2(neg(gamma), ng);
                orc_multi_miller_loop(pa, ng, nullptr, 1, 1, m1);
                EXPECT(dgpu_multi_miller_loop_prepared(pa, pvk.gamma_g2_neg_pc.data(), nullptr, 1, m2) == DGPU_OK && std::memcmp(m1, m2, 576) == 0);
            }
            std::vector<legogroth16::Proof> proofs(nl); std::vector<std::vector<BigInt256>> inputs(nl);
            for (size_t i = 0; i < nl; i++) {
                const BigInt256 a_ = T(3 + 5 * i), b_ = T(4 + 5 * i), x1 = T(5 + 5 * i), x2 = T(6 + 5 * i), dd = T(7 + 5 * i);
                const BigInt256 dt = add_mod(add_mod(gk[0], mul_mod(x1, gk[1])), add_mod(mul_mod(x2, gk[2]), dd));
                const BigInt256 c_ = mul_mod(add_mod(mul_mod(a_, b_), neg(add_mod(ab, mul_mod(dt, gamma)))), dinv);
                proofs[i].a = g1a(a_); proofs[i].b = g2a(b_); proofs[i].c = g1a(c_); proofs[i].d = g1a(dd); inputs[i] = {x1, x2};
            }
            for (size_t i = 0; i < 3; i++) EXPECT(legogroth16::verify_proof(pvk, proofs[i], inputs[i]));
            const BigInt256 rb{0xabcdef12345ULL, 0x77, 3, 0};
            EXPECT(legogroth16::verify_proofs_batch(pvk, proofs, inputs, rb));
            EXPECT(legogroth16::verify_proofs_batch(pvk, {}, {}, rb));                                        // nothing to check
            { auto bad = inputs; bad[7][1][0] ^= 1; EXPECT(!legogroth16::verify_proof(pvk, proofs[7], bad[7])); EXPECT(!legogroth16::verify_proofs_batch(pvk, proofs, bad, rb)); }
            { auto bad = proofs; std::swap(bad[2].d, bad[3].d); EXPECT(!legogroth16::verify_proof(pvk, bad[2], inputs[2])); EXPECT(!legogroth16::verify_proofs_batch(pvk, bad, inputs, rb)); }
            { auto bad = proofs; bad[nl - 1].c = proofs[0].c; EXPECT(!legogroth16::verify_proofs_batch(pvk, bad, inputs, rb)); }
            { auto other = lvk; std::swap(other.gamma_g2, other.delta_g2); const auto pv2 = legogroth16::prepare_verifying_key(other);
              EXPECT(!legogroth16::verify_proof(pv2, proofs[0], inputs[0])); EXPECT(!legogroth16::verify_proofs_batch(pv2, proofs, inputs, rb)); }
            // argument errors as the reference's: more inputs than gamma_abc has rows (MalformedVerifyingKey), ragged rows, a zero batching scalar
            { bool thrown = false; try { legogroth16::verify_proof(pvk, proofs[0], {T(5), T(6), T(7), T(8)}); } catch (const Error &e) { thrown = e.code == DGPU_E_BADARG; } EXPECT(thrown); }
            { bool thrown = false; auto bad = inputs; bad[4].pop_back(); try { legogroth16::verify_proofs_batch(pvk, proofs, bad, rb); } catch (const Error &e) { thrown = e.code == DGPU_E_LENGTH; } EXPECT(thrown); }
            { bool thrown = false; try { legogroth16::verify_proofs_batch(pvk, proofs, inputs, BigInt256{}); } catch (const Error &e) { thrown = e.code == DGPU_E_BADARG; } EXPECT(thrown); }
            (void)np;
        }
    }
    // several device contexts in this one process (a Rust host is one process): two contexts on the box's one GPU, every MSM chunked
    // over them inside the library (dgpu_msm_*_sharded*), same point as the single-context call
    init_devices({0, 0});
    EXPECT(dgpu_context_count() == 2);
    {
        uint64_t one[18], sh[18], h = 0, hs = 0;
        orc_g1_msm(b1.data(), inf1.data(), sc.data(), n, 8, e1);
        EXPECT(dgpu_msm_g1(b1.data(), inf1.data(), sc.data(), n, one) == DGPU_OK);
        EXPECT(dgpu_msm_g1_sharded(b1.data(), inf1.data(), sc.data(), n, 2, sh) == DGPU_OK && std::memcmp(one, sh, sizeof one) == 0);
        EXPECT(dgpu_msm_g1_sharded(b1.data(), inf1.data(), sc.data(), n, 3, sh) == DGPU_E_BADARG);      // only two contexts exist
        EXPECT(dgpu_bases_upload_g1_sharded(b1.data(), inf1.data(), n, 0, &h) == DGPU_OK);
        EXPECT(dgpu_msm_g1_sharded_handle(h, sc.data(), n, 0, sh) == DGPU_OK && std::memcmp(one, sh, sizeof one) == 0);
        EXPECT(dgpu_msm_g1_sharded_handle(h, scm.data(), n, 1, sh) == DGPU_OK && std::memcmp(one, sh, sizeof one) == 0);
        EXPECT(dgpu_scalars_upload_sharded(sc.data(), n, 0, h, &hs) == DGPU_OK);
        EXPECT(dgpu_msm_g1_sharded_resident(h, hs, sh) == DGPU_OK && std::memcmp(one, sh, sizeof one) == 0);
        uint64_t j[18]; std::memcpy(j, sh, sizeof j);
        G1::Projective pr; std::memcpy(&pr.x, j, 48); std::memcpy(&pr.y, j + 6, 48); std::memcpy(&pr.z, j + 12, 48);
        EXPECT(same_point<G1>(pr, e1, orc_g1_to_affine));
        // the first 1000 terms only (truncation) against a handle that holds 3000 bases
        uint64_t part[18]; orc_g1_msm(b1.data(), inf1.data(), sc.data(), 1000, 8, e1);
        EXPECT(dgpu_msm_g1_sharded_handle(h, sc.data(), 1000, 0, part) == DGPU_OK);
        std::memcpy(&pr.x, part, 48); std::memcpy(&pr.y, part + 6, 48); std::memcpy(&pr.z, part + 12, 48);
        EXPECT(same_point<G1>(pr, e1, orc_g1_to_affine));
        EXPECT(dgpu_scalars_free(hs) == DGPU_OK && dgpu_bases_free(h) == DGPU_OK && dgpu_bases_free(h) == DGPU_E_BADARG);
        uint64_t one2[36], sh2[36];
        EXPECT(dgpu_msm_g2(b2.data(), inf2.data(), sc.data(), n, one2) == DGPU_OK);
        EXPECT(dgpu_msm_g2_sharded(b2.data(), inf2.data(), sc.data(), n, 0, sh2) == DGPU_OK && std::memcmp(one2, sh2, sizeof one2) == 0);
    }
    // RandomizedPairingChecker from a compiled host (utils/src/randomized_pairing_check.rs:273-419's test shape): several true equations of both
    // kinds pass, in lazy and in eager mode; one wrong target or one swapped point makes the whole batch fail; unequal lengths are refused
    {
        auto pt1 = [&](size_t i) { return P1[20 + i]; }; auto pt2 = [&](size_t i) { return P2[20 + i]; };
        auto gt_of = [&](size_t lo, size_t cnt) { Fq12 m{}, g{}; orc_multi_miller_loop(b1.data() + 12 * (20 + lo), b2.data() + 24 * (20 + lo), nullptr, cnt, 4, m.data()); orc_final_exponentiation(m.data(), g.data()); return g; };
        BigInt256 rnd{}; orc_rand_scalars(99, 1, rnd.data());
        // e(x P, Q) == e(P, x Q): both sides through the oracle's scalar multiplication
        const uint64_t xk[4] = {0x1234567890abcdefULL, 0x0fedcba987654321ULL, 0x1111, 0};
        G1::Affine xP; G2::Affine xQ; xP.infinity = xQ.infinity = false;
        { uint64_t j1[18], a1[12], j2[36], a2[24]; orc_g1_mul(b1.data() + 12 * 40, 0, xk, j1); orc_g1_to_affine(j1, a1); std::memcpy(&xP.x, a1, 48); std::memcpy(&xP.y, a1 + 6, 48);
          orc_g2_mul(b2.data() + 24 * 41, 0, xk, j2); orc_g2_to_affine(j2, a2); std::memcpy(&xQ.x, a2, 96); std::memcpy(&xQ.y, a2 + 12, 96); }
        for (int lazy = 0; lazy < 2; lazy++) {
            for (int wrong = 0; wrong < 3; wrong++) {
                RandomizedPairingChecker chk(rnd, lazy != 0);
                chk.add_sources_and_target(pt1(0), pt2(0), gt_of(0, 1));
                chk.add_multiple_sources_and_target({pt1(1), pt1(2), pt1(3)}, {pt2(1), pt2(2), pt2(3)}, wrong == 1 ? gt_of(1, 2) : gt_of(1, 3));
                chk.add_sources(xP, P2[41], P1[40], wrong == 2 ? P2[42] : xQ);
                chk.add_multiple_sources({pt1(4), pt1(5)}, {pt2(4), pt2(5)}, {pt1(5), pt1(4)}, {pt2(5), pt2(4)});
                EXPECT(chk.verify() == (wrong == 0));
            }
        }
        bool refused = false;
        try { RandomizedPairingChecker chk(rnd, true); chk.add_multiple_sources_and_target({pt1(0), pt1(1)}, {pt2(0)}, gt_of(0, 1)); } catch (const Error &e) { refused = e.code == DGPU_E_LENGTH; }
        EXPECT(refused);
    }
    if (fails) std::printf("cpp_api_driver: %d FAILED\n", fails); else std::printf("cpp_api_driver: all equal\n");
    return fails ? 1 : 0;
}
