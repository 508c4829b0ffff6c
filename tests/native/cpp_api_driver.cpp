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
    if (fails) std::printf("cpp_api_driver: %d FAILED\n", fails); else std::printf("cpp_api_driver: all equal\n");
    return fails ? 1 : 0;
}
