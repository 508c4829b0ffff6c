// tests/native/host_tower_driver.cpp — the host's Fq12 tower (crypto_amd/csrc/host_field.hpp) over a fixed pseudo-random sequence of operands
// (one in five components near p or near zero): products, squarings, cyclotomic squarings, inversions, final exponentiations and
// exponentiations by x, all folded into one hash.  tests/test_host_tower.py builds it three ways — the lazy-reduction tower (the product's),
// the same with -DHOSTF_CHECK (every wide value carries its bound, every operation asserts its precondition) and the eager tower
// (-DHOSTF_EAGER_REDUCTION: one Montgomery reduction per Fq product, the statement the lazy formulas must equal bit for bit) — and compares
// the hashes.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include "../../crypto_amd/csrc/host_field.hpp"
using namespace hostf;
static uint64_t S = 88172645463325252ULL;
static uint64_t rnd() { S ^= S << 13; S ^= S >> 7; S ^= S << 17; return S; }
static Fq rfq(int kind) {
    Fq a;
    for (int i = 0; i < 6; i++) a.l[i] = rnd();
    a.l[5] &= 0x0fffffffffffffffULL;
    if (kind == 1) { memcpy(a.l, Fq::P, 48); a.l[0] -= 1 + (rnd() & 3); }          // p - 1 .. p - 4
    if (kind == 2) { memset(a.l, 0, 48); a.l[0] = rnd() & 3; }                     // 0 .. 3
    while (Fq::geq_p(a.l)) Fq::sub_p(a.l);
    return a;
}
static Fq12 r12() { Fq12 f; Fq *q = (Fq *)&f; for (int i = 0; i < 12; i++) q[i] = rfq(rnd() % 5 == 0 ? 1 + (int)(rnd() % 2) : 0); return f; }
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    uint64_t h = 0;
    auto mix = [&](const Fq12 &f) { const uint64_t *w = (const uint64_t *)&f; for (int i = 0; i < 72; i++) h = h * 1099511628211ULL ^ w[i]; };
    for (int it = 0; it < iters; it++) {
        const Fq12 a = r12(), b = r12();
        mix(a * b); mix(a.sqr()); mix(a.cyclotomic_sqr()); mix(a.inv());
        { const Fq2 x = a.c0.c1 * b.c1.c2; const uint64_t *w = (const uint64_t *)&x; for (int i = 0; i < 12; i++) h = h * 1099511628211ULL ^ w[i]; }
        if (it % 100 == 0) { Fq12 o; final_exponentiation(o, a); mix(o); mix(o.cyclotomic_sqr()); mix(exp_by_x(o)); }
    }
    // extremes: every component p - 1, every component zero but one
    { Fq12 m; Fq *q = (Fq *)&m; for (int i = 0; i < 12; i++) { memcpy(q[i].l, Fq::P, 48); q[i].l[0] -= 1; } mix(m * m); mix(m.sqr()); mix(m.cyclotomic_sqr());
      Fq12 z; memset(&z, 0, sizeof z); z.c1.c2.c1 = Fq::one(); mix(z * m); mix(z.sqr()); mix(z * z); }
    printf("host_tower_driver: %016llx\n", (unsigned long long)h);
    return 0;
}
