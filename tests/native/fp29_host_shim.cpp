// tests/native/fp29_host_shim.cpp — host build (g++) of the DEVICE field/group code with the
// FP29_CHECK worst-case bound tracker enabled.  Test-only: lets `-m "not gpu"` tests check the exact
// arithmetic the kernels run, and proves the lazy-limb overflow bounds, without a GPU.
#define FP29_CHECK 1
#include "../../crypto_amd/csrc/fp29.hip.h"
#include "../../crypto_amd/csrc/ec29.hip.h"
#include "../../crypto_amd/csrc/fp30s.hip.h"
#include "../../crypto_amd/csrc/fs2_pair.hip.h"
#include "../../crypto_amd/csrc/pairing29.hip.h"
#include "../../crypto_amd/csrc/host_field.hpp"
#include "../../crypto_amd/csrc/fr29.hip.h"
#include "../../crypto_amd/csrc/fp_safegcd.hip.h"
#include <vector>
#include <string.h>
using namespace bls29;
extern "C" {
void shim_fp_mul(const uint32_t *a, const uint32_t *b, uint32_t *out) { Fp x, y, r; fp_from_abi(x, a); fp_from_abi(y, b); fp_mul(r, x, y); fp_to_abi(out, r); }
void shim_fp_sqr(const uint32_t *a, uint32_t *out) { Fp x, r; fp_from_abi(x, a); fp_sqr(r, x); fp_to_abi(out, r); }
// 1 / (k a) by the division-step inversion the kernels use; k a is formed by unreduced additions (a lazy-class operand)
void shim_fp_inv(const uint32_t *a, int k, uint32_t *out) { Fp x, t, r; fp_from_abi(x, a); t = x; for (int i = 1; i < k; i++) fp_add(t, t, x); fp_inv_safegcd(r, t); fp_to_abi(out, r); }
void shim_fp_roundtrip(const uint32_t *a, uint32_t *out) { Fp x; fp_from_abi(x, a); fp_to_abi(out, x); }
// (a - b) * c with the lazy subtraction + norm, exercising K_M domination
void shim_fp_submul(const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *out) {
    Fp x, y, z, t; fp_from_abi(x, a); fp_from_abi(y, b); fp_from_abi(z, c);
    fp_sub<4>(t, x, y); fp_norm(t, t); fp_mul(t, t, z);
    Fp u; fp_sub<16>(u, t, x); fp_norm(u, u); fp_add(u, u, x); fp_norm(u, u);   // back to t (mod p)
    fp_to_abi(out, u);
}
int shim_fp_is_zero(const uint32_t *a, const uint32_t *b) { Fp x, y, t; fp_from_abi(x, a); fp_from_abi(y, b); fp_sub<4>(t, x, y); fp_norm(t, t); return (fp_maybe_zero(t) ? 1 : 0) | (fp_is_zero_exact(t) ? 2 : 0); }

static void load_aff(Aff<Fp> &p, const uint32_t *xy) { fp_from_abi(p.x, xy); fp_from_abi(p.y, xy + 12); }
static void store_xyzz(uint32_t *out, const Xyzz<Fp> &a, bool inf) {
    if (inf) { memset(out, 0, 4 * 48); return; }
    fp_to_abi(out, a.x); fp_to_abi(out + 12, a.y); fp_to_abi(out + 24, a.zz); fp_to_abi(out + 36, a.zzz);
}
// sum_{i<n} (+/-) pts[i] by repeated mixed addition; out = X,Y,ZZ,ZZZ in ABI form (all-zero = identity)
void shim_g1_madd_chain(const uint32_t *pts, const uint8_t *neg, int n, uint32_t *out) {
    Xyzz<Fp> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (int i = 0; i < n; i++) { Aff<Fp> p; load_aff(p, pts + 24 * i); xyzz_madd(acc, inf, p, neg && neg[i]); }
    store_xyzz(out, acc, inf);
}
// tree: (p0 + p1) + (p2 + p3) ... using the general XYZZ addition
void shim_g1_add_tree(const uint32_t *pts, int n, uint32_t *out) {
    Xyzz<Fp> *v = new Xyzz<Fp>[n > 0 ? n : 1]; bool *f = new bool[n > 0 ? n : 1];
    for (int i = 0; i < n; i++) { Aff<Fp> p; load_aff(p, pts + 24 * i); f[i] = true; fzero(v[i].x); fzero(v[i].y); fzero(v[i].zz); fzero(v[i].zzz); xyzz_madd(v[i], f[i], p, false); }
    int m = n;
    while (m > 1) { int h = (m + 1) / 2; for (int i = 0; i + h < m; i++) xyzz_add(v[i], f[i], v[i + h], f[i + h]); m = h; }
    if (n == 0) { memset(out, 0, 4 * 48); } else store_xyzz(out, v[0], f[0]);
    delete[] v; delete[] f;
}

// ---- the 13 x 30-bit signed field of the G1 MSM kernels (fp30s.hip.h), same entry shapes ----
void shim_fs_mul(const uint32_t *a, const uint32_t *b, uint32_t *out) { Fs x, y, r; fs_from_abi(x, a); fs_from_abi(y, b); fs_mul(r, x, y); fs_to_abi(out, r); }
void shim_fs_sqr(const uint32_t *a, uint32_t *out) { Fs x, r; fs_from_abi(x, a); fs_sqr(r, x); fs_to_abi(out, r); }
void shim_fs_roundtrip(const uint32_t *a, uint32_t *out) { Fs x; fs_from_abi(x, a); fs_to_abi(out, x); }
// a b - c d through the fused two-product reduction
void shim_fs_mul2(const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d, uint32_t *out) {
    Fs x, y, z, w, r; fs_from_abi(x, a); fs_from_abi(y, b); fs_from_abi(z, c); fs_from_abi(w, d);
    fmul_sub<0>(r, x, y, z, w); fs_to_abi(out, r);
}
// ((a - b) c - 3 a + 3 a) with lazy subtractions, both carry passes and a negation on the way
void shim_fs_submul(const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *out) {
    Fs x, y, z, t, u, v; fs_from_abi(x, a); fs_from_abi(y, b); fs_from_abi(z, c);
    fs_sub(t, x, y); fs_bal(t, t); fs_mul(t, t, z);
    fs_add(u, x, x); fs_add(u, u, x); fs_sub(v, t, u); fs_bal_wide(v, v); fs_neg(v, v); fs_neg(v, v); fs_bal(u, u); fs_add(v, v, u); fs_bal(v, v);
    fs_to_abi(out, v);
}
int shim_fs_is_zero(const uint32_t *a, const uint32_t *b) { Fs x, y, t; fs_from_abi(x, a); fs_from_abi(y, b); fs_sub(t, x, y); fs_bal(t, t); return (fs_maybe_zero(t) ? 1 : 0) | (fs_is_zero_exact(t) ? 2 : 0); }
// Fs -> Fp -> Fs and back out: the conversions around the division-step inversion
void shim_fs_via_fp(const uint32_t *a, uint32_t *out_fp, uint32_t *out_fs) { Fs x, y; Fp f; fs_from_abi(x, a); fp_from_fs(f, x); fp_to_abi(out_fp, f); fs_from_fp(y, f); fs_to_abi(out_fs, y); }
static void load_aff_s(Aff<Fs> &p, const uint32_t *xy) { fs_from_abi(p.x, xy); fs_from_abi(p.y, xy + 12); }
static void store_xyzz_s(uint32_t *out, const Xyzz<Fs> &a, bool inf) {
    if (inf) { memset(out, 0, 4 * 48); return; }
    fs_to_abi(out, a.x); fs_to_abi(out + 12, a.y); fs_to_abi(out + 24, a.zz); fs_to_abi(out + 36, a.zzz);
}
void shim_g1s_madd_chain(const uint32_t *pts, const uint8_t *neg, int n, uint32_t *out) {
    Xyzz<Fs> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (int i = 0; i < n; i++) { Aff<Fs> p; load_aff_s(p, pts + 24 * i); xyzz_madd(acc, inf, p, neg && neg[i]); }
    store_xyzz_s(out, acc, inf);
}
void shim_g1s_add_tree(const uint32_t *pts, int n, uint32_t *out) {
    Xyzz<Fs> *v = new Xyzz<Fs>[n > 0 ? n : 1]; bool *f = new bool[n > 0 ? n : 1];
    for (int i = 0; i < n; i++) { Aff<Fs> p; load_aff_s(p, pts + 24 * i); f[i] = true; fzero(v[i].x); fzero(v[i].y); fzero(v[i].zz); fzero(v[i].zzz); xyzz_madd(v[i], f[i], p, false); }
    int m = n;
    while (m > 1) { int h = (m + 1) / 2; for (int i = 0; i + h < m; i++) xyzz_add(v[i], f[i], v[i + h], f[i + h]); m = h; }
    if (n == 0) { memset(out, 0, 4 * 48); } else store_xyzz_s(out, v[0], f[0]);
    delete[] v; delete[] f;
}
// the same tree with the round-by-round form of the addition (ec29.hip.h xyzz_add_rounds: what k_reduce_top's four lanes per point run),
// alternating with xyzz_add level by level so that each form must accept the other's outputs
void shim_g1s_add_tree_rounds(const uint32_t *pts, int n, uint32_t *out) {
    Xyzz<Fs> *v = new Xyzz<Fs>[n > 0 ? n : 1]; bool *f = new bool[n > 0 ? n : 1];
    for (int i = 0; i < n; i++) { Aff<Fs> p; load_aff_s(p, pts + 24 * i); f[i] = true; fzero(v[i].x); fzero(v[i].y); fzero(v[i].zz); fzero(v[i].zzz); xyzz_madd(v[i], f[i], p, false); }
    int m = n, lvl = 0; QuadSerial q4;
    while (m > 1) { int h = (m + 1) / 2; for (int i = 0; i + h < m; i++) { if (lvl & 1) xyzz_add(v[i], f[i], v[i + h], f[i + h]); else xyzz_add_rounds(v[i], f[i], v[i + h], f[i + h], q4); } m = h; lvl++; }
    if (n == 0) { memset(out, 0, 4 * 48); } else store_xyzz_s(out, v[0], f[0]);
    delete[] v; delete[] f;
}
void shim_g1s_dbl_chain_rounds(const uint32_t *pt, int k, uint32_t *out) {
    Aff<Fs> p; load_aff_s(p, pt);
    Xyzz<Fs> acc; xyzz_dbl_affine(acc, p); QuadSerial q4;
    for (int i = 1; i < k; i++) { Xyzz<Fs> d; if (i & 1) xyzz_dbl_rounds(d, acc, q4); else xyzz_dbl(d, acc); acc = d; }
    store_xyzz_s(out, acc, false);
}
// doubling chain 2^k P through xyzz_dbl_affine / xyzz_dbl (the table construction's step)
void shim_g1s_dbl_chain(const uint32_t *pt, int k, uint32_t *out) {
    Aff<Fs> p; load_aff_s(p, pt);
    Xyzz<Fs> acc; xyzz_dbl_affine(acc, p);
    for (int i = 1; i < k; i++) { Xyzz<Fs> d; xyzz_dbl(d, acc); acc = d; }
    store_xyzz_s(out, acc, false);
}

// ---- G2 over the signed field (fs2_pair.hip.h, one-lane form; the lane-pair form runs the same component arithmetic) ----
static void load_aff2s(Aff<Fs2> &p, const uint32_t *xy) { fs_from_abi(p.x.c0, xy); fs_from_abi(p.x.c1, xy + 12); fs_from_abi(p.y.c0, xy + 24); fs_from_abi(p.y.c1, xy + 36); }
static void store_xyzz2s(uint32_t *out, const Xyzz<Fs2> &a, bool inf) {
    if (inf) { memset(out, 0, 4 * 96); return; }
    const Fs *f = reinterpret_cast<const Fs *>(&a);
    for (int k = 0; k < 8; k++) fs_to_abi(out + 12 * k, f[k]);
}
void shim_fs2_mul(const uint32_t *a, const uint32_t *b, uint32_t *out) {
    Fs2 x, y, r; fs_from_abi(x.c0, a); fs_from_abi(x.c1, a + 12); fs_from_abi(y.c0, b); fs_from_abi(y.c1, b + 12);
    fmul(r, x, y); fs_to_abi(out, r.c0); fs_to_abi(out + 12, r.c1);
}
void shim_fs2_sqr(const uint32_t *a, uint32_t *out) {
    Fs2 x, r; fs_from_abi(x.c0, a); fs_from_abi(x.c1, a + 12);
    fsqr(r, x); fs_to_abi(out, r.c0); fs_to_abi(out + 12, r.c1);
}
// early = 1: the early-return form of the mixed addition (what the lane-pair kernels instantiate)
void shim_g2s_madd_chain(const uint32_t *pts, const uint8_t *neg, int n, int early, uint32_t *out) {
    Xyzz<Fs2> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (int i = 0; i < n; i++) { Aff<Fs2> p; load_aff2s(p, pts + 48 * i); if (early) xyzz_madd_early(acc, inf, p, neg && neg[i]); else xyzz_madd(acc, inf, p, neg && neg[i]); }
    store_xyzz2s(out, acc, inf);
}
void shim_g2s_add_tree(const uint32_t *pts, int n, uint32_t *out) {
    Xyzz<Fs2> *v = new Xyzz<Fs2>[n > 0 ? n : 1]; bool *f = new bool[n > 0 ? n : 1];
    for (int i = 0; i < n; i++) { Aff<Fs2> p; load_aff2s(p, pts + 48 * i); f[i] = true; fzero(v[i].x); fzero(v[i].y); fzero(v[i].zz); fzero(v[i].zzz); xyzz_madd_early(v[i], f[i], p, false); }
    int m = n;
    while (m > 1) { int h = (m + 1) / 2; for (int i = 0; i + h < m; i++) xyzz_add(v[i], f[i], v[i + h], f[i + h]); m = h; }
    if (n == 0) { memset(out, 0, 4 * 96); } else store_xyzz2s(out, v[0], f[0]);
    delete[] v; delete[] f;
}
void shim_g2s_add_tree_rounds(const uint32_t *pts, int n, uint32_t *out) {
    Xyzz<Fs2> *v = new Xyzz<Fs2>[n > 0 ? n : 1]; bool *f = new bool[n > 0 ? n : 1];
    for (int i = 0; i < n; i++) { Aff<Fs2> p; load_aff2s(p, pts + 48 * i); f[i] = true; fzero(v[i].x); fzero(v[i].y); fzero(v[i].zz); fzero(v[i].zzz); xyzz_madd_early(v[i], f[i], p, false); }
    int m = n, lvl = 0; QuadSerial q4;
    while (m > 1) { int h = (m + 1) / 2; for (int i = 0; i + h < m; i++) { if (lvl & 1) xyzz_add(v[i], f[i], v[i + h], f[i + h]); else xyzz_add_rounds(v[i], f[i], v[i + h], f[i + h], q4); } m = h; lvl++; }
    if (n == 0) { memset(out, 0, 4 * 96); } else store_xyzz2s(out, v[0], f[0]);
    delete[] v; delete[] f;
}
void shim_g2s_dbl_chain_rounds(const uint32_t *pt, int k, uint32_t *out) {
    Aff<Fs2> p; load_aff2s(p, pt);
    Xyzz<Fs2> acc; xyzz_dbl_affine(acc, p); QuadSerial q4;
    for (int i = 1; i < k; i++) { Xyzz<Fs2> d; if (i & 1) xyzz_dbl_rounds(d, acc, q4); else xyzz_dbl(d, acc); acc = d; }
    store_xyzz2s(out, acc, false);
}
void shim_g2s_dbl_chain(const uint32_t *pt, int k, uint32_t *out) {
    Aff<Fs2> p; load_aff2s(p, pt);
    Xyzz<Fs2> acc; xyzz_dbl_affine(acc, p);
    for (int i = 1; i < k; i++) { Xyzz<Fs2> d; xyzz_dbl(d, acc); acc = d; }
    store_xyzz2s(out, acc, false);
}

// ---- G2 ----
static void load_aff2(Aff<Fp2> &p, const uint32_t *xy) { fp_from_abi(p.x.c0, xy); fp_from_abi(p.x.c1, xy + 12); fp_from_abi(p.y.c0, xy + 24); fp_from_abi(p.y.c1, xy + 36); }
static void store_xyzz2(uint32_t *out, const Xyzz<Fp2> &a, bool inf) {
    if (inf) { memset(out, 0, 4 * 96); return; }
    const Fp *f = reinterpret_cast<const Fp *>(&a);
    for (int k = 0; k < 8; k++) fp_to_abi(out + 12 * k, f[k]);
}
void shim_fp2_mul(const uint32_t *a, const uint32_t *b, uint32_t *out) {
    Fp2 x, y, r; fp_from_abi(x.c0, a); fp_from_abi(x.c1, a + 12); fp_from_abi(y.c0, b); fp_from_abi(y.c1, b + 12);
    fmul(r, x, y); fp_to_abi(out, r.c0); fp_to_abi(out + 12, r.c1);
}
void shim_fp2_sqr(const uint32_t *a, uint32_t *out) {
    Fp2 x, r; fp_from_abi(x.c0, a); fp_from_abi(x.c1, a + 12);
    fsqr(r, x); fp_to_abi(out, r.c0); fp_to_abi(out + 12, r.c1);
}
void shim_g2_madd_chain(const uint32_t *pts, const uint8_t *neg, int n, uint32_t *out) {
    Xyzz<Fp2> acc; bool inf = true;
    fzero(acc.x); fzero(acc.y); fzero(acc.zz); fzero(acc.zzz);
    for (int i = 0; i < n; i++) { Aff<Fp2> p; load_aff2(p, pts + 48 * i); xyzz_madd(acc, inf, p, neg && neg[i]); }
    store_xyzz2(out, acc, inf);
}
void shim_g2_add_tree(const uint32_t *pts, int n, uint32_t *out) {
    Xyzz<Fp2> *v = new Xyzz<Fp2>[n > 0 ? n : 1]; bool *f = new bool[n > 0 ? n : 1];
    for (int i = 0; i < n; i++) { Aff<Fp2> p; load_aff2(p, pts + 48 * i); f[i] = true; fzero(v[i].x); fzero(v[i].y); fzero(v[i].zz); fzero(v[i].zzz); xyzz_madd(v[i], f[i], p, false); }
    int m = n;
    while (m > 1) { int h = (m + 1) / 2; for (int i = 0; i + h < m; i++) xyzz_add(v[i], f[i], v[i + h], f[i + h]); m = h; }
    if (n == 0) { memset(out, 0, 4 * 96); } else store_xyzz2(out, v[0], f[0]);
    delete[] v; delete[] f;
}

// ---- pairing tower / Miller-loop lines ----
static void load_f12(Fp12d &f, const uint32_t *w) { Fp *c = reinterpret_cast<Fp *>(&f); for (int k = 0; k < 12; k++) fp_from_abi(c[k], w + 12 * k); }
static void store_f12(uint32_t *w, const Fp12d &f) { const Fp *c = reinterpret_cast<const Fp *>(&f); for (int k = 0; k < 12; k++) fp_to_abi(w + 12 * k, c[k]); }
static void load_f2(Fp2 &f, const uint32_t *w) { fp_from_abi(f.c0, w); fp_from_abi(f.c1, w + 12); }
void shim_f12_mul(const uint32_t *a, const uint32_t *b, int reps, uint32_t *out) {
    Fp12d x, y, r; load_f12(x, a); load_f12(y, b);
    f12_mul(r, x, y);
    for (int i = 1; i < reps; i++) { Fp12d t; f12_mul(t, r, y); r = t; }     // feed outputs back in: bounds must close
    store_f12(out, r);
}
// the same product regrouped into 18 roles + 6 outputs (k_product_tree18's dataflow); every other repetition through f12_mul: both forms
// must accept each other's outputs
void shim_f12_mul_roles(const uint32_t *a, const uint32_t *b, int reps, uint32_t *out) {
    Fp12d x, y, r; load_f12(x, a); load_f12(y, b);
    f12_mul_roles(r, x, y);
    for (int i = 1; i < reps; i++) { Fp12d t; if (i & 1) f12_mul(t, r, y); else f12_mul_roles(t, r, y); r = t; }
    store_f12(out, r);
}
void shim_f12_mul_by_014(const uint32_t *a, const uint32_t *c0, const uint32_t *c1, const uint32_t *c4, int reps, uint32_t *out) {
    Fp12d x; load_f12(x, a); Fp2 k0, k1, k4; load_f2(k0, c0); load_f2(k1, c1); load_f2(k4, c4);
    for (int i = 0; i < reps; i++) f12_mul_by_014(x, k0, k1, k4);
    store_f12(out, x);
}
static void pair_lines(std::vector<Line> &ls, const uint32_t *p, const uint32_t *q) {
    Fp px, py; fp_from_abi(px, p); fp_from_abi(py, p + 12);
    Aff<Fp2> Q; load_aff2(Q, q);
    G2Proj R; R.x = Q.x; R.y = Q.y; fset_one(R.z);
    for (int i = 62; i >= 0; i--) {
        Line l; line_dbl_step(R, l); line_eval(l, px, py); ls.push_back(l);
        if ((BLS_X_ABS >> i) & 1) { line_add_step(R, Q, l); line_eval(l, px, py); ls.push_back(l); }
    }
}
// the same lines through line_dbl_step_fast (what k_miller_lines_hex computes): un-normalised coefficients are carry-passed by the consumer
static void pair_lines_fast(std::vector<Line> &ls, const uint32_t *p, const uint32_t *q) {
    Fp px, py; fp_from_abi(px, p); fp_from_abi(py, p + 12);
    Aff<Fp2> Q; load_aff2(Q, q);
    G2Proj R; R.x = Q.x; R.y = Q.y; fset_one(R.z);
    for (int i = 62; i >= 0; i--) {
        const bool add = (BLS_X_ABS >> i) & 1;
        Line l; line_dbl_step_fast(R, l, add || i == 0);
        fnorm(l.c0, l.c0); fnorm(l.c1, l.c1); fnorm(l.c2, l.c2);
        line_eval(l, px, py); ls.push_back(l);
        if (add) { line_add_step(R, Q, l); line_eval(l, px, py); ls.push_back(l); }
    }
}
void shim_miller_lines_fast(const uint32_t *p, const uint32_t *q, uint32_t *out) {
    std::vector<Line> ls; pair_lines_fast(ls, p, q);
    for (size_t s = 0; s < ls.size(); s++) { const Fp *c = reinterpret_cast<const Fp *>(&ls[s]); for (int k = 0; k < 6; k++) fp_to_abi(out + (s * 6 + k) * 12, c[k]); }
}
// ... and through line_dbl_step_ws / line_add_step_ws (what k_miller_lines_ws computes: a wave per role, four-lane products)
void shim_miller_lines_ws(const uint32_t *p, const uint32_t *q, uint32_t *out) {
    Fp px, py; fp_from_abi(px, p); fp_from_abi(py, p + 12);
    Aff<Fp2> Q; load_aff2(Q, q);
    G2Proj R; R.x = Q.x; R.y = Q.y; fset_one(R.z);
    std::vector<Line> ls;
    for (int i = 62; i >= 0; i--) {
        const bool add = (BLS_X_ABS >> i) & 1;
        Line l; line_dbl_step_ws(R, l, add || i == 0);
        fnorm(l.c0, l.c0); fnorm(l.c1, l.c1); fnorm(l.c2, l.c2);
        line_eval(l, px, py); ls.push_back(l);
        if (add) { line_add_step_ws(R, Q, l); line_eval(l, px, py); ls.push_back(l); }
    }
    for (size_t s = 0; s < ls.size(); s++) { const Fp *c = reinterpret_cast<const Fp *>(&ls[s]); for (int k = 0; k < 6; k++) fp_to_abi(out + (s * 6 + k) * 12, c[k]); }
}
// 12 a by the scaled carry pass, from a lazily reduced operand (k a, carry-passed)
void shim_fp_mul12(const uint32_t *a, int k, uint32_t *out) {
    Fp x, t, r; fp_from_abi(x, a); t = x; for (int i = 1; i < k; i++) fp_add(t, t, x); fp_norm(t, t);
    fp_mul12_norm(r, t); fp_norm(r, r); fp_to_abi(out, r);
}
// out: 68 x (c0, c1, c2) in ABI form (6 Fp each)
void shim_miller_lines(const uint32_t *p, const uint32_t *q, uint32_t *out) {
    std::vector<Line> ls; pair_lines(ls, p, q);
    for (size_t s = 0; s < ls.size(); s++) { const Fp *c = reinterpret_cast<const Fp *>(&ls[s]); for (int k = 0; k < 6; k++) fp_to_abi(out + (s * 6 + k) * 12, c[k]); }
}
// the kernels' dataflow on the host: per-step products over the pairs (two halves combined with the dense product),
// then the host-side square-and-multiply over the 68 step products, conjugation at the end
void shim_multi_miller(const uint32_t *ps, const uint32_t *qs, int n, uint32_t *out) {
    std::vector<std::vector<Line>> all(n);
    for (int i = 0; i < n; i++) pair_lines(all[i], ps + 24 * i, qs + 48 * i);
    hostf::Fq12 L[N_LINES];
    for (int s = 0; s < N_LINES; s++) {
        Fp12d half[2]; f12_set_one(half[0]); f12_set_one(half[1]);
        for (int i = 0; i < n; i++) { Fp12d &h = half[i & 1]; f12_mul_by_014(h, all[i][s].c0, all[i][s].c1, all[i][s].c2); }
        Fp12d prod; f12_mul(prod, half[0], half[1]);
        uint32_t w[144]; store_f12(w, prod); memcpy(&L[s], w, 576);
    }
    hostf::Fq12 f = hostf::Fq12::one(); int idx = 0;
    for (int i = 62; i >= 0; i--) { f = f.sqr() * L[idx++]; if ((BLS_X_ABS >> i) & 1) f = f * L[idx++]; }
    f = f.conj();
    memcpy(out, &f, 576);
}

// ---- Fr (scalar field, NTT path) ----
void shim_fr_mul(const uint32_t *a, const uint32_t *b, int mont, uint32_t *out) { fr29::Fr x, y, r; fr29::fr_from_words(x, a, mont); fr29::fr_from_words(y, b, mont); fr29::fr_mul(r, x, y); fr29::fr_to_words(out, r, mont); }
// a chain of butterflies exercising lazy add / sub / norm:  (x, y) -> (x + w y, x - w y) repeated `reps` times
void shim_fr_butterflies(const uint32_t *a, const uint32_t *b, const uint32_t *w, int reps, uint32_t *out_x, uint32_t *out_y) {
    fr29::Fr x, y, tw, t, u, v; fr29::fr_from_words(x, a, 0); fr29::fr_from_words(y, b, 0); fr29::fr_from_words(tw, w, 0);
    for (int i = 0; i < reps; i++) { fr29::fr_mul(t, y, tw); fr29::fr_add(u, x, t); fr29::fr_sub(v, x, t); fr29::fr_norm(x, u); fr29::fr_norm(y, v); }
    fr29::fr_to_words(out_x, x, 0); fr29::fr_to_words(out_y, y, 0);
}
}
