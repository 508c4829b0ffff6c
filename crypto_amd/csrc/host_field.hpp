// crypto_amd/csrc/host_field.hpp — host-side (x86-64) BLS12-381 field and group arithmetic used by
// libdock_gpu.so for the O(window-count) tail of an MSM (Horner fold of the per-window sums, final
// normalisation) and for scalar preparation.  A lone wave on the GPU needs ~15 us per group addition;
// a host core needs ~0.3 us, so the few hundred strictly sequential operations at the end of an MSM
// belong here.  Product code: it does not use anything under oracle/.
//
// Representation = the C-ABI one: 6 x u64 little-endian, value * 2^384 mod p (ark-ff Fp<MontBackend,6>).
#pragma once
#include <stdint.h>
#include <string.h>

namespace hostf {

typedef unsigned __int128 u128;

struct Fq {
    uint64_t l[6];
    static constexpr uint64_t P[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
    static constexpr uint64_t ONE[6] = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
    static constexpr uint64_t INV = 0x89f3fffcfffcfffdULL;

    static Fq zero() { Fq r; memset(&r, 0, sizeof r); return r; }
    static Fq one() { Fq r; memcpy(r.l, ONE, sizeof ONE); return r; }
    bool is_zero() const { uint64_t t = 0; for (int i = 0; i < 6; i++) t |= l[i]; return t == 0; }
    bool operator==(const Fq &o) const { uint64_t t = 0; for (int i = 0; i < 6; i++) t |= l[i] ^ o.l[i]; return t == 0; }

    static bool geq_p(const uint64_t *a) {
        for (int i = 5; i >= 0; i--) { if (a[i] > P[i]) return true; if (a[i] < P[i]) return false; }
        return true;
    }
    static void sub_p(uint64_t *a) {
        uint64_t br = 0;
        for (int i = 0; i < 6; i++) { u128 d = (u128)a[i] - P[i] - br; a[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    }
    Fq operator+(const Fq &b) const {
        Fq r; uint64_t c = 0;
        for (int i = 0; i < 6; i++) { u128 s = (u128)l[i] + b.l[i] + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        if (geq_p(r.l)) sub_p(r.l);
        return r;
    }
    Fq operator-(const Fq &b) const {
        Fq r; uint64_t br = 0;
        for (int i = 0; i < 6; i++) { u128 d = (u128)l[i] - b.l[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
        if (br) { uint64_t c = 0; for (int i = 0; i < 6; i++) { u128 s = (u128)r.l[i] + P[i] + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
        return r;
    }
    Fq neg() const { return is_zero() ? *this : zero() - *this; }
    // Montgomery product (coarsely integrated operand scanning)
    Fq operator*(const Fq &b) const {
        uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; i++) {
            uint64_t c = 0; u128 s;
            for (int j = 0; j < 6; j++) { s = (u128)l[j] * b.l[i] + t[j] + c; t[j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
            s = (u128)t[6] + c; t[6] = (uint64_t)s; t[7] = (uint64_t)(s >> 64);
            uint64_t m = t[0] * INV;
            s = (u128)m * P[0] + t[0]; c = (uint64_t)(s >> 64);
            for (int j = 1; j < 6; j++) { s = (u128)m * P[j] + t[j] + c; t[j - 1] = (uint64_t)s; c = (uint64_t)(s >> 64); }
            s = (u128)t[6] + c; t[5] = (uint64_t)s; t[6] = t[7] + (uint64_t)(s >> 64);
        }
        Fq r; for (int i = 0; i < 6; i++) r.l[i] = t[i];
        if (t[6] || geq_p(r.l)) sub_p(r.l);
        return r;
    }
    Fq sqr() const { return (*this) * (*this); }
    Fq dbl() const { return (*this) + (*this); }
    Fq inv() const {   // Fermat: a^(p-2)
        static constexpr uint64_t E[6] = {0xb9feffffffffaaa9ULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
        Fq acc = one();
        for (int i = 380; i >= 0; i--) { acc = acc.sqr(); if ((E[i / 64] >> (i % 64)) & 1) acc = acc * (*this); }
        return acc;
    }
};

struct Fq2 {
    Fq c0, c1;
    static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
    static Fq2 one() { return {Fq::one(), Fq::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fq2 &o) const { return c0 == o.c0 && c1 == o.c1; }
    Fq2 operator+(const Fq2 &b) const { return {c0 + b.c0, c1 + b.c1}; }
    Fq2 operator-(const Fq2 &b) const { return {c0 - b.c0, c1 - b.c1}; }
    Fq2 neg() const { return {c0.neg(), c1.neg()}; }
    Fq2 operator*(const Fq2 &b) const {
        Fq t0 = c0 * b.c0, t1 = c1 * b.c1, t2 = (c0 + c1) * (b.c0 + b.c1);
        return {t0 - t1, t2 - t0 - t1};
    }
    Fq2 sqr() const { Fq t = c0 * c1; return {(c0 + c1) * (c0 - c1), t + t}; }
    Fq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    Fq2 inv() const { Fq n = (c0.sqr() + c1.sqr()).inv(); return {c0 * n, (c1 * n).neg()}; }
};

// Extended Jacobian point (x = X/ZZ, y = Y/ZZZ); identity: inf = true.
template <class F> struct HXyzz {
    F x, y, zz, zzz; bool inf;
    static HXyzz identity() { HXyzz r; r.x = F::zero(); r.y = F::zero(); r.zz = F::zero(); r.zzz = F::zero(); r.inf = true; return r; }
    void dbl_in_place() {
        if (inf) return;
        F U = y.dbl(), V = U.sqr(), W = U * V, S = x * V, M = x.sqr(); M = M + M + M;
        F X3 = M.sqr() - S - S, Y3 = M * (S - X3) - W * y;
        x = X3; y = Y3; zz = V * zz; zzz = W * zzz;
    }
    void add_in_place(const HXyzz &b) {
        if (b.inf) return;
        if (inf) { *this = b; return; }
        F U1 = x * b.zz, U2 = b.x * zz, S1 = y * b.zzz, S2 = b.y * zzz;
        F Pd = U2 - U1, Rd = S2 - S1;
        if (Pd.is_zero()) { if (Rd.is_zero()) dbl_in_place(); else *this = identity(); return; }
        F PP = Pd.sqr(), PPP = Pd * PP, Q = U1 * PP;
        F X3 = Rd.sqr() - PPP - Q - Q, Y3 = Rd * (Q - X3) - S1 * PPP;
        x = X3; y = Y3; zz = zz * b.zz * PP; zzz = zzz * b.zzz * PPP;
    }
    // canonical Jacobian triple: (x_aff, y_aff, 1) or (1, 1, 0) for the identity — ark-ec Projective::zero()
    void to_normalised_jacobian(F &X, F &Y, F &Z) const {
        if (inf) { X = F::one(); Y = F::one(); Z = F::zero(); return; }
        F i3 = zzz.inv();            // 1/ZZZ
        F i2 = i3 * i3 * zz * zz;    // ZZ^2/ZZZ^2 = ZZ^2/ZZ^3 = 1/ZZ
        X = x * i2; Y = y * i3; Z = F::one();
    }
};

// ---- Fr (scalar field): Montgomery (R = 2^256) -> canonical, i.e. ark-ff into_bigint ----
static inline void fr_from_mont(uint64_t out[4], const uint64_t a[4]) {
    static constexpr uint64_t MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    static constexpr uint64_t INV = 0xfffffffeffffffffULL;
    uint64_t t[6] = {a[0], a[1], a[2], a[3], 0, 0};
    for (int i = 0; i < 4; i++) {   // four reduction rounds: t = (t + m * MOD) / 2^64
        uint64_t m = t[0] * INV; u128 s = (u128)m * MOD[0] + t[0]; uint64_t c = (uint64_t)(s >> 64);
        for (int j = 1; j < 4; j++) { s = (u128)m * MOD[j] + t[j] + c; t[j - 1] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        s = (u128)t[4] + c; t[3] = (uint64_t)s; t[4] = (uint64_t)(s >> 64);
    }
    bool ge = t[4] != 0;
    if (!ge) { ge = true; for (int i = 3; i >= 0; i--) { if (t[i] > MOD[i]) break; if (t[i] < MOD[i]) { ge = false; break; } } }
    if (ge) { uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - MOD[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }
    for (int i = 0; i < 4; i++) out[i] = t[i];
}

}  // namespace hostf
