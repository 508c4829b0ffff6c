// crypto_amd/csrc/digit_codes.hip.h — the signed radix-2^c recoding of one scalar (shared by k_digit_codes, sort_kernels.hip.h, and by the
// small-MSM table kernel, small_kernels.hip.h).  Window w covers scalar bits [w c, w c + c); W = 255 / c + 1 windows, so the top window holds
// fewer than c bits and never carries out.  digit in [-(B-1), B], B = 2^(c-1); code = (|d| - 1) | sign << (CB-1), all-ones = zero digit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msm {
// the W codes of scalar i (i < n_pad; skip: a padding entry or an identity base contributes nothing)
template <class CODE>
__device__ __forceinline__ void digit_codes_one(const uint32_t *__restrict__ scalars, size_t i, size_t n, bool skip, size_t n_pad, int c, int W, CODE *__restrict__ dig, uint32_t *__restrict__ bad) {
    constexpr CODE ZERO = (CODE)~(CODE)0;
    constexpr int SIGN = sizeof(CODE) * 8 - 1;
    uint32_t s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(scalars + i * 8);
        uint4 a = p[0], b = p[1];
        // A scalar is a 255-bit value (Fr::MODULUS_BIT_SIZE).  Whether arkworks' digit extraction reads bit 255 depends on ITS window width
        // (it does unless that width divides 255, oracle/oracle.c ark_make_digits), so a scalar >= 2^255 has no width-independent meaning:
        // the call is refused (DGPU_E_BADARG) and the caller stays on its CPU path.  `into_bigint()` never produces one.
        if (b.w >> 31) atomicOr(bad, 1u);
        if (!skip) { s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w & 0x7fffffffu; }
    }
    const uint32_t B = 1u << (c - 1);
    uint32_t carry = 0;
    for (int w = 0; w < W; w++) {
        int bitpos = w * c;
        uint32_t raw = 0;
        if (bitpos < 256) {
            int wd = bitpos >> 5, sh = bitpos & 31;
            uint64_t v = 0;   // register array indexed through a select chain (no scratch)
#pragma unroll
            for (int k = 0; k < 8; k++) { if (k == wd) v |= s[k]; if (k == wd + 1) v |= (uint64_t)s[k] << 32; }
            raw = (uint32_t)(v >> sh) & ((1u << c) - 1u);
        }
        uint32_t v = raw + carry;
        uint32_t neg = v > B ? 1u : 0u;
        uint32_t mag = neg ? (2u * B - v) : v;
        carry = neg;
        CODE code = (mag == 0 || skip) ? ZERO : (CODE)((mag - 1) | (neg << SIGN));
        dig[(size_t)w * n_pad + i] = code;
    }
}
}  // namespace msm
