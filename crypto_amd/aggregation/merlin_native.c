/* crypto_amd/aggregation/merlin_native.c — dock_crypto_utils::transcript::MerlinTranscript in C, with the two callbacks of `dgpu_transcript`
 * (include/dock_gpu.h) as plain C functions: what a Rust host's merlin::Transcript costs the library (microseconds per call), where the Python
 * transcript of crypto_amd/aggregation/transcript.py costs ~0.35 ms per GIPA round of interpreter time.
 *
 * NOT part of the product ABI: Merlin is the caller's.  Test / bench helper, built into libkeccak_f1600.so next to keccak_f1600.c by
 * __graft_entry__.build(); pinned byte for byte against transcript.py (tests/test_transcript.py) and used by bench.py for the
 * `*_native_transcript_ms` figures.  Follows /root/reference/merlin/src/strobe.rs:60-190 (STROBE-128 subset: meta-AD, AD, PRF; rate 166),
 * /root/reference/merlin/src/transcript.rs:74-215 (`new`, `append_message`, `challenge_bytes`) and
 * /root/reference/utils/src/transcript.rs:103-122 (`challenge_scalar`: 64 PRF bytes -> the first 32 little-endian with the top bit shaved must be a
 * non-zero element < r, else resample; the INVERSE of the element is returned). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int keccak_f1600(uint8_t state[200]);
#define API __attribute__((visibility("default")))
#define STROBE_R 166
enum { FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };

typedef struct { uint8_t st[200]; uint8_t pos, pos_begin, cur_flags; } strobe;

static void run_f(strobe *s) {
    s->st[s->pos] ^= s->pos_begin; s->st[s->pos + 1] ^= 0x04; s->st[STROBE_R + 1] ^= 0x80;
    keccak_f1600(s->st);
    s->pos = 0; s->pos_begin = 0;
}
static void absorb(strobe *s, const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) { s->st[s->pos++] ^= d[i]; if (s->pos == STROBE_R) run_f(s); }
}
static void squeeze(strobe *s, uint8_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) { out[i] = s->st[s->pos]; s->st[s->pos] = 0; s->pos++; if (s->pos == STROBE_R) run_f(s); }
}
static void begin_op(strobe *s, uint8_t flags, int more) {
    if (more) return;                                   /* (the caller keeps cur_flags consistent) */
    const uint8_t old_begin = s->pos_begin;
    s->pos_begin = (uint8_t)(s->pos + 1);
    s->cur_flags = flags;
    const uint8_t hdr[2] = {old_begin, flags};
    absorb(s, hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && s->pos != 0) run_f(s);
}
static void meta_ad(strobe *s, const uint8_t *d, size_t n, int more) { begin_op(s, FLAG_M | FLAG_A, more); absorb(s, d, n); }
static void ad(strobe *s, const uint8_t *d, size_t n, int more) { begin_op(s, FLAG_A, more); absorb(s, d, n); }
static void prf(strobe *s, uint8_t *out, size_t n, int more) { begin_op(s, FLAG_I | FLAG_A | FLAG_C, more); squeeze(s, out, n); }

static void le32(uint8_t o[4], uint32_t v) { o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16); o[3] = (uint8_t)(v >> 24); }
static void merlin_append(strobe *s, const uint8_t *label, size_t ll, const uint8_t *msg, size_t n) {
    uint8_t len[4]; le32(len, (uint32_t)n);
    meta_ad(s, label, ll, 0); meta_ad(s, len, 4, 1); ad(s, msg, n, 0);
}
static void merlin_challenge(strobe *s, const uint8_t *label, size_t ll, uint8_t *out, size_t n) {
    uint8_t len[4]; le32(len, (uint32_t)n);
    meta_ad(s, label, ll, 0); meta_ad(s, len, 4, 1); prf(s, out, n, 0);
}

/* ---- 256-bit arithmetic mod r for the inverse: binary extended Euclid (r odd) ---- */
typedef struct { uint64_t w[5]; } u320;                /* one spare word for the intermediate x + r */
static const uint64_t RMOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static int ge4(const uint64_t a[4], const uint64_t b[4]) { for (int i = 3; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; } return 1; }
static void sub4(uint64_t a[4], const uint64_t b[4]) { unsigned __int128 br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 t = (unsigned __int128)a[i] - b[i] - (uint64_t)br; a[i] = (uint64_t)t; br = (t >> 64) & 1; } }
static int is_zero4(const uint64_t a[4]) { return !(a[0] | a[1] | a[2] | a[3]); }
static int is_one4(const uint64_t a[4]) { return a[0] == 1 && !(a[1] | a[2] | a[3]); }
static void shr1_4(uint64_t a[4], uint64_t top) { for (int i = 0; i < 3; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 63); a[3] = (a[3] >> 1) | (top << 63); }
/* x <- x / 2 mod r */
static void half_mod(uint64_t x[4]) {
    if (x[0] & 1) { unsigned __int128 c = 0; for (int i = 0; i < 4; i++) { c += (unsigned __int128)x[i] + RMOD[i]; x[i] = (uint64_t)c; c >>= 64; } shr1_4(x, (uint64_t)c); }
    else shr1_4(x, 0);
}
static void sub_mod(uint64_t a[4], const uint64_t b[4]) {      /* a <- a - b mod r (both < r) */
    if (ge4(a, b)) sub4(a, b);
    else { uint64_t t[4]; memcpy(t, RMOD, 32); sub4(t, b); unsigned __int128 c = 0; for (int i = 0; i < 4; i++) { c += (unsigned __int128)a[i] + t[i]; a[i] = (uint64_t)c; c >>= 64; } }
}
static void inv_mod_r(uint64_t out[4], const uint64_t a_in[4]) {
    uint64_t u[4], v[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
    memcpy(u, a_in, 32); memcpy(v, RMOD, 32);
    while (!is_one4(u) && !is_one4(v)) {
        while (!(u[0] & 1)) { shr1_4(u, 0); half_mod(x1); }
        while (!(v[0] & 1)) { shr1_4(v, 0); half_mod(x2); }
        if (ge4(u, v)) { sub4(u, v); sub_mod(x1, x2); } else { sub4(v, u); sub_mod(x2, x1); }
    }
    memcpy(out, is_one4(u) ? x1 : x2, 32);
}

/* ---- the transcript object and the callbacks of dgpu_transcript ---- */
API void *mt_new(const uint8_t *label, size_t n) {
    strobe *s = (strobe *)calloc(1, sizeof(strobe));
    if (!s) return NULL;
    static const uint8_t init[18] = {1, STROBE_R + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    memcpy(s->st, init, 18);
    keccak_f1600(s->st);
    meta_ad(s, (const uint8_t *)"Merlin v1.0", 11, 0);
    merlin_append(s, (const uint8_t *)"dom-sep", 7, label, n);
    return s;
}
API void *mt_clone(const void *t) { strobe *s = (strobe *)malloc(sizeof(strobe)); if (s) memcpy(s, t, sizeof(strobe)); return s; }
API void mt_free(void *t) { free(t); }
API void mt_append_message(void *ctx, const uint8_t *label, size_t label_len, const uint8_t *bytes, size_t len) { merlin_append((strobe *)ctx, label, label_len, bytes, len); }
API void mt_challenge_bytes(void *ctx, const uint8_t *label, size_t label_len, uint8_t *out, size_t n) { merlin_challenge((strobe *)ctx, label, label_len, out, n); }
API void mt_challenge_scalar(void *ctx, const uint8_t *label, size_t label_len, uint64_t out[4]) {
    for (;;) {
        uint8_t buf[64]; uint64_t v[4];
        merlin_challenge((strobe *)ctx, label, label_len, buf, 64);
        memcpy(v, buf, 32);                              /* little-endian host */
        v[3] &= 0x7fffffffffffffffULL;
        if (is_zero4(v) || ge4(v, RMOD)) continue;
        inv_mod_r(out, v);
        return;
    }
}
/* the state, for the byte-for-byte comparison with transcript.py */
API void mt_state(const void *t, uint8_t out[203]) { const strobe *s = (const strobe *)t; memcpy(out, s->st, 200); out[200] = s->pos; out[201] = s->pos_begin; out[202] = s->cur_flags; }
