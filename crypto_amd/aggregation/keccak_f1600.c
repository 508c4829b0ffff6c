/* crypto_amd/aggregation/keccak_f1600.c — Keccak-f[1600] for the Python transcript mirror (crypto_amd/aggregation/transcript.py).
 *
 * NOT part of the product ABI (include/dock_gpu.h): Merlin is out of scope; a Rust host keeps its own
 * merlin/src/strobe.rs:97-104 (`run_f` -> keccak::f1600).  Built by __graft_entry__.build() into libkeccak_f1600.so next to
 * this file; transcript.py falls back to its pure-Python statement of the permutation when the helper is absent. */
#include <stdint.h>
#include <string.h>

static uint64_t rol(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }

__attribute__((visibility("default"))) int keccak_f1600(uint8_t state[200]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL, 0x0000000080000001ULL,
        0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
        0x000000000000800AULL, 0x800000008000000AULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};   /* index x + 5y */
    uint64_t a[25];
    if (!state) return -1;
    memcpy(a, state, 200);                                  /* little-endian host */
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];
    }
    memcpy(state, a, 200);
    return 0;
}
