/* include/dock_gpu.h — C ABI of libdock_gpu.so, the MI355X (gfx950) backend for the one data-parallel
 * hot path of docknetwork/crypto: BLS12-381 variable-base MSM (G1/G2) and the batched Miller loop.
 *
 * The reference has no FFI for this path today: every call is a statically dispatched call into
 * arkworks (ark-ec 0.4).  Each entry point below names the reference call it replaces; the Rust-side
 * binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - return int32_t: 0 = DGPU_OK, negative = DGPU_E_*; never unwinds, aborts or prints.
 *   - field elements are ark-ff `Fp<MontBackend,N>` raw limbs: little-endian u64, value * R mod p,
 *     R = 2^384 for Fq (6 limbs), R = 2^256 for Fr (4 limbs)   (SURVEY.md A.5).
 *   - G1 affine = x,y (12 u64); G2 affine = x.c0,x.c1,y.c0,y.c1 (24 u64); identity is flagged out of
 *     band by `is_inf[i] != 0` (ark-ec `Affine{ x, y, infinity }` is not repr(C), the shim repacks);
 *     a point whose coordinate words are all zero is also treated as the identity.
 *   - group results are returned as a Jacobian triple X,Y,Z (ark-ec `Projective`): always the
 *     normalised representative (Z = R, i.e. one) or Z = 0 for the identity, so equal group elements
 *     give bit-identical output.
 *   - caller owns every buffer; nothing is retained after return except behind explicit handles.
 *   - thread-safe and re-entrant (the reference calls MSM from inside rayon workers,
 *     verifiable_encryption/src/tz_21/rdkgith.rs:140-147): up to six calls per device run concurrently (one
 *     slot = streams + workspace each); callers beyond six wait for a slot and are served first come, first served
 *     (the MSM rate does not depend on the number of calling threads: tests/native/inflight_threads.cpp).
 */
#ifndef DOCK_GPU_H
#define DOCK_GPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)      /* the library is built with -fvisibility=hidden: these entry points are its whole surface */

#define DGPU_OK            0
#define DGPU_E_NODEVICE   -1   /* no usable HIP device / dgpu_init not called or failed */
#define DGPU_E_OOM        -2   /* device or host allocation failed */
#define DGPU_E_BADARG     -3   /* NULL pointer, bad handle, offset + n out of range, n too large */
#define DGPU_E_HIP        -4   /* a HIP runtime call failed (see dgpu_last_hip_error) */
#define DGPU_E_ZERO       -5   /* final_exponentiation of 0 (arkworks returns None) */
#define DGPU_E_TOO_SMALL  -6   /* n below dgpu_set_min_gpu_n threshold: caller should stay on its CPU path */
#define DGPU_E_LENGTH     -7   /* multi_miller_loop with unequal lengths (arkworks zip_eq panics) */

/* ---- lifecycle ---- */
/* Optional, before anything else: process-level settings of the ROCm runtime that suit this library.  The library never changes its host's environment on
 * its own; a host that wants the setting calls this from its start-up code — before the process's first HIP call (the runtime reads its configuration once)
 * and while no other thread can be inside getenv / setenv.  DGPU_E_BADARG: unknown flag, or this library has already made a HIP call (too late).
 *   DGPU_HINT_EIGHT_HW_QUEUES   export GPU_MAX_HW_QUEUES=8 unless the variable is set: the runtime maps a process's streams onto that many hardware
 *                               queues (default 4); with up to six calls in flight, each on three streams, two streams on one queue run one after the
 *                               other (Miller loop -7 %, aggregation -5 %, MSM rate unchanged: profiles/r05_hwq_bench_ab.txt).  A host that has already
 *                               initialised HIP (PyTorch) exports the variable itself instead. */
#define DGPU_HINT_EIGHT_HW_QUEUES 1u
int32_t dgpu_runtime_hints(uint32_t flags);
/* fork(): HIP state does not survive it.  fork + exec is safe (the library's worker pool re-arms itself in the child); a child that goes on to USE the library
 * must bring it up again itself (dgpu_shutdown, then dgpu_init*) — handles, streams and resident tables of the parent mean nothing there. */
int32_t dgpu_init(int32_t device);           /* bind to a HIP device ordinal (one process per GPU); idempotent */
/* Several GPUs in ONE process (a Rust host is one process): one context per device, each with its own streams and
 * workspaces.  dgpu_init_devices(mask): bit d = use HIP device d; contexts are numbered 0.. in increasing device order.
 * dgpu_init_device_list: context k on physical[k] (a device may be listed twice: two contexts on one GPU, used by the tests on a
 * one-GPU box).  Entry points that take host pointers run on the calling thread's context (dgpu_set_device, thread-local, default =
 * context 0); entry points that take a handle run on the context that owns the handle. */
int32_t dgpu_init_devices(uint32_t device_mask);
int32_t dgpu_init_device_list(const int32_t *physical, int32_t count);
int32_t dgpu_context_count(void);
int32_t dgpu_set_device(int32_t context);
int32_t dgpu_shutdown(void);
int32_t dgpu_device_count(void);
const char *dgpu_strerror(int32_t code);
int32_t dgpu_last_hip_error(void);
/* below this many terms the MSM entry points return DGPU_E_TOO_SMALL without touching the device
 * (>= 95 % of the reference's call sites have n < 100).  Default: DGPU_DEFAULT_MIN_GPU_N, the measured
 * crossover against the CPU path; 0 = always run on the device. */
#define DGPU_DEFAULT_MIN_GPU_N 256
/* MSMs over a bases HANDLE (dgpu_msm_*_handle, dgpu_msm_*_resident) are refused only below min(threshold, DGPU_MIN_GPU_N_HANDLE) terms: the handle
 * carries the small path's table, a call is one launch (0.14 ms at 16 terms; one CPU thread: 0.64 ms), and a caller who uploaded bases wants them used. */
#define DGPU_MIN_GPU_N_HANDLE 8
int32_t dgpu_set_min_gpu_n(size_t n);
size_t dgpu_get_min_gpu_n(void);
/* Several devices behind the UNMODIFIED call: with more than one device context in the process (dgpu_init_devices) the one-shot entry points
 * (dgpu_msm_g1 / _g2 [_mont | _strided]) shard calls of at least n terms over all of them — context k takes the contiguous balanced chunk k of the terms
 * on a host thread inside the call (from its own resident copy of that chunk once the resident-bases cache holds it: each device caches ITS chunk of a
 * key), the partial points are folded on the host: BASELINE config 5's shape (2^24 terms over 8 GPUs) from `msm_bigint(&[G1Affine], ..)`.  Same group
 * element as on one device.  n = 0: off (the default — host-pointer calls then run on the calling thread's context, dgpu_set_device). */
int32_t dgpu_set_auto_shard_min_n(size_t n);
/* (tuning knobs of the kernels — window width, chunk length, reduction geometry, Miller-loop forms —, stage timers and self-test hooks are NOT part
 * of this ABI: include/dock_gpu_dev.h, served by the development twin libdock_gpu_dev.so.  The product runs every knob at its measured optimum.) */
/* MSMs of up to n terms (default and maximum 8192) over plain bases — one-shot calls, plain handles — run as 64 signed 4-bit windows, each a
 * tree over the terms' table entries (a table launch and a tree launch: crypto_amd/csrc/small_kernels.hip.h) instead of the ~15-launch bucket
 * pipeline: 0.27 - 0.6 ms instead of 0.7 - 0.9 ms per call.  A plain HANDLE of up to 8192 bases gets a table of its own when it is uploaded (one
 * made another way: at its second small MSM, or by dgpu_bases_precompute_*) — the eight multiples of P, 2^64 P, 2^128 P, 2^192 P per base, 6.5 KB (G1) / 13 KB (G2)
 * per base, released with the handle — and its calls are ONE launch of 16 trees over 4 n leaves and a 60-doubling fold on the host:
 * 0.14 - 0.4 ms (G1 n = 600 / 4096: 0.19 / 0.29 ms; G2: 0.37 / 0.57).  If that allocation fails the calls keep building their per-call table.
 * 0: always the bucket pipeline (the parity tests compare the two). */
int32_t dgpu_set_small_msm_max(size_t n);
/* Workspaces.  Every call in flight owns one of the context's slots (stream + grow-only device workspace).  The slots are sized AHEAD of
 * the calls so that no MSM path allocates in steady state (a hipMalloc costs 0.1 - 1 ms and the hipFree of the buffer it replaces waits for
 * the whole device, i.e. for every other call in flight): dgpu_bases_upload_* / dgpu_bases_precompute_* / dgpu_window_table_mul_to_bases_*
 * size every slot for MSMs over that handle; dgpu_reserve_g1/g2(n) does it for one-shot calls (dgpu_msm_*) of up to n terms on the calling
 * thread's context — what a Rust host calls once at start-up; without it the first one-shot call of a new size grows its own slot and
 * every idle one.  dgpu_device_alloc_count: hipMalloc calls issued by the library so far:
 * a steady-state delta of 0 is asserted by tests/test_gpu_reserve.py. */
int32_t dgpu_reserve_g1(size_t n);
int32_t dgpu_reserve_g2(size_t n);
uint64_t dgpu_device_alloc_count(void);

/* ---- one-shot MSM: host buffers in, one point out ----
 * replaces <G1Projective as VariableBaseMSM>::msm_bigint(bases, bigints)
 *   legogroth16/src/prover.rs:286,299,363,592 ; utils/src/pairs.rs:153-155 ; utils/src/owned_pairs.rs:103-105
 * Callers pass n = min(bases.len(), scalars.len()) — the truncation arkworks applies (prover.rs:286).
 * Preconditions (DGPU_E_BADARG otherwise, nothing is allocated): n < 2^31 and n * W < 2^32 where W = 255 / c + 1 is the number of
 * windows (c = 16 from n = 2^17 on: n < 2^28; split larger MSMs and fold the parts with dgpu_fold_*).  Scalars are 255-bit values
 * (Fr::MODULUS_BIT_SIZE): scalars >= r but < 2^255 are multiplied as the integers they are (the same group element as for the reduced
 * scalar, and what arkworks computes).  A scalar with bit 255 set has no meaning that is independent of arkworks' window width — its
 * `make_digits` reads bit 255 in the top window unless the width divides 255 (n <= 32, 2^18 < n <= 2^20, 2^21 < n <= 2^23
 * ignore it, every other n multiplies by the full 256-bit integer) — so every MSM entry point REFUSES such input with DGPU_E_BADARG and the
 * caller stays on its CPU path; `Fr::into_bigint()` never produces it.  (tests/test_gpu_msm.py::test_scalars_with_bit_255) */
int32_t dgpu_msm_g1(const uint64_t *bases_xy /* n*12 */, const uint8_t *is_inf /* n or NULL */,
                    const uint64_t *scalars /* n*4, canonical */, size_t n, uint64_t out_xyz[18]);
/* replaces <G1Projective as VariableBaseMSM>::msm_unchecked(bases, &[Fr]) (Fr in Montgomery form, R = 2^256)
 *   utils/src/pairs.rs:145-147 ; utils/src/randomized_mult_checker.rs:100 ; schnorr_pok/src/pok_generalized_pedersen.rs:97 */
int32_t dgpu_msm_g1_mont(const uint64_t *bases_xy, const uint8_t *is_inf,
                         const uint64_t *scalars_mont, size_t n, uint64_t out_xyz[18]);
/* The same straight from the caller's `&[G1Affine]` / `&[G2Affine]`: ark-ec 0.4 `Affine<P> { x, y, infinity }` is a Rust struct of
 * 104 (G1) / 200 (G2) bytes whose field offsets the shim reads with core::mem::offset_of! — no host-side repacking into x|y arrays (the
 * 96-MB copy loop of a 2^20-term call; utils/src/pairs.rs:143-156 and the 166 direct call sites hand over exactly this slice).  Point i
 * lives at bases + i * stride_bytes: x at x_off, y at y_off (48 / 96 bytes of Montgomery limbs each, 8-byte aligned), the `infinity`
 * bool at inf_off (DGPU_NO_INF_OFF: no flag byte; all-zero coordinates still mean the identity).  montgomery != 0: scalars are &[Fr].
 * The whole array crosses PCIe as it is (stride_bytes per point) and is converted on the device. */
#define DGPU_NO_INF_OFF (~(size_t)0)
int32_t dgpu_msm_g1_strided(const void *bases, size_t stride_bytes, size_t x_off, size_t y_off, size_t inf_off,
                            const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t out_xyz[18]);
int32_t dgpu_msm_g2_strided(const void *bases, size_t stride_bytes, size_t x_off, size_t y_off, size_t inf_off,
                            const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t out_xyz[36]);
int32_t dgpu_bases_upload_g1_strided(const void *bases, size_t stride_bytes, size_t x_off, size_t y_off, size_t inf_off, size_t n, uint64_t *handle);
int32_t dgpu_bases_upload_g2_strided(const void *bases, size_t stride_bytes, size_t x_off, size_t y_off, size_t inf_off, size_t n, uint64_t *handle);
/* same pair over G2 — legogroth16/src/prover.rs:344 (b_g2_query), delegatable_credentials/src/set_commitment.rs:566 */
int32_t dgpu_msm_g2(const uint64_t *bases_xy /* n*24 */, const uint8_t *is_inf,
                    const uint64_t *scalars, size_t n, uint64_t out_xyz[36]);
int32_t dgpu_msm_g2_mont(const uint64_t *bases_xy, const uint8_t *is_inf,
                         const uint64_t *scalars_mont, size_t n, uint64_t out_xyz[36]);

/* ---- the resident-bases cache behind the one-shot entry points above ----
 * The reference's call sites pass the same proving-key slices proof after proof (legogroth16/src/prover.rs:286,299,363,592 `msm_bigint(&pk.h_query, ..)`,
 * `&query[1..]`; utils/src/pairs.rs:143-156) and know nothing of handles.  So that the UNMODIFIED call reaches the resident path, dgpu_msm_g1 / _g2
 * [_mont | _strided] with n >= the cache's threshold (default 2^16) remember what they were given: the first sighting of (pointer, n, layout) runs one-shot
 * and keeps a 64-bit fingerprint of 32 evenly spaced records; the second sighting with the same fingerprint uploads the points once, keeps a fingerprint of
 * every record (8 B each, host memory) and turns them into a precomputed-multiples table (dgpu_bases_precompute_*: ~40 ms at 2^20 G1 points, once per key);
 * from then on the call is dgpu_msm_*_handle on that table — only the scalars cross PCIe (2^20 G1 terms: 5.3 -> ~2.8 ms).  A call whose points lie
 * INSIDE a resident entry of the same layout (`&query[1..]`, a shorter n) resolves to (entry, offset).  Same group element, limb for limb, either way.
 * Stale keys: before a resident entry's result is used the host re-fingerprints the records of the call's range and compares with what was uploaded; a difference
 * evicts the entry and the call runs one-shot on what the buffer holds now.  DEFAULT = DGPU_CACHE_VERIFY_FULL: every record, on the library's host threads, BESIDE the
 * MSM on the resident copy, whose result is discarded if the check fails (1.5 ms of host work per 2^20 G1 points; 3.21 -> 3.22 ms per call;
 * dgpu_legogroth16_prove_host checks its views beside the proof the same way: 10.0 -> 10.4 - 10.8 ms) — the unmodified call never answers from a key that has changed.
 * A host whose key cannot change between calls may select the sampled mode (s records per use, its first and last among them): a refilled buffer is still
 * noticed at once, an IN-PLACE edit of a few records only with probability s / n per call.  dgpu_bases_cache_invalidate drops entries by address range.
 * (Rust: a `&[G1Affine]` borrowed from a `ProvingKey` cannot change while borrowed; between calls it can.)
 *   dgpu_set_bases_cache_bytes(b)   device bytes the cache may hold (least recently used entries go first; an entry in use is never freed under its
 *                                   user); 0 = off and emptied.  Default (DGPU_CACHE_BYTES_AUTO): a quarter of the device memory that is free at the
 *                                   cache's first use.
 *                                   Whatever the budget, the cache gives way when the device is full: a failed allocation of the library releases
 *                                   least recently used entries and is tried again.
 *   dgpu_set_bases_cache_min_n(n)   calls below n terms are never cached (default 65536)
 *   dgpu_set_bases_cache_verify(s)  DGPU_CACHE_VERIFY_FULL (default), or s = 2 .. 4096 records sampled per use
 *   dgpu_bases_cache_invalidate     forget every entry that overlaps [p, p + bytes)
 *   dgpu_bases_cache_stats          out[0..7] = hits, misses (cacheable calls that ran one-shot), fills, stale evictions, budget evictions, bytes in use,
 *                                   byte budget (0 until resolved), resident entries */
#define DGPU_CACHE_VERIFY_FULL (-1)
#define DGPU_CACHE_BYTES_AUTO (~(size_t)0)
int32_t dgpu_set_bases_cache_bytes(size_t bytes);
int32_t dgpu_set_bases_cache_min_n(size_t n);
int32_t dgpu_set_bases_cache_verify(int32_t samples);
int32_t dgpu_bases_cache_invalidate(const void *p, size_t bytes);
int32_t dgpu_bases_cache_clear(void);
int32_t dgpu_bases_cache_stats(uint64_t out[8]);

/* ---- device-resident operands (proving-key queries live in HBM across proofs) ----
 * `offset` expresses `&query[1..]` (legogroth16/src/prover.rs:592). */
int32_t dgpu_bases_upload_g1(const uint64_t *bases_xy, const uint8_t *is_inf, size_t n, uint64_t *handle);
int32_t dgpu_bases_upload_g2(const uint64_t *bases_xy, const uint8_t *is_inf, size_t n, uint64_t *handle);
int32_t dgpu_bases_free(uint64_t handle);
int32_t dgpu_scalars_upload(const uint64_t *scalars /* n*4 */, size_t n, int32_t montgomery, uint64_t *handle);
int32_t dgpu_scalars_free(uint64_t handle);
/* one resident scalar vector from several host arrays, in order (the prover's `assignment` = inputs[1..] ++ witnesses,
 * legogroth16/src/prover.rs:319-321, without concatenating on the host) */
int32_t dgpu_scalars_upload_parts(const uint64_t *const *parts, const size_t *counts, size_t n_parts, int32_t montgomery, uint64_t *handle);
int32_t dgpu_msm_g1_handle(uint64_t bases, size_t offset, const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t out_xyz[18]);
int32_t dgpu_msm_g2_handle(uint64_t bases, size_t offset, const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t out_xyz[36]);
/* Precomputed-multiples mode for a resident query (in place; the handle keeps its id): the device builds table[w][i] = 2^(c w) P_i for the
 * W = 255 / c + 1 windows (W x the memory: 1.7 GB for a 2^20-point G1 query at c = 20 — sized for 288 GB of HBM), after which every MSM on
 * the handle adds digit w of scalar i into ONE bucket set shared by all windows: wider windows (13 instead of 16 additions per term at
 * n = 2^20), 16x fewer buckets to reduce, no Horner fold on the host.  Same group element as before, limb for limb.  window_bits: 0 =
 * chosen from n, else 16..22.  Works on plain (dgpu_bases_upload_*, dgpu_window_table_mul_to_bases_*) and sharded handles; offsets
 * (`&query[1..]`) and sub-ranges keep working.  While the table is being built the handle is unavailable (DGPU_E_BADARG).
 * A handle too small for a bucket table to pay (below 2^15 points with window_bits = 0) stays plain; with up to 8192 points it gets the small
 * path's table (dgpu_set_small_msm_max above) if its upload has not built it already.
 * The automatic width (20 from 320 000 points) is the optimum for full-width scalars.  A query that is multiplied by a WITNESS (a proving key's
 * a / b / l queries: half of a Groth16 witness is 0 or 1, much of the rest small) adds fewer points per MSM but reduces the same number of
 * buckets, and a proof reduces five bucket sets: DGPU_TABLE_C_WITNESS = 17 has the window count of 18 (15) with half the buckets — measured
 * at 2^20 constraints: 12.4 -> 11.75 ms per proof, 10.2 -> 9.6 with four in flight (19: 12.5, 18: 12.0, 16: 12.9). */
#define DGPU_TABLE_C_WITNESS 17
int32_t dgpu_bases_precompute_g1(uint64_t bases, int32_t window_bits);
int32_t dgpu_bases_precompute_g2(uint64_t bases, int32_t window_bits);
/* ---- one partition sort for several tables ----
 * A, B-in-G1 and B-in-G2 of a LegoGroth16 proof multiply the same assignment by three queries of equal length (legogroth16/src/prover.rs:325-344
 * through calculate_coeff :585-594).  With the queries held as precomputed tables of ONE shape (same window width, same row count) the sort of
 * the (window, bucket) keys is the same for all three: dgpu_scalars_sort does it once — for scalars [scalar_offset, scalar_offset + n) against rows
 * [base_offset, base_offset + n) of any table of that shape (G1 or G2) — and dgpu_msm_g1/g2_sorted run the rest of the MSM on a table of
 * that shape (DGPU_E_BADARG otherwise).  Identity rows of a table are passed over by the accumulation.  The result is the group element
 * dgpu_msm_*_resident returns for the same operands.  row_shift = k > 0: the table is k rows SHORTER than that shape and holds the points
 * of its rows k, k + 1, ... (the l_query against the list sorted for the a_query: the assignment minus its first k entries, prover.rs:292-299);
 * the terms of rows < k are passed over.  Free the sorted handle with dgpu_scalars_free once the MSMs have returned. */
int32_t dgpu_bases_table_shape(uint64_t table, size_t *rows, int32_t *window_bits, int32_t *windows);   /* DGPU_E_BADARG: not a precomputed table */
int32_t dgpu_scalars_sort(uint64_t table, size_t base_offset, uint64_t scalars, size_t scalar_offset, size_t n, uint64_t *sorted_handle);
int32_t dgpu_msm_g1_sorted(uint64_t table, uint64_t sorted, size_t row_shift, uint64_t out_xyz[18]);
int32_t dgpu_msm_g2_sorted(uint64_t table, uint64_t sorted, size_t row_shift, uint64_t out_xyz[36]);
/* both operands resident: the timed region of bench.py (inputs already in HBM) */
int32_t dgpu_msm_g1_resident(uint64_t bases, size_t base_offset, uint64_t scalars, size_t scalar_offset, size_t n, uint64_t out_xyz[18]);
int32_t dgpu_msm_g2_resident(uint64_t bases, size_t base_offset, uint64_t scalars, size_t scalar_offset, size_t n, uint64_t out_xyz[36]);

/* ---- several GPUs behind the ABI (SURVEY.md 8b `dgpu_msm_g1_sharded`, 8e): point-chunk sharding inside one process ----
 * Context k (see dgpu_init_devices) takes the contiguous balanced chunk k of the n terms, runs the whole single-GPU pipeline on it
 * from its own host thread inside the call, and the per-device partial points (144 / 288 B) are folded on the host — bit-identical
 * to the single-device result (the ABI returns the normalised representative).  ngpus = 0: every initialised context.
 *   dgpu_msm_*_sharded            one-shot: host bases + scalars in (each device pulls its own chunk over its own PCIe link)
 *   dgpu_bases_upload_*_sharded   a proving-key query resident across the devices; free with dgpu_bases_free
 *   dgpu_msm_*_sharded_handle     fresh host scalars against such a handle (n <= number of bases: the first n terms)
 *   dgpu_scalars_upload_sharded   scalars split the way the bases handle `like` is split; free with dgpu_scalars_free
 *   dgpu_msm_*_sharded_resident   both operands resident (BASELINE config 5's timed region: inputs pre-sharded) */
int32_t dgpu_msm_g1_sharded(const uint64_t *bases_xy, const uint8_t *is_inf, const uint64_t *scalars, size_t n, int32_t ngpus, uint64_t out_xyz[18]);
int32_t dgpu_msm_g2_sharded(const uint64_t *bases_xy, const uint8_t *is_inf, const uint64_t *scalars, size_t n, int32_t ngpus, uint64_t out_xyz[36]);
int32_t dgpu_bases_upload_g1_sharded(const uint64_t *bases_xy, const uint8_t *is_inf, size_t n, int32_t ngpus, uint64_t *handle);
int32_t dgpu_bases_upload_g2_sharded(const uint64_t *bases_xy, const uint8_t *is_inf, size_t n, int32_t ngpus, uint64_t *handle);
int32_t dgpu_msm_g1_sharded_handle(uint64_t bases, const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t out_xyz[18]);
int32_t dgpu_msm_g2_sharded_handle(uint64_t bases, const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t out_xyz[36]);
int32_t dgpu_scalars_upload_sharded(const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t like, uint64_t *handle);
/* scalars [lo, hi) of a resident vector as a new vector on device context `dst_context`: device to device (a peer copy over xGMI between two
 * GPUs; peer access is enabled between the process's devices at init).  dgpu_legogroth16_prove on a sharded key moves h this way. */
int32_t dgpu_scalars_copy_range(uint64_t scalars, size_t lo, size_t hi, int32_t dst_context, uint64_t *handle);
int32_t dgpu_msm_g1_sharded_resident(uint64_t bases, uint64_t scalars, uint64_t out_xyz[18]);
int32_t dgpu_msm_g2_sharded_resident(uint64_t bases, uint64_t scalars, uint64_t out_xyz[36]);

/* ---- multi-process multi-GPU: fold per-rank partial results (host code; EC addition is not an RCCL reduction op, so the
 * collective is an all-gather of k normalised Jacobian triples followed by this fold) ---- */
int32_t dgpu_fold_g1(const uint64_t *partials_xyz /* k*18 */, size_t k, uint64_t out_xyz[18]);
int32_t dgpu_fold_g2(const uint64_t *partials_xyz /* k*36 */, size_t k, uint64_t out_xyz[36]);
/* sum_i scalars[i] * points[i] over k <= DGPU_MAX_LINCOMB affine points, computed on the HOST (no device call, usable without a device).
 * This is the O(1) group arithmetic around the MSMs that the reference does with `mul_bigint` on the CPU, not a replacement for them:
 * delta_g1 * r, s g_a + r g1_b - rs delta - v eta/delta, g_d (legogroth16/src/prover.rs:309-313, 350-355, 361-368, 585-594).
 * points: k x 12 (G1) / 24 (G2) u64 Montgomery, is_inf: k bytes or NULL, scalars: k x 4 u64 canonical; out: normalised Jacobian. */
#define DGPU_MAX_LINCOMB 16
int32_t dgpu_lincomb_g1(const uint64_t *points_xy, const uint8_t *is_inf, const uint64_t *scalars, size_t k, uint64_t out_xyz[18]);
int32_t dgpu_lincomb_g2(const uint64_t *points_xy, const uint8_t *is_inf, const uint64_t *scalars, size_t k, uint64_t out_xyz[36]);

/* ---- pairings ----
 * replaces Bls12_381::multi_miller_loop(a, b) — utils/src/randomized_pairing_check.rs:207,
 * legogroth16/src/verifier.rs:69-76.  skip[i] != 0 marks a pair with an identity member (arkworks
 * filters those out).  Output: raw MillerLoopOutput, Fp12 as c0.c0.c0 ... c1.c2.c1 (72 u64). */
int32_t dgpu_multi_miller_loop(const uint64_t *p_xy /* n*12 */, const uint64_t *q_xy /* n*24 */,
                               const uint8_t *skip /* n or NULL */, size_t n, uint64_t out_f12[72]);
/* E::G2Prepared — the form in which the reference's verifier and pairing checker HOLD their G2 operands
 * (legogroth16/src/data_structures.rs:118-120 `gamma_g2_neg_pc` / `delta_g2_neg_pc`, verifier.rs:22-23,69-76;
 * utils/src/randomized_pairing_check.rs:35 `pending: (Vec<G1Prepared>, Vec<G2Prepared>)`, :119-138,:204-214).
 * One prepared point = arkworks' `ell_coeffs`: 68 triples (Fp2, Fp2, Fp2), each Fq 6 x u64 Montgomery limbs in the order
 * c0.c0 c0.c1 c1.c0 c1.c1 c2.c0 c2.c1 -> DGPU_G2_PREPARED_WORDS u64 = 19 584 B; `infinity` travels as a flag byte and the
 * block of an identity point is all zero.
 * dgpu_g2_prepare replaces `G2Prepared::from(G2Affine)` for a batch (the 63 doubling + 5 addition steps run on the GPU);
 * dgpu_multi_miller_loop_prepared replaces `E::multi_miller_loop(a, b)` where b is already prepared: the same raw
 * MillerLoopOutput, limb for limb, as dgpu_multi_miller_loop on the unprepared points (the reference asserts exactly this
 * for pairings, utils/src/msm.rs:261-276).  skip[i] != 0, an all-zero P or an all-zero coefficient block skip pair i. */
#define DGPU_G2_PREPARED_WORDS (68 * 36)
#define DGPU_MAX_PREPARED ((size_t)1 << 18)     /* pairs per call (5 GB of coefficients) */
int32_t dgpu_g2_prepare(const uint64_t *q_xy /* n*24 */, const uint8_t *is_inf /* n or NULL */, size_t n,
                        uint64_t *out_coeffs /* n*DGPU_G2_PREPARED_WORDS */, uint8_t *out_inf /* n */);
int32_t dgpu_multi_miller_loop_prepared(const uint64_t *p_xy /* n*12 */, const uint64_t *coeffs /* n*DGPU_G2_PREPARED_WORDS */,
                                        const uint8_t *skip /* n or NULL */, size_t n, uint64_t out_f12[72]);
/* The same product over n_aff pairs with affine Q and n_prep pairs with prepared Q in ONE call (no separate dgpu_g2_prepare, no coefficient
 * round trip for the affine members): what legogroth16/src/verifier.rs:69-76 passes to multi_miller_loop — `[proof.b.into(),
 * pvk.delta_g2_neg_pc.clone(), pvk.gamma_g2_neg_pc.clone()]` — and what a RandomizedPairingChecker's `pending` holds when some operands
 * came in prepared (utils/src/randomized_pairing_check.rs:119-138).  The product does not depend on the order of the pairs. */
int32_t dgpu_multi_miller_loop_mixed(const uint64_t *p_aff /* n_aff x 12 */, const uint64_t *q_aff /* n_aff x 24 */, const uint8_t *skip_aff, size_t n_aff,
                                     const uint64_t *p_prep /* n_prep x 12 */, const uint64_t *coeffs /* n_prep x DGPU_G2_PREPARED_WORDS */, const uint8_t *skip_prep, size_t n_prep,
                                     uint64_t out_f12[72]);
/* prod_i e([m_i] P_i, Q_i) x prod_j e(P'_j, prepared_j): the scalings of RandomizedPairingChecker and its Miller loop as ONE call
 * (utils/src/randomized_pairing_check.rs:125-134 `a.mul_bigint(m)` per source, then :204-214 the lazy multi_miller_loop) — limb for limb what
 * dgpu_g1_scale_batch followed by dgpu_multi_miller_loop_mixed returns.  The line coefficients of a pair depend on Q alone, so the chain of the Q_i
 * and the scaling chains of the P_i run side by side and the scaled points never visit the host (1024 pairs: 1.64 -> 1.34 ms).
 * scalars: n_aff x 4 canonical words (scalar_stride = 4) or ONE scalar for every pair (scalar_stride = 0), reduced mod r; a pair whose
 * scaled point is the identity (m = 0 mod r, P all zero), whose Q is all zero or whose skip flag is set contributes one.  P_i in the
 * prime-order subgroup (the invariant of arkworks' G1Affine: the scaling uses the endomorphism). */
int32_t dgpu_multi_miller_loop_scaled(const uint64_t *p_aff /* n_aff x 12 */, const uint64_t *scalars, size_t scalar_stride, const uint64_t *q_aff /* n_aff x 24 */,
                                      const uint8_t *skip_aff, size_t n_aff,
                                      const uint64_t *p_prep /* n_prep x 12 */, const uint64_t *coeffs /* n_prep x DGPU_G2_PREPARED_WORDS */, const uint8_t *skip_prep, size_t n_prep,
                                      uint64_t out_f12[72]);
/* nseg independent Miller loops in one call: segment g is the pairs [seg_end[g - 1], seg_end[g]) (ascending, seg_end[nseg - 1] == n; an empty
 * segment yields one); out_f12 = nseg x 72 words, each what dgpu_multi_miller_loop returns for that segment alone.  Serves the
 * mutually independent `E::multi_pairing` calls the aggregation issues one after another (legogroth16/src/aggregation/commitment.rs:30-31,54-67,
 * aggregation/utils.rs:95-96: ten per GIPA round, 1 ... n/2 pairs each): a line-kernel launch lasts as long as its 68 dependent steps
 * whatever the pair count, so the segments share one. */
int32_t dgpu_multi_miller_loop_segments(const uint64_t *p_xy, const uint64_t *q_xy, const uint8_t *skip, size_t n,
                                        const uint64_t *seg_end, size_t nseg, uint64_t *out_f12 /* nseg x 72 */);
/* the same followed by the final exponentiation of every segment (E::multi_pairing), on the host threads that assemble the Miller outputs;
 * DGPU_E_ZERO if a Miller output is zero (arkworks' multi_pairing would panic on the `None`: cannot happen for points of G1 x G2) */
int32_t dgpu_multi_pairing_segments(const uint64_t *p_xy, const uint64_t *q_xy, const uint8_t *skip, size_t n,
                                    const uint64_t *seg_end, size_t nseg, uint64_t *out_gt /* nseg x 72 */);
/* the same with the pairs chunked over the process's device contexts (ngpus = 0: all of them), raw outputs multiplied on the host */
int32_t dgpu_multi_miller_loop_sharded(const uint64_t *p_xy, const uint64_t *q_xy, const uint8_t *skip, size_t n, int32_t ngpus, uint64_t out_f12[72]);
/* replaces Bls12_381::final_exponentiation — utils/src/randomized_pairing_check.rs:213 (host code, once per batch) */
int32_t dgpu_final_exponentiation(const uint64_t in_f12[72], uint64_t out_f12[72]);

/* ---- pieces of utils::randomized_pairing_check::RandomizedPairingChecker (utils/src/randomized_pairing_check.rs:24-215) ----
 * out_i = s_i * P_i as affine points (the per-equation `a.mul_bigint(m)` scalings, :125-127,152-158), batched on the GPU.
 * scalar_stride = 4: one canonical scalar per point; scalar_stride = 0: the same scalar for every point.
 * negate[i] != 0 returns -(s_i P_i) (the `-c.mul_bigint(m)` of add_multiple_sources, :156-158). */
int32_t dgpu_g1_scale_batch(const uint64_t *p_xy /* n*12 */, const uint8_t *is_inf, const uint64_t *scalars, size_t scalar_stride,
                            const uint8_t *negate, size_t n, uint64_t *out_xy /* n*12 */, uint8_t *out_inf /* n */);
/* GT arithmetic on the host: PairingOutput `+` is the Fp12 product, `mul_bigint` the power (:136 `self.right += out.mul_bigint(m)`) */
int32_t dgpu_fp12_mul(const uint64_t a[72], const uint64_t b[72], uint64_t out[72]);
int32_t dgpu_fp12_pow(const uint64_t a[72], const uint64_t e[4], uint64_t out[72]);
/* ok[i] = a_i is an element of GT (order dividing r): `Valid::check` of ark-ec's PairingOutput (a proof deserialized with Validate::Yes), answered
 * with a Frobenius identity and one exponentiation by the 64-bit curve parameter (0.09 ms per element on a host thread) instead of f^r (1.3 ms) */
int32_t dgpu_gt_in_subgroup(const uint64_t *a /* n*72 */, size_t n, uint8_t *ok /* n */);
/* prod_i a_i^{e_i}: the fold of `PairingOutput::mul_bigint` + `add_assign` in the aggregation verifier
 * (legogroth16/src/aggregation/groth16/verifier.rs:272-370); host threads, generic Fp12 arithmetic */
int32_t dgpu_fp12_multi_pow(const uint64_t *a /* n*72 */, const uint64_t *e /* n*4 */, size_t n, uint64_t out[72]);

/* ---- fixed-base batch multiplication ----
 * replaces ark-ec FixedBase::{get_window_table, msm} as the reference calls them: WindowTable::new / multiply_many and
 * multiply_field_elems_with_same_group_elem (utils/src/msm.rs:8-62) and the six query computations of the LegoGroth16 CRS
 * generator followed by normalize_batch (legogroth16/src/generator.rs:335-399,424-431).
 * out_i = scalars_i * base as affine points (x, y Montgomery limbs; zeros and out_inf[i] = 1 for the identity).
 * montgomery != 0: scalars are Fr Montgomery limbs (what &[Fr] holds), else canonical.  The table handle keeps
 * 32 x 255 multiples of the base in device memory (1 MiB for G1, 2 MiB for G2). */
int32_t dgpu_window_table_g1(const uint64_t base_xy[12], uint64_t *handle);
int32_t dgpu_window_table_g2(const uint64_t base_xy[24], uint64_t *handle);
int32_t dgpu_window_table_free(uint64_t handle);
int32_t dgpu_window_table_mul_g1(uint64_t table, const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t *out_xy /* n*12 */, uint8_t *out_inf /* n */);
int32_t dgpu_window_table_mul_g2(uint64_t table, const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t *out_xy /* n*24 */, uint8_t *out_inf /* n */);
/* same products left in device memory as an MSM bases handle (dgpu_msm_g1/g2_handle, dgpu_bases_free): a CRS query goes
 * from the generator to the prover without crossing PCIe */
int32_t dgpu_window_table_mul_to_bases_g1(uint64_t table, const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t *bases_handle);
int32_t dgpu_window_table_mul_to_bases_g2(uint64_t table, const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t *bases_handle);
/* table + multiply + free in one call */
int32_t dgpu_fixed_base_g1(const uint64_t base_xy[12], const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t *out_xy, uint8_t *out_inf);
int32_t dgpu_fixed_base_g2(const uint64_t base_xy[24], const uint64_t *scalars, size_t n, int32_t montgomery, uint64_t *out_xy, uint8_t *out_inf);

/* ---- SnarkPack aggregation folding step ----
 * out_i = addend_i + scalars_i * P_i as affine points: `compress` (legogroth16/src/aggregation/utils.rs:34-49), Key::compress and
 * Key::scale (legogroth16/src/aggregation/key.rs:117-175), `b.mul_bigint(r_i)` (aggregation/groth16/prover.rs:107-112).
 * scalar_stride = 4: one canonical scalar per point; 0: one scalar for all.  addend_xy = NULL: plain scaling.
 * Identity inputs: all-zero words or a set flag byte; identity outputs: zero words and out_inf[i] = 1.
 * The points are elements of the prime-order subgroups (what arkworks' G1Affine / G2Affine hold after deserialization or arithmetic):
 * the scalars are split by the GLV / GLS endomorphisms, which act as multiplications only there.  Any 256-bit scalar (reduced mod r). */
int32_t dgpu_g1_mul_add_batch(const uint64_t *p_xy /* n*12 */, const uint8_t *p_inf, const uint64_t *scalars, size_t scalar_stride,
                              const uint64_t *addend_xy /* n*12 or NULL */, const uint8_t *addend_inf, size_t n, uint64_t *out_xy, uint8_t *out_inf);
int32_t dgpu_g2_mul_add_batch(const uint64_t *p_xy /* n*24 */, const uint8_t *p_inf, const uint64_t *scalars, size_t scalar_stride,
                              const uint64_t *addend_xy /* n*24 or NULL */, const uint8_t *addend_inf, size_t n, uint64_t *out_xy, uint8_t *out_inf);
/* The same step for ONE scalar that is not known yet (a GIPA round's challenge arrives after the round's pairings): prepare runs the doubling chains of
 * the points — the part of a double-and-add that does not depend on the scalar, 1 ms whatever n — and keeps 2^k P_i (G1; G2: 2^k |x|^j P_i, the GLS
 * bases) behind a handle (28.7 KB per G1 point, 115 KB per G2 point); apply then adds the entries at the set bits of the split scalar and the addend in a
 * tree nine additions deep: 0.15 - 0.25 ms instead of 1.5.  out_i = addend_i + scalar * P_i, affine, bit for bit what dgpu_g*_mul_add_batch returns
 * (identity: zero words and out_inf[i] = 1).  One handle serves any number of applies; dgpu_fold_free releases it.  crypto_amd/csrc/fold_kernels.hip.h */
int32_t dgpu_g1_fold_prepare(const uint64_t *p_xy /* n*12 */, size_t n, uint64_t *handle);
int32_t dgpu_g2_fold_prepare(const uint64_t *p_xy /* n*24 */, size_t n, uint64_t *handle);
/* both groups' point sets of a round in ONE launch (two prepare calls can land on one hardware queue and run one after the other); a set may be empty */
int32_t dgpu_fold_prepare_pair(const uint64_t *g1_xy, size_t n1, uint64_t *g1_handle, const uint64_t *g2_xy, size_t n2, uint64_t *g2_handle);
int32_t dgpu_g1_fold_apply(uint64_t handle, const uint64_t scalar[4], const uint64_t *addend_xy /* n*12 or NULL */, uint64_t *out_xy, uint8_t *out_inf);
int32_t dgpu_g2_fold_apply(uint64_t handle, const uint64_t scalar[4], const uint64_t *addend_xy /* n*24 or NULL */, uint64_t *out_xy, uint8_t *out_inf);
int32_t dgpu_fold_free(uint64_t handle);

/* ---- R1CS -> QAP witness map ----
 * replaces LibsnarkReduction::witness_map_from_matrices (legogroth16/src/r1cs_to_qap.rs:150-210): h = ((A z)(B z) - C z) / Z_D as the
 * D = next_pow2(num_constraints + num_inputs) coefficients the prover pairs with h_query (legogroth16/src/prover.rs:281-286).
 * Matrices in CSR (rowptr[num_constraints + 1], cols[nnz], vals[nnz * 4]); assignment = (1, instance..., witness...), num_vars values.
 * montgomery & 1: coefficients and assignment are ark-ff Fr limbs (R = 2^256), else canonical.  The result is canonical
 * (what `into_bigint` yields, prover.rs:281-283): copied to out_h (D * 4 limbs) and/or left in HBM as a scalars handle that
 * dgpu_msm_g1_resident consumes directly.  montgomery & DGPU_WM_H_MONTGOMERY: the copy in out_h is written as Fr limbs instead (the
 * `Vec<F>` witness_map_from_matrices returns, r1cs_to_qap.rs:150-210 — a drop-in for that function hands back exactly this; the conversion
 * runs on the device); a resident vector stays canonical.  *out_len = D. */
#define DGPU_WM_H_MONTGOMERY 2
int32_t dgpu_witness_map(const uint64_t *a_rowptr, const uint32_t *a_cols, const uint64_t *a_vals, size_t a_nnz,
                         const uint64_t *b_rowptr, const uint32_t *b_cols, const uint64_t *b_vals, size_t b_nnz,
                         const uint64_t *c_rowptr, const uint32_t *c_cols, const uint64_t *c_vals, size_t c_nnz,
                         const uint64_t *assignment, size_t num_vars, size_t num_inputs, size_t num_constraints, int32_t montgomery,
                         uint64_t *out_h, uint64_t *out_handle, size_t *out_len);

/* the same with the circuit's matrices resident in HBM (they are fixed per circuit; only the assignment changes per proof) */
int32_t dgpu_r1cs_upload(const uint64_t *a_rowptr, const uint32_t *a_cols, const uint64_t *a_vals, size_t a_nnz,
                         const uint64_t *b_rowptr, const uint32_t *b_cols, const uint64_t *b_vals, size_t b_nnz,
                         const uint64_t *c_rowptr, const uint32_t *c_cols, const uint64_t *c_vals, size_t c_nnz,
                         size_t num_vars, size_t num_inputs, size_t num_constraints, int32_t montgomery, uint64_t *handle);
int32_t dgpu_r1cs_free(uint64_t handle);
/* shape of a resident circuit (any pointer may be NULL); dgpu_legogroth16_prove refuses an (n_inst, num_vars) that disagrees with it */
int32_t dgpu_r1cs_shape(uint64_t r1cs, size_t *num_vars, size_t *num_inputs, size_t *num_constraints);
int32_t dgpu_witness_map_r1cs(uint64_t r1cs, const uint64_t *assignment, size_t num_vars, int32_t montgomery,
                              uint64_t *out_h, uint64_t *out_handle, size_t *out_len);
/* the same with the assignment z already resident (a dgpu_scalars_upload handle of num_vars scalars on the circuit's device): ONE upload of z
 * per proof then serves the witness map and — at scalar offset 1 — the prover's `assignment` = z[1..] of the a / b_g1 / b_g2 / l MSMs */
int32_t dgpu_witness_map_r1cs_resident(uint64_t r1cs, uint64_t assignment, uint64_t *out_h, uint64_t *out_handle, size_t *out_len);

/* ---- the LegoGroth16 prover as one call ----
 * replaces create_proof_and_committed_witnesses_with_assignment (legogroth16/src/prover.rs:267-383, with calculate_coeff :585-594) and — when
 * the circuit is resident (r1cs != 0) — the QAP::witness_map call in front of it (create_proof_with_reduction, :153-180): the whole schedule
 * (one upload of z, witness map, one partition sort shared by the A / B-in-G1 / B-in-G2 / l MSMs when the queries are tables of one shape, the
 * G2 MSM issued first, the O(1) scalar multiplications on a host core meanwhile, the final fold) runs on host threads inside the library.
 *   pk        the proving key: five query handles (dgpu_bases_upload_* / dgpu_window_table_mul_to_bases_*, plain or dgpu_bases_precompute_*d)
 *             on one device, and its O(1) elements on the host in the ABI's affine layout (an all-zero point is the identity)
 *   r1cs      dgpu_r1cs_upload handle of the circuit, or 0 with h_scalars = a resident vector of the D coefficients of h (dgpu_witness_map*)
 *   z         the full assignment (1, instance..., witness...), num_vars scalars; n_inst = number of instance variables incl. the leading 1;
 *             montgomery != 0: &[Fr] limbs.  The first commit_witness_count witnesses are the committed ones (proof.d).
 *   r, s, v   the prover's randomness (canonical limbs, reduced mod r inside)
 *   out       A (G1), B (G2), C (G1), D (G1) as affine points; out_inf[k] = 1 marks an identity (then zero words)
 * Same group elements as the reference computes (asserted limb for limb against the Python mirror and the toxic-waste closed form,
 * tests/test_gpu_prove_abi.py, and from the compiled C++ driver). */
typedef struct dgpu_lego_pk {
    uint64_t a_query, b_g1_query, b_g2_query, h_query, l_query;                                     /* bases handles (query[0] included) */
    const uint64_t *alpha_g1, *beta_g1, *delta_g1, *eta_delta_inv_g1, *eta_gamma_inv_g1;             /* 12 u64 each */
    const uint64_t *beta_g2, *delta_g2;                                                              /* 24 u64 each */
    const uint64_t *a0, *b1_0, *b2_0;                                                                /* query[0] of a / b_g1 (12) / b_g2 (24) */
    const uint64_t *gamma_abc_g1; size_t gamma_abc_len;                                              /* vk.gamma_abc_g1: (n_inst + commit_witness_count) x 12 */
    size_t commit_witness_count;
} dgpu_lego_pk;
int32_t dgpu_legogroth16_prove(const dgpu_lego_pk *pk, uint64_t r1cs, uint64_t h_scalars, const uint64_t *z, size_t num_vars, size_t n_inst,
                               int32_t montgomery, const uint64_t r[4], const uint64_t s[4], const uint64_t v[4],
                               uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12], uint64_t out_d[12], uint8_t out_inf[4]);
/* The same prover for a host that holds the proving key the way the reference does — `pk_common: &ProvingKeyCommon<E>` with its queries as `Vec<G1Affine>` /
 * `Vec<G2Affine>` (legogroth16/src/data_structures.rs, prover.rs:267-383) and h as the `&[E::ScalarField]` QAP::witness_map returned (prover.rs:153-180,577-581):
 * every query is a VIEW of host memory (dgpu_msm_*_strided's description of a slice of ark-ec Affine structs; for a_query / b_g1_query / b_g2_query the WHOLE
 * query, element 0 included) which the resident-bases cache above resolves — at a key's second proof the queries are uploaded once and become tables (the a / b / l
 * queries of DGPU_TABLE_C_WITNESS), later proofs run dgpu_legogroth16_prove's schedule on them and only the assignment and h cross PCIe.  A view the cache does not
 * hold (a key's first proof, the cache off or full) is uploaded for the duration of the call.  instance: n_inst scalars (the leading 1 included), witness: n_wit
 * scalars (montgomery != 0: &[Fr]).  Exactly one source of h: r1cs = a resident circuit (dgpu_r1cs_upload; the witness map then runs inside the call and h
 * never leaves the device — create_proof_with_reduction, prover.rs:153-180, as one call) or h = h_len coefficients in host memory (h_montgomery != 0: &[Fr] — the
 * prover's `into_bigint` then happens on the device; create_proof_with_assignment's contract, prover.rs:237-265).  Same proof, limb for limb, as
 * dgpu_legogroth16_prove on uploaded handles (tests/test_gpu_prove_abi.py). */
typedef struct dgpu_bases_view { const void *p; size_t stride, x_off, y_off, inf_off, n; } dgpu_bases_view;
typedef struct dgpu_lego_pk_host {
    dgpu_bases_view a_query, b_g1_query, b_g2_query, h_query, l_query;
    const uint64_t *alpha_g1, *beta_g1, *delta_g1, *eta_delta_inv_g1, *eta_gamma_inv_g1;             /* 12 u64 each */
    const uint64_t *beta_g2, *delta_g2;                                                              /* 24 u64 each */
    const uint64_t *a0, *b1_0, *b2_0;                                                                /* query[0] of a / b_g1 (12) / b_g2 (24) */
    const uint64_t *gamma_abc_g1; size_t gamma_abc_len;
    size_t commit_witness_count;
} dgpu_lego_pk_host;
int32_t dgpu_legogroth16_prove_host(const dgpu_lego_pk_host *pk, uint64_t r1cs, const uint64_t *h, size_t h_len, int32_t h_montgomery,
                                    const uint64_t *instance, size_t n_inst, const uint64_t *witness, size_t n_wit, int32_t montgomery,
                                    const uint64_t r[4], const uint64_t s[4], const uint64_t v[4],
                                    uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12], uint64_t out_d[12], uint8_t out_inf[4]);
/* The same call accepts a key that is resident across several device contexts (every query a dgpu_bases_upload_*_sharded handle, optionally
 * dgpu_bases_precompute_*d; the five queries over the same contexts): context g multiplies its rows by the matching slices of z and h from a host
 * thread of its own inside the call, the witness map runs once on the circuit's context (r1cs must be given), the partial points are folded on
 * the host.  Same proof as the single-device call (tests/test_gpu_prove_abi.py, two contexts on one GPU). */
/* ---- the LegoGroth16 verifier as one call (SURVEY.md 8a rows a4 - a7 as the reference uses them together) ----
 * replaces verify_proof (legogroth16/src/verifier.rs:62-99): d = gamma_abc[0] + sum x_j gamma_abc[1 + j] + proof.d (calculate_d, :29-50,:101-109),
 * then e(A, B) e(C, -delta) e(d, -gamma) == e(alpha, beta) (verify_qap_proof, :62-84) with the PreparedVerifyingKey's members (:17-25):
 *   alpha_beta_gt  pvk.alpha_g1_beta_g2 (72 words); delta_neg_pc / gamma_neg_pc: the two G2Prepared values (DGPU_G2_PREPARED_WORDS each,
 *   dgpu_g2_prepare's / arkworks' layout); gamma_abc_g1: vk.gamma_abc_g1, gamma_abc_len x 12 words
 *   proof (a, b, c, d) affine, proof_inf[k] != 0 (or all-zero words) marks an identity; public_inputs: n_pub scalars (montgomery != 0: &[Fr])
 * The host's share (the scalar multiplications of calculate_d) runs while the device computes the chain of the pair (A, B).
 * *ok = 1: the proof verifies, 0: it does not.  DGPU_E_BADARG: n_pub + 1 > gamma_abc_len (MalformedVerifyingKey) or n_pub + 2 >
 * DGPU_MAX_LINCOMB (then: dgpu_msm_g1 + dgpu_multi_miller_loop_mixed + dgpu_final_exponentiation); DGPU_E_ZERO where the reference answers
 * UnexpectedIdentity. */
int32_t dgpu_legogroth16_verify(const uint64_t alpha_beta_gt[72], const uint64_t *delta_neg_pc, const uint64_t *gamma_neg_pc, const uint64_t *gamma_abc_g1, size_t gamma_abc_len,
                                const uint64_t proof_a[12], const uint64_t proof_b[24], const uint64_t proof_c[12], const uint64_t proof_d[12], const uint8_t *proof_inf,
                                const uint64_t *public_inputs, size_t n_pub, int32_t montgomery, int32_t *ok);
/* n proofs of ONE verifying key in one call — the classical Groth16 batch check: what the reference reaches through RandomizedPairingChecker
 * (utils/src/randomized_pairing_check.rs:116-138,204-214: three pairs and one GT power per proof, proof_system/src/verifier.rs hands every statement to one lazy
 * checker) with the pairs that share -delta / -gamma merged BEFORE the pairing: n scalings by the powers of `random`, two variable-base MSMs, ONE Miller
 * loop over n + 2 pairs, one final exponentiation, one GT power (crypto_amd/csrc/dock_aggregation.cpp).  proofs_a / _c / _d: n x 12 words, proofs_b: n x 24
 * (all-zero words: identity); public_inputs: n rows of n_pub scalars; random: the batching scalar (drawn AFTER the proofs are fixed; non-zero mod r, else
 * DGPU_E_BADARG).  *ok = 1 iff every proof verifies (up to the 2^-255 soundness error of the random combination).  n = 0: *ok = 1. */
int32_t dgpu_legogroth16_verify_batch(const uint64_t alpha_beta_gt[72], const uint64_t *delta_neg_pc, const uint64_t *gamma_neg_pc, const uint64_t *gamma_abc_g1, size_t gamma_abc_len,
                                      const uint64_t *proofs_a, const uint64_t *proofs_b, const uint64_t *proofs_c, const uint64_t *proofs_d, size_t n,
                                      const uint64_t *public_inputs, size_t n_pub, int32_t montgomery, const uint64_t random[4], int32_t *ok);
/* ---- SnarkPack aggregation of Groth16 / LegoGroth16 proofs (crypto_amd/csrc/dock_aggregation.cpp) ----
 * replaces aggregate_proofs (legogroth16/src/aggregation/groth16/prover.rs:47-147; legogroth16/prover.rs:38-127 when `d` is given) and
 * verify_aggregate_proof (groth16/verifier.rs:36-100, legogroth16/verifier.rs:34-96, legogroth16/using_groth16.rs:45-128) as the reference
 * calls them; the group, pairing and GT work goes through the entry points above from host threads inside the call.
 * The Fiat-Shamir transcript is the caller's (`&mut impl Transcript`, utils/src/transcript.rs:45-63): append_message receives the label and the
 * bytes `Transcript::append` would serialize (`serialize_compressed`: 48 / 96-byte Zcash points, 32 / 48-byte little-endian canonical field
 * elements, a PairCommitment as t then u), challenge_scalar returns the transcript's `challenge_scalar::<Fr>(label)` as 4 canonical words.
 * Points are affine ABI words (identity: all-zero words).  The aggregate proof is a flat array of 64-bit words,
 *   nproofs, n_mipp | com_ab(t, u) com_c [com_d] | z_ab z_c [z_d] | comms_ab[L](l.t l.u r.t r.u) comms_c[L] [comms_d[L]] | z_ab[L](l r) z_c[L] [z_d[L]]
 *   | final_a final_b final_c [final_d] final_vkey(2) final_wkey(2) | vkey_opening(2) wkey_opening(2)          (L = log2 nproofs; GT 72, G1 12, G2 24 words)
 * i.e. the fields of AggregateProof / GipaProof / TippMippProof in their declaration order (groth16/proof.rs:12-24,76-84,117-121). */
#define DGPU_SNARKPACK_MAX_SRS_SIZE (((size_t)2 << 19) + 1)          /* srs.rs MAX_SRS_SIZE */
#define DGPU_SNARKPACK_VALIDATE_POINTS 2                             /* verify flag: every G1 / G2 element of the proof (and of d_list) must be on its curve and in the prime-order subgroup (the other half of Validate::Yes) */
#define DGPU_SNARKPACK_VALIDATE_GT 1                                 /* verify flag: every GT element of the proof must have order r (Validate::Yes on a deserialised proof) */
typedef struct dgpu_transcript {
    void *ctx;
    void (*append_message)(void *ctx, const uint8_t *label, size_t label_len, const uint8_t *bytes, size_t len);
    void (*challenge_scalar)(void *ctx, const uint8_t *label, size_t label_len, uint64_t out[4]);
} dgpu_transcript;
typedef struct dgpu_snarkpack_prover_srs {                           /* ProverSRS specialised to n proofs (srs.rs:60-93,180-237) */
    size_t n;
    const uint64_t *g_alpha_powers_table, *g_beta_powers_table;      /* 2n x 12 */
    const uint64_t *h_alpha_powers_table, *h_beta_powers_table;      /* n x 24 */
    const uint64_t *vkey_a, *vkey_b;                                 /* n x 24 (h^{alpha^i}, h^{beta^i}) */
    const uint64_t *wkey_a, *wkey_b;                                 /* n x 12 (g^{alpha^{n+i}}, g^{beta^{n+i}}) */
} dgpu_snarkpack_prover_srs;
typedef struct dgpu_snarkpack_verifier_srs { size_t n; const uint64_t *g, *h, *g_alpha, *g_beta, *h_alpha, *h_beta; } dgpu_snarkpack_verifier_srs;   /* srs.rs:95-110 */
typedef struct dgpu_groth16_vk { const uint64_t *alpha_g1, *beta_g2, *gamma_g2, *delta_g2, *gamma_abc_g1; size_t gamma_abc_len; } dgpu_groth16_vk;
/* words of an aggregate proof of n proofs (0: n is not a power of two in [2, MAX_SRS_SIZE]) */
size_t dgpu_snarkpack_proof_words(size_t n, int32_t with_d);
/* a, c (, d): n x 12; b: n x 24; d = NULL: Groth16 proofs.  *len_words is set to the proof's length even when cap_words is too small
 * (DGPU_E_LENGTH).  DGPU_E_BADARG: n < 2, not a power of two, or srs->n != n. */
int32_t dgpu_snarkpack_aggregate(const dgpu_snarkpack_prover_srs *srs, const uint64_t *a, const uint64_t *b, const uint64_t *c, const uint64_t *d,
                                 size_t n, const dgpu_transcript *transcript, uint64_t *proof, size_t cap_words, size_t *len_words);
/* variant 0: Groth16 proofs; 1: LegoGroth16 (the proof carries the MIPP for d); 2: LegoGroth16 proofs under the Groth16 aggregator, d_list = the
 * n commitments d (using_groth16.rs).  public_inputs: n_rows x inputs_per_proof canonical scalars (n_rows must equal nproofs); random: the
 * pairing checker's batching scalar (RandomizedPairingChecker::new_using_rng draws it; must be non-zero mod r — a zero would scale every equation after
 * the first out of the check: DGPU_E_BADARG).  flags: DGPU_SNARKPACK_VALIDATE_GT | DGPU_SNARKPACK_VALIDATE_POINTS for a proof from an untrusted source (what
 * CanonicalDeserialize with Validate::Yes checks before the reference ever sees an AggregateProof).  *ok = 1: the aggregate verifies, 0: it does not (a failed
 * pairing / final_z check, a GT / G1 / G2 element outside its subgroup under the validation flags).  DGPU_E_BADARG: a malformed proof (parsing_check), a key that does not match
 * the public inputs (MalformedVerifyingKey), a row count that is not nproofs. */
int32_t dgpu_snarkpack_verify(const dgpu_snarkpack_verifier_srs *srs, const dgpu_groth16_vk *vk, const uint64_t *public_inputs, size_t n_rows, size_t inputs_per_proof,
                              const uint64_t *proof, size_t len_words, int32_t variant, const uint64_t *d_list, const uint64_t random[4],
                              const dgpu_transcript *transcript, int32_t flags, int32_t *ok);
/* elements behind a bases / scalars / sorted handle; constraints of a resident circuit */
int32_t dgpu_handle_len(uint64_t handle, size_t *n);
int32_t dgpu_handle_context(uint64_t handle, int32_t *context);
/* layout of a sharded handle: number of parts (0: not sharded); part k holds elements [lo, hi) as a handle of its own on `context` */
int32_t dgpu_shard_count(uint64_t handle, int32_t *count);
int32_t dgpu_shard_part(uint64_t handle, size_t k, uint64_t *sub_handle, size_t *lo, size_t *hi, int32_t *context);

/* ---- canonical (de)serialisation of group elements (host code) ----
 * The format ark-bls12-381 0.4 emits for `CanonicalSerialize` (Zcash / IETF BLS12-381): big-endian coordinates, top three bits of
 * byte 0 = compressed / infinity / y-lexicographically-largest; G1 48 B (compressed) or 96 B, G2 96 or 192 B with c1 before c0.
 * Used to load keys and proofs written by the Rust side (legogroth16/src/data_structures.rs:7-186) into the ABI layout.
 * `mode` of the deserialisers: bit 0 = compressed; DGPU_SERDE_NO_VALIDATE = arkworks' Validate::No (skip the subgroup test).
 * DGPU_E_BADARG: wrong compression flag, coordinate >= p, not on the curve, a non-canonical infinity encoding, or — Validate::Yes,
 * the default, what `deserialize_compressed` does — a point outside the prime-order subgroup ([r]P != O). */
#define DGPU_SERDE_NO_VALIDATE 2
int32_t dgpu_g1_serialize(const uint64_t *xy /* n*12 */, const uint8_t *is_inf, size_t n, int32_t compressed, uint8_t *out);
int32_t dgpu_g1_deserialize(const uint8_t *in, size_t n, int32_t mode, uint64_t *xy /* n*12 */, uint8_t *is_inf /* n */);
int32_t dgpu_g2_serialize(const uint64_t *xy /* n*24 */, const uint8_t *is_inf, size_t n, int32_t compressed, uint8_t *out);
int32_t dgpu_g2_deserialize(const uint8_t *in, size_t n, int32_t mode, uint64_t *xy /* n*24 */, uint8_t *is_inf /* n */);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
