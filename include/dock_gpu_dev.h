/* include/dock_gpu_dev.h — DEVELOPMENT surface of the MI355X backend: tuning knobs of the kernels, stage timers, self-test and fault-injection
 * hooks.  Served by crypto_amd/libdock_gpu_dev.so only — the twin of the product library built from the same objects plus
 * crypto_amd/csrc/dock_dev.hip (and dock_core.hip compiled -DDGPU_DEV).  The product library libdock_gpu.so (include/dock_gpu.h: what a Rust host
 * binds) exports none of these; tests/, tools/ and the stage / roofline leg of bench.py load the twin.
 *
 * Every knob returns the SAME result limb for limb at any setting (the parity tests sweep each of them against the automatic choice); the defaults
 * are the measured optima on MI355X.  Process-wide. */
#ifndef DOCK_GPU_DEV_H
#define DOCK_GPU_DEV_H
#include "dock_gpu.h"
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* window width c (bits) used by the bucket method; 0 = automatic from n.  Any value gives the same point. */
int32_t dgpu_set_window_bits(int32_t c);
/* terms per lane of the bucket accumulation (16..4096; 0 = automatic).  Any value gives the same point (tests sweep it). */
int32_t dgpu_set_chunk(int32_t terms);
/* log2 of the buckets one lane of the bucket reduction sums serially on the table pipeline (0..6; -1 = automatic: 3 for a 2^19-bucket table when the
 * call runs alone, 4 when three or more calls are in flight on the device context).  Any value gives the same point. */
int32_t dgpu_set_reduce_shift(int32_t log2_buckets_per_lane);
/* form of the bucket reduction.  0 (default): the shared bucket set of the table pipeline by bit marginals (crypto_amd/csrc/reduce_kernels.hip.h:
 * one butterfly network per wave, class folds with four members per value; the plain pipeline's per-window reduction then runs as 4); 2: the same with
 * one lane per value in the class folds; 4 / 1: the scan form of rounds 1-4 (k_reduce_l0, then k_reduce_top_quad / k_reduce_top). */
int32_t dgpu_set_reduce_lanes(int32_t lanes);
/* Forms of the Miller-loop kernels, a bit mask (default 31).  Bit 0: dgpu_multi_miller_loop of up to 8192 pairs runs its line kernel in pieces
 * and overlaps the products / host share of a finished piece with the next (and dgpu_g2_prepare runs the same lanes-per-point chain
 * followed by a parallel conversion pass).  Bit 1: the product tree gives every node 18 lane pairs (one Fp2 product deep per level) instead
 * of three.  Bit 2: up to 4096 pairs the line kernel gives every (P, Q) sixteen lanes (a doubling step two Fp2 operations deep instead of
 * five).  Bit 3 (with bit 2): those sixteen lanes are four lanes in each of four waves, a wave per role (k_miller_lines_ws).  Bit 4: the sparse
 * products of a launch of up to 512 blocks run as three waves per 32 slices (k_line_products3).  Every combination gives the same Fp12 value
 * limb for limb (tests compare all 32).  Upper bits, zero = the default: bits 8-13 / 16-21 the bits of |x| at which the chain is cut into three
 * launches (40 and 17), bits 24-27 the slice length of the last piece's sparse products, bits 28-29 log2 of a factor on bit 4's block limit
 * (tools/dev/ml_cuts_sweep.py, ml_tail_sweep.py, ml_lp3_limit.py).  DGPU_E_BADARG for anything else. */
int32_t dgpu_set_miller_pipeline(int32_t mode);

/* ---- instrumentation (bench.py's stage breakdown and roofline leg; rocprofv3 cross-check) ----
 * When enabled, every stage of the next calls is bracketed by HIP events on the library's own stream. */
int32_t dgpu_prof_enable(int32_t on);
int32_t dgpu_prof_reset(void);
/* fills up to `cap` entries; returns the number of stages recorded.  names[i] points to a static string.  The last row, "hipMalloc", is
 * always present: calls = device allocations since dgpu_prof_reset (0 in steady state), total_ms = the time they took. */
int32_t dgpu_prof_read(const char **names, double *total_ms, uint64_t *calls, int32_t cap);

/* ---- self-test hooks (run the device field / group code on tiny inputs) ---- */
/* host: the GLV split of a G1 scalar used by dgpu_g1_scale_batch: k mod r = k1 + k2 * lambda, lambda = x_BLS^2 - 1, k1, k2 < 2^128 */
int32_t dgpu_selftest_glv_decompose(const uint64_t k[4], uint64_t k1[2], uint64_t k2[2]);
int32_t dgpu_selftest_fp_mul(const uint64_t *a /* n*6 */, const uint64_t *b /* n*6 */, size_t n, uint64_t *out /* n*6 */);
int32_t dgpu_selftest_g1_sum(const uint64_t *pts_xy /* n*12 */, const uint8_t *neg, size_t n, uint64_t out_xyz[18]);

/* ---- fault injection (tests/test_gpu_fault_paths.py): the k-th hipMalloc from now and the count - 1 after it fail ---- */
int32_t dgpu_dev_fail_alloc_after(int64_t k, int64_t count);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
