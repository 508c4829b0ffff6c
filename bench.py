"""bench.py — BLS12-381 G1 variable-base MSM throughput on MI355X (BASELINE.json metric).

One "step" = one n-term MSM over operands that are already resident in HBM, through the C ABI
(dgpu_msm_g1_resident): the bases are a proving-key-style resident handle converted once, outside the timed
region, to the library's precomputed-multiples table (dgpu_bases_precompute_g1: per-key setup, like the upload),
the scalars a resident canonical vector.  At N = 1 the workload is BASELINE.json configs[1]: n = 2^20 random
scalars / points.  At N > 1 (one process per GPU, launched by torch.distributed.run) the ranks jointly compute ONE
MSM of 2^24 terms per step (configs[4]) by point-chunk sharding: each rank runs the full pipeline on its own
2^24 / N terms, then an RCCL all_gather of the 144-byte partial points and a local fold (crypto_amd/sharded.py).
value = (terms per step / 2^20) / seconds per step, i.e. n = 2^20-MSM equivalents per second for the whole job.

Extra objects on the JSON line: "roofline" (dominant kernel = k_accumulate, algorithmic bytes = 128 B/term,
duration from HIP events on the library's stream), "valu_roofline" (the same kernel against the measured
v_mad_u64_u32 issue peak: the path is integer-multiply bound), "cpu_baseline" (the CPU oracle = arkworks-style
Pippenger, one thread per window like rayon, timed on this box's host cores; kind "port"; plus one thread and the
n = 2^16 plumbing config), "secondary" (N = 1 only, outside the timed region: the plain resident pipeline without a
table, H2D-inclusive and one-shot calls, the skewed scalar distributions of SURVEY 8d, one 2^24-term MSM on this one
GPU, G2 MSM, 1024-pair Miller loop, final exponentiation, verification of 1024 proofs, witness map, LegoGroth16 prove at 2^20 constraints).

Inputs and the closed-form check are produced by the library itself (fixed-base kernel, published generator
encodings); oracle/ is imported only inside the cpu_baseline leg, where it is the timed CPU baseline and the
cross-check of the GPU result.
"""
import argparse
import json
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the process's first HIP call (PyTorch's, here): what dgpu_runtime_hints(DGPU_HINT_EIGHT_HW_QUEUES) does for a host whose first HIP call is the library's (include/dock_gpu.h)
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
# arkworks / Zcash compressed encodings of the BLS12-381 generators (published constants; tests/test_serde_host.py pins the codec to them)
G1_GEN_COMPRESSED = "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
G2_GEN_COMPRESSED = ("93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
                     "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")
MADS_PER_MIXED_ADD = 6 * 338 + 507 + 2 * 260     # XYZZ mixed addition over the 13 x 30-bit signed field (fp30s.hip.h): 6 products, one fused two-product reduction, 2 squares (round 2, 14 x 29-bit limbs: 3542)
MADS_PER_G2_MIXED_ADD = 2 * (8 * 507 + 2 * 338)   # the same formula over Fp2 on a lane pair (fs2_pair.hip.h): per lane 8 fused two-product reductions + 2 products for the squares (round 2: 10976)
MAD_PEAK = 35.1                                   # Tmad/s: best measured stream of nothing but v_mad_i64_i32 (tools/ubench/clock_under_load.hip, profiles/r05_clock_under_load.txt: 35.1 at a shader clock of 2.44 GHz;
                                                  # rounds 1-4 used 31.8 from profiles/r01h_instr_rate_ubench.txt, a shorter run at a lower clock: the fractions of those rounds read 10 % higher for the same kernel)


def seeded_scalars(seed, n):
    """n scalars uniform in [0, r): 255-bit draws from a seeded PCG64 stream, draws >= r redrawn (SURVEY 8d config 1 input rule)"""
    rng = np.random.Generator(np.random.PCG64(seed))

    def draw(k):
        a = rng.integers(0, 1 << 64, size=(k, 4), dtype=np.uint64, endpoint=False)
        a[:, 3] &= np.uint64((1 << 63) - 1)
        return a
    out = draw(n)
    r = [np.uint64((R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF) for i in range(4)]
    while True:
        ge = np.zeros(n, dtype=bool); decided = np.zeros(n, dtype=bool)
        for i in (3, 2, 1, 0):
            gt = (out[:, i] > r[i]) & ~decided; lt = (out[:, i] < r[i]) & ~decided
            ge |= gt; decided |= gt | lt
        ge |= ~decided                       # equal to r
        k = int(ge.sum())
        if k == 0:
            return out
        out[ge] = draw(k)


def dot_mod_r(a, b):
    """sum a_i b_i mod r for two (n, 4) uint64 limb arrays, exactly: 16-bit pieces, float64 matrix products over chunks of 2^20 rows
    (every partial sum < 2^32 * 2^20 = 2^52 is exact in a double) — the closed form the timed result is checked against"""
    a16 = np.ascontiguousarray(a).view(np.uint16).reshape(len(a), 16)
    b16 = np.ascontiguousarray(b).view(np.uint16).reshape(len(b), 16)
    tot = 0
    for lo in range(0, len(a), 1 << 20):
        m = a16[lo:lo + (1 << 20)].astype(np.float64).T @ b16[lo:lo + (1 << 20)].astype(np.float64)
        for i in range(16):
            for j in range(16):
                tot += int(m[i, j]) << (16 * (i + j))
    return tot % R_MOD


TRACE_CALLS = bool(os.environ.get("DGPU_BENCH_TRACE_CALLS"))      # development: (index, start, end) of every call of an InFlight run, to stderr from the timed region


class InFlight:
    """`count` calls of fn() by `threads` host threads, each taking the next call from a shared counter.  The threads are created and parked at a start
    line BEFORE the clock starts, and the caller's thread blocks while they run: no thread of this process wakes another through the interpreter inside a
    timed region.  (A ThreadPoolExecutor does: every submit() wakes a parked worker that then waits out CPython's 5-ms switch interval for the GIL the
    submitting thread still holds — six workers cost the first 30 - 35 ms of a 50-ms timed region, or nothing, depending on who wins a race: the
    default bench line read 180 - 240 or 420 - 445 MSM/s for the same library and the same kernels, tools/dev/r05_host_timeline.py.)"""

    def __init__(self, fn, count, threads):
        import itertools
        import threading
        self.fn, self.count, self.err = fn, count, None
        self.results = [None] * count
        self.log = []
        self._next = itertools.count()                      # (next() on it is atomic under the GIL)
        self._start, self._done = threading.Barrier(threads + 1), threading.Barrier(threads + 1)
        self._threads = [threading.Thread(target=self._run, daemon=True) for _ in range(threads)]
        for t in self._threads:
            t.start()

    def _run(self):
        self._start.wait()
        try:
            while True:
                i = next(self._next)
                if i >= self.count:
                    break
                if TRACE_CALLS:
                    t0 = time.perf_counter(); self.results[i] = self.fn(); self.log.append((i, t0, time.perf_counter()))
                else:
                    self.results[i] = self.fn()
        except BaseException as e:          # noqa: BLE001
            self.err = e
        self._done.wait()

    def go(self):
        """release the threads, wait for the last call, return the results in call order"""
        self._start.wait()
        self._done.wait()
        for t in self._threads:
            t.join()
        if self.err is not None:
            raise self.err
        return self.results


def run_inflight(fn, count, threads):
    """(seconds, results) of `count` calls with `threads` in flight; thread start-up is outside the clock"""
    w = InFlight(fn, count, threads)
    t0 = time.perf_counter()
    r = w.go()
    return time.perf_counter() - t0, r


def single_gpu_2p24(ca, FB, gen1, pool):
    """BASELINE config 5's 2^24 terms on ONE GPU (table built once per key, resident scalars, four calls in flight): the 1-GPU rate the N > 1 lines
    are a multiple of.  Returns a dictionary; msm_2p20_equivalents_per_s is in the unit of the line's `value`."""
    n24 = 1 << 24
    k24 = seeded_scalars(0x5EED2400, n24); s24 = seeded_scalars(0x5EED2401, n24)
    with FB.WindowTable(ca.G1, gen1[0]) as t1:
        b24 = t1.multiply_many_to_bases(k24)
        exp_xy, _ = t1.multiply(dot_mod_r(k24, s24))
    t0 = time.perf_counter(); b24.precompute(); tab24 = (time.perf_counter() - t0) * 1e3
    d24 = ca.DeviceScalars(s24)
    ok = bool((b24.msm_resident(d24)[:12] == exp_xy).all())
    lat = timed(lambda: b24.msm_resident(d24), 3, warm=8)        # (warm-up calls: every slot's workspace grows on its first call of this size)
    run_inflight(lambda: b24.msm_resident(d24), 4, 4)
    thr = run_inflight(lambda: b24.msm_resident(d24), 8, 4)[0] / 8 * 1e3
    b24.free(); d24.free()
    return {"latency_ms": round(lat, 2), "ms_per_msm_4_in_flight": round(thr, 2), "msm_2p20_equivalents_per_s": round(16e3 / thr, 2),
            "table_build_ms": round(tab24, 1), "bit_exact_vs_closed_form": ok}


def timed(fn, k=5, warm=1):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    return (time.perf_counter() - t0) / k * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=0, help="terms per GPU = 2^log2n; default: 20 at 1 GPU (BASELINE config 2), 24 - log2(gpus) at N > 1 (config 5: 2^24 in total)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary timings appended to the JSON line at N = 1")
    ap.add_argument("--inflight", type=int, default=6, help="MSM calls in flight per GPU (host threads; each call owns a stream + workspace slot)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-legs", action="store_true", help="skip secondary.cpu (the CPU path timed beside configs 3 and 4: about 10 s of host work)")
    ap.add_argument("--no-table", action="store_true", help="time the plain resident pipeline (no precomputed-multiples table)")
    ap.add_argument("--reduce-shift", type=int, default=-1, help="development: dgpu_set_reduce_shift (log2 buckets per lane of the bucket reduction; -1 = automatic)")
    ap.add_argument("--reduce-lanes", type=int, default=-1, help="development: dgpu_set_reduce_lanes (0 = bit marginals, the default; 1 / 4 = the scan form of rounds 1-4)")
    ap.add_argument("--miller-pipeline", type=int, default=-1, help="development: dgpu_set_miller_pipeline (forms of the Miller kernels; -1 = the library's default, 31)")
    args = ap.parse_args()

    sys.setswitchinterval(1e-4)            # (a host thread that returns from the library gets the interpreter within 0.1 ms instead of CPython's default 5)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DGPU_BENCH_SAME_DEVICE"):      # plumbing test only: several ranks on one GPU
        local = 0
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    # DGPU_BENCH_STUB (tests/test_bench_multi_rank_cpu.py only): the library is replaced by tests/bench_stub.py (the group Z_r) and the ranks are
    # CPU processes under gloo, so that this file's N > 1 control flow is exercised where no 8-GPU node is at hand.  Never a measurement.
    STUB = bool(os.environ.get("DGPU_BENCH_STUB"))
    if STUB:
        os.environ["DGPU_BENCH_BACKEND"] = "gloo"
        dev = cdev = None
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        cdev = dev if os.environ.get("DGPU_BENCH_BACKEND", "nccl") == "nccl" else None     # where collective payloads live
    sync = (lambda: None) if STUB else torch.cuda.synchronize
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DGPU_BENCH_BACKEND", "nccl")       # "gloo" only for the one-GPU plumbing test (RCCL refuses two ranks on one device)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if STUB:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from bench_stub import ca, sharded, serde, FB
    else:
        import crypto_amd as ca
        from crypto_amd import sharded, serde, fixed_base as FB

    ca.init(local)
    # development knobs exist in the twin only (include/dock_gpu_dev.h): a run that sets one is a run of the TWIN from here on, and says so on its line
    import contextlib
    dev_run = (args.reduce_shift >= 0 or args.reduce_lanes >= 0 or args.miller_pipeline >= 0) and not STUB
    whole = contextlib.ExitStack()
    if dev_run:
        whole.enter_context(ca.twin())
        from crypto_amd._native import lib as _lib
        if args.reduce_shift >= 0:
            assert _lib().dgpu_set_reduce_shift(args.reduce_shift) == 0
        if args.reduce_lanes >= 0:
            assert _lib().dgpu_set_reduce_lanes(args.reduce_lanes) == 0
        if args.miller_pipeline >= 0:
            assert _lib().dgpu_set_miller_pipeline(args.miller_pipeline) == 0
    if args.log2n == 0:
        lg = 0
        while (1 << lg) < world:
            lg += 1
        args.log2n = 20 if world == 1 else max(16, 24 - lg)
    n = 1 << args.log2n
    ncpu = os.cpu_count() or 1
    # Synthetic inputs with known discrete logs: P_i = k_i G with seeded k_i, scalars uniform in [0, r).  Everything here comes from the
    # library itself (the bases from its fixed-base kernel, straight into a resident handle); the CPU oracle is only touched by the
    # cpu_baseline leg at the end.
    gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(G1_GEN_COMPRESSED))
    ks = seeded_scalars(0x5EED2000 + rank, n)
    scalars = seeded_scalars(0x5EED1000 + rank, n)
    t_setup = time.perf_counter()
    with FB.WindowTable(ca.G1, gen1[0]) as gtab:
        db = gtab.multiply_many_to_bases(ks)
    use_table = not args.no_table
    table_ms = 0.0
    if use_table:
        t0 = time.perf_counter()
        db.precompute()
        table_ms = (time.perf_counter() - t0) * 1e3
    ds = ca.DeviceScalars(scalars)
    t_setup = time.perf_counter() - t_setup

    def step():
        part = db.msm_resident(ds)
        return sharded.gather_and_fold(ca.G1, part, cdev) if world > 1 else part

    # correctness of what is timed: closed form (sum s_i k_i) G over ALL ranks' terms; the expected point comes from the fixed-base path
    # (a different kernel family), the comparison against the CPU oracle is part of the cpu_baseline leg
    res = step()
    loc = dot_mod_r(ks, scalars)
    if world > 1:
        allv = [None] * world
        dist.all_gather_object(allv, loc)
        tot = sum(allv) % R_MOD
    else:
        tot = loc
    with FB.WindowTable(ca.G1, gen1[0]) as gtab:
        exp_xy, exp_inf = gtab.multiply(tot)
    got_inf = not res[12:].any()
    bit_exact = bool(got_inf == exp_inf and (got_inf or (res[:12] == exp_xy).all()))
    assert bit_exact, "GPU MSM does not match the closed form"

    from concurrent.futures import ThreadPoolExecutor
    inflight = max(1, args.inflight)
    pool = ThreadPoolExecutor(max_workers=inflight)

    def run_steps(k):
        """k steps with up to `inflight` local MSMs in flight (untimed use: warm-up); partial points are gathered/folded in step order"""
        last = None
        for part in InFlight(lambda: db.msm_resident(ds), k, inflight).go():
            last = sharded.gather_and_fold(ca.G1, part, cdev) if world > 1 else part
        return last

    for _ in range(args.warmup):
        step()
    run_steps(2 * inflight)                           # untimed: every host thread / slot has run once (the workspaces were sized at upload / precompute)
    # sequential pass (one call in flight): the single-call latency
    tl = time.perf_counter()
    for _ in range(3):
        step()
    latency_ms = (time.perf_counter() - tl) / 3 * 1e3
    run_steps(2 * inflight)                           # untimed, again: with more caller threads than the library has slots (--inflight 8) the SECOND multi-threaded pass of a process
                                                      # reads 30 - 60 ms long whatever it computes (a harness effect: native threads and every later pass do not show it, tools/dev/native/inflight_threads.cpp)
    timed_calls = InFlight(lambda: db.msm_resident(ds), args.steps, inflight)       # the host threads wait at their start line: nothing is created inside the clock
    if world > 1:
        dist.barrier()
    sync()
    allocs0 = ca.device_alloc_count()
    t0 = time.perf_counter()
    last = None
    for part in timed_calls.go():                                                    # EXACTLY args.steps steps; the partial points are gathered / folded in step order
        last = sharded.gather_and_fold(ca.G1, part, cdev) if world > 1 else part
    sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if TRACE_CALLS:
        for i, a_, b_ in sorted(timed_calls.log):
            print("call %3d: %8.2f -> %8.2f ms" % (i, (a_ - t0) * 1e3, (b_ - t0) * 1e3), file=sys.stderr)
    allocs_timed = ca.device_alloc_count() - allocs0
    assert (last == res).all(), "result changed between runs"
    per_rank_table_ms = [round(table_ms, 1)]
    collective = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev if cdev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        got = [None] * world
        dist.all_gather_object(got, round(table_ms, 1))            # every rank builds the table of ITS 2^24 / N terms: the per-key setup of each
        per_rank_table_ms = got
        ones = torch.ones(1, dtype=torch.int64, device=cdev if cdev is not None else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)                 # the number of ranks the collective library itself saw (RCCL on a GPU box): must equal N
        collective = {"backend": dist.get_backend(), "ranks_seen_by_all_reduce": int(ones.item()), "world_size": world}
        assert collective["ranks_seen_by_all_reduce"] == world

    # Stage breakdown and the dominant kernel's duration (HIP events on the library's own stream around every stage): the stage timers are part of
    # the DEVELOPMENT surface (include/dock_gpu_dev.h), which the product library does not export.  So this leg — untimed, rank 0, after the
    # ranks' collectives — runs the same workload on the development twin (libdock_gpu_dev.so: the product's objects, kernels included, plus that surface) with its own copy of the
    # operands: one call in flight (stages_seq: the roofline's kernel duration) and the timed region's shape (stages: durations that overlap).
    stages_seq, stages = {}, {}
    if rank == 0:
        local_res = res if world == 1 else db.msm_resident(ds)        # this rank's own partial point (product library), before the twin takes over the wrappers
        with ca.twin():
            with FB.WindowTable(ca.G1, gen1[0]) as gtab_t:
                db_t = gtab_t.multiply_many_to_bases(ks)
            if use_table:
                db_t.precompute()
            ds_t = ca.DeviceScalars(scalars)
            assert (db_t.msm_resident(ds_t)[:18] == local_res[:18]).all(), "twin != product"
            run_inflight(lambda: db_t.msm_resident(ds_t), 2 * inflight, inflight)
            ca.prof.enable(True); ca.prof.reset()
            for _ in range(3):
                db_t.msm_resident(ds_t)
            stages_seq = ca.prof.read(); ca.prof.reset()
            run_inflight(lambda: db_t.msm_resident(ds_t), args.steps, inflight)
            stages = ca.prof.read(); ca.prof.enable(False)
            db_t.free(); ds_t.free()

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        terms = n * world
        value = (terms / float(1 << 20)) / (dt / args.steps)
        # dominant kernel: duration from the one-call-in-flight pass (HIP events on the library's stream); inside the
        # timed region up to `inflight` launches share the chip, so their event durations overlap and are reported separately
        acc_ms = stages_seq.get("msm.accumulate", (0.0, 1))
        acc_avg_ms = acc_ms[0] / max(1, acc_ms[1])
        acc_ov = stages.get("msm.accumulate", (0.0, 1))
        acc_ov_ms = acc_ov[0] / max(1, acc_ov[1])
        alg_bytes = 128.0 * n                      # 32 B scalar + 96 B affine base per term, one launch = n terms
        achieved = alg_bytes / (acc_avg_ms * 1e-3) / 1e9 if acc_avg_ms > 0 else 0.0
        # HBM bytes per launch from the PMC counters: collected by separate rocprofv3 --pmc passes of THIS command (tools/dev/round4_profiles.sh ->
        # tools/traffic_json.py), not inside this process; the file names the commit it was measured at
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, "profiles", "traffic_accumulate.json")
        if os.path.exists(tj):
            try:
                tr = json.load(open(tj))
                if tr.get("log2n") == args.log2n and bool(tr.get("table")) == use_table:
                    traffic = tr.get("hbm_bytes_per_launch")
                    traffic_src = "profiles/traffic_accumulate.json, measured at commit %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --inflight 1`)" % tr.get("commit", "of round 3")
            except Exception:
                pass
        shape = db.table_shape()                   # (rows, window bits, windows) when the handle is a table
        windows = shape[2] if shape else (16 if args.log2n >= 17 else None)
        out = {
            "metric": "BLS12-381 G1 MSM/s at n=2^20 (1 GPU) and n=2^24 (8 GPU); bit-exact vs CPU",
            "value": round(value, 3), "unit": "MSM/s (n=2^20-term equivalents, whole job)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None,
            "dtype": "i32 (13 signed 30-bit limbs, 381-bit modular integer; 64-bit column accumulators)",
            "data": "synthetic (seeded PCG64 scalars uniform in [0, r); bases k_i G with seeded known discrete logs from the fixed-base kernel)",
            "config": {"workload": "BLS12-381 G1 variable-base MSM, n=2^%d terms per GPU, operands resident in HBM (%s), %s" % (
                args.log2n, "bases as a precomputed-multiples table built once per key outside the timed region" if use_table else "plain prepared bases",
                "1xMI355X" if world == 1 else "one 2^%d-term MSM per step point-chunk sharded over %dxMI355X, RCCL all_gather of partial points" % (
                    args.log2n + (world - 1).bit_length(), world)),
                "terms_per_step": terms, "bit_exact_vs_closed_form": bit_exact,
                "parallelism": "1 process per GPU, %d ranks, %d calls in flight per GPU" % (world, inflight), "collective": collective,
                "per_key_setup_ms": {"fixed_base_bases_and_uploads": round(t_setup * 1e3 - table_ms, 1), "precomputed_table": round(table_ms, 1), "precomputed_table_per_rank": per_rank_table_ms},
                "scaling_note": "N = 1 is BASELINE config 2 (2^20 terms); every N > 1 computes config 5's 2^24 terms in total, so the N > 1 "
                                "values are a strong-scaling series; secondary.g1_2p24_single_gpu is the 1-GPU time of the same 2^24 terms"},
            "terms_per_s": round(terms / (dt / args.steps), 1),
            "inflight": inflight, "latency_ms_one_in_flight": round(latency_ms, 4), "device_allocations_in_timed_region": int(allocs_timed),
            "stages_ms": {k.replace("msm.", ""): round(v[0] / max(1, v[1]), 4) for k, v in stages.items()},
            "stages_ms_one_in_flight": {k.replace("msm.", ""): round(v[0] / max(1, v[1]), 4) for k, v in stages_seq.items()},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 6),
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "msm::k_accumulate<msm::G1S, false>" if use_table else "msm::k_accumulate<msm::G1S, false> (plain pipeline)", "avg_ms": round(acc_avg_ms, 4), "avg_ms_overlapped": round(acc_ov_ms, 4),
                         "note": "algorithmic 128 B/term x 2^log2n terms per launch; avg_ms = HIP-event duration on the library's stream with one call in flight "
                                 "(same process, untimed pass on the development twin — the product's kernel objects plus the stage timers, which the product "
                                 "library does not export; rocprof of `bench.py --inflight 1` agrees), avg_ms_overlapped = the same with the timed region's "
                                 "calls in flight, where launches share the chip; the kernel is integer-multiply bound: see valu_roofline"},
        }
        if acc_avg_ms > 0 and windows:
            mads = float(n) * windows * MADS_PER_MIXED_ADD
            out["valu_roofline"] = {"bound": "v_mad_i64_i32 (same issue rate as v_mad_u64_u32)", "achieved": round(mads / (acc_avg_ms * 1e-3) / 1e12, 3), "peak": MAD_PEAK, "unit": "Tmad/s",
                                    "frac": round(mads / (acc_avg_ms * 1e-3) / 1e12 / MAD_PEAK, 4), "mixed_additions_per_launch": int(n) * windows,
                                    "note": "%d windows x n mixed additions x %d 32 x 32 -> 64-bit multiply-adds each (13 x 30-bit signed limbs since round 3: 13.8 %% fewer than the 3542 of the 14 x 29-bit field, so the same kernel time is a LOWER fraction); peak = measured issue rate, profiles/r01h_instr_rate_ubench.txt" % (
                                        windows, MADS_PER_MIXED_ADD)}
        if dev_run:
            out["library"] = "libdock_gpu_dev.so (development twin: --reduce-shift / --reduce-lanes set a knob the product does not have)"
        if STUB:
            out["data"] = "STUB (tests/bench_stub.py: control-flow test on CPU ranks, not a measurement)"
        if not args.no_cpu_baseline and not STUB and world == 1:        # (the contract: on rank 0 at N = 1 only)
            out["cpu_baseline"] = cpu_baseline(ca, gen1, ks, scalars, db, ds, args.log2n, ncpu)
        if world == 1 and not args.no_secondary and not STUB:
            try:
                out["secondary"] = secondary_configs(args.log2n, ks, scalars, db, ds, pool, cpu_legs=not (args.no_cpu_baseline or args.no_cpu_legs), ncpu=ncpu)
            except Exception as e:                      # never let the secondary numbers take the headline line down
                out["secondary"] = {"error": repr(e)}
        # scaling_base: the 1-GPU rate on config 5's 2^24 terms, in the unit of `value` — what value(N > 1) must be divided by (the N = 1 line is config 2:
        # 2^20 terms with six independent calls in flight, a different workload).  At N = 1 it comes out of the secondary leg; at N > 1 rank 0 measures it
        # on its own GPU after the timed region (one 28-GB table, ~3 s), so that every line of the driver's SCALE run carries its own denominator.
        sb = None
        try:
            if world == 1 and isinstance(out.get("secondary"), dict):
                sb = out["secondary"].get("g1_2p24_single_gpu", {}).get("msm_2p20_equivalents_per_s")
            elif world > 1 and not STUB and not os.environ.get("DGPU_BENCH_NO_SCALING_BASE"):
                sb = single_gpu_2p24(ca, FB, gen1, pool).get("msm_2p20_equivalents_per_s")
        except Exception as e:                          # noqa: BLE001
            out["scaling_base_error"] = repr(e)
        out["scaling_base"] = sb
        out["scaling_base_note"] = "MSM/s (2^20-term equivalents) of ONE GPU computing config 5's 2^24 terms (table resident, four calls in flight): speed-up at N GPUs = value / scaling_base"
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(ca, gen1, ks, scalars, db, ds, log2n, ncpu):
    """arkworks-style Pippenger on the host cores (the oracle: test infrastructure, used here only as the timed CPU baseline and its cross-check):
    one task per window like rayon (17 windows at n = 2^20 => at most 17 busy threads), the same on ONE thread, and BASELINE config 1 (n = 2^16)."""
    from crypto_amd import fixed_base as FB
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c as O
    log2s = min(log2n, 20)
    ns = 1 << log2s
    with FB.WindowTable(ca.G1, gen1[0]) as gtab:
        bases, _ = gtab.multiply_many(ks[:ns])               # host copies of the first ns bases
    c = O.window_c(ns)
    nw = (255 + c - 1) // c
    thr = max(1, min(ncpu, nw))
    tb = time.perf_counter()
    ref = O.G1.msm(bases, scalars[:ns], threads=thr)
    tcpu = time.perf_counter() - tb
    chk = db.msm_resident(ds, n=ns)
    same = bool((O.G1.to_affine(ref)[0] == O.G1.to_affine(chk)[0]).all())
    res = {"value": round((ns / float(1 << 20)) / tcpu, 4), "unit": "MSM/s (n=2^20-term equivalents)", "cores": thr, "kind": "port",
           "sample": "one n=2^%d G1 MSM, %.2f s wall on %d threads (one per window, arkworks' rayon structure) of %d logical CPUs; result bit-exact vs GPU: %s" % (
               log2s, tcpu, thr, ncpu, same)}
    import ctypes as _C
    L = O.lib()
    L.orc_bench_fp_mul_ns.restype = _C.c_double; L.orc_bench_g1_madd_ns.restype = _C.c_double
    res["calibration"] = {"ns_per_fp_mul": round(min(L.orc_bench_fp_mul_ns(_C.c_size_t(2000000)) for _ in range(3)), 1),
                          "ns_per_g1_mixed_add": round(min(L.orc_bench_g1_madd_ns(_C.c_size_t(300000)) for _ in range(3)), 1),
                          "note": "one core, dependent chain; plain C (unsigned __int128, no-carry CIOS, -O3 -mbmi2 -madx).  ark-ff 0.4's asm Montgomery backend is "
                                  "quoted at ~25-30 ns per 381-bit product on current x86: scale `value` by ns_per_fp_mul / 27 for an estimate of the real arkworks path"}
    l1 = min(log2s, 18)                                      # one thread: a 2^18 sample (about 4 s), scaled per term
    tb = time.perf_counter()
    O.G1.msm(bases[:1 << l1], scalars[:1 << l1], threads=1)
    t1 = time.perf_counter() - tb
    res["one_thread"] = {"value": round(((1 << l1) / float(1 << 20)) / t1, 4), "unit": "MSM/s (n=2^20-term equivalents)", "cores": 1,
                         "sample": "one n=2^%d G1 MSM on one thread, %.2f s" % (l1, t1)}
    l16 = min(log2s, 16)
    c16 = O.window_c(1 << l16)
    thr16 = max(1, min(ncpu, (255 + c16 - 1) // c16))
    tb = time.perf_counter()
    r16 = O.G1.msm(bases[:1 << l16], scalars[:1 << l16], threads=thr16)
    t16 = time.perf_counter() - tb
    same16 = bool((O.G1.to_affine(r16)[0] == O.G1.to_affine(db.msm_resident(ds, n=1 << l16))[0]).all())
    res["config1_n_2p16"] = {"ms": round(t16 * 1e3, 2), "msm_per_s": round(1.0 / t16, 2), "cores": thr16,
                             "sample": "BASELINE config 1: one n=2^%d G1 MSM on the CPU path (%d threads); bit-exact vs GPU: %s" % (l16, thr16, same16)}
    return res


def secondary_configs(log2n, ks, scalars, db, ds, pool, cpu_legs=True, ncpu=1):
    """Everything BASELINE / SURVEY 8d ask for next to the headline, outside the timed region (N = 1 only).  Inputs are synthetic: bases are
    device fixed-base products of seeded scalars (parity of these paths is the GPU test-suite's job, not this one's).

    cpu_legs: res["cpu"] = the CPU path (the oracle: arkworks-shaped code, `kind: "port"`) timed on this box's host cores beside configs 3 and 4
    — G2 MSM, the 1024-pair Miller loop, one proof verified, the witness map, the five MSMs of a proof — each cross-checked bit for bit against
    what the GPU returned for the same inputs.  The oracle is only the timed baseline and the checker here, never the thing measured as `value`."""
    O = None
    if cpu_legs:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_c as O                       # noqa: N812
    cpu = {}

    def cpu_time(fn):
        t0 = time.perf_counter(); r = fn(); return r, (time.perf_counter() - t0) * 1e3

    def win_threads(k):                            # arkworks' rayon structure: one task per window (msm_bigint), so at most `windows` busy cores
        c = O.window_c(k)
        return max(1, min(ncpu, (255 + c - 1) // c))
    import crypto_amd as ca
    from crypto_amd import fixed_base as FB, qap, serde, legogroth16 as LG
    gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(G1_GEN_COMPRESSED))
    gen2, _ = serde.deserialize(ca.G2, bytes.fromhex(G2_GEN_COMPRESSED))
    n = 1 << log2n
    res = {}

    def thr4(fn, k=16):
        """ms per call with four calls in flight (four host threads, parked before the clock: InFlight); warmed concurrently first"""
        run_inflight(fn, 8, 4)
        return min(run_inflight(fn, k, 4)[0] / k * 1e3 for _ in range(2))      # (best of two passes: a box with busy host cores stretches a pass now and then)

    # -- the headline's workload from NATIVE host threads (tests/native/inflight_threads.cpp, a process of its own over the C ABI: what a Rust host's rayon workers
    #    are): resident table, 2^20 terms, 6 / 8 / 12 threads against the library's six slots — no interpreter between the calls
    try:
        import re
        import subprocess
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_cpp_api import build_inflight_driver
        exe = build_inflight_driver()
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "crypto_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([exe, "6,40", "8,40", "12,60"], capture_output=True, text=True, timeout=300, env=env)
        rows = re.findall(r"T=\s*(\d+) count=\s*(\d+):\s*([\d.]+) ms total, ([\d.]+) ms per call, longest call ([\d.]+) ms, mismatches (\d+)", r.stdout)
        nat = {}
        for T_, count_, total_, per_, longest_, bad_ in rows:
            k_ = "%s_threads" % T_
            if k_ not in nat or float(per_) < nat[k_]["ms_per_msm"]:
                nat[k_] = {"ms_per_msm": float(per_), "msm_per_s": round(1e3 / float(per_), 1), "longest_call_ms": float(longest_), "calls": int(count_), "mismatches": int(bad_)}
        nat["note"] = ("dgpu_msm_g1_resident on a precomputed table from T native threads (best of three passes each; results compared with the first call's); callers beyond the six "
                       "slots queue first come, first served, so the longest call stays near T x ms_per_msm")
        res["native_host_threads"] = nat
    except Exception as e:      # noqa: BLE001  (a box without g++: the figure is simply absent)
        res["native_host_threads"] = {"error": repr(e)[:200]}
    with FB.WindowTable(ca.G1, gen1[0]) as t1:
        plain = t1.multiply_many_to_bases(ks)
        host_bases, _ = t1.multiply_many(ks)
    # -- the plain resident pipeline (no table): what a handle costs before dgpu_bases_precompute_g1
    a0 = ca.device_alloc_count()
    nthr = pool._max_workers                        # the headline run's calls in flight

    def thr_pool(fn, k=24):
        """ms per call with as many calls in flight as the headline run uses"""
        run_inflight(fn, 12, nthr)
        return min(run_inflight(fn, k, nthr)[0] / k * 1e3 for _ in range(2))
    res["plain_resident"] = {"latency_ms": round(timed(lambda: plain.msm_resident(ds), 5, warm=3), 3), "ms_per_msm_4_in_flight": round(thr4(lambda: plain.msm_resident(ds), 24), 3),
                             "ms_per_msm_in_flight_like_headline": round(thr_pool(lambda: plain.msm_resident(ds)), 3)}
    res["plain_resident"]["msm_per_s"] = round(1e3 / res["plain_resident"]["ms_per_msm_in_flight_like_headline"], 2)
    res["plain_resident"]["device_allocations"] = int(ca.device_alloc_count() - a0)       # 0: the slots were sized when the handle was created
    # -- H2D-inclusive: fresh host scalars per call against the resident key (32 B/term over PCIe), and the full one-shot
    #    call (bases + scalars from host memory: 128 B/term) — never `value`
    res["h2d_inclusive"] = {
        "resident_table_fresh_scalars_ms": round(timed(lambda: db.msm_bigint(scalars)), 3),
        "resident_table_fresh_scalars_ms_per_msm_4_in_flight": round(thr4(lambda: db.msm_bigint(scalars), 12), 3),
        "resident_plain_fresh_scalars_ms": round(timed(lambda: plain.msm_bigint(scalars)), 3),
        "one_shot_bases_and_scalars_ms": round(timed(lambda: ca.msm_bigint(ca.G1, host_bases, scalars), 5, warm=2), 3),
        "one_shot_strided_affine_structs_ms": round(timed(lambda st=ca.to_affine_structs(ca.G1, host_bases): ca.msm_strided(ca.G1, st, scalars), 5, warm=2), 3),
        "note": "dgpu_msm_g1_handle (upload of n x 32 B scalars inside the call), dgpu_msm_g1 (n x 128 B inside the call) and dgpu_msm_g1_strided (the caller's 104-byte Affine structs: n x 136 B), pageable host memory; "
                "from n = 2^19 the operands cross PCIe in two term ranges: a range's scalars are sorted while its bases are in flight, its kernels run under the next range's copies, the ranges' bucket sets are merged before the one reduction"}
    # -- SURVEY 8d secondary scalar distributions, on the table path, one call in flight
    rng = np.random.Generator(np.random.PCG64(0x5EED0009))
    d = {}
    eq = np.tile(scalars[7], (n, 1))
    d["all_equal_scalars_ms"] = round(timed(lambda: db.msm_bigint(eq), 3), 3)
    s16 = np.zeros((n, 4), np.uint64); s16[:, 0] = rng.integers(0, 1 << 16, n, dtype=np.uint64)
    d["16_bit_scalars_ms"] = round(timed(lambda: db.msm_bigint(s16), 3), 3)
    zo = scalars.copy(); kind = rng.integers(0, 4, n); zo[kind <= 1] = 0; zo[kind == 1, 0] = 1
    d["half_zeros_ones_ms"] = round(timed(lambda: db.msm_bigint(zo), 3), 3)
    d["uniform_ms"] = round(timed(lambda: db.msm_bigint(scalars), 3), 3)
    inf1 = (rng.integers(0, 100, n) == 0).astype(np.uint8)                      # 1 % identity bases (zero QAP rows of a real key, prover.rs:198)
    dbi = ca.DeviceBases(ca.G1, host_bases, inf1).precompute()
    d["one_percent_identity_bases_ms"] = round(timed(lambda: dbi.msm_bigint(scalars), 3), 3)
    dbi.free()
    res["scalar_distributions_fresh_scalars"] = d
    plain.free()
    del eq, s16, zo, host_bases
    # -- BASELINE config 5's 2^24 terms on this ONE GPU: the denominator of the ">= 6x further at 8 GPUs" target
    try:
        res["g1_2p24_single_gpu"] = single_gpu_2p24(ca, FB, gen1, pool)
    except Exception as e:          # noqa: BLE001
        res["g1_2p24_single_gpu"] = {"error": repr(e)}
    # -- the per-rank workload of BASELINE config 5 on 8 GPUs: 2^21 terms on a table, resident, with the headline's calls in flight: the
    #    measured single-GPU quantity behind the ">= 6x at 8 GPUs" projection (8 ranks x this rate / the 1-GPU time of 2^24 terms above)
    try:
        n21 = 1 << 21
        k21 = seeded_scalars(0x5EED2100, n21); s21 = seeded_scalars(0x5EED2101, n21)
        with FB.WindowTable(ca.G1, gen1[0]) as t1:
            b21 = t1.multiply_many_to_bases(k21)
            exp21, _ = t1.multiply(dot_mod_r(k21, s21))
        b21.precompute(); d21 = ca.DeviceScalars(s21)
        ok21 = bool((b21.msm_resident(d21)[:12] == exp21).all())
        lat21 = timed(lambda: b21.msm_resident(d21), 5, warm=2)
        thr21 = thr_pool(lambda: b21.msm_resident(d21), 24)
        res["g1_2p21_per_gpu_share"] = {"latency_ms": round(lat21, 3), "ms_per_msm_in_flight_like_headline": round(thr21, 3), "bit_exact_vs_closed_form": ok21}
        if "g1_2p24_single_gpu" in res and "ms_per_msm_4_in_flight" in res["g1_2p24_single_gpu"]:
            res["g1_2p21_per_gpu_share"]["projected_speedup_8_gpus"] = round(res["g1_2p24_single_gpu"]["ms_per_msm_4_in_flight"] / thr21, 2)
            res["g1_2p21_per_gpu_share"]["note"] = "8 ranks each computing their 2^21-term share at this rate (the 144-byte all_gather is ~20 us) vs one GPU computing all 2^24 terms"
        b21.free(); d21.free(); del k21, s21
    except Exception as e:          # noqa: BLE001
        res["g1_2p21_per_gpu_share"] = {"error": repr(e)}
    # -- the MSM sizes of the reference's call sites above the GPU threshold (the aggregation's halving MSMs, the mult checker): the tree path
    #    (small_kernels.hip.h) one-shot from host memory and on a resident handle (its own table), with the CPU path beside them
    try:
        sm = {}
        with FB.WindowTable(ca.G1, gen1[0]) as ts1:
            for nn in (600, 4096):
                pts, _ = ts1.multiply_many(seeded_scalars(0x5EED0600 + nn, nn)); scn = seeded_scalars(0x5EED0700 + nn, nn)
                ref = ca.msm_bigint(ca.G1, pts, scn)
                hb = ca.DeviceBases(ca.G1, pts); hs = ca.DeviceScalars(scn)
                for _ in range(3):
                    assert (hb.msm_resident(hs) == ref).all()
                e = {"one_shot_ms": round(timed(lambda: ca.msm_bigint(ca.G1, pts, scn), 20, warm=3), 3), "resident_handle_ms": round(timed(lambda: hb.msm_resident(hs), 20, warm=3), 3)}
                if cpu_legs:
                    (rc_, ms1) = min((cpu_time(lambda: O.G1.msm(pts, scn, threads=1)) for _ in range(3)), key=lambda t: t[1])
                    (_, msw) = min((cpu_time(lambda: O.G1.msm(pts, scn, threads=win_threads(nn))) for _ in range(3)), key=lambda t: t[1])
                    e.update({"cpu_one_thread_ms": round(ms1, 3), "cpu_one_thread_per_window_ms": round(msw, 3), "cpu_threads": win_threads(nn),
                              "bit_exact_vs_gpu": bool((O.G1.to_affine(rc_)[0] == ref[:12]).all())})
                sm["g1_n%d" % nn] = e
                hb.free(); hs.free()
        sm["note"] = "dgpu_msm_g1 from host memory / dgpu_msm_g1_resident on a plain handle; below 8193 terms both take the tree path, the handle with its own table of pre-doubled multiples (one launch)"
        res["small_msm"] = sm
    except Exception as e:          # noqa: BLE001
        res["small_msm"] = {"error": repr(e)}
    # -- BASELINE config 3: G2 MSM at the same n (plain and table), 1024-pair Miller loop, final exponentiation
    def t2_twin_bases(nn):                            # the G2 bases of this leg, made again inside whichever library is current
        with FB.WindowTable(ca.G2, gen2[0]) as tt:
            return tt.multiply_many_to_bases(seeded_scalars(0x5EED0003, nn))

    with FB.WindowTable(ca.G2, gen2[0]) as t2, FB.WindowTable(ca.G1, gen1[0]) as t1:
        if cpu_legs:
            g2_host, _ = t2.multiply_many(seeded_scalars(0x5EED0003, n))
            db2 = ca.DeviceBases(ca.G2, g2_host)
        else:
            db2 = t2.multiply_many_to_bases(seeded_scalars(0x5EED0003, n))
        res["g2_msm_plain_ms"] = round(timed(lambda: db2.msm_resident(ds), 3, warm=7), 3)     # (warm-ups: one per slot, their workspaces grow on the first G2 call)
        db2.precompute()
        res["g2_msm_ms"] = round(timed(lambda: db2.msm_resident(ds), 3, warm=7), 3)
        with ca.twin():                                  # (stage timers: the development twin, its own copy of the table)
            db2_t = t2_twin_bases(n); db2_t.precompute(); ds_t = ca.DeviceScalars(scalars)
            for _ in range(2):
                db2_t.msm_resident(ds_t)
            ca.prof.enable(True); ca.prof.reset()
            for _ in range(3):
                db2_t.msm_resident(ds_t)
            st2 = ca.prof.read(); ca.prof.enable(False)
            db2_t.free(); ds_t.free()
        acc2 = st2.get("msm.accumulate", (0.0, 1)); acc2 = acc2[0] / max(1, acc2[1])
        sh2 = db2.table_shape()
        if acc2 > 0 and sh2:
            mads2 = float(n) * sh2[2] * MADS_PER_G2_MIXED_ADD
            res["g2_stages_ms_one_in_flight"] = {k.replace("msm.", ""): round(v[0] / max(1, v[1]), 4) for k, v in st2.items()}
            res["g2_valu_roofline"] = {"bound": "v_mad_u64_u32", "kernel": "msm::k_accumulate<msm::G2P, false>", "avg_ms": round(acc2, 4), "achieved": round(mads2 / (acc2 * 1e-3) / 1e12, 3), "peak": MAD_PEAK,
                                       "unit": "Tmad/s", "frac": round(mads2 / (acc2 * 1e-3) / 1e12 / MAD_PEAK, 4), "mixed_additions_per_launch": int(n) * sh2[2],
                                       "note": "%d windows x n mixed additions over Fp2 x %d v_mad_u64_u32 each (two lanes per point; schoolbook Fp2 product = two fused two-product reductions per lane)" % (sh2[2], MADS_PER_G2_MIXED_ADD),
                                       "hbm_roofline": {"achieved_GB_s": round(224.0 * n / (acc2 * 1e-3) / 1e9, 1), "frac": round(224.0 * n / (acc2 * 1e-3) / 1e9 / 8000.0, 5), "note": "algorithmic 224 B/term (SURVEY 8d)"}}
        res["g2_msm_ms_per_msm_4_in_flight"] = round(thr4(lambda: db2.msm_resident(ds), 8), 3)
        res["g2_msm_per_s"] = round(1e3 / res["g2_msm_ms_per_msm_4_in_flight"], 2)
        if cpu_legs:
            thr = win_threads(n)
            ref2, ms2 = cpu_time(lambda: O.G2.msm(g2_host, scalars, threads=thr))
            same2 = bool((O.G2.to_affine(ref2)[0] == O.G2.to_affine(db2.msm_resident(ds))[0]).all())
            cpu["g2_msm"] = {"cpu_ms": round(ms2, 1), "gpu_ms": res["g2_msm_ms"], "x": round(ms2 / res["g2_msm_ms"], 1), "x_4_in_flight": round(ms2 / res["g2_msm_ms_per_msm_4_in_flight"], 1),
                             "cores": thr, "bit_exact_vs_gpu": same2, "sample": "one n=2^%d G2 MSM, one thread per window (%d windows) of %d logical CPUs" % (log2n, thr, ncpu)}
            del g2_host
        P, _ = t1.multiply_many(seeded_scalars(0x5EED0005, 1024)); Q, _ = t2.multiply_many(seeded_scalars(0x5EED0006, 1024))
    f = ca.multi_miller_loop(P, Q)
    res["miller_loop_1024_pairs_ms"] = round(timed(lambda: ca.multi_miller_loop(P, Q), 20, warm=3), 3)      # (millisecond calls: 20 of them after 3 warm-ups)
    res["miller_loop_pairs_per_s"] = round(1024 / res["miller_loop_1024_pairs_ms"] * 1e3, 0)
    # the same 1024-pair loop from six host threads (one call is a chain of 68 dependent steps on 64 waves: the chip has room for several)
    run_inflight(lambda: ca.multi_miller_loop(P, Q), 12, 6)
    dt, fs = run_inflight(lambda: ca.multi_miller_loop(P, Q), 60, 6); dt = dt / 60 * 1e3
    assert all((g == f).all() for g in fs)
    res["miller_loop_1024_pairs_ms_per_call_6_in_flight"] = round(dt, 3)
    res["miller_loop_pairs_per_s_6_in_flight"] = round(1024 / dt * 1e3, 0)
    from crypto_amd import pairing
    pc = pairing.G2Prepared.from_affine(Q)
    assert (pairing.multi_miller_loop(P, pc) == f).all()
    res["g2_prepare_1024_ms"] = round(timed(lambda: pairing.G2Prepared.from_affine(Q), 10, warm=2), 3)
    res["miller_loop_1024_prepared_pairs_ms"] = round(timed(lambda: pairing.multi_miller_loop(P, pc), 20, warm=3), 3)
    res["final_exponentiation_ms"] = round(timed(lambda: ca.final_exponentiation(f), 20, warm=3), 3)
    if cpu_legs:
        thr_ml = max(1, min(ncpu, 64, 256))                                                  # 256 chunks of 4 pairs (ark-ec's rayon chunks_mut(4)): every core has work
        O.multi_miller_loop(P[:8], Q[:8], threads=thr_ml)
        (fc, ms_ml) = min((cpu_time(lambda: O.multi_miller_loop(P, Q, threads=thr_ml)) for _ in range(3)), key=lambda t: t[1])
        cpu["miller_1024"] = {"cpu_ms": round(ms_ml, 2), "gpu_ms": res["miller_loop_1024_pairs_ms"], "x": round(ms_ml / res["miller_loop_1024_pairs_ms"], 2),
                              "x_6_in_flight": round(ms_ml / res["miller_loop_1024_pairs_ms_per_call_6_in_flight"], 2), "cores": thr_ml, "bit_exact_vs_gpu": bool((fc == f).all()),
                              "sample": "multi_miller_loop over 1024 affine pairs (G2Prepared::from inside, like the GPU call), chunks of 4 pairs over %d threads, best of 3" % thr_ml}
        (fc1, ms_ml1) = cpu_time(lambda: O.multi_miller_loop(P, Q, threads=1))
        cpu["miller_1024"]["one_thread_ms"] = round(ms_ml1, 1)
        (ge, ms_fe) = min((cpu_time(lambda: O.final_exponentiation(f)) for _ in range(5)), key=lambda t: t[1])
        cpu["final_exponentiation"] = {"cpu_ms": round(ms_fe, 3), "gpu_lib_ms": res["final_exponentiation_ms"], "cores": 1, "bit_exact_vs_gpu": bool((ge == ca.final_exponentiation(f)).all()),
                                       "sample": "one final exponentiation on one core (host code on both sides: the library's runs on the host too, SURVEY 8a6)"}
    # -- the verifier's side of the same path (verifier.rs:62-99, randomized_pairing_check.rs): 1024 Groth16-shaped proofs with known discrete
    #    logs (a_i b_i = alpha beta + (g0 + x_i g1 + d_i) gamma + c_i delta, so every one of them verifies), one at a time and batched
    from crypto_amd import legogroth16 as LGv
    nv = 1024
    ints = lambda seed, k: [int(x[0]) | (int(x[1]) << 64) | (int(x[2]) << 128) | (int(x[3]) << 192) for x in seeded_scalars(seed, k)]
    al, be, ga, de, g0, g1x = ints(0x5EED0020, 6)
    av, bv, dv, xv = ints(0x5EED0021, nv), ints(0x5EED0022, nv), ints(0x5EED0023, nv), ints(0x5EED0024, nv)
    dinv = pow(de, R_MOD - 2, R_MOD)
    cv = [((a * b - al * be - (g0 + x * g1x + d) * ga) * dinv) % R_MOD for a, b, d, x in zip(av, bv, dv, xv)]
    lim = lambda vals: np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64)
    with FB.WindowTable(ca.G2, gen2[0]) as t2, FB.WindowTable(ca.G1, gen1[0]) as t1:
        A_, _ = t1.multiply_many(lim(av)); C_, _ = t1.multiply_many(lim(cv)); D_, _ = t1.multiply_many(lim(dv)); K_, _ = t1.multiply_many(lim([al, g0, g1x, 1]))
        B_, _ = t2.multiply_many(lim(bv)); V_, _ = t2.multiply_many(lim([be, ga, de]))
    vkv = LGv.VerifyingKey(K_[0], V_[0], V_[1], V_[2], K_[1:3], K_[3], 0)
    pvkv = LGv.prepare_verifying_key(vkv)
    proofs_v = [{"a": A_[i], "b": B_[i], "c": C_[i], "d": D_[i]} for i in range(nv)]
    pubs_v = [lim([x]) for x in xv]
    assert LGv.verify_proof(pvkv, proofs_v[0], pubs_v[0]) and not LGv.verify_proof(pvkv, proofs_v[0], pubs_v[1])
    # one proof verified: ONE call of the C ABI (dgpu_legogroth16_verify, verifier.rs:62-99) — and the three calls a host makes without it
    # (calculate_d, the mixed Miller loop, the final exponentiation), for comparison
    assert LGv.verify_proof_abi(pvkv, proofs_v[0], pubs_v[0]) and not LGv.verify_proof_abi(pvkv, proofs_v[0], pubs_v[1])
    res["verify_one_proof_ms"] = round(timed(lambda: LGv.verify_proof_abi(pvkv, proofs_v[1], pubs_v[1]), 20, warm=3), 3)
    res["verify_one_proof_three_calls_ms"] = round(timed(lambda: LGv.verify_proof(pvkv, proofs_v[1], pubs_v[1]), 20, warm=3), 3)
    if cpu_legs:
        # verifier.rs:62-99 on one core (three pairs are one rayon chunk): d = gamma_abc[0] + x gamma_abc[1] + proof.d, the Miller loop over
        # [(A, B affine), (C, -delta prepared), (d, -gamma prepared)], the final exponentiation, the comparison with e(alpha, beta)
        pre_cpu = np.stack([O.g2_prepare(LGv._neg_affine(ca.G2, V_[2])).reshape(-1), O.g2_prepare(LGv._neg_affine(ca.G2, V_[1])).reshape(-1)])   # -delta, -gamma: prepared once per key
        ab_cpu = O.final_exponentiation(O.multi_miller_loop(K_[0].reshape(1, 12), V_[0].reshape(1, 24)))

        def cpu_verify(pr, pub):
            acc = O.G1.add(np.concatenate([K_[1], pairing.FP_ONE_MONT]), O.G1.mul(K_[2], pub[0]))
            acc = O.G1.add(acc, np.concatenate([pr["d"], pairing.FP_ONE_MONT]))
            dpt, _ = O.G1.to_affine(acc)
            fm = O.multi_miller_loop_mixed(pr["a"].reshape(1, 12), pr["b"].reshape(1, 24), np.stack([pr["c"], dpt]), pre_cpu, threads=1)
            return bool((O.final_exponentiation(fm) == ab_cpu).all())
        assert cpu_verify(proofs_v[1], pubs_v[1]) and not cpu_verify(proofs_v[1], pubs_v[2])
        (okv, ms_v) = min((cpu_time(lambda: cpu_verify(proofs_v[1], pubs_v[1])) for _ in range(5)), key=lambda t: t[1])
        cpu["verify_one"] = {"cpu_ms": round(ms_v, 3), "gpu_ms": res["verify_one_proof_ms"], "x": round(ms_v / res["verify_one_proof_ms"], 2), "cores": 1,
                             "bit_exact_vs_gpu": bool((ab_cpu == pvkv["alpha_g1_beta_g2"]).all()) and okv,
                             "sample": "verify_proof of one LegoGroth16 proof (one public input; -delta, -gamma prepared in the key) on ONE core: three pairs are a single rayon chunk in ark-ec, best of 5"}
    assert LGv.verify_proofs_batch(pvkv, proofs_v, pubs_v, 0x5EED0025) and LGv.verify_proofs_batch_merged(pvkv, proofs_v, pubs_v, 0x5EED0026)
    swapped = list(proofs_v); swapped[7] = dict(swapped[7], c=proofs_v[8]["c"])
    assert not LGv.verify_proofs_batch_merged(pvkv, swapped, pubs_v, 0x5EED0027)
    res["verify_1024_proofs_pairing_checker_ms"] = round(timed(lambda: LGv.verify_proofs_batch(pvkv, proofs_v, pubs_v, 0x5EED0028), 3), 2)
    res["verify_1024_proofs_merged_ms"] = round(timed(lambda: LGv.verify_proofs_batch_merged(pvkv, proofs_v, pubs_v, 0x5EED0029), 3), 2)
    res["verify_proofs_per_s_merged"] = round(nv / res["verify_1024_proofs_merged_ms"] * 1e3, 0)
    # the same merged check as ONE call of the C ABI (dgpu_legogroth16_verify_batch: scalings, both MSMs and the GT power side by side inside the library)
    assert LGv.verify_proofs_batch_abi(pvkv, proofs_v, pubs_v, 0x5EED0029) and not LGv.verify_proofs_batch_abi(pvkv, swapped, pubs_v, 0x5EED0027)
    packed_v = LGv.pack_proofs(proofs_v, pubs_v)                        # (the column form the ABI takes: a Rust host's &[Proof] costs microseconds to lay out, 1024 Python dictionaries ~2 ms)
    res["verify_1024_proofs_one_call_ms"] = round(timed(lambda: LGv.verify_proofs_batch_abi(pvkv, None, None, 0x5EED0029, packed=packed_v), 24, warm=36), 3)      # (the call runs four pieces side by side since round 6 — scaled Miller loop, two MSMs, the prepared pairs' loop — and they rotate through the context's six slots: a few dozen calls until every slot has seen every buffer size)
    res["verify_1024_proofs_one_call_incl_python_packing_ms"] = round(timed(lambda: LGv.verify_proofs_batch_abi(pvkv, proofs_v, pubs_v, 0x5EED0029), 3, warm=1), 3)
    res["verify_proofs_per_s_one_call"] = round(nv / res["verify_1024_proofs_one_call_ms"] * 1e3, 0)
    if cpu_legs:
        # ---- the f-rows on the host's cores: the same algorithms with the oracle's pieces, threaded where the reference's rayon is ----
        thr = max(1, min(ncpu, 64))
        to_l = lambda v: np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        # (1) G2Prepared::from x 1024 (randomized_pairing_check.rs:132,163; verifier.rs:22-23): arkworks converts inside a serial `.map(Into::into)`, so one thread is
        #     the reference's shape; all cores beside it
        (pc_cpu, ms_p1) = cpu_time(lambda: O.g2_prepare_batch(Q, threads=1))
        (_, ms_pt) = min((cpu_time(lambda: O.g2_prepare_batch(Q, threads=thr)) for _ in range(3)), key=lambda t: t[1])
        cpu["g2_prepare_1024"] = {"cpu_ms_one_thread": round(ms_p1, 1), "cpu_ms": round(ms_pt, 2), "cores": thr, "gpu_ms": res["g2_prepare_1024_ms"], "x": round(ms_pt / res["g2_prepare_1024_ms"], 2),
                                  "x_vs_one_thread": round(ms_p1 / res["g2_prepare_1024_ms"], 1), "bit_exact_vs_gpu": bool((pc_cpu == pc.coeffs).all()),
                                  "sample": "G2Prepared::from of 1024 points (68 x 3 Fp2 line coefficients each); the reference converts serially (`b.into()` inside collect), all %d threads beside it" % thr}
        # (2) 1024 proofs through the lazy RandomizedPairingChecker exactly as the reference runs it (utils/src/randomized_pairing_check.rs:116-138,204-214): per proof three
        #     scalings by random^k (rayon), the target's e(alpha, beta)^(random^k), then ONE Miller loop over 3 x 1024 pairs (B_i prepared serially inside, -delta / -gamma
        #     prepared in the key) and one final exponentiation
        rnd_c = 0x5EED0028 % R_MOD
        ms_pow, mm = [], 1
        for _ in range(nv):
            ms_pow.append(mm); mm = mm * rnd_c % R_MOD
        m_l = np.stack([to_l(x) for x in ms_pow])
        gabc0 = np.concatenate([K_[1], pairing.FP_ONE_MONT])

        def cpu_checker():
            xg, _ = O.g1_scale_batch(np.repeat(K_[2].reshape(1, 12), nv, 0), np.stack([p[0] for p in pubs_v]), threads=thr)          # calculate_d: x_i gamma_abc[1] ...
            dps = []
            for i in range(nv):                                                                                                  # ... + gamma_abc[0] + proof.d
                acc = O.G1.add(O.G1.add(gabc0, np.concatenate([xg[i], pairing.FP_ONE_MONT])), np.concatenate([D_[i], pairing.FP_ONE_MONT]))
                dps.append(O.G1.to_affine(acc)[0])
            a_m, _ = O.g1_scale_batch(A_, m_l, threads=thr); c_m, _ = O.g1_scale_batch(C_, m_l, threads=thr); d_m, _ = O.g1_scale_batch(np.stack(dps), m_l, threads=thr)
            right = O.fp12_multi_pow(pvkv["alpha_g1_beta_g2"], m_l, threads=thr)
            co = np.concatenate([np.repeat(pre_cpu[0].reshape(1, -1), nv, 0), np.repeat(pre_cpu[1].reshape(1, -1), nv, 0)])
            fm = O.multi_miller_loop_mixed(a_m, B_, np.concatenate([c_m, d_m]), co, threads=thr)
            return bool((O.final_exponentiation(fm) == right).all())
        (ok_c, ms_chk) = cpu_time(cpu_checker)
        cpu["verify_1024_checker"] = {"cpu_ms": round(ms_chk, 1), "gpu_ms": res["verify_1024_proofs_pairing_checker_ms"], "x": round(ms_chk / res["verify_1024_proofs_pairing_checker_ms"], 1), "cores": thr,
                                      "accepts": ok_c, "sample": "1024 LegoGroth16 proofs through the lazy pairing checker as the reference runs it: 3 x 1024 scalings + 1024 calculate_d + 1024 GT powers on %d threads, "
                                                                 "1024 G2Prepared::from serially inside ONE 3072-pair Miller loop (chunks of 4 pairs on %d threads), one final exponentiation" % (thr, thr)}
        # (3) the merged form (what verify_1024_proofs_merged_ms times): N scalings, two N-term MSMs, ONE (N + 2)-pair Miller loop; every piece compared with the library's
        from crypto_amd.pairing_check import g1_scale_each
        c_sc = np.stack([to_l(sum(ms_pow) % R_MOD), to_l(sum(mi * xi for mi, xi in zip(ms_pow, xv)) % R_MOD)] + [m_l[i] for i in range(nv)])
        d_pts = np.concatenate([K_[1:3], D_])

        def cpu_merged():
            a_m, a_i = O.g1_scale_batch(A_, m_l, threads=thr)
            c_sum, _ = O.G1.to_affine(O.G1.msm(C_, m_l, threads=win_threads(nv)))
            d_sum, _ = O.G1.to_affine(O.G1.msm(d_pts, c_sc, threads=win_threads(nv)))
            fm = O.multi_miller_loop_mixed(a_m, B_, np.stack([c_sum, d_sum]), pre_cpu, threads=thr)
            right = O.fp12_pow(pvkv["alpha_g1_beta_g2"], sum(ms_pow) % R_MOD)
            return a_m, c_sum, d_sum, fm, bool((O.final_exponentiation(fm) == right).all())
        ((a_c, cs_c, ds_c, fm_c, ok_m), ms_mrg) = min((cpu_time(cpu_merged) for _ in range(2)), key=lambda t: t[1])
        a_g, _ = g1_scale_each(A_, m_l, None)
        cs_g = ca.msm_bigint(ca.G1, C_, m_l); ds_g = ca.msm_bigint(ca.G1, d_pts, c_sc)
        fm_g = pairing.multi_miller_loop(np.concatenate([a_g, cs_g[:12].reshape(1, 12), ds_g[:12].reshape(1, 12)]), [B_, pvkv["delta_g2_neg_pc"], pvkv["gamma_g2_neg_pc"]])
        cpu["verify_1024_merged"] = {"cpu_ms": round(ms_mrg, 1), "gpu_ms": res["verify_1024_proofs_merged_ms"], "x": round(ms_mrg / res["verify_1024_proofs_merged_ms"], 1),
                                     "gpu_one_call_ms": res["verify_1024_proofs_one_call_ms"], "x_one_call": round(ms_mrg / res["verify_1024_proofs_one_call_ms"], 1), "cores": thr, "accepts": ok_m,
                                     "bit_exact_vs_gpu": bool((a_c == a_g).all() and (cs_c == cs_g[:12]).all() and (ds_c == ds_g[:12]).all() and (fm_c == fm_g).all()),
                                     "sample": "the classical Groth16 batch verifier: 1024 scalings (%d threads), two 1024-term G1 MSMs (one thread per window), ONE 1026-pair Miller loop, one final exponentiation; "
                                               "scaled points, both MSM results and the raw Fp12 Miller output compared with the library's limb for limb" % thr}
        # (4) what a SnarkPack aggregation of these 1024 proofs spends its time in (legogroth16/src/aggregation/groth16/prover.rs:212-382, commitment.rs:23-69): per GIPA round of half
        #     length s = 512 .. 1, fourteen multi-pairings of s pairs (the left / right commitments T, U of (A, B), C, D and the cross terms z) and four s-term G1 MSMs (the MIPP
        #     cross terms of C and D against the challenge powers).  Round by round on the host's cores; the GPU number beside it is the WHOLE aggregation
        def cpu_gipa():
            gts = []
            s_ = nv // 2
            while s_ >= 1:
                for k in range(14):
                    lo = (k * 37) % (nv - s_ + 1)
                    gts.append(O.final_exponentiation(O.multi_miller_loop(P[lo:lo + s_], Q[lo:lo + s_], threads=thr)))
                for k in range(4):
                    O.G1.msm(C_[:s_], m_l[k:k + s_], threads=win_threads(s_) if s_ >= 32 else 1)
                s_ //= 2
            return gts
        (gts, ms_gipa) = cpu_time(cpu_gipa)
        gt_g = ca.final_exponentiation(ca.multi_miller_loop(P[:nv // 2], Q[:nv // 2]))
        cpu["snarkpack_aggregate_dominant_ops"] = {"cpu_ms": round(ms_gipa, 1), "cores": thr, "bit_exact_vs_gpu": bool((gts[0] == gt_g).all()),
                                                   "sample": "ten GIPA rounds of a 1024-proof aggregation: 14 multi-pairings (Miller loop on %d threads + final exponentiation) and 4 G1 MSMs per round at half "
                                                             "lengths 512 .. 1 — the prover's dominant operations only (no KZG openings, no folding, no transcript); first multi-pairing compared with the library's" % thr}
    # -- SnarkPack aggregation of the same 1024 LegoGroth16 proofs (aggregation/legogroth16/prover.rs:38-127: TIPP for (A, B), MIPP for
    #    C and D, the KZG openings) and verification of the aggregate (verifier.rs:34-96) — the producer of the segmented Miller loops and
    #    of the endomorphism-split folding steps
    from crypto_amd import aggregation as AGv
    from crypto_amd.aggregation import legogroth16 as ALv
    s_al, s_be = ints(0x5EED002A, 2)
    pk_ag, vsrs_ag = AGv.setup_fake_srs(s_al, s_be, nv, gen1[0], gen2[0]).specialize(nv)
    pubs_int = [[x] for x in xv]
    # (the verifier is timed the way the reference runs it: on an AggregateProof that deserialisation has already validated; the validating form — what a proof from an
    #  untrusted source needs, the wrappers' default — is timed beside it)
    TRUSTED = dict(validate_gt=False, validate_points=False)
    agg_v = ALv.aggregate_proofs(pk_ag, AGv.MerlinTranscript(b"bench"), proofs_v)
    ALv.verify_aggregate_proof(vsrs_ag, {"vk": vkv}, pubs_int, agg_v, 0x5EED002B, AGv.MerlinTranscript(b"bench"), validate_gt=False)      # raises if invalid
    # the library's own aggregator / verifier (dgpu_snarkpack_aggregate / _verify: the protocol in C++ inside libdock_gpu.so, the transcript called back)
    # is the product; the Python statement of the protocol above the ABI, which the tests compare it with, is timed beside it
    from crypto_amd.aggregation import native as ANv
    words_v = ANv.aggregate_proofs_words(pk_ag, AGv.MerlinTranscript(b"bench"), proofs_v, with_d=True)
    assert (ANv.proof_to_words(agg_v) == words_v).all()
    ANv.verify_aggregate_proof(vsrs_ag, {"vk": vkv}, pubs_int, words_v, 0x5EED002B, AGv.MerlinTranscript(b"bench"), with_d=True, **TRUSTED)       # raises if invalid
    res["snarkpack_aggregate_1024_proofs_ms"] = round(timed(lambda: ANv.aggregate_proofs_words(pk_ag, AGv.MerlinTranscript(b"bench"), proofs_v, with_d=True), 3), 2)
    res["snarkpack_verify_aggregate_ms"] = round(timed(lambda: ANv.verify_aggregate_proof(vsrs_ag, {"vk": vkv}, pubs_int, words_v, 0x5EED002C, AGv.MerlinTranscript(b"bench"), with_d=True, **TRUSTED), 3), 2)
    ANv.aggregate_proofs_words(pk_ag, AGv.MerlinTranscript(b"bench"), proofs_v, with_d=True); t_ag = ANv.LAST["transcript_ms"]
    ANv.verify_aggregate_proof(vsrs_ag, {"vk": vkv}, pubs_int, words_v, 0x5EED002C, AGv.MerlinTranscript(b"bench"), with_d=True, **TRUSTED); t_vf = ANv.LAST["transcript_ms"]
    # the same two calls with the caller's transcript as C callbacks (crypto_amd/aggregation/merlin_native.c, byte for byte the Python Merlin): what a Rust
    # host's merlin::Transcript costs the library — the figure to hold against the reference, whose transcript is compiled code too
    NT = AGv.NativeMerlinTranscript
    assert (ANv.aggregate_proofs_words(pk_ag, NT(b"bench"), proofs_v, with_d=True) == words_v).all()
    ANv.verify_aggregate_proof(vsrs_ag, {"vk": vkv}, pubs_int, words_v, 0x5EED002B, NT(b"bench"), with_d=True, **TRUSTED)
    res["snarkpack_aggregate_1024_proofs_native_transcript_ms"] = round(timed(lambda: ANv.aggregate_proofs_words(pk_ag, NT(b"bench"), proofs_v, with_d=True), 3), 2)
    res["snarkpack_verify_aggregate_native_transcript_ms"] = round(timed(lambda: ANv.verify_aggregate_proof(vsrs_ag, {"vk": vkv}, pubs_int, words_v, 0x5EED002C, NT(b"bench"), with_d=True, **TRUSTED), 3), 2)
    res["snarkpack_verify_aggregate_validating_native_transcript_ms"] = round(timed(lambda: ANv.verify_aggregate_proof(vsrs_ag, {"vk": vkv}, pubs_int, words_v, 0x5EED002C, NT(b"bench"), with_d=True), 3), 2)
    res["snarkpack_python_transcript_ms"] = {"aggregate": round(t_ag, 2), "verify": round(t_vf, 2),
                                             "note": "part of the two numbers above spent inside the Python Merlin transcript the library calls back (a Rust caller's merlin::Transcript costs microseconds)"}
    if cpu_legs and "snarkpack_aggregate_dominant_ops" in cpu:
        cpu["snarkpack_aggregate_dominant_ops"]["gpu_whole_aggregation_ms"] = res["snarkpack_aggregate_1024_proofs_ms"]
        cpu["snarkpack_aggregate_dominant_ops"]["x"] = round(cpu["snarkpack_aggregate_dominant_ops"]["cpu_ms"] / res["snarkpack_aggregate_1024_proofs_ms"], 1)
    res["snarkpack_aggregate_1024_proofs_python_host_ms"] = round(timed(lambda: ALv.aggregate_proofs(pk_ag, AGv.MerlinTranscript(b"bench"), proofs_v), 3), 2)
    res["snarkpack_verify_aggregate_python_host_ms"] = round(timed(lambda: ALv.verify_aggregate_proof(vsrs_ag, {"vk": vkv}, pubs_int, agg_v, 0x5EED002C, AGv.MerlinTranscript(b"bench"), validate_gt=False), 3), 2)
    # -- BASELINE config 4: witness map on the x_i = x_{i-1}^2 + i circuit shape (m + 1 constraints + 2 instance variables = D), circuit resident,
    #    and LegoGroth16 create_proof (prover.rs:267-383) on a synthetic key of that size with every query a precomputed table
    m = n - 3
    idx = np.arange(m, dtype=np.uint32)
    one = np.zeros((1, 4), np.uint64); one[0, 0] = 1
    a_rp = np.arange(m + 2, dtype=np.uint64); a_cl = np.concatenate([2 + idx, [2 + m]]).astype(np.uint32); a_vl = np.repeat(one, m + 1, 0)
    b_cl = np.concatenate([2 + idx, [0]]).astype(np.uint32)
    c_rp = np.concatenate([2 * np.arange(m + 1, dtype=np.uint64), [2 * m + 1]]).astype(np.uint64)
    c_cl = np.concatenate([np.stack([3 + idx, np.zeros(m, np.uint32)], 1).reshape(-1), [1]]).astype(np.uint32)
    make_circuit = lambda: qap.DeviceR1cs((a_rp, a_cl, a_vl), (a_rp, b_cl, a_vl), (c_rp, c_cl, np.repeat(one, 2 * m + 1, 0)), m + 3, 2, m + 1)
    circ = make_circuit()
    z = seeded_scalars(0x5EED0007, m + 3)
    rngw = np.random.Generator(np.random.PCG64(0x5EED0008))       # Groth16-like witness: half in {0, 1}, a quarter 16-bit, a quarter full-size
    kd = rngw.integers(0, 4, m + 3)
    z[kd <= 1] = 0; z[kd == 1, 0] = rngw.integers(0, 2, int((kd == 1).sum()), dtype=np.uint64)
    mk = kd == 2; z[mk, 1:] = 0; z[mk, 0] &= np.uint64(0xFFFF)

    def wm():
        _, dh = circ.witness_map(z, to_host=False, resident=True)
        dh.free()
    res["witness_map_ms"] = round(timed(wm, 3), 3)
    if cpu_legs:
        mats_cpu = ((a_rp, a_cl, a_vl), (a_rp, b_cl, a_vl), (c_rp, c_cl, np.repeat(one, 2 * m + 1, 0)))
        thr_wm = max(1, min(ncpu, 64))
        h_cpu, ms_wm = cpu_time(lambda: O.witness_map(mats_cpu, z, 2, m + 1, threads=thr_wm))
        h_gpu, _ = circ.witness_map(z, to_host=True)
        cpu["witness_map"] = {"cpu_ms": round(ms_wm, 1), "gpu_ms": res["witness_map_ms"], "x": round(ms_wm / res["witness_map_ms"], 1), "cores": thr_wm, "bit_exact_vs_gpu": bool((h_cpu == h_gpu).all()),
                              "sample": "r1cs_to_qap witness map, D = 2^%d (three sparse mat-vecs, 3 iFFT + 3 coset FFT + pointwise + coset iFFT), rows / butterflies split over %d threads" % (log2n, thr_wm)}
        del h_gpu
    with ca.twin():                                      # (stage timers: the development twin, its own resident circuit)
        circ_t = make_circuit()
        for _ in range(2):
            _, dh = circ_t.witness_map(z, to_host=False, resident=True); dh.free()
        ca.prof.enable(True); ca.prof.reset()
        for _ in range(3):
            _, dh = circ_t.witness_map(z, to_host=False, resident=True); dh.free()
        st = ca.prof.read(); ca.prof.enable(False)
        circ_t.free()
    ntt_ms = st.get("qap.ntt", (0.0, 1)); ntt_ms = ntt_ms[0] / max(1, ntt_ms[1])
    if ntt_ms > 0:
        alg = 7 * 2 * 32.0 * n                                # 7 transforms x (read + write) x 32 B per element
        res["ntt_roofline"] = {"bound": "hbm", "kernel": "k_ntt_r4: the seven transforms of one witness map (9 launches: a, b, c go through every pass together; (ab - c)/Z and the final scaling / un-reversal are fused into the last transform)", "achieved": round(alg / (ntt_ms * 1e-3) / 1e9, 1),
                               "peak": 8000.0, "unit": "GB/s", "frac": round(alg / (ntt_ms * 1e-3) / 1e9 / 8000.0, 4), "avg_ms": round(ntt_ms, 3),
                               "matvec_incl_upload_ms": round(st.get("qap.matvec", (0.0, 1))[0] / max(1, st.get("qap.matvec", (0.0, 1))[1]), 3),
                               "note": "algorithmic 448 B per domain element; the transforms are VALU-issue bound (73 M butterflies of one 10-limb Fr product each; rocprof: vector ALU issuing ~89 % of the time in the batched passes), not HBM bound"}
    cw = 2
    V = m + 2                                          # variables 1 .. m + 2 pair with query[1..]
    with FB.WindowTable(ca.G2, gen2[0]) as t2, FB.WindowTable(ca.G1, gen1[0]) as t1:
        small1, _ = t1.multiply_many(seeded_scalars(0x5EED0010, 8 + 2 + cw)); small2, _ = t2.multiply_many(seeded_scalars(0x5EED0011, 4))
        # the queries that meet the witness take DGPU_TABLE_C_WITNESS (fewer buckets to reduce per MSM: include/dock_gpu.h), the h query the automatic width
        wc_ = ca.TABLE_C_WITNESS
        hostq = {}

        def query(tab, curve, name, seed, k, width=None):
            """a proving-key query of k points: straight into HBM, or (CPU legs) through a host copy the CPU path multiplies too"""
            # (a host copy of every query: the CPU path multiplies it, and the drop-in leg below hands it over the way a Rust host holds its key)
            hostq[name], _ = tab.multiply_many(seeded_scalars(seed, k))
            q = ca.DeviceBases(curve, hostq[name])
            return q.precompute(width) if width else q.precompute()
        qa = query(t1, ca.G1, "a", 0x5EED0012, V + 1, wc_)
        qb1 = query(t1, ca.G1, "b1", 0x5EED0013, V + 1, wc_)
        qb2 = query(t2, ca.G2, "b2", 0x5EED0014, V + 1, wc_)
        qh = query(t1, ca.G1, "h", 0x5EED0015, n - 1)
        ql = query(t1, ca.G1, "l", 0x5EED0016, m + 1 - cw, wc_)
    vk = LG.VerifyingKey(small1[0], small2[0], small2[1], small2[2], small1[8:8 + 2 + cw], small1[1], cw)
    pk = LG.ProvingKey.from_device(vk, small1[2], small1[3], small1[4], small1[5], small1[6], small2[3], qa, qb1, qb2, qh, ql)

    def prove():
        return LG.create_proof_with_reduction(pk, circ, 123456789, 987654321, 555, z)
    p0 = prove()
    ms = timed(prove, 6, warm=6)                     # (warm-ups: every slot's workspace grows on its first call of a size)
    assert all((prove()[k] == p0[k]).all() for k in p0)
    res["prove_2p20_ms"] = round(ms, 2)
    res["prove_constraints_per_s"] = round((m + 1) / (ms * 1e-3), 1)
    # a proving service's rate: four proofs in flight from host threads (the library queues calls beyond its six slots); every proof equal to p0
    run_inflight(prove, 8, 4)
    ms4, many = run_inflight(prove, 16, 4); ms4 = ms4 / 16 * 1e3
    assert all((q[k] == p0[k]).all() for q in many for k in p0)
    res["prove_2p20_ms_per_proof_4_in_flight"] = round(ms4, 2)
    res["prove_constraints_per_s_4_in_flight"] = round((m + 1) / (ms4 * 1e-3), 1)
    if cpu_legs:
        # the reference's prover on the host cores (prover.rs:284-359): witness map, then the five MSMs ONE AFTER THE OTHER (each one rayon-parallel
        # over its windows), in the reference's order; the O(1) finish is left out on the CPU side (it is inside the GPU number).  Every MSM is
        # cross-checked against the GPU's result for the same query and scalars.
        n_aux, aux_at = (m + 1) - cw, 1 + 1 + cw
        dzc = ca.DeviceScalars(z)
        _, dhc = circ.witness_map(dzc, to_host=False, resident=True)
        legs = [("h_query MSM (D - 1 terms)", O.G1, hostq["h"], h_cpu[:n - 1], lambda: qh.msm_resident(dhc, n=min(qh.n, dhc.n))),
                ("l_query MSM", O.G1, hostq["l"][:n_aux], z[aux_at:aux_at + n_aux], lambda: ql.msm_resident(dzc, n=min(ql.n, n_aux), scalar_offset=aux_at)),
                ("a_query MSM", O.G1, hostq["a"][1:], z[1:], lambda: qa.msm_resident(dzc, n=V, base_offset=1, scalar_offset=1)),
                ("b_g1_query MSM", O.G1, hostq["b1"][1:], z[1:], lambda: qb1.msm_resident(dzc, n=V, base_offset=1, scalar_offset=1)),
                ("b_g2_query MSM (G2)", O.G2, hostq["b2"][1:], z[1:], lambda: qb2.msm_resident(dzc, n=V, base_offset=1, scalar_offset=1))]
        spans_cpu = {"R1CS to QAP witness map": round(ms_wm, 1)}
        same_all = cpu["witness_map"]["bit_exact_vs_gpu"]
        tot_cpu = ms_wm
        for name, G_, bq, sq, gpu_fn in legs:
            k_ = min(len(bq), len(sq))
            thr = win_threads(k_)
            rj, msj = cpu_time(lambda: G_.msm(bq[:k_], sq[:k_], threads=thr))
            same_all = same_all and bool((G_.to_affine(rj)[0] == G_.to_affine(gpu_fn())[0]).all())
            spans_cpu[name] = round(msj, 1); tot_cpu += msj
        dhc.free(); dzc.free()
        cpu["prove"] = {"cpu_ms": round(tot_cpu, 1), "gpu_ms": res["prove_2p20_ms"], "x": round(tot_cpu / res["prove_2p20_ms"], 1), "x_4_in_flight": round(tot_cpu / ms4, 1),
                        "cpu_constraints_per_s": round((m + 1) / (tot_cpu * 1e-3), 1), "gpu_constraints_per_s": res["prove_constraints_per_s"],
                        "cores": max(win_threads(V), thr_wm), "bit_exact_vs_gpu": same_all, "spans_cpu_ms": spans_cpu,
                        "sample": "one 2^%d-constraint LegoGroth16 proof: witness map on %d threads + the five MSMs sequentially, each with one thread per window (<= %d busy cores), same Groth16-like witness as the GPU leg; O(1) finish not included on the CPU side" % (log2n, thr_wm, win_threads(V))}
    # -- the DROP-IN: exactly what the reference executes per proof once rust/patches are applied (feature "gpu") — nothing resident that the host put there,
    #    no handle in the host's hands: witness_map_from_matrices (r1cs_to_qap.rs:150-210, patch 0009) with the matrices and the assignment in host memory and
    #    h handed back as &[Fr]; then create_proof_and_committed_witnesses_with_assignment (prover.rs:267-383, patch 0005) with the key's queries as slices of
    #    ark-ec Affine structs in host memory, h and the assignment as &[Fr].  The library's resident-bases cache is what makes the second and later proofs fast.
    structs = {k_: ca.to_affine_structs(ca.G2 if k_ == "b2" else ca.G1, hostq[k_]) for k_ in ("a", "b1", "b2", "h", "l")}
    hpk = LG.HostProvingKey(vk, small1[2], small1[3], small1[4], structs["a"], structs["b1"], structs["b2"], structs["h"], structs["l"], a0=small1[5], b1_0=small1[6], b2_0=small2[3])
    mats_host = ((a_rp, a_cl, a_vl), (a_rp, b_cl, a_vl), (c_rp, c_cl, np.repeat(one, 2 * m + 1, 0)))
    z_inst, z_wit = np.ascontiguousarray(z[:2]), np.ascontiguousarray(z[2:])
    ca.bases_cache_clear()
    di = {}

    def wm_host(circuit_cached):
        """h as &[Fr] in host memory: the matrices cross PCIe with every call (what the patch does for a constraint system synthesised per proof), or
        the circuit resident (the Rust glue keys it by a content hash it computes while flattening `ConstraintMatrices`: rust/dock_gpu/src/lib.rs)"""
        if circuit_cached:
            return circ.witness_map(z, h_montgomery=True)[0]
        return qap.witness_map(*mats_host, z, 2, m + 1, h_montgomery=True)[0]

    def drop_in_proof(circuit_cached):
        h_fr = wm_host(circuit_cached)
        return LG.prove_host(hpk, 123456789, 987654321, 555, h_fr, z_inst, z_wit, h_montgomery=True)
    t0 = time.perf_counter(); p_cold = drop_in_proof(False); di["proof_1_cold_ms"] = round((time.perf_counter() - t0) * 1e3, 2)        # every query uploaded for the call
    t0 = time.perf_counter(); p_fill = drop_in_proof(False); di["proof_2_fill_ms"] = round((time.perf_counter() - t0) * 1e3, 2)        # the cache makes the five queries resident tables
    assert all((p_cold[k_] == p0[k_]).all() and (p_fill[k_] == p0[k_]).all() for k_ in p0)
    di["proof_warm_ms"] = round(timed(lambda: drop_in_proof(False), 5, warm=3), 2)
    di["proof_warm_circuit_cached_ms"] = round(timed(lambda: drop_in_proof(True), 5, warm=3), 2)
    assert all((drop_in_proof(True)[k_] == p0[k_]).all() for k_ in p0)
    # create_proof_with_reduction as patched (prover.rs:153-180): the synthesised constraint system's matrices resolve to a resident circuit (content hash in the
    # Rust glue), the witness map runs inside the prover call, h never leaves the device, z crosses PCIe once
    p_red = LG.prove_host(hpk, 123456789, 987654321, 555, None, z_inst, z_wit, circuit=circ)
    assert all((p_red[k_] == p0[k_]).all() for k_ in p0)
    di["proof_warm_with_reduction_ms"] = round(timed(lambda: LG.prove_host(hpk, 123456789, 987654321, 555, None, z_inst, z_wit, circuit=circ), 6, warm=3), 2)
    di["witness_map_host_matrices_ms"] = round(timed(lambda: wm_host(False), 3, warm=1), 2)
    di["witness_map_circuit_cached_ms"] = round(timed(lambda: wm_host(True), 3, warm=1), 2)
    h_fr = wm_host(True)
    di["prove_host_warm_ms"] = round(timed(lambda: LG.prove_host(hpk, 123456789, 987654321, 555, h_fr, z_inst, z_wit, h_montgomery=True), 5, warm=2), 2)
    # the five MSMs as the per-call-site patches of round 5 issue them (one after the other, each its own dgpu_msm_*_strided call): cache on, then off
    h_can = circ.witness_map(z)[0]
    n_aux_d, aux_at_d = (m + 1) - cw, 1 + 1 + cw

    def five_msms():
        return (ca.msm_strided(ca.G1, structs["h"], h_can[:n - 1]), ca.msm_strided(ca.G1, structs["l"][:n_aux_d], z[aux_at_d:aux_at_d + n_aux_d]),
                ca.msm_strided(ca.G1, structs["a"][1:], z[1:]), ca.msm_strided(ca.G1, structs["b1"][1:], z[1:]), ca.msm_strided(ca.G2, structs["b2"][1:], z[1:]))
    five_msms(); five_msms()
    di["five_sequential_strided_msms_warm_ms"] = round(timed(five_msms, 3, warm=1), 2)
    # the stale-key check: every figure above ran under the library's default, the EXACT mode (DGPU_CACHE_VERIFY_FULL: every record of the call's range re-fingerprinted on the
    # library's host threads beside the work on the resident copy); the sampled mode a host may select when its key cannot change under the library's feet, timed beside it
    st_h = structs["h"]
    di["strided_msm_warm_full_check_ms"] = round(timed(lambda: ca.msm_strided(ca.G1, st_h, h_can[:n - 1]), 5, warm=2), 3)
    ca.bases_cache(verify=24)
    di["strided_msm_warm_sampled_check_ms"] = round(timed(lambda: ca.msm_strided(ca.G1, st_h, h_can[:n - 1]), 5, warm=2), 3)
    di["proof_warm_with_reduction_sampled_check_ms"] = round(timed(lambda: LG.prove_host(hpk, 123456789, 987654321, 555, None, z_inst, z_wit, circuit=circ), 6, warm=2), 2)
    ca.bases_cache(verify=ca.CACHE_VERIFY_FULL)
    st_c = ca.bases_cache_stats()
    di["cache"] = {k_: st_c[k_] for k_ in ("hits", "misses", "fills", "stale", "evictions", "entries")}; di["cache"]["resident_GB"] = round(st_c["bytes"] / 1e9, 2)
    ca.bases_cache(bytes=0)
    di["five_sequential_strided_msms_cache_off_ms"] = round(timed(five_msms, 2, warm=1), 2)
    di["proof_cache_off_ms"] = round(timed(lambda: drop_in_proof(False), 2, warm=1), 2)
    ca.bases_cache(bytes=(1 << 64) - 1)
    di["vs_prove_2p20_ms"] = {"warm_with_reduction": round(di["proof_warm_with_reduction_ms"] / res["prove_2p20_ms"], 2), "warm": round(di["proof_warm_ms"] / res["prove_2p20_ms"], 2),
                              "warm_circuit_cached": round(di["proof_warm_circuit_cached_ms"] / res["prove_2p20_ms"], 2)}
    if cpu_legs and "prove" in cpu:
        di["cpu_ms"] = cpu["prove"]["cpu_ms"]; di["x_vs_cpu_warm"] = round(cpu["prove"]["cpu_ms"] / di["proof_warm_ms"], 1)
    di["note"] = ("one 2^%d-constraint proof through dgpu_witness_map[_r1cs] (h back to the host as &[Fr]) + dgpu_legogroth16_prove_host (queries = host slices of Affine structs, resolved by the "
                  "resident-bases cache; z and h cross PCIe per proof); bit-identical to prove_2p20_ms's proof; Rust-side costs that cannot be measured here (flattening ConstraintMatrices to CSR) are not included" % log2n)
    res["drop_in"] = di
    del structs, hpk
    hostq.clear()
    # the reference's timer spans (prover.rs:284-369, :578), each stage alone and in the reference's order (one call in flight)
    from crypto_amd import sharded as SH
    dz = ca.DeviceScalars(z)
    spans = {}

    def span(name, fn, k=3):
        spans[name] = round(timed(fn, k), 3)
    keep = {}

    def wm_keep():
        if "dh" in keep:
            keep["dh"].free()
        _, keep["dh"] = circ.witness_map(dz, to_host=False, resident=True)
    span("R1CS to QAP witness map", wm_keep)
    n_aux, aux_at = (m + 1) - cw, 1 + 1 + cw
    span("Compute C (h_query and l_query MSMs)", lambda: (qh.msm_resident(keep["dh"], n=min(qh.n, keep["dh"].n)), ql.msm_resident(dz, n=min(ql.n, n_aux), scalar_offset=aux_at)))
    span("Compute A", lambda: LG._calculate_coeff(ca.G1, pk.delta_g1, 123456789, qa, pk.a0, vk.alpha_g1, dz, 1))
    span("Compute B in G1", lambda: LG._calculate_coeff(ca.G1, pk.delta_g1, 987654321, qb1, pk.b1_0, pk.beta_g1, dz, 1))
    span("Compute B in G2", lambda: LG._calculate_coeff(ca.G2, vk.delta_g2, 987654321, qb2, pk.b2_0, vk.beta_g2, dz, 1))
    span("Finish C", lambda: SH.fold(ca.G1, np.stack([LG.lincomb(ca.G1, [p0["a"], p0["a"], pk.delta_g1, pk.eta_delta_inv_g1], [3, 5, 7, 11]), np.concatenate([p0["c"], pairing.FP_ONE_MONT]), np.concatenate([p0["d"], pairing.FP_ONE_MONT])])))
    span("Compute D", lambda: LG.lincomb(ca.G1, [small1[8], small1[9], small1[1]], [3, 5, 7]))
    keep["dh"].free(); dz.free()
    spans["sum_of_spans"] = round(sum(spans.values()), 3)
    res["prove_spans_ms"] = spans
    res["prove_note"] = ("LegoGroth16 create_proof (witness map + 4 G1 MSMs + 1 G2 MSM + finish), m + 1 = %d constraints, D = 2^%d, Groth16-like witness, "
                         "circuit and key (precomputed tables: window width %d for the a / b / l queries, automatic = 20 for the h query) resident, assignment uploaded per proof; the A / B-in-G1 / B-in-G2 / l MSMs share one partition sort" % (m + 1, log2n, ca.TABLE_C_WITNESS))
    res["note"] = "n = D = 2^%d; one call in flight unless stated; host-visible wall time per call" % log2n
    if cpu_legs:
        cpu["note"] = ("CPU path = oracle/ (arkworks-shaped C: the same window rule, signed digits, chunks-of-4 Miller loop, final-exponentiation chain; kind 'port', %d logical CPUs on this box); "
                       "x = cpu_ms / gpu_ms for one call in flight.  See cpu_baseline.calibration for how this C compares with ark-ff's assembly backend." % ncpu)
        res["cpu"] = cpu
    return res


if __name__ == "__main__":
    main()
