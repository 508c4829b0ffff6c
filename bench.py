"""bench.py — BLS12-381 G1 variable-base MSM throughput on MI355X (BASELINE.json metric).

One "step" = one n-term MSM over operands that are already resident in HBM (prepared bases + canonical
scalars), through the C ABI (dgpu_msm_g1_resident).  At N = 1 the workload is BASELINE.json configs[1]:
n = 2^20 random scalars / points.  At N > 1 (one process per GPU, launched by torch.distributed.run) the N
ranks jointly compute ONE MSM of N * 2^log2n terms per step by point-chunk sharding: each rank runs the full
pipeline on its own chunk, then an RCCL all_gather of the 144-byte partial points and a local fold
(crypto_amd/sharded.py).  Per-GPU work is fixed => "scaling": "weak".  value = (terms per step / 2^20) /
seconds per step, i.e. n=2^20-MSM equivalents per second for the whole job.

Extra objects on the JSON line: "roofline" (dominant kernel = k_accumulate, algorithmic bytes = 128 B/term,
duration from HIP events on the library's stream), "cpu_baseline" (the CPU oracle = arkworks-style Pippenger,
one thread per window like rayon, timed on this box's host cores; kind "port"), "secondary" (G2 MSM, 1024-pair
Miller loop, final exponentiation, witness map — N = 1 only, outside the timed region).

Inputs and the closed-form check are produced by the library itself (fixed-base kernel, published generator
encodings); oracle/ is imported only inside the cpu_baseline leg, where it is the timed CPU baseline and the
cross-check of the GPU result.
"""
import argparse
import json
import os
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


def limbs_to_ints(a):
    return [int(w0) | (int(w1) << 64) | (int(w2) << 128) | (int(w3) << 192) for w0, w1, w2, w3 in a.tolist()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=20, help="terms per GPU = 2^log2n (BASELINE: 20 at 1 GPU, 21 per GPU for 2^24 on 8)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the G2 / Miller-loop / witness-map timings appended to the JSON line at N = 1")
    ap.add_argument("--inflight", type=int, default=4, help="MSM calls in flight per GPU (host threads; each call owns a stream + workspace slot)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DGPU_BENCH_SAME_DEVICE"):      # plumbing test only: several ranks on one GPU
        local = 0
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if os.environ.get("DGPU_BENCH_BACKEND", "nccl") == "nccl" else None     # where collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DGPU_BENCH_BACKEND", "nccl")       # "gloo" only for the one-GPU plumbing test (RCCL refuses two ranks on one device)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import crypto_amd as ca
    from crypto_amd import sharded, serde, fixed_base as FB

    ca.init(local)
    n = 1 << args.log2n
    ncpu = os.cpu_count() or 1
    # Synthetic inputs with known discrete logs: P_i = (k0 + (off + i) d) G, scalars uniform in [0, r).  Everything here comes from the
    # library itself (the bases from its fixed-base kernel); the CPU oracle is only touched by the cpu_baseline leg at the end.
    K0 = int.from_bytes(np.random.Generator(np.random.PCG64(0x5EED0002)).bytes(40), "little") % R_MOD
    D = int.from_bytes(np.random.Generator(np.random.PCG64(0x5EED0003)).bytes(40), "little") % R_MOD
    off = rank * n
    gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(G1_GEN_COMPRESSED))
    kints, k = [], (K0 + off * D) % R_MOD
    for _ in range(n):
        kints.append(k)
        k += D
        if k >= R_MOD:
            k -= R_MOD
    with FB.WindowTable(ca.G1, gen1[0]) as gtab:
        bases, binf = gtab.multiply_many(kints)
    assert not binf.any()
    scalars = seeded_scalars(0x5EED1000 + rank, n)
    db = ca.DeviceBases(ca.G1, bases)
    ds = ca.DeviceScalars(scalars)

    def step():
        part = db.msm_resident(ds)
        return sharded.gather_and_fold(ca.G1, part, cdev) if world > 1 else part

    # correctness of what is timed: closed form (sum s_i k_i) G over ALL ranks' terms; the expected point comes from the fixed-base path
    # (a different kernel family), the comparison against the CPU oracle is part of the cpu_baseline leg
    res = step()
    loc = sum(sv * kv for sv, kv in zip(limbs_to_ints(scalars), kints)) % R_MOD
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
        """k steps with up to `inflight` local MSMs in flight; partial points are gathered/folded in step order"""
        futs = [pool.submit(db.msm_resident, ds) for _ in range(k)]
        last = None
        for f in futs:
            part = f.result()
            last = sharded.gather_and_fold(ca.G1, part, cdev) if world > 1 else part
        return last

    for _ in range(args.warmup):
        step()
    run_steps(min(args.steps, 2 * inflight))          # warm every slot's workspace (untimed)
    # sequential pass (one call in flight): per-stage HIP-event times without overlap, and the single-call latency
    ca.prof.enable(True)
    ca.prof.reset()
    tl = time.perf_counter()
    for _ in range(3):
        step()
    latency_ms = (time.perf_counter() - tl) / 3 * 1e3
    stages_seq = ca.prof.read()
    ca.prof.reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    assert (last == res).all(), "result changed between runs"
    stages = ca.prof.read()
    ca.prof.enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev if cdev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

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
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic_accumulate.json")
        if os.path.exists(tj):
            try:
                tr = json.load(open(tj))
                if tr.get("log2n") == args.log2n:
                    traffic = tr.get("hbm_bytes_per_launch")
            except Exception:
                pass
        # integer-throughput view of the same kernel (the path is VALU-bound, SURVEY.md 8d honesty note)
        mads_per_term_window = 6 * 392 + 588 + 2 * 301     # mixed add: 6 products, one fused two-product reduction, 2 squares
        out = {
            "metric": "BLS12-381 G1 MSM/s at n=2^20 (1 GPU) and n=2^24 (8 GPU); bit-exact vs CPU",
            "value": round(value, 3), "unit": "MSM/s (n=2^20-term equivalents, whole job)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (29-bit limbs, 381-bit modular integer)",
            "data": "synthetic (seeded PCG64 scalars uniform in [0, r); bases (k0 + i d) G with known discrete logs from the fixed-base kernel)",
            "config": {"workload": "BLS12-381 G1 variable-base MSM, n=2^%d terms per GPU, operands resident in HBM, %s" % (
                args.log2n, "1xMI355X" if world == 1 else "%dxMI355X point-chunk sharded, RCCL all_gather of partial points" % world),
                "terms_per_step": terms, "bit_exact_vs_closed_form": bit_exact, "parallelism": "1 process per GPU, %d ranks, %d calls in flight per GPU" % (world, inflight)},
            "terms_per_s": round(terms / (dt / args.steps), 1),
            "inflight": inflight, "latency_ms_one_in_flight": round(latency_ms, 4),
            "stages_ms": {k.replace("msm.", ""): round(v[0] / max(1, v[1]), 4) for k, v in stages.items()},
            "stages_ms_one_in_flight": {k.replace("msm.", ""): round(v[0] / max(1, v[1]), 4) for k, v in stages_seq.items()},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 6),
                         "traffic": traffic, "kernel": "k_accumulate<G1>", "avg_ms": round(acc_avg_ms, 4), "avg_ms_overlapped": round(acc_ov_ms, 4),
                         "note": "algorithmic 128 B/term x 2^log2n terms per launch; avg_ms = HIP-event duration with one call in flight (same process, "
                                 "untimed pass; rocprof of `bench.py --inflight 1` agrees), avg_ms_overlapped = inside the timed region where launches "
                                 "share the chip; the kernel is integer-multiply bound: see valu_roofline"},
        }
        if acc_avg_ms > 0 and "msm.accumulate" in stages:
            W = 16 if args.log2n >= 17 else None
            if W:
                mads = float(n) * W * mads_per_term_window
                out["valu_roofline"] = {"bound": "v_mad_u64_u32", "achieved": round(mads / (acc_avg_ms * 1e-3) / 1e12, 3),
                                        "peak": 31.8, "unit": "Tmad/s", "frac": round(mads / (acc_avg_ms * 1e-3) / 1e12 / 31.8, 4),
                                        "note": "peak = measured v_mad_u64_u32 issue rate, profiles/r01_instr_rate_ubench.txt"}
        if not args.no_cpu_baseline:
            # arkworks-style Pippenger, one task per window (17 windows at n=2^20 => at most 17 busy threads)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_c as O             # test infrastructure, used here only as the timed CPU baseline and its cross-check
            log2s = min(args.log2n, 20)
            ns = 1 << log2s
            c = O.window_c(ns)
            nw = (255 + c - 1) // c
            thr = max(1, min(ncpu, nw))
            tb = time.perf_counter()
            ref = O.G1.msm(bases[:ns], scalars[:ns], threads=thr)
            tcpu = time.perf_counter() - tb
            chk = db.msm_resident(ds, n=ns)
            same = bool((O.G1.to_affine(ref)[0] == O.G1.to_affine(chk)[0]).all())
            out["cpu_baseline"] = {"value": round((ns / float(1 << 20)) / tcpu, 4), "unit": "MSM/s (n=2^20-term equivalents)", "cores": thr,
                                   "kind": "port", "sample": "one n=2^%d G1 MSM, %.2f s wall on %d threads of %d logical CPUs; result bit-exact vs GPU: %s" % (
                                       log2s, tcpu, thr, ncpu, same)}
        if world == 1 and not args.no_secondary:
            try:
                out["secondary"] = secondary_configs(args.log2n)
            except Exception as e:                      # never let the secondary numbers take the headline line down
                out["secondary"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def secondary_configs(log2n):
    """BASELINE configs 3 and 4 next to the headline, outside the timed region (N = 1 only, a few seconds): G2 MSM at the same n,
    the 1024-pair Miller loop + final exponentiation, the R1CS->QAP witness map at D = 2^log2n.  Inputs are synthetic: G2 bases and the
    pairing inputs are device fixed-base products of seeded scalars (parity of these paths is the GPU test-suite's job, not this one's)."""
    import numpy as np
    import crypto_amd as ca
    from crypto_amd import fixed_base as FB, qap, serde
    gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(G1_GEN_COMPRESSED))
    gen2, _ = serde.deserialize(ca.G2, bytes.fromhex(G2_GEN_COMPRESSED))

    def timed(fn, k=5):
        fn()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        return (time.perf_counter() - t0) / k * 1e3

    n = 1 << log2n
    res = {}
    with FB.WindowTable(ca.G2, gen2[0]) as t2, FB.WindowTable(ca.G1, gen1[0]) as t1:
        db2 = t2.multiply_many_to_bases(seeded_scalars(0x5EED0003, n))
        ds = ca.DeviceScalars(seeded_scalars(0x5EED0004, n))
        res["g2_msm_ms"] = round(timed(lambda: db2.msm_resident(ds), 3), 3)
        res["g2_msm_per_s"] = round(1e3 / res["g2_msm_ms"], 2)
        db2.free(); ds.free()
        P, _ = t1.multiply_many(seeded_scalars(0x5EED0005, 1024)); Q, _ = t2.multiply_many(seeded_scalars(0x5EED0006, 1024))
    f = ca.multi_miller_loop(P, Q)
    res["miller_loop_1024_pairs_ms"] = round(timed(lambda: ca.multi_miller_loop(P, Q)), 3)
    res["miller_loop_pairs_per_s"] = round(1024 / res["miller_loop_1024_pairs_ms"] * 1e3, 0)
    res["final_exponentiation_ms"] = round(timed(lambda: ca.final_exponentiation(f)), 3)
    # witness map on the x_i = x_{i-1}^2 + i circuit shape (m + 1 constraints + 2 instance variables = D), circuit resident
    m = n - 3
    idx = np.arange(m, dtype=np.uint32)
    one = np.zeros((1, 4), np.uint64); one[0, 0] = 1
    a_rp = np.arange(m + 2, dtype=np.uint64); a_cl = np.concatenate([2 + idx, [2 + m]]).astype(np.uint32); a_vl = np.repeat(one, m + 1, 0)
    b_cl = np.concatenate([2 + idx, [0]]).astype(np.uint32)
    c_rp = np.concatenate([2 * np.arange(m + 1, dtype=np.uint64), [2 * m + 1]]).astype(np.uint64)
    c_cl = np.concatenate([np.stack([3 + idx, np.zeros(m, np.uint32)], 1).reshape(-1), [1]]).astype(np.uint32)
    circ = qap.DeviceR1cs((a_rp, a_cl, a_vl), (a_rp, b_cl, a_vl), (c_rp, c_cl, np.repeat(one, 2 * m + 1, 0)), m + 3, 2, m + 1)
    z = seeded_scalars(0x5EED0007, m + 3)

    def wm():
        _, dh = circ.witness_map(z, to_host=False, resident=True)
        dh.free()
    res["witness_map_ms"] = round(timed(wm, 3), 3)
    res["note"] = "n = D = 2^%d; one call in flight; host-visible wall time per call" % log2n
    return res


if __name__ == "__main__":
    main()
