"""CPU: the C-ABI library loads and exports every symbol include/dock_gpu.h declares; the entry points fail
loudly (error codes, no CPU fallback) without a device; the host-side pieces that need no GPU (partial-point
fold used by the multi-GPU gather, the reference-interface mirror's length semantics) are checked."""
import ctypes as C
import os
import re
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import crypto_amd as ca
from crypto_amd._native import lib, dev_lib, SYMBOLS, DEV_SYMBOLS
from crypto_amd import sharded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAS_GPU = torch.cuda.is_available()


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "dock_gpu.h")).read()
    declared = sorted(set(re.findall(r"\b(dgpu_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    L = lib()
    for name in declared:
        assert hasattr(L, name), "libdock_gpu.so does not export %s" % name
    assert sorted(SYMBOLS) == declared


def test_development_surface_lives_in_the_twin_only():
    """include/dock_gpu_dev.h (tuning knobs, stage timers, self-test and fault-injection hooks) is served by libdock_gpu_dev.so, which also
    exports everything the product does; the product exports exactly its own header — nothing a Rust host would not bind"""
    import subprocess
    hdr = open(os.path.join(ROOT, "include", "dock_gpu_dev.h")).read()
    declared = sorted(set(re.findall(r"\b(dgpu_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(DEV_SYMBOLS)
    L, T = lib(), dev_lib()
    for name in declared:
        assert hasattr(T, name), "libdock_gpu_dev.so does not export %s" % name
        assert not hasattr(L, name), "the product library exports the development symbol %s" % name
    for name in SYMBOLS:
        assert hasattr(T, name)
    exported = lambda so: sorted(l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "crypto_amd", so)], text=True).splitlines() if " T dgpu_" in l)
    assert exported("libdock_gpu.so") == sorted(SYMBOLS)
    assert exported("libdock_gpu_dev.so") == sorted(SYMBOLS + DEV_SYMBOLS)


def test_error_strings():
    L = lib()
    for code in range(0, -8, -1):
        assert L.dgpu_strerror(code)
    assert b"unknown" in L.dgpu_strerror(-99)
    T = dev_lib()
    assert T.dgpu_set_window_bits(3) == -3 and T.dgpu_set_window_bits(0) == 0


@pytest.mark.skipif(HAS_GPU, reason="checks the no-device behaviour")
def test_fails_loudly_without_device():
    L = lib()
    assert L.dgpu_device_count() == 0
    assert L.dgpu_init(0) == -1
    out = np.zeros(18, np.uint64)
    b = O.G1.generator().reshape(1, 12)
    s = np.ones((1, 4), np.uint64)
    rc = L.dgpu_msm_g1(b.ctypes.data_as(C.c_void_p), None, s.ctypes.data_as(C.c_void_p), 1, out.ctypes.data_as(C.c_void_p))
    assert rc == -1                      # DGPU_E_NODEVICE — never a silent CPU result
    with pytest.raises(ca.DockGpuError):
        ca.msm_bigint(ca.G1, b, s)
    # every device-backed entry point refuses the same way: pairing, witness map, fixed base, folding step, uploads
    p_ = lambda a: a.ctypes.data_as(C.c_void_p)
    f12 = np.zeros(72, np.uint64); q = O.G2.generator().reshape(1, 24); o12 = np.zeros(12, np.uint64); oi = np.zeros(1, np.uint8)
    h = C.c_uint64(0)
    assert L.dgpu_multi_miller_loop(p_(b), p_(q), None, 1, p_(f12)) == -1
    co = np.zeros(68 * 36, np.uint64)
    assert L.dgpu_g2_prepare(p_(q), None, 1, p_(co), p_(oi)) == -1
    assert L.dgpu_multi_miller_loop_prepared(p_(b), p_(co), None, 1, p_(f12)) == -1
    assert L.dgpu_fixed_base_g1(p_(b), p_(s), 1, 0, p_(o12), p_(oi)) == -1
    assert L.dgpu_window_table_g1(p_(b), C.byref(h)) == -1
    assert L.dgpu_g1_mul_add_batch(p_(b), None, p_(s), 4, None, None, 1, p_(o12), p_(oi)) == -1
    assert L.dgpu_g1_scale_batch(p_(b), None, p_(s), 4, None, 1, p_(o12), p_(oi)) == -1
    assert L.dgpu_bases_upload_g1(p_(b), None, 1, C.byref(h)) == -1
    assert L.dgpu_scalars_upload(p_(s), 1, 0, C.byref(h)) == -1
    # the in-library multi-GPU entry points and the table conversion refuse the same way
    assert L.dgpu_msm_g1_sharded(p_(b), None, p_(s), 1, 0, p_(out)) == -1
    assert L.dgpu_multi_miller_loop_sharded(p_(b), p_(q), None, 1, 0, p_(f12)) == -1
    two = np.array([1, 2], np.uint64); b2 = np.concatenate([b, b]); q2 = np.concatenate([q, q]); f24 = np.zeros(144, np.uint64)
    assert L.dgpu_multi_miller_loop_segments(p_(b2), p_(q2), None, 2, p_(two), 2, p_(f24)) == -1          # (two segments: the batched path)
    assert L.dgpu_multi_pairing_segments(p_(b2), p_(q2), None, 2, p_(two), 2, p_(f24)) == -1
    assert L.dgpu_bases_upload_g1_sharded(p_(b), None, 1, 0, C.byref(h)) == -1
    assert L.dgpu_context_count() == 0
    assert L.dgpu_set_device(0) == -3                  # no such context
    from crypto_amd import fixed_base, qap
    with pytest.raises(ca.DockGpuError):
        fixed_base.multiply_field_elems_with_same_group_elem(ca.G1, b[0], s)


def test_bad_arguments():
    L = lib()
    out = np.zeros(18, np.uint64)
    assert L.dgpu_msm_g1(None, None, None, 5, out.ctypes.data_as(C.c_void_p)) == -3
    assert L.dgpu_msm_g1(None, None, None, 0, None) == -3
    assert L.dgpu_bases_free(12345) == -3
    assert L.dgpu_r1cs_shape(12345, None, None, None) == -3
    # argument checks come before the device check on every entry point
    b = O.G1.generator().reshape(1, 12); sc = np.ones((1, 4), np.uint64); o = np.zeros(12, np.uint64); oi = np.zeros(1, np.uint8)
    p_ = lambda a: a.ctypes.data_as(C.c_void_p)
    assert L.dgpu_g1_mul_add_batch(p_(b), None, p_(sc), 3, None, None, 1, p_(o), p_(oi)) == -3       # scalar_stride must be 0 or 4
    assert L.dgpu_g1_mul_add_batch(None, None, p_(sc), 4, None, None, 1, p_(o), p_(oi)) == -3
    assert L.dgpu_g1_scale_batch(p_(b), None, p_(sc), 2, None, 1, p_(o), p_(oi)) == -3
    assert L.dgpu_fp12_multi_pow(None, None, 2, p_(np.zeros(72, np.uint64))) == -3
    assert L.dgpu_window_table_g1(None, C.byref(C.c_uint64(0))) == -3
    assert L.dgpu_multi_miller_loop(None, None, None, 3, p_(np.zeros(72, np.uint64))) == -3
    assert L.dgpu_multi_miller_loop_prepared(None, None, None, 3, p_(np.zeros(72, np.uint64))) == -3
    assert L.dgpu_g2_prepare(None, None, 2, None, None) == -3
    g2 = O.G2.generator().reshape(1, 24); seg = np.array([1, 1], np.uint64); o144 = np.zeros(144, np.uint64)
    assert L.dgpu_multi_miller_loop_segments(p_(b), p_(g2), None, 1, p_(seg), 0, p_(o144)) == -3             # no segments
    assert L.dgpu_multi_miller_loop_segments(p_(b), p_(g2), None, 1, None, 2, p_(o144)) == -3
    assert L.dgpu_multi_miller_loop_segments(p_(b), p_(g2), None, 1, p_(np.array([1, 0], np.uint64)), 2, p_(o144)) == -3    # ends not ascending
    assert L.dgpu_multi_pairing_segments(p_(b), p_(g2), None, 1, p_(np.array([0, 2], np.uint64)), 2, p_(o144)) == -3        # an end past n
    assert L.dgpu_multi_pairing_segments(None, None, None, 0, p_(np.array([0, 0], np.uint64)), 2, p_(o144)) == 0 and o144[0] != 0   # empty products: one, no device needed
    assert L.dgpu_window_table_free(999) == -3
    assert L.dgpu_bases_precompute_g1(424242, 0) == -3 and L.dgpu_bases_precompute_g2(424242, 20) == -3      # unknown handle
    assert L.dgpu_bases_precompute_g1(1, 15) == -3 and L.dgpu_bases_precompute_g1(1, 23) == -3                # width outside 16..22
    assert L.dgpu_init_device_list(None, 2) == -3 and L.dgpu_init_devices(0) == -3
    assert L.dgpu_msm_g1_sharded(None, None, None, 4, 0, p_(np.zeros(18, np.uint64))) == -3
    assert L.dgpu_msm_g1_sharded_handle(777, p_(sc), 1, 0, p_(np.zeros(18, np.uint64))) == -3
    assert L.dgpu_scalars_upload_sharded(p_(sc), 1, 0, 777, C.byref(C.c_uint64(0))) == -3
    assert L.dgpu_witness_map_r1cs_resident(5, 6, p_(np.zeros(8, np.uint64)), None, None) in (-1, -3)


def test_malformed_r1cs_is_refused_before_it_reaches_the_device():
    """rowptr monotonic, rowptr[0] == 0, rowptr[rows] == nnz, cols < num_vars — validated inside the native entry points"""
    L = lib()
    p_ = lambda a: a.ctypes.data_as(C.c_void_p)
    vals = np.ones((2, 4), np.uint64); z = np.ones((3, 4), np.uint64); h = C.c_uint64(0); out = np.zeros((4, 4), np.uint64); ol = C.c_size_t(0)
    good = (np.array([0, 1, 2], np.uint64), np.array([0, 2], np.uint32))
    bads = [(np.array([0, 2, 1], np.uint64), good[1]),        # decreasing
            (np.array([1, 1, 2], np.uint64), good[1]),        # does not start at 0
            (np.array([0, 1, 3], np.uint64), good[1]),        # rowptr[rows] != nnz
            (good[0], np.array([0, 3], np.uint32))]           # column >= num_vars
    def args(m):
        a = []
        for k in range(3):
            rp, cl = m if k == 1 else good
            a += [p_(rp), p_(cl), p_(vals), 2]
        return a
    for m in bads:
        assert L.dgpu_r1cs_upload(*args(m), 3, 1, 2, 0, C.byref(h)) == -3
        assert L.dgpu_witness_map(*args(m), p_(z), 3, 1, 2, 0, p_(out), None, C.byref(ol)) == -3
    assert L.dgpu_r1cs_upload(*args(good), 3, 1, 2, 0, C.byref(h)) == -1          # well-formed: only the missing device stops it


def test_checked_msm_and_pairs_length_semantics():
    # ark-ec msm(): Err(min_len) on mismatch;  utils/src/pairs.rs: Pairs::new -> None / TryFrom Err((l, r))
    b = np.zeros((5, 12), np.uint64)
    s = np.zeros((3, 4), np.uint64)
    assert ca.msm(ca.G1, b, s) == (False, 3)
    with pytest.raises(ValueError):
        ca.Pairs(ca.G1, b, s)
    # OwnedPairs (utils/src/owned_pairs.rs): same length rule, owns copies, extend / split / as_ref
    with pytest.raises(ValueError):
        ca.OwnedPairs(ca.G1, b, s)
    op = ca.OwnedPairs(ca.G1)
    assert op.is_empty() and len(op) == 0
    op.extend([(b[0], s[0]), (b[1], s[1])])
    l, r = op.split()
    assert len(op) == 2 and l.shape == (2, 12) and r.shape == (2, 4) and isinstance(op.as_ref(), ca.Pairs)


@pytest.mark.parametrize("curve,G", [(ca.G1, O.G1), (ca.G2, O.G2)])
def test_fold_partials_on_host(curve, G):
    gen = G.generator()
    ks = [5, 7, 11, 0, 13]
    parts = []
    for k in ks:
        j = G.mul(gen, O.int_to_limbs(k, 4))
        a, inf = G.to_affine(j)
        one = O.fp_to_mont(np.array([[1, 0, 0, 0, 0, 0]], np.uint64)).reshape(-1)
        z = np.zeros(G.AW // 2, np.uint64)
        if not inf:
            z[:6] = one
        parts.append(np.concatenate([a, z]) if not inf else np.concatenate([np.zeros(G.AW, np.uint64), z]))
    parts.append(G.mul(gen, O.int_to_limbs(17, 4)))      # a non-normalised Jacobian triple is accepted too
    got = sharded.fold(curve, np.stack(parts))
    exp = G.mul(gen, O.int_to_limbs(sum(ks) + 17, 4))
    assert U.jac_to_model(G, got) == U.jac_to_model(G, exp)
    assert (got[-(G.AW // 2):][:6] == O.fp_to_mont(np.array([[1, 0, 0, 0, 0, 0]], np.uint64)).reshape(-1)).all()  # Z == one
    # identity + identity == identity (Z = 0)
    zero = np.zeros((2, curve.JW), np.uint64)
    assert not sharded.fold(curve, zero)[-(G.AW // 2):].any()
    # P + (-P)
    a, _ = G.to_affine(parts[0])
    neg = a.copy()
    h = G.AW // 2
    for k in range(h // 6):
        neg[h + 6 * k:h + 6 * k + 6] = U.fp_abi((-U.fp_int(a[h + 6 * k:h + 6 * k + 6])) % U.P)
    z = np.zeros(h, np.uint64); z[:6] = O.fp_to_mont(np.array([[1, 0, 0, 0, 0, 0]], np.uint64)).reshape(-1)
    both = np.stack([np.concatenate([a, z]), np.concatenate([neg, z])])
    assert not sharded.fold(curve, both)[-h:].any()


@pytest.mark.parametrize("curve,G", [(ca.G1, O.G1), (ca.G2, O.G2)])
def test_lincomb_on_host_matches_the_oracle(curve, G):
    """dgpu_lincomb_*: the O(1) scalar multiplications around the MSMs (prover.rs:309-313,350-355 use `mul_bigint` on the CPU) — host
    arithmetic inside the library, no device: sum s_i P_i == the oracle's MSM, for 0 .. 16 terms, zero scalars, identity points, repeated
    and opposite points, scalars r - 1 and 2^255 - 1 (not reduced); more than DGPU_MAX_LINCOMB terms are refused."""
    from crypto_amd import legogroth16 as LG
    from crypto_amd.aggregation.ops import neg
    k0 = O.rand_scalars(5, 1)[0]; d = O.rand_scalars(6, 1)[0]
    pts = G.gen_seq(k0, d, 16, threads=4)
    sc = O.rand_scalars(7, 16)
    sc[1] = 0
    sc[2] = O.int_to_limbs(U.R - 1, 4)
    sc[3] = np.array([0xFFFFFFFFFFFFFFFF] * 3 + [0x7FFFFFFFFFFFFFFF], np.uint64)
    sc[4] = O.int_to_limbs(1, 4)
    pts[6] = pts[5]; pts[8] = neg(curve, pts[7]); sc[8] = sc[7]
    inf = np.zeros(16, np.uint8); inf[9] = 1
    for k in (0, 1, 2, 3, 4, 9, 10, 16):
        p = pts[:k].copy(); p[inf[:k] == 1] = 0
        got = LG.lincomb(curve, list(p), [O.limbs_to_int(x) for x in sc[:k]]) if k else None
        if k == 0:
            out = np.zeros(curve.JW, np.uint64)
            assert curve.fn("dgpu_lincomb_%s")(None, None, None, 0, out.ctypes.data_as(C.c_void_p)) == 0
            got = out
        exp = G.msm(pts[:k], sc[:k], inf[:k], threads=1) if k else G.msm(pts[:0], sc[:0], None, threads=1)
        assert U.jac_to_model(G, got) == U.jac_to_model(G, exp), k
    out = np.zeros(curve.JW, np.uint64)
    big_p = np.concatenate([pts, pts[:1]]); big_s = np.concatenate([sc, sc[:1]])
    rc = curve.fn("dgpu_lincomb_%s")(big_p.ctypes.data_as(C.c_void_p), None, big_s.ctypes.data_as(C.c_void_p), 17, out.ctypes.data_as(C.c_void_p))
    assert rc == ca.DGPU_E_BADARG if hasattr(ca, "DGPU_E_BADARG") else rc != 0


def test_chunk_bounds_partition():
    for n in (0, 1, 7, 8, 1 << 20, (1 << 24) + 3):
        for world in (1, 2, 3, 8):
            spans = [sharded.chunk_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_cpp_mirror_header_compiles():
    """include/dock_gpu.hpp (the compiled-host mirror of the reference's interface) and its parity driver are valid C++17"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "native", "cpp_api_driver.cpp")])


def test_glv_decomposition_on_host():
    """k mod r = k1 + k2 lambda with k1, k2 < 2^128 (what the G1 scaling kernel is fed with) against Python's divmod"""
    import random
    L = dev_lib()                        # (a self-test hook: include/dock_gpu_dev.h)
    lam = 0xac45a4010001a40200000000ffffffff
    assert (lam * lam + lam + 1) % U.R == 0
    random.seed(5)
    cases = [0, 1, lam - 1, lam, lam + 1, U.R - 1, U.R, U.R + 1, 2 * U.R, (1 << 256) - 1] + [random.randrange(1 << 256) for _ in range(3000)]
    for k in cases:
        a = O.int_to_limbs(k, 4); k1 = np.zeros(2, np.uint64); k2 = np.zeros(2, np.uint64)
        assert L.dgpu_selftest_glv_decompose(a.ctypes.data_as(C.c_void_p), k1.ctypes.data_as(C.c_void_p), k2.ctypes.data_as(C.c_void_p)) == 0
        q, rem = divmod(k % U.R, lam)
        assert O.limbs_to_int(k1) == rem and O.limbs_to_int(k2) == q, hex(k)


def test_prepared_batches_do_not_pose_as_arrays():
    """A G2Prepared batch indexes to a G2Prepared batch; numpy used to walk such an object as an endless sequence of sequences when a mixed
    operand list reached np.asarray (a batch verifier that passed [b, delta_pc, gamma_pc] per proof to the pairing checker never returned
    and ate the host's memory).  Now: TypeError at once, IndexError past the end."""
    from crypto_amd import pairing
    g = pairing.G2Prepared(np.zeros((2, pairing.PREPARED_WORDS), np.uint64), np.zeros(2, np.uint8))
    with pytest.raises(TypeError):
        np.asarray([np.zeros((1, 24), np.uint64), g])
    with pytest.raises(TypeError):
        np.ascontiguousarray([np.zeros((1, 24), np.uint64), g, g], dtype=np.uint64)
    with pytest.raises(IndexError):
        g[2]
    assert len(g[1]) == 1 and len(g[-1]) == 1 and len(g[0:2]) == 2
