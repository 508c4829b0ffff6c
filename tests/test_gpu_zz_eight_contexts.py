"""GPU (-m gpu): BASELINE config 5's shape inside ONE process with EIGHT device contexts — the box has one GPU, so the device is listed eight times
(dgpu_init_device_list): every code path of the 8-GPU in-process form (one host thread per context inside the call, per-context tables and
workspaces, peer copies between contexts, partial points folded on the host) except seven more physical devices.  2^24 terms in total, 2^21 per
context on precomputed tables, closed form over known discrete logs; and the LegoGroth16 prover with its key sharded eight ways.
(File name: the library's contexts only ever grow, and tests/test_gpu_multi_context.py asserts it finds two — so this module sorts last.)
The measured 1 -> 8 GPU curve is the driver's (SCALE_rNN.json); this file is about correctness of the sharded path at its real shape."""
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import lego_setup as LS
import crypto_amd as ca
from crypto_amd import fixed_base as fb, legogroth16 as LG, qap
from crypto_amd._native import lib
import bench as B

pytestmark = pytest.mark.gpu
R = U.R


@pytest.fixture(scope="module", autouse=True)
def _contexts():
    assert torch.cuda.is_available()
    ca.init_devices([0] * 8)
    assert lib().dgpu_context_count() == 8
    yield
    lib().dgpu_set_device(0)         # (contexts are never taken away again: this module is named to run after every other GPU test)


def _to_int(a):
    return [int(w0) | (int(w1) << 64) | (int(w2) << 128) | (int(w3) << 192) for w0, w1, w2, w3 in a.tolist()]


def test_2_24_terms_over_eight_contexts_closed_form():
    n = 1 << 24
    ks = O.rand_scalars(8101, n); sc = O.rand_scalars(8102, n)
    with fb.WindowTable(ca.G1, O.G1.generator()) as t:
        bases, inf = t.multiply_many(ks)                                    # host array: the sharded upload cuts it into eight ranges
        tot_all = B.dot_mod_r(ks, sc)                                       # sum k_i s_i mod r (bench.py's exact limb arithmetic)
        exp_xy, exp_inf = t.multiply(tot_all)
        pre = sum(x * y for x, y in zip(_to_int(ks[: 1 << 12]), _to_int(sc[: 1 << 12]))) % R         # the same by Python big integers on a prefix
        assert B.dot_mod_r(ks[: 1 << 12], sc[: 1 << 12]) == pre
    sh = ca.ShardedDeviceBases(ca.G1, bases, inf, ngpus=8)
    cnt = lib().dgpu_shard_count
    import ctypes as C
    k = C.c_int32(0); assert cnt(sh.handle, C.byref(k)) == 0 and k.value == 8
    plain = sh.msm_bigint(sc)
    assert not exp_inf and (plain[:12] == exp_xy).all() and plain[12:].any()
    sh.precompute()                                                         # eight tables of 2^21 rows (13 windows of width 20 each)
    assert (sh.msm_bigint(sc) == plain).all()
    ds = sh.upload_scalars(sc)
    assert (sh.msm_resident(ds) == plain).all()
    # fewer scalars than bases (prover.rs:286): a prefix that ends inside the sixth shard
    m = 5 * (n // 8) + 12345
    part = sh.msm_bigint(sc[:m])
    one = ca.DeviceBases(ca.G1, bases[:m], inf[:m])
    assert (part == one.msm_bigint(sc[:m])).all()
    one.free(); ds.free(); sh.free()
    # config 5's shape from the UNMODIFIED call (dgpu_set_auto_shard_min_n): msm_bigint(&[G1Affine], ..) over the caller's structs shards itself over the eight
    # contexts, each of which makes ITS chunk of the key resident at the chunk's second sighting (eight tables of 2^21 rows) — cold, fill, warm: the same point
    st = ca.to_affine_structs(ca.G1, bases, inf)
    L = lib()
    ca.bases_cache_clear()
    assert L.dgpu_set_auto_shard_min_n(1 << 20) == 0
    try:
        s0 = ca.bases_cache_stats()
        for call in range(3):
            assert (ca.msm_strided(ca.G1, st, sc) == plain).all(), call
        s1 = ca.bases_cache_stats()
        assert s1["fills"] - s0["fills"] == 8 and s1["hits"] - s0["hits"] == 16 and s1["entries"] == 8, (s0, s1)
    finally:
        L.dgpu_set_auto_shard_min_n(0)
        ca.bases_cache_clear()


def test_prover_with_its_key_sharded_eight_ways():
    m, cw = 400, 3
    cs = LS.circuit(m, x0=7)
    key = LS.setup(cs, cw, seed=1700 + m)
    vk = LG.VerifyingKey(key["alpha_g1"], key["beta_g2"], key["gamma_g2"], key["delta_g2"], key["gamma_abc_g1"], key["eta_gamma_inv_g1"], cw)
    small = (vk, key["beta_g1"], key["delta_g1"], key["eta_delta_inv_g1"])
    pk1 = LG.ProvingKey(*small, key["a_query"], key["b_g1_query"], key["b_g2_query"], key["h_query"], key["l_query"])
    sh = lambda curve, q: ca.ShardedDeviceBases(curve, q, ngpus=8)
    pk8 = LG.ProvingKey.from_device(*small, key["a_query"][0].copy(), key["b_g1_query"][0].copy(), key["b_g2_query"][0].copy(),
                                    sh(ca.G1, key["a_query"]), sh(ca.G1, key["b_g1_query"]), sh(ca.G2, key["b_g2_query"]), sh(ca.G1, key["h_query"]), sh(ca.G1, key["l_query"]))
    z = LS.scalars(cs["z"]); n_inst = cs["n_inst"]
    dr = qap.DeviceR1cs(*[qap.csr(cs[k]) for k in "ABC"], len(cs["z"]), n_inst, len(cs["A"]))
    r, s, v = 0x1234567 * 0x9E3779B97F4A7C15 % R, 0xABCDEF01 * 0xBF58476D1CE4E5B9 % R, 0x55AA55 * 0x94D049BB133111EB % R
    pvk = LG.prepare_verifying_key(vk)
    ref = LG.prove_abi(pk1, r, s, v, z, n_inst, circuit=dr)
    for tables in (False, True):
        if tables:
            for q in (pk8.a_query, pk8.b_g1_query, pk8.b_g2_query, pk8.h_query, pk8.l_query):
                q.precompute(16)
        got = LG.prove_abi(pk8, r, s, v, z, n_inst, circuit=dr)
        assert all((got[k] == ref[k]).all() for k in ref), tables
        assert LG.verify_proof(pvk, got, z[1:n_inst])
    dr.free()
