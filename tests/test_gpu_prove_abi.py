"""GPU (-m gpu): dgpu_legogroth16_prove — legogroth16/src/prover.rs:267-383 (with the witness map of :153-180 in front) as ONE call of the
C ABI.  The schedule that used to live in Python above the ABI runs on host threads inside the library; the proof must equal, limb for limb,
what the Python mirror of the reference's function computes from the individual entry points (which tests/test_gpu_legogroth16.py checks
against the CPU oracle's evaluation of the same equations), for plain queries and for precomputed tables (shared sort), Montgomery and
canonical assignments, r = 0, no committed witnesses and all witnesses committed."""
import numpy as np
import pytest
import torch
import oracle_c as O
import lego_setup as LS
import crypto_amd as ca
from crypto_amd import legogroth16 as LG, qap

pytestmark = pytest.mark.gpu
R = LS.R


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available()
    ca.init(0)


@pytest.mark.parametrize("m,cw", [(20, 2), (117, 0), (117, 3), (300, 298)])
def test_one_call_equals_the_python_schedule(m, cw):
    cs = LS.circuit(m, x0=7)
    nw = len(cs["z"]) - cs["n_inst"]
    cw = min(cw, nw)
    key = LS.setup(cs, cw, seed=900 + m)
    vk = LG.VerifyingKey(key["alpha_g1"], key["beta_g2"], key["gamma_g2"], key["delta_g2"], key["gamma_abc_g1"], key["eta_gamma_inv_g1"], cw)
    pk = LG.ProvingKey(vk, key["beta_g1"], key["delta_g1"], key["eta_delta_inv_g1"], key["a_query"], key["b_g1_query"], key["b_g2_query"], key["h_query"], key["l_query"])
    z = LS.scalars(cs["z"])
    n_inst = cs["n_inst"]
    dr = qap.DeviceR1cs(*[qap.csr(cs[k]) for k in "ABC"], len(cs["z"]), n_inst, len(cs["A"]))
    r, s, v = 0x1234567 * 0x9E3779B97F4A7C15 % R, 0xABCDEF01 * 0xBF58476D1CE4E5B9 % R, 0x55AA55 * 0x94D049BB133111EB % R
    pvk = LG.prepare_verifying_key(vk)
    for tables in (False, True):
        if tables:
            for q in (pk.a_query, pk.b_g1_query, pk.b_g2_query, pk.h_query, pk.l_query):
                if q.n:
                    q.precompute(16)
        for (rr, ss) in ((r, s), (0, s), (r + R, s)):                      # r = 0: no B in G1 (prover.rs:330); r + R: reduced inside
            ref = LG.create_proof_with_reduction_py(pk, dr, rr % R, ss, v, z)
            got = LG.prove_abi(pk, rr, ss, v, z, n_inst, circuit=dr)
            assert all((got[k] == ref[k]).all() for k in ref), (m, cw, tables)
            assert LG.verify_proof(pvk, got, z[1:n_inst])
        # &[Fr] (Montgomery) assignment, and h handed over as a resident vector instead of the circuit
        gm = LG.prove_abi(pk, r, s, v, O.fr_to_mont(z), n_inst, circuit=dr, montgomery=True)
        ref = LG.create_proof_with_reduction_py(pk, dr, r, s, v, z)
        assert all((gm[k] == ref[k]).all() for k in ref)
        _, dh = dr.witness_map(z, to_host=False, resident=True)
        gh = LG.prove_abi(pk, r, s, v, z, n_inst, h=dh)
        assert all((gh[k] == ref[k]).all() for k in ref)
        dh.free()
    # the same proof for a host that holds the key the way the reference does (dgpu_legogroth16_prove_host): queries as arrays of Affine structs in host
    # memory, h as the host vector the witness map returned — first proof (views uploaded for the call), second (the cache makes them resident), later ones
    hpk = LG.HostProvingKey(vk, key["beta_g1"], key["delta_g1"], key["eta_delta_inv_g1"], key["a_query"], key["b_g1_query"], key["b_g2_query"], key["h_query"], key["l_query"])
    h_host, _ = dr.witness_map(z, to_host=True)
    ref = LG.create_proof_with_reduction_py(pk, dr, r, s, v, z)
    ca.bases_cache_clear(); ca.bases_cache(min_n=2)
    try:
        s0 = ca.bases_cache_stats()
        for call in range(4):
            gh = LG.prove_host(hpk, r, s, v, h_host, z[:n_inst], z[n_inst:])
            assert all((gh[k] == ref[k]).all() for k in ref), (m, cw, call)
        gm = LG.prove_host(hpk, r, s, v, O.fr_to_mont(h_host), O.fr_to_mont(z[:n_inst]), O.fr_to_mont(z[n_inst:]), montgomery=True, h_montgomery=True)
        assert all((gm[k] == ref[k]).all() for k in ref)
        gc = LG.prove_host(hpk, r, s, v, None, z[:n_inst], z[n_inst:], circuit=dr)          # the circuit resident instead of h: create_proof_with_reduction as one call
        assert all((gc[k] == ref[k]).all() for k in ref)
        with pytest.raises(ca.DockGpuError):
            LG.prove_host(hpk, r, s, v, h_host, z[:n_inst], z[n_inst:], circuit=dr)           # both sources of h
        with pytest.raises(ca.DockGpuError):
            LG.prove_host(hpk, r, s, v, None, z[:n_inst], z[n_inst:])                          # neither
        g0 = LG.prove_host(hpk, 0, s, v, h_host, z[:n_inst], z[n_inst:])
        r0 = LG.create_proof_with_reduction_py(pk, dr, 0, s, v, z)
        assert all((g0[k] == r0[k]).all() for k in r0)
        s1 = ca.bases_cache_stats()
        nq = sum(1 for q in (hpk.a_query, hpk.b_g1_query, hpk.b_g2_query, hpk.h_query, hpk.l_query) if len(q) >= 2)
        assert s1["fills"] - s0["fills"] == nq and s1["hits"] - s0["hits"] == 6 * nq, (s0, s1)
        # the exact stale-key mode (every record of every view re-fingerprinted BESIDE the proof): same proof; and an in-place edit of ONE record in the middle of a
        # resident query — which the sampled mode would most likely miss — gives the edited key's proof, not the resident copy's
        ca.bases_cache(verify=ca.CACHE_VERIFY_FULL)
        gx = LG.prove_host(hpk, r, s, v, h_host, z[:n_inst], z[n_inst:])
        assert all((gx[k] == ref[k]).all() for k in ref)
        if len(hpk.a_query) > 8:
            keep = hpk.a_query[5].copy()
            hpk.a_query[5] = hpk.a_query[6]
            st0 = ca.bases_cache_stats()["stale"]
            g_edit = LG.prove_host(hpk, r, s, v, h_host, z[:n_inst], z[n_inst:])
            assert ca.bases_cache_stats()["stale"] == st0 + 1
            ca.bases_cache(bytes=0)
            g_plain = LG.prove_host(hpk, r, s, v, h_host, z[:n_inst], z[n_inst:])
            ca.bases_cache(bytes=(1 << 64) - 1)
            assert all((g_edit[k] == g_plain[k]).all() for k in g_plain) and not all((g_edit[k] == ref[k]).all() for k in ref)
            hpk.a_query[5] = keep
        # the cache off: every view is uploaded for the call
        ca.bases_cache(bytes=0)
        gh = LG.prove_host(hpk, r, s, v, h_host, z[:n_inst], z[n_inst:])
        assert all((gh[k] == ref[k]).all() for k in ref)
    finally:
        ca.bases_cache(bytes=(1 << 64) - 1, min_n=1 << 16)
    # argument checks: both / neither source of h, n_inst out of range
    with pytest.raises(ca.DockGpuError):
        LG.prove_abi(pk, r, s, v, z, n_inst)
    with pytest.raises(ca.DockGpuError):
        LG.prove_abi(pk, r, s, v, z, len(z) + 1, circuit=dr)
    # an n_inst that disagrees with the resident circuit's own num_inputs (it decides D and the gamma_abc / l offsets) is refused, not
    # turned into a wrong proof or a short h buffer; the circuit's shape is readable through the ABI
    assert dr.shape() == (len(cs["z"]), n_inst, len(cs["A"]))
    if n_inst + cw + 1 <= len(z) and cw == 0:
        with pytest.raises(ca.DockGpuError) as ei:
            LG.prove_abi(pk, r, s, v, z, n_inst + 1, circuit=dr)
        assert ei.value.code == -3
    if n_inst > 1:
        with pytest.raises(ca.DockGpuError):
            LG.prove_abi(pk, r, s, v, z, n_inst - 1, circuit=dr)
    dr.free()


@pytest.mark.parametrize("m,cw", [(117, 3), (300, 0)])
def test_key_resident_across_two_contexts(m, cw):
    """dgpu_legogroth16_prove on a key whose five queries are sharded handles over two device contexts (two contexts on the box's one GPU: every
    code path of the multi-GPU form except a second physical device): the same proof as the single-context call, plain shards and tables."""
    ca.init_devices([0, 0])
    cs = LS.circuit(m, x0=5)
    key = LS.setup(cs, cw, seed=700 + m)
    vk = LG.VerifyingKey(key["alpha_g1"], key["beta_g2"], key["gamma_g2"], key["delta_g2"], key["gamma_abc_g1"], key["eta_gamma_inv_g1"], cw)
    small = (vk, key["beta_g1"], key["delta_g1"], key["eta_delta_inv_g1"])
    pk1 = LG.ProvingKey(*small, key["a_query"], key["b_g1_query"], key["b_g2_query"], key["h_query"], key["l_query"])
    sh = lambda curve, q: ca.ShardedDeviceBases(curve, q, ngpus=2)
    pk2 = LG.ProvingKey.from_device(*small, key["a_query"][0].copy(), key["b_g1_query"][0].copy(), key["b_g2_query"][0].copy(),
                                    sh(ca.G1, key["a_query"]), sh(ca.G1, key["b_g1_query"]), sh(ca.G2, key["b_g2_query"]), sh(ca.G1, key["h_query"]), sh(ca.G1, key["l_query"]))
    z = LS.scalars(cs["z"]); n_inst = cs["n_inst"]
    dr = qap.DeviceR1cs(*[qap.csr(cs[k]) for k in "ABC"], len(cs["z"]), n_inst, len(cs["A"]))
    r, s, v = 0x1234567 * 0x9E3779B97F4A7C15 % R, 0xABCDEF01 * 0xBF58476D1CE4E5B9 % R, 0x55AA55 * 0x94D049BB133111EB % R
    pvk = LG.prepare_verifying_key(vk)
    for tables in (False, True):
        if tables:
            for q in (pk2.a_query, pk2.b_g1_query, pk2.b_g2_query, pk2.h_query, pk2.l_query):
                q.precompute(16)
        for rr in (r, 0):
            ref = LG.prove_abi(pk1, rr, s, v, z, n_inst, circuit=dr)
            got = LG.prove_abi(pk2, rr, s, v, z, n_inst, circuit=dr)
            assert all((got[k] == ref[k]).all() for k in ref), (tables, rr)
            assert LG.verify_proof(pvk, got, z[1:n_inst])
        gm = LG.prove_abi(pk2, r, s, v, O.fr_to_mont(z), n_inst, circuit=dr, montgomery=True)
        assert all((gm[k] == ref[k]).all() for k in ref) if False else all((gm[k] == LG.prove_abi(pk1, r, s, v, z, n_inst, circuit=dr)[k]).all() for k in gm)
    dr.free()
