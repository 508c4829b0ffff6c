"""GPU (-m gpu): LegoGroth16 prove with the five large MSMs chunked over two ranks — two processes on the one GPU of the box,
gloo for the 864-byte exchange (RCCL refuses two ranks on one device; on a multi-GPU node the same code runs over "nccl").  Both ranks
must return the proof the single-process prover returns, and it must verify."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import lego_setup as LS
    import crypto_amd as ca
    from crypto_amd import legogroth16 as LG
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ca.init(0)
    m, cw = 150, 3
    cs = LS.circuit(m, x0=9)
    key = LS.setup(cs, cw, seed=77)
    vk = LG.VerifyingKey(key["alpha_g1"], key["beta_g2"], key["gamma_g2"], key["delta_g2"], key["gamma_abc_g1"], key["eta_gamma_inv_g1"], cw)
    args = (vk, key["beta_g1"], key["delta_g1"], key["eta_delta_inv_g1"], key["a_query"], key["b_g1_query"], key["b_g2_query"], key["h_query"], key["l_query"])
    spk = LG.ShardedProvingKey(*args, world, rank)
    z = cs["z"]
    inp, wit = LS.scalars(z[:cs["n_inst"]]), LS.scalars(z[cs["n_inst"]:])
    h = LS.scalars(LS.witness_map(cs))
    proof = LG.create_proof_sharded(spk, 111, 222, 333, h, inp, wit)
    ref = LG.create_proof(LG.ProvingKey(*args), 111, 222, 333, h, inp, wit)
    same = all((proof[k] == ref[k]).all() for k in ref)
    ok = LG.verify_proof(LG.prepare_verifying_key(vk), proof, inp[1:])
    q.put((rank, bool(same), bool(ok)))
    dist.destroy_process_group()


def test_sharded_prove_two_ranks():
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] for r in res), res


def _worker_msm_ml(rank, world, port, q):
    """HIP partials through the collective: every rank runs the device pipeline on its chunk, all_gather + fold; compared with the
    single-device call on the whole input and with the CPU oracle."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_c as O
    import crypto_amd as ca
    from crypto_amd import sharded, pairing
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ca.init(0)
    ok = True
    for curve, G, n in ((ca.G1, O.G1, 20001), (ca.G2, O.G2, 3001)):
        k0 = O.rand_scalars(1, 1)[0]; d = O.rand_scalars(2, 1)[0]
        bases = G.gen_seq(k0, d, n, threads=4)
        sc = O.rand_scalars(3, n)
        lo, hi = sharded.chunk_bounds(n, world, rank)
        full = sharded.msm_sharded(curve, lambda: ca.msm_bigint(curve, bases[lo:hi], sc[lo:hi]))
        single = ca.msm_bigint(curve, bases, sc)
        ref = G.msm(bases, sc, threads=4)
        ok = ok and bool((full == single).all()) and bool((G.to_affine(ref)[0] == full[:G.AW]).all())
    npairs = 301
    ps = O.G1.gen_seq(O.rand_scalars(4, 1)[0], O.rand_scalars(5, 1)[0], npairs, threads=4)
    qs = O.G2.gen_seq(O.rand_scalars(6, 1)[0], O.rand_scalars(7, 1)[0], npairs, threads=4)
    plo, phi = sharded.chunk_bounds(npairs, world, rank)
    f = sharded.multi_miller_loop_sharded(lambda: pairing.multi_miller_loop(ps[plo:phi], qs[plo:phi]))
    ok_ml = bool((f == pairing.multi_miller_loop(ps, qs)).all()) and bool((f == O.multi_miller_loop(ps, qs, threads=4)).all())
    q.put((rank, ok, ok_ml))
    dist.destroy_process_group()


def test_hip_partials_through_the_collective_two_ranks():
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_msm_ml, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] for r in res), res
