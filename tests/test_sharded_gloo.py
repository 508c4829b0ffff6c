"""CPU, world_size 2, gloo: the multi-GPU path (point-chunk sharding + all_gather of the per-rank partial point
+ local fold) is correct by construction.  On the GPU box each rank's partial comes from the HIP pipeline; here
the per-rank partial is produced by the oracle (test infrastructure) so that the collective, the partition and
the product's host fold (dgpu_fold_g1) are what is under test."""
import os
import socket
import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_c as O
    import crypto_amd as ca
    from crypto_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k0 = O.rand_scalars(1, 1)[0]; d = O.rand_scalars(2, 1)[0]
    bases = O.G1.gen_seq(k0, d, n, threads=2)          # same inputs on every rank (seeded)
    sc = O.rand_scalars(3, n)
    lo, hi = sharded.chunk_bounds(n, world, rank)
    full = sharded.msm_sharded(ca.G1, lambda: O.G1.msm(bases[lo:hi], sc[lo:hi]))
    ref = O.G1.msm(bases, sc)
    ok = (O.G1.to_affine(full)[0] == O.G1.to_affine(ref)[0]).all()
    # every rank must hold the identical, normalised result
    gathered = [None] * world
    dist.all_gather_object(gathered, full.tobytes())
    same = all(g == gathered[0] for g in gathered)
    # Miller loop over pair chunks: product of the per-rank raw Fp12 outputs == the single multi_miller_loop (bit-exact)
    npairs = 9
    ps = np.stack([O.G1.to_affine(O.G1.mul(O.G1.generator(), s))[0] for s in O.rand_scalars(4, npairs)])
    qs = np.stack([O.G2.to_affine(O.G2.mul(O.G2.generator(), s))[0] for s in O.rand_scalars(5, npairs)])
    plo, phi = sharded.chunk_bounds(npairs, world, rank)
    f = sharded.multi_miller_loop_sharded(lambda: O.multi_miller_loop(ps[plo:phi], qs[plo:phi]))
    ok_ml = bool((f == O.multi_miller_loop(ps, qs)).all())
    q.put((rank, bool(ok) and ok_ml, same))
    dist.destroy_process_group()


def test_sharded_msm_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 1001, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] for r in res), res
