"""GPU (-m gpu): the ABI is re-entrant across entry points — MSM (G1, G2), Miller loop, fixed-base products, the folding kernel and
the witness map issued concurrently from host threads (the reference calls them from rayon workers, SURVEY 8b "Threading") must each
return exactly what they return alone.  Slots (stream + workspace) are shared by all entry points, so this exercises their reuse."""
import threading
import numpy as np
import pytest
import torch
import oracle_c as O
import crypto_amd as ca
from crypto_amd import fixed_base as fb, qap
from crypto_amd.aggregation import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available()
    ca.init(0)


def test_mixed_entry_points_from_eight_threads():
    n = 6000
    k0 = O.rand_scalars(1, 1)[0]; d = O.rand_scalars(2, 1)[0]
    b1 = O.G1.gen_seq(k0, d, n, threads=8); b2 = O.G2.gen_seq(d, k0, 2000, threads=8)
    sc = O.rand_scalars(3, n)
    P, Q = b1[:200], b2[:200]
    tab = fb.WindowTable(ca.G1, O.G1.generator())
    # a small R1CS: x_i = x_{i-1}^2 + i  (tests/lego_setup.py shape) through the CSR entry point
    import lego_setup as LS
    cs = LS.circuit(100, x0=5)
    A, B, Cm = (qap.csr(cs[k]) for k in ("A", "B", "C"))
    z = LS.scalars(cs["z"])
    jobs = {
        "msm_g1": lambda: ca.msm_bigint(ca.G1, b1, sc),
        "msm_g2": lambda: ca.msm_bigint(ca.G2, b2, sc[:2000]),
        "miller": lambda: ca.multi_miller_loop(P, Q),
        "fixed": lambda: tab.multiply_many(sc[:500])[0],
        "fold": lambda: ops.mul_add(ca.G2, b2[:64], 0x123456789ABCDEF, b2[64:128]),
    }
    jobs["qap"] = lambda: qap.witness_map(A, B, Cm, z, cs["n_inst"], cs["n_cons"])[0]
    # resident handles (plain and precomputed-multiples table) and a Miller loop large enough for the tree's two levels; the two-launch form
    # of the line kernel is taken only while at most two Miller loops are in flight, so both forms run here
    plain_h = ca.DeviceBases(ca.G1, b1); table_h = ca.DeviceBases(ca.G1, b1).precompute(0)
    P2, Q2 = np.concatenate([b1[:600]] * 2)[:1100], np.concatenate([b2[:600]] * 2)[:1100]
    jobs["msm_plain_handle"] = lambda: plain_h.msm_bigint(sc)
    jobs["msm_table_handle"] = lambda: table_h.msm_bigint(sc)
    jobs["miller_1100"] = lambda: ca.multi_miller_loop(P2, Q2)
    ref = {k: np.array(f(), copy=True) for k, f in jobs.items()}
    errors = []

    def worker(seed):
        rng = np.random.default_rng(seed)
        names = list(jobs)
        for _ in range(20):
            k = names[int(rng.integers(0, len(names)))]
            got = np.asarray(jobs[k]())
            if got.shape != ref[k].shape or not (got == ref[k]).all():
                errors.append(k)
    ths = [threading.Thread(target=worker, args=(s,)) for s in range(8)]
    for t in ths: t.start()
    for t in ths: t.join()
    tab.free(); plain_h.free(); table_h.free()
    assert not errors, errors
    assert (ref["msm_plain_handle"] == ref["msm_g1"]).all() and (ref["msm_table_handle"] == ref["msm_g1"]).all()
