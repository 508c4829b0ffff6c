"""GPU (-m gpu): the R1CS -> QAP witness map through the C ABI vs the CPU oracle's restatement of
r1cs_to_qap.rs:150-210 (oracle/oracle.c orc_witness_map, itself pinned against naive big-integer polynomial
arithmetic in tests/test_oracle_golden.py), bit-exact on the canonical coefficients; the HBM-resident result feeds
the h_query MSM without leaving the device; and the full prove -> verify round trip with h computed on the GPU."""
import ctypes as C
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import lego_setup as LS
import crypto_amd as ca
from crypto_amd import qap, legogroth16 as LG

pytestmark = pytest.mark.gpu
R = LS.R


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available()
    ca.init(0)


def oracle_map(cs, mats):
    L = O.lib(); L.orc_witness_map.restype = C.c_int
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    D = 1
    while D < cs["n_cons"] + cs["n_inst"]:
        D *= 2
    out = np.zeros((D, 4), np.uint64)
    z = LS.scalars(cs["z"])
    args = []
    for rp, cl, vl in mats:
        args += [p(rp), p(cl), p(vl)]
    L.orc_witness_map(*args, p(z), C.c_size_t(len(cs["z"])), C.c_size_t(cs["n_inst"]), C.c_size_t(cs["n_cons"]), p(out))
    return out


@pytest.mark.parametrize("m", [1, 5, 40, 1000, (1 << 14) - 3, 40000])
def test_witness_map_vs_oracle(m):
    cs = LS.circuit(m, x0=3)
    mats = [qap.csr(cs[k]) for k in "ABC"]
    z = LS.scalars(cs["z"])
    ref = oracle_map(cs, mats)
    h, _ = qap.witness_map(*mats, z, cs["n_inst"], cs["n_cons"])
    assert h.shape == ref.shape and (h == ref).all()
    assert not h[-1].any()                     # deg h <= D - 2: the prover pairs h[..D-1] with h_query (prover.rs:286)
    if m == 40:
        assert [O.limbs_to_int(x) for x in h] == LS.witness_map(cs)       # and the naive big-integer version
    # the same through a device-resident circuit handle (matrices uploaded once, assignment per proof)
    dr = qap.DeviceR1cs(*mats, len(cs["z"]), cs["n_inst"], cs["n_cons"])
    h2, _ = dr.witness_map(z)
    assert (h2 == ref).all()
    with pytest.raises(ca.DockGpuError):
        dr.witness_map(z[:-1])
    dr.free()
    # Montgomery inputs (what the reference holds in memory) give the same canonical output
    matsm = [(rp, cl, O.fr_to_mont(vl)) for rp, cl, vl in mats]
    hm, _ = qap.witness_map(*matsm, O.fr_to_mont(z), cs["n_inst"], cs["n_cons"], montgomery=True)
    assert (hm == ref).all()
    # ... and DGPU_WM_H_MONTGOMERY hands h back as the &[Fr] witness_map_from_matrices returns (r1cs_to_qap.rs:150-210), the resident copy staying canonical
    hmm, dh = qap.witness_map(*matsm, O.fr_to_mont(z), cs["n_inst"], cs["n_cons"], montgomery=True, h_montgomery=True, resident=True)
    assert (hmm == O.fr_to_mont(ref)).all()
    dr = qap.DeviceR1cs(*mats, len(cs["z"]), cs["n_inst"], cs["n_cons"])
    h3, _ = dr.witness_map(z, h_montgomery=True)
    assert (h3 == hmm).all()
    if m >= 1000:
        bases, _, _ = U.seq_bases(O.G1, len(ref), 17, threads=16)
        assert (ca.DeviceBases(ca.G1, bases).msm_resident(dh) == ca.msm_bigint(ca.G1, bases, ref)).all()
    dh.free(); dr.free()


def test_resident_h_feeds_the_msm_and_the_proof_verifies():
    m, cw = 300, 2
    cs = LS.circuit(m, x0=9)
    key = LS.setup(cs, cw, seed=5)
    mats = [qap.csr(cs[k]) for k in "ABC"]
    z = LS.scalars(cs["z"])
    h, dh = qap.witness_map(*mats, z, cs["n_inst"], cs["n_cons"], resident=True)
    vk = LG.VerifyingKey(key["alpha_g1"], key["beta_g2"], key["gamma_g2"], key["delta_g2"], key["gamma_abc_g1"], key["eta_gamma_inv_g1"], cw)
    pk = LG.ProvingKey(vk, key["beta_g1"], key["delta_g1"], key["eta_delta_inv_g1"], key["a_query"], key["b_g1_query"], key["b_g2_query"], key["h_query"], key["l_query"])
    # h_acc straight from HBM == h_acc from host scalars (truncation to D - 1 terms, prover.rs:286)
    acc_res = pk.h_query.msm_resident(dh)
    assert (acc_res == pk.h_query.msm_bigint(h)).all()
    inp, wit = z[:cs["n_inst"]], z[cs["n_inst"]:]
    proof = LG.create_proof(pk, 12345, 67890, 424242, h, inp, wit)
    pvk = LG.prepare_verifying_key(vk)
    assert LG.verify_proof(pvk, proof, inp[1:])
    dr = qap.DeviceR1cs(*mats, len(cs["z"]), cs["n_inst"], cs["n_cons"])
    pr2 = LG.create_proof_with_reduction(pk, dr, 12345, 67890, 424242, z)          # prover.rs:153-180: witness map inside
    assert all((pr2[k] == proof[k]).all() for k in proof)
    dr.free()
    h_bad = h.copy(); h_bad[3][0] ^= np.uint64(1)
    assert not LG.verify_proof(pvk, LG.create_proof(pk, 12345, 67890, 424242, h_bad, inp, wit), inp[1:])


@pytest.mark.parametrize("logd,k,skew", [(16, 160, False), (18, 48, False), (18, 96, True), (17, 400, True)])
def test_dense_rows_do_not_outgrow_the_lazy_representation(logd, k, skew):
    """Rows with many terms (hash-function circuits have linear combinations of tens to hundreds of variables): the matrix-vector products
    and the partial sums of the decimation-in-frequency stages stay inside what the lazy Fr representation and its subtraction
    constants can hold (D x k >= 2^23 terms; the inputs need not satisfy the constraints — both sides interpolate the same coset
    evaluations, r1cs_to_qap.rs:186-205).  skew: only the odd rows carry terms, so the last inverse stages subtract a partial sum of
    2^(logd-1) x k products from (nearly) nothing — the worst case for a subtraction constant sized for balanced operands."""
    n_inst = 2
    m = (1 << logd) - n_inst
    nv = 5000
    rng = np.random.default_rng(logd)
    per_row = np.full(m, k, dtype=np.uint64)
    if skew:
        per_row[0::2] = 0
    rp = np.concatenate([[0], np.cumsum(per_row)]).astype(np.uint64)
    nnz = int(rp[-1])
    mats = []
    for i in range(3):
        cols = rng.integers(0, nv, nnz, dtype=np.uint32)
        vals = O.rand_scalars(900 + 3 * logd + i, nnz)
        mats.append((rp, cols, vals))
    z = O.rand_scalars(77 + logd, nv)
    cs = {"n_cons": m, "n_inst": n_inst, "z": [O.limbs_to_int(x) for x in z]}
    ref = oracle_map(cs, mats)
    h, _ = qap.witness_map(*mats, z, n_inst, m)
    assert (h == ref).all()


@pytest.mark.parametrize("logd", [9, 10, 11, 12, 13, 15, 17, 19])
def test_every_pass_schedule_of_the_ntt(logd):
    """One domain size per way the transform is cut into passes (per-stage kernels below 2^10, then one flat pass, flat + one strided pass
    with an odd / even number of stages, flat + two strided passes): random sparse matrices (2 terms per row), same coset interpolation on
    both sides."""
    n_inst = 3
    m = (1 << logd) - n_inst
    nv = 700
    rng = np.random.default_rng(100 + logd)
    rp = (np.arange(m + 1, dtype=np.uint64) * np.uint64(2))
    mats = [(rp, rng.integers(0, nv, 2 * m, dtype=np.uint32), O.rand_scalars(300 + 3 * logd + i, 2 * m)) for i in range(3)]
    z = O.rand_scalars(55 + logd, nv)
    cs = {"n_cons": m, "n_inst": n_inst, "z": [O.limbs_to_int(x) for x in z]}
    ref = oracle_map(cs, mats)
    h, _ = qap.witness_map(*mats, z, n_inst, m)
    assert (h == ref).all()
