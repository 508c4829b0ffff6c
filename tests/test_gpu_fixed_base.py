"""GPU parity: fixed-base batch multiplication (WindowTable, utils/src/msm.rs:8-62; FixedBase::msm of
legogroth16/src/generator.rs:335-399) against the CPU oracle's double-and-add, through the C ABI.

Follows the reference's own test (utils/src/msm.rs:196-231: `table.multiply_many(&scalars)[i] == g * scalars[i]`)."""
import numpy as np
import pytest
import torch
import oracle_c as O
import crypto_amd as ca
from crypto_amd import fixed_base as fb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available()
    ca.init(0)

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def _expect(grp, base, scalars):
    exp = [grp.to_affine(grp.mul(base, s)) for s in scalars]
    return np.stack([e[0] for e in exp]), np.array([e[1] for e in exp], dtype=np.uint8)


@pytest.mark.parametrize("curve", ["g1", "g2"])
def test_multiply_many_matches_double_and_add(curve):
    grp, cv = (O.G1, ca.G1) if curve == "g1" else (O.G2, ca.G2)
    base = grp.to_affine(grp.mul(grp.generator(), O.int_to_limbs(0xB16B00B5, 4)))[0]
    n = 300
    s = O.rand_scalars(0x51DE + (curve == "g2"), n)
    # edge scalars: 0, 1, r-1, a single top byte, all bytes 0xff below r, zero bytes inside
    edge = [0, 1, R - 1, 0x73 << 248, (1 << 248) - 1, 0x0100000000000000FF, 2, 255, 256]
    for k, v in enumerate(edge):
        s[k] = O.int_to_limbs(v, 4)
    exp_xy, exp_inf = _expect(grp, base, s)
    with fb.WindowTable(cv, base, n) as t:
        out, inf = t.multiply_many(s)
        assert (inf == exp_inf).all() and inf[0] == 1 and inf[1:].sum() == 0
        assert (out == exp_xy).all()
        one, one_inf = t.multiply(5)
        assert not one_inf and (one == grp.to_affine(grp.mul(base, O.int_to_limbs(5, 4)))[0]).all()
        # Montgomery-form scalars (what &[Fr] holds)
        out_m, inf_m = t.multiply_many(O.fr_to_mont(s), montgomery=True)
        assert (out_m == exp_xy).all() and (inf_m == exp_inf).all()
    out2, inf2 = fb.multiply_field_elems_with_same_group_elem(cv, base, s)
    assert (out2 == exp_xy).all() and (inf2 == exp_inf).all()


def test_identity_base_and_empty():
    s = O.rand_scalars(7, 10)
    out, inf = fb.multiply_field_elems_with_same_group_elem(ca.G1, np.zeros(12, dtype=np.uint64), s)
    assert inf.all() and not out.any()
    out, inf = fb.multiply_field_elems_with_same_group_elem(ca.G2, O.G2.generator(), np.zeros((0, 4), dtype=np.uint64))
    assert out.shape == (0, 24) and inf.shape == (0,)


def test_large_batch_sum_property():
    """n = 2^16: sum_i (s_i * B) == (sum_i s_i) * B  (checked with the oracle's point addition through an MSM of ones)"""
    n = 1 << 16
    s = O.rand_scalars(99, n)
    base = O.G1.generator()
    out, inf = fb.multiply_field_elems_with_same_group_elem(ca.G1, base, s)
    assert not inf.any()
    ones = np.zeros((n, 4), dtype=np.uint64); ones[:, 0] = 1
    total = ca.msm_bigint(ca.G1, out, ones)
    ssum = sum(O.limbs_to_int(x) for x in s) % R
    exp = O.G1.to_affine(O.G1.mul(base, O.int_to_limbs(ssum, 4)))[0]
    assert (O.G1.to_affine(total)[0] == exp).all()
    # spot-check individual outputs
    for i in (0, 1, n // 2, n - 1):
        assert (out[i] == O.G1.to_affine(O.G1.mul(base, s[i]))[0]).all()


@pytest.mark.parametrize("curve", ["g1", "g2"])
def test_products_as_resident_bases(curve):
    """multiply_many_to_bases + MSM on the handle == MSM over the downloaded products == (sum_i t_i s_i) * B"""
    grp, cv = (O.G1, ca.G1) if curve == "g1" else (O.G2, ca.G2)
    n = 5000
    s = O.rand_scalars(11, n); t = O.rand_scalars(12, n)
    s[3] = 0                                  # an identity among the bases
    with fb.WindowTable(cv, grp.generator()) as tab:
        db = tab.multiply_many_to_bases(s)
        got = db.msm_bigint(t)
        out, inf = tab.multiply_many(s)
        assert inf[3] == 1
        assert (got == ca.msm_bigint(cv, out, t, is_inf=inf)).all()
        acc = sum(O.limbs_to_int(a) * O.limbs_to_int(b) for a, b in zip(s, t)) % R
        assert (grp.to_affine(got)[0] == grp.to_affine(grp.mul(grp.generator(), O.int_to_limbs(acc, 4)))[0]).all()
        db.free()
