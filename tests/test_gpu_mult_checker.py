"""GPU (-m gpu): RandomizedMultChecker mirror (crypto_amd/mult_checker.py) — the reference's own test shape
(/root/reference/utils/src/randomized_mult_checker.rs:120-275: valid relations of the add_1 / add_2 / add_3 / add_many kinds verify,
one wrong target makes the batch fail; a point and its negative share an entry)."""
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import crypto_amd as ca
from crypto_amd.mult_checker import RandomizedMultChecker

pytestmark = pytest.mark.gpu
R = U.R


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available()
    ca.init(0)


@pytest.mark.parametrize("curve,grp", [(ca.G1, O.G1), (ca.G2, O.G2)])
def test_mult_checker(curve, grp):
    rng = np.random.default_rng(3)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (R - 1) + 1
    pt = lambda k: grp.to_affine(grp.mul(grp.generator(), O.int_to_limbs(k % R, 4)))[0]
    ks = [rnd() for _ in range(6)]
    g = [pt(k) for k in ks]
    a = [rnd() for _ in range(6)]
    c1 = pt(ks[0] * a[0])
    c2 = pt(ks[1] * a[1] + ks[2] * a[2])
    c3 = pt(ks[3] * a[3] + ks[4] * a[4] + ks[5] * a[5])
    big_n = 300
    bk = [rnd() for _ in range(big_n)]; bs = [rnd() for _ in range(big_n)]
    bp = [pt(k) for k in bk]
    c4 = pt(sum(k * s for k, s in zip(bk, bs)))

    def filled(wrong=None):
        ch = RandomizedMultChecker(curve, rnd())
        ch.add_1(g[0], a[0], c1 if wrong != 1 else pt(5))
        ch.add_2(g[1], a[1], g[2], a[2], c2 if wrong != 2 else pt(6))
        ch.add_3(g[3], a[3], g[4], a[4], g[5], a[5], c3 if wrong != 3 else pt(7))
        ch.add_many(bp, bs, c4 if wrong != 4 else pt(8))
        return ch
    ok = filled()
    assert ok.verify()
    assert len(ok) == 6 + 3 + big_n + 1
    for w in (1, 2, 3, 4):
        assert not filled(w).verify()
    # a point and its negative share one entry; the identity is ignored
    ch = RandomizedMultChecker(curve, rnd())
    from crypto_amd.aggregation.ops import neg
    ch.add_1(g[0], a[0], c1)
    ch.add_1(neg(curve, g[0]), a[0], neg(curve, c1))
    ch.add_1(np.zeros(curve.AW, np.uint64), 12345, np.zeros(curve.AW, np.uint64))
    assert len(ch) == 2 and ch.verify()
    assert RandomizedMultChecker(curve, 7).verify()          # nothing to check
