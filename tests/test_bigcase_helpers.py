"""CPU: the helpers of the full-size GPU tests agree with their plain-Python definitions."""
import oracle_c as O
import lego_setup as LS
from crypto_amd import qap
from bigcase import dot_mod_r, big_circuit, R


def test_dot_mod_r_helper():
    a = O.rand_scalars(1, 3000); b = O.rand_scalars(2, 3000)
    exp = sum(O.limbs_to_int(x) * O.limbs_to_int(y) for x, y in zip(a, b)) % R
    assert dot_mod_r(a, b) == exp


def test_big_circuit_builder_matches_the_list_form():
    z, A, B, Cm, n_inst, nc = big_circuit(50, 3)
    cs = LS.circuit(50, 3)
    assert z == cs["z"] and nc == cs["n_cons"]
    for got, key in ((A, "A"), (B, "B"), (Cm, "C")):
        rp, cl, vl = qap.csr(cs[key])
        assert (got[0] == rp).all() and (got[1] == cl).all() and (got[2] == vl).all()
