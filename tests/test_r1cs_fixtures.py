"""The reference's own Circom fixtures (tests/golden/r1cs/*.r1cs are data files copied from
/root/reference/legogroth16/test-vectors/bls12-381/, the inputs of legogroth16/src/circom/tests.rs): the `.r1cs` reader
parses them like r1cs_reader.rs does, a witness computed from the circuit's definition satisfies every constraint, and the
witness map over these REAL matrices agrees between the CPU oracle and (on a GPU box) the HIP path."""
import ctypes as C
import os
import numpy as np
import pytest
import torch
import oracle_c as O
import lego_setup as LS
from crypto_amd.r1cs_file import R1csFile, BLS12_381_ORDER
from crypto_amd import qap

HERE = os.path.dirname(os.path.abspath(__file__))
R = BLS12_381_ORDER
fx = lambda name: R1csFile.from_path(os.path.join(HERE, "golden", "r1cs", name))


def witness_multiply2(a, b):                       # out = a * b; wires: 1, out, a, b
    return [1, a * b % R, a, b]


def witness_test1(x):                              # y = x^3 + x + 5; wires: 1, y, x, t1  (circom folds t2 into the last constraint)
    t1 = x * x % R
    return [1, (t1 * x + x + 5) % R, x, t1]


def witness_nconstraints(f, x):                    # intermediate[i] = intermediate[i-1]^2 + i   (nconstraints.circom, n = 2500)
    n = f.n_constraints + 1
    vals = [x % R]
    for i in range(1, n):
        vals.append((vals[-1] * vals[-1] + i) % R)
    w = [1, vals[-1], x % R] + vals[1:-1]          # out aliases intermediate[n-1], in aliases intermediate[0]
    return w


def witness_multiply_n(f, ins):                    # out = prod in[i]; wires: 1, out, in[0..n), intermediates
    acc, inter = ins[0] % R, []
    for v in ins[1:]:
        acc = acc * v % R; inter.append(acc)
    return [1, acc] + [v % R for v in ins] + inter[:-1]


def test_reader_matches_reference_header_semantics():
    f = fx("multiply2.r1cs")
    assert f.prime == R and (f.n_wires, f.n_pub_out, f.n_pub_in, f.n_prv_in, f.n_constraints) == (4, 1, 0, 2, 1)
    assert f.is_satisfied(witness_multiply2(3, 11)) and not f.is_satisfied([1, 34, 3, 11])
    t = fx("test1.r1cs")
    assert t.n_constraints == 2 and t.n_wires == 4 and t.is_satisfied(witness_test1(3)) and witness_test1(3)[1] == 35
    n = fx("nconstraints.r1cs")
    assert n.n_constraints == 2499 and n.n_pub_out == 1 and n.n_prv_in == 1
    w = witness_nconstraints(n, 7)
    assert len(w) == n.n_wires and n.is_satisfied(w)
    # the reference's test: the public output equals the iterated x^2 + i (legogroth16/src/circom/tests.rs:233-258,916-929)
    x = 7
    for i in range(1, 2500):
        x = (x * x + i) % R
    assert w[1] == x
    m = fx("multiply_n.r1cs")
    wm = witness_multiply_n(m, list(range(2, 302)))
    assert len(wm) == m.n_wires and m.is_satisfied(wm)
    with pytest.raises(ValueError):
        R1csFile(b"r1cx" + bytes(100))


def test_reader_rejects_a_file_without_wire2label_like_the_reference():
    """r1cs_reader.rs:91-96 looks section 3 up unconditionally (R1CSFileParsing when it has no offset / size): drop it from a good file"""
    import struct
    data = open(os.path.join(HERE, "golden", "r1cs", "multiply2.r1cs"), "rb").read()
    (nsec,) = struct.unpack_from("<I", data, 8)
    off, kept = 12, []
    for _ in range(nsec):
        typ, size = struct.unpack_from("<IQ", data, off)
        if typ != 3:
            kept.append(data[off:off + 12 + size])
        off += 12 + size
    assert len(kept) == nsec - 1
    R1csFile(data)                                                      # the whole file parses
    with pytest.raises(ValueError, match="wire2label"):
        R1csFile(data[:8] + struct.pack("<I", nsec - 1) + b"".join(kept))


def test_reference_held_known_answer_vector_bn_254():
    """The one known-answer vector the reference's tests hold for a row of SURVEY 8: the `bn_254` sample of
    legogroth16/src/circom/r1cs_reader.rs:283-340 (816 bytes of test DATA, stored as tests/golden/r1cs/bn254_sample.r1cs) and every value that test
    asserts about it (:341-356, basic_checks :250-281).  Coefficients are (wire id, value) pairs in the reference, (value, wire id) here."""
    f = fx("bn254_sample.r1cs")
    assert f.curve == "bn128"
    assert f.prime == 21888242871839275222246405745257275088548364400416034343698204186575808495617
    assert (f.n_wires, f.n_pub_out, f.n_pub_in, f.n_prv_in, f.n_labels, f.n_constraints) == (7, 1, 2, 3, 0x03E8, 3)
    assert len(f.constraints) == 3
    a0, b0, c0 = f.constraints[0]
    assert len(a0) == 2 and a0[0] == (3, 5)                       # constraints[0].a.0[0] = (wire 5, coefficient 3)
    assert f.constraints[2][1][0] == (6, 0)                       # constraints[2].b.0[0] = (wire 0, coefficient 6)
    assert len(f.constraints[1][2]) == 0                          # constraints[1].c is empty
    assert len(f.wire_mapping) == 7 and f.wire_mapping[1] == 3 and f.n_labels >= f.n_wires
    # the other terms of the sample, read off the bytes
    assert a0[1] == (8, 6) and b0 == [(2, 0), (20, 2), (12, 3)] and c0 == [(5, 0), (7, 2)]
    assert f.constraints[2][2] == [(600, 6)] and f.wire_mapping == [0, 3, 10, 11, 12, 15, 324]
    # input_validation (:359-385): a file for the other curve is recognisable as such
    assert fx("multiply2.r1cs").curve == "bls12_381"
    bad = bytearray(open(os.path.join(HERE, "golden", "r1cs", "bn254_sample.r1cs"), "rb").read())
    bad[-56] = 1                                                   # wire 0 mapped to label 1
    with pytest.raises(ValueError, match="Wire 0"):
        R1csFile(bytes(bad))


def _oracle_map(f, w):
    L = O.lib(); L.orc_witness_map.restype = C.c_int
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    D = 1
    while D < f.n_constraints + f.num_inputs:
        D *= 2
    out = np.zeros((D, 4), np.uint64)
    args = []
    for rp, cl, vl in f.csr():
        args += [p(rp), p(cl), p(vl)]
    L.orc_witness_map(*args, p(LS.scalars(w)), C.c_size_t(len(w)), C.c_size_t(f.num_inputs), C.c_size_t(f.n_constraints), p(out))
    return out


def test_oracle_witness_map_on_reference_circuits():
    for name, w in (("multiply2.r1cs", witness_multiply2(3, 11)), ("test1.r1cs", witness_test1(5))):
        f = fx(name)
        h = _oracle_map(f, w)
        assert not h[-1].any()
    f = fx("nconstraints.r1cs")
    h = _oracle_map(f, witness_nconstraints(f, 3))
    assert h.shape == (4096, 4) and not h[-1].any() and h[:100].any()
    # an unsatisfying assignment leaves a non-zero top coefficient (a*b - c is no longer divisible by Z)
    bad = witness_nconstraints(f, 3); bad[5] = (bad[5] + 1) % R
    assert _oracle_map(f, bad)[-1].any()


@pytest.mark.gpu
def test_gpu_witness_map_on_reference_circuits():
    assert torch.cuda.is_available()
    import crypto_amd as ca
    ca.init(0)
    for name, w in (("multiply2.r1cs", witness_multiply2(3, 11)), ("test1.r1cs", witness_test1(5)),
                    ("nconstraints.r1cs", None), ("multiply_n.r1cs", None)):
        f = fx(name)
        if name == "nconstraints.r1cs":
            w = witness_nconstraints(f, 3)
        if name == "multiply_n.r1cs":
            w = witness_multiply_n(f, list(range(2, 302)))
        dr = qap.DeviceR1cs(*f.csr(), f.n_wires, f.num_inputs, f.n_constraints)
        h, _ = dr.witness_map(LS.scalars(w))
        assert (h == _oracle_map(f, w)).all(), name


@pytest.mark.gpu
def test_gpu_prove_and_verify_on_reference_circuits():
    """legogroth16/src/circom/tests.rs (e.g. :206-258): parameters from the `.r1cs` file's matrices, a proof for a witness of the circuit,
    verification against the circuit's public output, rejection of another output — multiply2, test1 and the 2499-constraint nconstraints."""
    assert torch.cuda.is_available()
    import crypto_amd as ca
    from crypto_amd import legogroth16 as LG
    ca.init(0)
    rng = np.random.default_rng(77)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (R - 1) + 1
    g1 = lambda k: O.G1.to_affine(O.G1.mul(O.G1.generator(), O.int_to_limbs(k % R, 4)))[0]
    g2 = lambda k: O.G2.to_affine(O.G2.mul(O.G2.generator(), O.int_to_limbs(k % R, 4)))[0]
    # (test3 / test4: commit_witness_count = 4 as in tests.rs:162-168,199-205 `generate_params_prove_and_verify(.., 4, ..)`)
    for name, cw in (("multiply2.r1cs", 2), ("test1.r1cs", 1), ("nconstraints.r1cs", 1), ("test2.r1cs", 1), ("test3.r1cs", 4), ("test4.r1cs", 4)):
        f = fx(name)
        w = {"multiply2.r1cs": lambda: witness_multiply2(rnd(), rnd()), "test1.r1cs": lambda: witness_test1(rnd()),
             "nconstraints.r1cs": lambda: witness_nconstraints(f, rnd()), "test2.r1cs": lambda: witness_test2(f, rnd(), rnd()),
             "test3.r1cs": lambda: witness_test3(f, *[rnd() for _ in range(6)]), "test4.r1cs": lambda: witness_test4(f, *[rnd() for _ in range(8)])}[name]()
        assert f.is_satisfied(w)
        n_inst, n_wit = f.num_inputs, f.n_wires - f.num_inputs
        pk, _ = LG.generate_parameters(f.rows(0), f.rows(1), f.rows(2), n_inst, n_wit, cw, rnd(), rnd(), rnd(), rnd(), rnd(), rnd(), g1(rnd()), g2(rnd()))
        circ = qap.DeviceR1cs(*f.csr(), f.n_wires, n_inst, f.n_constraints)
        z = LS.scalars(w)
        v = rnd()
        proof = LG.create_proof_with_reduction(pk, circ, rnd(), rnd(), v, z)
        pvk = LG.prepare_verifying_key(pk.vk)
        assert LG.verify_proof(pvk, proof, z[1:n_inst]), name
        bad = z[1:n_inst].copy(); bad[0][0] ^= np.uint64(1)
        assert not LG.verify_proof(pvk, proof, bad), name
        LG.verify_witness_commitment(pk.vk, proof, n_inst - 1, w[n_inst:n_inst + cw], v)
        circ.free()


# ---- test2 / test3 / test4 of legogroth16/src/circom/tests.rs:145-230,874-899: the reference's circuits, its input vectors, its asserted outputs ----
def solve_witness(f, known):
    """the wires of a Circom R1CS from its inputs, without the wasm witness calculator (out of scope): every constraint <A,w> <B,w> = <C,w> of these
    circuits defines one new wire, so propagate until nothing is unknown.  `known`: {wire: value}."""
    r = f.prime
    w = dict(known); w[0] = 1
    def split(lc):
        val, unk = 0, []
        for co, i in lc:
            if i in w:
                val = (val + co * w[i]) % r
            else:
                unk.append((co, i))
        return val, unk
    pending = list(f.constraints)
    while pending:
        rest = []
        for a, b, c in pending:
            (va, ua), (vb, ub), (vc, uc) = split(a), split(b), split(c)
            if not ua and not ub and len(uc) == 1:                      # new wire on the right-hand side
                co, i = uc[0]; w[i] = (va * vb - vc) * pow(co, r - 2, r) % r
            elif not ua and not ub and not uc:
                assert va * vb % r == vc
            elif len(ua) == 1 and not ub and not uc and vb:             # ... or inside one factor
                co, i = ua[0]; w[i] = (vc * pow(vb, r - 2, r) - va) * pow(co, r - 2, r) % r
            elif len(ub) == 1 and not ua and not uc and va:
                co, i = ub[0]; w[i] = (vc * pow(va, r - 2, r) - vb) * pow(co, r - 2, r) % r
            else:
                rest.append((a, b, c))
        assert len(rest) < len(pending), "no constraint determines a new wire"
        pending = rest
    assert len(w) == f.n_wires
    return [w[i] for i in range(f.n_wires)]


def _rand(seed, k):
    rng = np.random.default_rng(seed)
    return [int.from_bytes(rng.bytes(40), "little") % R for _ in range(k)]


def witness_test2(f, x, z):              # test2.circom: y = (x + z)^2 + z + 1; wires 1, y, x, z, ...
    return solve_witness(f, {2: x, 3: z})


def witness_test3(f, x, y, a, b, c, d):  # main {public [x, y]}: wires 1, z1, z2, x, y, a, b, c, d, ...
    return solve_witness(f, {3: x, 4: y, 5: a, 6: b, 7: c, 8: d})


def witness_test4(f, x, y, p, q, a, b, r, s):   # main {public [a, b, r, s]}: wires 1, z1, z2, a, b, r, s, x, y, p, q, ...
    return solve_witness(f, {3: a, 4: b, 5: r, 6: s, 7: x, 8: y, 9: p, 10: q})


def test_reference_circuits_test2_test3_test4_public_outputs():
    """The public wires the reference's tests assert (tests.rs:170-176, :207-230; the inputs of test2_input1.json / test3_input1.json are the
    reference's own vectors, the random cases follow its StdRng cases in shape)"""
    t2 = fx("test2.r1cs")
    for x, z in ((1, 2), (10, 20)) + tuple(tuple(_rand(50 + i, 2)) for i in range(2)):        # test2_input1.json = {"x": 1, "z": 2}
        w = witness_test2(t2, x, z)
        assert t2.is_satisfied(w) and w[1] == ((x + z) ** 2 + z + 1) % R
    t3 = fx("test3.r1cs")
    assert (t3.n_pub_out, t3.n_pub_in, t3.n_prv_in, t3.n_constraints) == (2, 2, 4, 5)
    for vals in ((10, 25, 4, 5, 105, 1000),) + tuple(tuple(_rand(60 + i, 6)) for i in range(3)):   # test3_input1.json
        x, y, a, b, c, d = vals
        w = witness_test3(t3, *vals)
        assert t3.is_satisfied(w)
        public = w[1:t3.num_inputs]
        assert len(public) == 4 and public == [(a * x + b * y + c * d) % R, (c * x + d * y) % R, x, y]
    t4 = fx("test4.r1cs")
    for vals in tuple(tuple(_rand(70 + i, 8)) for i in range(3)):
        x, y, p, q, a, b, r, s = vals
        w = witness_test4(t4, *vals)
        assert t4.is_satisfied(w)
        z1 = (a * x + b * y + 10 * p * q - 19 * r ** 3 * p + 55 * s ** 4 * q ** 3 - 3 * x * x + 6 * x * y - 13 * y ** 3 - r * s * x + 5 * a * b * y
              - 32 * a * x * y - 2 * x * y * p * q - 100) % R
        z2 = (a ** 3 * y + 3 * b * b * x - 20 * x * x * y * y + 45) % R
        public = w[1:t4.num_inputs]
        assert len(public) == 6 and public == [z1, z2, a, b, r, s]
    bad = witness_test3(t3, 1, 2, 3, 4, 5, 6); bad[1] = (bad[1] + 1) % R
    assert not t3.is_satisfied(bad)
