"""CPU: the aggregation's Fiat-Shamir transcript (crypto_amd/aggregation/transcript.py) — Keccak-f[1600] pinned against
hashlib's SHA3 (same permutation), the Merlin framing against the published Merlin "simple transcript" known answer
(dalek-cryptography/merlin, reproduced by its ports), challenge_scalar against the rule in
/root/reference/utils/src/transcript.rs:103-122; plus the KZG polynomial helpers of aggregation/kzg.rs:238-292."""
import hashlib
from crypto_amd.aggregation.transcript import keccak_f1600, keccak_f1600_py, Merlin, MerlinTranscript, R_MOD
from crypto_amd.aggregation import kzg


def _sha3_256(msg):
    rate, st = 136, bytearray(200)
    m = bytearray(msg) + b"\x06"
    m += b"\x00" * ((-len(m)) % rate)
    m[-1] |= 0x80
    for i in range(0, len(m), rate):
        for j in range(rate):
            st[j] ^= m[i + j]
        keccak_f1600(st)
    return bytes(st[:32])


def test_keccak_matches_hashlib_sha3():
    for msg in (b"", b"abc", b"q" * 135, b"q" * 136, b"x" * 1000):
        assert _sha3_256(msg) == hashlib.sha3_256(msg).digest()


def test_native_and_python_permutations_agree():
    st = bytearray(range(200))
    a, b = bytearray(st), bytearray(st)
    for _ in range(3):
        keccak_f1600(a); keccak_f1600_py(b)
        assert a == b


def test_merlin_simple_transcript_known_answer():
    t = Merlin(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_merlin_long_messages_cross_the_rate():
    """absorbing / squeezing more than one STROBE block (166 bytes) stays deterministic and sensitive to every byte"""
    a, b = Merlin(b"p"), Merlin(b"p")
    msg = bytes(range(256)) * 3
    a.append_message(b"m", msg); b.append_message(b"m", msg[:-1] + b"\x00")
    ca, cb = a.challenge_bytes(b"c", 400), b.challenge_bytes(b"c", 400)
    assert len(ca) == 400 and ca != cb
    c = Merlin(b"p"); c.append_message(b"m", msg)
    assert c.challenge_bytes(b"c", 400) == ca


def test_challenge_scalar_is_inverse_of_sampled_element():
    t1, t2 = MerlinTranscript(b"agg"), MerlinTranscript(b"agg")
    t1.append(b"x", b"\x01" * 48); t2.append(b"x", b"\x01" * 48)
    c = t1.challenge_scalar(b"r")
    raw = t2.challenge_bytes(b"r", 64)
    v = int.from_bytes(raw[:32], "little") & ((1 << 255) - 1)
    if 0 < v < R_MOD:
        assert c * v % R_MOD == 1
    assert 0 < c < R_MOD


def test_kzg_polynomial_helpers():
    tr = [3, 5, 7, 11]
    r_shift, z = 123456789, 987654321
    co = kzg.polynomial_coefficients_from_transcript(tr, r_shift)
    assert len(co) == 16
    ev = sum(c * pow(z, i, R_MOD) for i, c in enumerate(co)) % R_MOD
    assert ev == kzg.polynomial_evaluation_product_form_from_transcript(tr, z, r_shift)
    # quotient: (f(X) - f(z)) == q(X) (X - z)
    p = list(co); p[0] = (p[0] - ev) % R_MOD
    q = kzg._quotient_by_linear(p, z)
    back = [0] * len(p)
    for i, c in enumerate(q):
        back[i + 1] = (back[i + 1] + c) % R_MOD
        back[i] = (back[i] - c * z) % R_MOD
    assert back == p


def test_native_merlin_is_the_python_merlin_byte_for_byte():
    """merlin_native.c (the dgpu_transcript callbacks as plain C functions: what a Rust host's merlin::Transcript costs the library) against the Python
    statement: the same STROBE state after every operation, the same challenge bytes and scalars (the inverse of the sampled element), clones included"""
    import os
    import random
    from crypto_amd.aggregation import transcript as T
    T.build_helper()
    rnd = random.Random(5)
    for label in (b"", b"snarkpack", b"x" * 200):
        a = T.MerlinTranscript(label); b = T.NativeMerlinTranscript(label)
        for i in range(120):
            lab = bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 40)))
            op = rnd.randrange(4)
            if op <= 1:
                msg = os.urandom(rnd.choice([0, 1, 31, 32, 48, 96, 165, 166, 167, 332, 576, 1000]))
                a.append(lab, msg); b.append(lab, msg)
            elif op == 2:
                assert a.challenge_scalar(lab) == b.challenge_scalar(lab)
            else:
                k = rnd.choice([1, 32, 64, 166, 400])
                assert a.challenge_bytes(lab, k) == b.challenge_bytes(lab, k)
            st = b.state()
            assert bytes(a.merlin.strobe.state) == st[:200] and (a.merlin.strobe.pos, a.merlin.strobe.pos_begin) == (st[200], st[201]), (label, i)
        c = b.clone()
        assert c.challenge_scalar(b"after") == a.challenge_scalar(b"after") == b.challenge_scalar(b"after")
    # the published Merlin vector through the C path as well
    t = T.NativeMerlinTranscript(b"test protocol")
    t.append(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
