"""SRS, commitment keys and pair commitments of the aggregation —
/root/reference/legogroth16/src/aggregation/srs.rs (GenericSRS, specialize :180-237, setup_fake_srs :311-392,
structured_generators_scalar_power :394-412), key.rs (Key::{scale, split, compress, first} :96-187) and
commitment.rs (PairCommitment::{single, double} :23-69)."""
import numpy as np
from . import ops
from .ops import G1, G2, R_MOD
from ..fixed_base import WindowTable

MAX_SRS_SIZE = (2 << 19) + 1


class AggregationError(Exception):
    pass


def structured_generators_scalar_power(curve, num, g, s):
    """[g, s g, s^2 g, ...] (srs.rs:394-412): one device fixed-base batch"""
    assert num > 0
    pw, acc = [], 1
    for _ in range(num):
        pw.append(acc); acc = acc * s % R_MOD
    with WindowTable(curve, g, num) as t:
        out, _ = t.multiply_many(pw)
    return out


class Key:
    """Key<G> { a, b } — VKey over G2, WKey over G1 (key.rs:41-57)"""

    def __init__(self, curve, a, b):
        self.curve, self.a, self.b = curve, ops.pts(curve, a), ops.pts(curve, b)

    def __len__(self):
        return len(self.a)

    def has_correct_len(self, n):
        return len(self.a) == n and len(self.b) == n

    def ensure_sufficient_len(self, m):
        if len(self.a) < len(m):
            raise AggregationError("InsufficientKeyLength(%d)" % len(self.a))

    def scale(self, s_vec):                                  # key.rs:117-139
        if len(self.a) != len(s_vec):
            raise AggregationError("InvalidKeyLength")
        both = ops.mul_add(self.curve, np.concatenate([self.a, self.b]), list(s_vec) + list(s_vec))
        return Key(self.curve, both[:len(self.a)], both[len(self.a):])

    def split(self, at):                                     # key.rs:142-155
        return Key(self.curve, self.a[:at], self.b[:at]), Key(self.curve, self.a[at:], self.b[at:])

    def compress(self, right, scale):                        # key.rs:160-184: left + scale * right, both vectors in one launch
        if len(self.a) != len(right.a):
            raise AggregationError("InvalidKeyLength")
        both = ops.mul_add(self.curve, np.concatenate([right.a, right.b]), int(scale), np.concatenate([self.a, self.b]))
        return Key(self.curve, both[:len(self.a)], both[len(self.a):])

    def first(self):
        return self.a[0].copy(), self.b[0].copy()


class PairCommitment:
    """(t, u) in GT^2 (commitment.rs:14-19)"""

    def __init__(self, t, u):
        self.t, self.u = t, u

    def to_bytes(self):
        return ops.gt_bytes(self.t) + ops.gt_bytes(self.u)

    def __eq__(self, o):
        return bool((self.t == o.t).all() and (self.u == o.u).all())

    @staticmethod
    def single(vkey, a_vec):                                 # commitment.rs:23-33
        a_vec = ops.pts(G1, a_vec)
        vkey.ensure_sufficient_len(a_vec)
        n = len(a_vec)
        return PairCommitment(ops.multi_pairing(a_vec, vkey.a[:n]), ops.multi_pairing(a_vec, vkey.b[:n]))

    # the same commitments as lists of (G1 vector, G2 vector) pairing jobs, so that a caller can put several of them into one
    # ops.multi_pairings call; PairCommitment(*results) rebuilds the value
    @staticmethod
    def single_jobs(vkey, a_vec):
        a_vec = ops.pts(G1, a_vec)
        vkey.ensure_sufficient_len(a_vec)
        n = len(a_vec)
        return [(a_vec, vkey.a[:n]), (a_vec, vkey.b[:n])]

    @staticmethod
    def double_jobs(vkey, wkey, a, b):
        a, b = ops.pts(G1, a), ops.pts(G2, b)
        na, nb = len(a), len(b)
        return [(np.concatenate([a, wkey.a[:nb]]), np.concatenate([vkey.a[:na], b])), (np.concatenate([a, wkey.b[:nb]]), np.concatenate([vkey.b[:na], b]))]

    @staticmethod
    def double(vkey, wkey, a, b):                            # commitment.rs:36-69: prod e(a_i, v_i) e(w_i, b_i)
        a, b = ops.pts(G1, a), ops.pts(G2, b)
        na, nb = len(a), len(b)
        t = ops.multi_pairing(np.concatenate([a, wkey.a[:nb]]), np.concatenate([vkey.a[:na], b]))
        u = ops.multi_pairing(np.concatenate([a, wkey.b[:nb]]), np.concatenate([vkey.b[:na], b]))
        return PairCommitment(t, u)


class ProverSRS:
    def __init__(self, n, g_alpha_powers_table, h_alpha_powers_table, g_beta_powers_table, h_beta_powers_table, vkey, wkey):
        self.n = n
        self.g_alpha_powers_table, self.h_alpha_powers_table = g_alpha_powers_table, h_alpha_powers_table
        self.g_beta_powers_table, self.h_beta_powers_table = g_beta_powers_table, h_beta_powers_table
        self.vkey, self.wkey = vkey, wkey

    def has_correct_len(self, n):
        return self.vkey.has_correct_len(n) and self.wkey.has_correct_len(n)


class VerifierSRS:
    def __init__(self, n, g, h, g_alpha, g_beta, h_alpha, h_beta):
        self.n, self.g, self.h, self.g_alpha, self.g_beta, self.h_alpha, self.h_beta = n, g, h, g_alpha, g_beta, h_alpha, h_beta


class GenericSRS:
    def __init__(self, g_alpha_powers, h_alpha_powers, g_beta_powers, h_beta_powers):
        self.g_alpha_powers, self.h_alpha_powers = g_alpha_powers, h_alpha_powers
        self.g_beta_powers, self.h_beta_powers = g_beta_powers, h_beta_powers

    def specialize(self, num_proofs):                        # srs.rs:180-237
        n = int(num_proofs)
        assert n > 0 and n & (n - 1) == 0
        tn = 2 * n
        for v in (self.g_alpha_powers, self.h_alpha_powers, self.g_beta_powers, self.h_beta_powers):
            assert len(v) >= tn
        vkey = Key(G2, self.h_alpha_powers[:n], self.h_beta_powers[:n])
        wkey = Key(G1, self.g_alpha_powers[n:tn], self.g_beta_powers[n:tn])
        pk = ProverSRS(n, self.g_alpha_powers[:tn].copy(), self.h_alpha_powers[:n].copy(), self.g_beta_powers[:tn].copy(),
                       self.h_beta_powers[:n].copy(), vkey, wkey)
        vk = VerifierSRS(n, self.g_alpha_powers[0].copy(), self.h_alpha_powers[0].copy(), self.g_alpha_powers[1].copy(),
                         self.g_beta_powers[1].copy(), self.h_alpha_powers[1].copy(), self.h_beta_powers[1].copy())
        return pk, vk


def setup_fake_srs(alpha, beta, size, g, h):
    """srs.rs:311-392 with the two secrets and the generators passed in (the reference draws alpha, beta from `rng` and uses
    the curve generators)"""
    return GenericSRS(structured_generators_scalar_power(G1, 2 * size, g, alpha), structured_generators_scalar_power(G2, 2 * size, h, alpha),
                      structured_generators_scalar_power(G1, 2 * size, g, beta), structured_generators_scalar_power(G2, 2 * size, h, beta))
