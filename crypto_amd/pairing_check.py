"""Mirror of `dock_crypto_utils::randomized_pairing_check::RandomizedPairingChecker<Bls12_381>`
(/root/reference/utils/src/randomized_pairing_check.rs:24-215), running on the C ABI.

Same state, same methods, same laziness semantics:
    left    product of Miller-loop outputs accumulated so far              (:27)
    right   GT target, `right += out.mul_bigint(m)`  (GT is written additively in arkworks; it is the Fp12 product)  (:30)
    pending (G1, G2) pairs queued for one big multi_miller_loop in verify()  when lazy   (:34, :204-214)
    random / current_random  r and r^k; equation k is scaled by r^k         (:36-38)
The G1 scalings `a.mul_bigint(m)` run batched on the GPU (dgpu_g1_scale_batch), Miller loops through
dgpu_multi_miller_loop, GT arithmetic and the single final exponentiation on the host (dgpu_fp12_*,
dgpu_final_exponentiation).  Points are numpy uint64 arrays in the ABI layout; scalars are Python ints.
"""
import ctypes as C
import numpy as np
from ._native import lib, DockGpuError
from .msm import _ensure
from . import pairing

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_ONE6 = np.array([0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745, 0x5c071a97a256ec6d, 0x15f65ec3fa80e493], dtype=np.uint64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _limbs(v):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def fp12_one():
    o = np.zeros(72, dtype=np.uint64)
    o[:6] = _ONE6
    return o


def fp12_mul(a, b):
    out = np.zeros(72, dtype=np.uint64)
    rc = lib().dgpu_fp12_mul(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_fp12_mul")
    return out


def fp12_pow(a, e):
    out = np.zeros(72, dtype=np.uint64)
    rc = lib().dgpu_fp12_pow(_p(np.ascontiguousarray(a)), _p(_limbs(e % R_MOD)), _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_fp12_pow")
    return out


def fp12_multi_pow(bases, exps):
    """prod bases[i]^exps[i] (dgpu_fp12_multi_pow: host threads, shared squarings)"""
    n = len(bases)
    a = np.ascontiguousarray(np.stack([np.asarray(b, dtype=np.uint64).reshape(72) for b in bases]))
    e = np.ascontiguousarray(np.stack([_limbs(int(x) % R_MOD) for x in exps]))
    out = np.zeros(72, dtype=np.uint64)
    rc = lib().dgpu_fp12_multi_pow(_p(a), _p(e), n, _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_fp12_multi_pow")
    return out


def g1_scale(points, m, negate=False):
    """[m * P for P in points] as affine ABI points (identity -> zero words + flag), `-` if negate"""
    _ensure()
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
    n = len(pts)
    out = np.zeros((n, 12), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    neg = np.ones(n, dtype=np.uint8) if negate else None
    rc = lib().dgpu_g1_scale_batch(_p(pts), None, _p(_limbs(m % R_MOD)), 0, _p(neg), n, _p(out), _p(inf))
    if rc:
        raise DockGpuError(rc, "dgpu_g1_scale_batch")
    return out, inf


def g1_scale_each(points, scalar_limbs, negate=None):
    """[s_i * P_i] (or its negative where negate[i]) as affine ABI points: one launch with one scalar per point"""
    _ensure()
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
    sc = np.ascontiguousarray(scalar_limbs, dtype=np.uint64).reshape(-1, 4)
    n = len(pts)
    out = np.zeros((n, 12), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    neg = None if negate is None else np.ascontiguousarray(negate, dtype=np.uint8)
    rc = lib().dgpu_g1_scale_batch(_p(pts), None, _p(sc), 4, _p(neg), n, _p(out), _p(inf))
    if rc:
        raise DockGpuError(rc, "dgpu_g1_scale_batch")
    return out, inf


def _g2(b):
    """`impl Into<E::G2Prepared>`: an affine (n, 24) array stays affine (prepared inside the fused line kernel), a G2Prepared batch passes through"""
    if isinstance(b, pairing.G2Prepared):
        return b
    if isinstance(b, (list, tuple)) and any(isinstance(x, pairing.G2Prepared) for x in b):
        # a per-pair list of affine points and prepared values (verifier.rs:69-76 builds exactly that): kept as a list, the affine members of
        # everything queued are prepared together in verify() / by the Miller loop
        return _Mixed(x if isinstance(x, pairing.G2Prepared) else np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 24) for x in b)
    return np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 24)


class _Mixed(list):
    """operands of one equation, some affine (n, 24) arrays, some G2Prepared batches; len() = number of pairs"""
    def __len__(self):
        return sum(len(x) for x in list.__iter__(self))


def _g2_all(items):
    """the queued G2 operands as ONE operand of multi_miller_loop: a plain concatenation while nothing is prepared, else G2Prepared"""
    flat = []
    for b in items:
        flat.extend(list.__iter__(b)) if isinstance(b, _Mixed) else flat.append(b)
    if any(isinstance(b, pairing.G2Prepared) for b in flat):
        return flat                                   # a mixed list: multi_miller_loop sends both kinds to the device in one call
    return np.concatenate(flat)


class RandomizedPairingChecker:
    def __init__(self, random, lazy):                          # new(random, lazy)  :44-53
        self.left = fp12_one()
        self.right = fp12_one()                                # PairingOutput::zero() == Fp12 one
        self.lazy = bool(lazy)
        self.pending = ([], [])
        self.pending_targets = []
        self.random = random % R_MOD
        self.current_random = 1

    # -- single equations -------------------------------------------------------------------------------------
    def add_sources_and_target(self, a, b, out):               # e(a, b) == out   :61-77
        self.add_multiple_sources_and_target(np.asarray(a).reshape(1, 12), _g2(b), out)

    def add_sources(self, a, b, c, d):                         # e(a, b) == e(c, d)   :104-113
        self.add_multiple_sources(np.asarray(a).reshape(1, 12), _g2(b), np.asarray(c).reshape(1, 12), _g2(d))

    # -- products ---------------------------------------------------------------------------------------------
    def add_multiple_sources_and_target(self, a, b, out, lazy=None):    # prod e(a_i, b_i) == out   :116-138
        lazy = self.lazy if lazy is None else lazy
        m = self.current_random
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 12)
        b = _g2(b)
        if len(a) != len(b):
            raise DockGpuError(-7, "zip_eq")
        if lazy:
            self._queue(a, m, False, b)
        else:
            a_m, inf = g1_scale(a, m)
            self.left = fp12_mul(self.left, pairing.multi_miller_loop(a_m, b, inf))
        out = np.ascontiguousarray(out, dtype=np.uint64).reshape(72)
        if not (out == fp12_one()).all():                 # a target of one (e(..) e(..) = 1, the KZG checks) contributes nothing
            if lazy:
                self.pending_targets.append((out, m))     # right += out.mul_bigint(m): all of them as ONE multi-exponentiation in verify()
            else:
                self.right = fp12_mul(self.right, fp12_pow(out, m))
        self.current_random = self.current_random * self.random % R_MOD

    def add_multiple_sources(self, a, b, c, d, lazy=None):              # prod e(a_i, b_i) == prod e(c_i, d_i)   :142-173
        lazy = self.lazy if lazy is None else lazy
        m = self.current_random
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 12)
        c = np.ascontiguousarray(c, dtype=np.uint64).reshape(-1, 12)
        b, d = _g2(b), _g2(d)
        if len(a) != len(b) or len(c) != len(d):
            raise DockGpuError(-7, "zip_eq")
        if lazy:
            self._queue(a, m, False, b); self._queue(c, m, True, d)
        else:
            a_m, ainf = g1_scale(a, m)
            c_m, cinf = g1_scale(c, m, negate=True)
            self.left = fp12_mul(self.left, pairing.multi_miller_loop(a_m, b, ainf))
            self.left = fp12_mul(self.left, pairing.multi_miller_loop(c_m, d, cinf))
        self.current_random = self.current_random * self.random % R_MOD

    def _queue(self, a, m, negate, b):
        """lazy mode: the `a.mul_bigint(m)` scalings of every queued equation run as ONE batched launch in verify() (each
        call is latency-bound on its own); the pairs and their scalars are what the reference would have pushed to `pending`"""
        self.pending[0].append((a, m, negate)); self.pending[1].append(b)

    def verify(self):                                                   # :204-214
        left = self.left
        if self.pending_targets:
            self.right = fp12_mul(self.right, fp12_multi_pow([o for o, _ in self.pending_targets], [m for _, m in self.pending_targets]))
            self.pending_targets = []
        if self.pending[0]:
            pts = np.concatenate([a for a, _, _ in self.pending[0]])
            sc = np.concatenate([np.tile(_limbs(m), (len(a), 1)) for a, m, _ in self.pending[0]])
            ng = np.concatenate([np.full(len(a), 1 if neg else 0, dtype=np.uint8) for a, _, neg in self.pending[0]])
            qs = _g2_all(self.pending[1])
            if isinstance(qs, np.ndarray):
                # every G2 operand affine: the scalings and the Miller loop as ONE call (dgpu_multi_miller_loop_scaled: the scaling chains run beside
                # the line chain of the G2 members); a negated source is scaled by r - m
                ms = np.concatenate([np.tile(_limbs((R_MOD - m) % R_MOD if neg else m), (len(a), 1)) for a, m, neg in self.pending[0]])
                left = fp12_mul(pairing.multi_miller_loop_scaled(pts, ms, qs), left)
            else:
                ps, _ = g1_scale_each(pts, sc, ng)
                left = fp12_mul(pairing.multi_miller_loop(ps, qs), left)    # identity members are all-zero words: skipped on the device
        gt = pairing.final_exponentiation(left)
        if gt is None:
            raise ValueError("final_exponentiation of zero")           # arkworks: .unwrap() panics
        return bool((gt == self.right).all())
