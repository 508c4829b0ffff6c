"""Host-side mirror of the reference's pairing call surface over the C ABI.

  * E::multi_miller_loop(a, b)   -> multi_miller_loop()   (utils/src/randomized_pairing_check.rs:207, legogroth16/src/verifier.rs:69-76)
  * E::final_exponentiation(f)   -> final_exponentiation() returns None where arkworks returns None
  * E::multi_pairing(a, b)       -> multi_pairing()       (bbs_plus/src/signature.rs:284, legogroth16/src/link/snark.rs:157)
  * E::G2Prepared::from(q)       -> G2Prepared.from_affine(qs) / g2_prepare(qs)   (verifier.rs:22-23, randomized_pairing_check.rs:132)
    `b: impl Into<E::G2Prepared>`: every entry point below takes affine points (n, 24), a `G2Prepared` batch, or a list mixing both,
    the way the reference mixes `b.into()` with `pvk.delta_g2_neg_pc.clone()` (verifier.rs:69-76).
Arrays: numpy uint64 in the ABI layout (G1 affine 12, G2 affine 24, Fp12 72 limbs; Montgomery).
"""
import ctypes as C
import numpy as np
from ._native import lib, DockGpuError
from .msm import _ensure


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


PREPARED_WORDS = 68 * 36      # DGPU_G2_PREPARED_WORDS
FP_ONE_MONT = np.array([0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745, 0x5c071a97a256ec6d, 0x15f65ec3fa80e493], dtype=np.uint64)   # 1 in Fq Montgomery limbs


class G2Prepared:
    """A batch of ark-ec `G2Prepared` values: `coeffs` (n, 68*36) uint64 = the ell_coeffs triples in ABI limbs, `infinity` (n,) uint8."""

    def __init__(self, coeffs, infinity):
        self.coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, PREPARED_WORDS)
        self.infinity = np.ascontiguousarray(infinity, dtype=np.uint8).reshape(-1)
        assert len(self.coeffs) == len(self.infinity)

    def __len__(self):
        return len(self.coeffs)

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            if not -len(self) <= i < len(self):
                raise IndexError("G2Prepared index out of range")
            i = int(i) % len(self)
            return G2Prepared(self.coeffs[i:i + 1], self.infinity[i:i + 1])
        return G2Prepared(self.coeffs[i], self.infinity[i])

    def __array__(self, *args, **kwargs):
        # numpy would otherwise walk the batch as a sequence of sequences of ... (an element of a batch is a batch): converting a list that
        # mixes affine arrays and prepared batches must go through G2Prepared.concat, not through np.asarray
        raise TypeError("G2Prepared is not an array: use G2Prepared.concat([...]) for mixed operand lists")

    @classmethod
    def from_affine(cls, qs, is_inf=None):
        """G2Prepared::from for every point of `qs` (n, 24), on the device (dgpu_g2_prepare)"""
        _ensure()
        qs = np.ascontiguousarray(qs, dtype=np.uint64).reshape(-1, 24)
        n = len(qs)
        co = np.zeros((n, PREPARED_WORDS), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        fl = None if is_inf is None else np.ascontiguousarray(is_inf, dtype=np.uint8)
        if n:
            rc = lib().dgpu_g2_prepare(_p(qs), _p(fl), n, _p(co), _p(inf))
            if rc:
                raise DockGpuError(rc, "dgpu_g2_prepare")
        return cls(co, inf)

    @classmethod
    def concat(cls, items):
        """items: G2Prepared batches and/or affine arrays — `impl Into<E::G2Prepared>` for each"""
        items = list(items)
        aff = [i for i, it in enumerate(items) if not isinstance(it, G2Prepared)]
        if aff:                             # every affine operand of the list in ONE dgpu_g2_prepare call (a batch verifier queues one per proof)
            arrs = [np.ascontiguousarray(items[i], dtype=np.uint64).reshape(-1, 24) for i in aff]
            prep = cls.from_affine(np.concatenate(arrs))
            at = 0
            for i, a in zip(aff, arrs):
                items[i] = prep[at:at + len(a)]; at += len(a)
        return cls(np.concatenate([p.coeffs for p in items]), np.concatenate([p.infinity for p in items]))


def g2_prepare(qs, is_inf=None):
    return G2Prepared.from_affine(qs, is_inf)


def _mixed_miller_loop(ps, items, skip):
    items = [q if isinstance(q, G2Prepared) else np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, 24) for q in items]
    if len(ps) != sum(len(q) for q in items):
        raise DockGpuError(-7, "multi_miller_loop")
    sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.uint8)
    pa, qa, ska, pp, cp, skp = [], [], [], [], [], []
    at = 0
    for q in items:
        k = len(q)
        if isinstance(q, G2Prepared):
            pp.append(ps[at:at + k]); cp.append(q.coeffs); skp.append(q.infinity if sk is None else (sk[at:at + k] | q.infinity))
        else:
            pa.append(ps[at:at + k]); qa.append(q); ska.append(np.zeros(k, np.uint8) if sk is None else sk[at:at + k])
        at += k
    cat = lambda xs, w, dt=np.uint64: np.ascontiguousarray(np.concatenate(xs)) if xs else np.zeros((0, w), dt)
    pa, qa, pp, cp = cat(pa, 12), cat(qa, 24), cat(pp, 12), cat(cp, PREPARED_WORDS)
    ska = np.ascontiguousarray(np.concatenate(ska)) if ska else np.zeros(0, np.uint8)
    skp = np.ascontiguousarray(np.concatenate(skp)) if skp else np.zeros(0, np.uint8)
    out = np.zeros(72, dtype=np.uint64)
    rc = lib().dgpu_multi_miller_loop_mixed(_p(pa), _p(qa), _p(ska), len(pa), _p(pp), _p(cp), _p(skp), len(pp), _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_multi_miller_loop_mixed")
    return out


def multi_miller_loop(ps, qs, skip=None):
    _ensure()
    ps = np.ascontiguousarray(ps, dtype=np.uint64).reshape(-1, 12)
    if isinstance(qs, (list, tuple)) and any(isinstance(q, G2Prepared) for q in qs):
        # `impl Into<G2Prepared>` operands of both kinds: the affine ones and the prepared ones go to the device as they are, in one call
        # (dgpu_multi_miller_loop_mixed: the product does not depend on the order of the pairs)
        if not all(isinstance(q, G2Prepared) for q in qs):
            return _mixed_miller_loop(ps, list(qs), skip)
        qs = G2Prepared.concat(qs)
    if isinstance(qs, G2Prepared):
        if len(ps) != len(qs):
            raise DockGpuError(-7, "multi_miller_loop")
        sk = qs.infinity if skip is None else (np.ascontiguousarray(skip, dtype=np.uint8) | qs.infinity)
        out = np.zeros(72, dtype=np.uint64)
        rc = lib().dgpu_multi_miller_loop_prepared(_p(ps), _p(qs.coeffs), _p(np.ascontiguousarray(sk)), len(ps), _p(out))
        if rc:
            raise DockGpuError(rc, "dgpu_multi_miller_loop_prepared")
        return out
    qs = np.ascontiguousarray(qs, dtype=np.uint64).reshape(-1, 24)
    if len(ps) != len(qs):
        # arkworks zips with zip_eq and panics; the Rust shim would panic too — here: DGPU_E_LENGTH
        raise DockGpuError(-7, "multi_miller_loop")
    sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.uint8)
    out = np.zeros(72, dtype=np.uint64)
    rc = lib().dgpu_multi_miller_loop(_p(ps), _p(qs), _p(sk), len(ps), _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_multi_miller_loop")
    return out


def multi_miller_loop_scaled(ps, scalars, qs, skip=None, prepared_ps=None, prepared=None):
    """prod e([m_i] P_i, Q_i) (x prod e(P'_j, prepared_j)) in ONE call (dgpu_multi_miller_loop_scaled): the scalings of RandomizedPairingChecker
    (utils/src/randomized_pairing_check.rs:125-134) run beside the chain of the Q_i instead of in front of it.  scalars: one row of four canonical
    words per pair, or a single scalar (an int or one row) for every pair; prepared: a G2Prepared for the pairs with prepared_ps."""
    _ensure()
    ps = np.ascontiguousarray(ps, dtype=np.uint64).reshape(-1, 12)
    qs = np.ascontiguousarray(qs, dtype=np.uint64).reshape(-1, 24)
    if len(ps) != len(qs):
        raise DockGpuError(-7, "multi_miller_loop_scaled")
    if isinstance(scalars, int):
        scalars = np.array([(scalars >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)], dtype=np.uint64)
    sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    if len(sc) not in (1, len(ps)) and len(ps):
        raise DockGpuError(-7, "multi_miller_loop_scaled")
    stride = 4 if len(sc) == len(ps) else 0
    sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.uint8)
    pp = co = skp = None
    n_prep = 0
    if prepared is not None:
        pp = np.ascontiguousarray(prepared_ps, dtype=np.uint64).reshape(-1, 12)
        if len(pp) != len(prepared):
            raise DockGpuError(-7, "multi_miller_loop_scaled")
        co, skp, n_prep = prepared.coeffs, np.ascontiguousarray(prepared.infinity), len(pp)
    out = np.zeros(72, dtype=np.uint64)
    rc = lib().dgpu_multi_miller_loop_scaled(_p(ps), _p(sc), stride, _p(qs), _p(sk), len(ps), _p(pp), _p(co), _p(skp), n_prep, _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_multi_miller_loop_scaled")
    return out


def multi_miller_loop_sharded(ps, qs, skip=None, ngpus=0):
    """multi_miller_loop with the pairs chunked over the process's device contexts (dgpu_multi_miller_loop_sharded)"""
    _ensure()
    ps = np.ascontiguousarray(ps, dtype=np.uint64).reshape(-1, 12)
    qs = np.ascontiguousarray(qs, dtype=np.uint64).reshape(-1, 24)
    if len(ps) != len(qs):
        raise DockGpuError(-7, "multi_miller_loop")
    sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.uint8)
    out = np.zeros(72, dtype=np.uint64)
    rc = lib().dgpu_multi_miller_loop_sharded(_p(ps), _p(qs), _p(sk), len(ps), ngpus, _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_multi_miller_loop_sharded")
    return out


def multi_miller_loops(jobs, final_exp=False):
    """[(ps, qs), ...] -> the raw Miller output of every job, all of them in ONE call (dgpu_multi_miller_loop_segments): the
    mutually independent `multi_pairing`s the aggregation issues one after another (legogroth16/src/aggregation/commitment.rs:30-31,54-67).
    final_exp: the GT elements instead (dgpu_multi_pairing_segments)."""
    _ensure()
    P = [np.ascontiguousarray(ps, dtype=np.uint64).reshape(-1, 12) for ps, _ in jobs]
    Q = [np.ascontiguousarray(qs, dtype=np.uint64).reshape(-1, 24) for _, qs in jobs]
    if any(len(a) != len(b) for a, b in zip(P, Q)):
        raise DockGpuError(-7, "multi_miller_loops")
    if not jobs:
        return []
    ends = np.cumsum([len(a) for a in P]).astype(np.uint64)
    pa, qa = np.ascontiguousarray(np.concatenate(P)), np.ascontiguousarray(np.concatenate(Q))
    out = np.zeros((len(jobs), 72), dtype=np.uint64)
    fn = lib().dgpu_multi_pairing_segments if final_exp else lib().dgpu_multi_miller_loop_segments
    rc = fn(_p(pa), _p(qa), None, len(pa), _p(ends), len(jobs), _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_multi_pairing_segments" if final_exp else "dgpu_multi_miller_loop_segments")
    return list(out)


def multi_pairings(jobs):
    return multi_miller_loops(jobs, final_exp=True)


def final_exponentiation(f):
    f = np.ascontiguousarray(f, dtype=np.uint64)
    out = np.zeros(72, dtype=np.uint64)
    rc = lib().dgpu_final_exponentiation(_p(f), _p(out))
    if rc == -5:
        return None
    if rc:
        raise DockGpuError(rc, "dgpu_final_exponentiation")
    return out


def multi_pairing(ps, qs, skip=None):
    return final_exponentiation(multi_miller_loop(ps, qs, skip))
