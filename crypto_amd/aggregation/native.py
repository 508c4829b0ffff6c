"""The aggregation as the library runs it — dgpu_snarkpack_aggregate / dgpu_snarkpack_verify (crypto_amd/csrc/dock_aggregation.cpp: the protocol of
groth16.py / kzg.py in C++ inside libdock_gpu.so, host threads around the same device calls) — behind the call surface of this package:
same SRS objects, same proof dictionaries, same transcript objects (the library calls back into `transcript.append` / `challenge_scalar`,
as it calls the Rust caller's `impl Transcript`).  tests/test_gpu_aggregation_native.py compares the two element by element.

Reference: /root/reference/legogroth16/src/aggregation/groth16/{prover.rs:47-147, verifier.rs:36-100}, legogroth16/{prover.rs:38-127,
verifier.rs:34-96, using_groth16.rs:26-128}."""
import ctypes as C
import time
import numpy as np
from .._native import lib, DockGpuError, Transcript, APPEND_FN, CHALLENGE_FN, SnarkpackProverSrs, SnarkpackVerifierSrs, Groth16Vk
from . import ops
from .ops import G1, G2, R_MOD
from .srs import PairCommitment, AggregationError

VALIDATE_GT = 1
VALIDATE_POINTS = 2
LAST = {"transcript_ms": 0.0}          # of the last call: milliseconds spent inside the Python transcript's callbacks (bench.py reports it)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, w):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, w)


class _Callbacks:
    """dgpu_transcript over a Python transcript object; an exception raised by the transcript is kept and re-raised after the call"""

    def __init__(self, transcript):
        self.error = None
        self.seconds = 0.0               # time spent inside the Python transcript (a Rust caller's Merlin costs microseconds; this one does not)

        def append(_ctx, label, label_len, data, n):
            t0 = time.perf_counter()
            try:
                transcript.append(C.string_at(label, label_len), C.string_at(data, n))
            except BaseException as e:       # noqa: BLE001 (must not unwind into C)
                self.error = self.error or e
            self.seconds += time.perf_counter() - t0

        def challenge(_ctx, label, label_len, out):
            t0 = time.perf_counter()
            try:
                v = int(transcript.challenge_scalar(C.string_at(label, label_len))) % R_MOD
            except BaseException as e:       # noqa: BLE001
                self.error = self.error or e
                v = 1
            for i in range(4):
                out[i] = (v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
            self.seconds += time.perf_counter() - t0
        self._a, self._c = APPEND_FN(append), CHALLENGE_FN(challenge)
        self.struct = Transcript(None, self._a, self._c)

    def check(self):
        if self.error is not None:
            raise self.error


class _NativeCallbacks:
    """a transcript that already IS a dgpu_transcript with C callbacks (transcript.NativeMerlinTranscript): nothing of this process's interpreter runs
    inside the library call"""

    def __init__(self, transcript):
        self.struct, self.seconds, self._keep = transcript.struct, 0.0, transcript

    def check(self):
        pass


def _callbacks(transcript):
    return _NativeCallbacks(transcript) if hasattr(transcript, "struct") else _Callbacks(transcript)


# ---- the flat proof (include/dock_gpu.h: the word layout) <-> the dictionaries of groth16.py ------------------------------------------------
def proof_to_words(proof):
    names = [k for k in ("c", "d") if "com_" + k in proof]
    gipa = proof["tmipp"]["gipa"]
    out = [np.array([gipa["nproofs"], len(names)], dtype=np.uint64)]
    put = lambda a: out.append(np.asarray(a, dtype=np.uint64).reshape(-1))
    pc = lambda c: (put(c.t), put(c.u))
    pc(proof["com_ab"])
    for k in names:
        pc(proof["com_" + k])
    put(proof["z_ab"])
    for k in names:
        put(proof["z_" + k])
    for l, r in gipa["comms_ab"]:
        pc(l); pc(r)
    for k in names:
        for l, r in gipa["comms_" + k]:
            pc(l); pc(r)
    for l, r in gipa["z_ab"]:
        put(l); put(r)
    for k in names:
        for l, r in gipa["z_" + k]:
            put(l); put(r)
    put(gipa["final_a"]); put(gipa["final_b"])
    for k in names:
        put(gipa["final_" + k])
    for key in ("final_vkey", "final_wkey"):
        put(gipa[key][0]); put(gipa[key][1])
    for key in ("vkey_opening", "wkey_opening"):
        put(proof["tmipp"][key][0]); put(proof["tmipp"][key][1])
    return np.ascontiguousarray(np.concatenate(out))


def proof_from_words(words):
    w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
    n, nm = int(w[0]), int(w[1])
    if nm not in (1, 2) or n < 2 or n & (n - 1) or len(w) != lib().dgpu_snarkpack_proof_words(n, nm - 1):
        raise AggregationError("malformed proof words")
    names = ("c", "d")[:nm]
    L = n.bit_length() - 1
    pos = [2]

    def get(k):
        a = w[pos[0]:pos[0] + k].copy(); pos[0] += k
        return a
    pc = lambda: PairCommitment(get(72), get(72))
    proof, gipa = {}, {"nproofs": n}
    proof["com_ab"] = pc()
    for k in names:
        proof["com_" + k] = pc()
    proof["z_ab"] = get(72)
    for k in names:
        proof["z_" + k] = get(12)
    gipa["comms_ab"] = [(pc(), pc()) for _ in range(L)]
    for k in names:
        gipa["comms_" + k] = [(pc(), pc()) for _ in range(L)]
    gipa["z_ab"] = [(get(72), get(72)) for _ in range(L)]
    for k in names:
        gipa["z_" + k] = [(get(12), get(12)) for _ in range(L)]
    gipa["final_a"], gipa["final_b"] = get(12), get(24)
    for k in names:
        gipa["final_" + k] = get(12)
    gipa["final_vkey"] = (get(24), get(24)); gipa["final_wkey"] = (get(12), get(12))
    proof["tmipp"] = {"gipa": gipa, "vkey_opening": (get(24), get(24)), "wkey_opening": (get(12), get(12))}
    return proof


def aggregate_proofs_words(srs, transcript, proofs, with_d=False):
    """the aggregate proof as the ABI's flat words"""
    n = len(proofs)
    if n < 2:
        raise AggregationError("invalid proof size < 2")
    if n & (n - 1):
        raise AggregationError("invalid proof size: not power of two")
    if not srs.has_correct_len(n):
        raise AggregationError("SRS len %d != proofs len %d" % (len(srs.vkey), n))
    a = _c(np.stack([p["a"] for p in proofs]), 12); b = _c(np.stack([p["b"] for p in proofs]), 24); c = _c(np.stack([p["c"] for p in proofs]), 12)
    d = _c(np.stack([p["d"] for p in proofs]), 12) if with_d else None
    keep = [_c(srs.g_alpha_powers_table, 12), _c(srs.g_beta_powers_table, 12), _c(srs.h_alpha_powers_table, 24), _c(srs.h_beta_powers_table, 24),
            _c(srs.vkey.a, 24), _c(srs.vkey.b, 24), _c(srs.wkey.a, 12), _c(srs.wkey.b, 12)]
    if len(keep[0]) != 2 * n or len(keep[1]) != 2 * n or len(keep[2]) != n or len(keep[3]) != n:
        raise AggregationError("SRS tables do not match the number of proofs")
    S = SnarkpackProverSrs(n, *[x.ctypes.data for x in keep])
    cap = lib().dgpu_snarkpack_proof_words(n, int(with_d))
    out = np.zeros(cap, dtype=np.uint64)
    ln = C.c_size_t(0)
    cb = _callbacks(transcript)
    rc = lib().dgpu_snarkpack_aggregate(C.byref(S), _p(a), _p(b), _p(c), _p(d), n, C.byref(cb.struct), _p(out), cap, C.byref(ln))
    cb.check()
    LAST["transcript_ms"] = cb.seconds * 1e3
    if rc:
        raise DockGpuError(rc, "dgpu_snarkpack_aggregate")
    return out[:ln.value]


def aggregate_proofs(srs, transcript, proofs, with_d=False):
    """groth16.aggregate_proofs' result (the same dictionary), computed by the library"""
    return proof_from_words(aggregate_proofs_words(srs, transcript, proofs, with_d))


def verify_aggregate_proof(ip_verifier_srs, pvk, public_inputs, proof, random, transcript, with_d=False, d=None, validate_gt=True, validate_points=True):
    """groth16.verify_aggregate_proof / using_groth16.verify_aggregate_proof (d = the list of commitments): raises AggregationError on an
    invalid proof.  `proof`: the dictionary or the flat words.  validate_gt / validate_points: the two halves of `Validate::Yes` for a proof that
    arrives from an untrusted source as raw words (GT members of order r; G1 / G2 members on their curve and in the prime-order subgroup)."""
    vk = pvk["vk"]
    words = proof if isinstance(proof, np.ndarray) else proof_to_words(proof)
    words = np.ascontiguousarray(words, dtype=np.uint64)
    if not public_inputs or any(len(pub) != len(public_inputs[0]) for pub in public_inputs):
        raise AggregationError("public inputs of unequal length")
    l = len(public_inputs[0])
    pub = ops.limbs([x for row in public_inputs for x in row]) if l else np.zeros((0, 4), dtype=np.uint64)
    pub = np.ascontiguousarray(pub)
    v = ip_verifier_srs
    keep = [_c(v.g, 12), _c(v.h, 24), _c(v.g_alpha, 12), _c(v.g_beta, 12), _c(v.h_alpha, 24), _c(v.h_beta, 24)]
    S = SnarkpackVerifierSrs(v.n, *[x.ctypes.data for x in keep])
    kk = [_c(vk.alpha_g1, 12), _c(vk.beta_g2, 24), _c(vk.gamma_g2, 24), _c(vk.delta_g2, 24), _c(vk.gamma_abc_g1, 12)]
    K = Groth16Vk(*[x.ctypes.data for x in kk], len(kk[4]))
    variant = 2 if d is not None else (1 if with_d else 0)
    dl = None if d is None else _c(d, 12)
    rnd = ops.limbs([random]).reshape(4)
    ok = C.c_int32(0)
    cb = _callbacks(transcript)
    rc = lib().dgpu_snarkpack_verify(C.byref(S), C.byref(K), _p(pub) if l else None, len(public_inputs), l, _p(words), len(words), variant, _p(dl), _p(rnd), C.byref(cb.struct),
                                     (VALIDATE_GT if validate_gt else 0) | (VALIDATE_POINTS if validate_points else 0), C.byref(ok))
    cb.check()
    LAST["transcript_ms"] = cb.seconds * 1e3
    if rc == -3:
        raise AggregationError("malformed proof, key or public inputs (DGPU_E_BADARG)")
    if rc:
        raise DockGpuError(rc, "dgpu_snarkpack_verify")
    if not ok.value:
        raise AggregationError("Proof Verification Failed")
