"""Fiat-Shamir transcript of the aggregation protocol: `dock_crypto_utils::transcript::MerlinTranscript`
(/root/reference/utils/src/transcript.rs:16-170) over the vendored Merlin (/root/reference/merlin/src/transcript.rs:74-215:
`new`, `append_message`, `challenge_bytes`) and its STROBE-128 subset (/root/reference/merlin/src/strobe.rs:60-190:
meta-AD, AD, PRF over Keccak-f[1600], rate 166).  Host logic: a few hundred bytes are hashed per GIPA round.

Keccak-f[1600] is pinned against hashlib's SHA3-256 (same permutation) and the Merlin framing against the
published Merlin "simple transcript" vector in tests/test_transcript.py.
"""

_MASK = (1 << 64) - 1
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
    0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
    0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]   # [x][y]


def _rol(v, n):
    n %= 64
    return ((v << n) | (v >> (64 - n))) & _MASK if n else v


def keccak_f1600_py(state):
    """state: bytearray(200), permuted in place (pure-Python statement of the permutation; the transcript uses keccak_f1600.c when built)"""
    a = [[int.from_bytes(state[8 * (x + 5 * y):8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y):8 * (x + 5 * y) + 8] = (a[x][y] & _MASK).to_bytes(8, "little")


_HELPER = None


def build_helper():
    """gcc keccak_f1600.c -> libkeccak_f1600.so next to this file (called by __graft_entry__.build())"""
    import os, subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so, srcs = os.path.join(here, "libkeccak_f1600.so"), [os.path.join(here, "keccak_f1600.c"), os.path.join(here, "merlin_native.c")]
    if not os.path.exists(so) or any(os.path.getmtime(src) > os.path.getmtime(so) for src in srcs):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-fvisibility=hidden", "-o", so] + srcs)
    return so


def _helper():
    global _HELPER
    if _HELPER is None:
        import ctypes, os
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libkeccak_f1600.so")
        try:
            _HELPER = ctypes.CDLL(so).keccak_f1600
            _HELPER.argtypes = [ctypes.c_void_p]
        except OSError:
            _HELPER = False
    return _HELPER


def keccak_f1600(state):
    """state: bytearray(200), permuted in place.  Host-side helper of this Python mirror (keccak_f1600.c; the pure-Python
    statement above when the helper was not built).  Not part of the product ABI: Merlin is out of scope for the library."""
    f = _helper()
    if not f:
        return keccak_f1600_py(state)
    import ctypes
    buf = (ctypes.c_uint8 * 200).from_buffer(state)
    if f(ctypes.cast(buf, ctypes.c_void_p)):
        raise RuntimeError("keccak_f1600 helper failed")


STROBE_R = 166
FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32


class Strobe128:
    def __init__(self, protocol_label):
        st = bytearray(200)
        st[0:6] = bytes([1, STROBE_R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.state, self.pos, self.pos_begin, self.cur_flags = st, 0, 0, 0
        self.meta_ad(protocol_label, False)

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[STROBE_R + 1] ^= 0x80
        keccak_f1600(self.state)
        self.pos = self.pos_begin = 0

    def _absorb(self, data):
        data = bytes(data)
        off = 0
        while off < len(data):
            k = min(STROBE_R - self.pos, len(data) - off)
            chunk = int.from_bytes(self.state[self.pos:self.pos + k], "little") ^ int.from_bytes(data[off:off + k], "little")
            self.state[self.pos:self.pos + k] = chunk.to_bytes(k, "little")
            self.pos += k; off += k
            if self.pos == STROBE_R:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray(n)
        for i in range(n):
            out[i] = self.state[self.pos]
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        assert flags & FLAG_T == 0
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if flags & (FLAG_C | FLAG_K) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(FLAG_M | FLAG_A, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(FLAG_A, more)
        self._absorb(data)

    def prf(self, n, more):
        self._begin_op(FLAG_I | FLAG_A | FLAG_C, more)
        return self._squeeze(n)


class Merlin:
    def __init__(self, label):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label, message):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(len(message).to_bytes(4, "little"), True)
        self.strobe.ad(message, False)

    def challenge_bytes(self, label, n):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(n.to_bytes(4, "little"), True)
        return self.strobe.prf(n, False)


R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


class MerlinTranscript:
    """utils/src/transcript.rs:24-41,64-170.  `append` takes the element's `serialize_compressed` bytes (serialization is
    the caller's, crypto_amd/aggregation/ser.py)."""

    def __init__(self, label):
        self.merlin = Merlin(label)

    def append(self, label, element_bytes):
        self.merlin.append_message(label, element_bytes)

    append_message = append

    def challenge_bytes(self, label, n):
        return self.merlin.challenge_bytes(label, n)

    def challenge_scalar(self, label):
        """transcript.rs:103-122: 64 PRF bytes -> Fr::from_random_bytes (first 32 bytes little-endian, top bit shaved, must be
        < r) -> returns the INVERSE of the sampled element; resamples on None / zero."""
        buf = self.merlin.challenge_bytes(label, 64)
        while True:
            v = int.from_bytes(buf[:32], "little") & ((1 << 255) - 1)
            if v < R_MOD and v != 0:
                return pow(v, R_MOD - 2, R_MOD)
            buf = self.merlin.challenge_bytes(label, 64)



class NativeMerlinTranscript:
    """The same transcript in C (merlin_native.c, in the helper library next to keccak_f1600.c): `struct` is a `dgpu_transcript` whose two callbacks are
    plain C functions — what a Rust host's merlin::Transcript costs the library (microseconds), with no interpreter in the loop.  Test / bench helper,
    byte for byte the transcript above (tests/test_transcript.py)."""

    def __init__(self, label, _handle=None):
        import ctypes as C
        import os
        from .._native import Transcript, APPEND_FN, CHALLENGE_FN
        H = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libkeccak_f1600.so"))
        H.mt_new.restype = C.c_void_p; H.mt_new.argtypes = [C.c_char_p, C.c_size_t]
        H.mt_clone.restype = C.c_void_p; H.mt_clone.argtypes = [C.c_void_p]
        H.mt_free.argtypes = [C.c_void_p]
        H.mt_append_message.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        H.mt_challenge_scalar.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p]
        H.mt_challenge_bytes.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        H.mt_state.argtypes = [C.c_void_p, C.c_void_p]
        self._H, self._C = H, C
        self.ctx = _handle if _handle is not None else H.mt_new(bytes(label), len(label))
        if not self.ctx:
            raise MemoryError("mt_new")
        self.struct = Transcript(self.ctx, C.cast(H.mt_append_message, APPEND_FN), C.cast(H.mt_challenge_scalar, CHALLENGE_FN))

    def clone(self):
        h = self._H.mt_clone(self.ctx)
        if not h:                         # (ctypes turns NULL into None, which __init__ would read as "no handle given": a fresh, empty-label transcript)
            raise MemoryError("mt_clone")
        return NativeMerlinTranscript(b"", _handle=h)

    def append(self, label, element_bytes):
        self._H.mt_append_message(self.ctx, bytes(label), len(label), bytes(element_bytes), len(element_bytes))

    append_message = append

    def challenge_bytes(self, label, n):
        buf = self._C.create_string_buffer(n)
        self._H.mt_challenge_bytes(self.ctx, bytes(label), len(label), buf, n)
        return buf.raw

    def challenge_scalar(self, label):
        out = (self._C.c_uint64 * 4)()
        self._H.mt_challenge_scalar(self.ctx, bytes(label), len(label), out)
        return sum(int(out[i]) << (64 * i) for i in range(4))

    def state(self):
        buf = self._C.create_string_buffer(203)
        self._H.mt_state(self.ctx, buf)
        return buf.raw

    def __del__(self):
        try:
            if self.ctx:
                self._H.mt_free(self.ctx); self.ctx = None
        except Exception:      # noqa: BLE001
            pass
