"""Host-side mirror of the reference's variable-base MSM call surface, over the C ABI.

Mirrors (names, argument meaning, error behaviour):
  * ark_ec::VariableBaseMSM::msm(bases, scalars)          -> msm()            Err(min_len) on length mismatch
  * ark_ec::VariableBaseMSM::msm_unchecked(bases, &[Fr])  -> msm_unchecked()  Montgomery scalars, truncates to min(len)
  * ark_ec::VariableBaseMSM::msm_bigint(bases, &[BigInt]) -> msm_bigint()     canonical scalars, truncates to min(len)
  * dock_crypto_utils::pairs::Pairs::{msm, msm_bigint}    -> Pairs            /root/reference/utils/src/pairs.rs:143-156
Arrays are numpy uint64 in the ABI layout of include/dock_gpu.h (ark-ff Montgomery limbs).
"""
import ctypes as C
import numpy as np
from . import _native
from ._native import lib, DockGpuError


class _Curve:
    def __init__(self, tag, aff_words):
        self.tag = tag
        self.AW = aff_words           # u64 per affine point: 12 (G1) / 24 (G2)
        self.JW = aff_words * 3 // 2  # u64 per Jacobian point: 18 / 36

    def fn(self, name):
        return getattr(lib(), name % self.tag)


G1 = _Curve("g1", 12)
G2 = _Curve("g2", 24)

_inited = False


def init(device=0, min_gpu_n=0):
    """dgpu_init(device).  The library's default size threshold (DGPU_DEFAULT_MIN_GPU_N: below it the ABI answers DGPU_E_TOO_SMALL so that
    a Rust caller stays on arkworks) is lowered to `min_gpu_n` = 0 here: this mirror has no CPU path to stay on."""
    global _inited
    rc = lib().dgpu_init(device)
    if rc:
        raise DockGpuError(rc, "dgpu_init(%d)" % device)
    lib().dgpu_set_min_gpu_n(min_gpu_n)
    _native.note_init("init", device, min_gpu_n)
    _inited = True


def init_devices(physical, min_gpu_n=0):
    """several GPUs in this process: context k on HIP device physical[k] (dgpu_init_device_list; a device may repeat)"""
    global _inited
    arr = np.ascontiguousarray(physical, dtype=np.int32)
    rc = lib().dgpu_init_device_list(arr.ctypes.data_as(C.c_void_p), len(arr))
    if rc:
        raise DockGpuError(rc, "dgpu_init_device_list(%s)" % list(arr))
    lib().dgpu_set_min_gpu_n(min_gpu_n)
    _native.note_init("list", [int(x) for x in arr], min_gpu_n)
    _inited = True


def _ensure():
    if not _inited:
        init(0)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _prep(curve, bases, scalars, is_inf):
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, curve.AW)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    n = min(len(bases), len(scalars))     # arkworks truncates (legogroth16/src/prover.rs:286 relies on it)
    inf = None
    if is_inf is not None:
        inf = np.ascontiguousarray(is_inf, dtype=np.uint8)
        if len(inf) < n:
            raise ValueError("is_inf shorter than the truncated length")
    return bases, scalars, inf, n


def msm_bigint(curve, bases, scalars, is_inf=None):
    """G::msm_bigint(bases, bigints): canonical 4x64 scalars; returns the Jacobian triple (normalised)."""
    _ensure()
    bases, scalars, inf, n = _prep(curve, bases, scalars, is_inf)
    out = np.zeros(curve.JW, dtype=np.uint64)
    rc = curve.fn("dgpu_msm_%s")(_p(bases), _p(inf), _p(scalars), n, _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_msm_%s" % curve.tag)
    return out


def msm_unchecked(curve, bases, scalars_mont, is_inf=None):
    """G::msm_unchecked(bases, &[Fr]): scalars are Fr in Montgomery form (R = 2^256)."""
    _ensure()
    bases, scalars, inf, n = _prep(curve, bases, scalars_mont, is_inf)
    out = np.zeros(curve.JW, dtype=np.uint64)
    rc = curve.fn("dgpu_msm_%s_mont")(_p(bases), _p(inf), _p(scalars), n, _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_msm_%s_mont" % curve.tag)
    return out


def affine_struct_dtype(curve):
    """numpy dtype with the layout of ark-ec 0.4's in-memory `Affine<P> { x, y, infinity }` as rustc lays it out for BLS12-381 today
    (x, y, then the bool, padded to the 8-byte alignment of the limbs): 104 bytes for G1, 200 for G2.  The strided entry points take the
    offsets as arguments, so a different field order costs the shim nothing."""
    h = curve.AW // 2
    return np.dtype({"names": ["x", "y", "infinity"], "formats": [(np.uint64, h), (np.uint64, h), np.uint8],
                     "offsets": [0, 8 * h, 16 * h], "itemsize": 16 * h + 8})


def to_affine_structs(curve, bases, is_inf=None):
    """ABI arrays -> an array of `Affine` structs (what a Rust caller's &[G1Affine] looks like in memory)"""
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, curve.AW)
    h = curve.AW // 2
    a = np.zeros(len(bases), dtype=affine_struct_dtype(curve))
    a["x"], a["y"] = bases[:, :h], bases[:, h:]
    if is_inf is not None:
        a["infinity"] = np.asarray(is_inf, dtype=np.uint8)[:len(a)]
    return a


def msm_strided(curve, structs, scalars, montgomery=False):
    """dgpu_msm_*_strided: the MSM straight from an array of `Affine` structs (any numpy structured array with fields x, y and optionally infinity)"""
    _ensure()
    assert structs.flags["C_CONTIGUOUS"]
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    n = min(len(structs), len(scalars))
    f = structs.dtype.fields
    inf_off = f["infinity"][1] if "infinity" in f else (1 << 64) - 1
    out = np.zeros(curve.JW, dtype=np.uint64)
    rc = curve.fn("dgpu_msm_%s_strided")(_p(structs), structs.dtype.itemsize, f["x"][1], f["y"][1], inf_off, _p(scalars), n, int(montgomery), _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_msm_%s_strided" % curve.tag)
    return out


CACHE_VERIFY_FULL = -1     # include/dock_gpu.h DGPU_CACHE_VERIFY_FULL


def bases_cache(bytes=None, min_n=None, verify=None):
    """the resident-bases cache behind the one-shot entry points (include/dock_gpu.h: dgpu_set_bases_cache_bytes / _min_n / _verify); None = leave as it is"""
    _ensure()
    for val, name in ((bytes, "dgpu_set_bases_cache_bytes"), (min_n, "dgpu_set_bases_cache_min_n"), (verify, "dgpu_set_bases_cache_verify")):
        if val is not None:
            rc = getattr(lib(), name)(val)
            if rc:
                raise DockGpuError(rc, name)


def bases_cache_stats():
    """dgpu_bases_cache_stats as a dict"""
    out = np.zeros(8, dtype=np.uint64)
    rc = lib().dgpu_bases_cache_stats(_p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_bases_cache_stats")
    return dict(zip(("hits", "misses", "fills", "stale", "evictions", "bytes", "budget", "entries"), (int(v) for v in out)))


def bases_cache_clear():
    lib().dgpu_bases_cache_clear()


def bases_cache_invalidate(arr):
    """forget every cached entry that overlaps the memory of numpy array `arr`"""
    rc = lib().dgpu_bases_cache_invalidate(_p(arr), arr.nbytes)
    if rc:
        raise DockGpuError(rc, "dgpu_bases_cache_invalidate")


def reserve(curve, n):
    """dgpu_reserve_*: size every slot of the calling thread's context for one-shot MSMs of up to n terms"""
    _ensure()
    rc = curve.fn("dgpu_reserve_%s")(n)
    if rc:
        raise DockGpuError(rc, "dgpu_reserve")


def device_alloc_count():
    return lib().dgpu_device_alloc_count()


def msm(curve, bases, scalars_mont, is_inf=None):
    """G::msm(bases, scalars): checked variant — (False, min_len) on length mismatch, like Err(min_len)."""
    nb = np.asarray(bases).size // curve.AW
    ns = np.asarray(scalars_mont).size // 4
    if nb != ns:
        return False, min(nb, ns)
    return True, msm_unchecked(curve, bases, scalars_mont, is_inf)


class Pairs:
    """utils/src/pairs.rs `Pairs<'_, G, Fr>`: equal-length (left, right) slices."""

    def __init__(self, curve, left, right, is_inf=None):
        self.curve = curve
        self.left = np.ascontiguousarray(left, dtype=np.uint64).reshape(-1, curve.AW)
        self.right = np.ascontiguousarray(right, dtype=np.uint64).reshape(-1, 4)
        if len(self.left) != len(self.right):    # Pairs::new returns None / TryFrom Err((l, r))
            raise ValueError((len(self.left), len(self.right)))
        self.is_inf = is_inf

    def msm(self):           # pairs.rs:145-147
        return msm_unchecked(self.curve, self.left, self.right, self.is_inf)

    def msm_bigint(self):    # pairs.rs:153-155
        return msm_bigint(self.curve, self.left, self.right, self.is_inf)


class OwnedPairs(Pairs):
    """utils/src/owned_pairs.rs `OwnedPairs<G, Fr>` (:21-105): the owning counterpart of `Pairs` — same `msm` (:95-97) and
    `msm_bigint` (:103-105); `split`, `len`, `extend` as in the reference."""

    def __init__(self, curve, left=None, right=None, is_inf=None):
        left = np.zeros((0, curve.AW), dtype=np.uint64) if left is None else left
        right = np.zeros((0, 4), dtype=np.uint64) if right is None else right
        super().__init__(curve, np.array(left, dtype=np.uint64, copy=True), np.array(right, dtype=np.uint64, copy=True), is_inf)

    def split(self):                 # :42-44
        return self.left, self.right

    def as_ref(self):                # :47-49
        return Pairs(self.curve, self.left, self.right, self.is_inf)

    def __len__(self):
        return len(self.left)

    def is_empty(self):
        return len(self.left) == 0

    def extend(self, pairs):         # Extend<(Left, Right)>  :124-137
        for l, r in pairs:
            self.left = np.concatenate([self.left, np.asarray(l, dtype=np.uint64).reshape(1, self.curve.AW)])
            self.right = np.concatenate([self.right, np.asarray(r, dtype=np.uint64).reshape(1, 4)])


class DeviceBases:
    """Device-resident prepared bases (a proving-key query): dgpu_bases_upload_* / dgpu_msm_*_handle."""

    def __init__(self, curve, bases, is_inf=None):
        _ensure()
        self.curve = curve
        bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, curve.AW)
        self.n = len(bases)
        inf = None if is_inf is None else np.ascontiguousarray(is_inf, dtype=np.uint8)
        h = C.c_uint64(0)
        rc = curve.fn("dgpu_bases_upload_%s")(_p(bases), _p(inf), self.n, C.byref(h))
        if rc:
            raise DockGpuError(rc, "dgpu_bases_upload")
        self.handle = h.value

    @classmethod
    def from_structs(cls, curve, structs):
        """dgpu_bases_upload_*_strided: a resident query straight from an array of `Affine` structs"""
        _ensure()
        assert structs.flags["C_CONTIGUOUS"]
        f = structs.dtype.fields
        inf_off = f["infinity"][1] if "infinity" in f else (1 << 64) - 1
        h = C.c_uint64(0)
        rc = curve.fn("dgpu_bases_upload_%s_strided")(_p(structs), structs.dtype.itemsize, f["x"][1], f["y"][1], inf_off, len(structs), C.byref(h))
        if rc:
            raise DockGpuError(rc, "dgpu_bases_upload_strided")
        return cls.from_handle(curve, h.value, len(structs))

    @classmethod
    def from_handle(cls, curve, handle, n):
        """adopt a bases handle produced on the device (WindowTable.multiply_many_to_bases)"""
        self = cls.__new__(cls)
        self.curve, self.handle, self.n = curve, handle, n
        return self


    def precompute(self, window_bits=0):
        """dgpu_bases_precompute_*: in place; later MSMs on this handle run over the precomputed-multiples table"""
        rc = self.curve.fn("dgpu_bases_precompute_%s")(self.handle, window_bits)
        if rc:
            raise DockGpuError(rc, "dgpu_bases_precompute")
        return self

    def table_shape(self):
        """(rows, window bits, windows) of the precomputed table behind the handle, None if it is not one (dgpu_bases_table_shape)"""
        rows, c, w = C.c_size_t(0), C.c_int32(0), C.c_int32(0)
        if not getattr(self, "handle", 0) or lib().dgpu_bases_table_shape(self.handle, C.byref(rows), C.byref(c), C.byref(w)):
            return None
        return rows.value, c.value, w.value

    def same_table_shape(self, other):
        """both handles are precomputed tables of one shape (row count and window width): their MSMs over one scalar vector can share a sort"""
        a = self.table_shape()
        return a is not None and isinstance(other, DeviceBases) and a == other.table_shape()

    def msm_sorted(self, sorted_scalars, row_shift=0):
        """the MSM over a list dgpu_scalars_sort produced for a table of this shape (SortedScalars); row_shift = k: this table is k rows
        shorter than that shape and pairs with the scalars from the k-th on (the l_query against the a_query's list)"""
        out = np.zeros(self.curve.JW, dtype=np.uint64)
        rc = self.curve.fn("dgpu_msm_%s_sorted")(self.handle, sorted_scalars.handle, row_shift, _p(out))
        if rc:
            raise DockGpuError(rc, "dgpu_msm_sorted")
        return out

    def msm_bigint(self, scalars, offset=0, montgomery=False):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        n = min(len(scalars), self.n - offset)
        out = np.zeros(self.curve.JW, dtype=np.uint64)
        rc = self.curve.fn("dgpu_msm_%s_handle")(self.handle, offset, _p(scalars), n, int(montgomery), _p(out))
        if rc:
            raise DockGpuError(rc, "dgpu_msm_handle")
        return out

    def msm_resident(self, dscalars, n=None, base_offset=0, scalar_offset=0):
        if n is None:
            n = min(self.n - base_offset, dscalars.n - scalar_offset)
        out = np.zeros(self.curve.JW, dtype=np.uint64)
        rc = self.curve.fn("dgpu_msm_%s_resident")(self.handle, base_offset, dscalars.handle, scalar_offset, n, _p(out))
        if rc:
            raise DockGpuError(rc, "dgpu_msm_resident")
        return out

    def free(self):
        if self.handle:
            lib().dgpu_bases_free(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def msm_bigint_sharded(curve, bases, scalars, is_inf=None, ngpus=0):
    """one-shot MSM chunked over the process's device contexts (dgpu_msm_*_sharded)"""
    _ensure()
    bases, scalars, inf, n = _prep(curve, bases, scalars, is_inf)
    out = np.zeros(curve.JW, dtype=np.uint64)
    rc = curve.fn("dgpu_msm_%s_sharded")(_p(bases), _p(inf), _p(scalars), n, ngpus, _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_msm_%s_sharded" % curve.tag)
    return out


class ShardedDeviceBases:
    """A proving-key query resident across the process's device contexts (dgpu_bases_upload_*_sharded)."""

    def __init__(self, curve, bases, is_inf=None, ngpus=0):
        _ensure()
        self.curve = curve
        bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, curve.AW)
        self.n = len(bases)
        inf = None if is_inf is None else np.ascontiguousarray(is_inf, dtype=np.uint8)
        h = C.c_uint64(0)
        rc = curve.fn("dgpu_bases_upload_%s_sharded")(_p(bases), _p(inf), self.n, ngpus, C.byref(h))
        if rc:
            raise DockGpuError(rc, "dgpu_bases_upload_sharded")
        self.handle = h.value


    def precompute(self, window_bits=0):
        """dgpu_bases_precompute_*: in place; later MSMs on this handle run over the precomputed-multiples table"""
        rc = self.curve.fn("dgpu_bases_precompute_%s")(self.handle, window_bits)
        if rc:
            raise DockGpuError(rc, "dgpu_bases_precompute")
        return self

    def msm_bigint(self, scalars, montgomery=False):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        n = min(len(scalars), self.n)
        out = np.zeros(self.curve.JW, dtype=np.uint64)
        rc = self.curve.fn("dgpu_msm_%s_sharded_handle")(self.handle, _p(scalars), n, int(montgomery), _p(out))
        if rc:
            raise DockGpuError(rc, "dgpu_msm_sharded_handle")
        return out

    def upload_scalars(self, scalars, montgomery=False):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        h = C.c_uint64(0)
        rc = lib().dgpu_scalars_upload_sharded(_p(scalars), len(scalars), int(montgomery), self.handle, C.byref(h))
        if rc:
            raise DockGpuError(rc, "dgpu_scalars_upload_sharded")
        ds = DeviceScalars.__new__(DeviceScalars)
        ds.n, ds.handle = len(scalars), h.value
        return ds

    def msm_resident(self, dscalars):
        out = np.zeros(self.curve.JW, dtype=np.uint64)
        rc = self.curve.fn("dgpu_msm_%s_sharded_resident")(self.handle, dscalars.handle, _p(out))
        if rc:
            raise DockGpuError(rc, "dgpu_msm_sharded_resident")
        return out

    def free(self):
        if self.handle:
            lib().dgpu_bases_free(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class SortedScalars:
    """dgpu_scalars_sort: the partition sort of scalars [scalar_offset, scalar_offset + n) for rows [base_offset, ...) of a precomputed table;
    every table of the same shape (DeviceBases.same_table_shape) can run its MSM on it (DeviceBases.msm_sorted)"""

    def __init__(self, table, dscalars, n, base_offset=0, scalar_offset=0):
        h = C.c_uint64(0)
        rc = lib().dgpu_scalars_sort(table.handle, base_offset, dscalars.handle, scalar_offset, n, C.byref(h))
        if rc:
            raise DockGpuError(rc, "dgpu_scalars_sort")
        self.handle, self.n = h.value, n

    def free(self):
        if self.handle:
            lib().dgpu_scalars_free(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceScalars:
    def __init__(self, scalars, montgomery=False):
        _ensure()
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        self.n = len(scalars)
        h = C.c_uint64(0)
        rc = lib().dgpu_scalars_upload(_p(scalars), self.n, int(montgomery), C.byref(h))
        if rc:
            raise DockGpuError(rc, "dgpu_scalars_upload")
        self.handle = h.value

    @classmethod
    def from_parts(cls, parts, montgomery=False):
        """one resident vector = the given arrays back to back (no host concatenation)"""
        _ensure()
        arrs = [np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4) for a in parts]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        cnts = (C.c_size_t * len(arrs))(*[len(a) for a in arrs])
        self = cls.__new__(cls)
        self.n = sum(len(a) for a in arrs)
        h = C.c_uint64(0)
        rc = lib().dgpu_scalars_upload_parts(ptrs, cnts, len(arrs), int(montgomery), C.byref(h))
        if rc:
            raise DockGpuError(rc, "dgpu_scalars_upload_parts")
        self.handle = h.value
        return self

    def free(self):
        if self.handle:
            lib().dgpu_scalars_free(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


TABLE_C_WITNESS = 17      # include/dock_gpu.h DGPU_TABLE_C_WITNESS: table window width for queries that are multiplied by a witness


class prof:
    """Per-stage HIP-event timings recorded inside the library on its own stream.  Development surface (include/dock_gpu_dev.h): only inside
    `with crypto_amd.twin():` — the product library has no stage timers."""

    @staticmethod
    def enable(on=True):
        lib().dgpu_prof_enable(1 if on else 0)

    @staticmethod
    def reset():
        lib().dgpu_prof_reset()

    @staticmethod
    def read():
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        calls = (C.c_uint64 * cap)()
        k = lib().dgpu_prof_read(names, ms, calls, cap)
        return {names[i].decode(): (ms[i], calls[i]) for i in range(k)}
