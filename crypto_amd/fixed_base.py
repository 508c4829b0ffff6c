"""Mirror of `dock_crypto_utils::msm::WindowTable` (/root/reference/utils/src/msm.rs:8-62) on the C ABI.

    WindowTable::new(num_multiplications, group_elem)      -> WindowTable(curve, base)
    table.multiply(&s) / &table * &s                       -> table.multiply(s)
    table.multiply_many(&[Fr])                             -> table.multiply_many(scalars)
    multiply_field_elems_with_same_group_elem(g, &[Fr])    -> multiply_field_elems_with_same_group_elem(curve, g, scalars)

ark-ec sizes its window from `num_multiplications`; the device table always uses 8-bit windows (32 x 255 affine
multiples, built once per base), so the argument is accepted and ignored — it "does not impact correctness but only
performance" (msm.rs:17-18).  Results come back normalised (G::normalize_batch, as generator.rs:424-431 does right after):
an (n, 12) / (n, 24) uint64 array of affine Montgomery limbs plus a uint8 identity mask.
"""
import ctypes as C
import numpy as np
from ._native import lib, DockGpuError
from .msm import G1, G2, _ensure, DeviceBases


def _scalars_to_limbs(scalars):
    """(n, 4) uint64 limbs, from such an array or from Python ints"""
    if isinstance(scalars, np.ndarray) and scalars.dtype == np.uint64:
        return np.ascontiguousarray(scalars).reshape(-1, 4)
    return np.array([[(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in scalars], dtype=np.uint64).reshape(-1, 4)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class WindowTable:
    def __init__(self, curve, group_elem, num_multiplications=0):
        _ensure()
        self.curve = curve
        self.words = 12 if curve is G1 else 24
        base = np.ascontiguousarray(group_elem, dtype=np.uint64).reshape(self.words)
        h = C.c_uint64(0)
        fn = lib().dgpu_window_table_g1 if curve is G1 else lib().dgpu_window_table_g2
        rc = fn(_p(base), C.byref(h))
        if rc:
            raise DockGpuError(rc, "dgpu_window_table")
        self.handle = h.value

    def multiply_many(self, scalars, montgomery=False):
        s = _scalars_to_limbs(scalars)
        n = len(s)
        out = np.zeros((n, self.words), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        fn = lib().dgpu_window_table_mul_g1 if self.curve is G1 else lib().dgpu_window_table_mul_g2
        rc = fn(self.handle, _p(s), n, 1 if montgomery else 0, _p(out), _p(inf))
        if rc:
            raise DockGpuError(rc, "dgpu_window_table_mul")
        return out, inf

    def multiply_many_to_bases(self, scalars, montgomery=False):
        """multiply_many whose products stay in HBM as prepared MSM bases (a proving-key query straight from the generator)"""
        s = _scalars_to_limbs(scalars)
        h = C.c_uint64(0)
        fn = lib().dgpu_window_table_mul_to_bases_g1 if self.curve is G1 else lib().dgpu_window_table_mul_to_bases_g2
        rc = fn(self.handle, _p(s), len(s), 1 if montgomery else 0, C.byref(h))
        if rc:
            raise DockGpuError(rc, "dgpu_window_table_mul_to_bases")
        return DeviceBases.from_handle(self.curve, h.value, len(s))

    def multiply(self, scalar):
        out, inf = self.multiply_many([scalar])
        return out[0], bool(inf[0])

    def free(self):
        if self.handle:
            lib().dgpu_window_table_free(self.handle)
            self.handle = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.free()


def multiply_field_elems_with_same_group_elem(curve, group_elem, elements, montgomery=False):
    """utils/src/msm.rs:55-62"""
    _ensure()
    words = 12 if curve is G1 else 24
    base = np.ascontiguousarray(group_elem, dtype=np.uint64).reshape(words)
    s = _scalars_to_limbs(elements)
    n = len(s)
    out = np.zeros((n, words), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    fn = lib().dgpu_fixed_base_g1 if curve is G1 else lib().dgpu_fixed_base_g2
    rc = fn(_p(base), _p(s), n, 1 if montgomery else 0, _p(out), _p(inf))
    if rc:
        raise DockGpuError(rc, "dgpu_fixed_base")
    return out, inf
