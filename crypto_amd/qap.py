"""Mirror of `LibsnarkReduction::witness_map_from_matrices` (/root/reference/legogroth16/src/r1cs_to_qap.rs:150-210)
over the C ABI (dgpu_witness_map): h = ((A z)(B z) - C z) / Z_D, returned as canonical scalars — on the host, and/or
left in HBM as a `DeviceScalars` handle for the prover's h_query MSM (legogroth16/src/prover.rs:281-286)."""
import ctypes as C
import numpy as np
from ._native import lib, DockGpuError
from .msm import _ensure, DeviceScalars


def csr(rows):
    """rows: list of [(coeff_int, var_index), ...] -> (rowptr u64, cols u32, vals (nnz, 4) u64 canonical)"""
    rp, cols, vals = [0], [], []
    for row in rows:
        for co, idx in row:
            cols.append(idx)
            vals.append([(co >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)])
        rp.append(len(cols))
    return (np.array(rp, dtype=np.uint64), np.array(cols, dtype=np.uint32),
            np.array(vals, dtype=np.uint64).reshape(-1, 4) if vals else np.zeros((0, 4), np.uint64))


def _p(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)


def witness_map(A, B, Cm, assignment, num_inputs, num_constraints, montgomery=False, to_host=True, resident=False, h_montgomery=False):
    """A, B, Cm: (rowptr, cols, vals) CSR triples.  Returns (h or None, DeviceScalars or None)."""
    _ensure()
    z = np.ascontiguousarray(assignment, dtype=np.uint64).reshape(-1, 4)
    mats = []
    for rp, cl, vl in (A, B, Cm):
        rp = np.ascontiguousarray(rp, dtype=np.uint64); cl = np.ascontiguousarray(cl, dtype=np.uint32)
        vl = np.ascontiguousarray(vl, dtype=np.uint64).reshape(-1, 4)
        if len(rp) != num_constraints + 1 or len(cl) != len(vl) or int(rp[-1]) != len(cl):
            raise ValueError("malformed CSR matrix")
        if len(cl) and int(cl.max()) >= len(z):
            raise ValueError("column index out of range")
        mats.append((rp, cl, vl))
    D = 1
    while D < num_constraints + num_inputs:
        D *= 2
    D = max(D, 2)
    out = np.zeros((D, 4), dtype=np.uint64) if to_host else None
    handle = C.c_uint64(0)
    olen = C.c_size_t(0)
    args = []
    for rp, cl, vl in mats:
        args += [rp.ctypes.data_as(C.c_void_p), _p(cl), _p(vl), len(cl)]
    rc = lib().dgpu_witness_map(*args, z.ctypes.data_as(C.c_void_p), len(z), num_inputs, num_constraints, int(bool(montgomery)) | (2 if h_montgomery else 0),
                                None if out is None else out.ctypes.data_as(C.c_void_p), C.byref(handle) if resident else None, C.byref(olen))
    if rc:
        raise DockGpuError(rc, "dgpu_witness_map")
    ds = None
    if resident:
        ds = DeviceScalars.__new__(DeviceScalars)
        ds.n, ds.handle = olen.value, handle.value
    return out, ds


class DeviceR1cs:
    """The circuit's three constraint matrices resident in HBM (dgpu_r1cs_upload): fixed per circuit, reused by every proof."""

    def __init__(self, A, B, Cm, num_vars, num_inputs, num_constraints, montgomery=False):
        _ensure()
        args = []
        self._keep = []
        for rp, cl, vl in (A, B, Cm):
            rp = np.ascontiguousarray(rp, dtype=np.uint64); cl = np.ascontiguousarray(cl, dtype=np.uint32)
            vl = np.ascontiguousarray(vl, dtype=np.uint64).reshape(-1, 4)
            if len(rp) != num_constraints + 1 or len(cl) != len(vl) or int(rp[-1]) != len(cl) or (len(cl) and int(cl.max()) >= num_vars):
                raise ValueError("malformed CSR matrix")
            args += [rp.ctypes.data_as(C.c_void_p), _p(cl), _p(vl), len(cl)]
        self.num_vars, self.num_inputs, self.num_constraints = num_vars, num_inputs, num_constraints
        h = C.c_uint64(0)
        rc = lib().dgpu_r1cs_upload(*args, num_vars, num_inputs, num_constraints, int(montgomery), C.byref(h))
        if rc:
            raise DockGpuError(rc, "dgpu_r1cs_upload")
        self.handle = h.value

    def shape(self):
        """(num_vars, num_inputs, num_constraints) as the library holds them (dgpu_r1cs_shape)"""
        v, i, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        rc = lib().dgpu_r1cs_shape(self.handle, C.byref(v), C.byref(i), C.byref(c))
        if rc:
            raise DockGpuError(rc, "dgpu_r1cs_shape")
        return v.value, i.value, c.value

    def witness_map(self, assignment, montgomery=False, to_host=True, resident=False, h_montgomery=False):
        """`assignment`: host scalars, or a DeviceScalars holding z on the circuit's device (dgpu_witness_map_r1cs_resident)"""
        D = 2
        while D < self.num_constraints + self.num_inputs:
            D *= 2
        out = np.zeros((D, 4), dtype=np.uint64) if to_host else None
        handle = C.c_uint64(0); olen = C.c_size_t(0)
        if isinstance(assignment, DeviceScalars):
            rc = lib().dgpu_witness_map_r1cs_resident(self.handle, assignment.handle, None if out is None else out.ctypes.data_as(C.c_void_p),
                                                      C.byref(handle) if resident else None, C.byref(olen))
        else:
            z = np.ascontiguousarray(assignment, dtype=np.uint64).reshape(-1, 4)
            rc = lib().dgpu_witness_map_r1cs(self.handle, z.ctypes.data_as(C.c_void_p), len(z), int(bool(montgomery)) | (2 if h_montgomery else 0),
                                             None if out is None else out.ctypes.data_as(C.c_void_p), C.byref(handle) if resident else None, C.byref(olen))
        if rc:
            raise DockGpuError(rc, "dgpu_witness_map_r1cs")
        ds = None
        if resident:
            ds = DeviceScalars.__new__(DeviceScalars)
            ds.n, ds.handle = olen.value, handle.value
        return out, ds

    def free(self):
        if self.handle:
            lib().dgpu_r1cs_free(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
