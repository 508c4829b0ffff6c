"""arkworks `CanonicalSerialize` / `CanonicalDeserialize` for BLS12-381 group elements (Zcash format) over the C ABI:
what the reference's keys and proofs look like on disk (legogroth16/src/data_structures.rs:7-186, utils/src/serde_utils.rs:8-33)."""
import ctypes as C
import numpy as np
from ._native import lib, DockGpuError

_SZ = {("g1", True): 48, ("g1", False): 96, ("g2", True): 96, ("g2", False): 192}


def serialize(curve, points, is_inf=None, compressed=True):
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, curve.AW)
    inf = None if is_inf is None else np.ascontiguousarray(is_inf, dtype=np.uint8)
    out = np.zeros(len(pts) * _SZ[(curve.tag, compressed)], dtype=np.uint8)
    fn = lib().dgpu_g1_serialize if curve.tag == "g1" else lib().dgpu_g2_serialize
    rc = fn(pts.ctypes.data_as(C.c_void_p), None if inf is None else inf.ctypes.data_as(C.c_void_p), len(pts), int(compressed), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise DockGpuError(rc, "serialize")
    return out.tobytes()


def deserialize(curve, data, compressed=True, validate=True):
    """CanonicalDeserialize: validate=True is Validate::Yes (curve + prime-order subgroup), False is Validate::No (curve only)"""
    sz = _SZ[(curve.tag, compressed)]
    if len(data) % sz:
        raise ValueError("length is not a multiple of %d" % sz)
    n = len(data) // sz
    buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    pts = np.zeros((n, curve.AW), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    fn = lib().dgpu_g1_deserialize if curve.tag == "g1" else lib().dgpu_g2_deserialize
    rc = fn(buf.ctypes.data_as(C.c_void_p), n, int(compressed) | (0 if validate else 2), pts.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p))
    if rc:
        raise DockGpuError(rc, "deserialize")
    return pts, inf
