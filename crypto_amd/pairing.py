"""Host-side mirror of the reference's pairing call surface over the C ABI.

  * E::multi_miller_loop(a, b)   -> multi_miller_loop()   (utils/src/randomized_pairing_check.rs:207, legogroth16/src/verifier.rs:69-76)
  * E::final_exponentiation(f)   -> final_exponentiation() returns None where arkworks returns None
  * E::multi_pairing(a, b)       -> multi_pairing()       (bbs_plus/src/signature.rs:284, legogroth16/src/link/snark.rs:157)
Arrays: numpy uint64 in the ABI layout (G1 affine 12, G2 affine 24, Fp12 72 limbs; Montgomery).
"""
import ctypes as C
import numpy as np
from ._native import lib, DockGpuError
from .msm import _ensure


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def multi_miller_loop(ps, qs, skip=None):
    _ensure()
    ps = np.ascontiguousarray(ps, dtype=np.uint64).reshape(-1, 12)
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
