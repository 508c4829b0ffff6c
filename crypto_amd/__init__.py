"""crypto_amd — MI355X (gfx950) backend for the BLS12-381 MSM / multi-Miller-loop hot path of
docknetwork/crypto.  The product is the C-ABI library `libdock_gpu.so` (include/dock_gpu.h, built from
crypto_amd/csrc/); this package is the thin host-side mirror of the reference's call surface
(`VariableBaseMSM::{msm, msm_unchecked, msm_bigint}`, `utils::pairs::Pairs`) used by tests and bench.py.
There is no CPU fallback: every call goes through the HIP library and raises if it is missing.
"""
from ._native import lib, DockGpuError, build_native, twin, dev_lib  # noqa: F401
from .msm import (  # noqa: F401
    G1, G2, msm_bigint, msm_unchecked, msm, Pairs, OwnedPairs, DeviceBases, DeviceScalars, SortedScalars, init, prof, init_devices, msm_bigint_sharded, ShardedDeviceBases,
    msm_strided, to_affine_structs, affine_struct_dtype, reserve, device_alloc_count, TABLE_C_WITNESS,
    bases_cache, bases_cache_stats, bases_cache_clear, bases_cache_invalidate, CACHE_VERIFY_FULL,
)
from .pairing import multi_miller_loop, final_exponentiation, multi_pairing  # noqa: F401,E402
from .pairing_check import RandomizedPairingChecker  # noqa: F401,E402
from . import qap  # noqa: F401,E402
from . import serde  # noqa: F401,E402
