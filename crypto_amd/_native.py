"""ctypes binding of libdock_gpu.so (include/dock_gpu.h) and of its development twin libdock_gpu_dev.so (include/dock_gpu_dev.h: the product's
objects plus the tuning knobs, stage timers and self-test hooks).  Fails loudly when a library is absent.

`lib()` is the library every wrapper of this package calls: the PRODUCT unless the caller is inside `with twin():` (tests that sweep a knob,
tools/, the stage / roofline leg of bench.py).  The two libraries are separate images with their own contexts, streams and handle tables: a
handle made under one must be used and freed under the same one."""
import contextlib
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("DGPU_LIB") or os.path.join(_HERE, "libdock_gpu.so")   # DGPU_LIB: development override (make g1only)
_DEV_SO = os.environ.get("DGPU_DEV_LIB") or os.path.join(_HERE, "libdock_gpu_dev.so")   # DGPU_DEV_LIB: development override (A/B builds)

ERR = {0: "DGPU_OK", -1: "DGPU_E_NODEVICE", -2: "DGPU_E_OOM", -3: "DGPU_E_BADARG", -4: "DGPU_E_HIP",
       -5: "DGPU_E_ZERO", -6: "DGPU_E_TOO_SMALL", -7: "DGPU_E_LENGTH"}


class DockGpuError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = ERR.get(code, str(code))
        try:
            msg += ": " + lib().dgpu_strerror(code).decode()
            if code == -4:
                msg += " (hip error %d)" % lib().dgpu_last_hip_error()
        except Exception:
            pass
        super().__init__("%s %s" % (what, msg))


def build_native(jobs=3):
    """Compile libdock_gpu.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j%d" % jobs], stdout=subprocess.DEVNULL)
    return _SO


class LegoPk(C.Structure):
    """include/dock_gpu.h `dgpu_lego_pk`"""
    _fields_ = [(k, C.c_uint64) for k in ("a_query", "b_g1_query", "b_g2_query", "h_query", "l_query")] + \
               [(k, C.c_void_p) for k in ("alpha_g1", "beta_g1", "delta_g1", "eta_delta_inv_g1", "eta_gamma_inv_g1", "beta_g2", "delta_g2", "a0", "b1_0", "b2_0", "gamma_abc_g1")] + \
               [("gamma_abc_len", C.c_size_t), ("commit_witness_count", C.c_size_t)]


class BasesView(C.Structure):
    """include/dock_gpu.h `dgpu_bases_view`: a slice of ark-ec Affine structs in host memory"""
    _fields_ = [("p", C.c_void_p), ("stride", C.c_size_t), ("x_off", C.c_size_t), ("y_off", C.c_size_t), ("inf_off", C.c_size_t), ("n", C.c_size_t)]


class LegoPkHost(C.Structure):
    """include/dock_gpu.h `dgpu_lego_pk_host`"""
    _fields_ = [(k, BasesView) for k in ("a_query", "b_g1_query", "b_g2_query", "h_query", "l_query")] + \
               [(k, C.c_void_p) for k in ("alpha_g1", "beta_g1", "delta_g1", "eta_delta_inv_g1", "eta_gamma_inv_g1", "beta_g2", "delta_g2", "a0", "b1_0", "b2_0", "gamma_abc_g1")] + \
               [("gamma_abc_len", C.c_size_t), ("commit_witness_count", C.c_size_t)]


APPEND_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t)
CHALLENGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint64))


class Transcript(C.Structure):
    """include/dock_gpu.h `dgpu_transcript`"""
    _fields_ = [("ctx", C.c_void_p), ("append_message", APPEND_FN), ("challenge_scalar", CHALLENGE_FN)]


class SnarkpackProverSrs(C.Structure):
    """include/dock_gpu.h `dgpu_snarkpack_prover_srs`"""
    _fields_ = [("n", C.c_size_t)] + [(k, C.c_void_p) for k in ("g_alpha_powers_table", "g_beta_powers_table", "h_alpha_powers_table", "h_beta_powers_table", "vkey_a", "vkey_b", "wkey_a", "wkey_b")]


class SnarkpackVerifierSrs(C.Structure):
    """include/dock_gpu.h `dgpu_snarkpack_verifier_srs`"""
    _fields_ = [("n", C.c_size_t)] + [(k, C.c_void_p) for k in ("g", "h", "g_alpha", "g_beta", "h_alpha", "h_beta")]


class Groth16Vk(C.Structure):
    """include/dock_gpu.h `dgpu_groth16_vk`"""
    _fields_ = [(k, C.c_void_p) for k in ("alpha_g1", "beta_g2", "gamma_g2", "delta_g2", "gamma_abc_g1")] + [("gamma_abc_len", C.c_size_t)]


_lib = None          # the library the wrappers call right now
_loaded = {}         # path -> CDLL
_init_args = None    # ("init", device, min_gpu_n) / ("list", [devices], min_gpu_n): how the product was initialised — the twin is brought up the same way

# every symbol include/dock_gpu_dev.h adds (the twin only)
DEV_SYMBOLS = ["dgpu_set_window_bits", "dgpu_set_chunk", "dgpu_set_reduce_shift", "dgpu_set_reduce_lanes", "dgpu_set_miller_pipeline",
               "dgpu_prof_enable", "dgpu_prof_reset", "dgpu_prof_read", "dgpu_selftest_fp_mul", "dgpu_selftest_g1_sum", "dgpu_selftest_glv_decompose",
               "dgpu_dev_fail_alloc_after"]

# every symbol include/dock_gpu.h declares
SYMBOLS = [
    "dgpu_runtime_hints", "dgpu_init", "dgpu_init_devices", "dgpu_init_device_list", "dgpu_context_count", "dgpu_set_device", "dgpu_shutdown", "dgpu_device_count", "dgpu_strerror", "dgpu_last_hip_error",
    "dgpu_set_min_gpu_n", "dgpu_get_min_gpu_n", "dgpu_set_auto_shard_min_n", "dgpu_set_small_msm_max", "dgpu_reserve_g1", "dgpu_reserve_g2", "dgpu_device_alloc_count",
    "dgpu_set_bases_cache_bytes", "dgpu_set_bases_cache_min_n", "dgpu_set_bases_cache_verify", "dgpu_bases_cache_invalidate", "dgpu_bases_cache_clear", "dgpu_bases_cache_stats",
    "dgpu_msm_g1", "dgpu_msm_g1_mont", "dgpu_msm_g2", "dgpu_msm_g2_mont", "dgpu_msm_g1_strided", "dgpu_msm_g2_strided", "dgpu_bases_upload_g1_strided", "dgpu_bases_upload_g2_strided",
    "dgpu_bases_upload_g1", "dgpu_bases_upload_g2", "dgpu_bases_free", "dgpu_scalars_upload", "dgpu_scalars_upload_parts", "dgpu_scalars_free",
    "dgpu_msm_g1_handle", "dgpu_msm_g2_handle", "dgpu_msm_g1_resident", "dgpu_msm_g2_resident", "dgpu_bases_precompute_g1", "dgpu_bases_precompute_g2",
    "dgpu_msm_g1_sharded", "dgpu_msm_g2_sharded", "dgpu_bases_upload_g1_sharded", "dgpu_bases_upload_g2_sharded", "dgpu_msm_g1_sharded_handle", "dgpu_msm_g2_sharded_handle", "dgpu_scalars_upload_sharded", "dgpu_scalars_copy_range", "dgpu_msm_g1_sharded_resident", "dgpu_msm_g2_sharded_resident",
    "dgpu_fold_g1", "dgpu_fold_g2", "dgpu_lincomb_g1", "dgpu_lincomb_g2", "dgpu_multi_miller_loop", "dgpu_multi_miller_loop_sharded", "dgpu_bases_table_shape", "dgpu_scalars_sort", "dgpu_msm_g1_sorted", "dgpu_msm_g2_sorted", "dgpu_multi_miller_loop_segments", "dgpu_multi_pairing_segments", "dgpu_g2_prepare", "dgpu_multi_miller_loop_prepared", "dgpu_multi_miller_loop_mixed", "dgpu_multi_miller_loop_scaled", "dgpu_final_exponentiation", "dgpu_g1_scale_batch", "dgpu_fp12_mul", "dgpu_fp12_pow", "dgpu_fp12_multi_pow", "dgpu_gt_in_subgroup", "dgpu_g1_serialize", "dgpu_g1_deserialize", "dgpu_g2_serialize", "dgpu_g2_deserialize", "dgpu_witness_map", "dgpu_r1cs_upload", "dgpu_r1cs_free", "dgpu_r1cs_shape", "dgpu_witness_map_r1cs", "dgpu_witness_map_r1cs_resident",
    "dgpu_window_table_g1", "dgpu_window_table_g2", "dgpu_window_table_free", "dgpu_window_table_mul_g1", "dgpu_window_table_mul_g2", "dgpu_window_table_mul_to_bases_g1", "dgpu_window_table_mul_to_bases_g2", "dgpu_fixed_base_g1", "dgpu_fixed_base_g2", "dgpu_g1_mul_add_batch", "dgpu_g2_mul_add_batch",
    "dgpu_legogroth16_prove", "dgpu_legogroth16_prove_host", "dgpu_legogroth16_verify", "dgpu_legogroth16_verify_batch", "dgpu_handle_len", "dgpu_handle_context", "dgpu_shard_count", "dgpu_shard_part",
    "dgpu_snarkpack_proof_words", "dgpu_snarkpack_aggregate", "dgpu_snarkpack_verify",
    "dgpu_g1_fold_prepare", "dgpu_g2_fold_prepare", "dgpu_fold_prepare_pair", "dgpu_g1_fold_apply", "dgpu_g2_fold_apply", "dgpu_fold_free",
]


def lib():
    """the library the wrappers call: the product, or the twin inside `with twin():`"""
    global _lib
    if _lib is None:
        _lib = _load(_SO)
    return _lib


def dev_lib():
    """the development twin (loaded, not made current)"""
    return _load(_DEV_SO)


def is_twin():
    return _lib is not None and _lib is _loaded.get(_DEV_SO)


def note_init(*args):
    global _init_args
    if not is_twin():
        _init_args = args


@contextlib.contextmanager
def twin():
    """Inside the block every wrapper of this package calls the development twin (include/dock_gpu_dev.h: knobs, stage timers, self-tests), brought up
    on the same device(s) as the product.  Nests; handles do not cross the border."""
    global _lib
    prev = lib()
    T = dev_lib()
    if T is not prev:
        import numpy as np
        kind, devs, min_n = _init_args or ("init", 0, 0)
        if kind == "init":
            rc = T.dgpu_init(devs)
        else:
            arr = np.ascontiguousarray(devs, dtype=np.int32)
            rc = T.dgpu_init_device_list(arr.ctypes.data_as(C.c_void_p), len(arr)) if T.dgpu_context_count() != len(arr) else 0
        if rc:
            raise DockGpuError(rc, "twin: dgpu_init")
        T.dgpu_set_min_gpu_n(min_n)
    _lib = T
    try:
        yield T
    finally:
        _lib = prev


def _load(path):
    if path not in _loaded:
        if not os.path.exists(path):
            raise ImportError(
                "crypto_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % path)
        # A process that also uses torch (device memory, streams, torch.distributed) must load torch's bundled HIP runtime FIRST: with the library's
        # runtime loaded before it the process holds two of them and dgpu_init answers DGPU_E_NODEVICE.  So torch goes first whenever it is there.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        for s in SYMBOLS:
            getattr(L, s).restype = C.c_int32
        L.dgpu_strerror.restype = C.c_char_p
        L.dgpu_strerror.argtypes = [C.c_int32]
        L.dgpu_init.argtypes = [C.c_int32]
        L.dgpu_runtime_hints.argtypes = [C.c_uint32]
        L.dgpu_set_min_gpu_n.argtypes = [C.c_size_t]
        L.dgpu_set_auto_shard_min_n.argtypes = [C.c_size_t]
        L.dgpu_set_small_msm_max.argtypes = [C.c_size_t]
        L.dgpu_reserve_g1.argtypes = [C.c_size_t]
        L.dgpu_reserve_g2.argtypes = [C.c_size_t]
        L.dgpu_device_alloc_count.restype = C.c_uint64
        L.dgpu_device_alloc_count.argtypes = []
        vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
        L.dgpu_set_bases_cache_bytes.argtypes = [sz]
        L.dgpu_set_bases_cache_min_n.argtypes = [sz]
        L.dgpu_set_bases_cache_verify.argtypes = [C.c_int32]
        L.dgpu_bases_cache_invalidate.argtypes = [vp, sz]
        L.dgpu_bases_cache_clear.argtypes = []
        L.dgpu_bases_cache_stats.argtypes = [vp]
        for name in ("dgpu_msm_g1", "dgpu_msm_g1_mont", "dgpu_msm_g2", "dgpu_msm_g2_mont"):
            getattr(L, name).argtypes = [vp, vp, vp, sz, vp]
        for name in ("dgpu_msm_g1_strided", "dgpu_msm_g2_strided"):
            getattr(L, name).argtypes = [vp, sz, sz, sz, sz, vp, sz, C.c_int32, vp]
        for name in ("dgpu_bases_upload_g1_strided", "dgpu_bases_upload_g2_strided"):
            getattr(L, name).argtypes = [vp, sz, sz, sz, sz, sz, C.POINTER(u64)]
        for name in ("dgpu_bases_upload_g1", "dgpu_bases_upload_g2"):
            getattr(L, name).argtypes = [vp, vp, sz, C.POINTER(u64)]
        L.dgpu_bases_free.argtypes = [u64]
        L.dgpu_scalars_free.argtypes = [u64]
        L.dgpu_scalars_upload.argtypes = [vp, sz, C.c_int32, C.POINTER(u64)]
        L.dgpu_scalars_upload_parts.argtypes = [C.POINTER(vp), C.POINTER(sz), sz, C.c_int32, C.POINTER(u64)]
        for name in ("dgpu_msm_g1_handle", "dgpu_msm_g2_handle"):
            getattr(L, name).argtypes = [u64, sz, vp, sz, C.c_int32, vp]
        for name in ("dgpu_msm_g1_resident", "dgpu_msm_g2_resident"):
            getattr(L, name).argtypes = [u64, sz, u64, sz, sz, vp]
        L.dgpu_scalars_sort.argtypes = [u64, sz, u64, sz, sz, C.POINTER(u64)]
        L.dgpu_bases_table_shape.argtypes = [u64, C.POINTER(sz), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        for name in ("dgpu_msm_g1_sorted", "dgpu_msm_g2_sorted"):
            getattr(L, name).argtypes = [u64, u64, sz, vp]
        L.dgpu_init_devices.argtypes = [C.c_uint32]
        L.dgpu_init_device_list.argtypes = [vp, C.c_int32]
        L.dgpu_set_device.argtypes = [C.c_int32]
        L.dgpu_get_min_gpu_n.restype = C.c_size_t
        for name in ("dgpu_msm_g1_sharded", "dgpu_msm_g2_sharded"):
            getattr(L, name).argtypes = [vp, vp, vp, sz, C.c_int32, vp]
        for name in ("dgpu_bases_upload_g1_sharded", "dgpu_bases_upload_g2_sharded"):
            getattr(L, name).argtypes = [vp, vp, sz, C.c_int32, C.POINTER(u64)]
        for name in ("dgpu_msm_g1_sharded_handle", "dgpu_msm_g2_sharded_handle"):
            getattr(L, name).argtypes = [u64, vp, sz, C.c_int32, vp]
        L.dgpu_scalars_upload_sharded.argtypes = [vp, sz, C.c_int32, u64, C.POINTER(u64)]
        L.dgpu_scalars_copy_range.argtypes = [u64, sz, sz, C.c_int32, C.POINTER(u64)]
        for name in ("dgpu_msm_g1_sharded_resident", "dgpu_msm_g2_sharded_resident"):
            getattr(L, name).argtypes = [u64, u64, vp]
        L.dgpu_g2_prepare.argtypes = [vp, vp, sz, vp, vp]
        L.dgpu_multi_miller_loop_prepared.argtypes = [vp, vp, vp, sz, vp]
        L.dgpu_multi_miller_loop_mixed.argtypes = [vp, vp, vp, sz, vp, vp, vp, sz, vp]
        L.dgpu_multi_miller_loop_scaled.argtypes = [vp, vp, sz, vp, vp, sz, vp, vp, vp, sz, vp]
        L.dgpu_multi_miller_loop_segments.argtypes = [vp, vp, vp, sz, vp, sz, vp]
        L.dgpu_multi_pairing_segments.argtypes = [vp, vp, vp, sz, vp, sz, vp]
        L.dgpu_bases_precompute_g1.argtypes = [u64, C.c_int32]
        L.dgpu_bases_precompute_g2.argtypes = [u64, C.c_int32]
        L.dgpu_fold_g1.argtypes = [vp, sz, vp]
        L.dgpu_fold_g2.argtypes = [vp, sz, vp]
        L.dgpu_lincomb_g1.argtypes = [vp, vp, vp, sz, vp]
        L.dgpu_lincomb_g2.argtypes = [vp, vp, vp, sz, vp]
        L.dgpu_multi_miller_loop.argtypes = [vp, vp, vp, sz, vp]
        L.dgpu_multi_miller_loop_sharded.argtypes = [vp, vp, vp, sz, C.c_int32, vp]
        L.dgpu_final_exponentiation.argtypes = [vp, vp]
        L.dgpu_g1_scale_batch.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp]
        L.dgpu_fp12_mul.argtypes = [vp, vp, vp]
        L.dgpu_fp12_pow.argtypes = [vp, vp, vp]
        L.dgpu_fp12_multi_pow.argtypes = [vp, vp, sz, vp]
        L.dgpu_gt_in_subgroup.argtypes = [vp, sz, vp]
        for name in ("dgpu_g1_serialize", "dgpu_g2_serialize"):
            getattr(L, name).argtypes = [vp, vp, sz, C.c_int32, vp]
        for name in ("dgpu_g1_deserialize", "dgpu_g2_deserialize"):
            getattr(L, name).argtypes = [vp, sz, C.c_int32, vp, vp]
        L.dgpu_r1cs_upload.argtypes = [vp, vp, vp, sz] * 3 + [sz, sz, sz, C.c_int32, C.POINTER(u64)]
        L.dgpu_r1cs_free.argtypes = [u64]
        L.dgpu_r1cs_shape.argtypes = [u64, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
        L.dgpu_witness_map_r1cs.argtypes = [u64, vp, sz, C.c_int32, vp, C.POINTER(u64), C.POINTER(sz)]
        L.dgpu_witness_map_r1cs_resident.argtypes = [u64, u64, vp, C.POINTER(u64), C.POINTER(sz)]
        L.dgpu_witness_map.argtypes = [vp, vp, vp, sz] * 3 + [vp, sz, sz, sz, C.c_int32, vp, C.POINTER(u64), C.POINTER(sz)]
        for name in ("dgpu_window_table_g1", "dgpu_window_table_g2"):
            getattr(L, name).argtypes = [vp, C.POINTER(u64)]
        L.dgpu_window_table_free.argtypes = [u64]
        for name in ("dgpu_window_table_mul_g1", "dgpu_window_table_mul_g2"):
            getattr(L, name).argtypes = [u64, vp, sz, C.c_int32, vp, vp]
        for name in ("dgpu_window_table_mul_to_bases_g1", "dgpu_window_table_mul_to_bases_g2"):
            getattr(L, name).argtypes = [u64, vp, sz, C.c_int32, C.POINTER(u64)]
        for name in ("dgpu_fixed_base_g1", "dgpu_fixed_base_g2"):
            getattr(L, name).argtypes = [vp, vp, sz, C.c_int32, vp, vp]
        for name in ("dgpu_g1_mul_add_batch", "dgpu_g2_mul_add_batch"):
            getattr(L, name).argtypes = [vp, vp, vp, sz, vp, vp, sz, vp, vp]
        L.dgpu_legogroth16_prove.argtypes = [vp, u64, u64, vp, sz, sz, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
        L.dgpu_legogroth16_prove_host.argtypes = [vp, u64, vp, sz, C.c_int32, vp, sz, vp, sz, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
        for name in ("dgpu_g1_fold_prepare", "dgpu_g2_fold_prepare"):
            getattr(L, name).argtypes = [vp, sz, C.POINTER(u64)]
        for name in ("dgpu_g1_fold_apply", "dgpu_g2_fold_apply"):
            getattr(L, name).argtypes = [u64, vp, vp, vp, vp]
        L.dgpu_fold_free.argtypes = [u64]
        L.dgpu_fold_prepare_pair.argtypes = [vp, sz, C.POINTER(u64), vp, sz, C.POINTER(u64)]
        L.dgpu_snarkpack_proof_words.restype = C.c_size_t
        L.dgpu_snarkpack_proof_words.argtypes = [sz, C.c_int32]
        L.dgpu_snarkpack_aggregate.argtypes = [vp, vp, vp, vp, vp, sz, vp, vp, sz, C.POINTER(sz)]
        L.dgpu_snarkpack_verify.argtypes = [vp, vp, vp, sz, sz, vp, sz, C.c_int32, vp, vp, vp, C.c_int32, C.POINTER(C.c_int32)]
        L.dgpu_legogroth16_verify_batch.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp, vp, sz, vp, sz, C.c_int32, vp, C.POINTER(C.c_int32)]
        L.dgpu_legogroth16_verify.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, sz, C.c_int32, C.POINTER(C.c_int32)]
        L.dgpu_handle_len.argtypes = [u64, C.POINTER(sz)]
        L.dgpu_handle_context.argtypes = [u64, C.POINTER(C.c_int32)]
        L.dgpu_shard_count.argtypes = [u64, C.POINTER(C.c_int32)]
        L.dgpu_shard_part.argtypes = [u64, sz, C.POINTER(u64), C.POINTER(sz), C.POINTER(sz), C.POINTER(C.c_int32)]
        if hasattr(L, "dgpu_prof_read"):                      # the development twin (include/dock_gpu_dev.h)
            for s in DEV_SYMBOLS:
                getattr(L, s).restype = C.c_int32
            for name in ("dgpu_set_window_bits", "dgpu_set_chunk", "dgpu_set_miller_pipeline", "dgpu_set_reduce_shift", "dgpu_set_reduce_lanes", "dgpu_prof_enable"):
                getattr(L, name).argtypes = [C.c_int32]
            L.dgpu_prof_read.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(u64), C.c_int32]
            L.dgpu_selftest_fp_mul.argtypes = [vp, vp, sz, vp]
            L.dgpu_selftest_glv_decompose.argtypes = [vp, vp, vp]
            L.dgpu_selftest_g1_sum.argtypes = [vp, vp, sz, vp]
            L.dgpu_dev_fail_alloc_after.argtypes = [C.c_int64, C.c_int64]
        _loaded[path] = L
    return _loaded[path]
