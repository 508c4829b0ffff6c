"""Multi-GPU MSM and Miller loop: chunk sharding, one process per GPU.

An MSM is a sum, so rank g computes the full single-GPU pipeline over its own chunk of (base, scalar) pairs
and the only exchange is one normalised Jacobian point per rank (144 B for G1, 288 B for G2).  EC addition is
not an RCCL reduction op, therefore: all_gather of the partial points (backend "nccl" == RCCL over xGMI on
the GPU box, "gloo" in the CPU tests) followed by a local fold (`dgpu_fold_*`, host code).  Raw buckets are
never exchanged.
"""
import ctypes as C
import numpy as np
import torch
import torch.distributed as dist
from ._native import lib, DockGpuError


def chunk_bounds(n, world, rank):
    """Contiguous balanced partition of [0, n) — identical on every rank."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def fold(curve, partials):
    partials = np.ascontiguousarray(partials, dtype=np.uint64).reshape(-1, curve.JW)
    out = np.zeros(curve.JW, dtype=np.uint64)
    fn = lib().dgpu_fold_g1 if curve.tag == "g1" else lib().dgpu_fold_g2
    rc = fn(partials.ctypes.data_as(C.c_void_p), len(partials), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise DockGpuError(rc, "dgpu_fold")
    return out


def gather_and_fold(curve, local_partial, device=None):
    """all_gather the per-rank partial points and fold them; every rank returns the same full result."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return fold(curve, local_partial)
    world = dist.get_world_size()
    # int64 view of the u64 limbs: the collective only moves bytes
    t = torch.from_numpy(np.ascontiguousarray(local_partial, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    buf = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(buf, t)
    parts = np.stack([b.cpu().numpy().view(np.uint64) for b in buf])
    return fold(curve, parts)


def msm_sharded(curve, local_msm, device=None):
    """`local_msm()` returns this rank's partial (Jacobian limbs) over its own chunk."""
    return gather_and_fold(curve, local_msm(), device)


# ---- Miller loop: pairs are independent, the per-rank raw Fp12 outputs multiply (SURVEY.md 8e "Miller loop") ---------------
def fold_fp12(partials):
    """product of the per-rank MillerLoopOutput values (host code, dgpu_fp12_mul)"""
    parts = np.ascontiguousarray(partials, dtype=np.uint64).reshape(-1, 72)
    acc = parts[0].copy()
    out = np.zeros(72, dtype=np.uint64)
    for p in parts[1:]:
        rc = lib().dgpu_fp12_mul(acc.ctypes.data_as(C.c_void_p), np.ascontiguousarray(p).ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        if rc:
            raise DockGpuError(rc, "dgpu_fp12_mul")
        acc = out.copy()
    return acc


def multi_miller_loop_sharded(local_miller_loop, device=None):
    """`local_miller_loop()` returns this rank's raw Fp12 output over its own chunk of pairs (576 B); all_gather + product.
    The result is limb-for-limb the single-device multi_miller_loop value: squaring distributes over the per-step line products and
    Fp12 multiplication is exact and commutative, so the partition does not matter (then ONE final exponentiation, on every rank)."""
    local = np.ascontiguousarray(local_miller_loop(), dtype=np.uint64).reshape(72)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    t = torch.from_numpy(local.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    buf = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(buf, t)
    return fold_fp12(np.stack([b.cpu().numpy().view(np.uint64) for b in buf]))
