"""Stand-in for crypto_amd inside bench.py when DGPU_BENCH_STUB is set (tests/test_bench_multi_rank_cpu.py): the group is Z_r with generator 1,
a "point" k G is the integer k, an MSM is sum s_i k_i mod r — so bench.py's whole N > 1 control flow (per-rank seeds and term counts, the
closed-form check over all ranks' terms, calls in flight, barrier + max-over-ranks timing, the all_gather of partial results and their fold,
the one JSON line from rank 0) runs on CPU ranks under gloo, with no device and no library.  Test infrastructure only."""
import numpy as np
import torch
import torch.distributed as dist

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def enc(v):
    out = np.zeros(18, np.uint64)
    for i in range(4):
        out[i] = (v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
    out[12] = 1                        # "z != 0": not the identity
    return out


def dec(a):
    return sum(int(a[i]) << (64 * i) for i in range(4))


def _ints(a):
    return [sum(int(r[i]) << (64 * i) for i in range(4)) for r in np.asarray(a).reshape(-1, 4)]


class _Curve:
    JW, AW, tag = 18, 12, "g1"


class _Prof:
    def enable(self, on): pass
    def reset(self): pass
    def read(self): return {}


class _Scalars:
    def __init__(self, s): self.s, self.n = s, len(s)
    def free(self): pass


class _Bases:
    def __init__(self, ks): self.ks, self.n, self.built = ks, len(ks), False
    def precompute(self, c=None): self.built = True; return self
    def table_shape(self): return (self.n, 20, 13) if self.built else None
    def msm_resident(self, ds, n=None):
        n = min(self.n, ds.n) if n is None else n
        return enc(sum(a * b for a, b in zip(_ints(self.ks[:n]), _ints(ds.s[:n]))) % R_MOD)
    def free(self): pass


class _Table:
    def __init__(self, curve, gen): pass
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def multiply_many_to_bases(self, ks): return _Bases(ks)
    def multiply(self, tot): return enc(tot % R_MOD)[:12], False


class ca:                               # noqa: N801  (plays the module crypto_amd)
    G1 = _Curve()
    prof = _Prof()

    @staticmethod
    def twin():
        import contextlib
        return contextlib.nullcontext()
    DeviceScalars = _Scalars
    @staticmethod
    def init(device): pass
    @staticmethod
    def device_alloc_count(): return 0


class serde:                            # noqa: N801
    @staticmethod
    def deserialize(curve, raw): return np.zeros((1, 12), np.uint64), np.zeros(1, np.uint8)


class FB:                               # noqa: N801
    WindowTable = _Table


class sharded:                          # noqa: N801
    @staticmethod
    def gather_and_fold(curve, part, device=None):
        """the same collective as crypto_amd.sharded.gather_and_fold (all_gather of the fixed-size partials, then a local fold)"""
        t = torch.from_numpy(np.ascontiguousarray(part, dtype=np.uint64).view(np.int64).copy())
        buf = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(buf, t)
        return enc(sum(dec(b.numpy().view(np.uint64)) for b in buf) % R_MOD)
