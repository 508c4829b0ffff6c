"""Test infrastructure for the full-size GPU tests: exact big dot products with numpy, the nconstraints-shape circuit built
as CSR arrays at m = 2^20, and the oracle's witness map on CSR inputs."""
import ctypes as C
import numpy as np
import oracle_c as O
import lego_setup as LS

R = LS.R


def dot_mod_r(a, b):
    """sum a_i b_i mod r for two (n, 4) uint64 limb arrays, exactly: 16-bit pieces, float64 matrix products over chunks of
    2^20 rows (every partial sum < 2^32 * 2^20 = 2^52 is exact in a double)."""
    a16 = np.ascontiguousarray(a).view(np.uint16).reshape(len(a), 16)
    b16 = np.ascontiguousarray(b).view(np.uint16).reshape(len(b), 16)
    tot = 0
    for lo in range(0, len(a), 1 << 20):
        m = a16[lo:lo + (1 << 20)].astype(np.float64).T @ b16[lo:lo + (1 << 20)].astype(np.float64)
        for i in range(16):
            for j in range(16):
                tot += int(m[i, j]) << (16 * (i + j))
    return tot % R


def big_circuit(m, x0):
    """LS.circuit(m, x0) built with numpy (the list-of-lists form takes minutes at m = 2^20): x_i = x_{i-1}^2 + i.
    Returns (z as python ints, CSR triples A, B, C, n_inst, n_cons)."""
    xs = [x0 % R]
    for i in range(1, m + 1):
        xs.append((xs[-1] * xs[-1] + i) % R)
    z = [1, xs[-1]] + xs
    nc = m + 1
    one = np.zeros((1, 4), np.uint64); one[0, 0] = 1
    # A: row i-1 (i = 1..m): (1, w(i-1)); last row: (1, w(m)).  w(i) = 2 + i
    a_cols = np.arange(2, 2 + m + 1, dtype=np.uint32)
    a_rp = np.arange(0, nc + 1, dtype=np.uint64)
    a_vals = np.repeat(one, nc, axis=0)
    # B: rows 0..m-1 same as A; last row (1, 0)
    b_cols = a_cols.copy(); b_cols[-1] = 0
    # C: row i-1: (1, w(i)), (-i mod r, 0); last row: (1, 1)
    c_cols = np.empty(2 * m + 1, np.uint32)
    c_cols[0:2 * m:2] = np.arange(3, 3 + m, dtype=np.uint32); c_cols[1:2 * m:2] = 0; c_cols[-1] = 1
    c_rp = np.concatenate([np.arange(0, 2 * m + 1, 2, dtype=np.uint64), np.array([2 * m + 1], np.uint64)])
    c_vals = np.zeros((2 * m + 1, 4), np.uint64)
    c_vals[0:2 * m:2, 0] = 1; c_vals[-1, 0] = 1
    # r - i for i = 1..m: low limb borrows at most once (i < 2^32 <= low limb of r? r's low limb is 0xffffffff00000001)
    r_l = [int((R >> (64 * k)) & 0xFFFFFFFFFFFFFFFF) for k in range(4)]
    i_arr = np.arange(1, m + 1, dtype=np.uint64)
    assert m < r_l[0]
    c_vals[1:2 * m:2, 0] = np.uint64(r_l[0]) - i_arr
    for k in range(1, 4):
        c_vals[1:2 * m:2, k] = np.uint64(r_l[k])
    return z, (a_rp, a_cols, a_vals), (a_rp.copy(), b_cols, a_vals.copy()), (c_rp, c_cols, c_vals), 2, nc


def ints_to_limbs(vals):
    out = np.empty((len(vals), 4), np.uint64)
    mask = 0xFFFFFFFFFFFFFFFF
    for k in range(4):
        out[:, k] = np.fromiter(((v >> (64 * k)) & mask for v in vals), dtype=np.uint64, count=len(vals))
    return out


def oracle_map(mats, z_limbs, n_inst, n_cons):
    L = O.lib(); L.orc_witness_map.restype = C.c_int
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    D = 1
    while D < n_cons + n_inst:
        D *= 2
    out = np.zeros((D, 4), np.uint64)
    args = []
    for rp, cl, vl in mats:
        args += [p(rp), p(cl), p(vl)]
    L.orc_witness_map(*args, p(z_limbs), C.c_size_t(len(z_limbs)), C.c_size_t(n_inst), C.c_size_t(n_cons), p(out))
    return out


