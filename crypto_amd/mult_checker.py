"""Mirror of `dock_crypto_utils::randomized_mult_checker::RandomizedMultChecker`
(/root/reference/utils/src/randomized_mult_checker.rs:11-118) on the C ABI: many claimed scalar multiplications / small MSMs
`sum_i s_i P_i == T` are batched with powers of a random scalar into ONE variable-base MSM whose result must be the identity
(`G::Group::msm_unchecked(&points, &scalars).is_zero()`, :100 — one of the reference's large-n MSM call sites).

Same state and merging rule: `args` maps a point's x coordinate to (scalar, point), so a point and its negative share one
entry (:104-117).  Points are affine ABI arrays (identity = all-zero words, ignored as in the reference); scalars Python ints.
"""
import numpy as np
import importlib
M = importlib.import_module(__package__ + ".msm")

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


class RandomizedMultChecker:
    def __init__(self, curve, random):                       # new(random)  :20-26
        self.curve = curve
        self.args = {}                                       # x-coordinate bytes -> [scalar, point]
        self.random = random % R_MOD
        self.current_random = 1

    def __len__(self):
        return len(self.args)

    def _add(self, p, s):                                    # :104-117
        p = np.ascontiguousarray(p, dtype=np.uint64).reshape(self.curve.AW)
        if not p.any():
            return                                           # the point at infinity does not change the result
        h = self.curve.AW // 2
        key = p[:h].tobytes()
        ent = self.args.get(key)
        if ent is None:
            self.args[key] = [s % R_MOD, p.copy()]
        elif (ent[1] == p).all():
            ent[0] = (ent[0] + s) % R_MOD
        else:                                                # same x, other y: the stored point is -p
            ent[0] = (ent[0] - s) % R_MOD

    def add_1(self, p, s, t):                                # s p == t   :32-36
        self.add_many([p], [s], t)

    def add_2(self, p1, s1, p2, s2, t):                      # :39-44
        self.add_many([p1, p2], [s1, s2], t)

    def add_3(self, p1, s1, p2, s2, p3, s3, t):              # :47-61
        self.add_many([p1, p2, p3], [s1, s2, s3], t)

    def add_many(self, a, b, t):                             # sum b_i a_i == t   :64-75
        for a_i, b_i in zip(a, b):
            self._add(a_i, self.current_random * b_i)
        self._add(t, -self.current_random)
        self.current_random = self.current_random * self.random % R_MOD

    def verify(self):                                        # :78-86: one MSM, result must be the identity
        if not self.args:
            return True
        pts = np.stack([e[1] for e in self.args.values()])
        sc = np.array([[(e[0] >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for e in self.args.values()], dtype=np.uint64)
        res = M.msm_bigint(self.curve, pts, sc)
        return not res[self.curve.AW:].any()
