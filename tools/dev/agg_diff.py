import sys, os, numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R_ + "/oracle", R_ + "/tests", R_]
import torch
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd import aggregation as AG
from crypto_amd.aggregation import native as NA, legogroth16 as AL, groth16 as G16
from crypto_amd.fixed_base import WindowTable
ca.init(0)
rng = np.random.default_rng(3)
ints = lambda k: [int.from_bytes(rng.bytes(40), "little") % (U.R - 1) + 1 for _ in range(k)]
def fixed(curve, g, ks):
    with WindowTable(curve, g, len(ks)) as t:
        return t.multiply_many(ks)[0]
g, h = O.G1.generator(), O.G2.generator()
for n in (8, 64, 256):
    for with_d in (False, True):
        A, Cc, D = fixed(ca.G1, g, ints(n)), fixed(ca.G1, g, ints(n)), fixed(ca.G1, g, ints(n)); B = fixed(ca.G2, h, ints(n))
        proofs = [{"a": A[i], "b": B[i], "c": Cc[i], "d": D[i]} for i in range(n)]
        pk, vsrs = AG.setup_fake_srs(ints(1)[0], ints(1)[0], n, g, h).specialize(n)
        runs = []
        for who in ("py", "nat", "py", "nat", "nat", "py"):
            if who == "nat":
                runs.append(NA.aggregate_proofs_words(pk, AG.MerlinTranscript(b"t"), proofs, with_d=with_d))
            else:
                runs.append(NA.proof_to_words((AL if with_d else G16).aggregate_proofs(pk, AG.MerlinTranscript(b"t"), proofs)))
        print(n, with_d, [int((runs[0] != r).sum()) for r in runs], flush=True)

