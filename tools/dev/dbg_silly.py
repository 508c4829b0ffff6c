import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/oracle", ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np
import crypto_amd as ca
from crypto_amd import legogroth16 as LG, qap
import lego_setup as LS
import test_gpu_reference_circuits as T
ca.init(0)
for cw in (2, 1, 0):
    rng = np.random.default_rng(cw)
    shape = T.silly(1, 1)
    try:
        LG.generate_parameters(shape["A"], shape["B"], shape["C"], 2, 2, 3, *[T._rnd(rng) for _ in range(6)], T.g1(5), T.g2(7))
    except ValueError:
        pass
    pk, link = T._params(shape, cw, rng)
    for it in range(3):
        a, b = T._rnd(rng), T._rnd(rng)
        cs = T.silly(a, b)
        r, s, v, link_v = (T._rnd(rng) for _ in range(4))
        z = LS.scalars(cs["z"])
        circ = qap.DeviceR1cs(*[qap.csr(cs[k]) for k in "ABC"], len(cs["z"]), cs["n_inst"], cs["n_cons"])
        p_abi = LG.create_proof_with_reduction(pk, circ, r, s, v, z)
        p_py = LG.create_proof_with_reduction_py(pk, circ, r, s, v, z)
        h = LS.scalars(LS.witness_map(cs))
        p_h = LG.create_proof(pk, r, s, v, h, z[:2], z[2:])
        _, dh = circ.witness_map(z, to_host=True, resident=True)
        hd, _ = circ.witness_map(z, to_host=True)
        print(cw, it, {k: (bool((p_abi[k] == p_py[k]).all()), bool((p_py[k] == p_h[k]).all())) for k in p_abi}, "h equal:", bool((hd == h[:len(hd)]).all()), len(hd), len(h))
