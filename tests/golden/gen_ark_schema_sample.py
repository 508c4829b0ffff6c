"""Writes tests/golden/ark_sample/{msm,pairing}.json: files of the schema rust/dock_gpu/tests/parity.rs `write_golden` produces
(tests/golden/ark/README.md), computed by the CPU ORACLE — they exercise tests/test_ark_golden.py's consumer until someone with a Rust
toolchain writes the real tests/golden/ark/*.json.  Run from the repository root:  python tests/golden/gen_ark_schema_sample.py"""
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", "..", "oracle"), os.path.join(HERE, "..")]
import oracle_c as O  # noqa: E402
import util as U      # noqa: E402

PRODUCER = "oracle/ (schema sample written by tests/golden/gen_ark_schema_sample.py; NOT arkworks)"
hexw = lambda a: "".join("%016x" % int(x) for x in np.asarray(a, dtype=np.uint64).reshape(-1))
ONE = O.fp_to_mont(np.array([[1, 0, 0, 0, 0, 0]], np.uint64)).reshape(-1)


def normalised(G, jac):
    a, inf = G.to_affine(jac)
    h = G.AW // 2
    one = np.zeros(h, np.uint64); one[:6] = ONE
    return np.concatenate([one, one, np.zeros(h, np.uint64)]) if inf else np.concatenate([a, one])


def main():
    out = os.path.join(HERE, "ark_sample")
    os.makedirs(out, exist_ok=True)
    cases = []
    for gname, G, sizes in (("g1", O.G1, (1, 2, 33, 300)), ("g2", O.G2, (1, 2, 33))):
        for n in sizes:
            bases, _, _ = U.seq_bases(G, n, 9000 + n, threads=8)
            inf = np.zeros(n, np.uint8)
            if n >= 33:
                inf[5] = 1; bases[5] = 0
            sc = O.rand_scalars(9100 + n, n)
            cases.append({"kind": "msm_" + gname, "n": n, "bases": hexw(bases), "inf": "".join(str(int(b)) for b in inf), "scalars": hexw(sc),
                          "out": hexw(normalised(G, G.msm(bases, sc, inf, threads=8)))})
    json.dump({"schema": "dock_gpu/ark-golden/1", "producer": PRODUCER, "cases": cases}, open(os.path.join(out, "msm.json"), "w"), indent=0)
    cases = []
    for n in (1, 3, 5):
        p, _, _ = U.seq_bases(O.G1, n, 9200 + n, threads=8); q, _, _ = U.seq_bases(O.G2, n, 9300 + n, threads=8)
        f = O.multi_miller_loop(p, q)
        cases.append({"kind": "miller_loop", "n": n, "p": hexw(p), "q": hexw(q), "out": hexw(f), "final_exponentiation": hexw(O.final_exponentiation(f))})
    q, _, _ = U.seq_bases(O.G2, 1, 9400, threads=1)
    cases.append({"kind": "g2_prepared", "q": hexw(q[0]), "coeffs": hexw(O.g2_prepare(q[0]))})
    json.dump({"schema": "dock_gpu/ark-golden/1", "producer": PRODUCER, "cases": cases}, open(os.path.join(out, "pairing.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
