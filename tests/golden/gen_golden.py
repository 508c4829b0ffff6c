"""Generates the golden fixtures in this directory from the build's own big-integer model
(oracle/bls12_381_model.py) — NOT from reference code: the reference (Rust + arkworks) cannot be
built or imported here and keeps no known-answer vectors for this path.

Run:  python tests/golden/gen_golden.py      (takes ~1-2 min; pure Python big ints)

Every expected value is computed two independent ways inside the model (arkworks-style Pippenger vs the
closed form (sum s_i k_i) * G with naive double-and-add) and asserted equal before it is written.
All numbers are plain integers (canonical, NOT Montgomery) as hex strings.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import bls12_381_model as M  # noqa: E402

hx = lambda v: hex(v)


def enc_g1(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def enc_g2(p):
    return None if p is None else [[hx(p[0][0]), hx(p[0][1])], [hx(p[1][0]), hx(p[1][1])]]


def enc_f12(f):
    return [hx(c) for h in f for q in h for c in q]   # c0.c0.c0, c0.c0.c1, ... c1.c2.c1


def msm_case(group, n, seed, edge=True):
    rng = M.SplitMix64(seed)
    gen, mul, neg, msm_f, enc = ((M.G1_GEN, M.g1_mul, M.g1_neg, M.g1_msm, enc_g1) if group == 1
                                 else (M.G2_GEN, M.g2_mul, M.g2_neg, M.g2_msm, enc_g2))
    ks = [rng.scalar() for _ in range(n)]
    ss = [rng.scalar() for _ in range(n)]
    if edge:
        if n >= 3:
            ss[0] = 0            # zero scalar
            ss[1] = M.R - 1      # scalar r - 1
        if n >= 6:
            ks[3] = ks[2]        # duplicate base
            ks[4] = M.R - ks[2]  # P and -P
        if n >= 8:
            ks[6] = 0            # identity base
            ss[7] = 1
        if n >= 10:
            ss[8] = ss[9] = 12345  # equal small scalars
    bases = [mul(gen, k) if k else None for k in ks]
    expect = msm_f(bases, ss)
    closed = mul(gen, sum(a * b for a, b in zip(ks, ss)) % M.R)
    assert expect == closed, (group, n)
    return {"group": "G%d" % group, "n": n, "seed": seed, "dlogs": [hx(k) for k in ks], "scalars": [hx(s) for s in ss],
            "bases": [enc(b) for b in bases], "expected": enc(expect)}


def main():
    out = {}
    out["g1_msm"] = [msm_case(1, n, 1000 + n) for n in (0, 1, 2, 31, 32, 33, 100)]
    # truncation: 5 bases, 3 scalars -> result over the first 3 pairs (legogroth16/src/prover.rs:286)
    t = msm_case(1, 5, 77, edge=False)
    t["scalars"] = t["scalars"][:3]
    t["expected"] = enc_g1(M.g1_msm([None if b is None else (int(b[0], 16), int(b[1], 16)) for b in t["bases"]][:3],
                                    [int(s, 16) for s in t["scalars"]]))
    t["note"] = "truncation to min(len)"
    out["g1_msm"].append(t)
    out["g2_msm"] = [msm_case(2, n, 2000 + n) for n in (0, 1, 2, 33)]
    # signed-digit recoding vectors (A.2)
    rng = M.SplitMix64(5)
    dig = []
    for c in (3, 8, 13, 15, 16, 18):
        for s in (0, 1, M.R - 1, rng.scalar(), rng.scalar()):
            dig.append({"c": c, "scalar": hx(s), "digits": M.ark_make_digits(s, c)})
    out["digits"] = dig
    out["window_c"] = [[n, M.ark_window_c(n)] for n in (0, 1, 31, 32, 33, 1 << 10, 1 << 16, 1 << 20, 1 << 22, 1 << 24)]
    # pairings
    e = M.pairing(M.G1_GEN, M.G2_GEN)
    assert M.f12_pow(e, M.R) == M.F12_ONE and e != M.F12_ONE
    pr = {"e_g1_g2": enc_f12(e), "miller_g1_g2": enc_f12(M.multi_miller_loop([M.G1_GEN], [M.G2_GEN]))}
    co = M.g2_prepare(M.G2_GEN)
    pr["g2_prepared_gen_first"] = [[hx(c) for q in co[0] for c in q]]
    pr["g2_prepared_gen_last"] = [[hx(c) for q in co[-1] for c in q]]
    pr["n_coeffs"] = len(co)
    cases = []
    for npairs, seed in ((2, 31), (3, 32), (8, 33)):
        rng = M.SplitMix64(seed)
        a = [rng.scalar() for _ in range(npairs)]
        b = [rng.scalar() for _ in range(npairs)]
        ps = [M.g1_mul(M.G1_GEN, x) for x in a]
        qs = [M.g2_mul(M.G2_GEN, x) for x in b]
        f = M.multi_miller_loop(ps, qs)
        gt = M.final_exponentiation(f)
        assert gt == M.f12_pow(e, sum(x * y for x, y in zip(a, b)) % M.R)
        cases.append({"a": [hx(x) for x in a], "b": [hx(x) for x in b], "p": [enc_g1(p) for p in ps], "q": [enc_g2(q) for q in qs],
                      "miller": enc_f12(f), "gt": enc_f12(gt)})
    pr["cases"] = cases
    out["pairing"] = pr
    for k, v in out.items():
        with open(os.path.join(HERE, k + ".json"), "w") as f:
            json.dump(v, f, indent=0, separators=(",", ":"))
        print("wrote", k)


if __name__ == "__main__":
    main()
