"""Consumer of the arkworks golden files (schema dock_gpu/ark-golden/1, tests/golden/ark/README.md).

tests/golden/ark/*.json are written by `cargo test -- --ignored write_golden` in rust/dock_gpu (REAL arkworks: the pin this image cannot
produce, no Rust toolchain here); while that directory holds none, the arkworks-parametrised tests skip.  tests/golden/ark_sample/*.json
have the identical schema but were computed by the oracle (tests/golden/gen_ark_schema_sample.py): they keep this consumer exercised — the
CPU half on every run, the C-ABI half under -m gpu — so that dropping the real files in needs no code.

CPU half: the oracle (oracle/oracle.c) == the file.  GPU half: libdock_gpu.so through its C ABI == the file.  Bar: bit-exact."""
import glob
import json
import os
import numpy as np
import pytest
import torch  # noqa: F401  (before the library: one HIP runtime per process)
import oracle_c as O

HERE = os.path.dirname(os.path.abspath(__file__))
SETS = {"arkworks": os.path.join(HERE, "golden", "ark"), "schema-sample (oracle-made)": os.path.join(HERE, "golden", "ark_sample")}


def words(h):
    return np.array([int(h[i:i + 16], 16) for i in range(0, len(h), 16)], dtype=np.uint64)


def load(which):
    cases = []
    for f in sorted(glob.glob(os.path.join(SETS[which], "*.json"))):
        doc = json.load(open(f))
        assert doc["schema"] == "dock_gpu/ark-golden/1", f
        if which == "arkworks":
            assert "ark" in doc["producer"] and "oracle" not in doc["producer"], "tests/golden/ark/ is for files written by the Rust side only"
        cases += doc["cases"]
    if not cases:
        pytest.skip("no %s golden files (tests/golden/ark/README.md: run rust/dock_gpu's write_golden)" % which)
    return cases


def normalised(G, jac):
    a, inf = G.to_affine(jac)
    h = G.AW // 2
    one = np.zeros(h, np.uint64); one[:6] = O.fp_to_mont(np.array([[1, 0, 0, 0, 0, 0]], np.uint64)).reshape(-1)
    return np.concatenate([one, one, np.zeros(h, np.uint64)]) if inf else np.concatenate([a, one])


def check(case, msm, miller, final_exp, prepare):
    """one case against an implementation given as four callables"""
    k = case["kind"]
    if k in ("msm_g1", "msm_g2"):
        G = O.G1 if k == "msm_g1" else O.G2
        n = case["n"]
        bases = words(case["bases"]).reshape(n, G.AW); sc = words(case["scalars"]).reshape(n, 4)
        inf = np.array([int(c) for c in case["inf"]], dtype=np.uint8)
        assert (msm(G, bases, sc, inf) == words(case["out"])).all(), (k, n)
    elif k == "miller_loop":
        n = case["n"]
        f = miller(words(case["p"]).reshape(n, 12), words(case["q"]).reshape(n, 24))
        assert (f == words(case["out"])).all(), (k, n)
        assert (final_exp(words(case["out"])) == words(case["final_exponentiation"])).all(), (k, n)
    elif k == "g2_prepared":
        assert (prepare(words(case["q"])).reshape(-1) == words(case["coeffs"])).all(), k
    else:
        raise AssertionError("unknown kind " + k)


@pytest.mark.parametrize("which", list(SETS))
def test_cpu_oracle_equals_the_golden_files(which):
    for case in load(which):
        check(case, msm=lambda G, b, s, i: normalised(G, G.msm(b, s, i, threads=8)), miller=lambda p, q: O.multi_miller_loop(p, q, threads=4),
              final_exp=O.final_exponentiation, prepare=O.g2_prepare)


@pytest.mark.gpu
@pytest.mark.parametrize("which", list(SETS))
def test_hip_library_equals_the_golden_files(which):
    import torch
    assert torch.cuda.is_available()
    import crypto_amd as ca
    from crypto_amd import pairing
    ca.init(0)
    curve = lambda G: ca.G1 if G is O.G1 else ca.G2
    for case in load(which):
        check(case, msm=lambda G, b, s, i: ca.msm_bigint(curve(G), b, s, i), miller=lambda p, q: ca.multi_miller_loop(p, q),
              final_exp=ca.final_exponentiation, prepare=lambda q: pairing.G2Prepared.from_affine(q.reshape(1, 24)).coeffs)


def test_the_rust_side_is_committed_as_source():
    """rust/dock_gpu is source a maintainer can build: the files exist, the extern block names only symbols include/dock_gpu.h declares, and
    the golden writer targets this directory's schema"""
    import re
    root = os.path.join(HERE, "..", "rust", "dock_gpu")
    for f in ("Cargo.toml", "build.rs", "src/lib.rs", "tests/parity.rs"):
        assert os.path.exists(os.path.join(root, f)), f
    lib_rs = open(os.path.join(root, "src", "lib.rs")).read()
    ffi_rs = open(os.path.join(root, "src", "ffi.rs")).read()          # (the extern block: generated from the header since round 6, tools/gen_rust_ffi.py)
    header = open(os.path.join(HERE, "..", "include", "dock_gpu.h")).read()
    declared = set(re.findall(r"\b(dgpu_[a-z0-9_]+)\s*\(", header))
    used = set(re.findall(r"pub fn (dgpu_[a-z0-9_]+)\(", ffi_rs))
    assert used and used <= declared, used - declared
    from crypto_amd._native import SYMBOLS
    assert used <= set(SYMBOLS)
    assert "dock_gpu/ark-golden/1" in open(os.path.join(root, "tests", "parity.rs")).read()
    assert "offset_of!(G1Affine, x)" in lib_rs and "dgpu_msm_g1_strided" in lib_rs and "dgpu_legogroth16_prove" in lib_rs
