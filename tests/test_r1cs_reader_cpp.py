"""CPU: the compiled host mirror's `.r1cs` reader (include/dock_gpu.hpp dock_gpu::circom::R1CSFile, what a C++ host uses where the reference uses
/root/reference/legogroth16/src/circom/r1cs_reader.rs:16-140) against crypto_amd/r1cs_file.py on the reference's own fixtures
(tests/golden/r1cs/*.r1cs): same header fields, the same three CSR matrices word for word (hashed), and the reader's errors on truncated, foreign
and section-less files.  Host logic only: the driver links nothing."""
import json
import os
import subprocess
import numpy as np
import pytest
from crypto_amd.r1cs_file import R1csFile, BLS12_381_ORDER

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "r1cs")


def fnv(h, data):
    for b in bytes(data):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def csr_hash(rp, cols, vals):
    h = 0xCBF29CE484222325
    for a in (rp, cols, vals):
        h = fnv(h, np.ascontiguousarray(a).tobytes())
    return "%016x" % h


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("r1cs") / "r1cs_reader_driver")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "r1cs_reader_driver.cpp")])
    # the header's other classes reference the C ABI: link against the library so that the reader can be run (no entry point is called)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "r1cs_reader_driver.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "crypto_amd"), "-ldock_gpu", "-Wl,-rpath," + os.path.join(ROOT, "crypto_amd")])
    return exe


def run(driver, paths):
    r = subprocess.run([driver] + paths, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return [json.loads(l) for l in r.stdout.strip().splitlines()]


def test_cpp_reader_equals_python_reader(driver):
    names = sorted(n for n in os.listdir(FIX) if n.endswith(".r1cs"))
    assert len(names) >= 6
    out = run(driver, [os.path.join(FIX, n) for n in names])
    for n, got in zip(names, out):
        f = R1csFile.from_path(os.path.join(FIX, n))
        assert "error" not in got, (n, got)
        assert (got["n_wires"], got["n_pub_out"], got["n_pub_in"], got["n_prv_in"], got["n_constraints"], got["n_labels"]) == (f.n_wires, f.n_pub_out, f.n_pub_in, f.n_prv_in, f.n_constraints, f.n_labels), n
        assert got["num_inputs"] == f.num_inputs and got["bls12_381"] == (f.prime == BLS12_381_ORDER) and got["last_label"] == f.wire_mapping[-1], n
        mats = f.csr()
        assert got["nnz"] == [len(m[1]) for m in mats], n
        assert got["hash"] == [csr_hash(*m) for m in mats], n


def test_cpp_reader_errors(driver, tmp_path):
    good = open(os.path.join(FIX, "multiply2.r1cs"), "rb").read()
    cases = {"magic": b"r1cx" + good[4:], "version": good[:4] + (2).to_bytes(4, "little") + good[8:], "truncated": good[:len(good) // 2], "empty": b""}
    # no wire2label section: keep the first two sections only (the section table is sequential)
    off, secs = 12, []
    for _ in range(int.from_bytes(good[8:12], "little")):
        typ, size = int.from_bytes(good[off:off + 4], "little"), int.from_bytes(good[off + 4:off + 12], "little")
        secs.append((typ, good[off:off + 12 + size])); off += 12 + size
    cases["no wire2label"] = good[:8] + (len(secs) - 1).to_bytes(4, "little") + b"".join(b for t, b in secs if t != 3)
    paths = []
    for k, v in cases.items():
        p = tmp_path / (k.replace(" ", "_") + ".r1cs"); p.write_bytes(v); paths.append(str(p))
    out = run(driver, paths)
    want = {"magic": "Invalid magic number", "version": "Unsupported version", "truncated": "unexpected end of file", "empty": "unexpected end of file",
            "no wire2label": "No section offset for wire2label type found"}
    for k, got in zip(cases, out):
        assert got.get("error") == want[k], (k, got)
        with pytest.raises(ValueError):          # ... and the Python reader refuses the same files
            R1csFile(cases[k])
