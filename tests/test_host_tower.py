"""CPU: the host's Fq12 tower with lazy reduction (crypto_amd/csrc/host_field.hpp: an Fq6 product reduces 6 times instead of 18, the
Karatsuba sums are taken on unreduced 768-bit products) against the eager tower it replaced — the same canonical residues, bit for bit — and
under -DHOSTF_CHECK, where every wide value carries its worst-case bound and every operation asserts its precondition (offsets cover the
subtrahends, sums stay below 2^768, the reduction's conditional subtractions suffice).  The values themselves are pinned against the oracle
by tests/test_gt_host.py and the GPU pairing tests; this test pins the arithmetic identity and the bounds."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "host_tower_driver.cpp")


def _run(tmp_path, name, flags, iters):
    cxx = shutil.which("g++") or shutil.which("clang++")
    if not cxx:
        pytest.skip("needs a host C++ compiler")
    exe = str(tmp_path / name)
    subprocess.check_call([cxx, "-std=c++17", "-O2", "-mbmi2", "-madx", "-Wno-unknown-pragmas"] + flags + ["-o", exe, SRC])
    r = subprocess.run([exe, str(iters)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("host_tower_driver: "), (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    return r.stdout.split()[1]


def test_lazy_tower_equals_eager_tower_and_respects_its_bounds(tmp_path):
    iters = 1500
    lazy = _run(tmp_path, "lazy", [], iters)
    checked = _run(tmp_path, "checked", ["-DHOSTF_CHECK"], iters)
    eager = _run(tmp_path, "eager", ["-DHOSTF_EAGER_REDUCTION"], iters)
    assert lazy == eager == checked
