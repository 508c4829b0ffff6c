"""GPU (-m gpu): the C++ host mirror of the reference's interface (include/dock_gpu.hpp: VariableBaseMSM, Pairs, DeviceBases,
WindowTable, multi_miller_loop / final_exponentiation / multi_pairing) against the CPU oracle, from a compiled driver
(tests/native/cpp_api_driver.cpp) — the path a compiled host (the reference is Rust) takes: no Python between it and the C ABI."""
import os
import subprocess
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_driver():
    exe = os.path.join(ROOT, "tests", "native", "cpp_api_driver")
    src = exe + ".cpp"
    hdr = os.path.join(ROOT, "include", "dock_gpu.hpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", exe,
                               "-L" + os.path.join(ROOT, "crypto_amd"), "-ldock_gpu", "-L" + os.path.join(ROOT, "oracle"), "-loracle",
                               "-Wl,-rpath," + os.path.join(ROOT, "crypto_amd") + ":" + os.path.join(ROOT, "oracle")])
    return exe


def test_cpp_host_mirror_matches_oracle():
    import oracle_c
    oracle_c.build()
    exe = build_driver()
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "crypto_amd") + ":" + os.path.join(ROOT, "oracle") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "all equal" in r.stdout, r.stdout + r.stderr
