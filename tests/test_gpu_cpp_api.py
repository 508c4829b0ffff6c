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


def build_inflight_driver():
    """tests/native/inflight_threads.cpp: the headline's workload from native host threads (bench.py secondary.native_host_threads)"""
    exe = os.path.join(ROOT, "tests", "native", "inflight_threads")
    src = exe + ".cpp"
    hdr = os.path.join(ROOT, "include", "dock_gpu.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), src, "-o", exe,
                               "-L" + os.path.join(ROOT, "crypto_amd"), "-ldock_gpu", "-Wl,-rpath," + os.path.join(ROOT, "crypto_amd")])
    return exe


def test_native_host_threads_all_get_the_same_point_and_their_turn():
    """eight and twelve native threads against six slots: every result equals the first call's, and no call lasts longer than a fraction of the run
    (first come, first served: dock_ctx.hpp SlotLock)"""
    import re
    exe = build_inflight_driver()
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "crypto_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, "8,48", "12,72"], capture_output=True, text=True, timeout=600, env=env)
    rows = re.findall(r"T=\s*(\d+) count=\s*(\d+):\s*([\d.]+) ms total, ([\d.]+) ms per call, longest call ([\d.]+) ms, mismatches (\d+)", r.stdout)
    assert r.returncode == 0 and len(rows) == 6, r.stdout + r.stderr
    for T, count, total, per, longest, bad in rows:
        assert int(bad) == 0
        assert float(longest) < 0.5 * float(total), (T, count, total, longest)


def test_cpp_host_mirror_matches_oracle():
    import oracle_c
    oracle_c.build()
    exe = build_driver()
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "crypto_amd") + ":" + os.path.join(ROOT, "oracle") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "all equal" in r.stdout, r.stdout + r.stderr
