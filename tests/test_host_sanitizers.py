"""CPU: the library's host-side concurrency under ThreadSanitizer and AddressSanitizer + UBSan (the reference relies on Rust's
borrow checker and `cargo test` under rayon; a C++ host side needs the sanitizers to say the same).  tests/native/host_sanitize_driver.cpp is
built from the PRODUCT's sources — dock_prover.cpp (seven host threads per proof, sharded form included), dock_aggregation.cpp (the aggregation's nested
parallel sections on the worker pool), dock_gt.cpp, dock_serde.cpp,
host_par.hpp and dock_ctx.hpp's slot / handle machinery — with the device entry points replaced by stand-ins that delay and fail on request,
and run: six threads of concurrent proofs with failing stages, lockers / pinners / a freer racing on one handle, the shutdown race of
SlotLock, the threaded GT and codec entry points.  Pass = exit code 0 and no sanitizer report."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "native", "host_sanitize_driver.cpp")] + [os.path.join(ROOT, "crypto_amd", "csrc", f) for f in ("dock_prover.cpp", "dock_gt.cpp", "dock_serde.cpp", "dock_aggregation.cpp")]
HIP_INC = "/opt/rocm/include"


@pytest.mark.parametrize("name,flags,env", [
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1 exitcode=66"}),
    ("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], {"ASAN_OPTIONS": "detect_leaks=1 exitcode=67"}),
    ("plain", [], {}),          # no sanitizer: the one build that also forks with a live worker pool (pthread_atfork handlers of host_par.hpp)
])
def test_host_code_is_clean_under(name, flags, env, tmp_path):
    if not shutil.which("g++") or not os.path.exists(os.path.join(HIP_INC, "hip", "hip_runtime.h")):
        pytest.skip("needs g++ and the HIP headers")
    exe = str(tmp_path / ("hsd_" + name))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-mbmi2", "-madx", "-pthread", "-I" + HIP_INC] + flags + ["-o", exe] + SRC)
    r = subprocess.run([exe], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "host_sanitize_driver: ok" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-4000:])
    assert "Sanitizer" not in r.stderr, r.stderr[-4000:]
