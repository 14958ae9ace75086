"""CPU sanitizer build of the library's host-side helpers (the reference relies on Rust's memory safety;
here the fp64 host code of csrc/host_math.h runs under AddressSanitizer + UndefinedBehaviorSanitizer).  GPU
sanitizers are not available on this pool, so device code is covered by the parity tests only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_math_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_math_san")
    src = os.path.join(ROOT, "tests", "helpers", "host_math_san.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-fno-omit-frame-pointer", src, "-o", exe], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "sanitize" in (r.stderr or "") and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtimes not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.strip().endswith("ok")
