"""Host-side C++ of the product (the drop-in API's f64 math, types and KITTI readers) under AddressSanitizer +
UndefinedBehaviorSanitizer, running the reference's host-only test cases.  The reference builds with strict warnings only
(CMakeLists.txt:9); SURVEY.md section 5 asks for sanitizers on the host code."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib")


@pytest.mark.timeout(600)
def test_host_api_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "api_san")
    srcs = sorted(glob.glob(os.path.join(ROOT, "kitti_motion_compensation_amd", "csrc", "api", "*.cpp")))
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"), *srcs, os.path.join(ROOT, "tests", "cpp", "test_dropin_api.cpp"),
           "-L" + LIB, "-lkmc_hip", "-pthread", "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr.lower() or "ubsan" in r.stderr.lower()):
        pytest.skip("sanitizer runtimes not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, "host", os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "0 failures" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
