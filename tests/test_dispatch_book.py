"""The bookkeeping of barrier-free dispatch and of gathered calls (kmc_dispatch_book.hpp) is HIP-free on purpose: this test builds its
C++ unit test with the system compiler and runs it on the CPU -- a fake stream answers "have you run dry?" from a script
(VERDICT r04 #7); the direct queue's two-lane protocol (LaneWindow + LaneSync) and the one-queue window are run, as random programs of
frames, through a MODEL of AQL queues with random kernel durations: no frame may start before a frame it conflicts with has completed.
The GPU suite checks the same promises end to end (test_gathered_calls_keep_in_order_results, test_frame_queues_...)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dispatch_book_unit_tests_pass_on_the_cpu(tmp_path):
    exe = str(tmp_path / "test_dispatch_book")
    src = os.path.join(ROOT, "tests", "cpp", "test_dispatch_book.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsanitize=undefined", "-fno-sanitize-recover=all", "-o", exe, src],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
