import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The shared libraries are build artefacts (git-ignored).  If this checkout has none, or they are older than their
    # sources, build them once (hipcc cross-compiles gfx950 without a GPU; ~20 s).
    lib_dir = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib")
    needed = [os.path.join(lib_dir, n) for n in ("libkmc_hip.so", "libkitti_motion_compensation_lib.so", "kmc_api_tests")]
    needed.append(os.path.join(ROOT, "oracle", "libkmc_oracle.so"))
    src_dirs = [os.path.join(ROOT, "kitti_motion_compensation_amd", "csrc"), os.path.join(ROOT, "include"),
                os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "cpp")]
    newest_src = 0.0
    for d in src_dirs:
        for dp, _, files in os.walk(d):
            for fn in files:
                if fn.endswith((".hip", ".h", ".hpp", ".cpp", ".c", "Makefile")):
                    newest_src = max(newest_src, os.path.getmtime(os.path.join(dp, fn)))
    if any(not os.path.exists(p) or os.path.getmtime(p) < newest_src for p in needed):
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
        return json.load(f)
