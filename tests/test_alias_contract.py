"""The hazard check behind barrier-free dispatch and gathered lists compares virtual address ranges; aliased mappings of one physical
buffer are invisible to it (include/kmc_hip.h, "ALIASES").  tools/alias_probe.hip maps one buffer twice with the HIP virtual-memory API and
runs a two-frame chain THROUGH the alias: with the caller's own ordering (KMC_ANY_ORDER=0, or a join between the frames) every repetition
must give the serial result; without it the result is undefined -- the test records what this device did, it does not assert it."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib", "alias_probe")


@pytest.mark.gpu
def test_aliased_buffers_need_the_callers_own_ordering():
    if not os.path.exists(TOOL):
        pytest.skip("tools/alias_probe was not built (tools/Makefile is best effort)")
    r = subprocess.run([TOOL, "1000000", "40"], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, (r.returncode, r.stdout[-1000:], r.stderr[-2000:])
    d = json.loads(lines[-1])
    if "skipped" in d:
        pytest.skip("HIP virtual-memory API not usable here: " + d["skipped"])
    cfg = d["configurations"]
    assert cfg["KMC_ANY_ORDER=0"]["repetitions_that_differ_from_the_serial_result"] == 0
    assert cfg["gathered_plus_join_between"]["repetitions_that_differ_from_the_serial_result"] == 0
    assert r.returncode == 0
    # (cfg["default"] / cfg["gathered"]: undefined by contract -- on an MI355X with barrier-free dispatch verified they DO differ, see profiles/)
