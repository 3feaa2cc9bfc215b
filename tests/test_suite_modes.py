"""The GPU suite under the two non-default dispatch configurations of a context, as a test of the driver's suite ("a
parametrised CI run, not a notebook entry"):

    KMC_ANY_ORDER=0         every dispatch carries the AQL barrier bit (no barrier-free dispatch of independent frames)
    KMC_DIRECT_DISPATCH=1   every context starts opted in to the direct queue (frames as AQL packets below the HIP runtime)

Each run is a child pytest over the modules whose assertions are about RESULTS (parity against the oracle, bits, error behaviour): the
same assertions must hold whichever way the frames are dispatched.  Modules whose subject is the dispatch configuration itself
(test_dispatch_modes, test_direct_queue, test_hip_rules, the bench contract) set their own environment and are not repeated; the few
tests elsewhere that assert a DEFAULT context's counters are deselected by name below, with the reason."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODULES = ["tests/test_gpu_parity.py", "tests/test_trajectory.py", "tests/test_near_origin.py", "tests/test_special_values.py", "tests/test_projection.py",
           "tests/test_alias_contract.py", "tests/test_inplace_threads.py", "tests/test_cpp_dropin.py", "tests/test_c_boundary.py",
           "tests/test_run_driver.py"]
# asserts what a context created WITHOUT the knob reports (the probe's verdict 1, barrier-free launches > 0): true by construction only there
DESELECT = {
    "KMC_ANY_ORDER": ["tests/test_gpu_parity.py::test_any_order_dispatch_is_gated_by_the_runtime_probe"],
    "KMC_DIRECT_DISPATCH": [],
}


@pytest.mark.gpu
@pytest.mark.parametrize("knob,value", [("KMC_ANY_ORDER", "0"), ("KMC_DIRECT_DISPATCH", "1")])
def test_gpu_suite_under_a_non_default_dispatch_configuration(knob, value):
    if os.environ.get("KMC_SUITE_MODE_CHILD") == "1":
        pytest.skip("already inside a suite-mode child run")
    env = dict(os.environ, KMC_SUITE_MODE_CHILD="1")
    env[knob] = value
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *MODULES]
    for name in DESELECT[knob]:
        cmd += ["--deselect", name]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = r.stdout[-3000:] + r.stderr[-1000:]
    assert r.returncode == 0, f"{knob}={value}:\n{tail}"
    assert " passed" in r.stdout and " failed" not in r.stdout, tail
