"""Drives tests/cpp/test_dropin_api.cpp: the reference's gtest cases for the hot path restated against the C++ drop-in API
(include/kitti_motion_compensation/*.hpp -> libkitti_motion_compensation_lib.so -> libkmc_hip.so)."""
import os
import signal
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib", "kmc_api_tests")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _run(*args, timeout=600):
    assert os.path.exists(EXE), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    return subprocess.run([EXE, *args], capture_output=True, text=True, timeout=timeout)


def test_reference_cases_host_side():
    r = _run("host", GOLDEN)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout


def test_out_of_range_time_aborts_like_the_reference_assert():
    """test_trajectory_interpolation.cpp:77-81 EXPECT_DEATH: the reference keeps its assert in release builds."""
    r = _run("death_pose")
    assert r.returncode == -signal.SIGABRT, (r.returncode, r.stderr)
    assert "TimeIsInRange" in r.stderr


def test_drop_in_library_exports_the_reference_symbols():
    lib = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib", "libkitti_motion_compensation_lib.so")
    out = subprocess.run(["nm", "-DC", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    for sym in ("kmc::MotionCompensateFrame(kmc::Frame const&, double)",
                "kmc::MotionCompensatePoint(kmc::trajectory_interpolation::TrajectoryInterpolator const&, double, kmc::Vector4d const&, double)",
                "kmc::GetPseudoTimeStamps(kmc::MatrixX4d const&, double, double)",
                "kmc::trajectory_interpolation::TrajectoryInterpolator::GetPoseAtTime(double) const",
                "kmc::lie::Log(kmc::Affine3d const&)", "kmc::lie::Exp(kmc::Twist const&)",
                "kmc::OxtsToPose(kmc::Oxts const&, double)", "kmc::MotionCompensateRun("):
        assert sym in out, sym
    assert "kmo_" not in out  # never links the oracle


@pytest.mark.gpu
def test_reference_cases_on_the_gpu(tmp_path):
    r = _run("gpu", GOLDEN, str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout
    print(r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["death_frame", "death_point"])
def test_frame_and_point_abort_on_out_of_range_times(mode):
    r = _run(mode)
    assert r.returncode == -signal.SIGABRT, (r.returncode, r.stdout, r.stderr)
    assert "TimeIsInRange" in r.stderr


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = _run("gpu", GOLDEN, "/tmp")
    assert r.returncode != 0
    assert "KMC_ERR_NO_DEVICE" in (r.stdout + r.stderr)

