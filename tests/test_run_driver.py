"""kmc::MotionCompensateRun (the reference's handlers.cpp:41-65) end to end: the `motion_compensate_runs` CLI on a KITTI-raw-shaped
run directory, every file it writes checked against the ORACLE -- oracle MakeFrame (data_io.cpp:253-269) + the FAITHFUL
per-point sequence (motion_compensation.cpp:9-28) + the f32 cast of WritePointcloud (data_io.cpp:300-310) -- not against
another path of this repository.  Also: the first / last frame copies (handlers.cpp:19-39, slip included), the batch size
must not matter, the N-knot mode against the oracle's chain, and the multi-device driver (two device contexts on the one GPU
of the test box) must write the very same bytes as the single-device run."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as orc
from tests import util

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib", "motion_compensate_runs")
N_FRAMES = 24
REL_TOL = 1e-5  # north_star's bar on XYZ


@pytest.fixture(scope="module")
def run_dir(tmp_path_factory):
    out = tmp_path_factory.mktemp("kitti_raw")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_run.py"), str(out), str(N_FRAMES)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = r.stdout.strip().splitlines()[-1]
    assert os.path.isdir(run)
    return str(out), os.path.basename(run), run


def _run_cli(run_dir, **env):
    data_dir, name, run = run_dir
    out_dir = os.path.join(run, "velodyne_points", "data_motion_compensated")
    if os.path.isdir(out_dir):
        for f in os.listdir(out_dir):
            os.remove(os.path.join(out_dir, f))
    e = dict(os.environ)
    for k in ("KMC_RUN_BATCH_FRAMES", "KMC_RUN_KNOTS", "KMC_DEVICES", "KMC_FIX_LAST_FRAME_COPY"):
        e.pop(k, None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([CLI, data_dir + "/", name], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    files = sorted(os.listdir(out_dir))
    assert files == ["%010d.bin" % i for i in range(N_FRAMES)]
    return {i: np.fromfile(os.path.join(out_dir, "%010d.bin" % i), dtype=np.float32).reshape(-1, 4) for i in range(N_FRAMES)}, r.stdout


def _stamps(run):
    vp = os.path.join(run, "velodyne_points")
    return [[util.load_timestamp(os.path.join(vp, f), i) for i in range(N_FRAMES)] for f in ("timestamps_start.txt", "timestamps.txt", "timestamps_end.txt")]


def _oxts(run, i):
    return orc.oxts(**util.load_oxts_fields(run, i))


def _check_against_oracle(run, written, three_knots=False):
    t_start, t_mid, t_end = _stamps(run)
    worst = 0.0
    for i in range(1, N_FRAMES - 1):
        raw = util.load_velodyne_bin(run, i)
        got = written[i]
        assert got.shape == raw.shape, i
        assert np.array_equal(got[:, 3].view(np.uint32), raw[:, 3].view(np.uint32)), f"frame {i}: intensity not bit-identical"
        o = [_oxts(run, i - 1), _oxts(run, i), _oxts(run, i + 1)]
        if three_knots:
            ref = orc.deskew_xyzi_f32_traj(raw, t_start[i], t_end[i], [x.stamp for x in o], [orc.oxts_to_pose(x) for x in o], t_mid[i])
        else:
            rc, T_start, T_end = orc.make_frame_poses(o[0], o[1], o[2], t_start[i], t_end[i])  # MakeFrame, data_io.cpp:253-269
            assert rc == orc.OK
            ref = orc.deskew_xyzi_f32(raw, t_start[i], T_start, t_end[i], T_end, t_mid[i], mode=orc.FAITHFUL, want_f32=True)
        assert ref["rc"] == orc.OK
        err = util.rel_point_error(got[:, :3], ref["xyz_f64"])
        assert err.max() <= REL_TOL, f"frame {i}: {err.max():.3e}"
        worst = max(worst, float(err.max()))
        if not three_knots:  # the bytes the reference would have written (its f64 result cast to f32): within a few f32 ulps of the point's norm
            d = np.abs(got[:, :3].astype(np.float64) - ref["xyzi_f32"][:, :3].astype(np.float64)).max(axis=1)
            assert (d <= 8 * 2.0**-24 * np.maximum(np.linalg.norm(ref["xyz_f64"], axis=1), 1e-3)).all(), i
    return worst


def test_files_written_by_the_run_driver_match_the_oracle(run_dir):
    run = run_dir[2]
    baseline, stdout = _run_cli(run_dir)
    assert all(f"Motion compensated pointcloud number: {i}" in stdout for i in range(1, N_FRAMES - 1))  # handlers.cpp:63
    worst = _check_against_oracle(run, baseline)
    assert worst > 0.0  # the run does move points: a pass-through would not be a test
    first = util.load_velodyne_bin(run, 0)
    assert np.array_equal(baseline[0].view(np.uint32), first.view(np.uint32))                # handlers.cpp:25-33
    assert np.array_equal(baseline[N_FRAMES - 1].view(np.uint32), first.view(np.uint32))     # handlers.cpp:36-38: the reference's slip
    # the batch size is an implementation detail: same bytes whatever it is
    for batch in (1, 7, 64):
        again, _ = _run_cli(run_dir, KMC_RUN_BATCH_FRAMES=batch)
        for i in range(N_FRAMES):
            assert np.array_equal(again[i].view(np.uint32), baseline[i].view(np.uint32)), (batch, i)
    fixed, _ = _run_cli(run_dir, KMC_FIX_LAST_FRAME_COPY=1)
    assert np.array_equal(fixed[N_FRAMES - 1].view(np.uint32), util.load_velodyne_bin(run, N_FRAMES - 1).view(np.uint32))


def test_three_knot_run_matches_the_oracle_chain(run_dir):
    written, _ = _run_cli(run_dir, KMC_RUN_KNOTS=3, KMC_RUN_BATCH_FRAMES=5)
    _check_against_oracle(run_dir[2], written, three_knots=True)


def test_multi_device_run_writes_the_same_files(run_dir):
    """KMC_DEVICES=0,0(,0): several workers, each with its own device context and its own contiguous frame range -- the code
    path of a multi-GPU node, on the one GPU this box has.  Same bytes as the single-device run, every frame written once."""
    single, _ = _run_cli(run_dir, KMC_RUN_BATCH_FRAMES=4)
    for devices in ("0,0", "0,0,0", "0,0,0,0,0,0,0,0"):
        multi, stdout = _run_cli(run_dir, KMC_DEVICES=devices, KMC_RUN_BATCH_FRAMES=4, KMC_RUN_TIMING=1)
        for i in range(N_FRAMES):
            assert np.array_equal(multi[i].view(np.uint32), single[i].view(np.uint32)), (devices, i)
        lines = [l for l in stdout.splitlines() if l.startswith("Motion compensated pointcloud number:")]
        assert sorted(int(l.split(":")[1]) for l in lines) == list(range(1, N_FRAMES - 1)), devices
    _check_against_oracle(run_dir[2], multi)


def test_a_page_locking_failure_ends_the_run_with_an_error_or_a_fallback_never_a_hang(run_dir):
    """the run's page-locked buffers come from a helper thread; when it failed after its FIRST buffer (a memlock limit, a large
    batch) the reader and the GPU thread waited for each other for ever.  KMC_TEST_HOST_POOL_FAIL_AT=k makes the process's k-th pool
    allocation fail: whichever allocation that hits, the run must END -- with the very same files (the failure hit a buffer the run can
    make another way: the pool declining from the start falls back to buffers of the device context) or with the error on stderr and
    exit code 1.  At least one k must hit the part-way case."""
    data_dir, name, run = run_dir
    baseline, _ = _run_cli(run_dir, KMC_RUN_BATCH_FRAMES=4)
    out_dir = os.path.join(run, "velodyne_points", "data_motion_compensated")
    part_way = 0
    for k in range(1, 9):
        for f in os.listdir(out_dir):
            os.remove(os.path.join(out_dir, f))
        e = dict(os.environ, KMC_RUN_BATCH_FRAMES="4", KMC_TEST_HOST_POOL_FAIL_AT=str(k))
        r = subprocess.run([CLI, data_dir + "/", name], capture_output=True, text=True, env=e, timeout=120)  # (a hang ends here, as a TimeoutExpired)
        if r.returncode == 0:
            for i in range(N_FRAMES):
                got = np.fromfile(os.path.join(out_dir, "%010d.bin" % i), dtype=np.float32).reshape(-1, 4)
                assert np.array_equal(got.view(np.uint32), baseline[i].view(np.uint32)), (k, i)
        else:
            assert r.returncode == 1 and "error:" in r.stderr, (k, r.stderr[-500:])
            if "page-locking buffer" in r.stderr:
                part_way += 1
    assert part_way >= 1
