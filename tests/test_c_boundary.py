"""include/kmc_hip.h is a C header: a plain C99 translation unit must compile against it (no C++ types leaking through the
boundary), link libkmc_hip.so and run the GPU-free part; with a GPU the same program deskews the shipped KITTI frame."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cabi") / "deskew_bin")
    cmd = ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200112L", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "deskew_bin.c"), "-L" + LIB, "-lkmc_hip", "-lm", "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib",
           "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_plain_c99_client_compiles_links_and_runs_the_host_prestep(exe):
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host pre-step ok" in r.stdout


@pytest.mark.gpu
def test_plain_c_client_deskews_the_kitti_frame(exe, tmp_path, golden_dir):
    from oracle import oracle as orc
    from tests import util

    src = os.path.join(golden_dir, "kitti_2011_09_26_drive_0005", "velodyne_points", "data", "0000000000.bin")
    dst = str(tmp_path / "out.bin")
    twist = [1.3, 0.05, -0.02, 0.002, -0.004, 0.03]
    r = subprocess.run([exe, src, dst] + [repr(v) for v in twist], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "1 launch(es)" in r.stdout and "in place on page-locked buffers" in r.stdout  # pool buffers: ONE kernel over the link, no staging
    xyzi = np.fromfile(src, dtype=np.float32).reshape(-1, 4)
    got = np.fromfile(dst, dtype=np.float32).reshape(-1, 4)
    P1 = orc.Affine.identity()
    P2 = orc.se3_exp(twist)
    ref = orc.deskew_xyzi_f32(xyzi, 0.0, P1, 0.1, P2, 0.05, mode=orc.FAITHFUL)
    assert util.rel_point_error(got[:, :3], ref["xyz_f64"]).max() <= 1e-5
    assert np.array_equal(got[:, 3], xyzi[:, 3])


@pytest.fixture(scope="module")
def project_exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cabi") / "project_bin")
    cmd = ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200112L", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "project_bin.c"), "-L" + LIB, "-lkmc_hip", "-lm", "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib",
           "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_plain_c99_projection_client_compiles_and_links(project_exe):
    r = subprocess.run([project_exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_plain_c_client_projects_the_kitti_frame(project_exe, tmp_path, golden_dir):
    from oracle import oracle as orc
    from tests import util

    src = os.path.join(golden_dir, "kitti_2011_09_26_drive_0005", "velodyne_points", "data", "0000000000.bin")
    dst = str(tmp_path / "out.uv")
    r = subprocess.run([project_exe, src, golden_dir, dst, "15"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    xyzi = np.fromfile(src, dtype=np.float32).reshape(-1, 4)
    n = xyzi.shape[0]
    raw = np.fromfile(dst, dtype=np.uint8)
    uv = raw[:32 * n].view(np.int32).reshape(n, 4, 2)
    bgrv = raw[32 * n:].reshape(n, 4)
    tf, R_rect, P = util.load_kitti_calibration(golden_dir)
    uv_ref, bgrv_ref = orc.project_xyzi_f32(xyzi, orc.camera_rig(tf, R_rect, P, 15.0))
    assert np.array_equal(uv, uv_ref) and np.array_equal(bgrv, bgrv_ref)
    assert f"{n} points, {int(bgrv_ref[:, 3].sum())} drawn" in r.stdout
