"""Row N4: the LiDAR -> image projection of camera_model.cpp without the OpenCV drawing.
CPU part: the oracle's restatement against an independent numpy twin and against hand-computed cases (the reference has
no test for this file, so there is no known-answer vector to pin it to: "parity unpinned", see oracle/kmc_oracle.h).
GPU part (-m gpu): the HIP kernels through the C-ABI against the oracle, integers bit-exact."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import util

I32_MIN = np.iinfo(np.int32).min


@pytest.fixture(scope="module")
def calib(golden_dir):
    return util.load_kitti_calibration(golden_dir)


@pytest.fixture(scope="module")
def kitti_xyzi(golden_dir):
    return util.load_velodyne_bin(os.path.join(golden_dir, "kitti_2011_09_26_drive_0005"), 0)


def _pinhole(f=700.0, cx=600.0, cy=180.0):
    P = np.array([[f, 0, cx, 0], [0, f, cy, 0], [0, 0, 1, 0]], dtype=np.float64)
    return np.hstack([np.eye(3), np.zeros((3, 1))]), np.eye(3), [P, P, P, P]


# ---------------------------------------------------------------- CPU: the oracle itself
def test_calibration_fixture_values(calib):
    tf, R_rect, P = calib
    assert tf[0, 3] == -4.069766e-03 and tf[2, 0] == 9.998621e-01      # calib_velo_to_cam.txt
    assert R_rect[0, 0] == 9.999239e-01                                # R_rect_00
    assert P[0][0, 0] == 7.215377e+02 and P[0][0, 3] == 0.0            # P_rect_00
    assert P[1][0, 3] == -3.875744e+02 and P[2][0, 3] == 4.485728e+01  # P_rect_01 / _02 baselines
    assert P[3][0, 3] == -3.395242e+02 and P[3][2, 3] == 2.729905e-03


def test_oracle_hand_cases():
    tf, R, P = _pinhole()
    rig = orc.camera_rig(tf, R, P, 15.0)
    pts = np.array([
        [0.0, 0.0, 10.0],      # straight ahead -> principal point
        [1.0, -1.0, 10.0],     # (670, 110)
        [-10.0, 0.0, 10.0],    # u = -100
        [-8.6, 0.0, 10.0],     # u = -2.0000000000000568 -> trunc toward zero = -2
        [-8.58, 0.0, 10.0],    # u = -0.6 -> 0 (truncation, not floor)
        [0.0, 0.0, 0.01],      # nearest drawn depth
        [0.0, 0.0, 15.0],      # farthest drawn depth
        [0.0, 0.0, 0.00999],   # too close
        [0.0, 0.0, 15.0001],   # too far
        [0.0, 1.2500001, 5.0], # below the camera: ground filter
        [0.0, 1.25, 5.0],      # exactly on the limit: drawn
        [0.0, 0.0, -3.0],      # behind
    ])
    uv, bgrv = orc.project_points(pts, rig)
    assert uv[0, 0].tolist() == [600, 180]
    assert uv[1, 0].tolist() == [670, 110]
    assert uv[2, 0].tolist() == [-100, 180]
    assert uv[3, 0, 0] == int((700.0 * -8.6 + 600.0 * 10.0) / 10.0)
    assert uv[4, 0, 0] == 0
    assert bgrv[5].tolist() == [255, 0, 255, 1]     # cs = 0.17
    assert bgrv[6].tolist() == [0, 255, 0, 1]       # cs = 255.17 -> saturates
    for i in (7, 8, 9, 11):
        assert bgrv[i].tolist() == [0, 0, 0, 0] and np.all(uv[i] == I32_MIN)
    assert bgrv[10, 3] == 1
    assert np.all(uv[:, 1:] == uv[:, :1])           # four identical cameras


def test_oracle_half_to_even_colour():
    tf, R, P = _pinhole()
    rig = orc.camera_rig(tf, R, P, 15.0)
    # cs = 255 z / 14.99 ; z chosen so cs is within rounding of k + 0.5 is not constructible exactly in general: check
    # the rounding rule directly on a sweep instead
    z = np.linspace(0.01, 15.0, 20001)
    pts = np.stack([np.zeros_like(z), np.zeros_like(z), z], axis=1)
    _, bgrv = orc.project_points(pts, rig)
    cs = 255.0 * (z / (15.0 - 0.01))
    assert np.array_equal(bgrv[:, 1], np.clip(np.rint(cs), 0, 255).astype(np.uint8))
    assert np.array_equal(bgrv[:, 0], np.clip(np.rint(255.0 - cs), 0, 255).astype(np.uint8))
    assert np.array_equal(bgrv[:, 0], bgrv[:, 2])


def test_oracle_matches_numpy_twin_on_kitti(calib, kitti_xyzi):
    tf, R_rect, P = calib
    rig = orc.camera_rig(tf, R_rect, P, 15.0)
    uv, bgrv = orc.project_xyzi_f32(kitti_xyzi, rig)
    uv2, bgrv2 = util.project_numpy(kitti_xyzi[:, :3].astype(np.float64), tf, R_rect, P, 15.0)
    assert np.array_equal(uv, uv2) and np.array_equal(bgrv, bgrv2)
    drawn = bgrv[:, 3] == 1
    assert 5000 < drawn.sum() < 40000               # the slice of the scan in front of the cameras, within 15 m
    # camera 00 sees them around its 1242 x 375 image
    inside = (uv[drawn, 0, 0] >= 0) & (uv[drawn, 0, 0] < 1242) & (uv[drawn, 0, 1] >= 0) & (uv[drawn, 0, 1] < 375)
    assert inside.mean() > 0.2
    # the f64-column entry agrees with the f32 entry (widening is exact)
    uv3, bgrv3 = orc.project_points(kitti_xyzi[:, :3].astype(np.float64), rig)
    assert np.array_equal(uv, uv3) and np.array_equal(bgrv, bgrv3)


def test_oracle_special_values():
    tf, R, P = _pinhole()
    rig = orc.camera_rig(tf, R, P, 15.0)
    pts = np.array([[np.nan, 0, 5], [0, 0, np.nan], [1e300, 0, 5.0], [np.inf, 0, 5.0], [0.0, -np.inf, 5.0]])
    uv, bgrv = orc.project_points(pts, rig)
    uv2, bgrv2 = util.project_numpy(pts, tf, R, P, 15.0)
    assert np.array_equal(uv, uv2) and np.array_equal(bgrv, bgrv2)
    assert uv[0, 0, 0] == I32_MIN and bgrv[0, 3] == 1      # NaN x: every comparison of :21-24 is false -> "drawn" at INT_MIN
    assert uv[2, 0, 0] == I32_MIN                            # does not fit an int: cvttsd2si's indefinite value


# ---------------------------------------------------------------- GPU: HIP kernels vs the oracle
@pytest.fixture(scope="module")
def ctx():
    import torch

    assert torch.cuda.is_available(), "these tests need the GPU: there is no CPU fallback to test"
    from kitti_motion_compensation_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


def _rigs(calib, max_range=15.0):
    from kitti_motion_compensation_amd import capi

    tf, R_rect, P = calib
    return capi.CameraRig.make(tf, R_rect, P, max_range), orc.camera_rig(tf, R_rect, P, max_range)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 4097, 123397])
def test_gpu_project_f32_host_bit_exact(ctx, calib, kitti_xyzi, n):
    rig, orig = _rigs(calib)
    pts = np.ascontiguousarray(kitti_xyzi[:n])
    uv = np.full((n, 4, 2), 7, dtype=np.int32)
    bgrv = np.full((n, 4), 7, dtype=np.uint8)
    st = ctx.project_f32(pts, rig, uv, bgrv)
    assert st.n_points == n
    uv_ref, bgrv_ref = orc.project_xyzi_f32(pts, orig)
    assert np.array_equal(uv, uv_ref) and np.array_equal(bgrv, bgrv_ref)


@pytest.mark.gpu
def test_gpu_project_f32_against_numpy_alone_no_oracle(ctx, golden_dir, kitti_xyzi):
    """the HIP projection against an INDEPENDENT second implementation -- numpy f64 straight from the shipped calibration
    text files (tests/golden/calib_*.txt) and the shipped scan, camera_model.cpp:5-36's operations one rounded step at a time -- with
    nothing of oracle/ in the loop.  It does not pin the row (the reference holds no vector for camera_model.cpp) but it removes the
    "oracle against itself" structure: two implementations that share no code agree on every integer."""
    from kitti_motion_compensation_amd import capi

    tf, R_rect, P = util.load_kitti_calibration(golden_dir)  # parses the calibration text, no oracle
    for max_range in (15.0, 40.0):
        rig = capi.CameraRig.make(tf, R_rect, P, max_range)
        n = kitti_xyzi.shape[0]
        uv = np.zeros((n, 4, 2), dtype=np.int32)
        bgrv = np.zeros((n, 4), dtype=np.uint8)
        ctx.project_f32(kitti_xyzi, rig, uv, bgrv)
        uv_np, bgrv_np = util.project_numpy(kitti_xyzi[:, :3].astype(np.float64), tf, R_rect, P, max_range)
        drawn = bgrv_np[:, 3] == 1
        in_view = drawn & (uv_np[:, 0, 0] >= 0) & (uv_np[:, 0, 0] < 1242) & (uv_np[:, 0, 1] >= 0) & (uv_np[:, 0, 1] < 375)
        assert in_view.sum() >= 1000, in_view.sum()  # thousands of points of the shipped scan land inside camera 00's image
        assert np.array_equal(uv, uv_np) and np.array_equal(bgrv, bgrv_np)


@pytest.mark.gpu
def test_gpu_project_f32_device_and_f64cols(ctx, calib, kitti_xyzi):
    import torch

    rig, orig = _rigs(calib, 40.0)
    n = kitti_xyzi.shape[0]
    d_in = torch.from_numpy(kitti_xyzi).cuda()
    d_uv = torch.zeros((n, 4, 2), dtype=torch.int32, device="cuda")
    d_col = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.project_f32(d_in, rig, d_uv, d_col)
    torch.cuda.synchronize()
    uv_ref, bgrv_ref = orc.project_xyzi_f32(kitti_xyzi, orig)
    assert np.array_equal(d_uv.cpu().numpy(), uv_ref) and np.array_equal(d_col.cpu().numpy(), bgrv_ref)
    # Eigen-layout entry: three f64 columns
    cols = [np.ascontiguousarray(kitti_xyzi[:, k].astype(np.float64)) for k in range(3)]
    uv = np.zeros((n, 4, 2), dtype=np.int32)
    bgrv = np.zeros((n, 4), dtype=np.uint8)
    ctx.project_f64cols(cols[0], cols[1], cols[2], rig, uv, bgrv)
    assert np.array_equal(uv, uv_ref) and np.array_equal(bgrv, bgrv_ref)
    d_cols = [torch.from_numpy(c).cuda() for c in cols]
    d_uv.zero_(); d_col.zero_()
    ctx.project_f64cols(d_cols[0], d_cols[1], d_cols[2], rig, d_uv, d_col)
    torch.cuda.synchronize()
    assert np.array_equal(d_uv.cpu().numpy(), uv_ref) and np.array_equal(d_col.cpu().numpy(), bgrv_ref)
    ctx.set_stream(None)


@pytest.mark.gpu
@pytest.mark.parametrize("twist", [[1.3, 0.05, -0.02, 0.001, -0.002, 0.03], [0.4, 0.1, 0.0, 0.1, -0.3, 0.6], [0.4, 0.1, 0.0, 0.3, -0.9, 2.2]])
def test_gpu_fused_deskew_then_project(ctx, calib, kitti_xyzi, twist):
    """project(deskew(p)) in one kernel: the cloud it writes is bit-identical to kmc_hip_deskew_f32's, and the pixels are
    the oracle's projection of exactly that cloud.  Against the f64 oracle deskew (motion_compensation.cpp:16-28) the
    pixels may differ where the f32 result straddles a pixel boundary: at most 1 px (relative 1/4096 for the far
    off-image pixels of points next to the camera plane), on < 0.5 % of the drawn points."""
    from kitti_motion_compensation_amd import capi

    rig, orig = _rigs(calib)
    n = kitti_xyzi.shape[0]
    params = capi.FrameParams.make(twist, 0.5)
    plain = np.empty_like(kitti_xyzi)
    ctx.deskew_f32(kitti_xyzi, plain, params)
    cloud = np.empty_like(kitti_xyzi)
    uv = np.zeros((n, 4, 2), dtype=np.int32)
    bgrv = np.zeros((n, 4), dtype=np.uint8)
    ctx.project_f32(kitti_xyzi, rig, uv, bgrv, deskew=params, xyzi_out=cloud)
    assert np.array_equal(cloud.view(np.uint32), plain.view(np.uint32))
    uv_ref, bgrv_ref = orc.project_xyzi_f32(cloud, orig)
    assert np.array_equal(uv, uv_ref) and np.array_equal(bgrv, bgrv_ref)
    # without the cloud output
    uv2 = np.zeros_like(uv); bgrv2 = np.zeros_like(bgrv)
    ctx.project_f32(kitti_xyzi, rig, uv2, bgrv2, deskew=params)
    assert np.array_equal(uv2, uv) and np.array_equal(bgrv2, bgrv)
    # against the reference's own sequence in f64: deskew (oracle, faithful) then project
    T0, T1 = 47072.283701593, 47072.386973931
    ident = orc.se3_exp([0, 0, 0, 0, 0, 0])
    res = orc.deskew_xyzi_f32(kitti_xyzi, T0, ident, T1, orc.se3_exp(twist), T0 + 0.5 * (T1 - T0), mode=orc.FAITHFUL)
    uv64, bgrv64 = orc.project_points(res["xyz_f64"], orig)
    both = (bgrv[:, 3] == 1) & (bgrv64[:, 3] == 1)
    d = np.abs(uv[both].astype(np.int64) - uv64[both].astype(np.int64))
    mag = np.abs(uv64[both].astype(np.int64))
    assert np.all(d <= 1 + mag // 4096)          # 1 px; points centimetres from the camera plane land at |u| ~ 1e4..1e6 px
    assert (d > 0).mean() < 5e-3
    assert (bgrv[:, 3] != bgrv64[:, 3]).mean() < 1e-3


@pytest.mark.gpu
def test_gpu_project_synthetic_1m_and_special_values(ctx, kitti_xyzi):
    """1 M synthetic points, a random (but plausible) rig, plus NaN / Inf / out-of-int-range coordinates."""
    from kitti_motion_compensation_amd import capi

    rng = np.random.default_rng(7)
    n = 1_000_000
    pts = capi.synth_points_host(n, 0x4B4D43)
    pts[:8, 0] = [np.nan, np.inf, -np.inf, 1e30, -1e30, 0.0, 3e38, 1.0]
    pts[8:12, 2] = [np.nan, np.inf, -np.inf, 0.0]
    a = 0.02 * rng.standard_normal(3)
    Rz = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], dtype=np.float64)  # velodyne -> camera axes
    tf = np.hstack([Rz + 0.01 * rng.standard_normal((3, 3)), rng.standard_normal((3, 1)) * 0.3])
    R_rect = np.eye(3) + np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    P = [np.array([[721.5 + c, 0, 609.5, -387.0 * c], [0, 721.5 + c, 172.8, 0.3 * c], [0, 0, 1, 0.002 * c]]) for c in range(4)]
    rig, orig = capi.CameraRig.make(tf, R_rect, P, 80.0), orc.camera_rig(tf, R_rect, P, 80.0)
    uv = np.zeros((n, 4, 2), dtype=np.int32)
    bgrv = np.zeros((n, 4), dtype=np.uint8)
    ctx.project_f32(pts, rig, uv, bgrv)
    uv_ref, bgrv_ref = orc.project_xyzi_f32(pts, orig)
    assert np.array_equal(bgrv, bgrv_ref)
    assert np.array_equal(uv, uv_ref)
    assert (bgrv[:, 3] == 1).sum() > 100_000


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["shared", "pinhole", "dense"])
def test_gpu_project_exact_integer_pixels(ctx, kind):
    """Points that project EXACTLY onto integer pixel coordinates (and the int-range ends): the lanes whose approximate
    quotient sits on an integer must take the IEEE-division path and still agree bit for bit.  Three kernel variants: all four
    cameras with the same intrinsics (fx x + cx z shared), pinhole cameras with different intrinsics, dense P_rect."""
    from kitti_motion_compensation_amd import capi

    tf, R, P = _pinhole(f=512.0, cx=600.0, cy=180.0)
    dense = kind == "dense"
    if kind == "pinhole":  # pinhole zeros kept, but every camera its own focal lengths / principal point / baseline
        P = [p + np.array([[16.0 * c, 0, 4.0 * c, -30.0 * c], [0, 8.0 * c, 2.0 * c, 0.5 * c], [0, 0, 0, 0.001 * c]]) for c, p in enumerate(P)]
    if dense:  # a P_rect without the pinhole zeros -> the general kernel variant
        P = [p + np.array([[0, 0.25, 0, 0], [0.5, 0, 0, 0], [0, 0, 0, 0.0]]) for p in P]
        P[2] = P[2] + np.array([[0, 0, 0, 0], [0, 0, 0, 0], [0.001, 0.002, 0, 0.003]])
    u = np.arange(-300, 300, dtype=np.float64)
    z = np.array([0.5, 1.0, 2.0, 4.0, 8.0])
    X, Z = np.meshgrid(u / 512.0, z)                 # u_px = 512 * (X * Z) / Z + 600: integer for power-of-two Z
    pts = np.stack([(X * Z).ravel(), (X * Z).ravel() * 0.5, Z.ravel(), np.zeros(X.size)], axis=1).astype(np.float32)
    extra = np.array([[4194303.5 * 0.01 / 1.0, 0, 0.01, 0], [-4194305.0 * 0.01, 0, 0.01, 0], [1e7, 0, 0.01, 0]], dtype=np.float32)
    pts = np.ascontiguousarray(np.vstack([pts, extra]))
    n = pts.shape[0]
    rig, orig = capi.CameraRig.make(tf, R, P, 15.0), orc.camera_rig(tf, R, P, 15.0)
    uv = np.zeros((n, 4, 2), dtype=np.int32)
    bgrv = np.zeros((n, 4), dtype=np.uint8)
    ctx.project_f32(pts, rig, uv, bgrv)
    uv_ref, bgrv_ref = orc.project_xyzi_f32(pts, orig)
    assert np.array_equal(uv, uv_ref) and np.array_equal(bgrv, bgrv_ref)
    if not dense:
        assert np.array_equal(uv[:600, 0, 0], np.arange(300, 900))   # z = 0.5 row: exact integers (camera 0 keeps f = 512, cx = 600)


@pytest.mark.gpu
def test_gpu_project_colour_on_half_integers(ctx):
    """cs = 255 z / (max_range - 0.01) EXACTLY on k + 0.5 for k = 0 .. 254 (max_range - 0.01 == 127.5, z = (k + 0.5) / 2): the bytes
    are cv::saturate_cast's half-to-even roundings of cs and 255 - cs.  The kernel guesses the quotient through a reciprocal and
    must notice that these points sit on a rounding boundary and redo the IEEE division (camera_model.cpp:28-32)."""
    from kitti_motion_compensation_amd import capi

    max_range = 127.5 + 0.01
    assert max_range - 0.01 == 127.5
    tf, R, P = _pinhole()
    k = np.arange(255, dtype=np.float64)
    z = (k + 0.5) / 2.0
    pts = np.zeros((3 * 255, 4), dtype=np.float32)
    pts[:, 2] = np.concatenate([z, np.nextafter(z.astype(np.float32), np.float32(np.inf)), np.nextafter(z.astype(np.float32), np.float32(-np.inf))])
    pts[:, 0] = 0.01 * pts[:, 2]
    rig, orig = capi.CameraRig.make(tf, R, P, max_range), orc.camera_rig(tf, R, P, max_range)
    n = pts.shape[0]
    uv = np.zeros((n, 4, 2), dtype=np.int32)
    bgrv = np.zeros((n, 4), dtype=np.uint8)
    ctx.project_f32(pts, rig, uv, bgrv)
    uv_ref, bgrv_ref = orc.project_xyzi_f32(pts, orig)
    assert np.array_equal(uv, uv_ref) and np.array_equal(bgrv, bgrv_ref)
    assert bgrv_ref[:255, 3].all()
    # half to even on the exact half-integers: cs = k + 0.5 -> k if k is even, k + 1 if odd
    assert np.array_equal(bgrv_ref[:255, 1], np.where(k % 2 == 0, k, k + 1).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense", "pinhole"])
def test_gpu_project_dense_rig_1m(ctx, kind):
    from kitti_motion_compensation_amd import capi

    rng = np.random.default_rng(11)
    n = 1_000_000
    pts = capi.synth_points_host(n, 99)
    pts[:4, 1] = [np.nan, np.inf, -np.inf, 1e38]
    tf = np.hstack([np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]]) + 0.05 * rng.standard_normal((3, 3)), rng.standard_normal((3, 1))])
    R_rect = np.eye(3) + 0.02 * rng.standard_normal((3, 3))
    P = [np.array([[700.0, 0, 600, 40], [0, 700, 180, 0], [0, 0, 1, 0]]) + 0.01 * rng.standard_normal((3, 4)) for _ in range(4)]
    if kind == "pinhole":  # four different pinhole cameras: the zeros and the one of [fx 0 cx tx; 0 fy cy ty; 0 0 1 tz] restored
        for p in P:
            p[0, 1] = p[1, 0] = p[2, 0] = p[2, 1] = 0.0
            p[2, 2] = 1.0
    rig, orig = capi.CameraRig.make(tf, R_rect, P, 60.0), orc.camera_rig(tf, R_rect, P, 60.0)
    uv = np.zeros((n, 4, 2), dtype=np.int32)
    bgrv = np.zeros((n, 4), dtype=np.uint8)
    ctx.project_f32(pts, rig, uv, bgrv)
    uv_ref, bgrv_ref = orc.project_xyzi_f32(pts, orig)
    assert np.array_equal(uv, uv_ref) and np.array_equal(bgrv, bgrv_ref)


@pytest.mark.gpu
def test_gpu_project_argument_checks(ctx, calib, kitti_xyzi):
    from kitti_motion_compensation_amd import capi

    rig, _ = _rigs(calib)
    n = 128
    pts = np.ascontiguousarray(kitti_xyzi[:n])
    uv = np.zeros((n, 4, 2), dtype=np.int32)
    bgrv = np.zeros((n, 4), dtype=np.uint8)
    with pytest.raises(capi.KmcError) as e:   # a cloud output needs a deskew
        ctx.project_f32(pts, rig, uv, bgrv, xyzi_out=np.empty_like(pts))
    assert e.value.status == capi.ERR_INVALID_ARG
    bad = capi.CameraRig.make(calib[0], calib[1], calib[2], float("nan"))
    with pytest.raises(capi.KmcError) as e:
        ctx.project_f32(pts, bad, uv, bgrv)
    assert e.value.status == capi.ERR_INVALID_ARG
    with pytest.raises(capi.KmcError) as e:   # requested time outside the scan
        ctx.project_f32(pts, rig, uv, bgrv, deskew=capi.FrameParams.make([1, 0, 0, 0, 0, 0.1], 1.5))
    assert e.value.status == capi.ERR_TIME_OUT_OF_RANGE


def test_pixels_do_not_depend_on_the_summation_order_on_real_data(calib, kitti_xyzi):
    """The oracle sums every dot product left to right without FMA, which is how Eigen evaluates these small products in a
    plain -O3 build; the reference has no test that would pin that choice.  How much could it matter?  The same projection
    through BLAS matrix products (blocked / FMA order) must give the same integers on the shipped scan -- a pixel only changes
    if a quotient sits within ~1e-13 of an integer."""
    tf, R_rect, P = calib
    rig = orc.camera_rig(tf, R_rect, P, 15.0)
    uv, bgrv = orc.project_xyzi_f32(kitti_xyzi, rig)
    ph = np.hstack([kitti_xyzi[:, :3].astype(np.float64), np.ones((kitti_xyzi.shape[0], 1))])
    T4 = np.vstack([tf, [0, 0, 0, 1]])
    R4 = np.eye(4)
    R4[:3, :3] = R_rect
    rect = (R4 @ (T4 @ ph.T)).T
    drawn = ~((rect[:, 2] < 0.01) | (rect[:, 2] > 15.0) | (rect[:, 1] > 1.25))
    assert np.array_equal(drawn, bgrv[:, 3] == 1)
    mismatches = 0
    for cam in range(4):
        pix = (P[cam] @ rect.T).T
        q = pix[:, :2] / pix[:, 2:3]
        mismatches += int(np.count_nonzero(np.trunc(q[drawn]).astype(np.int64) != uv[drawn, cam]))
    assert mismatches == 0
