"""Pins the CPU oracle (oracle/kmc_oracle.c) against every known-answer test the reference holds for the
hot path.  Expected values live in tests/golden/reference_kats.json with the
reference test file:line they were read from.  CPU only."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import util


def _rot_x_pose(angle, tx):
    c, s = np.cos(angle), np.sin(angle)
    R = np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)
    return orc.Affine.from_Rt(R, [tx, 0.0, 0.0])


def _kat_frame(k):
    ox = [orc.oxts(**o) for o in k["oxts"]]
    rc, T_start, T_end = orc.make_frame_poses(ox[0], ox[1], ox[2], k["stamp_start"], k["stamp_end"])
    assert rc == orc.OK
    return T_start, T_end


# ---- test/test_motion_compensation.cpp:54-76 -------------------------------------------------------
def test_motion_compensate_frame_kat(kats):
    k = kats["motion_compensate_frame"]
    T_start, T_end = _kat_frame(k)
    cloud = np.array(k["cloud"])
    stamps = orc.pseudo_timestamps(cloud, k["stamp_start"], k["stamp_end"])
    rc, n_bad, out = orc.motion_compensate_frame(cloud, stamps, k["stamp_start"], T_start, k["stamp_end"], T_end,
                                                 k["requested_time"])
    assert rc == orc.OK and n_bad == 0
    util.assert_float_eq(out, np.array(k["expected"]), "MotionCompensateFrame KAT")


def test_motion_compensate_frame_kat_pipeline_modes(kats):
    """Same KAT through the KITTI-layout helper, FAITHFUL and HOISTED (closed form) modes."""
    k = kats["motion_compensate_frame"]
    T_start, T_end = _kat_frame(k)
    xyzi = np.array(k["cloud"], dtype=np.float32)
    xyzi[:, 3] = [0.1, 0.2, 0.3]
    exp = np.array(k["expected"])[:, :3]
    for mode in (orc.FAITHFUL, orc.HOISTED):
        for threads in (1, 2):
            r = orc.deskew_xyzi_f32(xyzi, k["stamp_start"], T_start, k["stamp_end"], T_end, k["requested_time"],
                                    mode=mode, threads=threads, want_f32=True, want_stamps=True)
            assert r["rc"] == orc.OK
            util.assert_float_eq(r["xyz_f64"], exp)
            util.assert_float_eq(r["xyzi_f32"][:, :3], exp)
            assert np.array_equal(r["xyzi_f32"][:, 3], xyzi[:, 3])  # intensity bit-identical
            util.assert_float_eq(r["stamps"], [0.125, 0.15, 0.175])


# ---- test/test_timestamp_mocking.cpp ---------------------------------------------------------------
def test_fraction_of_scan_completed_kat(kats):
    k = kats["timestamp_mocking"]
    for p, e in zip(k["cloud"], k["expected_fraction"]):
        util.assert_float_eq(orc.fraction_of_scan_completed(p), e)


def test_pseudo_timestamp_kat(kats):
    k = kats["timestamp_mocking"]
    for p, e in zip(k["cloud"], k["expected_stamp"]):
        util.assert_float_eq(orc.pseudo_timestamp(p, k["scan_start"], k["scan_end"]), e)
    util.assert_float_eq(orc.pseudo_timestamps(np.array(k["cloud"]), k["scan_start"], k["scan_end"]), k["expected_stamp"])


def test_fraction_edge_semantics():
    """timestamp_mocking.cpp:46 has no special cases: atan2(0,0)=0 -> 0.5; y=-0,x<0 -> 1.0; y=+0,x<0 -> 0.0"""
    assert orc.fraction_of_scan_completed([0.0, 0.0, 0.0, 1.0]) == 0.5
    assert orc.fraction_of_scan_completed([-1.0, -0.0, 0.0, 1.0]) == 1.0
    assert orc.fraction_of_scan_completed([-1.0, 0.0, 0.0, 1.0]) == 0.0


# ---- test/test_lie_algebra.cpp -----------------------------------------------------------------------
def test_hat_vee_inverse(kats):
    phi = np.array(kats["lie_algebra"]["phi"])
    util.assert_float_eq(orc.vee(orc.hat(phi)), phi)
    H = orc.hat(phi)
    assert np.array_equal(H, -H.T)
    assert H[0, 1] == -phi[2] and H[0, 2] == phi[1] and H[1, 2] == -phi[0]  # lie_algebra.cpp:10-15


def test_so3_log_exp_inverse(kats):
    phi = np.array(kats["lie_algebra"]["phi"])
    util.assert_float_eq(orc.so3_log(orc.so3_exp(phi)), phi)
    R = orc.so3_exp(phi)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-15)


def test_left_jacobian_inverse(kats):
    k = kats["lie_algebra"]
    phi = np.array(k["phi"])
    I = orc.left_jacobian(phi) @ orc.inverse_left_jacobian(phi)
    util.assert_float_eq(np.trace(I), k["jacobian_product_trace"])
    assert abs(np.float32(I.sum() - np.trace(I))) < 1e-6  # gtest FLOAT_EQ vs 0.0 -> tiny


def test_se3_log_exp_inverse(kats):
    xi = np.array(kats["lie_algebra"]["xi"])
    util.assert_float_eq(orc.se3_log(orc.se3_exp(xi)), xi)


def test_small_angle_branches():
    """lie_algebra.cpp:25-28, :43-46, :54-57, :70-73: first-order forms below 1e-6 rad."""
    phi = np.array([3e-7, -2e-7, 1e-7])
    assert np.array_equal(orc.so3_exp(phi), np.eye(3) + orc.hat(phi))
    assert np.array_equal(orc.left_jacobian(phi), np.eye(3) + 0.5 * orc.hat(phi))
    assert np.array_equal(orc.inverse_left_jacobian(phi), np.eye(3) - 0.5 * orc.hat(phi))
    assert np.allclose(orc.so3_log(np.eye(3) + orc.hat(phi)), phi, atol=1e-18)


# ---- test/test_trajectory_interpolation.cpp (artificial poses) -------------------------------------
def test_interpolation_midpoint(kats):
    k = kats["trajectory_interpolation_artificial"]
    p0, p1, p2 = (_rot_x_pose(a, t) for a, t in zip(k["x_rotation"], k["x_translation"]))
    ti = orc.interpolator_from_poses(k["times"][0], p0, k["times"][2], p2)
    rc, mid = orc.get_pose_at_time(ti, k["times"][1])
    assert rc == orc.OK
    assert util.transformation_matrices_are_the_same(mid.matrix(), p1.matrix())
    assert np.allclose(mid.matrix(), p1.matrix(), atol=1e-14)


def test_relative_pose_between_times(kats):
    k = kats["trajectory_interpolation_artificial"]
    p0, _, p2 = (_rot_x_pose(a, t) for a, t in zip(k["x_rotation"], k["x_translation"]))
    ti = orc.interpolator_from_poses(k["times"][0], p0, k["times"][2], p2)
    rc1, tf01 = orc.relative_pose_between_times(ti, k["times"][0], k["times"][1])
    rc2, tf12 = orc.relative_pose_between_times(ti, k["times"][1], k["times"][2])
    assert rc1 == orc.OK and rc2 == orc.OK
    assert util.transformation_matrices_are_the_same(tf01.matrix(), tf12.matrix())


def test_out_of_range_time_is_an_error(kats, golden_dir):
    """test_trajectory_interpolation.cpp:77-81: GetPoseAtTime(0) on a real-OXTS interpolator dies."""
    run = os.path.join(golden_dir, "kitti_2011_09_26_drive_0005")
    o0 = orc.oxts(**{k: v for k, v in util.load_oxts_fields(run, 0).items()})
    o2 = orc.oxts(**{**util.load_oxts_fields(run, 0), "stamp": o0.stamp + 0.2})
    ti = orc.interpolator_from_oxts(o0, o2)
    rc, _ = orc.get_pose_at_time(ti, kats["trajectory_out_of_range"]["query_time"])
    assert rc == orc.ERR_TIME_OUT_OF_RANGE
    rc, _ = orc.get_pose_at_time(ti, o0.stamp)  # boundaries are inclusive (:47)
    assert rc == orc.OK
    rc, _ = orc.get_pose_at_time(ti, o2.stamp)
    assert rc == orc.OK


# ---- test/test_oxts_to_pose.cpp ------------------------------------------------------------------------
def test_oxts_to_pose_kat(kats, golden_dir):
    k = kats["oxts_to_pose"]
    run = os.path.join(golden_dir, "kitti_2011_09_26_drive_0005")
    f = util.load_oxts_fields(run, 0)
    pose = orc.oxts_to_pose(orc.oxts(**f), k["scale"])
    util.assert_float_eq(np.linalg.det(orc.affine_rotation(pose)), k["expected_det"])
    util.assert_float_eq(pose.tv(), k["expected_translation"])
    # quaternion path == Rz*Ry*Rx to rounding
    cy, sy = np.cos(f["yaw"]), np.sin(f["yaw"])
    cp, sp = np.cos(f["pitch"]), np.sin(f["pitch"])
    cr, sr = np.cos(f["roll"]), np.sin(f["roll"])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    assert np.allclose(pose.Rm(), Rz @ Ry @ Rx, atol=1e-15)


# ---- test/test_data_io.cpp (values of the shipped frame 0) -----------------------------------------------
def test_data_io_values_of_shipped_frame(kats, golden_dir):
    k = kats["data_io"]
    run = os.path.join(golden_dir, "kitti_2011_09_26_drive_0005")
    f = util.load_oxts_fields(run, 0)
    assert f["stamp"] == k["oxts_stamp"]
    for key, val in k["oxts"].items():
        assert f[key] == val, key
    vp = os.path.join(run, "velodyne_points")
    t0 = util.load_timestamp(os.path.join(vp, "timestamps_start.txt"), 0)
    tm = util.load_timestamp(os.path.join(vp, "timestamps.txt"), 0)
    t1 = util.load_timestamp(os.path.join(vp, "timestamps_end.txt"), 0)
    assert (t0, tm, t1) == (k["stamp_start"], k["stamp_middle"], k["stamp_end"])
    xyzi = util.load_velodyne_bin(run, 0)
    assert xyzi.shape[0] == k["num_points"]
    cloud = np.concatenate([xyzi[:, :3].astype(np.float64), np.ones((xyzi.shape[0], 1))], axis=1)
    util.assert_float_eq(cloud[0], k["first_point"])
    util.assert_float_eq(cloud[-1], k["last_point"])
    util.assert_float_eq(xyzi[0, 3], k["first_intensity"])
    util.assert_float_eq(xyzi[-1, 3], k["last_intensity"])
    stamps = orc.pseudo_timestamps(cloud, t0, t1)
    util.assert_float_eq(stamps[0], k["first_stamp"])
    util.assert_float_eq(stamps[-1], k["last_stamp"])
    assert stamps.min() >= t0 and stamps.max() <= t1


# ---- internal consistency of the restatement ---------------------------------------------------------------
def test_polar_rotation_matches_numpy_svd():
    rng = np.random.default_rng(7)
    for _ in range(50):
        A = rng.normal(size=(3, 3))
        if np.linalg.det(A) < 0:
            A[:, 0] = -A[:, 0]
        U, s, Vt = np.linalg.svd(A)
        Q = U @ Vt
        got = orc.affine_rotation(orc.Affine.from_Rt(A, np.zeros(3)))
        assert np.allclose(got, Q, atol=1e-12)
    # reflection case: rotation() flips the smallest singular direction so det = +1
    A = np.diag([1.0, 2.0, -0.5])
    got = orc.affine_rotation(orc.Affine.from_Rt(A, np.zeros(3)))
    assert np.isclose(np.linalg.det(got), 1.0)


def test_affine_inverse_and_mul():
    rng = np.random.default_rng(3)
    A = orc.Affine.from_Rt(orc.so3_exp(rng.normal(size=3)), rng.normal(size=3) * 1e3)
    B = orc.Affine.from_Rt(orc.so3_exp(rng.normal(size=3)), rng.normal(size=3))
    assert np.allclose(orc.affine_mul(A, B).matrix(), A.matrix() @ B.matrix(), atol=1e-12)
    assert np.allclose(orc.affine_inverse(A).matrix(), np.linalg.inv(A.matrix()), atol=1e-9)


@pytest.mark.parametrize("turn", [0.0, 0.02, 0.1, 0.6])
def test_faithful_equals_hoisted_closed_form(turn, golden_dir):
    """SURVEY.md section 3.2 identity: (P1 Exp(x_r f))^-1 (P1 Exp(x_i f)) == Exp((x_i - x_r) f).
    Both oracle modes must agree far inside the 1e-5 parity bar on the real KITTI frame with Mercator-scale poses."""
    run = os.path.join(golden_dir, "kitti_2011_09_26_drive_0005")
    xyzi = util.load_velodyne_bin(run, 0)[::31]
    f = util.load_oxts_fields(run, 0)
    P1 = orc.oxts_to_pose(orc.oxts(**f))
    step = orc.se3_exp([1.3, 0.05, -0.02, 0.01 * turn, -0.02 * turn, turn])
    P2 = orc.affine_mul(P1, step)
    t0, t1 = 47072.283701593, 47072.386973931
    treq = 47072.335337762
    a = orc.deskew_xyzi_f32(xyzi, t0, P1, t1, P2, treq, mode=orc.FAITHFUL, threads=2)
    b = orc.deskew_xyzi_f32(xyzi, t0, P1, t1, P2, treq, mode=orc.HOISTED, threads=2)
    assert a["rc"] == orc.OK and b["rc"] == orc.OK
    err = util.rel_point_error(b["xyz_f64"], a["xyz_f64"])
    assert err.max() < 1e-8, err.max()


def test_real_odometry_kat_when_kitti_root_has_the_packets():
    """test_trajectory_interpolation.cpp:83-100: interpolating OXTS packets 0 and 2 of drive 0005 at packet 1's stamp differs from
    packet 1's own pose by a matrix whose elements sum to -0.01504419 (ASSERT_FLOAT_EQ).  The reference ships only packet 0, so the
    number can be checked only where KITTI_ROOT holds the raw drive ("not reproducible" offline) -- the one
    reference-held number for the path the oracle is not otherwise held to.  Both the oracle and the product's host pre-step."""
    import os

    from kitti_motion_compensation_amd import capi
    from tests import workloads

    run = workloads.find_drive("0005")
    if run is None or not all(os.path.exists(os.path.join(run, "oxts", "data", f"{i:010d}.txt")) for i in (0, 1, 2)):
        pytest.skip("needs OXTS packets 0-2 of 2011_09_26_drive_0005 under KITTI_ROOT (the reference ships packet 0 only)")
    f = [util.load_oxts_fields(run, i) for i in (0, 1, 2)]
    o = [orc.oxts(**x) for x in f]
    rc, interp = orc.get_pose_at_time(orc.interpolator_from_oxts(o[0], o[2]), o[1].stamp)
    assert rc == orc.OK
    truth = orc.oxts_to_pose(o[1])
    util.assert_float_eq((interp.matrix() - truth.matrix()).sum(), -0.01504419, "oracle, real odometry")
    c = [capi.Oxts(**x) for x in f]
    got = capi.interpolate_trajectory(c[0], c[2], f[1]["stamp"])
    ref = capi.oxts_to_pose(c[1])
    util.assert_float_eq((got - ref).sum(), -0.01504419, "host pre-step, real odometry")
