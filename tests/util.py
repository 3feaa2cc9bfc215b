"""Test helpers (not product code): reference comparison rules and tiny KITTI text parsers."""
import os

import numpy as np


def float_ulp_diff(a, b):
    """ULP distance between a and b after casting both to IEEE float (gtest's ASSERT_FLOAT_EQ metric)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)

    def biased(x):
        i = x.view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)

    return np.abs(biased(a) - biased(b))


def assert_float_eq(a, b, msg=""):
    """gtest ASSERT_FLOAT_EQ: equal as float within 4 ULP."""
    d = float_ulp_diff(a, b)
    assert np.all(d <= 4), f"ASSERT_FLOAT_EQ failed {msg}: {a} vs {b} ({d} ulp)"


def float_equal_eps(a, b, eps=1e-10):
    """utilities_for_testing.hpp:4  FloatEqual(float a, float b, float eps)"""
    return abs(np.float32(a) - np.float32(b)) <= np.float32(eps)


def transformation_matrices_are_the_same(M1, M2):
    """utilities_for_testing.hpp:6-11 criterion on 4x4 matrices."""
    I = M1 @ np.linalg.inv(M2)
    return float_equal_eps(np.trace(I), 4.0) and float_equal_eps(I.sum() - np.trace(I), 0.0)


def hhmmss_to_seconds(tok):
    """utils.cpp:31-38 restated: 'HH:MM:SS.nnnnnnnnn' -> seconds since midnight (double)."""
    hours = int(tok[0:2])
    minutes = int(tok[3:5])
    seconds = float(tok[6:24])
    return float(60 * hours * 60 + minutes * 60) + seconds


def load_timestamp(path, frame_id):
    """data_io.cpp:18-35 restated."""
    with open(path) as f:
        lines = f.read().splitlines()
    return hhmmss_to_seconds(lines[frame_id].split(" ")[1])


def load_oxts_fields(run_dir, frame_id):
    """data_io.cpp:37-66 restated -> dict(stamp, lat, lon, alt, roll, pitch, yaw, vf, vl, vu)."""
    stamp = load_timestamp(os.path.join(run_dir, "oxts", "timestamps.txt"), frame_id)
    with open(os.path.join(run_dir, "oxts", "data", f"{frame_id:010d}.txt")) as f:
        tok = f.readline().split(" ")
    v = [float(t) for t in tok[:11]]
    return dict(stamp=stamp, lat=v[0], lon=v[1], alt=v[2], roll=v[3], pitch=v[4], yaw=v[5], vf=v[8], vl=v[9], vu=v[10])


def load_velodyne_bin(run_dir, frame_id):
    """KITTI velodyne .bin: f32 AoS x,y,z,intensity (data_io.cpp:101-138)."""
    return np.fromfile(os.path.join(run_dir, "velodyne_points", "data", f"{frame_id:010d}.bin"), dtype=np.float32).reshape(-1, 4)


def rel_point_error(p, ref):
    """SURVEY.md section 8(d) parity gate: ||p - ref||_2 / max(||ref||_2, 1e-3) per point."""
    p = np.asarray(p, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return np.linalg.norm(p - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-3)


def real_cadence_drive(run_dir, n_frames=108, speed=13.0, yaw_rate=0.3):
    """SURVEY.md section 8(d) config 3 twin: the shipped timestamps*.txt (real KITTI cadence) and OXTS packets placed on a
    constant-twist track (arc of radius speed / yaw_rate, small roll / pitch / altitude rates), through the inverse of the
    Mercator map that OxtsToPose applies.  Returns (t_start, t_mid, t_end, oxts_dicts)."""
    vp = os.path.join(run_dir, "velodyne_points")
    t_start = [load_timestamp(os.path.join(vp, "timestamps_start.txt"), i) for i in range(n_frames)]
    t_mid = [load_timestamp(os.path.join(vp, "timestamps.txt"), i) for i in range(n_frames)]
    t_end = [load_timestamp(os.path.join(vp, "timestamps_end.txt"), i) for i in range(n_frames)]
    t_oxts = [load_timestamp(os.path.join(run_dir, "oxts", "timestamps.txt"), i) for i in range(n_frames)]
    f0 = load_oxts_fields(run_dir, 0)
    R_E = 6378137.0
    north0 = R_E * np.log(np.tan(np.pi * (90.0 + f0["lat"]) / 360.0))
    east0 = R_E * np.pi * f0["lon"] / 180.0
    oxts = []
    for i in range(n_frames):
        dt = t_oxts[i] - t_oxts[0]
        yaw = f0["yaw"] + yaw_rate * dt
        east = east0 + (speed / yaw_rate) * (np.sin(yaw) - np.sin(f0["yaw"]))
        north = north0 - (speed / yaw_rate) * (np.cos(yaw) - np.cos(f0["yaw"]))
        oxts.append(dict(stamp=t_oxts[i], lat=360.0 / np.pi * np.arctan(np.exp(north / R_E)) - 90.0, lon=east * 180.0 / (np.pi * R_E),
                         alt=f0["alt"] + 0.02 * dt, roll=f0["roll"] + 0.01 * dt, pitch=f0["pitch"] - 0.008 * dt, yaw=yaw))
    return t_start, t_mid, t_end, oxts


def load_kitti_calibration(date_dir):
    """calib_velo_to_cam.txt + calib_cam_to_cam.txt restated (data_io.cpp:168-210, :321-406) ->
    (tf_c00_lo 3x4, R_rect_00 3x3, [P_rect_00..03] 3x4 each)."""
    def values(line):
        return [float(t) for t in line.strip().split(" ")[1:]]

    with open(os.path.join(date_dir, "calib_velo_to_cam.txt")) as f:
        lines = f.read().splitlines()
    R = np.array(values(lines[1])).reshape(3, 3)
    T = np.array(values(lines[2])).reshape(3, 1)
    tf = np.hstack([R, T])
    with open(os.path.join(date_dir, "calib_cam_to_cam.txt")) as f:
        lines = f.read().splitlines()[2:]
    R_rect_00 = np.array(values(lines[6])).reshape(3, 3)
    P = [np.array(values(lines[8 * c + 7])).reshape(3, 4) for c in range(4)]
    return tf, R_rect_00, P


def project_numpy(xyz, tf, R_rect, P_rects, max_range=15.0):
    """camera_model.cpp:5-95 (without the drawing) in numpy f64, one rounded operation at a time in the reference's order;
    an independent twin of oracle kmo_project_points.  -> (uv (N,4,2) int32, bgrv (N,4) uint8)"""
    p = np.asarray(xyz, dtype=np.float64)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    with np.errstate(all="ignore"):
        c = [((tf[k, 0] * x + tf[k, 1] * y) + tf[k, 2] * z) + tf[k, 3] * 1.0 for k in range(3)]
        r = [((R_rect[k, 0] * c[0] + R_rect[k, 1] * c[1]) + R_rect[k, 2] * c[2]) + 0.0 for k in range(3)]
        drawn = ~((r[2] < 0.01) | (r[2] > max_range) | (r[1] > 1.25))
        n = p.shape[0]
        uv = np.full((n, 4, 2), np.iinfo(np.int32).min, dtype=np.int32)
        for cam in range(4):
            P = P_rects[cam]
            h = [((P[k, 0] * r[0] + P[k, 1] * r[1]) + P[k, 2] * r[2]) + P[k, 3] * 1.0 for k in range(3)]
            for j in range(2):
                q = h[j] / h[2]
                fits = (q > -2147483649.0) & (q < 2147483648.0)
                t = np.where(fits, np.trunc(np.where(fits, q, 0.0)), -2147483648.0).astype(np.int64).astype(np.int32)
                uv[:, cam, j] = np.where(drawn, t, np.iinfo(np.int32).min)
        cs = 255.0 * (r[2] / (max_range - 0.01))

        def sat(v):
            rr = np.rint(v)
            rr = np.where(rr != rr, 0.0, np.clip(rr, 0.0, 255.0))
            return rr.astype(np.uint8)

        bgrv = np.stack([sat(255.0 - cs), sat(cs), sat(255.0 - cs), np.ones(n, dtype=np.uint8)], axis=1)
        bgrv[~drawn] = 0
    return uv, bgrv
