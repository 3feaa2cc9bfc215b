"""ctypes binding of the CPU oracle (oracle/libkmc_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under kitti_motion_compensation_amd/ may import this module
(tests/test_abi_exports.py::test_product_never_references_the_oracle enforces it).  See oracle/kmc_oracle.h for the reference file:line each function restates.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkmc_oracle.so")


class Affine(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3)]

    @staticmethod
    def from_Rt(R, t) -> "Affine":
        a = Affine()
        R = np.asarray(R, dtype=np.float64).reshape(9)
        t = np.asarray(t, dtype=np.float64).reshape(3)
        for i in range(9):
            a.R[i] = R[i]
        for i in range(3):
            a.t[i] = t[i]
        return a

    @staticmethod
    def identity() -> "Affine":
        return Affine.from_Rt(np.eye(3), np.zeros(3))

    def Rm(self) -> np.ndarray:
        return np.array(list(self.R), dtype=np.float64).reshape(3, 3)

    def tv(self) -> np.ndarray:
        return np.array(list(self.t), dtype=np.float64)

    def matrix(self) -> np.ndarray:
        M = np.eye(4)
        M[:3, :3] = self.Rm()
        M[:3, 3] = self.tv()
        return M

    def rt12(self) -> np.ndarray:
        """row-major 3x4 [R|t] -- the pose layout of the C-ABI (include/kmc_hip.h)."""
        return np.ascontiguousarray(self.matrix()[:3, :4]).reshape(12)


class Oxts(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("stamp", "lat", "lon", "alt", "roll", "pitch", "yaw", "vf", "vl", "vu")]


class Interpolator(C.Structure):
    _fields_ = [("time_1", C.c_double), ("pose_1", Affine), ("time_2", C.c_double), ("pose_2", Affine)]


OK = 0
ERR_TIME_OUT_OF_RANGE = -1

_lib = None


def build(force: bool = False) -> str:
    """(Re)build the oracle shared library with the committed Makefile."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(os.path.join(_HERE, "kmc_oracle.c")), os.path.getmtime(os.path.join(_HERE, "kmc_oracle.h"))
    ):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    dp = C.POINTER(C.c_double)
    fp = C.POINTER(C.c_float)
    ap = C.POINTER(Affine)
    op = C.POINTER(Oxts)
    ip = C.POINTER(Interpolator)
    sig = {
        "kmo_hat": (None, [dp, dp]),
        "kmo_vee": (None, [dp, dp]),
        "kmo_so3_exp": (None, [dp, dp]),
        "kmo_so3_log": (None, [dp, dp]),
        "kmo_left_jacobian": (None, [dp, dp]),
        "kmo_inverse_left_jacobian": (None, [dp, dp]),
        "kmo_se3_exp": (None, [dp, ap]),
        "kmo_se3_log": (None, [ap, dp]),
        "kmo_affine_inverse": (None, [ap, ap]),
        "kmo_affine_mul": (None, [ap, ap, ap]),
        "kmo_affine_rotation": (None, [ap, dp]),
        "kmo_affine_apply4": (None, [ap, dp, dp]),
        "kmo_mat3_det": (C.c_double, [dp]),
        "kmo_interpolator_from_poses": (None, [C.c_double, ap, C.c_double, ap, ip]),
        "kmo_interpolator_from_oxts": (None, [op, op, ip]),
        "kmo_get_pose_at_time": (C.c_int, [ip, C.c_double, ap]),
        "kmo_relative_pose_between_times": (C.c_int, [ip, C.c_double, C.c_double, ap]),
        "kmo_interpolate_trajectory": (C.c_int, [op, op, C.c_double, ap]),
        "kmo_fraction_of_scan_completed": (C.c_double, [dp]),
        "kmo_pseudo_timestamp": (C.c_double, [dp, C.c_double, C.c_double]),
        "kmo_pseudo_timestamps": (None, [dp, C.c_size_t, C.c_double, C.c_double, dp]),
        "kmo_motion_compensate_point": (C.c_int, [ip, C.c_double, dp, C.c_double, dp]),
        "kmo_motion_compensate_frame": (
            C.c_int,
            [dp, dp, C.c_size_t, C.c_double, ap, C.c_double, ap, C.c_double, dp, C.POINTER(C.c_size_t)],
        ),
        "kmo_oxts_to_pose": (None, [op, C.c_double, ap]),
        "kmo_make_frame_poses": (C.c_int, [op, op, op, C.c_double, C.c_double, ap, ap]),
        "kmo_deskew_xyzi_f32": (
            C.c_int,
            [fp, C.c_size_t, C.c_double, ap, C.c_double, ap, C.c_double, C.c_int, C.c_int, dp, fp, dp,
             C.POINTER(C.c_size_t)],
        ),
        "kmo_num_threads": (C.c_int, []),
        "kmo_traj_pose_at_time": (C.c_int, [dp, ap, C.c_size_t, C.c_double, ap]),
        "kmo_deskew_xyzi_f32_traj": (
            C.c_int,
            [fp, C.c_size_t, C.c_double, C.c_double, dp, ap, C.c_size_t, C.c_double, C.c_int, dp, C.POINTER(C.c_uint32),
             C.POINTER(C.c_size_t)],
        ),
        "kmo_motion_compensate_frame_traj": (
            C.c_int,
            [dp, dp, C.c_size_t, dp, ap, C.c_size_t, C.c_double, dp, C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)],
        ),
        "kmo_bracket_indices_f32": (None, [fp, C.c_size_t, dp, C.c_size_t, C.c_double, C.c_double, C.POINTER(C.c_uint32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L.kmo_set_num_threads.restype = None
    L.kmo_set_num_threads.argtypes = [C.c_int]
    L.kmo_set_num_threads(default_threads())
    _lib = L
    return L


def default_threads() -> int:
    """OpenMP team size for the oracle helpers: the cores this process may really use (affinity mask capped by the cgroup
    CPU quota -- GPU boxes expose hundreds of logical CPUs but grant far fewer; oversubscribed OpenMP teams crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


# ---- thin numpy wrappers -------------------------------------------------------------------------
def hat(a):
    a_, ap_ = _d(a)
    M = np.zeros(9)
    lib().kmo_hat(ap_, M.ctypes.data_as(C.POINTER(C.c_double)))
    return M.reshape(3, 3)


def vee(M):
    M_, mp = _d(M)
    a = np.zeros(3)
    lib().kmo_vee(mp, a.ctypes.data_as(C.POINTER(C.c_double)))
    return a


def _mat_fn(name, v):
    v_, vp = _d(v)
    M = np.zeros(9)
    getattr(lib(), name)(vp, M.ctypes.data_as(C.POINTER(C.c_double)))
    return M.reshape(3, 3)


def so3_exp(phi):
    return _mat_fn("kmo_so3_exp", phi)


def left_jacobian(phi):
    return _mat_fn("kmo_left_jacobian", phi)


def inverse_left_jacobian(phi):
    return _mat_fn("kmo_inverse_left_jacobian", phi)


def so3_log(R):
    R_, rp = _d(R)
    phi = np.zeros(3)
    lib().kmo_so3_log(rp, phi.ctypes.data_as(C.POINTER(C.c_double)))
    return phi


def se3_exp(xi) -> Affine:
    xi_, xp = _d(xi)
    T = Affine()
    lib().kmo_se3_exp(xp, C.byref(T))
    return T


def se3_log(T: Affine):
    xi = np.zeros(6)
    lib().kmo_se3_log(C.byref(T), xi.ctypes.data_as(C.POINTER(C.c_double)))
    return xi


def affine_inverse(T: Affine) -> Affine:
    out = Affine()
    lib().kmo_affine_inverse(C.byref(T), C.byref(out))
    return out


def affine_mul(A: Affine, B: Affine) -> Affine:
    out = Affine()
    lib().kmo_affine_mul(C.byref(A), C.byref(B), C.byref(out))
    return out


def affine_rotation(T: Affine):
    R = np.zeros(9)
    lib().kmo_affine_rotation(C.byref(T), R.ctypes.data_as(C.POINTER(C.c_double)))
    return R.reshape(3, 3)


def oxts(stamp, lat, lon, alt, roll, pitch, yaw, vf=0.0, vl=0.0, vu=0.0) -> Oxts:
    return Oxts(stamp, lat, lon, alt, roll, pitch, yaw, vf, vl, vu)


def oxts_to_pose(o: Oxts, scale: float = 1.0) -> Affine:
    out = Affine()
    lib().kmo_oxts_to_pose(C.byref(o), scale, C.byref(out))
    return out


def interpolator_from_poses(t1, p1: Affine, t2, p2: Affine) -> Interpolator:
    ti = Interpolator()
    lib().kmo_interpolator_from_poses(t1, C.byref(p1), t2, C.byref(p2), C.byref(ti))
    return ti


def interpolator_from_oxts(o0: Oxts, o1: Oxts) -> Interpolator:
    ti = Interpolator()
    lib().kmo_interpolator_from_oxts(C.byref(o0), C.byref(o1), C.byref(ti))
    return ti


def get_pose_at_time(ti: Interpolator, t: float):
    out = Affine()
    rc = lib().kmo_get_pose_at_time(C.byref(ti), t, C.byref(out))
    return rc, out


def relative_pose_between_times(ti: Interpolator, anchor: float, query: float):
    out = Affine()
    rc = lib().kmo_relative_pose_between_times(C.byref(ti), anchor, query, C.byref(out))
    return rc, out


def make_frame_poses(o_nm1: Oxts, o_n: Oxts, o_np1: Oxts, stamp_start: float, stamp_end: float):
    a, b = Affine(), Affine()
    rc = lib().kmo_make_frame_poses(C.byref(o_nm1), C.byref(o_n), C.byref(o_np1), stamp_start, stamp_end,
                                    C.byref(a), C.byref(b))
    return rc, a, b


def fraction_of_scan_completed(p):
    p_, pp = _d(p)
    return lib().kmo_fraction_of_scan_completed(pp)


def pseudo_timestamp(p, start, end):
    p_, pp = _d(p)
    return lib().kmo_pseudo_timestamp(pp, start, end)


def pseudo_timestamps(cloud_nx4, start, end):
    """cloud_nx4: (N,4) array; converted to Eigen's column-major layout internally."""
    cm = np.asfortranarray(np.asarray(cloud_nx4, dtype=np.float64))
    n = cm.shape[0]
    out = np.zeros(n)
    lib().kmo_pseudo_timestamps(cm.ctypes.data_as(C.POINTER(C.c_double)), n, start, end,
                                out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def motion_compensate_point(ti: Interpolator, stamp, p, requested):
    p_, pp = _d(p)
    out = np.zeros(4)
    rc = lib().kmo_motion_compensate_point(C.byref(ti), stamp, pp, requested,
                                           out.ctypes.data_as(C.POINTER(C.c_double)))
    return rc, out


def motion_compensate_frame(cloud_nx4, stamps, stamp_start, T_start: Affine, stamp_end, T_end: Affine, requested):
    """Faithful MotionCompensateFrame on Eigen-layout data. Returns (rc, n_bad, out (N,4))."""
    cm = np.asfortranarray(np.asarray(cloud_nx4, dtype=np.float64))
    n = cm.shape[0]
    st_, sp = _d(stamps)
    out = np.zeros((n, 4), order="F")
    nbad = C.c_size_t(0)
    rc = lib().kmo_motion_compensate_frame(cm.ctypes.data_as(C.POINTER(C.c_double)), sp, n, stamp_start,
                                           C.byref(T_start), stamp_end, C.byref(T_end), requested,
                                           out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nbad))
    return rc, nbad.value, np.ascontiguousarray(out)


FAITHFUL = 0
HOISTED = 1


def deskew_xyzi_f32(xyzi, stamp_start, T_start: Affine, stamp_end, T_end: Affine, requested, mode=HOISTED,
                    threads=0, want_f64=True, want_f32=False, want_stamps=False, out_f32=None):
    """KITTI-layout pipeline: (N,4) f32 AoS in -> dict(xyz_f64 (N,3), xyzi_f32 (N,4), stamps, rc, n_bad).
    out_f32: optional preallocated (N,4) float32 array to write into (implies want_f32)."""
    a = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    n = a.shape[0]
    if threads <= 0:
        threads = default_threads() if n >= 20000 else 1
    o64 = np.empty((n, 3), dtype=np.float64) if want_f64 else None
    if out_f32 is not None:
        assert out_f32.dtype == np.float32 and out_f32.shape == (n, 4) and out_f32.flags.c_contiguous
        want_f32 = True
        o32 = out_f32
    else:
        o32 = np.empty((n, 4), dtype=np.float32) if want_f32 else None
    st = np.empty(n, dtype=np.float64) if want_stamps else None
    nbad = C.c_size_t(0)
    rc = lib().kmo_deskew_xyzi_f32(
        a.ctypes.data_as(C.POINTER(C.c_float)), n, stamp_start, C.byref(T_start), stamp_end, C.byref(T_end),
        requested, mode, threads,
        o64.ctypes.data_as(C.POINTER(C.c_double)) if want_f64 else None,
        o32.ctypes.data_as(C.POINTER(C.c_float)) if want_f32 else None,
        st.ctypes.data_as(C.POINTER(C.c_double)) if want_stamps else None,
        C.byref(nbad))
    return {"rc": rc, "n_bad": nbad.value, "xyz_f64": o64, "xyzi_f32": o32, "stamps": st}


def num_threads() -> int:
    return lib().kmo_num_threads()


# ---- N-knot trajectory (this project's generalisation; see kmc_oracle.h) ------------------------------
def _affine_array(poses):
    arr = (Affine * len(poses))()
    for i, p in enumerate(poses):
        arr[i] = p
    return arr


def traj_pose_at_time(times, poses, t):
    t_, tp = _d(times)
    out = Affine()
    rc = lib().kmo_traj_pose_at_time(tp, _affine_array(poses), len(poses), t, C.byref(out))
    return rc, out


def deskew_xyzi_f32_traj(xyzi, stamp_start, stamp_end, times, poses, requested, threads=0):
    """-> dict(rc, n_bad, xyz_f64 (N,3), bracket_by_time (N,) uint32)"""
    a = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    n = a.shape[0]
    if threads <= 0:
        threads = default_threads() if n >= 20000 else 1
    t_, tp = _d(times)
    o64 = np.empty((n, 3), dtype=np.float64)
    br = np.empty(n, dtype=np.uint32)
    nbad = C.c_size_t(0)
    rc = lib().kmo_deskew_xyzi_f32_traj(a.ctypes.data_as(C.POINTER(C.c_float)), n, stamp_start, stamp_end, tp,
                                        _affine_array(poses), len(poses), requested, threads,
                                        o64.ctypes.data_as(C.POINTER(C.c_double)), br.ctypes.data_as(C.POINTER(C.c_uint32)),
                                        C.byref(nbad))
    return {"rc": rc, "n_bad": nbad.value, "xyz_f64": o64, "bracket_by_time": br}


def motion_compensate_frame_traj(cloud_nx4, stamps, times, poses, requested):
    cm = np.asfortranarray(np.asarray(cloud_nx4, dtype=np.float64))
    n = cm.shape[0]
    st_, sp = _d(stamps)
    t_, tp = _d(times)
    out = np.zeros((n, 4), order="F")
    br = np.empty(n, dtype=np.uint32)
    nbad = C.c_size_t(0)
    rc = lib().kmo_motion_compensate_frame_traj(cm.ctypes.data_as(C.POINTER(C.c_double)), sp, n, tp, _affine_array(poses),
                                                len(poses), requested, out.ctypes.data_as(C.POINTER(C.c_double)),
                                                br.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(nbad))
    return rc, nbad.value, np.ascontiguousarray(out), br


def bracket_indices_f32(xyzi, times, stamp_start, stamp_end):
    a = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    t_, tp = _d(times)
    out = np.empty(a.shape[0], dtype=np.uint32)
    lib().kmo_bracket_indices_f32(a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], tp, len(t_), stamp_start, stamp_end,
                                  out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


# ---- N4: projection (camera_model.cpp without the drawing; parity unpinned, see kmc_oracle.h) ------------
class CameraRig(C.Structure):
    _fields_ = [("tf_c00_lo", C.c_double * 12), ("R_rect_00", C.c_double * 9), ("P_rect", (C.c_double * 12) * 4),
                ("max_range", C.c_double)]


def camera_rig(tf_c00_lo_3x4, R_rect_00, P_rects, max_range=15.0) -> CameraRig:
    g = CameraRig()
    g.tf_c00_lo[:] = [float(v) for v in np.asarray(tf_c00_lo_3x4, dtype=np.float64).reshape(12)]
    g.R_rect_00[:] = [float(v) for v in np.asarray(R_rect_00, dtype=np.float64).reshape(9)]
    for c in range(4):
        g.P_rect[c][:] = [float(v) for v in np.asarray(P_rects[c], dtype=np.float64).reshape(12)]
    g.max_range = float(max_range)
    return g


def project_points(xyz, rig):
    """xyz (N,3) f64 -> (uv (N,4,2) int32: per point, per camera, (u, v); bgrv (N,4) uint8)"""
    a = np.asarray(xyz, dtype=np.float64)
    n = a.shape[0]
    cols = [np.ascontiguousarray(a[:, k]) for k in range(3)]
    uv = np.empty((n, 4, 2), dtype=np.int32)
    bgrv = np.empty((n, 4), dtype=np.uint8)
    f = lib().kmo_project_points
    f.restype = None
    f.argtypes = [C.c_void_p] * 3 + [C.c_size_t, C.POINTER(CameraRig), C.c_void_p, C.c_void_p]
    f(cols[0].ctypes.data, cols[1].ctypes.data, cols[2].ctypes.data, n, C.byref(rig), uv.ctypes.data, bgrv.ctypes.data)
    return uv, bgrv


def project_xyzi_f32(xyzi, rig):
    a = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    n = a.shape[0]
    uv = np.empty((n, 4, 2), dtype=np.int32)
    bgrv = np.empty((n, 4), dtype=np.uint8)
    f = lib().kmo_project_xyzi_f32
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(CameraRig), C.c_void_p, C.c_void_p]
    f(a.ctypes.data, n, C.byref(rig), uv.ctypes.data, bgrv.ctypes.data)
    return uv, bgrv
