"""ctypes binding of the C-ABI in include/kmc_hip.h (libkmc_hip.so).

Plumbing only: the product is the HIP library; this module lets pytest and bench.py drive it with numpy
arrays (KMC_MEM_HOST) or torch device tensors (KMC_MEM_DEVICE, zero-copy through data_ptr()).

There is NO fallback: if the shared library has not been built, importing the symbols raises; if there is
no HIP device, Context() raises KmcError(KMC_ERR_NO_DEVICE).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libkmc_hip.so")

OK = 0
ERR_INVALID_ARG = -1
ERR_HIP = -2
ERR_NO_DEVICE = -3
ERR_TIME_OUT_OF_RANGE = -4
ERR_ALLOC = -5
ERR_DEGENERATE = -6

MEM_HOST = 0
MEM_DEVICE = 1
MEM_HOST_MAPPED = 2

TIER_SERIES3, TIER_SERIES5, TIER_WIDE, TIER_TRIG = 0, 1, 2, 3


class FrameParams(C.Structure):
    _fields_ = [("twist", C.c_double * 6), ("x_req", C.c_double)]

    @staticmethod
    def make(twist, x_req) -> "FrameParams":
        p = FrameParams()
        for i, v in enumerate(np.asarray(twist, dtype=np.float64).reshape(6)):
            p.twist[i] = float(v)
        p.x_req = float(x_req)
        return p

    def twist_np(self) -> np.ndarray:
        return np.array(list(self.twist), dtype=np.float64)


class Oxts(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("stamp", "lat", "lon", "alt", "roll", "pitch", "yaw", "vf", "vl", "vu")]


class Stats(C.Structure):
    _fields_ = [
        ("n_points", C.c_uint64),
        ("n_out_of_range", C.c_uint64),
        ("n_launches", C.c_uint32),
        ("variant", C.c_uint32),
        ("kernel_ms", C.c_float),
        ("total_ms", C.c_float),
    ]


class DeviceInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 128),
        ("arch", C.c_char * 64),
        ("device_id", C.c_int),
        ("compute_units", C.c_int),
        ("wavefront_size", C.c_int),
        ("hbm_bytes", C.c_uint64),
        ("clock_khz", C.c_int),
        ("any_order_dispatch", C.c_int),
    ]


class CameraRig(C.Structure):
    """kmc_camera_rig: the calibration of camera_model.hpp:10-11 as plain row-major doubles."""
    _fields_ = [("tf_c00_lo", C.c_double * 12), ("R_rect_00", C.c_double * 9), ("P_rect", (C.c_double * 12) * 4),
                ("max_range", C.c_double)]

    @staticmethod
    def make(tf_c00_lo_3x4, R_rect_00, P_rects, max_range=15.0) -> "CameraRig":
        g = CameraRig()
        g.tf_c00_lo[:] = [float(v) for v in np.asarray(tf_c00_lo_3x4, dtype=np.float64).reshape(12)]
        g.R_rect_00[:] = [float(v) for v in np.asarray(R_rect_00, dtype=np.float64).reshape(9)]
        for c in range(4):
            g.P_rect[c][:] = [float(v) for v in np.asarray(P_rects[c], dtype=np.float64).reshape(12)]
        g.max_range = float(max_range)
        return g


class TrajFrame(C.Structure):
    """kmc_traj_frame: one frame's trajectory for the batched N-knot entry point (pointers into HOST arrays)."""
    _fields_ = [("knot_times", C.POINTER(C.c_double)), ("knot_poses", C.POINTER(C.c_double)), ("n_knots", C.c_uint32),
                ("reserved", C.c_uint32), ("stamp_start", C.c_double), ("stamp_end", C.c_double), ("requested_time", C.c_double)]


class KmcError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        msg = f"{where}: {status_string(status)}"
        if detail:
            msg += f" [{detail}]"
        super().__init__(msg)


class CallTrace(C.Structure):
    """kmc_call_trace (kmc_hip.h): stage times of the last in-place call, microseconds (host: steady_clock; device: its own 100 MHz clock)."""
    _fields_ = [("issue_begin_us", C.c_double), ("issue_end_us", C.c_double), ("wait_begin_us", C.c_double), ("wait_end_us", C.c_double),
                ("dev_first_wave_us", C.c_double), ("dev_last_store_us", C.c_double), ("waves", C.c_uint32), ("route", C.c_uint32)]


_lib = None

# every symbol include/kmc_hip.h declares: (restype, argtypes)
_vp = C.c_void_p
_dp = C.POINTER(C.c_double)
SIGNATURES = {
    "kmc_abi_version": (C.c_int, []),
    "kmc_status_string": (C.c_char_p, [C.c_int]),
    "kmc_hip_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "kmc_hip_destroy": (None, [_vp]),
    "kmc_hip_set_stream": (C.c_int, [_vp, _vp]),
    "kmc_hip_use_own_stream": (C.c_int, [_vp]),
    "kmc_hip_synchronize": (C.c_int, [_vp]),
    "kmc_hip_enable_timing": (C.c_int, [_vp, C.c_int]),
    "kmc_hip_last_error": (C.c_char_p, [_vp]),
    "kmc_hip_device_info": (C.c_int, [_vp, C.POINTER(DeviceInfo)]),
    "kmc_hip_enable_call_trace": (C.c_int, [_vp, C.c_int]),
    "kmc_hip_last_call_trace": (C.c_int, [_vp, C.POINTER(CallTrace)]),
    "kmc_hip_force_tier": (C.c_int, [_vp, C.c_int]),
    "kmc_hip_timer_begin": (C.c_int, [_vp]),
    "kmc_hip_timer_end": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "kmc_hip_host_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "kmc_hip_host_free": (C.c_int, [_vp, _vp]),
    "kmc_host_pool_alloc": (C.c_int, [C.c_size_t, C.POINTER(_vp)]),
    "kmc_host_pool_alloc_near": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(_vp)]),
    "kmc_host_pool_free": (C.c_int, [_vp]),
    "kmc_host_pool_owns": (C.c_int, [_vp, C.c_size_t]),
    "kmc_host_pool_trim": (C.c_int, []),
    "kmc_hip_bind_thread_near_device": (C.c_int, [C.c_int]),
    "kmc_frame_params_from_poses": (C.c_int, [_dp, _dp, C.c_double, C.c_double, C.c_double, C.POINTER(FrameParams)]),
    "kmc_oxts_to_pose": (C.c_int, [C.POINTER(Oxts), C.c_double, _dp]),
    "kmc_interpolate_trajectory": (C.c_int, [C.POINTER(Oxts), C.POINTER(Oxts), C.c_double, _dp]),
    "kmc_make_frame_poses": (C.c_int, [C.POINTER(Oxts), C.POINTER(Oxts), C.POINTER(Oxts), C.c_double, C.c_double, _dp, _dp]),
    "kmc_frame_ranges_balanced": (C.c_int, [C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "kmc_hip_deskew_f32": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.POINTER(FrameParams), C.c_int, C.POINTER(Stats)]),
    "kmc_hip_set_frame_queues": (C.c_int, [_vp, C.c_int]),
    "kmc_hip_frame_queue_join": (C.c_int, [_vp]),
    "kmc_hip_any_order_launches": (C.c_uint64, [_vp]),
    "kmc_hip_completion_word_fallbacks": (C.c_uint64, [_vp, C.POINTER(C.c_uint32)]),
    "kmc_hip_frame_queue_dropped": (C.c_uint64, [_vp]),
    "kmc_hip_direct_frames": (C.c_uint64, [_vp]),
    "kmc_hip_set_direct_dispatch": (C.c_int, [_vp, C.c_int]),
    "kmc_hip_direct_dispatch_active": (C.c_int, [_vp]),
    "kmc_hip_set_frame_queue_order": (C.c_int, [_vp, C.c_int]),
    "kmc_hip_deskew_frames_f32": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(FrameParams), C.c_uint32, C.POINTER(Stats)]),
    "kmc_hip_deskew_batch_f32": (
        C.c_int,
        [_vp, _vp, _vp, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(FrameParams), _vp, C.c_int, C.POINTER(Stats)],
    ),
    "kmc_hip_deskew_f64cols": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, C.c_double, C.c_double, C.POINTER(FrameParams), _vp, _vp, _vp, _vp,
         C.c_int, C.POINTER(Stats)],
    ),
    "kmc_hip_deskew_f64cols_begin": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, C.c_double, C.c_double, C.POINTER(FrameParams), _vp, _vp, _vp, _vp, C.c_int],
    ),
    "kmc_hip_deskew_f64cols_end": (C.c_int, [_vp, C.POINTER(Stats)]),
    "kmc_hip_deskew_traj_f32": (
        C.c_int,
        [_vp, _vp, _vp, C.c_uint64, _dp, _dp, C.c_uint32, C.c_double, C.c_double, C.c_double, _vp, C.c_int, C.POINTER(Stats)],
    ),
    "kmc_hip_deskew_traj_batch_f32": (
        C.c_int,
        [_vp, _vp, _vp, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(TrajFrame), _vp, _vp, C.c_int, C.POINTER(Stats)],
    ),
    "kmc_hip_deskew_traj_f64cols": (
        C.c_int,
        [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _dp, _dp, C.c_uint32, C.c_double, _vp, _vp, _vp, _vp, _vp, C.c_int,
         C.POINTER(Stats)],
    ),
    "kmc_hip_pseudo_timestamps_f64": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_double, C.c_double, _vp, C.c_int]),
    "kmc_hip_project_f32": (
        C.c_int,
        [_vp, _vp, C.c_uint64, C.POINTER(CameraRig), C.POINTER(FrameParams), _vp, _vp, _vp, C.c_int, C.POINTER(Stats)],
    ),
    "kmc_hip_project_f64cols": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64, C.POINTER(CameraRig), _vp, _vp, C.c_int, C.POINTER(Stats)]),
    "kmc_hip_synth_points": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64]),
    "kmc_synth_points_host": (C.c_int, [_vp, C.c_uint64, C.c_uint64]),
}


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP runtimes in one process
    do not both see the GPU, so whichever is loaded first must be the only one: if torch is installed but not imported
    yet, load ITS runtime now; libkmc_hip.so's DT_NEEDED libamdhip64.so.7 then binds to it by SONAME, and a later
    `import torch` finds the very same file.  (Pure C/C++ users link /opt/rocm's runtime as usual.)"""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib() -> C.CDLL:
    """Loads libkmc_hip.so (built by `make -C kitti_motion_compensation_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        _preload_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the deskew path is HIP-only; there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # raises AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def status_string(status: int) -> str:
    return lib().kmc_status_string(status).decode()


def _ptr(a, dtype=None):
    """numpy array / torch tensor / int / None -> void* value.  Arrays must be C-contiguous (and of `dtype` if given):
    the C-ABI takes bare pointers, a strided view would silently be read as if it were dense."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        if not a.flags.c_contiguous:
            raise ValueError("kmc: array must be C-contiguous")
        if dtype is not None and a.dtype != np.dtype(dtype):
            raise TypeError(f"kmc: expected {np.dtype(dtype)}, got {a.dtype}")
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        if not a.is_contiguous():
            raise ValueError("kmc: tensor must be contiguous")
        if dtype is not None and str(a.dtype).replace("torch.", "") != np.dtype(dtype).name:
            raise TypeError(f"kmc: expected {np.dtype(dtype).name}, got {a.dtype}")
        return a.data_ptr()
    raise TypeError(type(a))


def _mem_kind(a) -> int:
    if isinstance(a, np.ndarray):
        return MEM_HOST
    if hasattr(a, "is_cuda"):
        return MEM_DEVICE if a.is_cuda else MEM_HOST
    raise TypeError(type(a))


def _pose12(T) -> np.ndarray:
    T = np.asarray(T, dtype=np.float64)
    if T.shape == (4, 4):
        T = T[:3, :4]
    return np.ascontiguousarray(T).reshape(12)


# ---- host pre-step (no GPU needed) ------------------------------------------------------------------
def frame_params_from_poses(T_start, T_end, stamp_start, stamp_end, requested_time) -> FrameParams:
    a, b = _pose12(T_start), _pose12(T_end)
    p = FrameParams()
    rc = lib().kmc_frame_params_from_poses(a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), stamp_start, stamp_end,
                                           requested_time, C.byref(p))
    if rc != OK:
        raise KmcError(rc, "kmc_frame_params_from_poses")
    return p


def oxts_to_pose(o: Oxts, scale: float = 1.0) -> np.ndarray:
    T = np.zeros(12)
    rc = lib().kmc_oxts_to_pose(C.byref(o), scale, T.ctypes.data_as(_dp))
    if rc != OK:
        raise KmcError(rc, "kmc_oxts_to_pose")
    return T.reshape(3, 4)


def interpolate_trajectory(o1: Oxts, o2: Oxts, time: float) -> np.ndarray:
    T = np.zeros(12)
    rc = lib().kmc_interpolate_trajectory(C.byref(o1), C.byref(o2), time, T.ctypes.data_as(_dp))
    if rc != OK:
        raise KmcError(rc, "kmc_interpolate_trajectory")
    return T.reshape(3, 4)


def make_frame_poses(o_nm1: Oxts, o_n: Oxts, o_np1: Oxts, stamp_start: float, stamp_end: float):
    a, b = np.zeros(12), np.zeros(12)
    rc = lib().kmc_make_frame_poses(C.byref(o_nm1), C.byref(o_n), C.byref(o_np1), stamp_start, stamp_end,
                                    a.ctypes.data_as(_dp), b.ctypes.data_as(_dp))
    if rc != OK:
        raise KmcError(rc, "kmc_make_frame_poses")
    return a.reshape(3, 4), b.reshape(3, 4)


def frame_ranges_balanced(frame_points, n_parts: int) -> np.ndarray:
    """kmc_frame_ranges_balanced: bounds[0..n_parts] of the contiguous, point-balanced frame ranges (host code, no GPU)."""
    sizes = np.ascontiguousarray(frame_points, dtype=np.uint64)
    bounds = np.zeros(n_parts + 1, dtype=np.uint32)
    rc = lib().kmc_frame_ranges_balanced(sizes.ctypes.data_as(C.POINTER(C.c_uint64)), len(sizes), n_parts,
                                         bounds.ctypes.data_as(C.POINTER(C.c_uint32)))
    if rc != OK:
        raise KmcError(rc, "kmc_frame_ranges_balanced")
    return bounds


class PooledArray:
    """A float64 / float32 / uint32 numpy array living in a block of the C-ABI's page-locked host pool (kmc_host_pool_alloc):
    the f64 entry points run IN PLACE on such arrays (no staging copies).  Keep the object alive while `.a` is in use."""

    def __init__(self, shape, dtype=np.float64):
        dt = np.dtype(dtype)
        n = int(np.prod(shape))
        p = _vp()
        rc = lib().kmc_host_pool_alloc(max(n * dt.itemsize, 1), C.byref(p))
        if rc != OK:
            raise KmcError(rc, "kmc_host_pool_alloc")
        self._p = p
        buf = (C.c_char * (n * dt.itemsize)).from_address(p.value)
        self.a = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)

    def close(self):
        if getattr(self, "_p", None):
            self.a = None
            lib().kmc_host_pool_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bind_thread_near_device(device: int = 0) -> None:
    """The calling thread onto the CPUs of the device's NUMA node (kmc_hip_bind_thread_near_device): what numactl does for a deployment."""
    rc = lib().kmc_hip_bind_thread_near_device(device)
    if rc != OK:
        raise KmcError(rc, "kmc_hip_bind_thread_near_device")


def host_pool_owns(a: np.ndarray) -> bool:
    return bool(lib().kmc_host_pool_owns(a.ctypes.data, a.nbytes))


def synth_points_host(n: int, seed: int) -> np.ndarray:
    out = np.empty((n, 4), dtype=np.float32)
    rc = lib().kmc_synth_points_host(out.ctypes.data, n, seed)
    if rc != OK:
        raise KmcError(rc, "kmc_synth_points_host")
    return out


def params_array(params_list):
    """list of FrameParams -> contiguous ctypes array (build once, reuse across steps)."""
    arr = (FrameParams * max(len(params_list), 1))()
    for i, p in enumerate(params_list):
        arr[i] = p
    return arr


# ---- device context -----------------------------------------------------------------------------------
class Context:
    """One kmc_ctx: one GPU, one stream.  Raises KmcError(ERR_NO_DEVICE) when there is no HIP device."""

    def __init__(self, device_id: int = 0, stream=None):
        self._h = _vp()
        rc = lib().kmc_hip_create(C.byref(self._h), device_id)
        if rc != OK:
            self._h = None
            raise KmcError(rc, "kmc_hip_create")
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_h", None):
            lib().kmc_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, where):
        if rc != OK:
            detail = lib().kmc_hip_last_error(self._h).decode() if rc == ERR_HIP else ""
            raise KmcError(rc, where, detail)

    def set_stream(self, stream):
        """stream: an int hipStream_t handle, taken literally -- 0 is HIP's legacy default stream, which is what
        torch.cuda.current_stream().cuda_stream returns unless a side stream is active; None = the ctx's own stream."""
        if stream is None:
            self._check(lib().kmc_hip_use_own_stream(self._h), "kmc_hip_use_own_stream")
        else:
            self._check(lib().kmc_hip_set_stream(self._h, C.c_void_p(int(stream))), "kmc_hip_set_stream")

    def synchronize(self):
        self._check(lib().kmc_hip_synchronize(self._h), "kmc_hip_synchronize")

    def enable_timing(self, on=True):
        self._check(lib().kmc_hip_enable_timing(self._h, 1 if on else 0), "kmc_hip_enable_timing")

    def force_tier(self, tier=-1):
        self._check(lib().kmc_hip_force_tier(self._h, tier), "kmc_hip_force_tier")

    def device_info(self) -> dict:
        d = DeviceInfo()
        self._check(lib().kmc_hip_device_info(self._h, C.byref(d)), "kmc_hip_device_info")
        return dict(name=d.name.decode(), arch=d.arch.decode(), device_id=d.device_id, compute_units=d.compute_units,
                    wavefront_size=d.wavefront_size, hbm_bytes=d.hbm_bytes, clock_khz=d.clock_khz, any_order_dispatch=d.any_order_dispatch)

    def timer_begin(self):
        self._check(lib().kmc_hip_timer_begin(self._h), "kmc_hip_timer_begin")

    def timer_end(self) -> float:
        ms = C.c_float(0)
        self._check(lib().kmc_hip_timer_end(self._h, C.byref(ms)), "kmc_hip_timer_end")
        return ms.value

    # -- hot path ------------------------------------------------------------------------------------
    def deskew_f32(self, xyzi_in, xyzi_out, params: FrameParams, n=None) -> Stats:
        kind = _mem_kind(xyzi_in)
        assert kind == _mem_kind(xyzi_out)
        if n is None:
            n = int(xyzi_in.shape[0])
        st = Stats()
        rc = lib().kmc_hip_deskew_f32(self._h, _ptr(xyzi_in, np.float32), _ptr(xyzi_out, np.float32), n, C.byref(params), kind, C.byref(st))
        self._check(rc, "kmc_hip_deskew_f32")
        return st

    def set_frame_queues(self, queues: int):
        """queues > 1: device-resident deskew_f32 calls are GATHERED on the host and issued as list launches (deferred issue, in-order
        results; see kmc_hip.h); 1 = every call issues its own launch."""
        self._check(lib().kmc_hip_set_frame_queues(self._h, int(queues)), "kmc_hip_set_frame_queues")

    def set_frame_queue_order(self, after_producers: bool = True):
        """False: the caller's word that nothing is produced on its stream between two calls (kmc_hip_set_frame_queue_order)."""
        self._check(lib().kmc_hip_set_frame_queue_order(self._h, 1 if after_producers else 0), "kmc_hip_set_frame_queue_order")

    def frame_queue_join(self):
        self._check(lib().kmc_hip_frame_queue_join(self._h), "kmc_hip_frame_queue_join")

    def enable_call_trace(self, on: bool = True):
        self._check(lib().kmc_hip_enable_call_trace(self._h, 1 if on else 0), "kmc_hip_enable_call_trace")

    def last_call_trace(self) -> "CallTrace":
        t = CallTrace()
        self._check(lib().kmc_hip_last_call_trace(self._h, C.byref(t)), "kmc_hip_last_call_trace")
        return t

    def direct_frames(self) -> int:
        """Frames dispatched through the context's direct queue (kmc_hip.h, "THE DIRECT QUEUE"); 0: HIP launches."""
        return int(lib().kmc_hip_direct_frames(self._h))

    def set_direct_dispatch(self, on: bool = True):
        """Opt in to (or out of) the direct queue (kmc_hip.h, "THE DIRECT QUEUE"): with it on, device-resident single-frame calls on the context's
        own stream are in NO HIP stream -- torch.cuda.synchronize() does not wait for them, only self.synchronize() (or any other call on the
        context) does, and a frame's tensors must stay alive and untouched until then."""
        self._check(lib().kmc_hip_set_direct_dispatch(self._h, 1 if on else 0), "kmc_hip_set_direct_dispatch")

    def direct_dispatch_active(self) -> bool:
        """True if eligible frames of this context really go through the direct queue now (asked for, and the device / runtime can)."""
        return bool(lib().kmc_hip_direct_dispatch_active(self._h))

    def frame_queue_dropped(self) -> int:
        """Gathered frames lost to a failed join over the context's life (kmc_hip.h, "Gathered frames and errors")."""
        return int(lib().kmc_hip_frame_queue_dropped(self._h))

    def completion_word_fallbacks(self):
        """(count, [sequence number expected, word seen, ticket seen] of the last event): in-place calls that ended on a stream
        synchronisation because the kernel's completion word had not come (kmc_hip.h); (0, [0, 0, 0]) is the normal state."""
        st = (C.c_uint32 * 3)()
        n = int(lib().kmc_hip_completion_word_fallbacks(self._h, st))
        return n, [int(v) for v in st]

    def any_order_launches(self) -> int:
        """Frames dispatched without the barrier bit so far (see kmc_hip_set_frame_queues in kmc_hip.h)."""
        return int(lib().kmc_hip_any_order_launches(self._h))

    def prepare_frames(self, pairs, params_list):
        """pairs: [(device_in, device_out), ...] of (n_f, 4) float32 device tensors.  -> opaque argument pack for deskew_frames_f32
        (built once, reusable: the per-call cost is then one ctypes call)."""
        nf = len(pairs)
        ins = (_vp * nf)(*[p[0].data_ptr() for p in pairs])
        outs = (_vp * nf)(*[p[1].data_ptr() for p in pairs])
        ns = (C.c_uint64 * nf)(*[int(p[0].shape[0]) for p in pairs])
        prm = params_list if isinstance(params_list, C.Array) else params_array(params_list)
        return (ins, outs, ns, prm, nf, pairs)

    def deskew_frames_f32(self, pack) -> Stats:
        """A stream of separate device-resident frames in one call (kmc_hip_deskew_frames_f32); `pack` from prepare_frames."""
        ins, outs, ns, prm, nf, _ = pack
        st = Stats()
        self._check(lib().kmc_hip_deskew_frames_f32(self._h, ins, outs, ns, prm, nf, C.byref(st)), "kmc_hip_deskew_frames_f32")
        return st

    def deskew_batch_f32(self, xyzi_in, xyzi_out, offsets, params_list, frame_idx_out=None) -> Stats:
        kind = _mem_kind(xyzi_in)
        offs = offsets if (isinstance(offsets, np.ndarray) and offsets.dtype == np.uint64 and offsets.flags.c_contiguous) \
            else np.ascontiguousarray(offsets, dtype=np.uint64)
        nf = len(offs) - 1
        arr = params_list if isinstance(params_list, C.Array) else params_array(params_list)
        st = Stats()
        rc = lib().kmc_hip_deskew_batch_f32(self._h, _ptr(xyzi_in, np.float32), _ptr(xyzi_out, np.float32),
                                            offs.ctypes.data_as(C.POINTER(C.c_uint64)), nf, arr, _ptr(frame_idx_out), kind,
                                            C.byref(st))
        self._check(rc, "kmc_hip_deskew_batch_f32")
        return st

    def deskew_f64cols(self, x, y, z, w, stamps, stamp_start, stamp_end, params: FrameParams, ox, oy, oz, ow=None,
                       raise_on_range=True):
        kind = _mem_kind(x)
        n = int(x.shape[0])
        st = Stats()
        rc = lib().kmc_hip_deskew_f64cols(self._h, _ptr(x, np.float64), _ptr(y, np.float64), _ptr(z, np.float64),
                                          _ptr(w, np.float64), _ptr(stamps, np.float64), n, stamp_start,
                                          stamp_end, C.byref(params), _ptr(ox, np.float64), _ptr(oy, np.float64),
                                          _ptr(oz, np.float64), _ptr(ow, np.float64), kind,
                                          C.byref(st))
        if rc == ERR_TIME_OUT_OF_RANGE and not raise_on_range:
            return rc, st
        self._check(rc, "kmc_hip_deskew_f64cols")
        return rc, st

    def deskew_f64cols_begin(self, x, y, z, w, stamps, stamp_start, stamp_end, params: FrameParams, ox, oy, oz, ow=None):
        """kmc_hip_deskew_f64cols_begin: issues the call (returns at once for device-addressable buffers); pair with deskew_f64cols_end."""
        rc = lib().kmc_hip_deskew_f64cols_begin(self._h, _ptr(x, np.float64), _ptr(y, np.float64), _ptr(z, np.float64), _ptr(w, np.float64),
                                                _ptr(stamps, np.float64), int(x.shape[0]), stamp_start, stamp_end, C.byref(params),
                                                _ptr(ox, np.float64), _ptr(oy, np.float64), _ptr(oz, np.float64), _ptr(ow, np.float64), _mem_kind(x))
        self._check(rc, "kmc_hip_deskew_f64cols_begin")

    def deskew_f64cols_end(self, raise_on_range=True):
        st = Stats()
        rc = lib().kmc_hip_deskew_f64cols_end(self._h, C.byref(st))
        if rc == ERR_TIME_OUT_OF_RANGE and not raise_on_range:
            return rc, st
        self._check(rc, "kmc_hip_deskew_f64cols_end")
        return rc, st

    def deskew_traj_f32(self, xyzi_in, xyzi_out, knot_times, knot_poses, stamp_start, stamp_end, requested_time,
                        bracket_idx_out=None, n=None) -> Stats:
        """knot_poses: (K, 3, 4) or (K, 12) row-major [R|t]."""
        kind = _mem_kind(xyzi_in)
        if n is None:
            n = int(xyzi_in.shape[0])
        t = np.ascontiguousarray(knot_times, dtype=np.float64)
        P = np.ascontiguousarray(np.asarray(knot_poses, dtype=np.float64).reshape(len(t), 12))
        st = Stats()
        rc = lib().kmc_hip_deskew_traj_f32(self._h, _ptr(xyzi_in, np.float32), _ptr(xyzi_out, np.float32), n, t.ctypes.data_as(_dp), P.ctypes.data_as(_dp),
                                           len(t), stamp_start, stamp_end, requested_time, _ptr(bracket_idx_out), kind, C.byref(st))
        self._check(rc, "kmc_hip_deskew_traj_f32")
        return st

    @staticmethod
    def prepare_traj_frames(frames):
        """list of dicts(times, poses (K,3,4)|(K,12), stamp_start, stamp_end, requested_time) -> (ctypes kmc_traj_frame array, keepalive).
        Build once and pass the result as `frames` to deskew_traj_batch_f32 when the same trajectories are used repeatedly:
        converting the dicts costs several microseconds per frame in Python, the C-ABI's own pre-step 0.4 us."""
        keep = []
        arr = (TrajFrame * max(len(frames), 1))()
        for i, fr in enumerate(frames):
            t = np.ascontiguousarray(fr["times"], dtype=np.float64)
            P = np.ascontiguousarray(np.asarray(fr["poses"], dtype=np.float64).reshape(len(t), 12))
            keep.append((t, P))
            arr[i].knot_times = t.ctypes.data_as(_dp)
            arr[i].knot_poses = P.ctypes.data_as(_dp)
            arr[i].n_knots = len(t)
            arr[i].stamp_start = float(fr["stamp_start"])
            arr[i].stamp_end = float(fr["stamp_end"])
            arr[i].requested_time = float(fr["requested_time"])
        return (arr, keep, len(frames))

    def deskew_traj_batch_f32(self, xyzi_in, xyzi_out, offsets, frames, frame_idx_out=None, bracket_idx_out=None) -> Stats:
        """frames: list of dicts(times, poses (K,3,4)|(K,12), stamp_start, stamp_end, requested_time), one per frame, or the
        result of prepare_traj_frames()."""
        kind = _mem_kind(xyzi_in)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64) if not (isinstance(offsets, np.ndarray) and offsets.dtype == np.uint64 and offsets.flags.c_contiguous) else offsets
        arr, keep, _ = frames if isinstance(frames, tuple) else self.prepare_traj_frames(frames)
        st = Stats()
        rc = lib().kmc_hip_deskew_traj_batch_f32(self._h, _ptr(xyzi_in, np.float32), _ptr(xyzi_out, np.float32),
                                                 offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(offs) - 1, arr,
                                                 _ptr(frame_idx_out), _ptr(bracket_idx_out), kind, C.byref(st))
        self._check(rc, "kmc_hip_deskew_traj_batch_f32")
        return st

    def deskew_traj_f64cols(self, x, y, z, w, stamps, knot_times, knot_poses, requested_time, ox, oy, oz, ow=None,
                            bracket_idx_out=None, raise_on_range=True):
        kind = _mem_kind(x)
        t = np.ascontiguousarray(knot_times, dtype=np.float64)
        P = np.ascontiguousarray(np.asarray(knot_poses, dtype=np.float64).reshape(len(t), 12))
        st = Stats()
        rc = lib().kmc_hip_deskew_traj_f64cols(self._h, _ptr(x), _ptr(y), _ptr(z), _ptr(w), _ptr(stamps), int(x.shape[0]),
                                               t.ctypes.data_as(_dp), P.ctypes.data_as(_dp), len(t), requested_time, _ptr(ox),
                                               _ptr(oy), _ptr(oz), _ptr(ow), _ptr(bracket_idx_out), kind, C.byref(st))
        if rc == ERR_TIME_OUT_OF_RANGE and not raise_on_range:
            return rc, st
        self._check(rc, "kmc_hip_deskew_traj_f64cols")
        return rc, st

    def pseudo_timestamps_f64(self, x, y, scan_start, scan_end, out):
        kind = _mem_kind(x)
        rc = lib().kmc_hip_pseudo_timestamps_f64(self._h, _ptr(x), _ptr(y), int(x.shape[0]), scan_start, scan_end, _ptr(out), kind)
        self._check(rc, "kmc_hip_pseudo_timestamps_f64")

    # -- N4: projection ------------------------------------------------------------------------------
    def project_f32(self, xyzi_in, rig: CameraRig, uv_out, bgrv_out, deskew: FrameParams = None, xyzi_out=None, n=None) -> Stats:
        """uv_out: (n, 4, 2) int32 -- per point, per camera, (u, v); bgrv_out: (n, 4) uint8; deskew: fuse the motion compensation in front (optional)."""
        kind = _mem_kind(xyzi_in)
        if n is None:
            n = int(xyzi_in.shape[0])
        st = Stats()
        rc = lib().kmc_hip_project_f32(self._h, _ptr(xyzi_in, np.float32), n, C.byref(rig), C.byref(deskew) if deskew is not None else None,
                                       _ptr(xyzi_out, np.float32), _ptr(uv_out, np.int32), _ptr(bgrv_out, np.uint8), kind, C.byref(st))
        self._check(rc, "kmc_hip_project_f32")
        return st

    def project_f64cols(self, x, y, z, rig: CameraRig, uv_out, bgrv_out) -> Stats:
        kind = _mem_kind(x)
        st = Stats()
        rc = lib().kmc_hip_project_f64cols(self._h, _ptr(x, np.float64), _ptr(y, np.float64), _ptr(z, np.float64), int(x.shape[0]),
                                           C.byref(rig), _ptr(uv_out, np.int32), _ptr(bgrv_out, np.uint8), kind, C.byref(st))
        self._check(rc, "kmc_hip_project_f64cols")
        return st

    def synth_points(self, out_device, n, seed):
        self._check(lib().kmc_hip_synth_points(self._h, _ptr(out_device), n, seed), "kmc_hip_synth_points")
