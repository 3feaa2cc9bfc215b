"""A default context under the ORDINARY HIP rules: a caller who knows nothing about the library's queues
issues frames with kmc_hip_deskew_f32(KMC_MEM_DEVICE), waits with hipDeviceSynchronize() ONLY, reads the results, hipFree()s the frame's
buffers, allocates again (the allocator hands the same addresses back) and overwrites them with the next frame.  That is legal HIP, and
until round 5 it was a use-after-free on the device here: the frames of a fresh context went through the direct queue, which no HIP
synchronisation covers.  Since round 6 (ABI 7) the direct queue is opt-in (kmc_hip_set_direct_dispatch) and a default context's frames sit
in its HIP stream.  10 000 frames, every one compared bit for bit with what the same frame gave at its first appearance, and each distinct
frame's first result compared with the FAITHFUL oracle (motion_compensation.cpp:16-28 is pure: the same input gives the same output,
whatever ran before).  The same loop on an opted-in context that waits through kmc_hip_synchronize (the queue's own rule) is held to the
same bar."""
import ctypes as C
import os

import numpy as np
import pytest

from kitti_motion_compensation_amd import capi
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

ITERATIONS = 10_000
DISTINCT = 12


def _hip():
    import torch  # noqa: F401  (loads libamdhip64 into the process: the library below is the one torch and libkmc_hip use)

    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"):
        try:
            h = C.CDLL(name)
            break
        except OSError:
            continue
    else:
        pytest.skip("libamdhip64 not loadable through ctypes")
    h.hipMalloc.argtypes, h.hipMalloc.restype = [C.POINTER(C.c_void_p), C.c_size_t], C.c_int
    h.hipFree.argtypes, h.hipFree.restype = [C.c_void_p], C.c_int
    h.hipMemcpy.argtypes, h.hipMemcpy.restype = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int], C.c_int
    h.hipDeviceSynchronize.argtypes, h.hipDeviceSynchronize.restype = [], C.c_int
    h.hipSetDevice.argtypes, h.hipSetDevice.restype = [C.c_int], C.c_int
    return h


def _frames():
    """DISTINCT frames of different sizes and twists -> [(points, params, oracle xyz)]"""
    out = []
    T0, T1 = 47072.283701593, 47072.386973931
    P1 = orc.Affine.identity()
    for k in range(DISTINCT):
        n = 9_000 + 4_513 * k  # 144 KB .. 940 KB per buffer: the allocator recycles the blocks among the sizes
        pts = capi.synth_points_host(n, 0x600D + k)
        tw = np.array([1.0 + 0.05 * k, 0.03, -0.01, 0.002, -0.003, 0.02 + 0.004 * k])
        P2 = orc.se3_exp(tw)
        tr = T0 + (T1 - T0) * (0.2 + 0.05 * k)
        prm = capi.frame_params_from_poses(P1.rt12().reshape(3, 4), P2.rt12().reshape(3, 4), T0, T1, tr)
        ref = orc.deskew_xyzi_f32(pts, T0, P1, T1, P2, tr, mode=orc.FAITHFUL)
        assert ref["rc"] == orc.OK
        out.append((pts, prm, ref["xyz_f64"]))
    return out


def _loop(ctx, hip, frames, wait, iterations):
    """-> (corrupted frames, worst relative error of the first results against the oracle)"""
    L = capi.lib()
    first = [None] * len(frames)
    corrupted, worst = 0, 0.0
    HOST_TO_DEVICE, DEVICE_TO_HOST = 1, 2
    poison = np.full((frames[-1][0].shape[0] + 8_192, 4), np.nan, dtype=np.float32)
    for i in range(iterations):
        k = (i * 7 + i // 5) % len(frames)
        pts, prm, ref = frames[k]
        n = pts.shape[0]
        d_in, d_out = C.c_void_p(), C.c_void_p()
        assert hip.hipMalloc(C.byref(d_in), n * 16) == 0 and hip.hipMalloc(C.byref(d_out), n * 16) == 0
        # the caller OVERWRITES what it was given: if a frame of an earlier iteration were still running on these addresses, either it
        # would read this frame's points / the poison, or its stores would land in this frame's output
        assert hip.hipMemcpy(d_out, poison.ctypes.data, n * 16, HOST_TO_DEVICE) == 0
        assert hip.hipMemcpy(d_in, pts.ctypes.data, n * 16, HOST_TO_DEVICE) == 0
        rc = L.kmc_hip_deskew_f32(ctx._h, d_in.value, d_out.value, n, C.byref(prm), capi.MEM_DEVICE, None)
        assert rc == capi.OK, capi.status_string(rc)
        wait()
        got = np.empty((n, 4), dtype=np.float32)
        assert hip.hipMemcpy(got.ctypes.data, d_out, n * 16, DEVICE_TO_HOST) == 0
        assert hip.hipFree(d_in) == 0 and hip.hipFree(d_out) == 0  # (hipFree's own implicit wait covers HIP streams, nothing else)
        if first[k] is None:
            first[k] = got
            err = np.linalg.norm(got[:, :3].astype(np.float64) - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-3)
            worst = max(worst, float(err.max()))
            assert np.array_equal(got[:, 3].view(np.uint32), pts[:, 3].view(np.uint32))
        elif not np.array_equal(got.view(np.uint32), first[k].view(np.uint32)):
            corrupted += 1
    return corrupted, worst


def test_free_and_reuse_after_hipDeviceSynchronize_only_on_a_default_context():
    hip = _hip()
    frames = _frames()
    saved = os.environ.pop("KMC_DIRECT_DISPATCH", None)  # (the suite may be running under KMC_DIRECT_DISPATCH=1: THIS test is about the default)
    try:
        ctx = capi.Context(0)
    finally:
        if saved is not None:
            os.environ["KMC_DIRECT_DISPATCH"] = saved
    try:
        assert not ctx.direct_dispatch_active()
        corrupted, worst = _loop(ctx, hip, frames, lambda: hip.hipDeviceSynchronize(), ITERATIONS)
        assert ctx.direct_frames() == 0  # every frame was a launch in the context's HIP stream
        assert corrupted == 0 and worst <= 1e-5, (corrupted, worst)
    finally:
        ctx.close()


def test_free_and_reuse_after_kmc_hip_synchronize_on_an_opted_in_context():
    """The direct queue's own rule: wait through the context.  Same loop, same bar."""
    hip = _hip()
    frames = _frames()
    ctx = capi.Context(0)
    try:
        ctx.set_direct_dispatch(True)
        corrupted, worst = _loop(ctx, hip, frames, ctx.synchronize, ITERATIONS // 2)
        if ctx.direct_dispatch_active():
            assert ctx.direct_frames() == ITERATIONS // 2
        assert corrupted == 0 and worst <= 1e-5, (corrupted, worst)
        # ... and opting out again waits for what is in the queue and returns the context to its HIP stream
        before = ctx.direct_frames()
        ctx.set_direct_dispatch(False)
        corrupted, _ = _loop(ctx, hip, frames, lambda: hip.hipDeviceSynchronize(), 500)
        assert corrupted == 0 and ctx.direct_frames() == before and not ctx.direct_dispatch_active()
    finally:
        ctx.close()
