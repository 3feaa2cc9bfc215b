#!/usr/bin/env python3
"""Times the batched N-knot trajectory kernel (every frame with its own 3-knot trajectory, one launch) next to the batched
2-pose kernel on the same device-resident points.   python tools/measure_traj_batch.py [frames=64] [points_per_frame=1000000]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kitti_motion_compensation_amd import capi  # noqa: E402


def rt(yaw, tx, ty):
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0, tx], [s, c, 0, ty], [0, 0, 1, 0.0]])


def main():
    import torch

    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    iters = 30
    n = nf * per
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    ctx.synth_points(d_in, n, 0x4B4D43)
    d_out = torch.empty_like(d_in)
    offsets = (np.arange(nf + 1, dtype=np.uint64) * per)
    T0 = 47072.0
    frames, params = [], []
    for f in range(nf):
        t0, t1 = T0 + 0.1 * f + 0.05, T0 + 0.1 * f + 0.15
        times = [t0 - 0.05, 0.5 * (t0 + t1) + 0.003, t1 + 0.05]
        poses = np.stack([rt(0.03 * (f + k), 1.3 * (f + k), 0.02 * k) for k in range(3)])
        frames.append(dict(times=times, poses=poses, stamp_start=t0, stamp_end=t1, requested_time=0.5 * (t0 + t1)))
        params.append(capi.FrameParams.make([1.3, 0.02, 0, 0, 0, 0.03], 0.5))
    parr = capi.params_array(params)
    frames = ctx.prepare_traj_frames(frames)  # the ctypes pack, built once (Python's dict -> struct conversion is not the library's cost)
    cases = {
        "batched 2-pose (deskew_batch_f32)": lambda: ctx.deskew_batch_f32(d_in, d_out, offsets, parr),
        "batched 3-knot (deskew_traj_batch_f32)": lambda: ctx.deskew_traj_batch_f32(d_in, d_out, offsets, frames),
    }
    for name, fn in cases.items():
        for _ in range(5):
            fn()
        ctx.timer_begin()
        for _ in range(iters):
            fn()
        ms = ctx.timer_end() / iters
        ctx.enable_timing(True)
        k_ms = np.median([fn().kernel_ms for _ in range(10)])
        ctx.enable_timing(False)
        print(f"{name:42s} {nf} x {per}: call-to-call {ms * 1e3:9.1f} us; kernel alone {k_ms * 1e3:8.1f} us = {n / k_ms / 1e6:8.2f} G pts/s  {n * 32 / k_ms / 1e9:6.3f} TB/s", flush=True)


if __name__ == "__main__":
    main()
