#!/usr/bin/env python3
"""Times the N4 projection kernels on device-resident synthetic points (HIP events through the ctx timer).
  python tools/measure_projection.py [n_points=16777216] [iters=50]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kitti_motion_compensation_amd import capi  # noqa: E402
from tests import util  # noqa: E402


def main():
    import torch

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    tf, R_rect, P = util.load_kitti_calibration(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    rig = capi.CameraRig.make(tf, R_rect, P, 15.0)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    data = os.environ.get("KMC_PROJ_DATA", "synth")
    if data == "kitti":      # the shipped frame repeated: the validity pattern of a real scan (6 % of the points are drawn)
        frame = util.load_velodyne_bin(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kitti_2011_09_26_drive_0005"), 0)
        reps = (n + frame.shape[0] - 1) // frame.shape[0]
        d_in.copy_(torch.from_numpy(np.tile(frame, (reps, 1))[:n]))
    elif data == "front":    # worst case: every point in front of the cameras and drawn
        g = torch.Generator(device="cuda").manual_seed(1)
        d_in[:, 0] = torch.empty(n, device="cuda").uniform_(1.0, 14.0, generator=g)
        d_in[:, 1] = torch.empty(n, device="cuda").uniform_(-5.0, 5.0, generator=g)
        d_in[:, 2] = torch.empty(n, device="cuda").uniform_(-1.0, 1.0, generator=g)
        d_in[:, 3] = 0.5
    else:
        ctx.synth_points(d_in, n, 0x4B4D43)
    print("data:", data)
    d_out = torch.empty_like(d_in)
    d_uv = torch.empty((n, 4, 2), dtype=torch.int32, device="cuda")
    d_col = torch.empty((n, 4), dtype=torch.uint8, device="cuda")
    params = capi.FrameParams.make([1.3, 0.05, -0.02, 0.001, -0.002, 0.03], 0.5)
    cases = {
        "project_f32 (52 B/pt)": (lambda: ctx.project_f32(d_in, rig, d_uv, d_col), 52),
        "deskew+project (52 B/pt)": (lambda: ctx.project_f32(d_in, rig, d_uv, d_col, deskew=params), 52),
        "deskew+project+cloud (68 B/pt)": (lambda: ctx.project_f32(d_in, rig, d_uv, d_col, deskew=params, xyzi_out=d_out), 68),
    }
    for name, (fn, bpp) in cases.items():
        for _ in range(300):  # ~45 ms: clocks settled (a handful of launches after a pause times a transient)
            fn()
        ctx.synchronize()
        ctx.timer_begin()
        for _ in range(iters):
            fn()
        ms = ctx.timer_end() / iters
        print(f"{name:34s} n={n}  {ms * 1e3:9.1f} us  {n / ms / 1e6:8.2f} G pts/s  {n * bpp / ms / 1e9:7.3f} TB/s", flush=True)
    x = torch.empty(n, dtype=torch.float64, device="cuda").uniform_(-40, 40)
    y = torch.empty(n, dtype=torch.float64, device="cuda").uniform_(-40, 40)
    z = torch.empty(n, dtype=torch.float64, device="cuda").uniform_(-3, 1)
    for _ in range(300):
        ctx.project_f64cols(x, y, z, rig, d_uv, d_col)
    ctx.synchronize()
    ctx.timer_begin()
    for _ in range(iters):
        ctx.project_f64cols(x, y, z, rig, d_uv, d_col)
    ms = ctx.timer_end() / iters
    print(f"{'project_f64cols (60 B/pt)':34s} n={n}  {ms * 1e3:9.1f} us  {n / ms / 1e6:8.2f} G pts/s  {n * 60 / ms / 1e9:7.3f} TB/s")


if __name__ == "__main__":
    main()
