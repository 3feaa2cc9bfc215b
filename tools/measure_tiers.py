#!/usr/bin/env python3
"""Kernel time of every coefficient tier (series3 / series5 / wide polynomial / any-angle trig, kmc_device_math.hip.h) on the same
device-resident points, through the C-ABI with the tier pinned by kmc_hip_force_tier: single frame, batched, batched N-knot.
   python tools/measure_tiers.py [frames=64] [points_per_frame=1000000]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kitti_motion_compensation_amd import capi  # noqa: E402

NAMES = {capi.TIER_SERIES3: "series3", capi.TIER_SERIES5: "series5", capi.TIER_WIDE: "wide", capi.TIER_TRIG: "trig"}


def rt(yaw, tx, ty):
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0, tx], [s, c, 0, ty], [0, 0, 1, 0.0]])


def main():
    import torch

    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    n = nf * per
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    ctx.synth_points(d_in, n, 0x4B4D43)
    d_out = torch.empty_like(d_in)
    offsets = np.arange(nf + 1, dtype=np.uint64) * per
    one = capi.FrameParams.make([1.3, 0.05, -0.02, 0.002, -0.004, 0.03], 0.5)
    parr = capi.params_array([capi.FrameParams.make([1.3, 0.02 * (f % 3), 0, 0.001, 0, 0.03], 0.5) for f in range(nf)])
    T0 = 47072.0
    frames = []
    for f in range(nf):
        t0, t1 = T0 + 0.1 * f + 0.05, T0 + 0.1 * f + 0.15
        frames.append(dict(times=[t0 - 0.05, 0.5 * (t0 + t1) + 0.003, t1 + 0.05], poses=np.stack([rt(0.03 * (f + k), 1.3 * (f + k), 0.02 * k) for k in range(3)]),
                           stamp_start=t0, stamp_end=t1, requested_time=0.5 * (t0 + t1)))
    frames = ctx.prepare_traj_frames(frames)
    cases = {
        "single frame  (deskew_f32)": lambda: ctx.deskew_f32(d_in, d_out, one),
        "batched       (deskew_batch_f32)": lambda: ctx.deskew_batch_f32(d_in, d_out, offsets, parr),
        "batched 3-knot(deskew_traj_batch_f32)": lambda: ctx.deskew_traj_batch_f32(d_in, d_out, offsets, frames),
    }
    ctx.enable_timing(True)
    for rnd in range(2):  # two interleaved rounds: clocks and neighbours drift
        for name, fn in cases.items():
            for tier in sorted(NAMES):
                ctx.force_tier(tier)
                for _ in range(3):
                    fn()
                k_ms = float(np.median([fn().kernel_ms for _ in range(15)]))
                print(f"round {rnd}  {name:40s} {NAMES[tier]:8s} {nf} x {per}: kernel {k_ms * 1e3:8.1f} us = {n / k_ms / 1e6:8.2f} G pts/s  {n * 32 / k_ms / 1e9:6.3f} TB/s", flush=True)
    ctx.force_tier(-1)
    ctx.enable_timing(False)


if __name__ == "__main__":
    main()
