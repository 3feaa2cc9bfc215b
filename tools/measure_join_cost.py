#!/usr/bin/env python3
"""What a join every K frames costs a stream of 1 M-point frames over the context's four frame queues (kmc_hip_deskew_frames_f32 forks
once per call and joins at its end): K frames per call, K = 4 ... 480.   python tools/measure_join_cost.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kitti_motion_compensation_amd import capi  # noqa: E402


def main():
    import torch

    n = 1_000_000
    ctx = capi.Context(0)
    prm = capi.FrameParams.make([1.3, 0.02, -0.01, 0.001, -0.002, 0.03], 0.5)
    bufs = []
    for k in range(24):
        a = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        ctx.synth_points(a, n, 99 + k)
        bufs.append((a, torch.empty_like(a)))
    ctx.synchronize()
    torch.cuda.synchronize()
    for queues in (1, 4):
        ctx.set_frame_queues(queues)
        for per_call in (480, 4, 8, 16, 24, 32, 96, 480, 240, 480):
            packs = [ctx.prepare_frames([bufs[(c * per_call + k) % 24] for k in range(per_call)], [prm] * per_call) for c in range(480 // per_call)]

            def once():
                for p in packs:
                    ctx.deskew_frames_f32(p)

            for _ in range(2):
                once()
            ctx.timer_begin()
            for _ in range(6):
                once()
            ms = ctx.timer_end() / 6 / (len(packs) * per_call)
            print(f"queues={queues} frames_per_call={per_call:4d}: {ms * 1e3:6.3f} us per frame = {32 * n / ms / 1e9:6.3f} TB/s", flush=True)


if __name__ == "__main__":
    main()
