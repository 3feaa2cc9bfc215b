#!/usr/bin/env python3
"""What ONE device-resident single-frame call costs end to end (host issue + GPU), in order and over four frame queues:
the two-pose kmc_hip_deskew_f32, the N-knot kmc_hip_deskew_traj_f32 with three knots (records in the kernel arguments) and with
five knots (records through a device table: slot, upload, host wait).  KITTI-sized and 1 M-point frames.
   python tools/measure_call_latency.py [reps=2000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kitti_motion_compensation_amd import capi  # noqa: E402


def rt(yaw, tx, ty):
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0, tx], [s, c, 0, ty], [0, 0, 1, 0.0]])


def main():
    import torch

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    T0 = 47072.0
    t0, t1 = T0 + 0.05, T0 + 0.15
    times3 = np.array([t0 - 0.05, 0.5 * (t0 + t1) + 0.003, t1 + 0.05])
    poses3 = np.stack([rt(0.03 * k, 1.3 * k, 0.02 * k) for k in range(3)])
    times5 = np.linspace(t0 - 0.05, t1 + 0.05, 5)
    poses5 = np.stack([rt(0.03 * k, 1.3 * k, 0.02 * k) for k in range(5)])
    one = capi.FrameParams.make([1.3, 0.05, -0.02, 0.002, -0.004, 0.03], 0.5)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def bench(fn):
        for _ in range(200):
            fn()
        ctx.frame_queue_join()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        issue = time.perf_counter() - t
        ctx.frame_queue_join()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e6, issue / reps * 1e6

    for n in (123_397, 1_000_000):
        d_in = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        ctx.synth_points(d_in, n, 5)
        d_out = torch.empty_like(d_in)
        cases = {
            "2-pose            ": lambda: ctx.deskew_f32(d_in, d_out, one),
            "3 knots (kernarg) ": lambda: ctx.deskew_traj_f32(d_in, d_out, times3, poses3, t0, t1, 0.5 * (t0 + t1)),
            "5 knots (table)   ": lambda: ctx.deskew_traj_f32(d_in, d_out, times5, poses5, t0, t1, 0.5 * (t0 + t1)),
        }
        for _ in range(2):  # the first pass also pays HIP's one-off costs per kernel; report the second
            rows = []
            for queues in (1, 4):
                ctx.set_frame_queues(queues)
                for name, fn in cases.items():
                    total, issue = bench(fn)
                    rows.append(f"n={n:8d} queues={queues}  {name} {total:6.1f} us per call ({issue:5.1f} us of host issue; Python wrapper included)")
            ctx.set_frame_queues(1)
        print("\n".join(rows), flush=True)


if __name__ == "__main__":
    main()
