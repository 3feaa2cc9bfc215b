#!/usr/bin/env python3
"""Times the f64 Eigen-layout kernel (the device side of kmc::MotionCompensateFrame(Frame const&, Time)) on device-resident
columns: 40 B read (x, y, z, w, stamp) + 32 B written per point.   python tools/measure_f64.py [n=16000000]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kitti_motion_compensation_amd import capi  # noqa: E402


def main():
    import torch

    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.enable_timing(True)
    turn = capi.FrameParams.make([1.3, 0.05, -0.02, 0.001, -0.002, 0.03], 0.5)
    for n in ([int(sys.argv[1])] if len(sys.argv) > 1 else [16_000_000, 16_000_001, 123_397]):
        cols = [torch.rand(n, dtype=torch.float64, device="cuda") * 80 - 40 for _ in range(3)]
        w = torch.ones(n, dtype=torch.float64, device="cuda")
        stamps = torch.rand(n, dtype=torch.float64, device="cuda") * 0.1 + 100.0
        outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
        ks = []
        for _ in range(25):
            rc, st = ctx.deskew_f64cols(cols[0], cols[1], cols[2], w, stamps, 100.0, 100.1, turn, *outs)
            ks.append(st.kernel_ms)
        k = float(np.median(ks[5:]))
        print(f"deskew_f64cols n={n}: kernel {k * 1e3:8.1f} us  {n / k / 1e6:7.2f} G pts/s  {n * 72 / k / 1e9:6.3f} TB/s (72 B/pt)")


if __name__ == "__main__":
    main()
