#!/usr/bin/env python3
"""Times GetPseudoTimeStamps on device-resident f64 columns (16 B read + 8 B written per point).   python tools/measure_pseudo_stamps.py [n=16000000]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kitti_motion_compensation_amd import capi  # noqa: E402


def main():
    import torch

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    x = torch.rand(n, dtype=torch.float64, device="cuda") * 80 - 40
    y = torch.rand(n, dtype=torch.float64, device="cuda") * 80 - 40
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(5):
        ctx.pseudo_timestamps_f64(x, y, 100.0, 100.1, out)
    ctx.timer_begin()
    for _ in range(30):
        ctx.pseudo_timestamps_f64(x, y, 100.0, 100.1, out)
    ms = ctx.timer_end() / 30
    print(f"pseudo_timestamps_f64 n={n}: {ms * 1e3:8.1f} us  {n / ms / 1e6:7.2f} G pts/s  {n * 24 / ms / 1e9:6.3f} TB/s (24 B/pt)")


if __name__ == "__main__":
    main()
