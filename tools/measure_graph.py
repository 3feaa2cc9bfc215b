#!/usr/bin/env python3
"""One 1 M-point frame per launch (BASELINE.json config 2 taken literally): plain back-to-back launches of
kmc_hip_deskew_f32 vs the same launches captured once into a HIP graph (torch.cuda.CUDAGraph on ROCm) and replayed.
  python tools/measure_graph.py [frames=256] [points_per_frame=1000000]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kitti_motion_compensation_amd import capi  # noqa: E402


def main():
    import torch

    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    n = nf * per
    ctx = capi.Context(0)
    d_in = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.synth_points(d_in, n, 0x4B4D43)
    d_out = torch.empty_like(d_in)
    params = [capi.FrameParams.make([1.0 + 0.001 * f, 0.02, 0, 0, 0, 0.03], 0.5) for f in range(nf)]
    views_in = [d_in[f * per:(f + 1) * per] for f in range(nf)]
    views_out = [d_out[f * per:(f + 1) * per] for f in range(nf)]

    def plain():
        for f in range(nf):
            ctx.deskew_f32(views_in[f], views_out[f], params[f], n=per)

    def timed(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    ms = timed(plain)
    print(f"plain launches : {nf} x {per}: {ms * 1e3 / nf:7.2f} us per frame  {n / ms / 1e6:8.2f} G pts/s  {n * 32 / ms / 1e9:6.3f} TB/s", flush=True)
    ref = d_out.clone()

    d_out.zero_()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        ctx.set_stream(side.cuda_stream)
        plain()          # warm-up on the capture stream
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            plain()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    d_out.zero_()
    g.replay()
    torch.cuda.synchronize()
    same = bool(torch.equal(d_out.view(torch.int32), ref.view(torch.int32)))
    ms = timed(g.replay)
    print(f"graph replay   : {nf} x {per}: {ms * 1e3 / nf:7.2f} us per frame  {n / ms / 1e6:8.2f} G pts/s  {n * 32 / ms / 1e9:6.3f} TB/s  (bit-identical: {same})", flush=True)


if __name__ == "__main__":
    main()
