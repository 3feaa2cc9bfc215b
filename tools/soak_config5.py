#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU: five synthetic drives (108/154/340/312/660 frames of 90-130 k points), batches of
<= 64 M points through the batched kernel, device-resident, for a wall-clock budget (soak >= 60 s,
aggregate M points/s and its variance).  Under torch.distributed.run every rank soaks its own contiguous share of every
drive (sharding.multi_drive_ranges).   python tools/soak_config5.py [seconds=60]  -> one JSON object"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kitti_motion_compensation_amd import capi, sharding  # noqa: E402


def main():
    import torch

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    ctx = capi.Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    counts = [108, 154, 340, 312, 660]
    rng = np.random.default_rng(5)
    sizes_all = [rng.integers(90_000, 130_001, size=c) for c in counts]
    sizes = np.concatenate([sizes_all[d][a:b] for d, a, b in sharding.multi_drive_ranges(counts, rank, world)])
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(offs[-1])
    a = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    ctx.synth_points(a, n, 500 + rank)
    b = torch.empty_like(a)
    turn = capi.FrameParams.make([1.3, 0.05, -0.02, 0.001, -0.002, 0.03], 0.5)
    prepared = []
    for (i, j) in sharding.make_batches(sizes.tolist(), 0, len(sizes), max_points=int(os.environ.get("KMC_SOAK_BATCH_POINTS", "64000000"))):
        prepared.append((int(offs[i]), int(offs[j]), (offs[i:j + 1] - offs[i]).astype(np.uint64), capi.params_array([turn] * (j - i))))

    def one_pass():
        for s, e, o, p in prepared:
            ctx.deskew_batch_f32(a[s:e], b[s:e], o, p, None)

    for _ in range(5):
        one_pass()
    torch.cuda.synchronize()
    rates = []
    t_end = time.time() + budget
    t_all = time.perf_counter()
    passes = 0
    while time.time() < t_end:
        ctx.timer_begin()
        for _ in range(20):
            one_pass()
        ms = ctx.timer_end() / 20
        rates.append(n / ms / 1e3)
        passes += 20
    wall = time.perf_counter() - t_all
    r = np.array(rates)
    print(json.dumps({"rank": rank, "world": world, "frames": int(len(sizes)), "points_per_pass": n, "batches_per_pass": len(prepared),
                      "seconds": round(wall, 1), "passes": passes, "Mpts_s_wall": round(passes * n / wall / 1e6, 1),
                      "Mpts_s_mean": round(float(r.mean()), 1), "Mpts_s_min": round(float(r.min()), 1),
                      "Mpts_s_max": round(float(r.max()), 1), "Mpts_s_std": round(float(r.std()), 1),
                      "GBps_mean": round(float(r.mean()) * 32 / 1e3, 1)}))


if __name__ == "__main__":
    main()
