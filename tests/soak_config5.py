#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU (or, under torch.distributed.run, on every rank's share): the five drives of
tests/workloads.five_drives() -- 108/154/340/312/660 frames of 90-130 k points, EVERY frame with its own twist and request time --
in batches of <= 64 M points through the batched kernel, device-resident, for a wall-clock budget (soak
>= 60 s, aggregate M points/s and its variance).  Before and after the timed soak ONE pass is checked point by point against the
oracle (HOISTED, all cores) -- the soak's last output is what is checked at the end, so a kernel that drifted, or a table slot that
was recycled too early somewhere in the run, does not go unnoticed.

    python tests/soak_config5.py [seconds=60]      -> one JSON object per rank

Test infrastructure: imports the oracle as the checker (outside every timed region)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kitti_motion_compensation_amd import capi, sharding  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests import util, workloads  # noqa: E402

T0, T1 = 47072.283701593, 47072.386973931


def main():
    import torch

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("KMC_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    ctx = capi.Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    drives = workloads.five_drives()
    counts = [len(d["sizes"]) for d in drives]
    mine = sharding.multi_drive_ranges(counts, rank, world)
    sizes = np.concatenate([drives[d]["sizes"][a:b] for d, a, b in mine])
    twists = np.concatenate([drives[d]["twists"][a:b] for d, a, b in mine])
    x_req = np.concatenate([drives[d]["x_req"][a:b] for d, a, b in mine])
    seeds = np.concatenate([drives[d]["seeds"][a:b] for d, a, b in mine])
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(offs[-1])
    a = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    for f in range(len(sizes)):
        ctx.synth_points(a[int(offs[f]):int(offs[f + 1])], int(sizes[f]), int(seeds[f]))
    b = torch.empty_like(a)
    poses = [workloads.frame_poses(orc, tw) for tw in twists]
    params = [capi.frame_params_from_poses(A.rt12().reshape(3, 4), B.rt12().reshape(3, 4), T0, T1, T0 + x * (T1 - T0)) for (A, B), x in zip(poses, x_req)]
    prepared = []
    max_points = int(os.environ.get("KMC_SOAK_BATCH_POINTS", "64000000"))
    for d_begin, d_end in _drive_spans(mine):  # batches never cross a drive boundary (a drive's frames are what a rank is handed)
        for (i, j) in sharding.make_batches(sizes.tolist(), d_begin, d_end, max_points=max_points):
            prepared.append((int(offs[i]), int(offs[j]), (offs[i:j + 1] - offs[i]).astype(np.uint64), capi.params_array(params[i:j])))

    def one_pass():
        for s, e, o, p in prepared:
            ctx.deskew_batch_f32(a[s:e], b[s:e], o, p, None)

    def check(label):
        torch.cuda.synchronize()
        pts, got = a.cpu().numpy(), b.cpu().numpy()
        assert np.array_equal(got[:, 3].view(np.uint32), pts[:, 3].view(np.uint32)), f"{label}: intensity not bit-identical"
        worst = 0.0
        for f in range(len(sizes)):
            s, e = int(offs[f]), int(offs[f + 1])
            A, B = poses[f]
            r = orc.deskew_xyzi_f32(pts[s:e], T0, A, T1, B, T0 + x_req[f] * (T1 - T0), mode=orc.HOISTED)
            assert r["rc"] == orc.OK
            worst = max(worst, float(util.rel_point_error(got[s:e, :3], r["xyz_f64"]).max()))
        assert worst <= 1e-5, f"{label}: parity violated, {worst:.3e}"
        return worst

    b.zero_()
    one_pass()
    worst_before = check("first pass")
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
    worst_after = check("last pass of the soak")
    r = np.array(rates)
    print(json.dumps({"rank": rank, "world": world, "frames": int(len(sizes)), "points_per_pass": n, "batches_per_pass": len(prepared),
                      "distinct_twists": int(len({tuple(t) for t in twists.round(9).tolist()})),
                      "seconds": round(wall, 1), "passes": passes, "Mpts_s_wall": round(passes * n / wall / 1e6, 1),
                      "Mpts_s_mean": round(float(r.mean()), 1), "Mpts_s_min": round(float(r.min()), 1),
                      "Mpts_s_max": round(float(r.max()), 1), "Mpts_s_std": round(float(r.std()), 1),
                      "GBps_mean": round(float(r.mean()) * 32 / 1e3, 1), "frac_of_8TBps": round(float(r.mean()) * 32 / 1e3 / 8000, 4),
                      "parity_every_point_first_pass": worst_before, "parity_every_point_last_pass": worst_after, "bar": 1e-5}))


def _drive_spans(mine):
    """[(first frame, end frame)] of this rank's concatenated frame list, one span per drive range"""
    spans, o = [], 0
    for _, a, b in mine:
        spans.append((o, o + (b - a)))
        o += b - a
    return spans


if __name__ == "__main__":
    main()
