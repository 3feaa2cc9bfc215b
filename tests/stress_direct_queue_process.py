#!/usr/bin/env python3
"""ONE process of the two-process direct-queue stress (tests/test_race_hunters.py starts two of these at once on one GPU): a context that
opted in to the direct queue (kmc_hip_set_direct_dispatch) issues the reference's calling pattern -- one kmc_hip_deskew_f32 call per frame
(handlers.cpp:55-64) -- over independent frames, a chain and in-place repeats, sweep after sweep for `seconds`, waiting through the context
only.  The first sweep's results are held to the FAITHFUL oracle; every later sweep must reproduce the first bit for bit
(motion_compensation.cpp:16-28 is pure).  Prints one JSON object.

    python tests/stress_direct_queue_process.py [seconds=6] [seed=1]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kitti_motion_compensation_amd import capi  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (the checker: this is a test tool)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch

    T0, T1 = 47072.283701593, 47072.386973931
    P1 = orc.Affine.identity()
    frames = []
    for k in range(24):
        n = 30_000 + 7_919 * ((k * 5 + seed) % 13)
        pts = capi.synth_points_host(n, 0xD12EC7 + 100 * seed + k)
        tw = np.array([1.0 + 0.03 * k, 0.02, -0.01, 0.001, -0.002, 0.02 + 0.003 * (k % 7)])
        P2 = orc.se3_exp(tw)
        tr = T0 + (T1 - T0) * (0.1 + 0.03 * k)
        prm = capi.frame_params_from_poses(P1.rt12().reshape(3, 4), P2.rt12().reshape(3, 4), T0, T1, tr)
        frames.append((pts, prm, P2, tr))
    ctx = capi.Context(0)
    ctx.set_direct_dispatch(True)
    d_in = [torch.from_numpy(f[0]).cuda() for f in frames]
    d_out = [torch.empty_like(a) for a in d_in]
    chain = [torch.empty_like(d_in[0]) for _ in range(4)]
    inplace = torch.empty_like(d_in[1])
    torch.cuda.synchronize()

    def sweep():
        for k, (a, b) in enumerate(zip(d_in, d_out)):          # independent frames: both lanes, no barrier bit
            ctx.deskew_f32(a, b, frames[k][1])
        ctx.deskew_f32(d_in[0], chain[0], frames[0][1])        # a chain: every link reads what the one before wrote
        for j in range(3):
            ctx.deskew_f32(chain[j], chain[j + 1], frames[j + 1][1])
        inplace.copy_(d_in[1])                                  # torch's stream ...
        torch.cuda.synchronize()                                # ... which nothing orders against the context but this
        for j in range(3):
            ctx.deskew_f32(inplace, inplace, frames[5 + j][1])  # in place, three times
        ctx.synchronize()                                       # the queue's rule: wait through the context
        return [b.cpu().numpy() for b in d_out] + [c.cpu().numpy() for c in chain] + [inplace.cpu().numpy()]

    first = sweep()
    worst = 0.0
    for k in range(len(frames)):  # the independent frames against the oracle
        pts, _, P2, tr = frames[k]
        ref = orc.deskew_xyzi_f32(pts, T0, P1, T1, P2, tr, mode=orc.FAITHFUL)
        assert ref["rc"] == orc.OK
        err = np.linalg.norm(first[k][:, :3].astype(np.float64) - ref["xyz_f64"], axis=1) / np.maximum(np.linalg.norm(ref["xyz_f64"], axis=1), 1e-3)
        worst = max(worst, float(err.max()))
    sweeps = bad = 0
    t_end = time.time() + seconds
    while time.time() < t_end:
        got = sweep()
        sweeps += 1
        bad += sum(0 if np.array_equal(g.view(np.uint32), f.view(np.uint32)) else 1 for g, f in zip(got, first))
    n_fb, _ = ctx.completion_word_fallbacks()
    out = {"seed": seed, "sweeps": sweeps, "frames_per_sweep": len(frames) + 7, "mismatching_buffers": bad, "max_rel_err_vs_oracle": worst,
           "direct_dispatch_active": ctx.direct_dispatch_active(), "direct_frames": ctx.direct_frames(), "frames_without_barrier_bit": ctx.any_order_launches(),
           "completion_word_fallbacks": n_fb}
    ctx.close()
    print(json.dumps(out))
    return 0 if bad == 0 and worst <= 1e-5 else 1


if __name__ == "__main__":
    sys.exit(main())
