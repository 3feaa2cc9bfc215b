"""BASELINE.json's configurations at THEIR stated shapes, through the C-ABI, against the oracle (-m gpu).

  configs[4]  five concurrent KITTI drives (0001/0005/0091/0104/0117: 108/154/340/312/660 frames), mixed frame sizes
              90-130 k points, a different twist and request time for every frame, frame-sharded for world in {1, 8} through
              sharding.multi_drive_ranges -> sharding.make_batches -> kmc_hip_deskew_batch_f32 (handlers.cpp:41-65 is the loop
              being sharded; frames are independent, motion_compensation.cpp:22-25).  EVERY point (~173 M) against the oracle,
              every per-point frame index against the offsets.
  configs[2]  one full drive (108 frames of ~121 k points at the real cadence of the shipped timestamps) -- in
              tests/test_gpu_parity.py::test_config3_drive_twin_with_real_cadence, at full frame size since round 3.
  KITTI_ROOT  when the box has the raw drives ("use KITTI_ROOT if present") the same checks run on the
              REAL frames, OXTS packets and stamps (data_io.cpp:253-285); a miniature KITTI_ROOT assembled from the shipped
              drive-0005 files keeps that code path exercised everywhere.
"""
import os
import shutil

import numpy as np
import pytest

from kitti_motion_compensation_amd import capi, sharding
from oracle import oracle as orc
from tests import util, workloads

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5
T0, T1 = 47072.283701593, 47072.386973931
MAX_BATCH_POINTS = 64_000_000


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "these tests need the GPU: there is no CPU fallback to test"
    return torch


@pytest.fixture(scope="module")
def ctx(torch_mod):
    c = capi.Context(0)
    c.set_stream(torch_mod.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def _check_frames(got, pts, frames, what):
    """got / pts: (n, 4) float32 of consecutive frames; frames: [(size, P_start, P_end, t0, t1, t_req)] -> worst literal error.
    FAITHFUL oracle on all cores: the reference's own per-point sequence -- two GetPoseAtTime calls, each with its Log and Exp
    (trajectory_interpolation.cpp:31-45) -- for every one of the ~173 M points, ~3 s per world on the GPU box's 16 cores."""
    worst, o = 0.0, 0
    assert np.array_equal(got[:, 3].view(np.uint32), pts[:, 3].view(np.uint32)), f"{what}: intensity not bit-identical"
    for n, A, B, t0, t1, tr in frames:
        r = orc.deskew_xyzi_f32(pts[o:o + n], t0, A, t1, B, tr, mode=orc.FAITHFUL)
        assert r["rc"] == orc.OK
        err = util.rel_point_error(got[o:o + n, :3], r["xyz_f64"])
        k = int(np.argmax(err)) if n else 0
        assert n == 0 or err[k] <= REL_TOL, f"{what}: {err[k]:.3e} at point {o + k} ({pts[o + k]}) > {REL_TOL}"
        worst = max(worst, float(err[k]) if n else 0.0)
        o += n
    assert o == got.shape[0]
    return worst


def test_config5_five_drives_every_point_world_1_and_8(torch_mod, ctx):
    torch = torch_mod
    drives = workloads.five_drives()
    counts = [len(d["sizes"]) for d in drives]
    assert counts == [108, 154, 340, 312, 660]
    # per drive: resident input / output, the per-frame poses (oracle side) and parameters (product side, through the C-ABI's Log)
    dev = []
    for d in drives:
        offs = np.concatenate([[0], np.cumsum(d["sizes"])]).astype(np.uint64)
        n = int(offs[-1])
        a = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        for f, (sz, seed) in enumerate(zip(d["sizes"], d["seeds"])):
            ctx.synth_points(a[int(offs[f]):int(offs[f + 1])], int(sz), int(seed))
        poses = [workloads.frame_poses(orc, tw) for tw in d["twists"]]
        params = [capi.frame_params_from_poses(A.rt12().reshape(3, 4), B.rt12().reshape(3, 4), T0, T1, T0 + x * (T1 - T0)) for (A, B), x in zip(poses, d["x_req"])]
        dev.append(dict(offs=offs, a=a, poses=poses, params=params))
    total_points = sum(int(x["offs"][-1]) for x in dev)
    assert 140_000_000 < total_points < 200_000_000

    worst, tiers = {}, set()
    for world in (1, 8):
        seen = [np.zeros(c, dtype=np.int32) for c in counts]
        want_idx = [np.full(int(x["offs"][-1]), -1, dtype=np.int32) for x in dev]
        outs = [torch.zeros_like(x["a"]) for x in dev]
        idxs = [torch.full((x["a"].shape[0],), -1, dtype=torch.int32, device="cuda") for x in dev]
        for rank in range(world):
            for (d, begin, end) in sharding.multi_drive_ranges(counts, rank, world):
                x, sizes = dev[d], drives[d]["sizes"].tolist()
                for (i, j) in sharding.make_batches(sizes, begin, end, max_points=MAX_BATCH_POINTS):
                    s, e = int(x["offs"][i]), int(x["offs"][j])
                    o = (x["offs"][i:j + 1] - x["offs"][i]).astype(np.uint64)
                    st = ctx.deskew_batch_f32(x["a"][s:e], outs[d][s:e], o, x["params"][i:j], idxs[d][s:e])
                    assert st.n_points == e - s and st.n_launches == 1
                    tiers.add(int(st.variant))
                    seen[d][i:j] += 1
                    want_idx[d][s:e] = np.repeat(np.arange(j - i, dtype=np.int32), np.diff(o).astype(np.int64))  # index = frame within ITS batch
        torch.cuda.synchronize()
        assert all((s == 1).all() for s in seen), f"world {world}: the ranks' ranges must cover every frame exactly once"
        worst[world] = 0.0
        for d, x in enumerate(dev):  # every point against the oracle, every index against the offsets
            pts = x["a"].cpu().numpy()
            got = outs[d].cpu().numpy()
            frames = [(int(sz), A, B, T0, T1, T0 + xr * (T1 - T0)) for sz, (A, B), xr in zip(drives[d]["sizes"], x["poses"], drives[d]["x_req"])]
            worst[world] = max(worst[world], _check_frames(got, pts, frames, f"world {world}, drive {drives[d]['id']}"))
            moved = np.linalg.norm(got[:, :3].astype(np.float64) - pts[:, :3], axis=1)
            assert moved.max() > 0.2, "the corrections are far above the tolerance: the check is not vacuous"
            assert np.array_equal(idxs[d].cpu().numpy(), want_idx[d]), f"world {world}, drive {drives[d]['id']}: per-point frame indices"
        del outs, idxs
    # a batch runs the coefficient tier of its widest frame: the single rank's 64 M-point batches all contain a violent frame,
    # most of the eight ranks' shorter ranges do not -- both tiers were exercised (and both were held to the bar above)
    assert len(tiers) >= 2, f"the batches were meant to mix coefficient tiers, saw {tiers}"
    print(f"configs[4]: {sum(counts)} frames, {total_points} points, worst literal error world 1: {worst[1]:.3e}, world 8: {worst[8]:.3e}, tiers {sorted(tiers)}")


# ---- KITTI_ROOT ------------------------------------------------------------------------------------------------------
def _run_real_drive(torch, ctx, drive, what):
    """frames 1 .. n-2 of a real run folder (handlers.cpp:55), requested = stamp_middle (:59), ONE batched launch, every point
    against the oracle's MakeFrame + FAITHFUL loop.  -> (frames, points, worst error)"""
    n = drive["n_frames"]
    clouds, params, frames = [], [], []
    for i in range(1, n - 1):
        pts = drive["load_bin"](i)
        co = [capi.Oxts(**drive["oxts"][i + d]) for d in (-1, 0, 1)]
        T_s, T_e = capi.make_frame_poses(co[0], co[1], co[2], drive["t_start"][i], drive["t_end"][i])
        params.append(capi.frame_params_from_poses(T_s, T_e, drive["t_start"][i], drive["t_end"][i], drive["t_mid"][i]))
        oo = [orc.oxts(**drive["oxts"][i + d]) for d in (-1, 0, 1)]
        rc, A, B = orc.make_frame_poses(oo[0], oo[1], oo[2], drive["t_start"][i], drive["t_end"][i])
        assert rc == orc.OK
        clouds.append(pts)
        frames.append((pts.shape[0], A, B, drive["t_start"][i], drive["t_end"][i], drive["t_mid"][i]))
    xyzi = np.ascontiguousarray(np.concatenate(clouds))
    offsets = np.concatenate([[0], np.cumsum([c.shape[0] for c in clouds])]).astype(np.uint64)
    d_in = torch.from_numpy(xyzi).cuda()
    d_out = torch.empty_like(d_in)
    ctx.deskew_batch_f32(d_in, d_out, offsets, params, None)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    worst, o = 0.0, 0
    assert np.array_equal(got[:, 3].view(np.uint32), xyzi[:, 3].view(np.uint32))
    for m, A, B, t0, t1, tr in frames:
        r = orc.deskew_xyzi_f32(xyzi[o:o + m], t0, A, t1, B, tr, mode=orc.FAITHFUL)
        assert r["rc"] == orc.OK, f"{what}: the reference would abort on frame at offset {o}"
        err = util.rel_point_error(got[o:o + m, :3], r["xyz_f64"])
        assert err.max() <= REL_TOL, f"{what}: {err.max():.3e}"
        worst = max(worst, float(err.max()))
        o += m
    return len(frames), xyzi.shape[0], worst


def test_kitti_root_layout_is_honoured(torch_mod, ctx, golden_dir, tmp_path, monkeypatch):
    """A miniature KITTI_ROOT (<root>/2011_09_26/2011_09_26_drive_0005_sync) assembled from the shipped drive-0005 files: the first
    five stamps / OXTS lines of the real drive, the shipped frame standing in for every scan.  find_drive / load_drive resolve it
    the way the reference's loaders address a run folder, and the frames go through the same check a real drive gets."""
    src = os.path.join(golden_dir, "kitti_2011_09_26_drive_0005")
    run = tmp_path / "2011_09_26" / "2011_09_26_drive_0005_sync"
    (run / "velodyne_points" / "data").mkdir(parents=True)
    (run / "oxts" / "data").mkdir(parents=True)
    n = 5
    for name in ("timestamps.txt", "timestamps_start.txt", "timestamps_end.txt"):
        with open(os.path.join(src, "velodyne_points", name)) as f:
            (run / "velodyne_points" / name).write_text("\n".join(f.read().splitlines()[:n]) + "\n")
    with open(os.path.join(src, "oxts", "timestamps.txt")) as f:
        (run / "oxts" / "timestamps.txt").write_text("\n".join(f.read().splitlines()[:n]) + "\n")
    with open(os.path.join(src, "oxts", "data", "0000000000.txt")) as f:
        packet = f.readline().split(" ")
    for i in range(n):
        shutil.copy(os.path.join(src, "velodyne_points", "data", "0000000000.bin"), run / "velodyne_points" / "data" / f"{i:010d}.bin")
        p = list(packet)
        p[1] = repr(float(packet[1]) + 1.2e-5 * i)  # the car drives east: ~0.9 m per packet
        p[5] = repr(float(packet[5]) + 0.01 * i)    # and turns
        (run / "oxts" / "data" / f"{i:010d}.txt").write_text(" ".join(p))
    monkeypatch.setenv("KITTI_ROOT", str(tmp_path))
    assert workloads.find_drive("0005") == str(run)
    assert workloads.find_drive("0001") is None
    drive = workloads.load_drive(str(run))
    assert drive["n_frames"] == n
    frames, points, worst = _run_real_drive(torch_mod, ctx, drive, "miniature KITTI_ROOT")
    assert frames == n - 2 and points == 3 * 123397
    print(f"miniature KITTI_ROOT: {frames} frames, {points} points, worst {worst:.3e}")


@pytest.mark.parametrize("drive_id", [d for d, _ in workloads.FIVE_DRIVES])
def test_real_kitti_drive_if_present(torch_mod, ctx, drive_id):
    """BASELINE.json configs[2] (drive 0001) and the other four drives of configs[4] on REAL data when KITTI_ROOT has them."""
    run = workloads.find_drive(drive_id)
    if run is None:
        pytest.skip(f"KITTI_ROOT has no {workloads.DATE}_drive_{drive_id}_sync (KITTI_ROOT={os.environ.get('KITTI_ROOT', '<unset>')}); the synthetic twins cover the shape")
    drive = workloads.load_drive(run)
    expected = dict(workloads.FIVE_DRIVES)[drive_id]
    assert drive["n_frames"] == expected, f"drive {drive_id}: {drive['n_frames']} frames, KITTI's metadata says {expected}"
    frames, points, worst = _run_real_drive(torch_mod, ctx, drive, f"drive {drive_id}")
    print(f"real drive {drive_id}: {frames} frames, {points} points, worst {worst:.3e}")
