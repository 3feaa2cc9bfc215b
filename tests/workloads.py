"""Workload builders shared by the -m gpu tests, tests/soak_config5.py and tools/measure_configs.py (test infrastructure:
this module drives the ORACLE as the checker, so it lives under tests/).

  * five_drives(): BASELINE.json configs[4] -- five concurrent KITTI drives 0001/0005/0091/0104/0117 with their real frame
    counts (108/154/340/312/660), mixed frame sizes 90-130 k points, EVERY frame with its own
    twist and its own request time -- as synthetic twins, or the real drives when KITTI_ROOT holds them;
  * kitti_root helpers: SURVEY.md section 8(d) says configs 1 and 3 "use KITTI_ROOT if present": find_drive() / load_drive()
    resolve <KITTI_ROOT>/<date>/<date>_drive_<id>_sync (or _extract) the way the reference's loaders expect a run folder
    (data_io.cpp:18-66, :97-138, :253-285).
"""
import os

import numpy as np

from tests import util

FIVE_DRIVES = (("0001", 108), ("0005", 154), ("0091", 340), ("0104", 312), ("0117", 660))
DATE = "2011_09_26"


# ---- KITTI_ROOT ------------------------------------------------------------------------------------------------------
def kitti_root():
    r = os.environ.get("KITTI_ROOT", "")
    return r if r and os.path.isdir(r) else None


def find_drive(drive_id, date=DATE):
    """Path of the run folder of drive `drive_id` under KITTI_ROOT, or None."""
    root = kitti_root()
    if not root:
        return None
    for base in (os.path.join(root, date), root):
        for suffix in ("_sync", "_extract", ""):
            p = os.path.join(base, f"{date}_drive_{drive_id}{suffix}")
            if os.path.isdir(os.path.join(p, "velodyne_points", "data")) and os.path.isdir(os.path.join(p, "oxts", "data")):
                return p
    return None


def load_drive(run_dir, max_frames=None):
    """The pieces MakeFrame needs for every frame of a run folder (data_io.cpp:253-269): scan stamps, OXTS packets, and a
    loader for the velodyne payload.  -> dict(n_frames, t_start, t_mid, t_end, oxts [dict], load_bin(i) -> (N,4) f32)."""
    vp = os.path.join(run_dir, "velodyne_points")
    with open(os.path.join(vp, "timestamps.txt")) as f:
        n = len([ln for ln in f.read().splitlines() if ln.strip()])
    n = min(n, len([fn for fn in os.listdir(os.path.join(vp, "data")) if fn.endswith(".bin")]))
    if max_frames:
        n = min(n, max_frames)
    t_start = [util.load_timestamp(os.path.join(vp, "timestamps_start.txt"), i) for i in range(n)]
    t_mid = [util.load_timestamp(os.path.join(vp, "timestamps.txt"), i) for i in range(n)]
    t_end = [util.load_timestamp(os.path.join(vp, "timestamps_end.txt"), i) for i in range(n)]
    oxts = [util.load_oxts_fields(run_dir, i) for i in range(n)]
    return dict(n_frames=n, t_start=t_start, t_mid=t_mid, t_end=t_end, oxts=oxts, run_dir=run_dir,
                load_bin=lambda i: util.load_velodyne_bin(run_dir, i))


# ---- configs[4]: five drives -------------------------------------------------------------------------------------------
def five_drives(seed=5, scale=1.0):
    """-> list of drives; drive d = dict(id, sizes (F,) int64, twists (F,6), x_req (F,), seeds (F,)).
    Frame sizes ~U[90 k, 130 k] * scale; twist of frame f: a vehicle-like screw motion that DIFFERS from frame to frame (speed
    3-25 m/s over the 0.1 s scan, yaw rate up to +-1 rad/s, small roll / pitch rates, and every 37th frame a violent one of
    |phi| ~ 0.3-0.6 rad so that a batch also mixes coefficient tiers); request time of frame f: anywhere in the scan."""
    rng = np.random.default_rng(seed)
    drives = []
    for d, (drive_id, count) in enumerate(FIVE_DRIVES):
        sizes = (rng.integers(90_000, 130_001, size=count) * scale).astype(np.int64)
        speed = rng.uniform(3.0, 25.0, size=count) * 0.1
        twists = np.stack([speed, rng.normal(0, 0.02, count), rng.normal(0, 0.01, count),
                           rng.normal(0, 0.002, count), rng.normal(0, 0.002, count), rng.uniform(-0.1, 0.1, count)], axis=1)
        wild = np.arange(count) % 37 == 11
        twists[wild, 3:] = rng.normal(0, 0.25, size=(int(wild.sum()), 3))
        x_req = rng.uniform(0.0, 1.0, size=count)
        x_req[rng.random(count) < 0.3] = 0.5  # handlers.cpp:59: requested = stamp_middle is the production caller's choice
        drives.append(dict(id=drive_id, sizes=sizes, twists=twists, x_req=x_req,
                           seeds=(1000 * (d + 1) + np.arange(count)).astype(np.int64)))
    return drives


def frame_poses(orc, twist):
    """(P_start, P_end) for a frame whose motion over the scan is Exp(twist): local-frame poses (the Mercator-scale case is covered
    by the drive twins), P_start a fixed non-trivial pose so that the host Log really works on a product."""
    P1 = orc.se3_exp([0.3, -0.2, 0.1, 0.01, -0.02, 0.4])
    return P1, orc.affine_mul(P1, orc.se3_exp(list(twist)))
