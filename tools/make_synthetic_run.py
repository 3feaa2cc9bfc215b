#!/usr/bin/env python3
"""Builds a KITTI-raw-shaped run directory from the one shipped frame (tests/golden): N frames of ~123 k points, 10 Hz cadence,
OXTS packets along a gently turning track.  Used to time kmc::MotionCompensateRun / the motion_compensate_runs CLI end to end.
  python tools/make_synthetic_run.py <out_dir> [n_frames=108]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "golden", "kitti_2011_09_26_drive_0005")


def stamp(sec):
    h, rem = divmod(sec, 3600.0)
    m, s = divmod(rem, 60.0)
    return "2011-09-26 %02d:%02d:%012.9f" % (int(h), int(m), s)


def main():
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 108
    run = os.path.join(out, "2011_09_26_drive_9999_sync")
    os.makedirs(os.path.join(run, "velodyne_points", "data"), exist_ok=True)
    os.makedirs(os.path.join(run, "oxts", "data"), exist_ok=True)
    xyzi = np.fromfile(os.path.join(SRC, "velodyne_points", "data", "0000000000.bin"), dtype=np.float32).reshape(-1, 4)
    with open(os.path.join(SRC, "oxts", "data", "0000000000.txt")) as f:
        tok = f.readline().split(" ")
    rng = np.random.default_rng(1)
    t0 = 47072.283701593
    lat, lon, yaw = float(tok[0]), float(tok[1]), float(tok[5])
    ts, tm, te, to = [], [], [], []
    for i in range(n):
        keep = rng.random(xyzi.shape[0]) < rng.uniform(0.9, 1.0)
        xyzi[keep].tofile(os.path.join(run, "velodyne_points", "data", "%010d.bin" % i))
        base = t0 + 0.1033 * i
        ts.append(stamp(base)); tm.append(stamp(base + 0.051636169)); te.append(stamp(base + 0.103272338)); to.append(stamp(base + 0.065958371))
        t = list(tok)
        yaw_i = yaw + 0.02 * i
        lat += 1.2e-5 * np.sin(yaw_i) * 0.9
        lon += 1.8e-5 * np.cos(yaw_i) * 0.9
        t[0], t[1], t[5] = "%.13f" % lat, "%.13f" % lon, "%.13f" % yaw_i
        with open(os.path.join(run, "oxts", "data", "%010d.txt" % i), "w") as f:
            f.write(" ".join(t))
    for name, rows in (("velodyne_points/timestamps_start.txt", ts), ("velodyne_points/timestamps.txt", tm),
                       ("velodyne_points/timestamps_end.txt", te), ("oxts/timestamps.txt", to)):
        with open(os.path.join(run, name), "w") as f:
            f.write("\n".join(rows) + "\n")
    print(run)


if __name__ == "__main__":
    main()
