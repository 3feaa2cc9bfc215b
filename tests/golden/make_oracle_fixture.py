#!/usr/bin/env python3
"""Generates tests/golden/oracle_outputs_kitti_sample.npz: the CPU oracle's f64 result (FAITHFUL mode = the reference's
operation sequence) on a strided sample of the shipped KITTI frame for four synthetic trajectories (SURVEY.md section 8(c):
straight, gentle turn, hard turn |phi| ~ 0.1, stationary).  The fixture freezes the oracle: tests/test_golden_fixture.py
requires today's oracle to reproduce it and the HIP path to match it within the 1e-5 bar.

    python tests/golden/make_oracle_fixture.py        (run from the repository root)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from tests import util  # noqa: E402

T0, T1, TREQ = 47072.283701593, 47072.386973931, 47072.335337762
TRAJECTORIES = {
    "stationary": [0, 0, 0, 0, 0, 0],
    "straight": [1.3, 0.05, -0.02, 0, 0, 0],
    "gentle_turn": [1.3, 0.05, -0.02, 0.001, -0.002, 0.03],
    "hard_turn": [2.9, -0.3, 0.1, 0.02, 0.01, -0.1],
}


def main():
    run = os.path.join(ROOT, "tests", "golden", "kitti_2011_09_26_drive_0005")
    xyzi = util.load_velodyne_bin(run, 0)[::31]  # 3981 points
    P1 = orc.oxts_to_pose(orc.oxts(**util.load_oxts_fields(run, 0)))
    out = {"stride": np.int64(31), "stamp_start": np.float64(T0), "stamp_end": np.float64(T1), "requested_time": np.float64(TREQ),
           "T_start_rt12": P1.rt12()}
    for name, step in TRAJECTORIES.items():
        P2 = orc.affine_mul(P1, orc.se3_exp(step))
        r = orc.deskew_xyzi_f32(xyzi, T0, P1, T1, P2, TREQ, mode=orc.FAITHFUL, threads=1, want_stamps=True)
        assert r["rc"] == orc.OK
        out[f"{name}_step"] = np.asarray(step, dtype=np.float64)
        out[f"{name}_T_end_rt12"] = P2.rt12()
        out[f"{name}_xyz"] = r["xyz_f64"]
        out["stamps"] = r["stamps"]
    path = os.path.join(ROOT, "tests", "golden", "oracle_outputs_kitti_sample.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
