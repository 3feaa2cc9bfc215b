#!/usr/bin/env python3
"""The f64 Eigen-layout kernel's variants (tools/build_f64_variants.sh) against each other on ONE box, INTERLEAVED -- variant a, b, c, a, b, c
... -- so that the box's drift (clock, temperature: the nine-stream copy itself moves by +-3 % within a minute) hits every variant alike;
the nine-stream copy ceiling (tools/copy_ceiling ... cols) before the first and after the last round.  Every run is its own process
(a variant is a different libkmc_hip.so).  64 M device-resident points, 72 B per point, 4 bursts of 20 launches per run.

    python tools/ab_f64_variants.py [rounds=3] [variants=base,old,tpw2,w6,w8,tpw2w6]
"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def ceiling():
    exe = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib", "copy_ceiling")
    if not os.path.exists(exe):
        return None
    r = subprocess.run([exe, "67108864", "3", "10", "0", "cols"], capture_output=True, text=True, timeout=600)
    rows = [l.split(",") for l in r.stdout.splitlines() if "," in l and not l.startswith("kernel,")]
    return {row[0]: {"us_median": float(row[3]), "GBps_median": float(row[5])} for row in rows} or r.stdout[-400:]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    variants = (sys.argv[2] if len(sys.argv) > 2 else "base,old,tpw2,w6,w8,tpw2w6").split(",")
    out = {"points": 64_000_000, "bytes_per_point": 72, "rounds": rounds, "ceiling_before": ceiling(), "runs": {v: [] for v in variants}}
    for _ in range(rounds):
        for v in variants:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_f64_waves.py"), v], capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                out["runs"][v].append({"error": r.stderr[-300:]})
                continue
            out["runs"][v].append(json.loads(r.stdout.strip().splitlines()[-1])["us_per_64M"])
    out["ceiling_after"] = ceiling()
    out["median_us"] = {v: round(statistics.median([x for run in rs if isinstance(run, list) for x in run]), 1) for v, rs in out["runs"].items() if any(isinstance(run, list) for run in rs)}
    out["TBps_at_median"] = {v: round(72 * 64e6 / us / 1e6, 3) for v, us in out["median_us"].items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
