#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of one profiling session (gpurun_out/<tag>/, scratch) into the small tracked summaries
under profiles/ that bench.py and DESIGN.md cite.

  python tools/summarize_profiles.py gpurun_out/r01 r01

Inputs expected in the session directory (see profiles/README.md for the exact gpurun command):
  kt/bench_kernel_stats.csv                rocprofv3 --kernel-trace --stats            -- python bench.py
  pmc_fetch/bench_counter_collection.csv   rocprofv3 --pmc FETCH_SIZE                  -- python bench.py --steps 4 --warmup 1
  pmc_write/bench_counter_collection.csv   rocprofv3 --pmc WRITE_SIZE                  -- python bench.py --steps 4 --warmup 1
  cal_fetch/, cal_write/                   the same two counters on tools/copy_ceiling (copy kernel of known byte count)

HBM-byte derivation (MI355X_MICROARCH.md section HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly 1/2 of the bytes of a wide coalesced stream, WRITE_SIZE must be calibrated -- both factors are re-derived here from
the float4 copy kernel whose byte count is known, in the same access pattern as the deskew kernel.
"""
import collections
import csv
import json
import os
import shutil
import sys


def counters(path, name):
    d = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == name:
                d[(r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]))].append(float(r["Counter_Value"]))
    return d


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(src, "kt", "bench_kernel_stats.csv"), os.path.join(out, f"{tag}_bench_kernel_stats.csv"))
    for extra_src, extra_dst in (("kt_legs/legs_kernel_stats.csv", f"{tag}_legs_kernel_stats.csv"), ("ceilings.csv", f"{tag}_ceilings.csv"),
                                 ("kt_legs_serial/legs_kernel_stats.csv", f"{tag}_legs_serialized_kernel_stats.csv"), ("kt_stream/stream_kernel_stats.csv", f"{tag}_frame_stream_kernel_stats.csv"),
                                 ("legs_kt.log", f"{tag}_legs_profiled_run.json"), ("bench_plain.json", f"{tag}_bench_default.json"), ("bench_detail.json", f"{tag}_bench_detail.json")):
        if os.path.exists(os.path.join(src, extra_src)):
            if extra_src.endswith(".log"):  # the legs' own JSON line of the profiled run (HIP-event figures next to the profiler's)
                with open(os.path.join(src, extra_src)) as fh:
                    lines = [ln for ln in fh.read().splitlines() if ln.startswith("{")]
                if lines:
                    with open(os.path.join(out, extra_dst), "w") as fh:
                        fh.write(lines[-1] + "\n")
            else:
                shutil.copy(os.path.join(src, extra_src), os.path.join(out, extra_dst))

    # calibration on the copy kernel: 64 Mi points x 16 B each way
    cal_f = counters(os.path.join(src, "cal_fetch", "tune_counter_collection.csv"), "FETCH_SIZE")
    cal_w = counters(os.path.join(src, "cal_write", "tune_counter_collection.csv"), "WRITE_SIZE")

    def pick(d, frag):
        for (k, grid, wg), v in d.items():
            if frag in k and grid == 67108864 and wg == 256:
                return sum(v) / len(v)
        raise KeyError(frag)

    known_kib = 67108864 * 16 / 1024.0
    fetch_factor = known_kib / pick(cal_f, "copy_points")
    write_factor = known_kib / pick(cal_w, "copy_points")

    f = counters(os.path.join(src, "pmc_fetch", "bench_counter_collection.csv"), "FETCH_SIZE")
    w = counters(os.path.join(src, "pmc_write", "bench_counter_collection.csv"), "WRITE_SIZE")
    (kname, grid, wg), fv = next((k, v) for k, v in f.items() if "deskew_batch_f32" in k[0])
    wv = next(v for k, v in w.items() if "deskew_batch_f32" in k[0])
    fetch_kib, write_kib = sum(fv) / len(fv), sum(wv) / len(wv)
    points = grid  # one lane per point, one tile per workgroup: Grid_Size == points (rounded up to the 64-lane tile)
    hbm_bytes = (fetch_kib * fetch_factor + write_kib * write_factor) * 1024.0

    stats = {}
    with open(os.path.join(src, "kt", "bench_kernel_stats.csv")) as fh:
        for r in csv.DictReader(fh):
            if "deskew_batch_f32" in r["Name"]:
                stats = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                         "max_ns": float(r["MaxNs"])}
    # the timed region alone: the per-launch trace of the same run, in launch order -- bench.py's defaults are
    # 12 spin-up + 20 warm-up + 200 timed launches, so the LAST 200 rows of the kernel are the timed region (no other leg ran here)
    timed = {}
    trace = os.path.join(src, "kt", "bench_kernel_trace.csv")
    if os.path.exists(trace):
        rows = []
        with open(trace) as fh:
            for r in csv.DictReader(fh):
                if "deskew_batch_f32" in r.get("Kernel_Name", ""):
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        rows.sort()
        last = rows[-200:]
        if last:
            d = [e - b for b, e in last]
            gaps = [last[i + 1][0] - last[i][1] for i in range(len(last) - 1)]
            timed = {"launches": len(last), "of_launches_in_trace": len(rows), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d),
                     "avg_gap_to_next_launch_ns": (sum(gaps) / len(gaps)) if gaps else None,
                     "start_to_start_avg_ns": ((last[-1][0] - last[0][0]) / (len(last) - 1)) if len(last) > 1 else None}
    # the frame-list kernel on 256 separate 1 M-point frames (tools/time_frame_stream under the same two counters): its traffic per launch
    list_traffic = {}
    lf_path, lw_path = (os.path.join(src, d, "stream_counter_collection.csv") for d in ("pmc_stream_fetch", "pmc_stream_write"))
    if os.path.exists(lf_path) and os.path.exists(lw_path):
        lf, lw = counters(lf_path, "FETCH_SIZE"), counters(lw_path, "WRITE_SIZE")
        # (round 5: the list's records ride in the kernel arguments -- deskew_list_f32<0, 256>; round 4: a device table -- <0, false>)
        pick_list = lambda d: [v for (k, grid, wg), v in d.items() if ("deskew_list_f32<0, 256>" in k or "deskew_list_f32<0, false>" in k) and grid == 256000000]
        if pick_list(lf) and pick_list(lw):
            fv2, wv2 = pick_list(lf)[0], pick_list(lw)[0]
            b = (sum(fv2) / len(fv2) * fetch_factor + sum(wv2) / len(wv2) * write_factor) * 1024.0
            list_traffic = {"kernel": "deskew_list_f32<0, 256>, 2-D grid 15625 x 256 (256 separate 1 M-point frames, records in the kernel arguments)", "launches": len(fv2),
                            "hbm_bytes_per_launch": b, "traffic_over_algorithmic": b / (32.0 * 256000000)}
        # the direct queue's per-frame kernels (AQL packets the library writes itself; rocprofv3 sees the HSA queue): traffic per 1 M-point frame
        for kern, key in (("kmc_direct_frame_t0", "direct_queue_frame"), ("kmc_direct_traj_t0", "direct_queue_nknot_frame")):
            pick = lambda d: [v for (k, grid, wg), v in d.items() if kern in k and grid == 1000000]
            if pick(lf) and pick(lw):
                fv3, wv3 = pick(lf)[0], pick(lw)[0]
                b = (sum(fv3) / len(fv3) * fetch_factor + sum(wv3) / len(wv3) * write_factor) * 1024.0
                list_traffic[key] = {"kernel": kern + " (one 1 M-point frame per AQL packet)", "dispatches": len(fv3), "hbm_bytes_per_dispatch": b, "traffic_over_algorithmic": b / 32.0e6}
    summary = {
        "tag": tag,
        "kernel": kname,
        "points_per_launch": points,
        "algorithmic_bytes_per_launch": 32 * points,
        "FETCH_SIZE_KiB_raw": fetch_kib,
        "WRITE_SIZE_KiB_raw": write_kib,
        "fetch_correction_factor": fetch_factor,
        "write_correction_factor": write_factor,
        "calibration": "tools/copy_ceiling.hip copy_points over 67108864 points (1 GiB read + 1 GiB written), same 16 B/lane nt access pattern",
        "hbm_bytes_per_launch": hbm_bytes,
        "hbm_bytes_per_point": hbm_bytes / points,
        "traffic_over_algorithmic": hbm_bytes / (32.0 * points),
        "kernel_trace_stats": stats,
        "kernel_trace_timed_region": timed,
        "frame_list_kernel_traffic": list_traffic,
    }
    with open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    with open(os.path.join(out, "pmc_traffic.json"), "w") as fh:  # the one bench.py reads
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
