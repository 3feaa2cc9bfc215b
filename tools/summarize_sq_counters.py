#!/usr/bin/env python3
"""Per-launch averages of the SQ / TCC counters tools/collect_sq_counters.sh collected for the bench kernel (deskew_batch_f32),
plus the ratios worth reading.   python tools/summarize_sq_counters.py gpurun_out/r02 r02   -> profiles/r02_pmc_sq_tcc.json
Another kernel of another profiled command:   ... gpurun_out/r04sqlegs r04 "deskew_f64cols<false>" f64cols   -> profiles/r04_pmc_sq_tcc_f64cols.json"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src, tag = sys.argv[1], sys.argv[2]
    kernel = sys.argv[3] if len(sys.argv) > 3 else "deskew_batch_f32"
    suffix = "_" + sys.argv[4] if len(sys.argv) > 4 else ""
    acc = collections.defaultdict(list)
    for path in sorted(glob.glob(os.path.join(src, "sq_*", "**", "*counter_collection.csv"), recursive=True)):
        with open(path) as f:
            for r in csv.DictReader(f):
                if kernel in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
    if not out:
        raise SystemExit(f"no {kernel} rows found under " + src)
    w = out.get("SQ_WAVES")
    derived = {}
    if w:
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if k in out:
                derived[k.lower().replace("sq_insts_", "") + "_per_wave"] = round(out[k] / w, 2)
    if "SQ_ACTIVE_INST_VALU" in out and "SQ_WAVE_CYCLES" in out:  # share of a resident wave's time in which it issues VALU work
        derived["valu_active_share_of_wave_cycles"] = round(out["SQ_ACTIVE_INST_VALU"] / out["SQ_WAVE_CYCLES"], 4)
    if "SQ_WAIT_ANY" in out and "SQ_WAVE_CYCLES" in out:
        derived["wave_time_waiting"] = round(out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"], 4)
    if "TCC_HIT_sum" in out and "TCC_MISS_sum" in out:
        derived["l2_hit_rate"] = round(out["TCC_HIT_sum"] / max(1.0, out["TCC_HIT_sum"] + out["TCC_MISS_sum"]), 5)
    res = {"kernel": "deskew_batch_f32 (bench.py --steps 4 --warmup 1, 256 M points per launch)" if kernel == "deskew_batch_f32" else kernel, "per_launch": out, "derived": derived,
           "launches_averaged": {k: len(v) for k, v in acc.items()}}
    os.makedirs("profiles", exist_ok=True)
    with open(os.path.join("profiles", f"{tag}_pmc_sq_tcc{suffix}.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
