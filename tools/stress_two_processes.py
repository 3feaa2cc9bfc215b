#!/usr/bin/env python3
"""The C++ test binary (kmc_api_tests gpu: in-place calls, four threads with a context each, the run driver) run again and again WHILE
this process keeps the same GPU busy with resident deskew launches of its own -- two processes sharing one device, as in the GPU suite,
where the binary's one completion-word event of round 5 happened (profiles/NOTES.md, section 10).  Prints the binary's exit codes and the
completion-word fallbacks it reported.

    python tools/stress_two_processes.py [runs=40] [busy=1] [max_seconds=0]

max_seconds > 0: no new run of the binary is started after that many seconds (the short form tests/test_race_hunters.py runs under -m gpu).
"""
import json
import os
import re
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from kitti_motion_compensation_amd import capi  # noqa: E402


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    busy = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    max_seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    import torch

    dev = torch.device("cuda", 0)
    n = 4_000_000
    pts = torch.randn(n, 4, device=dev, dtype=torch.float32)
    out = torch.empty_like(pts)
    ctx = capi.Context(0)
    params = capi.FrameParams.make(np.array([1.0, 0.1, 0.0, 0.001, 0.002, 0.03]), 0.5)
    stop = threading.Event()
    launches = [0]

    def keep_busy():
        while not stop.is_set():
            for _ in range(200):
                ctx.deskew_f32(pts, out, params)
                launches[0] += 1
            ctx.synchronize()

    th = threading.Thread(target=keep_busy)
    if busy:
        th.start()
    exe = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib", "kmc_api_tests")
    golden = os.path.join(ROOT, "tests", "golden")
    codes, fallbacks, fails, states = [], [], [], []
    with tempfile.TemporaryDirectory() as tmp:
        t_start = time.time()
        for _ in range(runs):
            if max_seconds > 0 and time.time() - t_start > max_seconds:
                break
            r = subprocess.run([exe, "gpu", golden, tmp], capture_output=True, text=True, timeout=120)
            codes.append(r.returncode)
            m = re.search(r"completion-word fallbacks of this process: (\d+)", r.stdout)
            fallbacks.append(int(m.group(1)) if m else None)
            if m and int(m.group(1)):
                states.append(r.stdout[m.start():].splitlines()[0])
            if r.returncode != 0:
                fails.append((r.stdout[-300:] + r.stderr[-600:]))
    stop.set()
    if busy:
        th.join()
    print(json.dumps({"runs": len(codes), "other_process_keeps_the_gpu_busy": bool(busy), "its_launches_meanwhile": launches[0], "nonzero_exit_codes": sum(1 for c in codes if c != 0),
                      "completion_word_fallbacks_reported": [f for f in fallbacks if f], "their_lines": states, "runs_without_a_report": sum(1 for f in fallbacks if f is None), "failures": fails[:3]}))


if __name__ == "__main__":
    main()
