#!/usr/bin/env python3
"""Several host threads, each with its own context, run the in-place f64 call (kmc_hip_deskew_f64cols on page-locked containers: the
kernel works over the link and its last wave raises the completion word the call waits for) back to back -- the pattern of the C++ test
binary's re-entrancy case, where a wait once found the stream idle without the word.  Every call's result is compared bit for bit with
the thread's first one; at the end: calls, mismatching calls, completion-word fallbacks and the state of the last one per thread.

    python tools/stress_inplace_threads.py [threads=4] [seconds=20] [points=123397] [churn=0]

churn = 1: one more thread creates and destroys contexts all the while (hipMalloc / hipFree / hipHostFree / stream creation next to the
running kernels -- what the C++ binary's threads do when they start and end).
"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kitti_motion_compensation_amd import capi  # noqa: E402


def worker(k, seconds, n, out, start):
    rng = np.random.default_rng(100 + k)
    cols = [capi.PooledArray((n,), np.float64) for _ in range(8)]  # x y z stamps | ox oy oz + a spare
    x, y, z, ts, ox, oy, oz, _ = (c.a for c in cols)
    x[:] = rng.uniform(-60, 60, n); y[:] = rng.uniform(-60, 60, n); z[:] = rng.uniform(-3, 3, n)
    ts[:] = np.sort(rng.uniform(10.0, 10.1, n))
    twist = np.array([1.2 + 0.1 * k, 0.05, -0.02, 0.004, -0.003, 0.03 + 0.002 * k])
    params = capi.FrameParams.make(twist, 0.5)
    ctx = capi.Context(0)
    ctx.deskew_f64cols(x, y, z, None, ts, 10.0, 10.1, params, ox, oy, oz)
    ref = [ox.copy(), oy.copy(), oz.copy()]
    calls = bad = 0
    start.wait()
    t_end = time.time() + seconds
    while time.time() < t_end:
        for _ in range(50):
            ox[:64] = 0.0; oz[-64:] = 0.0  # a result that never arrived would show
            ctx.deskew_f64cols(x, y, z, None, ts, 10.0, 10.1, params, ox, oy, oz)
            calls += 1
            if not (np.array_equal(ox, ref[0]) and np.array_equal(oy, ref[1]) and np.array_equal(oz, ref[2])):
                bad += 1
    n_fb, state = ctx.completion_word_fallbacks()
    out[k] = {"thread": k, "calls": calls, "mismatching_calls": bad, "completion_word_fallbacks": n_fb,
              "last_fallback_seq_word_ticket": state}
    ctx.close()
    for c in cols:
        c.close()


def churner(stop, count):
    while not stop.is_set():
        capi.Context(0).close()
        count[0] += 1


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 123397
    churn = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    out = [None] * threads
    start = threading.Event()
    stop = threading.Event()
    churned = [0]
    ch = threading.Thread(target=churner, args=(stop, churned)) if churn else None
    ws = [threading.Thread(target=worker, args=(k, seconds, n, out, start)) for k in range(threads)]
    for w in ws:
        w.start()
    time.sleep(0.5)
    start.set()
    if ch:
        ch.start()
    for w in ws:
        w.join()
    stop.set()
    if ch:
        ch.join()
    print(json.dumps({"threads": threads, "seconds": seconds, "points": n, "contexts_created_and_destroyed_meanwhile": churned[0],
                      "calls": sum(o["calls"] for o in out if o), "mismatching_calls": sum(o["mismatching_calls"] for o in out if o),
                      "completion_word_fallbacks": sum(o["completion_word_fallbacks"] for o in out if o),
                      "completion_word_fallbacks_of_the_process": int(capi.lib().kmc_hip_completion_word_fallbacks(None, None)), "per_thread": out}))


if __name__ == "__main__":
    main()
