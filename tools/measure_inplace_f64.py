#!/usr/bin/env python3
"""The f64 Eigen-layout kernel working IN PLACE on page-locked host containers (kmc_host_pool_alloc) against the staged route on
ordinary host memory: call time (host clock) and the kernel's own time (HIP events), i.e. the link rate a kernel achieves when it
streams 40 B per point up and 32 B down at once.   python tools/measure_inplace_f64.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kitti_motion_compensation_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(0)
    params = capi.FrameParams.make([1.3, 0.05, -0.02, 0.002, -0.004, 0.03], 0.5)
    rng = np.random.default_rng(3)
    for n in (123_397, 1_000_000, 10_000_000):
        cols = rng.uniform(-40, 40, size=(4, n))
        cols[3] = 1.0
        stamps = 100.0 + rng.uniform(0, 0.1, size=n)
        pin, pst, pout = capi.PooledArray((4, n)), capi.PooledArray((n,)), capi.PooledArray((4, n))
        pin.a[:] = cols
        pst.a[:] = stamps
        ord_in = [np.ascontiguousarray(cols[j]) for j in range(4)]
        ord_out = [np.empty(n) for _ in range(4)]
        routes = {
            "in place (page-locked pool)": lambda: ctx.deskew_f64cols(pin.a[0], pin.a[1], pin.a[2], pin.a[3], pst.a, 100.0, 100.1, params, pout.a[0], pout.a[1], pout.a[2], pout.a[3]),
            "in place, no w column": lambda: ctx.deskew_f64cols(pin.a[0], pin.a[1], pin.a[2], None, pst.a, 100.0, 100.1, params, pout.a[0], pout.a[1], pout.a[2], None),
            "staged (ordinary memory)": lambda: ctx.deskew_f64cols(ord_in[0], ord_in[1], ord_in[2], ord_in[3], stamps, 100.0, 100.1, params, *ord_out),
        }
        iters = max(5, min(300, 40_000_000 // n))
        for name, fn in routes.items():
            for _ in range(5):
                fn()
            t = time.perf_counter()
            for _ in range(iters):
                fn()
            call_us = (time.perf_counter() - t) / iters * 1e6
            ctx.enable_timing(True)
            k = np.median([fn()[1].kernel_ms for _ in range(10)]) * 1e3
            ctx.enable_timing(False)
            nb = 72 if "no w" not in name else 56
            print(f"n = {n:9d}  {name:30s}: {call_us:9.1f} us per call ({n / call_us:7.1f} M pts/s); kernel {k:9.1f} us = {nb * n / k / 1e3:6.1f} GB/s of its {nb} B/point", flush=True)
        pin.close(); pst.close(); pout.close()
    ctx.close()


if __name__ == "__main__":
    main()
