#!/usr/bin/env python3
"""What the host link of this box sustains (pinned memory, torch copies): H2D alone, D2H alone, both at once.
Context for the PCIe-inclusive numbers in DESIGN.md section 6."""
import json
import time

import torch

n = 1 << 28  # 256 MiB
h_a = torch.empty(n, dtype=torch.uint8).pin_memory()
h_b = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(h2d, d2h, reps=8):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        if h2d:
            with torch.cuda.stream(s1):
                d_a.copy_(h_a, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2):
                h_b.copy_(d_b, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    return n * reps / dt / 1e9


run(True, True, 2)
print(json.dumps({"h2d_alone_GBps": run(True, False), "d2h_alone_GBps": run(False, True), "both_each_direction_GBps": run(True, True)}))
