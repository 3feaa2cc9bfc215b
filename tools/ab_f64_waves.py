#!/usr/bin/env python3
"""A/B of the f64 Eigen-layout kernel's waves-per-SIMD bound (KMC_F64_WAVES in kmc_kernels.hip.h, default 4) on 64 M device-resident
points, 4 bursts of 20 launches between one event pair.  Variant libraries are built next to the product's objects:

    cd kitti_motion_compensation_amd && mkdir -p lib/ab && for W in 6 8; do
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -I../include -DKMC_F64_WAVES=$W \
            -c -o lib/ab/kmc_capi_f64_w$W.o csrc/kmc_capi_f64.hip
      hipcc --offload-arch=gfx950 -fPIC -shared -o lib/ab/libkmc_hip_w$W.so lib/obj/kmc_capi_{core,deskew,traj,project,synth,hostpool,direct}.o \
            lib/ab/kmc_capi_f64_w$W.o -lhsa-runtime64; done
    python tools/ab_f64_waves.py 4|6|8          (4 = the product's library)
Round 6: tools/build_f64_variants.sh builds the variants (old = a git revision's kernel, w6, w8, tpw2, tpw2w6); any of those names is
accepted here ("base" / "4" = the product's library), and tools/ab_f64_variants.py runs them interleaved on one box.

Round 5, one box whose nine-stream copy ran at 6.20 TB/s: 4 waves 744-760 us (6.08-6.20 TB/s), 6 waves 768-801, 8 waves 783-788."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, "/root/repo")
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
which = sys.argv[1]
from kitti_motion_compensation_amd import capi
if which not in ("4", "base"):
    name = f"w{which}" if which.isdigit() else which
    capi.LIB_PATH = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib", "ab", f"libkmc_hip_{name}.so")
    assert os.path.exists(capi.LIB_PATH), capi.LIB_PATH
import torch
ctx = capi.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
turn = capi.FrameParams.make([1.3, 0.05, -0.02, 0.001, -0.002, 0.03], 0.5)
n = 64_000_000
cols = [torch.rand(n, dtype=torch.float64, device="cuda") * 80 - 40 for _ in range(3)]
w = torch.ones(n, dtype=torch.float64, device="cuda")
stamps = torch.rand(n, dtype=torch.float64, device="cuda") * 0.1 + 100.0
outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
def burst(K=20):
    for _ in range(K):
        ctx.deskew_f64cols_begin(cols[0], cols[1], cols[2], w, stamps, 100.0, 100.1, turn, *outs)
    ctx.deskew_f64cols_end()
burst(5)
res = []
for _ in range(4):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); burst(20); e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 20 * 1e3)
print(json.dumps({"waves": which, "us_per_64M": [round(r, 1) for r in res], "TBps_best": round(72 * n / min(res) / 1e6, 3)}))
