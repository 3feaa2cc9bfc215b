"""The process-wide pool of page-locked host memory (kmc_host_pool_*, include/kmc_hip.h) and its three switches, on the GPU: the pool is
what lets kmc::MotionCompensateFrame(Frame const&, Time) run as ONE kernel on the caller's containers (motion_compensation.cpp:16-28 takes
host containers).  Each switch is exercised in a child process (the pool reads its environment once) that reports what it saw; results
are always compared with the default route's bits -- a switch changes the route, never the result."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r'''
import ctypes as C, faulthandler, json, sys
faulthandler.dump_traceback_later(60, repeat=True)  # a stall names its line on stderr every minute (and may still resolve: the parent waits 15 minutes)
import numpy as np
sys.path.insert(0, %r)
from kitti_motion_compensation_amd import capi
import time
T0 = time.time()
def mark(what):  # progress on stderr (unbuffered), with the seconds since the start: a child that hangs says where
    sys.stderr.write("step: %%.2f s " %% (time.time() - T0) + what + "\n"); sys.stderr.flush()
L = capi.lib()
n = 60_000
rng = np.random.default_rng(11)
out = {}
p = C.c_void_p()
rc = L.kmc_host_pool_alloc(1 << 20, C.byref(p))
out["alloc_rc"] = rc
mark("pool alloc done")
if rc == capi.OK:
    out["owns"] = L.kmc_host_pool_owns(p, 1 << 20)
    out["freed"] = L.kmc_host_pool_free(p)
    q = C.c_void_p()
    L.kmc_host_pool_alloc(1 << 20, C.byref(q))
    out["reused_cached_block"] = q.value == p.value   # a cached free block is handed out again; with KMC_HOST_POOL_MAX_MB=0 nothing is cached
    L.kmc_host_pool_free(q)
    out["trimmed"] = L.kmc_host_pool_trim()
    r = C.c_void_p()
    out["alloc_near_rc"] = L.kmc_host_pool_alloc_near(1 << 20, 0, C.byref(r))
    out["alloc_near_owned"] = L.kmc_host_pool_owns(r, 1 << 20)
    L.kmc_host_pool_free(r)
mark("pool calls done")
import torch
mark("torch imported")
pts = capi.synth_points_host(n, 77)
prm = capi.FrameParams.make(np.array([1.1, 0.02, -0.01, 0.002, -0.001, 0.03]), 0.4)
with capi.Context(0) as ctx:
    ctx.enable_call_trace(True)
    mark("context created")
    # (a) ordinary pageable memory: staged copies
    a = pts.copy(); b = np.zeros_like(a)
    ctx.deskew_f32(a, b, prm)
    out["pageable_route"] = ctx.last_call_trace().route
    mark("pageable call done")
    # (b) page-locked memory the CALLER made (torch's pin_memory): in place unless KMC_HOST_DETECT_PINNED=0
    ta = torch.from_numpy(pts.copy()).pin_memory(); tb = torch.zeros_like(ta).pin_memory()
    mark("torch pinned tensors made")
    ctx.deskew_f32(ta.numpy(), tb.numpy(), prm)
    out["pinned_route"] = ctx.last_call_trace().route
    mark("pinned call done")
    out["pinned_equals_pageable"] = bool(np.array_equal(tb.numpy().view(np.uint32), b.view(np.uint32)))
    # (c) pool memory, when there is a pool
    if rc == capi.OK:
        pa = capi.PooledArray((n, 4), np.float32); pb = capi.PooledArray((n, 4), np.float32)
        pa.a[:] = pts; pb.a[:] = 0
        ctx.deskew_f32(pa.a, pb.a, prm)
        out["pool_route"] = ctx.last_call_trace().route
        out["pool_equals_pageable"] = bool(np.array_equal(pb.a.view(np.uint32), b.view(np.uint32)))
        pa.close(); pb.close()
        mark("pool call done")
mark("context closed")
print(json.dumps(out))
sys.stdout.flush()
mark("result printed")
import os
os._exit(0)  # the subject is the pool and the routes, not the teardown of torch's and HIP's pinned-memory caches in one process
'''


def _child(**env):
    e = {k: v for k, v in os.environ.items() if k not in ("KMC_HOST_POOL", "KMC_HOST_POOL_MAX_MB", "KMC_HOST_DETECT_PINNED")}
    e.update(env)
    try:
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, env=e, timeout=900)
    except subprocess.TimeoutExpired as t:
        raise AssertionError("the child hung; its last steps: " + str((t.stderr or b"")[-600:])) from None
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_default_pool_in_place_routes():
    from kitti_motion_compensation_amd import capi

    d = _child()
    assert d["alloc_rc"] == capi.OK and d["owns"] == 1 and d["freed"] == 1 and d["reused_cached_block"] is True and d["trimmed"] >= 1
    assert d["alloc_near_rc"] == capi.OK and d["alloc_near_owned"] == 1
    assert d["pageable_route"] == 0                                # staged copies: not an in-place call
    assert d["pinned_route"] == 1 and d["pool_route"] == 1         # in place over the link
    assert d["pinned_equals_pageable"] and d["pool_equals_pageable"]


def test_kmc_host_pool_0_declines_and_callers_fall_back():
    from kitti_motion_compensation_amd import capi

    d = _child(KMC_HOST_POOL="0")
    assert d["alloc_rc"] == capi.ERR_NO_DEVICE and "pool_route" not in d   # an allocation is not a computation: the deskew itself still runs on the GPU
    assert d["pageable_route"] == 0 and d["pinned_equals_pageable"]


def test_kmc_host_pool_max_mb_0_caches_nothing():
    from kitti_motion_compensation_amd import capi

    d = _child(KMC_HOST_POOL_MAX_MB="0")
    assert d["alloc_rc"] == capi.OK and d["freed"] == 1 and d["trimmed"] == 0   # the freed block was unpinned at once: nothing left to trim
    assert d["pool_route"] == 1 and d["pool_equals_pageable"]


def test_kmc_host_detect_pinned_0_keeps_foreign_pinned_memory_on_the_staged_route():
    d = _child(KMC_HOST_DETECT_PINNED="0")
    assert d["pinned_route"] == 0 and d["pinned_equals_pageable"]   # torch's pinned tensors: staged copies now
    assert d["pool_route"] == 1 and d["pool_equals_pageable"]       # the pool's own blocks stay in place
