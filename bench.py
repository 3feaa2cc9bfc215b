#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X deskew engine (contract: see the round prompt / DESIGN.md "Measurement").

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Metric (BASELINE.json): M points/s deskewed, with the achieved HBM GB/s of the per-point transform kernel.
Workload = BASELINE.json configs[1]: synthetic 1 M-point frames, straight-line constant-velocity trajectory.  One STEP is
one pass of the hot path over one batch of FRAMES_PER_STEP distinct such frames (256 x 1 M points = 4 GB in + 4 GB out,
far beyond the 256 MiB Infinity Cache, so the GB/s are HBM GB/s) issued as ONE launch of the batched kernel through the
C-ABI (kmc_hip_deskew_batch_f32, KMC_MEM_DEVICE: inputs resident in HBM before the timed region starts).
Every rank processes its own batch (frame-sharded, weak scaling); RCCL is used only to reduce the counters.

The same invocation then measures, at every N, BASELINE.json configs[3] -- the stream north_star scales on: 10 M-point
frames x 8 000, rank r owning the contiguous frame range sharding.frame_range(8000, r, N) -- on a stated timed subset of
each rank's range, and reports it as the "configs3" object of the same JSON line (the headline stays configs[1] at every N,
so the N = 1 line of a scaling sweep is the BENCH line).

At N = 1 the same line also carries the secondary configurations as driver-observed legs, each with its time,
GB/s, fraction of the 8 TB/s peak, kernel name and an oracle spot check outside the timing:
  configs1_literal  BASELINE.json configs[1] literally: ONE 1 M-point frame per kmc_hip_deskew_f32 call -- in order on one stream,
                    and over the context's four frame queues;
  configs2_drive    configs[2]'s shape: a drive of 108 frames of ~121 k points in ONE batched launch, steady state;
  nknot3            north_star's three bracketing poses used directly: 10 M-point frames through kmc_hip_deskew_traj_f32;
  f64cols           the reference's own layout (Eigen column-major f64 + per-point stamps), 64 M points, 72 B per point.
At N > 1 the line reports the slowest and the fastest rank's own rate next to the aggregate.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured float4-copy ceiling
BYTES_PER_POINT = 32    # algorithmic: 16 B {x,y,z,intensity} read + 16 B written
POINTS_PER_FRAME = 1_000_000
SEED = 0x4B4D43
SPIN_UP_STEPS = 12  # untimed, before the W warm-up steps (see main)
C3_FRAMES_TOTAL = 8000          # BASELINE.json configs[3]: 10 M points per frame x 8 k frames, frame-sharded
C3_POINTS_PER_FRAME = 10_000_000
C3_YAW_PER_FRAME = 0.03         # |phi| ~ 0.03 rad per frame
C3_FRAMES_PER_LAUNCH = 24       # one batched launch = 24 frames = 240 M points = 7.68 GB of traffic
C3_BUFFER_GROUPS = 2            # rotating buffer groups of C3_FRAMES_PER_LAUNCH distinct frames each (15.4 GB in + out)


def make_workload(capi, n_frames, rank, yaw_per_frame=0.0, first_frame=None):
    """configs[1] trajectory: yaw = roll = pitch = 0, 10 m/s east; OXTS at T0 + {.05,.15,.25}, scan T0 + {.10,.15,.20}.
    yaw_per_frame != 0 turns it into configs[3]'s constant-twist track (|phi| ~ 0.03 per frame).  Frame f of the call is frame
    first_frame + f of the drive (default: rank * n_frames)."""
    v = 10.0
    dlon = v * 0.1 * 180.0 / (np.pi * 6378137.0)
    params = []
    if first_frame is None:
        first_frame = rank * n_frames
    for f in range(n_frames):
        Tz = 47072.0 + 0.1 * ((f + first_frame) % 8000)  # seconds since midnight stay in the reference's range
        k0 = f + first_frame
        ox = [capi.Oxts(stamp=Tz + 0.05 + 0.1 * i, lat=0.0, lon=dlon * (k0 + i), alt=0, roll=0, pitch=0,
                        yaw=((yaw_per_frame * (k0 + i) + np.pi) % (2 * np.pi)) - np.pi if yaw_per_frame else 0) for i in range(3)]
        t0, tm, t1 = Tz + 0.10, Tz + 0.15, Tz + 0.20
        T_start, T_end = capi.make_frame_poses(ox[0], ox[1], ox[2], t0, t1)
        params.append((capi.frame_params_from_poses(T_start, T_end, t0, t1, tm), (t0, tm, t1), (ox[0], ox[1], ox[2])))
    return params


def effective_cores():
    """CPU cores this process may actually use (affinity mask capped by the cgroup CPU quota; the GPU box exposes 256
    logical CPUs but grants 16 through cpu.max, and OpenMP beyond the quota only oversubscribes)."""
    from oracle import oracle as orc

    return orc.default_threads()


def cpu_baseline(xyzi_sample, frame_meta, n_frames_sample, gpu_out_frame0):
    """The oracle timed on this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).  The same leg
    also checks the GPU's output of the sample's first frame against the oracle's (parity spot check, outside any timing).
    This function is the ONLY place bench.py touches oracle/."""
    from oracle import oracle as orc

    cores = effective_cores()
    per = POINTS_PER_FRAME

    out = np.zeros((per, 4), dtype=np.float32)  # preallocated and touched: no page faults inside the timed loops

    def run(mode, threads, frames):
        t = time.perf_counter()
        pts = 0
        for f in range(frames):
            (t0, tm, t1), oxs = frame_meta[f]
            oo = [orc.oxts(o.stamp, o.lat, o.lon, o.alt, o.roll, o.pitch, o.yaw) for o in oxs]
            rc, A, B = orc.make_frame_poses(oo[0], oo[1], oo[2], t0, t1)
            r = orc.deskew_xyzi_f32(xyzi_sample[f * per:(f + 1) * per], t0, A, t1, B, tm, mode=mode, threads=threads,
                                    want_f64=False, out_f32=out)
            assert rc == orc.OK and r["rc"] == orc.OK
            pts += per
        return pts / (time.perf_counter() - t) / 1e6

    run(orc.HOISTED, cores, 1)  # spin the OpenMP team up once
    # parity spot check: the reference's op sequence (FAITHFUL) on frame 0 against what the GPU wrote for frame 0
    (t0, tm, t1), oxs = frame_meta[0]
    oo = [orc.oxts(o.stamp, o.lat, o.lon, o.alt, o.roll, o.pitch, o.yaw) for o in oxs]
    rc, A, B = orc.make_frame_poses(oo[0], oo[1], oo[2], t0, t1)
    sel = slice(0, 100_000)
    ref = orc.deskew_xyzi_f32(xyzi_sample[sel], t0, A, t1, B, tm, mode=orc.FAITHFUL)
    assert rc == orc.OK and ref["rc"] == orc.OK
    err = np.linalg.norm(gpu_out_frame0[sel, :3] - ref["xyz_f64"], axis=1) / np.maximum(np.linalg.norm(ref["xyz_f64"], axis=1), 1e-3)
    parity = {"max_rel_err": float(err.max()), "bar": 1e-5,
              "intensity_bit_identical": bool(np.array_equal(gpu_out_frame0[sel, 3], xyzi_sample[sel, 3]))}
    assert parity["max_rel_err"] <= 1e-5 and parity["intensity_bit_identical"], parity
    b1 = run(orc.FAITHFUL, 1, n_frames_sample)            # B1: the reference's op sequence, 1 thread (it is single-threaded)
    b2 = run(orc.FAITHFUL, cores, n_frames_sample)        # B2: same, OpenMP over points
    b3 = run(orc.HOISTED, cores, n_frames_sample)         # B3: hoisted closed form, all cores
    return {
        "value": round(b1, 3), "unit": "Mpts/s", "cores": 1, "kind": "port",
        "sample": f"{n_frames_sample} of the step's {per}-point frames ({n_frames_sample * per} points), oracle FAITHFUL mode "
                  "(reference op sequence incl. per-point Log/Exp), f64, 1 thread like the reference",
        "all_cores": {"cores": cores, "logical_cpus_visible": os.cpu_count(), "faithful_Mpts_s": round(b2, 3),
                      "hoisted_closed_form_Mpts_s": round(b3, 3)},
        "note": "a plain-C restatement at -O3 -ffp-contract=off: probably FASTER than the Eigen build it stands in for (SURVEY.md 3.2 estimates "
                "0.2-0.5 Mpts/s for the reference; Eigen + OpenCV are not in this image, so the reference itself cannot be timed: kind = port)",
    }, parity


def live_traffic(frames_per_step, points_per_frame, yaw_per_frame, calibration):
    """HBM bytes per launch of the bench kernel, measured NOW: two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE and
    WRITE_SIZE in separate passes, no trace domain), corrected with the factors profiles/pmc_traffic.json derived from the copy
    kernel of known size (gfx950: FETCH_SIZE counts half of a wide coalesced stream).  Returns bytes per launch or None."""
    import csv
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    total = 0.0
    for counter, factor in (("FETCH_SIZE", calibration["fetch_correction_factor"]), ("WRITE_SIZE", calibration["write_correction_factor"])):
        out = tempfile.mkdtemp(prefix="kmc_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "bench", "--", sys.executable, os.path.abspath(__file__),
               "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-live-traffic", "--no-configs3", "--no-legs", "--sustained-seconds", "0", "--frames-per-step", str(frames_per_step),
               "--points-per-frame", str(points_per_frame), "--yaw-per-frame", str(yaw_per_frame)]
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "KMC_BENCH_FORCE_DIST"):
            env.pop(k, None)
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=os.environ.get("TMPDIR", "/tmp"))
        vals = []
        for dp, _, fns in os.walk(out):
            for fn in fns:
                if fn.endswith("counter_collection.csv"):
                    with open(os.path.join(dp, fn)) as fh:
                        vals += [float(row["Counter_Value"]) for row in csv.DictReader(fh)
                                 if row["Counter_Name"] == counter and "deskew_batch_f32" in row["Kernel_Name"]]
        shutil.rmtree(out, ignore_errors=True)
        if r.returncode != 0 or not vals:
            return None
        total += sum(vals) / len(vals) * 1024.0 * factor  # counters are in KiB
    return total

def run_configs3(capi, sharding, torch, ctx, dist, rank, world, dev, timed_frames, frames_per_launch, check):
    """BASELINE.json configs[3] -- the stream north_star scales on: 10 M-point frames x 8 000, rank r owns the contiguous
    frame range sharding.frame_range(8000, r, world) (motion_compensation.cpp:22-25's callers make frames independent).
    Timed: the first `timed_frames` frames of the rank's range, C3_FRAMES_PER_LAUNCH frames per batched launch, over >= 2
    rotating buffer groups of distinct device-generated frames (their own resident buffers), constant-
    twist track with |phi| ~ 0.03 rad per frame.  The per-frame f64 host pre-step (MakeFrame + Log, ~1 us per frame in the C++
    driver) is done before the timed region, like the headline's.  Outside the timed region the rank's first and last timed
    frames are checked against the oracle (full 10 M points each).  -> dict of this rank's counters."""
    per, B, groups = C3_POINTS_PER_FRAME, frames_per_launch, C3_BUFFER_GROUPS
    d_in = torch.empty((groups * B * per, 4), dtype=torch.float32, device=dev)
    d_out = torch.empty_like(d_in)
    begin, end = sharding.frame_range(C3_FRAMES_TOTAL, rank, world)
    T = min(timed_frames, end - begin)
    T -= T % B
    assert T >= B
    n_data = groups * B  # distinct frames resident per rank
    for j in range(n_data):
        ctx.synth_points(d_in[j * per:(j + 1) * per], per, SEED + 0xC3000000 + begin + j)
    work = make_workload(capi, T, rank, C3_YAW_PER_FRAME, first_frame=begin)
    launches = []
    offsets = np.arange(B + 1, dtype=np.uint64) * per
    for l in range(T // B):
        g = l % groups
        launches.append((d_in[g * B * per:(g + 1) * B * per], d_out[g * B * per:(g + 1) * B * per],
                         capi.params_array([w[0] for w in work[l * B:(l + 1) * B]])))

    def sweep():
        for a, b, prm in launches:
            ctx.deskew_batch_f32(a, b, offsets, prm, None)

    for a, b, prm in launches[:min(len(launches), 4 * groups)]:  # untimed: table ring sized, clocks up, TLBs of the fresh buffers warm
        ctx.deskew_batch_f32(a, b, offsets, prm, None)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.timer_begin()
    sweep()
    ev_ms = ctx.timer_end()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    wall = time.perf_counter() - t0
    res = {"range": (begin, end), "timed_frames": T, "points": float(T * per), "wall": wall, "ev_s": ev_ms * 1e-3, "parity_err": 0.0,
           "groups": groups, "frames_per_launch": B}
    if check:  # parity, outside any timing: first and last timed frame of the rank, whole frames, FAITHFUL oracle
        from oracle import oracle as orc

        # the ranks share the host's cores: every rank's oracle team is sized to, and BOUND to, its own share of them (the team is
        # created by the first call below and inherits the mask), so that eight checks run side by side instead of on top of each other
        share = max(1, orc.default_threads() // world)
        allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
        if world > 1 and allowed:
            os.sched_setaffinity(0, {allowed[(rank * share + j) % len(allowed)] for j in range(share)})
        worst = 0.0
        for f in (0, T - 1):
            j = ((f // B) % groups) * B + f % B  # where frame f's points live
            src = d_in[j * per:(j + 1) * per]
            dst = d_out[:per]
            ctx.deskew_batch_f32(src, dst, np.array([0, per], dtype=np.uint64), capi.params_array([work[f][0]]), None)
            torch.cuda.synchronize()
            pts, got = src.cpu().numpy(), dst.cpu().numpy()
            (t_s, t_m, t_e), oxs = work[f][1], work[f][2]
            oo = [orc.oxts(o.stamp, o.lat, o.lon, o.alt, o.roll, o.pitch, o.yaw) for o in oxs]
            rc, A, Bp = orc.make_frame_poses(oo[0], oo[1], oo[2], t_s, t_e)
            ref = orc.deskew_xyzi_f32(pts, t_s, A, t_e, Bp, t_m, mode=orc.FAITHFUL, threads=share)
            assert rc == orc.OK and ref["rc"] == orc.OK
            err = np.linalg.norm(got[:, :3] - ref["xyz_f64"], axis=1) / np.maximum(np.linalg.norm(ref["xyz_f64"], axis=1), 1e-3)
            assert np.array_equal(got[:, 3].view(np.uint32), pts[:, 3].view(np.uint32)), "configs3: intensity not bit-identical"
            worst = max(worst, float(err.max()))
        if world > 1 and allowed:
            os.sched_setaffinity(0, set(allowed))
        assert worst <= 1e-5, f"configs3 parity violated on rank {rank}: {worst:.3e}"
        res["parity_err"] = worst
    return res


def _frac(gbps):
    return round(gbps / HBM_PEAK_GBPS, 4)


class ToolMissing(Exception):
    """A C++ benchmark client under kitti_motion_compensation_amd/lib was not built (tools/Makefile is best effort)."""


def unprofiled_env(env=None):
    """The environment of a child process without an inherited rocprofv3 tool: the C++ clients run un-profiled even when this process is
    being profiled (a counter pass, --pmc, inherited through the environment crashes a second process on the same device; the tools have
    their own passes, profiles/README.md)."""
    return {k: v for k, v in (os.environ if env is None else env).items()
            if not (k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES") and "rocprof" in v.lower()) and not k.startswith(("ROCPROF", "ROCPROFILER_"))}


def run_secondary_legs(capi, torch, ctx, dev, check):
    """N = 1 only: the secondary configurations as legs of the ONE bench line.  Every leg: device-resident inputs generated before its
    timed region, HIP events on the launch stream around it (kmc_hip_timer_begin / _end), rotating buffers beyond the 256 MiB
    Infinity Cache, and -- outside the timing -- an oracle spot check of what the GPU wrote."""
    out = {}
    orc = None
    if check:
        from oracle import oracle as orc

    def timed(fn, iters, warm, on=None):
        """ms per call.  Warm-up: `warm` calls AND at least ~40 ms of them -- an MI355X that sat idle while the previous leg was set up
        needs ~10 ms of launches to settle its clocks (round 4: the 10 M-point N-knot kernel drifted between 49 and 64 us per launch
        during a 7 ms leg that followed a pause; `profiles/r04_legs_serialized_kernel_stats.csv`)."""
        on = on or ctx
        t_warm, done = time.perf_counter(), 0
        while done < warm or time.perf_counter() - t_warm < 0.04:
            fn()
            done += 1
            if done % 16 == 0:
                on.synchronize()  # (keeps the host from running seconds ahead of the device)
        on.synchronize()
        torch.cuda.synchronize()
        on.timer_begin()
        for _ in range(iters):
            fn()
        return on.timer_end() / iters  # ms per call

    def rel_err(got, ref):
        return float((np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-3)).max())

    def run_tool(cmd, env=None, timeout=900):
        """A C++ client of the product libraries (built by __graft_entry__.build() under kitti_motion_compensation_amd/lib): its one JSON line."""
        import subprocess

        if not os.path.exists(cmd[0]):  # a benchmark CLIENT (tools/Makefile, best effort) did not build: its leg is skipped, the line still comes out
            raise ToolMissing(f"{os.path.basename(cmd[0])} is missing (tools/Makefile builds it; `python -c 'import __graft_entry__ as g; g.build()'`)")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=unprofiled_env(env))
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            raise SystemExit(f"{' '.join(cmd)} failed ({r.returncode}): {r.stdout[-1000:]} {r.stderr[-2000:]}")
        return json.loads(lines[-1])

    # ---- ceilings: the box's own memory ceilings for the kernels' access patterns, without their arithmetic (tools/copy_ceiling.hip) ----
    # (driver-observed, in the same line as the kernels they bound)
    def run_ceilings(cols_only=False):
        import subprocess

        exe = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib", "copy_ceiling")
        if not os.path.exists(exe):
            raise ToolMissing("copy_ceiling is missing (tools/Makefile builds it)")
        r = subprocess.run([exe, "67108864", "3", "10"] + (["0", "cols"] if cols_only else []), capture_output=True, text=True, timeout=600, env=unprofiled_env())
        rows = [l.split(",") for l in r.stdout.splitlines() if "," in l and not l.startswith("kernel,")]
        if r.returncode != 0 or not rows:
            raise SystemExit(f"copy_ceiling failed ({r.returncode}): {r.stdout[-500:]} {r.stderr[-1000:]}")
        return {row[0]: {"us_median": float(row[3]), "GBps_median": float(row[5]), "frac_of_peak": _frac(float(row[5]))} for row in rows}

    try:
        ceil = run_ceilings()
        out["ceilings"] = {
            "what": "tools/copy_ceiling.hip on this box, 64 Mi points per launch, median of 3 rounds x 10 launches: the access patterns of the deskew kernels with the arithmetic taken out",
            "f32_one_stream_in_one_out": {"kernel": "copy_tiles (one 64-point tile per one-wave workgroup, nt load, nt + sc1 store: the f32 kernels' pattern, 32 B/point)", **ceil.get("copy_tiles", {})},
            "f32_read_alone": ceil.get("read_points"), "f32_write_alone": ceil.get("write_points"), "f32_copy_256_thread_workgroups": ceil.get("copy_points"),
            "f64_nine_column_streams": {"kernel": "copy_cols9 (five double[n] columns read, four written, 16 B per lane and column, one 128-point tile per one-wave workgroup, nt: "
                                                  "deskew_f64cols' pattern, 72 B/point); _w4 = 4 resident waves per SIMD like the kernel, _sc1 = nt + sc1 stores",
                                        "copy_cols9": ceil.get("copy_cols9"), "copy_cols9_w4": ceil.get("copy_cols9_w4"), "copy_cols9_sc1": ceil.get("copy_cols9_sc1")},
            "f64_seven_column_streams": {"kernel": "copy_cols7 (no homogeneous column: four read, three written, 56 B/point)", "copy_cols7": ceil.get("copy_cols7"), "copy_cols7_w4": ceil.get("copy_cols7_w4")},
        }
    except ToolMissing as e:
        ceil = {}
        out["ceilings"] = {"skipped": str(e)}

    # ---- configs1_literal: one 1 M-point frame per call, driven from C++ (tools/time_frame_stream.hip) ---------------------------
    # 256 frames of 1 M points, every frame its OWN allocation (8.2 GB of distinct in / out buffers: every frame comes from HBM).  The
    # C-ABI is driven by a C++ loop, so the host language does not set the pace (a ctypes call costs ~8 us, a launch ~2).
    n = 1_000_000
    torch.cuda.synchronize()
    stream_tool = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib", "time_frame_stream")
    try:
        fs = run_tool([stream_tool, "256", str(n), "1", "8"])
    except ToolMissing as e:
        fs = None
        out["configs1_literal"] = {"skipped": str(e)}

    def stream_leg(d, key, npts, note):
        us = d[key]["us_per_frame"]
        o = {"us_per_frame": round(us, 3), "Mpts_s": round(npts / us, 1), "GBps": round(32 * npts / us / 1e3, 1), "frac": _frac(32 * npts / us / 1e3), "note": note}
        for k in ("host_us_per_call", "host_us_per_frame", "dispatched_without_barrier_bit", "through_the_direct_queue", "same_bits_as_hip_launches"):
            if k in d[key]:
                o[k] = d[key][k]
        return o

    if fs is not None:
        leg = {
            "workload": "configs[1] literally: synthetic 1 M-point frames, each in its own allocation (256 distinct frames, 8.2 GB), per-frame twists; driven from C++ through the C-ABI (tools/time_frame_stream.hip), HIP events on the context's stream, 8 timed sweeps",
            "kernel": "kmc_dev::deskew_frame_f32<series3, ppt=1, nt loads + nt|sc1 stores, block=64> per call; kmc_dev::deskew_list_f32 for the list",
            "any_order_dispatch_verdict": fs["any_order_dispatch"],
            "in_order": stream_leg(fs, "per_call", n, "ONE kmc_hip_deskew_f32 call per frame, in order on a default context's own stream: one HIP launch per frame; frames that share "
                                   "no buffer with one in flight go out without the barrier bit where the run-time probe verified it"),
            "in_order_direct_queue": stream_leg(fs, "per_call_direct_queue", n, "the same calls on a context that opted in with kmc_hip_set_direct_dispatch(ctx, 1): an AQL packet per frame in the context's "
                                                "direct queue, below the HIP runtime's launch path (through_the_direct_queue = share of the frames)") if "per_call_direct_queue" in fs else None,
            "in_order_nknot3": stream_leg(fs, "per_call_nknot3", n, "ONE kmc_hip_deskew_traj_f32 call per frame (north_star's three bracketing poses, every frame its own knots): the segment records ride in the "
                                          "launch's argument block") if "per_call_nknot3" in fs else None,
            "in_order_nknot3_direct_queue": stream_leg(fs, "per_call_nknot3_direct_queue", n, "the same calls on the opted-in context") if "per_call_nknot3_direct_queue" in fs else None,
            "in_order_drained": stream_leg(fs, "per_call_drained", n, "the same calls on a context created with KMC_ANY_ORDER=0: every dispatch waits for the last wave of the one before it"),
            "gathered_calls": stream_leg(fs, "per_call_gathered", n, "the same calls, one per frame, with kmc_hip_set_frame_queues(ctx, 4): the library gathers them on the host and issues "
                                         "ONE launch of the frame-list kernel per up to 16 frames (deferred issue, in-order results)"),
            "list_one_launch": stream_leg(fs, "list_one_launch", n, "kmc_hip_deskew_frames_f32: the 256 separate frames handed over as ONE list -> the frame-list kernel (2-D grid: frame x tile) "
                                          "in chained kernel-argument launches of 16 frames, barrier-free behind the first where verified: nothing uploaded, the host never waits "
                                          "(round 5; key name kept); bit-identical to the per-call outputs (checked by the tool: list_equals_per_call_bitwise)"),
            "batch_packed": stream_leg(fs, "batch_packed", n, "the same frames packed into one buffer, kmc_hip_deskew_batch_f32 (the headline's kernel): the ceiling for this frame mix"),
            "list_equals_per_call_bitwise": fs["list_equals_per_call_bitwise"],
        }
        assert fs["list_equals_per_call_bitwise"] is True, fs
        leg["list_launches"] = fs["list_launches"]
        if check:  # oracle spot check of the per-call entry point on this workload's first frame shape (outside any timing)
            work = make_workload(capi, 1, 0, yaw_per_frame=0.03)[0]
            prm, (t0, tm, t1), oxs = work
            a = torch.empty((100_000, 4), dtype=torch.float32, device=dev)
            ctx.synth_points(a, 100_000, SEED + 0xC1000000)
            b = torch.empty_like(a)
            ctx.deskew_f32(a, b, prm)
            torch.cuda.synchronize()
            pts, got = a.cpu().numpy(), b.cpu().numpy()
            oo = [orc.oxts(o.stamp, o.lat, o.lon, o.alt, o.roll, o.pitch, o.yaw) for o in oxs]
            rc, A, B = orc.make_frame_poses(oo[0], oo[1], oo[2], t0, t1)
            ref = orc.deskew_xyzi_f32(pts, t0, A, t1, B, tm, mode=orc.FAITHFUL)
            assert rc == orc.OK and ref["rc"] == orc.OK
            leg["parity"] = {"max_rel_err": rel_err(got[:, :3], ref["xyz_f64"]), "bar": 1e-5, "points": 100_000}
            assert leg["parity"]["max_rel_err"] <= 1e-5, leg
            del a, b
        out["configs1_literal"] = leg
    caller_stream = torch.cuda.current_stream().cuda_stream
    state = {"k": 0}

    # ---- configs2_drive: 108 frames of ~121 k points, one batched launch ---------------------------------------------------------
    rng = np.random.default_rng(SEED + 2)
    sizes = np.clip(rng.normal(121_000, 3_000, size=108), 90_000, 140_000).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    ntot = int(offs[-1])
    dwork = make_workload(capi, 108, 0, yaw_per_frame=0.03)  # a turning drive: every frame its own poses through the whole host pre-step
    dparams = capi.params_array([w[0] for w in dwork])
    sets = []
    for k in range(3):
        a = torch.empty((ntot, 4), dtype=torch.float32, device=dev)
        ctx.synth_points(a, ntot, SEED + 0xC2000000 + k)
        sets.append((a, torch.empty_like(a)))
    state["k"] = 0

    def drive():
        a, b = sets[state["k"] % 3]
        state["k"] += 1
        ctx.deskew_batch_f32(a, b, offs, dparams, None)

    ms = min(timed(drive, 300, 300 if r == 0 else 0) for r in range(3))
    leg = {"workload": f"configs[2] shape: a drive of 108 frames ~N(121 k, 3 k) points ({ntot} points) in ONE batched launch, 3 rotating buffer sets, best of 3 x 300 launches",
           "kernel": "kmc_dev::deskew_batch_f32<series3, ppt=1, nt loads + nt|sc1 stores, block=64>",
           "us_per_drive": round(ms * 1e3, 2), "Mpts_s": round(ntot / ms / 1e3, 1), "GBps": round(32 * ntot / ms / 1e6, 1), "frac": _frac(32 * ntot / ms / 1e6)}
    if check:
        f = 57
        s0, s1 = int(offs[f]), int(offs[f + 1])
        a, b = sets[(state["k"] - 1) % 3]
        pts, got = a[s0:s1].cpu().numpy(), b[s0:s1].cpu().numpy()
        (t0, tm, t1), oxs = dwork[f][1], dwork[f][2]
        oo = [orc.oxts(o.stamp, o.lat, o.lon, o.alt, o.roll, o.pitch, o.yaw) for o in oxs]
        rc, A, B = orc.make_frame_poses(oo[0], oo[1], oo[2], t0, t1)
        ref = orc.deskew_xyzi_f32(pts, t0, A, t1, B, tm, mode=orc.FAITHFUL)
        assert rc == orc.OK and ref["rc"] == orc.OK
        leg["parity"] = {"max_rel_err": rel_err(got[:, :3], ref["xyz_f64"]), "bar": 1e-5, "points": int(s1 - s0), "frame": f}
        assert leg["parity"]["max_rel_err"] <= 1e-5, leg
    # the same drive shape FRAME BY FRAME from C++ (the reference's calling pattern, handlers.cpp:55-64): 108 frames ~N(121 k, 3 k), each in
    # its own allocation, 3 rotating sets; one kmc_hip_deskew_f32 call per frame, the list call, and the packed batch beside them
    del sets
    torch.cuda.empty_cache()
    try:
        fd = run_tool([stream_tool, "108", "kitti", "3", "50"])
        npts = fd["mean_points_per_frame"]
        leg["frame_by_frame_from_c"] = {
            "workload": "108 separate frames ~N(121 k, 3 k) points, each in its own allocation, 3 rotating sets, per-frame twists; tools/time_frame_stream.hip, 50 timed sweeps",
            "mean_points_per_frame": npts, "any_order_dispatch_verdict": fd["any_order_dispatch"],
            "per_call": stream_leg(fd, "per_call", npts, "one kmc_hip_deskew_f32 call per frame, in order on a default context's own stream: one HIP launch per frame (the runtime's launch path)"),
            "per_call_direct_queue": stream_leg(fd, "per_call_direct_queue", npts, "the same calls after kmc_hip_set_direct_dispatch(ctx, 1): an AQL packet per frame in the context's direct queue") if "per_call_direct_queue" in fd else None,
            "per_call_nknot3": stream_leg(fd, "per_call_nknot3", npts, "one kmc_hip_deskew_traj_f32 call per frame (three knots, the records in the launch's argument block)") if "per_call_nknot3" in fd else None,
            "per_call_nknot3_direct_queue": stream_leg(fd, "per_call_nknot3_direct_queue", npts, "the same calls on the opted-in context") if "per_call_nknot3_direct_queue" in fd else None,
            "per_call_drained": stream_leg(fd, "per_call_drained", npts, "KMC_ANY_ORDER=0: the barrier bit on every dispatch"),
            "per_call_gathered": stream_leg(fd, "per_call_gathered", npts, "the same calls with kmc_hip_set_frame_queues(ctx, 4): gathered on the host, one list launch per up to 16 frames"),
            "list_one_launch": stream_leg(fd, "list_one_launch", npts, "kmc_hip_deskew_frames_f32: the 108 separate frames as one list -> 7 chained kernel-argument launches of <= 16 frames, "
                                          "barrier-free behind the first where verified (round 5: no table upload, no host wait; key name kept)"),
            "list_launches": fd.get("list_launches"),
            "batch_packed": stream_leg(fd, "batch_packed", npts, "the same frames packed into one buffer, one batched launch"),
            "per_call_rate_vs_batched": round(fd["batch_packed"]["us_per_frame"] / fd["per_call"]["us_per_frame"], 3),
            "direct_queue_rate_vs_batched": round(fd["batch_packed"]["us_per_frame"] / fd["per_call_direct_queue"]["us_per_frame"], 3) if "per_call_direct_queue" in fd else None,
            "gathered_rate_vs_batched": round(fd["batch_packed"]["us_per_frame"] / fd["per_call_gathered"]["us_per_frame"], 3),
            "list_rate_vs_batched": round(fd["batch_packed"]["us_per_frame"] / fd["list_one_launch"]["us_per_frame"], 3),
            "list_equals_per_call_bitwise": fd["list_equals_per_call_bitwise"],
        }
        assert fd["list_equals_per_call_bitwise"] is True, fd
    except ToolMissing as e:
        leg["frame_by_frame_from_c"] = {"skipped": str(e)}
    out["configs2_drive"] = leg

    # ---- nknot3: three bracketing poses used directly, 10 M-point frames ---------------------------------------------------------
    n = 10_000_000
    Tz = 47072.0
    knot_t = [Tz + 0.05, Tz + 0.15, Tz + 0.25]
    P = []
    for k in range(3):
        M = np.eye(4)[:3].copy()
        c, s_ = np.cos(0.03 * k), np.sin(0.03 * k)
        M[:2, :2] = [[c, -s_], [s_, c]]
        M[:, 3] = [1.3 * k, 0.02 * k * k, 0.0]
        P.append(M)
    P = np.stack(P)
    bufs = []
    for k in range(3):
        a = torch.empty((n, 4), dtype=torch.float32, device=dev)
        ctx.synth_points(a, n, SEED + 0xC4000000 + k)
        bufs.append((a, torch.empty_like(a)))
    state["k"] = 0

    def traj():
        a, b = bufs[state["k"] % 3]
        state["k"] += 1
        ctx.deskew_traj_f32(a, b, knot_t, P, Tz + 0.10, Tz + 0.20, Tz + 0.15, None)

    torch.cuda.synchronize()
    ctx.set_stream(None)  # the context's own stream, like configs1_literal
    ao_before, calls_before = ctx.any_order_launches(), state["k"]
    ms = timed(traj, 120, 12)
    ao_share = (ctx.any_order_launches() - ao_before) / max(1, state["k"] - calls_before)  # (the warm-up runs for at least 40 ms: count the calls made)
    ctx.synchronize()
    ctx.set_stream(caller_stream)
    leg = {"workload": "north_star's three bracketing poses used directly (piecewise geodesic, 2 segments): one synthetic 10 M-point frame per kmc_hip_deskew_traj_f32 call, 3 rotating buffer pairs",
           "kernel": "kmc_dev::deskew_traj_f32<series3, nt loads + nt|sc1 stores, inline records> (no LDS: records through scalar loads)",
           "us_per_frame": round(ms * 1e3, 2), "Mpts_s": round(n / ms / 1e3, 1), "GBps": round(32 * n / ms / 1e6, 1), "frac": _frac(32 * n / ms / 1e6),
           "dispatched_without_barrier_bit": round(ao_share, 3),
           "note": "call to call on one stream: the kernel itself (rocprofv3 row in profiles/) plus the drain / launch gap between two frames (three rotating buffer pairs: two of three frames go out without the barrier bit)"}
    if check:
        sel = slice(4_950_000, 5_050_000)  # around mid-scan: both segments
        a, b = bufs[(state["k"] - 1) % 3]
        pts, got = a[sel].cpu().numpy(), b[sel].cpu().numpy()
        poses = [orc.Affine.from_Rt(M[:, :3], M[:, 3]) for M in P]
        ref = orc.deskew_xyzi_f32_traj(pts, Tz + 0.10, Tz + 0.20, knot_t, poses, Tz + 0.15)
        assert ref["rc"] == orc.OK
        leg["parity"] = {"max_rel_err": rel_err(got[:, :3], ref["xyz_f64"]), "bar": 1e-5, "points": 100_000,
                         "segments_seen": sorted(set(ref["bracket_by_time"].tolist()))}
        assert leg["parity"]["max_rel_err"] <= 1e-5, leg
    out["nknot3"] = leg
    del bufs

    # ---- f64cols: the reference's own layout, device resident ----------------------------------------------------------------------
    n = 64_000_000  # 4.6 GB of columns per launch: far beyond the 256 MiB Infinity Cache
    turn = capi.FrameParams.make([1.3, 0.05, -0.02, 0.002, -0.004, 0.03], 0.5)
    g = torch.Generator(device=dev)
    g.manual_seed(SEED)
    cols = [torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 80 - 40 for _ in range(3)]
    w = torch.ones(n, dtype=torch.float64, device=dev)
    stamps = torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 0.1 + 100.0
    outs = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(4)]
    # K launches back to back between ONE event pair, like every other leg: kmc_hip_deskew_f64cols_begin x K (device-resident columns:
    # each only enqueues its launch), then one _end (the combined out-of-range verdict).  Next to it the per-call figure: one
    # kmc_hip_deskew_f64cols call, which ends with the host waiting for its verdict, timed by the library's own event pair around the kernel.
    K64 = 20

    def f64_burst():
        for _ in range(K64):
            ctx.deskew_f64cols_begin(cols[0], cols[1], cols[2], w, stamps, 100.0, 100.1, turn, *outs)
        ctx.deskew_f64cols_end()

    def adjacent_ceiling():  # the nine- and seven-stream copies in the minute (clock, temperature) the kernel is measured in
        try:
            return run_ceilings(cols_only=True)
        except (ToolMissing, SystemExit):
            return {}

    torch.cuda.synchronize()
    ceil_before = adjacent_ceiling()
    ms = timed(f64_burst, 3, 1) / K64

    def f64_burst_ones():  # the homogeneous column KNOWN to be ones (what the C++ drop-in passes for every cloud its loaders made): neither read nor written
        for _ in range(K64):
            ctx.deskew_f64cols_begin(cols[0], cols[1], cols[2], None, stamps, 100.0, 100.1, turn, outs[0], outs[1], outs[2], None)
        ctx.deskew_f64cols_end()

    ms_ones = timed(f64_burst_ones, 3, 1) / K64
    torch.cuda.synchronize()
    ceil_after = adjacent_ceiling()
    ctx.enable_timing(True)
    for _ in range(3):
        ctx.deskew_f64cols(cols[0], cols[1], cols[2], w, stamps, 100.0, 100.1, turn, *outs)
    ms_call = float(np.median([ctx.deskew_f64cols(cols[0], cols[1], cols[2], w, stamps, 100.0, 100.1, turn, *outs)[1].kernel_ms for _ in range(20)]))
    ctx.enable_timing(False)
    leg = {"workload": "the reference's layout: four f64 columns (Eigen::MatrixX4d, column-major) + f64 per-point stamps, 64 M points, device resident; 40 B read + 32 B written per point",
           "kernel": "kmc_dev::deskew_f64cols (one wave per workgroup, two points per lane)", "bytes_per_point": 72,
           "us_per_call": round(ms * 1e3, 2), "Mpts_s": round(n / ms / 1e3, 1), "GBps": round(72 * n / ms / 1e6, 1), "frac": _frac(72 * n / ms / 1e6),
           "timed_as": f"{K64} launches back to back (kmc_hip_deskew_f64cols_begin x {K64}, one _end) between ONE HIP-event pair on the launch stream, 3 bursts after a warm-up burst",
           "homogeneous_column_known_to_be_ones": {
               "bytes_per_point": 56, "us_per_call": round(ms_ones * 1e3, 2), "Mpts_s": round(n / ms_ones / 1e3, 1), "GBps": round(56 * n / ms_ones / 1e6, 1), "frac": _frac(56 * n / ms_ones / 1e6),
               "note": "w == NULL, ow == NULL: Affine3d * (x, y, z, 1) needs no w (motion_compensation.cpp:13); what kmc::MotionCompensateFrame passes for clouds whose column is known to be ones (data_io.cpp:130)"},
           "single_call": {"us_per_call": round(ms_call * 1e3, 2), "GBps": round(72 * n / ms_call / 1e6, 1), "frac": _frac(72 * n / ms_call / 1e6),
                           "note": "one kmc_hip_deskew_f64cols call at a time (each ends with the host waiting for the out-of-range verdict): the library's own event pair around the lone launch, median of 20"}}
    if check:
        m = 50_000
        cl = np.stack([c[:m].cpu().numpy() for c in cols] + [np.ones(m)], axis=1)
        st = stamps[:m].cpu().numpy()
        A = orc.se3_exp([0.0] * 6)
        B = orc.se3_exp(list(turn.twist))
        rc, nbad, want = orc.motion_compensate_frame(cl, st, 100.0, A, 100.1, B, 100.05)
        got = np.stack([o[:m].cpu().numpy() for o in outs[:3]], axis=1)
        assert rc == orc.OK
        leg["parity"] = {"max_rel_err": rel_err(got, want[:, :3]), "bar": 1e-11, "points": m}
        assert leg["parity"]["max_rel_err"] <= 1e-11, leg
    c9 = [v["GBps_median"] for k, v in (ceil or {}).items() if k.startswith("copy_cols9") and v]
    c7 = [v["GBps_median"] for k, v in (ceil or {}).items() if k.startswith("copy_cols7") and v]
    if c9:  # the kernel against the box's own ceiling for nine column streams (best of the copy variants measured in this run)
        leg["ceiling_9_streams_GBps"] = max(c9)
        leg["frac_of_9_stream_ceiling"] = round(leg["GBps"] / max(c9), 4)
        leg["l2_hit_rate_explained"] = ("profiles/r04_pmc_sq_tcc_f64cols.json: TCC hits 16.0 M = exactly half of the 32.0 M write requests -- a 128-byte line arrives as two 64-byte "
                                        "write requests, the second meets the first in the L2; every one of the 20.0 M read requests (128 B each = the 2.56 GB read) misses: nothing is re-read")
    adj9 = [v["GBps_median"] for cc in (ceil_before, ceil_after) for k, v in cc.items() if k.startswith("copy_cols9")]
    adj7 = [v["GBps_median"] for cc in (ceil_before, ceil_after) for k, v in cc.items() if k.startswith("copy_cols7")]
    if adj9:  # the same copies run right before and right after the kernel's bursts: no half minute of clock / temperature drift in between
        leg["ceiling_9_streams_adjacent_GBps"] = {"before": max(v["GBps_median"] for k, v in ceil_before.items() if k.startswith("copy_cols9")) if ceil_before else None,
                                                  "after": max(v["GBps_median"] for k, v in ceil_after.items() if k.startswith("copy_cols9")) if ceil_after else None,
                                                  "is": "best copy variant (copy_cols9 / _w4 / _sc1, median of 3 rounds x 10 launches) of tools/copy_ceiling ... cols, run directly before and directly after the bursts above"}
        leg["frac_of_adjacent_9_stream_ceiling"] = round(leg["GBps"] / max(adj9), 4)
    if adj7:
        leg["homogeneous_column_known_to_be_ones"]["frac_of_adjacent_7_stream_ceiling"] = round(leg["homogeneous_column_known_to_be_ones"]["GBps"] / max(adj7), 4)
    if c7:
        leg["homogeneous_column_known_to_be_ones"]["ceiling_7_streams_GBps"] = max(c7)
        leg["homogeneous_column_known_to_be_ones"]["frac_of_7_stream_ceiling"] = round(leg["homogeneous_column_known_to_be_ones"]["GBps"] / max(c7), 4)
    out["f64cols"] = leg
    del cols, w, stamps, outs
    torch.cuda.empty_cache()
    try:
        out["dropin_cpp"] = run_dropin_cpp_leg(run_tool, rel_err, orc if check else None)
    except ToolMissing as e:
        out["dropin_cpp"] = {"skipped": str(e)}
    # ---- sharded_cpp: north_star's multi-GPU shape driven from C++ in one process (tools/run_sharded.hip): frame ranges per device, ONE native
    # RCCL reduction of the counters (ncclCommInitAll + ncclAllReduce sum / max).  At N = 1 a world of one; the driver's 8-GPU box can run
    # `run_sharded 0,1,2,3,4,5,6,7` -- unmeasured on hardware so far (SCALE was skipped in every round).
    try:
        sh = run_tool([os.path.join(ROOT, "kitti_motion_compensation_amd", "lib", "run_sharded"), "all", "64", "1000000", "8"])  # one rank per VISIBLE device: 1 here, a real RCCL group on a multi-GPU node
        out["sharded_cpp"] = {"what": "tools/run_sharded.hip: 64 synthetic 1 M-point frames in contiguous frame ranges per rank (kmc_frame_ranges_balanced), one device context per rank, "
                                      "8 frames per batched launch; counters reduced by ONE RCCL group (librccl dlopen'ed by the tool only)", **sh}
        assert sh["reduction_agrees_with_host_arithmetic"] is True, sh
    except ToolMissing as e:
        out["sharded_cpp"] = {"skipped": str(e)}
    return out


PCIE_GBPS_PER_DIRECTION = 63.0  # stated peak of the box's link: PCIe 5.0 x16 = 32 GT/s x 16 lanes x 128/130 = 63.0 GB/s each way (full duplex)
PCIE_MEASURED_DUPLEX_GBPS = 48.5  # what the copy engines move EACH way when both directions run (profiles/r04_pcie_probe.txt; alone: 57) -- the denominator VERDICT r04 #2 asks for


def run_dropin_cpp_leg(run_tool, rel_err, orc):
    """the API north_star names, through the C++ drop-in library (libkitti_motion_compensation_lib.so -- no ctypes, no
    torch): kmc::MotionCompensateFrame(Frame const&, Time) (motion_compensation.cpp:16-28) and hip::MotionCompensateKittiCloud on the
    shipped 123 397-point KITTI frame held in HOST containers -- the drop-in's own page-locked containers, and ordinary pageable ones
    (KMC_HOST_POOL=0) --, and kmc::MotionCompensateRun (handlers.cpp:41-65) on a synthetic KITTI-raw-shaped drive through the reference's
    CLI.  PCIe-inclusive by construction (the API hands over host memory): reported against the link's stated peak, never the headline.
    Parity: the clouds the C++ calls returned, and a frame the run driver wrote, against the oracle (outside any timing)."""
    import shutil
    import subprocess
    import tempfile

    lib_dir = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib")
    golden = os.path.join(ROOT, "tests", "golden")
    tmp = tempfile.mkdtemp(prefix="kmc_dropin_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        prefix = os.path.join(tmp, "frame")
        pooled = run_tool([os.path.join(lib_dir, "time_dropin_frame"), golden, "300", prefix])
        env = dict(os.environ, KMC_HOST_POOL="0")
        pageable = run_tool([os.path.join(lib_dir, "time_dropin_frame"), golden, "100"], env=env)
        n = pooled["points"]

        def link(us, up_bytes, down_bytes):
            up, down = up_bytes * n / us / 1e3, down_bytes * n / us / 1e3
            return {"up_GBps": round(up, 1), "down_GBps": round(down, 1), "busier_direction_frac_of_peak": round(max(up, down) / PCIE_GBPS_PER_DIRECTION, 3),
                    "busier_direction_frac_of_measured_duplex": round(max(up, down) / PCIE_MEASURED_DUPLEX_GBPS, 3)}

        trace = pooled.get("trace")
        leg = {
            "trace_of_MotionCompensateFrame_f64": trace,
            "trace_is": "hip::FrameTrace (motion_compensation.hpp) + kmc_call_trace (kmc_hip.h): medians of the stages of one call, microseconds; profiles/r05_dropin_trace.json",
            "workload": f"the shipped KITTI frame (tests/golden, {n} points) in HOST containers through libkitti_motion_compensation_lib.so (C++ client tools/time_dropin_frame.cpp); "
                        "T_start = I, T_end = [Rz(0.03) | (1.3, 0.05, -0.02)], requested = stamp_middle; wall clock around the calls, result by value like the reference",
            "pcie_peak_GBps_per_direction": PCIE_GBPS_PER_DIRECTION, "pcie_measured_duplex_GBps_per_direction": PCIE_MEASURED_DUPLEX_GBPS,
            "pcie_peak_is": "PCIe 5.0 x16, 32 GT/s x 16 lanes x 128/130, each way, full duplex (stated, not measured; tools/pcie_probe.py measured ~50 GB/s each way alone on this kind of box)",
            "MotionCompensateFrame_f64": {
                "api": "kmc::MotionCompensateFrame(Frame const&, Time) -> Pointcloud (motion_compensation.hpp:13)",
                "page_locked_containers": {"us_per_frame": pooled["MotionCompensateFrame_f64_us_per_frame"], "us_best_call": pooled["MotionCompensateFrame_f64_us_best_call"],
                                           "Mpts_s": round(n / pooled["MotionCompensateFrame_f64_us_per_frame"], 1), "route": pooled["route"],
                                           "link": link(pooled["MotionCompensateFrame_f64_us_per_frame"], 32, 24),
                                           "link_bytes_per_point": "32 up (x, y, z, stamp; the homogeneous column of ones is not sent) + 24 down (x, y, z; the host fills w)"},
                "pageable_containers": {"us_per_frame": pageable["MotionCompensateFrame_f64_us_per_frame"], "Mpts_s": round(n / pageable["MotionCompensateFrame_f64_us_per_frame"], 1),
                                        "route": pageable["route"]},
            },
            "MotionCompensateFrame_3arg_f64": {
                "api": "kmc::MotionCompensateFrame(Frame const&, Trajectory const&, Time) -- north_star's signature: three knots (scan start / middle / end poses), the frame's own stamps",
                "page_locked_containers": {"us_per_frame": pooled.get("MotionCompensateFrame_3arg_3knots_f64_us_per_frame"),
                                           "link": link(pooled["MotionCompensateFrame_3arg_3knots_f64_us_per_frame"], 40, 32) if pooled.get("MotionCompensateFrame_3arg_3knots_f64_us_per_frame") else None,
                                           "link_bytes_per_point": "40 up (x, y, z, w, stamp) + 32 down (x, y, z, w): this overload moves the homogeneous column"},
                "pageable_containers": {"us_per_frame": pageable.get("MotionCompensateFrame_3arg_3knots_f64_us_per_frame")},
            },
            "MotionCompensateKittiCloud_f32": {
                "api": "kmc::hip::MotionCompensateKittiCloud(float const*, n, T_start, T_end, stamps..., float*) -- the .bin layout, no f64 round trip",
                "page_locked_containers": {"us_per_frame": pooled["MotionCompensateKittiCloud_f32_us_per_frame"], "Mpts_s": round(n / pooled["MotionCompensateKittiCloud_f32_us_per_frame"], 1),
                                           "link": link(pooled["MotionCompensateKittiCloud_f32_us_per_frame"], 16, 16)},
                "pageable_containers": {"us_per_frame": pageable["MotionCompensateKittiCloud_f32_us_per_frame"], "Mpts_s": round(n / pageable["MotionCompensateKittiCloud_f32_us_per_frame"], 1)},
            },
        }
        if orc is not None:  # parity of what the C++ calls returned + the oracle's own time per frame (FAITHFUL, 1 thread: the reference's loop)
            cin = np.fromfile(prefix + ".cloud_in.f64", dtype=np.float64).reshape(4, n).T  # column-major N x 4
            stamps = np.fromfile(prefix + ".stamps.f64", dtype=np.float64)
            cout = np.fromfile(prefix + ".cloud_out.f64", dtype=np.float64).reshape(4, n).T
            kin = np.fromfile(prefix + ".kitti_in.f32", dtype=np.float32).reshape(n, 4)
            kout = np.fromfile(prefix + ".kitti_out.f32", dtype=np.float32).reshape(n, 4)
            c_, s_ = np.cos(pooled["T_end"]["yaw_z"]), np.sin(pooled["T_end"]["yaw_z"])
            A = orc.se3_exp([0.0] * 6)
            B = orc.Affine.from_Rt(np.array([[c_, -s_, 0.0], [s_, c_, 0.0], [0.0, 0.0, 1.0]]), np.array(pooled["T_end"]["t"]))
            t_s, t_m, t_e = pooled["stamp_start"], pooled["stamp_middle"], pooled["stamp_end"]
            t0 = time.perf_counter()
            rc, nbad, want = orc.motion_compensate_frame(np.ascontiguousarray(cin), stamps, t_s, A, t_e, B, t_m)  # the reference's loop, f64, one thread
            oracle_ms = (time.perf_counter() - t0) * 1e3
            assert rc == orc.OK and nbad == 0
            e64 = rel_err(cout[:, :3], want[:, :3])
            ref32 = orc.deskew_xyzi_f32(kin, t_s, A, t_e, B, t_m, mode=orc.FAITHFUL)
            assert ref32["rc"] == orc.OK
            e32 = rel_err(kout[:, :3].astype(np.float64), ref32["xyz_f64"])
            leg["parity"] = {"MotionCompensateFrame_f64_max_rel_err": e64, "f64_bar": 1e-11, "homogeneous_column_is_ones": bool((cout[:, 3] == 1.0).all()),
                             "MotionCompensateKittiCloud_f32_max_rel_err": e32, "f32_bar": 1e-5,
                             "intensity_bit_identical": bool(np.array_equal(kout[:, 3].view(np.uint32), kin[:, 3].view(np.uint32))), "points": int(n)}
            assert e64 <= 1e-11 and e32 <= 1e-5 and leg["parity"]["intensity_bit_identical"] and leg["parity"]["homogeneous_column_is_ones"], leg["parity"]
            leg["oracle_faithful_1_thread_ms_per_frame"] = round(oracle_ms, 2)
            leg["oracle_note"] = "oracle/ FAITHFUL mode (the reference's per-point Log / Exp sequence, f64) on the same frame, one thread like the reference; kind = port (Eigen + OpenCV are not in the image)"
        # ---- kmc::MotionCompensateRun through the reference's CLI on a synthetic KITTI-raw-shaped drive ----
        n_run = 216
        data_dir = os.path.join(tmp, "raw")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_run.py"), data_dir, str(n_run)], capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit("tools/make_synthetic_run.py failed: " + r.stderr[-1000:])
        run = r.stdout.strip().splitlines()[-1]
        cli = os.path.join(lib_dir, "motion_compensate_runs")
        walls, stages, stage_rows = [], None, []
        for rep in range(3):
            shutil.rmtree(os.path.join(run, "velodyne_points", "data_motion_compensated"), ignore_errors=True)
            t0 = time.perf_counter()
            rr = subprocess.run([cli, data_dir + "/", os.path.basename(run)], capture_output=True, text=True, env=unprofiled_env(dict(os.environ, KMC_RUN_TIMING="1")), timeout=900)
            walls.append(time.perf_counter() - t0)
            if rr.returncode != 0:
                raise SystemExit("motion_compensate_runs failed: " + rr.stderr[-2000:])
            stages = [l for l in rr.stderr.splitlines() if l.startswith("kmc run timing")]
            stage_rows.append(_run_stage_numbers(stages))
        pts_run = sum(os.path.getsize(os.path.join(run, "velodyne_points", "data", "%010d.bin" % i)) // 16 for i in range(1, n_run - 1))
        best = min(walls)
        leg["MotionCompensateRun"] = {
            "api": "kmc::MotionCompensateRun(Path) through the reference's CLI (tools/motion_compensate_runs.cpp <- examples/motion_compensate_runs.cpp:9-46); handlers.cpp:41-65",
            "workload": f"tools/make_synthetic_run.py: {n_run} frames of ~117 k points (the shipped frame, randomly thinned), 10 Hz cadence, OXTS along a turning track; files on the box's local disk (page cache warm)",
            "frames_compensated": n_run - 2, "points_compensated": int(pts_run),
            "process_wall_s_best_of_3": round(best, 3), "process_wall_s_all": [round(x, 3) for x in walls],
            "frames_per_s": round((n_run - 2) / best, 1), "Mpts_s": round(pts_run / best / 1e6, 1),
            "stages_ms_per_run": stage_rows,
            "stages_are": "per run, milliseconds: busy time of the pipeline's stages (they overlap) and wall-clock marks since kmc::MotionCompensateRun was entered -- "
                          "the rest of process_wall is the process itself: loading the HIP runtime before main, its teardown after",
            "spread_of_3": round(max(walls) / min(walls) - 1.0, 3),
            "spread_is": "the HIP runtime's own start-up differs from run to run and box to box (stages_ms_per_run.hip_runtime_up_and_first_buffer_page_locked_at: 55-235 ms observed); "
                         "everything after it (run_after_the_hip_runtime_is_up) is this library's and steady",
            "note": "whole process: HIP runtime start-up and page-locking on a helper thread WHILE the text files are parsed, context, reading, one batched GPU round trip per 8 frames, writing",
        }
        if orc is not None:  # one frame the driver wrote, against oracle MakeFrame + the FAITHFUL loop
            from tests import util

            i = n_run // 2
            raw = util.load_velodyne_bin(run, i)
            got = np.fromfile(os.path.join(run, "velodyne_points", "data_motion_compensated", "%010d.bin" % i), dtype=np.float32).reshape(-1, 4)
            vp = os.path.join(run, "velodyne_points")
            ts, tm_, te = (util.load_timestamp(os.path.join(vp, f), i) for f in ("timestamps_start.txt", "timestamps.txt", "timestamps_end.txt"))
            o = [orc.oxts(**util.load_oxts_fields(run, j)) for j in (i - 1, i, i + 1)]
            rc, T_s, T_e = orc.make_frame_poses(o[0], o[1], o[2], ts, te)
            ref = orc.deskew_xyzi_f32(raw, ts, T_s, te, T_e, tm_, mode=orc.FAITHFUL)
            assert rc == orc.OK and ref["rc"] == orc.OK and got.shape == raw.shape
            e = rel_err(got[:, :3].astype(np.float64), ref["xyz_f64"])
            leg["MotionCompensateRun"]["parity"] = {"frame": i, "max_rel_err": e, "bar": 1e-5, "points": int(raw.shape[0]),
                                                    "intensity_bit_identical": bool(np.array_equal(got[:, 3].view(np.uint32), raw[:, 3].view(np.uint32)))}
            assert e <= 1e-5 and leg["MotionCompensateRun"]["parity"]["intensity_bit_identical"], leg["MotionCompensateRun"]["parity"]
        return leg
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _run_stage_numbers(lines):
    """The `kmc run timing` lines of one motion_compensate_runs process (KMC_RUN_TIMING=1) as numbers, milliseconds."""
    import re

    out = {}
    for l in lines:
        m = re.search(r"text files parsed ([0-9.eE+-]+) ms", l)
        if m:
            out["text_files_parsed_at"] = round(float(m.group(1)), 2)
        m = re.search(r"MotionCompensateRun returns ([0-9.eE+-]+) ms after entry; HIP runtime up and first buffer page-locked ([0-9.eE+-]+) ms", l)
        if m:
            out["run_returns_at"] = round(float(m.group(1)), 2)
            out["hip_runtime_up_and_first_buffer_page_locked_at"] = round(float(m.group(2)), 2)
            out["run_after_the_hip_runtime_is_up"] = round(float(m.group(1)) - float(m.group(2)), 2)
        m = re.search(r"context ([0-9.eE+-]+)\s+read ([0-9.eE+-]+)\s+gpu round trip ([0-9.eE+-]+)\s+write ([0-9.eE+-]+)\s+\| wall seconds since the range started: first batch read ([0-9.eE+-]+)\s+all written ([0-9.eE+-]+)", l)
        if m:
            v = [round(float(x) * 1e3, 2) for x in m.groups()]
            out.update({"busy_context": v[0], "busy_read": v[1], "busy_gpu_round_trips": v[2], "busy_write": v[3], "first_batch_read_after_range_start": v[4],
                        "all_written_after_range_start": v[5]})
    return out


LINE_TARGET_BYTES = 4096  # the printed line stays below this ...
LINE_LIMIT_BYTES = 8192   # ... and bench.py fails rather than print a line the driver cannot hold (BENCH_r05: a 21 KB line, parsed = null)
DETAIL_NAME = "bench_detail.json"


def _get(d, *path):
    """d[path[0]][path[1]]... or None: a leg that did not run leaves no hole in the line."""
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _short(s, limit=160):
    s = " ".join(str(s).split())
    return s if len(s) <= limit else s[:limit - 3] + "..."


def compact_line(full, detail_path=None):
    """The ONE line bench.py prints, made from the full record `full` (which goes to bench_detail.json): the contract's keys, `roofline`,
    `cpu_baseline`, and one scalar per leg -- numbers and short identifiers only, no prose.  Raises when the result would exceed
    LINE_LIMIT_BYTES."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: full[k] for k in keep}
    cfg = full["config"]
    line["config"] = {
        "workload": _short(cfg["workload"], 200),
        "points_per_frame": cfg["points_per_frame"], "frames_per_step_per_gpu": cfg["frames_per_step_per_gpu"],
        "points_per_step_per_gpu": cfg["points_per_step_per_gpu"], "parallelism": _short(cfg["parallelism"], 60),
        "kernel": _short(cfg["kernel"], 60), "device": _short(cfg["device"], 40), "arch": _short(cfg["arch"], 40),
    }
    rf = full["roofline"]
    line["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_point", "points_per_launch", "kernel_ms_avg")}
    line["roofline"]["traffic_source"] = rf.get("traffic_kind")
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": _short(cb["sample"], 120)}
        ac = cb.get("all_cores")
        if ac:
            line["cpu_baseline"]["all_cores"] = {"cores": ac["cores"], "faithful_Mpts_s": ac["faithful_Mpts_s"], "hoisted_Mpts_s": ac["hoisted_closed_form_Mpts_s"]}
    if "parity_spot_check" in full:
        line["parity_spot_check"] = {"max_rel_err": float("%.3g" % full["parity_spot_check"]["max_rel_err"]), "bar": full["parity_spot_check"]["bar"]}
    rk = full.get("ranks")
    if rk:
        line["ranks"] = {"backend": _short(rk["collective_backend"], 24), "rccl_world_size": rk["rccl_world_size"], "world_size": rk["world_size"],
                         "distinct_devices": rk["distinct_devices"], "host_cpus_allowed_min": rk.get("host_cpus_allowed_min")}
    pr = full.get("per_rank")
    if pr:
        line["per_rank"] = {"Mpts_s_min": pr["Mpts_s_min"], "Mpts_s_max": pr["Mpts_s_max"], "slowest_rank": pr["slowest_rank"]}
        if "configs3_Mpts_s_min" in pr:
            line["per_rank"]["configs3_Mpts_s_min"], line["per_rank"]["configs3_Mpts_s_max"] = pr["configs3_Mpts_s_min"], pr["configs3_Mpts_s_max"]
    c3 = full.get("configs3")
    if c3:
        line["configs3"] = {"value": c3["value"], "unit": c3["unit"], "points_per_frame": c3["points_per_frame"], "frames_total": c3["frames_total"],
                            "timed_frames_per_rank": c3["timed_frames_per_rank"], "ms_per_frame": c3["ms_per_frame"], "frac_rank0": c3["frac_of_peak_rank0"],
                            "parity_max_rel_err": (float("%.3g" % c3["parity_first_last_frame_per_rank"]["max_rel_err"]) if c3.get("parity_first_last_frame_per_rank") else None)}
    su = full.get("sustained")
    if su:
        line["sustained"] = {k: su[k] for k in ("seconds", "launches", "Mpts_s_mean", "Mpts_s_min", "Mpts_s_max", "frac_mean")}
    # one scalar per leg: fractions of the 8 TB/s peak unless the name says otherwise
    legs = {}
    for name, path in LEG_SCALARS:
        v = _get(full, *path)
        if isinstance(v, bool) or isinstance(v, int):
            legs[name] = v
        elif isinstance(v, float):
            legs[name] = float("%.4g" % v) if abs(v) < 1e-3 else v
    if legs:
        line["legs"] = legs
        if legs.get("ceiling_f32_copy_frac"):  # the headline against the same access pattern without arithmetic, measured on this box in this run
            line["roofline"]["frac_of_same_run_copy_ceiling"] = round(rf["frac"] / legs["ceiling_f32_copy_frac"], 4)
    if "secondary_legs_error" in full:
        line["legs_error"] = _short(full["secondary_legs_error"], 200)
    line["peak_device_GiB_per_rank"] = full.get("peak_device_GiB_per_rank")
    line["detail"] = detail_path
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_TARGET_BYTES:  # shed the optional parts, legs last
        for k in ("sustained", "per_rank", "legs"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= LINE_TARGET_BYTES:
                break
    if len(text) > LINE_LIMIT_BYTES:
        raise RuntimeError(f"bench line of {len(text)} bytes exceeds {LINE_LIMIT_BYTES}: the driver would not hold it")
    return text


# name in the line's "legs" object -> where the number lives in the full record (bench_detail.json)
LEG_SCALARS = (
    ("ceiling_f32_copy_frac", ("ceilings", "f32_one_stream_in_one_out", "frac_of_peak")),
    ("ceiling_f64_9stream_frac", ("ceilings", "f64_nine_column_streams", "copy_cols9", "frac_of_peak")),
    ("c1_per_call_frac", ("configs1_literal", "in_order", "frac")),
    ("c1_per_call_direct_queue_frac", ("configs1_literal", "in_order_direct_queue", "frac")),
    ("c1_gathered_frac", ("configs1_literal", "gathered_calls", "frac")),
    ("c1_list_frac", ("configs1_literal", "list_one_launch", "frac")),
    ("c1_batch_frac", ("configs1_literal", "batch_packed", "frac")),
    ("c1_nknot3_per_call_frac", ("configs1_literal", "in_order_nknot3", "frac")),
    ("c1_nknot3_direct_queue_frac", ("configs1_literal", "in_order_nknot3_direct_queue", "frac")),
    ("c1_parity_max_rel_err", ("configs1_literal", "parity", "max_rel_err")),
    ("c2_batched_frac", ("configs2_drive", "frac")),
    ("c2_per_call_us", ("configs2_drive", "frame_by_frame_from_c", "per_call", "us_per_frame")),
    ("c2_per_call_frac", ("configs2_drive", "frame_by_frame_from_c", "per_call", "frac")),
    ("c2_per_call_direct_queue_us", ("configs2_drive", "frame_by_frame_from_c", "per_call_direct_queue", "us_per_frame")),
    ("c2_per_call_direct_queue_frac", ("configs2_drive", "frame_by_frame_from_c", "per_call_direct_queue", "frac")),
    ("c2_gathered_frac", ("configs2_drive", "frame_by_frame_from_c", "per_call_gathered", "frac")),
    ("c2_list_frac", ("configs2_drive", "frame_by_frame_from_c", "list_one_launch", "frac")),
    ("c2_list_equals_per_call_bitwise", ("configs2_drive", "frame_by_frame_from_c", "list_equals_per_call_bitwise")),
    ("c2_parity_max_rel_err", ("configs2_drive", "parity", "max_rel_err")),
    ("nknot3_frac", ("nknot3", "frac")),
    ("nknot3_parity_max_rel_err", ("nknot3", "parity", "max_rel_err")),
    ("f64cols_frac", ("f64cols", "frac")),
    ("f64cols_frac_of_adjacent_9_stream_ceiling", ("f64cols", "frac_of_adjacent_9_stream_ceiling")),
    ("f64cols_ones_frac", ("f64cols", "homogeneous_column_known_to_be_ones", "frac")),
    ("f64cols_parity_max_rel_err", ("f64cols", "parity", "max_rel_err")),
    ("dropin_frame_f64_us", ("dropin_cpp", "MotionCompensateFrame_f64", "page_locked_containers", "us_per_frame")),
    ("dropin_frame_f64_pageable_us", ("dropin_cpp", "MotionCompensateFrame_f64", "pageable_containers", "us_per_frame")),
    ("dropin_frame_3arg_f64_us", ("dropin_cpp", "MotionCompensateFrame_3arg_f64", "page_locked_containers", "us_per_frame")),
    ("dropin_kitti_cloud_f32_us", ("dropin_cpp", "MotionCompensateKittiCloud_f32", "page_locked_containers", "us_per_frame")),
    ("dropin_f64_parity_max_rel_err", ("dropin_cpp", "parity", "MotionCompensateFrame_f64_max_rel_err")),
    ("dropin_oracle_1_thread_ms_per_frame", ("dropin_cpp", "oracle_faithful_1_thread_ms_per_frame")),
    ("run_frames_per_s", ("dropin_cpp", "MotionCompensateRun", "frames_per_s")),
    ("run_wall_s_best_of_3", ("dropin_cpp", "MotionCompensateRun", "process_wall_s_best_of_3")),
    ("run_parity_max_rel_err", ("dropin_cpp", "MotionCompensateRun", "parity", "max_rel_err")),
    ("sharded_cpp_world", ("sharded_cpp", "world")),
    ("sharded_cpp_Mpts_s", ("sharded_cpp", "reduced", "Mpts_s")),
)


def write_detail(full):
    """The full record (every leg's numbers, notes and workload descriptions) as a file: $KMC_BENCH_DETAIL, else gpurun_out/ under the
    repo (what gpurun brings back), else the repo root.  Returns the path relative to the repo, or None when nothing was writable."""
    cands = [os.environ["KMC_BENCH_DETAIL"]] if os.environ.get("KMC_BENCH_DETAIL") else []
    cands += [os.path.join(ROOT, "gpurun_out", DETAIL_NAME), os.path.join(ROOT, DETAIL_NAME)]
    for path in cands:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as fh:
                json.dump(full, fh, indent=1)
                fh.write("\n")
            return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
        except OSError:
            continue
    return None


def relaunch_under_launcher(n, torch):
    """-> exit code.  The driver's N > 1 launch line, built here when bench.py was started bare with --gpus N: one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 at a free port.  KMC_BENCH_LAUNCH_DRY=1 prints the line as JSON instead of running it
    (the CPU test of this path); KMC_BENCH_DEVICE (the one-GPU test knob) waives the device-count check."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    if os.environ.get("KMC_BENCH_LAUNCH_DRY") == "1":
        print(json.dumps({"relaunch": cmd}), flush=True)
        return 0
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and "KMC_BENCH_DEVICE" not in os.environ:
        sys.stderr.write(f"bench.py: --gpus {n} but this box shows {have} GPU(s): not measuring one GPU under an N-GPU label.  The launch line would be:\n  {' '.join(cmd)}\n")
        return 2
    sys.stderr.write(f"bench.py: --gpus {n} without a launcher (WORLD_SIZE unset): re-executing as\n  {' '.join(cmd)}\n")
    return subprocess.call(cmd)


def main():
    global POINTS_PER_FRAME
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames-per-step", type=int, default=256)
    ap.add_argument("--points-per-frame", type=int, default=POINTS_PER_FRAME,
                    help="1000000 = BASELINE.json configs[1] (the default, the headline); 10000000 with --frames-per-step 24 "
                         "--yaw-per-frame 0.03 = the per-GPU share of configs[3]'s 10 M-point-per-frame stream")
    ap.add_argument("--yaw-per-frame", type=float, default=0.0)
    ap.add_argument("--live-traffic", action="store_true", help="(default at N = 1; kept for older command lines)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child runs of the same workload, ~1 minute, "
                         "N = 1 only); the per-point figure of the committed PMC passes (profiles/pmc_traffic.json) is scaled instead")
    ap.add_argument("--no-configs3", action="store_true", help="skip the configs[3] leg (10 M-point frames, frame-sharded)")
    ap.add_argument("--configs3-frames", type=int, default=960, help="timed frames per rank of the configs[3] leg")
    ap.add_argument("--configs3-frames-per-launch", type=int, default=C3_FRAMES_PER_LAUNCH)
    ap.add_argument("--sustained-seconds", type=float, default=10.0, help="N = 1: repeat the headline launch for this long after the timed region and report the rate per ~0.1 s window (0 = skip)")
    ap.add_argument("--legs-only", action="store_true", help="run ONLY the secondary legs and print their JSON (for a profiler pass whose kernel rows then belong to the legs alone)")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary legs (configs1_literal, configs2_drive, nknot3, f64cols; N = 1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=32)
    ap.add_argument("--rotate", type=int, default=1, help="number of in/out buffer pairs cycled through by the steps")
    args = ap.parse_args()
    headline = args.points_per_frame == POINTS_PER_FRAME and args.yaw_per_frame == 0.0
    POINTS_PER_FRAME = args.points_per_frame

    import torch

    from kitti_motion_compensation_amd import capi, sharding

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: ONE process would measure one GPU and label it N.  Re-execute
        # under the contract's launcher instead -- one process per GPU -- or refuse when the box does not have N GPUs.
        raise SystemExit(relaunch_under_launcher(args.gpus, torch))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} "
                         f"--master-addr 127.0.0.1 --master-port <port> bench.py --gpus {args.gpus} ... (or run `python bench.py --gpus {args.gpus}` without WORLD_SIZE set: it re-executes itself that way)")
    dist = None
    if world > 1 or os.environ.get("KMC_BENCH_FORCE_DIST") == "1":  # the env knob exercises the RCCL path on one GPU
        import torch.distributed as dist  # backend "nccl" IS RCCL on ROCm

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        # Test knobs (tests/test_z_bench_contract.py runs the whole N = 2 flow on a one-GPU box): KMC_BENCH_BACKEND=gloo reduces the
        # counters over gloo on CPU tensors, KMC_BENCH_DEVICE=<id> puts every rank on that device.  Unset = the contract: RCCL, one
        # rank per GPU.
        backend = os.environ.get("KMC_BENCH_BACKEND", "nccl")
        if "KMC_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["KMC_BENCH_DEVICE"])
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            # eager initialisation (device_id) + one collective: a group RCCL cannot form -- two ranks on one GPU, a rank without a device -- is
            # reported HERE, as what it is, not minutes later as a hang or a wrong number (exit code 3, the reason on stderr)
            try:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
                dist.barrier()
            except Exception as e:
                print(f"bench: RCCL could not form a group of {world} rank(s) (this is rank {rank} on device {local_rank}): {type(e).__name__}: {str(e)[:1500]}", file=sys.stderr, flush=True)
                os._exit(3)
        else:
            dist.init_process_group(backend=backend)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the deskew path has no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    reduce_dev = dev if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")

    F = args.frames_per_step
    n = F * POINTS_PER_FRAME
    # this rank's host side on its GPU's NUMA node (what numactl does for a deployment; the C++ clients the legs start inherit the mask):
    # on the two-socket MI355X boxes a process that lands on the other socket pays the inter-socket hop on every argument block, read-back
    # and in-place access -- 2.2 instead of 1.9 us per direct-queue call, 104-125 instead of 92 us per in-place KITTI frame
    try:
        capi.bind_thread_near_device(local_rank)
    except Exception as e:  # a placement hint: the bench runs wherever it is
        print(f"bench: not bound to the GPU's NUMA node ({e})", file=sys.stderr)
    ctx = capi.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    info = ctx.device_info()
    if args.legs_only:
        assert world == 1 and dist is None, "--legs-only is an N = 1 profiling aid"
        print(json.dumps(run_secondary_legs(capi, torch, ctx, dev, check=not args.no_cpu_baseline)), flush=True)
        ctx.close()
        return

    # ---- workload: generated on the device (identical generator on the host for the oracle), resident in HBM ----
    R = max(1, args.rotate)
    d_ins = [torch.empty((n, 4), dtype=torch.float32, device=dev) for _ in range(R)]
    d_outs = [torch.empty_like(d_ins[0]) for _ in range(R)]
    for f in range(F):
        ctx.synth_points(d_ins[0][f * POINTS_PER_FRAME:(f + 1) * POINTS_PER_FRAME], POINTS_PER_FRAME, SEED + f + rank * F)
    for r in range(1, R):
        d_ins[r].copy_(d_ins[0])
    d_in, d_out = d_ins[0], d_outs[0]
    state = {"k": 0}
    work = make_workload(capi, F, rank, args.yaw_per_frame)
    params = capi.params_array([w[0] for w in work])
    offsets = np.arange(F + 1, dtype=np.uint64) * POINTS_PER_FRAME
    torch.cuda.synchronize()

    def step():
        k = state["k"] % R
        state["k"] += 1
        ctx.deskew_batch_f32(d_ins[k], d_outs[k], offsets, params, None)

    # Setup, not measurement: the first call sizes the table ring (allocations + a stream sync) and an idle MI355X needs ~8
    # launches (~10 ms) to ramp its clocks (measured: 1.45 -> 1.24 ms per step).  A fixed spin-up precedes the caller's W
    # warm-up steps so that a small W does not put allocation or ramp time into the K timed steps.
    for _ in range(SPIN_UP_STEPS):
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides, HIP events on the launch stream ----
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_begin = time.perf_counter()
    ctx.timer_begin()
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_end()  # hipEventRecord + hipEventSynchronize on the same stream
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    wall = time.perf_counter() - t_begin

    # ---- sustained leg (N = 1): the headline launch repeated for a wall-clock budget, in windows, so that the driver's run holds
    # more than a few milliseconds of evidence: mean / min / max / std over the windows ----
    sustained = None
    if world == 1 and dist is None and headline and args.sustained_seconds > 0:
        window = max(1, int(round(0.1 / max(ev_ms / args.steps * 1e-3, 1e-6))))  # ~0.1 s of launches per window
        rates, t_stop = [], time.perf_counter() + args.sustained_seconds
        while time.perf_counter() < t_stop:
            ctx.timer_begin()
            for _ in range(window):
                step()
            rates.append(n * window / (ctx.timer_end() * 1e-3) / 1e6)
        r = np.array(rates)
        sustained = {"seconds": args.sustained_seconds, "windows": int(r.size), "launches_per_window": window, "launches": int(r.size) * window,
                     "Mpts_s_mean": round(float(r.mean()), 1), "Mpts_s_min": round(float(r.min()), 1), "Mpts_s_max": round(float(r.max()), 1),
                     "Mpts_s_std": round(float(r.std()), 1), "GBps_mean": round(float(r.mean()) * BYTES_PER_POINT / 1e3, 1),
                     "frac_mean": round(float(r.mean()) * BYTES_PER_POINT / 1e3 / HBM_PEAK_GBPS, 4),
                     "note": "the headline's step (one 256 M-point batched launch) back to back, HIP events per window on the launch stream"}

    # the sample the cpu_baseline leg needs, before the configs[3] leg reuses the buffers
    cpu_sample = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        k_cpu = min(args.cpu_sample_frames, F, max(1, 32_000_000 // POINTS_PER_FRAME))  # ~10-30 s of single-thread oracle work
        cpu_sample = (k_cpu, d_in[:k_cpu * POINTS_PER_FRAME].cpu().numpy(), d_outs[(state["k"] - 1) % R][:POINTS_PER_FRAME].cpu().numpy())

    # ---- configs[3] leg: the 10 M-point-per-frame stream, rank r on frames frame_range(8000, r, N) ----
    c3 = None
    if not args.no_configs3:
        del d_in, d_out, d_ins, d_outs  # the headline's 8 GB; the leg allocates its own buffers
        torch.cuda.empty_cache()
        c3 = run_configs3(capi, sharding, torch, ctx, dist, rank, world, dev, args.configs3_frames, args.configs3_frames_per_launch,
                          check=not args.no_cpu_baseline)

    # ---- secondary legs (N = 1: the BENCH line), each with its own timed region and oracle spot check ----
    legs = None
    if world == 1 and dist is None and headline and not args.no_legs:
        torch.cuda.empty_cache()
        try:  # a secondary leg that cannot run (a tool that did not build, a box without rocprof...) must not take the headline line with it
            legs = run_secondary_legs(capi, torch, ctx, dev, check=not args.no_cpu_baseline)
        except AssertionError:
            raise  # a parity check that fails is not a leg that could not run
        except (Exception, SystemExit) as e:
            legs = {"secondary_legs_error": f"{type(e).__name__}: {e}"[:2000]}

    # the job's ONLY data collective (RCCL when N > 1): one all_gather of every rank's counters; SUM of points, MAX of times.
    # (The four dist.barrier() calls around the two timed regions are collectives too -- one-element all-reduces on the nccl
    # backend -- and are what the timing contract asks for.)
    props = torch.cuda.get_device_properties(dev)
    pci_code = (int(getattr(props, "pci_domain_id", 0)) << 16) | ((int(getattr(props, "pci_bus_id", 0)) & 0xFF) << 8) | (int(getattr(props, "pci_device_id", 0)) & 0xFF)
    cpus_now = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []  # after bind_thread_near_device: this rank's host side
    sums, maxes, rows = sharding.reduce_counters(
        dist, reduce_dev, [float(n * args.steps), c3["points"] if c3 else 0.0],
        [wall, ev_ms * 1e-3, c3["wall"] if c3 else 0.0, c3["ev_s"] if c3 else 0.0, c3["parity_err"] if c3 else 0.0,
         torch.cuda.max_memory_allocated(dev) / 2**30, float(dev.index or 0), float(pci_code), float(len(cpus_now)), float(min(cpus_now) if cpus_now else -1)], with_rows=True)
    pts_total, t_max = sums[0], maxes[0]

    if rank == 0:
        step_ms = ev_ms / args.steps  # HIP-event time of the timed region / steps on this rank's launch stream: the kernel plus
        # whatever the stream waited for between two launches (per-step table upload, host-side table build)
        achieved = BYTES_PER_POINT * n / (step_ms * 1e-3) / 1e9
        # HBM bytes per launch: measured in THIS run at N = 1 (two rocprofv3 --pmc child runs of the same workload, FETCH_SIZE
        # and WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md's HBM section prescribes with the factors
        # profiles/pmc_traffic.json derived from the copy kernel); if the box does not allow counter collection, or N > 1, the
        # per-point figure of the committed passes is scaled to this launch size -- traffic_source says which.
        traffic, traffic_source, traffic_kind = None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fjson:
                tj = json.load(fjson)
            traffic = round(tj["hbm_bytes_per_point"] * n)
            traffic_source = "profiles/pmc_traffic.json (committed rocprofv3 --pmc passes of this command), per point x points per launch"
            traffic_kind = "committed_pmc_per_point_scaled"
            if world == 1 and dist is None and not args.no_live_traffic:
                live = live_traffic(F, POINTS_PER_FRAME, args.yaw_per_frame, tj)
                if live:
                    traffic = round(live)
                    traffic_source = "measured in this run: rocprofv3 --pmc child runs of this invocation (FETCH_SIZE, WRITE_SIZE passes)"
                    traffic_kind = "live_pmc_this_run"
                else:
                    traffic_source += " -- counter collection was not possible on this box"
                    traffic_kind += "_live_collection_not_possible"
        out = {
            "metric": "M points/sec deskewed",
            "value": round(pts_total / t_max / 1e6, 1),
            "unit": "Mpts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(t_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"configs[1]: synthetic 1M-point frames, straight-line constant-velocity trajectory; " if headline else
                             f"NOT the headline -- configs[3]-shaped stream: synthetic {POINTS_PER_FRAME}-point frames, yaw {args.yaw_per_frame} rad per frame; ")
                            + f"{F} distinct frames per step in one batched launch (per GPU), device-resident",
                "points_per_frame": POINTS_PER_FRAME, "frames_per_step_per_gpu": F, "points_per_step_per_gpu": n,
                "parallelism": f"frame-sharded x{world} (no data-path collective; one all_gather of the counters + the contract's four barriers)",
                "kernel": "kmc_dev::deskew_batch_f32<series3, ppt=1, nt loads + nt|sc1 stores, block=64>, one 64-point tile (one wave) per workgroup", "device": info["name"], "arch": info["arch"],
                "host_threads": "every rank (and the C++ clients it starts) on the CPUs of its GPU's NUMA node: kmc_hip_bind_thread_near_device",
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source, "traffic_kind": traffic_kind,
                "bytes_per_point": BYTES_PER_POINT, "points_per_launch": n, "kernel_ms_avg": round(step_ms, 4),
                "kernel_ms_avg_is": "HIP-event time of the timed region on the launch stream / steps (one launch per step)",
            },
        }
        # who measured: the collective backend and its world size as the process group reports them, and every rank's device (ordinal +
        # PCI address) out of the same all_gather -- the record itself shows that N ranks sat on N distinct GPUs
        nsum = 2  # width of the SUM block in a gathered row
        devs = [{"rank": r, "device_ordinal": int(row[nsum + 6]), "pci": "%04x:%02x:%02x" % (int(row[nsum + 7]) >> 16, (int(row[nsum + 7]) >> 8) & 0xFF, int(row[nsum + 7]) & 0xFF),
                 "host_cpus_allowed": int(row[nsum + 8]), "first_host_cpu": int(row[nsum + 9])}  # the rank's CPU set after it was bound to its GPU's NUMA node
                for r, row in enumerate(rows)]
        backend_name = dist.get_backend() if dist else None
        out["ranks"] = {
            "collective_backend": ("nccl (= RCCL on ROCm)" if backend_name == "nccl" else backend_name) if dist else "none (single process)",
            "rccl_world_size": dist.get_world_size() if (dist and backend_name == "nccl") else None,
            "world_size": dist.get_world_size() if dist else 1,
            "devices": devs, "distinct_devices": len({d["pci"] for d in devs}),
            "host_cpus_allowed_min": min(d["host_cpus_allowed"] for d in devs),
        }
        out["peak_device_GiB_per_rank"] = round(maxes[5], 2)  # torch's allocator high-water mark, MAX over the ranks (8 ranks of the default run: ~8 x 15.4 GiB of the 288)
        if world > 1:  # a straggler is invisible in SUM / MAX: every rank's own rate (its points / its own wall time of the timed region)
            rates = [r[0] / r[2] / 1e6 for r in rows]  # row = [points, c3 points | wall, ev_s, c3 wall, c3 ev_s, c3 parity]
            out["per_rank"] = {"Mpts_s_min": round(min(rates), 1), "Mpts_s_max": round(max(rates), 1), "Mpts_s": [round(v, 1) for v in rates],
                               "slowest_rank": int(np.argmin(rates)), "note": "each rank's points / its own wall time between the two barriers"}
            if c3:
                rates3 = [r[1] / r[4] / 1e6 for r in rows]  # c3 points / c3 wall
                out["per_rank"]["configs3_Mpts_s_min"] = round(min(rates3), 1)
                out["per_rank"]["configs3_Mpts_s_max"] = round(max(rates3), 1)
                out["per_rank"]["configs3_Mpts_s"] = [round(v, 1) for v in rates3]
        if c3:
            t3 = maxes[2]
            out["configs3"] = {
                "workload": f"configs[3]: synthetic {C3_POINTS_PER_FRAME}-point frames x {C3_FRAMES_TOTAL}, |phi| ~ {C3_YAW_PER_FRAME} rad per frame, "
                            f"frame-sharded: rank r owns frames sharding.frame_range({C3_FRAMES_TOTAL}, r, {world}); timed = the first "
                            f"{c3['timed_frames']} frames of every rank's range, {c3['frames_per_launch']} frames per batched launch over {c3['groups']} rotating "
                            "buffer groups of distinct device-generated frames; no data-path collective",
                "frames_total": C3_FRAMES_TOTAL, "points_per_frame": C3_POINTS_PER_FRAME,
                "rank_frame_ranges": [list(sharding.frame_range(C3_FRAMES_TOTAL, r, world)) for r in range(world)],
                "timed_frames_per_rank": c3["timed_frames"], "frames_per_launch": c3["frames_per_launch"],
                "value": round(sums[1] / t3 / 1e6, 1), "unit": "Mpts/s",
                "ms_per_frame": round(t3 / c3["timed_frames"] * 1e3, 4),
                "achieved_GBps_rank0": round(BYTES_PER_POINT * c3["points"] / c3["ev_s"] / 1e9, 1),
                "frac_of_peak_rank0": round(BYTES_PER_POINT * c3["points"] / c3["ev_s"] / 1e9 / HBM_PEAK_GBPS, 4),
                "parity_first_last_frame_per_rank": ({"max_rel_err": maxes[4], "bar": 1e-5, "oracle": "FAITHFUL, whole 10 M-point frames"}
                                                     if not args.no_cpu_baseline else None),
            }
        if sustained:
            out["sustained"] = sustained
        if legs:
            out.update(legs)
        out["config"]["kitti_root"] = ("present (not used by the synthetic headline; tests/test_configs_at_size.py and tools/measure_configs.py run the real drives)"
                                        if os.environ.get("KITTI_ROOT") and os.path.isdir(os.environ["KITTI_ROOT"]) else "absent: every workload is a synthetic twin")
        if cpu_sample is not None:
            k_cpu, sample, gpu_frame0 = cpu_sample
            out["cpu_baseline"], out["parity_spot_check"] = cpu_baseline(sample, [(w[1], w[2]) for w in work], k_cpu, gpu_frame0)
        # the full record goes to a file; the ONE stdout line is compact (numbers and identifiers, < 4 KB): a line the driver cannot hold is
        # an unmeasured round (BENCH_r05)
        line = compact_line(out, write_detail(out))
    # The line is the LAST thing on stdout -- of the whole job: native libraries write there too (RCCL's "Librccl path ..." sits in the C
    # library's buffer of EVERY rank until that process flushes or ends, and the launcher merges the ranks' streams).  So: every rank flushes
    # what it has buffered, every rank but 0 then closes its stdout for good, a barrier, rank 0 writes the line and closes its own.  Whatever
    # a teardown prints afterwards goes nowhere.  (The process ends the ordinary way: a profiler attached to it -- rocprofv3's counter passes
    # of live_traffic() among them -- writes its results from an exit handler.)
    import ctypes

    def seal_stdout():
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        null_fd = os.open(os.devnull, os.O_WRONLY)
        os.dup2(null_fd, 1)
        os.close(null_fd)

    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if rank != 0:
        seal_stdout()
    if dist:
        dist.barrier()
    if rank == 0:
        os.write(1, (line + "\n").encode())
        seal_stdout()
    if dist:
        # every rank is past the last barrier and the line is out: nothing is left that needs the group.  Its teardown (communicator
        # destruction, the watchdog's 10-minute patience) is where multi-rank jobs are known to linger; a job that has printed its result
        # does not wait for it.  (The single-process run below ends the ordinary way -- profilers write their output from exit handlers.)
        sys.stderr.flush()
        os._exit(0)
    ctx.close()


if __name__ == "__main__":
    main()
