"""Compiler evidence, guarded: the register / scratch / occupancy figures DESIGN.md quotes are not prose -- this
test re-runs hipcc's kernel-resource-usage analysis on the shipped translation units (cross-compiling gfx950 needs no GPU) and
holds every kernel to its budget, so that a compiler update or an innocent edit cannot silently cost a wave slot:

  * every shipped f32 deskew kernel: no scratch, <= 64 VGPRs, 8 waves per SIMD -- these kernels need every wave slot (occupancy
    sweep, profiles/r02_tune_occ_ppt.csv);
  * the N-knot kernels use no LDS since round 3 (records travel through scalar loads);
  * the f64 Eigen-layout kernels: no scratch, 4 waves per SIMD on purpose (nine streams per wave);
  * nothing anywhere spills to scratch;
  * round 4: the product compiles ONE geometry (one 64-point tile per one-wave workgroup, no tile loops, one cache policy) -- at most
    93 kernel instantiations (round 3: 170; round 5 added the list kernel's two larger argument blocks), VERDICT r03 #7.
The committed summary profiles/r06_resource_usage.txt must list the same kernels (it is regenerated with
`make -C kitti_motion_compensation_amd/csrc resource-usage 2>&1 | python tools/summarize_resource_usage.py`)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import summarize_resource_usage as sru  # noqa: E402


@pytest.fixture(scope="module")
def usage():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "kitti_motion_compensation_amd", "csrc"), "resource-usage"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = sru.parse(r.stdout + r.stderr)
    names = sru.demangle([x["mangled"] for x in rows])
    out = {}
    for x, n in zip(rows, names):
        out[n.replace("void ", "").replace("kmc_dev::", "").split("(")[0]] = {k: int(v) for k, v in x.items() if k != "mangled"}
    assert 60 < len(out) <= 94, len(out)  # half of round 3's 170: no tile-loop twins, no points-per-lane / policy / block-size variants (round 5: +8, the list kernel's 64- and 256-frame argument blocks)
    return out


def test_nothing_spills_to_scratch(usage):
    bad = {k: v["scratch"] for k, v in usage.items() if v["scratch"] != 0}
    assert not bad, bad


def test_f32_kernels_keep_eight_waves_per_simd(usage):
    hot = {k: v for k, v in usage.items() if k.startswith(("deskew_frame_f32<", "deskew_batch_f32<", "deskew_list_f32<", "deskew_traj_f32<", "deskew_traj_batch_f32<",
                                                            "deskew_frame_streamed_f32<"))}
    assert len(hot) == 4 + 16 + 12 + 16 + 8 + 4, sorted(hot)  # (12: the frame-list kernel in its three argument-block capacities; the device-table form went in round 6)
    for k, v in hot.items():
        assert v["occupancy"] == 8 and v["vgprs"] <= 64 and v["agprs"] == 0, (k, v)
    # the headline kernel by name: <series3, no index output, device tables>
    bench = usage["deskew_batch_f32<0, false, false>"]
    assert bench["vgprs"] <= 48 and bench["sgprs"] <= 56 and bench["sgpr_spills"] == 0 and bench["lds"] == 1024, bench
    # kernel-argument tables: same body, no extra registers
    inline = usage["deskew_batch_f32<0, false, true>"]
    assert inline["vgprs"] <= bench["vgprs"] + 2 and inline["occupancy"] == 8, inline
    # single frames and lists of frames share one tile body
    for k in ("deskew_frame_f32<0>", "deskew_list_f32<0, 16>", "deskew_list_f32<0, 64>", "deskew_list_f32<0, 256>"):
        assert usage[k]["vgprs"] <= 40 and usage[k]["sgpr_spills"] == 0 and usage[k]["lds"] == 0, (k, usage[k])
    # the batched N-knot kernel: without a tile loop (no loop-carried copies of its twelve arguments) it does not live on spills
    for idx in ("false", "true"):
        assert usage[f"deskew_traj_batch_f32<0, {idx}>"]["sgpr_spills"] <= 8


def test_nknot_kernels_use_no_lds(usage):
    for k, v in usage.items():
        if k.startswith(("deskew_traj_f32<", "deskew_traj_batch_f32<")):
            assert v["lds"] == 0, (k, v)


def test_f64_kernels_as_documented(usage):
    for k in ("deskew_f64cols<false>", "deskew_f64cols<true>", "deskew_traj_f64cols<false>", "deskew_traj_f64cols<true>"):
        assert usage[k]["occupancy"] == 4 and usage[k]["scratch"] == 0, (k, usage[k])
    assert usage["deskew_f64cols<false>"]["vgprs"] <= 64 and usage["deskew_f64cols<false>"]["sgpr_spills"] == 0


def test_committed_summary_lists_the_same_kernels(usage):
    path = os.path.join(ROOT, "profiles", "r06_resource_usage.txt")
    with open(path) as f:
        committed = {ln.split(" | ")[0]: ln.strip().split(" | ")[1:] for ln in f if ln.strip() and not ln.startswith("#")}
    assert set(committed) == set(usage), (sorted(set(committed) ^ set(usage))[:5], "regenerate profiles/r06_resource_usage.txt")
    stale = [k for k, v in usage.items() if [str(v[x]) for x in ("vgprs", "agprs", "sgprs", "sgpr_spills", "scratch", "occupancy", "lds")] != committed[k]]
    assert not stale, (stale[:5], "the committed figures differ from what hipcc produces now: regenerate profiles/r06_resource_usage.txt")
