"""north_star's multi-GPU shape driven from C++ (tools/run_sharded.hip): contiguous frame ranges per rank, no point data between devices,
and ONE native RCCL reduction of the counters -- ncclCommInitAll + ncclAllReduce(sum of points) + ncclAllReduce(max of seconds) -- whose
result must equal plain host arithmetic on the ranks' own counters.  World 1 on the box's GPU goes through RCCL; world 2
on ONE GPU ("0,0": two ranks, two device contexts, one device) exercises the sharding and the barrier, and documents what RCCL does with a
duplicate device: it refuses the communicator, the tool says so and reduces on the host (an 8-GPU node passes eight distinct ids)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "kitti_motion_compensation_amd", "lib", "run_sharded")


def _run(devices, frames, points, per_launch):
    if not os.path.exists(TOOL):
        pytest.skip("tools/run_sharded was not built (tools/Makefile is best effort)")
    r = subprocess.run([TOOL, devices, str(frames), str(points), str(per_launch)], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_world_1_reduces_its_counters_through_rccl():
    d = _run("0", 24, 200_000, 8)
    assert d["world"] == 1 and d["ranks"][0]["frames"] == [0, 24]
    assert d["reduced_by"].startswith("rccl: ncclCommInitAll + ncclAllReduce"), d
    assert d["reduction_agrees_with_host_arithmetic"] is True
    assert d["reduced"]["points"] == 24 * 200_000 and d["reduced"]["seconds_max"] == d["ranks"][0]["seconds"]


@pytest.mark.gpu
def test_world_2_on_one_device_shards_contiguously_and_says_who_reduced():
    d = _run("0,0", 25, 150_000, 4)
    assert d["world"] == 2
    (a0, a1), (b0, b1) = d["ranks"][0]["frames"], d["ranks"][1]["frames"]
    assert a0 == 0 and a1 == b0 and b1 == 25 and abs((a1 - a0) - (b1 - b0)) <= 1  # contiguous, balanced on points (equal frames: on count)
    assert d["reduced"]["points"] == 25 * 150_000
    assert d["reduced"]["seconds_max"] == max(r["seconds"] for r in d["ranks"])
    assert d["reduction_agrees_with_host_arithmetic"] is True
    # one GPU twice: RCCL either builds the communicator or refuses the duplicate -- the line must say which path reduced
    assert d["reduced_by"].startswith("rccl") or ("host" in d["reduced_by"] and "ncclCommInitAll refused" in d["rccl_note"]), d


@pytest.mark.gpu
def test_all_means_one_rank_per_visible_device():
    """`run_sharded all` (what bench.py's leg runs): a rank per device the process can see.  One device on the test box; on a multi-GPU node
    the same command forms a real RCCL group -- world == visible devices, every device once, reduced by RCCL (which no box this repository
    has run on could show: ncclAllReduce across more than one rank has never executed, README.md says so)."""
    import torch

    visible = torch.cuda.device_count()
    d = _run("all", max(24, 8 * visible), 200_000, 8)
    assert d["world"] == visible and d["devices"] == list(range(visible))
    assert d["reduced_by"].startswith("rccl: ncclCommInitAll + ncclAllReduce"), d
    assert d["reduction_agrees_with_host_arithmetic"] is True
