"""The N > 1 path on CPU: contiguous frame-range sharding and the counter reduction, world_size 2 over gloo
(the GPU box uses the same code with backend nccl == RCCL)."""
import os
import socket

import numpy as np
import pytest

from kitti_motion_compensation_amd import sharding


def test_frame_range_partitions_contiguously():
    for n in (0, 1, 7, 8, 108, 8000, 8001):
        for world in (1, 2, 3, 4, 8):
            ranges = [sharding.frame_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.frame_range(8000, 3, 8) == (3000, 4000)  # configs[3]: frames [r*1000, (r+1)*1000) on rank r


def test_balanced_ranges_on_mixed_frame_sizes():
    rng = np.random.default_rng(0)
    sizes = rng.integers(90_000, 130_000, size=660).tolist()
    for world in (1, 2, 4, 8):
        ranges = [sharding.frame_range_balanced(sizes, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == len(sizes)
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        pts = [sum(sizes[b:e]) for b, e in ranges]
        assert max(pts) - min(pts) <= 2 * max(sizes)
    assert sharding.frame_range_balanced([0, 0, 0], 1, 2) == (2, 3)
    assert sharding.frame_range_balanced([5, 0, 0, 5], 0, 2)[0] == 0


def test_multi_drive_ranges_and_batches():
    counts = [108, 154, 340, 312, 660]  # configs[4]
    seen = {d: [] for d in range(len(counts))}
    for r in range(8):
        for d, b, e in sharding.multi_drive_ranges(counts, r, 8):
            seen[d].append((b, e))
    for d, n in enumerate(counts):
        spans = sorted(seen[d])
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
    sizes = [100, 200, 5000, 50, 50, 50, 0, 7000]
    batches = list(sharding.make_batches(sizes, 1, 8, max_points=5000, max_frames=3))
    assert batches[0][0] == 1 and batches[-1][1] == 8
    assert all(batches[i][1] == batches[i + 1][0] for i in range(len(batches) - 1))
    for b, e in batches:
        assert e - b <= 3
        assert sum(sizes[b:e]) <= 5000 or e - b == 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        # a sharded "drive": every rank walks its own contiguous frame range; the work itself is GPU-only, so the CPU
        # test stands in a deterministic point count and a per-rank time for it
        sizes = [100_000 + 37 * i for i in range(108)]
        b, e = sharding.frame_range(len(sizes), rank, world)
        pts = float(sum(sizes[b:e]))
        secs = 1.0 + 0.5 * rank
        dist.barrier()
        total, tmax, kmax = sharding.reduce_throughput(dist, torch.device("cpu"), pts, secs, secs / 2)
        sums, maxes = sharding.reduce_counters(dist, torch.device("cpu"), [pts, 1.0], [secs, float(rank), -1.0 - rank])
        assert sums == [float(sum(sizes)), 2.0] and maxes == [1.5, 1.0, -1.0]  # one all_gather, SUM block and MAX block
        q.put((rank, b, e, total, tmax, kmax))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world_size_2_gloo_reduction():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    sizes = [100_000 + 37 * i for i in range(108)]
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 54, 54, 108)
    for _, _, _, total, tmax, kmax in res:  # every rank sees the same reduced values
        assert total == float(sum(sizes))
        assert tmax == 1.5 and kmax == 0.75


def test_single_process_reduction_is_identity():
    assert sharding.reduce_throughput(None, None, 10, 2.0, 1.0) == (10.0, 2.0, 1.0)


def test_c_abi_split_is_the_python_split():
    """kmc_frame_ranges_balanced (what MotionCompensateRun's multi-device driver cuts a run with) against
    sharding.balanced_bounds (what the per-rank launch uses): one definition, two implementations."""
    from kitti_motion_compensation_amd import capi

    rng = np.random.default_rng(5)
    cases = [[], [0, 0, 0], [5, 0, 0, 5], [1], [7] * 9, [0, 9, 0], list(rng.integers(0, 3, 40)),
             list(rng.integers(90_000, 130_000, size=660)), list(rng.integers(1, 2**40, size=97)), [2**63, 2**63, 2**63 - 1, 1]]
    for sizes in cases:
        for parts in (1, 2, 3, 4, 8, 13):
            got = capi.frame_ranges_balanced(sizes, parts).tolist()
            want = sharding.balanced_bounds([int(s) for s in sizes], parts)
            assert got == want, (sizes[:8], parts)
            assert got[0] == 0 and got[-1] == len(sizes) and all(a <= b for a, b in zip(got, got[1:]))
    # near-equal point counts per part on KITTI-like sizes
    sizes = [int(s) for s in rng.integers(90_000, 130_000, size=660)]
    b = capi.frame_ranges_balanced(sizes, 8).tolist()
    pts = [sum(sizes[b[r]:b[r + 1]]) for r in range(8)]
    assert max(pts) - min(pts) <= 2 * max(sizes)
