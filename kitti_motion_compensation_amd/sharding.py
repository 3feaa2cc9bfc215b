"""Frame-range sharding for the one-process-per-GPU launch.

The deskew path shards trivially: points within a frame are independent and frames are independent given their own
(T_start, T_end).  A drive is split into CONTIGUOUS frame ranges, one per rank; no point data ever crosses GPUs.  The
collectives of a job: ONE all_gather of the throughput counters (RCCL on the GPU box, gloo in the CPU tests) -- plus, in
bench.py, the four barriers the timing contract asks for around its two timed regions (on the nccl backend a barrier is a
one-element all-reduce).  Plumbing only -- no compute here.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def frame_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of frames owned by `rank`; sizes differ by at most one frame, earlier ranks get the extra."""
    if world <= 0 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError((n_frames, rank, world))
    base, extra = divmod(n_frames, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def frame_range_balanced(frame_sizes: Sequence[int], rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range chosen on POINT counts (mixed frame sizes, BASELINE.json configs[4]): rank r owns the frames whose
    point-prefix midpoint falls into [r, r+1) * total / world.  Ranges are contiguous, disjoint and cover every frame.
    Integer arithmetic, the same definition as the C-ABI's kmc_frame_ranges_balanced (which MotionCompensateRun's multi-device
    driver uses); tests/test_sharding_gloo.py holds the two against each other."""
    b = balanced_bounds(frame_sizes, world)
    if not (0 <= rank < world):
        raise ValueError((rank, world))
    return b[rank], b[rank + 1]


def balanced_bounds(frame_sizes: Sequence[int], world: int) -> List[int]:
    """bounds[0..world]: part r = frames [bounds[r], bounds[r+1])."""
    if world <= 0:
        raise ValueError(world)
    n = len(frame_sizes)
    total = sum(int(s) for s in frame_sizes)
    if total == 0:
        return [frame_range(n, r, world)[0] for r in range(world)] + [n]
    bounds = [0] * (world + 1)
    bounds[world] = n
    acc, r = 0, 1
    for i, s in enumerate(frame_sizes):
        lhs = (2 * acc + int(s)) * world  # midpoint acc + s/2 >= r total / world, without rounding
        while r < world and lhs >= 2 * r * total:
            bounds[r] = i
            r += 1
        acc += int(s)
    while r < world:
        bounds[r] = n
        r += 1
    return bounds


def multi_drive_ranges(drive_frame_counts: Sequence[int], rank: int, world: int) -> List[Tuple[int, int, int]]:
    """configs[4] (several drives at once): every drive is split into contiguous ranges per rank.
    Returns [(drive_index, begin, end), ...] for this rank, empty ranges dropped."""
    out = []
    for d, n in enumerate(drive_frame_counts):
        b, e = frame_range(n, rank, world)
        if e > b:
            out.append((d, b, e))
    return out


def make_batches(frame_sizes: Sequence[int], begin: int, end: int, max_points: int, max_frames: int = 1 << 16):
    """Greedy packing of the frames [begin, end) into batches of at most max_points points / max_frames frames (a frame
    larger than max_points travels alone).  Yields (first_frame, last_frame_exclusive)."""
    i = begin
    while i < end:
        j, pts = i, 0
        while j < end and j - i < max_frames and (j == i or pts + frame_sizes[j] <= max_points):
            pts += frame_sizes[j]
            j += 1
        yield i, j
        i = j


def gather_counters(dist, device, values: Sequence[float]):
    """ONE all_gather of a small vector per rank (RCCL on the GPU box, gloo on CPU; ~100 bytes per rank: latency-bound, link
    bandwidth is irrelevant).  -> rows[r] = rank r's vector, identical on every rank.  `dist` is torch.distributed or None for a
    single process."""
    values = [float(v) for v in values]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [values]
    import torch

    world = dist.get_world_size()
    mine = torch.tensor(values, dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return torch.stack(parts).cpu().tolist()


def reduce_counters(dist, device, sums: Sequence[float], maxes: Sequence[float], with_rows: bool = False):
    """The job's ONLY data collective: every rank contributes one small vector, ONE all_gather (gather_counters); SUM over the first
    block, MAX over the second, computed identically on every rank.  -> (sums, maxes) as float lists; with_rows=True also returns
    the per-rank rows (sums block followed by maxes block), from which a caller can report per-rank minima / maxima -- a
    straggler is invisible in SUM / MAX alone."""
    sums = [float(v) for v in sums]
    maxes = [float(v) for v in maxes]
    rows = gather_counters(dist, device, sums + maxes)
    k = len(sums)
    tot = [sum(r[i] for r in rows) for i in range(k)]
    mx = [max(r[k + i] for r in rows) for i in range(len(maxes))]
    return (tot, mx, rows) if with_rows else (tot, mx)


def reduce_throughput(dist, device, points: float, seconds: float, kernel_seconds: float = 0.0):
    """SUM of points, MAX of times through reduce_counters.  Returns (total_points, max_seconds, max_kernel_seconds)."""
    s, m = reduce_counters(dist, device, [points], [seconds, kernel_seconds])
    return s[0], m[0], m[1]
