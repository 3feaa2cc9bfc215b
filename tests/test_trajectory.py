"""N-knot trajectories: the 3-argument MotionCompensateFrame(Frame, Trajectory, Time) that BASELINE.json's north_star names.
The reference has only the 2-pose geodesic; the N-knot form chains the reference's own interpolator over consecutive
knots (oracle/kmc_oracle.h).  Checked here: (CPU) the oracle's chain reduces to the reference algorithm for 2 knots and
is continuous across knots; (GPU) the HIP kernels match it, reduce BIT-FOR-BIT to the 2-argument kernels for 2 knots, and
emit per-point bracket indices that are BIT-EXACT against the oracle's trig-free definition."""
import os

import numpy as np
import pytest

from kitti_motion_compensation_amd import capi
from oracle import oracle as orc
from tests import util

T0, T1, TREQ = 47072.283701593, 47072.386973931, 47072.335337762
REL_TOL = 1e-5


@pytest.fixture(scope="module")
def kitti(golden_dir):
    run = os.path.join(golden_dir, "kitti_2011_09_26_drive_0005")
    return util.load_velodyne_bin(run, 0), orc.oxts_to_pose(orc.oxts(**util.load_oxts_fields(run, 0)))


def _chain(P0, steps):
    poses = [P0]
    for s in steps:
        poses.append(orc.affine_mul(poses[-1], orc.se3_exp(s)))
    return poses


def _rt(poses):
    return np.stack([p.rt12().reshape(3, 4) for p in poses])


def _oxts_like_trajectory(P0):
    """Three bracketing poses at T0-0.05 / mid / T1+0.05: the boundary between the two segments sits inside the scan."""
    times = [T0 - 0.0520, 0.5 * (T0 + T1) + 0.0031, T1 + 0.0490]
    poses = _chain(P0, [[1.31, 0.04, -0.015, 0.002, -0.004, 0.035], [1.36, -0.02, 0.01, -0.003, 0.002, 0.05]])
    return times, poses


# ---- CPU: the oracle's chain --------------------------------------------------------------------------------------
def test_oracle_two_knots_is_the_reference_algorithm(kitti):
    xyzi, P1 = kitti
    xyzi = xyzi[::23]
    P2 = orc.affine_mul(P1, orc.se3_exp([2.9, -0.3, 0.1, 0.02, 0.01, -0.1]))
    a = orc.deskew_xyzi_f32(xyzi, T0, P1, T1, P2, TREQ, mode=orc.FAITHFUL, threads=1)
    b = orc.deskew_xyzi_f32_traj(xyzi, T0, T1, [T0, T1], [P1, P2], TREQ, threads=1)
    assert a["rc"] == orc.OK and b["rc"] == orc.OK
    assert np.array_equal(a["xyz_f64"], b["xyz_f64"])  # same operations, bit for bit
    assert np.all(b["bracket_by_time"] == 0)


def test_oracle_chain_is_continuous_and_in_range(kitti):
    _, P1 = kitti
    times, poses = _oxts_like_trajectory(P1)
    eps = 1e-9
    rc1, A = orc.traj_pose_at_time(times, poses, times[1] - eps)
    rc2, B = orc.traj_pose_at_time(times, poses, times[1])
    assert rc1 == orc.OK and rc2 == orc.OK
    assert np.allclose(A.matrix(), B.matrix(), atol=1e-6)
    assert np.allclose(B.matrix(), poses[1].matrix(), atol=1e-8)  # a knot reproduces its pose
    assert orc.traj_pose_at_time(times, poses, times[0] - 1e-3)[0] == orc.ERR_TIME_OUT_OF_RANGE
    assert orc.traj_pose_at_time(times, poses, times[2] + 1e-3)[0] == orc.ERR_TIME_OUT_OF_RANGE
    assert orc.traj_pose_at_time(times, poses, times[2])[0] == orc.OK


def test_trig_free_bracket_matches_time_bracket_away_from_knots(kitti):
    xyzi, P1 = kitti
    times, poses = _oxts_like_trajectory(P1)
    r = orc.deskew_xyzi_f32_traj(xyzi, T0, T1, times, poses, TREQ)
    idx = orc.bracket_indices_f32(xyzi, times, T0, T1)
    frac = (np.pi - np.arctan2(xyzi[:, 1].astype(np.float64), xyzi[:, 0].astype(np.float64))) / (2 * np.pi)
    c1 = (times[1] - T0) / (T1 - T0)
    away = np.abs(frac - c1) > 1e-6
    assert np.array_equal(idx[away], r["bracket_by_time"][away])
    assert set(np.unique(idx)) == {0, 1}


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    import torch

    assert torch.cuda.is_available()
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_two_knots_reduce_bit_for_bit_to_the_two_argument_kernels(ctx, kitti):
    xyzi, P1 = kitti
    for step in ([1.3, 0.05, -0.02, 0, 0, 0], [2.9, -0.3, 0.1, 0.02, 0.01, -0.1], [0.4, 0.1, 0.0, 0.1, -0.3, 0.6], [0.4, 0.1, 0.0, 0.3, -0.9, 2.2]):
        P2 = orc.affine_mul(P1, orc.se3_exp(step))
        params = capi.frame_params_from_poses(_rt([P1])[0], _rt([P2])[0], T0, T1, TREQ)
        a = np.empty_like(xyzi)
        b = np.empty_like(xyzi)
        br = np.full(xyzi.shape[0], 99, dtype=np.uint32)
        ctx.deskew_f32(xyzi, a, params)
        ctx.deskew_traj_f32(xyzi, b, [T0, T1], _rt([P1, P2]), T0, T1, TREQ, br)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), step
        assert np.all(br == 0)
        # f64 Eigen layout
        n = 5000
        cloud = np.concatenate([xyzi[:n, :3].astype(np.float64), np.ones((n, 1))], axis=1)
        stamps = orc.pseudo_timestamps(cloud, T0, T1)
        cols = [np.ascontiguousarray(cloud[:, j]) for j in range(4)]
        o1 = [np.empty(n) for _ in range(4)]
        o2 = [np.empty(n) for _ in range(4)]
        ctx.deskew_f64cols(*cols, stamps, T0, T1, params, *o1)
        ctx.deskew_traj_f64cols(*cols, stamps, [T0, T1], _rt([P1, P2]), TREQ, *o2)
        for u, v in zip(o1, o2):
            assert np.array_equal(u.view(np.uint64), v.view(np.uint64)), step


@pytest.mark.gpu
def test_three_bracketing_poses_vs_oracle_and_bit_exact_indices(ctx, kitti):
    xyzi, P1 = kitti
    times, poses = _oxts_like_trajectory(P1)
    out = np.empty_like(xyzi)
    br = np.empty(xyzi.shape[0], dtype=np.uint32)
    st = ctx.deskew_traj_f32(xyzi, out, times, _rt(poses), T0, T1, TREQ, br)
    assert st.n_points == xyzi.shape[0]
    ref = orc.deskew_xyzi_f32_traj(xyzi, T0, T1, times, poses, TREQ)
    assert ref["rc"] == orc.OK
    assert np.array_equal(out[:, 3].view(np.uint32), xyzi[:, 3].view(np.uint32))
    assert util.rel_point_error(out[:, :3], ref["xyz_f64"]).max() <= REL_TOL
    assert np.array_equal(br, orc.bracket_indices_f32(xyzi, times, T0, T1)), "bracket indices must be bit-exact"
    # without the index output, and device-resident: same bits
    import torch

    n = xyzi.shape[0]
    d_in = torch.from_numpy(xyzi).cuda()
    d_out = torch.full((n + 64, 4), 7.0, dtype=torch.float32, device="cuda")  # guard rows: 123397 is not a multiple of 64
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.deskew_traj_f32(d_in, d_out, times, _rt(poses), T0, T1, TREQ, None, n=n)
    torch.cuda.synchronize()
    assert bool((d_out[n:] == 7.0).all()), "wrote past the end"
    assert np.array_equal(d_out[:n].cpu().numpy().view(np.uint32), out.view(np.uint32))
    ctx.set_stream(None)


@pytest.mark.gpu
def test_bracket_decision_around_the_knots_is_the_exact_one(ctx, kitti):
    """Round 3 decides a point's bracket from the scan fraction the deskew computes anyway and falls back to the exact trig-free
    half-plane test only for waves that hold a lane within 1e-4 turn of a knot (or a coordinate that is not a normal number).
    The claim "outside the margin the two decisions are the same" is put under load here: hundreds of thousands of points placed
    at angular distances from 1e-9 to 3e-3 turn on BOTH sides of every knot -- inside the margin, on its edge, outside it --, at
    radii from 1e-30 to 1e30 m, through the single-frame kernel (2, 3, 4 and 7 knots: kernel-argument records, the preloaded knot
    path and the knot loop): every bracket index equals the oracle's (which only knows the exact test), bit for bit."""
    xyzi, P1 = kitti
    rng = np.random.default_rng(31)
    steps = [[0.7, 0.02, -0.01, 0.001, -0.002, 0.018], [0.66, 0.02, 0.0, -0.001, 0.001, 0.02], [0.71, -0.01, 0.01, 0.0, 0.002, 0.022],
             [0.69, 0.0, 0.0, 0.001, 0.0, 0.019], [0.7, 0.01, 0.0, 0.0, 0.001, 0.02], [0.68, 0.0, 0.01, 0.001, 0.0, 0.021]]
    for n_knots in (2, 3, 4, 7):
        # knots: the first / last just outside the scan, the interior ones at "ugly" fractions plus the exact quarter turns
        interior = {3: [0.5003], 4: [0.25, 0.75], 7: [0.1234, 0.25, 0.5, 0.6180339, 0.9]}.get(n_knots, [])
        fr = np.array([-0.04] + interior + [1.04])
        times = list(T0 + fr * (T1 - T0))
        poses = _chain(P1, steps[:n_knots - 1])
        pts = []
        for c in interior:
            d = np.concatenate([10.0 ** rng.uniform(-9, -2.5, size=40_000), [0.0, 1e-4, 0.99e-4, 1.01e-4, 5e-5]])
            d = d * rng.choice([-1.0, 1.0], size=d.size)
            frac = np.clip(c + d, 1e-9, 1 - 1e-9)
            az = np.pi - 2 * np.pi * frac                        # timestamp_mocking.cpp:46 inverted
            r = 10.0 ** rng.choice([0.0, 1.0, 1.7, -3.0, 6.0, -30.0, 30.0], size=d.size, p=[0.3, 0.3, 0.3, 0.04, 0.04, 0.01, 0.01])
            pts.append(np.stack([r * np.cos(az), r * np.sin(az), rng.normal(0, 1, d.size), rng.uniform(0, 1, d.size)], axis=1))
        cloud = np.ascontiguousarray(np.concatenate(pts + [xyzi[:5000]]).astype(np.float32))
        cloud = cloud[rng.permutation(cloud.shape[0])]           # knot neighbours scattered over the waves, not sorted by angle
        out = np.empty_like(cloud)
        br = np.empty(cloud.shape[0], dtype=np.uint32)
        ctx.deskew_traj_f32(cloud, out, times, _rt(poses), T0, T1, TREQ, br)
        want = orc.bracket_indices_f32(cloud, times, T0, T1)
        bad = np.flatnonzero(br != want)
        assert bad.size == 0, (n_knots, bad.size, cloud[bad[:3]], br[bad[:3]], want[bad[:3]])
        assert len(set(want.tolist())) == n_knots - 1            # every bracket really occurs
        import torch

        d_in = torch.from_numpy(cloud).cuda()                    # device-resident: records in the kernel arguments up to four knots
        d_out = torch.empty_like(d_in)
        d_br = torch.zeros(cloud.shape[0], dtype=torch.int32, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.deskew_traj_f32(d_in, d_out, times, _rt(poses), T0, T1, TREQ, d_br)
        torch.cuda.synchronize()
        ctx.set_stream(None)
        assert np.array_equal(d_br.cpu().numpy().astype(np.uint32), want), n_knots
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), out.view(np.uint32)), n_knots


@pytest.mark.gpu
def test_short_trajectories_ride_in_the_kernel_arguments(kitti):
    """Up to four knots (three segments) on device-resident points: the segment records are kernel arguments -- no table slot, no
    upload, no host wait -- and the call may go over the frame queues like kmc_hip_deskew_f32.  Longer trajectories, and host
    buffers, take the table path.  Same records, same kernel body: the two paths give the same bits (points and bracket
    indices), in order and over four queues, at 16-byte-aligned sub-range offsets too."""
    import torch

    xyzi, P1 = kitti
    n = xyzi.shape[0]
    steps = [[0.7, 0.02, -0.01, 0.001, -0.002, 0.018], [0.66, 0.02, 0.0, -0.001, 0.001, 0.02], [0.71, -0.01, 0.01, 0.0, 0.002, 0.022],
             [0.69, 0.0, 0.0, 0.001, 0.0, 0.019]]
    c = capi.Context(0)
    try:
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        d_in = torch.from_numpy(xyzi).cuda()
        for n_knots in (2, 3, 4, 5):  # 5 knots = 4 segments: beyond the inline limit, table path on the device as well
            lo, hi = T0 - 0.004, T1 + 0.004
            times = list(np.linspace(lo, hi, n_knots))
            poses = _chain(P1, steps[:n_knots - 1])
            want = np.empty_like(xyzi)
            want_k = np.empty(n, dtype=np.uint32)
            c.deskew_traj_f32(xyzi, want, times, _rt(poses), T0, T1, TREQ, want_k)  # host buffers: table path
            assert np.array_equal(want_k, orc.bracket_indices_f32(xyzi, times, T0, T1))
            assert len(np.unique(want_k)) == n_knots - 1
            for queues in (1, 4):
                c.set_frame_queues(queues)
                outs = [torch.zeros((n + 8, 4), dtype=torch.float32, device="cuda") for _ in range(6)]
                idx = [torch.full((n,), 99, dtype=torch.int32, device="cuda") for _ in range(6)]
                for k, (o, ix) in enumerate(zip(outs, idx)):
                    c.deskew_traj_f32(d_in, o[k:k + n], times, _rt(poses), T0, T1, TREQ, ix if k % 2 == 0 else None, n=n)
                c.frame_queue_join()
                torch.cuda.synchronize()
                for k, (o, ix) in enumerate(zip(outs, idx)):
                    assert np.array_equal(o[k:k + n].cpu().numpy().view(np.uint32), want.view(np.uint32)), (n_knots, queues, k)
                    assert bool((o[:k] == 0).all()) and bool((o[k + n:] == 0).all()), "wrote outside the frame"
                    if k % 2 == 0:
                        assert np.array_equal(ix.cpu().numpy().view(np.uint32), want_k), (n_knots, queues, k)
            c.set_frame_queues(1)
    finally:
        c.close()


@pytest.mark.gpu
def test_many_knots_quarter_turn_boundaries_and_edge_points(ctx, kitti):
    """Knots exactly on the quarter turns (exact directions), knots outside the scan, and points on the axes / signed zeros."""
    xyzi, P1 = kitti
    D = T1 - T0
    times = [T0 - 0.3 * D, T0 - 0.1 * D, T0 + 0.25 * D, T0 + 0.4 * D, T0 + 0.5 * D, T0 + 0.75 * D, T0 + 0.9 * D, T1, T1 + 0.2 * D]
    steps = [[0.3 + 0.02 * i, 0.01 * (-1) ** i, 0.0, 0.001 * i, -0.002, 0.01 + 0.004 * i] for i in range(len(times) - 1)]
    poses = _chain(P1, steps)
    edge = np.array([
        [0, 0, 1, 1], [-0.0, 0.0, 1, 2], [-0.0, -0.0, 1, 3], [0.0, -0.0, 1, 4], [-1, 0, 0, 5], [-1, -0.0, 0, 6], [1, 0, 0, 7],
        [1, -0.0, 0, 8], [0, 5, 0, 9], [0, -5, 0, 10], [-0.0, 5, 0, 11], [-0.0, -5, 0, 12], [3, 3, 1, 13], [-3, 3, 1, 14],
        [-3, -3, 1, 15], [3, -3, 1, 16], [1e-30, 1e-30, 0, 17], [-5, 1e-7, 0, 18], [-5, -1e-7, 0, 19], [5, 1e-7, 0, 20], [5, -1e-7, 0, 21],
    ], dtype=np.float32)
    pts = np.concatenate([edge, xyzi[::3]])
    out = np.empty_like(pts)
    br = np.empty(pts.shape[0], dtype=np.uint32)
    treq = T0 + 0.62 * D
    ctx.deskew_traj_f32(pts, out, times, _rt(poses), T0, T1, treq, br)
    want_idx = orc.bracket_indices_f32(pts, times, T0, T1)
    assert np.array_equal(br, want_idx)
    assert br.min() >= 1 and br.max() <= 7  # knots 0,1 lie before the scan (always >=), knot 8 after it
    ref = orc.deskew_xyzi_f32_traj(pts, T0, T1, times, poses, treq)
    assert ref["rc"] == orc.OK
    err = util.rel_point_error(out[:, :3], ref["xyz_f64"])
    assert err.max() <= REL_TOL, err.max()
    # the bracket the oracle derives from f64 stamps agrees wherever the point is not within rounding of a knot
    frac = (np.pi - np.arctan2(pts[:, 1].astype(np.float64), pts[:, 0].astype(np.float64))) / (2 * np.pi)
    cs = (np.array(times) - T0) / D
    away = np.min(np.abs(frac[:, None] - cs[None, :]), axis=1) > 1e-6
    assert np.array_equal(br[away], ref["bracket_by_time"][away])


@pytest.mark.gpu
def test_f64_trajectory_vs_oracle(ctx, kitti):
    xyzi, _ = kitti
    xyzi = xyzi[::9]
    n = xyzi.shape[0]
    P_local = orc.Affine.from_Rt(orc.so3_exp([0.01, -0.02, 0.7]), [12.5, -3.0, 0.4])
    times, poses = _oxts_like_trajectory(P_local)
    cloud = np.concatenate([xyzi[:, :3].astype(np.float64), np.ones((n, 1))], axis=1)
    stamps = orc.pseudo_timestamps(cloud, T0, T1)
    cols = [np.ascontiguousarray(cloud[:, j]) for j in range(4)]
    outs = [np.empty(n) for _ in range(4)]
    br = np.empty(n, dtype=np.uint32)
    rc, st = ctx.deskew_traj_f64cols(*cols, stamps, times, _rt(poses), TREQ, *outs, bracket_idx_out=br)
    assert rc == capi.OK and st.n_out_of_range == 0
    rc_o, nbad, want, br_o = orc.motion_compensate_frame_traj(cloud, stamps, times, poses, TREQ)
    assert rc_o == orc.OK
    assert np.array_equal(br, br_o)  # f64 compares of the same stamps: exact
    got = np.stack(outs, axis=1)
    assert util.rel_point_error(got[:, :3], want[:, :3]).max() <= 1e-11
    # a stamp beyond the last knot is reported like the reference's assert
    stamps2 = stamps.copy()
    stamps2[7] = times[-1] + 1e-3
    rc, st = ctx.deskew_traj_f64cols(*cols, stamps2, times, _rt(poses), TREQ, *outs, raise_on_range=False)
    assert rc == capi.ERR_TIME_OUT_OF_RANGE and st.n_out_of_range == 1


@pytest.mark.gpu
def test_f64_trajectory_long_table_and_the_ones_column(ctx, kitti):
    """(a) More than four knots: the records go through the device table instead of the kernel arguments -- same oracle bar.
    (b) w = None says "the homogeneous column is all ones": same XYZ bits as an explicit column of ones; an output column that is
    wanted all the same comes back as ones on every route -- staged (pageable numpy), in place (the page-locked pool: the HOST fills
    it while the kernel runs, the column never crosses the link) and device-resident (the kernel writes it)."""
    import torch

    xyzi, _ = kitti
    xyzi = xyzi[::5]
    n = xyzi.shape[0]
    P_local = orc.Affine.from_Rt(orc.so3_exp([0.01, -0.02, 0.7]), [12.5, -3.0, 0.4])
    cloud = np.concatenate([xyzi[:, :3].astype(np.float64), np.ones((n, 1))], axis=1)
    stamps = orc.pseudo_timestamps(cloud, T0, T1)
    cols = [np.ascontiguousarray(cloud[:, j]) for j in range(4)]
    for n_knots in (3, 7):
        times = list(np.linspace(T0 - 0.05, T1 + 0.05, n_knots))
        poses = _chain(P_local, [[0.4 + 0.01 * k, 0.01, -0.005, 0.001, -0.002, 0.012 + 0.001 * k] for k in range(n_knots - 1)])
        rc_o, nbad, want, br_o = orc.motion_compensate_frame_traj(cloud, stamps, times, poses, TREQ)
        assert rc_o == orc.OK
        full = [np.empty(n) for _ in range(4)]
        br = np.empty(n, dtype=np.uint32)
        rc, st = ctx.deskew_traj_f64cols(*cols, stamps, times, _rt(poses), TREQ, *full, bracket_idx_out=br)
        assert rc == capi.OK and st.n_out_of_range == 0 and np.array_equal(br, br_o)
        assert util.rel_point_error(np.stack(full, axis=1)[:, :3], want[:, :3]).max() <= 1e-11
        assert np.array_equal(full[3], cols[3])
        # staged, no w
        outs = [np.full(n, -7.0) for _ in range(4)]
        ctx.deskew_traj_f64cols(cols[0], cols[1], cols[2], None, stamps, times, _rt(poses), TREQ, *outs)
        for j in range(3):
            assert np.array_equal(outs[j].view(np.uint64), full[j].view(np.uint64)), (n_knots, "staged", j)
        assert np.all(outs[3] == 1.0)
        outs3 = [np.full(n, -7.0) for _ in range(3)]  # ... and no output column either
        ctx.deskew_traj_f64cols(cols[0], cols[1], cols[2], None, stamps, times, _rt(poses), TREQ, *outs3)
        assert np.array_equal(outs3[0].view(np.uint64), full[0].view(np.uint64))
        # in place on the page-locked pool
        pin, pst, pout = capi.PooledArray((3, n)), capi.PooledArray((n,)), capi.PooledArray((4, n))
        try:
            pin.a[:] = np.stack(cols[:3])
            pst.a[:] = stamps
            pout.a[:] = -7.0
            rc, st = ctx.deskew_traj_f64cols(pin.a[0], pin.a[1], pin.a[2], None, pst.a, times, _rt(poses), TREQ, pout.a[0], pout.a[1], pout.a[2], pout.a[3])
            assert rc == capi.OK and st.n_launches == 1
            for j in range(3):
                assert np.array_equal(pout.a[j].view(np.uint64), full[j].view(np.uint64)), (n_knots, "in place", j)
            assert np.all(pout.a[3] == 1.0)
        finally:
            pin.close(), pst.close(), pout.close()
        # device-resident
        d = [torch.from_numpy(c).cuda() for c in cols[:3]] + [torch.from_numpy(stamps).cuda()]
        do = [torch.full((n,), -7.0, dtype=torch.float64, device="cuda") for _ in range(4)]
        ctx.deskew_traj_f64cols(d[0], d[1], d[2], None, d[3], times, _rt(poses), TREQ, *do)
        ctx.synchronize()
        for j in range(3):
            assert np.array_equal(do[j].cpu().numpy().view(np.uint64), full[j].view(np.uint64)), (n_knots, "device", j)
        assert bool((do[3] == 1.0).all())


@pytest.mark.gpu
def test_trajectory_argument_checks(ctx, kitti):
    xyzi, P1 = kitti
    pts = np.ascontiguousarray(xyzi[:64])
    out = np.empty_like(pts)
    times, poses = _oxts_like_trajectory(P1)
    with pytest.raises(capi.KmcError) as e:  # trajectory does not cover the scan
        ctx.deskew_traj_f32(pts, out, [T0 + 0.01, T1], _rt(poses[:2]), T0, T1, TREQ)
    assert e.value.status == capi.ERR_TIME_OUT_OF_RANGE
    with pytest.raises(capi.KmcError) as e:  # knots not increasing
        ctx.deskew_traj_f32(pts, out, [times[1], times[0], times[2]], _rt(poses), T0, T1, TREQ)
    assert e.value.status == capi.ERR_DEGENERATE
    with pytest.raises(capi.KmcError) as e:  # too many knots
        ctx.deskew_traj_f32(pts, out, np.linspace(T0 - 1, T1 + 1, 19), _rt(poses * 7)[:19], T0, T1, TREQ)
    assert e.value.status == capi.ERR_INVALID_ARG


# ---- batched N-knot form: every frame with its own trajectory, one launch ----------------------------------------------
def _random_frames(rng, P1, sizes):
    frames = []
    for _ in sizes:
        k = int(rng.integers(2, 7))  # 2..6 knots
        t0 = T0 + float(rng.uniform(-1e-3, 1e-3))
        t1 = t0 + float(rng.uniform(0.09, 0.11))
        inner = np.sort(rng.uniform(t0 + 0.005, t1 - 0.005, k - 2)) if k > 2 else np.array([])
        if k > 2 and rng.random() < 0.3:
            inner[0] = t0 + 0.25 * (t1 - t0)  # a knot exactly on a quarter turn
            inner = np.sort(inner)
        times = np.concatenate([[t0 - float(rng.uniform(0, 0.05))], inner, [t1 + float(rng.uniform(0, 0.05))]])
        steps = [list(rng.normal(0, [0.5, 0.05, 0.02, 0.002, 0.003, 0.02])) for _ in range(k - 1)]
        start = orc.affine_mul(P1, orc.se3_exp(list(rng.normal(0, [3, 3, 0.1, 0.01, 0.01, 0.3]))))
        poses = _chain(start, steps)
        frames.append(dict(times=times, poses=_rt(poses), oracle_poses=poses, stamp_start=t0, stamp_end=t1,
                           requested_time=t0 + float(rng.random()) * (t1 - t0)))
    return frames


def _check_traj_batch(ctx, xyzi, sizes, frames, mem="host"):
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(offsets[-1])
    pts = np.ascontiguousarray(xyzi[:n])
    out = np.full_like(pts, 7.0)
    fidx = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    bidx = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    if mem == "host":
        ctx.deskew_traj_batch_f32(pts, out, offsets, frames, fidx, bidx)
    else:
        import torch

        d_in = torch.from_numpy(pts).cuda()
        d_out = torch.full((n + 64, 4), 7.0, dtype=torch.float32, device="cuda")
        d_f = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_b = torch.zeros(n, dtype=torch.int32, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.deskew_traj_batch_f32(d_in, d_out, offsets, frames, d_f, d_b)
        torch.cuda.synchronize()
        ctx.set_stream(None)
        assert bool((d_out[n:] == 7.0).all()), "wrote past the end"
        out = d_out[:n].cpu().numpy()
        fidx = d_f.cpu().numpy().view(np.uint32)
        bidx = d_b.cpu().numpy().view(np.uint32)
    assert np.array_equal(fidx, np.repeat(np.arange(len(sizes), dtype=np.uint32), sizes)), "frame indices must be bit-exact"
    assert np.array_equal(out[:, 3].view(np.uint32), pts[:, 3].view(np.uint32))
    for f, fr in enumerate(frames):
        a, b = int(offsets[f]), int(offsets[f + 1])
        if a == b:
            continue
        # the single-frame entry point on the same frame: same arithmetic, so the same bits
        one = np.empty_like(pts[a:b])
        br1 = np.empty(b - a, dtype=np.uint32)
        ctx.deskew_traj_f32(np.ascontiguousarray(pts[a:b]), one, fr["times"], fr["poses"], fr["stamp_start"], fr["stamp_end"],
                            fr["requested_time"], br1)
        assert np.array_equal(out[a:b].view(np.uint32), one.view(np.uint32)), f"frame {f}"
        assert np.array_equal(bidx[a:b], br1), f"frame {f}"
        assert np.array_equal(bidx[a:b], orc.bracket_indices_f32(pts[a:b], fr["times"], fr["stamp_start"], fr["stamp_end"]))
        if f % 7 == 0 or b - a < 2000:
            ref = orc.deskew_xyzi_f32_traj(pts[a:b], fr["stamp_start"], fr["stamp_end"], fr["times"], fr["oracle_poses"], fr["requested_time"])
            assert ref["rc"] == orc.OK
            assert util.rel_point_error(out[a:b, :3], ref["xyz_f64"]).max() <= REL_TOL
    return out, fidx, bidx


@pytest.mark.gpu
def test_batched_trajectories_vs_single_frame_kernel_and_oracle(ctx, kitti):
    xyzi, P1 = kitti
    rng = np.random.default_rng(5)
    big = np.tile(xyzi, (4, 1))
    sizes = [20000, 1, 63, 0, 64, 65, 30000, 0, 0, 5, 16384, 16385, 100, 40000, 127, 129]  # ragged, empty, tiny, chunk-aligned
    frames = _random_frames(rng, P1, sizes)
    host = _check_traj_batch(ctx, big, sizes, frames, "host")
    dev = _check_traj_batch(ctx, big, sizes, frames, "device")
    for h, d in zip(host, dev):
        assert np.array_equal(h.view(np.uint32), d.view(np.uint32))


@pytest.mark.gpu
def test_batched_trajectories_many_tiny_frames(ctx, kitti):
    """More frames than points per tile: a 64-point tile walks through dozens of frames."""
    xyzi, P1 = kitti
    rng = np.random.default_rng(6)
    sizes = [int(v) for v in rng.integers(0, 9, 300)] + [5000] + [int(v) for v in rng.integers(0, 4, 100)]
    frames = _random_frames(rng, P1, sizes)
    _check_traj_batch(ctx, xyzi, sizes, frames, "host")


@pytest.mark.gpu
def test_batched_trajectories_two_knots_equal_the_batched_two_pose_kernel(ctx, kitti):
    xyzi, P1 = kitti
    sizes = [30000, 12345, 50000]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(offsets[-1])
    pts = np.ascontiguousarray(xyzi[:n])
    steps = [[1.3, 0.05, -0.02, 0.001, -0.002, 0.03], [0.9, 0.0, 0.0, 0.0, 0.0, -0.04], [2.9, -0.3, 0.1, 0.02, 0.01, -0.1]]
    frames, params = [], []
    for s in steps:
        P2 = orc.affine_mul(P1, orc.se3_exp(s))
        frames.append(dict(times=[T0, T1], poses=_rt([P1, P2]), stamp_start=T0, stamp_end=T1, requested_time=TREQ))
        params.append(capi.frame_params_from_poses(P1.rt12().reshape(3, 4), P2.rt12().reshape(3, 4), T0, T1, TREQ))
    a = np.empty_like(pts)
    b = np.empty_like(pts)
    ctx.deskew_traj_batch_f32(pts, a, offsets, frames)
    ctx.deskew_batch_f32(pts, b, offsets, params)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
def test_batched_trajectories_argument_checks(ctx, kitti):
    xyzi, P1 = kitti
    pts = np.ascontiguousarray(xyzi[:1000])
    out = np.empty_like(pts)
    good = dict(times=[T0, T1], poses=_rt([P1, P1]), stamp_start=T0, stamp_end=T1, requested_time=TREQ)
    short = dict(good, times=[T0 + 0.01, T1])          # does not cover the scan
    with pytest.raises(capi.KmcError) as e:
        ctx.deskew_traj_batch_f32(pts, out, [0, 500, 1000], [good, short])
    assert e.value.status == capi.ERR_TIME_OUT_OF_RANGE
    late = dict(good, requested_time=T1 + 1.0)
    with pytest.raises(capi.KmcError) as e:
        ctx.deskew_traj_batch_f32(pts, out, [0, 500, 1000], [late, good])
    assert e.value.status == capi.ERR_TIME_OUT_OF_RANGE
    with pytest.raises(capi.KmcError) as e:
        ctx.deskew_traj_batch_f32(pts, out, [0, 600, 500], [good, good])
    assert e.value.status == capi.ERR_INVALID_ARG
    st = ctx.deskew_traj_batch_f32(pts, out, [0], [])   # no frames: nothing to do
    assert st.n_points == 0


@pytest.mark.gpu
def test_nknot_frames_without_the_barrier_bit_same_bits(kitti, monkeypatch):
    """Short trajectories on device-resident points travel in the kernel arguments; on the context's own stream such frames are
    dispatched without the barrier bit when they share no buffer with the frames in flight (kmc_hip.h, kmc_hip_set_frame_queues).
    Same bits as a context created with KMC_ANY_ORDER=0; a frame that reads what the frame before it wrote stays ordered; two-pose
    and N-knot frames share one window."""
    import torch

    xyzi, P1 = kitti
    times, poses = _oxts_like_trajectory(P1)
    n, nf = xyzi.shape[0], 12
    monkeypatch.setenv("KMC_ANY_ORDER", "0")
    plain = capi.Context(0)
    monkeypatch.delenv("KMC_ANY_ORDER")
    fast = capi.Context(0)
    try:
        rng = np.random.default_rng(9)
        ins = [torch.from_numpy(np.ascontiguousarray(xyzi[rng.permutation(n)])).cuda() for _ in range(nf)]
        treq = [T0 + (T1 - T0) * (f + 0.5) / nf for f in range(nf)]
        want = [torch.empty_like(x) for x in ins]
        for f in range(nf):
            plain.deskew_traj_f32(ins[f], want[f], times, _rt(poses), T0, T1, treq[f], None, n=n)
        plain.synchronize()
        assert plain.any_order_launches() == 0
        outs = [torch.zeros_like(x) for x in ins]
        torch.cuda.synchronize()
        for f in range(nf):
            fast.deskew_traj_f32(ins[f], outs[f], times, _rt(poses), T0, T1, treq[f], None, n=n)
        assert fast.any_order_launches() == nf - 1
        fast.synchronize()
        for f in range(nf):
            assert torch.equal(outs[f].view(torch.int32), want[f].view(torch.int32)), f
        # a chain through both kinds of frame: N-knot -> two-pose -> N-knot, each reading the output of the one before
        prm = capi.FrameParams.make([1.3, 0.02, -0.01, 0.001, -0.002, 0.03], 0.5)
        ref = [ins[0], torch.empty_like(ins[0]), torch.empty_like(ins[0]), torch.empty_like(ins[0])]
        plain.deskew_traj_f32(ref[0], ref[1], times, _rt(poses), T0, T1, treq[3], None, n=n)
        plain.deskew_f32(ref[1], ref[2], prm)
        plain.deskew_traj_f32(ref[2], ref[3], times, _rt(poses), T0, T1, treq[7], None, n=n)
        plain.synchronize()
        got = [ins[0], torch.zeros_like(ins[0]), torch.zeros_like(ins[0]), torch.zeros_like(ins[0])]
        torch.cuda.synchronize()
        before = fast.any_order_launches()
        fast.deskew_traj_f32(got[0], got[1], times, _rt(poses), T0, T1, treq[3], None, n=n)
        fast.deskew_f32(got[1], got[2], prm)
        fast.deskew_traj_f32(got[2], got[3], times, _rt(poses), T0, T1, treq[7], None, n=n)
        assert fast.any_order_launches() == before
        fast.deskew_f32(ins[4], outs[4], prm)  # independent of the chain: may overtake it
        assert fast.any_order_launches() == before + 1
        fast.synchronize()
        assert torch.equal(got[3].view(torch.int32), ref[3].view(torch.int32))
    finally:
        plain.close()
        fast.close()
